"""Per-dispatch summary of a rocprofv3 --kernel-trace CSV of tools/train_bench.py: time per kernel and, for gemm_kernel, per launch geometry
(grid = (row tiles, column tiles, batch x k-split) work-groups) -- which layer shapes the training step spends its GEMM time in.
    python tools/train_trace.py <dir with *kernel_trace.csv> [steps] [kernel name to break down by grid ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
per = defaultdict(lambda: [0, 0.0])
gem = defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("lsdm::", "").replace("void ", "").split("(")[0][:60]
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per[name][0] += 1; per[name][1] += dt
    if "gemm_kernel" in name:
        g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        gem[g][0] += 1; gem[g][1] += dt
tot = sum(v[1] for v in per.values())
print("total kernel time %.2f ms per step (%d steps)" % (tot / steps / 1e3, steps))
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-60s %6d calls  %8.3f ms/step" % (k, v[0], v[1] / steps / 1e3))
for extra in sys.argv[3:]:      # any further argument: a kernel name to break down by launch grid (work-groups)
    by = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if extra in r["Kernel_Name"]:
            g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            by[g][0] += 1; by[g][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%s by grid:" % extra)
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  grid %-22s %5d calls  avg %8.1f us  %7.3f ms/step" % (k, v[0], v[1] / v[0], v[1] / steps / 1e3))
print("gemm_kernel by grid (row tiles of 64, column tiles of 64, batch x k-split):")
for k, v in sorted(gem.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  grid %-22s %5d calls  avg %8.1f us  %7.3f ms/step" % (k, v[0], v[1] / v[0], v[1] / steps / 1e3))
