"""Condenses rocprofv3 output directories into small committed summaries (profiles/).

    python tools/summarize_rocprof.py stats  <dir> <out.md>     kernel-trace --stats  -> per-kernel table
    python tools/summarize_rocprof.py pmc    <dir> <out.json>   --pmc counter CSVs   -> per-kernel mean counter per launch
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("lsdm::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:90]


def stats(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("TotalDuration", 0)) or 0))
    with open(out, "w") as fh:
        fh.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            tot = float(r.get("TotalDurationNs", 0))
            fh.write("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |\n" % (
                short(r.get("Name", "")), r.get("Calls", ""), tot / 1e6, float(r.get("AverageNs", 0)) / 1e3,
                float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3, r.get("Percentage", "")))
    print(open(out).read())


def pmc(d, out):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
