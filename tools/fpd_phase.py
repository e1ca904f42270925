"""Per-phase s_memtime sums of fp_d_kernel's two roles (variant built with -DLSDM_PHASE_TIMING: tools/build_variant.sh phase "-DLSDM_PHASE_TIMING";
run with the variant copied over lsdm_amd/liblsdm_hip.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner  # noqa: E402
from lsdm_amd import _lib  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
K = 8
r = Runner("bf16x3", 64, 13, K, 1, 0, dev)
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_fpd_phase(buf, 1)
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_fpd_phase(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
mhz = cal[0] / (cal[2] / 100.0)
print("s_memtime = %.1f MHz" % mhz)
for base, role, names in ((0, "A", ["p1 split + fetch issue", "rows -> accumulators", "layer 1", "ReLU+split->bufB", "wait rows (vmcnt)", "barrier"]),
                          (8, "B", ["layer 2", "stores", "barrier", "-", "-", "-"])):
    n = buf[base + 7]
    if not n:
        continue
    tot = sum(buf[base + i] for i in range(6))
    print("role %s: %d tile-iterations over %d launches; us per tile: " % (role, n, K) +
          ", ".join("%s %.2f" % (names[i], buf[base + i] / n / mhz) for i in range(6)) + " | total %.2f" % (tot / n / mhz))
