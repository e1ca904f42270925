"""Per-phase s_memtime sums of sa_w_kernel (sa4; variant built with -DLSDM_PHASE_TIMING: tools/build_variant.sh phase "-DLSDM_PHASE_TIMING"; run with the
variant copied over lsdm_amd/liblsdm_hip.so).  One wave per SIMD: the phases of wave 0 ARE the timeline of its SIMD."""
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
buf = (C.c_ulonglong * 8)()
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_saw_phase(buf, 1)
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_saw_phase(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
mhz = cal[0] / (cal[2] / 100.0)
names = ["prologue (V row, U - V, ReLU, split)", "layer-2 k loops (4 panels x 96 MFMAs)", "layer-2 epilogues (ReLU + split)", "layer-3 k loops (8 panels x 96 MFMAs)",
         "layer-3 max + store", "panel wait + barrier (12 per round)"]
n = buf[7]
tot = sum(buf[i] for i in range(6))
print("s_memtime = %.1f MHz; %d rounds of 128 rows over %d launches; us per round (share):" % (mhz, n, K))
for i in range(6):
    print("   %-44s %7.2f  (%4.1f %%)" % (names[i], buf[i] / n / mhz, 100.0 * buf[i] / tot))
print("   %-44s %7.2f" % ("total", tot / n / mhz))
