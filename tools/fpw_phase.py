"""Per-phase s_memtime sums of fp_w_kernel (fp3 / fp4; variant built with -DLSDM_PHASE_TIMING: tools/build_variant.sh phase "-DLSDM_PHASE_TIMING"; run with the
variant copied over lsdm_amd/liblsdm_hip.so).  One wave per SIMD: the phases of wave 0 ARE the timeline of its SIMD.  fp3 and fp4 launches are summed."""
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
lib.lsdm_debug_fpw_phase(buf, 1)
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_fpw_phase(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
mhz = cal[0] / (cal[2] / 100.0)
names = ["round top (operands of the round, neighbour lists)", "layer-1 panel top (accumulator init, row requests)", "k loops (12 panels: 48 / 96 MFMAs)",
         "epilogues (bias + ReLU + split)", "layer-3 stores", "panel wait + barrier (12 per round)"]
n = buf[7]
tot = sum(buf[i] for i in range(6))
print("s_memtime = %.1f MHz; %d rounds of 128 rows over %d steps (fp3: 5 rounds per work-group and step, fp4: 1.1); us per round (share):" % (mhz, n, K))
for i in range(6):
    print("   %-52s %7.2f  (%4.1f %%)" % (names[i], buf[i] / n / mhz, 100.0 * buf[i] / tot))
print("   %-52s %7.2f" % ("total", tot / n / mhz))
