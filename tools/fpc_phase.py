"""Per-phase s_memtime sums of fp_c_kernel's roles (variant built with -DLSDM_PHASE_TIMING: tools/build_variant.sh phase "-DLSDM_PHASE_TIMING";
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
buf = (C.c_ulonglong * 32)()
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_fpc_phase(buf, 1)
r.run(r.T - 1, 0, K, invariant=True)
torch.cuda.synchronize()
lib.lsdm_debug_fpc_phase(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
mhz = cal[0] / (cal[2] / 100.0)
print("s_memtime = %.1f MHz" % mhz)
names = ["row requests / top", "stage step", "interpolation -> X0", "barrier", "-", "-"]
for role, what in ((0, "waves 0,1: gather + stage 0"), (1, "waves 2,3: gather + stage 1"), (2, "waves 4,5: gather + stage 2"), (3, "waves 6,7: conv1 + conv2")):
    n = buf[role * 8 + 7]
    if not n:
        continue
    tot = sum(buf[role * 8 + i] for i in range(6))
    print("%s: %d iterations; us per iteration: " % (what, n) + ", ".join("%s %.2f" % (names[i], buf[role * 8 + i] / n / mhz) for i in range(4)) + " | total %.2f" % (tot / n / mhz))
