"""Per-phase s_memtime sums of the sa_h kernels (debug build: make -C lsdm_amd/csrc clean all PHASE=1)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd import _lib  # noqa: E402
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
B = int(os.environ.get("B", 64))
m, diff = create_model_and_diffusion("proxd", precision=os.environ.get("LSDM_PRECISION", "bf16x3"))
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.cuda() for k, v in make_synthetic_batch(B, seed=1, num_objects=8).items()}
x = torch.randn(B, 1024, 3).cuda()
t = torch.full((B,), 500).cuda()
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
for it in range(3):
    m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"])
    torch.cuda.synchronize()
    lib.lsdm_debug_phase_counters(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
print("calibration over %d wall ticks (100 MHz): s_memtime %d, clock64 %d -> s_memtime = %.1f MHz" % (cal[2], cal[0], cal[1], cal[0] / (cal[2] / 100.0)))
names = ["gather", "bar0", "layer1", "bar1", "layer2", "bar2", "layer3+pool", "wgs"]
for lvl in range(4):
    n = buf[lvl * 8 + 7]
    if not n:
        continue
    print("sa%d: %d WGs; mean s_memtime ticks per WG (100 MHz ticks? see ratio): " % (lvl + 1, n) +
          ", ".join("%s %.0f" % (names[i], buf[lvl * 8 + i] / n) for i in range(7)) + "  | total %.0f" % (sum(buf[lvl * 8 + i] for i in range(7)) / n))
