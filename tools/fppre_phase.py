"""Per-phase s_memtime sums of fp_pre_kernel (fp4 / fp3; fp2 under LSDM_FP2_WS=0) from a PHASE build of the library
(tools/build_variant.sh phase "-DLSDM_PHASE_TIMING", copied over lsdm_amd/liblsdm_hip.so on the GPU box).
Wave 0 of every work-group stamps s_memtime at the phase boundaries (its rate is calibrated against the 100 MHz wall clock by
lsdm_debug_calibrate); sums per phase / work-groups = mean time of a work-group in that phase."""
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
prec = os.environ.get("LSDM_PRECISION", "bf16x3")
m, diff = create_model_and_diffusion("proxd", precision=prec)
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.cuda() for k, v in make_synthetic_batch(B, seed=1, num_objects=8).items()}
x = torch.randn(B, 1024, 3).cuda()
t = torch.full((B,), 500).cuda()
lib = _lib.load()
buf = (C.c_ulonglong * 24)()
lib.lsdm_debug_fppre_phase(buf, 1)
REPS = 5
for it in range(REPS):
    m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"])
torch.cuda.synchronize()
lib.lsdm_debug_fppre_phase(buf, 1)
cal = (C.c_ulonglong * 3)()
lib.lsdm_debug_calibrate(cal)
MHZ = cal[0] / (cal[2] / 100.0)
names = ["requests+interp->tile", "acc<-tile", "p1 tile + L1 k loops", "L1 epilogue->bufB", "L2 k loops", "L2 epilogue->bufC", "L3 / plain L2 + stores"]
print("precision %s, B = %d, %d forwards; s_memtime runs at %.1f MHz here" % (prec, B, REPS, MHZ))
for slot, nm in ((0, "D1=64 (fp2, reference form)"), (1, "D1=128 (fp3)"), (2, "D1=256 (fp4)")):
    n = buf[slot * 8 + 7]
    if not n:
        continue
    tot = sum(buf[slot * 8 + i] for i in range(7)) / n
    print("%s: %d work-groups per forward, wave 0 mean %.2f us per work-group" % (nm, n // REPS, tot / MHZ))
    for i in range(7):
        v = buf[slot * 8 + i] / n
        print("    %-26s %7.2f us  %5.1f %%" % (names[i], v / MHZ, 100.0 * v / tot))
