"""Wall time of one training step on the HIP path (diffusion.training_losses + backward + AdamW step), B = the reference's 6 by default.

    python tools/train_bench.py [--batch 6] [--steps 5]          (rocprofv3 --kernel-trace --stats over it: profiles/r03_train_kernel_stats.md)
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--host-rng", action="store_true", help="draws (noise, FPS starts, the 7 M-element dropout mask) from the HOST generator in the order "
                "a CPU run of the reference makes them (the parity mode of the tests: the mask alone costs 25 - 65 ms of host time per step); default: the "
                "device generator, as run/train_sdm.py does")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model, diffusion = create_model_and_diffusion("proxd", device=0)
if not a.host_rng:
    from lsdm_amd.util.model_util import create_gaussian_diffusion, get_default_diffusion
    diffusion = create_gaussian_diffusion(get_default_diffusion(), host_rng=False)
model.load_state_dict(make_synthetic_state_dict(seed=0))
model.requires_grad_(True)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)        # (as lsdm_amd/run/train_sdm.py builds it)
B = a.batch
bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=3, num_objects=8).items()}
x_start = bt["given_objs"][:, 1].clone()
target_cat = torch.zeros(B, 13, device=dev)
target_cat[:, 3] = 1
torch.manual_seed(0)
times = []
for i in range(a.steps + 1):
    t = torch.randint(0, 1000, (B,), device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    terms = diffusion.training_losses(model, x_start, bt["mask"], t, bt["given_objs"], bt["given_cats"], target_cat, y=bt["text_features"])
    terms["loss"].backward()
    opt.step()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    print("step %d: loss %.4f  %.1f ms" % (i, float(terms["loss"]), times[-1] * 1e3), flush=True)
ts = sorted(times[1:])
print("(%s RNG) B = %d: median %.1f ms per training step (forward with all activations kept + backward + AdamW), %.2f samples/s; algorithmic %.1f GFLOP "
      "(3 x 14.58 per sample) -> %.1f TFLOP/s fp32" % ("host" if a.host_rng else "device", B, ts[len(ts) // 2] * 1e3, B / ts[len(ts) // 2], 3 * 14.58 * B, 3 * 14.58 * B / ts[len(ts) // 2] / 1e3))
