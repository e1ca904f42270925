"""Wall time of the drop-in call itself: diffusion.p_sample_loop(model, ...) for a full T = 1000 chain batch (B = 64) -- GPU steps
plus whatever the host does per chunk -- per schedule, in the parity mode (host RNG in the reference's order: draws + uploads) and in
the throughput mode (host_rng=False: device generator, nothing drawn or uploaded by the host)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_gaussian_diffusion, create_model_and_diffusion, get_default_diffusion
torch.set_grad_enabled(False)
B = int(os.environ.get("B", 64)); T = int(os.environ.get("T", 1000))
m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
m.load_state_dict(make_synthetic_state_dict(seed=0))
d = {k: v.cuda() for k, v in make_synthetic_batch(B, seed=1, num_objects=8).items()}
for host_rng in (True, False):
  diff = create_gaussian_diffusion(get_default_diffusion(), host_rng=host_rng)
  for name, kw in (("strict", {"hoist_invariants": False}), ("hoist_invariants (exact, default)", {}), ("hoist_conditioning", {"hoist_conditioning": True})):
    name = ("host RNG / " if host_rng else "device RNG / ") + name
    for rep in range(2):
        torch.manual_seed(5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = diff.p_sample_loop(m, [B, 1024, 3], d["mask"], d["given_objs"], d["given_cats"], d["text_features"], clip_denoised=False,
                               skip_timesteps=1000 - T, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-52s T=%d B=%d: %.2f s wall = %.1f steps/s (%.3f ms/step)" % (name, T, B, dt, T / dt, dt / T * 1e3), flush=True)
