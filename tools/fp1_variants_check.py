"""fp1 + head's two implementations (LSDM_FP1_PIPE = 0 fp_s_kernel, 1 fp_c_kernel) on the same inputs through lsdm_scene_encoder: the encoder output
pairwise max-abs and bit equality, plus run-to-run equality of each.  (The switch is read when the context is created: one model per setting.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
B = int(os.environ.get("B", 3))
bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=5).items()}
clouds = bt["given_objs"].reshape(B * 9, 1024, 3).contiguous()
torch.manual_seed(0)
starts = torch.stack([torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]).to(dev)
res = {}
for v in ("0", "1"):
    os.environ["LSDM_FP1_PIPE"] = v
    m, _ = create_model_and_diffusion("proxd", precision="bf16x3")
    m.load_state_dict(make_synthetic_state_dict(seed=0))
    ctx = m._context()
    m._reserve(B)
    for rep in ("", "b"):
        out = torch.empty(B * 9, 1024, 3, device=dev)
        ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), B * 9, out.data_ptr(), None)
        torch.cuda.synchronize()
        res[v + rep] = out.clone()
    del m, ctx
for a, b in (("0", "1"), ("0", "0b"), ("1", "1b")):
    d = (res[a] - res[b]).abs()
    print("encoder out %s vs %s: max-abs %.3e, differing elements %d of %d (finite: %s)" % (a, b, float(d.max()), int((d > 0).sum()), d.numel(), bool(torch.isfinite(res[b]).all())))
