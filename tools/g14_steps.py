"""Per-step max-abs of the HIP path against golden G14 (configs[1], B = 16, C = 43, first / last five steps), f32 and bf16x3."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
DEV = "cuda:0"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in ("first", "last"):
    g = np.load(os.path.join(root, "tests", "golden", "g14_config2_%s.npz" % tag))
    B, C, K, seed, bt_seed = [int(v) for v in g["meta"]]
    kw = {}
    if tag == "last":
        gen = torch.Generator().manual_seed(int(g["init_seed"]))
        kw = dict(init_image=(0.7 * torch.randn(B, 1024, 3, generator=gen)).to(DEV), skip_timesteps=1000 - K)
    for prec in ("f32", "bf16x3"):
        m, diff = create_model_and_diffusion("proxd", max_cats=C, precision=prec)
        m.load_state_dict(make_synthetic_state_dict(seed=0, max_cats=C))
        bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=bt_seed, max_cats=C).items()}
        torch.manual_seed(seed)
        got = []
        for o in diff.p_sample_loop_progressive(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False, **kw):
            got.append(o["sample"].cpu())
            if len(got) == K:
                break
        per = [float((got[k][:, ::8] - torch.from_numpy(g["steps_sub"][k])).abs().max()) for k in range(K)]
        cnt = [int(((got[k][:, ::8] - torch.from_numpy(g["steps_sub"][k])).abs() > 1e-4).sum()) for k in range(K)]
        import hashlib
        print("   digest of the 5 samples:", hashlib.md5(torch.stack(got).numpy().tobytes()).hexdigest()[:12], " elements > 1e-4 per step:", cnt)
        d = (got[-1] - torch.from_numpy(g["final"])).abs()
        chain_worst = d.view(B, -1).max(dim=1)[0]
        print("%s %-6s per-step max-abs %s; final per chain: max %.2e, chains above 3e-4: %s" % (
            tag, prec, " ".join("%.2e" % v for v in per), float(chain_worst.max()), [(i, "%.1e" % float(v)) for i, v in enumerate(chain_worst) if v > 3e-4]))
