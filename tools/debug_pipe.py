import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
torch.set_grad_enabled(False)
DEV = "cuda:0"
for prec in ("f32", "bf16x3"):
    B, K = 2, 4
    m, diff = create_model_and_diffusion("proxd", precision=prec)
    m.load_state_dict(make_synthetic_state_dict(seed=0))
    bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=12).items()}
    res = {}
    for name, pipe in (("p1", 1), ("s1", 0), ("p2", 1), ("s2", 0), ("p3", 1)):
        m._context().call("lsdm_set_pipeline", pipe)
        torch.manual_seed(8)
        res[name] = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"],
                                       clip_denoised=False, skip_timesteps=1000 - K, chunk_steps=K).clone()
        torch.cuda.synchronize()
    names = list(res)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            d = (res[names[i]] - res[names[j]]).abs()
            print(prec, names[i], names[j], "max %.3e  nnz %d" % (float(d.max()), int((d > 0).sum())))
