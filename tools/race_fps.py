"""Stress: lsdm_fps on one stream while lsdm_scene_encoder runs on another; FPS outputs must never change."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
torch.set_grad_enabled(False)
DEV = "cuda:0"
prec = os.environ.get("LSDM_PRECISION", "bf16x3")
B = int(os.environ.get("B", 2))
m, _ = create_model_and_diffusion("proxd", precision=prec)
m.load_state_dict(make_synthetic_state_dict(seed=0))
ctx = m._context(); m._reserve(B)
bt = make_synthetic_batch(B, seed=12)
clouds = bt["given_objs"].reshape(B * 9, 1024, 3).to(DEV).contiguous()
nc = B * 9
torch.manual_seed(3)
starts = torch.stack([torch.randint(0, n, (nc,)) for n in (1024, 1024, 256, 64)]).to(DEV)
S = (1024, 256, 64, 16)
def fps(stream):
    idx = [torch.empty(nc, s, dtype=torch.int32, device=DEV) for s in S]
    xyz = [torch.empty(nc, s, 3, device=DEV) for s in S]
    ctx.call("lsdm_fps", clouds.data_ptr(), starts.data_ptr(), nc, *[i.data_ptr() for i in idx], *[x.data_ptr() for x in xyz], stream.cuda_stream)
    return idx
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
ref = fps(sA); torch.cuda.synchronize()
out = torch.empty(nc, 1024, 3, device=DEV)
bad = 0
for it in range(int(os.environ.get("N", 30))):
    for _ in range(3):
        ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), nc, out.data_ptr(), sB.cuda_stream)
    cur = fps(sA)
    torch.cuda.synchronize()
    d = [int((a != b).sum()) for a, b in zip(ref, cur)]
    if any(d):
        bad += 1
        print("iter %d: FPS differs %s" % (it, d), flush=True)
print("precision %s B=%d: %d of %d concurrent FPS runs differ" % (prec, B, bad, int(os.environ.get("N", 30))))
