"""Stage outputs of one scene-encoder call (B = 7, ragged scenes) and a 6-step chain's sample with the library as it is in lsdm_amd/ (and the LSDM_*
switches of the environment), saved to the given file; `--compare a b` prints max-abs / differing elements per tensor.  For same-box A/B of library
VARIANTS (tools/build_variant.sh) that must be bit-identical."""
import ctypes
import os
import sys

import torch

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k] - b[k]).abs()
        print("%-6s max-abs %.3e (|ref|max %.2f), differing elements %d of %d" % (k, float(d.max()), float(a[k].abs().max()), int((d > 0).sum()), d.numel()))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
B = 7
hip = ctypes.CDLL("libamdhip64.so")
bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=5).items()}
clouds = bt["given_objs"].reshape(B * 9, 1024, 3).contiguous()
torch.manual_seed(0)
starts = torch.stack([torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]).to(dev)
m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
m.load_state_dict(make_synthetic_state_dict(seed=0))
ctx = m._context()
m._reserve(B)
out = torch.empty(B * 9, 1024, 3, device=dev)
ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), B * 9, out.data_ptr(), None)
torch.cuda.synchronize()


def grab(name, numel):
    ptr, n = ctx.workspace_tensor(name)
    t = torch.zeros(numel, device=dev)
    assert numel <= n and hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(numel * 4), 3) == 0
    return t.cpu()


res = {"fp4": grab("fp4", B * 9 * 64 * 256), "fp3": grab("fp3", B * 9 * 256 * 256), "fp3_y": grab("fp3_y", B * 9 * 64 * 256),
       "fp2_y": grab("fp2_y", B * 9 * 256 * 256), "fp2": grab("fp2", B * 9 * 1024 * 128), "out": out.flatten().cpu()}
torch.manual_seed(3)
x = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False,
                       skip_timesteps=diff.num_timesteps - 6, chunk_steps=6)
res["chain"] = x.flatten().cpu()
torch.save(res, sys.argv[1])
