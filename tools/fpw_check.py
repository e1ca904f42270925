"""fp3 / fp4 on fp_w_kernel (encoder_w.hip, LSDM_FP_W=1: the default) against the tile-per-work-group form (fp_pre_kernel, LSDM_FP_W=0) on the
same inputs: the stage outputs (workspace tensors fp4 / fp3 / fp2), the encoder output, max-abs and differing elements, run-to-run equality; and
the whole multi-step call in both settings (final sample max-abs).  The switch is read when a context is created."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
B = int(os.environ.get("B", 7))
hip = ctypes.CDLL("libamdhip64.so")
bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=5).items()}
clouds = bt["given_objs"].reshape(B * 9, 1024, 3).contiguous()
torch.manual_seed(0)
starts = torch.stack([torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]).to(dev)


def grab(ctx, name, numel):
    ptr, n = ctx.workspace_tensor(name)
    t = torch.zeros(numel, device=dev)
    assert numel <= n and hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(numel * 4), 3) == 0
    return t


res = {}
for v in ("0", "1", "1", "0"):
    os.environ["LSDM_FP_W"] = v
    m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
    m.load_state_dict(make_synthetic_state_dict(seed=0))
    ctx = m._context()
    m._reserve(B)
    out = torch.empty(B * 9, 1024, 3, device=dev)
    ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), B * 9, out.data_ptr(), None)
    torch.cuda.synchronize()
    taps = {"fp4": grab(ctx, "fp4", B * 9 * 64 * 256), "fp3": grab(ctx, "fp3", B * 9 * 256 * 256), "fp3_y": grab(ctx, "fp3_y", B * 9 * 64 * 256),
            "fp2_y": grab(ctx, "fp2_y", B * 9 * 256 * 256), "fp2": grab(ctx, "fp2", B * 9 * 1024 * 128), "out": out.flatten().clone()}
    torch.manual_seed(3)
    x = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False,
                           skip_timesteps=diff.num_timesteps - 6, chunk_steps=6)
    taps["chain"] = x.flatten().clone()
    res[v + ("b" if v in res else "")] = taps
    del m, ctx
for a, b in (("0", "1"), ("1", "1b"), ("0", "0b")):
    for k in res[a]:
        d = (res[a][k] - res[b][k]).abs()
        print("%-6s LSDM_FP_W %s vs %s: max-abs %.3e (|ref|max %.2f), differing elements %d of %d" % (
            k, a, b, float(d.max()), float(res[a][k].abs().max()), int((d > 0).sum()), d.numel()))
