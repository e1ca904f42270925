"""fp2's implementations (LSDM_FP2_WS = 0 fp_pre_kernel, 4 fp_d_kernel) on the same inputs: the stage's output
(workspace tensor fpo2) and the encoder output, pairwise max-abs and bit equality, plus run-to-run equality of each."""
import os
import sys

import ctypes

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
B = int(os.environ.get("B", 3))
m, _ = create_model_and_diffusion("proxd", precision="bf16x3")
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.to(dev) for k, v in make_synthetic_batch(B, seed=5).items()}
ctx = m._context()
m._reserve(B)
clouds = bt["given_objs"].reshape(B * 9, 1024, 3).contiguous()
torch.manual_seed(0)
starts = torch.stack([torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]).to(dev)
res = {}
for v in ("0", "4", "0", "4"):
    os.environ["LSDM_FP2_WS"] = v
    out = torch.empty(B * 9, 1024, 3, device=dev)
    ctx.call("lsdm_scene_encoder", clouds.data_ptr(), starts.data_ptr(), B * 9, out.data_ptr(), None)
    torch.cuda.synchronize()
    ptr, n = ctx.workspace_tensor("fp2")
    fpo2 = torch.zeros(B * 9 * 1024 * 128, device=dev)
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(fpo2.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(fpo2.numel() * 4), 3) == 0
    key = v + ("b" if v in res else "")
    res[key] = (fpo2.clone(), out.clone())
for a, b in (("0", "4"), ("0", "0b"), ("4", "4b")):
    d = (res[a][0] - res[b][0]).abs()
    print("fpo2 %s vs %s: max-abs %.3e, differing elements %d of %d; encoder out max-abs %.3e" % (
        a, b, float(d.max()), int((d > 0).sum()), d.numel(), float((res[a][1] - res[b][1]).abs().max())))
bad = (res["0"][0] - res["4"][0]).abs().view(B * 9, 1024, 128)
if float(bad.max()) > 0:
    idx = (bad > 0).nonzero()
    print("first differing (cloud, point, channel):", idx[:8].tolist(), " clouds touched:", sorted(set(idx[:, 0].tolist()))[:12],
          " points per cloud hist (tile = point // 64):", torch.bincount((idx[:, 1] // 64), minlength=16).tolist())
    c0, t0 = int(idx[0, 0]), int(idx[0, 1]) // 64
    blk = bad[c0, t0 * 64:(t0 + 1) * 64]                       # (64 rows, 128 channels) of the first differing tile
    ref = res["0"][0].view(B * 9, 1024, 128)[c0, t0 * 64:(t0 + 1) * 64]
    print("tile (cloud %d, tile %d): %d differing; per channel block %s; per row block %s; max rel %.2e; rows %s" % (
        c0, t0, int((blk > 0).sum()), [(int((blk[:, q * 32:(q + 1) * 32] > 0).sum())) for q in range(4)],
        [int((blk[q * 32:(q + 1) * 32] > 0).sum()) for q in range(2)], float((blk / (ref.abs() + 1e-6)).max()),
        sorted(set((blk > 0).nonzero()[:, 0].tolist()))[:20]))
    # is the difference already in the stage's INPUT-side (interpolated + layer 1) or only in layer 2?  compare with the zero pattern of ReLU
    print("values at first diffs:", [(float(ref[r, c]), float(res["4"][0].view(B * 9, 1024, 128)[c0, t0 * 64 + r, c])) for r, c in (blk > 0).nonzero()[:4].tolist()])
