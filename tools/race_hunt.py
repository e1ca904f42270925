"""Determinism stress: repeats one forward while an unrelated load runs on a side stream; reports the first workspace
tensor that differs from the first run (debug aid for intra-kernel races that only show under contention)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
torch.set_grad_enabled(False)
DEV = "cuda:0"
hip = C.CDLL("libamdhip64.so")
def ws(m, name, numel):
    ptr, n = m._ctx.workspace_tensor(name)
    out = torch.empty(numel, dtype=torch.float32)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(numel * 4), 2) == 0
    return out
prec = os.environ.get("LSDM_PRECISION", "bf16x3")
B = int(os.environ.get("B", 16))
m, diff = create_model_and_diffusion("proxd", precision=prec)
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=12, num_objects=8).items()}
torch.manual_seed(1)
x0 = torch.randn(B, 1024, 3)
starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
t = torch.full((B,), 500, device=DEV)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=DEV)
nc = B * 9
names = [("feat1", nc * 1024 * 64), ("feat2", nc * 256 * 128), ("feat3", nc * 64 * 256), ("feat4", nc * 16 * 512), ("fp4", nc * 64 * 256),
         ("fp3", nc * 256 * 256), ("fp4_y", nc * 16 * 256), ("fp3_y", nc * 64 * 256), ("fp2_y", nc * 256 * 256), ("fp2", nc * 1024 * 128), ("pcd", nc * 3072), ("o12", nc * 12), ("hm", B * 3072)]
ref = None
GOOD = os.environ.get("GOOD_WS")          # e.g. GOOD_WS=0: run 0 (the reference) is made with LSDM_FP2_WS=0, the others with the environment's setting
keep_ws = os.environ.get("LSDM_FP2_WS")
for it in range(int(os.environ.get("N", 12))):
    if GOOD is not None:
        if it == 0:
            os.environ["LSDM_FP2_WS"] = GOOD
        elif keep_ws is None:
            os.environ.pop("LSDM_FP2_WS", None)
        else:
            os.environ["LSDM_FP2_WS"] = keep_ws
    if it > 0:
        with torch.cuda.stream(side):
            for _ in range(40):
                b = torch.sin(a) * 1.0001 + a
                b = b @ a[:, :512]
    x = x0.clone().to(DEV)
    _, out = m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=starts)
    cur = {n: ws(m, n, k) for n, k in names}
    cur["out"] = out.cpu().flatten(); cur["x"] = x.cpu().flatten(); cur["guid"] = m.saved_guiding_points.cpu().flatten()
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    bad = [(n, int((cur[n] != ref[n]).sum()), float((cur[n] - ref[n]).abs().max())) for n in cur if not torch.equal(cur[n], ref[n])]
    print("iter %d:" % it, bad if bad else "identical", flush=True)
    if "fp2" in dict((b[0], 1) for b in bad) and os.environ.get("WHERE"):
        d = (cur["fp2"] != ref["fp2"]).view(nc, 1024, 128)
        rows = d.any(dim=2).nonzero()
        tl = sorted(set(c * 16 + n // 64 for c, n in rows.tolist()))
        print("    bad tiles (round, wg-in-xcd): %s" % [(t_, (t_ % 288) // 32, (t_ % 288) % 32) for t_ in tl], flush=True)
        for c, n in rows.tolist()[:int(os.environ.get("WHERE"))]:
            ch = d[c, n].nonzero().flatten().tolist()
            tile = c * 16 + n // 64
            print("    cloud %d point %d (tile %d, row %d): %d channels, first %s; max err %.3g" % (c, n, tile, n % 64, len(ch), ch[:6],
                  float((cur["fp2"].view(nc, 1024, 128)[c, n] - ref["fp2"].view(nc, 1024, 128)[c, n]).abs().max())), flush=True)
