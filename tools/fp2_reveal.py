"""Debug aid: makes fp2's interpolated rows T visible in its output without touching the kernel.  Layer 2 is set to a selection of 128
of the 256 hidden channels (identity BN), so out = ReLU(T + W1a p1 + b1)[:, half]; a wrong T entry shows up as the same difference in
one output channel.  Compares fp_ws_kernel runs with the fp_pre_kernel result and tests where wrong values come from."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
torch.set_grad_enabled(False)
DEV = "cuda:0"
B = int(os.environ.get("B", 16)); N = int(os.environ.get("N", 30)); HALF = int(os.environ.get("HALF", 0))
sd = make_synthetic_state_dict(seed=0)
w2 = torch.zeros(128, 256, 1); w2[torch.arange(128), torch.arange(128) + 128 * HALF, 0] = 1.0
sd["pcd_backbone.fp2.mlp_convs.1.weight"] = w2
sd["pcd_backbone.fp2.mlp_convs.1.bias"] = torch.zeros(128)
sd["pcd_backbone.fp2.mlp_bns.1.weight"] = torch.ones(128); sd["pcd_backbone.fp2.mlp_bns.1.bias"] = torch.zeros(128)
sd["pcd_backbone.fp2.mlp_bns.1.running_mean"] = torch.zeros(128); sd["pcd_backbone.fp2.mlp_bns.1.running_var"] = torch.full((128,), 1.0 - 1e-5)
m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
m.load_state_dict(sd)
bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=12, num_objects=8).items()}
torch.manual_seed(1)
x0 = torch.randn(B, 1024, 3, device=DEV)
starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
t = torch.full((B,), 500, device=DEV)
nc = B * 9
hip = C.CDLL("libamdhip64.so")
def ws(name, numel, dtype=torch.float32):
    ptr, n = m._ctx.workspace_tensor(name)
    out = torch.empty(numel, dtype=dtype)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(numel * 4), 2) == 0
    return out
def run(wsmode):
    if wsmode is None: os.environ.pop("LSDM_FP2_WS", None)
    else: os.environ["LSDM_FP2_WS"] = wsmode
    m(x0.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=starts)
    return ws("fp2", nc * 1024 * 128).view(nc, 1024, 128)
good = run("0")
Y = ws("fp2_y", nc * 256 * 256).view(nc, 256, 256)
idx = ws("nn_idx3", nc * 1024 * 3, torch.int32).view(nc, 1024, 3).long()
w = ws("nn_w3", nc * 1024 * 3).view(nc, 1024, 3)
def T_of(c, n):       # interpolated row (all 256 channels), float64 for reference
    return sum(Y[c, idx[c, n, k]].double() * w[c, n, k].double() for k in range(3))
shown = 0
for it in range(N):
    out = run(None)
    bad = (out != good).any(dim=2).nonzero().tolist()
    if not bad:
        continue
    print("run %d: %d wrong rows" % (it, len(bad)), flush=True)
    for c, n in bad[:6]:
        d = out[c, n] - good[c, n]
        ch = d.nonzero().flatten().tolist()
        tile = c * 16 + n // 64
        print("  cloud %d point %d (tile %d row %d): %d of 128 channels differ: %s" % (c, n, tile, n % 64, len(ch), [x + 128 * HALF for x in ch][:40]))
        Tr = T_of(c, n)
        for x in ch[:6]:
            cc = x + 128 * HALF
            got = float(Tr[cc]) + float(d[x])          # (valid where the hidden unit is positive in both runs)
            cands = {}
            for name, (c2, n2) in {"tile-32 same slot": divmod((tile - 32) * 64 + n % 64, 1024), "tile+32 same slot": divmod((tile + 32) * 64 + n % 64, 1024),
                                   "tile+64 same slot": divmod((tile + 64) * 64 + n % 64, 1024)}.items():
                if 0 <= c2 < nc:
                    cands[name] = float(T_of(c2, n2)[cc])
                    cands[name + " raw Y0"] = float(Y[c2, idx[c2, n2, 0], cc])
            terms = [float(Y[c, idx[c, n, k], cc]) for k in range(3)]
            print("     ch %d: T want %.6f, implied got %.6f (out %.6f vs %.6f); neighbours' Y %s w %s; %s" % (cc, float(Tr[cc]), got, float(out[c, n, x]), float(good[c, n, x]),
                  ["%.4f" % v for v in terms], ["%.4f" % float(v) for v in w[c, n]], {k_: "%.4f" % v for k_, v in cands.items()}))
    shown += 1
    if shown >= 3:
        break
