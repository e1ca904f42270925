"""Stage taps of one forward at the bench batch size (B = 64) vs the CPU oracle: which tensor carries the largest deviation."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
from oracle import cpu_ref
torch.set_grad_enabled(False)
hip = C.CDLL("libamdhip64.so")
def ws(m, name, shape):
    ptr, n = m._ctx.workspace_tensor(name)
    out = torch.empty(int(np.prod(shape)), dtype=torch.float32)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(out.numel() * 4), 2) == 0
    return out.view(*shape)
B = int(os.environ.get("B", 64))
sd = make_synthetic_state_dict(seed=0)
bt = make_synthetic_batch(B, seed=1, num_objects=8)
t = torch.tensor([(991 - 7 * b) % 1000 for b in range(B)])
torch.manual_seed(1)
x0 = torch.randn(B, 1024, 3)
torch.manual_seed(2)
starts = cpu_ref.draw_fps_starts(B * 9)
torch.set_num_threads(min(16, os.cpu_count()))
taps = {}
xr = x0.clone()
t0 = time.time()
cat_o, out_o, guid_o = cpu_ref.forward(sd, xr, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"], fps_starts=starts, taps=taps)
print("oracle %.1f s; taps: %s" % (time.time() - t0, sorted(taps)[:40]), flush=True)
d = {k: v.cuda() for k, v in bt.items()}
m, _ = create_model_and_diffusion("proxd", precision=os.environ.get("LSDM_PRECISION", "f32"))
m.load_state_dict(sd)
x = x0.clone().cuda()
out_cat, out = m(x, d["mask"], t.cuda(), d["given_objs"], d["given_cats"], y=d["text_features"], fps_starts=starts)
torch.cuda.synchronize()
def rep(name, got, ref):
    e = (got.cpu().double() - ref.double()).abs()
    per = e.reshape(B, -1).amax(dim=1)
    print("%-8s max-abs %.3e (|ref|max %.3g) worst chain %d" % (name, float(e.max()), float(ref.abs().max()), int(per.argmax())), flush=True)
rep("hm", ws(m, "hm", (B, 1024, 3)), taps["hm"])
rep("pcd", ws(m, "pcd", (B, 9, 3072)), taps["pcd"])
rep("o12", ws(m, "o12", (B, 9, 12)), taps["o"])
rep("w", ws(m, "w", (B, 9)), taps["w"])
rep("tr", ws(m, "tr", (B, 9, 12)), taps["tr"])
if "g" in taps: rep("g", (x.cpu() - x0), taps["g"] if taps["g"].shape == x0.shape else (xr - x0))
rep("x_mut", x, xr)
rep("out", out, out_o)

# discrete decisions of the geometry at this batch size
def wsi(name, shape):
    ptr, n = m._ctx.workspace_tensor(name)
    out = torch.empty(int(np.prod(shape)), dtype=torch.int32)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(out.numel() * 4), 2) == 0
    return out.view(*shape).long()
nc = B * 9
S = (1024, 256, 64, 16)
for l in range(4):
    fi = wsi("fps_idx%d" % (l + 1), (nc, S[l])); bi = wsi("ball%d" % (l + 1), (nc, S[l], 32))
    print("level %d: fps mismatches %d, ball mismatches %d" % (l + 1, int((fi != taps["sa%d.fps_idx" % (l + 1)]).sum()), int((bi != taps["sa%d.ball_idx" % (l + 1)]).sum())), flush=True)
nn_n = {"fp4": (1, 64), "fp3": (2, 256), "fp2": (3, 1024), "fp1": (4, 1024)}
for name, (lv, n) in nn_n.items():
    gi = wsi("nn_idx%d" % lv, (nc, n, 3)); ri = taps[name + ".nn_idx"]
    same_set = (gi.sort(dim=-1)[0] == ri.sort(dim=-1)[0]).all(dim=-1)
    bad = (~same_set).nonzero()
    print("%s: points whose 3-NN SET differs from the oracle: %d of %d; clouds: %s" % (name, bad.shape[0], nc * n, sorted(set((bad[:, 0] // 9).tolist()))[:10]), flush=True)
for name in ("sa1", "sa2", "sa3", "sa4"):
    lv = int(name[2]); C_ = (64, 128, 256, 512)[lv - 1]
    got = ws(m, "feat%d" % lv, (nc, S[lv - 1], C_)); e = (got - taps[name + ".new_points"]).abs().reshape(nc, -1).amax(dim=1)
    print("%s features: max-abs %.3e (cloud %d of chain %d)" % (name, float(e.max()), int(e.argmax()), int(e.argmax()) // 9), flush=True)
for name, lv, shape in (("fp4", 4, (nc, 64, 256)), ("fp3", 3, (nc, 256, 256)), ("fp2", 2, (nc, 1024, 128))):
    got = ws(m, name, shape); e = (got - taps[name + ".out"]).abs().reshape(nc, -1).amax(dim=1)
    print("%s out: max-abs %.3e (cloud %d of chain %d)" % (name, float(e.max()), int(e.argmax()), int(e.argmax()) // 9), flush=True)
