import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
torch.set_grad_enabled(False)
DEV = "cuda:0"
hip = C.CDLL("libamdhip64.so")
def ws(m, name, dtype=torch.float32):
    ptr, n = m._ctx.workspace_tensor(name)
    out = torch.empty(n, dtype=dtype)
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * out.element_size()), 2) == 0
    return out
B, K = 2, 2
m, diff = create_model_and_diffusion("proxd", precision="bf16x3")
m.load_state_dict(make_synthetic_state_dict(seed=0))
bt = {k: v.to(DEV) for k, v in make_synthetic_batch(B, seed=12).items()}
names_f = ["xyz1", "xyz2", "xyz3", "xyz4", "nn_w1", "nn_w2", "nn_w3", "nn_w4", "e0", "w", "tr"]
names_i = ["fps_idx1", "fps_idx2", "fps_idx3", "fps_idx4", "ball1", "ball2", "ball3", "ball4", "nn_idx1", "nn_idx2", "nn_idx3", "nn_idx4"]
ref = None
for run in range(6):
    m._context().call("lsdm_set_pipeline", 1)
    torch.manual_seed(8)
    x = diff.p_sample_loop(m, [B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], clip_denoised=False,
                           skip_timesteps=1000 - K, chunk_steps=K).clone()
    cur = {"x": x.cpu().flatten()}
    for sx in ("", "_b"):
        for n in names_f:
            cur[n + sx] = ws(m, n + sx)
        for n in names_i:
            cur[n + sx] = ws(m, n + sx, torch.int32)
    for n in ("feat1", "feat2", "feat3", "feat4", "fp4", "fp3", "fp2", "pcd", "o12", "hm"):
        cur[n] = ws(m, n)
    if ref is None:
        ref = cur
        continue
    bad = [(n, int((cur[n] != ref[n]).sum())) for n in cur if not torch.equal(cur[n], ref[n])]
    print("run %d:" % run, bad if bad else "identical", flush=True)
# where do the FPS index sequences diverge?
a, b = ref["fps_idx1_b"].view(-1, 1024), cur["fps_idx1_b"].view(-1, 1024)
for c in range(a.shape[0]):
    d = (a[c] != b[c]).nonzero().flatten()
    if len(d):
        i = int(d[0])
        print("cloud %d: first diff at %d, %d diffs; ref %s cur %s" % (c, i, len(d), a[c, max(0, i - 2):i + 4].tolist(), b[c, max(0, i - 2):i + 4].tolist()))
xa, xb = ref["xyz1_b"].view(-1, 1024, 3), cur["xyz1_b"].view(-1, 1024, 3)
print("xyz1_b rows differing per cloud:", [(c, int((xa[c] != xb[c]).any(dim=1).sum())) for c in range(xa.shape[0]) if not torch.equal(xa[c], xb[c])])
