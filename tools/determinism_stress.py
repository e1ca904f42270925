"""Determinism stress: the same forward N times on one context, every stage's workspace tensor and the outputs compared ON THE DEVICE
with the first run (and fp2 with the tile-per-work-group kernel's result, LSDM_FP2_WS=0).  Prints, per tensor, how many runs differed.
Found the packed-fp32 hazard of fp_ws_kernel's gather waves in round 4; tests/test_hip_determinism.py runs a short form."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402


def stress(B=16, N=200, precision="bf16x3", device="cuda:0", verbose=False):
    torch.set_grad_enabled(False)
    m, _ = create_model_and_diffusion("proxd", precision=precision)
    m.load_state_dict(make_synthetic_state_dict(seed=0))
    bt = {k: v.to(device) for k, v in make_synthetic_batch(B, seed=12, num_objects=8).items()}
    torch.manual_seed(1)
    x0 = torch.randn(B, 1024, 3, device=device)
    starts = [torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]
    t = torch.full((B,), 500, device=device)
    nc = B * 9
    hip = C.CDLL("libamdhip64.so")
    names = [("feat1", nc * 1024 * 64), ("feat2", nc * 256 * 128), ("feat3", nc * 64 * 256), ("feat4", nc * 16 * 512), ("fp4", nc * 64 * 256),
             ("fp3", nc * 256 * 256), ("fp2", nc * 1024 * 128), ("pcd", nc * 3072), ("o12", nc * 12), ("hm", B * 3072)]

    def grab(name, numel):
        ptr, _n = m._ctx.workspace_tensor(name)
        out = torch.empty(numel, dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(numel * 4), 3) == 0
        return out

    def run():
        x = x0.clone()
        _, out = m(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], y=bt["text_features"], fps_starts=starts)
        cur = {n: grab(n, k) for n, k in names}
        cur["out"] = out.flatten().clone()
        cur["x"] = x.flatten().clone()
        cur["guiding"] = m.saved_guiding_points.flatten().clone()
        return cur

    keep = os.environ.get("LSDM_FP2_WS")
    os.environ["LSDM_FP2_WS"] = "0"
    fp2_tile_form = run()["fp2"]
    if keep is None:
        os.environ.pop("LSDM_FP2_WS")
    else:
        os.environ["LSDM_FP2_WS"] = keep
    ref = run()
    counts = {n: 0 for n in ref}
    counts["fp2 vs LSDM_FP2_WS=0"] = int(not torch.equal(ref["fp2"], fp2_tile_form))
    for it in range(1, N):
        cur = run()
        for n in ref:
            if not torch.equal(cur[n], ref[n]):
                counts[n] += 1
                if verbose:
                    print("run %d: %s differs in %d elements" % (it, n, int((cur[n] != ref[n]).sum())), flush=True)
        if not torch.equal(cur["fp2"], fp2_tile_form):
            counts["fp2 vs LSDM_FP2_WS=0"] += 1
    return counts


def stress_loop(B=64, K=6, N=30, precision="bf16x3", device="cuda:0", entry="invariant"):
    """The benched entry points: N calls of K sampler steps from the same x_T and RNG streams (lsdm_sample_steps_invariant, lsdm_sample_steps
    or lsdm_sample_steps_hoisted); the final sample, predicted x_0, guiding points and class logits compared with the first call's."""
    torch.set_grad_enabled(False)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import Runner
    dev = torch.device(device)
    r = Runner(precision, B, 13, K, 1, dev.index or 0, dev)
    x0 = r.x.clone()
    ref, bad = None, 0
    for it in range(N):
        x = x0.clone()
        r.run(r.T - 1, 0, K, x=x, invariant=entry == "invariant", hoisted=entry == "hoisted")
        torch.cuda.synchronize()
        cur = [x.clone(), r.out_x0.clone(), r.out_g.clone(), r.out_cat.clone()]
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
    return bad


if __name__ == "__main__":
    B = int(os.environ.get("B", 16))
    N = int(os.environ.get("N", 200))
    prec = os.environ.get("LSDM_PRECISION", "bf16x3")
    c = stress(B, N, prec, verbose=bool(os.environ.get("VERBOSE")))
    bad = {k: v for k, v in c.items() if v}
    print("determinism stress (%s, B=%d, %d runs): %s" % (prec, B, N, "all tensors identical in every run" if not bad else "DIFFERENCES (runs per tensor): %s" % bad), flush=True)
    nb = 0
    for entry in ("invariant", "strict", "hoisted"):
        k = stress_loop(B=int(os.environ.get("LOOP_B", 64)), K=6, N=int(os.environ.get("LOOP_N", 30)), precision=prec, entry=entry)
        print("loop stress (%s, %s entry point, B=%s, 6 steps per call): %d calls differ from the first" % (prec, entry, os.environ.get("LOOP_B", 64), k), flush=True)
        nb += k
    sys.exit(1 if bad or nb else 0)
