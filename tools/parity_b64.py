"""One p_sample step of the BENCH workload (B = 64, 8 objects + human, C = 13) through the HIP path vs the CPU oracle on the
same seeds -- the batch-size dependent quirks (Q2/Q3) at the size the headline number is quoted on.  ~1-2 min of CPU oracle."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
from lsdm_amd.util.model_util import create_model_and_diffusion
from oracle import cpu_ref

torch.set_grad_enabled(False)
B = int(os.environ.get("B", 64))
sd = make_synthetic_state_dict(seed=0)
bt = make_synthetic_batch(B, seed=1, num_objects=8)
t = torch.tensor([(991 - 7 * b) % 1000 for b in range(B)])
torch.manual_seed(1)
x0 = torch.randn(B, 1024, 3)
torch.set_num_threads(min(16, os.cpu_count()))
t0 = time.time()
torch.manual_seed(2)
smp = cpu_ref.Sampler(sd)
xr = x0.clone()
ref = smp.p_sample(xr, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"])
print("oracle: %.1f s" % (time.time() - t0), flush=True)
d = {k: v.cuda() for k, v in bt.items()}
for precision in ("f32", "bf16x3"):
    m, diff = create_model_and_diffusion("proxd", precision=precision)
    m.load_state_dict(sd)
    torch.manual_seed(2)
    xg = x0.clone().cuda()
    out = diff.p_sample(m, xg, d["mask"], t.cuda(), d["given_objs"], d["given_cats"], d["text_features"], clip_denoised=False)
    torch.cuda.synchronize()
    print("B=%d %s: |sample|max %.1f, |pred_xstart|max %.1f; max-abs sample %.3e, pred_xstart %.3e, saved_cat %.3e, guiding %.3e" % (
        B, precision, float(ref["sample"].abs().max()), float(ref["pred_xstart"].abs().max()), float((out["sample"].cpu() - ref["sample"]).abs().max()), float((out["pred_xstart"].cpu() - ref["pred_xstart"]).abs().max()),
        float((m.saved_cat.cpu() - smp.saved_cat).abs().max()), float((m.saved_guiding_points.cpu() - smp.saved_guiding_points).abs().max())),
        flush=True)

    es = (out["sample"].cpu() - ref["sample"]).abs().amax(dim=(1, 2))
    ex = (xg.cpu() - xr).abs().amax(dim=(1, 2))
    worst = int(es.argmax())
    print("   per-chain sample err: worst chain %d (t=%d) %.3e; x_mut err there %.3e, max x_mut err %.3e; |x_mut|max %.1f" % (
        worst, int(t[worst]), float(es[worst]), float(ex[worst]), float(ex.max()), float(xr.abs().max())), flush=True)
    noise_term = ref["sample"] - (out["sample"].cpu())
    idx = (out["sample"].cpu() - ref["sample"]).abs().flatten().argmax()
    print("   worst element: ref %.6f got %.6f" % (float(ref["sample"].flatten()[idx]), float(out["sample"].cpu().flatten()[idx])))
