"""Per-stack study of a TWO-term split product for the dense stacks.  CPU only: emulation on the oracle.

bf16x3 evaluates a * w as a_lo*w_hi + a_hi*w_lo + a_hi*w_hi (three MFMAs per product, dropped term 2^-18 relative).  Two MFMAs per
product are only possible with ONE activation plane: a16 * (w_hi + w_lo), where a16 is the activation rounded to a 16-bit format and
the weight keeps both planes (a 16-bit plane pair carries >= 16 mantissa bits).  The error of that scheme is the rounding of the layer's
INPUT activation -- 2^-12 relative for an fp16 plane (11-bit significand), 2^-9 for a bf16 plane -- so it is emulated exactly by
rounding the input of the chosen stacks' conv / linear calls on the fp32 oracle (weights and accumulation stay fp32).

For every stack alone and for groups of stacks: max-abs deviation from the exact oracle of one p_sample step (B = 2, t = 900: the
encoder output `pcd`, the mutated x, pred_xstart, the sample) and of the final sample of a K = 4 step chain.  The bar is 1e-3 on the
sample with >= 3x margin (<= 3.3e-4) for a mode that claims fp32-class results.

    python tools/two_term_study.py [--plane fp16|bf16] [--chain 4]
"""
import argparse
import contextlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict  # noqa: E402
from oracle import cpu_ref as R  # noqa: E402

torch.set_grad_enabled(False)
ap = argparse.ArgumentParser()
ap.add_argument("--plane", default="fp16", choices=["fp16", "bf16"])
ap.add_argument("--chain", type=int, default=4)
a = ap.parse_args()
PLANE = torch.float16 if a.plane == "fp16" else torch.bfloat16

STATE = {"tag": "cond", "on": set()}


class QF:
    """torch.nn.functional with the inputs of conv / linear rounded to one 16-bit plane inside the active stacks."""

    def __getattr__(self, k):
        return getattr(torch.nn.functional, k)

    @staticmethod
    def _q(x):
        return x.to(PLANE).float() if STATE["tag"] in STATE["on"] else x

    def conv1d(self, x, w, b=None, *r, **k):
        return torch.nn.functional.conv1d(self._q(x), w, b, *r, **k)

    def conv2d(self, x, w, b=None, *r, **k):
        return torch.nn.functional.conv2d(self._q(x), w, b, *r, **k)

    def linear(self, x, w, b=None):
        return torch.nn.functional.linear(self._q(x), w, b)


R.F = QF()


def tagged(fn, tag_of):
    def wrap(*args, **kw):
        prev = STATE["tag"]
        STATE["tag"] = tag_of(*args, **kw)
        try:
            return fn(*args, **kw)
        finally:
            STATE["tag"] = prev
    return wrap


R.set_abstraction = tagged(R.set_abstraction, lambda sd, name, *r, **k: name)
R.feature_propagation = tagged(R.feature_propagation, lambda sd, name, *r, **k: name)
R.denoise_mlp = tagged(R.denoise_mlp, lambda *r, **k: "denoiser")
R.human_backbone = tagged(R.human_backbone, lambda *r, **k: "posa")
_enc = R.scene_encoder
# (the encoder head conv1 / conv2 run inside scene_encoder itself: tag "head"; the SA / FP calls inside re-tag themselves)
R.scene_encoder = tagged(_enc, lambda *r, **k: "head")

STACKS = ["sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2", "fp1", "head", "denoiser", "posa", "cond"]
GROUPS = {"all SA": ["sa1", "sa2", "sa3", "sa4"], "all FP + head": ["fp4", "fp3", "fp2", "fp1", "head"],
          "encoder (SA + FP + head)": STACKS[:9], "everything": STACKS}

B = 2
sd = make_synthetic_state_dict(seed=0)
bt = make_synthetic_batch(B, seed=3)
t = torch.tensor([900] * B)
torch.manual_seed(1)
x0 = torch.randn(B, 1024, 3)


def step(on):
    STATE["on"] = set(on)
    smp = R.Sampler(sd)
    taps = {}
    torch.manual_seed(2)
    x = x0.clone()
    out = smp.p_sample(x, bt["mask"], t, bt["given_objs"], bt["given_cats"], bt["text_features"], taps=taps)
    return {"pcd": taps.get("pcd"), "x_mut": x, "pred_xstart": out["pred_xstart"], "sample": out["sample"]}


def chain(on, K):
    STATE["on"] = set(on)
    smp = R.Sampler(sd)
    torch.manual_seed(5)
    return smp.sample_loop([B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], skip_timesteps=1000 - K)


ref = step([])
ref_chain = chain([], a.chain)
mx = lambda u, v: float((u - v).abs().max()) if u is not None and v is not None else float("nan")
print("one activation plane = %s (relative rounding 2^-%d); B = %d, t = 900; chain of K = %d last steps" % (a.plane, 12 if a.plane == "fp16" else 9, B, a.chain))
print("| stacks on two-term products | pcd | mutated x | pred_xstart | sample | chain final |")
print("|---|---|---|---|---|---|")
rows = [(s, [s]) for s in STACKS] + list(GROUPS.items())
for name, on in rows:
    r = step(on)
    c = chain(on, a.chain)
    print("| %s | %.1e | %.1e | %.1e | %.1e | %.1e |" % (name, mx(r["pcd"], ref["pcd"]), mx(r["x_mut"], ref["x_mut"]), mx(r["pred_xstart"], ref["pred_xstart"]),
                                                      mx(r["sample"], ref["sample"]), mx(c, ref_chain)), flush=True)
