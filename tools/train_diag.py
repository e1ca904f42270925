"""Diagnostic (GPU box): the G12 batch through the HIP training path vs the CPU oracle in float64 (max-pool picks of the HIP path
injected) and in fp32; per PointNet++ tensor, in network order, the relative l2 / entry-wise deviation of both from the float64 gradient.
Test infrastructure only.      python tools/train_diag.py eval|train"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from lsdm_amd.synthetic import make_synthetic_state_dict  # noqa: E402
from lsdm_amd.util.model_util import create_model_and_diffusion  # noqa: E402
from oracle import cpu_ref, train_ref  # noqa: E402
from test_oracle_golden import T, g12_inputs  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
g = load_golden("g12_training")
B, C, bt, x_start, noise, t, target_cat = g12_inputs(g)
sd = make_synthetic_state_dict(seed=0, max_cats=C)
starts = [s.long() for s in T(g[mode + "_fps_starts"])]
keep = None
mk = lambda: None
if mode == "train":
    keep = torch.from_numpy(np.unpackbits(g["train_dropout_mask"])[:B * 9 * 128 * 1024].reshape(B * 9, 128, 1024).copy())
    mk = lambda: cpu_ref.TrainMode(dropout_mask=keep.float())
torch.set_num_threads(16)
dev = torch.device("cuda", 0)
with torch.enable_grad():
    model, diffusion = create_model_and_diffusion("proxd", device=0)
    model.load_state_dict(sd)
    model.requires_grad_(True)
    model.train(mode == "train")
    d = {k: v.to(dev) for k, v in bt.items()}
    terms = diffusion.training_losses(model, x_start.to(dev), d["mask"], t.to(dev), d["given_objs"], d["given_cats"], target_cat.to(dev),
                                      y=d["text_features"], noise=noise.to(dev), fps_starts=starts, dropout_keep=keep)
    terms["loss"].backward()
picks = model._train_session.pool_choices(B)
inject = {"sa%d" % (l + 1): p for l, p in enumerate(picks)}
args = (sd, cpu_ref.Sampler(sd).tab, x_start.clone(), bt["mask"], t, bt["given_objs"], bt["given_cats"], target_cat, bt["text_features"], noise, starts)
l64, r64 = train_ref.losses_and_grads(*args, lambda_cat=0.1, train=mk(), pool_inject=inject, dtype=torch.float64)
l32, r32 = train_ref.losses_and_grads(*args, lambda_cat=0.1, train=mk(), pool_inject=inject)
print("losses f64", l64, "f32", l32, "hip", {k: float(terms[k]) for k in ("loss", "mse", "cat_loss")})
print("%-46s %10s %10s   %10s %10s" % ("parameter", "hip l2", "hip max", "f32 l2", "f32 max"))
for n, p in model.named_parameters():
    if r64[n] is None or float(r64[n].norm()) < 1e-6:
        continue
    sc, nr = float(r64[n].abs().max()), float(r64[n].norm())
    a, b = p.grad.cpu().double() - r64[n], r32[n].double() - r64[n]
    if n.startswith("pcd_backbone") or float(a.norm()) / nr > 1e-4:
        print("%-46s %10.2e %10.2e   %10.2e %10.2e" % (n, float(a.norm()) / nr, float(a.abs().max()) / sc, float(b.norm()) / nr, float(b.abs().max()) / sc))
