"""CPU oracle of the TRAINING objective of the path: ``GaussianDiffusion.training_losses`` + its gradients.

*** TEST INFRASTRUCTURE (see oracle/cpu_ref.py): only tests/ and oracle/make_goldens.py import this file.

Restates diffusion/gaussian_diffusion.py:1256-1342 (START_X, FIXED_SMALL, LossType.MSE -- the configuration of
util/model_util.py:127-163):

    x_t      = q_sample(x_start, t, noise)                         (:233-256)
    out_cat, model_output = model(x_t, mask, t, given_objs, given_cats, y)
    cat_loss = lambda_cat * CrossEntropy(out_cat.squeeze(1), argmax(target_cat))     (:1296-1301; out_cat is ALREADY a softmax)
    mse      = chamfer_distance(model_output, x_start)[0]           (:1331; pytorch3d)
    loss     = mse + cat_loss

The network is oracle/cpu_ref.forward (plain torch ops), so ``torch.autograd`` through it IS the gradient oracle.  Pinned by G12
(oracle/make_goldens.py::g12): the three loss terms and d loss / d theta of every parameter recorded from the REAL reference in
``model.eval()`` and in ``model.train()`` (batch-statistics BatchNorm + the Dropout(0.5) of the encoder head with the recorded
mask).  ``pytorch3d`` is absent from /root/reference and from this image: ``chamfer_distance`` below restates its published
definition (point_reduction = batch_reduction = "mean", squared L2, both directions summed) and G12 was recorded with the same
restatement patched into the reference -- **the chamfer term's parity is unpinned**; everything else of the objective is pinned.
"""
import torch
import torch.nn.functional as F

from . import cpu_ref


def chamfer_distance(x, y):
    """pytorch3d.loss.chamfer_distance(x, y)[0] with its defaults: x (B,P1,3), y (B,P2,3) ->
    mean_b [ mean_i min_j |x_i - y_j|^2 + mean_j min_i |x_i - y_j|^2 ]."""
    d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)
    return (d.min(dim=2)[0].mean(dim=1) + d.min(dim=1)[0].mean(dim=1)).mean()


def q_sample(tab, x_start, t, noise):
    """gaussian_diffusion.py:233-256."""
    return (cpu_ref._coef(tab["sqrt_alphas_cumprod"], t, x_start) * x_start
            + cpu_ref._coef(tab["sqrt_one_minus_alphas_cumprod"], t, x_start) * noise)


def training_losses(sd, tab, x_start, mask, t, given_objs, given_cats, target_cat, text_features, noise, fps_starts, lambda_cat=0.1,
                    train=None, pool_inject=None, taps=None):
    """Returns {"loss", "mse", "cat_loss"} (scalars).  ``sd`` may hold tensors with requires_grad; ``train`` = cpu_ref.TrainMode or
    None (eval-mode BatchNorm, no dropout)."""
    x_t = q_sample(tab, x_start, t, noise).to(x_start.dtype)
    out_cat, out, _ = cpu_ref.forward(sd, x_t, mask, t, given_objs, given_cats, text_features, fps_starts, train=train,
                                      pool_inject=pool_inject, taps=taps)
    cat_loss = F.cross_entropy(out_cat.squeeze(1), torch.argmax(target_cat, dim=1)) * lambda_cat
    mse = chamfer_distance(out.float(), x_start.float())
    return {"loss": mse + cat_loss, "mse": mse, "cat_loss": cat_loss}


def trainable_names(sd):
    """state_dict keys that are nn.Parameters of the reference model (everything but buffers)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
                                  or k.endswith(".pe"))]


def losses_and_grads(sd, tab, *args, dtype=None, **kw):
    """training_losses + backward.  Returns (losses: dict of floats, grads: dict name -> tensor; parameters the loss does not
    depend on -- the value / output projections of the two attention modules' dead branches, Q6 -- map to None).
    ``dtype=torch.float64`` evaluates the same graph in double precision with the discrete geometry decisions still taken in fp32
    (cpu_ref.set_abstraction / feature_propagation): the mathematically exact gradient of the reference's objective, against which
    the conditioning of an fp32 evaluation can be measured (in training mode torch's own fp32 result is 6e-4 .. 1e-3 away from it)."""
    cast = (lambda v: v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) if dtype is not None else (lambda v: v)
    names = set(trainable_names(sd))
    leaves = {k: (cast(v).detach().clone().requires_grad_(True) if k in names else cast(v)) for k, v in sd.items()}
    args = [cast(a) for a in args]
    if kw.get("train") is not None and kw["train"].dropout_mask is not None:
        kw["train"].dropout_mask = cast(kw["train"].dropout_mask)
    with torch.enable_grad():
        terms = training_losses(leaves, tab, *args, **kw)
        terms["loss"].backward()
    return {k: float(v) for k, v in terms.items()}, {k: leaves[k].grad for k in trainable_names(sd)}
