"""Training objective of the path on the HIP device: ``diffusion.training_losses(model, ...)`` + ``loss.backward()``.

Mirror of the reference's training step (run/train_sdm.py:62-80 around diffusion/gaussian_diffusion.py:1256-1342): the loss terms
AND ``d loss / d theta`` of every parameter are computed by ``lsdm_train_step`` (lsdm_amd/csrc/train.hip: the network layer by layer
with every activation kept, then the reverse sweep -- hand-written HIP kernels, exact fp32).  ``HipObjective`` hands the result to
``torch.autograd`` so that the reference's loop runs unchanged::

    losses = diffusion.training_losses(model, target_obj, mask, t, given_objs, given_cats, target_cat, y=y)
    loss = (losses["loss"] * weights).mean()
    loss.backward()                # routes the library's gradients into p.grad
    optimizer.step()               # torch.optim.AdamW on the device tensors the library reads in place

``model.training`` selects the reference's ``model.train()`` behaviour on this path: batch-statistics BatchNorm in PointNet++ (running
statistics updated in place, ``num_batches_tracked`` incremented) and the Dropout(0.5) of the encoder head; ``model.eval()`` uses the
running statistics and no dropout.  RNG (host generator, reference order on CPU): ``randn_like(x_start)`` when ``noise`` is None, the
four FPS start draws (pointnet2_utils.py:72), then the dropout mask (``empty_like(h).bernoulli_(0.5)``, what ``F.dropout`` draws);
with ``create_gaussian_diffusion(..., host_rng=False)`` all three come from the device generator (throughput mode).

Only ``terms["loss"]`` carries the graph (the library differentiates mse + cat_loss as one scalar); ``mse`` / ``cat_loss`` are values.
"""
import ctypes as C

import torch

from . import _lib

NUM_SLOTS = 9
FPS_RANGES = (1024, 1024, 256, 64)
# parameters that do not enter the loss (dead value / output projections of attn_layer): torch gives them no gradient
DEAD = ("attn_layer.v_proj_weight", "attn_layer.out_proj.weight", "attn_layer.out_proj.bias")


class TrainSession:
    """Owns one ``lsdm_train`` handle and the gradient buffers of a model's parameters."""

    def __init__(self, model):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LsdmError("training needs a HIP device: the objective and its gradients run only in liblsdm_hip.so")
        self.model = model
        h = C.c_void_p()
        rc = self.lib.lsdm_train_create(C.byref(h), int(model._device_index), int(model.max_cats))
        if rc != 0:
            raise _lib.LsdmError("lsdm_train_create failed (%d)" % rc)
        self.handle = h
        self.grads = {}

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.LsdmError("%s failed (%d): %s" % (what, rc, self.lib.lsdm_train_last_error(self.handle).decode()))

    def _bind(self):
        dev = torch.device("cuda", self.model._device_index)
        for name, p in self.model.state_dict(keep_vars=True).items():
            if name.endswith("num_batches_tracked") or name.startswith("embed_timestep.sequence_pos_encoder") or name in DEAD:
                continue
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.LsdmError("parameter %s must be a contiguous float32 tensor on %s (model.to(device))" % (name, dev))
            g = None
            if isinstance(p, torch.nn.Parameter):
                g = self.grads.get(name)
                if g is None or g.shape != p.shape:
                    g = self.grads[name] = torch.zeros_like(p.data)
            self._check(self.lib.lsdm_train_bind(self.handle, name.encode(), p.data_ptr(), g.data_ptr() if g is not None else None, p.numel()),
                        "lsdm_train_bind(%s)" % name)

    def step(self, x_start, qa, qb, noise, t, mask, given_objs, given_cats, target, text, fps_starts, dropout_keep, lambda_cat, training,
             grad_scale=1.0, want_grads=True):
        """-> (losses (3,) = [loss, mse, cat_loss], model_output (B,1024,3), out_cat (B,1,C)); fills ``self.grads`` unless ``want_grads`` is
        False (flags & 2 of lsdm_train_step: the objective only, no reverse sweep -- about a third of the step).  ``step_id`` counts the
        calls that rewrote ``self.grads``: a backward() that runs after a LATER such call would read the other batch's gradients and raises."""
        self._bind()
        if want_grads:
            self.step_id = getattr(self, "step_id", 0) + 1
        dev = x_start.device
        B = x_start.shape[0]
        f32 = lambda v: v.to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda v: v.to(device=dev, dtype=torch.long).contiguous()
        x_start, qa, qb, noise, mask, given_objs, given_cats, text = (f32(v) for v in (x_start, qa, qb, noise, mask, given_objs, given_cats, text))
        t, target, starts = i64(t), i64(target), i64(fps_starts)
        if x_start.shape != (B, 1024, 3) or given_objs.shape != (B, NUM_SLOTS, 1024, 3) or given_cats.shape != (B, NUM_SLOTS, self.model.max_cats) \
                or mask.shape != (B, NUM_SLOTS) or text.shape != (B, self.model.clip_dim) or starts.shape != (4, B * NUM_SLOTS) or t.shape != (B,) or target.shape != (B,):
            raise ValueError("training batch: expected x_start (B,1024,3), given_objs (B,9,1024,3), given_cats (B,9,C), mask (B,9), text (B,clip_dim)")
        keep = None
        if dropout_keep is not None:
            keep = dropout_keep.to(device=dev, dtype=torch.uint8).contiguous()
            if keep.shape != (B * NUM_SLOTS, 128, 1024):
                raise ValueError("dropout keep-mask must be (B*9,128,1024)")
        losses = torch.empty(3, device=dev)
        out = torch.empty(B, 1024, 3, device=dev)
        out_cat = torch.empty(B, 1, self.model.max_cats, device=dev)
        p = lambda v: None if v is None else v.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._check(self.lib.lsdm_train_step(self.handle, B, (1 if training else 0) | (0 if want_grads else 2), p(x_start), p(qa), p(qb), p(noise), p(t), p(mask), p(given_objs),
                                             p(given_cats), p(target), p(text), p(starts), p(keep), float(lambda_cat), float(grad_scale), p(losses),
                                             p(out), p(out_cat), stream), "lsdm_train_step")
        return losses, out, out_cat

    def pool_choices(self, B):
        """Which neighbour the four max-pools of the last step picked per (cloud, centroid, channel): 4 uint8 tensors (B*9, S, C)."""
        out = []
        for l, (S, Cc) in enumerate(((1024, 64), (256, 128), (64, 256), (16, 512))):
            host = torch.empty(B * NUM_SLOTS * S * Cc, dtype=torch.uint8)
            n = C.c_int64()
            self._check(self.lib.lsdm_train_read(self.handle, ("sa%d.arg" % (l + 1)).encode(), host.data_ptr(), host.numel(), C.byref(n)), "lsdm_train_read")
            assert n.value == host.numel()
            out.append(host.view(B * NUM_SLOTS, S, Cc))
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lsdm_train_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipObjective(torch.autograd.Function):
    """loss = mse + cat_loss as a node of torch's graph: forward runs lsdm_train_step (values + gradients), backward scales the
    stored gradients by the incoming one and returns them for the parameters."""

    @staticmethod
    def forward(ctx, session, names, run, *params):
        losses, out, out_cat = run()
        ctx.session, ctx.names, ctx.step_id = session, names, session.step_id
        values = losses.clone()
        ctx.mark_non_differentiable(values, out, out_cat)
        return losses[0].clone(), values, out, out_cat

    @staticmethod
    def backward(ctx, g_loss, _g_all, _g_out, _g_cat):
        if ctx.session.step_id != ctx.step_id:
            raise RuntimeError("HipObjective.backward: the session's gradient buffers were rewritten by a later training_losses() call "
                               "(step %d, this loss is step %d) -- call backward() before the next objective, or evaluate the other one under "
                               "torch.no_grad()" % (ctx.session.step_id, ctx.step_id))
        # one multi-tensor launch per ~30 parameters instead of one elementwise kernel per parameter (157 of them per step); the products
        # are fresh tensors, which torch's AccumulateGrad adopts as p.grad without another copy
        live = [n for n in ctx.names if n not in DEAD]
        scaled = dict(zip(live, torch._foreach_mul([ctx.session.grads[n] for n in live], g_loss)))
        return (None, None, None) + tuple(scaled.get(n) for n in ctx.names)


def training_losses(diffusion, model, x_start, mask, t, given_objs, given_cats, target_cat, y=None, noise=None, *, fps_starts=None,
                    dropout_keep=None):
    """See the module docstring.  Extra keyword arguments inject the draws (tests): ``fps_starts`` (4 tensors of B*9 indices) and
    ``dropout_keep`` ((B*9,128,1024) 0/1, used when ``model.training``)."""
    dev = x_start.device
    model._check_device(x_start)
    B = x_start.shape[0]
    if noise is None:
        noise = diffusion._noise_like(x_start)
    t_host = t.detach().cpu().long()
    qa = torch.from_numpy(diffusion.sqrt_alphas_cumprod)[t_host].float()                 # _extract_into_tensor: gather in f64, cast
    qb = torch.from_numpy(diffusion.sqrt_one_minus_alphas_cumprod)[t_host].float()
    text = model._encode_text(y, B, dev)
    # draws: host generator in the reference's CPU order (parity), or -- diffusion.host_rng False -- the device generator (the 7 M-element
    # dropout mask of a B = 6 batch costs ~25 ms on the host and nothing on the device)
    on_device = not getattr(diffusion, "host_rng", True)
    if fps_starts is None:
        fps_starts = [torch.randint(0, n, (B * NUM_SLOTS,), dtype=torch.long, device=dev if on_device else None) for n in FPS_RANGES]
        starts = torch.stack([s.reshape(-1) for s in fps_starts]).long()
    else:
        starts = torch.stack([s.reshape(-1) for s in fps_starts]).long()
        model._check_starts(starts, B * NUM_SLOTS)
    training = bool(model.training)
    if training and dropout_keep is None:
        dropout_keep = (torch.empty(B * NUM_SLOTS, 128, 1024, device=dev).bernoulli_(0.5).to(torch.uint8) if on_device
                        else torch.empty(B * NUM_SLOTS, 128, 1024).bernoulli_(0.5))
    if not training:
        dropout_keep = None
    session = getattr(model, "_train_session", None)
    if session is None:
        session = model._train_session = TrainSession(model)
    target = torch.argmax(target_cat, dim=1)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    want_grads = bool(named) and torch.is_grad_enabled()
    run = lambda: session.step(x_start, qa, qb, noise, diffusion._scale_timesteps(t), mask, given_objs, given_cats, target, text, starts,
                               dropout_keep, diffusion.lambda_cat, training, want_grads=want_grads)
    if want_grads:
        loss, all_terms, out, out_cat = HipObjective.apply(session, tuple(n for n, _ in named), run, *[p for _, p in named])
    else:
        all_terms, out, out_cat = run()
        loss = all_terms[0]
    if training:
        with torch.no_grad():      # (one multi-tensor launch for the 23 counters)
            counters = [buf for name, buf in model.named_buffers() if name.endswith("num_batches_tracked")]
            if counters:
                torch._foreach_add_(counters, 1)
    if training or want_grads:   # the sampling context re-reads the parameters after an optimizer step / running-statistics update
        model._dirty = True      # (a pure evaluation changes neither)
    model.saved_cat = out_cat
    return {"loss": loss, "mse": all_terms[1], "cat_loss": all_terms[2], "model_output": out}
