"""Sampler side of the path: ``GaussianDiffusion`` with the reference's method signatures.

Mirrors diffusion/gaussian_diffusion.py of the reference for the functions on the sampling path:
coefficient tables (:166-202), ``p_mean_variance`` (:282-393), ``p_sample`` (:501-561),
``p_sample_loop`` / ``p_sample_loop_progressive`` (:611-759), ``q_sample`` (:240-256) and the DDIM
update (:790-811).  The reference's own ``ddim_sample_loop`` cannot run for LSDM (it does not thread
mask/given_objs/given_cats/y), so the DDIM entry points here take the same conditioning
arguments as ``p_sample_loop`` and compose the working ``p_mean_variance`` with the DDIM arithmetic.
Training losses, PLMS and bits-per-dim utilities are outside this path.

When the model is the HIP ``SceneDiffusionModel`` the network evaluation and the posterior update run
as ONE fused launch sequence (``lsdm_step``); the host only draws the random numbers, in the
reference's order: 4 FPS ``randint`` (inside the model) and then ``randn`` for the step.
"""
import enum
import math
from copy import deepcopy

import numpy as np
import torch as th

from .. import _lib
from ..model.sdm import SceneDiffusionModel


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    """'linear' and 'cosine' schedules (gaussian_diffusion.py:22-66), float64."""
    T = num_diffusion_timesteps
    if schedule_name == "linear":
        scale = scale_betas * 1000 / T
        return np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)
    if schedule_name == "cosine":
        abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / T) / abar(i / T), 0.999) for i in range(T)])
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """f64 table -> gather -> fp32 -> broadcast (gaussian_diffusion.py:1585-1598)."""
    res = th.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


def _unwrap(model):
    return getattr(model, "model", model) if not isinstance(model, SceneDiffusionModel) else model


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False, lambda_rcxyz=0.,
                 lambda_vel=0., lambda_pose=1., lambda_orient=1., lambda_loc=1., data_rep='rot6d', lambda_root_vel=0.,
                 lambda_vel_rcxyz=0., lambda_fc=0., lambda_cat=0.05, host_rng=True):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        self.data_rep = data_rep
        self.lambda_cat = lambda_cat
        # host_rng=True draws x_T, the per-step FPS starts and the per-step noise from the CPU generator in the reference's order
        # (parity with the CPU reference).  False = throughput mode: all of them from the DEVICE generator (the reference on an
        # accelerator draws its noise there too; its FPS starts stay on the host, pointnet2_utils.py:72) -- nothing is drawn or
        # uploaded by the host inside the loop.
        self.host_rng = host_rng
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    # ------------------------------------------------------------------ helpers
    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    def _schedule_key(self):
        """Digest of the seven tables the library gathers from (content, not identity)."""
        if getattr(self, "_sched_digest", None) is None:
            import hashlib
            h = hashlib.sha1()
            for a in self._device_tables():
                h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
            self._sched_digest = h.hexdigest()
        return self._sched_digest

    def _device_tables(self):
        return (self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_log_variance_clipped,
                self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.alphas_cumprod,
                self.alphas_cumprod_prev)

    def _fusable(self, model, clip_denoised, denoised_fn, cond_fn):
        return (isinstance(_unwrap(model), SceneDiffusionModel) and not clip_denoised and denoised_fn is None
                and cond_fn is None and self.model_mean_type == ModelMeanType.START_X
                and self.model_var_type == ModelVarType.FIXED_SMALL and not self.rescale_timesteps)

    def _noise_like(self, x, const_noise=False):
        noise = th.randn(x.shape).to(x.device) if self.host_rng else th.randn_like(x)
        if const_noise:
            noise = noise[[0]].repeat(x.shape[0], *([1] * (x.dim() - 1)))
        return noise.contiguous()

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = self._noise_like(x_start)
        return (_extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        mean = (_extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, _extract_into_tensor(self.posterior_variance, t, x_t.shape),
                _extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape))

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    # ------------------------------------------------------------------ reference API
    def p_mean_variance(self, model, x, mask, t, given_objs, given_cats, y, clip_denoised=True, denoised_fn=None,
                        model_kwargs=None):
        """Unfused form: one network evaluation (HIP) + the posterior arithmetic as tensor ops.
        ``x`` is mutated by the model and the mutated value enters the posterior mean (Q5)."""
        if self.model_mean_type != ModelMeanType.START_X or self.model_var_type != ModelVarType.FIXED_SMALL:
            raise NotImplementedError("LSDM uses START_X / FIXED_SMALL at every call site (util/model_util.py:127-163)")
        B = x.shape[0]
        assert t.shape == (B,)
        _, model_output = model(x, mask, self._scale_timesteps(t), given_objs, given_cats, y)
        pred_xstart = model_output
        if denoised_fn is not None:
            pred_xstart = denoised_fn(pred_xstart)
        if clip_denoised:
            pred_xstart = pred_xstart.clamp(-1, 1)
        mean, var, logvar = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        assert mean.shape == logvar.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": pred_xstart}

    def p_sample(self, model, x, mask, t, given_objs, given_cats, y, clip_denoised=True, denoised_fn=None, cond_fn=None,
                 model_kwargs=None, const_noise=False):
        if self._fusable(model, clip_denoised, denoised_fn, cond_fn):
            return self._fused_step(_lib.MODE_P_SAMPLE, 0.0, model, x, mask, t, given_objs, given_cats, y, const_noise)
        out = self.p_mean_variance(model, x, mask, t, given_objs, given_cats, y, clip_denoised=clip_denoised,
                                   denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = self._noise_like(x, const_noise)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        if cond_fn is not None:
            gradient = cond_fn(x, self._scale_timesteps(t), **(model_kwargs or {}))
            out["mean"] = out["mean"].float() + out["variance"] * gradient.float()
        sample = out["mean"] + nonzero_mask * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, mask, t, given_objs, given_cats, y, clip_denoised=True, denoised_fn=None,
                    cond_fn=None, model_kwargs=None, eta=0.0):
        if self._fusable(model, clip_denoised, denoised_fn, cond_fn):
            return self._fused_step(_lib.MODE_DDIM, eta, model, x, mask, t, given_objs, given_cats, y, False)
        out = self.p_mean_variance(model, x, mask, t, given_objs, given_cats, y, clip_denoised=clip_denoised,
                                   denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        alpha_bar = _extract_into_tensor(self.alphas_cumprod, t, x.shape)
        alpha_bar_prev = _extract_into_tensor(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * th.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * th.sqrt(1 - alpha_bar / alpha_bar_prev)
        noise = self._noise_like(x)
        mean_pred = out["pred_xstart"] * th.sqrt(alpha_bar_prev) + th.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        return {"sample": mean_pred + nonzero_mask * sigma * noise, "pred_xstart": out["pred_xstart"]}

    def _fused_step(self, mode, eta, model, x, mask, t, given_objs, given_cats, y, const_noise):
        inner = _unwrap(model)
        inner._set_schedule(self._schedule_key(), self._device_tables())
        starts = None
        if not self.host_rng and x.is_cuda:       # device-RNG mode: the FPS starts come from the device generator as well
            from ..model.sdm import FPS_RANGES, NUM_SLOTS
            starts = [th.randint(0, n, (x.shape[0] * NUM_SLOTS,), device=x.device, dtype=th.long) for n in FPS_RANGES]
        _, x0, sample = inner._run(mode, eta, x, mask, t, given_objs, given_cats, y, fps_starts=starts, trusted_starts=starts is not None,
                                   noise_fn=lambda: self._noise_like(x, const_noise))
        return {"sample": sample, "pred_xstart": x0}

    def _bulk_loop(self, mode, eta, model, shape, mask, given_objs, given_cats, y, noise, device, skip_timesteps, init_image,
                   const_noise, hoist_conditioning, chunk_steps, hoist_invariants=False):
        """Whole sampling loop with ``chunk_steps`` steps per library call.  The random numbers are drawn in the
        reference's order (x_T; then per step the 4 FPS randints of pointnet2_utils.py:72 and randn_like of
        gaussian_diffusion.py:545), only ahead of time for a chunk.  With ``hoist_conditioning`` the FPS starts are the
        first step's draws for every step and the scene conditioning is evaluated once per chunk."""
        inner = _unwrap(model)
        if device is None:
            device = next(inner.parameters()).device
        B = shape[0]
        if noise is not None:
            img = noise
        else:
            img = th.randn(*shape).to(device) if self.host_rng else th.randn(*shape, device=device)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        t_first = self.num_timesteps - skip_timesteps - 1
        if init_image is not None:
            my_t = th.ones([B], device=device, dtype=th.long) * t_first
            img = self.q_sample(init_image, my_t, img)
        x = img.to(device=device, dtype=th.float32).contiguous().clone()
        inner._set_schedule(self._schedule_key(), self._device_tables())
        text = inner._encode_text(y, B, x.device)      # the same features every step: encoded once
        from ..model.sdm import FPS_RANGES, NUM_SLOTS
        draw_starts = lambda: th.stack([th.randint(0, n, (B * NUM_SLOTS,), dtype=th.long) for n in FPS_RANGES])
        pinned, done, total = None, 0, t_first + 1
        # Host-RNG (parity) mode: the draws of a chunk go straight into one of two page-locked staging buffers and are
        # uploaded with ONE asynchronous copy per chunk, so the host draws chunk k+1 while the GPU runs chunk k (a blocking
        # per-step upload would wait behind the whole previous chunk and leave the GPU idle while the host draws).
        if chunk_steps is None:
            chunk_steps = 56 if (self.host_rng or not x.is_cuda) else 256
        kmax = min(int(chunk_steps), total)
        stage = None
        if self.host_rng and x.is_cuda:
            stage = [dict(noise=th.empty(kmax, *x.shape, dtype=th.float32).pin_memory(),
                          starts=th.empty(kmax, len(FPS_RANGES), B * NUM_SLOTS, dtype=th.long).pin_memory(), ev=None) for _ in range(2)]
        chunk_i = 0
        if hoist_conditioning and not self.host_rng and x.is_cuda:
            # the hoisted entry point evaluates the scene conditioning once per CALL; with nothing to draw or upload on the host the whole
            # chain is one call (the noise of 1000 steps at B = 64 is 0.8 GB of the 288 GB)
            chunk_steps = total
        while done < total:
            k = min(int(chunk_steps), total - done)
            if stage is not None:
                st = stage[chunk_i & 1]
                if st["ev"] is not None:
                    st["ev"].synchronize()       # the upload that last read this staging buffer has finished
                for j in range(k):
                    if hoist_conditioning:
                        if pinned is None:
                            pinned = draw_starts()
                    else:
                        for l, n in enumerate(FPS_RANGES):
                            th.randint(0, n, (B * NUM_SLOTS,), dtype=th.long, out=st["starts"][j, l])
                    th.randn(x.shape, out=st["noise"][j])
                    if const_noise:
                        st["noise"][j] = st["noise"][j][[0]].repeat(B, 1, 1)
                noises_d = st["noise"][:k].to(x.device, non_blocking=True)
                starts_d = (pinned if hoist_conditioning else st["starts"][:k]).to(x.device, non_blocking=True)
                st["ev"] = th.cuda.Event()
                st["ev"].record(th.cuda.current_stream(x.device))
            elif not self.host_rng and x.is_cuda:
                # Device-RNG (throughput) mode: the FPS starts and the noise of the chunk come from the DEVICE generator -- two
                # launches per chunk, no host draw, no upload; deterministic under torch.manual_seed (which seeds the device
                # generator too), statistically the same process, not comparable draw by draw with the CPU reference.
                if hoist_conditioning:
                    if pinned is None:
                        pinned = th.stack([th.randint(0, n, (B * NUM_SLOTS,), device=x.device, dtype=th.long) for n in FPS_RANGES])
                    starts_d = pinned
                else:
                    starts_d = th.stack([th.randint(0, n, (k, B * NUM_SLOTS), device=x.device, dtype=th.long) for n in FPS_RANGES], dim=1)
                noises_d = th.randn(k, *x.shape, device=x.device, dtype=th.float32)
                if const_noise:
                    noises_d = noises_d[:, :1].expand(-1, B, -1, -1).contiguous()
            else:
                starts, noises = [], []
                for _ in range(k):
                    if hoist_conditioning:
                        if pinned is None:
                            pinned = draw_starts()
                    else:
                        starts.append(draw_starts())
                    noises.append(self._noise_like(x, const_noise))
                noises_d, starts_d = th.stack(noises), (pinned if hoist_conditioning else th.stack(starts))
            inner._run_steps(mode, eta, x, mask, t_first - done, k, given_objs, given_cats, text, starts_d, noises_d,
                             hoisted=hoist_conditioning, invariant=hoist_invariants and not hoist_conditioning)
            done += k
            chunk_i += 1
        return x

    def p_sample_loop(self, model, shape, mask, given_objs, given_cats, y, noise=None, clip_denoised=True,
                      denoised_fn=None, cond_fn=None, model_kwargs=None, device=None, progress=False, skip_timesteps=0,
                      init_image=None, randomize_class=False, cond_fn_with_grad=False, dump_steps=None,
                      const_noise=False, *, hoist_conditioning=False, chunk_steps=None, hoist_invariants=True):
        """Reference signature (gaussian_diffusion.py:611-632) + keyword-only extensions: ``hoist_invariants`` (default on: evaluate the
        loop-invariant level-1 encoder work once per library call -- bit-identical results, lsdm_sample_steps_invariant; off = every
        stage recomputed every step, lsdm_sample_steps), ``chunk_steps`` (steps per
        library call on the fused path; with host-side draws (``host_rng``, the parity mode) results do not depend on it -- with DEVICE draws a
        chunk's noise and FPS starts are one ``th.randn(k, ...)`` / ``th.randint(..., (k, ...))`` each, whose Philox stream depends on k, so seeded
        device-RNG samples change with ``chunk_steps`` (and changed when the default went from 56 to 256); default 56 with host-side draws -- the host draws chunk k + 1 while the
        GPU runs chunk k -- and 256 with device draws, where a call's once-per-call work is all a longer chunk saves) and ``hoist_conditioning`` (pin the FPS starts to
        the first step's draws and evaluate the scene conditioning once -- not the stock reference's behaviour)."""
        if (dump_steps is None and not progress and not randomize_class and not cond_fn_with_grad
                and self._fusable(model, clip_denoised, denoised_fn, cond_fn)):
            return self._bulk_loop(_lib.MODE_P_SAMPLE, 0.0, model, shape, mask, given_objs, given_cats, y, noise, device,
                                   skip_timesteps, init_image, const_noise, hoist_conditioning, chunk_steps, hoist_invariants)
        if hoist_conditioning:
            raise NotImplementedError("hoist_conditioning needs the fused path (clip_denoised=False, no denoised_fn/cond_fn/dump_steps)")
        final, dump = None, []
        for i, sample in enumerate(self.p_sample_loop_progressive(
                model, shape, mask, given_objs, given_cats, y, noise=noise, clip_denoised=clip_denoised,
                denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                skip_timesteps=skip_timesteps, init_image=init_image, randomize_class=randomize_class,
                cond_fn_with_grad=cond_fn_with_grad, const_noise=const_noise)):
            if dump_steps is not None and i in dump_steps:
                dump.append(deepcopy(sample["sample"]))
            final = sample
        return dump if dump_steps is not None else final["sample"]

    def p_sample_loop_progressive(self, model, shape, mask, given_objs, given_cats, y, noise=None, clip_denoised=True,
                                  denoised_fn=None, cond_fn=None, model_kwargs=None, device=None, progress=False,
                                  skip_timesteps=0, init_image=None, randomize_class=False, cond_fn_with_grad=False,
                                  const_noise=False, _step="p_sample", _eta=0.0):
        if cond_fn_with_grad or randomize_class:
            raise NotImplementedError("gradient-guided / class-randomised sampling is unused by LSDM")
        if device is None:
            device = next(_unwrap(model).parameters()).device
        assert isinstance(shape, (tuple, list))
        if noise is not None:
            img = noise
        else:
            img = th.randn(*shape).to(device) if self.host_rng else th.randn(*shape, device=device)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(img)
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]
        if init_image is not None:
            my_t = th.ones([shape[0]], device=device, dtype=th.long) * indices[0]
            img = self.q_sample(init_image, my_t, img)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        inner = _unwrap(model)
        failed = False
        try:
            for i in indices:
                t = th.tensor([i] * shape[0], device=device)
                with th.no_grad():
                    img = img.contiguous()
                    if _step == "ddim":
                        out = self.ddim_sample(model, img, mask, t, given_objs, given_cats, y, clip_denoised=clip_denoised,
                                               denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs, eta=_eta)
                    else:
                        out = self.p_sample(model, img, mask, t, given_objs, given_cats, y, clip_denoised=clip_denoised,
                                            denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                            const_noise=const_noise)
                    yield out
                    img = out["sample"]
        except GeneratorExit:
            raise                # close() / a caller's break: the check below still runs, and its ValueError reaches the caller of close()
        except BaseException:
            failed = True        # the loop body is already raising: the deferred check must not replace that exception
            raise
        finally:
            # the deferred range checks of the loop's device-resident timesteps (model/sdm.py::_run) are resolved when the generator ends --
            # also when the caller abandons it early.  A caller that breaks out of the loop should close() the generator (or call
            # model.check_timesteps()) itself: a generator left to the garbage collector can only PRINT the error ("Exception ignored").
            if not failed and hasattr(inner, "check_timesteps"):
                inner.check_timesteps()

    def ddim_sample_loop(self, model, shape, mask, given_objs, given_cats, y, noise=None, clip_denoised=True,
                         denoised_fn=None, cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0,
                         skip_timesteps=0, init_image=None, randomize_class=False, cond_fn_with_grad=False,
                         dump_steps=None, const_noise=False, *, hoist_conditioning=False, chunk_steps=None, hoist_invariants=True):
        """Argument list of the reference's ``ddim_sample_loop`` (gaussian_diffusion.py:957-1000) with the conditioning
        arguments threaded.  ``dump_steps`` behaves as in ``p_sample_loop`` (:661-680); ``const_noise`` is accepted for
        signature compatibility -- the reference's DDIM step has no such argument (``ddim_sample`` :761-775 draws
        ``randn_like``), and with the default eta = 0 the noise does not enter the sample at all."""
        if (dump_steps is None and not progress and not randomize_class and not cond_fn_with_grad
                and self._fusable(model, clip_denoised, denoised_fn, cond_fn)):
            return self._bulk_loop(_lib.MODE_DDIM, eta, model, shape, mask, given_objs, given_cats, y, noise, device,
                                   skip_timesteps, init_image, False, hoist_conditioning, chunk_steps, hoist_invariants)
        if hoist_conditioning:
            raise NotImplementedError("hoist_conditioning needs the fused path (clip_denoised=False, no denoised_fn/cond_fn/dump_steps)")
        final, dump = None, []
        for i, sample in enumerate(self.ddim_sample_loop_progressive(
                model, shape, mask, given_objs, given_cats, y, noise=noise, clip_denoised=clip_denoised,
                denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, progress=progress,
                eta=eta, skip_timesteps=skip_timesteps, init_image=init_image)):
            if dump_steps is not None and i in dump_steps:
                dump.append(deepcopy(sample["sample"]))
            final = sample
        return dump if dump_steps is not None else final["sample"]

    # ------------------------------------------------------------------ training objective
    def training_losses(self, model, cf, mask, t, given_objs, given_cats, target_cat, y=None, noise=None, **draws):
        """The loss terms of one training batch AND their gradient (gaussian_diffusion.py:1256-1342; MSE loss type, START_X, fixed
        variance): x_t = q_sample(cf, t, noise); (out_cat, x0_hat) = model(x_t, ...);
        ``cat_loss`` = lambda_cat * CrossEntropyLoss()(out_cat, argmax(target_cat)) -- out_cat are SOFTMAX outputs used as logits,
        as in the reference (:1296-1300); ``mse`` = pytorch3d chamfer_distance(x0_hat, cf) (:1334; mean squared nearest-neighbour
        distance, both directions summed, batch mean); ``loss`` = mse + cat_loss.
        Everything runs in liblsdm_hip (``lsdm_train_step``, lsdm_amd/csrc/train.hip); ``terms["loss"]`` is a node of torch's graph whose
        backward hands the library's d loss / d theta to the parameters (lsdm_amd/training.py), so ``loss.backward(); optimizer.step()``
        of run/train_sdm.py:76-80 works unchanged.  ``model.training`` selects batch-statistics BatchNorm + the encoder head's dropout
        (the reference's ``model.train()``); ``model.eval()`` gives the running-statistics objective ``validate()`` style."""
        from ..training import training_losses
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE) or self.model_var_type != ModelVarType.FIXED_SMALL \
                or self.model_mean_type != ModelMeanType.START_X:
            raise NotImplementedError("LSDM trains with MSE / START_X / FIXED_SMALL (util/model_util.py:127-163)")
        m = _unwrap(model)
        if not isinstance(m, SceneDiffusionModel):
            raise NotImplementedError("training_losses runs on the HIP SceneDiffusionModel only (no eager fallback)")
        return training_losses(self, m, cf, mask, t, given_objs, given_cats, target_cat, y=y, noise=noise, **draws)

    def ddim_sample_loop_progressive(self, model, shape, mask, given_objs, given_cats, y, noise=None, clip_denoised=True,
                                     denoised_fn=None, cond_fn=None, model_kwargs=None, device=None, progress=False,
                                     eta=0.0, skip_timesteps=0, init_image=None, **unused):
        return self.p_sample_loop_progressive(model, shape, mask, given_objs, given_cats, y, noise=noise,
                                              clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                              model_kwargs=model_kwargs, device=device, progress=progress,
                                              skip_timesteps=skip_timesteps, init_image=init_image, _step="ddim", _eta=eta)
