"""``SceneDiffusionModel`` -- drop-in for the reference's ``model.sdm.SceneDiffusionModel`` on MI355X.

Same constructor signature, ``forward`` signature, side-effect attributes (``saved_cat``,
``saved_guiding_points``) and ``state_dict`` key names/shapes as the reference (model/sdm.py:19-22,
131, 158, 217), so ``model.load_state_dict(ckpt['model_state_dict'])``
(run/test_sdm.py:124) works unchanged.  The module owns parameters only; every computation of
``forward`` runs in hand-written HIP kernels behind the C ABI of ``liblsdm_hip.so``
(include/lsdm_hip.h).  There is no eager/CPU fallback: calling ``forward`` without the library or
with host tensors raises.

Extensions (keyword-only, defaults preserve reference behaviour):
  * ``y`` may be a ``(B, clip_dim)`` tensor of precomputed text features -- the boundary of this
    path (the reference re-runs its text encoder every step, model/sdm.py:146-149); a list of strings is encoded
    through ``self.text_encoder`` (by ``text_encoder_type``: the ``clip`` package, or ``transformers``' BERT pooler
    output with ``clip_dim=768``, when importable).
  * ``precision=`` ("f32" | "bf16x3" | "bf16", also ``model.set_precision``; default from ``LSDM_PRECISION`` or
    "f32") selects the arithmetic of the dense 1x1-conv / Linear stacks: exact fp32 MFMA, fp32 operands split into
    bf16 hi+lo with 3 bf16 MFMAs per product (fp32-class results, same <= 1e-3 parity bar), or plain bf16 products.
    Geometry, normalisations, softmaxes and the sampler arithmetic are fp32 in every mode.
  * ``forward(..., fps_starts=...)`` injects the four FPS start draws; by default they are drawn
    from the global CPU generator exactly where the reference draws them
    (model/pcd_backbone/pointnet2_utils.py:72 -- ``torch.randint`` without a device argument).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..state_spec import state_dict_spec

NUM_SLOTS = 9
FPS_RANGES = (1024, 1024, 256, 64)  # N of the cloud each SA level samples from


def _attach(root, dotted, tensor, kind):
    """Registers ``tensor`` under a dotted state_dict name, creating plain container modules on the way."""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if kind in ("pe", "bn_mean", "bn_var", "bn_count"):
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class SceneDiffusionModel(nn.Module):
    def __init__(self, seg_len=256, modality='text', clip_version='ViT-B/32', clip_dim=768, dropout=0.1, n_layer=6,
                 n_head=8, f_vert=64, dim_ff=512, cat_emb=32, mesh_ds_dir="data/mesh_ds", posa_path=None,
                 latent_dim=128, cond_mask_prob=1.0, device=0, vert_dims=655, obj_cat=8, data_rep='rot6d', njoints=251,
                 use_cuda=True, pcd_points=1024, pcd_dim=128, xyz_dim=3, max_cats=13, translation_params=12,
                 pcd_backbone_type="PNT2", human_backbone_type="POSA", text_encoder_type="CLIP", **kwargs):
        super().__init__()
        self.precision = kwargs.pop("precision", None) or os.environ.get("LSDM_PRECISION", "f32")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        if pcd_backbone_type != "PNT2" or human_backbone_type != "POSA":
            raise NotImplementedError("only the default PNT2 / POSA backbones are on the accelerated path "
                                      "(DGCNN / P2R are ablation options of the reference, SURVEY.md §2)")
        if (latent_dim, cat_emb, pcd_points, xyz_dim, translation_params, n_head) != (128, 32, 1024, 3, 12, 8) or pcd_dim != 3 \
                or clip_dim % 2 or not 2 <= clip_dim <= 1024:
            raise NotImplementedError("kernels are specialised for the proxd/humanise defaults of util/model_util.py:26-72 "
                                      "(latent_dim=128, pcd_dim=3, n_head=8; clip_dim even and <= 1024: 512 for CLIP, 768 for BERT)")
        if text_encoder_type not in ("CLIP", "BERT"):
            raise ValueError("text_encoder_type must be 'CLIP' or 'BERT' (model/sdm.py:146-149)")
        self.seg_len, self.modality, self.clip_version, self.clip_dim = seg_len, modality, clip_version, clip_dim
        self.latent_dim, self.pcd_dim, self.pcd_points, self.xyz_dim = latent_dim, pcd_dim, pcd_points, xyz_dim
        self.max_cats, self.n_head, self.translation_params = max_cats, n_head, translation_params
        self.text_encoder_type = text_encoder_type
        self.device = "cuda:{}".format(device) if use_cuda else "cpu"
        self._device_index = int(device)
        self.saved_cat = None
        self.saved_guiding_points = None
        self.text_encoder = None  # callable(list[str]) -> (B,512) tensor
        gen = torch.Generator().manual_seed(0)
        for name, (shape, kind) in state_dict_spec(max_cats=max_cats, clip_dim=clip_dim).items():
            if kind == "pe":
                from ..synthetic import sinusoid_table
                t = sinusoid_table(shape[0], shape[2])
            elif kind == "bn_count":
                t = torch.tensor(0, dtype=torch.long)
            elif kind in ("norm_w", "bn_var"):
                t = torch.ones(shape)
            elif kind in ("norm_b", "bn_mean", "bias"):
                t = torch.zeros(shape)
            else:
                fan_in = int(np.prod(shape[1:]))
                t = (torch.rand(shape, generator=gen) * 2 - 1) / float(np.sqrt(fan_in))
            _attach(self, name, t, kind)
        self._ctx = None
        self._dirty = True
        self._schedule_key = None
        self._cap = 0
        self.requires_grad_(False)
        if use_cuda and torch.cuda.is_available():
            self.to(self.device)

    # ------------------------------------------------------------------ parameter plumbing
    def load_state_dict(self, state_dict, strict=True, **kw):
        # real checkpoints also carry the CLIP weights (clip_model.*), which live outside this path
        sd = {k: v for k, v in state_dict.items() if not k.startswith("clip_model.")}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._dirty = True
        return out

    def set_precision(self, precision):
        """Arithmetic of the dense stacks: "f32", "bf16x3" or "bf16" (see the module docstring)."""
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        if precision != self.precision:
            self.precision = precision
            self._dirty = True
            self._cap = 0
        return self

    def refresh(self):
        """Re-uploads the parameters after they were modified in place."""
        self._dirty = True

    def _context(self):
        if self._ctx is None:
            if not torch.cuda.is_available():
                raise _lib.LsdmError("SceneDiffusionModel.forward needs a HIP device: the hot path runs only in "
                                     "liblsdm_hip.so (no CPU fallback)")
            self._ctx = _lib.Context(self._device_index)
        if self._dirty:
            if self._ctx.lib.lsdm_get_precision(self._ctx.handle) != _lib.PRECISIONS[self.precision]:
                self._ctx.call("lsdm_set_precision", _lib.PRECISIONS[self.precision])   # drops the workspace
                self._cap = 0
            for name, tensor in self.state_dict().items():
                if name.endswith("num_batches_tracked") or name.startswith("embed_timestep.sequence_pos_encoder"):
                    continue
                self._ctx.set_param(name, np.ascontiguousarray(tensor.detach().cpu().float().numpy()))
            self._ctx.call("lsdm_finalize_params", self.max_cats)
            self._dirty = False
        return self._ctx

    def _reserve(self, batch):
        if batch > self._cap:
            self._context().call("lsdm_reserve", batch)
            self._cap = batch

    def _set_schedule(self, key, tables):
        """tables: 7 numpy arrays (see lsdm_set_schedule); ``key`` identifies their CONTENT (a digest, never ``id()``:
        two diffusion objects with the same length and different betas must not share an upload).  The library keeps the
        tables in their own allocations, so re-finalising the parameters (load_state_dict / set_precision) leaves them
        valid; a fresh context has none."""
        ctx = self._context()
        if self._schedule_key != (id(ctx), key):
            arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in tables]
            ctx.call("lsdm_set_schedule", len(arrs[0]), *[a.ctypes.data_as(_lib._vp) for a in arrs])
            self._schedule_key = (id(ctx), key)
            self._schedule_len = len(arrs[0])

    # ------------------------------------------------------------------ text boundary
    def _encode_text(self, y, batch, device):
        if isinstance(y, torch.Tensor):
            feats = y
        else:
            if self.text_encoder is None:       # model/sdm.py:146-149: the encoder is chosen by text_encoder_type
                self.text_encoder = (_default_clip_encoder(self.clip_version, device) if self.text_encoder_type == "CLIP"
                                     else _default_bert_encoder(device))
            feats = self.text_encoder(list(y))
        feats = feats.to(device=device, dtype=torch.float32).contiguous()
        if feats.shape != (batch, self.clip_dim):
            raise ValueError("text features must be (B, %d) = (B, clip_dim), got %s" % (self.clip_dim, tuple(feats.shape)))
        return feats

    # ------------------------------------------------------------------ the hot path
    def _check_device(self, x):
        if not x.is_cuda:
            raise _lib.LsdmError("SceneDiffusionModel runs on the HIP device only; got a %s tensor" % x.device)
        if x.device.index != self._device_index:
            raise _lib.LsdmError("tensor on cuda:%d but the model's context is on cuda:%d (construct the model with device=%d)"
                                 % (x.device.index, self._device_index, x.device.index))

    @staticmethod
    def _check_starts(starts, n_clouds):
        """(…,4,n_clouds) start indices: level l draws from [0, FPS_RANGES[l]) (pointnet2_utils.py:72)."""
        if starts.shape[-2:] != (4, n_clouds):
            raise ValueError("fps_starts must be 4 vectors of B*9 = %d indices, got %s" % (n_clouds, tuple(starts.shape)))
        lo = starts.amin(dim=-1).reshape(-1, 4).amin(dim=0).tolist()
        hi = starts.amax(dim=-1).reshape(-1, 4).amax(dim=0).tolist()
        for l, n in enumerate(FPS_RANGES):
            if lo[l] < 0 or hi[l] >= n:
                raise ValueError("fps_starts of level %d must lie in [0, %d), got [%d, %d]" % (l + 1, n, lo[l], hi[l]))

    def _run(self, mode, eta, x, mask, timesteps, given_objs, given_cats, y, fps_starts=None, noise_fn=None, trusted_starts=False):
        self._check_device(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("x must be a contiguous float32 tensor (it is updated in place)")
        B, O, N, _ = given_objs.shape
        if (O, N) != (NUM_SLOTS, self.pcd_points) or x.shape != (B, N, 3):
            raise ValueError("expected given_objs (B,9,1024,3) and x (B,1024,3)")
        dev = x.device
        ctx = self._context()
        self._reserve(B)
        f32 = lambda v: v.to(device=dev, dtype=torch.float32).contiguous()
        mask, given_objs, given_cats = f32(mask), f32(given_objs), f32(given_cats)
        if given_cats.shape != (B, O, self.max_cats):
            raise ValueError("given_cats must be (B,9,%d)" % self.max_cats)
        t = timesteps.to(device=dev, dtype=torch.long).contiguous()
        if t.shape != (B,):
            raise ValueError("timesteps must be (B,)")
        # the kernels index the positional table (5000 rows) and, in the sampler modes, the schedule tables by t.  Host tensors are
        # range-checked here; DEVICE tensors are not read back (that would stall the stream once per step): LSDM_CHECK_TIMESTEPS=1
        # opts into the synchronising check, otherwise an out-of-range device index is clamped into the table.
        t_max = 5000 if mode == _lib.MODE_FORWARD else getattr(self, "_schedule_len", 0)
        if not timesteps.is_cuda or os.environ.get("LSDM_CHECK_TIMESTEPS", "0") == "1":
            t_lo, t_hi = (int(v) for v in torch.aminmax(timesteps))
            if t_lo < 0 or t_hi >= t_max:
                raise ValueError("timesteps must lie in [0, %d), got [%d, %d]" % (t_max, t_lo, t_hi))
        else:
            # checked WITHOUT a stall: the verdict of this call travels to a pinned host byte by an asynchronous copy and is looked at by the
            # next call (or by check_timesteps(), which the sampling loops call when they finish) -- an out-of-range index raises one call
            # late instead of never; the kernels meanwhile see the index clamped into the table, so nothing reads out of bounds
            self._queue_timestep_check(t, t_max)
            t = t.clamp(0, t_max - 1)
        text = self._encode_text(y, B, dev)
        if fps_starts is None:
            # reference draw order and generator: CPU global RNG, sa1..sa4
            fps_starts = [torch.randint(0, n, (B * O,), dtype=torch.long) for n in FPS_RANGES]
        elif not trusted_starts:      # (device-generated starts are in range by construction; checking them would read them back)
            self._check_starts(torch.stack([s.reshape(-1) for s in fps_starts]).long(), B * O)
        starts = torch.stack([s.reshape(-1) for s in fps_starts]).to(device=dev, dtype=torch.long).contiguous()
        noise = noise_fn() if noise_fn is not None else None
        out_cat = torch.empty(B, 1, self.max_cats, device=dev)
        out_x0 = torch.empty_like(x)
        out_guid = torch.empty_like(x)
        out_sample = torch.empty_like(x) if mode != _lib.MODE_FORWARD else None
        p = lambda v: None if v is None else v.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        ctx.call("lsdm_step", mode, float(eta), B, p(x), p(mask), p(t), p(given_objs), p(given_cats), p(text), p(starts),
                 p(noise), p(out_cat), p(out_x0), p(out_guid), p(out_sample), stream)
        self.saved_cat = out_cat
        self.saved_guiding_points = out_guid
        return out_cat, out_x0, out_sample

    def _queue_timestep_check(self, t, t_max):
        self.check_timesteps(block=False)
        ring = getattr(self, "_t_ring", None)
        if ring is None:
            ring = self._t_ring = [torch.zeros(1, dtype=torch.uint8).pin_memory() for _ in range(8)]
            self._t_pending = []
        if len(self._t_pending) >= len(ring):          # every slot still in flight: wait for all of them (never in practice)
            self.check_timesteps(block=True)
        # a slot that NO pending check refers to (checks of calls on different streams can retire out of order: "the next in the ring"
        # could still be in flight and its verdict would be overwritten)
        busy = {id(h) for _, h, _ in self._t_pending}
        host = next(h for h in ring if id(h) not in busy)
        host.copy_(((t < 0) | (t >= t_max)).any().to(torch.uint8).reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        self._t_pending.append((ev, host, t_max))

    def check_timesteps(self, block=True):
        """Resolves the deferred range checks of device-resident timesteps (see _run): raises ValueError if a call since the last check was
        handed an index outside its table.  ``block=False`` looks only at the checks whose copy has already landed."""
        pending, keep, bad = getattr(self, "_t_pending", []), [], None
        for ev, host, t_max in pending:
            if block:
                ev.synchronize()
            if block or ev.query():
                if int(host[0]) and bad is None:
                    bad = t_max
            else:
                keep.append((ev, host, t_max))
        self._t_pending = keep
        if bad is not None:
            raise ValueError("timesteps must lie in [0, %d): an earlier call received a device-resident index outside that range "
                             "(its kernels saw the index clamped)" % bad)

    def _run_steps(self, mode, eta, x, mask, t_first, K, given_objs, given_cats, text, starts, noise, hoisted=False, invariant=False):
        """K consecutive sampler steps t = t_first .. t_first-K+1 in ONE library call (lsdm_sample_steps; lsdm_sample_steps_invariant:
        same results with the loop-invariant level-1 encoder work evaluated once per call; or
        lsdm_sample_steps_hoisted with one pinned (4,B*9) start tensor).  ``x`` (device, contiguous) holds the result;
        ``starts`` (K,4,B*9) int64 and ``noise`` (K,B,1024,3) are the pre-drawn RNG streams of the K steps."""
        self._check_device(x)
        B = x.shape[0]
        dev = x.device
        ctx = self._context()
        self._reserve(B)
        f32 = lambda v: v.to(device=dev, dtype=torch.float32).contiguous()
        mask, given_objs, given_cats = f32(mask), f32(given_objs), f32(given_cats)
        if given_objs.shape != (B, NUM_SLOTS, self.pcd_points, 3) or x.shape != (B, self.pcd_points, 3) \
                or given_cats.shape != (B, NUM_SLOTS, self.max_cats):
            raise ValueError("expected given_objs (B,9,1024,3), x (B,1024,3), given_cats (B,9,%d)" % self.max_cats)
        starts = starts.to(device=dev, dtype=torch.long).contiguous()
        noise = f32(noise)
        assert noise.shape == (K, B, self.pcd_points, 3) and starts.shape == ((4, B * NUM_SLOTS) if hoisted else (K, 4, B * NUM_SLOTS))
        out_cat = torch.empty(B, 1, self.max_cats, device=dev)
        out_x0, out_guid = torch.empty_like(x), torch.empty_like(x)
        p = lambda v: v.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        entry = "lsdm_sample_steps_hoisted" if hoisted else ("lsdm_sample_steps_invariant" if invariant else "lsdm_sample_steps")
        ctx.call(entry, mode, float(eta), B, int(t_first), int(K), p(x),
                 p(mask), p(given_objs), p(given_cats), p(text), p(starts), p(noise), p(out_cat), p(out_x0), p(out_guid), stream)
        self.saved_cat, self.saved_guiding_points = out_cat, out_guid
        return out_x0

    def forward(self, x, mask, timesteps, given_objs, given_cats, y=None, force_mask=False, *, fps_starts=None):
        """x (B,1024,3) is mutated in place (x += guidance, model/sdm.py:204).  Returns (out_cat, x_out).
        ``timesteps`` on the HOST are range-checked here ([0, 5000): the positional table).  DEVICE-resident timesteps are not read back (that
        would stall the stream every step): the kernels see the index clamped into the table, the verdict travels to the host asynchronously and
        an out-of-range index raises ValueError from the NEXT call or from ``check_timesteps()`` (which the sampling loops call when they end,
        also when abandoned early) -- late, never silently.  ``LSDM_CHECK_TIMESTEPS=1`` checks synchronously."""
        out_cat, out_x0, _ = self._run(_lib.MODE_FORWARD, 0.0, x, mask, timesteps, given_objs, given_cats, y, fps_starts)
        return out_cat, out_x0


def _default_clip_encoder(clip_version, device):
    """CLIP text tower as the reference uses it (model/sdm.py:245-277): tokenize with context 22, zero-pad to 77."""
    try:
        import clip  # third party: git+https://github.com/openai/CLIP.git (reference README.md:27)
    except ImportError as e:
        raise RuntimeError("the `clip` package is not installed: pass precomputed (B,512) text features as `y` "
                           "or set model.text_encoder") from e
    clip_model, _ = clip.load(clip_version, device=device, jit=False)
    clip_model.eval()

    def encode(texts):
        tok = clip.tokenize(texts, context_length=22, truncate=True).to(device)
        tok = torch.cat([tok, torch.zeros(tok.shape[0], 77 - 22, dtype=tok.dtype, device=tok.device)], dim=1)
        with torch.no_grad():
            return clip_model.encode_text(tok).float()

    return encode


def _default_bert_encoder(device):
    """The reference's other text tower (model/sdm.py:261-264, 279-285): ``bert-base-uncased`` tokenizer with padding, the model's
    ``pooler_output`` (B,768) -- so the model must be built with ``clip_dim=768``.  Third party (``transformers`` + downloaded weights):
    parity unpinned, like CLIP; everything downstream of the (B,768) feature is pinned (golden G13)."""
    try:
        from transformers import BertModel, BertTokenizer
        tokenizer = BertTokenizer.from_pretrained("bert-base-uncased")
        bert = BertModel.from_pretrained("bert-base-uncased").to(device)
    except Exception as e:      # no package, or no weights reachable offline
        raise RuntimeError("the BERT text encoder is not available (%s): pass precomputed (B,768) text features as `y` or set "
                           "model.text_encoder" % type(e).__name__) from e
    bert.eval()
    for p in bert.parameters():
        p.requires_grad = False

    def encode(texts):
        enc = tokenizer(list(texts), padding=True, return_tensors='pt').to(device)
        with torch.no_grad():
            return bert(**enc).pooler_output.float()

    return encode
