"""Deterministic synthetic weights and inputs.

Real LSDM checkpoints are a Google-Drive download (reference README.md:50) and the datasets are not
redistributable, so parity and benchmarks run on seeded synthetic data of the reference's shapes:

* ``make_synthetic_state_dict``: a ``state_dict`` with the reference's key names/shapes
  (lsdm_amd/state_spec.py) whose >=2-D weights are PyTorch-default-like uniform draws scaled by a
  gain, with perturbed norm statistics, so that no stage of the network is numerically dead
  (with default-init weights a <=1e-3 check on the output is vacuous).
* ``make_synthetic_batch``: inputs with the layout of posa/dataset.py:444-474 (slot 0 = human with
  mask 0, k object clouds with mask 1, zero-padded absent slots, one-hot categories) plus the
  (B,512) text feature that stands at the CLIP boundary.

The same generator feeds the real reference (oracle/make_goldens.py), the CPU oracle and the HIP path.
"""
import math

import numpy as np
import torch

from .state_spec import state_dict_spec

DEFAULT_GAIN = 2.5
# Per-layer gain overrides (longest matching prefix wins): keep the guidance term g = O(1) and the
# sigmoid stack of the per-point denoiser out of saturation so the output stays sensitive to x.
GAIN_OVERRIDES = {
    "pcd_backbone.conv2": 0.6,
    "input_process.pose_embedding.0": 2.0,
    "input_process.pose_embedding.2": 6.0,
    "input_process.combination_extraction": 6.0,
    "output_process": 3.0,
}


def sinusoid_table(max_len=5000, d_model=128):
    """The fixed ``pe`` buffer: pe[p,0,2i] = sin(p*w_i), pe[p,0,2i+1] = cos(p*w_i),
    w_i = exp(-2i*ln(10000)/d)  (model/diffusion_utils.py:29-35), evaluated in fp32 with torch."""
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    w = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * w)
    pe[:, 1::2] = torch.cos(pos * w)
    return pe.unsqueeze(1).contiguous()


def make_synthetic_state_dict(seed=0, gain=DEFAULT_GAIN, max_cats=13, **spec_kwargs):
    rng = np.random.Generator(np.random.Philox(seed))
    spec = state_dict_spec(max_cats=max_cats, **spec_kwargs)
    sd = {}
    pe = None
    for name, (shape, kind) in spec.items():
        if kind == "pe":
            if pe is None:
                pe = sinusoid_table(shape[0], shape[2])
            sd[name] = pe  # the reference registers the same buffer under both names
            continue
        if kind == "bn_count":
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        if kind.endswith("_w") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            g = gain
            best = -1
            for prefix, og in GAIN_OVERRIDES.items():
                if name.startswith(prefix) and len(prefix) > best:
                    g, best = og * gain / DEFAULT_GAIN, len(prefix)
            a = g / math.sqrt(fan_in)
            v = rng.uniform(-a, a, size=shape)
        elif kind == "bias":
            v = rng.uniform(-0.1, 0.1, size=shape)
        elif kind == "norm_w":
            v = rng.uniform(0.6, 1.4, size=shape)
        elif kind == "norm_b":
            v = rng.uniform(-0.2, 0.2, size=shape)
        elif kind == "bn_mean":
            v = rng.normal(0.0, 0.2, size=shape)
        elif kind == "bn_var":
            v = rng.uniform(0.5, 1.5, size=shape)
        else:
            raise ValueError(kind)
        sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).contiguous()
    return sd


def make_synthetic_batch(batch, seed=1, max_cats=13, num_slots=9, num_points=1024, all_ones_mask=False,
                         num_objects=None, text_dim=512, surface=False):
    """Returns dict(mask (B,O), given_objs (B,O,N,3), given_cats (B,O,C), text_features (B,text_dim)), fp32.

    ``num_objects``: fixed k for every sample (e.g. 8 = BASELINE config 3's "8192-pt scene cloud");
    default k ~ U{1..8} per sample.  ``surface``: the object points lie ON the six faces of their box (area-weighted) instead of
    in its volume -- PROX objects are surface scans (posa/dataset.py:424-440 loads sampled mesh vertices), sparser within a ball than
    the volume-filling default (a different random stream from the first object on)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    B, O, N, C = batch, num_slots, num_points, max_cats
    objs = np.zeros((B, O, N, 3), dtype=np.float32)
    mask = np.zeros((B, O), dtype=np.float32)
    cats = np.zeros((B, O, C), dtype=np.float32)
    for b in range(B):
        objs[b, 0] = rng.normal(0.0, 1.0, size=(N, 3)) * np.array([0.25, 0.25, 0.45])
        cats[b, 0, 0] = 1.0
        k = int(rng.integers(1, O)) if num_objects is None else int(num_objects)
        for o in range(1, k + 1):
            centre = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0, 1)])
            half = rng.uniform(0.2, 0.8, size=3)
            pts = rng.uniform(-1, 1, size=(N, 3))
            if surface:
                # one coordinate of every point pinned to +-1: the face's axis drawn with probability proportional to the face area
                area = np.array([half[1] * half[2], half[0] * half[2], half[0] * half[1]])
                axis = rng.choice(3, size=N, p=area / area.sum())
                pts[np.arange(N), axis] = rng.choice([-1.0, 1.0], size=N)
            objs[b, o] = centre + pts * half
            mask[b, o] = 1.0
            cats[b, o, int(rng.integers(1, C))] = 1.0
    if all_ones_mask:
        mask[:] = 1.0
    text = rng.normal(0.0, 1.0, size=(B, text_dim)).astype(np.float32)
    return {
        "mask": torch.from_numpy(mask),
        "given_objs": torch.from_numpy(objs),
        "given_cats": torch.from_numpy(cats),
        "text_features": torch.from_numpy(text),
    }
