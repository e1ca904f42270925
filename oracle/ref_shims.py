"""Import harness for the read-only reference tree (/root/reference).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_goldens.py *in the build container* to run the real
reference (andvg3/LSDM) on CPU and record golden vectors.  It never runs on the GPU box (the
reference does not exist there) and nothing in the product package imports it.

The reference needs third-party modules that are absent here: clip, pytorch3d,
trimesh, openmesh, open3d, tensorboard.  They are replaced by minimal stand-ins injected into
``sys.modules`` -- these are *Python-level import stubs for modules off the hot path*, not
re-implementations of the reference:

* ``clip``      -> ``encode_text`` returns features injected by the caller (the drop-in boundary is the
                   (B,512) text feature, SURVEY.md §8c "Third-party arithmetic").
* ``trimesh``   -> tiny OBJ ``v``/``f`` parser (only vertex count 655 matters).
* ``openmesh``  -> ``TriMesh`` placeholder; ``posa_utils.extract_spirals`` is patched to the identity
                   spiral ``[[i] for i in range(nv)]``, exact for ``seq_length == 1``
                   (posa/posa_utils.py:142-170: spiral starts with ``[vh0.idx()]`` and is cut to 1).
* ``pytorch3d.loss.chamfer_distance`` / ``open3d`` -> import-only placeholders.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

REF_ROOT = os.environ.get("LSDM_REFERENCE_ROOT", "/root/reference")


class _InjectedClipModel:
    """Stands in for the CLIP model object: encode_text returns whatever was injected."""

    def __init__(self):
        self.features = None

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def encode_text(self, tokens):
        assert self.features is not None, "inject text features first"
        assert self.features.shape[0] == tokens.shape[0]
        return self.features


def _install_stubs():
    import numpy as np
    import torch

    if "clip" in sys.modules and getattr(sys.modules["clip"], "_lsdm_stub", False):
        return
    clip = types.ModuleType("clip")
    clip._lsdm_stub = True
    clip._model = _InjectedClipModel()

    def load(name, device=None, jit=False):
        return clip._model, None

    def tokenize(texts, context_length=77, truncate=False):
        return torch.zeros(len(texts), context_length, dtype=torch.long)

    clip.load = load
    clip.tokenize = tokenize
    clip.model = types.ModuleType("clip.model")
    clip.model.convert_weights = lambda m: None
    sys.modules["clip"] = clip
    sys.modules["clip.model"] = clip.model

    p3d = types.ModuleType("pytorch3d")
    p3d_loss = types.ModuleType("pytorch3d.loss")

    def chamfer_distance(*a, **k):
        raise RuntimeError("pytorch3d stub: not on the sampling path")

    p3d_loss.chamfer_distance = chamfer_distance
    p3d.loss = p3d_loss
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.loss"] = p3d_loss

    trimesh = types.ModuleType("trimesh")

    class _Mesh:
        def __init__(self, v, f):
            self.vertices = v
            self.faces = f

    def tm_load(path, process=False):
        vs, fs = [], []
        with open(path) as fh:
            for line in fh:
                if line.startswith("v "):
                    vs.append([float(t) for t in line.split()[1:4]])
                elif line.startswith("f "):
                    fs.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
        return _Mesh(np.asarray(vs, dtype=np.float64), np.asarray(fs, dtype=np.int64))

    trimesh.load = tm_load
    sys.modules["trimesh"] = trimesh

    om = types.ModuleType("openmesh")

    class TriMesh:
        def __init__(self, v, f):
            self.v, self.f = v, f

    om.TriMesh = TriMesh
    om.VertexHandle = lambda i: i
    sys.modules["openmesh"] = om

    import transformers  # noqa: F401  (real package; import before any placeholder is registered)
    from transformers import BertTokenizer, BertModel  # noqa: F401
    for name in ("open3d", "smplx", "blobfile"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def import_reference():
    """Returns (SceneDiffusionModel, create_gaussian_diffusion, get_default_diffusion, defaults, clip_stub)."""
    _install_stubs()
    # The repo root carries drop-in alias packages with the reference's own names (model/, diffusion/, util/,
    # run/).  They are regular packages and would shadow the reference's namespace packages wherever they sit
    # on sys.path, so the repo root leaves sys.path while the reference is imported and returns at the END.
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in [m for m in sys.modules if m.split(".")[0] in ("model", "diffusion", "util", "run", "posa")]:
        del sys.modules[name]
    sys.path[:] = [p for p in sys.path if p not in ("", ".", repo_root, REF_ROOT)]
    sys.path.insert(0, REF_ROOT)
    os.chdir(REF_ROOT)  # posa_models.Decoder reads ./mesh_ds relative to the CWD (posa_models.py:293)
    import posa.posa_utils as posa_utils
    import posa.posa_models as posa_models

    def extract_spirals_identity(mesh, seq_length, dilation=1):
        assert seq_length == 1 and dilation == 1
        return [[i] for i in range(len(mesh.v))]

    posa_utils.extract_spirals = extract_spirals_identity
    _orig = posa_models.load_ds_us_param

    def load_cpu(ds_us_dir, level, seq_length, use_cuda=True):
        return _orig(ds_us_dir, level, seq_length, use_cuda=False)  # :229 hard-codes cuda

    posa_models.load_ds_us_param = load_cpu
    from model.sdm import SceneDiffusionModel
    import util.model_util as mu
    import diffusion.gaussian_diffusion  # noqa: F401  (cached in sys.modules for the golden generator)
    import diffusion.respace  # noqa: F401
    import model.pcd_backbone.pointnet2_utils  # noqa: F401
    sys.path.append(repo_root)
    return SceneDiffusionModel, mu, sys.modules["clip"]
