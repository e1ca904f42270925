# model/sdm_hip.py  (new file in the reference tree; INTEGRATION.md path B shows this file verbatim)
"""ctypes binding of liblsdm_hip.so underneath the reference's own ``SceneDiffusionModel.forward``."""
import ctypes as C
import os

import numpy as np
import torch

_lib = C.CDLL(os.environ.get("LSDM_HIP_LIB", "liblsdm_hip.so"))
_vp = C.c_void_p
_lib.lsdm_create.argtypes = [C.POINTER(_vp), C.c_int]
_lib.lsdm_destroy.argtypes = [_vp]
_lib.lsdm_set_param.argtypes = [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), C.c_int]
_lib.lsdm_finalize_params.argtypes = [_vp, C.c_int]
_lib.lsdm_reserve.argtypes = [_vp, C.c_int]
_lib.lsdm_step.argtypes = [_vp, C.c_int, C.c_float, C.c_int] + [_vp] * 13
_lib.lsdm_last_error.restype = C.c_char_p
_lib.lsdm_last_error.argtypes = [_vp]


class HipForward:
    """Drop-in for SceneDiffusionModel.forward (model/sdm.py:131): same arguments, same return, x mutated in place."""

    def __init__(self, torch_model, device=0, max_batch=64):
        self.h = _vp()
        self._ok(_lib.lsdm_create(C.byref(self.h), device))
        for name, t in torch_model.state_dict().items():            # run/test_sdm.py:124 keys
            if name.startswith("clip_model.") or name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(t.detach().cpu().float().numpy())
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._ok(_lib.lsdm_set_param(self.h, name.encode(), a.ctypes.data, shape, a.ndim))
        self._ok(_lib.lsdm_finalize_params(self.h, torch_model.max_cats))
        self._ok(_lib.lsdm_reserve(self.h, max_batch))
        self.clip = torch_model._encode_text_clip                   # model/sdm.py:245 stays in Python
        self.C = torch_model.max_cats

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(_lib.lsdm_last_error(self.h).decode())

    def __call__(self, x, mask, timesteps, given_objs, given_cats, y=None, force_mask=False):
        B = x.shape[0]
        text = self.clip(y).float().contiguous()                    # (B,512): the path boundary
        starts = torch.stack([torch.randint(0, n, (B * 9,)) for n in (1024, 1024, 256, 64)]).to(x.device)  # pointnet2_utils.py:72
        out_cat = torch.empty(B, 1, self.C, device=x.device)
        out, guid = torch.empty_like(x), torch.empty_like(x)
        keep = [mask.float().contiguous(), timesteps.long().contiguous(), given_objs.float().contiguous(),
                given_cats.float().contiguous()]                    # alive until the launches were enqueued
        p = lambda t: t.data_ptr()
        self._ok(_lib.lsdm_step(self.h, 0, 0.0, B, p(x), p(keep[0]), p(keep[1]), p(keep[2]), p(keep[3]),
                                p(text), p(starts), None, p(out_cat), p(out), p(guid), None,
                                torch.cuda.current_stream().cuda_stream))
        self.saved_cat, self.saved_guiding_points = out_cat, guid
        return out_cat, out

    def close(self):
        if self.h:
            _lib.lsdm_destroy(self.h)
            self.h = None
