"""ctypes binding of liblsdm_hip.so (include/lsdm_hip.h).

The product path has no fallback: if the shared library is missing or no HIP device is usable the
constructors below raise ``LsdmError`` -- nothing here ever routes to a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblsdm_hip.so")

MODE_FORWARD, MODE_P_SAMPLE, MODE_DDIM = 0, 1, 2
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}


class LsdmError(RuntimeError):
    pass


_lib = None

_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_SIGNATURES = {
    "lsdm_create": (_i, [C.POINTER(_vp), _i]),
    "lsdm_destroy": (_i, [_vp]),
    "lsdm_last_error": (C.c_char_p, [_vp]),
    "lsdm_set_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    "lsdm_set_precision": (_i, [_vp, _i]),
    "lsdm_get_precision": (_i, [_vp]),
    "lsdm_finalize_params": (_i, [_vp, _i]),
    "lsdm_set_schedule": (_i, [_vp, _i] + [_vp] * 7),
    "lsdm_reserve": (_i, [_vp, _i]),
    "lsdm_step": (_i, [_vp, _i, _f, _i] + [_vp] * 12 + [_vp]),
    "lsdm_sample_steps": (_i, [_vp, _i, _f, _i, _i, _i] + [_vp] * 10 + [_vp]),
    "lsdm_sample_steps_hoisted": (_i, [_vp, _i, _f, _i, _i, _i] + [_vp] * 10 + [_vp]),
    "lsdm_sample_steps_invariant": (_i, [_vp, _i, _f, _i, _i, _i] + [_vp] * 10 + [_vp]),
    "lsdm_fps": (_i, [_vp, _vp, _vp, _i] + [_vp] * 8 + [_vp]),
    "lsdm_ball_query": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp]),
    "lsdm_three_nn": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "lsdm_scene_encoder": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "lsdm_human_backbone": (_i, [_vp, _vp, _i, _vp, _vp]),
    "lsdm_nn_distance": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "lsdm_workspace_tensor": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "lsdm_profile_begin": (_i, [_vp]),
    "lsdm_profile_end": (_i, [_vp, _i, _vp, _vp, _vp]),
    "lsdm_profile_executed": (_i, [_vp, _i, _vp]),
    "lsdm_profile_kernel_name": (C.c_char_p, [_i]),
    "lsdm_train_create": (_i, [C.POINTER(_vp), _i, _i]),
    "lsdm_train_destroy": (_i, [_vp]),
    "lsdm_train_last_error": (C.c_char_p, [_vp]),
    "lsdm_train_bind": (_i, [_vp, C.c_char_p, _vp, _vp, C.c_int64]),
    "lsdm_train_read": (_i, [_vp, C.c_char_p, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "lsdm_train_step": (_i, [_vp, _i, _i] + [_vp] * 12 + [_f, _f] + [_vp] * 4),
}
EXPORTS = tuple(_SIGNATURES)


def load():
    """Loads liblsdm_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C lsdm_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64.so.7; it must be the HIP runtime of the process (device pointers and
    # streams are shared with torch), so torch is imported before the library resolves its DT_NEEDED entries.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LsdmError("liblsdm_hip.so not found at %s -- build it with `make -C lsdm_amd/csrc` "
                        "(the HIP library is mandatory; there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """Owns one ``lsdm_ctx`` (one per process and device)."""

    def __init__(self, device_index=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.lsdm_create(C.byref(h), int(device_index))
        if rc != 0:
            raise LsdmError("lsdm_create(device=%d) failed: %s" % (device_index, self.lib.lsdm_last_error(None).decode()))
        self.handle = h
        self.device_index = device_index

    def check(self, rc, what):
        if rc != 0:
            raise LsdmError("%s failed (%d): %s" % (what, rc, self.lib.lsdm_last_error(self.handle).decode()))

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self.handle, *args), name)

    def set_param(self, name, array):
        """array: contiguous fp32 numpy array on the host."""
        shape = (C.c_int64 * max(array.ndim, 1))(*array.shape)
        self.call("lsdm_set_param", name.encode(), array.ctypes.data_as(_vp), shape, array.ndim)

    def workspace_tensor(self, name):
        ptr, n = _vp(), C.c_int64()
        self.call("lsdm_workspace_tensor", name.encode(), C.byref(ptr), C.byref(n))
        return ptr.value, n.value

    def profile_begin(self):
        self.call("lsdm_profile_begin")

    def profile_end(self):
        """-> list of dict(name, total_ms, launches, macs_per_unit, executed_macs) per kernel of the hot path (executed_macs: total
        over the window, see include/lsdm_hip.h)."""
        cap = 64
        ms, n, macs = (C.c_double * cap)(), (C.c_int64 * cap)(), (C.c_double * cap)()
        k = self.lib.lsdm_profile_end(self.handle, cap, ms, n, macs)
        if k < 0:
            self.check(k, "lsdm_profile_end")
        ex = (C.c_double * cap)()
        self.check(min(0, self.lib.lsdm_profile_executed(self.handle, cap, ex)), "lsdm_profile_executed")
        return [dict(name=self.lib.lsdm_profile_kernel_name(i).decode(), total_ms=ms[i], launches=n[i], macs_per_unit=macs[i],
                     executed_macs=ex[i]) for i in range(k)]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lsdm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
