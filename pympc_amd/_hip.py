"""ctypes binding of the C ABI in include/pympc_hip.h, plus the torch hand-off helpers.

PyTorch is plumbing here: it owns device memory and streams; every kernel is ours.  There is NO CPU fallback:
if the HIP library or a GPU is missing the product path raises.
"""
import ctypes
import threading
import os

import numpy as np
import torch

from .build import LIB_PATH

OPTIMAL, INFEASIBLE, MAXITER, NOT_PD, BAD_MODE = 0, 1, 3, 4, 5

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_d = ctypes.c_double

# name -> (restype, argtypes): exactly the entry points include/pympc_hip.h declares
SIGNATURES = {
    "pympc_hip_abi_version": (_i, []),
    "pympc_hip_last_error": (ctypes.c_char_p, []),
    "pympc_hip_tune": (_i, [ctypes.c_char_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "pympc_hip_limits": (None, [ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "pympc_mpqp_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i]),
    "pympc_mpqp_prepare": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_mpqp_solve_batch": (_i, [_i, _i, _i, _vp, _vp, _i64, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_mpqp_cuts_doubles": (ctypes.c_size_t, [_i]),
    "pympc_mpqp_cuts_reset": (_i, [_i, _vp, _vp]),
    "pympc_mpqp_cuts_learn": (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _d, _vp, _vp]),
    "pympc_mpqp_cuts_mark": (_i, [_i, _vp, _vp]),
    "pympc_mpqp_cuts_restore": (_i, [_i, _vp, _vp]),
    "pympc_mpqp_solve_batch_cuts": (_i, [_i, _i, _i, _vp, _vp, _vp, _i64, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_mpqp_solve_batch_static": (_i, [_i, _i, _i, _vp, _vp, _i64, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_nnls_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i64]),
    "pympc_nnls_batch": (_i, [_i, _i, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_pnnls_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i, _i64]),
    "pympc_pnnls_batch": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pympc_qp_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i64]),
    "pympc_qp_solve_batch": (_i, [_i, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp]),
    "pympc_mpqp_multi_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i, _i64]),
    "pympc_mpqp_multi_solve": (_i, [_i, _i, _i, _i64] + [_vp] * 10 + [_i64, _d] + [_vp] * 8),
    "pympc_condense_limits": (None, [ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "pympc_condense_batch": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i64] + [_vp] * 23),
    "pympc_lp_workspace_doubles": (ctypes.c_size_t, [_i, _i, _i64, _i]),
    "pympc_lp_limits": (None, [ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "pympc_lp_solve_batch": (_i, [_i, _i, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp]),
    "pympc_point_location": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i,
                                  _vp, _i64, _d, _vp, _vp, _vp, _vp]),
}


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load_library(path=LIB_PATH):
    """dlopen the HIP library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise HipLibraryError(
                "%s is missing: build it with `python -m pympc_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback for the pympc_amd solvers." % path)
        # torch's bundled libamdhip64.so (SONAME libamdhip64.so.7) is already mapped by `import torch`, so the
        # library binds to the same HIP runtime the tensors and streams come from.
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load_library().pympc_hip_last_error().decode()
        if rc == -2:
            raise NotImplementedError("pympc_hip: " + msg)
        raise HipLibraryError("pympc_hip error %d: %s" % (rc, msg))


def device():
    if not torch.cuda.is_available():
        raise HipLibraryError("pympc_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Kernel scratch of the batched calls (the `ws` argument of the C ABI: overwritten from its first use on, never read by
# the caller): one grow-only tensor per (device, stream) instead of an allocation per call.  Launches on one stream run
# in order, so the next call's kernels only touch it after this call's are done; calls on different streams, or from
# different Python threads (ctypes drops the GIL during a call, so two threads' launches can interleave on one stream),
# get different tensors.  Large requests (a whole batch of per-problem workspaces) are not kept.
_SCRATCH = {}
SCRATCH_KEEP_DOUBLES = 1 << 23     # 64 MiB


def scratch(nd):
    dev = device()
    if nd > SCRATCH_KEEP_DOUBLES:
        return torch.empty(nd, dtype=torch.float64, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nd:
        t = torch.empty(min(SCRATCH_KEEP_DOUBLES, max(nd, 2 * (0 if t is None else t.numel()))), dtype=torch.float64, device=dev)
        _SCRATCH[key] = t
    return t


def to_device(a, dtype=torch.float64):
    """numpy / torch -> contiguous device tensor (no copy if already there)."""
    if isinstance(a, torch.Tensor):
        t = a.to(device=device(), dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device())
    return t.contiguous()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def limits():
    n = _i(0)
    m = _i(0)
    load_library().pympc_hip_limits(ctypes.byref(n), ctypes.byref(m))
    return n.value, m.value


def lp_limits():
    n = _i(0)
    m = _i(0)
    load_library().pympc_lp_limits(ctypes.byref(n), ctypes.byref(m))
    return n.value, m.value


def tune(key, value=-1):
    """pympc_hip_tune: set a launch-policy knob (value >= 0) and return its previous value; value < 0 only queries."""
    prev = ctypes.c_longlong(0)
    check(load_library().pympc_hip_tune(key.encode(), int(value), ctypes.byref(prev)))
    return prev.value


class tuned(object):
    """Context manager: `with tuned(lp_tile_min_rows=1, lp_tile_min_batch=1): ...` restores the knobs on exit."""

    def __init__(self, **knobs):
        self.knobs = knobs
        self.old = {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.old[k] = tune(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tune(k, v)
        return False
