"""Bindings of libmma_amd.so (include/mma_amd.h).  Loud failure, no fallback.

Two bindings of the SAME extern "C" entry points, both generated from the header by tools/gen_bindings.py:
  * torch ops (default when built): `torch.ops.mma_amd.<entry point>` registered by csrc/libmma_amd_torch.so
    (`TORCH_LIBRARY(mma_amd, ...)`, csrc/torch_ops.cpp) - tensors in, the current HIP stream taken in C++;
  * ctypes (MMA_BINDING=ctypes, or when the op library is not built): what a non-torch host binds.
Either way the work happens in the HIP kernels of libmma_amd.so; a missing or stale library raises MMALibraryError."""
import ctypes
import os

import torch

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMA_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "libmma_amd.so")   # override: A/B builds in development
OPS_PATH = os.path.join(_HERE, "csrc", "libmma_amd_torch.so")
ABI_VERSION = _abi.ABI_VERSION

_c = ctypes
_P, _I64, _I32, _U32, _U64 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_uint32, _c.c_uint64
_CT = {"T": _P, "H": _P, "S": _P, "i64": _I64, "i32": _I32, "u32": _U32, "u64": _U64, "f32": _c.c_float, "f64": _c.c_double}
_RET = {"int": _I32, "int64_t": _I64, "const char*": _c.c_char_p}

# name -> ctypes argtypes of the entry points that take a stream (the launchers), exactly the prototypes of include/mma_amd.h
PROTOTYPES = {name: [_CT[c] for _, c, _ in params] for name, (ret, params) in _abi.FUNCTIONS.items()
              if ret == "int" and params and params[-1][0] == "S"}
_KINDS = {name: [k if k in "THS" else c for k, c, _ in params] for name, (ret, params) in _abi.FUNCTIONS.items()}

_lib = None
_ops = None        # torch.ops.mma_amd once the op library is loaded; False: ctypes binding


class MMALibraryError(RuntimeError):
    pass


class _Stream:      # placeholder the call sites pass for `void* stream`
    pass


STREAM = _Stream()


def lib():
    """The loaded library; raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMALibraryError(
                "mma_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mma_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        L.mma_abi_version.restype = _I32
        L.mma_last_error.restype = _c.c_char_p
        if L.mma_abi_version() != ABI_VERSION:
            raise MMALibraryError("mma_amd: %s has ABI %d, expected %d: rebuild" % (LIB_PATH, L.mma_abi_version(), ABI_VERSION))
        for name, (ret, params) in _abi.FUNCTIONS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.argtypes, fn.restype = [_CT[c] for _, c, _ in params], _RET[ret]
        _lib = L
    return _lib


def ops():
    """torch.ops.mma_amd (the TORCH_LIBRARY binding) or False when the ctypes binding is in use."""
    global _ops
    if _ops is None:
        _ops = False
        want = os.environ.get("MMA_BINDING", "")
        if want not in ("", "torch", "ctypes"):
            raise MMALibraryError("MMA_BINDING=%r: expected torch or ctypes" % want)
        if want != "ctypes" and not os.environ.get("MMA_LIB_OVERRIDE"):
            lib()                                   # version check first; also makes the symbols global for the op library
            if os.path.exists(OPS_PATH):
                torch.ops.load_library(OPS_PATH)
                _ops = torch.ops.mma_amd
            elif want == "torch":
                raise MMALibraryError("mma_amd: MMA_BINDING=torch but %s is not built (python -c 'import __graft_entry__ as g; g.build()')"
                                      % OPS_PATH)
    return _ops


def binding():
    return "torch" if ops() else "ctypes"


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def call(name, *args):
    """Invoke a launcher.  Pointer parameters take a tensor (or None), `*_host` parameters a sequence of small ints, the
    stream parameter is _lib.STREAM (or anything: torch's current stream is used)."""
    kinds = _KINDS[name]
    if len(args) != len(kinds):
        raise TypeError("%s takes %d arguments, got %d" % (name, len(kinds), len(args)))
    o = ops()
    if o:
        conv = []
        for k, a in zip(kinds, args):
            if k == "S":
                continue
            if k == "H":
                conv.append(None if a is None else [int(x) for x in a])
            elif k == "u64":
                conv.append(_signed64(int(a)))
            elif k in ("f32", "f64"):
                conv.append(float(a))
            elif k == "T":
                conv.append(a)
            else:
                conv.append(int(a))
        try:
            getattr(o, name)(*conv)
        except RuntimeError as e:
            raise MMALibraryError(str(e).split("\n")[0]) from None
        return
    L = lib()
    conv = []
    for k, a in zip(kinds, args):
        if k == "T":
            conv.append(None if a is None else (a.data_ptr() if torch.is_tensor(a) else a))
        elif k == "H":
            conv.append(None if a is None else (ctypes.c_uint8 * len(a))(*a))
        elif k == "S":
            conv.append(torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None)
        else:
            conv.append(a)
    rc = getattr(L, name)(*conv)
    if rc != 0:
        raise MMALibraryError("%s failed (code %d): %s" % (name, rc, L.mma_last_error().decode()))


def query(name, *args):
    """The host-only size helpers (mma_*_workspace_*, mma_nc_crow_floats, ...): plain ctypes, no stream."""
    conv = [(ctypes.c_uint8 * len(a))(*a) if isinstance(a, (list, tuple)) else a for a in args]
    return int(getattr(lib(), name)(*conv))


def ptr(t):
    """A pointer argument: the tensor itself (None -> NULL); the binding takes its device pointer."""
    return t


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MMALibraryError("mma_amd: got a %s tensor; this path runs on the GPU only (no CPU fallback)" % t.device)


def stream_ptr():
    return STREAM


def host_codes(codes):
    return tuple(int(c) for c in codes)
