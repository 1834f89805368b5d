"""ctypes binding of libmma_amd.so (include/mma_amd.h).  Loud failure, no fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMA_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "libmma_amd.so")   # override: A/B builds in development
ABI_VERSION = 16

_c = ctypes
_P, _I64, _I32, _U32, _U64 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_uint32, _c.c_uint64

# name -> argtypes, exactly the prototypes of include/mma_amd.h
PROTOTYPES = {
    "mma_nc_fused_fwd": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _I64,
                         _I64, _I64, _I32, _I32, _P, _P, _I32, _U32, _U64, _P, _I64, _P, _P],
    "mma_nc_bwd_node": [_P, _I64, _I64, _P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P],
    "mma_nc_fused_bwd": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64,
                         _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I32, _I32, _P, _I32, _U32, _U64, _P, _I64, _P, _P],
    "mma_csr_spmm": [_P, _P, _P, _P, _I64, _I64, _I32, _P, _P, _I64, _I64, _I32, _P],
    "mma_build_csr": [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _I64, _P],
    "mma_gr_fused_fwd": [_P, _P, _P, _P, _P, _I64, _P, _I64, _I32, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P,
                         _I64, _I64, _I32, _I32, _P, _I32, _P, _I32, _c.c_float, _c.c_float, _I32, _U32, _U64, _P, _P],
    "mma_gr_fused_bwd": [_P, _P, _P, _P, _P, _I64, _P, _I64, _I32, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _I64,
                         _I64, _I64, _I32, _I32, _P, _I32, _P, _I32, _c.c_float, _c.c_float, _I32, _U32, _U64, _P, _P],
    "mma_csr_spmm_items": [_P, _P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _P],
    "mma_split_bf16x3": [_P, _I64, _P, _P],
    "mma_gemm_bf16x3": [_P, _I64, _P, _P, _I64, _I64, _I32, _I32, _I32, _P],
    "mma_pack_rows": [_P, _I64, _P, _I64, _P, _I64, _I32, _P],
    "mma_unpack_add_rows": [_P, _I64, _P, _I64, _P, _I64, _I32, _P],
    "mma_unpack_add_rows_csr": [_P, _I64, _P, _P, _P, _I64, _P, _I64, _I32, _P],
    "mma_col_sum": [_P, _I64, _I64, _I32, _P, _P, _I64, _P],
    "mma_gemm_bf16x3_tn": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P],
    "mma_tower_linear_bwd": [_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P],
}

_lib = None


class MMALibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMALibraryError(
                "mma_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mma_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.mma_abi_version.restype = _I32
        L.mma_last_error.restype = _c.c_char_p
        if L.mma_abi_version() != ABI_VERSION:
            raise MMALibraryError("mma_amd: %s has ABI %d, expected %d: rebuild" % (LIB_PATH, L.mma_abi_version(), ABI_VERSION))
        L.mma_nc_aux_row_floats.argtypes, L.mma_nc_aux_row_floats.restype = [_I32, _I32, _P], _I64
        L.mma_csr_workspace_bytes.argtypes, L.mma_csr_workspace_bytes.restype = [_I64, _I64], _I64
        L.mma_gr_arg_side_rows.argtypes, L.mma_gr_arg_side_rows.restype = [_I64], _I64
        L.mma_gr_long_nodes_len.argtypes, L.mma_gr_long_nodes_len.restype = [_I64], _I64
        L.mma_col_sum_workspace_floats.argtypes, L.mma_col_sum_workspace_floats.restype = [_I64, _I32], _I64
        L.mma_gemm_bf16x3_tn_workspace_floats.argtypes, L.mma_gemm_bf16x3_tn_workspace_floats.restype = [_I64, _I32, _I32], _I64
        L.mma_tower_linear_bwd_blocks.argtypes, L.mma_tower_linear_bwd_blocks.restype = [_I64], _I64
        for name, args in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.argtypes, fn.restype = args, _I32
        _lib = L
    return _lib


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise MMALibraryError("%s failed (code %d): %s" % (name, rc, L.mma_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MMALibraryError("mma_amd: got a %s tensor; this path runs on the GPU only (no CPU fallback)" % t.device)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def host_codes(codes):
    return (ctypes.c_uint8 * len(codes))(*codes)
