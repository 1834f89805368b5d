#!/usr/bin/env python3
"""Host-side sanitizer run of the C-ABI library ("sanitizers on the CPU build").

Runs inside a process that has the AddressSanitizer runtime preloaded and loads a build of mma_amd/csrc whose HOST code is
instrumented (`hipcc -fsanitize=address,undefined -fno-gpu-sanitize`, made by tools/sanitize_host.sh): every entry point of
include/mma_amd.h is called many times with structured random arguments - sizes from an edge-heavy set (0, 1, powers of two
+-1, int32 limits), strides, code lists valid and invalid, NULL and non-NULL pointers.  "Device" pointers are fake aligned
addresses: the host side of the library never dereferences them (only the code lists marked H are host memory), and
without a GPU / device code every launch fails with a HIP error that the entry point must turn into a non-zero return.
What is under test is therefore exactly the host logic: argument validation, grid / LDS / magic-number planning, workspace
arithmetic - any out-of-bounds access, signed overflow, bad shift or misaligned access there aborts the process.

    python tests/sanitize_driver.py <libmma_amd.so> [n_calls_per_function] [seed]
prints 'SANITIZE_OK <calls> <returned_zero>'."""
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mma_amd import _abi  # noqa: E402  (the generated table of include/mma_amd.h; imports nothing else)

CT = {"T": ctypes.c_void_p, "H": ctypes.c_void_p, "S": ctypes.c_void_p, "i64": ctypes.c_int64, "i32": ctypes.c_int32,
      "u32": ctypes.c_uint32, "u64": ctypes.c_uint64, "f32": ctypes.c_float, "f64": ctypes.c_double}
RET = {"int": ctypes.c_int32, "int64_t": ctypes.c_int64, "const char*": ctypes.c_char_p}
SIZES = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 32, 33, 63, 64, 65, 75, 76, 96, 127, 128, 129, 255, 256, 380, 456, 511, 512, 1000,
         1024, 4096, 65535, 65536, 204552, 427376, 1 << 20, (1 << 20) + 17, 10864894, (1 << 24) - 1, 1 << 24, (1 << 31) - 1,
         1 << 31, (1 << 31) + 5, 1 << 40, -1, -7]
SMALL = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 32, 64, 128, 255, 256, -1]
HIST = {} if os.environ.get("SANITIZE_HIST") else None      # SANITIZE_HIST=1: print where the consistent calls ended
FAKE = 0x7F0000000000           # a 2 MB-aligned address in no mapping: a host dereference would fault under any build


def pick_int(rng, name, code):
    small_names = ("K", "S", "T", "F", "O", "C", "H", "N32", "accumulate", "by_pos", "drop_mode", "n_tensors", "KA", "NC", "kinds")
    if code == "i32" or name in small_names:
        if rng.random() < 0.75:
            return rng.choice(SMALL + [75, 76, 380, 456, 512, 1024])
        return rng.choice(SIZES)
    return rng.choice(SIZES) if rng.random() < 0.8 else rng.randrange(0, 1 << 22)


def recipe(lib, name, rng, keep):
    """Arguments that agree with each other (name -> value; pointers not listed are non-NULL fakes), so that a call gets through
    the validation and into the host-side planning code (grids, LDS layout, magic numbers, splits) before its launch fails."""
    def codes(vals):
        arr = (ctypes.c_uint8 * 8)(*(list(vals) + [0] * (8 - len(vals))))
        keep.append(arr)
        return ctypes.cast(arr, ctypes.c_void_p)
    N = rng.choice([1, 7, 300, 1000, 204552, 1 << 20, (1 << 20) + 3, (1 << 31) - (1 << 20) - 1])
    E = rng.choice([0, 5, 2100, 427376, 10864894, (1 << 31) - 1])
    H = rng.choice([1, 4, 16, 64, 75, 128, 256])
    K = rng.choice([1, 2, 3, 4, 8])
    pad = rng.choice([0, 0, 4, 1])
    if name.startswith("mma_nc_"):
        kinds = [rng.randrange(0, 6) for _ in range(K)]
        n_items = rng.choice([0, 1, 100, 70000, 1 << 22])
        hubs = rng.random() < 0.5
        d = dict(ldx=H + pad, ldp=K * H + pad, ldq=K * H + pad, n_items=n_items, n_wave_items=rng.choice([0, n_items // 2, n_items]),
                 n_hubs=5 if hubs else 0, n_slots=40 if hubs else 0, ldms=H + pad, ldt=K * H, N=N, E=E, H=H, K=K, kind_host=codes(kinds),
                 act_host=codes([rng.randrange(0, 2) for _ in range(K)]), drop_mode=rng.choice([0, 1, 2]), drop_thr=rng.choice([0, 128, 255, 32768, 39322, 65535]),
                 drop_edge_base=rng.choice([0, 0, 1 << 20]), g_kstride=rng.choice([0, 4 * ((N * H + 3) // 4)]), ldgr=H + pad, ldgs=K * H,
                 ldgp=K * H, ldgx=H + pad, ldg=K * H, ldgq=K * H, ldgxo=H + pad)
        lib.mma_nc_crow_floats.restype = ctypes.c_int64
        d["ldc"] = int(lib.mma_nc_crow_floats(H, K, d["kind_host"]))
        d.update(ldgg=H + pad, n_targets=rng.choice([0, N // 2, N]))
        if not hubs:
            d.update(hubs=None, partial=None)
        if rng.random() < 0.3:
            d.update(T=None, sel=None, crow=None)
        if rng.random() < 0.5:
            d.update(sync=None)          # else a fake counter: the one-launch plan is laid out on the host
        return d
    if name.startswith("mma_gr_fused"):
        T, F = rng.choice([(1, 1), (1, 4), (2, 3), (5, 76), (5, 75), (1, 128), (4, 16), (8, 8)])
        D = T * F
        aggs = rng.choice([[2, 3], [0], [1], [2], [3], [0, 1, 2, 3], [0, 1, 2, 3, 4, 5], [4], [5, 0]])
        scal = rng.choice([[0], [0, 1], [0, 1, 2], [0, 1, 2, 3, 4], [3]])
        Nn = min(N, (1 << 31) - (1 << 20) - 1)
        d = dict(lduv=2 * D, ldz=D, by_pos=rng.choice([0, 1]), ldi=D, ldsave=D, ldg=D, ldgu=2 * D, N=Nn, E=E, T=T, F=F, aggr_host=codes(aggs),
                 K=len(aggs), scaler_host=codes(scal), S=len(scal), avg_log=1.2, avg_lin=2.1, drop_mode=rng.choice([0, 1]),
                 drop_thr=rng.choice([0, 128, 255, 32768, 39322, 65535]))
        if rng.random() < 0.4:
            d.update(U=None, V=None, Z=None)          # given-messages form
        else:
            d.update(inputs=None)
            if rng.random() < 0.4:
                d.update(Z=None)
        if 2 not in aggs:
            d.update(amin8=None, amin_side=None)
        if 3 not in aggs:
            d.update(amax8=None, amax_side=None)
        if not any(a >= 4 for a in aggs):
            d.update(mean=None, var=None)
        if rng.random() < 0.3:
            d.update(long_nodes=None)
        if rng.random() < 0.7 or d.get("Z", 1) is None:
            d.update(z_index=None)
        return d
    if name == "mma_gemm_bf16x3":
        M, Nc, Kc = rng.choice([(1 << 20, 1024, 128), (70001, 512, 128), (9001, 128, 1024), (1 << 20, 128, 512), (5, 32, 128), (0, 64, 256),
                                (300000, 128, 128), (40037, 1024, 128), (66001, 128, 256), (100, 96, 384), ((1 << 31) + 7, 128, 128)])
        return dict(lda=Kc + rng.choice([0, 4, 128]), ldc=Nc + rng.choice([0, 4]), M=M, N=Nc, K=Kc, accumulate=rng.choice([0, 1]))
    if name == "mma_gemm_bf16x3_tn":
        M, KA, NC = rng.choice([(1 << 20, 128, 1024), (5000, 64, 96), (33, 32, 32), (262161, 96, 160), (1, 128, 512), ((1 << 31) + 9, 128, 1024)])
        lib.mma_gemm_bf16x3_tn_workspace_floats.restype = ctypes.c_int64
        ws = int(lib.mma_gemm_bf16x3_tn_workspace_floats(M, KA, NC))
        return dict(ldx=KA + rng.choice([0, 128]), ldg=NC + rng.choice([0, 32]), ws_floats=ws, M=M, KA=KA, NC=NC)
    if name == "mma_gemm_f16x2_tn":
        M, KA, NC = rng.choice([(1 << 20, 128, 1024), (5000, 64, 96), (33, 32, 32), (262161, 96, 160), (1, 128, 512), ((1 << 31) + 9, 128, 1024),
                                ((1 << 30) - 1, 128, 1024)])
        lib.mma_gemm_f16x2_tn_workspace_floats.restype = ctypes.c_int64
        ws = int(lib.mma_gemm_f16x2_tn_workspace_floats(M, KA, NC))
        d = dict(ldx=KA + rng.choice([0, 128]), ldg=NC + rng.choice([0, 32]), ws_floats=ws, M=M, KA=KA, NC=NC)
        if rng.random() < 0.5:
            d.update(x_row_max=None)
        if rng.random() < 0.3:
            d.update(g_row_max=None)
        return d
    if name == "mma_gemm_f16x2":
        M, Nc = rng.choice([(1 << 20, 1024), (70001, 512), (5, 128), (0, 256), ((1 << 31) + 7, 128)])
        d = dict(lda=128 + rng.choice([0, 4, 128]), ldc=Nc + rng.choice([0, 4]), M=M, N=Nc)
        if rng.random() < 0.5:
            d.update(a_row_max=None)
        return d
    if name == "mma_gemm_f16x2_k256":
        M, Nc = rng.choice([(1 << 20, 4096), (70001, 512), (5, 128), (0, 256), ((1 << 31) + 7, 128)])
        return dict(lda=256 + rng.choice([0, 4, 128]), ldc=Nc + rng.choice([0, 4]), M=M, N=Nc)
    if name == "mma_row_absmax":
        M, C = rng.choice([(0, 7), (1, 1), (70001, 256), (1 << 20, 128), ((1 << 31) + 3, 75)])
        return dict(lda=C + rng.choice([0, 5]), M=M, cols=C)
    if name == "mma_split_f16x2":
        K, Nc = rng.choice([(128, 1024), (1024, 128), (256, 4096), (1, 1), (1 << 20, 4096)])
        return dict(stride_k=rng.choice([Nc, 1]), stride_n=rng.choice([1, K]), K=K, N=Nc, plain_lo=rng.choice([0, 1]))
    if name == "mma_gemm_f16x2_nlp":
        M, Nc, K = rng.choice([(0, 128, 256), (1, 128, 1024), (1 << 20, 256, 4096), (70001, 128, 1024), (1000, 256, 256), (5, 128, 192), (5, 64, 256)])
        return dict(lda=K + rng.choice([0, 64]), ldc=Nc + rng.choice([0, 4]), M=M, N=Nc, K=K, accumulate=rng.choice([0, 1]))
    if name == "mma_col_sum":
        R, C = rng.choice([(0, 7), (1, 1), (1000, 375), (204552, 375), (427376, 75), (300001, 130), (1 << 31, 16)])
        lib.mma_col_sum_workspace_floats.restype = ctypes.c_int64
        return dict(ldg=C + rng.choice([0, 5]), R=R, C=C, ws_floats=int(lib.mma_col_sum_workspace_floats(R, C)))
    if name == "mma_tower_linear_bwd":
        Nn, T, O, C = rng.choice([(1, 1, 1, 4), (1000, 5, 15, 456), (70001, 3, 16, 260), (204552, 5, 15, 456), (1 << 30, 5, 15, 456)])
        lib.mma_tower_linear_bwd_blocks.restype = ctypes.c_int64
        return dict(n_blocks=int(lib.mma_tower_linear_bwd_blocks(Nn)), N=Nn, T=T, O=O, C=C)
    if name == "mma_build_csr":
        Ee, Nn = rng.choice([(0, 5), (5, 5), (427376, 204552), (10864894, 1 << 20), ((1 << 31) - 1, (1 << 31) - 1), (100, (1 << 30) + 1)])
        lib.mma_csr_workspace_bytes.restype = ctypes.c_int64
        return dict(E=Ee, N=Nn, workspace_bytes=int(lib.mma_csr_workspace_bytes(Ee, Nn)))
    if name == "mma_adam_step":
        return dict(n_tensors=rng.choice([0, 1, 8, 25]), total_chunks=rng.choice([0, 1, 100, 1 << 20]), lr=0.01, beta1=0.9, beta2=0.999,
                    eps=1e-8, weight_decay=5e-4)
    if name.startswith("mma_logsoftmax"):
        C = rng.choice([1, 7, 100, 1000])
        return dict(ldx=C, ldo=C, ldg=C, n_idx=rng.choice([0, 1, 140, N]), N=N, C=C)
    if name.startswith("mma_csr_spmm"):
        C = rng.choice([1, 4, 16, 7, 64, 380])
        n_items = rng.choice([0, 1, 100, 70000])
        hubs = rng.random() < 0.5
        d = dict(ldb=C + pad, ldo=C + pad, n_items=n_items, n_wave_items=rng.choice([0, n_items]), n_hubs=3 if hubs else 0,
                 n_slots=12 if hubs else 0, C=C, K=K, rows_per_block=rng.choice([1, 4, 64]), n_rows=N)
        if not hubs:
            d.update(hubs=None, partial=None)
        return d
    if name.startswith("mma_pack_rows") or name.startswith("mma_unpack"):
        w = rng.choice([1, 4, 16, 128, 130])
        return dict(lds=w + pad, ldd=w + pad, width=w, n_idx=rng.choice([0, 1, 1000, 1 << 22]), n_rows=rng.choice([0, 1, 1000]))
    if name == "mma_split_bf16x3":
        return dict(n=rng.choice([0, 1, 128 * 1024, 1 << 31]))
    return None


def refuse_if_a_gpu_is_visible():
    """Launchers are called with arguments that PASS validation and fake device pointers: harmless only while every launch
    fails.  tools/sanitize_host.sh hides the devices; this is the second lock."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return          # no HIP runtime at all: nothing can launch
    n = ctypes.c_int(0)
    rc = hip.hipGetDeviceCount(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        sys.exit("sanitize_driver: %d GPU(s) visible - refusing to call launchers with fake device pointers "
                 "(run through tools/sanitize_host.sh, which sets HIP_VISIBLE_DEVICES=-1)" % n.value)


def main():
    refuse_if_a_gpu_is_visible()
    lib = ctypes.CDLL(sys.argv[1])
    n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    lib.mma_last_error.restype = ctypes.c_char_p
    assert lib.mma_abi_version() == _abi.ABI_VERSION
    total = zero = 0
    keep = []
    for name, (ret, params) in _abi.FUNCTIONS.items():
        fn = getattr(lib, name)
        fn.restype = RET[ret]
        fn.argtypes = [CT[c] for _, c, _ in params]
        if not params:
            fn()
            continue
        for _ in range(n_calls):
            # one "shape" per call so that related arguments agree often enough to get past the first checks
            base = rng.choice(SIZES[:30])
            wide = rng.choice([4, 16, 32, 64, 75, 76, 128, 256, 380, 512, 1024])
            args = []
            rec = recipe(lib, name, rng, keep) if rng.random() < 0.6 else None
            if rec is not None:
                for i, (_, code, pname) in enumerate(params):
                    if pname in rec:
                        args.append(rec[pname])
                    elif code == "T":
                        args.append(FAKE + 4096 * (i + 1))
                    elif code == "S":
                        args.append(None)
                    elif code == "u64":
                        args.append(rng.getrandbits(64))
                    else:
                        raise AssertionError("%s: recipe has no %s" % (name, pname))
                if rng.random() < 0.35:              # ... and one argument pushed off
                    j = rng.randrange(len(params))
                    code = params[j][1]
                    if code in ("i64", "i32"):
                        v = rng.choice(SIZES if code == "i64" else SMALL + [(1 << 31) - 1, -(1 << 31)])
                        args[j] = max(-(1 << 31), min((1 << 31) - 1, v)) if code == "i32" else v
                    elif code == "T":
                        args[j] = None if rng.random() < 0.7 else FAKE + 4
                rc = fn(*args)
                total += 1
                if ret == "int":
                    zero += rc == 0
                    if rc != 0 and HIST is not None:
                        HIST[(name, (lib.mma_last_error() or b"").decode()[:60])] = HIST.get((name, (lib.mma_last_error() or b"").decode()[:60]), 0) + 1
                continue
            for _, code, pname in params:
                if code == "H":          # host code list: real memory, valid codes most of the time
                    n = 8
                    vals = [rng.randrange(0, 6) if rng.random() < 0.9 else rng.randrange(0, 256) for _ in range(n)]
                    arr = (ctypes.c_uint8 * n)(*vals)
                    keep.append(arr)
                    args.append(ctypes.cast(arr, ctypes.c_void_p) if rng.random() < 0.97 else None)
                elif code == "T":
                    r = rng.random()
                    args.append(None if r < 0.12 else FAKE + 256 * rng.randrange(0, 1 << 20) + (rng.choice([1, 2, 4, 8]) if r > 0.97 else 0))
                elif code == "S":
                    args.append(None)
                elif code in ("f32", "f64"):
                    args.append(rng.choice([0.0, 1.0, 0.5, 1e-8, 0.9, 0.999, -1.0, 1e30, float("inf"), float("nan")]))
                elif code in ("u32", "u64"):
                    args.append(rng.choice([0, 1, 127, 128, 255, 256, (1 << 32) - 1]) if code == "u32" else rng.getrandbits(64))
                else:
                    r = rng.random()
                    if pname.startswith("ld") or pname.endswith("pitch"):
                        v = wide * rng.choice([1, 1, 1, 2, 4]) if r < 0.85 else pick_int(rng, pname, code)
                    elif pname in ("N", "M", "R", "E", "n_items", "n_rows", "n"):
                        v = base if r < 0.6 else pick_int(rng, pname, code)
                    elif pname in ("H", "C", "F", "D"):
                        v = rng.choice([4, 16, 64, 75, 76, 128, 256]) if r < 0.8 else pick_int(rng, pname, code)
                    else:
                        v = pick_int(rng, pname, code)
                    if code == "i32":
                        v = max(-(1 << 31), min((1 << 31) - 1, v))
                    args.append(v)
            rc = fn(*args)
            total += 1
            if ret == "int":
                zero += rc == 0
                if rc != 0:
                    assert lib.mma_last_error() is not None
            del keep[:-64]
    for (fname, msg), c in sorted((HIST or {}).items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("%6d %-28s %s" % (c, fname, msg))
    print("SANITIZE_OK %d %d" % (total, zero), flush=True)


if __name__ == "__main__":
    main()
