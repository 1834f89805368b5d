#!/usr/bin/env python3
"""Identity of the build a measurement belongs to.  tools/prof_*.sh write it next to the rocprofv3 passes on the GPU box,
tools/make_profiles.py copies it into profiles/r*_pmc_traffic*.json, and bench.py REFUSES a recorded PMC traffic figure whose
stamp differs from the running tree's (round-2 VERDICT item 8: a traffic figure must not outlive the kernel it was measured on).

Round 4: kernel traffic also depends on the host-side planners and switches, and a stale .so must not pass.  The stamp is
  abi_version  include/mma_amd.h
  kernel_sha   the SHA-256 over the kernel sources (mma_amd/csrc/*.hip, common.h, the header) EMBEDDED IN THE LOADED .so at build
               time (csrc/Makefile writes build_stamp.h from `--kernel-sha`; libmma_amd.so returns it from mma_build_stamp()); when
               no library is loadable the tree's own value is used and `library` says so
  host_sha     SHA-256 over the Python planners whose choices change what the kernels move (graph.py: chunking / grouping,
               functional.py, layers.py, mma_conv.py, dense.py, sharded.py)
  env          the MMA_* switches in effect (SMALL_* / ONE_LAUNCH / FUSE_NODE_BWD / FACTOR_SCALERS / FOLD_ROW_FACTOR ... are all MMA_* variables)
  stale_build  True when the loaded library was built from other sources than the tree holds

    python tools/build_stamp.py                # prints the stamp (JSON)
    python tools/build_stamp.py --kernel-sha   # prints the source SHA only (what the Makefile embeds)"""
import ctypes
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_FILES = ["graph.py", "functional.py", "layers.py", "mma_conv.py", "dense.py", "sharded.py"]


def _sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel_source_sha():
    hdr = os.path.join(ROOT, "include", "mma_amd.h")
    return _sha(sorted(glob.glob(os.path.join(ROOT, "mma_amd", "csrc", "*.hip"))) + [os.path.join(ROOT, "mma_amd", "csrc", "common.h"), hdr])


def loaded_kernel_sha():
    """The stamp the built library carries (None: no library, or one from before round 4)."""
    path = os.environ.get("MMA_LIB_OVERRIDE") or os.path.join(ROOT, "mma_amd", "csrc", "libmma_amd.so")
    try:
        L = ctypes.CDLL(path)
        L.mma_build_stamp.restype = ctypes.c_char_p
        return L.mma_build_stamp().decode()
    except (OSError, AttributeError):
        return None


def build_stamp():
    hdr = os.path.join(ROOT, "include", "mma_amd.h")
    m = re.search(r"#define\s+MMA_ABI_VERSION\s+(\d+)", open(hdr).read())
    src, lib = kernel_source_sha(), loaded_kernel_sha()
    return {"abi_version": int(m.group(1)) if m else -1, "kernel_sha": lib if lib is not None else src,
            "library": "loaded" if lib is not None else "not loadable: the tree's source SHA",
            "stale_build": bool(lib is not None and lib != src),
            "host_sha": _sha([os.path.join(ROOT, "mma_amd", f) for f in HOST_FILES]),
            # MMA_LIB_OVERRIDE included (round-4 ADVICE): a measurement library (tools/build_ablation.sh: wrong results on purpose) must not
            # pass for the product build - its abi.o also carries a "+abl..." suffix in kernel_sha, and bench.py refuses to run on one
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MMA_") and k != "MMA_BENCH_DETAIL"}}


if __name__ == "__main__":
    if "--kernel-sha" in sys.argv:
        print(kernel_source_sha())
    else:
        print(json.dumps(build_stamp()))
