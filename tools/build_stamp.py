#!/usr/bin/env python3
"""Identity of the kernel build a measurement belongs to: the ABI version of include/mma_amd.h and one SHA-256 over the kernel
sources (mma_amd/csrc/*.hip, common.h, the header).  tools/prof_*.sh write it next to the rocprofv3 passes on the GPU box,
tools/make_profiles.py copies it into profiles/r*_pmc_traffic*.json, and bench.py REFUSES a recorded PMC traffic figure whose
stamp differs from the running tree's (round-2 VERDICT item 8: a traffic figure must not outlive the kernel it was measured on).

    python tools/build_stamp.py            # prints {"abi_version": N, "kernel_sha": "..."}"""
import glob
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_stamp():
    hdr = os.path.join(ROOT, "include", "mma_amd.h")
    files = sorted(glob.glob(os.path.join(ROOT, "mma_amd", "csrc", "*.hip"))) + [os.path.join(ROOT, "mma_amd", "csrc", "common.h"), hdr]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    m = re.search(r"#define\s+MMA_ABI_VERSION\s+(\d+)", open(hdr).read())
    return {"abi_version": int(m.group(1)) if m else -1, "kernel_sha": h.hexdigest()[:16]}


if __name__ == "__main__":
    print(json.dumps(build_stamp()))
