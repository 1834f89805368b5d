#!/usr/bin/env python3
"""ISA audit of the HIP kernels (no GPU needed: hipcc -S cross-compiles): the two patterns that cost time without showing in the source.

  1. FLAT memory operations.  `cond ? lds_ptr[i] : global_ptr[j]` compiles into ONE flat load from a selected address; a flat load counts in
     vmcnt AND lgkmcnt, so the wait for it is `s_waitcnt vmcnt(0) lgkmcnt(0)` - inside a loop that also stores, a wait for the
     acknowledgement of every store issued before it (round 5: K4's per-edge loop, 0.39 -> 0.34 ms once the loop was split on the condition).
  2. Innermost loops that store AND wait for vmcnt(0): gfx950 has one in-order counter for loads and stores, so a load issued behind
     stores is a load that waits for them (K16's forward epilogue read its bias per stored element: 0.074 -> 0.063 ms with the bias in
     registers before the first store).

    python tools/isa_audit.py [file.hip ...]        (default: every mma_amd/csrc/*.hip; ~3 min for all of them on 8 cores)
    python tools/isa_audit.py --filter 'gr_bwd_block|segsum' mma_amd/csrc/gr_fused.hip mma_amd/csrc/spmm_rows.hip

Exit code 1 when a kernel has flat operations that are not on the allow list below (kernels that take pointer tables by design)."""
import argparse
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -S --cuda-device-only".split()
# flat operations by design: Adam walks a table of tensor pointers; the halo pack kernels select between two row sources per block;
# the block backward of GR keeps ONE flat load in its generic (un-staged or by-edge-id) form of the edge loop
ALLOW = re.compile(r"adam_kernel|pack_blocks_kernel|gr_bwd_block_kernel")


def asm_of(src):
    out = os.path.join(tempfile.gettempdir(), "isa_audit_" + os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return src, open(out).read()


def kernels(asm):
    for part in re.split(r"\n(?=_Z[\w]+:\s*(?:;.*)?\n)", asm):
        m = re.match(r"(_Z\w+):", part)
        if m:
            yield m.group(1), part.split(".Lfunc_end")[0].split("\n")


def innermost_loops(lines):
    labels, loops = {}, []
    for n, line in enumerate(lines):
        m = re.match(r"(\.LBB\w+):", line.strip())
        if m:
            labels[m.group(1)] = n
    for n, line in enumerate(lines):
        t = line.strip()
        if t.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < n:
                loops.append((labels[tgt], n))
    return [(a, b) for (a, b) in loops if not any(a <= c and d <= b and (c, d) != (a, b) for (c, d) in loops)]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.split("\n")))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--filter", default="", help="regular expression on the demangled kernel name")
    ap.add_argument("--loops", action="store_true", help="also list innermost loops that store and wait for vmcnt(0)")
    args = ap.parse_args()
    files = [os.path.abspath(f) for f in args.files] or sorted(glob.glob(os.path.join(ROOT, "mma_amd", "csrc", "*.hip")))
    bad = 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        for src, asm in ex.map(asm_of, files):
            ks = list(kernels(asm))
            names = demangle([k for k, _ in ks])
            n_flat = n_k = 0
            for k, lines in ks:
                name = names[k]
                if args.filter and not re.search(args.filter, name):
                    continue
                n_k += 1
                body = [line.strip() for line in lines]
                flat = [line for line in body if re.match(r"flat_(load|store|atomic)", line)]
                if flat:
                    n_flat += 1
                    ok = bool(ALLOW.search(name))
                    bad += 0 if ok else 1
                    print("%s  %s: %d flat operation(s)%s" % (os.path.basename(src), name[:110], len(flat), " (allowed)" if ok else "  <-- check"))
                if args.loops:
                    for a, b in innermost_loops(lines):
                        lb = [x for x in body[a:b + 1] if x and not x.startswith(";")]
                        st = sum(1 for x in lb if re.match(r"(global|buffer|flat)_(store|atomic)", x))
                        w0 = sum(1 for x in lb if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
                        if st and w0 and len(lb) < 400:
                            print("%s  %s: loop of %d instructions with %d store(s) and %d vmcnt(0) wait(s)" % (
                                os.path.basename(src), name[:90], len(lb), st, w0))
            print("%s: %d kernels, %d with flat operations" % (os.path.basename(src), n_k, n_flat))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
