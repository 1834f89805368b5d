#!/usr/bin/env python3
"""Audit of hand-placed memory waits in the compiled kernels (cdna_hip_programming.md 5.7: an inline-asm load is invisible to hipcc's
s_waitcnt bookkeeping, its destination registers count as written at ;;#ASMEND, so the compiler may copy, spill or reuse them before the
data has landed - "no fault, no message, a passing check is not evidence").

For every kernel of a .s file that contains inline-asm loads, walk the control-flow graph from the kernel entry and simulate the ONE
in-order vector-memory counter (vmcnt: loads, stores and atomics count together, in issue order; `s_waitcnt vmcnt(N)` retires all but the
N youngest).  An instruction that mentions a VGPR which is the destination of a load still in flight - a v_mov copy at a loop back edge,
a spill, an early use - is a violation.  States are (pc, in-flight queue) pairs, each visited once, so loops converge.

    hipcc ... -save-temps=obj -c gemm_x3.hip -o /tmp/x/gemm_x3.o ; python tools/check_asm_waits.py /tmp/x/gemm_x3-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-substring]
Exit status 1 when any kernel has a violation.

--asm-only keeps ONLY the inline-asm loads in the modelled queue (compiler-issued loads and stores are left out).  That model is
conservative - an asm load has at least as many younger operations in the real queue as in the model, so whatever a wait retires in the
model it retires on the machine - and it is what makes kernels with many conditional stores tractable: the number of stores on a path no
longer multiplies the states."""
import re
import sys

VM_OP = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|scratch_load|scratch_store|flat_load|flat_store)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
MAXQ = 80


def regs(text):
    out = []
    for m in REG.finditer(text):
        out.append((int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3))))
    return out


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line.rstrip("\n"))
            if "s_endpgm" in line:
                yield name, body
                name = None


ASM_ONLY = False


def check(name, body):
    ins, labels = [], {}
    in_asm = False
    for line in body:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if t:
            ins.append((t, in_asm))
    if not any(a and t.startswith(("global_load", "buffer_load")) for t, a in ins):
        return None
    viol, seen, stack = [], set(), [(0, ())]
    while stack:
        pc, q = stack.pop()
        while pc < len(ins):
            key = (pc, q)
            if key in seen:
                break
            seen.add(key)
            t, a = ins[pc]
            op = t.split()[0]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n < len(q) else q
                    if n == 0:
                        q = ()
            elif VM_OP.match(t):
                dst = None
                if "load" in op and not t.rstrip().endswith(" lds"):
                    r = regs(t.split(",")[0])
                    dst = r[0] if r else None
                # the address / data operands of this very instruction must not be in flight either
                for lo, hi in regs(",".join(t.split(",")[1:]) if dst else t):
                    for d in q:
                        if d and lo <= d[1] and hi >= d[0]:
                            viol.append((pc, t, d))
                if dst is not None and not a:
                    dst = None                                  # a compiler-issued load: the compiler waits for it itself
                if not (ASM_ONLY and not a):
                    q = (q + (dst,))[-MAXQ:]
            else:
                for lo, hi in regs(t):
                    for d in q:
                        if d and lo <= d[1] and hi >= d[0]:
                            viol.append((pc, t, d))
                            break
                if op.startswith("s_cbranch"):
                    tgt = t.split()[-1]
                    if tgt in labels:
                        stack.append((labels[tgt], q))
                elif op == "s_branch":
                    tgt = t.split()[-1]
                    pc = labels.get(tgt, len(ins))
                    continue
                elif op == "s_endpgm":
                    break
            pc += 1
    uniq = sorted({(pc, t, d) for pc, t, d in viol})
    return uniq, len(seen), sum(1 for t, a in ins if a and "load" in t)


def main():
    global ASM_ONLY
    args = [a for a in sys.argv[1:] if a != "--asm-only"]
    ASM_ONLY = "--asm-only" in sys.argv[1:]
    path = args[0]
    want = args[1] if len(args) > 1 else ""
    bad = 0
    for name, body in kernels(path):
        if want and want not in name:
            continue
        r = check(name, body)
        if r is None:
            continue
        v, states, n_loads = r
        print("%s: %d asm loads, %d states walked, %d violation(s)" % (name, n_loads, states, len(v)))
        for pc, t, d in v[:12]:
            print("    instruction %d `%s` touches v[%d:%d] while its load is in flight" % (pc, t, d[0], d[1]))
        bad += len(v)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
