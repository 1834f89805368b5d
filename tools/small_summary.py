"""profiles/rNN_small_graph_kernels.md from the traces of tools/prof_small.sh: the kernels of ONE hipGraph replay per configuration
(the last complete one in the trace), grouped by kernel, with the replay time the same process measured un-traced."""
import csv
import glob
import json
import os
import re
import sys
from collections import OrderedDict

MARK = {"C1": "seed_advance_kernel", "C3": "seed_advance_kernel", "C2": "pack_blocks_kernel<false>", "C2net": "adam_kernel"}
TITLE = {"C1": "C1: MMA layer fwd+bwd on the Cora structure (2 708 nodes / 10 556 edges, H=64, mean,mean2, p=0.75)",
         "C3": "C3: MMA layer fwd+bwd on the PubMed structure (19 717 nodes / 88 651 edges, H=16, min,min2,min3,min4, p=0.5)",
         "C2": "C2: MMAConv layer fwd+bwd on a ZINC-like batch of 64 molecules (T=5, F=75, min,max x identity,amplification,linear)",
         "C2net": "C2net: one TRAINING step of the graph-regression Net (4 MMAConv layers + BatchNorm + pooling + MLP, L1 loss, Adam) on a "
                  "padded batch of 64 molecules, CSR build included"}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:96]


def one_replay(rows, mark):
    idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
    if len(idx) < 3:
        return rows[-40:]
    gaps = [b - a for a, b in zip(idx, idx[1:])]
    per = max(set(gaps[-8:]), key=gaps[-8:].count)          # kernels per replay = the usual distance between two markers
    ends = [i for i, g in zip(idx[1:], gaps) if g == per]
    b = ends[-1]
    return rows[b - per:b]


def main():
    src, out = sys.argv[1], sys.argv[2]
    stamp = json.load(open(os.path.join(src, "stamp.json")))
    lines = ["# Small-graph hipGraph replays, kernel by kernel (rocprofv3 --kernel-trace)", "",
             "Build: `%s`.  One replay = the kernels between two occurrences of the marker kernel at the usual distance; durations are the "
             "profiler's start-to-end times of each kernel (a kernel that moves a few hundred KB still takes 4-5 us start to end: the step "
             "is the NUMBER of kernels).  `traced replay` = the replay time measured inside the traced process (`python tools/small_replay.py "
             "<config>` under rocprofv3: 20-30 %% slower than un-traced - the un-traced figures are in the bench line's `extra`)." % json.dumps(stamp), ""]
    for cfg in ("C1", "C3", "C2", "C2net"):
        files = glob.glob(os.path.join(src, cfg, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            continue
        rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))      # gpurun_out accumulates earlier rounds' files
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        rep = one_replay(rows, MARK[cfg])
        log = open(os.path.join(src, cfg + ".log")).read()
        m = re.search(r"replay ms ([0-9.eE+-]+) eager ms ([0-9.eE+-]+|None)", log)
        agg = OrderedDict()
        for r in rep:
            k = short(r["Kernel_Name"])
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            c, t = agg.get(k, (0, 0.0))
            agg[k] = (c + 1, t + d)
        total = sum(t for _, t in agg.values())
        lines += ["## " + TITLE[cfg], "",
                  "%d kernels per replay, sum of kernel durations %.1f us; traced replay %s ms, traced eager %s ms." %
                  (len(rep), total, ("%.3f" % float(m.group(1))) if m else "?", ("%.3f" % float(m.group(2))) if m and m.group(2) != "None" else "?"), "",
                  "| kernel | launches | total us | avg us |", "|---|---:|---:|---:|"]
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append("| `%s` | %d | %.1f | %.1f |" % (k.replace("|", "\\|"), c, t, t / c))
        lines.append("")
    open(out, "w").write("\n".join(lines))
    print("wrote", out)


if __name__ == "__main__":
    main()
