#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_stats, prof_fetch, prof_write, prof_gr, bench_default.log)
into the committed summaries under profiles/:

    python tools/make_profiles.py <tag>          # e.g. r01_v4

  profiles/<round>_bench_kernel_stats_<v>.{md,csv}   per-kernel time of `bench.py --steps 5 --warmup 2 --cpu-sample 0`
  profiles/<round>_pmc_traffic.json                  HBM traffic per kernel launch (FETCH_SIZE / WRITE_SIZE passes)
  profiles/<round>_bench_default.json                the un-profiled default `python bench.py` line, traffic patched in
  profiles/<round>_gr_c2l_kernel_stats_<v>.md        per-kernel time of the GR layer on the 10 000-graph batch
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKLOAD = ("C4 R-MAT 1,048,576 nodes / 10,864,894 directed edges, H=128, K=4 [sum,mean,max,min], p=0.5")


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def stats_table(path, calls_per_step, title, command, notes, out_md, out_csv=None, top=26):
    rows = list(csv.DictReader(open(path)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(out_md, "w") as f:
        f.write("# %s\n\nCommand (MI355X box): `%s`\n\n" % (title, command))
        f.write(notes + " All GPU kernels: %.2f ms per step.\n\n" % (total / calls_per_step / 1e6))
        f.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:top]:
            f.write("| `%s` | %s | %.3f | %.3f | %s |\n" % (r["Name"][:110], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e6, r["Percentage"]))
    if out_csv:
        shutil.copy(path, out_csv)
    return total / calls_per_step / 1e6


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_v4"
    rnd, ver = tag.split("_")
    P = os.path.join(ROOT, "profiles")
    ms = stats_table(newest("gpurun_out/prof_stats/**/*kernel_stats.csv"), 7,
                     "rocprofv3 --kernel-trace --stats, round %s, build %s" % (rnd[1:], ver),
                     "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 5 --warmup 2 --cpu-sample 0",
                     "Workload: " + WORKLOAD + "; 7 layer calls (2 warmup + 5 timed). One `mma_nc_fused_fwd` / `mma_nc_fused_bwd` call = two "
                     "launches of the kernel (last template flag false: one work item per wavefront, long segments; true: one item per "
                     "32-lane group, short segments) + the hub finalize; bench.py's HIP-event time of a call is their sum. Un-profiled "
                     "default run of the same build: profiles/%s_bench_default.json." % rnd,
                     os.path.join(P, "%s_bench_kernel_stats_%s.md" % (rnd, ver)), os.path.join(P, "%s_bench_kernel_stats_%s.csv" % (rnd, ver)))
    print("C4 profiled step: %.2f ms" % ms)
    fd = os.path.dirname(newest("gpurun_out/prof_fetch/**/*counter_collection.csv"))
    wd = os.path.dirname(newest("gpurun_out/prof_write/**/*counter_collection.csv"))
    out = os.path.join(P, "%s_pmc_traffic.json" % rnd)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), fd, wd, out,
                           "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 5 --warmup 2 "
                           "--cpu-sample 0; MI355X, round %s, build %s" % (rnd[1:], ver)])
    d = json.load(open(out))
    d["workload"] = {"nodes": 1048576, "edges": 10864894, "hidden": 128, "K": 4}      # what bench.pmc_traffic matches on
    json.dump(d, open(out, "w"), indent=1)
    import bench
    line = [l for l in open(os.path.join(ROOT, "gpurun_out", "bench_default.log")) if l.startswith("{")][-1]
    b = json.loads(line)
    c = b["config"]
    b["roofline"]["traffic"] = bench.pmc_traffic(b["roofline"]["kernel"], c["nodes"], c["edges"], c["hidden"], c["K"])
    json.dump(b, open(os.path.join(P, "%s_bench_default.json" % rnd), "w"))
    print("default bench: %.2f ms/step, roofline frac %.3f, traffic %s" % (b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["traffic"]))
    try:
        g = newest("gpurun_out/prof_gr/**/*kernel_stats.csv")
    except SystemExit:
        return
    ms = stats_table(g, 35, "rocprofv3 --kernel-trace --stats, GR layer on the 10 000-graph batch (C2L), round %s, build %s" % (rnd[1:], ver),
                     "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gr -- python tools/gr_c2l_step.py",
                     "Workload: MMAConv 75->75, towers=5, edge_dim=50, aggregators [min,max], scalers [identity,amplification,linear] on 10 000 "
                     "ZINC-like molecules (204,552 nodes / 427,376 edges); 35 layer forward+backward calls.",
                     os.path.join(P, "%s_gr_c2l_kernel_stats_%s.md" % (rnd, ver)))
    print("C2L profiled step: %.2f ms" % ms)


if __name__ == "__main__":
    main()
