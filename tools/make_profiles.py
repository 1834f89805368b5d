#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/prof_c4_round_*, prof_c2l_round_*, bench_default.log,
bench_c2l.log, parity_strict_report_gpu.jsonl) into the committed summaries under profiles/:

    python tools/make_profiles.py <round> <version>          # e.g. r02 v3

  profiles/<round>_bench_kernel_stats_<v>.{md,csv}   per-kernel time of `bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra` (C4)
  profiles/<round>_pmc_traffic.json                  C4: HBM traffic per kernel launch (FETCH_SIZE / WRITE_SIZE passes)
  profiles/<round>_gr_c2l_kernel_stats_<v>.{md,csv}  per-kernel time of `bench.py --workload c2l --steps 5 --warmup 2 --cpu-sample 0`
  profiles/<round>_gr_pmc_traffic.json               C2L: HBM traffic per kernel launch
  profiles/<round>_bench_default.json / _bench_c2l.json   the un-profiled bench lines of the same build (traffic looked up again)
  profiles/<round>_parity_strict_report.md           strict-bar (1e-5 + 1e-5|ref|) failure counts of every -m gpu comparison
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = os.path.join(ROOT, "profiles")
C4 = "C4 R-MAT 1,048,576 nodes / 10,864,894 directed edges, H=128, K=4 [sum,mean,max,min], p=0.5"
C2L = ("MMAConv 75->75, towers=5, edge_dim=50, aggregators [min,max], scalers [identity,amplification,linear] on 10 000 ZINC-like "
       "molecules (204,552 nodes / 427,376 edges)")


def find(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def stats_table(path, calls, title, command, notes, out_md, out_csv, top=28):
    rows = list(csv.DictReader(open(path)))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(out_md, "w") as f:
        f.write("# %s\n\nCommand (MI355X box): `%s`\n\n" % (title, command))
        f.write(notes + " All GPU kernels: %.2f ms per step (%d steps incl. warm-up).\n\n" % (total / calls / 1e6, calls))
        f.write("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:top]:
            f.write("| `%s` | %s | %.3f | %.4f | %s |\n" % (r["Name"][:110], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e6, r["Percentage"]))
    shutil.copy(path, out_csv)
    return total / calls / 1e6


def traffic(base, out, workload, note):
    def per_kernel(d, counter):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(find(d + "/**/*counter_collection.csv"))):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
    F, W = per_kernel(base + "_fetch", "FETCH_SIZE"), per_kernel(base + "_write", "WRITE_SIZE")
    stamp_file = os.path.join(ROOT, base + "_stamp.json")
    from tools.build_stamp import build_stamp
    stamp = json.load(open(stamp_file)) if os.path.exists(stamp_file) and os.path.getsize(stamp_file) else build_stamp()
    # the average duration of every kernel in the --kernel-trace --stats pass of the same command: bench.py's roofline.frac_rocprof
    avg_ns = {}
    try:
        for r in csv.DictReader(open(find(base + "_stats/**/*kernel_stats.csv"))):
            avg_ns[r["Name"]] = float(r["AverageNs"])
    except SystemExit:
        pass
    res = {"note": note, "workload": workload, "build": stamp,
           "correction": "traffic_bytes = 2*FETCH_SIZE_KiB*1024 + WRITE_SIZE_KiB*1024 (gfx950: FETCH_SIZE halves wide reads; MI355X_MICROARCH.md HBM)",
           "kernels": {}}
    for k in sorted(F):
        if "mma::" in k:
            n, f = F[k]
            w = W.get(k, (0, 0.0))[1]
            res["kernels"][k] = {"launches_sampled": n, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes": 2 * f * 1024 + w * 1024,
                                 "rocprof_avg_ns": avg_ns.get(k)}
    json.dump(res, open(out, "w"), indent=1)
    return res


def bench_line(log, out):
    import bench
    # the log holds stdout + stderr: the verbose {"detail": ...} record (stderr) and the compact metric line (stdout, last)
    lines = [l for l in open(os.path.join(ROOT, "gpurun_out", log)) if l.startswith("{")]
    det = [l for l in lines if l.startswith('{"detail"')]
    b = json.loads(det[-1])["detail"] if det else json.loads(lines[-1])
    b["compact_line"] = json.loads(lines[-1]) if det else None
    c = b["config"]
    for key in ("roofline", "roofline_other"):
        r = b.get(key)
        if r:
            wl = {"workload": "c2l", "nodes": c["nodes"], "edges": c["edges"]} if "towers" in c else \
                {"nodes": c["nodes"], "edges": c["edges"], "hidden": c["hidden"], "K": c["K"]}
            t, src, rms = bench.pmc_traffic(r["kernel"], wl)
            r["traffic"], r["traffic_source"] = t, bench._traffic_source(t, src)
            if rms and r.get("algorithmic_bytes"):
                r["frac_rocprof"] = r["algorithmic_bytes"] / (rms * 1e-3) / 1e9 / r["peak"]
    json.dump(b, open(out, "w"))
    return b


def parity_report(out):
    """One row per COMPARISON, keyed by its full name, with the bar that comparison asserted (round-2 VERDICT: the round-2 table
    merged forward outputs - asserted strict - with gradients of the same aggregator name under the first bar seen)."""
    src = os.path.join(ROOT, "gpurun_out", "parity_strict_report_gpu.jsonl")
    if not os.path.exists(src):
        return
    rows = [json.loads(l) for l in open(src)]
    bad = [r for r in rows if r["strict_outside"]]
    noise_c = max([float(r["bar"].split("+")[-1].split("*")[0]) for r in rows if "rowmax" in r["bar"]] or [0.0])

    def is_forward(r):          # aggregator outputs m_k, layer outputs, log-probabilities: everything that is not a gradient
        w = r["what"]
        leaf = w.split("/")[-1] if "/" in w else w
        return not (leaf.startswith("g") or "/g" in w or "grad" in w)
    fwd_agg = [r for r in rows if r["bar"] == "strict"]
    with open(out, "w") as f:
        f.write("# Strict-bar report of the `-m gpu` parity comparisons\n\nEvery comparison of the HIP path with a golden vector or the CPU oracle "
                "counts the elements outside the STRICT bar `|got - want| <= 1e-5 + 1e-5 |want|` (tests/golden_util.py), whatever bar it "
                "asserts. %d comparisons, %d elements; %d comparisons have elements outside the strict bar, %d elements in all (%.4f %%). "
                "`need` = the largest multiple of the reference's own fp32 noise (rowmax |reference - float64 oracle|) any element "
                "needs on top of the strict bar (asserted: %g).\n\n" % (len(rows), sum(r["n"] for r in rows), len(bad),
                                                                     sum(r["strict_outside"] for r in bad),
                                                                     100.0 * sum(r["strict_outside"] for r in bad) / max(1, sum(r["n"] for r in rows)), noise_c))
        f.write("**Comparisons ASSERTED at the strict bar: %d (%d elements), outside strict: %d.** These are the forward aggregator outputs "
                "(`m/<aggregator>`, per-aggregator `learnable_*` results, selection codes) and every other quantity that is not a long signed sum.\n\n"
                % (len(fwd_agg), sum(r["n"] for r in fwd_agg), sum(r["strict_outside"] for r in fwd_agg)))
        gold_fwd = [r for r in rows if (r["what"].startswith("set/") and "/m/" in r["what"]) or r["what"].startswith("single/")]
        f.write("**Forward aggregator outputs on the reference goldens (`set/<act>/<p>/<aggregators>/m/<a>` and `single/...`): %d comparisons, "
                "%d elements, outside strict: %d.**\n\n" % (len(gold_fwd), sum(r["n"] for r in gold_fwd), sum(r["strict_outside"] for r in gold_fwd)))
        f.write("Largest noise multiple needed by any comparison: %.2f.\n\n" % max([r.get("noise_multiple_needed") or 0.0 for r in rows] or [0.0]))
        f.write("## Comparisons with elements outside the strict bar (each asserts the bar in its last column)\n\n")
        f.write("| comparison | elements | outside strict | max err | max ref | need | asserted bar |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(bad, key=lambda r: -r["strict_outside"]):
            f.write("| %s | %d | %d | %.3g | %.3g | %.2f | %s |\n" % (r["what"], r["n"], r["strict_outside"], r["max_err"], r["max_ref"],
                                                                    r.get("noise_multiple_needed") or 0.0, r["bar"]))
        f.write("\n## All comparisons, grouped by asserted bar and test family\n\n| family | asserted bar | comparisons | elements | outside strict | max need |\n|---|---|---|---|---|---|\n")
        fam = collections.OrderedDict()
        for r in rows:
            w = r["what"]
            family = w.split(":")[0] if ":" in w else (w.split("/")[0] if "/" in w else w)
            a = fam.setdefault((family, r["bar"]), [0, 0, 0, 0.0])
            a[0] += 1; a[1] += r["n"]; a[2] += r["strict_outside"]; a[3] = max(a[3], r.get("noise_multiple_needed") or 0.0)
        for (family, bar), a in fam.items():
            f.write("| %s | %s | %d | %d | %d | %.2f |\n" % (family, bar, a[0], a[1], a[2], a[3]))


def main():
    rnd, ver = sys.argv[1], sys.argv[2]
    cmd4 = "python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra"
    ms = stats_table(find("gpurun_out/prof_c4_round_stats/**/*kernel_stats.csv"), 7,
                     "rocprofv3 --kernel-trace --stats, round %s, build %s" % (rnd[1:], ver),
                     "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd4,
                     "Workload: " + C4 + ". One `mma_nc_fused_fwd` / `mma_nc_fused_bwd` call = two launches of the kernel (last template "
                     "flag false: one work item per wavefront, long segments; true: one item per 32-lane group, short segments) + the hub "
                     "finalize; bench.py's HIP-event time of a call is their sum. Un-profiled default run of the same build: profiles/%s_bench_default.json." % rnd,
                     os.path.join(P, "%s_bench_kernel_stats_%s.md" % (rnd, ver)), os.path.join(P, "%s_bench_kernel_stats_%s.csv" % (rnd, ver)))
    print("C4 profiled step: %.2f ms" % ms)
    traffic("gpurun_out/prof_c4_round", os.path.join(P, "%s_pmc_traffic.json" % rnd), {"nodes": 1048576, "edges": 10864894, "hidden": 128, "K": 4},
            "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- %s; MI355X, round %s, build %s" % (cmd4, rnd[1:], ver))
    cmd2 = "python bench.py --workload c2l --steps 5 --warmup 2 --cpu-sample 0"
    ms = stats_table(find("gpurun_out/prof_c2l_round_stats/**/*kernel_stats.csv"), 7,
                     "rocprofv3 --kernel-trace --stats, GR layer on the 10 000-molecule batch (C2L), round %s, build %s" % (rnd[1:], ver),
                     "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd2, "Workload: " + C2L + ".",
                     os.path.join(P, "%s_gr_c2l_kernel_stats_%s.md" % (rnd, ver)), os.path.join(P, "%s_gr_c2l_kernel_stats_%s.csv" % (rnd, ver)))
    print("C2L profiled step: %.2f ms" % ms)
    traffic("gpurun_out/prof_c2l_round", os.path.join(P, "%s_gr_pmc_traffic.json" % rnd), {"workload": "c2l", "nodes": 204552, "edges": 427376},
            "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- %s; MI355X, round %s, build %s" % (cmd2, rnd[1:], ver))
    b = bench_line("bench_default.log", os.path.join(P, "%s_bench_default.json" % rnd))
    print("default bench: %.2f ms/step, roofline %s frac %.3f, traffic %s" % (b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline"]["traffic"]))
    b = bench_line("bench_c2l.log", os.path.join(P, "%s_bench_c2l.json" % rnd))
    print("c2l bench: %.2f ms/step, roofline %s frac %.3f / %.3f" % (b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline_other"]["frac"]))
    if os.path.isdir(os.path.join(ROOT, "gpurun_out", "prof_small_round")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "small_summary.py"), os.path.join(ROOT, "gpurun_out", "prof_small_round"),
                        os.path.join(P, "%s_small_graph_kernels.md" % rnd)], check=True)
    parity_report(os.path.join(P, "%s_parity_strict_report.md" % rnd))
    sq_clock(os.path.join(P, "%s_sq_clock.json" % rnd), rnd, ver, cmd4)
    gp = os.path.join(ROOT, "gpurun_out", "gemm_power.log")
    if os.path.exists(gp):
        with open(os.path.join(P, "%s_gemm_power.md" % rnd), "w") as f:
            f.write("# bf16x3 GEMMs on random vs zero operands, round %s, build %s\n\nCommand (MI355X box): `python tools/gemm_power.py`\n\n" % (rnd[1:], ver))
            f.write("".join(l for l in open(gp) if not l.startswith("/opt/amdgpu")))


def sq_clock(out, rnd, ver, cmd):
    """SQ counters and the effective clock per mma:: kernel of the C4 bench (tools/prof_c4.sh <tag> sq -> prof_summary.py)."""
    src = os.path.join(ROOT, "gpurun_out", "prof_c4_round_summary.json")
    if not os.path.exists(src):
        return
    d = json.load(open(src))
    res = {"note": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES "
                   "SQ_BUSY_CYCLES and, in its own pass, --pmc GRBM_GUI_ACTIVE -- %s; MI355X, round %s, build %s.  clock_GHz = GRBM_GUI_ACTIVE / 8 XCDs / "
                   "kernel time of that pass (MI355X_MICROARCH.md, DVFS give-back); wait / issue_stall / active = fractions of SQ_WAVE_CYCLES" % (cmd, rnd[1:], ver),
           "kernels": {}}
    for k, e in d.items():
        if not e.get("SQ_WAVE_CYCLES"):
            continue
        wc = e["SQ_WAVE_CYCLES"]
        res["kernels"][k] = {"avg_us": e["avg_us"], "clock_GHz": e.get("clock_GHz"), "wait": e["SQ_WAIT_ANY"] / wc,
                             "issue_stall": e["SQ_WAIT_INST_ANY"] / wc, "active": e["SQ_ACTIVE_INST_ANY"] / wc,
                             "valu_per_wave": e["SQ_INSTS_VALU"] / max(e["SQ_WAVES"], 1),
                             "mfma_busy_over_sq_busy": (e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]) if e.get("SQ_VALU_MFMA_BUSY_CYCLES") else None}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
