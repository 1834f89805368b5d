#!/usr/bin/env python3
"""Summarise the rocprofv3 passes tools/prof_*.sh wrote: per mma:: kernel the average duration (kernel trace), HBM traffic
(2*FETCH_SIZE + WRITE_SIZE, KiB counters, gfx950 correction of MI355X_MICROARCH.md) and the SQ counters per launch.
    python tools/prof_summary.py gpurun_out/prof_c2l_<tag> [out.json]"""
import collections
import csv
import glob
import json
import sys


def find(d, pat):
    f = glob.glob(d + "/**/*" + pat, recursive=True)
    return f[0] if f else None


def trace(d):
    f = find(d, "kernel_trace.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg


def counters(d):
    f = find(d, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    base = sys.argv[1]
    t = trace(base + "_stats")
    cf, cw, cs = counters(base + "_fetch"), counters(base + "_write"), counters(base + "_sq")
    ck, tk = counters(base + "_clk"), trace(base + "_clk")
    res = {}
    for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
        if "mma::" not in k:
            continue
        short = k.split("(")[0]
        avg = lambda xs: sum(xs) / len(xs) if xs else None
        e = {"launches": len(v), "avg_us": avg(v), "total_us": sum(v)}
        f, w = avg(cf[k].get("FETCH_SIZE", [])), avg(cw[k].get("WRITE_SIZE", []))
        if f is not None and w is not None:
            e.update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, traffic_bytes=2 * f * 1024 + w * 1024)
        for c, xs in cs[k].items():
            e[c] = avg(xs)
        gui = avg(ck[k].get("GRBM_GUI_ACTIVE", []))
        if gui and tk.get(k):          # cycles summed over the 8 XCDs / the duration seen in the SAME pass
            e["clock_GHz"] = gui / 8 / (avg(tk[k]) * 1e3)
        res[short] = e
    out = sys.argv[2] if len(sys.argv) > 2 else base + "_summary.json"
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res.items():
        line = "%-62s n=%3d avg %9.1f us" % (k[:62], e["launches"], e["avg_us"])
        if "traffic_bytes" in e:
            line += "  traffic %.3f GB (rd %.3f wr %.3f)" % (e["traffic_bytes"] / 1e9, 2 * e["FETCH_SIZE_KiB"] * 1024 / 1e9, e["WRITE_SIZE_KiB"] * 1024 / 1e9)
        if e.get("clock_GHz"):
            line += "  clk %.2f GHz" % e["clock_GHz"]
        if e.get("SQ_WAVE_CYCLES"):
            line += "  valu/wave %.0f wait %.2f act %.2f" % (e["SQ_INSTS_VALU"] / max(e["SQ_WAVES"], 1), e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"],
                                                           e["SQ_ACTIVE_INST_ANY"] / e["SQ_WAVE_CYCLES"])
            if e.get("SQ_WAIT_INST_ANY") is not None:
                line += " inst-stall %.2f" % (e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"])
            if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("SQ_BUSY_CYCLES"):
                # MFMA busy counts cycles summed over SIMDs; SQ_BUSY_CYCLES sums over the XCDs' SQs (MI355X_MICROARCH.md): relative only
                line += " mfma_busy/sq_busy %.3f" % (e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"])
        print(line)


if __name__ == "__main__":
    main()
