#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY as MI355X_MICROARCH.md prescribes:
FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2) into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [note]

Units/corrections (MI355X_MICROARCH.md "HBM"): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half the
bytes of a wide (16 B/lane) coalesced read, which is how the fused kernels read their rows, so it is doubled;
WRITE_SIZE reads the bytes exactly for 16-B-per-lane stores.   traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {"note": note, "correction": "traffic_bytes = 2*FETCH_SIZE_KiB*1024 + WRITE_SIZE_KiB*1024 (gfx950: FETCH_SIZE halves wide reads)",
           "kernels": {}}
    for k in sorted(F):
        if "mma::" not in k:
            continue
        n, f = F[k]
        w = W.get(k, (0, 0.0))[1]
        res["kernels"][k] = {"launches_sampled": n, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                             "traffic_bytes": 2 * f * 1024 + w * 1024}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-80s traffic %.2f GB" % (k[:80], v["traffic_bytes"] / 1e9))


if __name__ == "__main__":
    main()
