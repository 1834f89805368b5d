"""One small-graph configuration of bench.py (C1 / C3 layer replays, C2 layer replay, C2net graphed Net step) on its own, for
`tools/prof_small.sh` (rocprofv3 --kernel-trace of the hipGraph replays):   python tools/small_replay.py C1|C3|C2|C2net"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C1"
    dev = torch.device("cuda:0")
    if which == "C1":
        r = bench.nc_config("C1", "cora_h64", 64, ["mean", "mean2"], 7, 0.75, dev, reps=5)
    elif which == "C3":
        r = bench.nc_config("C3", "pubmed_h16", 16, ["min", "min2", "min3", "min4"], 3, 0.5, dev, reps=5)
    elif which == "C2":
        r = bench.gr_config("C2", 64, dev, reps=5)
    elif which == "C2net":
        r = bench.gr_model_config("C2net", 64, dev, reps=5, graphed=True)
    else:
        raise SystemExit("unknown configuration %r" % which)
    print(which, "replay ms", r["ms_per_step_hipgraph"], "eager ms", r.get("ms_per_step_eager"))


if __name__ == "__main__":
    main()
