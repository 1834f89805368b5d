#!/usr/bin/env python3
"""The GR layer (MMAConv 75->75, towers=5, edge_dim=50, [min,max] x [identity,amplification,linear]) forward+backward on a
10 000-molecule batch - the program tools/profile_round.sh puts under rocprofv3 for profiles/*_gr_c2l_*
(= `python bench.py --workload c2l --steps 5 --warmup 2 --cpu-sample 0`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    sys.argv = [sys.argv[0], "--workload", "c2l", "--steps", "5", "--warmup", "2", "--cpu-sample", "0"]
    bench.main()
