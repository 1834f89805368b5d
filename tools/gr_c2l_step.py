#!/usr/bin/env python3
"""The GR layer (MMAConv 75->75, towers=5, edge_dim=50, [min,max] x [identity,amplification,linear]) forward+backward on a
10 000-molecule batch - the program tools/profile_round.sh puts under rocprofv3 for profiles/*_gr_c2l_kernel_stats_*."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402

bench_configs.gr_config("C2L", 10000, reps=5)
