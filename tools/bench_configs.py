#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations that are NOT the bench.py headline (launch-bound on any GPU; one JSON line
each) - the same entries bench.py reports under `extra`:
  C1  Cora structure, H=64, aggregators mean,mean2, p=0.75      (tests/golden/cora_h64.npz holds the CSR)
  C3  Pubmed structure, H=16, aggregators min,min2,min3,min4    (tests/golden/pubmed_h16.npz)
  C2  ZINC-like molecule batch (64 graphs), MMAConv 75->75, towers=5, edge_dim=50, min,max x identity,amplification,linear
  C2L the same layer on a 10 000-graph batch (the whole ZINC-subset training split in one batch)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    import torch
    for v in bench.extra_configs(torch.device("cuda", 0)).values():
        print(json.dumps(v), flush=True)
