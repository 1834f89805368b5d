#!/bin/bash
# Kernel traces of the small-graph hipGraph replays (C1 Cora layer, C3 PubMed layer, C2 ZINC-batch layer, C2net graphed Net step),
# run on the GPU box through gpurun:   gpurun --timeout 900 -- 'bash tools/prof_small.sh [tag]'
# then here:   python tools/small_summary.py gpurun_out/prof_small_<tag> profiles/rNN_small_graph_kernels.md
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_small_$TAG
rm -rf $O && mkdir -p $O
python tools/build_stamp.py > $O/stamp.json
for c in C1 C3 C2 C2net; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$c -- python3 tools/small_replay.py $c > $O/$c.log 2>&1 || exit 1
  grep "replay ms" $O/$c.log
done
find $O -type f ! -name '*kernel_trace.csv' ! -name '*.log' ! -name 'stamp.json' -delete 2>/dev/null
exit 0
