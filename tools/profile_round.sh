#!/bin/bash
# Collect the raw rocprofv3 output behind profiles/ on the GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'   then, here:   python tools/make_profiles.py r01_v4
# Passes are separate on purpose: --kernel-trace --stats alone, then one --pmc pass per counter (FETCH_SIZE takes 3 of the
# 4 TCC slots), then the un-profiled default bench, then the GR layer on the 10 000-molecule batch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 5 --warmup 2 --cpu-sample 0"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_gr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B > gpurun_out/prof_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- $B > gpurun_out/prof_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- $B > gpurun_out/prof_write.log 2>&1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gr -- python tools/gr_c2l_step.py > gpurun_out/prof_gr.log 2>&1 || exit 1
tail -c 600 gpurun_out/bench_default.log
