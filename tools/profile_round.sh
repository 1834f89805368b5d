#!/bin/bash
# Collect the raw material behind profiles/ on the GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'   then, here:   python tools/make_profiles.py r02 v4
# Passes are separate on purpose (MI355X_MICROARCH.md): --kernel-trace --stats alone, then one --pmc pass per TCC counter
# (FETCH_SIZE takes 3 of the 4 TCC slots) - for the headline bench (C4) and for the GR layer on the 10 000-molecule batch
# (C2L) - then the two un-profiled default bench lines and the GEMMs on random vs zero operands (tools/gemm_power.py).
# The C4 round includes the SQ counters and the effective-clock pass (GRBM_GUI_ACTIVE).
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof_c4.sh round sq || exit 1
bash tools/prof_c2l.sh round || exit 1
bash tools/prof_small.sh round || exit 1
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --workload c2l > gpurun_out/bench_c2l.log 2>&1 || exit 1
timeout -k 10 200 python tools/gemm_power.py > gpurun_out/gemm_power.log 2>&1 || exit 1
tail -c 400 gpurun_out/bench_default.log
