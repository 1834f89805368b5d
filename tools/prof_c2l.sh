#!/bin/bash
# rocprofv3 passes over `bench.py --workload c2l` (the GR layer on the 10 000-molecule batch), run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'bash tools/prof_c2l.sh [tag]'
# Separate passes on purpose (MI355X_MICROARCH.md): --kernel-trace --stats alone, then one --pmc pass per TCC counter
# (FETCH_SIZE takes 3 of the 4 TCC slots), then the SQ counters.  Output: gpurun_out/prof_c2l_<tag>_{stats,fetch,write,sq}/
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --workload c2l --steps 5 --warmup 2 --cpu-sample 0"
O=gpurun_out/prof_c2l_$TAG
python tools/build_stamp.py > ${O}_stamp.json      # which kernel build these passes belong to (bench.py refuses a stale traffic figure)
rm -rf ${O}_stats ${O}_fetch ${O}_write ${O}_sq
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats -- $B > ${O}_stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -- $B > ${O}_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -- $B > ${O}_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d ${O}_sq -- $B > ${O}_sq.log 2>&1 || exit 1
# keep what is merged back small: drop everything but the csv files
find ${O}_stats ${O}_fetch ${O}_write ${O}_sq -type f ! -name '*.csv' -delete
python tools/prof_summary.py ${O}
