#!/bin/bash
# rocprofv3 passes over the dense products of the layer step on their own (tools/gemm_micro.py: the C4 shape and the C5 shard shape),
# run on the GPU box through gpurun:   gpurun --timeout 900 -- 'bash tools/prof_gemm.sh [tag]'
# Round-3 VERDICT item 1a: PMC + SQ passes of the C5-shape GEMMs, not just C4.  Separate passes on purpose (MI355X_MICROARCH.md):
# --kernel-trace --stats alone, one --pmc pass per TCC counter (FETCH_SIZE takes 3 of the 4 TCC slots), the SQ counters, the clock.
# Output: gpurun_out/prof_gemm_<tag>_{stats,fetch,write,sq,clk}/ + _summary.json (tools/prof_summary.py), _micro.log (un-profiled table)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 tools/gemm_micro.py --shape both --rounds 2 --reps 3"
O=gpurun_out/prof_gemm_$TAG
python tools/build_stamp.py > ${O}_stamp.json
rm -rf ${O}_stats ${O}_fetch ${O}_write ${O}_sq ${O}_clk
timeout -k 10 200 python tools/gemm_micro.py > ${O}_micro.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats -- $B > ${O}_stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -- $B > ${O}_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -- $B > ${O}_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d ${O}_sq -- $B > ${O}_sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${O}_clk -- $B > ${O}_clk.log 2>&1 || exit 1
find ${O}_stats ${O}_fetch ${O}_write ${O}_sq ${O}_clk -type f ! -name '*.csv' -delete 2>/dev/null
python tools/prof_summary.py ${O}
