#!/bin/bash
# rocprofv3 passes over the headline bench (`bench.py`, C4), run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'bash tools/prof_c4.sh [tag] [sq]'
# Separate passes on purpose (MI355X_MICROARCH.md): --kernel-trace --stats alone, then one --pmc pass per TCC counter
# (FETCH_SIZE takes 3 of the 4 TCC slots); `sq` adds a pass with the SQ wait / MFMA-busy counters.
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extra"
O=gpurun_out/prof_c4_$TAG
python tools/build_stamp.py > ${O}_stamp.json      # which kernel build these passes belong to (bench.py refuses a stale traffic figure)
rm -rf ${O}_stats ${O}_fetch ${O}_write ${O}_sq ${O}_clk
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats -- $B > ${O}_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -- $B > ${O}_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -- $B > ${O}_write.log 2>&1 || exit 1
if [ "$2" = "sq" ]; then
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d ${O}_sq -- $B > ${O}_sq.log 2>&1 || exit 1
  # effective clock under load = GRBM_GUI_ACTIVE / 8 XCDs / kernel time (MI355X_MICROARCH.md "DVFS give-back"), its own pass
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${O}_clk -- $B > ${O}_clk.log 2>&1 || exit 1
fi
find ${O}_stats ${O}_fetch ${O}_write ${O}_sq ${O}_clk -type f ! -name '*.csv' -delete 2>/dev/null
python tools/prof_summary.py ${O}
