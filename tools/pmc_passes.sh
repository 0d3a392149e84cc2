#!/bin/bash
# The four rocprofv3 --pmc passes behind profiles/<round>_pmc_summary.txt, each a run of its own with its own time limit
# (run on the GPU box): tools/pmc_passes.sh <out dir>
set -e
out=$1; mkdir -p "$out"
export TMPDIR=/tmp
pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out" -o "$name" -- python3 tools/stage_times.py 512 3 > "$out/$name.log" 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
# the Hamming kNN-2 call (2000 x 1M, tools/knn_time.py): instruction counts and busy cycles of its kernels
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d "$out" -o knn -- python3 tools/knn_time.py > "$out/knn.log" 2>&1
python3 tools/pmc_summary.py "$out" > "$out/summary.txt"
