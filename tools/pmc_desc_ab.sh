#!/bin/bash
# counters of the descriptor kernels (run on the GPU box): tools/pmc_desc_ab.sh <out dir> <tiled 0|1>
out=$1; v=$2; mkdir -p "$out"
export TMPDIR=/tmp ORBFE_DESCRIBE_TILED=$v
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out" -o "t${v}_$name" -- python3 tools/stage_times.py 512 3 > "$out/t${v}_$name.log" 2>&1 || echo "pass $name failed"
}
pass sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "t${v}_trace" -- python3 tools/stage_times.py 512 3 > "$out/t${v}_trace.log" 2>&1
python3 tools/pmc_summary.py "$out" describe
python3 tools/pmc_summary.py "$out" bucket
grep -h "describe\|bucket" "$out"/t${v}_trace_kernel_stats.csv
