#!/bin/bash
# A/B counters of the FAST kernel variants (run on the GPU box): tools/pmc_fast_ab.sh <out dir> <variant>
out=$1; v=$2; mkdir -p "$out"
export TMPDIR=/tmp ORBFE_FAST_VARIANT=$v
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out" -o "v${v}_$name" -- python3 tools/stage_times.py 512 3 > "$out/v${v}_$name.log" 2>&1 || echo "pass $name failed"
}
pass sq1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
python3 tools/pmc_summary.py "$out" fast_cells
