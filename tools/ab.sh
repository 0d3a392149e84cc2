#!/bin/bash
# A/B of two builds of liborbfe.so on the per-stage device times: tools/ab.sh <base.so> <out file> [batches...]
# (run on the GPU box; every step under its own time limit, a failing step ends the script)
set -e
base=$1; out=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for b in "${@:-512 64}"; do
  for n in $b; do
    ORBFE_LIB=$base timeout -k 10 120 python tools/stage_times.py $n 10 >> "$out" 2>&1
    timeout -k 10 120 python tools/stage_times.py $n 10 >> "$out" 2>&1
  done
done
grep crc "$out"
