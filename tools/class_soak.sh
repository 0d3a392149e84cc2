#!/bin/bash
# Soak of the C++ class tests (ORBmatcher / ORBextractor adapters through tests/cpp/*_main) over shifted generator seeds:
#   tools/class_soak.sh <first seed> <last seed> <log>      (on the GPU box; one pytest process per seed, stops at the first failure)
set -e
for s in $(seq "$1" "$2"); do
  echo "== ORBFE_CLASS_SEED=$s" >> "$3"
  ORBFE_CLASS_SEED=$s timeout -k 10 300 python -u -m pytest tests/test_matcher_class.py tests/test_guided_search.py -x -q --timeout 120 --timeout-method thread >> "$3" 2>&1
  tail -1 "$3"
done
