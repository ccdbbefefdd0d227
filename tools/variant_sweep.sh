#!/bin/bash
N=${1:-3e9}
for sv in 0 1 0; do
  echo -n "== SCATTER_VARIANT=$sv : "
  DNAGPU_SCATTER_VARIANT=$sv timeout -k 5 120 python tools/count_once.py $N 31 4 2>&1 | tail -1 | cut -c1-400
done
