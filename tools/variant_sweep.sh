#!/bin/bash
N=${1:-1e9}
for sv in 0 1; do
  echo -n "== SCATTER_VARIANT=$sv : "
  DNAGPU_SCATTER_VARIANT=$sv timeout -k 5 120 python tools/phase_probe.py $N 2>&1 | grep -E "k=31|scatter|rror" | head -3 | tr '\n' ' '; echo
done
