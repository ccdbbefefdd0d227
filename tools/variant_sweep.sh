#!/bin/bash
N=${1:-1e9}
for lv in 0 1 2 3; do for gm in 1 2 4; do
  echo -n "== LEAVES_VARIANT=$lv GRIDMULT=$gm : "
  DNAGPU_LEAVES_VARIANT=$lv DNAGPU_LEAVES_GRIDMULT=$gm timeout -k 5 120 python tools/phase_probe.py $N 2>&1 | grep -E "leaves|rror" | head -2 | tr '\n' ' '; echo
done; done
for dbg in 1 2 4 7; do
  echo -n "== VARIANT=1 DEBUG=$dbg : "
  DNAGPU_LEAVES_VARIANT=1 DNAGPU_DEBUG_LEAVES=$dbg timeout -k 5 120 python tools/phase_probe.py $N 2>&1 | grep -E "leaves|rror" | head -2 | tr '\n' ' '; echo
done
