#!/bin/bash
N=${1:-3e9}
for lv in 3 0 2 3; do
  echo -n "== LEAVES_VARIANT=$lv : "
  DNAGPU_LEAVES_VARIANT=$lv timeout -k 5 120 python tools/count_once.py $N 31 3 2>&1 | tail -1 | tr ',' '\n' | grep -A1 -E "leaves" | tr '\n' ' '; echo
done
