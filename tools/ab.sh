#!/bin/bash
# tools/ab.sh <tag> libA.so libB.so ...: the headline bench (config 4) with each library in turn, twice, on ONE box
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
  for L in "$@"; do
    N=$(basename $L .so)
    DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/${N}_$rep.json 2> gpurun_out/$TAG/${N}_$rep.err
    python - <<PY
import json
l=json.loads(open('gpurun_out/$TAG/${N}_$rep.json').read().strip().splitlines()[-1])
p=l['phases_ms']
print('$N rep $rep: %.2f ms  %.1f G' % (l['ms_per_step'], l['value']/1e9), {k:round(v,2) for k,v in p.items() if v>0.3})
PY
  done
done
