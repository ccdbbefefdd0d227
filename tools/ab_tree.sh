#!/bin/bash
# tools/ab_tree.sh <tag> libA.so libB.so ...: the ordered count (--engine tree, config 4) with each library in turn, REPS times, on one box
TAG=$1; shift
mkdir -p gpurun_out/$TAG
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
  for L in "$@"; do
    N=$(basename $L .so)
    DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python bench.py --engine tree --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/${N}_$rep.json 2> gpurun_out/$TAG/${N}_$rep.err
    python - <<PY
import json
l=json.loads(open('gpurun_out/$TAG/${N}_$rep.json').read().strip().splitlines()[-1])
p=l['phases_ms']
print('$N rep $rep: %.2f ms  %.1f G digest_ok=%s' % (l['ms_per_step'], l['value']/1e9, l.get('digest_ok')), {k:round(v,2) for k,v in p.items() if v>0.5})
PY
  done
done
