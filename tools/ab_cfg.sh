#!/bin/bash
# tools/ab_cfg.sh <tag> "<bench args>" libA.so libB.so ...: one bench line (any config / variant) with each library in turn, REPS times, on one box
TAG=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/$TAG
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
  for L in "$@"; do
    N=$(basename $L .so)
    DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline > gpurun_out/$TAG/${N}_$rep.json 2> gpurun_out/$TAG/${N}_$rep.err || { echo "$N failed"; tail -2 gpurun_out/$TAG/${N}_$rep.err; continue; }
    python - <<PY
import json
l=json.loads(open('gpurun_out/$TAG/${N}_$rep.json').read().strip().splitlines()[-1])
p=l['phases_ms']
print('$N rep $rep [$ARGS]: %.2f ms  %.1f G ok=%s/%s' % (l['ms_per_step'], l['value']/1e9, l.get('digest_ok'), l.get('digest_total_ok')), {k:round(v,2) for k,v in p.items() if v>0.2})
PY
  done
done
