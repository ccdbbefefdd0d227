#!/bin/bash
# tools/run_quick.sh <tag> [bench args...]: a short check on the GPU box -- a few parity tests of the unordered engine, a short
# fuzz soak, the headline bench line (+ config 3 and 2).  Every step only if the one before passed.
TAG=${1:-quick}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "test_count_unordered_superkmers or level0_slabs or level1_spec or level1_sampled or repeat or near_copies" > $O/pytest.log 2>&1; rc=$?
echo "rc=$rc" >> $O/pytest.log; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 tools/fuzz_unordered.py 120 > $O/fuzz.log 2>&1; rc=$?; tail -1 $O/fuzz.log
[ $rc -eq 0 ] || exit 1
B="timeout -k 10 300 python3 bench.py --no-cpu-baseline"
for cfg in "" "--config 3" "--config 2" "$@"; do
  n=$(echo "bench$cfg" | tr -d ' -')
  $B $cfg > $O/$n.json 2> $O/$n.err || { echo "bench $cfg failed"; tail -3 $O/$n.err; exit 1; }
  python3 - <<PY
import json
l=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
p=l['phases_ms']
print('$n: %.2f ms  %.1f G  digest_ok=%s' % (l['ms_per_step'], l['value']/1e9, l.get('digest_ok')), {k:round(v,2) for k,v in p.items() if v>0.12})
PY
done
