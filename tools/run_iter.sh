#!/bin/bash
# tools/run_iter.sh <tag> [fuzz cases]: one development iteration on the GPU box -- the unordered engine's parity tests, a fuzz
# soak, then bench lines (headline, configs 3 and 2, one repeat-rich variant).  Every step only if the one before passed.
TAG=${1:-iter}; FZ=${2:-300}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x -k "unordered or superkmer or records or level0 or level1 or smoke or config or batch or merge" > $O/pytest.log 2>&1; rc=$?
echo "rc=$rc" >> $O/pytest.log; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 tools/fuzz_unordered.py $FZ > $O/fuzz.log 2>&1; rc=$?; tail -2 $O/fuzz.log
[ $rc -eq 0 ] || exit 1
B="timeout -k 10 300 python3 bench.py --no-cpu-baseline"
for cfg in "" "--config 3" "--config 2" "--motif 1000" "--config 3 --motif 1000"; do
  n=$(echo "bench$cfg" | tr -d ' -')
  $B $cfg > $O/$n.json 2> $O/$n.err || { echo "bench $cfg failed"; tail -3 $O/$n.err; exit 1; }
  python3 - <<PY
import json
l=json.loads(open('$O/$n.json').read().strip().splitlines()[-1])
p=l['phases_ms']
print('$n: %.2f ms  %.1f G  digest_ok=%s' % (l['ms_per_step'], l['value']/1e9, l.get('digest_ok')), {k:round(v,2) for k,v in p.items() if v>0.15})
PY
done
