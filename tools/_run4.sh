#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03n
cd $GRAFT_REPO_ROOT
for cfg in 2 3; do
timeout -k 10 300 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r03n/bench_cfg$cfg.json 2> gpurun_out/r03n/err.log
python - <<PY
import json
l=json.loads(open('gpurun_out/r03n/bench_cfg$cfg.json').read().strip().splitlines()[-1])
print('cfg$cfg', round(l['ms_per_step'],3),'ms', round(l['value']/1e9,1),'G', l['phases_ms'])
PY
done
timeout -k 10 300 python bench.py --config 2 --engine tree --steps 50 --warmup 5 --no-cpu-baseline | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 tree', round(l['ms_per_step'],3),'ms', round(l['value']/1e9,1),'G', l['phases_ms'])"
