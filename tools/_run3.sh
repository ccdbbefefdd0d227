#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03m
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter or contains or config5" > gpurun_out/r03m/pytest_filter.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03m/pytest_filter.log
tail -4 gpurun_out/r03m/pytest_filter.log
for pat in NNNNNNNNNNWSNNNNNNNNN ACGNNNNNNNNNNNNNNNNNN; do
for rep in 1 2; do
timeout -k 10 300 python bench.py --config 5 --pattern $pat --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03m/bench_cfg5_${pat}_$rep.json 2> gpurun_out/r03m/err.log
python - <<PY
import json
l=json.loads(open('gpurun_out/r03m/bench_cfg5_${pat}_$rep.json').read().strip().splitlines()[-1])
print('$pat', round(l['ms_per_step']*1e3,1),'us', round(l['value']/1e12,3),'T rows/s', l['phases_ms'], l['roofline']['frac'])
PY
done
done
timeout -k 10 300 python bench.py --config 5 --n-bases 1e9 --steps 50 --warmup 5 --no-cpu-baseline | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1G', round(l['ms_per_step'],3),'ms', round(l['value']/1e12,3),'T', l['phases_ms'])"
