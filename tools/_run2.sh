#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03b
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unordered or records or multi_unordered" > gpurun_out/r03b/pytest_sk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03b/pytest_sk.log
tail -4 gpurun_out/r03b/pytest_sk.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03b/bench_n1.json 2> gpurun_out/r03b/bench_n1.err && python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03b/bench_n1.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['value']/1e9, l['config']['distinct'])
print(l['phases_ms'])
PY
