#!/bin/bash
# round 3, first GPU call: new multi path tests, digests of configs 2/3, bench N=1 and --gpus 2/8 rehearsal, overhead probe
set -o pipefail
mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi or config2 or config3 or bench_gpus2 or records_exchange" > gpurun_out/r03a/pytest_multi.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest_multi.log
tail -5 gpurun_out/r03a/pytest_multi.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r03a/bench_n1.json 2> gpurun_out/r03a/bench_n1.err && cat gpurun_out/r03a/bench_n1.json | cut -c1-1500
for p in 1 2 3; do
timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --parts $p > gpurun_out/r03a/bench_g2_p$p.json 2> gpurun_out/r03a/bench_g2_p$p.err; tail -c 1200 gpurun_out/r03a/bench_g2_p$p.json
done
timeout -k 10 300 python bench.py --gpus 8 --steps 3 --warmup 1 --parts 2 > gpurun_out/r03a/bench_g8_p2.json 2> gpurun_out/r03a/bench_g8_p2.err; tail -c 1200 gpurun_out/r03a/bench_g8_p2.json
timeout -k 10 300 python tools/records_probe.py 3e9 31 8 > gpurun_out/r03a/records_probe.log 2>&1; cat gpurun_out/r03a/records_probe.log
timeout -k 10 300 python tools/overhead_probe.py > gpurun_out/r03a/overhead_probe.log 2>&1; cat gpurun_out/r03a/overhead_probe.log
