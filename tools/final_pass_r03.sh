#!/bin/bash
# usage: tools/final_pass_r03.sh <tag> [suite] -- the round's evidence in one GPU call: bench lines of every config (and the
# repeat-rich / tree / multi-rank rehearsal variants), rocprofv3 kernel stats of the default bench command, PMC passes of one
# count, the record-exchange and fixed-cost probes; with "suite": the whole GPU test suite, plain and with the poisoned pool.
TAG=${1:-r03_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="timeout -k 10 400 python3 bench.py"
$B > $O/bench.json 2> $O/bench.err || exit 1
echo "headline: $(python3 -c "import json;l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(round(l['ms_per_step'],2),'ms',round(l['value']/1e9,1),'G',l['roofline']['kernel'],l['roofline']['frac'])")"
$B --engine tree --no-cpu-baseline > $O/bench_tree.json 2>> $O/bench.err || exit 1
for m in 1000 100000 64 1; do $B --motif $m --no-cpu-baseline > $O/bench_cfg4_motif$m.json 2>> $O/bench.err || exit 1; done
$B --config 2 > $O/bench_cfg2.json 2>> $O/bench.err || exit 1
$B --config 2 --motif 1000 --no-cpu-baseline > $O/bench_cfg2_motif1000.json 2>> $O/bench.err || exit 1
$B --config 3 > $O/bench_cfg3.json 2>> $O/bench.err || exit 1
$B --config 3 --motif 1000 --no-cpu-baseline > $O/bench_cfg3_motif1000.json 2>> $O/bench.err || exit 1
$B --config 5 --steps 200 --warmup 20 > $O/bench_cfg5.json 2>> $O/bench.err || exit 1
$B --config 5 --steps 200 --warmup 20 --pattern ACGNNNNNNNNNNNNNNNNNN --no-cpu-baseline > $O/bench_cfg5_sel64.json 2>> $O/bench.err || exit 1
$B --config 5 --n-bases 1000000000 --steps 50 --no-cpu-baseline > $O/bench_cfg5_1e9.json 2>> $O/bench.err || exit 1
$B --config 5 --n-bases 1000000000 --steps 50 --pattern ACGNNNNNNNNNNNNNNNNNN --no-cpu-baseline > $O/bench_cfg5_sel64_1e9.json 2>> $O/bench.err || exit 1
for g in 2 4 8; do $B --gpus $g --steps 3 --warmup 1 > $O/bench_gpus${g}_rehearsal.json 2>> $O/bench.err || exit 1; done
echo "bench lines done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err ) || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
echo "kernel stats done"
bash tools/pmc_cmd.sh $TAG sk_once.py 3e9 31 2 > $O/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary_3e9.txt
echo "pmc done"
timeout -k 10 300 python3 tools/records_probe.py 3e9 31 1,2,4,8 350 > $O/records_probe.log 2>&1
timeout -k 10 120 python3 tools/overhead_probe.py > $O/overhead_probe.log 2>&1
echo "probes done"
timeout -k 10 600 python3 tools/fuzz_unordered.py ${FUZZ_U:-2500} > $O/fuzz_unordered.log 2>&1; tail -1 $O/fuzz_unordered.log
timeout -k 10 300 python3 tools/fuzz_count.py ${FUZZ_C:-600} > $O/fuzz_count.log 2>&1; tail -1 $O/fuzz_count.log
if [ "$2" = "suite" ]; then
  timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  DNAGPU_TEST_POISON=1 timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/pytest_gpu_poison.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_poison.log; tail -3 $O/pytest_gpu_poison.log
fi
