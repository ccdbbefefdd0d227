#!/bin/bash
# usage: tools/final_pass_r04.sh <tag> [bench|prof|suite|all] -- the round's evidence on the GPU box.
#   bench: bench lines of every config (+ repeat-rich / tree / table-of-reads / multi-rank rehearsal variants)
#   prof:  rocprofv3 kernel stats of the default bench command, PMC passes of one count, the probes, the fuzz soaks
#   suite: the whole GPU test suite, plain, with the poisoned pool and with guard bands
# Every part stops at its first failing step.
TAG=${1:-r04_final}; PART=${2:-all}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
B="timeout -k 10 400 python3 bench.py"
if [ "$PART" = "bench" ] || [ "$PART" = "all" ]; then
  $B > $O/bench.json 2> $O/bench.err || exit 1
  echo "headline: $(python3 -c "import json;l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(round(l['ms_per_step'],2),'ms',round(l['value']/1e9,1),'G',l['roofline']['kernel'],l['roofline']['frac'],'digest_ok',l['digest_ok'])")"
  $B --engine tree --no-cpu-baseline > $O/bench_tree.json 2>> $O/bench.err || exit 1
  for m in 1000 100000 64 1; do $B --motif $m --no-cpu-baseline > $O/bench_cfg4_motif$m.json 2>> $O/bench.err || exit 1; done
  $B --config 2 > $O/bench_cfg2.json 2>> $O/bench.err || exit 1
  $B --config 2 --motif 1000 --no-cpu-baseline > $O/bench_cfg2_motif1000.json 2>> $O/bench.err || exit 1
  $B --config 3 > $O/bench_cfg3.json 2>> $O/bench.err || exit 1
  $B --config 3 --motif 1000 --no-cpu-baseline > $O/bench_cfg3_motif1000.json 2>> $O/bench.err || exit 1
  $B --config 3 --genome-like --no-cpu-baseline > $O/bench_cfg3_genome_like.json 2>> $O/bench.err || exit 1
  $B --config 2 --genome-like --no-cpu-baseline > $O/bench_cfg2_genome_like.json 2>> $O/bench.err || exit 1
  $B --config 5 --steps 200 --warmup 20 > $O/bench_cfg5.json 2>> $O/bench.err || exit 1
  $B --config 5 --steps 200 --warmup 20 --pattern ACGNNNNNNNNNNNNNNNNNN --no-cpu-baseline > $O/bench_cfg5_sel64.json 2>> $O/bench.err || exit 1
  $B --config 5 --steps 200 --warmup 20 --keys-only --no-cpu-baseline > $O/bench_cfg5_keys_only.json 2>> $O/bench.err || exit 1
  $B --config 6 --no-cpu-baseline > $O/bench_cfg6.json 2>> $O/bench.err || exit 1
  $B --config 6 --table-host-starts --no-cpu-baseline > $O/bench_cfg6_host_starts.json 2>> $O/bench.err || exit 1
  $B --config 6 --k 21 --no-cpu-baseline > $O/bench_cfg6_k21.json 2>> $O/bench.err || exit 1
  for g in 2 8; do $B --gpus $g --steps 3 --warmup 1 > $O/bench_gpus${g}_rehearsal.json 2>> $O/bench.err || exit 1; done
  python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/bench*.json')):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print('%-34s %8.3f ms %8.1f G  digest_ok=%s total_ok=%s' % (os.path.basename(f), l['ms_per_step'], l['value']/1e9, l.get('digest_ok'), l.get('digest_total_ok')))
    except Exception as e:
        print(f, 'unreadable', e)
PY
  echo "bench lines done"
fi
if [ "$PART" = "prof" ] || [ "$PART" = "all" ]; then
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err ) || exit 1
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
  rm -rf $O/prof
  echo "kernel stats done"
  bash tools/pmc_cmd.sh $TAG sk_once.py 3e9 31 2 > $O/pmc.log 2>&1
  cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary_3e9.txt
  python3 tools/pmc_traffic.py gpurun_out/pmc_$TAG/summary.txt profiles/${TAG}_pmc_summary_3e9.txt > $O/traffic.json
  echo "pmc done"
  bash tools/run_pmc_sq.sh ${TAG}_sq 1e9 31 > $O/pmc_sq_1e9.txt 2>&1
  echo "sq counters done"
  timeout -k 10 300 python3 tools/records_probe.py 3e9 31 8 350 > $O/records_probe.log 2>&1
  timeout -k 10 120 python3 tools/overhead_probe.py > $O/overhead_probe.log 2>&1
  echo "probes done"
  timeout -k 10 900 python3 tools/fuzz_unordered.py ${FUZZ_U:-2500} > $O/fuzz_unordered.log 2>&1; tail -1 $O/fuzz_unordered.log
  timeout -k 10 300 python3 tools/fuzz_count.py ${FUZZ_C:-600} > $O/fuzz_count.log 2>&1; tail -1 $O/fuzz_count.log
fi
if [ "$PART" = "suite" ] || [ "$PART" = "all" ]; then
  timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
  grep -q "rc=0" $O/pytest_gpu.log || exit 1
  if [ "$3" = "debug" ]; then
    DNAGPU_TEST_POISON=1 timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu_poison.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_poison.log; tail -3 $O/pytest_gpu_poison.log
  fi
fi
