#!/bin/bash
# usage: tools/final_pass.sh <tag> -- the round's evidence in one GPU call: bench lines of every config (and the
# repeat-rich / tree variants), rocprofv3 kernel stats of the default bench command, PMC passes of one count.
TAG=${1:-r02_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
set -x
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 bench.py --engine tree --no-cpu-baseline > $O/bench_tree.json 2>> $O/bench.err || exit 1
python3 bench.py --motif 1000 --no-cpu-baseline > $O/bench_cfg4_motif1000.json 2>> $O/bench.err || exit 1
python3 bench.py --motif 100000 --no-cpu-baseline > $O/bench_cfg4_motif100000.json 2>> $O/bench.err || exit 1
python3 bench.py --motif 64 --no-cpu-baseline > $O/bench_cfg4_motif64.json 2>> $O/bench.err || exit 1
python3 bench.py --motif 1 --no-cpu-baseline > $O/bench_cfg4_polyA_half.json 2>> $O/bench.err || exit 1
python3 bench.py --config 2 > $O/bench_cfg2.json 2>> $O/bench.err || exit 1
python3 bench.py --config 2 --motif 1000 --no-cpu-baseline > $O/bench_cfg2_motif1000.json 2>> $O/bench.err || exit 1
python3 bench.py --config 3 > $O/bench_cfg3.json 2>> $O/bench.err || exit 1
python3 bench.py --config 3 --motif 1000 --no-cpu-baseline > $O/bench_cfg3_motif1000.json 2>> $O/bench.err || exit 1
python3 bench.py --config 5 > $O/bench_cfg5.json 2>> $O/bench.err || exit 1
python3 bench.py --config 5 --motif 1000 --no-cpu-baseline > $O/bench_cfg5_motif1000.json 2>> $O/bench.err || exit 1
python3 bench.py --config 5 --pattern ACGNNNNNNNNNNNNNNNNNN --no-cpu-baseline > $O/bench_cfg5_sel64.json 2>> $O/bench.err || exit 1
python3 bench.py --config 5 --n-bases 1000000000 --no-cpu-baseline > $O/bench_cfg5_1e9.json 2>> $O/bench.err || exit 1
python3 bench.py --config 5 --n-bases 1000000000 --pattern ACGNNNNNNNNNNNNNNNNNN --no-cpu-baseline > $O/bench_cfg5_sel64_1e9.json 2>> $O/bench.err || exit 1
echo "bench lines done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err ) || exit 1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
echo "kernel stats done"
bash tools/pmc_cmd.sh $TAG sk_once.py 3e9 31 2 > $O/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt $O/pmc_summary_3e9.txt
rm -rf $O/prof
echo "pmc done"
