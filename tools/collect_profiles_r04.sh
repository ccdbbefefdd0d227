#!/bin/bash
# tools/collect_profiles_r04.sh [tag]: copies the evidence of tools/final_pass_r04.sh from gpurun_out/<tag>/ (scratch) into
# profiles/ (tracked) under the names DESIGN.md cites.  Run in the repo root after the GPU passes.
TAG=${1:-r04_final}
S=gpurun_out/$TAG
[ -d $S ] || { echo "no $S"; exit 1; }
for f in $S/bench*.json; do [ -f $f ] && cp $f profiles/${TAG}_$(basename $f); done
for f in kernel_stats.csv pmc_summary_3e9.txt records_probe.log overhead_probe.log fuzz_unordered.log fuzz_count.log pytest_gpu.log pytest_gpu_poison.log prof_bench.json; do
  [ -f $S/$f ] && cp $S/$f profiles/${TAG}_$f
done
[ -f $S/pmc_sq_1e9.txt ] && cp $S/pmc_sq_1e9.txt profiles/r04_pmc_sq_1e9.txt
[ -s $S/traffic.json ] && cp $S/traffic.json profiles/traffic_latest.json
ls profiles | grep $TAG | wc -l
