#!/bin/bash
# tools/run_pmc_sq.sh <tag> [n_bases] [k]: the two SQ counter groups over one unordered count (instructions per kernel)
TAG=${1:-sq}; N=${2:-1e9}; K=${3:-31}
bash $GRAFT_REPO_ROOT/tools/pmc_sq.sh $TAG dna-sequences-pg-extension_amd/libdnagpu.so sk_once.py $N $K 2
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/summary.txt
