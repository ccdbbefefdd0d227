#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <lib.so> <python script under tools/> [args...] -- the two SQ counter groups only
TAG=$1; shift
LIB=$1; shift
SCRIPT=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/$SCRIPT "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summarize.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt 2>&1
