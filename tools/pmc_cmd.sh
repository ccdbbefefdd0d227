#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <python script under tools/> [args...] -- one rocprofv3 --pmc pass per counter
# group (kernel-trace only, separate passes: FETCH_SIZE and WRITE_SIZE cannot share one), then a summary.
TAG=$1; shift
SCRIPT=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/tools/$SCRIPT "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summarize.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt 2>&1
cat gpurun_out/pmc_$TAG/summary.txt
