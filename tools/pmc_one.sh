#!/bin/bash
# SQ counter passes over ONE layer / variant (tools/ws_one.py; PMC_PROG=<script> runs another program):  bash tools/pmc_one.sh <tag> <variant> <cin> <cout> <h> <w> <n>
#   -> gpurun_out/<tag>_sq.json + a per-kernel line of fractions of SQ_WAVE_CYCLES
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_sq_$i -o ${tag} -- python3 $root/${PMC_PROG:-tools/ws_one.py} "$@" > /dev/null 2> $out/${tag}_sq_$i.err || echo "pass $i failed"
done
cd $root
python3 tools/pmc_waits.py $tag $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 $out/${tag}_sq_5
rm -rf $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 $out/${tag}_sq_5
