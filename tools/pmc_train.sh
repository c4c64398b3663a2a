#!/bin/bash
# Counters for the amp-O2 training step (bench.py --workload hrnet_w32_train):  bash tools/pmc_train.sh <tag> [bench args]
#   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE / MFMA-busy,
#      separate passes as MI355X_MICROARCH.md prescribes), <tag>_sq_counters.json (SQ wait / active counters).  Copy into profiles/.
set -e
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp MINDPOSE_TUNE_CACHE=$out/${tag}_tune.json
args="--workload hrnet_w32_train --amp O2 --batch 128 --leg --no-roofline $*"
python3 bench.py $args --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_tune.err   # fills the tuner cache
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 $root/bench.py $args --steps 10 --warmup 3 > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err
cp "$(find $out/${tag}_prof -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
rm -rf $out/${tag}_prof
echo "stats pass done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/${tag}_pmc_$i -o ${tag} -- python3 $root/bench.py $args --steps 1 --warmup 1 > /dev/null 2> $out/${tag}_pmc_$i.err || echo "pass $i failed"
  echo "pmc pass $i done"
done
cd $root
python3 tools/pmc_train.py $tag $out/${tag}_kernel_stats.csv $out/${tag}_pmc_1 $out/${tag}_pmc_2 $out/${tag}_pmc_3 $out/${tag}_pmc_4 $out/${tag}_pmc_5
rm -rf $out/${tag}_pmc_1 $out/${tag}_pmc_2 $out/${tag}_pmc_3 $out/${tag}_pmc_4 $out/${tag}_pmc_5
