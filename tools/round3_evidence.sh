#!/bin/bash
# round-3 evidence (gpurun_out/r03_*; copy what should be judged into profiles/)
#   bash tools/round3_evidence.sh 1     the bench line as the driver runs it + the headline's rocprofv3 / PMC set
#   bash tools/round3_evidence.sh 2     amp-O2 training step: rocprofv3 kernel stats + bench line (fused BatchNorm), per-cell A/B
#   bash tools/round3_evidence.sh 3     the same training evidence after the weight-gradient / BatchNorm-apply work (r03_j_*)
#   bash tools/round3_evidence.sh 1 r03_l / 4 r03_m_o2     end-of-round headline set / O2 inference set (captured multi-lane replay)
set -e
part=${1:-1}
out=gpurun_out
mkdir -p $out
if [ "$part" = "1" ]; then
    tag=${2:-r03_h}  # r03_h: mid-round; r03_l: end of round (captured multi-lane plan replay)
    python3 bench.py > $out/${tag}_bench_full.json 2> $out/${tag}_bench_full.err
    tail -c 300 $out/${tag}_bench_full.json; echo
    bash tools/profile_round.sh $tag --no-extra 2>&1 | tail -12
elif [ "$part" = "4" ]; then
    bash tools/profile_round.sh ${2:-r03_m_o2} --amp O2 --no-extra 2>&1 | tail -8
elif [ "$part" = "3" ]; then
    bash tools/profile_train.sh r03_j_train_o2 --amp O2 --batch 128 --steps 10 --warmup 3 --leg --no-roofline 2>&1 | tail -8
    MINDPOSE_BENCH_TRAIN_SHAPES=$out/r03_j_train_o2_shapes.csv MINDPOSE_TUNE_CACHE=$out/r03_j_train_o2_tune.json python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > $out/r03_j_train_o2_bench.json 2> $out/r03_j.err
    tail -c 300 $out/r03_j_train_o2_bench.json; echo
    MINDPOSE_BN_FUSE=0 python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > $out/r03_j_train_o2_percell_bench.json 2> $out/r03_j2.err
    tail -c 300 $out/r03_j_train_o2_percell_bench.json; echo
else
    bash tools/profile_train.sh r03_d_train_o2 --amp O2 --batch 128 --steps 10 --warmup 3 --leg --no-roofline 2>&1 | tail -8
    MINDPOSE_TUNE_CACHE=$out/r03_d_train_o2_tune.json python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > $out/r03_d_train_o2_bench.json 2> $out/r03_d.err
    tail -c 300 $out/r03_d_train_o2_bench.json; echo
    MINDPOSE_BN_FUSE=0 python3 bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg --no-roofline > $out/r03_d_train_o2_percell_bench.json 2> $out/r03_d2.err
    tail -c 300 $out/r03_d_train_o2_percell_bench.json; echo
    bash tools/profile_round.sh r03_i_o2 --amp O2 --no-extra 2>&1 | tail -8
fi
