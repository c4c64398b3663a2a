#!/bin/bash
# round 3, call A: new parity tests + baseline O2 training profile (per-shape table)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_headline_plan.py "tests/test_gpu_train_full.py::test_o2_training_step_vs_oracle_amp_emulation" tests/test_gpu_dp.py -x -q -s > gpurun_out/r3a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a_tests.log
python -m pytest tests/test_gpu_conv.py -x -q -k "resnet101 or resnet152 or gemm or winograd" >> gpurun_out/r3a_tests.log 2>&1
echo "tests2 rc=$?" >> gpurun_out/r3a_tests.log
MINDPOSE_BENCH_TRAIN_SHAPES=gpurun_out/r3a_train_shapes.csv python bench.py --workload hrnet_w32_train --amp O2 --batch 128 --steps 20 --warmup 5 --leg > gpurun_out/r3a_train_o2.json 2> gpurun_out/r3a_train_o2.err
echo "bench rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
