#!/bin/bash
# the whole -m gpu suite, log to gpurun_out/full_gpu_tests.log  (extra pytest arguments are passed through)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu "$@" > gpurun_out/full_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/full_gpu_tests.log
grep -v "amdgpu.ids" gpurun_out/full_gpu_tests.log | tail -25 | cut -c1-300
