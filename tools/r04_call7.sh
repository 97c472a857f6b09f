#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/r04_c7_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r04_c7_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/gemm_lib_ab.py tools/ab_libs/libp4p_r04_before_epi.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c7_gemm_ab.log
timeout -k 10 300 python tools/gemm_lib_ab.py tools/ab_libs/libp4p_r04_before_epi.so 131328 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_c7_gemm_ab.log
