#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_t513 or fused_kernels_against or fewer_cus or reproducible_under_load" > gpurun_out/r04_c4_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r04_c4_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/attn_f2_ab.py > gpurun_out/r04_c4_f2_ab.log 2>&1
echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r04_c4_f2_ab.log
F2_CS=1 timeout -k 10 200 python tools/attn_f2_stamps.py > gpurun_out/r04_c4_f2_stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_c4_f2_stamps.log
