#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_t513 or fused_kernels_against or fewer_cus or reproducible_under_load or attention_fwd_bwd" > gpurun_out/r04_c3_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r04_c3_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/attn_f2_ab.py --sweep > gpurun_out/r04_c3_f2_ab.log 2>&1
echo "ab rc=$?"; cat gpurun_out/r04_c3_f2_ab.log | grep -v amdgpu.ids
