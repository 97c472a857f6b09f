#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matched_parity.py -x -q -m gpu -s -k "teacher_forced or attention_fwd_bwd or free_running" > gpurun_out/r04_c6_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "per-block|worst links|gradient rel|passed|failed|Error|assert|flips" gpurun_out/r04_c6_tests.log | tail -30
