#!/bin/bash
# round 4, call 8: the K/V-resident T = 513 forward: parity first, then the A/B and the stamps
set -e
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "forward_kv_resident or test_attention_fwd_bwd" > gpurun_out/call8_tests.log 2>&1 || { tail -40 gpurun_out/call8_tests.log; exit 1; }
tail -3 gpurun_out/call8_tests.log
timeout -k 10 200 python3 tools/attn_fwd2_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/call8_ab.log
timeout -k 10 300 python3 tools/attn_fwd2_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/call8_stamps.log
