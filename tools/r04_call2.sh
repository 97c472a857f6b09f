#!/bin/bash
# round 4, call 2: host-overhead fix (FusedAdamW tables), wgrad splits following reserved CUs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_optim.py tests/test_gpu_trainer.py -x -q -m gpu > gpurun_out/r04_c2_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04_c2_tests.log
timeout -k 10 200 python tools/host_profile.py --top 25 > gpurun_out/r04_c2_hostprof_n8192.log 2>&1
echo "hostprof rc=$?"; head -12 gpurun_out/r04_c2_hostprof_n8192.log
timeout -k 10 200 python tools/host_profile.py --points 1024 --batch 256 --vit vit_base_patch16_384 --top 12 > gpurun_out/r04_c2_hostprof_n1024.log 2>&1
echo "hostprof rc=$?"; head -8 gpurun_out/r04_c2_hostprof_n1024.log
timeout -k 10 300 python tools/contention_rehearsal.py --steps 8 --rs 0,8,32 > gpurun_out/r04_c2_contention_n1024.log 2>&1
echo "rehearsal rc=$?"; tail -6 gpurun_out/r04_c2_contention_n1024.log
