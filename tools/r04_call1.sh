#!/bin/bash
# round 4, call 1: new DDP / reserved-CU tests, the contention rehearsal, and a baseline bench line on this box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_kernels.py -x -q -m gpu -k "rccl or fewer_cus or two_rank" > gpurun_out/r04_c1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04_c1_tests.log
tail -3 gpurun_out/r04_c1_tests.log
timeout -k 10 300 python tools/contention_rehearsal.py --steps 8 > gpurun_out/r04_c1_contention_n1024.log 2>&1
echo "rehearsal rc=$?"; tail -8 gpurun_out/r04_c1_contention_n1024.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r04_c1_bench.json 2> gpurun_out/r04_c1_bench.log
echo "bench rc=$?"; tail -c 1500 gpurun_out/r04_c1_bench.json
