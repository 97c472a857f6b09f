mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "wgrad or t513_workspace" > gpurun_out/r05_t_wgrad.log 2>&1 || { tail -30 gpurun_out/r05_t_wgrad.log; exit 1; }
tail -2 gpurun_out/r05_t_wgrad.log
for rep in 1 2 3; do for v in 0 1; do
  timeout -k 10 200 python tools/ab_flag.py functional._DEFER_SLAB_SUMS=$v --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$v', round(d['ms_per_step'],3), d['config'].get('final_loss'))"
done; done
timeout -k 10 300 python tools/cold_vs_warm.py 2>&1 | grep -v amdgpu
