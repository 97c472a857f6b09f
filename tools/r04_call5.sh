#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_c5_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/r04_c5_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/r04_c5_bench.json 2> gpurun_out/r04_c5_bench.log
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c5_bench.json'))
print({k:d[k] for k in ['value','ms_per_step','host_enqueue_ms_per_step','mfma_frac_whole_step']})
r=d['roofline']; print('wgrad',r['achieved'], r['avg_launch_us'], 'gemms', r['all_mfma_gemms'], 'attn', r['attention']['fwd_us_per_launch'], r['attention']['bwd_us_per_launch'])
n=d['north_star_point']; print({k:n[k] for k in ['clouds_per_sec','ms_per_step','host_enqueue_ms_per_step','mfma_frac_whole_step']}, n['all_mfma_gemms'], n['attention']['fwd_us_per_launch'], n['attention']['bwd_us_per_launch'], n['attention'].get('bwd_TFLOPs'))
n=d['n8192_point']; print({k:n[k] for k in ['clouds_per_sec','ms_per_step','ms_per_step_runs','scatter_pct','host_enqueue_ms_per_step','mfma_frac_whole_step']})
PY
