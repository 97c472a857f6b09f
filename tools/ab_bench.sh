#!/bin/bash
# A/B inside ONE gpurun call (box-to-box variance is +-3%): tools/ab_bench.sh "ENV_A" "ENV_B" ...  each arg = "K=V K=V"
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'])"
  done
done
