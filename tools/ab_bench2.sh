#!/bin/bash
# like ab_bench.sh with extra bench arguments: tools/ab_bench2.sh "<bench args>" "ENV_A" "ENV_B" ...
args="$1"; shift
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'])"
  done
done
