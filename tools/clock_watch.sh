#!/bin/bash
# Samples rocm-smi (shader clock, power) every 0.5 s while a bench.py run is in flight: what clock does the package hold under this step?
# usage: bash tools/clock_watch.sh [bench args]     (NOTES R5.6: the bf16 MFMA loops sit at a power / clock ceiling)
python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline "$@" > /tmp/cw_bench.json 2>/dev/null &
BP=$!
sleep 20
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/cw_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],2))"
