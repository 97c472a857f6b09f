#!/bin/bash
# Collects the per-round evidence under gpurun_out/ (copied into profiles/ by hand): un-profiled bench lines, rocprofv3
# kernel stats of the same commands, and the two PMC passes (FETCH_SIZE / WRITE_SIZE).  usage: tools/collect_profiles.sh r02
set -e
R=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 10 --warmup 3 > $OUT/bench_N1.json 2> $OUT/bench_N1.err
echo "bench done"
python3 $ROOT/bench.py --workload apf --steps 10 --warmup 3 > $OUT/apf_bench_N1.json 2> $OUT/apf_bench_N1.err
echo "apf bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n1024 -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-north-star --no-pmc > $OUT/prof_n1024.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n2048 -- python3 $ROOT/bench.py --points 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $OUT/prof_n2048.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_n8192 -- python3 $ROOT/bench.py --points 8192 --batch 8 --vit vit_large_patch16_384 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline > $OUT/prof_n8192.json 2> /dev/null
echo "n8192 kernel stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_apf -- python3 $ROOT/bench.py --workload apf --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_apf.json 2> /dev/null
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-north-star > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-north-star > /dev/null 2>&1
echo "pmc done"
cd $ROOT
for d in prof_n1024 prof_n2048 prof_n8192 prof_apf; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv; rm -rf $OUT/$d; done
python3 tools/pmc_summary.py $OUT/pmc_fetch/*/*counter_collection.csv $OUT/pmc_write/*/*counter_collection.csv 5 30 > $OUT/pmc_fetch_write_table.md
rm -rf $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
