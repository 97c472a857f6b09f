#!/bin/bash
# round 4, call 10: model-level random sweeps after the lazy rescale, then SQ counters of the attention forward kernels
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
FUZZ_SECONDS=200 FUZZ_SEED=77 timeout -k 10 400 python3 tests/tools/fuzz_model.py > gpurun_out/fuzz_model_r04.log 2>&1; echo "fuzz_model rc=$?"; tail -2 gpurun_out/fuzz_model_r04.log
FUZZ_SECONDS=120 FUZZ_SEED=78 timeout -k 10 300 python3 tests/tools/fuzz_apf.py > gpurun_out/fuzz_apf_r04.log 2>&1; echo "fuzz_apf rc=$?"; tail -2 gpurun_out/fuzz_apf_r04.log
FUZZ_SECONDS=100 FUZZ_SEED=79 timeout -k 10 300 python3 tests/tools/fuzz_walk.py > gpurun_out/fuzz_walk_r04.log 2>&1; echo "fuzz_walk rc=$?"; tail -2 gpurun_out/fuzz_walk_r04.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  AF_RUNS=5 timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/sq_$tag -- python3 $R/tools/attn_fwd2_ab.py > /dev/null 2>&1
  python3 $R/tools/pmc_sq.py $R/gpurun_out/sq_$tag/*/*counter_collection.csv attn_fwd >> $R/gpurun_out/sq_attn_fwd.log 2>&1
  rm -rf $R/gpurun_out/sq_$tag
done
cat $R/gpurun_out/sq_attn_fwd.log | head -60
