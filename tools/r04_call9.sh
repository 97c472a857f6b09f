#!/bin/bash
# round 4, call 9: knock-out pricing of the K/V-resident forward's key-tile loop (diagnostic builds, wrong results by design)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for ko in ${KOS:-0 1 2 4 8 16 32 64 6 14 49 127}; do
  timeout -k 10 120 python3 tools/attn_fwd2_stamps.py -DP4P_FW2_KO=$ko 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/call9_ko.log || exit 1
done
