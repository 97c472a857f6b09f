#!/bin/bash
# compile attention.hip to assembly and cut out attn_fwd_kv_kernel<16, true> (-> /tmp/fwd2.s); prints registers / spills
cd /root/repo/adapting-2d-vits-for-3d-point-cloud-understanding_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-pass-failed -fno-honor-nans -fno-slp-vectorize --cuda-device-only -S attention.hip -o /tmp/att_test.s -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -A 10 "attn_fwd_kv_kernelILi16ELb1" | grep -E "VGPRs|Spill|LDS|error"
awk '/^_Z18attn_fwd_kv_kernelILi16ELb1EEv10AttnParams:/,/s_endpgm/' /tmp/att_test.s > /tmp/fwd2.s
wc -l /tmp/fwd2.s
grep -n "Loop Header\|Inner Loop" /tmp/fwd2.s
