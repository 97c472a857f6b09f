"""Random pair counts above the CU count for the WALKING attention backward kernels (T = 257: attn_bwd_fused_kernel<8, true>; T = 513:
attn_bwd_dkdv_ks_kernel<16, true>): bit-equal to a workgroup per pair / item (variant bit 7), repeated launches bit-equal to each
other, a slice against fp32 autograd; plus the kernels of round 3's host-side changes (LayerNorm backward out of place, slab sum,
global_feat) at random shapes through the bodies of the parametrised tests.  FUZZ_SECONDS (default 120), FUZZ_SEED."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import p4p_amd
from p4p_amd import ops, _lib
import test_gpu_01_kernels as T

lib = _lib.lib()
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 0)))
budget = float(os.environ.get('FUZZ_SECONDS', 120))
ncu = _lib.call_int('p4p_device_cu_count')
t0, n, fails = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = int(rng.integers(0, 4))
    try:
        if kind == 0:
            Tn = int(rng.choice([257, 513]))
            H = int(rng.choice([1, 2, 3, 6, 12, 16]))
            pairs = int(rng.integers(ncu + 1, 3 * ncu)) if Tn == 257 else int(rng.integers(ncu // 2 + 1, 2 * ncu))
            B = max(1, (pairs + H - 1) // H)
            D = 64 * H
            g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
            qkv = (torch.randn(B * Tn, 3 * D, generator=g) * float(rng.uniform(0.5, 2.0))).to(torch.bfloat16).cuda()
            dout = torch.randn(B * Tn, D, generator=g).to(torch.bfloat16).cuda()
            out, lse = ops.attention_fwd(qkv, B, Tn, H)
            lib.p4p_set_attention_variant(128)
            a, acs = ops.attention_bwd(qkv, out, dout, lse, B, Tn, H, want_colsum=True)
            a, acs = a.clone(), acs.clone()
            lib.p4p_set_attention_variant(0)
            b, bcs = ops.attention_bwd(qkv, out, dout, lse, B, Tn, H, want_colsum=True)
            b, bcs = b.clone(), bcs.clone()
            c, ccs = ops.attention_bwd(qkv, out, dout, lse, B, Tn, H, want_colsum=True)
            assert torch.equal(a, b) and torch.equal(acs, bcs), f'walk != per pair at B={B} T={Tn} H={H}'
            assert torch.equal(b, c) and torch.equal(bcs, ccs), f'walk not reproducible at B={B} T={Tn} H={H}'
            sl = slice((B - 1) * Tn, B * Tn)
            qr = qkv[sl].float().requires_grad_(True)
            T._attn_ref(qr, 1, Tn, H).backward(dout[sl].float())
            assert (b[sl].float() - qr.grad).abs().max().item() <= 3e-2 * max(qr.grad.abs().max().item(), 1.0)
            desc = f'walk B={B} T={Tn} H={H}'
        elif kind == 1:
            M, D = int(rng.integers(1, 5000)), int(rng.choice([64, 192, 256, 384, 768, 1024]))
            T.test_layernorm_fwd_bwd(M, D)
            desc = f'layernorm M={M} D={D}'
        elif kind == 2:
            T.test_sum_slabs_adds_in_slab_order()
            desc = 'sum_slabs'
        else:
            B, Tn, D = int(rng.integers(2, 40)), int(rng.integers(2, 600)), int(rng.choice([8, 64, 192, 260, 768, 1024, 6]))
            T.test_global_feat_forward_edge_cases(B, Tn, D)
            desc = f'global_feat B={B} T={Tn} D={D}'
    except Exception as e:               # noqa: BLE001
        fails += 1
        print('FAIL', e, flush=True)
        lib.p4p_set_attention_variant(0)
    n += 1
    if n % 20 == 0:
        print(f'{n} cases, {fails} failures, {time.time() - t0:.0f} s (last: {desc})', flush=True)
print(f'done: {n} cases, {fails} failures in {time.time() - t0:.0f} s')
sys.exit(1 if fails else 0)
