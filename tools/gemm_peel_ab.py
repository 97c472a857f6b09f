"""Grid quantisation of the wide GEMMs: M = 256 x 257 rows and N = 2304 / 3072 columns are 9.04 / 12.05 rounds of 256 x 256 tiles on 256 CUs.  The narrow
ones (N = 768: 3 rounds + 3 tiles) have had their last row panel peeled into a small-tile launch since round 2; is it worth it at 9-12 rounds?
p4p_set_gemm_peel_rounds(v): peel when the big tiles run at most v rounds.  Interleaved rounds; outputs compared (equal, or how far apart)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()
D = 768
g = torch.Generator().manual_seed(0)
for M in (256 * 257, 256 * 513):
    for name, epi, N, K in (('qkv', ops.EPI_BF16, 3 * D, D), ('fc1 (GELU)', ops.EPI_BF16_GELU, 4 * D, D), ('dfc2 (dGELU)', ops.EPI_BF16_DGELU, 4 * D, D)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
        bias = torch.randn(N, generator=g).cuda()
        pre = torch.rand(M, N, generator=g).to(torch.bfloat16).cuda() if epi in (ops.EPI_BF16_GELU, ops.EPI_BF16_DGELU) else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, pre=pre, out=out)
        times, outs = {4: [], 32: []}, {}
        for rnd in range(4):
            for v in (4, 32):
                lib.p4p_set_gemm_peel_rounds(v)
                f(); torch.cuda.synchronize()
                outs[v] = out.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 100)
        lib.p4p_set_gemm_peel_rounds(4)
        d = (outs[4].float() - outs[32].float()).abs().max().item()
        print(f'{name:14s} M={M}: whole {min(times[4]):.1f} us   tail peeled {min(times[32]):.1f} us   max |diff| {d:.3g}', flush=True)
        del a, w, out
