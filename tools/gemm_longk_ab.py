"""K = 3072 GEMMs (fc2 with two residual streams, dfc1): one workgroup per tile (gemm_tn4.h, the default) against the persistent loop
(p4p_set_gemm_persistent bit 2) with start-time shifts.  Interleaved, bit-compared."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()
D = 768
g = torch.Generator().manual_seed(0)
variants = [('tile/WG', 1, 0), ('persistent, no shift', 5, 2), ('persistent 16k', 5, 16 << 8), ('persistent 32k', 5, 32 << 8), ('persistent 64k', 5, 64 << 8)]
for M in (256 * 257, 256 * 513):
    for name, epi, N, K, nres in (('fc2 (resid + pos)', ops.EPI_F32_RESID, D, 4 * D, 2), ('dfc1', ops.EPI_BF16, D, 4 * D, 0)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
        bias = torch.randn(N, generator=g).cuda()
        r1 = torch.randn(M, N, generator=g).cuda() if nres else None
        r2 = torch.randn(M, N, generator=g).cuda() if nres > 1 else None
        out = torch.empty(M, N, dtype=torch.float32 if nres else torch.bfloat16, device='cuda')
        f = lambda: ops.gemm_tn(a, w, epi, bias=bias, resid=r1, resid2=r2, out=out)
        times, ref = {v[0]: [] for v in variants}, None
        for rnd in range(4):
            for tag, pers, eflags in variants:
                lib.p4p_set_gemm_persistent(pers); lib.p4p_set_gemm_epilogue(eflags)
                f(); torch.cuda.synchronize()
                if ref is None: ref = out.clone()
                else: assert torch.equal(out, ref), (name, tag)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                times[tag].append(e0.elapsed_time(e1) * 100)
        lib.p4p_set_gemm_persistent(1); lib.p4p_set_gemm_epilogue(0)
        print(f'{name:18s} M={M}: ' + '   '.join(f'{t}: {min(x):.1f}' for t, x in times.items()) + ' us', flush=True)
        del a, w, out
