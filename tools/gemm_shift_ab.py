"""Start-time shift of the persistent GEMM (gemm_tn4p.h; p4p_set_gemm_epilogue bits 8..): every other workgroup of an XCD starts late, so
that the HBM-bound epilogues of one half fall into the K loops of the other.  Interleaved rounds, outputs compared bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()
D = 768
g = torch.Generator().manual_seed(0)
shifts = [0, 8, 16, 24, 32, 48]          # 0 here = NO shift (bit 1); the library's default is per epilogue (gemm.hip)
for M in (256 * 257, 256 * 513):
    for name, epi, N, K in (('proj (resid)', ops.EPI_F32_RESID, D, D), ('dfc2 (dGELU)', ops.EPI_BF16_DGELU, 4 * D, D), ('qkv', ops.EPI_BF16, 3 * D, D), ('fc1 (GELU)', ops.EPI_BF16_GELU, 4 * D, D)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
        bias = torch.randn(N, generator=g).cuda()
        resid = torch.randn(M, N, generator=g).cuda() if epi == ops.EPI_F32_RESID else None
        pre = torch.rand(M, N, generator=g).to(torch.bfloat16).cuda() if epi in (ops.EPI_BF16_GELU, ops.EPI_BF16_DGELU) else None
        out = torch.empty(M, N, dtype=torch.float32 if epi == ops.EPI_F32_RESID else torch.bfloat16, device='cuda')
        f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, resid=resid, pre=pre, out=out)
        times, ref = {s: [] for s in shifts}, None
        for rnd in range(4):
            for s in shifts:
                lib.p4p_set_gemm_epilogue((s << 8) if s else 2)
                f(); torch.cuda.synchronize()
                if ref is None: ref = out.clone()
                else: assert torch.equal(out, ref), (name, s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                times[s].append(e0.elapsed_time(e1) * 100)
        lib.p4p_set_gemm_epilogue(0)
        print(f'{name:14s} M={M}: ' + '   '.join(f'shift {s}k: {min(t):.1f}' for s, t in times.items()) + ' us', flush=True)
        del a, w, out
