"""A/B of the persistent tile loop (gemm_tn4p.h) against one workgroup per tile (gemm_tn4.h); exact compare, interleaved rounds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
dev = 'cuda'; lib = _lib.lib(); D = 768
g = torch.Generator().manual_seed(0)
for M in (256 * 257, 256 * 513):
    shapes = {'qkv': (M, 3 * D, D, ops.EPI_BF16), 'proj': (M, D, D, ops.EPI_F32_RESID), 'fc1': (M, 4 * D, D, ops.EPI_BF16_GELU),
              'fc2': (M, D, 4 * D, ops.EPI_F32_RESID), 'dfc2': (M, 4 * D, D, ops.EPI_BF16_DGELU), 'dfc1': (M, D, 4 * D, ops.EPI_BF16), 'dqkv': (M, D, 3 * D, ops.EPI_BF16), 'dproj': (M, D, D, ops.EPI_BF16)}
    for name, (m, n, k, epi) in shapes.items():
        a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        resid = torch.randn(m, n, device=dev) if epi == ops.EPI_F32_RESID else None
        pre = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi in (ops.EPI_BF16_GELU, ops.EPI_BF16_DGELU) else None
        out = torch.empty(m, n, dtype=torch.float32 if epi == ops.EPI_F32_RESID else torch.bfloat16, device=dev)
        f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, resid=resid, pre=pre, out=out)
        times, ref = {0: [], 1: []}, None
        for rnd in range(4):
            for v in (0, 1):
                lib.p4p_set_gemm_persistent(v)
                f(); torch.cuda.synchronize()
                if rnd == 0:
                    if ref is None: ref = out.clone()
                    else: assert torch.equal(out, ref), (name, 'persistent kernel differs', float((out.float() - ref.float()).abs().max()))
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): f()
                e.record(); torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / 10 * 1e3)
        print(M, name, {v: round(min(t), 1) for v, t in times.items()}, 'us', flush=True)
lib.p4p_set_gemm_persistent(1)
