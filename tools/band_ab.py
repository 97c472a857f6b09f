"""A/B of the banded tile order of the persistent GEMM (gemm_tn.h: band_remap) on the two N = 3072 shapes; exact compare."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
dev = 'cuda'; lib = _lib.lib(); D = 768
g = torch.Generator().manual_seed(0)
for M in (256 * 257, 256 * 513):
    shapes = {'fc1': (M, 4 * D, D, ops.EPI_BF16_GELU), 'dfc2': (M, 4 * D, D, ops.EPI_BF16_DGELU)}
    for name, (m, n, k, epi) in shapes.items():
        a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        pre = torch.randn(m, n, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, pre=pre, out=out)
        times, ref = {3: [], 1: []}, None
        for rnd in range(4):
            for v in (3, 1):                      # 3 = persistent, plain order; 1 = persistent, banded
                lib.p4p_set_gemm_persistent(v)
                f(); torch.cuda.synchronize()
                if rnd == 0:
                    if ref is None: ref = out.clone()
                    else: assert torch.equal(out, ref), (name, 'banded order differs', float((out.float() - ref.float()).abs().max()))
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): f()
                e.record(); torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / 10 * 1e3)
        print(M, name, {{3: 'plain', 1: 'banded'}[v]: round(min(t), 1) for v, t in times.items()}, 'us', flush=True)
lib.p4p_set_gemm_persistent(1)
