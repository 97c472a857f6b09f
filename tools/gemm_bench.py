"""A/B of gemm_tn tile configurations at the ViT-B shapes (interleaved rounds in ONE process, random data)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
dev = 'cuda'
if os.environ.get('P4P_DEBUG_LIB'):          # A/B of two builds of the library in one gpurun call
    _lib.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
lib = _lib.lib()
M, D = int(os.environ.get("GB_M", 256 * 257)), 768
shapes = {'qkv': (M, 3 * D, D, ops.EPI_BF16), 'proj': (M, D, D, ops.EPI_F32_RESID), 'fc1': (M, 4 * D, D, ops.EPI_BF16_GELU),
          'fc2': (M, D, 4 * D, ops.EPI_F32_RESID), 'dfc2': (M, 4 * D, D, ops.EPI_BF16_DGELU), 'dfc1': (M, D, 4 * D, ops.EPI_BF16)}
variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,2,3,11,12,13'.split(','))]
g = torch.Generator().manual_seed(0)
res = {}
only = os.environ.get('GB_SHAPES')
for name, (m, n, k, epi) in shapes.items():
    if only and name not in only.split(','): continue
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    resid = torch.randn(m, n, device=dev) if epi == ops.EPI_F32_RESID else None
    pre = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi in (ops.EPI_BF16_GELU, ops.EPI_BF16_DGELU) else None
    out = torch.empty(m, n, dtype=torch.float32 if epi == ops.EPI_F32_RESID else torch.bfloat16, device=dev)
    ref = None
    times = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            lib.p4p_set_gemm_variant(v)
            f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, resid=resid, pre=pre, out=out)
            f(); torch.cuda.synchronize()
            if rnd == 0:
                o = out.float().clone()
                if ref is None: ref = o
                elif v < 20 and not os.environ.get('GB_NOCHECK'): assert torch.allclose(o, ref, rtol=2e-2, atol=2e-2), (name, v, float((o - ref).abs().max()))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): f()
            e.record(); torch.cuda.synchronize()
            times[v].append(s.elapsed_time(e) / 10 * 1e-3)
    res[name] = {v: 2 * m * n * k / min(ts) * 1e-12 for v, ts in times.items()}
    print(name, {v: round(x) for v, x in res[name].items()}, flush=True)
lib.p4p_set_gemm_variant(0)
json.dump(res, open('gpurun_out/gemm_bench.json', 'w'))
