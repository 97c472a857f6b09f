import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
if os.environ.get('P4P_DEBUG_LIB'):          # A/B of two builds of the library in one gpurun call
    _lib.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
lib = _lib.lib()
M, D = int(os.environ.get('GB_M', 256 * 257)), 768
g = torch.Generator().manual_seed(0)
shapes = {'qkv': (3 * D, D), 'proj': (D, D), 'fc1': (4 * D, D), 'fc2': (D, 4 * D)}
for name, (n, k) in shapes.items():
    G = torch.randn(M, n, generator=g).to(torch.bfloat16).cuda()
    A = torch.randn(M, k, generator=g).to(torch.bfloat16).cuda()
    dw = torch.empty(n, k, device='cuda')
    res, ref = {}, None
    for rnd in range(3):
        for v in (2, 3, 6, 10):           # gemm_wgrad3.h (hand-interleaved ring) / gemm_wgrad4.h (staggered 4-phase schedule) / gemm_wgrad5.h (4 waves of 128 x 128: hand-interleaved, compiler-scheduled)
            lib.p4p_set_wgrad_wide(v)
            ops.gemm_wgrad(G, A, out=dw); torch.cuda.synchronize()
            if rnd == 0:                   # different M-splits, different summation order: close, not equal
                if ref is None: ref = dw.clone()
                else: assert (ref - dw).abs().max().item() <= 2e-4 * ref.abs().max().item(), (name, (ref - dw).abs().max().item())
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm_wgrad(G, A, out=dw)
            e.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(s.elapsed_time(e) / 10 * 1e-3)
    print(name, {v: (round(2 * M * n * k / min(t) * 1e-12), f'{min(t) * 1e6:.1f} us') for v, t in res.items()}, '(main kernel + slab sum)', flush=True)
lib.p4p_set_wgrad_wide(2)
