"""How much of the weight-gradient GEMM's time is per-workgroup overhead (prologue, 256-KB slab write, launch tail) and the slab sum?
The same four ViT-B shapes at M = 65,792 (9 / 28 / 7 / 7 M-splits of 229-294 steps: what the step runs) and at M = 8 x 65,792 (the same
grids, 8 x longer K loops per workgroup): TFLOP/s of the GEMM kernel + slab sum together.  The long-M figure is what a launch without
M-splits (whole-M tiles, batched over the blocks) could approach."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops
dev = 'cuda'
D = 768
g = torch.Generator().manual_seed(0)
for mult in (1, 8):
    M = 65792 * mult
    for name, (N, K) in {'qkv': (3 * D, D), 'proj': (D, D), 'fc1': (4 * D, D), 'fc2': (D, 4 * D)}.items():
        G = torch.randn(65792, N, generator=g).to(torch.bfloat16).to(dev).repeat(mult, 1)
        A = torch.randn(65792, K, generator=g).to(torch.bfloat16).to(dev).repeat(mult, 1)
        out = torch.empty(N, K, dtype=torch.float32, device=dev)
        f = lambda: ops.gemm_wgrad(G, A, out=out)
        f(); torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): f()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 5 * 1e-3)
        print(f'M = {M:7d} {name:5s} N={N:5d} K={K:5d}: {best * 1e6 / mult:8.1f} us per 65,792 rows, {2.0 * M * N * K / best * 1e-12:7.1f} TFLOP/s', flush=True)
        del G, A
