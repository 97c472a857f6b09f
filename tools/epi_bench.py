"""Epilogue-only timing: the ViT-B GEMM shapes with K = 64 (one ring step), so the call is prologue + epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
dev = 'cuda'
lib = _lib.lib()
M, D = 256 * 257, 768
g = torch.Generator().manual_seed(0)
lib.p4p_set_gemm_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
for K in (64, 128, 768):
    for name, n, epi in (('bf16 N=2304', 3 * D, ops.EPI_BF16), ('resid N=768', D, ops.EPI_F32_RESID), ('gelu N=3072', 4 * D, ops.EPI_BF16_GELU),
                         ('dgelu N=3072', 4 * D, ops.EPI_BF16_DGELU), ('bf16 N=768', D, ops.EPI_BF16)):
        a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = torch.randn(n, K, generator=g).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        resid = torch.randn(M, n, device=dev) if epi == ops.EPI_F32_RESID else None
        pre = torch.randn(M, n, device=dev).to(torch.bfloat16) if epi in (ops.EPI_BF16_GELU, ops.EPI_BF16_DGELU) else None
        out = torch.empty(M, n, dtype=torch.float32 if epi == ops.EPI_F32_RESID else torch.bfloat16, device=dev)
        f = lambda: ops.gemm_tn(a, w, epi, bias=None if epi == ops.EPI_BF16_DGELU else bias, resid=resid, pre=pre, out=out)
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 10 * 1e3
        mb = (M * K * 2 + M * n * (4 if epi == ops.EPI_F32_RESID else 2) * (2 if epi != ops.EPI_BF16 else 1)) / 1e6
        print(f'K={K:4d} {name:14s} {t:8.1f} us   {mb:7.0f} MB moved -> {mb / t / 1e3 * 1e3:6.2f} GB/ms = TB/s x1e-3... {mb/t:6.2f} TB/s'.replace('GB/ms = TB/s x1e-3... ', ''), flush=True)
