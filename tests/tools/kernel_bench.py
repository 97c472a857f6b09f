"""Per-kernel timing at the BASELINE config-2 shapes (B=256, N=1024, ViT-B): prints TFLOP/s or GB/s
per kernel.  Development aid; bench.py is the contract benchmark."""
import json
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import p4p_amd                      # noqa: E402
from p4p_amd import ops             # noqa: E402
from oracle import pix4point_oracle as O    # noqa: E402

dev = 'cuda'


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    res = {}
    B, N, T, D, H = 256, 1024, 257, 768, 12
    M = B * T
    g = torch.Generator().manual_seed(0)
    pts = O.synthetic_clouds(B, N, seed=1).to(dev)
    start = torch.zeros(B, dtype=torch.long, device=dev)
    t = timeit(lambda: ops.fps(pts, N // 4, start))
    res['fps_B256_N1024'] = {'ms': t * 1e3}
    idx = ops.fps(pts, N // 4, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    t = timeit(lambda: ops.knn(pts, cen, 16))
    res['knn_B256_N1024_k16'] = {'ms': t * 1e3}
    for name, (m, n, k, epi) in {
        'qkv': (M, 3 * D, D, ops.EPI_BF16), 'proj': (M, D, D, ops.EPI_F32_RESID),
        'fc1': (M, 4 * D, D, ops.EPI_BF16_GELU), 'fc2': (M, D, 4 * D, ops.EPI_F32_RESID),
        'embed_conv2a': (B * 256 * 16, 512, 256, ops.EPI_BF16), 'embed_conv2b': (B * 256 * 16, 256, 512, ops.EPI_BF16),
    }.items():
        a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        resid = torch.randn(m, n, device=dev) if epi == ops.EPI_F32_RESID else None
        out = torch.empty(m, n, dtype=torch.float32 if epi == ops.EPI_F32_RESID else torch.bfloat16, device=dev)
        t = timeit(lambda: ops.gemm_tn(a, w, epi, bias=bias, resid=resid, out=out))
        res[f'gemm_tn_{name}_{m}x{n}x{k}'] = {'ms': t * 1e3, 'TFLOPs': 2 * m * n * k / t * 1e-12}
        a2 = a.float().to(torch.bfloat16)
        t2 = timeit(lambda: torch.matmul(a2, w.t()))
        res[f'torch_matmul_{name}'] = {'ms': t2 * 1e3, 'TFLOPs': 2 * m * n * k / t2 * 1e-12}
        gg = torch.randn(m, n, generator=g).to(torch.bfloat16).to(dev)
        dw = torch.empty(n, k, device=dev)
        t = timeit(lambda: ops.gemm_wgrad(gg, a, out=dw))
        res[f'gemm_wgrad_{name}_{n}x{k}_M{m}'] = {'ms': t * 1e3, 'TFLOPs': 2 * m * n * k / t * 1e-12}
        del a, w, gg, out, resid
    qkv = torch.randn(M, 3 * D, generator=g).to(torch.bfloat16).to(dev)
    t = timeit(lambda: ops.attention_fwd(qkv, B, T, H))
    fl = 4 * B * H * T * T * 64
    res['attn_fwd'] = {'ms': t * 1e3, 'TFLOPs': fl / t * 1e-12}
    out, lse = ops.attention_fwd(qkv, B, T, H)
    dout = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    t = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
    res['attn_bwd'] = {'ms': t * 1e3, 'TFLOPs(5 products)': 2.5 * fl / t * 1e-12}
    x = torch.randn(M, D, device=dev)
    pos = torch.randn(M, D, device=dev)
    gamma = torch.ones(D, device=dev)
    beta = torch.zeros(D, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, gamma, beta, 1e-6, pos=pos, want_sum=True))
    res['layernorm_fwd'] = {'ms': t * 1e3, 'GB/s': (M * D * (4 + 4 + 4 + 2)) / t * 1e-9}
    y, xs, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, pos=pos, want_sum=True)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dx = torch.zeros(M, D, device=dev)
    t = timeit(lambda: ops.layernorm_bwd(dy, xs, mean, rstd, gamma, dx=dx))
    res['layernorm_bwd'] = {'ms': t * 1e3, 'GB/s': (M * D * (2 + 4 + 4 + 4)) / t * 1e-9}
    for k, v in res.items():
        print(k, json.dumps(v))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/kernel_bench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
