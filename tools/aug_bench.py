"""Timing of the batched augmentation kernel (development aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import augment as GA
for B, N in ((512, 1024), (256, 2048), (64, 8192)):
    pts = torch.rand(B, N, 3, device='cuda')
    aug = GA.TrainAugment(seed=1)
    for _ in range(3): aug(pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): aug(pts)
    e1.record(); torch.cuda.synchronize()
    full = e0.elapsed_time(e1) / 20 * 1e3
    flags = torch.full((B,), 15, dtype=torch.int32, device='cuda')
    W = GA._affine(None, None, B, 'cuda'); j = torch.randn(B, N, 3, device='cuda') * 0.03
    m = torch.rand(B, N, device='cuda') < 0.2; u = torch.rand(B, N, 3, device='cuda'); g = torch.randn(B, N, 3, device='cuda')
    from p4p_amd import ops
    for _ in range(3): ops.augment(pts, flags, affine=W, jitter=j, replace_mask=m, replace_u=u, replace_g=g)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): ops.augment(pts, flags, affine=W, jitter=j, replace_mask=m, replace_u=u, replace_g=g)
    e1.record(); torch.cuda.synchronize()
    k = e0.elapsed_time(e1) / 20 * 1e3
    byt = B * N * 3 * 4 * 5 + B * N
    print(f'B={B} N={N}: TrainAugment (draws + kernel) {full:.0f} us, kernel alone {k:.1f} us = {byt / k * 1e-3:.0f} GB/s')
