"""FPS timing at several (B, N) (development aid; P4P_FPS_WIDE_FROM switches the 1024-thread variant)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops
for B, N in ((256, 1024), (256, 2048), (64, 4096), (8, 8192), (64, 8192)):
    pts = torch.rand(B, N, 3, device='cuda'); start = torch.zeros(B, dtype=torch.long, device='cuda')
    for _ in range(3): ops.fps(pts, N // 4, start)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.fps(pts, N // 4, start)
    e1.record(); torch.cuda.synchronize()
    print(f'B={B} N={N}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us')
