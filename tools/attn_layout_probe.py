"""Does the row stride of q | k | v / dO cost bandwidth?  The same 3,072 (batch, head) pairs at T = 257 (or AT_T) as B = 256 x H = 12
(rows of 128 B at a stride of 4,608 B; dO / O at 1,536 B) and as B = 3,072 x H = 1 (stride 384 B; dO / O contiguous): identical work
and bytes per workgroup, only the addresses differ."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops

T = int(os.environ.get('AT_T', 257))


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for rnd in range(2):
    for B, H in ((256, 12), (3072, 1), (512, 6), (1536, 2)):
        D = 64 * H
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
        dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
        out, lse = ops.attention_fwd(qkv, B, T, H)
        f = timeit(lambda: ops.attention_fwd(qkv, B, T, H))
        b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
        print(f'round {rnd} T={T} B={B:5d} H={H:2d}: forward {f:7.1f} us   backward {b:7.1f} us', flush=True)
        del qkv, dout, out, lse
