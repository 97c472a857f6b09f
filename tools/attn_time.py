"""Attention forward / backward times at a few shapes (for A/Bs of two builds: P4P_DEBUG_LIB=... python tools/attn_time.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
if os.environ.get('P4P_DEBUG_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
lib = _lib.lib()


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


tag = os.environ.get('P4P_DEBUG_LIB', 'product')[-20:]
for B, T, H in ((256, 257, 12), (256, 513, 12), (8, 2049, 16), (16, 2049, 16), (64, 1025, 12)):
    D = 64 * H
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
    out, lse = ops.attention_fwd(qkv, B, T, H)
    res = []
    for rnd in range(3):
        f = timeit(lambda: ops.attention_fwd(qkv, B, T, H))
        b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
        res.append(f'{f:.1f}/{b:.1f}')
    print(f'{tag} B={B} T={T} H={H}: forward/backward us: ' + '  '.join(res), flush=True)
    del qkv, dout, out, lse
