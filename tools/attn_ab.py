import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
if os.environ.get('P4P_DEBUG_LIB'):          # A/B of two builds of the library in one gpurun call
    _lib.LIB_PATH = os.path.abspath(os.environ['P4P_DEBUG_LIB'])
_lib.lib().p4p_set_attention_variant(int(os.environ.get('AT_VARIANT', '0')))   # bit 0: no resident fwd/dQ, 1: no resident dK/dV, 2: no two-pass fwd
B, T, H, D = 256, int(os.environ.get('AT_T', 257)), 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
out, lse = ops.attention_fwd(qkv, B, T, H)
print('fwd us', timeit(lambda: ops.attention_fwd(qkv, B, T, H)))
print('bwd us', timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H)))
