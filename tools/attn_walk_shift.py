"""Start-time shift of the walking attention backward (T = 257): phases x cycles sweep through the variant switch
(bits 8-9: phases - 1, bits 10..: shift in units of 1,024 cycles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()
T = int(os.environ.get('AT_T', 257))
B, H, D = 256, 12, 768
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out, lse = ops.attention_fwd(qkv, B, T, H)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


cases = [(0, 'default (3 x 10k)'), (32, 'no shift')]
for ph in (2, 3, 4):
    for kc in (6, 10, 14, 20, 28):
        cases.append((((ph - 1) << 8) | (kc << 10), f'{ph} x {kc}k'))
for rnd in range(2):
    res = []
    for v, name in cases:
        lib.p4p_set_attention_variant(v)
        a = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
        b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True))
        res.append(f'{name}: {a:.1f}/{b:.1f}')
    print(f'round {rnd} T={T} bwd / with colsum (us): ' + '  '.join(res), flush=True)
lib.p4p_set_attention_variant(0)
