"""Attention forward / backward with one workgroup (or two) per CU walking the (batch, head) pairs -- the default where there are more
pairs than workgroup slots -- against a workgroup per pair (variant bit 7): results bit for bit, then time.  AT_T=257 | 513."""
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


def run(v):
    lib.p4p_set_attention_variant(v)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    dq = ops.attention_bwd(qkv, out, dout, lse, B, T, H)
    lib.p4p_set_attention_variant(0)
    return out.clone(), lse.clone(), dq.clone()


a, b = run(0), run(128)
print('walk == workgroup per pair, bit for bit (out, lse, dqkv):', [torch.equal(x, y) for x, y in zip(a, b)], flush=True)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


out, lse = ops.attention_fwd(qkv, B, T, H)
for rnd in range(3):
    for v, name in ((128, 'a workgroup per pair'), (0, 'walk')):
        lib.p4p_set_attention_variant(v)
        f = timeit(lambda: ops.attention_fwd(qkv, B, T, H))
        bw = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H))
        print(f'round {rnd} T={T} {name:22s}: forward {f:7.1f} us   backward {bw:7.1f} us', flush=True)
lib.p4p_set_attention_variant(0)
