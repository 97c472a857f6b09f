"""T = 513 attention backward: the fused kernel of attention_fused2.h against the dQ + key-stationary dK/dV pair it replaces
(variant bit 4), interleaved rounds in one process, with and without the qkv-bias partial rows; optional start-shift sweep.
    python tools/attn_f2_ab.py [--sweep]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd
from p4p_amd import ops, _lib
lib = _lib.lib()


def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


B, T, H = int(os.environ.get('F2_B', 256)), 513, 12
D = 64 * H
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * T, 3 * D, generator=g).to(torch.bfloat16).cuda()
dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
out, lse = ops.attention_fwd(qkv, B, T, H)
variants = [('fused2', 0), ('dq_q + dkdv_ks', 16)]
if '--sweep' in sys.argv:
    variants += [(f'fused2 shift {ph}x{cyc}k', ((ph - 1) << 8) | (cyc << 10)) for ph, cyc in ((1, 0), (2, 12), (3, 6), (3, 20), (4, 10))]
for cs in (False, True):
    res = {n: [] for n, _ in variants}
    for rnd in range(4):
        for n, v in variants:
            lib.p4p_set_attention_variant(v)
            res[n].append(timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=cs)))
    lib.p4p_set_attention_variant(0)
    for n, _ in variants:
        print(f'B={B} T={T} H={H} colsum={cs} {n:24s}: ' + '  '.join(f'{x:.1f}' for x in res[n]) + ' us', flush=True)
