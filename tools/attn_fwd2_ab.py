"""T = 513 attention forward: the K/V-resident kernel (attention_fwd2.h) against the streaming kernel it replaces, same process, same box.
usage (on the GPU box): python3 tools/attn_fwd2_ab.py            env: AF_B (256), AF_H (12), AF_RUNS (30)"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('adapting-2d-vits-for-3d-point-cloud-understanding_amd')
from importlib import import_module
ops = import_module(pkg.__name__ + '.ops'); _lib = import_module(pkg.__name__ + '._lib')
lib = _lib.lib()
B, H, T = int(os.environ.get('AF_B', 256)), int(os.environ.get('AF_H', 12)), 513
RUNS = int(os.environ.get('AF_RUNS', 30))
D = 64 * H
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.0).to(torch.bfloat16).cuda()


def timed(v):
    lib.p4p_set_attention_fwd_variant(v)
    for _ in range(3):
        out, lse = ops.attention_fwd(qkv, B, T, H)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(RUNS + 1)]
    ev[0].record()
    for i in range(RUNS):
        ops.attention_fwd(qkv, B, T, H)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(RUNS))
    lib.p4p_set_attention_fwd_variant(0)
    return ts[len(ts) // 2], ts[0], out, lse


flop = 4.0 * B * H * T * T * 64
res = {}
for rep in range(2):
    for name, v in [('streaming q<2>', 0), ('kv-resident walk', 1), ('kv-resident wg/pair', 3), ('kv-resident walk, no shift', 5)]:
        med, mn, out, lse = timed(v)
        res[name] = (out, lse)
        print(f'{name:28s} median {med:7.1f} us  min {mn:7.1f} us  {flop / med * 1e-6:6.0f} TFLOP/s', flush=True)
a, b = res['streaming q<2>'], res['kv-resident walk']
print('max |out diff| between the two contracts:', (a[0].float() - b[0].float()).abs().max().item(), ' lse:', (a[1] - b[1]).abs().max().item())
print('walk == wg/pair bitwise:', torch.equal(res['kv-resident walk'][0], res['kv-resident wg/pair'][0]), torch.equal(res['kv-resident walk'][1], res['kv-resident wg/pair'][1]))
