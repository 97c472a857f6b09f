"""Lazy rescale in the streaming attention forward (attn_fwd_q_kernel: the default, threshold 2^4) against the exact running
maximum (p4p_set_attention_fwd_variant bit 3), same process, same box.  usage (GPU box): python3 tools/attn_fwd_thr_ab.py       env: AF_RUNS (30)"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('adapting-2d-vits-for-3d-point-cloud-understanding_amd')
ops = importlib.import_module(pkg.__name__ + '.ops'); _lib = importlib.import_module(pkg.__name__ + '._lib')
lib = _lib.lib()
RUNS = int(os.environ.get('AF_RUNS', 30))


def timed(qkv, B, T, H, v):
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
    return ts[len(ts) // 2], ts[0], out.clone(), lse.clone()


for (B, T, H, scale) in [(256, 513, 12, 1.0), (8, 2049, 16, 1.0), (256, 513, 12, 2.0)]:
    D = 64 * H
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * scale).to(torch.bfloat16).cuda()
    for rep in range(2):
        res = {}
        for name, v in [('exact maximum', 8), ('lazy rescale 2^4', 0), ('lazy, 4 tiles / wave', 16), ('lazy, 2 tiles / wave', 32)]:
            med, mn, out, lse = timed(qkv, B, T, H, v)
            res[name] = (out, lse)
            print(f'B={B} T={T} H={H} x{scale} {name:22s} median {med:7.1f} us  min {mn:7.1f} us', flush=True)
    a, b = res['exact maximum'], res['lazy rescale 2^4']
    print('   max |out diff|', (a[0].float() - b[0].float()).abs().max().item(), ' lse diff', (a[1] - b[1]).abs().max().item())
