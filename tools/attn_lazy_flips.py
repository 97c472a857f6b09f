"""Does the lazy rescale change how far the attention forward is from its rounding-matched statement?  Random shapes, the same inputs through
the default (ATT_LAZY_THR = 2^4) and the exact-maximum form (p4p_set_attention_fwd_variant(8)), each against tests/emulate.py with the matching
threshold; prints the worst number of flipped numerators beyond half an output ulp (the statistic of LINK_LIMITS['att'] = 3.0) and the error
against fp32 softmax attention.      usage (GPU box): python3 tools/attn_lazy_flips.py      env: LF_CASES (40), LF_SEED (0)"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module('adapting-2d-vits-for-3d-point-cloud-understanding_amd')
ops = importlib.import_module(pkg.__name__ + '.ops'); _lib = importlib.import_module(pkg.__name__ + '._lib')
from tests import emulate as E
lib = _lib.lib()
rng = np.random.default_rng(int(os.environ.get('LF_SEED', 0)))
worst = {'lazy': [], 'exact': []}
err32 = {'lazy': [], 'exact': []}
for case in range(int(os.environ.get('LF_CASES', 40))):
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 13))
    T = int(rng.choice([int(rng.integers(300, 2100)), 513, 1728, 2049]))
    if B * H * T * T > 2.5e8:
        B, H = 1, min(H, 4)
    D = 64 * H
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    qkv = (torch.randn(B * T, 3 * D, generator=g) * float(rng.choice([1.0, 1.5, 2.5]))).to(torch.bfloat16).cuda()
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q.double() @ k.double().transpose(-1, -2)) * 0.125, -1) @ v.double()).permute(0, 2, 1, 3).reshape(B * T, D)
    unit = E.attention_flip_unit(qkv, B, T, H)
    for name, variant, thr in (('lazy', 0, 4.0), ('exact', 8, 0.0)):
        lib.p4p_set_attention_fwd_variant(variant)
        E.ONLINE_THR = thr
        out, _ = ops.attention_fwd(qkv, B, T, H)
        with torch.no_grad():
            emu = E.attention_emulated(qkv.float(), B, T, H, round_output=False)
        worst[name].append(E.link_stats(out, emu, 'att', unit))
        err32[name].append(float((out.double() - ref).abs().max()))
    lib.p4p_set_attention_fwd_variant(0)
    E.ONLINE_THR = 4.0
    print(f'B={B} T={T} H={H}: flips lazy {worst["lazy"][-1]:.2f} exact {worst["exact"][-1]:.2f}   |out - fp64 softmax| lazy {err32["lazy"][-1]:.2e} exact {err32["exact"][-1]:.2e}', flush=True)
for name in ('lazy', 'exact'):
    w = np.array(worst[name]); e = np.array(err32[name])
    print(f'{name:6s}: flips median {np.median(w):.2f}  90 % {np.quantile(w, 0.9):.2f}  max {w.max():.2f};  error vs fp64 softmax: median {np.median(e):.2e} max {e.max():.2e}')
