"""'att' link statistic (kernel output vs the rounding-matched statement) over a sweep of sequence lengths."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import p4p_amd
from p4p_amd import ops
from tests import emulate as E
DEV = 'cuda'
B, H = int(os.environ.get('SB', 2)), int(os.environ.get('SH', 2))
Ts = list(range(1, 340)) + [384, 385, 400, 449, 512, 513, 514, 600, 641, 700, 1024, 1025, 1100, 2049]
bad = []
for T in Ts:
    D = 64 * H
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    with torch.no_grad():
        emu = E.attention_emulated(qkv.float(), B, T, H, round_output=False)
        v = E.link_stats(out, emu, 'att')
        d = (out.float() - emu).abs()
    flag = v > (1.0 if T < 1024 else 2.0)
    if flag or T % 16 == 0:
        print(T, E.attention_lone_mode(T), round(v, 3), 'max abs', float(d.max()), 'FAIL' if flag else '', flush=True)
    if flag:
        bad.append(T)
print('failing T:', bad)
