"""Development aid: which rows / columns of the attention forward differ from fp32 softmax attention."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p4p_amd import ops
B, T, H = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 65, 1
D = 64 * H
g = torch.Generator().manual_seed(T)
qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.0).to(torch.bfloat16).cuda()
out, lse = ops.attention_fwd(qkv, B, T, H)
q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
a = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1)
ref = (a @ v).transpose(1, 2).reshape(B * T, D)
err = (out.float() - ref).abs()
print('rows with max err > 0.02:', (err.max(1).values > 0.02).nonzero().flatten().tolist()[:80])
print('err by 16-col block:', [round(float(err[:, c:c + 16].max()), 3) for c in range(0, 64, 16)])
print('err by col mod 16:', [round(float(err[:, c::16].max()), 3) for c in range(16)])
print('row 0 out', out[0, :8].float().tolist(), 'ref', ref[0, :8].tolist())
lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * 0.125, -1) * 1.4426950408889634
print('lse err', float((lse - lse_ref).abs().max()))
s = (q @ k.transpose(-1, -2)) * 0.125
def cand(name, a):
    r = (a @ v).transpose(1, 2).reshape(B * T, D)
    print(f'{name:40s} max err main rows {float((out.float() - r)[:T-1].abs().max()):.4f}  last row {float((out.float() - r)[T-1:].abs().max()):.4f}')
cand('exact', s.softmax(-1))
sx = s.clone(); sx[..., T - 1] = float('-inf'); cand('last key excluded', sx.softmax(-1))
sx = s.clone(); sx[..., T - 1] += 0.6931; cand('last key doubled', sx.softmax(-1))
sx = s.clone(); sx[..., T - 1] *= 4; cand('last key score x4', sx.softmax(-1))
sx = s.clone(); sx[..., T - 1] /= 4; cand('last key score /4', sx.softmax(-1))
cand('unscaled scores', (s * 8).softmax(-1))
perm = torch.arange(64).view(4, 16).t().reshape(-1)
print('lse (kernel) row0..3', lse.flatten()[:4].tolist(), 'ref', lse_ref.flatten()[:4].tolist())
