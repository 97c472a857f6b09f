"""Development aid: relative L2 error of the attention forward / backward against fp32 autograd."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from p4p_amd import ops
for B, T, H in ((4, 65, 3), (2, 257, 3), (2, 196, 3)):
    D = 64 * H
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * D, generator=g)).to(torch.bfloat16).cuda()
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).cuda()
    out, lse = ops.attention_fwd(qkv, B, T, H)
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B * T, D)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv, out, dout, lse, B, T, H)
    rl = lambda x, y: float((x.float() - y).norm() / y.norm())
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * 0.125, -1) * 1.4426950408889634
    gq = qr.grad.view(B, T, 3, D)
    dq = dqkv.float().view(B, T, 3, D)
    print(f'T={T}: out {rl(out, ref):.2e}  lse {float((lse - lse_ref).abs().max()):.2e}  dq {rl(dq[:, :, 0], gq[:, :, 0]):.2e} '
          f'dk {rl(dq[:, :, 1], gq[:, :, 1]):.2e} dv {rl(dq[:, :, 2], gq[:, :, 2]):.2e}  last-row dq {rl(dq[:, -1, 0], gq[:, -1, 0]):.2e} '
          f'dk {rl(dq[:, -1, 1], gq[:, -1, 1]):.2e} dv {rl(dq[:, -1, 2], gq[:, -1, 2]):.2e}')
