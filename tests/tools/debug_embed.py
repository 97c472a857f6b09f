import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import p4p_amd
from p4p_amd import ops
from oracle import pix4point_oracle as O
from tests.emulate import p3embed_emulated

def rel(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

B, N, k = 3, 512, 16
ora = O.P3Embed(in_channels=3, k=k)
sd = O.seeded_state_dict(ora, seed=1)
m = p4p_amd.P3Embed(in_channels=3, k=k); m.load_state_dict(sd); m = m.cuda()
pts = O.synthetic_clouds(B, N, seed=2).cuda()
start = (torch.arange(B) * 3).cuda()
m.train()
cen, tok = m.tokens(pts, pts, start)
w = torch.randn(tok.shape, generator=torch.Generator().manual_seed(3)).cuda()
(tok * w).sum().backward()
idx = ops.knn(pts, cen, k)
params = {n.replace('convs.0.', ''): p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
te = p3embed_emulated(pts, idx, params, k)
(te * w.view(-1, 256)).sum().backward()
print('tokens vs emulation', rel(tok.reshape(-1, 256), te))
for n, p in m.named_parameters():
    q = params[n.replace('convs.0.', '')]
    print(f'{n:24s} rel={rel(p.grad, q.grad):.4f} |ref|={float(q.grad.norm()):.4g} |got|={float(p.grad.norm()):.4g}')
