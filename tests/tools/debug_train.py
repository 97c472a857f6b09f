import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import p4p_amd
from oracle import pix4point_oracle as O
from tests.test_gpu_02_model import build_pair, rel_l2

D, L, H, B, N = 192, 2, 3, 32, 256
oracle, model = build_pair(40, D, L, H, seed=11)
for m in list(oracle.modules()) + list(model.modules()):
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
oracle.train(); model.train()
pts = O.synthetic_clouds(B, N, seed=5); start = torch.arange(B) * 3
labels = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2))
mode = sys.argv[1] if len(sys.argv) > 1 else 'ce'
def loss_fn(m, pts, start, labels):
    if mode == 'ce':
        return torch.nn.functional.cross_entropy(m(pts, start), labels)
    feat = m.model.forward_cls_feat(pts, start_idx=start) if isinstance(m, p4p_amd.Pix4Point) else m.model.forward_cls_feat(pts, start)
    w = torch.rand(feat.shape, generator=torch.Generator().manual_seed(9)).to(feat.device) + 0.5
    return (feat * w).sum() / feat.shape[0]
lo = loss_fn(oracle, pts, start, labels); lo.backward()
lg = loss_fn(model, pts.cuda(), start.cuda(), labels.cuda()); lg.backward()
print('loss', float(lo), float(lg))
og = dict(oracle.named_parameters())
for name, p in model.named_parameters():
    r = og[name].grad
    if r is None or p.grad is None: continue
    print(f'{name:48s} rel={rel_l2(p.grad, r):.4f} |ref|={float(r.norm()):.3e}')
