"""Which gradients differ between two identical train steps (diagnostic for the bitwise-reproducibility test)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import p4p_amd
from p4p_amd import ops
from oracle import pix4point_oracle as O

DEV = 'cuda'
B, N = int(os.environ.get('RB', 128)), int(os.environ.get('RN', 2048))
torch.manual_seed(0)
VIT = os.environ.get('RVIT', 'vit_base_patch16_384')
model = p4p_amd.Pix4Point(num_classes=40, embed_dim=1024 if 'large' in VIT else 768, pretrained_model=VIT, pretrained=False).to(DEV)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
    if hasattr(m, 'drop_prob'):
        m.drop_prob = 0.0
model.train()
pts = O.synthetic_clouds(B, N, seed=1).to(DEV)
start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(3)).to(DEV)
labels = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(4)).to(DEV)
bufs = [b.detach().clone() for b in model.buffers()]

def run():
    for b, v in zip(model.buffers(), bufs):
        b.copy_(v)
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(pts, start), labels)
    loss.backward()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

RUNS = int(os.environ.get('RUNS', 30))
l1, g1 = run()
nbad = 0
for trial in range(RUNS):
    l2, g2 = run()
    bad = [(n, float((g1[n] - g2[n]).abs().max()), float(g1[n].abs().max())) for n in g1 if not torch.equal(g1[n], g2[n])]
    if bad or not torch.equal(l1, l2):
        nbad += 1
        print('run', trial, 'loss equal', torch.equal(l1, l2), 'differing', len(bad), 'of', len(g1), 'last differing:', bad[-1] if bad else None)
print('B', B, 'N', N, 'differing runs', nbad, 'of', RUNS)
