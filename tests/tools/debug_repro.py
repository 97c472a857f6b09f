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
model = p4p_amd.Pix4Point(num_classes=40, embed_dim=768, pretrained_model='vit_base_patch16_384', pretrained=False).to(DEV)
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

for trial in range(3):
    l1, g1 = run(); l2, g2 = run()
    bad = [(n, float((g1[n] - g2[n]).abs().max()), float(g1[n].abs().max())) for n in g1 if not torch.equal(g1[n], g2[n])]
    print('trial', trial, 'loss equal', torch.equal(l1, l2), 'differing', len(bad), 'of', len(g1))
    for b in bad[:60]:
        print('   ', b)

# attention alone
T, H = N // 4 + 1, 12
qkv = torch.randn(B * T, 3 * 768, device=DEV).bfloat16()
do = torch.randn(B * T, 768, device=DEV).bfloat16()
for (b, t) in ((B, T), (8, 2049), (B, 257)):
    qkv = torch.randn(b * t, 3 * 768, device=DEV).bfloat16()
    do = torch.randn(b * t, 768, device=DEV).bfloat16()
    res = []
    for i in range(4):
        o, lse = ops.attention_fwd(qkv, b, t, H)
        dqkv, cs = ops.attention_bwd(qkv, o, do, lse, b, t, H, want_colsum=True)
        res.append((o.clone(), lse.clone(), dqkv.clone(), cs.clone()))
    for i in range(1, 4):
        print('attention T', t, 'run', i, [bool(torch.equal(a, c)) for a, c in zip(res[0], res[i])],
              [float((a.float() - c.float()).abs().max()) for a, c in zip(res[0], res[i])])
        d = (res[0][2].float() - res[i][2].float()).abs().reshape(b, t, 3, 768).amax(dim=(0, 3))
        nz = d.nonzero()
        print('    differing (row, which) count', len(nz), nz[:10].tolist())
