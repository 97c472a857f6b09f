"""Debug aid: stage-by-stage forward agreement of the tiny APF model (train mode) with the CPU oracle."""
import sys, torch
sys.path.insert(0, '.')
import p4p_amd
from p4p_amd import apf as PA, functional_apf as FA
from oracle import apf_oracle as A

def rel_l2(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

cfg = dict(num_classes=15, embedding_dim=192, npoint=64, nsample=32, depth=2, num_heads=3, dropout_rate=0.0, dropout_path_rate=0.0)
o = A.AdaptPointFormer(**cfg); sd = A.seeded_state_dict(o, seed=21); o.load_state_dict(sd)
m = PA.AdaptPointFormer(vit_name=None, pretrained=False, **cfg); m.load_state_dict(sd); m = m.to('cuda')
for mod in (o, m):
    mod.train()
    mod.head.eval()
    for d in mod.head.mlp_head:
        if isinstance(d, torch.nn.Dropout): d.p = 0.0
B, N = 24, 256
pts = A.synthetic_clouds(B, N, seed=22) * (0.3 + 0.7 * torch.rand(B, 1, 3, generator=torch.Generator().manual_seed(23))); start = torch.arange(B) * 7; labels = torch.arange(B) % 15
to = o.point_encoder(pts, start); tg = m.point_encoder(pts.cuda(), start.cuda())
print('tokens', rel_l2(tg, to))
bo = o.encoder_norm(o.blocks(to)); bg = FA.apf_blocks(tg, list(m.blocks), m.encoder_norm, 3)
print('blocks+norm', rel_l2(bg, bo))
bg2 = FA.apf_blocks(to.detach().cuda().requires_grad_(True), list(m.blocks), m.encoder_norm, 3)
print('blocks+norm on oracle tokens', rel_l2(bg2, bo))
po, pg = bo.max(-2)[0], bg.max(-2)[0]
print('pooled', rel_l2(pg, po))
lo, lg = o.head(po), m.head(pg)
print('logits', rel_l2(lg, lo), float(lo.abs().max()))
# head-only sensitivity: oracle head on GPU features
lo2 = o.head(pg.detach().cpu())
print('oracle head on gpu pooled vs oracle', rel_l2(lo2, lo))
losso = torch.nn.functional.cross_entropy(lo, labels, label_smoothing=0.3); losso.backward()
lossg = torch.nn.functional.cross_entropy(lg, labels.cuda(), label_smoothing=0.3); lossg.backward()
print('loss', float(losso), float(lossg))
go = dict(o.named_parameters())
for n, p in m.named_parameters():
    if p.grad is not None and go[n].grad is not None:
        print('  %-45s %.4g  |ref|=%.3g' % (n, rel_l2(p.grad, go[n].grad), float(go[n].grad.norm())))
