"""Debug aid: where a trained checkpoint loaded into the oracle starts to differ from the trained HIP model (eval mode)."""
import os, sys, tempfile, torch
sys.path.insert(0, '.')
import p4p_amd
from oracle import pix4point_oracle as O
mc = {'model_name': 'Pix4Point', 'num_classes': 8, 'pretrained_model': 'vit_custom_d192_l2_h3', 'pretrained': False, 'frozen': False, 'k_neighbors': 16, 'embed_dim': 192}
dc = {'root_dir': 'synthetic', 'split': 'training', 'variant': 'main_split', 'augmentation': 'none', 'num_points': 256, 'num_samples': 64, 'normalize': True}
tc = {'batch_size': 16, 'learning_rate': 2e-3, 'weight_decay': 5e-2, 'save_interval': 1, 'epochs': 2, 't_max': 10, 'warmup_epochs': 1, 'min_lr': 1e-6, 'grad_norm_clip': 10}
out = tempfile.mkdtemp()
torch.manual_seed(0)
tr = p4p_amd.Pix4PointTrainer(mc, dc, tc, device='cuda', output_dir=out)
tr.train()
sd = {k: v.cpu() for k, v in tr.model.state_dict().items()}
oracle = O.Pix4Point(num_classes=8, embed_dim=192, depth=2, heads=3, k_neighbors=16)
print(oracle.load_state_dict(sd, strict=True))
pts, _ = next(iter(tr.test_loader)); start = torch.zeros(pts.shape[0], dtype=torch.long)
oracle.eval(); tr.model.eval()
def rel(a, b): return float((a.cpu().double() - b.double()).norm() / b.double().norm())
with torch.no_grad():
    po, fo = oracle.model.patch_embed(pts, pts.transpose(1, 2).contiguous(), start)
    pg, fg = tr.model.model.patch_embed(pts.cuda(), pts.cuda().transpose(1, 2).contiguous(), start.cuda())
    print('tokens', rel(fg[-1], fo[-1]))
    print('logits', rel(tr.model(pts.cuda(), start.cuda()), oracle(pts, start)))
for n, b in tr.model.named_buffers():
    print(n, tuple(b.shape), float(b.float().abs().mean()))
