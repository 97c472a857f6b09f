"""Development aid: N steps of the contract workload on one batch, printing the loss (memorisation check at full size)."""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = types.SimpleNamespace(classes=40, dim=768, vit='vit_base_patch16_384')
dev = torch.device('cuda', 0)
for name, cls, extra in (('pix4point', bench.Stepper, dict(points=1024, batch=256)),
                         ('apf', bench.ApfStepper, dict(points=2048, batch=128, vit='vit_base_patch16_224', classes=15))):
    a = types.SimpleNamespace(**{**vars(args), **extra})
    st = cls(a, dev, False)
    batch = bench.synthetic_batch(a.batch, a.points, a.classes, 1234, dev)
    losses = []
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        losses.append(float(st.step(*batch).detach()))
    print(name, ' '.join(f'{l:.3f}' for l in losses[::5]), 'finite:', all(l == l for l in losses))
    del st
    torch.cuda.empty_cache()
