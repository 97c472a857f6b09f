"""Diagnostic for tests/test_gpu_09_ddp.py: the same tiny model, the same batch, four backward passes in ONE
process -- reserved CUs 0, 0, 32, 32 -- and per parameter max|g|, max|delta| between the repeats (run-to-run determinism) and
between r = 0 and r = 32 (the weight-gradient M-split order follows the reserved CUs).  Prints the parameters sorted by
delta / max(max|g|, 1e-6), the metric that failed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p4p_amd                                      # noqa: E402
from p4p_amd import ops                             # noqa: E402
from oracle import pix4point_oracle as O            # noqa: E402  (synthetic clouds only)


def grads(model, pts, start, labels, r):
    ops.set_reserved_cus(r)
    bufs = [b.detach().clone() for b in model.buffers()]
    model.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(model(pts, start), labels).backward()
    out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    for b, v in zip(model.buffers(), bufs):
        b.copy_(v)
    ops.set_reserved_cus(0)
    return out


def main():
    dev = 'cuda:0'
    torch.manual_seed(10)
    model = p4p_amd.Pix4Point(num_classes=40, embed_dim=192, pretrained_model='vit_custom_d192_l2_h3', pretrained=False).to(dev)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    pts = O.synthetic_clouds(8, 256, seed=50).to(dev)
    start = (torch.arange(8) * 3).to(dev)
    labels = torch.randint(0, 40, (8,), generator=torch.Generator().manual_seed(0)).to(dev)
    a = grads(model, pts, start, labels, 0)         # cold: first call of every stock-torch op
    b = grads(model, pts, start, labels, 0)
    c = grads(model, pts, start, labels, 32)
    d = grads(model, pts, start, labels, 32)
    gmax = max(float(v.abs().max()) for v in a.values())
    rows = []
    for n in a:
        w = float(a[n].abs().max())
        dab, dcd, dac = (float((x[n] - y[n]).abs().max()) for x, y in ((a, b), (c, d), (b, c)))
        rows.append((max(dab, dcd, dac) / max(w, 1e-6), n, w, dab, dcd, dac))
    rows.sort(reverse=True)
    print(f'global max|g| = {gmax:.3e}; columns: old metric, name, max|g|, cold-vs-warm r=0, repeat r=32, r=0 vs r=32')
    for r in rows[:25]:
        print('%.3e  %-44s %.3e  %.3e  %.3e  %.3e' % r)
    print('parameters with any difference:', sum(1 for r in rows if r[0] > 0), 'of', len(rows))


if __name__ == '__main__':
    main()
