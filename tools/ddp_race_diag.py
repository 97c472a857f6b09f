"""Diagnostic: is a backward pass of the tiny model bit-reproducible while ANOTHER process runs the same
work on the same GPU (what the two-rank test does)?  Two spawned processes, each 10 forward + backward passes of its own model
and batch with no synchronisation between them; every pass is compared with the process's first.  Prints, per process, the
parameters whose gradient ever differed (max|delta|, max|g|) and whether the gradient the stock-torch head hands the encoder
(captured at the top block's input) differed -- which separates "the head's library kernels are timing dependent" from "one of
this library's kernels is"."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, q):
    sys.path.insert(0, ROOT)
    import p4p_amd
    from p4p_amd import functional as F
    from oracle import pix4point_oracle as O            # synthetic clouds only
    dev = 'cuda:0'
    torch.manual_seed(10)
    model = p4p_amd.Pix4Point(num_classes=40, embed_dim=192, pretrained_model='vit_custom_d192_l2_h3', pretrained=False).to(dev)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    pts = O.synthetic_clouds(8, 256, seed=50 + rank).to(dev)
    start = (torch.arange(8) * 3).to(dev)
    labels = torch.randint(0, 40, (8,), generator=torch.Generator().manual_seed(rank)).to(dev)
    bufs = [b.detach().clone() for b in model.buffers()]
    first, first_up, lines = None, None, []
    diff = {}
    for it in range(10):
        model.zero_grad(set_to_none=True)
        F._CAPTURE_BWD = cap = {}
        logits = model(pts, start)
        torch.nn.functional.cross_entropy(logits, labels).backward()
        F._CAPTURE_BWD = None
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        up = cap['block_dout'][1].clone()
        lg = logits.detach().clone()
        for b, v in zip(model.buffers(), bufs):
            b.copy_(v)
        if first is None:
            first, first_up, first_lg = g, up, lg
            continue
        du, dl = float((up - first_up).abs().max()), float((lg - first_lg).abs().max())
        if du or dl:
            lines.append(f'rank {rank} pass {it}: logits delta {dl:.3e}, encoder upstream gradient delta {du:.3e} (max {float(first_up.abs().max()):.3e})')
        for n in g:
            d = float((g[n] - first[n]).abs().max())
            if d:
                diff[n] = max(diff.get(n, 0.0), d)
    for n, d in sorted(diff.items(), key=lambda kv: -kv[1] / max(float(first[kv[0]].abs().max()), 1e-30)):
        lines.append(f'rank {rank}: {n:44s} max|delta| {d:.3e}  max|g| {float(first[n].abs().max()):.3e}')
    lines.append(f'rank {rank}: {len(diff)} of {len(first)} gradients ever differed from the first pass')
    q.put('\n'.join(lines))


if __name__ == '__main__':
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, q)) for r in range(nproc)]
    for p in ps:
        p.start()
    for _ in ps:
        print(q.get(timeout=300))
    for p in ps:
        p.join(timeout=60)
