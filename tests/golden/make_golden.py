"""Generates the golden fixtures in this directory from the REFERENCE itself.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It loads ``/root/reference/src/models/pix4point.py`` by file path (SURVEY.md section 8c recipe:
the file's ``import timm`` is satisfied by an inert placeholder module because timm 1.0.16 is
not installed; ``timm.create_model`` is never called, so only the torch-only functions
``farthest_point_sampling``, ``group_knn``, ``P3Embed`` and ``ClsHead`` are usable) and records
their outputs on seeded inputs.  The fixtures are DATA (inputs + expected outputs); no
reference source travels.  Nothing here runs on the GPU box.

The reference's FPS draws its start index with an unseeded ``torch.randint``
(pix4point.py:30); to pin it we patch ``torch.randint`` for the duration of the call so the
draw returns the explicit start index stored in the fixture.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pix4point_oracle as O   # noqa: E402

REF = '/root/reference/src/models/pix4point.py'


def load_reference():
    sys.modules.setdefault('timm', types.ModuleType('timm'))
    spec = importlib.util.spec_from_file_location('ref_pix4point', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class fixed_randint:
    """Make the reference's start draw return a chosen tensor."""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self._orig = torch.randint
        v = self.value

        def fake(low, high, size, **kw):
            assert tuple(size) == tuple(v.shape)
            return v.clone().to(kw.get('dtype', torch.long))
        torch.randint = fake

    def __exit__(self, *a):
        torch.randint = self._orig


def main():
    ref = load_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------------- FPS: N in {64, 1024, 2048, 8192}, plus a duplicate-points cloud ----------
    fps = {}
    for tag, (B, N, dup) in {'n64': (4, 64, 0.0), 'n1024': (3, 1024, 0.0), 'n2048': (2, 2048, 0.0),
                             'n8192': (1, 8192, 0.0), 'dup1024': (2, 1024, 0.10),
                             'n100': (2, 100, 0.0)}.items():
        pts = O.synthetic_clouds(B, N, seed=100 + N, dup_frac=dup)
        g = torch.Generator().manual_seed(7 + N)
        start = torch.randint(0, N, (B,), generator=g)
        with fixed_randint(start):
            idx = ref.farthest_point_sampling(pts, N // 4)
        fps[f'{tag}_points'] = pts.numpy()
        fps[f'{tag}_start'] = start.numpy()
        fps[f'{tag}_idx'] = idx.numpy().astype(np.int32)      # < 2^31, stored compactly
    np.savez_compressed(os.path.join(HERE, 'fps.npz'), **fps)

    # ---------------- kNN: reference cdist+topk indices, plus the gap that flags near ties ------
    knn = {}
    for tag, (B, N, k) in {'n256k16': (2, 256, 16), 'n1024k16': (2, 1024, 16),
                           'n1024k32': (1, 1024, 32), 'n2048k16': (1, 2048, 16)}.items():
        pts = O.synthetic_clouds(B, N, seed=200 + N + k)
        start = torch.zeros(B, dtype=torch.long)
        with fixed_randint(start):
            cidx = ref.farthest_point_sampling(pts, N // 4)
        cen = torch.gather(pts, 1, cidx.unsqueeze(-1).expand(-1, -1, 3))
        gp, gf = ref.group_knn(pts, cen, pts, k)
        dist = torch.cdist(cen, pts)
        ref_idx = dist.topk(k=k, dim=-1, largest=False, sorted=True).indices.int()
        assert torch.equal(gp, pts[torch.arange(B).view(B, 1, 1), ref_idx.long()])
        knn[f'{tag}_points'] = pts.numpy()
        knn[f'{tag}_cidx'] = cidx.numpy().astype(np.int32)
        knn[f'{tag}_ref_idx'] = ref_idx.numpy().astype(np.int16)
        knn[f'{tag}_min_gap'] = O.knn_min_gap(cen, pts, k).numpy()
    np.savez_compressed(os.path.join(HERE, 'knn.npz'), **knn)

    # ---------------- P3Embed: eval tokens, train tokens + a few gradients ----------------------
    emb = {}
    for tag, (B, N, k) in {'b2n256': (2, 256, 16), 'b3n512': (3, 512, 16)}.items():
        m = ref.P3Embed(in_channels=3, k=k)
        sd = O.seeded_state_dict(m, seed=300 + N)
        m.load_state_dict(sd)
        pts = O.synthetic_clouds(B, N, seed=310 + N)
        start = torch.arange(B, dtype=torch.long) * 3
        f = pts.clone().transpose(1, 2).contiguous()
        m.eval()
        with torch.no_grad(), fixed_randint(start):
            p_list, f_list = m(pts, f)
        emb[f'{tag}_eval_tokens'] = f_list[-1].numpy()
        emb[f'{tag}_centroids'] = p_list[-1].numpy()
        m.train()
        with fixed_randint(start):
            p_list, f_list = m(pts, f)
        tok = f_list[-1]
        gw = torch.Generator().manual_seed(320 + N)
        # positive upstream weights: a well-conditioned gradient (no random-sign cancellation), so the
        # ReLU / max-pool gating flips that bf16 activation storage causes stay a ~1% effect
        w = torch.rand(tok.shape, generator=gw) + 0.5
        (tok * w).sum().backward()
        emb[f'{tag}_train_tokens'] = tok.detach().numpy()
        emb[f'{tag}_train_gradw'] = w.numpy()
        emb[f'{tag}_g_conv1a'] = m.convs[0][0][0].weight.grad.numpy().reshape(256, 6)
        emb[f'{tag}_g_conv1b_bias'] = m.convs[0][0][1].bias.grad.numpy()
        emb[f'{tag}_g_bn1_w'] = m.convs[0][0][2].weight.grad.numpy()
        emb[f'{tag}_g_conv2b_sum'] = m.convs[0][1][3].weight.grad.numpy().reshape(256, 512).sum(1)
        emb[f'{tag}_g_conv2a_sum'] = m.convs[0][1][0].weight.grad.numpy().reshape(512, 512).sum(1)
        emb[f'{tag}_bn_running'] = np.stack([m.convs[0][0][2].running_mean.numpy(),
                                             m.convs[0][0][2].running_var.numpy()])
        emb[f'{tag}_meta'] = np.array([B, N, k, 300 + N, 310 + N, 320 + N], dtype=np.int64)
        emb[f'{tag}_weight_checksum'] = np.array(
            [float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point()))])
    np.savez_compressed(os.path.join(HERE, 'p3embed.npz'), **emb)

    # ---------------- ClsHead eval logits ----------------------------------------------------------
    head = ref.ClsHead(in_channels=2 * 192, num_classes=40)
    sd = O.seeded_state_dict(head, seed=400)
    head.load_state_dict(sd)
    head.eval()
    x = torch.randn(8, 2 * 192, generator=torch.Generator().manual_seed(401))
    with torch.no_grad():
        y = head(x)
    np.savez_compressed(os.path.join(HERE, 'clshead.npz'), x=x.numpy(), logits=y.numpy())

    # ---------------- Full model (RESTATEMENT, unpinned for the ViT block): tiny config ----------
    torch.manual_seed(0)
    model = O.Pix4Point(num_classes=40, embed_dim=192, depth=2, heads=3)
    sd = O.seeded_state_dict(model, seed=500)
    model.load_state_dict(sd)
    model.eval()
    pts = O.synthetic_clouds(4, 256, seed=501)
    start = torch.tensor([0, 5, 17, 100])
    with torch.no_grad():
        logits = model(pts, start)
    np.savez_compressed(os.path.join(HERE, 'full_tiny_restatement.npz'),
                        points=pts.numpy(), start=start.numpy(), logits=logits.numpy(),
                        meta=np.array([40, 192, 2, 3, 500, 501], dtype=np.int64))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
