"""Generates the APF golden fixtures (``apf_*.npz``) in this directory from the REFERENCE itself.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden_apf.py

Three reference files are loaded by file path: ``src/data/sampler.py`` (torch only),
``src/models/apf_utils.py`` and ``src/models/apf.py``.  Their package-level imports (``models``, ``data``,
``timm.models.layers``) are satisfied with inert placeholder modules - timm 1.0.16, torchvision and h5py are
not installed - so only the torch-only classes are usable: ``furthest_point_sample``,
``knn_point``, ``index_points``, ``MortonEncoder``, ``Group``, ``Encoder``, ``PointNet``,
``ClassificationHead``, ``AttentionLayer``, ``AdapterLayer``.  ``APFViTLayer`` / ``AdaptPointFormer`` need
timm's ``Mlp`` / ``DropPath`` / weights and are NOT constructed; the whole-model fixture at the end comes
from the oracle's restatement and says so in its name.

Fixtures are DATA (inputs + expected outputs); no reference source travels.
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
from oracle import apf_oracle as A   # noqa: E402
from make_golden import fixed_randint   # noqa: E402

REF = '/root/reference/src'


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for n in ('timm', 'timm.models', 'timm.models.layers'):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules['timm.models.layers'].DropPath = None      # names only; never called
    sys.modules['timm.models.layers'].Mlp = None
    models = types.ModuleType('models')
    models.__path__ = []
    models.get_vit = models.get_timm_vit = None
    sys.modules['models'] = models
    data = types.ModuleType('data')
    data.__path__ = []
    sys.modules['data'] = data
    sampler = _load('data.sampler', f'{REF}/data/sampler.py')
    utils = _load('models.apf_utils', f'{REF}/models/apf_utils.py')
    apf = _load('ref_apf', f'{REF}/models/apf.py')
    return sampler, utils, apf


def main():
    sampler, utils, apf = load_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}

    # ---------------- FPS (sampler.py:4-30) at the APF shapes ----------------------------------------
    for tag, (B, N, S) in {'n2048g196': (2, 2048, 196), 'n1024g196': (2, 1024, 196), 'n333g64': (3, 333, 64)}.items():
        pts = A.synthetic_clouds(B, N, seed=500 + N)
        start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(17 + N))
        with fixed_randint(start):
            idx = sampler.furthest_point_sample(pts, S)
        out[f'fps_{tag}_points'] = pts.numpy()
        out[f'fps_{tag}_start'] = start.numpy()
        out[f'fps_{tag}_idx'] = idx.numpy().astype(np.int32)

    # ---------------- knn_point (sampler.py:47-75): neighbour sets + the near-tie flag ------------------
    for tag, (B, N, S, k) in {'n2048k32': (1, 2048, 196, 32), 'n512k32': (2, 512, 64, 32), 'n256k8': (2, 256, 32, 8)}.items():
        pts = A.synthetic_clouds(B, N, seed=600 + N)
        start = torch.zeros(B, dtype=torch.long)
        with fixed_randint(start):
            cidx = sampler.furthest_point_sample(pts, S)
        cen = sampler.index_points(pts, cidx)
        ref_idx = sampler.knn_point(k, pts, cen)
        out[f'knn_{tag}_points'] = pts.numpy()
        out[f'knn_{tag}_cidx'] = cidx.numpy().astype(np.int32)
        out[f'knn_{tag}_ref_sorted'] = np.sort(ref_idx.numpy(), axis=-1).astype(np.int16)
        out[f'knn_{tag}_min_gap'] = A.knn_min_gap(cen, pts, k).numpy()

    # ---------------- Morton order (apf_utils.py:67-104) -------------------------------------------------
    for tag, (B, G) in {'g196': (3, 196), 'g64': (4, 64)}.items():
        cen = A.synthetic_clouds(B, G, seed=700 + G)
        order = utils.MortonEncoder.points_to_morton(cen)
        x = ((cen - cen.min(1, keepdim=True)[0]) / (cen.max(1, keepdim=True)[0] - cen.min(1, keepdim=True)[0] + 1e-8)
             * 1023).long()
        codes = utils.MortonEncoder.encode_morton3_vectorized(x[..., 0], x[..., 1], x[..., 2])
        out[f'morton_{tag}_points'] = cen.numpy()
        out[f'morton_{tag}_codes'] = codes.numpy().astype(np.int32)
        out[f'morton_{tag}_order'] = order.numpy().astype(np.int16)
    np.savez_compressed(os.path.join(HERE, 'apf_index.npz'), **out)

    # ---------------- Group (apf.py:12-112) --------------------------------------------------------------
    grp = {}
    for tag, (B, N, G, gs) in {'n512g64': (2, 512, 64, 32), 'n2048g196': (1, 2048, 196, 32)}.items():
        pts = A.synthetic_clouds(B, N, seed=800 + N)
        start = torch.arange(B, dtype=torch.long) * 5 + 1
        g = apf.Group(G, gs)
        with fixed_randint(start):
            nb, cen = g(pts, pts[:, :, :3])
        # neighbours inside a group are unordered in the reference (topk sorted=False): store them sorted
        # lexicographically by their 6 channels so the comparison is a set comparison
        nbs = nb.numpy()
        key = np.lexsort([nbs[..., c] for c in range(5, -1, -1)], axis=-1)
        nbs = np.take_along_axis(nbs, key[..., None], axis=2)
        grp[f'{tag}_points'] = pts.numpy()
        grp[f'{tag}_start'] = start.numpy()
        grp[f'{tag}_center'] = cen.numpy()
        grp[f'{tag}_neigh_sorted'] = nbs.astype(np.float32)
        cen0 = sampler.index_points(pts, A.furthest_point_sample(pts, G, start))
        grp[f'{tag}_min_gap'] = A.knn_min_gap(cen0, pts, gs).numpy()
        grp[f'{tag}_meta'] = np.array([B, N, G, gs], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'apf_group.npz'), **grp)

    # ---------------- Encoder (apf.py:114-181): eval tokens, train tokens + gradients ------------------------
    enc = {}
    for tag, (B, G, gs, ec) in {'ec128': (2, 24, 32, 128), 'ec768': (1, 12, 32, 768)}.items():
        m = apf.Encoder(ec, 6)
        m.load_state_dict(A.seeded_state_dict(m, seed=900 + ec))
        x = torch.randn(B, G, gs, 6, generator=torch.Generator().manual_seed(910 + ec)) * 0.3
        m.eval()
        with torch.no_grad():
            enc[f'{tag}_eval_tokens'] = m(x).numpy()
        m.train()
        tok = m(x)
        w = torch.rand(tok.shape, generator=torch.Generator().manual_seed(920 + ec)) + 0.5
        (tok * w).sum().backward()
        enc[f'{tag}_x'] = x.numpy()
        enc[f'{tag}_train_tokens'] = tok.detach().numpy()
        enc[f'{tag}_train_gradw'] = w.numpy()
        enc[f'{tag}_g_conv0_w'] = m.first_conv[0].weight.grad.numpy().reshape(256, 6)
        enc[f'{tag}_g_bn1_w'] = m.first_conv[1].weight.grad.numpy()
        enc[f'{tag}_g_bn1_b'] = m.first_conv[1].bias.grad.numpy()
        enc[f'{tag}_g_conv3_wsum'] = m.first_conv[3].weight.grad.numpy().reshape(512, 256).sum(1)
        enc[f'{tag}_g_conv6_b'] = m.first_conv[6].bias.grad.numpy()
        enc[f'{tag}_g_conv6_wsum'] = m.first_conv[6].weight.grad.numpy().reshape(ec, 512).sum(1)
        enc[f'{tag}_g_sconv0_wsum'] = m.second_conv[0].weight.grad.numpy().reshape(2 * ec, 2 * ec).sum(1)
        enc[f'{tag}_g_sbn_w'] = m.second_conv[1].weight.grad.numpy()
        enc[f'{tag}_g_sconv3_wsum'] = m.second_conv[3].weight.grad.numpy().reshape(ec, 2 * ec).sum(1)
        enc[f'{tag}_g_sconv3_b'] = m.second_conv[3].bias.grad.numpy()
        enc[f'{tag}_bn1_running'] = np.stack([m.first_conv[1].running_mean.numpy(), m.first_conv[1].running_var.numpy()])
        enc[f'{tag}_sbn_running'] = np.stack([m.second_conv[1].running_mean.numpy(), m.second_conv[1].running_var.numpy()])
        enc[f'{tag}_meta'] = np.array([B, G, gs, ec], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'apf_encoder.npz'), **enc)

    # ---------------- layers: ClassificationHead, AttentionLayer, AdapterLayer ------------------------------
    lay = {}
    head = apf.ClassificationHead(192, 15)
    head.load_state_dict(A.seeded_state_dict(head, seed=1000))
    head.eval()
    x = torch.randn(6, 192, generator=torch.Generator().manual_seed(1001))
    with torch.no_grad():
        lay['head_x'], lay['head_logits'] = x.numpy(), head(x).numpy()
    att = utils.AttentionLayer(192, 3)
    att.load_state_dict(A.seeded_state_dict(att, seed=1010))
    x = torch.randn(2, 50, 192, generator=torch.Generator().manual_seed(1011))
    with torch.no_grad():
        lay['attn_x'], lay['attn_y'] = x.numpy(), att(x).numpy()
    ad = utils.AdapterLayer(192, 64, 0.1)
    ad.load_state_dict(A.seeded_state_dict(ad, seed=1020))
    ad.eval()
    with torch.no_grad():
        lay['adapter_x'], lay['adapter_y'] = x.numpy(), ad(x).numpy()
    np.savez_compressed(os.path.join(HERE, 'apf_layers.npz'), **lay)

    # ---------------- whole model, from the RESTATEMENT (timm Mlp/DropPath absent) ------------------------------
    full = {}
    m = A.AdaptPointFormer(num_classes=15, embedding_dim=192, npoint=64, nsample=32, depth=2, num_heads=3)
    m.load_state_dict(A.seeded_state_dict(m, seed=1100))
    m.eval()
    pts = A.synthetic_clouds(4, 512, seed=1101)
    start = torch.tensor([0, 7, 100, 511])
    with torch.no_grad():
        full['points'], full['start'], full['logits'] = pts.numpy(), start.numpy(), m(pts, start).numpy()
    np.savez_compressed(os.path.join(HERE, 'apf_full_tiny_restatement.npz'), **full)
    print('written:', sorted(f for f in os.listdir(HERE) if f.startswith('apf_')))


if __name__ == '__main__':
    main()
