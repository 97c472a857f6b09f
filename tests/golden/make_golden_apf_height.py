"""Generates ``apf_height.npz`` from the REFERENCE: the shipped APF configuration with the height channel
(/root/reference/train_apf.py:5,29: in_channels=4, use_height=True), i.e. Group on (B,N,4) clouds [xyz, height] giving
8-channel neighbourhoods (src/models/apf.py:83-95,293), Encoder(E, 8) and the whole PointNet tokenizer.

    python tests/golden/make_golden_apf_height.py        (build container only: needs /root/reference)

Same loading recipe as make_golden_apf.py (file-path import, inert placeholders for the absent timm / package imports);
the heights are the reference's own: center_and_normalize_point_cloud (src/data/augment.py:228-259, numpy only, loaded
by file path as well).  Fixtures are DATA (inputs + expected outputs); no reference source travels.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import apf_oracle as A            # noqa: E402
from make_golden import fixed_randint         # noqa: E402
from make_golden_apf import load_reference, _load, REF   # noqa: E402


def clouds_with_height(aug, B, N, seed):
    """(B,N,4) = the reference's center_and_normalize_point_cloud on a synthetic cloud: normalised xyz + heights."""
    raw = A.synthetic_clouds(B, N, seed=seed).numpy() * 1.7 + 0.3
    out = []
    for b in range(B):
        d = aug.center_and_normalize_point_cloud({'xyz': raw[b].astype(np.float32)})
        out.append(np.concatenate([d['xyz'], d['heights']], axis=1).astype(np.float32))
    return torch.from_numpy(np.stack(out))


def main():
    sampler, utils, apf = load_reference()
    aug = _load('ref_augment', f'{REF}/data/augment.py')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    # ---------------- Group on 4-channel clouds (apf.py:52-112) ---------------------------------------------
    for tag, (B, N, G, gs) in {'n512g64': (2, 512, 64, 32), 'n2048g196': (1, 2048, 196, 32)}.items():
        x = clouds_with_height(aug, B, N, seed=1300 + N)
        start = torch.arange(B, dtype=torch.long) * 7 + 2
        with fixed_randint(start):
            nb, cen = apf.Group(G, gs)(x, x[:, :, :3])
        nbs = nb.numpy()
        key = np.lexsort([nbs[..., c] for c in range(7, -1, -1)], axis=-1)      # neighbours are unordered in the reference
        out[f'group_{tag}_x'] = x.numpy()
        out[f'group_{tag}_start'] = start.numpy()
        out[f'group_{tag}_center'] = cen.numpy()
        out[f'group_{tag}_neigh_sorted'] = np.take_along_axis(nbs, key[..., None], axis=2).astype(np.float32)
        cen0 = sampler.index_points(x[:, :, :3].contiguous(), A.furthest_point_sample(x[:, :, :3].contiguous(), G, start))
        out[f'group_{tag}_min_gap'] = A.knn_min_gap(cen0, x[:, :, :3].contiguous(), gs).numpy()
        out[f'group_{tag}_meta'] = np.array([B, N, G, gs], dtype=np.int64)
    # ---------------- Encoder(E, 8) (apf.py:114-181): eval tokens, train tokens + gradients --------------------
    for tag, (B, G, gs, ec) in {'ec128': (2, 24, 32, 128), 'ec768': (1, 12, 32, 768)}.items():
        m = apf.Encoder(ec, 8)
        m.load_state_dict(A.seeded_state_dict(m, seed=1400 + ec))
        x = torch.randn(B, G, gs, 8, generator=torch.Generator().manual_seed(1410 + ec)) * 0.3
        m.eval()
        with torch.no_grad():
            out[f'enc_{tag}_eval_tokens'] = m(x).numpy()
        m.train()
        tok = m(x)
        w = torch.rand(tok.shape, generator=torch.Generator().manual_seed(1420 + ec)) + 0.5
        (tok * w).sum().backward()
        out[f'enc_{tag}_x'] = x.numpy()
        out[f'enc_{tag}_train_tokens'] = tok.detach().numpy()
        out[f'enc_{tag}_train_gradw'] = w.numpy()
        out[f'enc_{tag}_g_conv0_w'] = m.first_conv[0].weight.grad.numpy().reshape(256, 8)
        out[f'enc_{tag}_g_bn1_w'] = m.first_conv[1].weight.grad.numpy()
        out[f'enc_{tag}_g_conv3_wsum'] = m.first_conv[3].weight.grad.numpy().reshape(512, 256).sum(1)
        out[f'enc_{tag}_g_sconv3_b'] = m.second_conv[3].bias.grad.numpy()
        out[f'enc_{tag}_bn1_running'] = np.stack([m.first_conv[1].running_mean.numpy(), m.first_conv[1].running_var.numpy()])
        out[f'enc_{tag}_meta'] = np.array([B, G, gs, ec], dtype=np.int64)
    # ---------------- PointNet(E, G, gs, 8) (apf.py:183-217): the whole tokenizer, eval mode ---------------------
    B, N, G, gs, ec = 2, 512, 64, 32, 192
    pn = apf.PointNet(ec, G, gs, 8)
    pn.load_state_dict(A.seeded_state_dict(pn, seed=1500))
    pn.eval()
    x = clouds_with_height(aug, B, N, seed=1501)
    start = torch.tensor([3, 400])
    with fixed_randint(start), torch.no_grad():
        out['pointnet_tokens'] = pn(x).numpy()
    out['pointnet_x'], out['pointnet_start'] = x.numpy(), start.numpy()
    out['pointnet_meta'] = np.array([B, N, G, gs, ec], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'apf_height.npz'), **out)
    print('written apf_height.npz:', len(out), 'arrays')


if __name__ == '__main__':
    main()
