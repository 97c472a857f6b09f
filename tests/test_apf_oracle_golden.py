"""CPU: the APF oracle restatement against the fixtures the REFERENCE produced (tests/golden/make_golden_apf.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import apf_oracle as A


@pytest.fixture(scope='module')
def index(golden_dir):
    return np.load(os.path.join(golden_dir, 'apf_index.npz'))


@pytest.mark.parametrize('tag,S', [('n2048g196', 196), ('n1024g196', 196), ('n333g64', 64)])
def test_fps_matches_reference(index, tag, S):
    pts = torch.from_numpy(index[f'fps_{tag}_points'])
    got = A.furthest_point_sample(pts, S, torch.from_numpy(index[f'fps_{tag}_start']))
    assert got.dtype == torch.int64
    assert np.array_equal(got.numpy(), index[f'fps_{tag}_idx'].astype(np.int64))


@pytest.mark.parametrize('tag', ['n2048k32', 'n512k32', 'n256k8'])
def test_knn_sets_vs_reference(index, tag):
    pts = torch.from_numpy(index[f'knn_{tag}_points'])
    cidx = torch.from_numpy(index[f'knn_{tag}_cidx'].astype(np.int64))
    ref = index[f'knn_{tag}_ref_sorted'].astype(np.int64)
    k = ref.shape[-1]
    cen = A.index_points(pts, cidx)
    got = np.sort(A.knn_point(k, pts, cen).numpy(), axis=-1)
    clear = index[f'knn_{tag}_min_gap'] > 2e-6          # rows where the -2ab+a^2+b^2 rounding cannot reorder
    assert clear.mean() > 0.9
    assert np.array_equal(got[clear], ref[clear])
    assert (got == ref).mean() > 0.995


@pytest.mark.parametrize('tag', ['g196', 'g64'])
def test_morton_codes_and_order(index, tag):
    cen = torch.from_numpy(index[f'morton_{tag}_points'])
    codes = A.morton_codes(cen).numpy()
    assert np.array_equal(codes, index[f'morton_{tag}_codes'].astype(np.int64))
    order = A.points_to_morton(cen).numpy()
    ref_order = index[f'morton_{tag}_order'].astype(np.int64)
    # same code sequence always; same permutation wherever the codes are distinct (argsort ties are unspecified)
    assert np.array_equal(np.take_along_axis(codes, order, 1), np.take_along_axis(codes, ref_order, 1))
    for b in range(codes.shape[0]):
        if len(np.unique(codes[b])) == codes.shape[1]:
            assert np.array_equal(order[b], ref_order[b])


def _lexsorted(nb):
    key = np.lexsort([nb[..., c] for c in range(5, -1, -1)], axis=-1)
    return np.take_along_axis(nb, key[..., None], axis=2)


@pytest.mark.parametrize('tag', ['n512g64', 'n2048g196'])
def test_group_matches_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_group.npz'))
    B, N, G, gs = z[f'{tag}_meta']
    pts = torch.from_numpy(z[f'{tag}_points'])
    nb, cen = A.Group(int(G), int(gs))(pts, pts[:, :, :3], torch.from_numpy(z[f'{tag}_start']))
    assert np.array_equal(cen.numpy(), z[f'{tag}_center'])
    got, want = _lexsorted(nb.numpy()), z[f'{tag}_neigh_sorted']
    # the gap is recorded in FPS order; move it to Morton order like the groups
    order = A.points_to_morton(A.index_points(pts, A.furthest_point_sample(pts, int(G), torch.from_numpy(z[f'{tag}_start'])))).numpy()
    clear = np.take_along_axis(z[f'{tag}_min_gap'], order, 1) > 2e-6
    assert clear.mean() > 0.9
    assert np.array_equal(got[clear], want[clear])


@pytest.mark.parametrize('tag', ['ec128', 'ec768'])
def test_encoder_matches_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_encoder.npz'))
    B, G, gs, ec = (int(v) for v in z[f'{tag}_meta'])
    m = A.Encoder(ec, 6)
    m.load_state_dict(A.seeded_state_dict(m, seed=900 + ec))
    x = torch.from_numpy(z[f'{tag}_x'])
    m.eval()
    with torch.no_grad():
        assert np.allclose(m(x).numpy(), z[f'{tag}_eval_tokens'], rtol=1e-5, atol=1e-5)
    m.train()
    tok = m(x)
    assert np.allclose(tok.detach().numpy(), z[f'{tag}_train_tokens'], rtol=1e-5, atol=1e-5)
    (tok * torch.from_numpy(z[f'{tag}_train_gradw'])).sum().backward()
    close = lambda a, b: np.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(b).max())))
    assert close(m.first_conv[0].weight.grad.numpy().reshape(256, 6), z[f'{tag}_g_conv0_w'])
    assert close(m.first_conv[1].weight.grad.numpy(), z[f'{tag}_g_bn1_w'])
    assert close(m.first_conv[6].bias.grad.numpy(), z[f'{tag}_g_conv6_b'])
    assert close(m.second_conv[0].weight.grad.numpy().reshape(2 * ec, 2 * ec).sum(1), z[f'{tag}_g_sconv0_wsum'])
    assert close(m.second_conv[3].bias.grad.numpy(), z[f'{tag}_g_sconv3_b'])
    assert np.allclose(np.stack([m.first_conv[1].running_mean.numpy(), m.first_conv[1].running_var.numpy()]),
                       z[f'{tag}_bn1_running'], rtol=1e-5, atol=1e-6)


def test_layers_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'apf_layers.npz'))
    head = A.ClassificationHead(192, 15)
    head.load_state_dict(A.seeded_state_dict(head, seed=1000))
    head.eval()
    att = A.AttentionLayer(192, 3)
    att.load_state_dict(A.seeded_state_dict(att, seed=1010))
    ad = A.AdapterLayer(192, 64, 0.1)
    ad.load_state_dict(A.seeded_state_dict(ad, seed=1020))
    ad.eval()
    with torch.no_grad():
        assert np.allclose(head(torch.from_numpy(z['head_x'])).numpy(), z['head_logits'], rtol=1e-5, atol=1e-5)
        assert np.allclose(att(torch.from_numpy(z['attn_x'])).numpy(), z['attn_y'], rtol=1e-5, atol=1e-5)
        assert np.allclose(ad(torch.from_numpy(z['adapter_x'])).numpy(), z['adapter_y'], rtol=1e-5, atol=1e-5)


def test_full_model_restatement_fixture_and_freeze(golden_dir):
    z = np.load(os.path.join(golden_dir, 'apf_full_tiny_restatement.npz'))
    m = A.AdaptPointFormer(num_classes=15, embedding_dim=192, npoint=64, nsample=32, depth=2, num_heads=3)
    m.load_state_dict(A.seeded_state_dict(m, seed=1100))
    m.eval()
    with torch.no_grad():
        got = m(torch.from_numpy(z['points']), torch.from_numpy(z['start'])).numpy()
    assert np.allclose(got, z['logits'], rtol=1e-5, atol=1e-5)
    # apf.py:335-346: trainable = head, encoder_norm, point_encoder; blocks (adapters included) frozen
    train = {n.split('.')[0] for n, p in m.named_parameters() if p.requires_grad}
    assert train == {'head', 'encoder_norm', 'point_encoder'}
    assert not any(p.requires_grad for p in m.blocks.parameters())


# ---------------------------------------------------------------------------------------------
# the reference's shipped configuration: in_channels = 4 (xyz + height), 8-channel groups   (tests/golden/make_golden_apf_height.py)
# ---------------------------------------------------------------------------------------------
def _lexsorted_c(nb):
    key = np.lexsort([nb[..., c] for c in range(nb.shape[-1] - 1, -1, -1)], axis=-1)
    return np.take_along_axis(nb, key[..., None], axis=2)


@pytest.mark.parametrize('tag', ['n512g64', 'n2048g196'])
def test_group_with_height_channel_matches_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, N, G, gs = (int(v) for v in z[f'group_{tag}_meta'])
    x = torch.from_numpy(z[f'group_{tag}_x'])
    start = torch.from_numpy(z[f'group_{tag}_start'])
    nb, cen = A.Group(G, gs)(x, x[:, :, :3], start)
    assert nb.shape == (B, G, gs, 8)
    assert np.array_equal(cen.numpy(), z[f'group_{tag}_center'])
    xyz = x[:, :, :3].contiguous()
    order = A.points_to_morton(A.index_points(xyz, A.furthest_point_sample(xyz, G, start))).numpy()
    clear = np.take_along_axis(z[f'group_{tag}_min_gap'], order, 1) > 2e-6
    assert clear.mean() > 0.9
    assert np.array_equal(_lexsorted_c(nb.numpy())[clear], z[f'group_{tag}_neigh_sorted'][clear])


@pytest.mark.parametrize('tag', ['ec128', 'ec768'])
def test_encoder_8_channels_matches_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, G, gs, ec = (int(v) for v in z[f'enc_{tag}_meta'])
    m = A.Encoder(ec, 8)
    m.load_state_dict(A.seeded_state_dict(m, seed=1400 + ec))
    x = torch.from_numpy(z[f'enc_{tag}_x'])
    m.eval()
    with torch.no_grad():
        assert np.allclose(m(x).numpy(), z[f'enc_{tag}_eval_tokens'], rtol=1e-5, atol=1e-5)
    m.train()
    tok = m(x)
    assert np.allclose(tok.detach().numpy(), z[f'enc_{tag}_train_tokens'], rtol=1e-5, atol=1e-5)
    (tok * torch.from_numpy(z[f'enc_{tag}_train_gradw'])).sum().backward()
    close = lambda a, b: np.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(b).max())))
    assert close(m.first_conv[0].weight.grad.numpy().reshape(256, 8), z[f'enc_{tag}_g_conv0_w'])
    assert close(m.first_conv[1].weight.grad.numpy(), z[f'enc_{tag}_g_bn1_w'])
    assert close(m.second_conv[3].bias.grad.numpy(), z[f'enc_{tag}_g_sconv3_b'])


def test_pointnet_with_height_channel_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, N, G, gs, ec = (int(v) for v in z['pointnet_meta'])
    pn = A.PointNet(ec, G, gs, 8)
    pn.load_state_dict(A.seeded_state_dict(pn, seed=1500))
    pn.eval()
    with torch.no_grad():
        tok = pn(torch.from_numpy(z['pointnet_x']), torch.from_numpy(z['pointnet_start'])).numpy()
    # a group whose neighbour set differs on a near-tie (gap < 2e-6) would move one token; none does on this fixture
    assert np.allclose(tok, z['pointnet_tokens'], rtol=1e-4, atol=1e-4)
