"""GPU parity of the AdaptPointFormer path through the reference-shaped Python API, against
the reference-generated fixtures (tests/golden/apf_*.npz) and the CPU oracle (oracle/apf_oracle.py).

Index outputs (FPS picks, Morton codes and order, group membership) are bit-exact.  Floating point is bf16 MFMA operands
with fp32 accumulation against an fp32 oracle; bounds follow tests/test_gpu_02_model.py: activations / logits 1e-2 relative
L2, gradients of smooth pieces 3e-2, gradients through hard gates (ReLU, max-pool) 5e-2 with a well-conditioned upstream
gradient.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import p4p_amd                                   # noqa: E402
from p4p_amd import apf as PA                    # noqa: E402
from p4p_amd import ops                          # noqa: E402
from p4p_amd import functional_apf as FA         # noqa: E402
from oracle import apf_oracle as A               # noqa: E402
from tests.emulate import apf_encoder_emulated   # noqa: E402

DEV = 'cuda'


def rel_l2(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope='module')
def index(golden_dir):
    return np.load(os.path.join(golden_dir, 'apf_index.npz'))


# --------------------------------------------------------------------------------------------- index ops
@pytest.mark.parametrize('tag,S', [('n2048g196', 196), ('n1024g196', 196), ('n333g64', 64)])
def test_fps_bit_exact_vs_reference(index, tag, S):
    pts = torch.from_numpy(index[f'fps_{tag}_points']).to(DEV)
    got = PA.furthest_point_sample(pts, S, torch.from_numpy(index[f'fps_{tag}_start']).to(DEV))
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), index[f'fps_{tag}_idx'].astype(np.int64))


@pytest.mark.parametrize('tag', ['n2048k32', 'n512k32', 'n256k8'])
def test_knn_sets_vs_reference(index, tag):
    pts = torch.from_numpy(index[f'knn_{tag}_points']).to(DEV)
    cidx = torch.from_numpy(index[f'knn_{tag}_cidx'].astype(np.int64)).to(DEV)
    ref = index[f'knn_{tag}_ref_sorted'].astype(np.int64)
    cen = PA.index_points(pts, cidx)
    got = PA.knn_point(ref.shape[-1], pts, cen)
    assert got.dtype == torch.int64
    # bit-exact against the oracle's exact-distance policy, order included
    want = A.knn_point(ref.shape[-1], pts.cpu(), cen.cpu())
    assert torch.equal(got.cpu(), want)
    clear = index[f'knn_{tag}_min_gap'] > 2e-6
    assert np.array_equal(np.sort(got.cpu().numpy(), -1)[clear], ref[clear])


@pytest.mark.parametrize('tag', ['g196', 'g64'])
def test_morton_bit_exact_vs_reference(index, tag):
    cen = torch.from_numpy(index[f'morton_{tag}_points']).to(DEV)
    order, codes = ops.morton_order(cen, want_codes=True)
    assert np.array_equal(codes.cpu().numpy(), index[f'morton_{tag}_codes'])
    want = A.points_to_morton(cen.cpu())
    assert torch.equal(order.cpu().long(), want)
    ref_order = index[f'morton_{tag}_order'].astype(np.int64)
    c = codes.cpu().numpy()
    for b in range(c.shape[0]):
        if len(np.unique(c[b])) == c.shape[1]:
            assert np.array_equal(order[b].cpu().numpy(), ref_order[b])
    assert torch.equal(PA.MortonEncoder.points_to_morton(cen).cpu(), want)


def test_morton_ties_and_degenerate_clouds():
    cen = torch.rand(3, 50, 3)
    cen[0, 10:20] = cen[0, 0]                   # equal codes: lowest index first
    cen[1] = 0.25                               # all points equal: every code 0, identity order
    cen[2, :, 2] = 0.5                          # flat axis
    got = ops.morton_order(cen.to(DEV)).cpu().long()
    assert torch.equal(got, A.points_to_morton(cen))
    assert torch.equal(got[1], torch.arange(50))


def _lexsorted(nb):
    key = np.lexsort([nb[..., c] for c in range(5, -1, -1)], axis=-1)
    return np.take_along_axis(nb, key[..., None], axis=2)


@pytest.mark.parametrize('tag', ['n512g64', 'n2048g196'])
def test_group_vs_reference_fixture(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_group.npz'))
    B, N, G, gs = (int(v) for v in z[f'{tag}_meta'])
    pts = torch.from_numpy(z[f'{tag}_points']).to(DEV)
    start = torch.from_numpy(z[f'{tag}_start']).to(DEV)
    grp = PA.Group(G, gs)
    nb, cen = grp(pts, pts[:, :, :3], start)
    assert nb.shape == (B, G, gs, 6) and cen.shape == (B, G, 3)
    assert np.array_equal(cen.cpu().numpy(), z[f'{tag}_center']), 'centres (FPS + Morton order) must be bit-exact'
    # bit-exact against the oracle (same neighbour policy), set-exact against the reference on clear rows
    nb_o, cen_o = A.Group(G, gs)(pts.cpu(), pts.cpu(), start.cpu())
    assert torch.equal(nb.cpu(), nb_o) and torch.equal(cen.cpu(), cen_o)
    fps_idx, _, order = grp.indices(pts, start)
    clear = np.take_along_axis(z[f'{tag}_min_gap'], order.cpu().numpy().astype(np.int64), 1) > 2e-6
    assert np.array_equal(_lexsorted(nb.cpu().numpy())[clear], z[f'{tag}_neigh_sorted'][clear])
    # the bf16 hi/lo operand of the fused path carries the same rows
    _, x0, _ = ops.apf_group(pts, fps_idx, ops.knn(pts, PA.index_points(pts, fps_idx), gs), order)
    rec = (x0[:, 0:6].float() + x0[:, 6:12].float()).view(B, G, gs, 6)
    assert float((rec - nb).abs().max()) < 1e-4 and torch.equal(x0[:, 0:6], x0[:, 12:18]) and not x0[:, 18:].any()


def test_group_max_forward_backward():
    G, k, C = 37, 32, 192
    torch.manual_seed(0)
    H = torch.randn(G * k, C, device=DEV).clamp(-3, 3).bfloat16()
    H[5 * k + 3] = H[5 * k + 9] = 4.0            # a tie: the first maximum wins
    bias = torch.randn(C, device=DEV)
    tok, arg = ops.group_max(H, G, k, bias=bias)
    want, warg = H.float().view(G, k, C).max(1)
    assert torch.equal(tok, want + bias)
    assert torch.equal(arg[5].long(), torch.full((C,), 3, device=DEV))
    assert torch.equal(H.float().view(G, k, C).gather(1, arg.long().unsqueeze(1)).squeeze(1), want)
    d = torch.randn(G, C, device=DEV)
    dH = ops.group_max_bwd(d, arg, G, k)
    ref = torch.zeros(G, k, C, device=DEV).scatter_(1, arg.long().unsqueeze(1), d.unsqueeze(1)).view(G * k, C)
    assert torch.equal(dH, ref.bfloat16())
    base = torch.randn(G * k, C, device=DEV).bfloat16()
    acc = ops.group_max_bwd(d.bfloat16(), arg, G, k, into=base.clone())
    want_acc = (base.float() + torch.zeros(G, k, C, device=DEV).scatter_(
        1, arg.long().unsqueeze(1), d.bfloat16().float().unsqueeze(1)).view(G * k, C)).bfloat16()
    assert torch.equal(acc, want_acc)


# --------------------------------------------------------------------------------------------- encoder MLP
@pytest.mark.parametrize('tag', ['ec128', 'ec768'])
def test_encoder_against_reference_fixture(golden_dir, tag):
    """Fixture values come from the REFERENCE's Encoder (tests/golden/make_golden_apf.py)."""
    z = np.load(os.path.join(golden_dir, 'apf_encoder.npz'))
    B, G, gs, ec = (int(v) for v in z[f'{tag}_meta'])
    m = PA.Encoder(ec, 6)
    m.load_state_dict(A.seeded_state_dict(A.Encoder(ec, 6), seed=900 + ec))
    m = m.to(DEV)
    x = torch.from_numpy(z[f'{tag}_x']).to(DEV)
    m.eval()
    with torch.no_grad():
        tok = m(x)
    assert tok.shape == (B, G, ec)
    assert rel_l2(tok, torch.from_numpy(z[f'{tag}_eval_tokens'])) < 1e-2
    m.train()
    tok = m(x)
    assert rel_l2(tok, torch.from_numpy(z[f'{tag}_train_tokens'])) < 1e-2
    (tok * torch.from_numpy(z[f'{tag}_train_gradw']).to(DEV)).sum().backward()
    fc, sc = m.first_conv, m.second_conv
    g = lambda name: torch.from_numpy(z[f'{tag}_{name}'])
    errs = {
        'conv0_w': rel_l2(fc[0].weight.grad.reshape(256, 6), g('g_conv0_w')),
        'bn1_w': rel_l2(fc[1].weight.grad, g('g_bn1_w')),
        'bn1_b': rel_l2(fc[1].bias.grad, g('g_bn1_b')),
        'conv3_wsum': rel_l2(fc[3].weight.grad.reshape(512, 256).sum(1), g('g_conv3_wsum')),
        'conv6_wsum': rel_l2(fc[6].weight.grad.reshape(ec, 512).sum(1), g('g_conv6_wsum')),
        'sconv0_wsum': rel_l2(sc[0].weight.grad.reshape(2 * ec, 2 * ec).sum(1), g('g_sconv0_wsum')),
        'sbn_w': rel_l2(sc[1].weight.grad, g('g_sbn_w')),
        'sconv3_wsum': rel_l2(sc[3].weight.grad.reshape(ec, 2 * ec).sum(1), g('g_sconv3_wsum')),
        'sconv3_b': rel_l2(sc[3].bias.grad, g('g_sconv3_b')),
    }
    # against the fp32 reference the ReLU / max-pool decisions of a few of the 1,536 (384) points flip under bf16 storage;
    # five layers and two pools amplify that towards the first layer (kernel logic is checked tightly in
    # test_encoder_backward_vs_rounding_matched_statement)
    assert max(errs.values()) < 0.25 and sorted(errs.values())[len(errs) // 2] < 0.15, errs
    # first_conv[6].bias shifts f, its pooled copy and hence H4 by one constant per channel, which the BatchNorm behind
    # removes: its gradient is zero in exact arithmetic (the reference's value is fp32 noise)
    assert float(fc[6].bias.grad.abs().max()) == 0.0
    assert float(g('g_conv6_b').abs().max()) < 1e-3 * float(fc[6].weight.grad.abs().max())
    for conv in (fc[0], fc[3], sc[0]):          # a bias in front of train-mode BatchNorm has zero gradient
        assert float(conv.bias.grad.abs().max()) == 0.0
    # first layer: fp32-class (hi/lo split operands); fourth layer: its input went through three bf16 GEMMs
    for bn, key, rtol, atol in ((fc[1], 'bn1_running', 1e-4, 1e-6), (sc[1], 'sbn_running', 2e-3, 1.5e-3)):
        run = np.stack([bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy()])
        np.testing.assert_allclose(run, z[f'{tag}_{key}'], rtol=rtol, atol=atol)
        assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize('G,ec', [(48, 128), (40, 768)])
def test_encoder_backward_vs_rounding_matched_statement(G, ec):
    gs = 32
    m = PA.Encoder(ec, 6)
    m.load_state_dict(A.seeded_state_dict(A.Encoder(ec, 6), seed=ec))
    m = m.to(DEV).train()
    x = (torch.randn(1, G, gs, 6, generator=torch.Generator().manual_seed(5)) * 0.3).to(DEV)
    FA._CAPTURE = {}
    try:
        tok = m(x)
    finally:
        gates, FA._CAPTURE = FA._CAPTURE, None
    w = torch.randn(tok.shape, generator=torch.Generator().manual_seed(6)).to(DEV)      # random sign: hard case
    (tok * w).sum().backward()
    params = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    # the statement takes the kernels' ReLU / arg-max decisions (bf16 storage in front of a 32-way max moves ~0.3 % of the
    # picks on one-ulp differences, which alone is a ~10 % gradient change under a random-sign upstream gradient)
    te = apf_encoder_emulated(x.reshape(G * gs, 6), params, G, gs, gates=gates)
    (te * w.view(G, ec)).sum().backward()
    assert rel_l2(tok.reshape(G, ec), te) < 6e-3
    errs = {}
    for n, p in m.named_parameters():
        if n in ('first_conv.0.bias', 'first_conv.3.bias', 'second_conv.0.bias', 'first_conv.6.bias'):
            assert float(p.grad.abs().max()) == 0.0            # exactly zero in exact arithmetic (see the fixture test)
            continue
        errs[n] = rel_l2(p.grad, params[n].grad)
    assert max(errs.values()) < 2e-2, errs


def test_pointnet_fused_path_equals_group_then_encoder():
    B, N, G, gs, E = 3, 512, 64, 32, 192
    pn = PA.PointNet(E, G, gs, 6)
    pn.load_state_dict(A.seeded_state_dict(A.PointNet(E, G, gs, 6), seed=5))
    pn = pn.to(DEV).eval()
    pts = A.synthetic_clouds(B, N, seed=6).to(DEV)
    start = torch.tensor([0, 5, 9], device=DEV)
    with torch.no_grad():
        fused = pn(pts, start)
        nb, _ = pn.group(pts, pts, start)
        two_step = pn.encoder(nb)
        want = A.PointNet(E, G, gs, 6)
        want.load_state_dict(A.seeded_state_dict(want, seed=5))
        ref = want.eval()(pts.cpu(), start.cpu())
    assert torch.equal(fused, two_step)
    assert rel_l2(fused, ref) < 1e-2


# --------------------------------------------------------------------------------------------- transformer layer
def _layer_pair(D, H, seed, drop_path=0.0, dropout=0.0):
    o = A.APFViTLayer(D, H, drop_path, dropout)
    sd = A.seeded_state_dict(o, seed=seed)
    sd['adapter.scale'] = torch.tensor([0.7])
    o.load_state_dict(sd)
    m = PA.APFViTLayer(D, H, drop_path, dropout)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.requires_grad_(False)                      # as inside AdaptPointFormer (apf.py:335-346): block weights are frozen
    return o, m.to(DEV)


@pytest.mark.parametrize('B,T', [(2, 50), (3, 196), (1, 64)])
def test_apf_layer_forward_backward_vs_oracle(B, T):
    D, H = 192, 3
    o, m = _layer_pair(D, H, seed=T)
    o.eval(), m.eval()
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1))
    w = torch.randn(B, T, D, generator=torch.Generator().manual_seed(2))
    xo = x.clone().requires_grad_(True)
    yo = o(xo)
    (yo * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yg = m(xg)
    (yg * w.to(DEV)).sum().backward()
    assert rel_l2(yg, yo) < 1e-2
    assert rel_l2(xg.grad, xo.grad) < 3e-2
    assert all(p.grad is None for p in m.parameters()), 'block weights take no gradient on this path'


def test_apf_layer_pinned_pieces_vs_reference_fixture(golden_dir):
    """AttentionLayer / AdapterLayer outputs recorded from the REFERENCE classes: assemble one layer from those two
    fixtures' weights and check the two sub-results through the composed kernel path."""
    z = np.load(os.path.join(golden_dir, 'apf_layers.npz'))
    D, H = 192, 3
    x = torch.from_numpy(z['attn_x'])
    o, m = _layer_pair(D, H, seed=77)
    att_sd = A.seeded_state_dict(A.AttentionLayer(D, H), seed=1010)
    ad_sd = A.seeded_state_dict(A.AdapterLayer(D, 64, 0.1), seed=1020)
    sd = o.state_dict()
    sd.update({f'attention.{k}': v for k, v in att_sd.items()})
    sd.update({f'adapter.{k}': v for k, v in ad_sd.items()})
    # neutral norm1 / mlp so that the layer output isolates the pinned pieces:
    #   norm1 = identity-ish is impossible (LN always normalises), so compare against the oracle layer instead and pin
    #   the oracle's sub-modules to the fixtures here
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    o.eval(), m.eval()
    with torch.no_grad():
        assert np.allclose(o.attention(x).numpy(), z['attn_y'], rtol=1e-5, atol=1e-5)
        assert np.allclose(o.adapter(x).numpy(), z['adapter_y'], rtol=1e-5, atol=1e-5)
        assert rel_l2(m(x.to(DEV)), o(x)) < 1e-2


def test_apf_layer_drop_path_and_adapter_dropout(monkeypatch):
    D, H, B, T = 192, 3, 4, 50
    o, m = _layer_pair(D, H, seed=9, drop_path=0.5, dropout=0.0)
    m.train()
    s_a = torch.tensor([0.0, 2.0, 2.0, 0.0])
    s_m = torch.tensor([2.0, 0.0, 2.0, 0.0])
    monkeypatch.setattr(PA, 'drop_factors', lambda blk, b, dev, training: ((s_a > 0).to(dev), (s_m > 0).to(dev), 2.0))
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(3))
    w = torch.randn(B, T, D, generator=torch.Generator().manual_seed(4))
    xo = x.clone().requires_grad_(True)
    o.eval()
    x1 = xo + s_a.view(B, 1, 1) * o.attention(o.norm1(xo))
    yo = s_m.view(B, 1, 1) * o.mlp(o.norm2(x1)) + o.adapter(x1) + x1
    (yo * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    yg = m(xg)
    (yg * w.to(DEV)).sum().backward()
    assert rel_l2(yg, yo) < 1e-2 and rel_l2(xg.grad, xo.grad) < 3e-2
    monkeypatch.undo()
    # stochastic pieces: train mode differs from eval mode, eval is deterministic, factors are 0 or 1/keep
    _, m2 = _layer_pair(D, H, seed=9, drop_path=0.5, dropout=0.5)
    m2.train()
    with torch.no_grad():
        a, b = m2(x.to(DEV)), m2(x.to(DEV))
        m2.eval()
        c, d = m2(x.to(DEV)), m2(x.to(DEV))
    assert torch.isfinite(a).all() and not torch.equal(a, b) and torch.equal(c, d)
    fa, fm, c = PA.drop_factors(m2.train(), 1000, DEV, True)
    assert fa.dtype == torch.bool and c == 2.0 and 0.4 < float(fa.float().mean()) < 0.6 and not torch.equal(fa, fm)


# --------------------------------------------------------------------------------------------- whole model
def _model_pair(seed, **kw):
    cfg = dict(num_classes=15, embedding_dim=192, npoint=64, nsample=32, depth=2, num_heads=3)
    cfg.update(kw)
    o = A.AdaptPointFormer(**cfg)
    sd = A.seeded_state_dict(o, seed=seed)
    o.load_state_dict(sd)
    m = PA.AdaptPointFormer(vit_name=None, pretrained=False, **cfg)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return o, m.to(DEV)


def test_full_model_eval_logits_vs_fixture_and_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, 'apf_full_tiny_restatement.npz'))
    o, m = _model_pair(1100)
    m.eval()
    with torch.no_grad():
        got = m(torch.from_numpy(z['points']).to(DEV), torch.from_numpy(z['start']).to(DEV)).cpu()
    scale = float(np.abs(z['logits']).max())
    assert float((got - torch.from_numpy(z['logits'])).abs().max()) < 1e-2 * max(scale, 1.0)
    assert rel_l2(got, torch.from_numpy(z['logits'])) < 1e-2


def test_full_model_freeze_and_state_dict_names():
    o, m = _model_pair(3)
    assert {n for n, p in m.named_parameters() if p.requires_grad} == {n for n, p in o.named_parameters() if p.requires_grad}
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    train = {n.split('.')[0] for n, p in m.named_parameters() if p.requires_grad}
    assert train == {'head', 'encoder_norm', 'point_encoder'}
    # the timm container's weights land in norm1/norm2/mlp only (apf.py:329-331 with strict=False)
    m2 = PA.AdaptPointFormer(num_classes=15, embedding_dim=192, vit_name='vit_custom_d192_l2_h3', pretrained=False,
                             npoint=64, nsample=32, depth=2, num_heads=3)
    assert not any(p.requires_grad for p in m2.blocks.parameters())
    with pytest.raises(RuntimeError, match='never\\s+fetches|P4P_PRETRAINED_DIR'):
        PA.AdaptPointFormer(num_classes=15, embedding_dim=192, vit_name='vit_custom_d192_l2_h3', pretrained=True, depth=2,
                            num_heads=3)


def test_blocks_and_final_norm_backward_vs_oracle():
    """Two layers + encoder_norm with a smooth loss: dx and the (trainable) encoder_norm gradients, tight."""
    o, m = _model_pair(31)
    o.eval(), m.eval()
    B, T, D = 3, 64, 192
    x = torch.randn(B, T, D, generator=torch.Generator().manual_seed(1))
    w = torch.randn(B, T, D, generator=torch.Generator().manual_seed(2))
    xo = x.clone().requires_grad_(True)
    (o.encoder_norm(o.blocks(xo)) * w).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = FA.apf_blocks(xg, list(m.blocks), m.encoder_norm, 3)
    (y * w.to(DEV)).sum().backward()
    assert rel_l2(xg.grad, xo.grad) < 3e-2
    assert rel_l2(m.encoder_norm.weight.grad, o.encoder_norm.weight.grad) < 3e-2
    assert rel_l2(m.encoder_norm.bias.grad, o.encoder_norm.bias.grad) < 3e-2


def test_full_model_train_step_gradients_vs_oracle():
    """dropout / DropPath rates 0 so that both sides are deterministic; train-mode BatchNorm in the encoder.  The head's
    BatchNorm runs on its running statistics here: over a batch of 24 near-identical synthetic clouds its batch
    statistics amplify the 0.5 % bf16 difference of the pooled features twelve-fold in the fp32 oracle itself (measured:
    the ORACLE's head maps the build's pooled features to logits 6 % away from its own), which would drown the
    comparison.  End to end the encoder gradients still cross every hard gate (3 ReLU+BN, two 32-way pools over bf16
    values, the max over tokens) under a mixed-sign upstream gradient: bounded loosely here, tightly with matched gates
    in test_encoder_backward_vs_rounding_matched_statement."""
    o, m = _model_pair(21, dropout_rate=0.0, dropout_path_rate=0.0)
    for mod in (o, m):
        mod.train()
        mod.head.eval()
    B, N = 24, 256
    pts = A.synthetic_clouds(B, N, seed=22) * (0.3 + 0.7 * torch.rand(B, 1, 3, generator=torch.Generator().manual_seed(23)))
    start = torch.arange(B) * 7
    labels = torch.arange(B) % 15
    lo = torch.nn.functional.cross_entropy(o(pts, start), labels, label_smoothing=0.3)
    lo.backward()
    tokens = {}

    def keep_tokens(mod, inp, out):
        out.retain_grad()
        tokens['t'] = out

    hook = m.point_encoder.register_forward_hook(keep_tokens)
    FA._CAPTURE = {}
    try:
        lg = torch.nn.functional.cross_entropy(m(pts.to(DEV), start.to(DEV)), labels.to(DEV), label_smoothing=0.3)
    finally:
        gates, FA._CAPTURE = FA._CAPTURE, None
        hook.remove()
    lg.backward()
    assert abs(float(lg.detach()) - float(lo.detach())) < 5e-3 * abs(float(lo.detach()))
    go = dict(o.named_parameters())
    # zero in exact arithmetic (a per-channel constant in front of a train-mode BatchNorm)
    zero = ('first_conv.0.bias', 'first_conv.3.bias', 'second_conv.0.bias', 'first_conv.6.bias')
    head, enc = {}, {}
    for n, p in m.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        if n.endswith(zero):
            assert float(p.grad.abs().max()) < 1e-3
            continue
        (enc if n.startswith('point_encoder') else head)[n] = rel_l2(p.grad, go[n].grad)
    assert len(head) == 12 and len(enc) == 12
    assert max(head.values()) < 8e-2, head                      # head + encoder_norm: smooth
    vals = sorted(enc.values())
    assert vals[len(vals) // 2] < 0.3 and vals[-1] < 0.4, enc   # against the fp32 oracle: its own gates (see the docstring)
    # THE tight check: the same encoder gradients against the statement that takes the kernels' ReLU / arg-max decisions
    # (tests/emulate.py) and the model's OWN upstream gradient d(loss)/d(tokens) -- arithmetic only, every layer's scale
    G, gs = m.point_encoder.group.num_group, m.point_encoder.group.group_size
    nb, _ = m.point_encoder.group(pts.to(DEV), pts.to(DEV), start.to(DEV))
    params = {n: p.detach().clone().requires_grad_(True) for n, p in m.point_encoder.encoder.named_parameters()}
    te = apf_encoder_emulated(nb.reshape(B * G * gs, -1), params, B * G, gs, gates=gates)
    assert rel_l2(tokens['t'].reshape(B * G, -1), te) < 6e-3
    te.backward(tokens['t'].grad.reshape(B * G, -1))
    tight = {n: rel_l2(p.grad, params[n].grad) for n, p in m.point_encoder.encoder.named_parameters()
             if not n.endswith(zero)}
    assert len(tight) == 12 and max(tight.values()) < 3e-2, tight


def test_full_model_reference_shape_train_step_runs():
    """The configuration BASELINE config 4 names (ViT-B width, 196 groups of 32, N=2048), small batch."""
    m = PA.AdaptPointFormer(num_classes=15, vit_name='vit_base_patch16_224', pretrained=False).to(DEV).train()
    pts = A.synthetic_clouds(4, 2048, seed=1).to(DEV)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.05)
    before = m.point_encoder.encoder.second_conv[3].weight.detach().clone()
    frozen = m.blocks[3].mlp.fc1.weight.detach().clone()
    loss = torch.nn.functional.cross_entropy(m(pts), torch.tensor([1, 2, 3, 4], device=DEV), label_smoothing=0.3)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss)
    assert not torch.equal(before, m.point_encoder.encoder.second_conv[3].weight)
    assert torch.equal(frozen, m.blocks[3].mlp.fc1.weight)


def test_full_size_train_step_is_bitwise_reproducible():
    """BASELINE config 4 at the benchmark's size (B=128, N=2048, xyz + height, ViT-B frozen): no kernel on this path uses
    atomics and every split reduction is summed in a fixed order, so repeated steps from the same state agree bit for bit
    (every dropout / DropPath rate 0: their masks are the only intended randomness)."""
    B, N = 128, 2048
    torch.manual_seed(0)
    m = PA.AdaptPointFormer(num_classes=15, in_channels=4, vit_name='vit_base_patch16_224', pretrained=False, dropout_rate=0.0,
                            dropout_path_rate=0.0).to(DEV).train()
    for mod in m.modules():                                 # the head's Dropout(0.4) layers (apf.py:222-223)
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    xyz = A.synthetic_clouds(B, N, seed=5)
    pts = torch.cat([xyz, xyz[:, :, 1:2] - xyz[:, :, 1:2].min(dim=1, keepdim=True).values], dim=-1).contiguous().to(DEV)
    start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(6)).to(DEV)
    labels = (torch.arange(B) % 15).to(DEV)
    bufs = [b.detach().clone() for b in m.buffers()]

    def run():
        for b, v in zip(m.buffers(), bufs):
            b.copy_(v)
        m.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(pts, start), labels, label_smoothing=0.3)
        loss.backward()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    l1, g1 = run()
    assert bool(torch.isfinite(l1)) and len(g1) > 20
    for r in range(4):
        l2, g2 = run()
        assert torch.equal(l1, l2), f'loss differs in run {r}'
        for n in g1:
            assert bool(torch.isfinite(g1[n]).all()), n
            assert torch.equal(g1[n], g2[n]), f'{n} differs in run {r}'


# --------------------------------------------------------------------------------------------- in_channels = 4 (xyz + height)
# the reference's shipped configuration (train_apf.py:5,29); fixtures: tests/golden/make_golden_apf_height.py
def _lexsorted_c(nb):
    key = np.lexsort([nb[..., c] for c in range(nb.shape[-1] - 1, -1, -1)], axis=-1)
    return np.take_along_axis(nb, key[..., None], axis=2)


@pytest.mark.parametrize('tag', ['n512g64', 'n2048g196'])
def test_group_with_height_channel_vs_reference_fixture(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, N, G, gs = (int(v) for v in z[f'group_{tag}_meta'])
    x = torch.from_numpy(z[f'group_{tag}_x']).to(DEV)
    start = torch.from_numpy(z[f'group_{tag}_start']).to(DEV)
    grp = PA.Group(G, gs)
    nb, cen = grp(x, x[:, :, :3], start)
    assert nb.shape == (B, G, gs, 8) and cen.shape == (B, G, 3)
    assert np.array_equal(cen.cpu().numpy(), z[f'group_{tag}_center']), 'centres (FPS + Morton order) must be bit-exact'
    nb_o, cen_o = A.Group(G, gs)(x.cpu(), x.cpu()[:, :, :3], start.cpu())
    assert torch.equal(nb.cpu(), nb_o) and torch.equal(cen.cpu(), cen_o)
    fps_idx, knn_idx, order = grp.indices(x[:, :, :3], start)
    clear = np.take_along_axis(z[f'group_{tag}_min_gap'], order.cpu().numpy().astype(np.int64), 1) > 2e-6
    assert clear.mean() > 0.9
    assert np.array_equal(_lexsorted_c(nb.cpu().numpy())[clear], z[f'group_{tag}_neigh_sorted'][clear])
    _, x0, _ = ops.apf_group(x, fps_idx, knn_idx, order)
    rec = (x0[:, 0:8].float() + x0[:, 8:16].float()).view(B, G, gs, 8)
    assert float((rec - nb).abs().max()) < 1e-4 and torch.equal(x0[:, 0:8], x0[:, 16:24]) and not x0[:, 24:].any()


@pytest.mark.parametrize('tag', ['ec128', 'ec768'])
def test_encoder_8_channels_against_reference_fixture(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, G, gs, ec = (int(v) for v in z[f'enc_{tag}_meta'])
    m = PA.Encoder(ec, 8)
    m.load_state_dict(A.seeded_state_dict(A.Encoder(ec, 8), seed=1400 + ec))
    m = m.to(DEV)
    x = torch.from_numpy(z[f'enc_{tag}_x']).to(DEV)
    m.eval()
    with torch.no_grad():
        assert rel_l2(m(x), torch.from_numpy(z[f'enc_{tag}_eval_tokens'])) < 1e-2
    m.train()
    FA._CAPTURE = {}
    try:
        tok = m(x)
    finally:
        gates, FA._CAPTURE = FA._CAPTURE, None
    assert rel_l2(tok, torch.from_numpy(z[f'enc_{tag}_train_tokens'])) < 1e-2
    w = torch.from_numpy(z[f'enc_{tag}_train_gradw']).to(DEV)
    (tok * w).sum().backward()
    # first layer: fp32-class through the hi/lo split (8 channels x 3 = 24 of the 64 padded K columns)
    run = np.stack([m.first_conv[1].running_mean.cpu().numpy(), m.first_conv[1].running_var.cpu().numpy()])
    np.testing.assert_allclose(run, z[f'enc_{tag}_bn1_running'], rtol=1e-4, atol=1e-6)
    assert rel_l2(m.second_conv[3].bias.grad, torch.from_numpy(z[f'enc_{tag}_g_sconv3_b'])) < 1e-5     # = column sums of w
    # every gradient against the gate-matched statement (arithmetic only) ...
    params = {n: p.detach().clone().requires_grad_(True) for n, p in m.named_parameters()}
    te = apf_encoder_emulated(x.reshape(B * G * gs, 8), params, B * G, gs, gates=gates)
    (te * w.view(B * G, ec)).sum().backward()
    zero = ('first_conv.0.bias', 'first_conv.3.bias', 'second_conv.0.bias', 'first_conv.6.bias')
    errs = {n: rel_l2(p.grad, params[n].grad) for n, p in m.named_parameters() if n not in zero}
    assert max(errs.values()) < 2e-2, errs
    assert m.first_conv[0].weight.grad.shape == (256, 8, 1)
    # ... and the first layer against the reference fixture (fp32 gates: bounded as in the 6-channel test)
    assert rel_l2(m.first_conv[0].weight.grad.reshape(256, 8), torch.from_numpy(z[f'enc_{tag}_g_conv0_w'])) < 0.25


def test_pointnet_with_height_channel_vs_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, 'apf_height.npz'))
    B, N, G, gs, ec = (int(v) for v in z['pointnet_meta'])
    pn = PA.PointNet(ec, G, gs, 8)
    pn.load_state_dict(A.seeded_state_dict(A.PointNet(ec, G, gs, 8), seed=1500))
    pn = pn.to(DEV).eval()
    with torch.no_grad():
        tok = pn(torch.from_numpy(z['pointnet_x']).to(DEV), torch.from_numpy(z['pointnet_start']).to(DEV))
    assert tok.shape == (B, G, ec)
    assert rel_l2(tok, torch.from_numpy(z['pointnet_tokens'])) < 1e-2
    with pytest.raises(RuntimeError, match='input channels'):
        pn(torch.zeros(B, N, 3, device=DEV))


def test_full_model_with_height_channel_vs_oracle_and_train_step():
    o, m = _model_pair(41, in_channels=4, dropout_rate=0.0, dropout_path_rate=0.0)
    B, N = 6, 512
    xyz = A.synthetic_clouds(B, N, seed=42)
    x = torch.cat([xyz, xyz[:, :, 1:2] - xyz[:, :, 1:2].min(dim=1, keepdim=True).values], dim=-1)     # augment.py:247-248
    start = torch.arange(B) * 3
    o.eval(), m.eval()
    with torch.no_grad():
        want, got = o(x, start), m(x.to(DEV), start.to(DEV)).cpu()
    assert float((got - want).abs().max()) < 1e-2 * max(1.0, float(want.abs().max()))
    m.train()
    loss = torch.nn.functional.cross_entropy(m(x.to(DEV), start.to(DEV)), torch.arange(B, device=DEV) % 15, label_smoothing=0.3)
    loss.backward()
    g = m.point_encoder.encoder.first_conv[0].weight.grad
    assert g.shape == (256, 8, 1) and bool(torch.isfinite(g).all()) and float(g[:, 3].abs().sum()) > 0     # the height column learns
