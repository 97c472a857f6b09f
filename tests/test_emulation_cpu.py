"""The rounding-matched statement of the whole model (tests/emulate.py) is test infrastructure for the GPU parity tests;
here it is tied to the oracle on the CPU: with every rounding switched off it IS the fp32 oracle (so the only thing it
adds is where the HIP pipeline rounds to bf16), with the roundings on it stays within the bf16 operand noise of it."""
import pytest
import torch

from oracle import pix4point_oracle as O
from tests import emulate as E


def _setup(D, L, H, B, N, seed):
    o = O.Pix4Point(40, D, L, H)
    o.load_state_dict(O.seeded_state_dict(o, seed=seed))
    pts = O.synthetic_clouds(B, N, seed=seed + 1)
    start = torch.arange(B) * 5
    idx = O.farthest_point_sampling(pts, N // 4, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    return o, pts, start, idx, O.knn_indices(cen, pts, 16)


@pytest.mark.parametrize('training', [False, True])
def test_unrounded_statement_is_the_oracle(training):
    o, pts, start, idx, nb = _setup(192, 2, 3, 4, 320, seed=3)        # T = 81: two key tiles, the second ragged
    for m in o.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    o.train(training)
    E.ROUND = False
    try:
        with torch.no_grad():
            got = E.pix4point_emulated(o, pts, idx, nb, training=training)
    finally:
        E.ROUND = True
    if training:        # the oracle's train-mode forward updates running statistics: compare on a copy
        import copy
        o = copy.deepcopy(o)
    with torch.no_grad():
        want = o(pts, start)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_rounded_statement_stays_within_bf16_noise_of_the_oracle():
    o, pts, start, idx, nb = _setup(192, 2, 3, 2, 256, seed=5)
    o.eval()
    with torch.no_grad():
        want = o(pts, start)
        got = E.pix4point_emulated(o, pts, idx, nb)
    err = float((got - want).abs().max())
    assert 1e-5 < err < 1e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('T', [150, 129, 257, 321])     # online; two-pass (64 n + 1 <= 257); online again
def test_attention_statement_equals_softmax_attention_when_unrounded(T):
    g = torch.Generator().manual_seed(0)
    B, H = 2, 3
    qkv = torch.randn(B * T, 3 * H * 64, generator=g)
    E.ROUND = False
    try:
        got = E.attention_emulated(qkv, B, T, H)
    finally:
        E.ROUND = True
    q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert float((got - want).abs().max()) < 1e-5


def test_attention_statement_of_the_kv_resident_forward_equals_softmax_attention_when_unrounded():
    """'online32' (csrc/attention_fwd2.h, T = 513): 32-key tiles, the odd query as eight partial chains combined in wave order."""
    g = torch.Generator().manual_seed(1)
    B, H, T = 1, 2, 513
    qkv = torch.randn(B * T, 3 * H * 64, generator=g) * 1.5
    E.ROUND = False
    try:
        got = E.attention_emulated(qkv, B, T, H, mode='online32')
    finally:
        E.ROUND = True
    q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert float((got - want).abs().max()) < 2e-5


def test_lazy_rescale_of_the_online_statement():
    """Round 4: the reference m of a 16-row query tile follows the running maximum only when a row outgrew it by more than 2^ONLINE_THR.
    (i) Threshold 0 is the exact running maximum (rounds 1-3).  (ii) Some tile of these inputs does skip an update -- the statement
    differs from the exact form in roundings -- and (iii) it is as close to fp32 softmax attention as the exact form (same bf16 noise).
    (iv) A late dominant key (beyond the threshold) still moves every row of its tile."""
    g = torch.Generator().manual_seed(2)
    B, H, T = 1, 2, 321
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.5).bfloat16().float()
    q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    lazy = E.attention_emulated(qkv, B, T, H)
    thr = E.ONLINE_THR
    try:
        E.ONLINE_THR = 0.0
        exact = E.attention_emulated(qkv, B, T, H)
        c = float(torch.tensor(64 ** -0.5) * torch.tensor(E.LOG2E))
        s = q @ k.transpose(-1, -2)
        groups = [list(range(k0, min(k0 + E.KEY_TILE, T))) for k0 in range(0, T, E.KEY_TILE)]
        m0, l0, o0 = E._online_chain(s, v, c, groups)                                  # no threshold argument: rounds 1-3
        m1, l1, o1 = E._online_chain(s, v, c, groups, thr=0.0, rows=16)
        assert torch.equal(m0, m1) and torch.equal(l0, l1) and torch.equal(o0, o1)
    finally:
        E.ONLINE_THR = thr
    assert not torch.equal(lazy, exact)
    e_lazy, e_exact = float((lazy - want).abs().max()), float((exact - want).abs().max())
    assert e_lazy < 1e-2 and e_exact < 1e-2 and e_lazy < 1.5 * e_exact + 1e-4
    spike = qkv.clone().view(B, T, 3, H, 64)
    spike[:, T - 1, 1] = spike[:, 5, 0] * 4                                           # key T-1 dominates query 5 (and its tile decides together)
    spike = spike.view(B * T, -1)
    q, k, v = spike.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert float((E.attention_emulated(spike, B, T, H) - want).abs().max()) < 2e-2
