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
