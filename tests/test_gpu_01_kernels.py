"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

Integer / index work is compared bit-exactly with the oracle and the reference-generated fixtures;
floating-point kernels against a plain PyTorch fp32 computation of the same op on the same
(bf16-rounded) inputs, with the tolerance stated in each test.  No test reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import p4p_amd                                   # noqa: E402
from p4p_amd import ops                          # noqa: E402
from oracle import pix4point_oracle as O         # noqa: E402  (checker only)

DEV = 'cuda'


def _loaded():
    assert os.path.exists(p4p_amd._lib.LIB_PATH)
    p4p_amd._lib.lib()


# ------------------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize('tag', ['n64', 'n100', 'n1024', 'n2048', 'n8192', 'dup1024'])
def test_fps_matches_reference_fixture(golden_dir, tag):
    _loaded()
    z = np.load(os.path.join(golden_dir, 'fps.npz'))
    pts = torch.from_numpy(z[f'{tag}_points']).to(DEV)
    start = torch.from_numpy(z[f'{tag}_start']).to(DEV)
    got = ops.fps(pts, pts.shape[1] // 4, start)
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), z[f'{tag}_idx'].astype(np.int64))


@pytest.mark.parametrize('B,N', [(5, 300), (3, 1000), (2, 4096), (1, 12288)])
def test_fps_matches_oracle_ragged_sizes(B, N):
    pts = O.synthetic_clouds(B, N, seed=B * 7 + N)
    start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(N))
    want = O.farthest_point_sampling(pts, N // 4, start)
    got = ops.fps(pts.to(DEV), N // 4, start.to(DEV)).cpu()
    assert torch.equal(got, want)


def test_fps_full_size_properties():
    """BASELINE config sizes (B=256, N=1024 / N=2048): size-independent properties + a sampled oracle check."""
    for N in (1024, 2048):
        B = 256
        pts = O.synthetic_clouds(B, N, seed=99 + N).to(DEV)
        start = torch.randint(0, N, (B,), device=DEV)
        idx = ops.fps(pts, N // 4, start)
        assert idx.shape == (B, N // 4)
        assert torch.equal(idx[:, 0], start)
        assert int(idx.min()) >= 0 and int(idx.max()) < N
        s = torch.sort(idx, dim=1).values
        assert bool((s[:, 1:] != s[:, :-1]).all()), 'a point was sampled twice'
        # running min distance to the chosen set must be non-increasing along the sampling order
        c = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        d = torch.cdist(c[:8].double(), c[:8].double())
        tri = torch.tril(torch.ones(N // 4, N // 4, device=DEV, dtype=torch.bool), -1)
        dmin = torch.where(tri, d, torch.full_like(d, 1e9)).min(dim=2).values[:, 1:]
        assert bool((dmin[:, 1:] <= dmin[:, :-1] + 1e-6).all())
        sel = [0, 17, 255]
        want = O.farthest_point_sampling(pts[sel].cpu(), N // 4, start[sel].cpu())
        assert torch.equal(idx[sel].cpu(), want)
        again = ops.fps(pts, N // 4, start)
        assert torch.equal(idx, again), 'not deterministic'


def test_fps_more_samples_than_points_is_clamped():
    pts = O.synthetic_clouds(2, 16, seed=1).to(DEV)
    got = ops.fps(pts, 64, torch.tensor([3, 5], device=DEV))
    assert got.shape == (2, 16)
    assert all(sorted(r.tolist()) == list(range(16)) for r in got.cpu())


def test_fps_argument_errors():
    pts = O.synthetic_clouds(2, 64, seed=1)
    with pytest.raises(RuntimeError):
        ops.fps(pts, 16)                                   # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.fps(pts.to(DEV).double(), 16)
    with pytest.raises(RuntimeError):
        ops.fps(torch.zeros(2, 64, 4, device=DEV), 16)


# ------------------------------------------------------------------------------------------ kNN
@pytest.mark.parametrize('tag', ['n256k16', 'n1024k16', 'n1024k32', 'n2048k16'])
def test_knn_matches_oracle_and_reference_fixture(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'knn.npz'))
    pts = torch.from_numpy(z[f'{tag}_points'])
    cidx = torch.from_numpy(z[f'{tag}_cidx'].astype(np.int64))
    ref = z[f'{tag}_ref_idx'].astype(np.int32)
    k = ref.shape[-1]
    cen = torch.gather(pts, 1, cidx.unsqueeze(-1).expand(-1, -1, 3))
    got = ops.knn(pts.to(DEV), cen.to(DEV), k).cpu()
    assert got.dtype == torch.int32
    assert torch.equal(got, O.knn_indices(cen, pts, k)), 'differs from the exact-distance oracle'
    clear = z[f'{tag}_min_gap'] > 2e-6
    assert np.array_equal(got.numpy()[clear], ref[clear]), 'differs from reference cdist+topk on well-separated rows'


@pytest.mark.parametrize('B,N,S,k', [(3, 100, 25, 16), (2, 1500, 375, 16), (1, 8192, 2048, 16), (2, 64, 16, 64),
                                     (1, 3000, 10, 1)])
def test_knn_matches_oracle_ragged(B, N, S, k):
    pts = O.synthetic_clouds(B, N, seed=N + k, dup_frac=0.05)
    cen = pts[:, torch.randperm(N, generator=torch.Generator().manual_seed(1))[:S]].contiguous()
    want = O.knn_indices(cen, pts, k)
    got = ops.knn(pts.to(DEV), cen.to(DEV), k).cpu()
    assert torch.equal(got, want)


def test_knn_ties_lowest_index_and_errors():
    pts = torch.zeros(1, 8, 3)
    pts[0, :, 0] = torch.tensor([0., 1., -1., 1., 2., -2., 0., 3.])
    got = ops.knn(pts.to(DEV), pts[:, :1].contiguous().to(DEV), 6).cpu()[0, 0].tolist()
    assert got == [0, 6, 1, 2, 3, 4]
    with pytest.raises(RuntimeError):
        ops.knn(pts.to(DEV), pts[:, :1].contiguous().to(DEV), 9)        # k > N


def test_knn_full_size_properties():
    B, N, k = 256, 1024, 16
    pts = O.synthetic_clouds(B, N, seed=5).to(DEV)
    idx = ops.fps(pts, N // 4, torch.zeros(B, dtype=torch.long, device=DEV))
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    nb = ops.knn(pts, cen, k)
    assert nb.shape == (B, N // 4, k)
    assert torch.equal(nb[..., 0].long(), idx), 'nearest neighbour of a centroid is itself'
    g = ops.group_gather(pts, nb)
    d = ((g - cen.unsqueeze(2)) ** 2).sum(-1)
    assert bool((d[..., 1:] >= d[..., :-1]).all()), 'neighbours not sorted by distance'
    # the k-th distance is a lower bound for every unselected point (checked on a few clouds)
    full = torch.cdist(cen[:4], pts[:4]) ** 2
    kth = d[:4, :, -1:]
    assert int((full < kth - 1e-6).sum()) <= 4 * (N // 4) * (k - 1)
    sel = [3, 200]
    assert torch.equal(nb[sel].cpu(), O.knn_indices(cen[sel].cpu(), pts[sel].cpu(), k))
    bi = torch.arange(B, device=DEV).view(B, 1, 1)
    assert torch.equal(g, pts[bi, nb.long()])


# ------------------------------------------------------------------------------------------ GEMM
def _int_bf16(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 384, 192), (100, 64, 128), (2056, 576, 192), (515, 192, 768),
                                   (1000, 768, 3072)])
def test_gemm_tn_exact_on_integer_data(M, N, K):
    """Small-integer operands make every product and partial sum exact in fp32 and in the bf16
    output, so any layout / swizzle / k-order mistake shows up as an exact mismatch."""
    a = _int_bf16((M, K), -2, 2, 1).to(DEV)
    w = _int_bf16((N, K), -1, 1, 2).to(DEV)
    if K > 512:
        w = w * (torch.rand(N, K, device=DEV) < 0.1)      # keep |sum| < 256 so bf16 output is exact
        w = w.to(torch.bfloat16).contiguous()
    want = a.float() @ w.float().t()
    assert float(want.abs().max()) <= 256
    got = ops.gemm_tn(a, w, ops.EPI_BF16)
    assert torch.equal(got.float(), want)
    bias = torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2
    got32 = ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=want.clone())
    assert torch.equal(got32, 2 * want + bias)


@pytest.mark.parametrize('variant', [3, 4, 8, 10])
@pytest.mark.parametrize('M', [300 * 256, 300 * 256 - 100])
def test_gemm_tn_big_tile_kernels_exact_on_integer_data(M, variant):
    """The 256-row tile kernels (mid-tile-barrier, 256x128 ring, 256x256 ring) on a grid of 600 tiles
    (with a ragged last panel): exact against fp32 on integer operands, and the
    GELU / dGELU epilogues bit-identical to the 128x128 kernel's."""
    from p4p_amd import _lib
    N, K = 512, 128
    a = _int_bf16((M, K), -2, 2, 11).to(DEV)
    w = _int_bf16((N, K), -1, 1, 12).to(DEV)
    want = a.float() @ w.float().t()
    bias = torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2
    pre = (torch.arange(M * N, device=DEV) % 7 - 3).to(torch.bfloat16).view(M, N)
    lib = _lib.lib()
    old = lib.p4p_set_gemm_variant(1)
    try:
        ref_gelu, ref_pre = ops.gemm_tn(a, w, ops.EPI_BF16_GELU, bias=bias, want_pre=True)
        ref_dgelu = ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre)
        lib.p4p_set_gemm_variant(variant)
        got = ops.gemm_tn(a, w, ops.EPI_BF16)
        got32 = ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=want.clone())
        gelu, gpre = ops.gemm_tn(a, w, ops.EPI_BF16_GELU, bias=bias, want_pre=True)
        dgelu = ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre)
    finally:
        lib.p4p_set_gemm_variant(old)
    assert torch.equal(got.float(), want)
    assert torch.equal(got32, 2 * want + bias)
    assert torch.equal(gelu, ref_gelu) and torch.equal(gpre, ref_pre)
    assert torch.equal(dgelu, ref_dgelu)


@pytest.mark.parametrize('M,N,K', [(300 * 256, 768, 256), (300 * 256 - 100, 768, 768), (513 * 128, 1024, 384), (257 * 256, 2304, 768),
                                   (700 * 256 + 17, 256, 2304), (1100 * 256, 256, 128)])
def test_gemm_tn_persistent_tile_loop_is_exact_and_equals_the_per_tile_kernel(M, N, K):
    """gemm_tn4p.h (several tiles per workgroup, the next tile's first K-tiles staged across the seam, counted waits over the
    epilogue's own stores): exact on integer operands, bit-identical to the one-workgroup-per-tile kernel for every epilogue,
    with a ragged last row panel (no seam there), with 2 .. 4+ tiles per workgroup, and K that is not eligible (128: plain path)."""
    from p4p_amd import _lib
    lib = _lib.lib()
    a = _int_bf16((M, K), -2, 2, 21).to(DEV)
    w = (_int_bf16((N, K), -1, 1, 22) * (torch.rand(N, K, generator=torch.Generator().manual_seed(5)) < (0.25 if K > 512 else 1.0))).to(torch.bfloat16).to(DEV)
    want = a.float() @ w.float().t()
    bias = torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2
    resid = (torch.arange(M * N, device=DEV) % 11 - 5).float().view(M, N)
    pre = (torch.arange(M * N, device=DEV) % 7 - 3).to(torch.bfloat16).view(M, N)
    outs = {}
    for mode in (0, 1):
        old = lib.p4p_set_gemm_persistent(mode)
        try:
            outs[mode] = (ops.gemm_tn(a, w, ops.EPI_BF16, bias=bias),
                          ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=resid),
                          ops.gemm_tn(a, w, ops.EPI_BF16_GELU, bias=bias, want_pre=True),
                          ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre, want_colsum=True))
        finally:
            lib.p4p_set_gemm_persistent(old)
    assert torch.equal(outs[1][0], (want + bias).to(torch.bfloat16))
    assert torch.equal(outs[1][1], (want + bias) + resid)
    for x, y in zip(outs[0], outs[1]):
        if isinstance(x, tuple):
            assert all(torch.equal(u, v) for u, v in zip(x, y))
        else:
            assert torch.equal(x, y)


def test_gemm_tn_persistent_tile_loop_token_stream_rows():
    """The tokenizer -> token-stream projection at the benchmark's size (B = 256 clouds of S = 256 tokens, 256 -> 768): the fp32
    epilogue writes row m of the compact [B*S] rows to row m + m/S + 1 of the [B*(S+1)] stream (one cls row in front of every
    sequence, pix4point.py:248-252 without the torch.cat).  768 tiles, three per workgroup: the persistent kernel; exact on
    integer data, cls rows untouched, identical to the per-tile kernel."""
    from p4p_amd import _lib
    lib = _lib.lib()
    B, S, N, K = 256, 256, 768, 256
    a = _int_bf16((B * S, K), -2, 2, 31).to(DEV)
    w = _int_bf16((N, K), -1, 1, 32).to(DEV)
    bias = torch.arange(N, device=DEV, dtype=torch.float32) % 7 - 3
    want = (a.float() @ w.float().t() + bias).view(B, S, N)
    outs = []
    for mode in (0, 1):
        old = lib.p4p_set_gemm_persistent(mode)
        try:
            out = torch.full((B * (S + 1), N), -777.0, device=DEV)
            ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, group=S, out=out)
        finally:
            lib.p4p_set_gemm_persistent(old)
        outs.append(out.view(B, S + 1, N))
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[1][:, 1:], want)
    assert bool((outs[1][:, 0] == -777.0).all())


@pytest.mark.parametrize('K', [2304, 3072])
def test_gemm_tn_split_k_tail_exact_on_integer_data(K):
    """M = 257 panels of 256 rows, N = 768: three full rounds of 256x256 tiles plus one panel, which is peeled off and --
    for K >= 1536 -- contracted in K slices by the 64x64 kernel into fp32 planes that a finishing kernel adds in a fixed
    order (bias / residual / bf16 cast applied there).  Exact on integer operands, including the peeled rows."""
    M, N = 257 * 256, 768
    a = _int_bf16((M, K), -2, 2, 21).to(DEV)
    w = (_int_bf16((N, K), -1, 1, 22).to(DEV) * (torch.rand(N, K, device=DEV) < 0.05)).to(torch.bfloat16).contiguous()
    want = a.float() @ w.float().t()
    assert float(want.abs().max()) <= 256
    bias = torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2
    got = ops.gemm_tn(a, w, ops.EPI_BF16, bias=bias)
    assert torch.equal(got.float(), want + bias)
    got32 = ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=want.clone())
    assert torch.equal(got32, 2 * want + bias)


@pytest.mark.parametrize('M', [70000, 256 * 300])
def test_gemm_tn_statistics_epilogues_on_the_big_tile_kernel(M):
    """EPI_BF16_ROWGROUP / EPI_BF16_STATS at tokenizer-like sizes take the 256x256 staggered kernel: same outputs as the
    128x128 kernel (bit-identical: same k order), statistics rows per 128-row block equal up to summation order."""
    from p4p_amd import _lib
    N, K, grp = 512, 256, 16
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    rg = torch.randn(M // grp, N, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    lib = _lib.lib()
    old = lib.p4p_set_gemm_variant(1)
    try:
        o1, s1 = ops.gemm_tn(a, w, ops.EPI_BF16_ROWGROUP, rowgrp=rg, group=grp)
        o2, s2 = ops.gemm_tn(a, w, ops.EPI_BF16_STATS, bias=bias)
        lib.p4p_set_gemm_variant(0)
        p1, t1 = ops.gemm_tn(a, w, ops.EPI_BF16_ROWGROUP, rowgrp=rg, group=grp)
        p2, t2 = ops.gemm_tn(a, w, ops.EPI_BF16_STATS, bias=bias)
    finally:
        lib.p4p_set_gemm_variant(old)
    assert torch.equal(o1, p1) and torch.equal(o2, p2)
    assert s1.shape == t1.shape and s2.shape == t2.shape
    torch.testing.assert_close(t1, s1, rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(t2, s2, rtol=1e-4, atol=1e-2)
    full = a.float() @ w.float().t() + bias
    torch.testing.assert_close(t2.sum(0)[0], full.sum(0), rtol=1e-4, atol=0.5)
    torch.testing.assert_close(t2.sum(0)[1], (full * full).sum(0), rtol=1e-4, atol=0.5)


def test_gemm_tn_epilogues_random():
    M, N, K = 777, 320, 256
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = a.float() @ w.float().t() + bias
    out, pre = ops.gemm_tn(a, w, ops.EPI_BF16_GELU, bias=bias, want_pre=True)
    # tolerance: one bf16 rounding of the output (2^-8 relative) on top of fp32 accumulation
    xg = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(xg).sum().backward()
    torch.testing.assert_close(pre.float(), xg.grad, rtol=1e-2, atol=1e-2)        # `pre` holds gelu'(h)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(ref), rtol=1e-2, atol=1e-2)
    # the fast erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7) against libm on a dense grid incl. the tails
    xs = torch.linspace(-9, 9, 128 * 64, device=DEV).view(128, 64)
    eye = torch.eye(64, device=DEV).to(torch.bfloat16)
    xb = xs.to(torch.bfloat16)
    gl = ops.gemm_tn(xb, eye, ops.EPI_BF16_GELU, bias=torch.zeros(64, device=DEV))
    torch.testing.assert_close(gl.float(), torch.nn.functional.gelu(xb.float()), rtol=8e-3, atol=1e-6)
    dg = ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre)
    torch.testing.assert_close(dg.float(), (a.float() @ w.float().t()) * pre.float(), rtol=2e-2, atol=2e-2)
    # row-group broadcast + column statistics
    grp = 7
    rg = torch.randn(M // grp, N, generator=g).to(DEV)
    out, st = ops.gemm_tn(a, w, ops.EPI_BF16_ROWGROUP, rowgrp=rg, group=grp)
    full = a.float() @ w.float().t() + rg.repeat_interleave(grp, 0)
    torch.testing.assert_close(out.float(), full, rtol=1e-2, atol=1e-2)
    s = st.sum(0)
    torch.testing.assert_close(s[0], full.sum(0), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(s[1], (full * full).sum(0), rtol=1e-4, atol=1e-2)
    out, st = ops.gemm_tn(a, w, ops.EPI_BF16_STATS, bias=bias)
    torch.testing.assert_close(st.sum(0)[0], ref.sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize('variant', [0, 1, 3, 4, 6, 8, 10])
@pytest.mark.parametrize('M,N,K', [(777, 512, 256), (66 * 256, 768, 256), (2049, 256, 128), (300 * 256, 512, 128)])
def test_gemm_tn_dgelu_fused_bias_gradient(M, N, K, variant):
    """EPI_BF16_DGELU with want_colsum: the column sums of the stored (bf16) output, from the GEMM epilogue on the ring
    kernels (incl. the peeled tail of the 66-panel case) and from the streaming fallback on the other tile shapes."""
    g = torch.Generator().manual_seed(M + variant)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    pre = torch.rand(M, N, generator=g).to(torch.bfloat16).to(DEV)
    from p4p_amd import _lib
    old = _lib.lib().p4p_set_gemm_variant(variant)
    try:
        plain = ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre)
        out, cs = ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre, want_colsum=True)
    finally:
        _lib.lib().p4p_set_gemm_variant(old)
    assert torch.equal(out, plain)
    want = out.double().sum(0)
    assert (cs.double() - want).abs().max().item() <= 1e-5 * max(out.float().abs().sum(0).max().item(), 1.0)


@pytest.mark.parametrize('M,N,K', [(64, 128, 128), (200, 256, 192), (2056, 576, 192), (4099, 192, 768), (1000, 8, 64)])
def test_gemm_wgrad_exact_on_integer_data(M, N, K):
    g = _int_bf16((M, N), -1, 1, 3).to(DEV)
    a = _int_bf16((M, K), -2, 2, 4).to(DEV)
    want = g.float().t() @ a.float()
    got = ops.gemm_wgrad(g, a)
    assert torch.equal(got, want)
    acc = ops.gemm_wgrad(g, a, out=got.clone(), beta=1.0)
    assert torch.equal(acc, 2 * want)
    dw, db = ops.gemm_wgrad(g, a, want_bias=True)          # bias gradient from the same pass
    assert torch.equal(dw, want) and torch.equal(db, g.float().sum(0))
    assert torch.equal(ops.colsum(g), g.float().sum(0))


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize('M,D', [(33, 192), (257, 384), (1000, 768), (64, 1024)])
def test_layernorm_fwd_bwd(M, D):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(M, D, generator=g).to(DEV)
    pos = torch.randn(M, D, generator=g).to(DEV)
    gamma = (torch.randn(D, generator=g) * 0.1 + 1).to(DEV)
    beta = (torch.randn(D, generator=g) * 0.1).to(DEV)
    y, xs, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6, pos=pos, want_sum=True, out_f32=True)
    xr = (x + pos).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gamma, beta, 1e-6)
    assert torch.equal(xs, x + pos)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)          # fp32 in, fp32 out
    yb, _, _, _ = ops.layernorm_fwd(x, gamma, beta, 1e-6, pos=pos)
    torch.testing.assert_close(yb.float(), ref, rtol=8e-3, atol=8e-3)  # one bf16 rounding
    dy = torch.randn(M, D, generator=g).to(DEV)
    gr = torch.nn.Parameter(gamma.clone())
    br = torch.nn.Parameter(beta.clone())
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy)
    base = torch.randn(M, D, generator=g).to(DEV)
    dx, dg, db, dsum = ops.layernorm_bwd(dy, xs, mean, rstd, gamma, dx=base.clone(), want_dxsum=True)
    torch.testing.assert_close(dx, base + xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dsum, dx.sum(0), rtol=1e-4, atol=1e-3)     # bias gradient by-product
    torch.testing.assert_close(dg, gr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, br.grad, rtol=1e-4, atol=1e-3)
    dx2, _, _ = ops.layernorm_bwd(dy.to(torch.bfloat16), xs, mean, rstd, gamma)
    torch.testing.assert_close(dx2, xr.grad, rtol=2e-2, atol=2e-2)
    # out of place (dx = dx_in + LN'(dy), dx_in left alone): the same values, bit for bit, as in place -- for both factors
    for fac in (1, 2):
        inplace, _, _ = ops.layernorm_bwd(dy, xs, mean, rstd, gamma, dx=base.clone(), dx_factor=fac)
        src, dst = base.clone(), torch.full_like(base, float('nan'))
        out, _, _ = ops.layernorm_bwd(dy, xs, mean, rstd, gamma, dx=dst, dx_in=src, dx_factor=fac)
        assert out.data_ptr() == dst.data_ptr() and torch.equal(dst, inplace) and torch.equal(src, base)


def test_sum_slabs_adds_in_slab_order():
    """p4p_sum_slabs_f32: the positional-embedding gradient as ONE pass over the block-input gradients, summed in slab order (the order
    the read-modify-write it replaces had): bit-equal to a sequential fp32 sum, not just close to it."""
    g = torch.Generator().manual_seed(3)
    for S, shape in ((1, (8, 4)), (3, (5, 12)), (12, (257, 768)), (24, (33, 1024))):
        slabs = (torch.randn(S, *shape, generator=g) * torch.logspace(-3, 3, S).view(S, 1, 1)).to(DEV)
        want = slabs[0].clone()
        for k in range(1, S):
            want = want + slabs[k]
        got = ops.sum_slabs(slabs)
        assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(RuntimeError):
        ops.sum_slabs(torch.zeros(2, 3, 5, device=DEV))            # slab size not a multiple of 4


# ------------------------------------------------------------------------------------------ attention
ATTN_BWD_FLIPS = 4.0        # numerators rounded the other way that one gradient element may be away from a correct rounding of the statement


def _attn_ref(qkv, B, T, H):
    D = qkv.shape[1] // 3
    x = qkv.float().reshape(B, T, 3, H, D // H).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    a = ((q @ k.transpose(-2, -1)) * (D // H) ** -0.5).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize('B,T,H', [(2, 1, 1), (2, 17, 3), (3, 64, 2), (2, 65, 3), (2, 100, 2), (1, 129, 1), (2, 192, 2),
                                   (2, 256, 2), (2, 257, 3), (2, 258, 1), (2, 321, 2), (1, 513, 2), (1, 2049, 1),
                                   (2, 2049, 16)])          # BASELINE configs[4]: ViT-L heads at N=8192
def test_attention_fwd_bwd(B, T, H):
    D = 64 * H
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, B, T, H)
    # tolerance: P and the output are rounded to bf16 (2^-8); values are O(1)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    # against the rounding-matched statement (same tile order, P rounded to bf16): a correct rounding of it, up to the
    # P roundings inside the kernel (tests/emulate.py: link_stats 'att'; these inputs make a PEAKED softmax, where one
    # flipped numerator moves an output by up to 2^-9 p |v|: measured 0.63, against 0.51 on the model's own activations)
    from tests import emulate as E
    with torch.no_grad():
        assert E.link_stats(out, E.attention_emulated(qkv.float(), B, T, H, round_output=False), 'att',
                            E.attention_flip_unit(qkv, B, T, H)) <= E.LINK_LIMITS['att']
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).to(DEV)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv, out, dout, lse, B, T, H)
    err = (dqkv.float() - qr.grad).abs().max().item()
    scale = qr.grad.abs().max().item()
    assert err <= 3e-2 * max(scale, 1.0), (err, scale)           # against fp32 autograd: the bf16 noise of P / dS (loose by nature)
    # against the rounding-matched statement (tests/emulate.py: attention_backward_emulated, the forward's lse2 teacher-forced): every
    # stored gradient element a correct bf16 rounding of the stated value up to a few numerators (P for dV, dS for dQ / dK) rounded
    # the other way -- a PER-ELEMENT bound, where the line above allows 3 % of the tensor's maximum everywhere
    if B * H * T * T <= 2.2e7:
        with torch.no_grad():
            want, unit = E.attention_backward_emulated(qkv, out, dout, lse, B, T, H)
            flips = E.flips_beyond_rounding(dqkv, want, unit)
        assert flips <= ATTN_BWD_FLIPS, (flips, B, T, H)
    # fused qkv-bias gradient: per-block column sums of the (rounded) dqkv from inside the backward kernels
    dqkv2, dbias = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    assert torch.equal(dqkv2, dqkv)
    want = dqkv.double().sum(0)
    assert (dbias.double() - want).abs().max().item() <= 1e-5 * max(dqkv.float().abs().sum(0).max().item(), 1.0)
    # forces the online-softmax rescale: one key dominates late in the sequence
    if T >= 65:
        spike = qkv.clone().view(B, T, 3, H, 64)
        spike[:, T - 1, 1] = spike[:, 0, 0] * 4
        spike = spike.view(B * T, 3 * D).contiguous()
        out2, _ = ops.attention_fwd(spike, B, T, H)
        torch.testing.assert_close(out2.float(), _attn_ref(spike, B, T, H), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('B,T,H', [(3, 65, 2), (7, 129, 5), (2, 257, 12), (37, 257, 3), (2, 513, 12), (9, 513, 2)])
def test_attention_backward_fused_kernels_against_the_two_kernel_path(B, T, H):
    """Round 3: attn_bwd_fused_kernel (T = 65 / 129 / 257: one kernel for dQ, dK, dV) and attn_bwd_dkdv_ks_kernel (T = 513) against
    the kernels they replace, switched back in through the tools' variant switch: the same arithmetic contract (P and dS rounded to
    bf16 where they enter a product), so every element within a couple of bf16 ulps of the other path's (a single flipped
    numerator), the same error against fp32 autograd, and the column sums = the sums of what was stored."""
    from p4p_amd import _lib
    lib = _lib.lib()
    D = 64 * H
    g = torch.Generator().manual_seed(1000 * T + B)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    try:
        lib.p4p_set_attention_variant(8 | 64)           # bit 3: no fused backward; bit 6: no key-stationary dK/dV
        old, old_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
        old, old_cs = old.clone(), old_cs.clone()
    finally:
        lib.p4p_set_attention_variant(0)
    new, new_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    assert torch.isfinite(new.float()).all()
    d = (new.float() - old.float()).abs()
    tol = 4 * 2.0 ** -8 * torch.maximum(old.float().abs(), new.float().abs()) + 1e-3
    assert (d <= tol).all(), float((d - tol).max())
    assert float((d > 0).float().mean()) < 0.02        # and almost all elements are bit-identical
    qr = qkv.float().requires_grad_(True)
    _attn_ref(qr, B, T, H).backward(dout.float())
    sc = max(qr.grad.abs().max().item(), 1.0)
    assert (new.float() - qr.grad).abs().max().item() <= 3e-2 * sc
    want = new.double().sum(0)
    assert (new_cs.double() - want).abs().max().item() <= 1e-5 * max(new.float().abs().sum(0).max().item(), 1.0)
    # the odd row (cls token: the last one) has the same treatment as every other row
    last = d.view(B, T, 3 * D)[:, T - 1]
    assert (last <= tol.view(B, T, 3 * D)[:, T - 1]).all()


@pytest.mark.parametrize('B,H', [(1, 1), (2, 12), (23, 12), (50, 6)])
def test_attention_backward_fused_t513_against_the_kernels_it_replaces(B, H):
    """Round 4: attn_bwd_fused2_kernel (T = 513: dQ, dK, dV in one kernel; a workgroup takes the two key halves of a pair one after the
    other, half 0's dQ chains parked in a scratch) against attn_bwd_dq_q_kernel + attn_bwd_dkdv_ks_kernel (variant bit 4): the same
    arithmetic contract, every element within single bf16 flips of the other path; the column sums are the sums of what
    was stored; repeated launches are bitwise equal (the scratch is rewritten per pair); pair counts below / above / not a multiple
    of the workgroup count; and the usual error against fp32 autograd."""
    from p4p_amd import _lib
    lib = _lib.lib()
    T, D = 513, 64 * H
    g = torch.Generator().manual_seed(513 + 7 * B + H)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    try:
        lib.p4p_set_attention_variant(16)
        old, old_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
        old, old_cs = old.clone(), old_cs.clone()
    finally:
        lib.p4p_set_attention_variant(0)
    new, new_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    new, new_cs = new.clone(), new_cs.clone()
    assert torch.isfinite(new.float()).all()
    # delta = rowsum(dO . O) is summed in another order here (fma chains over two half rows, as in attn_bwd_fused_kernel; the
    # streaming dQ kernel adds four 16-column groups), so dP differs in its last fp32 bit here and there and flips a bf16 rounding
    # of dS now and then: dK / dV equal the key-stationary kernel's up to such single flips, dQ (another summation order of its
    # keys on top) likewise
    d = (new.float() - old.float()).abs()
    tol = 4 * 2.0 ** -8 * torch.maximum(old.float().abs(), new.float().abs()) + 1e-3
    assert (d <= tol).all(), float((d - tol).max())
    assert float((d[:, D:] > 0).float().mean()) < 0.02        # dK, dV: almost all elements bit-identical
    assert float((d[:, :D] > 0).float().mean()) < 0.10
    want = new.double().sum(0)
    assert (new_cs.double() - want).abs().max().item() <= 1e-5 * max(new.float().abs().sum(0).max().item(), 1.0)
    for _ in range(3):
        again, again_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
        assert torch.equal(again, new) and torch.equal(again_cs, new_cs)
    nb = min(B, 2)
    for sl in (slice(0, nb * T), slice((B - nb) * T, B * T)):
        qr = qkv[sl].float().requires_grad_(True)
        _attn_ref(qr, nb, T, H).backward(dout[sl].float())
        sc = max(qr.grad.abs().max().item(), 1.0)
        assert (new[sl].float() - qr.grad).abs().max().item() <= 3e-2 * sc


def test_attention_backward_t513_workspace_belongs_to_the_call():
    """the fused T = 513 backward parked its dQ^T chains in a function-static, library-owned hipMalloc that every
    stream shared.  The parking space is now the caller's workspace (p4p_attention_bwd_workspace floats, taken per call from
    torch's stream-ordered allocator): two backward calls in flight on two streams, each walking 256 workgroups' worth of pairs,
    give the bits of the same calls run one after the other."""
    from p4p_amd import _lib
    B, T, H, D = 64, 513, 12, 768
    assert _lib.call_int('p4p_attention_bwd_workspace', B, T, H) > B * H * T          # the parking space, not just delta
    assert _lib.call_int('p4p_attention_bwd_workspace', B, 257, H) == B * H * 257
    g = torch.Generator(device=DEV).manual_seed(11)
    ins = []
    for _ in range(2):
        qkv = torch.randn(B * T, 3 * D, device=DEV, generator=g).bfloat16()
        do = torch.randn(B * T, D, device=DEV, generator=g).bfloat16()
        o, lse = ops.attention_fwd(qkv, B, T, H)
        ins.append((qkv, o, do, lse))
    want = [tuple(x.clone() for x in ops.attention_bwd(q, o, do, lse, B, T, H, want_colsum=True)) for q, o, do, lse in ins]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for rep in range(4):                                   # interleaved enqueue: the two streams' kernels overlap on the chip
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                q, o, do, lse = ins[i]
                # the column-sum partials go through ops._workspace: a buffer per (purpose, device, dtype, stream)
                got[i].append(ops.attention_bwd(q, o, do, lse, B, T, H, want_colsum=True))
    torch.cuda.synchronize()
    for i in range(2):
        for rep, (x, cs) in enumerate(got[i]):
            assert torch.equal(x, want[i][0]), f'stream {i}, launch {rep}: a concurrent call disturbed the parked dQ chains'
            assert torch.equal(cs, want[i][1]), f'stream {i}, launch {rep}: a concurrent call disturbed the column-sum partials'


@pytest.mark.parametrize('B,T,H', [(2, 2049, 3), (1, 1300, 2), (3, 513, 4)])
def test_attention_forward_two_and_four_query_tiles_per_wave_are_bit_identical(B, T, H):
    """attn_fwd_q_kernel<2> / <4>: the number of 16-query tiles a wave carries is a scheduling choice (round 4: taken by how many times the
    256-row workgroups fill the chip) -- every query tile runs the same chain over the same key tiles."""
    from p4p_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(T + B)
    qkv = (torch.randn(B * T, 3 * 64 * H, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    try:
        lib.p4p_set_attention_fwd_variant(16)
        o4, l4 = ops.attention_fwd(qkv, B, T, H)
        o4, l4 = o4.clone(), l4.clone()
        lib.p4p_set_attention_fwd_variant(32)
        o2, l2 = ops.attention_fwd(qkv, B, T, H)
        o2, l2 = o2.clone(), l2.clone()
    finally:
        lib.p4p_set_attention_fwd_variant(0)
    od, ld = ops.attention_fwd(qkv, B, T, H)
    assert torch.equal(o2, o4) and torch.equal(l2, l4)
    assert torch.equal(od, o2) and torch.equal(ld, l2)


@pytest.mark.parametrize('B,H', [(1, 1), (2, 12), (23, 12), (50, 6)])
def test_attention_forward_kv_resident_t513(B, H):
    """Round 4, not the default path (p4p_set_attention_fwd_variant(1)): attn_fwd_kv_kernel (T = 513: K and V of the sequence resident in LDS, 32-key tiles, the query on the lane) against its
    rounding-matched statement (tests/emulate.py 'online32': per-element, a correct bf16 rounding up to flipped numerators), against
    fp32 softmax attention, and lse2 against the exact log-sum-exp; the walk (pairs > CUs) bitwise equal to a workgroup per pair;
    the streaming kernel it replaces (another tile order) agrees within bf16 rounding of the output."""
    from p4p_amd import _lib
    from tests import emulate as E
    lib = _lib.lib()
    T, D = 513, 64 * H
    g = torch.Generator().manual_seed(77 + 3 * B + H)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    old, old_lse = ops.attention_fwd(qkv, B, T, H)
    old, old_lse = old.clone(), old_lse.clone()
    try:
        lib.p4p_set_attention_fwd_variant(1)
        _kv_resident_checks(lib, qkv, old, old_lse, B, T, H, D)
    finally:
        lib.p4p_set_attention_fwd_variant(0)


def _kv_resident_checks(lib, qkv, old, old_lse, B, T, H, D):
    from tests import emulate as E
    out, lse = ops.attention_fwd(qkv, B, T, H)
    out, lse = out.clone(), lse.clone()
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    lib.p4p_set_attention_fwd_variant(1 | 2)
    pp, pp_lse = ops.attention_fwd(qkv, B, T, H)
    assert torch.equal(pp, out) and torch.equal(pp_lse, lse)
    lib.p4p_set_attention_fwd_variant(1)
    for _ in range(2):
        again, again_lse = ops.attention_fwd(qkv, B, T, H)
        assert torch.equal(again, out) and torch.equal(again_lse, lse)
    assert (old.float() - out.float()).abs().max().item() <= 2e-2
    assert (old_lse - lse).abs().max().item() <= 2e-2
    nb = min(B, 2)
    with torch.no_grad():
        for b0 in (0, B - nb):
            sl = slice(b0 * T, (b0 + nb) * T)
            q = qkv[sl].float()
            torch.testing.assert_close(out[sl].float(), _attn_ref(q, nb, T, H), rtol=2e-2, atol=2e-2)
            assert E.link_stats(out[sl], E.attention_emulated(q, nb, T, H, round_output=False, mode='online32'), 'att',
                                E.attention_flip_unit(qkv[sl], nb, T, H)) <= E.LINK_LIMITS['att']
            qq, kk, _ = q.view(nb, T, 3, H, 64).permute(2, 0, 3, 1, 4)
            s = (qq.double() @ kk.double().transpose(-1, -2)) * (64 ** -0.5)
            want = torch.logsumexp(s, -1) * 1.4426950408889634          # lse2: base-2 domain
            assert (lse[b0:b0 + nb].double() - want).abs().max().item() <= 5e-3
        # the rescale: one key dominates late in the sequence (and for the odd query: in another wave's share)
        spike = qkv.clone().view(B, T, 3, H, 64)
        spike[:, T - 1, 1] = spike[:, 0, 0] * 4
        spike[:, 300, 1] = spike[:, T - 1, 0] * 4
        spike = spike.view(B * T, 3 * D).contiguous()
        out2, _ = ops.attention_fwd(spike, B, T, H)
        sl = slice(0, nb * T)
        torch.testing.assert_close(out2[sl].float(), _attn_ref(spike[sl].float(), nb, T, H), rtol=2e-2, atol=2e-2)
        assert E.link_stats(out2[sl], E.attention_emulated(spike[sl].float(), nb, T, H, round_output=False, mode='online32'), 'att',
                            E.attention_flip_unit(spike[sl], nb, T, H)) <= E.LINK_LIMITS['att']


@pytest.mark.parametrize('B,H', [(43, 6), (25, 12), (300, 1), (23, 12)])
def test_attention_backward_persistent_walk_equals_a_workgroup_per_pair(B, H):
    """T = 257 with more (batch, head) pairs than CUs: one workgroup per CU WALKS its pairs (the default).  Bit for bit what a
    workgroup per pair computes (variant bit 7), pair counts that are not a multiple of the CU count (some workgroups walk one pair
    more than others; 23 x 12 = 276: a second round of 20), and the usual error against fp32 autograd on a slice."""
    from p4p_amd import _lib
    lib = _lib.lib()
    T, D = 257, 64 * H
    g = torch.Generator().manual_seed(B * 100 + H)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B * T, D, generator=g).to(torch.bfloat16).to(DEV)
    out, lse = ops.attention_fwd(qkv, B, T, H)
    assert B * H > _lib.call_int('p4p_device_cu_count')
    try:
        lib.p4p_set_attention_variant(128)
        per_pair, per_pair_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
        per_pair, per_pair_cs = per_pair.clone(), per_pair_cs.clone()
    finally:
        lib.p4p_set_attention_variant(0)
    walk, walk_cs = ops.attention_bwd(qkv, out, dout, lse, B, T, H, want_colsum=True)
    assert torch.equal(walk, per_pair) and torch.equal(walk_cs, per_pair_cs)
    nb = min(B, 3)                                       # reference on the first and the last sequences (the walk's last round)
    for sl in (slice(0, nb * T), slice((B - nb) * T, B * T)):
        qr = qkv[sl].float().requires_grad_(True)
        _attn_ref(qr, nb, T, H).backward(dout[sl].float())
        sc = max(qr.grad.abs().max().item(), 1.0)
        assert (walk[sl].float() - qr.grad).abs().max().item() <= 3e-2 * sc


@pytest.mark.parametrize('r', [8, 13, 32, -1])
def test_persistent_kernels_on_fewer_cus_are_bit_identical(r):
    """p4p_set_reserved_cus(r): the persistent GEMM tile loop and the walking attention backward kernels launch CUs - r workgroups
    (r rounded up to a multiple of 8; r < 0: a workgroup per tile / pair) so that a concurrent long-lived kernel -- RCCL's all-reduce
    under data parallelism -- does not leave r of them waiting with their whole static share to do.  Same bits as with all CUs:
    every GEMM epilogue at a shape with a ragged last panel (the peel arithmetic follows the grid size) and at the token-stream
    shape, the fused backward at T = 257 and the key-stationary dK / dV at T = 513."""
    from p4p_amd import _lib
    ncu = _lib.call_int('p4p_device_cu_count')

    def run_all():
        outs = []
        for (M, N, K) in [(257 * 256, 768, 768), (300 * 256 - 100, 2304, 768), (257 * 256, 3072, 768)]:
            a = _int_bf16((M, K), -2, 2, 21).to(DEV)
            w = _int_bf16((N, K), -1, 1, 22).to(DEV)
            bias = torch.arange(N, device=DEV, dtype=torch.float32) % 5 - 2
            resid = (torch.arange(M * N, device=DEV) % 11 - 5).float().view(M, N)
            pre = (torch.arange(M * N, device=DEV) % 7 - 3).to(torch.bfloat16).view(M, N)
            outs += [ops.gemm_tn(a, w, ops.EPI_BF16, bias=bias), ops.gemm_tn(a, w, ops.EPI_F32_RESID, bias=bias, resid=resid)]
            outs += list(ops.gemm_tn(a, w, ops.EPI_BF16_GELU, bias=bias, want_pre=True))
            outs += list(ops.gemm_tn(a, w, ops.EPI_BF16_DGELU, pre=pre, want_colsum=True))
            if N == 768:
                assert torch.equal(outs[-6], ((a.float() @ w.float().t()) + bias).to(torch.bfloat16))
        for (B, T, H) in [(43, 257, 12), (30, 513, 12)]:
            g = torch.Generator().manual_seed(B + T)
            qkv = (torch.randn(B * T, 3 * 64 * H, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
            dout = torch.randn(B * T, 64 * H, generator=g).to(torch.bfloat16).to(DEV)
            o, lse = ops.attention_fwd(qkv, B, T, H)
            outs += [x.clone() for x in ops.attention_bwd(qkv, o, dout, lse, B, T, H, want_colsum=True)]
        return outs

    assert ops.persistent_grid_size() == ncu
    base = run_all()
    old = ops.set_reserved_cus(r)
    try:
        assert old == 0
        assert ops.persistent_grid_size() == (0 if r < 0 else ncu - (r + 7) // 8 * 8)
        got = run_all()
    finally:
        ops.set_reserved_cus(0)
    assert len(base) == len(got)
    for j, (x, y) in enumerate(zip(base, got)):
        if r < 0 and j >= len(base) - 2:
            # without persistent kernels T = 513 runs the streaming dQ + key-stationary dK/dV pair instead of the fused kernel
            # (attention_fused2.h walks by construction): the same contract, single bf16 flips apart
            dd = (x.float() - y.float()).abs()
            lim = 4 * 2.0 ** -8 * torch.maximum(x.float().abs(), y.float().abs()) + (1e-3 if x.dtype == torch.bfloat16 else 0.5)
            assert (dd <= lim).all(), (j, float((dd - lim).max()))
        else:
            assert torch.equal(x, y), j


@pytest.mark.parametrize('B,T,runs', [(128, 513, 40), (256, 257, 20), (8, 2049, 10)])
def test_attention_backward_is_bitwise_reproducible_under_load(B, T, runs):
    """A chip-filling launch repeated many times gives the same bits every time.  Regression for a barrier in the streaming
    dK/dV kernel that did not wait for the ds_write of the next tile's lse / -delta: about one launch in ten at this size had
    one 16-key dK tile off by ~1e-3 (timing dependent, so only a loaded chip and many launches show it)."""
    H, D = 12, 768
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = torch.randn(B * T, 3 * D, device=DEV, generator=g).bfloat16()
    do = torch.randn(B * T, D, device=DEV, generator=g).bfloat16()
    o, lse = ops.attention_fwd(qkv, B, T, H)
    ref, ref_cs = ops.attention_bwd(qkv, o, do, lse, B, T, H, want_colsum=True)
    ref, ref_cs = ref.clone(), ref_cs.clone()
    for r in range(runs):
        o2, lse2 = ops.attention_fwd(qkv, B, T, H)
        got, cs = ops.attention_bwd(qkv, o2, do, lse2, B, T, H, want_colsum=True)
        assert torch.equal(o2, o) and torch.equal(lse2, lse), f'forward differs in run {r}'
        assert torch.equal(got, ref), f'dqkv differs in run {r}'
        assert torch.equal(cs, ref_cs), f'column sums differ in run {r}'


# weight-gradient kernels behind p4p_set_wgrad_wide: 2 = gemm_wgrad3.h (8 waves of 128 x 64, the default), 3 = gemm_wgrad4.h (the staggered
# 4-phase schedule), 6 / 10 = gemm_wgrad5.h (4 waves of 128 x 128, hand-interleaved / compiler-scheduled); where K % 256 != 0 all fall back to 2
@pytest.mark.parametrize('variant', [2, 3, 6, 10])
@pytest.mark.parametrize('M,N,K', [(4096, 256, 128), (5000, 768, 768), (4099, 512, 256), (65792, 768, 256), (4160, 256, 256), (8191, 768, 3072)])
def test_gemm_wgrad_ring_kernel_exact_on_integer_data(M, N, K, variant):
    """Shapes that take the ring kernels (N%256==0, K%128==0, M>=4096), incl. ragged M tails (M % 64 = 8, 3, 63) and M % 64 = 0."""
    from p4p_amd import _lib
    g = _int_bf16((M, N), -1, 1, 5).to(DEV)
    a = _int_bf16((M, K), -2, 2, 6).to(DEV)
    if M > 20000:
        g = (g * (torch.rand(M, N, device=DEV) < 0.05)).to(torch.bfloat16).contiguous()   # keep sums exact in fp32
    want = g.float().t() @ a.float()
    old = _lib.call_int('p4p_set_wgrad_wide', variant)
    try:
        dw, db = ops.gemm_wgrad(g, a, want_bias=True)
        assert torch.equal(dw, want)
        assert torch.equal(db, g.float().sum(0))
        acc = ops.gemm_wgrad(g, a, out=dw.clone(), beta=1.0)
        assert torch.equal(acc, 2 * want)
    finally:
        _lib.call_int('p4p_set_wgrad_wide', old)


def test_gemm_wgrad_deferred_slab_sums_are_one_launch_and_bit_identical():
    """Round 5: a ViT block's four weight-gradient GEMMs queue their slab sums (gemm_wgrad(..., defer=queue)) and one
    p4p_sum_slabs_multi launch adds them at the end of the block -- the same additions in the same order as the sum that ran
    behind every GEMM.  Shapes: the four of a ViT-B block at a ragged M, an accumulate (beta = 1) into an existing gradient, and
    a shape the ring kernel does not take (it is computed at once and never enters the queue)."""
    M, D = 4099 + 32 * 7, 768
    g = torch.Generator().manual_seed(77)
    shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D), (64, 128)]
    ops_in = [((torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16).to(DEV), torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV))
              for N, K in shapes]
    base = [torch.randn(N, K, generator=g).to(DEV) for N, K in shapes]
    for beta in (0.0, 1.0):
        want = [ops.gemm_wgrad(G, A, out=b.clone(), beta=beta) for (G, A), b in zip(ops_in, base)]
        queue, got = [], []
        for (G, A), b in zip(ops_in, base):
            got.append(ops.gemm_wgrad(G, A, out=b.clone(), beta=beta, defer=queue))
        assert len(queue) == 4                                  # the small shape did not queue
        assert torch.equal(got[4], want[4])
        if beta == 0.0:
            assert not torch.equal(got[0], want[0])             # nothing has been summed yet
        ops.wgrad_flush(queue)
        assert queue == []
        for x, y, (N, K) in zip(got, want, shapes):
            assert torch.equal(x, y), (N, K, beta)


@pytest.mark.parametrize('variant', [2, 3, 6])
@pytest.mark.parametrize('M,N,K', [(11009, 768, 256), (4099, 3072, 1024), (32771, 512, 128), (16392, 1024, 1024)])
def test_gemm_wgrad_ring_kernel_never_reads_past_a_ragged_end(M, N, K, variant):
    """Regression (round 2): with an ODD number (>= 5) of 32-row steps in the last M-split and M % 32 != 0 the unchecked
    fast staging path of gemm_wgrad3_kernel fetched the ragged last step, i.e. up to 31 rows past the end of both
    operands (first seen as a run-to-run difference of the ViT-L / N=8192 weight gradients, M = 16392).  The operands
    here are the leading rows of larger buffers whose remaining rows are poisoned."""
    gf = torch.full((M + 64, N), 64.0, dtype=torch.bfloat16, device=DEV)
    af = torch.full((M + 64, K), 64.0, dtype=torch.bfloat16, device=DEV)
    gf[:M] = (_int_bf16((M, N), -1, 1, 5).to(DEV) * (torch.rand(M, N, device=DEV) < 0.1)).to(torch.bfloat16)
    af[:M] = _int_bf16((M, K), -2, 2, 6).to(DEV)
    g, a = gf[:M], af[:M]
    assert g.is_contiguous() and a.is_contiguous()
    from p4p_amd import _lib
    old = _lib.call_int('p4p_set_wgrad_wide', variant)
    try:
        dw = ops.gemm_wgrad(g, a)
    finally:
        _lib.call_int('p4p_set_wgrad_wide', old)
    assert torch.equal(dw, g.float().t() @ a.float())


@pytest.mark.parametrize('G,k,C', [(4096, 16, 512), (1000, 16, 512), (777, 32, 256), (300, 3, 512), (64, 16, 520)])
def test_relu_bn_backward_emits_the_group_sums_of_its_own_output(G, k, C):
    """Round 5: the P3Embed backward (autograd of src/models/pix4point.py:149-156,184-188) needs dU = the sum of dH2 over each group's
    k neighbours; the BatchNorm / ReLU backward pass that writes dH2 now emits it (one 1.07-GB read less at the benchmark shape).
    Contract: the bits of group_sum on the finished tensor, whatever the shape (shapes the fused form does not take fall back to it)."""
    g = torch.Generator().manual_seed(G + k + C)
    rows = G * k
    H = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(DEV)
    dY = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    mean = H.float().mean(0)
    invstd = (H.float().var(0, unbiased=False) + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    a = dY.clone()
    sums_a = ops.relu_bn_bwd_(a, H, scale, shift, mean, invstd)
    want = ops.group_sum(a, G, k)
    b = dY.clone()
    sums_b, got = ops.relu_bn_bwd_(b, H, scale, shift, mean, invstd, group=k)
    assert torch.equal(a, b) and torch.equal(sums_a, sums_b)
    assert torch.equal(got, want)
    ref = a.float().view(G, k, C).sum(1)
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6


def test_global_feat_matches_torch():
    """max over patch tokens + cls token (src/models/pix4point.py:260-271) and their gradient, against torch."""
    from p4p_amd import functional as F
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 33, 64, generator=g).to(DEV)
    x[0, 7, 3] = x[0, 9, 3] = 9.0                              # a tie: the first maximum takes the gradient (torch.max)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    got = F.global_feat(xa, ('max', 'cls'))
    want = torch.cat([xb[:, 1:, :].max(dim=1)[0], xb[:, 0, :]], dim=1)
    assert torch.equal(got, want)
    w = torch.randn(5, 128, generator=g).to(DEV)
    (got * w).sum().backward()
    (want * w).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    got2 = F.global_feat(x, ('cls', 'max'))
    assert torch.equal(got2, torch.cat([x[:, 0, :], x[:, 1:, :].max(dim=1)[0]], dim=1))


@pytest.mark.parametrize('B,T,D', [(3, 2, 64), (2, 3, 128), (2, 5, 64), (4, 6, 192), (2, 257, 768), (3, 130, 260), (2, 33, 6)])
def test_global_feat_forward_edge_cases(B, T, D):
    """The vectorised forward (four token slices per column quad, merged in LDS) against torch.max: value AND the index of the FIRST
    maximum, with ties inside a slice and across slices, a column of -inf, a NaN (it wins), and a width that takes the
    scalar kernel (D % 4 != 0)."""
    from p4p_amd import ops as O
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, T, D, generator=g)
    if T > 4:
        x[0, 1, 0] = x[0, 2, 0] = 50.0            # tie across slices: token 1 wins
        x[0, 4, 1] = x[0, 3, 1] = 50.0            # tie, the later slice holds the smaller index
        x[1, 1:, 2] = float('-inf')               # nothing but -inf: the first token
        x[1, 3, 3] = float('nan'); x[1, 2, 3] = 60.0          # a NaN beats everything
    if T > 9:
        x[0, 9, 5] = x[0, 5, 5] = 50.0            # tie inside one slice (tokens 5 and 9)
    x = x.to(DEV)
    mx, arg, cls = O.global_feat_fwd(x, want_max=True, want_cls=True)
    wv, wi = x[:, 1:, :].max(dim=1)
    assert torch.equal(torch.nan_to_num(mx, nan=12345.0), torch.nan_to_num(wv, nan=12345.0))
    assert torch.equal(arg.long(), wi + 1)
    assert torch.equal(cls, x[:, 0, :])


def test_pos_mlp0_matches_torch():
    """GELU(Linear(3,128)) of PointViT.pos_embed and its weight / bias gradients against torch (fp32)."""
    from p4p_amd import functional as F
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(3, 700, 3, generator=g) * 2 - 1).to(DEV)
    lin_a, lin_b = torch.nn.Linear(3, 128).to(DEV), torch.nn.Linear(3, 128).to(DEV)
    lin_b.load_state_dict(lin_a.state_dict())
    got = F.pos_mlp0(x, lin_a.weight, lin_a.bias)
    want = torch.nn.functional.gelu(lin_b(x))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6)
    w = torch.randn(3, 700, 128, generator=g).to(DEV)
    (got * w).sum().backward()
    (want * w).sum().backward()
    torch.testing.assert_close(lin_a.weight.grad, lin_b.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(lin_a.bias.grad, lin_b.bias.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('R,C,col0,cols', [(768, 768, 0, None), (3072, 768, 0, None), (512, 512, 256, 256), (1536, 1536, 768, 768),
                                            (256, 6, 0, None), (40, 256, 0, None), (130, 36, 4, 28), (64, 128, 2, 64)])
def test_cast_weight_and_transpose_are_exact(R, C, col0, cols):
    """fp32 master weight (or a column slice of it) -> bf16 W and W^T: exactly torch's round-to-nearest-even cast, on the
    vectorised 64x64 path (sizes and offsets multiples of 4) and on the general 32x32 one."""
    w = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).to(DEV)
    wb, wt = ops.cast_weight(w, True, True, cols=cols, col0=col0)
    ref = w[:, col0:col0 + (cols if cols is not None else C - col0)].bfloat16()
    assert torch.equal(wb, ref) and torch.equal(wt, ref.t().contiguous())
    only_t = ops.cast_weight(w, False, True, cols=cols, col0=col0)[1]
    assert torch.equal(only_t, ref.t().contiguous())


def test_refresh_weight_operands_is_one_exact_launch_for_all_stale_weights():
    """functional.refresh_weight_operands: the per-step refresh of the encoder's bf16 operand copies (W and W^T of every
    weight an optimizer step has touched) as ONE launch, in place, exactly torch's cast; clean weights are left alone."""
    from p4p_amd import functional as F
    g = torch.Generator().manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in ((2304, 768), (768, 768), (3072, 768), (768, 3072),
                                                                            (192, 64), (260, 132))]
    assert F.refresh_weight_operands(ws, True) == 6                      # first use: nothing cached yet
    for w in ws:
        wb, wt = F._bf16_weights(w, True)
        assert torch.equal(wb, w.detach().bfloat16()) and torch.equal(wt, w.detach().t().contiguous().bfloat16())
    assert F.refresh_weight_operands(ws, True) == 0                      # all current
    ptrs = [F._bf16_weights(w, True)[0].data_ptr() for w in ws]
    with torch.no_grad():
        for w in ws[:4]:
            w.mul_(1.5)                                                  # an "optimizer step" on four of them
    assert F.refresh_weight_operands(ws, True) == 4
    for w, ptr in zip(ws, ptrs):
        wb, wt = F._bf16_weights(w, True)
        assert wb.data_ptr() == ptr                                      # refreshed IN PLACE
        assert torch.equal(wb, w.detach().bfloat16()) and torch.equal(wt, w.detach().t().contiguous().bfloat16())
    # eval first (W only), training later (W^T as well)
    v = [torch.nn.Parameter(torch.randn(128, 64, generator=g).to(DEV)) for _ in range(3)]
    assert F.refresh_weight_operands(v, False) == 3 and F._bf16_weights(v[0], False)[1] is None
    assert F.refresh_weight_operands(v, True) == 3
    assert torch.equal(F._bf16_weights(v[2], True)[1], v[2].detach().t().contiguous().bfloat16())
