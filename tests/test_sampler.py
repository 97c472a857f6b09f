"""The drop-in for the reference's ``src/data/sampler.py`` (``p4p_amd.sampler``): fixtures generated from the reference
(tests/golden/make_golden_sampler.py).  CPU: the oracle's FPS + exact-distance kNN restate the reference's
``fps_sampling_with_knn`` (pins the oracle for this API).  GPU: the HIP path against the same fixture, bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import pix4point_oracle as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sampler.npz'))
CASES = ['b2n1024s256k16', 'b1n2048s196k32', 'b3n300s64k8_dup', 'b2n50s80k4_clamp']


def _rows_sorted(g):
    """(..., k, 3) -> neighbours of every row in lexicographic coordinate order (for rows with exact distance ties)."""
    a = np.asarray(g)
    key = np.lexsort([a[..., 2], a[..., 1], a[..., 0]], axis=-1)
    return np.take_along_axis(a, key[..., None], axis=-2)


def _check(tag, grouped, centroids, flat=None):
    ties = GOLD[f'{tag}_tie_rows']
    assert np.array_equal(np.asarray(centroids), GOLD[f'{tag}_centroids'].astype(np.int64))
    want = GOLD[f'{tag}_grouped']
    got = np.asarray(grouped)
    assert got.shape == want.shape
    assert np.array_equal(got[~ties], want[~ties])                       # same points in the same order
    if ties.any():                                                        # exact ties: torch.topk's order is unspecified
        d_got = ((got[ties] - got[ties][:, :1]) ** 2).sum(-1)            # distances to the centroid (= neighbour 0 or a twin)
        d_want = ((want[ties] - want[ties][:, :1]) ** 2).sum(-1)
        assert np.array_equal(np.sort(d_got, -1), np.sort(d_want, -1))
    if flat is not None:
        assert np.array_equal(np.asarray(flat)[~ties], GOLD[f'{tag}_flat_idx'][~ties].astype(np.int64))


@pytest.mark.parametrize('tag', CASES)
def test_oracle_restates_the_reference_sampler(tag):
    pts = torch.from_numpy(GOLD[f'{tag}_points'])
    B, N, S, k = (int(v) for v in GOLD[f'{tag}_meta'])
    cidx = O.farthest_point_sampling(pts, S, torch.from_numpy(GOLD[f'{tag}_start']))
    assert cidx.shape == (B, min(S, N))
    cen = torch.gather(pts, 1, cidx.unsqueeze(-1).expand(-1, -1, 3))
    idx = O.knn_indices(cen, pts, k).long()
    grouped = pts[torch.arange(B).view(B, 1, 1), idx]
    _check(tag, grouped.numpy(), cidx.numpy(), (idx + torch.arange(B).view(B, 1, 1) * N).numpy())


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
@pytest.mark.parametrize('fn', ['fps_sampling_with_knn', 'fps_sampling_with_knn_optimized', 'fps_sampling_with_knn_topk',
                                'fps_sampling_with_knn_cuda_optimized'])
def test_fps_sampling_with_knn_matches_the_reference_fixture(tag, fn):
    import p4p_amd
    from p4p_amd import sampler
    assert getattr(p4p_amd, fn) is getattr(sampler, fn)                   # the names src/data/__init__.py exports
    pts = torch.from_numpy(GOLD[f'{tag}_points']).cuda()
    B, N, S, k = (int(v) for v in GOLD[f'{tag}_meta'])
    start = torch.from_numpy(GOLD[f'{tag}_start']).cuda()
    grouped, cidx = getattr(sampler, fn)(pts, S, k, return_centroids=True, start_idx=start)
    assert cidx.dtype == torch.int64 and grouped.dtype == torch.float32
    g2, flat = sampler._get_knn_points(pts, cidx, k)
    assert torch.equal(g2, grouped)
    _check(tag, grouped.cpu().numpy(), cidx.cpu().numpy(), flat.cpu().numpy())
    only = getattr(sampler, fn)(pts, S, k, start_idx=start)               # return_centroids=False: the grouped points alone
    assert torch.equal(only, grouped)


@pytest.mark.gpu
def test_unbatched_input_fps_and_loud_failures():
    from p4p_amd import sampler
    pts = torch.from_numpy(GOLD['unbatched_points']).cuda()
    g = sampler.fps_sampling_with_knn(pts, 32, 12, start_idx=torch.from_numpy(GOLD['unbatched_start']).cuda())
    assert np.array_equal(g.cpu().numpy(), GOLD['unbatched_grouped'])     # (N,3) -> (1,S,k,3), sampler.py:214-215
    data = torch.from_numpy(GOLD['fps_data']).cuda()
    f = sampler.fps(data, 100, start_idx=torch.from_numpy(GOLD['fps_start']).cuda())
    assert np.array_equal(f.cpu().numpy(), GOLD['fps_out'])               # all four columns gathered, sampled on xyz
    # without start_idx the first sample is drawn (as in the reference); everything else is determined by it
    g1, c1 = sampler.fps_sampling_with_knn(pts, 16, 4, return_centroids=True)
    g2, c2 = sampler.fps_sampling_with_knn(pts, 16, 4, return_centroids=True, start_idx=c1[:, 0].contiguous())
    assert torch.equal(c1, c2) and torch.equal(g1, g2)
    with pytest.raises(RuntimeError):
        sampler.fps_sampling_with_knn(data, 16, 4)                         # 4 columns: the kernels are xyz kernels
    with pytest.raises(RuntimeError):
        sampler.fps_sampling_with_knn(pts, 16, 500)                        # k > N (torch.topk raises in the reference)
