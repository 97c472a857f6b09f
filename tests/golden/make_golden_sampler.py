"""Generates ``sampler.npz`` from the REFERENCE's ``src/data/sampler.py``: ``fps_sampling_with_knn`` (and the agreement of
its ``_optimized`` / ``_topk`` siblings on the same inputs), ``fps`` and the flattened neighbour indices of
``_get_knn_points``.

    python tests/golden/make_golden_sampler.py        (build container only: needs /root/reference)

The reference draws the FPS start with an unseeded ``torch.randint`` (sampler.py:120); ``fixed_randint`` makes that draw
return the recorded ``start``.  ``min_gap`` flags neighbour lists whose k-th and (k+1)-th distances (or two inside the
list) are closer than fp32 can separate reliably: torch.topk leaves exact ties unspecified, so those rows are compared as
sets.  Fixtures are DATA (inputs + expected outputs); no reference source travels.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import pix4point_oracle as O          # noqa: E402
from make_golden import fixed_randint             # noqa: E402
from make_golden_apf import load_reference        # noqa: E402


def exact_tie_rows(points, cidx, k):
    """bool (B,S): the exact distances of the k+1 nearest points hold a repeated value (order unspecified in topk)."""
    B = points.shape[0]
    cen = points[torch.arange(B).view(B, 1), cidx]
    d = ((points.unsqueeze(1) - cen.unsqueeze(2)) ** 2).sum(-1)
    kk = min(k + 1, points.shape[1])
    v = d.topk(kk, dim=-1, largest=False, sorted=True).values
    return (v[..., 1:] == v[..., :-1]).any(-1)


def main():
    sampler, _, _ = load_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    cases = {'b2n1024s256k16': (2, 1024, 256, 16, 0.0), 'b1n2048s196k32': (1, 2048, 196, 32, 0.0),
             'b3n300s64k8_dup': (3, 300, 64, 8, 0.1), 'b2n50s80k4_clamp': (2, 50, 80, 4, 0.0)}
    for tag, (B, N, S, k, dup) in cases.items():
        pts = O.synthetic_clouds(B, N, seed=700 + N, dup_frac=dup)
        start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(11 + N))
        res = {}
        for name in ('fps_sampling_with_knn', 'fps_sampling_with_knn_optimized', 'fps_sampling_with_knn_topk'):
            with fixed_randint(start):
                res[name] = getattr(sampler, name)(pts, S, k, return_centroids=True)
        g0, c0 = res['fps_sampling_with_knn']
        ties = exact_tie_rows(pts, c0, k)
        for name, (g, c) in res.items():          # the siblings compute the same function
            assert torch.equal(c, c0), (tag, name)
            assert torch.equal(g[~ties], g0[~ties]), (tag, name)
        _, flat = sampler._get_knn_points(pts, c0, k)
        out[f'{tag}_points'] = pts.numpy()
        out[f'{tag}_start'] = start.numpy()
        out[f'{tag}_centroids'] = c0.numpy().astype(np.int32)
        out[f'{tag}_grouped'] = g0.numpy()
        out[f'{tag}_flat_idx'] = flat.numpy().astype(np.int32)
        out[f'{tag}_tie_rows'] = ties.numpy()
        out[f'{tag}_meta'] = np.array([B, N, S, k], dtype=np.int64)
    # unbatched input (N,3) -> batch of one (sampler.py:214-215)
    pts = O.synthetic_clouds(1, 400, seed=77)[0]
    start = torch.tensor([5])
    with fixed_randint(start):
        g = sampler.fps_sampling_with_knn(pts, 32, 12)
    out['unbatched_points'], out['unbatched_start'], out['unbatched_grouped'] = pts.numpy(), start.numpy(), g.numpy()
    # fps(data, number) on 4-column data (sampler.py:33-45: sampled on the first three columns, all columns gathered)
    data = torch.cat([O.synthetic_clouds(2, 512, seed=88), torch.rand(2, 512, 1, generator=torch.Generator().manual_seed(3))], -1)
    start = torch.tensor([3, 400])
    with fixed_randint(start):
        f = sampler.fps(data, 100)
    out['fps_data'], out['fps_start'], out['fps_out'] = data.numpy(), start.numpy(), f.numpy()
    np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)
    print('wrote sampler.npz:', {k: v.shape for k, v in out.items() if k.endswith('_grouped')},
          'tie rows:', {k: int(v.sum()) for k, v in out.items() if k.endswith('_tie_rows')})


if __name__ == '__main__':
    main()
