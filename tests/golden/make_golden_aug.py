"""Generates ``augment.npz`` from the REFERENCE's ``src/data/augment.py`` (numpy + scipy only: it loads as it is).

    python tests/golden/make_golden_aug.py

``np.random``'s functions are wrapped for the duration of each call so that every draw the reference makes is recorded next to
the output.  Fixtures are DATA: input cloud, the draws, the function's output.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src/data/augment.py'


class record_draws:
    NAMES = ('random', 'uniform', 'randn', 'normal', 'choice', 'permutation')

    def __enter__(self):
        self.log = []
        self.orig = {n: getattr(np.random, n) for n in self.NAMES}
        for n in self.NAMES:
            def wrap(*a, _n=n, **k):
                v = self.orig[_n](*a, **k)
                self.log.append((_n, np.array(v)))
                return v
            setattr(np.random, n, wrap)
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(np.random, n, f)


def main():
    spec = importlib.util.spec_from_file_location('ref_augment', REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    np.random.seed(7)
    N = 300
    pts = (np.random.rand(N, 3).astype(np.float32) * 2 - 1) * np.array([1.0, 0.6, 0.3], dtype=np.float32) + 0.2
    out = {'points': pts}

    out['normalize'] = ref.normalize_point_cloud(pts.copy())
    for name, fn, kw in [('rotate_y', ref.rotate_point_cloud_y, {}), ('rotate_z', ref.rotate_point_cloud_z, {}),
                         ('random_rotate', ref.random_rotate_point_cloud, {}),
                         ('random_scale', ref.random_scale_point_cloud, dict(scale_low=0.8, scale_high=1.2)),
                         ('random_shift', ref.random_shift_point_cloud, {}),
                         ('jitter', ref.random_jitter_point_cloud, dict(sigma=0.03, clip=0.05)),
                         ('drop_replace', ref.drop_and_replace_with_noise, dict(drop_ratio=0.2, noise_std=0.05)),
                         ('point_dropout', ref.random_point_dropout, {})]:
        with record_draws() as r:
            res = fn(pts.copy(), **kw)
        out[name] = np.asarray(res)
        for j, (n, v) in enumerate(r.log):
            out[f'{name}_draw{j}_{n}'] = v
    # the dict-style transforms of the APF trainer (src/train/apf_trainer.py:88-98)
    with record_draws() as r:
        d = ref.scale_point_cloud({'xyz': pts.copy()})
    out['scale_point_cloud'] = d['xyz']
    out['scale_point_cloud_draw0_uniform'] = r.log[0][1]
    d = ref.center_and_normalize_point_cloud({'xyz': pts.copy()})
    out['center_and_normalize'], out['center_and_normalize_heights'] = d['xyz'], d['heights']
    with record_draws() as r:
        d = ref.rotate_point_cloud({'xyz': pts.copy()}, angle=(0.3, 1.0, 0.2))
    out['rotate_point_cloud'] = d['xyz']
    out['rotate_point_cloud_thetas'] = np.array([v for n, v in r.log if n == 'uniform']).reshape(3)
    out['rotate_point_cloud_order'] = [v for n, v in r.log if n == 'permutation'][0]
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **out)
    print('written augment.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
