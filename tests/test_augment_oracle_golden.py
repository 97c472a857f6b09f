"""CPU: the augmentation oracle against the fixtures the REFERENCE's numpy functions produced (tests/golden/make_golden_aug.py)."""
import os

import numpy as np
import pytest

from oracle import augment_oracle as G


@pytest.fixture(scope='module')
def z(golden_dir):
    return np.load(os.path.join(golden_dir, 'augment.npz'))


def test_every_function_reproduces_the_reference_given_its_draws(z):
    p = z['points']
    eq = lambda a, b: np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 if b.dtype == np.float64 else 1e-7)
    eq(G.normalize_point_cloud(p), z['normalize'])
    eq(G.rotate_point_cloud_y(p, z['rotate_y_draw0_uniform']), z['rotate_y'])
    eq(G.rotate_point_cloud_z(p, z['rotate_z_draw0_uniform']), z['rotate_z'])
    eq(G.random_rotate_point_cloud(p, z['random_rotate_draw0_uniform'], z['random_rotate_draw1_uniform'],
                                   z['random_rotate_draw2_uniform']), z['random_rotate'])
    eq(G.random_scale_point_cloud(p, z['random_scale_draw0_uniform']), z['random_scale'])
    eq(G.random_shift_point_cloud(p, z['random_shift_draw0_uniform']), z['random_shift'])
    eq(G.random_jitter_point_cloud(p, 0.03 * z['jitter_draw0_randn'], 0.05), z['jitter'])
    lo, hi = p.min(0), p.max(0)
    u01 = (z['drop_replace_draw1_uniform'] - lo) / (hi - lo)
    np.testing.assert_allclose(G.drop_and_replace_with_noise(p, z['drop_replace_draw0_choice'], u01, z['drop_replace_draw2_normal']),
                               z['drop_replace'], atol=1e-6)
    mask = z['point_dropout_draw1_random'] <= z['point_dropout_draw0_random'] * 0.875
    eq(G.random_point_dropout(p, mask), z['point_dropout'])
    eq(G.scale_point_cloud(p, z['scale_point_cloud_draw0_uniform']), z['scale_point_cloud'])
    xyz, h = G.center_and_normalize_point_cloud(p)
    eq(xyz, z['center_and_normalize'])
    eq(h, z['center_and_normalize_heights'])
    np.testing.assert_allclose(G.rotate_point_cloud(p, z['rotate_point_cloud_thetas'], z['rotate_point_cloud_order']),
                               z['rotate_point_cloud'], atol=1e-12)
