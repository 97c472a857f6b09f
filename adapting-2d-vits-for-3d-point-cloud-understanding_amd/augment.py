"""Batched, GPU-resident counterpart of the reference's per-cloud numpy augmentation (``src/data/augment.py``, applied one cloud
at a time inside DataLoader workers at ``src/data/scanobjectnn.py:186-212``).  A whole batch (B,N,3) already in HBM goes through
ONE kernel (``csrc/augment.hip``: one workgroup per cloud, the cloud in LDS for the whole chain); the random draws are made here
with a ``torch.Generator`` on the device and handed to the kernel as inputs, so results are reproducible and parity with the
numpy functions is checked draw for draw (``tests/test_gpu_06_augment.py``).

Function names and arguments follow ``src/data/augment.py``; the ``random_*`` functions take the draw they would make as an
explicit argument (``None`` = draw it), clouds are (B,N,3) fp32 device tensors.
"""
import math
from typing import Optional

import torch

from . import ops


def _flags(B, value, device):
    return torch.full((B,), int(value), dtype=torch.int32, device=device)


def _affine(W=None, t=None, B=1, device='cuda'):
    """[B,12] = (W row-major for p' = p W, then t)."""
    if W is None:
        W = torch.eye(3, device=device).expand(B, 3, 3)
    if t is None:
        t = torch.zeros(B, 3, device=device)
    return torch.cat([W.reshape(B, 9), t.reshape(B, 3)], dim=1).float().contiguous()


def rot_x(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([o, z, z, z, c, -s, z, s, c], -1).view(-1, 3, 3)


def rot_y(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([c, z, s, z, o, z, -s, z, c], -1).view(-1, 3, 3)


def rot_z(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([c, -s, z, s, c, z, z, z, o], -1).view(-1, 3, 3)


def _uniform(B, lo, hi, device, generator, shape=()):
    return torch.rand((B, *shape), device=device, generator=generator, dtype=torch.float64) * (hi - lo) + lo


def normalize_point_cloud(points):
    """augment.py:4-18 per cloud."""
    return ops.augment(points.contiguous(), _flags(points.shape[0], ops.AUG_NORMALIZE, points.device))


def _apply_affine(points, W, t=None):
    B = points.shape[0]
    return ops.augment(points.contiguous(), _flags(B, ops.AUG_AFFINE, points.device), affine=_affine(W, t, B, points.device))


def rotate_point_cloud_y(points, angle=None, generator=None):
    """augment.py:76-91 (``points @ R``); angle (B,) in radians."""
    a = _uniform(points.shape[0], 0, 2 * math.pi, points.device, generator) if angle is None else angle.double()
    return _apply_affine(points, rot_y(a))


def rotate_point_cloud_z(points, angle=None, generator=None):
    """augment.py:93-108."""
    a = _uniform(points.shape[0], 0, 2 * math.pi, points.device, generator) if angle is None else angle.double()
    return _apply_affine(points, rot_z(a))


def random_rotation_matrix(B, device, generator=None, angles=None):
    """The W of augment.py:158-191 (``points @ (Rz Ry Rx)^T``): full turn about z, +-15 degrees about y and x.  angles (B,3) =
    (z, y, x) override the draw."""
    if angles is None:
        angles = torch.stack([_uniform(B, 0, 2 * math.pi, device, generator),
                              _uniform(B, -math.pi / 12, math.pi / 12, device, generator),
                              _uniform(B, -math.pi / 12, math.pi / 12, device, generator)], dim=1)
    angles = angles.double()
    R = rot_z(angles[:, 0]) @ rot_y(angles[:, 1]) @ rot_x(angles[:, 2])
    return R.transpose(1, 2)


def random_rotate_point_cloud(points, angles=None, generator=None):
    return _apply_affine(points, random_rotation_matrix(points.shape[0], points.device, generator, angles))


def random_scale_point_cloud(points, scale_low=0.8, scale_high=1.25, scale=None, generator=None):
    """augment.py:35-47; scale (B,)."""
    B = points.shape[0]
    s = _uniform(B, scale_low, scale_high, points.device, generator) if scale is None else scale.double()
    return _apply_affine(points, torch.eye(3, device=points.device, dtype=torch.float64) * s.view(B, 1, 1))


def random_shift_point_cloud(points, shift_range=0.1, shifts=None, generator=None):
    """augment.py:49-60; shifts (B,3)."""
    B = points.shape[0]
    t = _uniform(B, -shift_range, shift_range, points.device, generator, (3,)) if shifts is None else shifts
    return _apply_affine(points, None, t)


def random_jitter_point_cloud(points, sigma=0.01, clip=0.05, noise=None, generator=None):
    """augment.py:62-74; noise (B,N,3) = sigma * randn."""
    if noise is None:
        noise = sigma * torch.randn(points.shape, device=points.device, generator=generator)
    return ops.augment(points.contiguous(), _flags(points.shape[0], ops.AUG_JITTER, points.device),
                       jitter=noise.float().contiguous(), clip=clip)


def _choice_mask(B, N, count, device, generator):
    """`count` distinct indices per cloud (np.random.choice(N, count, replace=False)) as a bool mask."""
    order = torch.rand(B, N, device=device, generator=generator).argsort(dim=1)
    mask = torch.zeros(B, N, dtype=torch.bool, device=device)
    return mask.scatter_(1, order[:, :count], True)


def drop_and_replace_with_noise(point_cloud, drop_ratio=0.05, noise_std=0.02, drop_mask=None, uniform01=None, gauss=None,
                                generator=None):
    """augment.py:110-156; drop_mask (B,N) bool, uniform01 / gauss (B,N,3) (read where the mask is set)."""
    B, N, _ = point_cloud.shape
    dev = point_cloud.device
    if drop_mask is None:
        drop_mask = _choice_mask(B, N, int(N * drop_ratio), dev, generator)
    if uniform01 is None:
        uniform01 = torch.rand(B, N, 3, device=dev, generator=generator)
    if gauss is None:
        gauss = noise_std * torch.randn(B, N, 3, device=dev, generator=generator)
    return ops.augment(point_cloud.contiguous(), _flags(B, ops.AUG_REPLACE, dev), replace_mask=drop_mask.contiguous(),
                       replace_u=uniform01.float().contiguous(), replace_g=gauss.float().contiguous())


def random_point_dropout(points, max_dropout_ratio=0.875, drop_mask=None, generator=None):
    """augment.py:20-33; drop_mask (B,N) bool."""
    B, N, _ = points.shape
    if drop_mask is None:
        ratio = torch.rand(B, 1, device=points.device, generator=generator) * max_dropout_ratio
        drop_mask = torch.rand(B, N, device=points.device, generator=generator) <= ratio
    return ops.augment(points.contiguous(), _flags(B, ops.AUG_DROPOUT, points.device), dropout_mask=drop_mask.contiguous())


def scale_point_cloud(points, scale_range=(0.9, 1.1), anisotropic=True, scale_xyz=(True, True, True), scale=None,
                      generator=None):
    """augment.py:193-226 on the cloud itself; scale (B,3)."""
    B = points.shape[0]
    if scale is None:
        scale = _uniform(B, scale_range[0], scale_range[1], points.device, generator, (3,) if anisotropic else (1,))
        scale = scale.expand(B, 3)
    s = scale.double().clone()
    for i, on in enumerate(scale_xyz):
        if not on:
            s[:, i] = 1.0
    return _apply_affine(points, torch.diag_embed(s))


def center_and_normalize_point_cloud(points, gravity_dim=1):
    """augment.py:228-259 -> (xyz, heights (B,N,1))."""
    h = points[:, :, gravity_dim:gravity_dim + 1]
    return normalize_point_cloud(points), h - h.min(dim=1, keepdim=True).values


def rotate_point_cloud(points, angle=(0.0, 1.0, 0.0), thetas=None, order=None, generator=None):
    """augment.py:261-307: a rotation within +-angle[i] about every axis whose bound is non-zero, multiplied in a random
    order; thetas (B,3), order (B,3) permutations override the draws."""
    B, dev = points.shape[0], points.device
    if thetas is None:
        thetas = torch.stack([_uniform(B, -a, a, dev, generator) if a != 0 else torch.zeros(B, device=dev, dtype=torch.float64)
                              for a in angle], dim=1)
    if order is None:
        order = torch.rand(B, 3, device=dev, generator=generator).argsort(dim=1)
    thetas = thetas.double()
    mats = torch.stack([rot_x(thetas[:, 0]), rot_y(thetas[:, 1]), rot_z(thetas[:, 2])], dim=1)        # (B,3,3,3)
    R = torch.eye(3, device=dev, dtype=torch.float64).expand(B, 3, 3)
    for j in range(3):
        R = R @ mats[torch.arange(B, device=dev), order[:, j]]
    return _apply_affine(points, R.transpose(1, 2))


class TrainAugment:
    """The training-time chain of ``ScanObjectNN.__getitem__`` (src/data/scanobjectnn.py:186-205) for a batch: normalise,
    then random rotation, scaling in [0.8, 1.2], jitter (sigma 0.03, clip 0.05) and drop-and-replace (20 %, std 0.05), each
    applied to a cloud when its own uniform draw exceeds ``augmentation_probability`` - one kernel launch for all of it."""

    def __init__(self, normalize=True, use_custom_augmentation=True, augmentation_probability=0.2, seed=None):
        self.normalize, self.custom, self.p = normalize, use_custom_augmentation, augmentation_probability
        self.seed = seed
        self._gen = None

    def __call__(self, points: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        B, N, _ = points.shape
        dev = points.device
        if generator is None and self.seed is not None:
            if self._gen is None:
                self._gen = torch.Generator(device=dev).manual_seed(self.seed)
            generator = self._gen
        flags = torch.full((B,), ops.AUG_NORMALIZE if self.normalize else 0, dtype=torch.int32, device=dev)
        if not self.custom:
            return ops.augment(points.contiguous(), flags)
        coin = torch.rand(4, B, device=dev, generator=generator) > self.p
        flags = flags + (coin[0] | coin[1]) * ops.AUG_AFFINE + coin[2] * ops.AUG_JITTER + coin[3] * ops.AUG_REPLACE
        W = random_rotation_matrix(B, dev, generator)
        W = torch.where(coin[0].view(B, 1, 1), W, torch.eye(3, device=dev, dtype=W.dtype).expand(B, 3, 3))
        s = torch.where(coin[1], _uniform(B, 0.8, 1.2, dev, generator), torch.ones(B, device=dev, dtype=torch.float64))
        W = W * s.view(B, 1, 1)
        noise = 0.03 * torch.randn(B, N, 3, device=dev, generator=generator)
        mask = _choice_mask(B, N, int(N * 0.2), dev, generator)
        u = torch.rand(B, N, 3, device=dev, generator=generator)
        g = 0.05 * torch.randn(B, N, 3, device=dev, generator=generator)
        return ops.augment(points.contiguous(), flags.to(torch.int32), affine=_affine(W, None, B, dev), jitter=noise, clip=0.05,
                           replace_mask=mask, replace_u=u, replace_g=g)
