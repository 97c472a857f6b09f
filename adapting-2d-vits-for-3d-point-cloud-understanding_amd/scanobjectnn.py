"""Drop-in for the reference's ``src/data/scanobjectnn.py``: same class name, constructor arguments, attributes
(``data``, ``labels``, ``num_classes``), file-name rules and sampling modes.

Differences, stated:

* container: the reference reads ``<root>/<variant>[_nobg]/<split>_objectdataset[_<aug>].h5`` with h5py (:99-138).  h5py is not
  in this image, so the same path is tried with ``.h5`` when h5py can be imported and with ``.npz`` (arrays ``data`` (M,P,3)
  float32 and ``label`` (M,)) otherwise -- ``numpy.savez(path, data=f['data'][:], label=f['label'][:])`` converts a file;
* dataset-level FPS (``sampling_method='fps'``, :92-97: ``fps(points.cuda(), num_points)`` over the WHOLE array in one
  Python-loop call) runs through the HIP kernel (``p4p_fps_f32``, bit-identical algorithm: tests/test_gpu_05_apf.py) in chunks of
  clouds; the reference draws the start indices unseeded (sampler.py:20), here they come from ``fps_seed`` (None = unseeded; in a multi-rank run one seed drawn by rank 0 is shared, so all ranks subsample alike);
* the per-sample numpy augmentations of ``__getitem__`` (:186-212) and the dict transforms of the APF trainer are applied to
  the DEVICE BATCH by the trainers (p4p_amd.augment: one kernel launch per batch, draw-for-draw parity with the reference's
  functions in tests/test_gpu_06_augment.py), so ``__getitem__`` returns the sampled (and, if asked, normalised) cloud;
  ``use_custom_augmentation`` / ``augmentation_probability`` / ``transform`` / ``use_height`` are kept as attributes for them.
"""
import os
from typing import Callable, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


def normalize_point_cloud(points: np.ndarray) -> np.ndarray:
    """src/data/augment.py:4-18: centre on the mean, scale the farthest point to the unit sphere."""
    points = points - np.mean(points, axis=0)
    m = np.max(np.sqrt(np.sum(points ** 2, axis=1)))
    return points / m if m > 0 else points


def dataset_file(root_dir, split, variant, augmentation, background, use_newsplit):
    """The reference's path rule (:113-130) without the extension."""
    dir_name = variant + ('' if background else '_nobg')
    if augmentation == 'base':
        stem = f'{split}_objectdataset'
    elif augmentation == 'augmentedrot_scale75' and use_newsplit and split == 'test':
        stem = f'{split}_objectdataset_{augmentation}_newsplit'
    else:
        stem = f'{split}_objectdataset_{augmentation}'
    return os.path.join(root_dir, dir_name, stem)


class ScanObjectNN(Dataset):
    """Reference src/data/scanobjectnn.py:17-229."""

    def __init__(self, root_dir: str, split: str = 'training', variant: str = 'main_split', augmentation: str = 'base',
                 background: bool = True, num_points: Optional[int] = None, normalize: bool = False,
                 use_newsplit: bool = False, use_custom_augmentation: bool = False, augmentation_probability: float = 0.2,
                 sampling_method: str = 'all', transform: Optional[Callable] = None, use_height: bool = False,
                 fps_seed: Optional[int] = None, fps_device: str = 'cuda', fps_chunk: int = 256) -> None:
        self.root_dir, self.split, self.variant, self.augmentation = root_dir, split, variant, augmentation
        self.background, self.num_points, self.normalize, self.use_newsplit = background, num_points, normalize, use_newsplit
        self.use_custom_augmentation, self.sampling_method = use_custom_augmentation, sampling_method
        self.augmentation_probability, self.transform, self.use_height = augmentation_probability, transform, use_height
        if sampling_method not in ('all', 'first', 'random', 'fps'):
            raise ValueError(f"sampling_method={sampling_method!r}: expected 'all', 'first', 'random' or 'fps'")
        self.data, self.labels = self._load_data()
        self.num_classes = len(np.unique(self.labels))
        if self.num_points is not None and self.sampling_method == 'fps':
            print("Applying FPS sampling to point clouds...")
            self.data = self._fps_all(fps_seed, fps_device, fps_chunk)
            print(f"Sampled {self.num_points} points per point cloud.")

    # ---- reference :99-138
    def _load_data(self):
        stem = dataset_file(self.root_dir, self.split, self.variant, self.augmentation, self.background, self.use_newsplit)
        if os.path.exists(stem + '.h5'):
            try:
                import h5py
            except ImportError as e:
                if not os.path.exists(stem + '.npz'):
                    raise RuntimeError(f'{stem}.h5 needs h5py, which is not installed; convert it once with '
                                       "numpy.savez(stem + '.npz', data=f['data'][:], label=f['label'][:])") from e
            else:
                with h5py.File(stem + '.h5', 'r') as f:
                    return f['data'][:].astype(np.float32), f['label'][:].astype(np.int64).squeeze()
        if not os.path.exists(stem + '.npz'):
            raise FileNotFoundError(f"Dataset file not found: {stem}.h5 (or its .npz conversion)")
        with np.load(stem + '.npz') as z:
            return z['data'].astype(np.float32), z['label'].astype(np.int64).squeeze()

    # ---- reference :92-97 through the HIP kernel
    def _fps_all(self, seed, device, chunk):
        from . import ops
        M, P, C = self.data.shape
        n = min(self.num_points, P)
        if seed is None:
            # the reference is one process and draws unseeded; with several ranks every rank builds this dataset, and
            # per-rank draws would evaluate the test split on DIFFERENT point subsets: all ranks take rank 0's draw
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                box = [int(torch.randint(0, 2 ** 31 - 1, (1,))) if dist.get_rank() == 0 else None]
                dist.broadcast_object_list(box, src=0)
                seed = box[0]
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        out = np.empty((M, n, C), dtype=np.float32)
        for a in range(0, M, chunk):
            pts = torch.from_numpy(self.data[a:a + chunk]).to(device)
            start = torch.randint(0, P, (pts.shape[0],), generator=gen).to(device)
            idx = ops.fps(pts[:, :, :3].contiguous(), n, start)
            out[a:a + chunk] = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, C)).cpu().numpy()
        return out

    def __len__(self):
        return len(self.labels)

    # ---- reference :148-229 (sampling and normalisation; augmentation / transforms / height: see the module docstring)
    def __getitem__(self, idx):
        points, label = self.data[idx], self.labels[idx]
        if self.sampling_method == 'all' or self.num_points is None:
            pass
        elif self.sampling_method == 'first':
            if points.shape[0] < self.num_points:
                points = points[np.random.choice(points.shape[0], self.num_points, replace=True)]
            elif points.shape[0] > self.num_points:
                points = points[:self.num_points]
        elif self.sampling_method == 'random':
            points = points[np.random.choice(points.shape[0], self.num_points, replace=points.shape[0] < self.num_points)]
        if self.normalize:
            points = normalize_point_cloud(points)
        return torch.from_numpy(np.ascontiguousarray(points)).float(), torch.LongTensor([label]).squeeze()
