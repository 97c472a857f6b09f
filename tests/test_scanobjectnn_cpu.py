"""CPU: the ScanObjectNN drop-in (p4p_amd.scanobjectnn) against the reference's semantics (src/data/scanobjectnn.py):
file-name rules (:113-130), npz container, sampling modes (:169-185), normalisation, class count; and no CPU FPS."""
import os

import numpy as np
import pytest
import torch

from p4p_amd.scanobjectnn import ScanObjectNN, dataset_file, normalize_point_cloud


def _write(root, sub, stem, M=12, P=40, classes=3, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, sub), exist_ok=True)
    data = rng.normal(size=(M, P, 3)).astype(np.float32)
    label = (np.arange(M) % classes).astype(np.int64).reshape(M, 1)         # the h5 files store (M,1) labels
    np.savez(os.path.join(root, sub, stem + '.npz'), data=data, label=label)
    return data, label.squeeze()


def test_file_name_rules_follow_the_reference():
    f = lambda **kw: os.path.relpath(dataset_file('/r', **kw), '/r')
    base = dict(split='training', variant='main_split', augmentation='base', background=True, use_newsplit=False)
    assert f(**base) == 'main_split/training_objectdataset'
    assert f(**dict(base, background=False)) == 'main_split_nobg/training_objectdataset'
    assert f(**dict(base, augmentation='augmentedrot_scale75')) == 'main_split/training_objectdataset_augmentedrot_scale75'
    assert f(**dict(base, split='test', augmentation='augmentedrot_scale75', use_newsplit=True)) == \
        'main_split/test_objectdataset_augmentedrot_scale75_newsplit'
    assert f(**dict(base, split='training', augmentation='augmentedrot_scale75', use_newsplit=True)) == \
        'main_split/training_objectdataset_augmentedrot_scale75'
    assert f(**dict(base, variant='split2', augmentation='augmented25rot')) == 'split2/training_objectdataset_augmented25rot'


def test_loading_sampling_modes_and_normalisation(tmp_path):
    data, label = _write(str(tmp_path), 'main_split_nobg', 'training_objectdataset')
    ds = ScanObjectNN(str(tmp_path), split='training', background=False)                       # sampling 'all'
    assert len(ds) == 12 and ds.num_classes == 3 and ds.data.dtype == np.float32 and ds.labels.shape == (12,)
    p, l = ds[5]
    assert p.dtype == torch.float32 and l.dtype == torch.int64 and l.dim() == 0 and int(l) == label[5]
    assert np.array_equal(p.numpy(), data[5])
    first = ScanObjectNN(str(tmp_path), background=False, num_points=16, sampling_method='first')
    assert np.array_equal(first[2][0].numpy(), data[2][:16])                                     # :174-175
    np.random.seed(0)
    padded = ScanObjectNN(str(tmp_path), background=False, num_points=64, sampling_method='first')[2][0].numpy()
    assert padded.shape == (64, 3) and {tuple(r) for r in padded} <= {tuple(r) for r in data[2]}   # :171-173: with replacement
    rnd = ScanObjectNN(str(tmp_path), background=False, num_points=16, sampling_method='random')[2][0].numpy()
    rows = {tuple(r) for r in rnd}
    assert len(rows) == 16 and rows <= {tuple(r) for r in data[2]}                               # :181-183: without replacement
    nrm = ScanObjectNN(str(tmp_path), background=False, normalize=True)[3][0].numpy()
    assert np.allclose(nrm, normalize_point_cloud(data[3])) and abs(np.linalg.norm(nrm, axis=1).max() - 1) < 1e-6
    assert np.allclose(nrm.mean(0), 0, atol=1e-6)


def test_missing_file_and_bad_arguments(tmp_path):
    with pytest.raises(FileNotFoundError, match='Dataset file not found'):
        ScanObjectNN(str(tmp_path), split='test')
    _write(str(tmp_path), 'main_split', 'test_objectdataset')
    with pytest.raises(ValueError, match='sampling_method'):
        ScanObjectNN(str(tmp_path), split='test', sampling_method='grid')
    # dataset-level FPS runs through the HIP kernel only: there is no CPU path to fall back to
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ScanObjectNN(str(tmp_path), split='test', num_points=8, sampling_method='fps', fps_device='cpu')
