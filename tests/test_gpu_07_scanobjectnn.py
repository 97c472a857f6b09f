"""GPU: the ScanObjectNN drop-in's dataset-level FPS (reference src/data/scanobjectnn.py:92-97) through p4p_fps_f32, bit-exact
against the oracle's FPS with the same start indices, and both trainers reading a ScanObjectNN-shaped directory (npz
container) with the reference's dict configs."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import p4p_amd                                         # noqa: E402
from p4p_amd.scanobjectnn import ScanObjectNN          # noqa: E402
from oracle import pix4point_oracle as O               # noqa: E402


def _write(root, sub, stem, M, P, classes, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, sub), exist_ok=True)
    centres = rng.normal(size=(classes, 1, 3)).astype(np.float32)
    label = (np.arange(M) % classes).astype(np.int64)
    data = (rng.normal(size=(M, P, 3)) * 0.3 + centres[label]).astype(np.float32)        # class = blob position: learnable
    np.savez(os.path.join(root, sub, stem + '.npz'), data=data, label=label.reshape(M, 1))
    return data, label


def test_dataset_level_fps_is_the_reference_algorithm_bit_for_bit(tmp_path):
    data, _ = _write(str(tmp_path), 'main_split', 'training_objectdataset', M=37, P=300, classes=5, seed=1)
    ds = ScanObjectNN(str(tmp_path), num_points=64, sampling_method='fps', fps_seed=7, fps_chunk=16)
    assert ds.data.shape == (37, 64, 3)
    g = torch.Generator().manual_seed(7)
    want = []
    for a in range(0, 37, 16):                                                       # same chunking => same start draws
        pts = torch.from_numpy(data[a:a + 16])
        start = torch.randint(0, 300, (pts.shape[0],), generator=g)
        idx = O.farthest_point_sampling(pts, 64, start)                              # = sampler.py:4-30 (tests/test_oracle_golden.py)
        want.append(torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).numpy())
    assert np.array_equal(ds.data, np.concatenate(want))
    p, l = ds[3]
    assert p.shape == (64, 3) and int(l) == 3


def test_pix4point_trainer_reads_a_scanobjectnn_directory(tmp_path):
    root = str(tmp_path / 'h5_files')
    _write(root, 'main_split', 'training_objectdataset_augmentedrot_scale75', M=64, P=400, classes=4, seed=2)
    _write(root, 'main_split', 'test_objectdataset_augmentedrot_scale75', M=16, P=400, classes=4, seed=3)
    mc = {'model_name': 'Pix4Point', 'num_classes': 4, 'pretrained_model': 'vit_custom_d192_l2_h3', 'pretrained': False,
          'frozen': False, 'k_neighbors': 16, 'embed_dim': 192}
    dc = {'root_dir': root, 'split': 'training', 'variant': 'main_split', 'augmentation': 'augmentedrot_scale75',
          'num_points': 256, 'normalize': True, 'sampling_method': 'fps', 'fps_seed': 0}          # train_pix4point.py's keys
    tc = {'batch_size': 16, 'learning_rate': 2e-3, 'weight_decay': 5e-2, 'save_interval': 10, 'epochs': 4, 't_max': 10,
          'warmup_epochs': 1, 'min_lr': 1e-6, 'grad_norm_clip': 10}
    torch.manual_seed(0)
    tr = p4p_amd.Pix4PointTrainer(mc, dc, tc, device='cuda', output_dir=str(tmp_path / 'run'))
    assert len(tr.train_dataset) == 64 and len(tr.test_dataset) == 16 and tr.train_dataset.data.shape == (64, 256, 3)
    tr.train()
    rows = list(csv.reader(open(os.path.join(str(tmp_path / 'run'), 'training_metrics.csv'))))
    losses = [float(r[1]) for r in rows[1:]]
    assert len(losses) == 4 and losses[-1] < losses[0], losses
    with pytest.raises(AssertionError, match='Number of classes'):
        p4p_amd.Pix4PointTrainer(dict(mc, num_classes=5), dc, tc, device='cuda', output_dir=str(tmp_path / 'bad'))


def test_apf_trainer_reads_a_scanobjectnn_directory_with_height(tmp_path):
    root = str(tmp_path / 'h5_files')
    _write(root, 'main_split_nobg', 'training_objectdataset', M=32, P=300, classes=4, seed=4)
    _write(root, 'main_split_nobg', 'test_objectdataset', M=16, P=300, classes=4, seed=5)
    mc = {'num_classes': 4, 'in_channels': 4, 'vit_name': 'vit_custom_d192_l2_h3', 'pretrained': False, 'embedding_dim': 192,
          'npoint': 32, 'nsample': 16, 'dropout_rate': 0.1, 'dropout_path_rate': 0.1, 'depth': 2, 'num_heads': 3}
    dc = {'root_dir': root, 'variant': 'main_split', 'augmentation': 'base', 'background': False, 'use_newsplit': False,
          'train_num_points': 256, 'test_num_points': 128, 'sampling_method': 'fps', 'use_apf_augmentation': True,
          'use_custom_augmentation': False, 'augmentation_probability': 0.0, 'use_height': True, 'fps_seed': 1}   # train_apf.py's keys
    tc = {'batch_size': 16, 'save_interval': 100, 'epochs': 2, 'label_smoothing': 0.3, 'learning_rate': 2e-3,
          'weight_decay': 5e-2, 'warmup_epochs': 1, 'warmup_lr_init': 1e-3}
    torch.manual_seed(0)
    tr = p4p_amd.APFTrainer(mc, dc, tc, device='cuda', output_dir=str(tmp_path / 'apf'))
    assert tr.train_dataset.data.shape == (32, 256, 3) and tr.test_dataset.data.shape == (16, 128, 3)
    tr.train()
    rows = list(csv.reader(open(os.path.join(str(tmp_path / 'apf'), 'training_metrics.csv'))))
    assert len(rows) == 3 and all(float(r[1]) == float(r[1]) for r in rows[1:])
