"""Drop-in for the reference's ``src/train/apf_trainer.py`` (same constructor, dict-config keys, ``train()``, CSV schema
with the ``learning_rate`` column, checkpoint dict and file names), running the HIP AdaptPointFormer; one process per GPU
with an RCCL all-reduce of the (few) trainable gradients when launched under torch.distributed.

The training step is the reference's (:260-275): zero_grad, forward, CrossEntropy(label_smoothing), backward, AdamW over
``model.parameters()`` - no gradient clipping on this path.  ``CosineWarmupLR`` restates the per-epoch schedule the
reference takes from ``timm.scheduler.CosineLRScheduler`` (:177-183; timm is absent, so this piece is unpinned): linear
warm-up from ``warmup_lr_init`` over ``warmup_t`` epochs, then one cosine cycle to 0 over ``t_initial`` epochs.
"""
import csv
import math
import os
import time

import torch
import torch.nn as nn

from . import augment as GA
from .apf import AdaptPointFormer
from .ddp import GradReducer, broadcast_parameters
from .optim import FusedAdamW
from .trainer import Pix4PointTrainer, SyntheticClouds


class CosineWarmupLR:
    def __init__(self, optimizer, t_initial, warmup_t=0, warmup_lr_init=0.0, lr_min=0.0):
        self.optimizer, self.t_initial, self.warmup_t = optimizer, t_initial, warmup_t
        self.warmup_lr_init, self.lr_min = warmup_lr_init, lr_min
        self.base = [g['lr'] for g in optimizer.param_groups]
        self.step(0) if warmup_t > 0 else None           # timm starts at the warm-up value

    def lr_at(self, t, base):
        if t < self.warmup_t:
            return self.warmup_lr_init + t * (base - self.warmup_lr_init) / self.warmup_t
        if t >= self.t_initial:                          # past the single cycle
            return self.lr_min
        return self.lr_min + 0.5 * (base - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))

    def step(self, epoch):
        for g, b in zip(self.optimizer.param_groups, self.base):
            g['lr'] = self.lr_at(epoch, b)


def save_configs(model_config: dict, dataset_config: dict, train_config: dict, output_dir: str, device: str) -> str:
    """The experiment record the reference's APF trainer leaves in its output directory before anything else
    (src/train/train_utils.py:4-46, called at apf_trainer.py:53): ``experiment_config_<timestamp>.txt`` with one
    ``key: value`` line per entry of the three dicts, then device / directory / timestamp.  Returns the path."""
    stamp = time.strftime('%Y-%m-%d_%H-%M-%S')
    path = os.path.join(output_dir, f'experiment_config_{stamp}.txt')
    sections = (('MODEL', model_config), ('DATASET', dataset_config), ('TRAINING', train_config))
    lines = ['======== EXPERIMENT CONFIGURATION ========', '']
    for title, cfg in sections:
        lines.append(f'== {title} CONFIGURATION ==')
        lines += [f'{k}: {v}' for k, v in cfg.items()]
        lines.append('')
    lines += ['== ENVIRONMENT ==', f'Device: {device}', f'Output directory: {output_dir}', f'Timestamp: {stamp}']
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print(f'Configuration saved to {path}')
    return path


class APFTrainer(Pix4PointTrainer):
    """Trainer for AdaptPointFormer (reference apf_trainer.py:20-345); the epoch loops are the shared ones."""

    def __init__(self, model_config: dict, dataset_config: dict, train_config: dict, device: str = "cpu",
                 output_dir: str = "./output") -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:      # reference :49-53
            os.makedirs(output_dir, exist_ok=True)
            save_configs(model_config, dataset_config, train_config, output_dir, device if device else 'cuda')
        super().__init__(model_config, dataset_config, train_config, device, output_dir)

    # ---- reference :64-82
    def _init_model(self) -> None:
        c = self.model_config
        self.model = AdaptPointFormer(num_classes=c['num_classes'], in_channels=c['in_channels'], vit_name=c['vit_name'],
                                      pretrained=c.get('pretrained', True), embedding_dim=c.get('embedding_dim', 768),
                                      npoint=c.get('npoint', 196), nsample=c.get('nsample', 32),
                                      dropout_rate=c.get('dropout_rate', 0.1),
                                      dropout_path_rate=c.get('drop_path_rate', 0.1),
                                      depth=c.get('depth', 12), num_heads=c.get('num_heads', 12))   # additive keys
        self.model.to(self.device)
        broadcast_parameters(self.model)
        # one bucket (only the head and the point encoder train: quirk 10); the frozen parameters are remembered so that one
        # unfrozen later is refused instead of silently staying local to its rank.  No CUs are reserved for RCCL here: the one
        # all-reduce is launched AND waited for inside finish(), after the backward -- no kernel of this library runs beside it
        self.reducer = (GradReducer([[p for p in self.model.parameters() if p.requires_grad]],
                                    outside=list(self.model.parameters())) if self.distributed else None)

    # ---- reference :84-146
    def _init_datasets(self) -> None:
        c = self.dataset_config
        if c.get('root_dir') != 'synthetic':            # reference :100-141; transforms / height: on the device batch below
            from .scanobjectnn import ScanObjectNN
            common = dict(root_dir=c['root_dir'], variant=c['variant'], augmentation=c['augmentation'],
                          background=c.get('background', False), use_newsplit=c.get('use_newsplit', False), normalize=False,
                          sampling_method=c.get('sampling_method', 'fps'), use_height=c.get('use_height', False),
                          fps_seed=c.get('fps_seed'), fps_device=self.device)
            self.train_dataset = ScanObjectNN(split='training', num_points=c['train_num_points'], **common)
            self.test_dataset = ScanObjectNN(split='test', num_points=c['test_num_points'], **common)
            if self.rank == 0:
                print(f"Dataset loaded: {len(self.train_dataset)} train samples, {len(self.test_dataset)} test samples")
                print(f"Number of classes: {self.train_dataset.num_classes}")
        else:
            n = c.get('num_samples', 2048)
            k = self.model_config['num_classes']
            self.train_dataset = SyntheticClouds(n, c['train_num_points'], k, seed=1234)
            self.test_dataset = SyntheticClouds(max(n // 4, 1), c['test_num_points'], k, seed=4321)
        # the reference's per-cloud numpy transforms (:88-98: scale_point_cloud, center_and_normalize_point_cloud,
        # rotate_point_cloud for training; center_and_normalize_point_cloud for testing) on the device batch
        gen = [None]

        use_height = bool(c.get('use_height', False))
        want_c = self.model_config.get('in_channels', 3)
        if want_c != (4 if use_height else 3):
            raise ValueError(f"model_config['in_channels']={want_c} does not match dataset_config['use_height']={use_height} "
                             '(the reference appends one height column, src/data/scanobjectnn.py:218-224)')

        def with_height(xyz, heights):
            return torch.cat([xyz, heights], dim=-1) if use_height else xyz

        def train_aug(points):
            if gen[0] is None:
                gen[0] = torch.Generator(device=points.device).manual_seed(1234 + self.rank)
            points = GA.scale_point_cloud(points, generator=gen[0])
            xyz, heights = GA.center_and_normalize_point_cloud(points)       # heights of the scaled, un-normalised cloud
            return with_height(GA.rotate_point_cloud(xyz, generator=gen[0]), heights)

        def test_aug(points):
            return with_height(*GA.center_and_normalize_point_cloud(points))
        self.train_augment = train_aug
        self.test_augment = test_aug

    # ---- reference :164-183
    def _init_optimizer(self) -> None:
        t = self.train_config
        self.criterion = nn.CrossEntropyLoss(label_smoothing=t.get('label_smoothing', 0.3))
        self.optimizer = FusedAdamW(self.model.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=t['learning_rate'],
                                    weight_decay=t['weight_decay'])
        self.scheduler = CosineWarmupLR(self.optimizer, t_initial=t['epochs'], warmup_t=t.get('warmup_epochs', 10),
                                        warmup_lr_init=t.get('warmup_lr_init', 1e-3))

    # ---- reference :185-203
    def _init_metrics_csv(self) -> None:
        with open(self.metrics_csv_path, 'w', newline='') as f:
            csv.writer(f).writerow(['epoch', 'train_loss', 'train_accuracy', 'test_loss', 'test_accuracy', 'learning_rate'])

    def _update_metrics_csv(self, epoch, train_loss, train_acc, test_loss, test_acc) -> None:
        with open(self.metrics_csv_path, 'a', newline='') as f:
            csv.writer(f).writerow([epoch + 1, train_loss, train_acc, test_loss, test_acc,
                                    self.optimizer.param_groups[0]['lr']])

    # ---- one optimisation step (reference :260-275); also what bench.py --workload apf times
    def train_step(self, points: torch.Tensor, labels: torch.Tensor, start_idx=None):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        logits = self.model(points, start_idx)
        loss = self.criterion(logits, labels)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish(average=False)            # .grad holds SUMS; 1/world is applied inside the fused update
            self.optimizer.step(grad_scale=self.reducer.grad_scale)
        else:
            self.optimizer.step()
        return loss, logits

    # ---- reference :205-247: the scheduler is stepped with the epoch index after the test phase
    def train(self) -> None:
        best_acc = 0.0
        try:
            for epoch in range(self.train_config['epochs']):
                if self.rank == 0:
                    print(f"\nEpoch {epoch+1}/{self.train_config['epochs']}")
                    print(f"Current learning rate: {self.optimizer.param_groups[0]['lr']:.6f}")
                if self.distributed:
                    self.train_loader.sampler.set_epoch(epoch)
                train_loss, train_acc = self._train_epoch()
                test_loss, test_acc = self._test_epoch()
                if self.rank == 0:
                    self._update_metrics_csv(epoch, train_loss, train_acc, test_loss, test_acc)
                    if test_acc > best_acc:
                        best_acc = test_acc
                        self._save_checkpoint("model_best.pt", epoch, test_acc)
                        print(f"New best model saved with accuracy: {best_acc:.4f}")
                    if (epoch + 1) % self.train_config['save_interval'] == 0:
                        self._save_checkpoint(f"model_epoch_{epoch+1}.pt", epoch, test_acc)
                    print(f"Epoch {epoch+1} results:")
                    print(f"  Train Loss: {train_loss:.4f}, Train Accuracy: {train_acc:.4f}")
                    print(f"  Test Loss: {test_loss:.4f}, Test Accuracy: {test_acc:.4f}")
                self.scheduler.step(epoch)
            if self.rank == 0:
                print(f"\nTraining completed. Best test accuracy: {best_acc:.4f}")
                print(f"Training metrics saved to {self.metrics_csv_path}")
        finally:
            if self.reducer is not None:
                self.reducer.release_cus()

    def _train_epoch(self):
        sched, self.scheduler = self.scheduler, _NoStep()          # the shared loop steps a per-epoch scheduler itself
        try:
            return super()._train_epoch()
        finally:
            self.scheduler = sched


class _NoStep:
    def step(self, *a):
        pass
