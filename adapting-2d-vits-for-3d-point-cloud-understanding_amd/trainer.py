"""Drop-in for the reference's ``src/train/pix4point_trainer.py`` (same constructor, dict-config keys,
``train()`` / ``load_checkpoint()``, CSV schema and checkpoint dict), running the HIP model, plus what the
reference lacks for an 8-GPU node: one process per GPU with RCCL gradient all-reduce (ddp.py).

``dataset_config['root_dir'] == 'synthetic'`` selects ModelNet40-shaped synthetic clouds; any other value is a
ScanObjectNN directory read by p4p_amd.scanobjectnn (the reference's path rules, sampling modes and dataset-level FPS, on
an .h5 or its .npz conversion).  Normalisation / augmentation run on the device batch (augment.py).
"""
import csv
import os
from typing import Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from .augment import TrainAugment
from .ddp import broadcast_parameters, build_reducer
from .optim import FusedAdamW
from .pix4point import Pix4Point


class SyntheticClouds(torch.utils.data.Dataset):
    """(points (N,3) f32 in the unit ball, label int64): the shape ScanObjectNN.__getitem__ returns
    (reference src/data/scanobjectnn.py:148-229) without the h5 files."""

    def __init__(self, num_samples, num_points, num_classes, seed=1234):
        g = torch.Generator().manual_seed(seed)
        p = torch.rand(num_samples, num_points, 3, generator=g) * 2 - 1
        p = p - p.mean(dim=1, keepdim=True)
        self.points = p / p.norm(dim=2).max(dim=1).values.view(-1, 1, 1)
        self.labels = torch.randint(0, num_classes, (num_samples,), generator=g)
        self.num_classes = num_classes

    def __len__(self):
        return self.points.shape[0]

    def __getitem__(self, i):
        return self.points[i], self.labels[i]


class Pix4PointTrainer:
    """Trainer for the Pix4Point model (reference pix4point_trainer.py:16-217)."""

    def __init__(self, model_config: dict, dataset_config: dict, train_config: dict, device: str = "cpu",
                 output_dir: str = "./output_pix4point") -> None:
        self.model_config, self.dataset_config, self.train_config = model_config, dataset_config, train_config
        self.output_dir = output_dir
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        self.device = device if device else "cuda"
        if not str(self.device).startswith('cuda'):
            raise RuntimeError(f"device={device!r}: this trainer runs the HIP path only (no CPU fallback); "
                               "use device='cuda'")
        if self.rank == 0:
            print(f"Using Device: {self.device}")
            os.makedirs(output_dir, exist_ok=True)
        self.metrics_csv_path = os.path.join(output_dir, "training_metrics.csv")
        self._init_model()
        self._init_datasets()
        self._init_loaders()
        self._init_optimizer()
        if self.rank == 0:
            self._init_metrics_csv()

    # ---- reference :44-54
    def _init_model(self) -> None:
        c = self.model_config
        self.model = Pix4Point(num_classes=c['num_classes'], pretrained_model=c['pretrained_model'],
                               pretrained=c['pretrained'], k_neighbors=c['k_neighbors'], embed_dim=c['embed_dim'],
                               frozen=c['frozen'])
        self.model.to(self.device)
        broadcast_parameters(self.model)
        self.reducer = (build_reducer(self.model, bucket_mb=self.train_config.get('ddp_bucket_mb'))
                        if self.distributed else None)      # 'ddp_bucket_mb': additive key (the reference has no DDP)
        if self.rank == 0:
            self.model.print_trainable_params()

    # ---- reference :56-83
    def _init_datasets(self) -> None:
        c = self.dataset_config
        if c.get('root_dir') == 'synthetic':
            n = c.get('num_samples', 2048)
            self.train_dataset = SyntheticClouds(n, c['num_points'], self.model_config['num_classes'], seed=1234)
            self.test_dataset = SyntheticClouds(max(n // 4, 1), c['num_points'], self.model_config['num_classes'], seed=4321)
        else:       # reference :57-79; normalize / custom augmentation happen on the device batch below, not per sample
            from .scanobjectnn import ScanObjectNN
            common = dict(root_dir=c['root_dir'], variant=c['variant'], augmentation=c['augmentation'],
                          num_points=c['num_points'], normalize=False, sampling_method=c.get('sampling_method', 'all'),
                          fps_seed=c.get('fps_seed'), fps_device=self.device)
            if self.rank == 0:
                print("Loading training dataset...")
            self.train_dataset = ScanObjectNN(split='training', use_custom_augmentation=c.get('use_custom_augmentation', False),
                                              **common)
            if self.rank == 0:
                print("Loading test dataset...")
            self.test_dataset = ScanObjectNN(split='test', use_custom_augmentation=False, **common)
            if self.rank == 0:
                print(f"Dataset loaded: {len(self.train_dataset)} train samples, {len(self.test_dataset)} test samples")
        assert self.train_dataset.num_classes == self.model_config['num_classes'], \
            "Number of classes in model config must match dataset classes"
        # the reference normalises / augments one cloud at a time in numpy inside ScanObjectNN.__getitem__
        # (src/data/scanobjectnn.py:186-205); here the batch is augmented on the GPU in one launch (augment.py)
        self.train_augment = TrainAugment(normalize=c.get('normalize', False),
                                          use_custom_augmentation=c.get('use_custom_augmentation', False),
                                          augmentation_probability=c.get('augmentation_probability', 0.2),
                                          seed=1234 + self.rank)
        self.test_augment = TrainAugment(normalize=c.get('normalize', False), use_custom_augmentation=False)

    # ---- reference :85-98 (+ rank sharding)
    def _init_loaders(self) -> None:
        bs = self.train_config['batch_size']
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        ts = DistributedSampler(self.train_dataset, shuffle=True, drop_last=True) if self.distributed else None
        es = DistributedSampler(self.test_dataset, shuffle=False) if self.distributed else None
        self.train_loader = DataLoader(self.train_dataset, batch_size=bs, shuffle=ts is None, sampler=ts, num_workers=0,
                                       drop_last=True, pin_memory=True)
        self.test_loader = DataLoader(self.test_dataset, batch_size=bs, shuffle=False, sampler=es, num_workers=0,
                                      pin_memory=True)

    # ---- reference :100-122
    def _init_optimizer(self) -> None:
        t = self.train_config
        self.criterion = nn.CrossEntropyLoss()
        # same update rule, groups and state_dict layout as the reference's optim.AdamW (:104-106); clip + step are three
        # launches (p4p_amd.optim)
        self.optimizer = FusedAdamW(self.model.get_param_groups(), lr=t['learning_rate'], weight_decay=t['weight_decay'])
        warm = optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=lambda epoch: (epoch + 1) / t['warmup_epochs'])
        cos = optim.lr_scheduler.CosineAnnealingLR(self.optimizer, t['t_max'], eta_min=t['min_lr'])
        self.scheduler = optim.lr_scheduler.SequentialLR(self.optimizer, schedulers=[warm, cos],
                                                         milestones=[t['warmup_epochs']])

    # ---- reference :124-133
    def _init_metrics_csv(self) -> None:
        with open(self.metrics_csv_path, 'w', newline='') as f:
            csv.writer(f).writerow(['epoch', 'train_loss', 'train_accuracy', 'test_loss', 'test_accuracy'])

    def _update_metrics_csv(self, epoch, train_loss, train_acc, test_loss, test_acc) -> None:
        with open(self.metrics_csv_path, 'a', newline='') as f:
            csv.writer(f).writerow([epoch + 1, train_loss, train_acc, test_loss, test_acc])

    # ---- one optimisation step (reference :160-167); also what bench.py times
    def train_step(self, points: torch.Tensor, labels: torch.Tensor, start_idx=None):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        logits = self.model(points, start_idx)
        loss = self.criterion(logits, labels)
        loss.backward()
        fused = isinstance(self.optimizer, FusedAdamW)
        scale = 1.0
        if self.reducer is not None:
            self.reducer.finish(average=not fused)        # fused: .grad holds SUMS, 1/world rides in the clip coefficient
            scale = self.reducer.grad_scale if fused else 1.0
        if fused:
            self.optimizer.step(max_norm=self.train_config['grad_norm_clip'], grad_scale=scale)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.train_config['grad_norm_clip'], foreach=True)
            self.optimizer.step()
        return loss, logits

    # ---- reference :135-151
    def train(self) -> None:
        best_acc = 0.0
        if self.reducer is not None:
            self.reducer.reserve_cus()                  # the plan for the backward windows (build_reducer's, if still there)
        try:
            for epoch in range(getattr(self, 'start_epoch', 0), self.train_config['epochs']):      # > 0 after load_checkpoint
                if self.rank == 0:
                    print(f"\nEpoch {epoch+1}/{self.train_config['epochs']}")
                if self.distributed:
                    self.train_loader.sampler.set_epoch(epoch)
                train_loss, train_acc = self._train_epoch()
                test_loss, test_acc = self._test_epoch()
                if self.rank == 0:
                    self._update_metrics_csv(epoch, train_loss, train_acc, test_loss, test_acc)
                    if test_acc > best_acc:
                        best_acc = test_acc
                        self._save_checkpoint("best_model.pt", epoch, best_acc)
                    if (epoch + 1) % self.train_config['save_interval'] == 0:
                        self._save_checkpoint(f"model_epoch_{epoch+1}.pt", epoch, test_acc)
                    print(f"Epoch {epoch+1} results:")
                    print(f"  Train Loss: {train_loss:.4f}, Train Accuracy: {train_acc:.4f}")
                    print(f"  Test Loss: {test_loss:.4f}, Test Accuracy: {test_acc:.4f}")
            if self.rank == 0:
                print(f"\nTraining completed. Best test accuracy: {best_acc:.4f}")
                print(f"Training metrics saved to {self.metrics_csv_path}")
        finally:
            if self.reducer is not None:
                self.reducer.release_cus()              # whatever runs after training in this process gets all CUs back

    def _reduce_epoch(self, loss_sum, correct, total, batches) -> Tuple[float, float]:
        stats = torch.stack([loss_sum, correct.float(), torch.tensor(float(total), device=loss_sum.device),
                             torch.tensor(float(batches), device=loss_sum.device)])
        if self.distributed:
            dist.all_reduce(stats)
        s = stats.tolist()                      # ONE device->host sync per epoch (the reference syncs twice per step)
        return s[0] / max(s[3], 1.0), s[1] / max(s[2], 1.0)

    # ---- reference :153-176
    def _train_epoch(self) -> Tuple[float, float]:
        self.model.train()
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device, dtype=torch.long)
        total = batches = 0
        for points, labels in self.train_loader:
            points = points.to(self.device, non_blocking=True)
            labels = labels.to(self.device, non_blocking=True)
            points = self.train_augment(points)
            loss, logits = self.train_step(points, labels)
            loss_sum += loss.detach()
            correct += (logits.detach().argmax(1) == labels).sum()
            total += labels.size(0)
            batches += 1
        self.scheduler.step()
        return self._reduce_epoch(loss_sum, correct, total, batches)

    # ---- reference :178-197
    def _test_epoch(self) -> Tuple[float, float]:
        self.model.eval()
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device, dtype=torch.long)
        total = batches = 0
        with torch.no_grad():
            for points, labels in self.test_loader:
                points = points.to(self.device, non_blocking=True)
                labels = labels.to(self.device, non_blocking=True)
                logits = self.model(self.test_augment(points))
                loss_sum += self.criterion(logits, labels)
                correct += (logits.argmax(1) == labels).sum()
                total += labels.size(0)
                batches += 1
        return self._reduce_epoch(loss_sum, correct, total, batches)

    # ---- reference :199-217 (same dict keys)
    def _save_checkpoint(self, filename: str, epoch: int, accuracy: float) -> None:
        """The reference's six keys (what its load_checkpoint reads), plus what a true resume needs and the reference lacks
        (SURVEY.md section 8f rank 4): scheduler state and the RNG streams."""
        ck = {'epoch': epoch, 'model_state_dict': self.model.state_dict(),
              'optimizer_state_dict': self.optimizer.state_dict(), 'accuracy': accuracy,
              'model_config': self.model_config, 'dataset_config': self.dataset_config}
        sched = getattr(self.scheduler, 'state_dict', None)
        ck['resume'] = {'scheduler_state_dict': sched() if sched else None, 'torch_rng_state': torch.get_rng_state(),
                        'cuda_rng_state': torch.cuda.get_rng_state(self.device)}
        torch.save(ck, os.path.join(self.output_dir, filename))

    def load_checkpoint(self, checkpoint_path: str) -> None:
        checkpoint = torch.load(checkpoint_path, map_location=self.device, weights_only=False)
        self.model_config = checkpoint.get('model_config', self.model_config)
        self.dataset_config = checkpoint.get('dataset_config', self.dataset_config)
        self._init_model()
        self.model.load_state_dict(checkpoint['model_state_dict'])
        from .functional import invalidate_operand_caches
        invalidate_operand_caches(self.model)          # belt and braces: copy_ bumps the version counters anyway
        self._init_optimizer()
        self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        resume = checkpoint.get('resume')              # absent in checkpoints written by the reference
        if resume:
            if resume.get('scheduler_state_dict') is not None and hasattr(self.scheduler, 'load_state_dict'):
                self.scheduler.load_state_dict(resume['scheduler_state_dict'])
            torch.set_rng_state(resume['torch_rng_state'].cpu())
            torch.cuda.set_rng_state(resume['cuda_rng_state'].cpu(), self.device)
        self.start_epoch = checkpoint['epoch'] + 1
        print(f"Loaded checkpoint from epoch {checkpoint['epoch']} with accuracy {checkpoint['accuracy']:.4f}")
