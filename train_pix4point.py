"""Entry script with the reference's dict configs (reference train_pix4point.py:1-44), running the
MI355X-native trainer.  Single GPU: ``python train_pix4point.py``; 8 GPUs of one node:
``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_pix4point.py``.
"""
import os

import torch
import torch.distributed as dist

from p4p_amd import Pix4PointTrainer

model_config = {
    'model_name': 'Pix4Point',
    'num_classes': 40,
    'pretrained_model': 'vit_base_patch16_384.augreg_in21k_ft_in1k',
    'pretrained': False,           # True needs a local state dict in $P4P_PRETRAINED_DIR (no downloads)
    'frozen': False,
    'k_neighbors': 16,
    'embed_dim': 768,
}

dataset_config = {
    'root_dir': 'synthetic',       # or a ScanObjectNN directory (.h5 / .npz, p4p_amd.scanobjectnn: same file-name rules as the reference)
    'split': 'training',
    'variant': 'main_split',
    'augmentation': 'augmentedrot_scale75',
    'num_points': 1024,
    'num_samples': 4096,
    'normalize': True,
    'sampling_method': 'all',
    'use_custom_augmentation': True,
}

train_config = {
    'batch_size': 64,
    'learning_rate': 5e-4,
    'weight_decay': 5e-2,
    'save_interval': 10,
    'epochs': 2,
    't_max': 100,
    'warmup_epochs': 10,
    'min_lr': 1.0e-6,
    'grad_norm_clip': 10,
}

if __name__ == '__main__':
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        from p4p_amd.ddp import configure_rccl_env
        configure_rccl_env()               # bounds RCCL's channels (= CUs) so that the library knows how many CUs to leave it
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    trainer = Pix4PointTrainer(model_config=model_config, dataset_config=dataset_config, train_config=train_config,
                               device='cuda')
    trainer.train()
    if dist.is_initialized():
        dist.destroy_process_group()
