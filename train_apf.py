"""Entry script with the reference's dict configs (reference train_apf.py), running the MI355X-native APF trainer.
Single GPU: ``python train_apf.py``; 8 GPUs of one node:
``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_apf.py``.
"""
import os

import torch
import torch.distributed as dist

from p4p_amd.apf_trainer import APFTrainer

model_config = {
    'num_classes': 15,
    'in_channels': 4,              # xyz + height (reference train_apf.py:5)
    'vit_name': 'vit_base_patch16_224',
    'pretrained': False,           # True needs a local state dict in $P4P_PRETRAINED_DIR (no downloads)
    'embedding_dim': 768,
    'npoint': 196,
    'nsample': 32,
    'dropout_rate': 0.1,
    'dropout_path_rate': 0.1,      # the reference's key (train_apf.py:12); its trainer reads 'drop_path_rate' (apf_trainer.py:79)
}

dataset_config = {
    'root_dir': 'synthetic',       # or a ScanObjectNN directory (.h5 / .npz, p4p_amd.scanobjectnn: same file-name rules as the reference)
    'variant': 'main_split',
    'augmentation': 'augmentedrot_scale75',
    'train_num_points': 2048,
    'test_num_points': 1024,
    'num_samples': 1024,
    'use_height': True,            # reference train_apf.py:29
}

train_config = {
    'batch_size': 32,
    'epochs': 2,
    'save_interval': 10,
    'label_smoothing': 0.3,
    'learning_rate': 5e-4,
    'weight_decay': 5e-2,
    'warmup_epochs': 10,
    'warmup_lr_init': 1e-3,
}

if __name__ == '__main__':
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        from p4p_amd.ddp import configure_rccl_env
        configure_rccl_env()               # bounds RCCL's channels (= CUs) so that the library knows how many CUs to leave it
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    trainer = APFTrainer(model_config=model_config, dataset_config=dataset_config, train_config=train_config,
                         device='cuda', output_dir='./output_apf')
    trainer.train()
    if dist.is_initialized():
        dist.destroy_process_group()
