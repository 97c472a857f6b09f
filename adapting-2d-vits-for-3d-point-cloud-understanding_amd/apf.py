"""Drop-in for the reference's AdaptPointFormer path: ``src/models/apf.py`` (Group, Encoder, PointNet, ClassificationHead,
AdaptPointFormer), the layer classes of ``src/models/apf_utils.py`` (MortonEncoder, AttentionLayer, AdapterLayer,
APFViTLayer) and the index functions of ``src/data/sampler.py`` it calls.  Same class names, constructor signatures,
attribute names and state-dict keys; every hot op runs on the HIP library (no CPU fallback).

Additive differences: ``start_idx`` (explicit FPS start; the reference draws it unseeded, sampler.py:20) and the
``depth`` / ``num_heads`` arguments of ``AdaptPointFormer`` (the reference hard-codes 12 / 12, apf.py:295,316).
Quirks kept on purpose: ``_freeze`` leaves the adapters frozen ('adaptmlp' matches no
parameter, apf.py:345); timm's ``blocks.i.attn.*`` weights do not land in ``blocks.i.attention.*`` (strict=False,
apf.py:329-331); the block residual enters twice (apf_utils.py:232,292).
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from . import functional_apf as FA
from . import ops
from .vit import create_model


# ---------------------------------------------------------------------------------------------
# src/data/sampler.py:4-94
# ---------------------------------------------------------------------------------------------
def furthest_point_sample(xyz: torch.Tensor, npoint: int, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,N,3) f32 -> int64 (B,npoint).  sampler.py:4-30 (same algorithm as pix4point.py:8-53: one kernel serves both)."""
    if npoint > xyz.shape[1]:
        raise RuntimeError(f'furthest_point_sample: npoint={npoint} exceeds N={xyz.shape[1]}')
    return ops.fps(xyz.contiguous(), npoint, start_idx)


def knn_point(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """sampler.py:64-75: int64 (B,S,nsample) indices of the nsample nearest points.  The reference's order inside a
    group is unspecified (topk sorted=False); here it is ascending exact distance, ties to the lowest index."""
    return ops.knn(xyz.contiguous(), new_xyz.contiguous(), nsample).long()


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """sampler.py:77-94: points (B,N,C), idx (B,S) or (B,S,k) -> (B,S[,k],C)."""
    B, C = points.shape[0], points.shape[-1]
    flat = idx.reshape(B, -1, 1).expand(-1, -1, C)
    return torch.gather(points, 1, flat).view(*idx.shape, C)


# ---------------------------------------------------------------------------------------------
# src/models/apf_utils.py
# ---------------------------------------------------------------------------------------------
class MortonEncoder:
    """apf_utils.py:8-104; ``points_to_morton`` runs as one kernel (codes + stable rank per cloud)."""

    @staticmethod
    def points_to_morton(points: torch.Tensor, resolution: int = 1024) -> torch.Tensor:
        return ops.morton_order(points.contiguous().float(), resolution).long()


class AttentionLayer(nn.Module):
    """apf_utils.py:106-160 (parameter container; the arithmetic runs in functional_apf.apf_blocks)."""

    def __init__(self, dim: int, num_heads: int) -> None:
        super().__init__()
        if dim % num_heads or dim // num_heads != 64:
            raise ValueError(f'head_dim {dim / num_heads} unsupported (the attention kernels are built for 64)')
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class AdapterLayer(nn.Module):
    """apf_utils.py:162-233 (parameter container, same initialisation :191-195)."""

    def __init__(self, model_dimension: int = 768, bottleneck: int = 64, dropout: float = 0.0) -> None:
        super().__init__()
        if bottleneck != 64:
            raise NotImplementedError('the adapter GEMMs are built for the bottleneck the reference uses (64)')
        self.n_embd, self.down_size, self.dropout = model_dimension, bottleneck, dropout
        self.adapter_norm = nn.LayerNorm(self.n_embd)
        self.scale = nn.Parameter(torch.ones(1))
        self.down_proj = nn.Linear(self.n_embd, self.down_size)
        self.relu = nn.ReLU()
        self.up_proj = nn.Linear(self.down_size, self.n_embd)
        with torch.no_grad():
            nn.init.kaiming_uniform_(self.down_proj.weight, a=math.sqrt(5))
            nn.init.zeros_(self.up_proj.weight)
            nn.init.zeros_(self.down_proj.bias)
            nn.init.zeros_(self.up_proj.bias)


class _Mlp(nn.Module):
    """timm Mlp's parameter layout (fc1 / act / fc2) as constructed at apf_utils.py:259-262."""

    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class _DropPath(nn.Module):
    """timm DropPath's role at apf_utils.py:258: holds the rate; the per-sample factors are drawn in ``drop_factors``."""

    def __init__(self, drop_prob: float = 0.0) -> None:
        super().__init__()
        self.drop_prob = drop_prob


def drop_factors(blk, B, device, training):
    """(keep_attn, keep_mlp, 1/keep_prob): the per-sample Bernoulli(keep) draws of the block's two DropPath calls
    (independent, apf_utils.py:280,290) as bool (B,) tensors, or (None, None, 1.0) when DropPath is the identity."""
    p = getattr(blk.drop_path, 'drop_prob', 0.0)
    if not training or p <= 0.0:
        return None, None, 1.0
    keep = 1.0 - p
    draw = lambda: torch.rand(B, device=device) < keep
    return draw(), draw(), 1.0 / keep


class APFViTLayer(nn.Module):
    """apf_utils.py:236-293."""

    def __init__(self, dim: int = 768, num_heads: int = 12, drop_path: float = 0.0, dropout: float = 0.1) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.drop_path = _DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.mlp = _Mlp(dim, dim * 4)
        self.attention = AttentionLayer(dim=dim, num_heads=num_heads)
        self.adapter = AdapterLayer(model_dimension=dim, dropout=dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        masks = [drop_factors(self, x.shape[0], x.device, self.training)]
        p = self.adapter.dropout if self.training else 0.0
        return FA.apf_blocks(x.float().contiguous(), [self], None, self.attention.num_heads,
                             masks if masks[0][0] is not None else None, p)


# ---------------------------------------------------------------------------------------------
# src/models/apf.py
# ---------------------------------------------------------------------------------------------
class Group(nn.Module):
    """apf.py:12-112: FPS centres, kNN groups, centre-relative concat, Morton order of the groups."""

    def __init__(self, num_group: int, group_size: int):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size
        self.morton_encoder = MortonEncoder()

    def indices(self, xyz: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        """(fps_idx (B,G) i64, knn_idx (B,G,gs) i32, order (B,G) i32): the integer half of forward (apf.py:68-73,99)."""
        xyz = xyz.contiguous().float()
        fps_idx = furthest_point_sample(xyz, self.num_group, start_idx)
        center = torch.gather(xyz, 1, fps_idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        knn_idx = ops.knn(xyz, center, self.group_size)
        return fps_idx, knn_idx, ops.morton_order(center)

    def forward(self, x: torch.Tensor, xyz: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        """x (B,N,C), xyz (B,N,3) -> (neighborhood (B,G,gs,2C) f32, center (B,G,3)), Morton order (C = 3 or 4)."""
        fps_idx, knn_idx, order = self.indices(xyz, start_idx)
        nb, _, center = ops.apf_group(x.contiguous().float(), fps_idx, knn_idx, order, want_f32=True, want_x0=False)
        return nb, center


class Encoder(nn.Module):
    """apf.py:114-181 (module tree identical, so ``first_conv.*`` / ``second_conv.*`` keys match)."""

    def __init__(self, encoder_channel: int, in_channel: int):
        super().__init__()
        if in_channel not in (6, 8):
            raise NotImplementedError('the HIP encoder is built for in_channel = 6 (xyz groups) and 8 (xyz + height, the '
                                      "reference's shipped configuration: train_apf.py:5 in_channels=4, apf.py:293)")
        self.in_channel = in_channel
        if encoder_channel % 64:
            raise NotImplementedError('encoder_channel must be a multiple of 64 (GEMM K granularity)')
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(in_channel, 256, 1), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                        nn.Conv1d(256, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                        nn.Conv1d(512, self.encoder_channel, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(self.encoder_channel * 2, self.encoder_channel * 2, 1),
                                         nn.BatchNorm1d(self.encoder_channel * 2), nn.ReLU(inplace=True),
                                         nn.Conv1d(self.encoder_channel * 2, self.encoder_channel, 1))

    def encode_rows(self, x0: torch.Tensor, B: int, G: int, gs: int) -> torch.Tensor:
        """x0: the bf16 [B*G*gs, 64] operand p4p_apf_group writes -> tokens (B,G,E)."""
        return FA.apf_encoder(x0, B * G, gs, self.training, self).view(B, G, self.encoder_channel)

    def get_features(self, point_groups: torch.Tensor) -> torch.Tensor:
        B, G, n, C = point_groups.shape
        if C != self.in_channel:
            raise RuntimeError(f'point_groups: expected {self.in_channel} channels, got {C}')
        x0 = FA.split_hi_lo_input(point_groups.reshape(B * G * n, C).float())
        return self.encode_rows(x0, B, G, n)

    def forward(self, point_groups: torch.Tensor) -> torch.Tensor:
        return self.get_features(point_groups)


class PointNet(nn.Module):
    """apf.py:183-217; the fused path never materialises the fp32 neighbourhoods."""

    def __init__(self, embed_dim: int, num_group: int, group_size: int, in_channel: int):
        super().__init__()
        self.group = Group(num_group, group_size)
        self.encoder = Encoder(embed_dim, in_channel)

    def forward(self, x: torch.Tensor, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        if 2 * x.shape[-1] != self.encoder.in_channel:
            raise RuntimeError(f'PointNet: {x.shape[-1]} input channels, but the encoder was built for '
                               f'{self.encoder.in_channel // 2} (in_channels of AdaptPointFormer)')
        x = x.contiguous().float()
        fps_idx, knn_idx, order = self.group.indices(x[:, :, :3], start_idx)                 # apf.py:210: xyz = x[:,:,:3]
        _, x0, _ = ops.apf_group(x, fps_idx, knn_idx, order, want_center=False)
        return self.encoder.encode_rows(x0, x.shape[0], self.group.num_group, self.group.group_size)


class ClassificationHead(nn.Module):
    """apf.py:219-251; <0.01 % of the FLOPs: stock torch."""

    def __init__(self, in_channels: int, num_classes: int):
        super().__init__()
        self.mlp_head = nn.Sequential(nn.Linear(in_channels, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                      nn.Dropout(p=0.4), nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                      nn.Dropout(p=0.4), nn.Linear(256, num_classes))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.mlp_head(x)


def get_timm_vit(name: str, pretrained: bool = True, delete=None):
    """src/models/vit.py:40-66 on the build's own ViT container (timm is absent; weights are never fetched)."""
    model = create_model(name, pretrained=pretrained)
    if delete is None:
        return model
    sd = model.state_dict()
    for d in delete:
        sd.pop(d, None)
    return sd


class AdaptPointFormer(nn.Module):
    """apf.py:253-373."""

    def __init__(self, num_classes: int = 15, embedding_dim: int = 768, vit_name: str = 'vit_base_patch16_224',
                 pretrained: bool = True, npoint: int = 196, nsample: int = 32, in_channels: int = 3,
                 dropout_rate: float = 0.1, dropout_path_rate: float = 0.1, depth: int = 12, num_heads: int = 12) -> None:
        super().__init__()
        in_channels = in_channels * 2                                             # apf.py:293
        dpr = [v.item() for v in torch.linspace(0, dropout_path_rate, depth)]     # apf.py:298
        self.dropout = nn.Dropout(dropout_rate)
        self.encoder_norm = nn.LayerNorm(embedding_dim)
        self.point_encoder = PointNet(embedding_dim, npoint, nsample, in_channels)
        self.head = ClassificationHead(in_channels=embedding_dim, num_classes=num_classes)
        self.blocks = nn.Sequential(*[APFViTLayer(dim=embedding_dim, num_heads=num_heads, drop_path=dpr[i],
                                                  dropout=dropout_rate) for i in range(depth)])
        if vit_name:
            self.load_state_dict(get_timm_vit(vit_name, pretrained=pretrained, delete=['head.weight', 'head.bias']),
                                 strict=False)                                    # apf.py:323-331
        self._freeze()

    def _freeze(self):
        """apf.py:335-346, literally."""
        for param in self.parameters():
            param.requires_grad_(False)
        for name, param in self.named_parameters():
            if 'adaptmlp' in name or 'head' in name or 'enc_norm' in name or 'encoder' in name:
                param.requires_grad = True

    def forward(self, x: torch.Tensor, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.point_encoder(x, start_idx)                                      # apf.py:358
        blocks = list(self.blocks)
        masks = [drop_factors(b, x.shape[0], x.device, self.training) for b in blocks]
        if all(m[0] is None for m in masks):
            masks = None
        p = blocks[0].adapter.dropout if self.training else 0.0
        x = FA.apf_blocks(x, blocks, self.encoder_norm, blocks[0].attention.num_heads, masks, p)   # apf.py:361-364
        x = x.max(-2)[0]                                                          # apf.py:366
        return self.head(self.dropout(x))                                         # apf.py:368-371
