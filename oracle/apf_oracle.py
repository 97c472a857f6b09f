"""CPU oracle for the AdaptPointFormer (APF) path, SURVEY.md section 8(f) rank 1.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.

Plain fp32 PyTorch-on-CPU restatement; every function cites the reference lines it follows (paths relative
to the reference checkout):

* ``furthest_point_sample`` / ``knn_point`` / ``index_points``  - ``src/data/sampler.py:4-94``
* ``MortonEncoder``                                             - ``src/models/apf_utils.py:8-104``
* ``Group`` / ``Encoder`` / ``PointNet`` / ``ClassificationHead`` - ``src/models/apf.py:12-251``
* ``AttentionLayer`` / ``AdapterLayer``                         - ``src/models/apf_utils.py:106-233``
  PINNED: ``tests/golden/make_golden_apf.py`` loaded those reference files in the build container and the
  committed fixtures ``tests/golden/apf_*.npz`` hold their outputs; ``tests/test_apf_oracle_golden.py``
  checks this file against them.
* ``APFViTLayer`` / ``AdaptPointFormer`` (``apf_utils.py:236-293``, ``apf.py:253-373``) construct timm 1.0.16's
  ``Mlp`` and ``DropPath`` (not vendored, not installed): **parity unpinned** for those two pieces, restated
  from timm's public definition (fc1 -> exact GELU -> fc2; per-sample Bernoulli(keep) / keep).  The wiring
  around them is the reference's.

Deliberate, documented policy differences:

* ``furthest_point_sample`` takes the start index explicitly (the reference draws it with an unseeded
  ``torch.randint``, sampler.py:20).
* ``knn_point`` returns the neighbours ordered by the EXACT fp32 squared distance, ties to the lowest index.
  The reference's ``topk(sorted=False)`` on the ``-2ab+a^2+b^2`` form (sampler.py:59-61,74) leaves the order
  unspecified and its rounding is matmul dependent; everything downstream (BatchNorm statistics, max over the
  group) is invariant to the order inside a group, so the fixtures compare neighbour SETS and flag near ties.
* ``points_to_morton`` breaks equal Morton codes by the lowest index (``torch.argsort`` is not stable).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .pix4point_oracle import knn_indices, knn_min_gap, seeded_state_dict, synthetic_clouds   # noqa: F401


# --------------------------------------------------------------------------------------------
# index ops                                                   reference src/data/sampler.py:4-94
# --------------------------------------------------------------------------------------------
def furthest_point_sample(xyz: torch.Tensor, npoint: int, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,N,3) -> int64 (B,npoint).  sampler.py:16-30: running distance 1e10 (:19), first pick (:20, explicit
    here), per round record (:24), ``sum((xyz-c)**2,-1)`` (:26), ``min`` (:27), ``max(...)[1]`` (:28)."""
    B, N, _ = xyz.shape
    picked = torch.zeros(B, npoint, dtype=torch.long)
    running = torch.ones(B, N) * 1e10
    far = torch.randint(0, N, (B,), dtype=torch.long) if start_idx is None else start_idx.to(torch.long).clone()
    rows = torch.arange(B)
    for i in range(npoint):
        picked[:, i] = far
        c = xyz[rows, far, :].view(B, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        running = torch.min(running, d)
        far = torch.max(running, -1)[1]
    return picked


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """sampler.py:77-94: points (B,N,C), idx (B,S[,k]) -> (B,S[,k],C)."""
    B = points.shape[0]
    b = torch.arange(B).view(B, *([1] * (idx.dim() - 1))).expand_as(idx)
    return points[b, idx, :]


def knn_point(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """sampler.py:64-75 as a SET: int64 (B,S,nsample), here in ascending exact distance (see the header)."""
    return knn_indices(new_xyz, xyz, nsample).long()


def knn_point_literal(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """Literal sampler.py:47-75 (for cross-checks on well separated inputs)."""
    d = -2 * torch.matmul(new_xyz, xyz.permute(0, 2, 1))
    d += torch.sum(new_xyz ** 2, -1).view(new_xyz.shape[0], -1, 1)
    d += torch.sum(xyz ** 2, -1).view(xyz.shape[0], 1, -1)
    return torch.topk(d, nsample, dim=-1, largest=False, sorted=False)[1]


# --------------------------------------------------------------------------------------------
# Morton order                                                reference src/models/apf_utils.py:8-104
# --------------------------------------------------------------------------------------------
def part1by2(n: torch.Tensor) -> torch.Tensor:
    """apf_utils.py:43-48 on int64 tensors."""
    n = n & 0x000003ff
    n = (n ^ (n << 16)) & 0xff0000ff
    n = (n ^ (n << 8)) & 0x0300f00f
    n = (n ^ (n << 4)) & 0x030c30c3
    n = (n ^ (n << 2)) & 0x09249249
    return n


def morton_codes(points: torch.Tensor, resolution: int = 1024) -> torch.Tensor:
    """int64 (B,N): apf_utils.py:89-100.  fp32 min/max normalisation, ``* (resolution-1)``, truncation."""
    pmin = points.min(dim=1, keepdim=True)[0]
    pmax = points.max(dim=1, keepdim=True)[0]
    q = ((points - pmin) / (pmax - pmin + 1e-8) * (resolution - 1)).long()
    return (part1by2(q[..., 2]) << 2) + (part1by2(q[..., 1]) << 1) + part1by2(q[..., 0])


def points_to_morton(points: torch.Tensor, resolution: int = 1024) -> torch.Tensor:
    """int64 (B,N) sorting permutation (apf_utils.py:103), equal codes -> lowest index first."""
    return torch.sort(morton_codes(points, resolution), dim=1, stable=True).indices


# --------------------------------------------------------------------------------------------
# Group / Encoder / PointNet / head                           reference src/models/apf.py:12-251
# --------------------------------------------------------------------------------------------
class Group(nn.Module):
    def __init__(self, num_group: int, group_size: int):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size

    def forward(self, x: torch.Tensor, xyz: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        """apf.py:52-112.  x (B,N,C), xyz (B,N,3) -> neighbourhood (B,G,gs,2C) = [x[nbr]-x[centre], x[centre]],
        centre (B,G,3); both in Morton order of the centres."""
        B = xyz.shape[0]
        fps_idx = furthest_point_sample(xyz.contiguous(), self.num_group, start_idx)          # :68
        center = index_points(xyz, fps_idx)                                                   # :70
        new_points = index_points(x, fps_idx)                                                 # :71
        idx = knn_point(self.group_size, xyz, center)                                         # :73
        nb = index_points(x, idx)                                                             # :74-80
        nb = nb - new_points.unsqueeze(-2)                                                    # :83-84
        nb = torch.cat([nb, new_points.unsqueeze(2).repeat(1, 1, self.group_size, 1)], -1)    # :88-95
        order = points_to_morton(center)                                                      # :99 (-> :43)
        b = torch.arange(B).view(B, 1)
        return nb[b, order].contiguous(), center[b, order].contiguous()                       # :102-110


class Encoder(nn.Module):
    """apf.py:114-181 (same Sequential layout => same state-dict names)."""

    def __init__(self, encoder_channel: int, in_channel: int):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(in_channel, 256, 1), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                                        nn.Conv1d(256, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                        nn.Conv1d(512, encoder_channel, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(encoder_channel * 2, encoder_channel * 2, 1),
                                         nn.BatchNorm1d(encoder_channel * 2), nn.ReLU(inplace=True),
                                         nn.Conv1d(encoder_channel * 2, encoder_channel, 1))

    def forward(self, point_groups: torch.Tensor) -> torch.Tensor:
        B, G, n, _ = point_groups.shape
        f = self.first_conv(point_groups.reshape(B * G, n, -1).transpose(2, 1))               # :157-159
        fg = torch.max(f, dim=2, keepdim=True)[0]                                             # :160
        f = self.second_conv(torch.cat([fg.expand(-1, -1, n), f], dim=1))                     # :162-165
        return torch.max(f, dim=2)[0].reshape(B, G, self.encoder_channel)                     # :167-169


class PointNet(nn.Module):
    """apf.py:183-217."""

    def __init__(self, embed_dim: int, num_group: int, group_size: int, in_channel: int):
        super().__init__()
        self.group = Group(num_group, group_size)
        self.encoder = Encoder(embed_dim, in_channel)

    def forward(self, x: torch.Tensor, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        nb, _ = self.group(x, x[:, :, :3], start_idx)                                         # :212-214
        return self.encoder(nb)                                                               # :216


class ClassificationHead(nn.Module):
    """apf.py:219-251."""

    def __init__(self, in_channels: int, num_classes: int):
        super().__init__()
        self.mlp_head = nn.Sequential(nn.Linear(in_channels, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                      nn.Dropout(p=0.4), nn.Linear(512, 256), nn.BatchNorm1d(256),
                                      nn.ReLU(inplace=True), nn.Dropout(p=0.4), nn.Linear(256, num_classes))

    def forward(self, x):
        return self.mlp_head(x)


# --------------------------------------------------------------------------------------------
# transformer layer with the bottleneck adapter              reference src/models/apf_utils.py:106-293
# --------------------------------------------------------------------------------------------
class AttentionLayer(nn.Module):
    """apf_utils.py:106-160."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)   # :144
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)                          # :149-152
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, C))                            # :155-158


class AdapterLayer(nn.Module):
    """apf_utils.py:162-233: LN -> down(64) -> ReLU -> dropout -> up -> * scale -> + input."""

    def __init__(self, model_dimension: int = 768, bottleneck: int = 64, dropout: float = 0.0):
        super().__init__()
        self.dropout = dropout
        self.adapter_norm = nn.LayerNorm(model_dimension)
        self.scale = nn.Parameter(torch.ones(1))
        self.down_proj = nn.Linear(model_dimension, bottleneck)
        self.relu = nn.ReLU()
        self.up_proj = nn.Linear(bottleneck, model_dimension)
        with torch.no_grad():                                                                 # :191-195
            nn.init.kaiming_uniform_(self.down_proj.weight, a=math.sqrt(5))
            nn.init.zeros_(self.up_proj.weight)
            nn.init.zeros_(self.down_proj.bias)
            nn.init.zeros_(self.up_proj.bias)

    def forward(self, x):
        down = self.relu(self.down_proj(self.adapter_norm(x)))                                # :220-224
        down = nn.functional.dropout(down, p=self.dropout, training=self.training)            # :226
        return self.up_proj(down) * self.scale + x                                            # :228-232


class Mlp(nn.Module):
    """timm Mlp at its defaults as constructed at apf_utils.py:259-262 (restated; unpinned)."""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class DropPath(nn.Module):
    """timm DropPath (restated; unpinned): per-sample Bernoulli(keep) mask divided by keep, identity in eval."""

    def __init__(self, p: float):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0. or not self.training:
            return x
        keep = 1 - self.p
        mask = x.new_empty(x.shape[0], *([1] * (x.dim() - 1))).bernoulli_(keep)
        return x * mask.div_(keep)


class APFViTLayer(nn.Module):
    """apf_utils.py:236-293.  Note the residual enters twice: ``x_mlp + adapter(x) + x`` with
    ``adapter(x) = up + x`` (:232,292)."""

    def __init__(self, dim: int = 768, num_heads: int = 12, drop_path: float = 0.0, dropout: float = 0.1):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.mlp = Mlp(dim, dim * 4)
        self.attention = AttentionLayer(dim, num_heads)
        self.adapter = AdapterLayer(model_dimension=dim, dropout=dropout)

    def forward(self, x):
        x = x + self.drop_path(self.attention(self.norm1(x)))                                 # :278-281
        adapt_x = self.adapter(x)                                                             # :285
        return self.drop_path(self.mlp(self.norm2(x))) + adapt_x + x                          # :288-292


class AdaptPointFormer(nn.Module):
    """apf.py:253-373 without the timm weight download (:323-331); ``depth``/``num_heads`` are arguments here so
    small models can be built for tests (the reference fixes 12/12, :295,316)."""

    def __init__(self, num_classes: int = 15, embedding_dim: int = 768, npoint: int = 196, nsample: int = 32,
                 in_channels: int = 3, dropout_rate: float = 0.1, dropout_path_rate: float = 0.1, depth: int = 12,
                 num_heads: int = 12):
        super().__init__()
        dpr = [v.item() for v in torch.linspace(0, dropout_path_rate, depth)]                 # :298
        self.dropout = nn.Dropout(dropout_rate)
        self.encoder_norm = nn.LayerNorm(embedding_dim)
        self.point_encoder = PointNet(embedding_dim, npoint, nsample, in_channels * 2)        # :293,303-308
        self.head = ClassificationHead(embedding_dim, num_classes)
        self.blocks = nn.Sequential(*[APFViTLayer(embedding_dim, num_heads, dpr[i], dropout_rate)
                                      for i in range(depth)])
        self._freeze()

    def _freeze(self):
        """apf.py:335-346 literally: adapters stay FROZEN ('adaptmlp' matches no parameter name)."""
        for p in self.parameters():
            p.requires_grad_(False)
        for name, p in self.named_parameters():
            if 'adaptmlp' in name or 'head' in name or 'enc_norm' in name or 'encoder' in name:
                p.requires_grad = True

    def forward(self, x: torch.Tensor, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.point_encoder(x, start_idx)                                                  # :358
        for blk in self.blocks:                                                               # :361-362
            x = blk(x)
        x = self.encoder_norm(x)                                                              # :364
        x = x.max(-2)[0]                                                                      # :366
        return self.head(self.dropout(x))                                                     # :368-371
