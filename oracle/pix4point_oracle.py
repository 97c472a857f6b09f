"""CPU oracle for the Pix4Point hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product package never does (it fails loudly without its HIP library).

This is a plain fp32 PyTorch-on-CPU restatement of the reference algorithm, every function citing
the reference lines it follows (paths relative to the reference checkout):

* rows 1-3, 5 of SURVEY.md section 8a (FPS, kNN grouping, P3Embed, ClsHead) follow
  ``src/models/pix4point.py:8-191,294-325`` and are PINNED: ``tests/golden/make_golden.py`` imported
  those reference functions in the build container and the committed fixtures in ``tests/golden``
  hold their outputs; ``tests/test_oracle_golden.py`` checks this file against them.
* row 4 (the ViT encoder block) lives in timm 1.0.16, which is neither vendored in the reference
  nor installed: **parity unpinned** for the block arithmetic.  It is restated from the in-repo
  textual statement ``src/models/apf_utils.py:142-158`` (attention) and ``:277-293`` (block
  wiring, minus the adapter) with timm's public defaults (LayerNorm eps 1e-6, exact-erf GELU,
  qkv bias, MLP ratio 4).  Two partial pins narrow that gap: ``Attention`` reproduces the reference's own
  ``AttentionLayer`` fixture bit for bit (tests/test_apf_oracle_golden.py), and ``Block`` (forward and
  gradients, one block and a stack) agrees to fp32 rounding with Hugging Face ``transformers``' ``ViTLayer``
  -- an independent implementation of the same published model family, installed in this image, under the
  weight correspondence of its timm-checkpoint converter (tests/test_oracle_vs_hf_vit.py).  timm itself
  stays unchecked.

Policy differences that are deliberate and documented in DESIGN.md:

* ``farthest_point_sampling`` takes the start index as an explicit argument (the reference draws
  it with an unseeded ``torch.randint``, ``src/models/pix4point.py:30``).
* ``knn_indices`` orders neighbours by the EXACT fp32 squared distance ``((dx*dx+dy*dy)+dz*dz)``
  with ties going to the lowest point index.  The reference's ``torch.cdist`` takes a BLAS
  matmul path whose rounding is machine dependent (hard parts);
  ``knn_indices_cdist`` restates it literally for cross-checking on well-separated inputs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------------
# Row 1: farthest point sampling                      reference src/models/pix4point.py:8-53
# --------------------------------------------------------------------------------------------
def farthest_point_sampling(points: torch.Tensor, n_samples: int,
                            start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,N,D) fp32 -> int64 (B, min(n_samples, N)).

    Follows pix4point.py:21-53 step for step: running distance initialised to 1e10 (:27),
    first pick = ``start_idx`` (:30, made explicit), then per round record (:37), gather the
    centroid (:40-41), ``sum((p-c)**2, dim=2)`` (:44), masked min-update (:47-48) and
    ``torch.max(distance, dim=1)[1]`` (:51).
    """
    B, N, _ = points.shape
    n = min(int(n_samples), N)                                   # :23
    dev = points.device
    picked = torch.zeros(B, n, dtype=torch.long, device=dev)     # :26
    running = torch.ones(B, N, dtype=points.dtype, device=dev) * 1e10   # :27
    if start_idx is None:
        far = torch.randint(0, N, (B,), dtype=torch.long, device=dev)   # :30
    else:
        far = start_idx.to(torch.long).clone()
    rows = torch.arange(B, device=dev)
    for i in range(n):
        picked[:, i] = far                                       # :37
        c = points[rows, far, :].view(B, 1, -1)                  # :40-41
        d = torch.sum((points - c) ** 2, dim=2)                  # :44
        closer = d < running                                     # :47
        running[closer] = d[closer]                              # :48
        far = torch.max(running, dim=1)[1]                       # :51
    return picked


def fps_explicit_numpy(points: np.ndarray, n_samples: int, start_idx: np.ndarray) -> np.ndarray:
    """The same algorithm with the arithmetic spelled out, the way the HIP kernel does it:
    d = (dx*dx + dy*dy) + dz*dz in fp32 without fused multiply-add, min-update, and the arg-max
    taking the LOWEST index among equal maxima.  tests/ check it equals the torch form above on
    the golden clouds, which pins the op order and the tie rule the kernel implements."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    B, N, D = pts.shape
    assert D == 3
    n = min(int(n_samples), N)
    out = np.zeros((B, n), dtype=np.int64)
    for b in range(B):
        x, y, z = pts[b, :, 0], pts[b, :, 1], pts[b, :, 2]
        running = np.full(N, 1e10, dtype=np.float32)
        far = int(start_idx[b])
        for i in range(n):
            out[b, i] = far
            dx = x - x[far]
            dy = y - y[far]
            dz = z - z[far]
            d = (dx * dx + dy * dy) + dz * dz
            running = np.minimum(running, d)
            far = int(np.argmax(running))                        # first (lowest) index of the max
    return out


# --------------------------------------------------------------------------------------------
# Row 2: kNN grouping                                 reference src/models/pix4point.py:56-102
# --------------------------------------------------------------------------------------------
def squared_distances(centroids: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """(B,S,3),(B,N,3) -> (B,S,N): ((dx*dx + dy*dy) + dz*dz), fp32, one rounding per operation."""
    dx = centroids[:, :, None, 0] - points[:, None, :, 0]
    dy = centroids[:, :, None, 1] - points[:, None, :, 1]
    dz = centroids[:, :, None, 2] - points[:, None, :, 2]
    return (dx * dx + dy * dy) + dz * dz


def knn_indices(centroids: torch.Tensor, points: torch.Tensor, k: int) -> torch.Tensor:
    """int32 (B,S,k), ascending exact squared distance, ties -> lowest point index.

    Same result set and order as pix4point.py:87-89 (``cdist`` -> ``topk(largest=False,
    sorted=True)`` -> ``.int()``) whenever neighbouring distances are separated by more than the
    matmul-path rounding of ``cdist``; the tie rule is this build's (topk's is unspecified)."""
    d2 = squared_distances(centroids, points)
    order = torch.sort(d2, dim=-1, stable=True).indices           # stable: lowest index first
    return order[..., :k].to(torch.int32)


def knn_indices_cdist(centroids: torch.Tensor, points: torch.Tensor, k: int) -> torch.Tensor:
    """Literal pix4point.py:87-89."""
    dist = torch.cdist(centroids, points)
    return dist.topk(k=k, dim=-1, largest=False, sorted=True).indices.int()


def knn_min_gap(centroids: torch.Tensor, points: torch.Tensor, k: int) -> torch.Tensor:
    """(B,S) smallest gap between consecutive sorted squared distances among the first k+1
    neighbours.  Rows with a tiny gap are the ones where cdist's rounding may legitimately swap
    or replace neighbours; fixtures carry it so such rows are flagged instead of compared."""
    d2 = squared_distances(centroids, points)
    s = torch.sort(d2, dim=-1).values[..., :k + 1]
    return (s[..., 1:] - s[..., :-1]).min(dim=-1).values


def group_knn(pnts: torch.Tensor, cntrds: torch.Tensor, feats: torch.Tensor, k: int,
              idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """pix4point.py:56-102: gather xyz and features of the k nearest points of every centroid.
    Returns (B,S,k,3) and (B,S,k,C).  Neighbour coordinates stay ABSOLUTE (:99)."""
    if idx is None:
        idx = knn_indices(cntrds, pnts, k)
    B = pnts.shape[0]
    b = torch.arange(B, device=pnts.device).view(B, 1, 1).expand(-1, idx.shape[1], idx.shape[2])   # :94-98
    li = idx.long()
    return pnts[b, li], feats[b, li]                                          # :99-100


# --------------------------------------------------------------------------------------------
# Row 3: P3Embed tokenizer                            reference src/models/pix4point.py:105-191
# --------------------------------------------------------------------------------------------
class P3Embed(nn.Module):
    """Same module tree (and therefore state-dict names) as pix4point.py:106-164."""

    def __init__(self, in_channels: int = 3, sample_ratio: float = 0.25, scale: int = 4,
                 k: int = 32, layers: int = 4, embed_dim: int = 256):
        super().__init__()
        self.sample_ratio, self.k = sample_ratio, k
        stages = int(math.log(1 / sample_ratio, scale))                       # :123
        width = int(embed_dim // 2 ** (stages - 1))                           # :124
        self.convs = nn.ModuleList()
        self.channel_list = [in_channels]
        half = layers // 2
        for _ in range(stages):
            ch = [in_channels + 3] + [width] * half + [width * 2] * (half - 1) + [width]  # :128-133
            first: List[nn.Module] = []
            for i in range(half):                                             # :137-144
                last = i == half - 1
                first.append(nn.Conv2d(ch[i], ch[i + 1], 1, bias=last))
                if last:
                    first += [nn.BatchNorm2d(ch[i + 1]), nn.ReLU()]
            ch[half] *= 2                                                     # :147
            second: List[nn.Module] = []
            for i in range(half, layers):                                     # :149-154
                second += [nn.Conv2d(ch[i], ch[i + 1], 1, bias=False),
                           nn.BatchNorm2d(ch[i + 1]), nn.ReLU()]
            self.convs.append(nn.ModuleList([nn.Sequential(*first), nn.Sequential(*second)]))
            self.channel_list.append(width)
            in_channels = width
            width *= 2
        self.out_channels = self.channel_list[-1]

    def forward(self, p: torch.Tensor, f: torch.Tensor,
                start_idx: Optional[torch.Tensor] = None):
        """pix4point.py:166-191.  ``f`` is (B,C,N).  Returns ([p, centroids], [f, tokens (B,E,S)])."""
        out_p, out_f = [p], [f]
        n = p.shape[1]
        for convs in self.convs:
            cur_p, cur_f = out_p[-1], out_f[-1].transpose(1, 2)               # :173
            n = int(n // 4)                                                   # :174
            cidx = farthest_point_sampling(cur_p, n, start_idx)               # :175
            cpnt = torch.gather(cur_p, 1, cidx.unsqueeze(-1).expand(-1, -1, 3))   # :176
            dp, fj = group_knn(cur_p, cpnt, cur_f, self.k)                    # :178
            dp = dp.permute(0, 3, 1, 2).contiguous()                          # :179
            fj = fj.permute(0, 3, 1, 2).contiguous()                          # :180
            fj = torch.cat([dp, fj], dim=1)                                   # :182
            fj = convs[0](fj)                                                 # :183
            pooled = fj.max(dim=-1, keepdim=True)[0]                          # :163
            fj = torch.cat([pooled.expand(-1, -1, -1, self.k), fj], dim=1)    # :184-186
            out_f.append(convs[1](fj).max(dim=-1, keepdim=True)[0].squeeze(-1))   # :188
            out_p.append(cpnt)                                                # :189
        return out_p, out_f


# --------------------------------------------------------------------------------------------
# Row 4: ViT encoder (timm 1.0.16 restated; parity unpinned)
# --------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """src/models/apf_utils.py:123-158 (declared there to match timm so weights load)."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, C = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)          # :148-151
        return self.proj((a @ v).transpose(1, 2).reshape(B, T, C))            # :154-157


class Mlp(nn.Module):
    """timm Mlp as used at apf_utils.py:259-262: fc1 -> exact GELU -> fc2, hidden = 4*dim."""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    """apf_utils.py:277-293 without the adapter branch = timm Block at its defaults."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, dim * 4)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    """Parameter container with timm's names; only blocks/norm/cls_token/pos_embed are used by
    the hot path (pix4point.py:220-227), the rest exist so state dicts line up."""

    def __init__(self, dim: int, depth: int, heads: int, img: int = 384, patch: int = 16,
                 num_classes: int = 1000):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, 1 + (img // patch) ** 2, dim) * .02)
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, dim, patch, patch)
        self.blocks = nn.Sequential(*[Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)


class PointViT(nn.Module):
    """pix4point.py:194-271."""

    def __init__(self, embed_dim: int, depth: int, heads: int, k_neighbors: int = 16):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = P3Embed(in_channels=3, k=k_neighbors)              # :209-212
        self.proj = nn.Linear(self.patch_embed.out_channels, embed_dim)       # :213
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, embed_dim))  # :214-218
        self.vit = VisionTransformer(embed_dim, depth, heads)                 # :220
        self.vit_blocks = self.vit.blocks                                     # :223
        self.norm = self.vit.norm                                             # :224
        self.cls_token = self.vit.cls_token                                   # :226
        self.cls_pos = nn.Parameter(self.vit.pos_embed[:, :1, :])             # :227

    def forward(self, p: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        x = p.clone().transpose(1, 2).contiguous()                            # :238
        B = x.shape[0]
        p_list, x_list = self.patch_embed(p, x, start_idx)                    # :242
        center = p_list[-1]
        tok = self.proj(x_list[-1].transpose(1, 2))                           # :245
        pos = self.pos_embed(center)                                          # :246
        pos = torch.cat([self.cls_pos.expand(B, -1, -1), pos], dim=1)         # :249-251
        feats = torch.cat([self.cls_token.expand(B, -1, -1), tok], dim=1)     # :248,252
        for blk in self.vit_blocks:
            feats = blk(feats + pos)                                          # :254-255
        return p_list, x_list, self.norm(feats)                               # :256

    def forward_cls_feat(self, p, start_idx=None):
        _, _, x = self.forward(p, start_idx)
        return torch.cat([x[:, 1:, :].max(dim=1)[0], x[:, 0, :]], dim=1)      # :262-269 'max,cls'


# --------------------------------------------------------------------------------------------
# Row 5: head + wrapper                               reference src/models/pix4point.py:294-437
# --------------------------------------------------------------------------------------------
class ClsHead(nn.Module):
    """pix4point.py:294-325."""

    def __init__(self, in_channels: int, num_classes: int, mlps=(256, 256), dropout: float = 0.5):
        super().__init__()
        dims = [in_channels] + list(mlps) + [num_classes]
        layers: List[nn.Module] = []
        for i in range(len(dims) - 2):
            layers += [nn.Linear(dims[i], dims[i + 1], True), nn.BatchNorm1d(dims[i + 1]),
                       nn.ReLU(), nn.Dropout(dropout)]
        layers.append(nn.Linear(dims[-2], dims[-1], True))
        self.head = nn.Sequential(*layers)

    def forward(self, x):
        return self.head(x)


class Pix4Point(nn.Module):
    """pix4point.py:328-437 (forward path only; weights come from a state dict in tests)."""

    def __init__(self, num_classes: int, embed_dim: int, depth: int, heads: int, k_neighbors: int = 16):
        super().__init__()
        self.model = PointViT(embed_dim, depth, heads, k_neighbors)
        self.cls_head = ClsHead(2 * embed_dim, num_classes)

    def forward(self, points: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        return self.cls_head(self.model.forward_cls_feat(points, start_idx))  # :434-435


# --------------------------------------------------------------------------------------------
# Shared synthetic inputs: same generator in tests, bench and fixtures
# --------------------------------------------------------------------------------------------
def synthetic_clouds(B: int, N: int, seed: int, dup_frac: float = 0.0) -> torch.Tensor:
    """fp32 (B,N,3): uniform cube, centred, rescaled to max-norm 1 (the range
    ``normalize_point_cloud`` produces, src/data/augment.py:4-18).  ``dup_frac`` repeats that
    fraction of the points (tie stress)."""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, N, 3, generator=g) * 2 - 1
    if dup_frac > 0:
        nd = int(N * dup_frac)
        src = torch.randint(0, N, (B, nd), generator=g)
        dst = torch.randint(0, N, (B, nd), generator=g)
        for b in range(B):
            p[b, dst[b]] = p[b, src[b]]
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=2).max(dim=1).values.view(B, 1, 1)
    return p.contiguous()


def seeded_state_dict(module: nn.Module, seed: int, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic non-trivial weights for parity runs: every floating tensor ~ N(0, s) with a
    per-tensor s giving O(1) activations, BN running stats non-trivial, BN weight near 1."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    seen = {}                      # aliased entries (vit.blocks.* == vit_blocks.*) get one value
    for name, t in module.state_dict().items():
        if not t.is_floating_point():
            sd[name] = t.clone()
            continue
        key = (t.data_ptr(), tuple(t.shape))
        if key in seen:
            sd[name] = seen[key]
            continue
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'running_var':
            v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
        elif leaf == 'running_mean':
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() <= 1 or 'norm' in name and leaf == 'weight':
            v = torch.randn(t.shape, generator=g) * 0.05
            if leaf == 'weight':
                v = v + 1.0
        else:
            fan_in = t[0].numel() if t.dim() > 1 else t.numel()
            v = torch.randn(t.shape, generator=g) * (scale / math.sqrt(max(fan_in, 1)))
        sd[name] = seen[key] = v.to(t.dtype)
    return sd
