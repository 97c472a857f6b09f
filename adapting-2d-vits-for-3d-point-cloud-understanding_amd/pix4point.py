"""Drop-in for the reference's ``src/models/pix4point.py``: same free functions, module classes,
constructor signatures, attribute names and state-dict keys; every hot op runs on the HIP library.

Differences, all additive: ``start_idx`` (explicit FPS start, the reference draws it unseeded at
pix4point.py:30) and ``grad_sink`` (data-parallel gradient hand-off, the reference has no DDP).
"""
import math
from typing import Iterator, List, Optional, Tuple, cast

import torch
import torch.nn as nn

from . import functional as F
from . import ops
from .vit import create_model



def farthest_point_sampling(points: torch.Tensor, n_samples: int,
                            start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,N,3) f32 -> int64 (B, min(n_samples,N)).  Reference pix4point.py:8-53."""
    return ops.fps(points.contiguous(), n_samples, start_idx)


def group_knn(pnts: torch.Tensor, cntrds: torch.Tensor, feats: torch.Tensor, k: int
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference pix4point.py:56-102: ((B,S,k,3), (B,S,k,C)) neighbour coordinates (absolute) and features."""
    pnts, feats = pnts.contiguous(), feats.contiguous()
    idx = ops.knn(pnts, cntrds.contiguous(), k)
    gp = ops.group_gather(pnts, idx)
    gf = gp if feats.data_ptr() == pnts.data_ptr() and feats.shape == pnts.shape else ops.group_gather(feats, idx)
    return gp, gf


def _report_trainable(params) -> None:
    """The three lines the reference's print_trainable_params methods print (pix4point.py:282-291, 407-416): same wording and
    number formats (the trainer's log is compared by eye against reference runs)."""
    sizes = [(q.numel(), q.requires_grad) for q in params]
    total = sum(n for n, _ in sizes)
    trainable = sum(n for n, on in sizes if on)
    for label, value in (('Total parameters', f'{total:,}'), ('Trainable parameters', f'{trainable:,}'),
                         ('Trainable percentage', f'{100 * trainable / total:.2f}%')):
        print(f'{label}: {value}')


class P3Embed(nn.Module):
    """Reference pix4point.py:105-191 (module tree identical, so ``convs.*`` keys match)."""

    def __init__(self, in_channels: int = 3, sample_ratio: float = 0.25, scale: int = 4, k: int = 32,
                 layers: int = 4, embed_dim: int = 256, **kwargs):
        super().__init__()
        self.sample_ratio, self.k = sample_ratio, k
        self.sample_fn, self.grouper = farthest_point_sampling, group_knn
        stages = int(math.log(1 / sample_ratio, scale))
        width = int(embed_dim // 2 ** (stages - 1))
        if in_channels != 3 or layers != 4 or stages != 1:
            raise NotImplementedError('the HIP tokenizer is built for the configuration the reference uses: '
                                      'in_channels=3, layers=4, one stage (sample_ratio=1/scale)')
        if sample_ratio != 0.25 or scale != 4:
            # the reference's forward samples N // 4 centroids whatever sample_ratio says (pix4point.py:174: n = int(n // 4));
            # other one-stage settings (e.g. 0.5 / 2) would build one module tree and run another sampling rate
            raise NotImplementedError('sample_ratio / scale other than 0.25 / 4: the forward pass samples N // 4 centroids '
                                      '(reference pix4point.py:174), which only matches these values')
        self.convs = nn.ModuleList()
        self.channel_list = [in_channels]
        conv1 = nn.Sequential(nn.Conv2d(in_channels + 3, width, 1, bias=False), nn.Conv2d(width, width, 1, bias=True),
                              nn.BatchNorm2d(width), nn.ReLU())
        conv2 = nn.Sequential(nn.Conv2d(width * 2, width * 2, 1, bias=False), nn.BatchNorm2d(width * 2), nn.ReLU(),
                              nn.Conv2d(width * 2, width, 1, bias=False), nn.BatchNorm2d(width), nn.ReLU())
        self.convs.append(nn.ModuleList([conv1, conv2]))
        self.channel_list.append(width)
        self.pool = lambda x: torch.max(x, dim=-1, keepdim=True)[0]
        self.out_channels = self.channel_list[-1]

    def tokens(self, p: torch.Tensor, f_bnc: torch.Tensor, start_idx: Optional[torch.Tensor] = None):
        """(centroids (B,S,3), tokens (B,S,E)) -- the compact layout the encoder consumes."""
        B, N, _ = p.shape
        S = int(N // 4)
        p = p.contiguous()
        cidx = self.sample_fn(p, S, start_idx).long()
        centers = torch.gather(p, 1, cidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        idx = ops.knn(p, centers, self.k)
        tok = F.p3embed(p, f_bnc.contiguous(), idx, self.k, self.training, self.convs[0])
        return centers, tok.view(B, S, -1)

    def forward(self, p: torch.Tensor, f: torch.Tensor, start_idx: Optional[torch.Tensor] = None
                ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """p (B,N,3), f (B,C,N) -> ([p, centroids], [f, tokens (B,E,S)])  (reference :166-191)."""
        centers, tok = self.tokens(p, f.transpose(1, 2), start_idx)
        return [p, centers], [f, tok.transpose(1, 2)]


class PointViT(nn.Module):
    """Reference pix4point.py:194-291."""

    def __init__(self, in_channels: int = 3, embed_dim: int = 384,
                 pretrained_model: str = "vit_small_patch16_384.augreg_in21k_ft_in1k", pretrained: bool = True,
                 frozen: bool = False, k_neighbors: int = 16, global_features: str = 'max,cls'):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.global_features = global_features.split(',')
        self.patch_embed = P3Embed(in_channels=in_channels, k=k_neighbors)
        self.proj = nn.Linear(self.patch_embed.out_channels, self.embed_dim)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128, bias=True), nn.GELU(), nn.Linear(128, self.embed_dim))
        self.vit = create_model(pretrained_model, pretrained=pretrained)
        if self.vit.embed_dim != embed_dim:
            raise ValueError(f'embed_dim={embed_dim} does not match {pretrained_model} ({self.vit.embed_dim})')
        total_params = sum(p.numel() for p in self.vit.parameters())
        print(f"Total ViT Parameters: {total_params:,}")
        self.vit_blocks: nn.ModuleList = cast(nn.ModuleList, self.vit.blocks)
        self.norm: nn.Module = cast(nn.Module, self.vit.norm)
        self.cls_token: nn.Parameter = cast(nn.Parameter, self.vit.cls_token)
        self.cls_pos = nn.Parameter(cast(torch.Tensor, self.vit.pos_embed)[:, :1, :])
        self.grad_sink = None
        if frozen:
            for name, param in self.named_parameters():
                if 'vit' in name:
                    param.requires_grad = False

    def forward(self, p: torch.Tensor, x: Optional[torch.Tensor] = None, start_idx: Optional[torch.Tensor] = None
                ) -> Tuple[List[torch.Tensor], List[torch.Tensor], torch.Tensor]:
        feats_bnc = p if x is None else x.transpose(1, 2)
        B = p.shape[0]
        centers, tok = self.patch_embed.tokens(p, feats_bnc, start_idx)
        p_list = [p, centers]
        x_list = [p.transpose(1, 2) if x is None else x, tok.transpose(1, 2)]
        h = F.pos_mlp0(centers, self.pos_embed[0].weight, self.pos_embed[0].bias)          # K = 3: not a GEMM
        # proj / pos_embed[2] write their rows straight into the (B,T,D) streams behind the cls rows (reference :245-252)
        pos = F.token_stream(h, self.pos_embed[2].weight, self.pos_embed[2].bias, self.cls_pos)
        feats = F.token_stream(tok, self.proj.weight, self.proj.bias, self.cls_token)
        if F._CAPTURE is not None:
            F._CAPTURE['streams'] = dict(centers=centers, tok=tok, h=h, pos=pos, feats=feats)
        feats = F.encoder(feats, pos, list(self.vit_blocks), self.norm, self.vit.num_heads, self.grad_sink)
        return p_list, x_list, feats

    def forward_cls_feat(self, p: torch.Tensor, x: Optional[torch.Tensor] = None,
                         start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        _, _, x = self.forward(p, x, start_idx)
        # reference :262-269 walks the configured names and takes 'cls' before 'max' inside one name
        kinds = [k for name in self.global_features for k in ('cls', 'max') if k in name]
        if kinds and len(set(kinds)) == len(kinds) and x.dtype == torch.float32:
            return F.global_feat(x, kinds)             # one kernel each way (functional._GlobalFeatFn)
        # a pooling repeated in the configuration, or a non-fp32 stream: the stock-torch form of the same selection
        pooled = {'cls': lambda: x[:, 0, :], 'max': lambda: x[:, 1:, :].max(dim=1).values}
        return torch.cat([pooled[k]() for k in kinds], dim=1)

    def get_trainable_params(self) -> Iterator[nn.Parameter]:
        return (q for q in self.parameters() if q.requires_grad)

    def print_trainable_params(self) -> None:
        _report_trainable(self.parameters())


class ClsHead(nn.Module):
    """Reference pix4point.py:294-325 (negligible FLOPs: stock torch, as SURVEY.md section 8a row 5 allows)."""

    def __init__(self, in_channels: int = 768, num_classes: int = 15, mlps: List[int] = [256, 256],
                 dropout: float = 0.5, point_dim: int = 2):
        super().__init__()
        self.point_dim = point_dim
        dims = [in_channels] + list(mlps) + [num_classes]
        layers: List[nn.Module] = []
        for i in range(len(dims) - 2):
            layers += [nn.Linear(dims[i], dims[i + 1], True), nn.BatchNorm1d(dims[i + 1]), nn.ReLU(), nn.Dropout(dropout)]
        layers += [nn.Linear(dims[-2], dims[-1], True)]
        self.head = nn.Sequential(*layers)

    def forward(self, end_points: torch.Tensor) -> torch.Tensor:
        return self.head(end_points)


class Pix4Point(nn.Module):
    """Reference pix4point.py:328-437."""

    def __init__(self, num_classes: int = 15, embed_dim: int = 768,
                 pretrained_model: str = 'vit_small_patch16_384.augreg_in21k_ft_in1k', pretrained: bool = True,
                 frozen: bool = False, k_neighbors: int = 16) -> None:
        super().__init__()
        self.pretrained = pretrained
        self.model = PointViT(pretrained_model=pretrained_model, pretrained=pretrained, embed_dim=embed_dim,
                              k_neighbors=k_neighbors, frozen=frozen)
        self.cls_head = ClsHead(in_channels=2 * embed_dim, num_classes=num_classes)
        self.initialize_weights()

    def initialize_weights(self):
        if not self.pretrained:
            torch.nn.init.normal_(self.model.cls_token, std=.02)
            torch.nn.init.normal_(self.model.cls_pos, std=.02)
        for name, module in self.named_modules():
            # the reference's prefix test never matches ('model.vit...' does not start with 'vit'), so
            # every Linear inside the ViT is re-initialised as well; kept as is
            if name.startswith('vit') and self.pretrained:
                continue
            self._init_weights(module)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_param_groups(self):
        decay, no_decay = [], []
        for name, param in self.named_parameters():
            if not param.requires_grad:
                continue
            if 'cls_token' in name or 'cls_pos' in name or name.endswith('.bias') or 'norm' in name:
                no_decay.append(param)
            else:
                decay.append(param)
        return [{'params': decay}, {'params': no_decay, 'weight_decay': 0.0}]

    def get_trainable_params(self) -> Iterator[nn.Parameter]:
        return (q for q in self.model.parameters() if q.requires_grad)       # the encoder only, as the reference counts (:403-405)

    def print_trainable_params(self) -> None:
        _report_trainable(self.model.parameters())

    def forward(self, points: torch.Tensor, start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """points (B,N,3) -> logits (B,num_classes)."""
        return self.cls_head(self.model.forward_cls_feat(points, start_idx=start_idx))
