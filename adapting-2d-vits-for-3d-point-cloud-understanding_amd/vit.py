"""The ViT definition the build owns (timm 1.0.16 is not installed and not vendored by the reference).

``create_model`` stands in for ``timm.create_model`` at reference src/models/pix4point.py:220: it returns
a parameter container whose module tree and state-dict names are timm's ``VisionTransformer``
(``cls_token``, ``pos_embed``, ``patch_embed.proj``, ``blocks.i.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,
mlp.fc2}``, ``norm``, ``head``), so timm / reference checkpoints load by name.  The block arithmetic
itself runs in functional.encoder on the HIP kernels; these modules never execute a torch forward.
"""
import os
import re

import torch
import torch.nn as nn

# (embed_dim, depth, heads) of the timm families the reference's configs name
_FAMILIES = {'tiny': (192, 12, 3), 'small': (384, 12, 6), 'base': (768, 12, 12), 'large': (1024, 24, 16)}


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """timm Block at its defaults: pre-norm, LayerNorm eps 1e-6, qkv bias, MLP ratio 4, exact GELU,
    no LayerScale / DropPath."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))



class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, img_size=384, patch_size=16, num_classes=1000):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, 1 + (img_size // patch_size) ** 2, embed_dim) * .02)
        self.patch_embed = _PatchEmbed(embed_dim, patch_size)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)


def parse_model_name(name):
    """'vit_base_patch16_384.augreg_in21k_ft_in1k' -> (768, 12, 12, 384, 16).  Also accepts the explicit
    form 'vit_custom_d192_l2_h3' (embed dim, depth, heads) for the small plumbing configs."""
    m = re.match(r'vit_custom_d(\d+)_l(\d+)_h(\d+)', name)
    if m:
        return int(m.group(1)), int(m.group(2)), int(m.group(3)), 384, 16
    m = re.match(r'vit_(tiny|small|base|large)_patch(\d+)_(\d+)', name)
    if not m:
        raise ValueError(f'unknown ViT name {name!r}: expected vit_{{tiny,small,base,large}}_patch16_{{224,384}}[.tag] '
                         'or vit_custom_d<dim>_l<depth>_h<heads>')
    d, l, h = _FAMILIES[m.group(1)]
    return d, l, h, int(m.group(3)), int(m.group(2))


def create_model(name, pretrained=False):
    d, l, h, img, patch = parse_model_name(name)
    if d % h or d // h != 64:
        raise ValueError(f'{name}: head_dim {d // h} unsupported (the attention kernels are built for 64)')
    model = VisionTransformer(d, l, h, img, patch)
    if pretrained:
        root = os.environ.get('P4P_PRETRAINED_DIR', '')
        path = os.path.join(root, name + '.pt')
        if not root or not os.path.exists(path):
            raise RuntimeError(
                f'pretrained=True needs a local timm state dict at $P4P_PRETRAINED_DIR/{name}.pt: this build never '
                'fetches weights (timm.create_model(pretrained=True) downloads from a hub, pix4point.py:220)')
        sd = torch.load(path, map_location='cpu')
        model.load_state_dict(sd.get('state_dict', sd), strict=False)
    return model
