"""Row 4 of SURVEY.md section 8a (the ViT encoder block) against an independent third-party implementation.

The reference takes its blocks from timm 1.0.16 (`timm.create_model`, src/models/pix4point.py:220-227), which is neither
vendored nor installed, so the oracle's `Block` restates timm's published definition and the reference holds no fixture for it
("parity unpinned", oracle header).  What IS installed in this image is Hugging Face `transformers`, whose `ViTLayer` is the same
published architecture (Dosovitskiy et al.) written by other hands, and whose checkpoint converter for exactly this family
(`convert_vit_timm_to_pytorch.py`: timm `vit_base_patch16_384` -> HF ViT) fixes the weight correspondence used below:
`attn.qkv` rows [0:D | D:2D | 2D:3D] = query | key | value, `attn.proj` = the attention output projection, `norm1` / `norm2` =
`layernorm_before` / `layernorm_after`, `mlp.fc1` / `mlp.fc2` = the MLP's two Linears, LayerNorm eps from the config (timm: 1e-6),
exact-erf GELU.  This test loads ONE set of random weights into both and compares outputs and gradients in fp32: the oracle's block
wiring, head split, softmax scale, residual order and activation are thereby checked against a second implementation of the
model family the reference instantiates -- not against timm itself, which stays the stated gap."""
import pytest
import torch

from oracle import pix4point_oracle as O

vit = pytest.importorskip('transformers.models.vit.modeling_vit')


def _hf_layer(D, H, blk):
    cfg = vit.ViTConfig(hidden_size=D, num_attention_heads=H, intermediate_size=4 * D, hidden_act='gelu', layer_norm_eps=1e-6,
                        qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg._attn_implementation = 'eager'
    layer = vit.ViTLayer(cfg).eval()
    names = dict(layer.named_parameters())
    sd = blk.state_dict()

    def pick(*cands):
        for c in cands:
            if c in names:
                return names[c]
        raise KeyError(f'none of {cands} in ViTLayer: {sorted(names)}')

    with torch.no_grad():
        for i, part in enumerate(('q', 'k', 'v')):          # transformers 4.x: attention.attention.query ...; 5.x: attention.q_proj
            long = {'q': 'query', 'k': 'key', 'v': 'value'}[part]
            pick(f'attention.{part}_proj.weight', f'attention.attention.{long}.weight').copy_(sd['attn.qkv.weight'][i * D:(i + 1) * D])
            pick(f'attention.{part}_proj.bias', f'attention.attention.{long}.bias').copy_(sd['attn.qkv.bias'][i * D:(i + 1) * D])
        pick('attention.o_proj.weight', 'attention.output.dense.weight').copy_(sd['attn.proj.weight'])
        pick('attention.o_proj.bias', 'attention.output.dense.bias').copy_(sd['attn.proj.bias'])
        for a, b in (('norm1', 'layernorm_before'), ('norm2', 'layernorm_after')):
            names[f'{b}.weight'].copy_(sd[f'{a}.weight'])
            names[f'{b}.bias'].copy_(sd[f'{a}.bias'])
        pick('mlp.fc1.weight', 'intermediate.dense.weight').copy_(sd['mlp.fc1.weight'])
        pick('mlp.fc1.bias', 'intermediate.dense.bias').copy_(sd['mlp.fc1.bias'])
        pick('mlp.fc2.weight', 'output.dense.weight').copy_(sd['mlp.fc2.weight'])
        pick('mlp.fc2.bias', 'output.dense.bias').copy_(sd['mlp.fc2.bias'])
    return layer


def _run(layer, x):
    y = layer(x)
    return y[0] if isinstance(y, (tuple, list)) else y


@pytest.mark.parametrize('D,H,T', [(192, 3, 65), (768, 12, 33)])
def test_block_forward_and_gradients_match_hf_vit_layer(D, H, T):
    torch.manual_seed(D + T)
    blk = O.Block(D, H).eval()
    blk.load_state_dict(O.seeded_state_dict(blk, 11, scale=1.0))
    with torch.no_grad():                                   # LayerNorm affine and biases away from their (1, 0) defaults
        for n, p in blk.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    hf = _hf_layer(D, H, blk)
    x = torch.randn(2, T, D)
    xo, xh = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yo, yh = blk(xo), _run(hf, xh)
    assert float((yo - yh).detach().abs().max()) <= 2e-5 * max(1.0, float(yh.detach().abs().max()))
    g = torch.randn_like(yo)
    yo.backward(g)
    yh.backward(g)
    assert float((xo.grad - xh.grad).abs().max()) <= 2e-5 * max(1.0, float(xh.grad.abs().max()))
    hp = dict(hf.named_parameters())
    gq = torch.cat([hp[k].grad for k in hp if k.endswith('weight') and ('q_proj' in k or 'query' in k)]
                   + [hp[k].grad for k in hp if k.endswith('weight') and ('k_proj' in k or 'key' in k)]
                   + [hp[k].grad for k in hp if k.endswith('weight') and ('v_proj' in k or 'value' in k)])
    assert float((blk.attn.qkv.weight.grad - gq).abs().max()) <= 2e-5 * max(1.0, float(gq.abs().max()))
    gfc1 = next(hp[k].grad for k in ('mlp.fc1.weight', 'intermediate.dense.weight') if k in hp)
    assert float((blk.mlp.fc1.weight.grad - gfc1).abs().max()) <= 2e-5 * max(1.0, float(gfc1.abs().max()))


def test_stack_of_blocks_and_final_norm_match_hf():
    """pix4point.py:254-256: `for blk in vit_blocks: feats = blk(feats + pos_embed)`, then `norm` -- two blocks and the final
    LayerNorm of both implementations on the same token / positional streams."""
    D, H, T = 192, 3, 65
    torch.manual_seed(3)
    blocks = [O.Block(D, H).eval() for _ in range(2)]
    for i, b in enumerate(blocks):
        b.load_state_dict(O.seeded_state_dict(b, 20 + i, scale=1.0))
    hfs = [_hf_layer(D, H, b) for b in blocks]
    norm = torch.nn.LayerNorm(D, eps=1e-6)
    feats, pos = torch.randn(2, T, D), 0.3 * torch.randn(2, T, D)
    a = b_ = feats
    with torch.no_grad():
        for blk, hf in zip(blocks, hfs):
            a = blk(a + pos)
            b_ = _run(hf, b_ + pos)
        assert float((norm(a) - norm(b_)).abs().max()) <= 3e-5
