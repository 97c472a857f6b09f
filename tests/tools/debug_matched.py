"""Development aid: where does the HIP pipeline differ from tests/emulate.py?  Op by op with identical inputs, then the
model stage by stage.  python tests/tools/debug_matched.py  (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import p4p_amd                                    # noqa: E402
from p4p_amd import functional as F, ops          # noqa: E402
from oracle import pix4point_oracle as O          # noqa: E402
from tests import emulate as E                    # noqa: E402
from tests.test_gpu_02_model import build_pair       # noqa: E402

DEV = 'cuda'


def rep(name, got, want):
    got, want = got.float(), want.float()
    d = (got - want).abs()
    nz = int((d > 0).sum())
    print(f'{name:34s} max|d| {float(d.max()):.3e}  mismatching {nz}/{d.numel()} ({nz / d.numel():.2e})  '
          f'rel_l2 {float(d.norm() / want.norm()):.2e}', flush=True)


def main():
    g = torch.Generator().manual_seed(0)
    M, D = 514, 192
    x = torch.randn(M, D, generator=g).to(DEV)
    w, b = (torch.randn(D, generator=g) * 0.1 + 1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    y, _, _, _ = ops.layernorm_fwd(x, w, b, 1e-6)
    rep('layernorm -> bf16', y, E._r(E._ln(x, w, b, 1e-6)))
    W = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(DEV)
    bb = (torch.randn(3 * D, generator=g) * 0.1).to(DEV)
    Wb, _ = ops.cast_weight(W, True, False)
    rep('cast_weight', Wb, E._r(W))
    xb = y
    q = ops.gemm_tn(xb, Wb, ops.EPI_BF16, bias=bb)
    rep('gemm EPI_BF16 + bias', q, E._r(xb.float() @ Wb.float().t() + bb))
    gl = ops.gemm_tn(xb, Wb, ops.EPI_BF16_GELU, bias=bb)
    rep('gemm EPI_BF16_GELU', gl, E._r(E._gelu(xb.float() @ Wb.float().t() + bb)))
    W2 = (torch.randn(D, D, generator=g) / D ** 0.5).to(DEV)
    W2b, _ = ops.cast_weight(W2, True, False)
    r1 = ops.gemm_tn(xb, W2b, ops.EPI_F32_RESID, bias=b, resid=x)
    rep('gemm EPI_F32_RESID', r1, xb.float() @ W2b.float().t() + b + x)
    for B, T, H in ((2, 257, 3), (2, 513, 3), (1, 2049, 2), (3, 81, 3)):
        qkv = (torch.randn(B * T, 3 * H * 64, generator=g)).to(DEV).bfloat16()
        out, lse = ops.attention_fwd(qkv, B, T, H)
        rep(f'attention B{B} T{T} H{H}', out, E.attention_emulated(qkv.float(), B, T, H))
    # ---- model stages
    Dm, L, Hm, Bm, N = 192, 2, 3, 8, 1024
    oracle, model = build_pair(40, Dm, L, Hm, seed=Dm + L)
    oracle.eval(), model.eval()
    oracle = oracle.to(DEV)
    pts = O.synthetic_clouds(Bm, N, seed=N)
    start = torch.arange(Bm) * 7
    idx = O.farthest_point_sampling(pts, N // 4, start)
    cen = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    nb = O.knn_indices(cen, pts, 16).to(DEV)
    pts, start, idx = pts.to(DEV), start.to(DEV), idx.to(DEV)
    pv, mv = oracle.model, model.model
    with torch.no_grad():
        centers, tok = mv.patch_embed.tokens(pts, pts, start)
        convs = pv.patch_embed.convs[0]
        params = {f'{i}.{n}': p for i in (0, 1) for n, p in convs[i].named_parameters()}
        running = {f'{i}.{j}': (convs[i][j].running_mean, convs[i][j].running_var) for i, j in ((0, 2), (1, 1), (1, 4))}
        etok = E.p3embed_emulated(pts, nb, params, 16, eps=convs[0][2].eps, running=running)
        rep('tokens (eval)', tok.reshape(-1, 256), etok)
        S = N // 4
        h = F.pos_mlp0(centers, mv.pos_embed[0].weight, mv.pos_embed[0].bias)
        eh = E._gelu(centers.reshape(-1, 3) @ pv.pos_embed[0].weight.t() + pv.pos_embed[0].bias)
        rep('pos_mlp0', h.reshape(-1, 128), eh)
        pos = F.token_stream(h, mv.pos_embed[2].weight, mv.pos_embed[2].bias, mv.cls_pos)
        feats = F.token_stream(tok, mv.proj.weight, mv.proj.bias, mv.cls_token)
        ext = E._r(etok) @ E._r(pv.proj.weight).t() + pv.proj.bias
        epe = E._r(eh) @ E._r(pv.pos_embed[2].weight).t() + pv.pos_embed[2].bias
        efeats = torch.cat([pv.cls_token.expand(Bm, -1, -1), ext.view(Bm, S, Dm)], dim=1)
        epos = torch.cat([pv.cls_pos.expand(Bm, -1, -1), epe.view(Bm, S, Dm)], dim=1)
        rep('feats stream', feats, efeats)
        rep('pos stream', pos, epos)
        for l in range(L + 1):
            yk = F.encoder(feats, pos, list(mv.vit_blocks)[:l], mv.norm, Hm)
            ye = E.encoder_emulated(feats, pos, list(pv.vit_blocks)[:l], pv.norm, Hm)
            rep(f'encoder, {l} blocks (same inputs)', yk, ye)


if __name__ == '__main__':
    main()
