"""Autograd wiring of the HIP kernels.  Three differentiable units cover the hot path:

* ``p3embed``  - FPS + kNN + local MLP tokenizer   (reference src/models/pix4point.py:166-191)
* ``linear``   - nn.Linear on bf16 MFMA            (PointViT.proj / pos_embed[2], :213-218,245-246)
* ``encoder``  - L x timm Block + final LayerNorm  (:254-256), positional embedding re-added per block

Forward and backward both run hand-written kernels; torch is only the allocator, the stream and
the autograd graph.  Saved activations are bf16 except the fp32 residual stream.
"""
import torch

from . import ops

# tests set this to a dict to receive the tensors every stage of the next forward passes on (tokenizer activations, the two
# token streams, per block: xin / ln1 / qkv / att / x1 / ln2 / g / xout, and the final LayerNorm output), so that a PyTorch
# statement can be checked LINK BY LINK against the pipeline's own upstream tensors (tests/test_gpu_03_matched_parity.py)
_CAPTURE = None
_CAPTURE_BWD = None        # tests only: a dict that the encoder's backward fills with every block's upstream gradient
_DEFER_SLAB_SUMS = True       # encoder backward: the M-split slabs of a block's four weight-gradient GEMMs are summed by ONE launch at the end of the
                              # block (ops.wgrad_flush) instead of one 20-us launch behind every GEMM; bit-identical (same order of additions)
_FREE_DSTACK_EARLY = False    # encoder backward: copy the returned block-input gradient out of the slab stack so that the stack is freed before the
                              # embedding-stage backward.  Measured (tools/peak_memory.py): the step's peak is 29.0 GiB (N=1024) / 56.0 GiB (N=2048) of 288
                              # either way -- it is set inside the encoder backward, where the stack must exist -- so the copy is off


def _bf16_weights(w, want_t, col0=0, cols=None):
    """bf16 operand copies (W, W^T) of an fp32 master weight.  The cache lives ON the parameter object
    (so it dies with it) and is invalidated by the tensor's version counter (any in-place update)."""
    cache = getattr(w, '_p4p_bf16', None)
    if cache is None:
        cache = {}
        w._p4p_bf16 = cache
    ent = cache.get((col0, cols))
    if (ent is not None and ent[0] == w._version and ent[3] == w.data_ptr()
            and (ent[2] is not None or not want_t)):
        return ent[1], ent[2]
    wb, wt = ops.cast_weight(w.detach(), True, want_t, cols=cols, col0=col0)
    cache[(col0, cols)] = (w._version, wb, wt, w.data_ptr())
    return wb, wt


_multi_tables = {}        # tuple of pointers -> (device descriptor table, tiles)


def refresh_weight_operands(weights, want_t):
    """Re-make the bf16 operand copies (W and, if want_t, W^T) of every STALE weight in ``weights`` (2-D fp32 parameters, whole
    matrices) in ONE launch, in place in their existing buffers; afterwards _bf16_weights() hits the cache for all of them.
    After an optimizer step that is every weight of the encoder: one launch instead of one per weight."""
    stale = []
    for w in weights:
        cache = getattr(w, '_p4p_bf16', None)
        ent = cache.get((0, None)) if cache else None
        if (ent is not None and ent[0] == w._version and ent[3] == w.data_ptr() and (ent[2] is not None or not want_t)):
            continue
        R, C = w.shape[0], w[0].numel()
        if w.dim() != 2 or R % 4 or C % 4 or w.data_ptr() % 16 or not w.is_contiguous():
            continue                                   # left to the per-tensor path of _bf16_weights
        stale.append((w, ent))
    if len(stale) < 2:
        return 0
    recs = []
    for w, ent in stale:
        R, C = w.shape
        # the old buffers are reused only if they still fit this weight (after model.to(another GPU) or a replaced
        # .data they sit on the old device / have the old shape: writing through them would corrupt other memory)
        def fits(t, shape):
            return t is not None and t.device == w.device and tuple(t.shape) == shape and t.dtype == torch.bfloat16
        wb = ent[1] if (ent is not None and fits(ent[1], (R, C))) else torch.empty(R, C, dtype=torch.bfloat16, device=w.device)
        wt = ent[2] if (ent is not None and fits(ent[2], (C, R))) else (
            torch.empty(C, R, dtype=torch.bfloat16, device=w.device) if want_t else None)
        recs.append((w, wb, wt))
    key = tuple((w.data_ptr(), wb.data_ptr(), 0 if wt is None else wt.data_ptr(), w.shape[0], w.shape[1], w.device.index)
                for w, wb, wt in recs)
    tab = _multi_tables.get(key)
    if tab is None:
        import struct
        raw, tile0 = bytearray(), 0
        for w, wb, wt in recs:
            R, C = w.shape
            tc = (C + 63) // 64
            raw += struct.pack('<QQQqiiii', w.data_ptr(), wb.data_ptr(), 0 if wt is None else wt.data_ptr(), C, R, C, tile0, tc)
            tile0 += tc * ((R + 63) // 64)
        host = torch.frombuffer(raw, dtype=torch.uint8).clone().pin_memory()
        if len(_multi_tables) > 16:
            _multi_tables.clear()
        tab = (host.to(recs[0][0].device, non_blocking=True), tile0, host)        # host kept alive with its copy
        _multi_tables[key] = tab
    ops.cast_weights_multi(tab[0], len(recs), tab[1])
    for w, wb, wt in recs:
        cache = getattr(w, '_p4p_bf16', None)
        if cache is None:
            cache = {}
            w._p4p_bf16 = cache
        cache[(0, None)] = (w._version, wb, wt, w.data_ptr())
    return len(recs)


def invalidate_operand_caches(module):
    """Drop the cached bf16 operand copies of every parameter of ``module`` (they are re-made at the next forward).

    The caches are keyed on the parameter's version counter and data pointer, which every in-place torch op and this
    package's own raw-pointer writers (FusedAdamW.step, ddp.broadcast_parameters) advance.  A write that bypasses the
    counter -- ``p.data.copy_(...)``, a third-party kernel, EMA / SWA code that goes through ``.data`` -- leaves STALE
    weights in forward and backward without an error; call this after such a write (load_checkpoint does)."""
    n = 0
    for p in module.parameters():
        for attr in ('_p4p_bf16', '_p4p_scaled', '_p4p_up'):
            if hasattr(p, attr):
                delattr(p, attr)
                n += 1
    return n


# ================================================================================================
# tokenizer
# ================================================================================================
class _P3EmbedFn(torch.autograd.Function):
    """tokens f32 [B*S, E] = P3Embed local MLP over the kNN groups (one stage)."""

    @staticmethod
    def forward(ctx, points, feats, idx, k, training, need_bwd, bns, W1, W2, b2, g1, be1, W3, g2, be2, W4, g3, be3):
        bn1, bn2, bn3 = bns
        B, N, _ = points.shape
        S = idx.shape[1]
        G, P = B * S, B * S * k
        C1 = W2.shape[0]
        W1m, W2m = W1.detach().reshape(C1, 6), W2.detach().reshape(C1, C1)
        mom_part = ops.embed_moments(points, feats, idx) if training else None
        aeff, sc1, sh1, mean1, is1, mom = ops.embed_fold(W1m, W2m, b2.detach(), bn1, mom_part, P, training)
        F1, pooled, arg1 = ops.embed_stage1(points, feats, idx, aeff, sc1, sh1)
        W3a, W3aT = _bf16_weights(W3, need_bwd, 0, C1)
        W3b, W3bT = _bf16_weights(W3, need_bwd, C1, W3.shape[1] - C1)
        U = ops.gemm_tn(pooled, W3a, ops.EPI_F32_RESID)
        H2, st2 = ops.gemm_tn(F1, W3b, ops.EPI_BF16_ROWGROUP, rowgrp=U, group=k)
        C2 = H2.shape[1]
        sums2 = ops.colreduce(st2, st2.shape[0], 2 * C2) if training else None
        sc2, sh2, mean2, is2 = ops.bn_finalize(sums2, bn2, P, training)
        F2 = ops.bn_relu(H2, sc2, sh2)
        W4b, W4T = _bf16_weights(W4, need_bwd)
        H3, st3 = ops.gemm_tn(F2, W4b, ops.EPI_BF16_STATS)
        C3 = H3.shape[1]
        sums3 = ops.colreduce(st3, st3.shape[0], 2 * C3) if training else None
        sc3, sh3, mean3, is3 = ops.bn_finalize(sums3, bn3, P, training)
        tok, arg3 = ops.bn_relu_maxpool(H3, sc3, sh3, G, k)
        if _CAPTURE is not None:
            _CAPTURE['embed'] = dict(F1=F1, pooled=pooled, U=U, H2=H2, F2=F2, H3=H3, tok=tok)
        if training:
            for bn in bns:
                bn.num_batches_tracked += 1
        ctx.training, ctx.k, ctx.dims = training, k, (B, N, S, C1, C2, C3)
        ctx.save_for_backward(points, feats, idx, W1, W2, g1, F1, pooled, arg1, aeff, mom, is1, H2, F2, sc2, sh2,
                              mean2, is2, H3, arg3, sc3, sh3, mean3, is3, W3aT, W3bT, W4T)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        if not ctx.training:
            raise RuntimeError('P3Embed backward is implemented for train-mode BatchNorm only '
                               '(the reference never back-propagates in eval mode)')
        (points, feats, idx, W1, W2, g1, F1, pooled, arg1, aeff, mom, is1, H2, F2, sc2, sh2, mean2, is2, H3, arg3, sc3,
         sh3, mean3, is3, W3aT, W3bT, W4T) = ctx.saved_tensors
        B, N, S, C1, C2, C3 = ctx.dims
        k, G = ctx.k, B * S
        dtok = dtok.contiguous()
        dH3, s3 = ops.pool_bn_bwd(dtok, H3, arg3, sc3, sh3, mean3, is3, G, k)
        dW4 = ops.gemm_wgrad(dH3, F2)
        dF2 = ops.gemm_tn(dH3, W4T, ops.EPI_BF16)
        s2, dU = ops.relu_bn_bwd_(dF2, H2, sc2, sh2, mean2, is2, group=k)   # dF2 now holds dH2; dU = its sums over the k neighbours
        dW3b = ops.gemm_wgrad(dF2, F1)
        dF1 = ops.gemm_tn(dF2, W3bT, ops.EPI_BF16)
        dW3a = ops.gemm_wgrad(dU, pooled)
        dPooled = ops.gemm_tn(dU, W3aT, ops.EPI_BF16)
        dW1, dW2, db2, dg1, db1 = ops.embed_stage1_bwd(points, feats, idx, F1, dF1, dPooled, arg1, mom, aeff,
                                                       W1.detach().reshape(C1, 6), W2.detach().reshape(C1, C1),
                                                       g1.detach(), is1)
        dW3 = torch.cat([dW3a, dW3b], dim=1)
        return (None, None, None, None, None, None, None, dW1.view_as(W1), dW2.view_as(W2), db2, dg1, db1, dW3.view(C2, C2, 1, 1),
                s2[1].clone(), s2[0].clone(), dW4.view(C3, C2, 1, 1), s3[1].clone(), s3[0].clone())


def p3embed(points, feats_bnc, idx, k, training, convs):
    """convs = P3Embed.convs[i] (ModuleList[conv1 Sequential, conv2 Sequential], pix4point.py:135-158)."""
    c1, c2 = convs[0], convs[1]
    bns = (c1[2], c2[1], c2[4])
    need_bwd = bool(training and torch.is_grad_enabled() and any(p.requires_grad for p in convs.parameters()))
    return _P3EmbedFn.apply(points, feats_bnc, idx, k, training, need_bwd, bns, c1[0].weight, c1[1].weight, c1[1].bias,
                            c1[2].weight, c1[2].bias, c2[0].weight, c2[1].weight, c2[1].bias, c2[3].weight,
                            c2[4].weight, c2[4].bias)


# ================================================================================================
# Linear
# ================================================================================================
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        x2 = x.reshape(-1, x.shape[-1])
        xb = ops.cast_bf16(x2.contiguous()) if x2.dtype == torch.float32 else x2.contiguous()
        need_t = ctx.needs_input_grad[0]
        Wb, Wt = _bf16_weights(W, need_t)
        y = ops.gemm_tn(xb, Wb, ops.EPI_F32_RESID, bias=b.detach() if b is not None else None)
        ctx.save_for_backward(xb, Wt)
        ctx.has_bias, ctx.xshape = b is not None, x.shape
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, Wt = ctx.saved_tensors
        g = ops.cast_bf16(dy.reshape(-1, dy.shape[-1]).contiguous())
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_tn(g, Wt, ops.EPI_F32_RESID).view(ctx.xshape)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                dW, db = ops.gemm_wgrad(g, xb, want_bias=True)
            else:
                dW = ops.gemm_wgrad(g, xb)
        elif want_b:
            db = ops.colsum(g)
        return dx, dW, db


def linear(x, weight, bias):
    return _LinearFn.apply(x, weight, bias)


class _TokenStreamFn(torch.autograd.Function):
    """stream [B, S+1, N] f32 = cat([cls_vec expanded, Linear(x)], dim=1) (src/models/pix4point.py:245-252) with the
    GEMM writing the patch rows straight into the stream (no torch.cat), and the backward reading them back as the
    compact bf16 operand of the dgrad / wgrad GEMMs (no slice copy, no separate cast)."""

    @staticmethod
    def forward(ctx, x, W, b, cls_vec):
        B, S, K = x.shape
        N = W.shape[0]
        x2 = x.reshape(B * S, K)
        xb = ops.cast_bf16(x2.contiguous()) if x2.dtype == torch.float32 else x2.contiguous()
        need_t = ctx.needs_input_grad[0]
        Wb, Wt = _bf16_weights(W, need_t)
        out = torch.empty(B * (S + 1), N, dtype=torch.float32, device=x.device)
        ops.gemm_tn(xb, Wb, ops.EPI_F32_RESID, bias=b.detach() if b is not None else None, out=out, group=S)
        ops.cls_rows_set(out, cls_vec.detach().reshape(N).contiguous(), B, S + 1, N)
        ctx.save_for_backward(xb, Wt)
        ctx.dims = (B, S, K, N, b is not None)
        return out.view(B, S + 1, N)

    @staticmethod
    def backward(ctx, dy):
        xb, Wt = ctx.saved_tensors
        B, S, K, N, has_bias = ctx.dims
        dy = dy.contiguous()
        if dy.dtype != torch.float32:
            dy = dy.float()
        g = ops.token_rows_to_compact(dy, B, S, N)                      # bf16 [B*S, N]
        dx = dW = db = dcls = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_tn(g, Wt, ops.EPI_F32_RESID).view(B, S, K)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if want_b:
                dW, db = ops.gemm_wgrad(g, xb, want_bias=True)
            else:
                dW = ops.gemm_wgrad(g, xb)
        elif want_b:
            db = ops.colsum(g)
        if ctx.needs_input_grad[3]:
            dcls = ops.cls_rows_sum(dy, B, S + 1, N).view(1, 1, N)
        return dx, dW, db, dcls


class _PosMlp0Fn(torch.autograd.Function):
    """GELU(Linear(3, C)(centers)): first layer of PointViT.pos_embed (src/models/pix4point.py:214-218,246)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = x.reshape(-1, 3).contiguous().float()
        Wd, bd = W.detach().contiguous(), b.detach().contiguous()
        ctx.save_for_backward(x2, Wd, bd)
        return ops.pos_mlp0_fwd(x2, Wd, bd).view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dh):
        x2, Wd, bd = ctx.saved_tensors
        dW, db = ops.pos_mlp0_bwd(x2, Wd, bd, dh.reshape(-1, dh.shape[-1]).contiguous().float())
        return None, dW, db


def pos_mlp0(x, weight, bias):
    return _PosMlp0Fn.apply(x, weight, bias)


class _GlobalFeatFn(torch.autograd.Function):
    """cat of {max over patch tokens x[:,1:], cls token x[:,0]} in the given order (src/models/pix4point.py:260-271); the
    backward writes dx in one pass instead of torch's zeros + scatter + slice-pad + add over (B,T,D)."""

    @staticmethod
    def forward(ctx, x, kinds):
        x = x.contiguous()
        B, T, D = x.shape
        mx, arg, cls = ops.global_feat_fwd(x, 'max' in kinds, 'cls' in kinds)
        ctx.kinds, ctx.dims = kinds, (B, T, D)
        ctx.save_for_backward(arg if arg is not None else x.new_empty(0, dtype=torch.int32))
        return torch.cat([mx if k == 'max' else cls for k in kinds], dim=1)

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, T, D = ctx.dims
        dy = dy.contiguous().float()
        dmax = dcls = None
        for j, k in enumerate(ctx.kinds):
            piece = dy[:, j * D:(j + 1) * D].contiguous()
            if k == 'max':
                dmax = piece
            else:
                dcls = piece
        return ops.global_feat_bwd(dmax, arg if dmax is not None else None, dcls, B, T, D, dy.device), None


def global_feat(x, kinds):
    """kinds: tuple of 'max' / 'cls' (one of each at most), in output order."""
    return _GlobalFeatFn.apply(x, tuple(kinds))


def token_stream(x, weight, bias, cls_vec):
    """[B, S, K] -> [B, S+1, N]: cls_vec in row 0 of every sequence, Linear(x) behind it."""
    return _TokenStreamFn.apply(x, weight, bias, cls_vec)


# ================================================================================================
# ViT encoder: L blocks + final norm
# ================================================================================================
_BLOCK_PARAMS = 12     # n1w n1b qkvw qkvb projw projb n2w n2b fc1w fc1b fc2w fc2b
# Settled by in-step A/B measurements, fixed here rather than switchable:
# * the bias gradients of qkv / fc1 come out of the kernels that produce dqkv / dh (no separate column-sum pass); for the
#   qkv bias only up to T = 1024 (T = 513: 114.6 -> 113.35 ms per step fused; T = 2049: no difference, separate pass kept);
# * blk(x + pos) (src/models/pix4point.py:255): the fc2 epilogue of block l adds pos for block l + 1.
_FUSE_QKV_BIAS_MAXT = 1024


def block_params(blk):
    return [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight,
            blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
            blk.mlp.fc2.weight, blk.mlp.fc2.bias]


class _EncoderFn(torch.autograd.Function):
    """feats, pos: f32 [B,T,D].  For every block: x = blk(x + pos); then the final LayerNorm."""

    @staticmethod
    def forward(ctx, feats, pos, num_heads, eps, grad_sink, train, norm_w, norm_b, *bp):
        B, T, D = feats.shape
        M, H = B * T, num_heads
        L = len(bp) // _BLOCK_PARAMS
        x = feats.reshape(M, D)
        pos2 = pos.reshape(M, D).contiguous()
        x = x.contiguous()
        saved = []
        for l in range(L):
            n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = bp[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS]
            if l == 0:
                ln1, xin, mean1, rstd1 = ops.layernorm_fwd(x, n1w.detach(), n1b.detach(), eps, pos=pos2, want_sum=True,
                                                           want_stats=train)
            else:       # x already is x_l + pos: the fc2 epilogue of the block below added it (one stream in, no sum written)
                ln1, _, mean1, rstd1 = ops.layernorm_fwd(x, n1w.detach(), n1b.detach(), eps, want_stats=train)
                xin = x
            Wqkv, _ = _bf16_weights(qkvw, train)
            qkv = ops.gemm_tn(ln1, Wqkv, ops.EPI_BF16, bias=qkvb.detach())
            att, lse = ops.attention_fwd(qkv, B, T, H, want_lse=train)
            Wproj, _ = _bf16_weights(projw, train)
            x1 = ops.gemm_tn(att, Wproj, ops.EPI_F32_RESID, bias=projb.detach(), resid=xin)
            ln2, _, mean2, rstd2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), eps, want_stats=train)
            Wfc1, _ = _bf16_weights(fc1w, train)
            if train:
                g, hpre = ops.gemm_tn(ln2, Wfc1, ops.EPI_BF16_GELU, bias=fc1b.detach(), want_pre=True)
            else:
                g, hpre = ops.gemm_tn(ln2, Wfc1, ops.EPI_BF16_GELU, bias=fc1b.detach()), None
            Wfc2, _ = _bf16_weights(fc2w, train)
            x = ops.gemm_tn(g, Wfc2, ops.EPI_F32_RESID, bias=fc2b.detach(), resid=x1,
                            resid2=pos2 if l + 1 < L else None)
            if train:
                saved += [xin, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, hpre, g]
            if _CAPTURE is not None:
                _CAPTURE.setdefault('blocks', []).append(dict(xin=xin, ln1=ln1, qkv=qkv, att=att, x1=x1, ln2=ln2, g=g, xout=x,
                                                              pos_added=l + 1 < L, lse=lse))
        y, _, meanN, rstdN = ops.layernorm_fwd(x, norm_w.detach(), norm_b.detach(), eps, out_f32=True, want_stats=train)
        if _CAPTURE is not None:
            _CAPTURE['encoder_out'] = y
        if train:
            ctx.save_for_backward(x, meanN, rstdN, norm_w, *bp, *saved)
        ctx.meta = (B, T, D, H, L, eps)
        ctx.grad_sink = grad_sink
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        B, T, D, H, L, eps = ctx.meta
        M = B * T
        t = ctx.saved_tensors
        xN, meanN, rstdN, norm_w = t[:4]
        bp = t[4:4 + L * _BLOCK_PARAMS]
        saved = t[4 + L * _BLOCK_PARAMS:]
        need = ctx.needs_input_grad           # feats, pos, heads, eps, sink, train, norm_w, norm_b, *bp
        dev = dy.device
        dxb = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        # every LayerNorm backward also returns the column sums of the dx it writes: that is the bias gradient of
        # the Linear whose output gradient dx is (fc2.bias of the block below / proj.bias), with no extra pass
        dx, dnw, dnb, dx_colsum = ops.layernorm_bwd(dy.reshape(M, D).contiguous(), xN, meanN, rstdN, norm_w.detach(),
                                                    dx_bf16=dxb, want_dxsum=True)
        # the positional embedding is re-added before every block (pix4point.py:255): its gradient is the sum of the L block-input
        # gradients.  Each of them gets its own slab (the LayerNorm backward reads one buffer and writes another: the same traffic
        # as in place), summed in ONE pass at the end in the order blocks L-1 .. 0 -- 52 B per element for L = 12 instead of the
        # 96 B of a read-modify-write of the sum in every block.
        # (same box, three interleaved runs of the N=1024 step: 51.08-51.30 ms against 51.23-51.46 with the read-modify-write.)
        dstack = torch.empty(max(L, 1), M, D, dtype=torch.float32, device=dev)
        dwork = torch.empty(M, D, dtype=torch.float32, device=dev) if L else None
        grads = [None] * (L * _BLOCK_PARAMS)
        sink = ctx.grad_sink
        if sink is not None:
            sink.backward_begins()          # data parallel: from here to finish() all-reduces run beside these kernels (CUs - r grids)
        for l in reversed(range(L)):
            n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = bp[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS]
            xin, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, hpre, g = saved[l * 13:(l + 1) * 13]
            if _CAPTURE_BWD is not None:     # tests: the upstream gradient of every block (teacher-forced per-block backward checks)
                _CAPTURE_BWD.setdefault('block_dout', {})[l] = dx.detach().clone()
            wg = need[8 + l * _BLOCK_PARAMS + 2]       # block weights trainable? (frozen ViT => False)
            # data parallel: the weight-gradient GEMMs write straight into the reducer's flat buffer (ddp.GradReducer)
            dst = sink.grad_views(bp[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS]) if (wg and sink is not None) else None
            out_w = (lambda j: dst[j]) if dst is not None else (lambda j: None)
            wq = [] if _DEFER_SLAB_SUMS else None     # the block's four weight-gradient GEMMs queue their slab sums: ONE launch below
            # ---- mlp: x = x1 + fc2(gelu(fc1(ln2)))
            _, Wfc2T = _bf16_weights(fc2w, True)
            if wg:                           # fc1 bias gradient = column sums of dh, accumulated in the same epilogue
                dh, dfc1b = ops.gemm_tn(dxb, Wfc2T, ops.EPI_BF16_DGELU, pre=hpre, want_colsum=True)
            else:
                dh = ops.gemm_tn(dxb, Wfc2T, ops.EPI_BF16_DGELU, pre=hpre)
            _, Wfc1T = _bf16_weights(fc1w, True)
            dln2 = ops.gemm_tn(dh, Wfc1T, ops.EPI_BF16)
            if wg:
                dfc2w, dfc2b = ops.gemm_wgrad(dxb, g, out=out_w(10), defer=wq), dx_colsum
                dfc1w = ops.gemm_wgrad(dh, ln2, out=out_w(8), defer=wq)
            del dh
            _, dn2w, dn2b, dx_colsum = ops.layernorm_bwd(dln2, x1, mean2, rstd2, n2w.detach(), dx=dwork, dx_in=dx, dx_bf16=dxb,
                                                         want_dxsum=True)
            # ---- attention: x1 = xin + proj(attn(qkv(ln1)))
            _, WprojT = _bf16_weights(projw, True)
            datt = ops.gemm_tn(dxb, WprojT, ops.EPI_BF16)
            if wg:
                dprojw, dprojb = ops.gemm_wgrad(dxb, att, out=out_w(4), defer=wq), dx_colsum
            fuse_qkvb = T <= _FUSE_QKV_BIAS_MAXT
            if wg and fuse_qkvb:
                dqkv, dqkvb = ops.attention_bwd(qkv, att, datt, lse, B, T, H, want_colsum=True)
            else:
                dqkv = ops.attention_bwd(qkv, att, datt, lse, B, T, H)
            _, WqkvT = _bf16_weights(qkvw, True)
            dln1 = ops.gemm_tn(dqkv, WqkvT, ops.EPI_BF16)
            if wg:
                if fuse_qkvb:
                    dqkvw = ops.gemm_wgrad(dqkv, ln1, out=out_w(2), defer=wq)
                else:
                    dqkvw, dqkvb = ops.gemm_wgrad(dqkv, ln1, out=out_w(2), want_bias=True)
            del dqkv, datt
            dx = dstack[L - 1 - l]
            _, dn1w, dn1b, dx_colsum = ops.layernorm_bwd(dln1, xin, mean1, rstd1, n1w.detach(), dx=dx, dx_in=dwork, dx_bf16=dxb,
                                                         want_dxsum=True)
            if wg:
                ops.wgrad_flush(wq)                  # before the bucket is handed to the all-reduce
                grads[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS] = [dn1w, dn1b, dqkvw, dqkvb, dprojw, dprojb, dn2w, dn2b,
                                                                    dfc1w, dfc1b, dfc2w, dfc2b]
                if sink is not None:
                    grads[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS] = sink.block_ready(
                        l, bp[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS], grads[l * _BLOCK_PARAMS:(l + 1) * _BLOCK_PARAMS])
        if _CAPTURE_BWD is not None:
            _CAPTURE_BWD['encoder_din'] = dx.detach().clone()
        dpos = ops.sum_slabs(dstack) if L else torch.zeros(M, D, dtype=torch.float32, device=dev)
        if L and _FREE_DSTACK_EARLY:
            # the returned gradient would be a view of the stack and pin all L slabs (2.4 GB at N=1024 / B=256, 4.8 GB at N=2048) through the
            # embedding-stage backward: one 202-MB copy (~70 us) lets the stack go here
            dx = dx.clone()
            del dstack
        return (dx.view(B, T, D), dpos.view(B, T, D), None, None, None, None, dnw, dnb, *grads)


def encoder(feats, pos, blocks, norm, num_heads, grad_sink=None):
    bp = []
    for blk in blocks:
        bp += block_params(blk)
    train = bool(torch.is_grad_enabled() and any(t.requires_grad for t in (feats, pos, norm.weight, *bp)))
    # after an optimizer step all 4 L weight operand pairs are stale: one launch re-makes them (instead of 4 L)
    refresh_weight_operands([bp[l * _BLOCK_PARAMS + j] for l in range(len(blocks)) for j in (2, 4, 8, 10)], train)
    return _EncoderFn.apply(feats, pos, num_heads, norm.eps, grad_sink, train, norm.weight, norm.bias, *bp)
