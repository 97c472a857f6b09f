"""Autograd wiring of the HIP kernels for the AdaptPointFormer path.

* ``apf_encoder`` - the point-group MLP of ``Encoder`` (reference src/models/apf.py:114-181): five 1x1 convolutions as
  bf16 MFMA GEMMs over all B*G*gs points, train-mode BatchNorm statistics from the GEMM epilogues, the two max-pools
  and the ``cat([global, local])`` folded into a per-group accumulator init (no 1536-channel concat is ever written).
* ``apf_blocks``  - L x ``APFViTLayer`` (src/models/apf_utils.py:236-293) + ``encoder_norm`` (apf.py:364).  The block
  weights are frozen in the reference (apf.py:335-346, adapters included), so the backward is the dgrad chain only.

Everything here reuses the GEMM / attention / LayerNorm / BatchNorm kernels of the Pix4Point path; only
``csrc/apf.hip`` is new.  torch is the allocator, the stream, the autograd graph and the (tiny) stochastic masks.
"""
import torch

from . import ops
from .functional import _bf16_weights


# tests set this to a dict to receive the gating decisions of the next encoder forward (ReLU outputs F1/F2/F4 and the two
# pool arg-max tensors), so that a PyTorch statement with IDENTICAL gates can check the backward kernels tightly
_CAPTURE = None


def _bn_after_bias(bn, scale, shift, bias, training):
    """A conv bias in front of BatchNorm: cancels in train mode (it only moves running_mean), folds into the shift in
    eval mode.  The GEMMs therefore run without it."""
    if bias is None:
        return shift
    if training:
        bn.running_mean.add_(bias.detach(), alpha=bn.momentum)
        return shift
    return shift + scale * bias.detach()


def _scaled_operands(w, b, c):
    """bf16 (c W, (c W)^T) and fp32 c b of a FROZEN Linear, cached on the weight (DropPath's 1/keep lives in the weights)."""
    cache = getattr(w, '_p4p_scaled', None)
    if cache is None:
        cache = {}
        w._p4p_scaled = cache
    ent = cache.get(c)
    if ent is None or ent[0] != (w._version, b._version, w.data_ptr()):
        wb, wt = ops.cast_weight(w.detach() * c, True, True)
        ent = ((w._version, b._version, w.data_ptr()), wb, wt, b.detach() * c)
        cache[c] = ent
    return ent[1], ent[2], ent[3]


def _adapter_up_operands(uw, ub, scale):
    """bf16 (s W_up, (s W_up)^T) and fp32 s b_up of the adapter's up-projection with its `scale` folded in
    (apf_utils.py:228-230), cached on the weight until any of the three tensors changes."""
    key = (uw._version, ub._version, scale._version, uw.data_ptr())
    ent = getattr(uw, '_p4p_up', None)
    if ent is None or ent[0] != key:
        sc = scale.detach().reshape(())
        wb, wt = ops.cast_weight(uw.detach() * sc, True, True)
        ent = (key, wb, wt, ub.detach() * sc)
        uw._p4p_up = ent
    return ent[1], ent[2], ent[3]


def _split_hi_lo_weight(W1):
    """[C,c] fp32 (c = 6 or 8 input channels) -> bf16 [C,64] = [W_hi, W_hi, W_lo, 0]: partner of the x0 layout of
    p4p_apf_group."""
    C = W1.shape[0]
    w = W1.detach().reshape(C, -1)
    c = w.shape[1]
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    out = torch.zeros(C, 64, dtype=torch.bfloat16, device=w.device)
    out[:, 0:c], out[:, c:2 * c], out[:, 2 * c:3 * c] = hi, hi, lo
    return out


def split_hi_lo_input(pg):
    """[P,c] fp32 -> bf16 [P,64] = [x_hi, x_lo, x_hi, 0] (what p4p_apf_group writes directly on the fused path)."""
    c = pg.shape[1]
    hi = pg.bfloat16()
    lo = (pg - hi.float()).bfloat16()
    out = torch.zeros(pg.shape[0], 64, dtype=torch.bfloat16, device=pg.device)
    out[:, 0:c], out[:, c:2 * c], out[:, 2 * c:3 * c] = hi, lo, hi
    return out


class _ApfEncoderFn(torch.autograd.Function):
    """tokens f32 [B*G, E] from x0 bf16 [B*G*gs, 64]."""

    @staticmethod
    def forward(ctx, x0, G, gs, training, need_bwd, bns, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, W4, b4, g4, be4, W5, b5):
        bn1, bn2, bn4 = bns
        P = x0.shape[0]
        C1, C2, E = W1.shape[0], W2.shape[0], W3.shape[0]

        def bn(H, st, mod, bias):
            C = H.shape[1]
            sums = ops.colreduce(st, st.shape[0], 2 * C) if training else None
            sc, sh, mean, istd = ops.bn_finalize(sums, mod, P, training)
            sh = _bn_after_bias(mod, sc, sh, bias, training)
            return ops.bn_relu(H, sc, sh), sc, sh, mean, istd

        H1, st1 = ops.gemm_tn(x0, _split_hi_lo_weight(W1), ops.EPI_BF16_STATS)
        F1, sc1, sh1, mean1, is1 = bn(H1, st1, bn1, b1)
        W2b, W2T = _bf16_weights(W2, need_bwd)
        H2, st2 = ops.gemm_tn(F1, W2b, ops.EPI_BF16_STATS)
        F2, sc2, sh2, mean2, is2 = bn(H2, st2, bn2, b2)
        W3b, W3T = _bf16_weights(W3, need_bwd)
        f = ops.gemm_tn(F2, W3b, ops.EPI_BF16, bias=b3.detach())
        fg, arg3 = ops.group_max(f, G, gs, out_f32=False)                       # apf.py:160
        # second_conv[0] on cat([global.expand, local]) (apf.py:162-165): W4[:, :E] meets the pooled vector once per
        # group and enters the per-point GEMM as its accumulator init
        W4a, W4aT = _bf16_weights(W4, need_bwd, 0, E)
        W4b, W4bT = _bf16_weights(W4, need_bwd, E, E)
        U = ops.gemm_tn(fg, W4a, ops.EPI_F32_RESID)
        H4, st4 = ops.gemm_tn(f, W4b, ops.EPI_BF16_ROWGROUP, rowgrp=U, group=gs)
        F4, sc4, sh4, mean4, is4 = bn(H4, st4, bn4, b4)
        W5b, W5T = _bf16_weights(W5, need_bwd)
        H5 = ops.gemm_tn(F4, W5b, ops.EPI_BF16)
        tok, arg5 = ops.group_max(H5, G, gs, bias=b5.detach())                  # apf.py:167
        if training:
            for m in bns:
                m.num_batches_tracked += 1
        if _CAPTURE is not None:
            _CAPTURE.update(F1=F1, F2=F2, F4=F4, arg3=arg3, arg5=arg5)
        ctx.training, ctx.dims, ctx.cin = training, (G, gs, C1, C2, E), W1.shape[1]
        ctx.save_for_backward(x0, H1, F1, sc1, sh1, mean1, is1, H2, F2, sc2, sh2, mean2, is2, f, fg, arg3, H4, F4, sc4,
                              sh4, mean4, is4, arg5, W2T, W3T, W4aT, W4bT, W5T)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        if not ctx.training:
            raise RuntimeError('APF Encoder backward is implemented for train-mode BatchNorm only '
                               '(the reference never back-propagates in eval mode)')
        (x0, H1, F1, sc1, sh1, mean1, is1, H2, F2, sc2, sh2, mean2, is2, f, fg, arg3, H4, F4, sc4, sh4, mean4, is4, arg5,
         W2T, W3T, W4aT, W4bT, W5T) = ctx.saved_tensors
        G, gs, C1, C2, E = ctx.dims
        dtok = dtok.contiguous().float()
        db5 = dtok.sum(0)
        dH5 = ops.group_max_bwd(dtok, arg5, G, gs)
        dW5 = ops.gemm_wgrad(dH5, F4)
        dF4 = ops.gemm_tn(dH5, W5T, ops.EPI_BF16)
        del dH5
        s4 = ops.relu_bn_bwd_(dF4, H4, sc4, sh4, mean4, is4)                    # dF4 now holds dH4
        dW4b = ops.gemm_wgrad(dF4, f)
        df = ops.gemm_tn(dF4, W4bT, ops.EPI_BF16)
        dU = ops.group_sum(dF4, G, gs)
        del dF4
        dW4a = ops.gemm_wgrad(dU, fg)
        dfg = ops.gemm_tn(dU, W4aT, ops.EPI_BF16)
        ops.group_max_bwd(dfg, arg3, G, gs, into=df)                            # the pooled vector's share joins df
        # first_conv[6].bias moves f, its pooled copy and therefore H4 by one constant per channel, which the train-mode
        # BatchNorm behind removes: its gradient (the column sums of df) is zero in exact arithmetic -- no pass is spent on it
        dW3, db3 = ops.gemm_wgrad(df, F2), torch.zeros(E, dtype=torch.float32, device=dtok.device)
        dF2 = ops.gemm_tn(df, W3T, ops.EPI_BF16)
        del df
        s2 = ops.relu_bn_bwd_(dF2, H2, sc2, sh2, mean2, is2)
        dW2 = ops.gemm_wgrad(dF2, F1)
        dF1 = ops.gemm_tn(dF2, W2T, ops.EPI_BF16)
        del dF2
        s1 = ops.relu_bn_bwd_(dF1, H1, sc1, sh1, mean1, is1)
        dW1p = ops.gemm_wgrad(dF1, x0)                                          # [C1, 64]: hi and lo input columns
        cin = ctx.cin
        dW1 = dW1p[:, 0:cin] + dW1p[:, cin:2 * cin]
        zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=dtok.device)   # a bias in front of train-mode BN
        dW4 = torch.cat([dW4a, dW4b], dim=1)
        return (None, None, None, None, None, None,
                dW1.reshape(C1, cin, 1), zeros(C1), s1[1].clone(), s1[0].clone(),
                dW2.unsqueeze(-1), zeros(C2), s2[1].clone(), s2[0].clone(),
                dW3.unsqueeze(-1), db3,
                dW4.unsqueeze(-1), zeros(2 * E), s4[1].clone(), s4[0].clone(),
                dW5.unsqueeze(-1), db5)


def apf_encoder(x0, G, gs, training, enc):
    """enc = Encoder module (first_conv / second_conv Sequentials, apf.py:129-143); x0 from ops.apf_group."""
    fc, sc = enc.first_conv, enc.second_conv
    bns = (fc[1], fc[4], sc[1])
    need_bwd = bool(training and torch.is_grad_enabled() and any(p.requires_grad for p in enc.parameters()))
    return _ApfEncoderFn.apply(x0, G, gs, training, need_bwd, bns, fc[0].weight, fc[0].bias, fc[1].weight, fc[1].bias,
                               fc[3].weight, fc[3].bias, fc[4].weight, fc[4].bias, fc[6].weight, fc[6].bias,
                               sc[0].weight, sc[0].bias, sc[1].weight, sc[1].bias, sc[3].weight, sc[3].bias)


# ================================================================================================
# L x APFViTLayer (+ encoder_norm)
# ================================================================================================
_APF_BLOCK_PARAMS = 19   # n1w n1b qkvw qkvb projw projb n2w n2b fc1w fc1b fc2w fc2b | anw anb scale dw db uw ub


def apf_block_params(blk):
    a = blk.adapter
    return [blk.norm1.weight, blk.norm1.bias, blk.attention.qkv.weight, blk.attention.qkv.bias, blk.attention.proj.weight,
            blk.attention.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
            blk.mlp.fc2.weight, blk.mlp.fc2.bias, a.adapter_norm.weight, a.adapter_norm.bias, a.scale, a.down_proj.weight,
            a.down_proj.bias, a.up_proj.weight, a.up_proj.bias]


class _ApfBlocksFn(torch.autograd.Function):
    """x f32 [B,T,D] -> L blocks -> optional final LayerNorm.  Per block (apf_utils.py:277-293):
        x1  = x + keep_a/q * attention(norm1(x))
        r   = up(drop(relu(down(adapter_norm(x1))))) * scale + 2 * x1       (adapter output + the block's own residual)
        out = keep_m/q * mlp(norm2(x1)) + r
    keep_a / keep_m are the per-sample DropPath draws (bool [B], None = identity), q the keep probability; drop_masks[l] =
    (keep_a, keep_m, 1/q).  The up-projection GEMM writes r directly (two residual inputs), the proj / fc2 GEMMs add their
    residual in the epilogue with 1/q folded into the frozen weights; the dropped samples' rows are then put back
    (p4p_rows_restore_f32): no elementwise pass over the stream exists on this path."""

    @staticmethod
    def forward(ctx, x, num_heads, eps, train, drop_masks, adapter_p, norm_w, norm_b, *bp):
        B, T, D = x.shape
        M, H = B * T, num_heads
        L = len(bp) // _APF_BLOCK_PARAMS
        x = x.reshape(M, D).contiguous()
        saved, scal = [], []
        for l in range(L):
            (n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, anw, anb, ascale, dw, db, uw,
             ub) = bp[l * _APF_BLOCK_PARAMS:(l + 1) * _APF_BLOCK_PARAMS]
            keep_a, keep_m, c = drop_masks[l] if drop_masks is not None else (None, None, 1.0)
            xin = x
            ln1, _, mean1, rstd1 = ops.layernorm_fwd(xin, n1w.detach(), n1b.detach(), eps, want_stats=train)
            Wqkv, _ = _bf16_weights(qkvw, train)
            qkv = ops.gemm_tn(ln1, Wqkv, ops.EPI_BF16, bias=qkvb.detach())
            att, lse = ops.attention_fwd(qkv, B, T, H, want_lse=train)
            if keep_a is None:
                Wproj, _ = _bf16_weights(projw, train)
                x1 = ops.gemm_tn(att, Wproj, ops.EPI_F32_RESID, bias=projb.detach(), resid=xin)
            else:               # DropPath: weights carry 1/keep, dropped samples get their residual back
                Wproj, _, bproj = _scaled_operands(projw, projb, c)
                x1 = ops.rows_restore_(ops.gemm_tn(att, Wproj, ops.EPI_F32_RESID, bias=bproj, resid=xin), xin, keep_a, B)
            # ---- adapter (apf_utils.py:217-233)
            aln, _, meanA, rstdA = ops.layernorm_fwd(x1, anw.detach(), anb.detach(), eps, want_stats=train)
            Wd, _ = _bf16_weights(dw, train)
            pre = ops.gemm_tn(aln, Wd, ops.EPI_F32_RESID, bias=db.detach())        # [M, 64] f32
            down, dmask = ops.adapter_mid_fwd(pre, adapter_p)
            Wu, WuT, bu = _adapter_up_operands(uw, ub, ascale)                     # [D, 64] with `scale` folded in
            r = ops.gemm_tn(down, Wu, ops.EPI_F32_RESID, bias=bu, resid=x1, resid2=x1)
            # ---- mlp
            ln2, _, mean2, rstd2 = ops.layernorm_fwd(x1, n2w.detach(), n2b.detach(), eps, want_stats=train)
            Wfc1, _ = _bf16_weights(fc1w, train)
            if train:
                g, hpre = ops.gemm_tn(ln2, Wfc1, ops.EPI_BF16_GELU, bias=fc1b.detach(), want_pre=True)
            else:
                g, hpre = ops.gemm_tn(ln2, Wfc1, ops.EPI_BF16_GELU, bias=fc1b.detach()), None
            if keep_m is None:
                Wfc2, _ = _bf16_weights(fc2w, train)
                x = ops.gemm_tn(g, Wfc2, ops.EPI_F32_RESID, bias=fc2b.detach(), resid=r)
            else:
                Wfc2, _, bfc2 = _scaled_operands(fc2w, fc2b, c)
                x = ops.rows_restore_(ops.gemm_tn(g, Wfc2, ops.EPI_F32_RESID, bias=bfc2, resid=r), r, keep_m, B)
            if train:
                saved += [xin, mean1, rstd1, qkv, att, lse, x1, meanA, rstdA, dmask, WuT, mean2, rstd2, hpre]
                scal.append((keep_a, keep_m, c))
        if norm_w is not None:
            y, _, meanN, rstdN = ops.layernorm_fwd(x, norm_w.detach(), norm_b.detach(), eps, out_f32=True, want_stats=train)
        else:
            y, meanN, rstdN = x, None, None
        if train:
            nw = norm_w if norm_w is not None else x.new_empty(0)
            ctx.save_for_backward(x, meanN if meanN is not None else x.new_empty(0),
                                  rstdN if rstdN is not None else x.new_empty(0), nw, *bp, *saved)
        ctx.meta = (B, T, D, H, L, eps, norm_w is not None)
        ctx.scal = scal
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        B, T, D, H, L, eps, has_norm = ctx.meta
        M = B * T
        t = ctx.saved_tensors
        xN, meanN, rstdN, norm_w = t[:4]
        bp = t[4:4 + L * _APF_BLOCK_PARAMS]
        saved = t[4 + L * _APF_BLOCK_PARAMS:]
        if any(ctx.needs_input_grad[8:]):
            raise NotImplementedError('APFViTLayer weights are frozen in the reference (apf.py:335-346, adapters included); '
                                      'their weight gradients are not built')
        dev = dy.device
        dxb = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        dnw = dnb = None
        if has_norm:
            dx, dnw, dnb = ops.layernorm_bwd(dy.reshape(M, D).contiguous(), xN, meanN, rstdN, norm_w.detach(), dx_bf16=dxb)
        else:
            dx = dy.reshape(M, D).contiguous().float().clone()
            dxb.copy_(dx)

        NS = 14
        for l in reversed(range(L)):
            (n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b, anw, anb, ascale, dw, db, uw,
             ub) = bp[l * _APF_BLOCK_PARAMS:(l + 1) * _APF_BLOCK_PARAMS]
            xin, mean1, rstd1, qkv, att, lse, x1, meanA, rstdA, dmask, WuT, mean2, rstd2, hpre = saved[l * NS:(l + 1) * NS]
            keep_a, keep_m, c = ctx.scal[l]
            # ---- mlp branch
            if keep_m is None:
                _, Wfc2T = _bf16_weights(fc2w, True)
                dh = ops.gemm_tn(dxb, Wfc2T, ops.EPI_BF16_DGELU, pre=hpre)
            else:
                _, Wfc2T, _ = _scaled_operands(fc2w, fc2b, c)
                dh = ops.rows_zero_(ops.gemm_tn(dxb, Wfc2T, ops.EPI_BF16_DGELU, pre=hpre), keep_m, B)
            _, Wfc1T = _bf16_weights(fc1w, True)
            dln2 = ops.gemm_tn(dh, Wfc1T, ops.EPI_BF16)
            del dh
            # ---- adapter branch: r = up(down) + 2 x1
            ddown = ops.gemm_tn(dxb, WuT, ops.EPI_F32_RESID)                       # [M, 64] f32
            dpre = ops.adapter_mid_bwd(ddown, dmask)
            _, WdT = _bf16_weights(dw, True)
            daln = ops.gemm_tn(dpre, WdT, ops.EPI_BF16)
            ops.layernorm_bwd(dln2, x1, mean2, rstd2, n2w.detach(), dx=dx, dx_factor=2, want_params=False)     # r = ... + 2 x1
            ops.layernorm_bwd(daln, x1, meanA, rstdA, anw.detach(), dx=dx, dx_bf16=dxb, want_params=False)
            # ---- attention branch
            if keep_a is None:
                _, WprojT = _bf16_weights(projw, True)
                datt = ops.gemm_tn(dxb, WprojT, ops.EPI_BF16)
            else:
                _, WprojT, _ = _scaled_operands(projw, projb, c)
                datt = ops.rows_zero_(ops.gemm_tn(dxb, WprojT, ops.EPI_BF16), keep_a, B)
            dqkv = ops.attention_bwd(qkv, att, datt, lse, B, T, H)
            _, WqkvT = _bf16_weights(qkvw, True)
            dln1 = ops.gemm_tn(dqkv, WqkvT, ops.EPI_BF16)
            del dqkv, datt
            ops.layernorm_bwd(dln1, xin, mean1, rstd1, n1w.detach(), dx=dx, dx_bf16=dxb, want_params=False)
        return (dx.view(B, T, D), None, None, None, None, None, dnw, dnb, *([None] * (L * _APF_BLOCK_PARAMS)))


def apf_blocks(x, blocks, norm, num_heads, drop_masks=None, adapter_p=0.0):
    """blocks: iterable of APFViTLayer; norm: LayerNorm or None; drop_masks: per block (keep_attn, keep_mlp, 1/keep_prob) with bool (B,) draws, or None."""
    bp = []
    for blk in blocks:
        bp += apf_block_params(blk)
    eps = blocks[0].norm1.eps
    nw, nb = (norm.weight, norm.bias) if norm is not None else (None, None)
    train = bool(torch.is_grad_enabled() and (x.requires_grad or (nw is not None and nw.requires_grad)
                                              or any(p.requires_grad for p in bp)))
    return _ApfBlocksFn.apply(x, num_heads, eps, train, drop_masks, float(adapter_p), nw, nb, *bp)
