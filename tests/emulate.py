"""Test helper: a plain PyTorch fp32 statement of P3Embed's local MLP that rounds activations to bf16 at
the same places as the HIP pipeline (F1, H2, F2, H3 and the GEMM operands).  ReLU masks and max-pool
arg-max positions are then identical to the kernels', which separates kernel-logic errors (must agree
to ~1%) from the hard-gating flips that any bf16 activation storage causes against an fp32 oracle."""
import torch


ROUND = True          # False: every rounding becomes the identity -> the statements below ARE the fp32 oracle (CPU test)


def _r(x):
    """round to bf16, straight-through gradient"""
    if not ROUND:
        return x
    return x + (x.to(torch.bfloat16).float() - x).detach()


def _bn_train(h, gamma, beta, eps):
    mean = h.mean(0)
    var = h.var(0, unbiased=False)
    return mean, var, gamma / torch.sqrt(var + eps), beta - mean * gamma / torch.sqrt(var + eps)


LINK_LIMITS = {'bf16': 0.05,     # ulps beyond a correct rounding (after the fp32 summation slack, link_stats)
               'att': 3.0,       # flipped numerators of the row beyond half an output ulp (link_stats 'att'): the bound that holds for ALL
                                 # inputs (tests/tools/fuzz_kernels.py: 376 random shapes; the model's own activations measure 0.0-0.3);
                                 # rounds 2-3 had 0.6 x max(ulp, 2^-8 rms) here, a statistic of the suite's fixed seeds that one random
                                 # peaked softmax in ten exceeds through a single numerator rounded the other way
               'f32': 3e-5,      # relative to the tensor's rms
               'exact': 0.0}
NORTH_STAR_TOL = 1e-3            # BASELINE.json north_star: "logits within 1e-3"


def assert_links(oracle, cap, pts, idx_fps, idx_knn, logits, training=False, grad=False):
    """Runs the statement link by link against a capture of the pipeline's forward (functional._CAPTURE) and asserts
    every hand-off (LINK_LIMITS) and the logits (NORTH_STAR_TOL).  Returns (stated logits -- a function of the pipeline's
    own tensors carrying the statement's autograd graph --, worst value per link)."""
    tape = Tape(cap)
    with torch.set_grad_enabled(grad):
        stated = pix4point_emulated(oracle, pts, idx_fps, idx_knn, training=training, tape=tape)
    bad, worst = [], {}
    for name, got, want, kind, *unit in tape.links:
        v = link_stats(got, want, kind, *unit)
        worst[name] = v
        if not v <= LINK_LIMITS[kind]:
            bad.append((name, kind, v))
    err = float((logits.detach() - stated.detach()).abs().max())
    worst['logits'] = err
    if not err <= NORTH_STAR_TOL:
        bad.append(('logits', 'abs', err))
    assert not bad, bad
    assert len(tape.links) == 7 + 3 + 1 + 7 * len(cap['blocks']) + 1         # tokenizer, streams, xin, 7 per block, final norm
    return stated, worst


class Tape:
    """How the statements below hand a tensor from one stage to the next.

    Free running (``cap=None``): round to bf16 where the pipeline stores bf16.
    Link by link (``cap`` = functional._CAPTURE of a pipeline forward): record (name, pipeline tensor, stated value, kind)
    and CONTINUE FROM THE PIPELINE'S TENSOR -- the value is substituted, the autograd graph of the stated value is kept, so
    gradients are evaluated at exactly the pipeline's forward tensors (same ReLU masks, same arg-max picks)."""

    def __init__(self, cap=None):
        self.cap, self.links = cap, []

    def _captured(self, name):
        grp, key = name.split('.')
        if grp.startswith('block'):
            return self.cap['blocks'][int(grp[5:])][key]
        return self.cap[grp][key]

    def __call__(self, name, v, kind, unit=None):
        """kind: 'bf16' (stored bf16), 'att' (bf16, but with roundings of its own inside; unit = what one flipped numerator moves an
        element by, attention_flip_unit), 'f32', 'exact'."""
        if self.cap is None:
            return _r(v) if kind in ('bf16', 'att') else v
        got = (self.cap[name] if '.' not in name else self._captured(name)).detach().reshape(v.shape).float()
        self.links.append((name, got, v.detach(), kind) if unit is None else (name, got, v.detach(), kind, unit))
        return v + (got - v).detach()


def p3embed_emulated(points, idx, params, k, eps=1e-5, running=None, tape=None):
    """points (B,N,3) f32 cuda, idx (B,S,k) int32; params = dict of fp32 leaf tensors named like the module.
    Returns tokens [B*S, 256] (fp32) computed the way csrc/embed.hip + the GEMM epilogues do.  ``running`` (eval
    mode): dict name -> (running_mean, running_var) for '0.2', '1.1', '1.4' instead of the batch statistics."""
    tape = tape or Tape()
    B, N, _ = points.shape
    S = idx.shape[1]
    b = torch.arange(B, device=points.device).view(B, 1, 1)
    g = points[b, idx.long()]                                    # (B,S,k,3)
    x6 = torch.cat([g, g], dim=-1).reshape(B * S * k, 6)
    W1, W2, b2 = params['0.0.weight'].reshape(256, 6), params['0.1.weight'].reshape(256, 256), params['0.1.bias']

    def fold(h, name, bias=None):
        """(scale, shift) of the BatchNorm over the rows of h (csrc/embed.hip: embed_fold_kernel / bn_finalize_kernel)."""
        gamma, beta = params[name + '.weight'], params[name + '.bias']
        if running is None:
            mean, var = h.mean(0), h.var(0, unbiased=False)
            if bias is not None:
                mean = mean + bias
        else:
            mean, var = running[name]
        sc = gamma / torch.sqrt(var + eps)
        return sc, (beta - mean * sc) if bias is None else (beta + (bias - mean) * sc)

    aeff = W2 @ W1
    h1 = x6 @ aeff.t()                                           # the conv bias b2 lives in the folded shift
    sc1, sh1 = fold(h1, '0.2', b2)
    f1 = tape('embed.F1', torch.relu(h1 * sc1 + sh1), 'bf16')     # F1 stored bf16
    pooled = tape('embed.pooled', f1.view(B * S, k, 256).max(1).values, 'exact')
    W3 = params['1.0.weight'].reshape(512, 512)
    u = tape('embed.U', pooled @ _r(W3[:, :256]).t(), 'f32')
    h2 = f1 @ _r(W3[:, 256:]).t() + u.repeat_interleave(k, 0)    # BatchNorm statistics: from these fp32 accumulators
    sc2, sh2 = fold(h2, '1.1')
    f2 = tape('embed.F2', torch.relu(tape('embed.H2', h2, 'bf16') * sc2 + sh2), 'bf16')
    h3 = f2 @ _r(params['1.3.weight'].reshape(256, 512)).t()
    sc3, sh3 = fold(h3, '1.4')
    return tape('embed.tok', torch.relu(tape('embed.H3', h3, 'bf16') * sc3 + sh3).view(B * S, k, 256).max(1).values, 'f32')


# ------------------------------------------------------------------------------------------------
# Whole-model statement: every tensor the HIP pipeline stores in bf16 (GEMM operands: weights, LayerNorm outputs, qkv,
# the softmax numerators P, attention output, GELU output) is rounded here at the same place, everything else (residual
# stream, LayerNorm / softmax statistics, accumulators) is fp32 -- functional._EncoderFn, csrc/attention.hip,
# csrc/gemm_tn*.h.
#
# Two fp32 implementations of one bf16 pipeline cannot be compared END TO END below the bf16 noise floor: their summation
# orders differ at 1e-7, that flips ~1e-4 of the next bf16 roundings by one ulp (4e-3 relative), every flip moves a whole
# row of the following GEMM and flips ~5 % of its roundings, ... -- the difference grows like d -> sqrt(d * ulp) per
# rounding stage and saturates near the noise floor within one transformer block (measured: tests/tools/debug_matched.py,
# DESIGN.md section 2).  The whole-model check is therefore made LINK BY LINK (Tape with a capture): every tensor the
# pipeline hands to its next stage must be a correct bf16 rounding of the stated fp32 value computed from the pipeline's
# OWN upstream tensors (half an ulp + slack), fp32 hand-offs agree to fp32 accuracy, and the last link is the logits.
# A kernel-logic error above the slack shows up in exactly one link.
# ------------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634
KEY_TILE = 64            # csrc/attention.hip: keys are consumed in tiles of 64 rows, ascending
ONLINE_THR = 4.0         # csrc/attention.hip: ATT_LAZY_THR -- the reference m of a 16-query tile moves only when a query outgrew it by more than 2^4


def _exp2_fma(s, c, m):
    """exp2(fma(s, c, -m)) as the kernels evaluate it (one rounding of s*c - m), straight-through for autograd."""
    exact = (s.detach().double() * c - m.detach().double()).float()
    smooth = s * c - m
    return torch.exp2(smooth + (exact - smooth).detach())


def attention_lone_mode(T):
    """Which arithmetic contract p4p_attention_fwd follows for a sequence of T rows (csrc/attention.hip dispatch):
    'two_pass' = csrc/attention_lone.h (T = 64 n + 1 <= 257), 'online' = every other shape (attention.hip).
    (A streaming variant with the last row apart -- the last key as the first step of every query's online softmax, the
    last query as four chains of block 0 combined through LDS -- was built, verified against its own statement here and
    measured: 515 vs 490 us at T = 513, 4.30 vs 4.41 ms at T = 2049; not kept)"""
    return 'two_pass' if (T % 64 == 1 and 65 <= T <= 257) else 'online'


ONLINE32_THR = 4.0        # csrc/attention_fwd2.h: FW2_THR


def _attention_online32(s, v, c):
    """csrc/attention_fwd2.h (T = 513).  Rows 0 .. T-2: ONE online chain over the 17 key tiles of 32 in ascending order (the last tile
    holds key T-1 alone).  The last row: eight chains, wave w over key tiles 2 w, 2 w + 1 (wave 7: also the last tile), their
    (max, sum, accumulator) triples combined in wave order.  The reference m of a chain moves to the row maximum only when some query of
    its 32-row tile outgrew it by more than 2^ONLINE32_THR (numerators up to 2^thr; _online_chain)."""
    T = s.shape[-1]
    TL = T - 1
    groups = [list(range(k0, min(k0 + 32, T))) for k0 in range(0, T, 32)]
    _, lsum, o = _online_chain(s[:, :, :TL], v, c, groups, thr=ONLINE32_THR, rows=32)
    main = o * (1.0 / lsum)
    sl = s[:, :, TL:]
    parts = [_online_chain(sl, v, c, groups[2 * w:2 * w + 2] + ([groups[-1]] if w == 7 else []), thr=ONLINE32_THR, rows=1) for w in range(8)]
    M = torch.stack([m for m, _, _ in parts], 0).max(0).values
    l = sum(lw * torch.exp2(m - M) for m, lw, _ in parts)
    ol = sum(ow * torch.exp2(m - M) for m, _, ow in parts)
    return torch.cat([main, ol * (1.0 / l)], dim=2)


def _attention_two_pass(s, v, c):
    """csrc/attention_lone.h.  s [B,H,T,T] scores, v [B,H,T,hd]; returns the unrounded output [B,H,T,hd].
    Main query rows (all but the last): exact row maximum over ALL keys, numerators rounded to bf16 for the P.V product
    AND for the row sum (the sum comes out of the matrix pipe), no rescale.  The last row: the same per 64-key tile (tile
    0 also holds the last key), the per-tile (max, sum, accumulator) triples combined afterwards."""
    T = s.shape[-1]
    TL = T - 1
    sm = s[:, :, :TL, :]
    m = (sm.max(-1, keepdim=True).values * c).detach()
    pr = _r(_exp2_fma(sm, c, m))
    main = (pr @ v) * (1.0 / pr.sum(-1, keepdim=True))
    sl = s[:, :, TL:, :]                                              # [B,H,1,T]
    ms, os_, ls = [], [], []
    for k0 in range(0, TL, KEY_TILE):
        keys = list(range(k0, k0 + KEY_TILE)) + ([TL] if k0 == 0 else [])
        st = sl[..., keys]
        mt = (st.max(-1, keepdim=True).values * c).detach()
        pt = _r(_exp2_fma(st, c, mt))
        ms.append(mt), os_.append(pt @ v[:, :, keys]), ls.append(pt.sum(-1, keepdim=True))
    M = torch.stack(ms, 0).max(0).values
    o = sum(torch.exp2(mt - M) * ot for mt, ot in zip(ms, os_))
    l = sum(torch.exp2(mt - M) * lt for mt, lt in zip(ms, ls))
    return torch.cat([main, o * (1.0 / l)], dim=2)


def _online_chain(s, v, c, key_groups, m=None, lsum=None, o=None, thr=None, rows=None):
    """One online-softmax chain over the given key groups (lists of key indices), attention.hip: attn_fwd_tile: running
    maximum, numerators rounded to bf16 for P.V AND for the denominator (round 3: the row sums come out of the matrix pipe, one MFMA
    against an all-ones operand, as in attention_lone.h), accumulator rescaled per group.  A group of ONE key
    whose numerator is exactly 1 reproduces the 'last key first' start of attention_lone.h.
    thr / rows (attention_fwd2.h): the reference m moves to the running maximum only when some row of a block of `rows` consecutive query
    rows outgrew it by more than thr; then for every row of the block."""
    B, H, R, _ = s.shape
    if m is None:
        m = torch.full((B, H, R, 1), float('-inf'), device=s.device)
        lsum = torch.zeros(B, H, R, 1, device=s.device)
        o = torch.zeros(B, H, R, v.shape[-1], device=s.device)
    for keys in key_groups:
        st = s[..., keys]
        mx = (st.max(-1, keepdim=True).values * c).detach()
        mnew = torch.maximum(m, mx)
        if thr is not None:
            nb = -(-R // rows)
            g = torch.zeros(B, H, nb * rows, 1, dtype=torch.bool, device=s.device)
            g[:, :, :R] = mx > m + thr
            grew = g.view(B, H, nb, rows).any(-1, keepdim=True).expand(B, H, nb, rows).reshape(B, H, nb * rows, 1)[:, :, :R]
            mnew = torch.where(grew, mnew, m)
        alpha = torch.exp2(m - mnew)
        alpha = torch.where(mnew == m, torch.ones_like(alpha), alpha)      # (-inf) - (-inf): no key seen yet, nothing to scale
        p = _exp2_fma(st, c, mnew)
        lsum = lsum * alpha + _r(p).sum(-1, keepdim=True)
        o = o * alpha + _r(p) @ v[:, :, keys]
        m = mnew
    return m, lsum, o


def attention_emulated(qkv, B, T, H, round_output=True, mode=None):
    """qkv [B*T, 3*H*64] (bf16-valued fp32) -> attention output [B*T, H*64].  Two contracts, chosen as the library
    chooses its kernel (attention_lone_mode):
    'online'  (csrc/attention.hip: attn_fwd_tile): online softmax over 64-key tiles in ascending order: reference m in the base-2
              domain c*s (the running maximum, followed lazily: round 4, ONLINE_THR, decided per tile of 16 query rows), numerators
              p = exp2(c*s - m) ROUNDED TO bf16 for the P.V product and for the denominator alike, accumulator rescaled by
              exp2(m_old - m_new) when m moves, one division at the end;
    'two_pass' (csrc/attention_lone.h): see _attention_two_pass;
    'online32' (csrc/attention_fwd2.h, T = 513): see _attention_online32.
    The result is rounded to bf16 (round_output=False: the fp32 value in front of that rounding)."""
    hd = 64
    c = float(torch.tensor(hd ** -0.5, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32))
    q, k, v = qkv.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)          # [B,H,T,hd] each
    s = q @ k.transpose(-1, -2)                                        # [B,H,T,T] fp32 (exact bf16 products)
    mode = mode or attention_lone_mode(T)                              # 'online32': p4p_set_attention_fwd_variant(1), T = 513
    if mode == 'two_pass':
        out = _attention_two_pass(s, v, c)
    elif mode == 'online32':
        out = _attention_online32(s, v, c)
    else:
        _, lsum, o = _online_chain(s, v, c, [list(range(k0, min(k0 + KEY_TILE, T))) for k0 in range(0, T, KEY_TILE)],
                                   thr=ONLINE_THR, rows=16)
        out = o * (1.0 / lsum)
    if round_output:
        out = _r(out)
    return out.permute(0, 2, 1, 3).reshape(B * T, H * hd)


def attention_backward_emulated(qkv, o, dout, lse2, B, T, H, key_chunk=256):
    """The attention BACKWARD as every backward kernel of csrc/attention*.h evaluates it (one contract for the streaming, resident,
    fused and T = 513 kernels): with S = q k^T, c = scale log2(e) and the FORWARD's lse2 (teacher-forced: the kernels read it too),
        P  = exp2(fma(S, c, -lse2))           fp32          delta = rowsum(dO . O)   from the bf16 tensors, fp32
        dP = dO v^T - delta                   fp32          dS    = P dP             fp32 (the unrounded P)
        dV = round_bf16(P)^T dO               dK = scale round_bf16(dS)^T q          dQ = scale round_bf16(dS) k
    i.e. P and dS are rounded to bf16 exactly where they enter a matrix product, everything else is fp32 (scale = 2^-3 at head_dim 64
    commutes with the rounding).  qkv [B*T, 3D], o / dout [B*T, D] bf16-valued, lse2 [B,H,T].
    Returns (dqkv fp32 [B*T, 3D] BEFORE the final rounding to bf16, unit [B*T, 3D]): unit = what ONE numerator rounded the other way
    can move that element by, max over the contraction index of ulp(x_k) |y_k| (x = P or dS).  Two correct implementations differ by
    half an output ulp plus a few such flips (v_exp_f32 against torch.exp2, the summation order of delta and of the 64-term dot
    products in front of a rounding)."""
    hd = 64
    D = H * hd
    scale = hd ** -0.5
    c = float(torch.tensor(scale, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32))
    x = qkv.float().view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]                                          # [B,H,T,hd]
    do = dout.float().view(B, T, H, hd).permute(0, 2, 1, 3)
    oo = o.float().view(B, T, H, hd).permute(0, 2, 1, 3)
    s = q @ k.transpose(-1, -2)
    p = _exp2_fma(s, c, lse2.float().view(B, H, T, 1))
    delta = (do * oo).sum(-1, keepdim=True)
    dp = do @ v.transpose(-1, -2) - delta
    ds = p * dp
    pb, dsb = p.to(torch.bfloat16).float(), ds.to(torch.bfloat16).float()
    dq = (dsb @ k) * scale
    dk = (dsb.transpose(-1, -2) @ q) * scale
    dv = pb.transpose(-1, -2) @ do
    up, uds = bf16_ulp(p), bf16_ulp(ds)

    def maxprod(a, y):
        """[.., R, C] x [.., R, hd] -> [.., C, hd]: max over R of a[r, c] |y[r, d]| (chunked over C)"""
        outs = []
        for c0 in range(0, a.shape[-1], key_chunk):
            outs.append((a[..., :, c0:c0 + key_chunk].transpose(-1, -2).unsqueeze(-1) * y.abs().unsqueeze(-3)).amax(dim=-2))
        return torch.cat(outs, dim=-2)

    unit_q = maxprod(uds.transpose(-1, -2), k) * scale                  # rows = keys (contraction), columns = queries
    unit_k = maxprod(uds, q) * scale                                    # rows = queries (contraction), columns = keys
    unit_v = maxprod(up, do)
    pack = lambda t3: torch.stack(t3, 0).permute(1, 3, 0, 2, 4).reshape(B * T, 3 * D)      # [3,B,H,T,hd] -> [B*T, 3, H, hd]
    return pack((dq, dk, dv)), pack((unit_q, unit_k, unit_v))


def flips_beyond_rounding(got, want, unit):
    """How many single flipped numerators (units of `unit`) the worst element of a stored bf16 tensor is away from a correct rounding
    of the stated fp32 value `want`: max over elements of (|got - want| - ulp(want) / 2 - 2e-5 rms) / unit, never below 0."""
    got, want = got.detach().float(), want.detach().float()
    d = (got - want).abs()
    rms = float(want.pow(2).mean().sqrt())
    excess = (d - 0.5 * bf16_ulp(want) - 2e-5 * rms).clamp_min(0)
    return float((excess / unit.clamp_min(1e-30)).max())


def _ln(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _gelu(x):
    return torch.nn.functional.gelu(x)          # exact erf, as the GEMM epilogue (csrc/gemm_tn.h: gelu_f)


def encoder_emulated(feats, pos, blocks, norm, H, tape=None):
    """feats, pos [B,T,D] fp32; blocks: modules with timm's Block layout (norm1, attn.qkv, attn.proj, norm2, mlp.fc1,
    mlp.fc2); for every block x = blk(x + pos), then the final LayerNorm (functional._EncoderFn: the fc2 epilogue of
    block l already adds pos for block l+1)."""
    tape = tape or Tape()
    B, T, D = feats.shape
    x, pos2 = feats.reshape(B * T, D), pos.reshape(B * T, D)
    L = len(blocks)
    for l, blk in enumerate(blocks):
        xin = tape(f'block{l}.xin', x + pos2, 'f32') if l == 0 else x
        x = block_emulated(xin, blk, B, T, H, tape, l)
        x = tape(f'block{l}.xout', x + pos2 if l + 1 < L else x, 'f32')
    return tape('encoder_out', _ln(x, norm.weight, norm.bias, norm.eps), 'f32').view(B, T, D)


class _AttentionMatchedBackward(torch.autograd.Function):
    """attention_emulated in the forward; in the backward the kernels' own contract (attention_backward_emulated: P and dS rounded to
    bf16 where they enter a product, delta from the stored bf16 output, the forward's lse2) instead of the exact fp32 gradient.  The
    two differ by the rounding of dS, which the sum over the keys amplifies wherever the key rows share a large common component
    (sum_k dS_k = 0 cancels it in exact arithmetic, not in the rounding noise): 4-8 % on the qkv / norm1 gradients of a randomly
    initialised ViT-B -- a property of ANY bf16 flash-style backward, not of these kernels."""

    @staticmethod
    def forward(ctx, qkv, B, T, H, o_stored, lse2):
        ctx.save_for_backward(qkv.detach(), o_stored.detach(), lse2.detach())
        ctx.dims = (B, T, H)
        with torch.no_grad():
            return attention_emulated(qkv.detach(), B, T, H, round_output=False)

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse2 = ctx.saved_tensors
        B, T, H = ctx.dims
        with torch.no_grad():
            dqkv, _ = attention_backward_emulated(qkv, o, dout.to(torch.bfloat16).float(), lse2, B, T, H)
        return dqkv.to(torch.bfloat16).float(), None, None, None, None, None      # the pipeline stores dqkv in bf16


def block_emulated(xin, blk, B, T, H, tape, l=0, matched_attention_backward=None):
    """One transformer block (timm Block layout) on xin [B*T, D] = the block's input INCLUDING the positional embedding; returns
    x1 + fc2(gelu(fc1(ln2(x1)))) with x1 = xin + proj(attention(qkv(ln1(xin)))), hand-offs named block{l}.* on the tape.
    matched_attention_backward = (stored attention output bf16, lse2 [B,H,T]) of the pipeline: autograd then runs the attention
    backward in the kernels' contract (_AttentionMatchedBackward) instead of the exact fp32 gradient."""
    ln1 = tape(f'block{l}.ln1', _ln(xin, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps), 'bf16')
    qkv = tape(f'block{l}.qkv', ln1 @ _r(blk.attn.qkv.weight).t() + blk.attn.qkv.bias, 'bf16')
    if matched_attention_backward is not None:
        att = tape(f'block{l}.att', _AttentionMatchedBackward.apply(qkv, B, T, H, *matched_attention_backward), 'att', unit=attention_flip_unit(qkv, B, T, H))
    else:
        att = tape(f'block{l}.att', attention_emulated(qkv, B, T, H, round_output=False), 'att', unit=attention_flip_unit(qkv, B, T, H))
    x1 = tape(f'block{l}.x1', att @ _r(blk.attn.proj.weight).t() + blk.attn.proj.bias + xin, 'f32')
    ln2 = tape(f'block{l}.ln2', _ln(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), 'bf16')
    g = tape(f'block{l}.g', _gelu(ln2 @ _r(blk.mlp.fc1.weight).t() + blk.mlp.fc1.bias), 'bf16')
    return g @ _r(blk.mlp.fc2.weight).t() + blk.mlp.fc2.bias + x1


def pix4point_emulated(oracle, pts, idx_fps, idx_knn, training=False, tape=None):
    """Logits of oracle (an oracle.pix4point_oracle.Pix4Point ON THE SAME DEVICE as pts) evaluated the way the HIP
    pipeline evaluates them.  idx_fps (B,S) int64 / idx_knn (B,S,k) int32 come from the oracle's index functions (the
    kernels are bit-exact on those: tests/test_gpu_01_kernels.py).  training=True: batch statistics in the tokenizer's and
    the head's BatchNorm (set oracle.train() and the dropout probabilities to 0 first).  tape: see Tape."""
    tape = tape or Tape()
    pv = oracle.model
    B, N, _ = pts.shape
    S, k = idx_fps.shape[1], idx_knn.shape[2]
    centers = torch.gather(pts, 1, idx_fps.unsqueeze(-1).expand(-1, -1, 3))
    convs = pv.patch_embed.convs[0]
    params = {f'{i}.{n}': p for i in (0, 1) for n, p in convs[i].named_parameters()}
    running = None
    if not training:
        running = {f'{i}.{j}': (convs[i][j].running_mean, convs[i][j].running_var) for i, j in ((0, 2), (1, 1), (1, 4))}
    tok = p3embed_emulated(pts, idx_knn, params, k, eps=convs[0][2].eps, running=running, tape=tape)          # [B*S, 256]
    xt = _r(tok) @ _r(pv.proj.weight).t() + pv.proj.bias
    h = _gelu(centers.reshape(B * S, 3) @ pv.pos_embed[0].weight.t() + pv.pos_embed[0].bias)       # fp32 (K = 3: no MFMA)
    h = tape('streams.h', h, 'f32')
    pe = _r(h) @ _r(pv.pos_embed[2].weight).t() + pv.pos_embed[2].bias
    D = xt.shape[1]
    feats = tape('streams.feats', torch.cat([pv.cls_token.expand(B, -1, -1), xt.view(B, S, D)], dim=1), 'f32')
    pos = tape('streams.pos', torch.cat([pv.cls_pos.expand(B, -1, -1), pe.view(B, S, D)], dim=1), 'f32')
    heads = pv.vit_blocks[0].attn.num_heads
    x = encoder_emulated(feats, pos, list(pv.vit_blocks), pv.norm, heads, tape=tape)
    glob = torch.cat([x[:, 1:, :].max(dim=1)[0], x[:, 0, :]], dim=1)
    return oracle.cls_head(glob)


def bf16_ulp(v):
    """spacing of bf16 numbers at |v| (8 significand bits)"""
    return torch.exp2(torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -100))) - 7)


def attention_flip_unit(qkv, B, T, H):
    """[B*T, H*64]: what ONE softmax numerator rounded to bf16 the other way can move an attention output element by -- 2^-9 times the
    row's largest probability times the largest |v| of the column (tests/tools/fuzz_kernels.py: every output of 376 random shapes within
    half an ulp + three of these)."""
    with torch.no_grad():
        q, k, v = qkv.detach().float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        pmax = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1).amax(-1)              # (B,H,T)
        vmax = v.abs().amax(2, keepdim=True)                                            # (B,H,1,64)
        return (2.0 ** -9) * (pmax.unsqueeze(-1) * vmax).permute(0, 2, 1, 3).reshape(B * T, H * 64)


def link_stats(got, want, kind, unit=None):
    """One number per link, to be bounded by the caller.
    'bf16': got is a stored bf16 tensor, want the stated fp32 value before rounding -> worst excess over half an ulp, in
            ulps, after an absolute slack of 2e-5 x the tensor's rms (fp32 summation order on cancelling sums: 200 times
            below the bf16 spacing at the rms);
    'att' : attention output: its roundings of P happen inside the kernel, so one flipped P moves an output by up to
            2^-8 p |v|: with `unit` (attention_flip_unit) the number of flipped numerators of the row beyond half an output ulp;
    'f32' : worst |difference| relative to the tensor's rms;  'exact': number of differing elements."""
    got, want = got.detach().float(), want.detach().float()
    d = (got - want).abs()
    rms = float(want.pow(2).mean().sqrt())
    if kind == 'exact':
        return float((d > 0).sum())
    if kind == 'f32':
        return float(d.max()) / max(rms, 1e-30)
    if kind == 'att':
        if unit is None:                    # (rounds 2-3: worst difference in units of max(ulp, 2^-8 rms): a seed-dependent statistic)
            return float((d / torch.clamp(bf16_ulp(want), min=rms * 2.0 ** -8)).max())
        return float(((d - 0.5 * bf16_ulp(want) - 1e-6).clamp_min(0) / unit.clamp_min(1e-30)).max())
    excess = (d - 2e-5 * rms - 0.5 * bf16_ulp(want)).clamp_min(0) / bf16_ulp(want)
    return float(excess.max())


def apf_encoder_emulated(x6, params, G, gs, eps=1e-5, gates=None):
    """x6 [G*gs, 6] f32; params = dict of fp32 leaf tensors named like Encoder's state dict (first_conv.N.* /
    second_conv.N.*).  Tokens [G, E] computed the way functional_apf._ApfEncoderFn does: bf16 storage of H1, F1, H2, F2,
    f, H4, F4, H5 and of the GEMM weight operands; BatchNorm statistics from the fp32 accumulators; conv biases in front
    of a train-mode BatchNorm dropped (they cancel).  ``gates`` (functional_apf._CAPTURE of the kernel run) replaces every
    ReLU / arg-max decision by the kernels' own, so that the two backward passes differ by arithmetic only: with bf16
    storage in front of a 32-way max, one-ulp differences would otherwise move ~0.3 % of the arg-max picks."""
    p = params
    E = p['first_conv.6.weight'].shape[0]
    w = lambda n: p[n].reshape(p[n].shape[0], -1)

    def bn_relu(h, pre, key):
        _, _, sc, sh = _bn_train(h, p[pre + '.weight'], p[pre + '.bias'], eps)
        z = _r(h) * sc + sh
        return _r(torch.relu(z) if gates is None else z * (gates[key] > 0).float())

    def pool(h, key):
        h = h.view(G, gs, E)
        return h.max(1).values if gates is None else h.gather(1, gates[key].long().unsqueeze(1)).squeeze(1)

    f1 = bn_relu(x6 @ w('first_conv.0.weight').t(), 'first_conv.1', 'F1')      # hi/lo split: fp32-class product
    f2 = bn_relu(f1 @ _r(w('first_conv.3.weight')).t(), 'first_conv.4', 'F2')
    f = _r(f2 @ _r(w('first_conv.6.weight')).t() + p['first_conv.6.bias'])
    fg = pool(f, 'arg3')
    W4 = w('second_conv.0.weight')
    u = fg @ _r(W4[:, :E]).t()
    f4 = bn_relu(f @ _r(W4[:, E:]).t() + u.repeat_interleave(gs, 0), 'second_conv.1', 'F4')
    h5 = _r(f4 @ _r(w('second_conv.3.weight')).t())
    return pool(h5, 'arg5') + p['second_conv.3.bias']
