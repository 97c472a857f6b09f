"""Tensor-level wrappers over the C ABI: argument checks mirror the errors torch would raise in the
reference (shape / dtype / device / contiguity), outputs are allocated with torch's caching
allocator on the input's device, kernels are enqueued on torch's current HIP stream."""
import torch

from . import _lib as L

EPI_BF16, EPI_BF16_GELU, EPI_F32_RESID, EPI_BF16_DGELU, EPI_BF16_ROWGROUP, EPI_BF16_STATS = range(6)


try:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice

    def _stream():
        # the raw handle of torch's current HIP stream on the current device (torch.cuda.current_stream() builds a Stream object
        # through four Python frames: 9 us per call, ~200 calls per step)
        return L.ctypes.c_void_p(_raw_stream(_cur_device()))
except AttributeError:                                       # a torch that renamed the private hooks: the public, slower way
    def _stream():
        return L.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=None, dims=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name}: expected a HIP device tensor, got {t.device} (this path has no CPU fallback)')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'{name}: expected dtype {dtype}, got {t.dtype}')
    if dims is not None and t.dim() != dims:
        raise RuntimeError(f'{name}: expected {dims} dims, got shape {tuple(t.shape)}')
    if not t.is_contiguous():
        raise RuntimeError(f'{name}: expected a contiguous tensor')
    return t


# ---------------------------------------------------------------- tokenizer index ops
def fps(points, n_samples, start_idx=None):
    """farthest_point_sampling (reference src/models/pix4point.py:8-53) -> int64 (B, min(n,N))."""
    _chk(points, 'points', torch.float32, 3)
    B, N, D = points.shape
    if D != 3:
        raise RuntimeError(f'points: last dim must be 3 (xyz), got {D}')
    S = min(int(n_samples), N)
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long, device=points.device)
    _chk(start_idx, 'start_idx', torch.int64, 1)
    if start_idx.shape[0] != B:
        raise RuntimeError('start_idx: expected shape (B,)')
    out = torch.empty(B, S, dtype=torch.int64, device=points.device)
    L.call('p4p_fps_f32', L.ptr(points), L.ptr(start_idx), L.ptr(out), B, N, S, _stream())
    return out


def knn(points, centroids, k):
    """int32 (B,S,k) nearest-point indices (reference src/models/pix4point.py:79-89)."""
    _chk(points, 'points', torch.float32, 3)
    _chk(centroids, 'centroids', torch.float32, 3)
    B, N, _ = points.shape
    if points.shape[2] != 3 or centroids.shape[2] != 3 or centroids.shape[0] != B:
        raise RuntimeError('knn: expected points (B,N,3) and centroids (B,S,3)')
    S = centroids.shape[1]
    out = torch.empty(B, S, int(k), dtype=torch.int32, device=points.device)
    L.call('p4p_knn_f32', L.ptr(points), L.ptr(centroids), L.ptr(out), B, N, S, int(k), _stream())
    return out


def group_gather(src, idx):
    """out[b,s,j,:] = src[b, idx[b,s,j], :] (reference src/models/pix4point.py:94-100)."""
    _chk(src, 'src', torch.float32, 3)
    _chk(idx, 'idx', torch.int32, 3)
    B, N, C = src.shape
    _, S, k = idx.shape
    out = torch.empty(B, S, k, C, dtype=torch.float32, device=src.device)
    L.call('p4p_group_gather_f32', L.ptr(src), L.ptr(idx), L.ptr(out), B, N, S, k, C, _stream())
    return out


# ---------------------------------------------------------------- GEMMs
def gemm_tn(a, w, epilogue=EPI_BF16, bias=None, resid=None, pre=None, rowgrp=None, group=1, out=None,
            want_stats=False, want_pre=False, want_colsum=False, resid2=None):
    """out[M,N] = a[M,K] @ w[N,K]^T with a fused epilogue (see include/p4p_hip.h).  want_colsum (EPI_BF16_DGELU
    only) also returns the column sums of out = the bias gradient of the Linear that produced `pre`."""
    _chk(a, 'a', torch.bfloat16, 2)
    _chk(w, 'w', torch.bfloat16, 2)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f'gemm_tn: a is {tuple(a.shape)}, w is {tuple(w.shape)}')
    f32_out = epilogue == EPI_F32_RESID
    if resid2 is not None:                         # second fp32 residual: travels in the rowgrp slot of the C ABI
        if not f32_out or resid is None or rowgrp is not None:
            raise RuntimeError('gemm_tn: resid2 needs EPI_F32_RESID with resid')
        _chk(resid2, 'resid2', torch.float32)
        rowgrp = resid2
    if f32_out and group > 1:                      # token-stream rows: the caller provides the [B*(S+1), N] buffer
        if out is None or resid is not None or M % group or out.shape != (M // group * (group + 1), N):
            raise RuntimeError('gemm_tn: EPI_F32_RESID with group > 1 needs out [M/group*(group+1), N] and no resid')
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if f32_out else torch.bfloat16, device=a.device)
    stats = None
    if epilogue in (EPI_BF16_ROWGROUP, EPI_BF16_STATS):
        rows = L.call_int('p4p_gemm_tn_stats_rows', M)
        stats = torch.empty(rows, 2, N, dtype=torch.float32, device=a.device)
    if epilogue == EPI_BF16_GELU and want_pre and pre is None:
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    cs_rows = 0
    if want_colsum:
        if epilogue != EPI_BF16_DGELU:
            raise RuntimeError('gemm_tn: want_colsum is only built for EPI_BF16_DGELU')
        cs_rows = L.call_int('p4p_gemm_tn_colsum_rows', M)
        stats = _workspace('dgelu_colsum', cs_rows * N, a.device)
    for t, nm, dt in ((bias, 'bias', torch.float32), (resid, 'resid', torch.float32),
                      (pre, 'pre', torch.bfloat16), (rowgrp, 'rowgrp', torch.float32)):
        if t is not None:
            _chk(t, nm, dt)
    L.call('p4p_gemm_tn_bf16', int(epilogue), L.ptr(a), L.ptr(w), L.ptr(out), L.ptr(bias), L.ptr(resid),
           L.ptr(pre), L.ptr(rowgrp), L.ptr(stats), M, N, K, int(group), _stream())
    if want_colsum:
        return out, colreduce(stats, cs_rows, N)
    if stats is not None:
        return out, stats
    if epilogue == EPI_BF16_GELU and want_pre:
        return out, pre
    return out


_wgrad_ws = {}


def _workspace(key, numel, device, dtype=torch.float32):
    """A persistent scratch buffer per (purpose, device, dtype, STREAM): kernels of one stream run in order, so reusing the buffer
    behind the previous call on that stream is safe; two streams never share one (the buffers used to be keyed
    without the stream).  A buffer that grows is replaced; the old one is freed by torch's stream-ordered allocator."""
    k = (key, device, dtype, _stream().value)
    t = _wgrad_ws.get(k)
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel), dtype=dtype, device=device)
        _wgrad_ws[k] = t
    return t


class _SlabSum(L.ctypes.Structure):          # include/p4p_hip.h: the record of p4p_sum_slabs_multi
    _fields_ = [('slabs', L.ctypes.c_void_p), ('out', L.ctypes.c_void_p), ('n', L.ctypes.c_longlong),
                ('nslabs', L.ctypes.c_int), ('beta', L.ctypes.c_float)]


def wgrad_flush(defer):
    """Sum the slabs of every weight gradient queued by gemm_wgrad(..., defer=queue) in one launch and empty the queue; the
    `out` tensors those calls returned are valid afterwards (same bits as the undeferred call)."""
    if not defer:
        return
    recs = (_SlabSum * len(defer))()
    for r, (ws, out, n, nslabs, beta) in zip(recs, defer):
        r.slabs, r.out, r.n, r.nslabs, r.beta = ws.data_ptr(), out.data_ptr(), n, nslabs, beta
    L.call('p4p_sum_slabs_multi', recs, len(defer), _stream())
    defer.clear()


def gemm_wgrad(g, a, out=None, beta=0.0, want_bias=False, bias_out=None, defer=None):
    """dW[N,K] f32 = beta*dW + g[M,N]^T @ a[M,K]; with want_bias also returns the column sums of g
    (the bias gradient), produced by the same kernel.  defer: a list -- the result is complete only after wgrad_flush(list)."""
    _chk(g, 'g', torch.bfloat16, 2)
    _chk(a, 'a', torch.bfloat16, 2)
    M, N = g.shape
    K = a.shape[1]
    if a.shape[0] != M:
        raise RuntimeError('gemm_wgrad: row counts differ')
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=g.device)
        beta = 0.0
    _chk(out, 'out', torch.float32, 2)
    if defer is not None and not want_bias:
        # slab sum deferred: the GEMM leaves its M-split slabs in a workspace of its own (one per queue position, persistent),
        # wgrad_flush(defer) sums the slabs of everything queued in ONE launch (a ViT block's four matrices: 4 launches -> 1)
        nslabs = L.call_int('p4p_wgrad_slab_count', M, N, K)
        if nslabs > 0 and len(defer) < 8:
            ws = _workspace(('wgrad_deferred', len(defer)), nslabs * N * K, g.device)
            L.call('p4p_gemm_wgrad_slabs_bf16', L.ptr(g), L.ptr(a), L.ptr(ws), M, N, K, _stream())
            defer.append((ws, out, N * K, nslabs, float(beta)))
            return out
    splits = L.call_int('p4p_wgrad_splits', M, N, K)
    ws = _workspace('wgrad', splits * (N * K + N), g.device)
    if want_bias and bias_out is None:
        bias_out = torch.empty(N, dtype=torch.float32, device=g.device)
        assert beta == 0.0
    L.call('p4p_gemm_wgrad_bf16', L.ptr(g), L.ptr(a), L.ptr(out), L.ptr(bias_out), L.ptr(ws), M, N, K, float(beta),
           _stream())
    return (out, bias_out) if want_bias else out


# ---------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps=1e-6, pos=None, want_sum=False, out_f32=False, want_stats=True):
    _chk(x, 'x', torch.float32, 2)
    M, D = x.shape
    _chk(gamma, 'gamma', torch.float32, 1)
    _chk(beta, 'beta', torch.float32, 1)
    if pos is not None:
        _chk(pos, 'pos', torch.float32, 2)
    y = torch.empty(M, D, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    xsum = torch.empty_like(x) if (pos is not None and want_sum) else None
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    L.call('p4p_layernorm_fwd', L.ptr(x), L.ptr(pos), L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(xsum), L.ptr(y),
           int(out_f32), L.ptr(mean), L.ptr(rstd), M, D, _stream())
    return y, xsum, mean, rstd


def layernorm_bwd(dy, xin, mean, rstd, gamma, dx=None, dgamma=None, dbeta=None, beta_acc=0.0, dx_bf16=None,
                  dacc=None, dacc_init=False, want_dxsum=False, dx_factor=1, want_params=True, dx_in=None):
    """Returns (dx, dgamma, dbeta); if dx is given the LN gradient is ADDED to dx_factor * dx in place (dx_factor 1 or 2) -- or,
    with dx_in, dx = dx_factor * dx_in + the LN gradient (dx written, dx_in left alone).
    dx_bf16 gets a bf16 copy of the result, dacc accumulates it (positional-embedding gradient)."""
    M, D = xin.shape
    _chk(dy, 'dy', None, 2)
    if dy.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('layernorm_bwd: dy must be f32 or bf16')
    if dx_in is not None:
        _chk(dx_in, 'dx_in', torch.float32, 2)
        if dx is None or dx_in.shape != dx.shape or dx.shape != xin.shape:
            raise RuntimeError('layernorm_bwd: dx_in needs a dx of the same shape to write to')
    acc = dx is not None
    if dx_factor not in (1, 2):
        raise RuntimeError('layernorm_bwd: dx_factor must be 1 or 2')
    if dx is None:
        dx = torch.empty(M, D, dtype=torch.float32, device=xin.device)
    dxsum = None
    if not want_params:                   # frozen LayerNorm (APF blocks): no parameter gradients, no reduce launch
        dgamma = dbeta = None
    elif dgamma is None:
        dgb = torch.empty(3, D, dtype=torch.float32, device=xin.device)     # contiguous: one reduce launch
        dgamma, dbeta, dxsum = dgb[0], dgb[1], dgb[2]
        beta_acc = 0.0
    rows = L.call_int('p4p_layernorm_bwd_partial_rows')
    part = _workspace('lnbwd', rows * 3 * D, xin.device)
    L.call('p4p_layernorm_bwd', L.ptr(dy), int(dy.dtype == torch.float32), L.ptr(xin), L.ptr(mean), L.ptr(rstd),
           L.ptr(gamma), L.ptr(dx), L.ptr(dx_in), int(dx_factor) if acc else 0, L.ptr(dx_bf16), L.ptr(dacc), int(dacc_init), L.ptr(part), L.ptr(dgamma),
           L.ptr(dbeta), L.ptr(dxsum), float(beta_acc), M, D, _stream())
    if want_dxsum:
        return dx, dgamma, dbeta, dxsum
    return dx, dgamma, dbeta


def sum_slabs(slabs):
    """slabs f32 (S, ...) contiguous -> their sum over S in the order 0, 1, .. (one pass)."""
    _chk(slabs, 'slabs', torch.float32, None)
    S = slabs.shape[0]
    n = slabs[0].numel()
    if n % 4:
        raise RuntimeError('sum_slabs: slab size must be a multiple of 4')
    out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    L.call('p4p_sum_slabs_f32', L.ptr(slabs), L.ptr(out), L.i64(n), int(S), _stream())
    return out


# ---------------------------------------------------------------- attention
def attention_fwd(qkv, B, T, H, want_lse=True):
    _chk(qkv, 'qkv', torch.bfloat16, 2)
    D = qkv.shape[1] // 3
    if qkv.shape[0] != B * T or qkv.shape[1] != 3 * D:
        raise RuntimeError('attention_fwd: qkv must be (B*T, 3D)')
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device) if want_lse else None
    L.call('p4p_attention_fwd', L.ptr(qkv), L.ptr(out), L.ptr(lse), B, T, H, D, float((D // H) ** -0.5), _stream())
    return out, lse


def attention_bwd(qkv, o, dout, lse, B, T, H, want_colsum=False):
    """dqkv; with want_colsum also the column sums of dqkv (the qkv bias gradient), accumulated per 64-row block
    inside the backward kernels and reduced here (no extra pass over dqkv)."""
    _chk(qkv, 'qkv', torch.bfloat16, 2)
    _chk(o, 'o', torch.bfloat16, 2)
    _chk(dout, 'dout', torch.bfloat16, 2)
    _chk(lse, 'lse', torch.float32, 3)
    D = o.shape[1]
    dqkv = torch.empty_like(qkv)
    # per-call workspace from torch's caching allocator (stream-ordered): delta, or the dQ^T parking space of the T = 513 kernel
    delta = torch.empty(L.call_int('p4p_attention_bwd_workspace', B, T, H), dtype=torch.float32, device=qkv.device)
    part = None
    if want_colsum:
        rows = L.call_int('p4p_attention_colsum_rows', B, T)
        part = _workspace('attn_colsum', rows * 3 * D, qkv.device)
    L.call('p4p_attention_bwd', L.ptr(qkv), L.ptr(o), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), L.ptr(part),
           B, T, H, D, float((D // H) ** -0.5), _stream())
    if want_colsum:
        return dqkv, colreduce(part, rows, 3 * D)
    return dqkv


# ---------------------------------------------------------------- streaming helpers
def cast_weight(w, want=True, want_t=False, rows=None, cols=None, col0=0):
    """fp32 weight (2-D view, possibly a column slice [:, col0:col0+cols]) -> (bf16 W, bf16 W^T)."""
    w2 = w.reshape(w.shape[0], -1)
    R = w2.shape[0] if rows is None else rows
    C = (w2.shape[1] - col0) if cols is None else cols
    _chk(w2, 'w', torch.float32, 2)
    dst = torch.empty(R, C, dtype=torch.bfloat16, device=w.device) if want else None
    dst_t = torch.empty(C, R, dtype=torch.bfloat16, device=w.device) if want_t else None
    src_ptr = L.ctypes.c_void_p(w2.data_ptr() + 4 * col0)
    L.call('p4p_cast_transpose_f32_bf16', src_ptr, L.i64(w2.shape[1]), L.ptr(dst), L.ptr(dst_t), R, C, _stream())
    return dst, dst_t


def cast_weights_multi(desc_table, n, total_tiles):
    """One launch over a device table of cast records (include/p4p_hip.h: p4p_cast_transpose_multi)."""
    _chk(desc_table, 'desc_table', torch.uint8, 1)
    if desc_table.numel() != 48 * n:
        raise RuntimeError('cast_weights_multi: the table must hold n records of 48 bytes')
    L.call('p4p_cast_transpose_multi', L.ptr(desc_table), int(n), int(total_tiles), _stream())


def cast_bf16(x):
    _chk(x, 'x', torch.float32)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.call('p4p_cast_f32_bf16', L.ptr(x), L.ptr(out), L.i64(x.numel()), _stream())
    return out


def colsum(g, out=None, beta=0.0):
    _chk(g, 'g', torch.bfloat16, 2)
    M, N = g.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=g.device)
        beta = 0.0
    ws = _workspace('colsum', L.call_int('p4p_colsum_splits', M) * N, g.device)
    L.call('p4p_colsum_bf16', L.ptr(g), L.ptr(out), L.ptr(ws), M, N, float(beta), _stream())
    return out


def token_rows_to_compact(stream, B, S, D):
    out = torch.empty(B * S, D, dtype=torch.bfloat16, device=stream.device)
    L.call('p4p_token_rows', L.ptr(stream), L.ptr(out), B, S, D, 0, _stream())
    return out


def compact_to_token_rows(stream, compact, B, S, D):
    _chk(compact, 'compact', torch.float32)
    L.call('p4p_token_rows', L.ptr(stream), L.ptr(compact), B, S, D, 1, _stream())


def cls_rows_set(stream, vec, B, T, D):
    L.call('p4p_cls_rows', L.ptr(stream), L.ptr(vec), B, T, D, 1, _stream())


def cls_rows_sum(stream, B, T, D):
    out = torch.empty(D, dtype=torch.float32, device=stream.device)
    L.call('p4p_cls_rows', L.ptr(stream), L.ptr(out), B, T, D, 0, _stream())
    return out


def pos_mlp0_fwd(x, W, b):
    """GELU(x[M,3] @ W[C,3]^T + b) -> f32 [M, C]."""
    _chk(x, 'x', torch.float32, 2)
    M, C = x.shape[0], W.shape[0]
    h = torch.empty(M, C, dtype=torch.float32, device=x.device)
    L.call('p4p_pos_mlp0_fwd', L.ptr(x), L.ptr(W), L.ptr(b), L.ptr(h), 0, L.i64(M), C, _stream())
    return h


def pos_mlp0_bwd(x, W, b, dh):
    """-> (dW [C,3], db [C])."""
    M, C = x.shape[0], W.shape[0]
    rows = L.call_int('p4p_pos_mlp0_bwd_rows', L.i64(M))
    part = _workspace('pos0', rows * C * 4, x.device)
    L.call('p4p_pos_mlp0_bwd', L.ptr(x), L.ptr(W), L.ptr(b), L.ptr(dh), L.ptr(part), L.i64(M), C, _stream())
    s = colreduce(part, rows, C * 4).view(C, 4)
    return s[:, :3].contiguous(), s[:, 3].contiguous()


def global_feat_fwd(x, want_max, want_cls):
    _chk(x, 'x', torch.float32, 3)
    B, T, D = x.shape
    mx = torch.empty(B, D, dtype=torch.float32, device=x.device) if want_max else None
    arg = torch.empty(B, D, dtype=torch.int32, device=x.device) if want_max else None
    cls = torch.empty(B, D, dtype=torch.float32, device=x.device) if want_cls else None
    L.call('p4p_global_feat_fwd', L.ptr(x), L.ptr(mx), L.ptr(arg), L.ptr(cls), B, T, D, _stream())
    return mx, arg, cls


def global_feat_bwd(dmax, arg, dcls, B, T, D, device):
    dx = torch.empty(B, T, D, dtype=torch.float32, device=device)
    L.call('p4p_global_feat_bwd', L.ptr(dx), L.ptr(dmax), L.ptr(arg), L.ptr(dcls), B, T, D, _stream())
    return dx


def colreduce(part, nrows, ncols):
    out = torch.empty(ncols, dtype=torch.float32, device=part.device)
    L.call('p4p_colreduce_f32', L.ptr(part), L.ptr(out), int(nrows), L.i64(ncols), int(ncols), 0.0, _stream())
    return out


# ---------------------------------------------------------------- P3Embed local MLP
def embed_partial_rows():
    return L.call_int('p4p_embed_partial_rows')


def embed_moments(points, feats, idx):
    B, N, _ = points.shape
    _, S, k = idx.shape
    part = torch.empty(embed_partial_rows(), 27, dtype=torch.float32, device=points.device)
    L.call('p4p_embed_moments', L.ptr(points), L.ptr(feats), L.ptr(idx), L.ptr(part), B, N, S, k, _stream())
    return part


def embed_fold(W1, W2, b2, bn, mom_partial, Np, training):
    C1 = W2.shape[0]
    dev = W1.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    aeff, scale, shift, mean, invstd, mom = f(C1, 8), f(C1), f(C1), f(C1), f(C1), f(42)
    L.call('p4p_embed_fold', L.ptr(W1), L.ptr(W2), L.ptr(b2), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
           L.ptr(bn.running_var), L.ptr(mom_partial), L.ptr(aeff), L.ptr(scale), L.ptr(shift), L.ptr(mean),
           L.ptr(invstd), L.ptr(mom), L.i64(Np), float(bn.eps), float(bn.momentum), int(training), C1, _stream())
    return aeff, scale, shift, mean, invstd, mom


def embed_stage1(points, feats, idx, aeff, scale, shift):
    B, N, _ = points.shape
    _, S, k = idx.shape
    C1 = aeff.shape[0]
    dev = points.device
    F1 = torch.empty(B * S * k, C1, dtype=torch.bfloat16, device=dev)
    pooled = torch.empty(B * S, C1, dtype=torch.bfloat16, device=dev)
    arg = torch.empty(B * S, C1, dtype=torch.uint8, device=dev)
    L.call('p4p_embed_stage1', L.ptr(points), L.ptr(feats), L.ptr(idx), L.ptr(aeff), L.ptr(scale), L.ptr(shift),
           L.ptr(F1), L.ptr(pooled), L.ptr(arg), B, N, S, k, C1, _stream())
    return F1, pooled, arg


def bn_finalize(sums, bn, Np, training):
    C = bn.weight.shape[0]
    dev = bn.weight.device
    f = lambda: torch.empty(C, dtype=torch.float32, device=dev)
    scale, shift, mean, invstd = f(), f(), f(), f()
    L.call('p4p_bn_finalize', L.ptr(sums), L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean),
           L.ptr(bn.running_var), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.i64(Np), float(bn.eps),
           float(bn.momentum), int(training), C, _stream())
    return scale, shift, mean, invstd


def bn_relu(x, scale, shift):
    y = torch.empty_like(x)
    L.call('p4p_bn_relu_bf16', L.ptr(x), L.ptr(y), L.ptr(scale), L.ptr(shift), L.i64(x.shape[0]), x.shape[1], _stream())
    return y


def bn_relu_maxpool(H3, scale, shift, G, k, want_arg=True):
    C = H3.shape[1]
    tok = torch.empty(G, C, dtype=torch.float32, device=H3.device)
    arg = torch.empty(G, C, dtype=torch.uint8, device=H3.device) if want_arg else None
    L.call('p4p_bn_relu_maxpool', L.ptr(H3), L.ptr(scale), L.ptr(shift), None, L.ptr(tok), L.ptr(arg), L.i64(G), k, C,
           _stream())
    return tok, arg


def pool_bn_bwd(dtok, H3, arg, scale, shift, mean, invstd, G, k):
    C = H3.shape[1]
    part = _workspace('embed_part', embed_partial_rows() * 7 * 512, H3.device)
    sums = torch.empty(2, C, dtype=torch.float32, device=H3.device)
    dH3 = torch.empty_like(H3)
    L.call('p4p_pool_bn_bwd', L.ptr(dtok), int(dtok.dtype == torch.float32), L.ptr(H3), L.ptr(arg), L.ptr(scale),
           L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(part), L.ptr(sums), L.ptr(dH3), L.i64(G), k, C, _stream())
    return dH3, sums


def relu_bn_bwd_(dY, H, scale, shift, mean, invstd, group=None):
    """dY <- gradient w.r.t. the BatchNorm input (in place); returns sums [2][C]; with group = k also the sums of every k consecutive
    rows of the result ([rows / k, C] bf16: group_sum(dY, rows // k, k) out of the same pass)."""
    rows, C = H.shape
    part = _workspace('embed_part', embed_partial_rows() * 7 * 512, H.device)
    sums = torch.empty(2, C, dtype=torch.float32, device=H.device)
    if group:
        gs = torch.empty(rows // group, C, dtype=torch.bfloat16, device=H.device)
        L.call('p4p_relu_bn_bwd_gsum', L.ptr(dY), L.ptr(H), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(part),
               L.ptr(sums), L.ptr(gs), int(group), rows, C, _stream())
        return sums, gs
    L.call('p4p_relu_bn_bwd', L.ptr(dY), L.ptr(H), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(part),
           L.ptr(sums), L.i64(rows), C, _stream())
    return sums


def group_sum(x, G, k):
    out = torch.empty(G, x.shape[1], dtype=torch.bfloat16, device=x.device)
    L.call('p4p_group_sum_bf16', L.ptr(x), L.ptr(out), L.i64(G), k, x.shape[1], _stream())
    return out


def embed_stage1_bwd(points, feats, idx, F1, dF1, dPooled, arg, mom, aeff, W1, W2, gamma, invstd):
    B, N, _ = points.shape
    _, S, k = idx.shape
    C1 = aeff.shape[0]
    dev = points.device
    part = _workspace('embed_part', embed_partial_rows() * 7 * 512, dev)
    sums = torch.empty(7, C1, dtype=torch.float32, device=dev)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    dW1, dW2, db2, dg, db = f(C1, 6), f(C1, C1), f(C1), f(C1), f(C1)
    L.call('p4p_embed_stage1_bwd', L.ptr(points), L.ptr(feats), L.ptr(idx), L.ptr(F1), L.ptr(dF1), L.ptr(dPooled),
           L.ptr(arg), L.ptr(part), L.ptr(sums), L.ptr(mom), L.ptr(aeff), L.ptr(W1), L.ptr(W2), L.ptr(gamma),
           L.ptr(invstd), L.ptr(dW1), L.ptr(dW2), L.ptr(db2), L.ptr(dg), L.ptr(db), 0.0, B, N, S, k, C1, _stream())
    return dW1, dW2, db2, dg, db


# ---------------------------------------------------------------- AdaptPointFormer tokenizer glue
def morton_order(centers, resolution=1024, want_codes=False):
    """MortonEncoder.points_to_morton (reference src/models/apf_utils.py:67-104) -> int32 (B,G) permutation."""
    _chk(centers, 'centers', torch.float32, 3)
    B, G, D = centers.shape
    if D != 3:
        raise RuntimeError(f'centers: last dim must be 3 (xyz), got {D}')
    order = torch.empty(B, G, dtype=torch.int32, device=centers.device)
    codes = torch.empty(B, G, dtype=torch.int32, device=centers.device) if want_codes else None
    L.call('p4p_morton_order_f32', L.ptr(centers), L.ptr(order), L.ptr(codes), B, G, int(resolution), _stream())
    return (order, codes) if want_codes else order


def apf_group(x, fps_idx, knn_idx, order, want_f32=False, want_x0=True, want_center=True):
    """Group.forward after the index ops (reference src/models/apf.py:70-110), x (B,N,C) with C = 3 or 4 (xyz + height).
    Returns (nb_f32 (B,G,gs,2C) | None, x0_bf16 | None, center (B,G,3) | None), rows in Morton order."""
    _chk(x, 'x', torch.float32, 3)
    _chk(fps_idx, 'fps_idx', torch.int64, 2)
    _chk(knn_idx, 'knn_idx', torch.int32, 3)
    _chk(order, 'order', torch.int32, 2)
    B, N, C = x.shape
    if C not in (3, 4):
        raise RuntimeError(f'apf_group: built for in_channels 3 (xyz) and 4 (xyz + height), got {C} channels')
    G, gs = knn_idx.shape[1], knn_idx.shape[2]
    if fps_idx.shape != (B, G) or order.shape != (B, G) or knn_idx.shape[0] != B:
        raise RuntimeError('apf_group: index shapes do not match')
    dev = x.device
    nb = torch.empty(B, G, gs, 2 * C, dtype=torch.float32, device=dev) if want_f32 else None
    x0 = torch.empty(B * G * gs, 64, dtype=torch.bfloat16, device=dev) if want_x0 else None
    cen = torch.empty(B, G, 3, dtype=torch.float32, device=dev) if want_center else None
    L.call('p4p_apf_group', L.ptr(x), L.ptr(fps_idx), L.ptr(knn_idx), L.ptr(order), L.ptr(nb), L.ptr(x0), L.ptr(cen), B, N, C,
           G, gs, _stream())
    return nb, x0, cen


def group_max(H, G, k, bias=None, out_f32=True, want_arg=True):
    """max over the k rows of every group of H [G*k, C] bf16 (+ bias) -> (tok [G,C] f32 or bf16, arg u8 | None)."""
    _chk(H, 'H', torch.bfloat16, 2)
    C = H.shape[1]
    if H.shape[0] != G * k:
        raise RuntimeError('group_max: H must be (G*k, C)')
    tok = torch.empty(G, C, dtype=torch.float32 if out_f32 else torch.bfloat16, device=H.device)
    arg = torch.empty(G, C, dtype=torch.uint8, device=H.device) if want_arg else None
    L.call('p4p_group_max_bf16', L.ptr(H), L.ptr(bias), L.ptr(tok) if out_f32 else None, None if out_f32 else L.ptr(tok),
           L.ptr(arg), L.i64(G), int(k), C, _stream())
    return tok, arg


def group_max_bwd(dtok, arg, G, k, into=None):
    """dH [G*k, C] bf16 = dtok scattered to the arg-max rows; `into` accumulates in place instead."""
    _chk(dtok, 'dtok', None, 2)
    if dtok.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('group_max_bwd: dtok must be f32 or bf16')
    _chk(arg, 'arg', torch.uint8, 2)
    C = dtok.shape[1]
    dH = into if into is not None else torch.empty(G * k, C, dtype=torch.bfloat16, device=dtok.device)
    _chk(dH, 'dH', torch.bfloat16, 2)
    L.call('p4p_group_max_bwd', L.ptr(dtok), int(dtok.dtype == torch.float32), L.ptr(arg), L.ptr(dH), int(into is not None),
           L.i64(G), int(k), C, _stream())
    return dH


def rows_restore_(dst, src, keep, B):
    """dst[b] = src[b] for the samples with keep[b] == False (dst, src f32, B samples of equal size; keep bool [B])."""
    _chk(dst, 'dst', torch.float32)
    _chk(src, 'src', torch.float32)
    _chk(keep, 'keep', torch.bool, 1)
    if dst.shape != src.shape or dst.numel() % B or keep.shape[0] != B:
        raise RuntimeError('rows_restore_: shapes do not match')
    L.call('p4p_rows_restore_f32', L.ptr(dst), L.ptr(src), L.ptr(keep), B, L.i64(dst.numel() // B), _stream())
    return dst


def rows_zero_(x, keep, B):
    """x[b] = 0 for the samples with keep[b] == False (x bf16, B samples of equal size; keep bool [B])."""
    _chk(x, 'x', torch.bfloat16)
    _chk(keep, 'keep', torch.bool, 1)
    if x.numel() % B or keep.shape[0] != B:
        raise RuntimeError('rows_zero_: shapes do not match')
    L.call('p4p_rows_zero_bf16', L.ptr(x), L.ptr(keep), B, L.i64(x.numel() // B), _stream())
    return x


def adapter_mid_fwd(pre, p=0.0):
    """(down bf16, gate f32) = dropout_p(relu(pre)) and its derivative factor (reference src/models/apf_utils.py:222-226)."""
    _chk(pre, 'pre', torch.float32, 2)
    u = torch.rand_like(pre) if p > 0 else None
    down = torch.empty(pre.shape, dtype=torch.bfloat16, device=pre.device)
    gate = torch.empty_like(pre)
    L.call('p4p_adapter_mid_fwd', L.ptr(pre), L.ptr(u), float(p), L.ptr(down), L.ptr(gate), L.i64(pre.numel()), _stream())
    return down, gate


def adapter_mid_bwd(ddown, gate):
    _chk(ddown, 'ddown', torch.float32, 2)
    _chk(gate, 'gate', torch.float32, 2)
    dpre = torch.empty(ddown.shape, dtype=torch.bfloat16, device=ddown.device)
    L.call('p4p_adapter_mid_bwd', L.ptr(ddown), L.ptr(gate), L.ptr(dpre), L.i64(ddown.numel()), _stream())
    return dpre


# ---------------------------------------------------------------- batched augmentation
AUG_NORMALIZE, AUG_AFFINE, AUG_JITTER, AUG_REPLACE, AUG_DROPOUT = 1, 2, 4, 8, 16


def augment(points, flags, affine=None, jitter=None, clip=0.05, replace_mask=None, replace_u=None, replace_g=None,
            dropout_mask=None, out=None):
    """The reference's per-cloud augmentation chain (src/data/augment.py) for a batch, every random draw an input; see
    include/p4p_hip.h: p4p_augment_f32."""
    _chk(points, 'points', torch.float32, 3)
    B, N, C = points.shape
    if C != 3:
        raise RuntimeError(f'augment: expected (B,N,3) clouds, got {tuple(points.shape)}')
    _chk(flags, 'flags', torch.int32, 1)
    for t, nm, dt, shp in ((affine, 'affine', torch.float32, (B, 12)), (jitter, 'jitter', torch.float32, (B, N, 3)),
                           (replace_mask, 'replace_mask', torch.bool, (B, N)), (replace_u, 'replace_u', torch.float32, (B, N, 3)),
                           (replace_g, 'replace_g', torch.float32, (B, N, 3)), (dropout_mask, 'dropout_mask', torch.bool, (B, N))):
        if t is not None:
            _chk(t, nm, dt)
            if tuple(t.shape) != shp:
                raise RuntimeError(f'augment: {nm} must have shape {shp}, got {tuple(t.shape)}')
    if flags.shape[0] != B:
        raise RuntimeError('augment: flags must have shape (B,)')
    if out is None:
        out = torch.empty_like(points)
    L.call('p4p_augment_f32', L.ptr(points), L.ptr(out), L.ptr(flags), L.ptr(affine), L.ptr(jitter), float(clip),
           L.ptr(replace_mask), L.ptr(replace_u), L.ptr(replace_g), L.ptr(dropout_mask), B, N, _stream())
    return out


def set_reserved_cus(r):
    """Compute units the persistent kernels (short-K GEMM tile loop, walking attention backward) leave to a concurrent
    long-lived kernel -- RCCL's all-reduce workgroups under data parallelism.  0: all CUs; r > 0: grids of CUs - r
    (multiples of 8), results bit-identical; r < 0: a workgroup per tile / pair.  Returns the previous value."""
    return L.call_int('p4p_set_reserved_cus', int(r))


def device_cu_count():
    """Compute units of the current device (-1 if the runtime cannot say)."""
    return L.call_int('p4p_device_cu_count')


def persistent_grid_size():
    """Workgroups a persistent launch takes on the current device now (0: no persistent kernels are launched)."""
    return L.call_int('p4p_persistent_grid_size')

