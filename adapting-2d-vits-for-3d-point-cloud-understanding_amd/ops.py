"""Tensor-level wrappers over the C ABI: argument checks mirror the errors torch would raise in the
reference (shape / dtype / device / contiguity), outputs are allocated with torch's caching
allocator on the input's device, kernels are enqueued on torch's current HIP stream."""
import torch

from . import _lib as L

EPI_BF16, EPI_BF16_GELU, EPI_F32_RESID, EPI_BF16_DGELU, EPI_BF16_ROWGROUP, EPI_BF16_STATS = range(6)


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
            want_stats=False, want_pre=False):
    """out[M,N] = a[M,K] @ w[N,K]^T with a fused epilogue (see include/p4p_hip.h)."""
    _chk(a, 'a', torch.bfloat16, 2)
    _chk(w, 'w', torch.bfloat16, 2)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f'gemm_tn: a is {tuple(a.shape)}, w is {tuple(w.shape)}')
    f32_out = epilogue == EPI_F32_RESID
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if f32_out else torch.bfloat16, device=a.device)
    stats = None
    if epilogue in (EPI_BF16_ROWGROUP, EPI_BF16_STATS):
        rows = L.call_int('p4p_gemm_tn_stats_rows', M)
        stats = torch.empty(rows, 2, N, dtype=torch.float32, device=a.device)
    if epilogue == EPI_BF16_GELU and want_pre and pre is None:
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    for t, nm, dt in ((bias, 'bias', torch.float32), (resid, 'resid', torch.float32),
                      (pre, 'pre', torch.bfloat16), (rowgrp, 'rowgrp', torch.float32)):
        if t is not None:
            _chk(t, nm, dt)
    L.call('p4p_gemm_tn_bf16', int(epilogue), L.ptr(a), L.ptr(w), L.ptr(out), L.ptr(bias), L.ptr(resid),
           L.ptr(pre), L.ptr(rowgrp), L.ptr(stats), M, N, K, int(group), _stream())
    if stats is not None:
        return out, stats
    if epilogue == EPI_BF16_GELU and want_pre:
        return out, pre
    return out


_wgrad_ws = {}


def _workspace(key, numel, device, dtype=torch.float32):
    t = _wgrad_ws.get((key, device, dtype))
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel), dtype=dtype, device=device)
        _wgrad_ws[(key, device, dtype)] = t
    return t


def gemm_wgrad(g, a, out=None, beta=0.0):
    """dW[N,K] f32 = beta*dW + g[M,N]^T @ a[M,K]."""
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
    splits = L.call_int('p4p_wgrad_splits', M, N, K)
    ws = _workspace('wgrad', splits * N * K, g.device)
    L.call('p4p_gemm_wgrad_bf16', L.ptr(g), L.ptr(a), L.ptr(out), L.ptr(ws), M, N, K, float(beta), _stream())
    return out


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


def layernorm_bwd(dy, xin, mean, rstd, gamma, dx=None, dgamma=None, dbeta=None, beta_acc=0.0):
    """Returns (dx, dgamma, dbeta); if dx is given the LN gradient is ADDED to it in place."""
    M, D = xin.shape
    _chk(dy, 'dy', None, 2)
    if dy.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('layernorm_bwd: dy must be f32 or bf16')
    acc = dx is not None
    if dx is None:
        dx = torch.empty(M, D, dtype=torch.float32, device=xin.device)
    if dgamma is None:
        dgamma = torch.empty(D, dtype=torch.float32, device=xin.device)
        dbeta = torch.empty(D, dtype=torch.float32, device=xin.device)
        beta_acc = 0.0
    rows = L.call_int('p4p_layernorm_bwd_partial_rows')
    part = _workspace('lnbwd', rows * 2 * D, xin.device)
    L.call('p4p_layernorm_bwd', L.ptr(dy), int(dy.dtype == torch.float32), L.ptr(xin), L.ptr(mean), L.ptr(rstd),
           L.ptr(gamma), L.ptr(dx), int(acc), L.ptr(part), L.ptr(dgamma), L.ptr(dbeta), float(beta_acc), M, D,
           _stream())
    return dx, dgamma, dbeta


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


def attention_bwd(qkv, o, dout, lse, B, T, H):
    _chk(qkv, 'qkv', torch.bfloat16, 2)
    _chk(o, 'o', torch.bfloat16, 2)
    _chk(dout, 'dout', torch.bfloat16, 2)
    _chk(lse, 'lse', torch.float32, 3)
    D = o.shape[1]
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    L.call('p4p_attention_bwd', L.ptr(qkv), L.ptr(o), L.ptr(dout), L.ptr(lse), L.ptr(delta), L.ptr(dqkv), B, T, H, D,
           float((D // H) ** -0.5), _stream())
    return dqkv
