// Streaming helpers around the GEMMs (all HBM-bound, 8-16 B per lane):
//   * fp32 master weight -> bf16 operand copies W and W^T (once per optimizer step);
//   * bias gradients: column sums of a bf16 [M,N] gradient (autograd of nn.Linear.bias);
//   * row remaps between the token stream [B*T,D] (cls row first, src/models/pix4point.py:248-252)
//     and the compact patch rows [B*S,D].
#include "common.h"
#include "gemm_tn.h"        // gelu_f / dgelu_f (exact-erf GELU)

// dst[r][c] = bf16(src[r*ld + c]);  dstT[c][r] = same.  32x32 tiles through LDS.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, long long ld,
                                                             __bf16* __restrict__ dst, __bf16* __restrict__ dstT,
                                                             int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + j * 8, c = c0 + tx;
        float v = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.f;
        tile[ty + j * 8][tx] = v;
        if (dst && r < R && c < C) dst[(size_t)r * C + c] = f2bf(v);
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + j * 8, r = r0 + tx;
        if (r < R && c < C) dstT[(size_t)c * R + r] = f2bf(tile[tx][ty + j * 8]);
    }
}

// The same for R, C, ld multiples of 4 and 16-B aligned pointers: 64 x 64 tiles, float4 loads, 8-B bf16x4 stores both ways
// (the per-step refresh of every weight's bf16 operand copies: 53 launches over 0.69 GB at ViT-B).
__global__ __launch_bounds__(256) void cast_transpose4_kernel(const float* __restrict__ src, long long ld,
                                                              __bf16* __restrict__ dst, __bf16* __restrict__ dstT,
                                                              int R, int C) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 column quads x 16 rows
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + j * 16, c = c0 + tx * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R && c < C) v = *reinterpret_cast<const f32x4*>(src + (size_t)r * ld + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[ty + j * 16][tx * 4 + i] = v[i];
        if (dst && r < R && c < C) {
            const bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *reinterpret_cast<bf16x4*>(dst + (size_t)r * C + c) = w;
        }
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + j * 16, r = r0 + tx * 4;
        if (r < R && c < C) {
            const bf16x4 w = {f2bf(tile[tx * 4][ty + j * 16]), f2bf(tile[tx * 4 + 1][ty + j * 16]),
                              f2bf(tile[tx * 4 + 2][ty + j * 16]), f2bf(tile[tx * 4 + 3][ty + j * 16])};
            *reinterpret_cast<bf16x4*>(dstT + (size_t)c * R + r) = w;
        }
    }
}

// Many weights in ONE launch (the per-step refresh of the encoder's 4 L operand pairs: 48 launches of ~7 us -> 1).  Every
// tensor obeys the cast_transpose4 conditions (multiples of 4, aligned); block -> (tensor, 64 x 64 tile) through the prefix
// of tile counts.
struct P4PCastDesc { const float* src; __bf16* dst; __bf16* dstT; long long ld; int R, C; int tile0, tiles_c; };
__global__ __launch_bounds__(256) void cast_transpose4_multi_kernel(const P4PCastDesc* __restrict__ descs, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;                                  // last tensor whose first tile is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const P4PCastDesc d = descs[lo];
    const int tl = blockIdx.x - d.tile0;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = (tl / d.tiles_c) * 64, c0 = (tl % d.tiles_c) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + j * 16, c = c0 + tx * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < d.R && c < d.C) v = *reinterpret_cast<const f32x4*>(d.src + (size_t)r * d.ld + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[ty + j * 16][tx * 4 + i] = v[i];
        if (d.dst && r < d.R && c < d.C) {
            const bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *reinterpret_cast<bf16x4*>(d.dst + (size_t)r * d.C + c) = w;
        }
    }
    if (!d.dstT) return;                                     // block-uniform
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + j * 16, r = r0 + tx * 4;
        if (r < d.R && c < d.C) {
            const bf16x4 w = {f2bf(tile[tx * 4][ty + j * 16]), f2bf(tile[tx * 4 + 1][ty + j * 16]),
                              f2bf(tile[tx * 4 + 2][ty + j * 16]), f2bf(tile[tx * 4 + 3][ty + j * 16])};
            *reinterpret_cast<bf16x4*>(d.dstT + (size_t)c * d.R + r) = w;
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n) {
    long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e + 3 < n) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src + e);
        bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(dst + e) = w;
    } else {
        for (; e < n; ++e) dst[e] = f2bf(src[e]);
    }
}

// partial[split][n] = sum over the split's rows of G[m][n]; 256 threads = 64 column quads x 4 row lanes
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ G, float* __restrict__ partial,
                                                          int M, int N, int rows_per_split) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + tx * 4;
    const int m_begin = blockIdx.y * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (int m = m_begin + ty; m < m_end; m += 4) {
            bf16x4 w = *reinterpret_cast<const bf16x4*>(G + (size_t)m * N + n);
            acc += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
        }
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        f32x4 s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * N + n) = s;
    }
}

// 256 threads = 32 columns x 8 row lanes (fixed order: reproducible)
__global__ __launch_bounds__(256) void colreduce2_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int nrows, int ncols, float beta) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < ncols)
        for (int r = ty; r < nrows; r += 8) s += part[(size_t)r * ncols + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < ncols) {
        float v = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) + ((red[4][tx] + red[5][tx]) + (red[6][tx] + red[7][tx]));
        out[c] = beta != 0.f ? beta * out[c] + v : v;
    }
}

// Token stream <-> compact patch rows.  stream row of patch (b,s) = b*T + 1 + s  (T = S + 1).
// mode 0: compact_bf16[m] = bf16(stream_f32[row(m)])      (gradient of the patch tokens)
// mode 1: stream_f32[row(m)] = compact_f32[m]             (assemble tokens / positional embedding)
__global__ void token_rows_kernel(float* __restrict__ stream, void* __restrict__ compact, long long n_compact, int S,
                                  int D, int mode) {
    long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n_compact) return;
    long long m = e / D;
    int c = (int)(e - m * D);
    long long row = m + m / S + 1;
    if (mode == 0) {
        f32x4 v = *reinterpret_cast<const f32x4*>(stream + row * D + c);
        bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>((__bf16*)compact + e) = w;
    } else {
        *reinterpret_cast<f32x4*>(stream + row * D + c) = *reinterpret_cast<const f32x4*>((const float*)compact + e);
    }
}

// cls rows: mode 1: stream[b*T] = vec;  mode 0: out[c] = sum_b stream[b*T][c]  (256 threads = 64 columns x 4 b-lanes,
// fixed order: reproducible)
__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ stream, float* __restrict__ vec, int B, int T, int D,
                                                       int mode) {
    if (mode == 1) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c >= D) return;
        const float v = vec[c];
        for (int b = blockIdx.y; b < B; b += gridDim.y) stream[(size_t)b * T * D + c] = v;
        return;
    }
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < D)
        for (int b = ty; b < B; b += 4) s += stream[(size_t)b * T * D + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < D) vec[c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// Global feature of PointViT.forward_cls_feat (src/models/pix4point.py:260-271): max over the patch tokens x[:,1:] and
// the cls token x[:,0].  fwd: out_max[b][c], arg[b][c] (token index of the first maximum), out_cls[b][c].
// bwd: dx[b][t][c] = dmax[b][c] at t = arg (+ dcls[b][c] at t = 0), zero elsewhere -- ONE pass over dx instead of
// torch's zeros + scatter + slice-pad + add.
__global__ __launch_bounds__(256) void global_feat_fwd_kernel(const float* __restrict__ x, float* __restrict__ out_max,
                                                              int* __restrict__ arg, float* __restrict__ out_cls, int T, int D) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= D) return;
    const float* xb = x + (size_t)b * T * D + c;
    if (out_cls) out_cls[(size_t)b * D + c] = xb[0];
    if (out_max) {
        float m = -INFINITY;
        int am = 1;
        for (int t = 1; t < T; ++t) {
            const float v = xb[(size_t)t * D];
            if (v > m || v != v) { if (!(m != m)) { m = v; am = t; } }      // first maximum; a NaN wins and stays (torch.max)
        }
        out_max[(size_t)b * D + c] = m;
        arg[(size_t)b * D + c] = am;
    }
}
// The same for D % 4 == 0 with 16-byte loads: 256 threads = 64 column quads x 4 token slices (tokens 1 + s, 5 + s, ..), four
// independent loads in flight per thread, the slices' candidates meet in LDS.  (The kernel above walks all T tokens per thread with
// 4-byte loads behind a data-dependent branch: 129 us for the 202 MB of the N = 1024, B = 256 step; this one streams.)
__device__ __forceinline__ void gf_take(float& m, int& am, float v, int t) {      // scan order = increasing t inside a slice
    if (v > m || v != v) { if (!(m != m)) { m = v; am = t; } }
}
__device__ __forceinline__ bool gf_better(float va, int ta, float vb, int tb) {   // does candidate a beat b?  (first maximum; the first NaN wins)
    const bool na = va != va, nb = vb != vb;
    if (na || nb) return na && (!nb || ta < tb);
    return va > vb || (va == vb && ta < tb);
}
__global__ __launch_bounds__(256) void global_feat_fwd4_kernel(const float* __restrict__ x, float* __restrict__ out_max,
                                                               int* __restrict__ arg, float* __restrict__ out_cls, int T, int D) {
    __shared__ f32x4 sv[4][64];
    __shared__ int st[4][64][4];
    const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6, b = blockIdx.y;
    const int c = (blockIdx.x * 64 + tx) * 4;
    const bool live = c < D;
    const float* xb = x + (size_t)b * T * D + (live ? c : 0);
    if (live && sl == 0 && out_cls) *reinterpret_cast<f32x4*>(out_cls + (size_t)b * D + c) = *reinterpret_cast<const f32x4*>(xb);
    if (!out_max) return;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int am[4] = {1, 1, 1, 1};
    if (live) {
        int t = 1 + sl;
        for (; t + 12 < T; t += 16) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (size_t)t * D), v1 = *reinterpret_cast<const f32x4*>(xb + (size_t)(t + 4) * D);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(xb + (size_t)(t + 8) * D), v3 = *reinterpret_cast<const f32x4*>(xb + (size_t)(t + 12) * D);
#pragma unroll
            for (int i = 0; i < 4; ++i) { gf_take(m[i], am[i], v0[i], t); gf_take(m[i], am[i], v1[i], t + 4); gf_take(m[i], am[i], v2[i], t + 8); gf_take(m[i], am[i], v3[i], t + 12); }
        }
        for (; t < T; t += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (size_t)t * D);
#pragma unroll
            for (int i = 0; i < 4; ++i) gf_take(m[i], am[i], v0[i], t);
        }
    }
    sv[sl][tx] = f32x4{m[0], m[1], m[2], m[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) st[sl][tx][i] = am[i];
    __syncthreads();
    if (sl == 0 && live) {
        // a slice that saw no token (T - 1 < 4) holds (-inf, 1): it never beats a real candidate of the same value with a smaller index,
        // and if every token is -inf the answer is the first one, token 1 -- which is what the empty slices say too
        f32x4 bm = sv[0][tx];
        int bt[4] = {st[0][tx][0], st[0][tx][1], st[0][tx][2], st[0][tx][3]};
#pragma unroll
        for (int s = 1; s < 4; ++s) {
            const f32x4 v = sv[s][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (gf_better(v[i], st[s][tx][i], bm[i], bt[i])) { bm[i] = v[i]; bt[i] = st[s][tx][i]; }
        }
        *reinterpret_cast<f32x4*>(out_max + (size_t)b * D + c) = bm;
        *reinterpret_cast<int4*>(arg + (size_t)b * D + c) = int4{bt[0], bt[1], bt[2], bt[3]};
    }
}
__global__ __launch_bounds__(256) void global_feat_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dmax,
                                                              const int* __restrict__ arg, const float* __restrict__ dcls,
                                                              int T, int D) {
    const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4, t = blockIdx.y, b = blockIdx.z;
    if (c4 >= D) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const size_t bd = (size_t)b * D + c4;
    if (dmax) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (arg[bd + r] == t) v[r] = dmax[bd + r];
    }
    if (dcls && t == 0) v += *reinterpret_cast<const f32x4*>(dcls + bd);
    *reinterpret_cast<f32x4*>(dx + ((size_t)b * T + t) * D + c4) = v;
}

extern "C" int p4p_cast_transpose_f32_bf16(const float* src, long long ld_src, void* dst, void* dstT, int R, int C,
                                           void* stream) {
    P4P_CHECK(src && (dst || dstT) && R >= 1 && C >= 1 && ld_src >= C, "p4p_cast_transpose_f32_bf16: bad arguments");
    const bool vec = R % 4 == 0 && C % 4 == 0 && ld_src % 4 == 0 && (reinterpret_cast<size_t>(src) & 15) == 0 &&
                     (reinterpret_cast<size_t>(dst) & 7) == 0 && (reinterpret_cast<size_t>(dstT) & 7) == 0;
    if (vec)
        hipLaunchKernelGGL(cast_transpose4_kernel, dim3(ceil_div(C, 64), ceil_div(R, 64)), dim3(256), 0,
                           (hipStream_t)stream, src, ld_src, (__bf16*)dst, (__bf16*)dstT, R, C);
    else
        hipLaunchKernelGGL(cast_transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0,
                           (hipStream_t)stream, src, ld_src, (__bf16*)dst, (__bf16*)dstT, R, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

// descs: DEVICE array of n records {const float* src; bf16* dst; bf16* dstT; long long ld; int R, C; int tile0, tiles_c}
// (48 bytes each; tile0 = number of 64 x 64 tiles of the records before it, tiles_c = ceil(C / 64)); total_tiles = grid size.
// Every record must satisfy R, C, ld % 4 == 0, src 16-byte and dst / dstT 8-byte aligned (the caller checks: the table is
// device memory).
extern "C" int p4p_cast_transpose_multi(const void* descs, int n, int total_tiles, void* stream) {
    P4P_CHECK(descs && n >= 1 && total_tiles >= 1, "p4p_cast_transpose_multi: bad arguments");
    hipLaunchKernelGGL(cast_transpose4_multi_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const P4PCastDesc*)descs, n);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_cast_f32_bf16(const float* src, void* dst, long long n, void* stream) {
    P4P_CHECK(src && dst && n >= 0, "p4p_cast_f32_bf16: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ceil_div((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (__bf16*)dst, n);
    P4P_LAUNCH_CHECK();
    return 0;
}

#define COLSUM_MAX_SPLITS 256
extern "C" int p4p_colsum_splits(int M) {
    int s = ceil_div(M, 256);
    return s < 1 ? 1 : (s > COLSUM_MAX_SPLITS ? COLSUM_MAX_SPLITS : s);
}

// out[N] f32 = beta*out + column sums of G[M,N] bf16.  workspace: p4p_colsum_splits(M)*N floats.
extern "C" int p4p_colsum_bf16(const void* G, float* out, float* workspace, int M, int N, float beta, void* stream) {
    P4P_CHECK(G && out && workspace, "p4p_colsum_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 4 && N % 4 == 0, "p4p_colsum_bf16: bad sizes M=%d N=%d", M, N);
    hipStream_t st = (hipStream_t)stream;
    int splits = p4p_colsum_splits(M);
    int rps = ceil_div(M, splits);
    splits = ceil_div(M, rps);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(ceil_div(N, 256), splits), dim3(256), 0, st, (const __bf16*)G, workspace,
                       M, N, rps);
    P4P_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce2_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, st, workspace, out, splits, N, beta);
    P4P_LAUNCH_CHECK();
    return 0;
}

// partial[r][N] for r < ceil(M/256) = column sums of rows [256r, 256r+256) of G (bf16); returns through the C ABI so the
// GEMM launcher (another translation unit) can finish a fused-bias-gradient request on tile shapes without the fusion.
extern "C" int p4p_colsum_partial_bf16(const void* G, float* partial, int M, int N, void* stream) {
    P4P_CHECK(G && partial, "p4p_colsum_partial_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 4 && N % 4 == 0, "p4p_colsum_partial_bf16: bad sizes M=%d N=%d", M, N);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(ceil_div(N, 256), ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)G, partial, M, N, 256);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_token_rows(float* stream_f32, void* compact, int B, int S, int D, int mode, void* stream) {
    P4P_CHECK(stream_f32 && compact && B >= 1 && S >= 1 && D >= 4 && D % 4 == 0 && (mode == 0 || mode == 1),
              "p4p_token_rows: bad arguments");
    long long n = (long long)B * S * D;
    hipLaunchKernelGGL(token_rows_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, stream_f32,
                       compact, n, S, D, mode);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_cls_rows(float* stream_f32, float* vec, int B, int T, int D, int mode, void* stream) {
    P4P_CHECK(stream_f32 && vec && B >= 1 && T >= 1 && D >= 1 && (mode == 0 || mode == 1), "p4p_cls_rows: bad arguments");
    dim3 grid(mode == 1 ? ceil_div(D, 256) : ceil_div(D, 64), mode == 1 ? (B < 64 ? B : 64) : 1);
    hipLaunchKernelGGL(cls_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, stream_f32, vec, B, T, D, mode);
    P4P_LAUNCH_CHECK();
    return 0;
}

// First layer of PointViT.pos_embed (src/models/pix4point.py:214-218,246): h = GELU(Linear(3, C)(centers)).  K = 3 is no
// GEMM: one thread per output (fwd), and in the backward the C x 3 weight gradient is a reduction over the rows that
// rocBLAS runs with a 16x32 macro-tile in ~0.2 ms; here every workgroup reduces 256 rows into partial[blk][C][4]
// (dW[j][0..2], db[j]) and p4p_colreduce_f32 finishes (fixed order).  The input gets no gradient (points are data).
__global__ __launch_bounds__(256) void pos_mlp0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, void* __restrict__ h, int h_bf16,
                                                           long long M, int C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * C) return;
    const long long m = e / C;
    const int j = (int)(e - m * C);
    const float pre = (x[m * 3] * W[j * 3] + x[m * 3 + 1] * W[j * 3 + 1]) + (x[m * 3 + 2] * W[j * 3 + 2] + b[j]);
    const float y = gelu_f(pre);
    if (h_bf16) ((__bf16*)h)[e] = f2bf(y);
    else ((float*)h)[e] = y;
}
#define POS0_ROWS 256
__global__ __launch_bounds__(256) void pos_mlp0_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, const float* __restrict__ dh,
                                                           float* __restrict__ partial, long long M, int C) {
    // thread = (column j = tid % C, row lane = tid / C); C divides 256
    const int j = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
    const long long m0 = (long long)blockIdx.x * POS0_ROWS;
    const float w0 = W[j * 3], w1 = W[j * 3 + 1], w2 = W[j * 3 + 2], bj = b[j];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
    for (int r = rl; r < POS0_ROWS; r += nrl) {
        const long long m = m0 + r;
        if (m >= M) break;
        const float x0 = x[m * 3], x1 = x[m * 3 + 1], x2 = x[m * 3 + 2];
        const float pre = (x0 * w0 + x1 * w1) + (x2 * w2 + bj);
        const float g = dh[m * C + j] * dgelu_f(pre);
        a0 += g * x0; a1 += g * x1; a2 += g * x2; ab += g;
    }
    __shared__ f32x4 red[256];
    red[threadIdx.x] = f32x4{a0, a1, a2, ab};
    __syncthreads();
    if (rl == 0) {
        f32x4 s = red[j];
        for (int q = 1; q < nrl; ++q) s += red[q * C + j];
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * C + j) * 4) = s;
    }
}

extern "C" int p4p_pos_mlp0_fwd(const float* x, const float* W, const float* b, void* h, int h_is_bf16, long long M, int C,
                                void* stream) {
    P4P_CHECK(x && W && b && h && M >= 1 && C >= 1, "p4p_pos_mlp0_fwd: bad arguments");
    hipLaunchKernelGGL(pos_mlp0_fwd_kernel, dim3(ceil_div(M * C, 256)), dim3(256), 0, (hipStream_t)stream, x, W, b, h, h_is_bf16, M, C);
    P4P_LAUNCH_CHECK();
    return 0;
}
extern "C" int p4p_pos_mlp0_bwd_rows(long long M) { return ceil_div(M, POS0_ROWS); }
// partial: [p4p_pos_mlp0_bwd_rows(M)][C][4] floats; column sums over the first dimension = (dW[j][0..2], db[j])
extern "C" int p4p_pos_mlp0_bwd(const float* x, const float* W, const float* b, const float* dh, float* partial, long long M,
                                int C, void* stream) {
    P4P_CHECK(x && W && b && dh && partial && M >= 1, "p4p_pos_mlp0_bwd: bad arguments");
    P4P_CHECK(C >= 1 && C <= 256 && 256 % C == 0, "p4p_pos_mlp0_bwd: C=%d must divide 256", C);
    hipLaunchKernelGGL(pos_mlp0_bwd_kernel, dim3(ceil_div(M, POS0_ROWS)), dim3(256), 0, (hipStream_t)stream, x, W, b, dh, partial, M, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_global_feat_fwd(const float* x, float* out_max, int* arg, float* out_cls, int B, int T, int D, void* stream) {
    P4P_CHECK(x && (out_cls || (out_max && arg)) && B >= 1 && T >= 2 && D >= 1 && B <= 65535, "p4p_global_feat_fwd: bad arguments");
    const bool vec = D % 4 == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(out_max) | reinterpret_cast<size_t>(arg) |
                                     reinterpret_cast<size_t>(out_cls)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(global_feat_fwd4_kernel, dim3(ceil_div(D, 256), B), dim3(256), 0, (hipStream_t)stream, x, out_max, arg,
                           out_cls, T, D);
    else
        hipLaunchKernelGGL(global_feat_fwd_kernel, dim3(ceil_div(D, 256), B), dim3(256), 0, (hipStream_t)stream, x, out_max, arg,
                           out_cls, T, D);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_global_feat_bwd(float* dx, const float* dmax, const int* arg, const float* dcls, int B, int T, int D,
                                   void* stream) {
    P4P_CHECK(dx && (!dmax || arg) && B >= 1 && T >= 2 && D >= 4 && D % 4 == 0 && B <= 65535 && T <= 65535,
              "p4p_global_feat_bwd: bad arguments");
    hipLaunchKernelGGL(global_feat_bwd_kernel, dim3(ceil_div(D / 4, 256), T, B), dim3(256), 0, (hipStream_t)stream, dx, dmax, arg,
                       dcls, T, D);
    P4P_LAUNCH_CHECK();
    return 0;
}
