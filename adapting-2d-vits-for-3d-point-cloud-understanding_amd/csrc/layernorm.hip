// LayerNorm forward/backward for the ViT encoder rows, gfx950.
// Reference: timm Block.norm1/norm2 and VisionTransformer.norm (LayerNorm eps 1e-6) as called from
// src/models/pix4point.py:254-256; the positional embedding that the reference re-adds before every
// block (`blk(feats + pos_embed)`, :255) is fused into the forward kernel's prologue.
//
// HBM-bound streaming kernels: one wave per row, 16-B loads/stores, the row lives in registers
// (two-pass mean / variance in fp32), no LDS.  fp32 residual stream in, bf16 (MFMA operand) out.
#include "common.h"

#define LN_THREADS 256

template <int J>   // J 256-element slabs per row: D <= 256*J
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ pos,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps,
                                                                    float* __restrict__ xsum, void* __restrict__ y,
                                                                    int y_f32, float* __restrict__ mean_out,
                                                                    float* __restrict__ rstd_out, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const size_t base = (size_t)row * D;
    f32x4 v[J];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j * 256 + lane * 4;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[j] = *reinterpret_cast<const f32x4*>(x + base + c);
            if (pos) {
                v[j] += *reinterpret_cast<const f32x4*>(pos + base + c);
                if (xsum) *reinterpret_cast<f32x4*>(xsum + base + c) = v[j];
            }
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum_f32(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < D) {
            f32x4 d = v[j] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float var = wave_sum_f32(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < D) {
            f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o = (v[j] - mean) * rstd * g + b;
            if (y_f32) {
                *reinterpret_cast<f32x4*>((float*)y + base + c) = o;
            } else {
                bf16x4 w = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                *reinterpret_cast<bf16x4*>((__bf16*)y + base + c) = w;
            }
        }
    }
}

// dx = accumulate * dx_in + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat))  (dx_in may be dx itself: in place; no
// __restrict__ conflict is exploited on that pair -- each element is read once, then written);  partial[blk][0][c] = sum dy*xhat,
// partial[blk][1][c] = sum dy, partial[blk][2][c] = sum of the OUTPUT dx (the bias gradient of the Linear that
// produced this residual stream: proj.bias / fc2.bias) over the rows this block handled.
template <int J>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(const void* __restrict__ dy, int dy_f32,
                                                                    const float* __restrict__ xin,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma,
                                                                    float* dx, const float* dx_in,
                                                                    int accumulate, __bf16* __restrict__ dx_bf16,
                                                                    float* __restrict__ dacc, int dacc_init,
                                                                    float* __restrict__ partial, int M, int D) {
    __shared__ float red[LN_THREADS / 64][3][256 * J];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = gridDim.x * (LN_THREADS / 64);
    f32x4 ag[J], ab[J], ao[J], g[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        ag[j] = ab[j] = ao[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = j * 256 + lane * 4;
        g[j] = c < D ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x * (LN_THREADS / 64) + wave; row < M; row += nwaves) {
        const size_t base = (size_t)row * D;
        const float mu = mean[row], rs = rstd[row];
        f32x4 d[J], xh[J];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = j * 256 + lane * 4;
            d[j] = xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                if (dy_f32) {
                    d[j] = *reinterpret_cast<const f32x4*>((const float*)dy + base + c);
                } else {
                    bf16x4 w = *reinterpret_cast<const bf16x4*>((const __bf16*)dy + base + c);
                    d[j] = f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
                }
                xh[j] = (*reinterpret_cast<const f32x4*>(xin + base + c) - mu) * rs;
                ab[j] += d[j];
                ag[j] += d[j] * xh[j];
                f32x4 dg = d[j] * g[j];
                s1 += (dg[0] + dg[1]) + (dg[2] + dg[3]);
                f32x4 dgx = dg * xh[j];
                s2 += (dgx[0] + dgx[1]) + (dgx[2] + dgx[3]);
            }
        }
        const float m1 = wave_sum_f32(s1) / (float)D, m2 = wave_sum_f32(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < D) {
                f32x4 o = (d[j] * g[j] - m1 - xh[j] * m2) * rs;
                if (accumulate) o += (float)accumulate * *reinterpret_cast<const f32x4*>(dx_in + base + c);   // 1 or 2 (a residual used twice)
                *reinterpret_cast<f32x4*>(dx + base + c) = o;
                ao[j] += o;
                if (dx_bf16) {
                    bf16x4 w = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    *reinterpret_cast<bf16x4*>(dx_bf16 + base + c) = w;
                }
                if (dacc) {
                    f32x4 a = dacc_init ? o : o + *reinterpret_cast<const f32x4*>(dacc + base + c);
                    *reinterpret_cast<f32x4*>(dacc + base + c) = a;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = j * 256 + lane * 4;
        *reinterpret_cast<f32x4*>(&red[wave][0][c]) = ag[j];
        *reinterpret_cast<f32x4*>(&red[wave][1][c]) = ab[j];
        *reinterpret_cast<f32x4*>(&red[wave][2][c]) = ao[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += LN_THREADS) {
        float a = 0.f, b = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < LN_THREADS / 64; ++w) { a += red[w][0][c]; b += red[w][1][c]; o += red[w][2][c]; }
        partial[((size_t)blockIdx.x * 3 + 0) * D + c] = a;
        partial[((size_t)blockIdx.x * 3 + 1) * D + c] = b;
        partial[((size_t)blockIdx.x * 3 + 2) * D + c] = o;
    }
}

// out[c] = beta*out[c] + sum_r part[r*stride + c]   for c < ncols  (fixed order: reproducible).
// 256 threads = 16 columns x 16 row lanes (64-B row segments): ncols/16 blocks keep >100 CUs busy for the
// 768..2304-column reductions of the encoder; each lane strides the rows by 16, partials meet in LDS.
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                         int nrows, long long stride, int ncols, float beta) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < ncols) {
        int r = ty;
        for (; r + 16 < nrows; r += 32) {
            s0 += part[(size_t)r * stride + c];
            s1 += part[(size_t)(r + 16) * stride + c];
        }
        if (r < nrows) s0 += part[(size_t)r * stride + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < ncols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][tx];
        out[c] = beta != 0.f ? beta * out[c] + s : s;
    }
}

// Same reduction with 16-B loads: 256 threads = 4 column quads x 64 row lanes (16 rows per lane at 1024 partial rows, four
// independent loads in flight per lane).  The scalar kernel above walks 64 rows per lane two at a time and takes its
// ~16 us from that dependent-latency chain, not from the 9 MB it reads.
__global__ __launch_bounds__(256) void colreduce4_kernel(const float* __restrict__ part, float* __restrict__ out, int nrows,
                                                          long long stride, int ncols, float beta) {
    __shared__ f32x4 red[64][4];
    const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
    const int c = (blockIdx.x * 4 + tx) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < ncols) {
        const float* p0 = part + c;
        int r = ty;
        for (; r + 192 < nrows; r += 256) {
            s0 += *reinterpret_cast<const f32x4*>(p0 + (size_t)r * stride);
            s1 += *reinterpret_cast<const f32x4*>(p0 + (size_t)(r + 64) * stride);
            s2 += *reinterpret_cast<const f32x4*>(p0 + (size_t)(r + 128) * stride);
            s3 += *reinterpret_cast<const f32x4*>(p0 + (size_t)(r + 192) * stride);
        }
        for (; r < nrows; r += 64) s0 += *reinterpret_cast<const f32x4*>(p0 + (size_t)r * stride);
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int h = 32; h >= 1; h >>= 1) {
        if (ty < h) red[ty][tx] += red[ty + h][tx];
        __syncthreads();
    }
    if (ty == 0 && c < ncols) {
        f32x4 v = red[0][tx];
        if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(out + c);
        *reinterpret_cast<f32x4*>(out + c) = v;
    }
}
static void launch_colreduce(const float* part, float* out, int nrows, long long stride, int ncols, float beta, hipStream_t st) {
    const bool vec = ncols % 4 == 0 && stride % 4 == 0 && ((reinterpret_cast<size_t>(part) | reinterpret_cast<size_t>(out)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(colreduce4_kernel, dim3(ceil_div(ncols, 16)), dim3(256), 0, st, part, out, nrows, stride, ncols, beta);
    else hipLaunchKernelGGL(colreduce_kernel, dim3(ceil_div(ncols, 16)), dim3(256), 0, st, part, out, nrows, stride, ncols, beta);
}

#define LN_BWD_BLOCKS 1024

extern "C" int p4p_layernorm_bwd_partial_rows(void) { return LN_BWD_BLOCKS; }

extern "C" int p4p_layernorm_fwd(const float* x, const float* pos, const float* gamma, const float* beta, float eps,
                                 float* xsum, void* y, int y_is_f32, float* mean, float* rstd, int M, int D,
                                 void* stream) {
    P4P_CHECK(x && gamma && beta && y, "p4p_layernorm_fwd: null pointer");
    P4P_CHECK(M >= 1 && D >= 4 && D % 4 == 0 && D <= 2048, "p4p_layernorm_fwd: bad sizes M=%d D=%d", M, D);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(M, LN_THREADS / 64)), blk(LN_THREADS);
    int J = ceil_div(D, 256);
#define LN_FWD(JJ) hipLaunchKernelGGL(layernorm_fwd_kernel<JJ>, grid, blk, 0, st, x, pos, gamma, beta, eps, xsum, y, \
                                      y_is_f32, mean, rstd, M, D)
    if (J <= 1) LN_FWD(1); else if (J <= 2) LN_FWD(2); else if (J <= 3) LN_FWD(3); else if (J <= 4) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
    P4P_LAUNCH_CHECK();
    return 0;
}

// partial: [LN_BWD_BLOCKS][3][D] floats of workspace; dgamma/dbeta/dxsum = beta_acc * old + column sums
// (dgamma, dbeta, dxsum contiguous as [3][D] or [2][D] takes one reduce launch).
extern "C" int p4p_layernorm_bwd(const void* dy, int dy_is_f32, const float* xin, const float* mean,
                                 const float* rstd, const float* gamma, float* dx, const float* dx_in, int accumulate_dx,
                                 void* dx_bf16, float* dacc, int dacc_init, float* partial, float* dgamma,
                                 float* dbeta, float* dxsum, float beta_acc, int M, int D, void* stream) {
    if (!dx_in) dx_in = dx;                              // in place
    P4P_CHECK(dy && xin && mean && rstd && gamma && dx && partial, "p4p_layernorm_bwd: null pointer");
    P4P_CHECK(M >= 1 && D >= 4 && D % 4 == 0 && D <= 2048, "p4p_layernorm_bwd: bad sizes M=%d D=%d", M, D);
    hipStream_t st = (hipStream_t)stream;
    int blocks = LN_BWD_BLOCKS;
    if (blocks * (LN_THREADS / 64) > M) blocks = ceil_div(M, LN_THREADS / 64);
    dim3 grid(blocks), blk(LN_THREADS);
    int J = ceil_div(D, 256);
#define LN_BWD(JJ) hipLaunchKernelGGL(layernorm_bwd_kernel<JJ>, grid, blk, 0, st, dy, dy_is_f32, xin, mean, rstd, gamma, \
                                      dx, dx_in, accumulate_dx, (__bf16*)dx_bf16, dacc, dacc_init, partial, M, D)
    if (J <= 1) LN_BWD(1); else if (J <= 2) LN_BWD(2); else if (J <= 3) LN_BWD(3); else if (J <= 4) LN_BWD(4); else LN_BWD(8);
#undef LN_BWD
    P4P_LAUNCH_CHECK();
    if (dgamma && dbeta == dgamma + D && (!dxsum || dxsum == dgamma + 2 * D)) {   // contiguous destination: one launch
        const int rows = dxsum ? 3 : 2;
        launch_colreduce(partial, dgamma, blocks, (long long)3 * D, rows * D, beta_acc, st);
        P4P_LAUNCH_CHECK();
        return 0;
    }
    float* dst[3] = {dgamma, dbeta, dxsum};
    for (int i = 0; i < 3; ++i) {
        if (!dst[i]) continue;
        launch_colreduce(partial + (size_t)i * D, dst[i], blocks, (long long)3 * D, D, beta_acc, st);
        P4P_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int p4p_colreduce_f32(const float* part, float* out, int nrows, long long stride, int ncols, float beta,
                                 void* stream) {
    P4P_CHECK(part && out && nrows >= 1 && ncols >= 1, "p4p_colreduce_f32: bad arguments");
    launch_colreduce(part, out, nrows, stride, ncols, beta, (hipStream_t)stream);
    P4P_LAUNCH_CHECK();
    return 0;
}
