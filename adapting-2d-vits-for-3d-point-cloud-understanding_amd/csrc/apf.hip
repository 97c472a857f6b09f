// AdaptPointFormer tokenizer glue for gfx950.  Index ops reuse fps.hip / knn.hip; this file
// adds what the APF `Group` and `Encoder` need on top (reference src/models/apf.py:12-181, src/models/apf_utils.py:8-104):
//   p4p_morton_order_f32 : 30-bit Morton code of every group centre + the sorting permutation   (apf_utils.py:67-104)
//   p4p_apf_group        : gather + centre-relative concat + Morton reorder of the kNN groups   (apf.py:70-110)
//   p4p_group_max_bf16   : max over the points of a group (+ channel bias), with arg-max         (apf.py:160,167)
//   p4p_group_max_bwd    : the matching scatter of the token gradient
//   p4p_rows_restore_f32 / p4p_rows_zero_bf16 : DropPath (apf_utils.py:280,290) as a row fix-up that touches only the
//                          dropped samples; the 1/keep factor lives in the (frozen) weights
// All HBM-bound byte/integer work: one coalesced pass each, nothing reshaped into a GEMM.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Morton order.  One workgroup per cloud; codes live in LDS, rank = number of (code, index) pairs below one's own
// (G*G/256 compares per thread: 150 for G=196).  Bit-exactness contract (oracle/apf_oracle.py: morton_codes):
//   q = (long)(((p - min) / ((max - min) + 1e-8f)) * 1023.f), every op a separately rounded fp32 op (IEEE division),
//   truncation toward zero; code = part1by2(z) << 2 | part1by2(y) << 1 | part1by2(x); equal codes -> lowest index first.
// ---------------------------------------------------------------------------------------------
#define MORTON_THREADS 256
#define MORTON_MAX_G 4096

__device__ __forceinline__ u32 part1by2(u32 n) {
    n &= 0x000003ffu;
    n = (n ^ (n << 16)) & 0xff0000ffu;
    n = (n ^ (n << 8)) & 0x0300f00fu;
    n = (n ^ (n << 4)) & 0x030c30c3u;
    n = (n ^ (n << 2)) & 0x09249249u;
    return n;
}

__global__ __launch_bounds__(MORTON_THREADS) void morton_order_kernel(const float* __restrict__ centers,
                                                                      int32_t* __restrict__ order,
                                                                      int32_t* __restrict__ codes_out, int G, float res_m1) {
    __shared__ u32 code[MORTON_MAX_G];
    __shared__ float red[2][3][MORTON_THREADS / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* src = centers + (size_t)b * G * 3;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = t; i < G; i += MORTON_THREADS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = src[3 * i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = -wave_max_f32(-mn[c]), z = wave_max_f32(mx[c]);
        if (lane == 0) { red[0][c][wave] = a; red[1][c][wave] = z; }
    }
    __syncthreads();
    float lo[3], inv_den[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = red[0][c][0], z = red[1][c][0];
#pragma unroll
        for (int w = 1; w < MORTON_THREADS / 64; ++w) { a = fminf(a, red[0][c][w]); z = fmaxf(z, red[1][c][w]); }
        lo[c] = a;
        inv_den[c] = __fadd_rn(__fsub_rn(z, a), 1e-8f);       // the DIVISOR (kept as such: the division is IEEE below)
    }
    for (int i = t; i < G; i += MORTON_THREADS) {
        u32 q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float n = __fdiv_rn(__fsub_rn(src[3 * i + c], lo[c]), inv_den[c]);
            q[c] = (u32)(long long)__fmul_rn(n, res_m1);
        }
        const u32 cd = (part1by2(q[2]) << 2) + (part1by2(q[1]) << 1) + part1by2(q[0]);
        code[i] = cd;
        if (codes_out) codes_out[(size_t)b * G + i] = (int32_t)cd;
    }
    __syncthreads();
    for (int i = t; i < G; i += MORTON_THREADS) {
        const u32 mine = code[i];
        int rank = 0;
        for (int j = 0; j < G; ++j) {
            const u32 o = code[j];
            rank += (o < mine || (o == mine && j < i)) ? 1 : 0;
        }
        order[(size_t)b * G + rank] = i;
    }
}

extern "C" int p4p_morton_order_f32(const float* centers, int32_t* order, int32_t* codes, int B, int G, int resolution,
                                    void* stream) {
    P4P_CHECK(centers && order, "p4p_morton_order_f32: null pointer");
    P4P_CHECK(B >= 0 && G >= 1 && G <= MORTON_MAX_G, "p4p_morton_order_f32: bad sizes B=%d G=%d (G <= %d)", B, G, MORTON_MAX_G);
    P4P_CHECK(resolution >= 2 && resolution <= 1024, "p4p_morton_order_f32: resolution=%d outside [2,1024] (10 bits per axis)",
              resolution);
    if (B == 0) return 0;
    hipLaunchKernelGGL(morton_order_kernel, dim3(B), dim3(MORTON_THREADS), 0, (hipStream_t)stream, centers, order, codes, G,
                       (float)(resolution - 1));
    P4P_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Group: for output group g' of cloud b (Morton position), src = order[b][g'], centre point c = fps_idx[b][src]:
//   six[j] = [x[nbr_j] - x[c], x[c]]  (apf.py:83-95), rows in Morton order (apf.py:102-110).
// Outputs (each optional):
//   nb_f32  [B,G,gs,6]  fp32  - what the reference's Group.forward returns
//   x0_bf16 [B*G*gs,64] bf16  - the first GEMM's operand: cols 0-5 hi = bf16(six), 6-11 lo = bf16(six - hi), 12-17 hi again,
//                               rest 0.  With weight columns [W_hi, W_hi, W_lo, 0] the bf16 MFMA product carries
//                               x_hi*W_hi + x_lo*W_hi + x_hi*W_lo: fp32-class accuracy for the K=6 layer at no extra cost
//                               (the GEMM's K is padded to 64 anyway).
//   center  [B,G,3]     fp32  - centres in Morton order
// One thread per output point row; 128 B per row written as 8 x 16 B.
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ void apf_group_kernel(const float* __restrict__ x, const int64_t* __restrict__ fps_idx,
                                 const int32_t* __restrict__ knn_idx, const int32_t* __restrict__ order,
                                 float* __restrict__ nb_f32, __bf16* __restrict__ x0, float* __restrict__ center, int N,
                                 int G, int gs, long long total) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total) return;
    const int j = (int)(r % gs);
    const long long bg = r / gs;
    const int gq = (int)(bg % G);
    const long long b = bg / G;
    const int src = order[b * G + gq];
    const long long c = fps_idx[b * G + src];
    const int n = knn_idx[(b * G + src) * gs + j];
    const float* pc = x + (b * N + c) * C;
    const float* pn = x + (b * N + n) * C;
    float f[2 * C];                      // [x[nbr] - x[centre] (C), x[centre] (C)]   (apf.py:83-95)
#pragma unroll
    for (int i = 0; i < C; ++i) {
        f[C + i] = pc[i];
        f[i] = __fsub_rn(pn[i], pc[i]);
    }
    if (nb_f32) {
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) nb_f32[r * (2 * C) + i] = f[i];
    }
    if (center && j == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) center[bg * 3 + i] = pc[i];
    }
    if (x0) {
        // [hi (2C) | lo (2C) | hi (2C) | 0 ...]: partner of functional_apf._split_hi_lo_weight's [W_hi | W_hi | W_lo | 0]
        __bf16 row[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) row[i] = f2bf(0.f);
#pragma unroll
        for (int i = 0; i < 2 * C; ++i) {
            const __bf16 h = f2bf(f[i]);
            row[i] = h;
            row[2 * C + i] = f2bf(f[i] - bf2f(h));
            row[4 * C + i] = h;
        }
        bf16x8* dst = reinterpret_cast<bf16x8*>(x0 + r * 64);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            bf16x8 w;
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = row[8 * v + i];
            dst[v] = w;
        }
    }
}

extern "C" int p4p_apf_group(const float* x, const int64_t* fps_idx, const int32_t* knn_idx, const int32_t* order,
                             float* nb_f32, void* x0_bf16, float* center, int B, int N, int C, int G, int gs, void* stream) {
    P4P_CHECK(x && fps_idx && knn_idx && order, "p4p_apf_group: null pointer");
    P4P_CHECK(nb_f32 || x0_bf16 || center, "p4p_apf_group: no output requested");
    P4P_CHECK(B >= 0 && N >= 1 && G >= 1 && gs >= 1, "p4p_apf_group: bad sizes B=%d N=%d G=%d gs=%d", B, N, G, gs);
    P4P_CHECK(C == 3 || C == 4, "p4p_apf_group: built for C = 3 (xyz) and C = 4 (xyz + height), got C=%d", C);
    const long long total = (long long)B * G * gs;
    if (total == 0) return 0;
    if (C == 3)
        hipLaunchKernelGGL(apf_group_kernel<3>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, fps_idx,
                           knn_idx, order, nb_f32, (__bf16*)x0_bf16, center, N, G, gs, total);
    else
        hipLaunchKernelGGL(apf_group_kernel<4>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, fps_idx,
                           knn_idx, order, nb_f32, (__bf16*)x0_bf16, center, N, G, gs, total);
    P4P_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// tok[g][c] = max_j H[g*k+j][c] (+ bias[c]); first maximum wins (torch.max's CPU rule); thread = (g, channel quad)
// ---------------------------------------------------------------------------------------------
__global__ void group_max_kernel(const __bf16* __restrict__ H, const float* __restrict__ bias, float* __restrict__ tok_f32,
                                 __bf16* __restrict__ tok_bf16, uint8_t* __restrict__ arg, long long G, int k, int C) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= G * C) return;
    const long long g = e / C;
    const int c = (int)(e - g * C);
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int am[4] = {0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const bf16x4 w = *reinterpret_cast<const bf16x4*>(H + ((size_t)g * k + j) * C + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = bf2f(w[i]);
            if (v > mx[i]) { mx[i] = v; am[i] = j; }
        }
    }
    if (bias) mx += *reinterpret_cast<const f32x4*>(bias + c);
    if (tok_f32) *reinterpret_cast<f32x4*>(tok_f32 + e) = mx;
    if (tok_bf16) {
        const bf16x4 o = {f2bf(mx[0]), f2bf(mx[1]), f2bf(mx[2]), f2bf(mx[3])};
        *reinterpret_cast<bf16x4*>(tok_bf16 + e) = o;
    }
    if (arg) *reinterpret_cast<uint32_t*>(arg + e) = (u32)am[0] | ((u32)am[1] << 8) | ((u32)am[2] << 16) | ((u32)am[3] << 24);
}

// dH[g*k+j][c] = (accumulate ? dH : 0) + (j == arg[g][c] ? dtok[g][c] : 0)
__global__ void group_max_bwd_kernel(const void* __restrict__ dtok, int dtok_f32, const uint8_t* __restrict__ arg,
                                     __bf16* __restrict__ dH, int accumulate, long long G, int k, int C) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= G * C) return;
    const long long g = e / C;
    const int c = (int)(e - g * C);
    f32x4 d;
    if (dtok_f32) {
        d = *reinterpret_cast<const f32x4*>((const float*)dtok + e);
    } else {
        const bf16x4 w = *reinterpret_cast<const bf16x4*>((const __bf16*)dtok + e);
        d = f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
    }
    const u32 pk = *reinterpret_cast<const uint32_t*>(arg + e);
    for (int j = 0; j < k; ++j) {
        bf16x4* p = reinterpret_cast<bf16x4*>(dH + ((size_t)g * k + j) * C + c);
        bool hit[4], any = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hit[i] = j == (int)((pk >> (8 * i)) & 255);
            any |= hit[i];
        }
        if (accumulate && !any) continue;                 // nothing to add to this row
        bf16x4 o = {};
        if (accumulate) o = *p;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (hit[i]) o[i] = f2bf((accumulate ? bf2f(o[i]) : 0.f) + d[i]);
        *p = o;
    }
}

extern "C" int p4p_group_max_bf16(const void* H, const float* bias, float* tok_f32, void* tok_bf16, uint8_t* arg, long long G,
                                  int k, int C, void* stream) {
    P4P_CHECK(H && (tok_f32 || tok_bf16), "p4p_group_max_bf16: null pointer");
    P4P_CHECK(G >= 1 && k >= 1 && k <= 255 && C >= 4 && C % 4 == 0, "p4p_group_max_bf16: bad sizes G=%lld k=%d C=%d", G, k, C);
    hipLaunchKernelGGL(group_max_kernel, dim3(ceil_div(G * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)H,
                       bias, tok_f32, (__bf16*)tok_bf16, arg, G, k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_group_max_bwd(const void* dtok, int dtok_is_f32, const uint8_t* arg, void* dH, int accumulate, long long G,
                                 int k, int C, void* stream) {
    P4P_CHECK(dtok && arg && dH, "p4p_group_max_bwd: null pointer");
    P4P_CHECK(G >= 1 && k >= 1 && k <= 255 && C >= 4 && C % 4 == 0, "p4p_group_max_bwd: bad sizes G=%lld k=%d C=%d", G, k, C);
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(ceil_div(G * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, dtok,
                       dtok_is_f32, arg, (__bf16*)dH, accumulate, G, k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// DropPath without an elementwise pass.  x1 = x + (mask/keep) * branch(x): the GEMM that ends the branch runs with weights
// and bias pre-scaled by 1/keep and adds the residual for EVERY sample; afterwards the rows of the dropped samples
// (keep[b] == 0, ~p of them) are put back to the residual.  In the backward the branch gradient of those samples is zeroed
// the same way.  Workgroups of kept samples return after one byte load.
// ---------------------------------------------------------------------------------------------
#define ROWFIX_SLABS 32
__global__ void rows_restore_kernel(float* __restrict__ dst, const float* __restrict__ src, const uint8_t* __restrict__ keep,
                                    long long per_sample) {
    const int b = blockIdx.y;
    if (keep[b]) return;
    const long long slab = (per_sample / 4 + ROWFIX_SLABS - 1) / ROWFIX_SLABS;
    const long long lo = (long long)blockIdx.x * slab, hi = lo + slab < per_sample / 4 ? lo + slab : per_sample / 4;
    f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)b * per_sample);
    const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)b * per_sample);
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) d[i] = s[i];
}

__global__ void rows_zero_kernel(__bf16* __restrict__ x, const uint8_t* __restrict__ keep, long long per_sample) {
    const int b = blockIdx.y;
    if (keep[b]) return;
    const long long slab = (per_sample / 8 + ROWFIX_SLABS - 1) / ROWFIX_SLABS;
    const long long lo = (long long)blockIdx.x * slab, hi = lo + slab < per_sample / 8 ? lo + slab : per_sample / 8;
    bf16x8* d = reinterpret_cast<bf16x8*>(x + (size_t)b * per_sample);
    const bf16x8 z = {};
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) d[i] = z;
}

/* dst[b] = src[b] for every sample b with keep[b] == 0; dst, src [B, per_sample] f32, per_sample % 4 == 0 */
extern "C" int p4p_rows_restore_f32(float* dst, const float* src, const uint8_t* keep, int B, long long per_sample,
                                    void* stream) {
    P4P_CHECK(dst && src && keep, "p4p_rows_restore_f32: null pointer");
    P4P_CHECK(B >= 1 && B <= 65535 && per_sample >= 4 && per_sample % 4 == 0, "p4p_rows_restore_f32: bad sizes B=%d per_sample=%lld",
              B, per_sample);
    hipLaunchKernelGGL(rows_restore_kernel, dim3(ROWFIX_SLABS, B), dim3(256), 0, (hipStream_t)stream, dst, src, keep, per_sample);
    P4P_LAUNCH_CHECK();
    return 0;
}

/* x[b] = 0 for every sample b with keep[b] == 0; x [B, per_sample] bf16, per_sample % 8 == 0 */
extern "C" int p4p_rows_zero_bf16(void* x, const uint8_t* keep, int B, long long per_sample, void* stream) {
    P4P_CHECK(x && keep, "p4p_rows_zero_bf16: null pointer");
    P4P_CHECK(B >= 1 && B <= 65535 && per_sample >= 8 && per_sample % 8 == 0, "p4p_rows_zero_bf16: bad sizes B=%d per_sample=%lld", B,
              per_sample);
    hipLaunchKernelGGL(rows_zero_kernel, dim3(ROWFIX_SLABS, B), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, keep, per_sample);
    P4P_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// AdapterLayer bottleneck (apf_utils.py:222-226): down = dropout(relu(pre)) on the [M, 64] tensor between the two adapter
// GEMMs, as one launch each way.  u (may be NULL) = uniform [0,1) draws for the dropout (kept where u >= p, scaled by
// 1/(1-p)); gate = relu'(pre) * dropout factor is what the backward multiplies by.
// ---------------------------------------------------------------------------------------------
__global__ void adapter_mid_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ u, float p, float inv_keep,
                                       __bf16* __restrict__ down, float* __restrict__ gate, long long n) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(pre + e);
    f32x4 g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = v[i] > 0.f ? 1.f : 0.f;
    if (u) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(u + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = r[i] >= p ? g[i] * inv_keep : 0.f;
    }
    const bf16x4 o = {f2bf(v[0] * g[0]), f2bf(v[1] * g[1]), f2bf(v[2] * g[2]), f2bf(v[3] * g[3])};
    *reinterpret_cast<bf16x4*>(down + e) = o;
    *reinterpret_cast<f32x4*>(gate + e) = g;
}

__global__ void adapter_mid_bwd_kernel(const float* __restrict__ ddown, const float* __restrict__ gate,
                                       __bf16* __restrict__ dpre, long long n) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n) return;
    const f32x4 d = *reinterpret_cast<const f32x4*>(ddown + e), g = *reinterpret_cast<const f32x4*>(gate + e);
    const bf16x4 o = {f2bf(d[0] * g[0]), f2bf(d[1] * g[1]), f2bf(d[2] * g[2]), f2bf(d[3] * g[3])};
    *reinterpret_cast<bf16x4*>(dpre + e) = o;
}

extern "C" int p4p_adapter_mid_fwd(const float* pre, const float* u, float p, void* down_bf16, float* gate, long long n,
                                   void* stream) {
    P4P_CHECK(pre && down_bf16 && gate, "p4p_adapter_mid_fwd: null pointer");
    P4P_CHECK(n >= 4 && n % 4 == 0 && p >= 0.f && p < 1.f, "p4p_adapter_mid_fwd: bad arguments n=%lld p=%f", n, (double)p);
    hipLaunchKernelGGL(adapter_mid_fwd_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, pre,
                       p > 0.f ? u : nullptr, p, 1.f / (1.f - p), (__bf16*)down_bf16, gate, n);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_adapter_mid_bwd(const float* ddown, const float* gate, void* dpre_bf16, long long n, void* stream) {
    P4P_CHECK(ddown && gate && dpre_bf16, "p4p_adapter_mid_bwd: null pointer");
    P4P_CHECK(n >= 4 && n % 4 == 0, "p4p_adapter_mid_bwd: bad size n=%lld", n);
    hipLaunchKernelGGL(adapter_mid_bwd_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, ddown, gate,
                       (__bf16*)dpre_bf16, n);
    P4P_LAUNCH_CHECK();
    return 0;
}
