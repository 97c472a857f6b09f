// P3Embed local-MLP kernels (everything around the three big GEMMs), gfx950.
// Reference: P3Embed.forward, src/models/pix4point.py:166-191, module tree :137-156:
//   x6 = cat[grouped xyz, grouped feats]                      (:178-182, absolute coordinates)
//   h1 = Conv(256,256,bias)(Conv(6,256)(x6)); f1 = ReLU(BN1(h1))          (convs[0], :137-145)
//   h2 = Conv(512,512)(cat[max_k f1, f1]);    f2 = ReLU(BN2(h2))          (:184-186, convs[1][0:3])
//   h3 = Conv(512,256)(f2);  tokens = max_k ReLU(BN3(h3))                 (convs[1][3:6], :188)
//
// Algebra used (exact up to fp32 rounding, documented in DESIGN.md):
//   * no nonlinearity sits between the first two convs, so h1 = Aeff x6 + b2 with Aeff = W2 W1 (256x6);
//     BN1's batch statistics follow from the 6-vector mean and 6x6 second moments of x6:
//     mean_c = Aeff_c.mu + b2_c, var_c = Aeff_c^T Cov Aeff_c.  Stage 1 therefore costs 6 MACs per
//     (position, channel) instead of 262, and its backward needs only sum(dy) and sum(dy (x) x6).
//   * half of conv2a's input is max_k f1 broadcast over k: W3 [pooled; f1] = W3a pooled + W3b f1, the
//     first term computed once per group and added in the GEMM epilogue (P4P_EPI_BF16_ROWGROUP).
// Activations between kernels are bf16 [positions, channels] (channel-contiguous, position = (b,s,j)).
#include "common.h"

extern "C" int p4p_colreduce_f32(const float* part, float* out, int nrows, long long stride, int ncols, float beta,
                                 void* stream);

#define X6 6
#define NMOM 27          // 6 sums + 21 upper-triangular products

// ---------------------------------------------------------------------------------------------
// E1: first and second moments of the gathered 6-vectors.  partial[blk][27]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_x6(const float* __restrict__ points, const float* __restrict__ feats,
                                        const int32_t* __restrict__ idx, long long pos, int N, int SK, float* x) {
    const long long b = pos / SK;
    const size_t row = ((size_t)b * N + idx[pos]) * 3;
    x[0] = points[row]; x[1] = points[row + 1]; x[2] = points[row + 2];
    x[3] = feats[row];  x[4] = feats[row + 1];  x[5] = feats[row + 2];
}

__global__ __launch_bounds__(256) void embed_moments_kernel(const float* __restrict__ points,
                                                            const float* __restrict__ feats,
                                                            const int32_t* __restrict__ idx, float* __restrict__ partial,
                                                            long long P, int N, int SK) {
    __shared__ float red[4][NMOM];
    float acc[NMOM];
#pragma unroll
    for (int i = 0; i < NMOM; ++i) acc[i] = 0.f;
    for (long long pos = (long long)blockIdx.x * 256 + threadIdx.x; pos < P; pos += (long long)gridDim.x * 256) {
        float x[X6];
        load_x6(points, feats, idx, pos, N, SK, x);
        int q = X6;
#pragma unroll
        for (int a = 0; a < X6; ++a) {
            acc[a] += x[a];
#pragma unroll
            for (int b = a; b < X6; ++b) acc[q++] += x[a] * x[b];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NMOM; ++i) {
        float s = wave_sum_f32(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NMOM)
        partial[(size_t)blockIdx.x * NMOM + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// E2: fold conv1a, conv1b, BN1 into  f1 = relu(scale * (Aeff . x6) + shift).   One block, 256 threads.
// mom[42] = Sx[6] | M2[6][6] (raw sums over all positions), kept for backward.
// ---------------------------------------------------------------------------------------------
struct FoldParams {
    const float* W1;        // [C1,6]
    const float* W2;        // [C1,C1]
    const float* b2;        // [C1]
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    const float* mom_partial; int mom_rows;
    float* aeff;            // [C1,8]
    float* scale; float* shift; float* save_mean; float* save_invstd;
    float* mom;             // [42]
    double Np; float eps, momentum; int training, C1;
};

__global__ __launch_bounds__(256) void embed_fold_kernel(FoldParams p) {
    __shared__ double sm[NMOM];
    __shared__ double smp[8][32];
    __shared__ float w1s[256 * X6];
    const int c = threadIdx.x;
    for (int i = c; i < p.C1 * X6; i += 256) w1s[i] = p.W1[i];
    {   // partial moments -> sm: 32 moment slots x 8 row lanes, fixed order (reproducible), fp64
        const int mi = c & 31, rl = c >> 5;
        double s = 0.0;
        if (p.training && mi < NMOM)
            for (int r = rl; r < p.mom_rows; r += 8) s += (double)p.mom_partial[(size_t)r * NMOM + mi];
        smp[rl][mi] = s;
    }
    __syncthreads();
    if (c < NMOM) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += smp[k][c];
        sm[c] = s;
    }
    __syncthreads();
    if (c < 42 && p.training) {
        if (c < 6) p.mom[c] = (float)sm[c];
        else {
            int a = (c - 6) / 6, b = (c - 6) % 6;
            int lo = a < b ? a : b, hi = a < b ? b : a;
            int q = 6 + lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
            p.mom[c] = (float)sm[q];
        }
    }
    if (c >= p.C1) return;
    float a[X6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < p.C1; ++j) {
        const float w = p.W2[(size_t)c * p.C1 + j];
#pragma unroll
        for (int d = 0; d < X6; ++d) a[d] += w * w1s[j * X6 + d];
    }
#pragma unroll
    for (int d = 0; d < X6; ++d) p.aeff[c * 8 + d] = a[d];
    p.aeff[c * 8 + 6] = p.aeff[c * 8 + 7] = 0.f;
    float mean, var;
    if (p.training) {
        double mu[X6], am = 0.0, v = 0.0;
#pragma unroll
        for (int d = 0; d < X6; ++d) { mu[d] = sm[d] / p.Np; am += (double)a[d] * mu[d]; }
        int q = X6;
#pragma unroll
        for (int x = 0; x < X6; ++x)
#pragma unroll
            for (int y = x; y < X6; ++y) {
                double cov = sm[q++] / p.Np - mu[x] * mu[y];
                v += (x == y ? 1.0 : 2.0) * (double)a[x] * (double)a[y] * cov;
            }
        mean = (float)(am + (double)p.b2[c]);
        var = v > 0.0 ? (float)v : 0.f;
        const float unb = (float)((double)var * p.Np / (p.Np > 1.0 ? p.Np - 1.0 : 1.0));
        p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
        p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * unb;
    } else {
        mean = p.running_mean[c];
        var = p.running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + p.eps);
    const float sc = p.gamma[c] * invstd;
    p.scale[c] = sc;
    p.shift[c] = p.beta[c] + (p.b2[c] - mean) * sc;
    p.save_mean[c] = mean;
    p.save_invstd[c] = invstd;
}

// ---------------------------------------------------------------------------------------------
// E3: stage 1.  block = 256 threads = 256 channels, loops over chunks of GPB groups.
//     F1[pos][c] = relu(scale_c * Aeff_c.x6[pos] + shift_c)  (bf16), pooled[g][c] = max_j, arg[g][c] = argmax_j
// ---------------------------------------------------------------------------------------------
#define GPB 8
__global__ __launch_bounds__(256) void embed_stage1_kernel(const float* __restrict__ points,
                                                           const float* __restrict__ feats,
                                                           const int32_t* __restrict__ idx,
                                                           const float* __restrict__ aeff,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, __bf16* __restrict__ F1,
                                                           __bf16* __restrict__ pooled, uint8_t* __restrict__ arg,
                                                           long long G, int N, int S, int k, int C1) {
    extern __shared__ float xs[];          // [GPB*k][8]
    const int c = threadIdx.x;
    float a[X6];
#pragma unroll
    for (int d = 0; d < X6; ++d) a[d] = c < C1 ? aeff[c * 8 + d] : 0.f;
    const float sc = c < C1 ? scale[c] : 0.f, sh = c < C1 ? shift[c] : 0.f;
    const int SK = S * k;
    for (long long g0 = (long long)blockIdx.x * GPB; g0 < G; g0 += (long long)gridDim.x * GPB) {
        const int ng = (int)((G - g0) < GPB ? (G - g0) : GPB);
        __syncthreads();
        for (int i = threadIdx.x; i < ng * k; i += 256) {
            float x[X6];
            load_x6(points, feats, idx, g0 * k + i, N, SK, x);
#pragma unroll
            for (int d = 0; d < X6; ++d) xs[i * 8 + d] = x[d];
        }
        __syncthreads();
        if (c < C1) {
            for (int g = 0; g < ng; ++g) {
                float mx = -1.f;
                int am = 0;
                for (int j = 0; j < k; ++j) {
                    const float* x = xs + (g * k + j) * 8;
                    float h = ((a[0] * x[0] + a[1] * x[1]) + (a[2] * x[2] + a[3] * x[3])) + (a[4] * x[4] + a[5] * x[5]);
                    float v = fmaxf(sc * h + sh, 0.f);
                    F1[((size_t)(g0 + g) * k + j) * C1 + c] = f2bf(v);
                    if (v > mx) { mx = v; am = j; }
                }
                pooled[(size_t)(g0 + g) * C1 + c] = f2bf(mx);
                arg[(size_t)(g0 + g) * C1 + c] = (uint8_t)am;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize from column sums: sums[0][c] = sum x, sums[1][c] = sum x^2 over Np rows.
// ---------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd, double Np, float eps,
                                   float momentum, int training, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        const double m = (double)sums[c] / Np;
        double v = (double)sums[C + c] / Np - m * m;
        v = v > 0.0 ? v : 0.0;
        mean = (float)m;
        var = (float)v;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * Np / (Np > 1.0 ? Np - 1.0 : 1.0));
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    scale[c] = gamma[c] * invstd;
    shift[c] = beta[c] - mean * gamma[c] * invstd;
    save_mean[c] = mean;
    save_invstd[c] = invstd;
}

// y = relu(scale[c] * x + shift[c]), bf16 -> bf16, [rows, C], C % 4 == 0; V = 4 or 8 elements per thread
template <int V>
__global__ void bn_relu_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, const float* __restrict__ scale,
                                    const float* __restrict__ shift, long long n, int C) {
    typedef __attribute__((ext_vector_type(V))) __bf16 bvec;
    long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (e >= n) return;
    const int c = (int)(e % C);
    const bvec w = *reinterpret_cast<const bvec*>(x + e);
    bvec o;
#pragma unroll
    for (int q = 0; q < V; q += 4) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c + q), sh = *reinterpret_cast<const f32x4*>(shift + c + q);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[q + i] = f2bf(fmaxf(sc[i] * bf2f(w[q + i]) + sh[i], 0.f));
    }
    *reinterpret_cast<bvec*>(y + e) = o;
}

// Column-resident variants for C % 8 == 0 (every width on the path): a thread owns 8 fixed channels, keeps their BatchNorm
// constants in registers and walks down the rows, so the streams are pure 16-byte loads / stores with no per-element
// index division and no per-element constant loads.  grid = (C / 8 / tpr, row blocks); block = 256 threads =
// (256 / tpr) rows x tpr column octets, tpr = gcd(C / 8, 256).
#define BNC_ROWS 256
static int bnc_tpr(int C) {            // gcd(C / 8, 256): a power of two
    int t = C / 8, g = 1;
    while (g < 256 && t % (2 * g) == 0) g *= 2;
    return g;
}
// GSUM (backward only, round 5): the same pass also emits gsum[g][c] = sum over the k rows of group g of the bf16 values it writes (fp32
// sum in row order, rounded once: the bits of group_sum_bf16_kernel on the finished tensor, without its 1.07-GB read).  A thread then
// walks a CONTIGUOUS run of BNC_ROWS / (256 / tpr) rows (whole groups: the launcher checks the divisibility), a wave still reads one
// full row per iteration.
template <bool BWD, bool GSUM = false>
__global__ __launch_bounds__(256) void bn_colwise_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y,
                                                         const __bf16* __restrict__ H, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ sums,
                                                         float invNp, long long rows, int C, int tpr,
                                                         __bf16* __restrict__ gsum = nullptr, int k = 1) {
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr, rpp = 256 / tpr;
    const int c = (blockIdx.x * tpr + tc) * 8;
    float sc[8], sh[8], a[8], b[8];          // BWD: dH = sc*dy*mask - a - b*h  with a = sc*(m1 - mu*is*m2), b = sc*is*m2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = scale[c + i];
        sh[i] = shift[c + i];
        if (BWD) {
            const float m1 = sums[c + i] * invNp, m2 = sums[C + c + i] * invNp, mu = mean[c + i], is = invstd[c + i];
            a[i] = sc[i] * (m1 - mu * is * m2);
            b[i] = sc[i] * is * m2;
        }
    }
    const long long r0 = (long long)blockIdx.y * BNC_ROWS, r1 = r0 + BNC_ROWS < rows ? r0 + BNC_ROWS : rows;
    const int run = BNC_ROWS / rpp;                                   // GSUM: rows per thread, a multiple of k
    float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int inrun = 0;
    for (long long r = GSUM ? r0 + (long long)tr * run : r0 + tr; r < (GSUM ? (r0 + (long long)(tr + 1) * run < r1 ? r0 + (long long)(tr + 1) * run : r1) : r1);
         r += GSUM ? 1 : rpp) {
        const size_t o = (size_t)r * C + c;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + o);
        bf16x8 out;
        if (BWD) {
            const bf16x8 h = *reinterpret_cast<const bf16x8*>(H + o);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float hv = bf2f(h[i]);
                const float dy = (sc[i] * hv + sh[i] > 0.f) ? bf2f(v[i]) : 0.f;
                out[i] = f2bf(sc[i] * dy - a[i] - b[i] * hv);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = f2bf(fmaxf(sc[i] * bf2f(v[i]) + sh[i], 0.f));
        }
        *reinterpret_cast<bf16x8*>(y + o) = out;
        if (GSUM) {
#pragma unroll
            for (int i = 0; i < 8; ++i) gs[i] += bf2f(out[i]);
            if (++inrun == k) {
                bf16x8 g8;
#pragma unroll
                for (int i = 0; i < 8; ++i) { g8[i] = f2bf(gs[i]); gs[i] = 0.f; }
                *reinterpret_cast<bf16x8*>(gsum + (size_t)(r / k) * C + c) = g8;
                inrun = 0;
            }
        }
    }
}

// tokens[g][c] = max_j relu(scale*H3[g*k+j][c] + shift); thread = (g, channel quad)
__global__ void bn_relu_maxpool_kernel(const __bf16* __restrict__ H3, const float* __restrict__ scale,
                                       const float* __restrict__ shift, __bf16* __restrict__ tok_bf16,
                                       float* __restrict__ tok_f32, uint8_t* __restrict__ arg, long long G, int k, int C) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= G * C) return;
    const long long g = e / C;
    const int c = (int)(e - g * C);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 mx = {-1.f, -1.f, -1.f, -1.f};
    int am[4] = {0, 0, 0, 0};
    for (int j = 0; j < k; ++j) {
        bf16x4 w = *reinterpret_cast<const bf16x4*>(H3 + ((size_t)g * k + j) * C + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = fmaxf(sc[i] * bf2f(w[i]) + sh[i], 0.f);
            if (v > mx[i]) { mx[i] = v; am[i] = j; }
        }
    }
    if (tok_bf16) {
        bf16x4 o = {f2bf(mx[0]), f2bf(mx[1]), f2bf(mx[2]), f2bf(mx[3])};
        *reinterpret_cast<bf16x4*>(tok_bf16 + e) = o;
    }
    if (tok_f32) *reinterpret_cast<f32x4*>(tok_f32 + e) = mx;
    if (arg) {
        uint32_t pk = (uint32_t)am[0] | ((uint32_t)am[1] << 8) | ((uint32_t)am[2] << 16) | ((uint32_t)am[3] << 24);
        *reinterpret_cast<uint32_t*>(arg + e) = pk;
    }
}

// ---------------------------------------------------------------------------------------------
// backward B1/B2: max-pool + ReLU + BN3.  Only the arg-max position of each (g,c) carries gradient.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 load4_any(const void* p, long long e, int is_f32) {
    if (is_f32) return *reinterpret_cast<const f32x4*>((const float*)p + e);
    bf16x4 w = *reinterpret_cast<const bf16x4*>((const __bf16*)p + e);
    return f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
}

// partial[blk][2][C]: sum dy, sum dy*xhat.  256 threads = 64 channel quads x 4 group lanes.
__global__ __launch_bounds__(256) void pool_bn_bwd_stats_kernel(const void* __restrict__ dtok, int dtok_f32,
                                                                const __bf16* __restrict__ H3,
                                                                const uint8_t* __restrict__ arg,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                float* __restrict__ partial, long long G, int k, int C,
                                                                int groups_per_block) {
    __shared__ f32x4 red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + tx * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c < C) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        const long long g_begin = (long long)blockIdx.y * groups_per_block;
        const long long g_end = g_begin + groups_per_block < G ? g_begin + groups_per_block : G;
        for (long long g = g_begin + ty; g < g_end; g += 4) {
            const f32x4 d = load4_any(dtok, g * C + c, dtok_f32);
            const uint32_t pk = *reinterpret_cast<const uint32_t*>(arg + g * C + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = (pk >> (8 * i)) & 255;
                const float h = bf2f(H3[((size_t)g * k + j) * C + c + i]);
                const float dy = (sc[i] * h + sh[i] > 0.f) ? d[i] : 0.f;
                s0[i] += dy;
                s1[i] += dy * (h - mu[i]) * is[i];
            }
        }
    }
    red[0][ty][tx] = s0; red[1][ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && c < C) {
        f32x4 a = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        f32x4 b = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 0) * C + c) = a;
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 1) * C + c) = b;
    }
}

// dH3[pos][c] = gamma*invstd*(dy - sums[0]/Np - xhat*sums[1]/Np), dy nonzero only at the arg-max row
__global__ void pool_bn_bwd_apply_kernel(const void* __restrict__ dtok, int dtok_f32, const __bf16* __restrict__ H3,
                                         const uint8_t* __restrict__ arg, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                         const float* __restrict__ invstd, const float* __restrict__ sums, float invNp,
                                         __bf16* __restrict__ dH3, long long G, int k, int C) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= G * C) return;
    const long long g = e / C;
    const int c = (int)(e - g * C);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(sums + c) * invNp;
    const f32x4 m2 = *reinterpret_cast<const f32x4*>(sums + C + c) * invNp;
    const f32x4 d = load4_any(dtok, e, dtok_f32);
    const uint32_t pk = *reinterpret_cast<const uint32_t*>(arg + e);
    for (int j = 0; j < k; ++j) {
        const size_t o = ((size_t)g * k + j) * C + c;
        bf16x4 w = *reinterpret_cast<const bf16x4*>(H3 + o);
        bf16x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h = bf2f(w[i]);
            const float dy = (j == (int)((pk >> (8 * i)) & 255) && sc[i] * h + sh[i] > 0.f) ? d[i] : 0.f;
            r[i] = f2bf(sc[i] * (dy - m1[i] - (h - mu[i]) * is[i] * m2[i]));     // sc = gamma*invstd
        }
        *reinterpret_cast<bf16x4*>(dH3 + o) = r;
    }
}

// ---------------------------------------------------------------------------------------------
// backward B3/B4: ReLU + BN over a dense [rows, C] gradient.  The statistics pass only reads; the apply pass re-derives the
// ReLU mask from H and rewrites dY in place (one write of the gradient instead of two).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bn_bwd_stats_kernel(const __bf16* __restrict__ dY, const __bf16* __restrict__ H,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                float* __restrict__ partial, long long rows, int C,
                                                                int rows_per_block) {
    __shared__ f32x4 red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + tx * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c < C) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        const long long r_begin = (long long)blockIdx.y * rows_per_block;
        const long long r_end = r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows;
        for (long long r = r_begin + ty; r < r_end; r += 4) {
            const size_t o = (size_t)r * C + c;
            const bf16x4 d = *reinterpret_cast<const bf16x4*>(dY + o);
            const bf16x4 h = *reinterpret_cast<const bf16x4*>(H + o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hv = bf2f(h[i]);
                const float dy = (sc[i] * hv + sh[i] > 0.f) ? bf2f(d[i]) : 0.f;     // the apply pass re-derives the mask
                s0[i] += dy;
                s1[i] += dy * (hv - mu[i]) * is[i];
            }
        }
    }
    red[0][ty][tx] = s0; red[1][ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && c < C) {
        f32x4 a = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        f32x4 b = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 0) * C + c) = a;
        *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.y * 2 + 1) * C + c) = b;
    }
}

// dH = gamma*invstd*(relu'(y) dy - sums[0]/Np - xhat*sums[1]/Np), in place over dY; V = 4 or 8 elements per thread
template <int V>
__global__ void bn_bwd_apply_kernel(__bf16* __restrict__ dY, const __bf16* __restrict__ H,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ sums, float invNp, long long n, int C) {
    typedef __attribute__((ext_vector_type(V))) __bf16 bvec;
    long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (e >= n) return;
    const int c = (int)(e % C);
    bvec d = *reinterpret_cast<const bvec*>(dY + e);
    const bvec h = *reinterpret_cast<const bvec*>(H + e);
#pragma unroll
    for (int q = 0; q < V; q += 4) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c + q), sh = *reinterpret_cast<const f32x4*>(shift + c + q);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c + q), is = *reinterpret_cast<const f32x4*>(invstd + c + q);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(sums + c + q) * invNp;
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(sums + C + c + q) * invNp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hv = bf2f(h[q + i]);
            const float dy = (sc[i] * hv + sh[i] > 0.f) ? bf2f(d[q + i]) : 0.f;
            d[q + i] = f2bf(sc[i] * (dy - m1[i] - (hv - mu[i]) * is[i] * m2[i]));
        }
    }
    *reinterpret_cast<bvec*>(dY + e) = d;
}

// out[g][c] = sum_j X[g*k+j][c]   (bf16 in, bf16 out, fp32 sum)
__global__ void group_sum_bf16_kernel(const __bf16* __restrict__ X, __bf16* __restrict__ out, long long G, int k, int C) {
    const long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= G * C) return;
    const long long g = e / C;
    const int c = (int)(e - g * C);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
        bf16x4 w = *reinterpret_cast<const bf16x4*>(X + ((size_t)g * k + j) * C + c);
        s += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
    }
    bf16x4 o = {f2bf(s[0]), f2bf(s[1]), f2bf(s[2]), f2bf(s[3])};
    *reinterpret_cast<bf16x4*>(out + e) = o;
}

// ---------------------------------------------------------------------------------------------
// backward B6: stage 1.  dy = relu'(F1) * (dF1 + [j == arg] dPooled);  partial[blk][7][C1] = sum dy, sum dy*x6
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_stage1_bwd_kernel(const float* __restrict__ points,
                                                               const float* __restrict__ feats,
                                                               const int32_t* __restrict__ idx,
                                                               const __bf16* __restrict__ F1,
                                                               const __bf16* __restrict__ dF1,
                                                               const __bf16* __restrict__ dPooled,
                                                               const uint8_t* __restrict__ arg,
                                                               float* __restrict__ partial, long long G, int N, int S,
                                                               int k, int C1) {
    // thread = (channel quad cq = tid & 63, position lane pl = tid >> 6): 8-B loads of F1 / dF1 and 4 independent position
    // streams per channel instead of one 2-B load per lane and iteration (the kernel ran at 1.9 TB/s, latency-bound)
    extern __shared__ float xs[];
    const int cq = threadIdx.x & 63, pl = threadIdx.x >> 6, c4 = cq * 4;
    const bool cok = c4 < C1;                                   // C1 % 4 == 0 (checked by the launcher)
    f32x4 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int SK = S * k;
    for (long long g0 = (long long)blockIdx.x * GPB; g0 < G; g0 += (long long)gridDim.x * GPB) {
        const int ng = (int)((G - g0) < GPB ? (G - g0) : GPB);
        __syncthreads();
        for (int i = threadIdx.x; i < ng * k; i += 256) {
            float x[X6];
            load_x6(points, feats, idx, g0 * k + i, N, SK, x);
#pragma unroll
            for (int d = 0; d < X6; ++d) xs[i * 8 + d] = x[d];
        }
        __syncthreads();
        if (cok) {
            for (int g = 0; g < ng; ++g) {
                const bf16x4 dp = *reinterpret_cast<const bf16x4*>(dPooled + (size_t)(g0 + g) * C1 + c4);
                const u32 am4 = *reinterpret_cast<const u32*>(arg + (size_t)(g0 + g) * C1 + c4);
                for (int j = pl; j < k; j += 4) {
                    const size_t o = ((size_t)(g0 + g) * k + j) * C1 + c4;
                    const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dF1 + o);
                    const bf16x4 f4 = *reinterpret_cast<const bf16x4*>(F1 + o);
                    const float* x = xs + (g * k + j) * 8;
                    f32x4 dy;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = bf2f(d4[r]) + (j == (int)((am4 >> (8 * r)) & 255u) ? bf2f(dp[r]) : 0.f);
                        dy[r] = bf2f(f4[r]) > 0.f ? v : 0.f;
                    }
                    acc[0] += dy;
#pragma unroll
                    for (int d = 0; d < X6; ++d) acc[1 + d] += dy * x[d];
                }
            }
        }
    }
    // the four position lanes of a channel quad -> one partial row set (fixed order)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(xs);                   // >= 256 * 7 * 16 B = 28 KiB (see the launcher)
#pragma unroll
    for (int q = 0; q < 7; ++q) red[(q * 4 + pl) * 64 + cq] = acc[q];
    __syncthreads();
    if (pl == 0 && cok) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const f32x4 s = (red[(q * 4 + 0) * 64 + cq] + red[(q * 4 + 1) * 64 + cq]) + (red[(q * 4 + 2) * 64 + cq] + red[(q * 4 + 3) * 64 + cq]);
            *reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * 7 + q) * C1 + c4) = s;
        }
    }
}

// B7: closed-form BN1 + conv1 backward from the reduced sums.  One block of 256 threads (C1 <= 256).
struct Stage1BwdParams {
    const float* sums;      // [7][C1]: Sdy, Sdyx[6]
    const float* mom;       // [42]: Sx[6] | M2[6][6]
    const float* aeff;      // [C1][8]
    const float* W1; const float* W2; const float* gamma; const float* save_invstd;
    float* dW1; float* dW2; float* db2; float* dgamma; float* dbeta;
    double Np; float beta_acc; int C1;
};

__global__ __launch_bounds__(256) void embed_stage1_bwd_finalize_kernel(Stage1BwdParams p) {
    __shared__ float g6[256][X6];
    __shared__ float w1s[256 * X6];
    const int c = threadIdx.x, C1 = p.C1;
    for (int i = c; i < C1 * X6; i += 256) w1s[i] = p.W1[i];
    if (c < C1) {
        double a[X6], sx[X6], sdyx[X6];
        const double sdy = p.sums[c], is = p.save_invstd[c], gam = p.gamma[c];
        double a_sx = 0.0, a_sdyx = 0.0;
#pragma unroll
        for (int d = 0; d < X6; ++d) {
            a[d] = p.aeff[c * 8 + d]; sx[d] = p.mom[d]; sdyx[d] = p.sums[(size_t)(1 + d) * C1 + c];
            a_sx += a[d] * sx[d]; a_sdyx += a[d] * sdyx[d];
        }
        const double bm = -a_sx / p.Np;                       // b2_c - mean_c
        const double sdyxhat = is * (a_sdyx + bm * sdy);      // sum dy * xhat
        const double mdy = sdy / p.Np, mdyx = sdyxhat / p.Np;
#pragma unroll
        for (int d = 0; d < X6; ++d) {
            double m2a = 0.0;
#pragma unroll
            for (int e = 0; e < X6; ++e) m2a += (double)p.mom[6 + d * 6 + e] * a[e];
            const double sxhx = is * (m2a + bm * sx[d]);      // sum xhat_c * x6[d]
            g6[c][d] = (float)(gam * is * (sdyx[d] - mdy * sx[d] - mdyx * sxhx));
        }
        const float ba = p.beta_acc;
        p.dgamma[c] = (ba != 0.f ? ba * p.dgamma[c] : 0.f) + (float)sdyxhat;
        p.dbeta[c] = (ba != 0.f ? ba * p.dbeta[c] : 0.f) + (float)sdy;
        // sum dh1 = gamma*invstd*(Sdy - Np*mdy - mdyx*sum xhat) and sum xhat == 0: the bias under a
        // train-mode BatchNorm receives exactly zero gradient
        p.db2[c] = (ba != 0.f ? ba * p.db2[c] : 0.f);
    }
    __syncthreads();
    if (c < C1) {
        // dW1[j=c][d] = sum_cc W2[cc][c] * g6[cc][d]
        float s[X6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < C1; ++cc) {
            const float w = p.W2[(size_t)cc * C1 + c];
#pragma unroll
            for (int d = 0; d < X6; ++d) s[d] += w * g6[cc][d];
        }
        const float ba = p.beta_acc;
#pragma unroll
        for (int d = 0; d < X6; ++d) p.dW1[c * X6 + d] = (ba != 0.f ? ba * p.dW1[c * X6 + d] : 0.f) + s[d];
        // dW2[cc][j] = g6[cc] . W1[j]; thread c writes column j = c for all cc (coalesced across threads)
        for (int cc = 0; cc < C1; ++cc) {
            float v = 0.f;
#pragma unroll
            for (int d = 0; d < X6; ++d) v += g6[cc][d] * w1s[c * X6 + d];
            const size_t o = (size_t)cc * C1 + c;
            p.dW2[o] = (ba != 0.f ? ba * p.dW2[o] : 0.f) + v;
        }
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
#define EMBED_BLOCKS 1024

extern "C" int p4p_embed_partial_rows(void) { return EMBED_BLOCKS; }

extern "C" int p4p_embed_moments(const float* points, const float* feats, const int32_t* idx, float* partial, int B,
                                 int N, int S, int k, void* stream) {
    P4P_CHECK(points && feats && idx && partial, "p4p_embed_moments: null pointer");
    long long P = (long long)B * S * k;
    P4P_CHECK(P >= 1, "p4p_embed_moments: empty input");
    hipLaunchKernelGGL(embed_moments_kernel, dim3(EMBED_BLOCKS), dim3(256), 0, (hipStream_t)stream, points, feats, idx,
                       partial, P, N, S * k);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_embed_fold(const float* W1, const float* W2, const float* b2, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, const float* mom_partial, float* aeff,
                              float* scale, float* shift, float* save_mean, float* save_invstd, float* mom,
                              long long Np, float eps, float momentum, int training, int C1, void* stream) {
    P4P_CHECK(W1 && W2 && b2 && gamma && beta && running_mean && running_var && aeff && scale && shift && save_mean &&
                  save_invstd && mom, "p4p_embed_fold: null pointer");
    P4P_CHECK(!training || mom_partial, "p4p_embed_fold: training needs moments");
    P4P_CHECK(C1 >= 1 && C1 <= 256, "p4p_embed_fold: C1=%d outside [1,256]", C1);
    FoldParams p = {W1, W2, b2, gamma, beta, running_mean, running_var, mom_partial, EMBED_BLOCKS, aeff, scale, shift,
                    save_mean, save_invstd, mom, (double)Np, eps, momentum, training, C1};
    hipLaunchKernelGGL(embed_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_embed_stage1(const float* points, const float* feats, const int32_t* idx, const float* aeff,
                                const float* scale, const float* shift, void* F1, void* pooled, uint8_t* arg, int B,
                                int N, int S, int k, int C1, void* stream) {
    P4P_CHECK(points && feats && idx && aeff && scale && shift && F1 && pooled && arg, "p4p_embed_stage1: null pointer");
    P4P_CHECK(C1 >= 1 && C1 <= 256 && k >= 1 && k <= 255, "p4p_embed_stage1: bad C1=%d k=%d", C1, k);
    long long G = (long long)B * S;
    int blocks = (int)((G + GPB - 1) / GPB);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_stage1_kernel, dim3(blocks), dim3(256), GPB * k * 8 * sizeof(float), (hipStream_t)stream,
                       points, feats, idx, aeff, scale, shift, (__bf16*)F1, (__bf16*)pooled, arg, G, N, S, k, C1);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                               long long Np, float eps, float momentum, int training, int C, void* stream) {
    P4P_CHECK(gamma && beta && running_mean && running_var && scale && shift && save_mean && save_invstd,
              "p4p_bn_finalize: null pointer");
    P4P_CHECK(!training || sums, "p4p_bn_finalize: training needs sums");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta,
                       running_mean, running_var, scale, shift, save_mean, save_invstd, (double)Np, eps, momentum,
                       training, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_bn_relu_bf16(const void* x, void* y, const float* scale, const float* shift, long long rows, int C,
                                void* stream) {
    P4P_CHECK(x && y && scale && shift && rows >= 1 && C % 4 == 0, "p4p_bn_relu_bf16: bad arguments");
    long long n = rows * C;
    if (C % 8 == 0) {
        const int tpr = bnc_tpr(C);
        hipLaunchKernelGGL(bn_colwise_kernel<false>, dim3(C / 8 / tpr, ceil_div(rows, BNC_ROWS)), dim3(256), 0,
                           (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, (const __bf16*)nullptr, scale, shift,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, rows, C, tpr);
    } else
        hipLaunchKernelGGL(bn_relu_bf16_kernel<4>, dim3(ceil_div(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)x, (__bf16*)y, scale, shift, n, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_bn_relu_maxpool(const void* H3, const float* scale, const float* shift, void* tok_bf16,
                                   float* tok_f32, uint8_t* arg, long long G, int k, int C, void* stream) {
    P4P_CHECK(H3 && scale && shift && (tok_bf16 || tok_f32) && G >= 1 && k >= 1 && k <= 255 && C % 4 == 0,
              "p4p_bn_relu_maxpool: bad arguments");
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(ceil_div(G * C / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)H3, scale, shift, (__bf16*)tok_bf16, tok_f32, arg, G, k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

// sums[2][C] <- column sums; dH3 [G*k, C] bf16.  partial: [p4p_embed_partial_rows()][2][C] floats.
extern "C" int p4p_pool_bn_bwd(const void* dtok, int dtok_is_f32, const void* H3, const uint8_t* arg,
                               const float* scale, const float* shift, const float* mean, const float* invstd,
                               float* partial, float* sums, void* dH3, long long G, int k, int C, void* stream) {
    P4P_CHECK(dtok && H3 && arg && scale && shift && mean && invstd && partial && sums && dH3,
              "p4p_pool_bn_bwd: null pointer");
    P4P_CHECK(G >= 1 && k >= 1 && C % 4 == 0, "p4p_pool_bn_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)(G < EMBED_BLOCKS ? G : EMBED_BLOCKS);
    int gpb = (int)((G + nb - 1) / nb);
    nb = (int)((G + gpb - 1) / gpb);
    hipLaunchKernelGGL(pool_bn_bwd_stats_kernel, dim3(ceil_div(C, 256), nb), dim3(256), 0, st, dtok, dtok_is_f32,
                       (const __bf16*)H3, arg, scale, shift, mean, invstd, partial, G, k, C, gpb);
    P4P_LAUNCH_CHECK();
    if (p4p_colreduce_f32(partial, sums, nb, (long long)2 * C, 2 * C, 0.f, stream)) return 2;
    float invNp = (float)(1.0 / ((double)G * k));
    hipLaunchKernelGGL(pool_bn_bwd_apply_kernel, dim3(ceil_div(G * C / 4, 256)), dim3(256), 0, st, dtok, dtok_is_f32,
                       (const __bf16*)H3, arg, scale, shift, mean, invstd, sums, invNp, (__bf16*)dH3, G, k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_group_sum_bf16(const void* X, void* out, long long G, int k, int C, void* stream);
// dY [rows,C] bf16 is overwritten with the gradient w.r.t. the BN input; sums[2][C] <- sum dy, sum dy*xhat.
// p4p_relu_bn_bwd_gsum additionally writes group_out [rows / k, C] bf16 = the sum of every k consecutive rows of the result (what
// p4p_group_sum_bf16 computes from the finished tensor, bit for bit) -- from the same pass where the shape allows, else by that kernel.
static int relu_bn_bwd_impl(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                            const float* invstd, float* partial, float* sums, void* group_out, int k, long long rows, int C, void* stream);
extern "C" int p4p_relu_bn_bwd(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                               const float* invstd, float* partial, float* sums, long long rows, int C, void* stream) {
    return relu_bn_bwd_impl(dY, H, scale, shift, mean, invstd, partial, sums, nullptr, 1, rows, C, stream);
}
extern "C" int p4p_relu_bn_bwd_gsum(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, float* partial, float* sums, void* group_out, int k, long long rows, int C,
                                    void* stream) {
    P4P_CHECK(group_out && k >= 1 && rows % k == 0, "p4p_relu_bn_bwd_gsum: group_out / k (rows=%lld, k=%d)", rows, k);
    return relu_bn_bwd_impl(dY, H, scale, shift, mean, invstd, partial, sums, group_out, k, rows, C, stream);
}
static int relu_bn_bwd_impl(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                            const float* invstd, float* partial, float* sums, void* group_out, int k, long long rows, int C, void* stream) {
    P4P_CHECK(dY && H && scale && shift && mean && invstd && partial && sums, "p4p_relu_bn_bwd: null pointer");
    P4P_CHECK(rows >= 1 && C % 4 == 0, "p4p_relu_bn_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)(rows < EMBED_BLOCKS ? rows : EMBED_BLOCKS);
    int rpb = (int)((rows + nb - 1) / nb);
    nb = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(relu_bn_bwd_stats_kernel, dim3(ceil_div(C, 256), nb), dim3(256), 0, st, (const __bf16*)dY,
                       (const __bf16*)H, scale, shift, mean, invstd, partial, rows, C, rpb);
    P4P_LAUNCH_CHECK();
    if (p4p_colreduce_f32(partial, sums, nb, (long long)2 * C, 2 * C, 0.f, stream)) return 2;
    long long n = rows * C;
    bool fused_gsum = false;
    if (C % 8 == 0) {
        const int tpr = bnc_tpr(C);
        const int run = BNC_ROWS / (256 / tpr);                      // rows a thread walks in the group-sum form
        if (group_out && run % k == 0) {
            fused_gsum = true;
            hipLaunchKernelGGL((bn_colwise_kernel<true, true>), dim3(C / 8 / tpr, ceil_div(rows, BNC_ROWS)), dim3(256), 0, st,
                               (const __bf16*)dY, (__bf16*)dY, (const __bf16*)H, scale, shift, mean, invstd, sums,
                               (float)(1.0 / (double)rows), rows, C, tpr, (__bf16*)group_out, k);
        } else
            hipLaunchKernelGGL(bn_colwise_kernel<true>, dim3(C / 8 / tpr, ceil_div(rows, BNC_ROWS)), dim3(256), 0, st,
                               (const __bf16*)dY, (__bf16*)dY, (const __bf16*)H, scale, shift, mean, invstd, sums,
                               (float)(1.0 / (double)rows), rows, C, tpr, (__bf16*)nullptr, 1);
    } else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ceil_div(n / 4, 256)), dim3(256), 0, st, (__bf16*)dY,
                           (const __bf16*)H, scale, shift, mean, invstd, sums, (float)(1.0 / (double)rows), n, C);
    P4P_LAUNCH_CHECK();
    if (group_out && !fused_gsum) return p4p_group_sum_bf16(dY, group_out, rows / k, k, C, stream);
    return 0;
}

extern "C" int p4p_group_sum_bf16(const void* X, void* out, long long G, int k, int C, void* stream) {
    P4P_CHECK(X && out && G >= 1 && k >= 1 && C % 4 == 0, "p4p_group_sum_bf16: bad arguments");
    hipLaunchKernelGGL(group_sum_bf16_kernel, dim3(ceil_div(G * C / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)X, (__bf16*)out, G, k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}

// partial: [EMBED_BLOCKS][7][C1]; sums: [7][C1]
extern "C" int p4p_embed_stage1_bwd(const float* points, const float* feats, const int32_t* idx, const void* F1,
                                    const void* dF1, const void* dPooled, const uint8_t* arg, float* partial,
                                    float* sums, const float* mom, const float* aeff, const float* W1, const float* W2,
                                    const float* gamma, const float* save_invstd, float* dW1, float* dW2, float* db2,
                                    float* dgamma, float* dbeta, float beta_acc, int B, int N, int S, int k, int C1,
                                    void* stream) {
    P4P_CHECK(points && feats && idx && F1 && dF1 && dPooled && arg && partial && sums && mom && aeff && W1 && W2 &&
                  gamma && save_invstd && dW1 && dW2 && db2 && dgamma && dbeta, "p4p_embed_stage1_bwd: null pointer");
    P4P_CHECK(C1 >= 1 && C1 <= 256, "p4p_embed_stage1_bwd: C1=%d outside [1,256]", C1);
    hipStream_t st = (hipStream_t)stream;
    long long G = (long long)B * S;
    int blocks = (int)((G + GPB - 1) / GPB);
    if (blocks > EMBED_BLOCKS) blocks = EMBED_BLOCKS;
    P4P_CHECK(C1 % 4 == 0, "p4p_embed_stage1_bwd: C1=%d must be a multiple of 4", C1);
    size_t shm = (size_t)GPB * k * 8 * sizeof(float);
    if (shm < 7 * 4 * 64 * 16) shm = 7 * 4 * 64 * 16;               // the cross-lane reduction reuses the coordinate buffer
    hipLaunchKernelGGL(embed_stage1_bwd_kernel, dim3(blocks), dim3(256), shm, st, points, feats,
                       idx, (const __bf16*)F1, (const __bf16*)dF1, (const __bf16*)dPooled, arg, partial, G, N, S, k, C1);
    P4P_LAUNCH_CHECK();
    if (p4p_colreduce_f32(partial, sums, blocks, (long long)7 * C1, 7 * C1, 0.f, stream)) return 2;
    Stage1BwdParams p = {sums, mom, aeff, W1, W2, gamma, save_invstd, dW1, dW2, db2, dgamma, dbeta,
                         (double)G * k, beta_acc, C1};
    hipLaunchKernelGGL(embed_stage1_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, p);
    P4P_LAUNCH_CHECK();
    return 0;
}
