// bf16 MFMA GEMMs for gfx950 with fused epilogues: every Linear / 1x1-conv of the hot path.
//
//   gemm_tn   : C[M,N]  = A[M,K] . W[N,K]^T          (forward Linear, and dgrad with W^T stored)
//               reference: nn.Linear inside timm Block (qkv/proj/fc1/fc2, src/models/pix4point.py:254-255),
//               PointViT.proj / pos_embed (:213-218,245-246), P3Embed 1x1 convs (:137-156,183,188).
//   gemm_wgrad: dW[N,K] = G[M,N]^T . A[M,K]          (weight gradient; contraction over the M rows)
//
// Both use one workgroup = 4 waves = a 128x128 output tile, v_mfma_f32_16x16x32_bf16, fp32
// accumulation, LDS double buffering filled by global_load_lds_dwordx4 (LDS-DMA, 16 B per lane,
// 1 KiB per wave-instruction).  LDS-DMA writes lane-linear, so bank-conflict swizzles are applied to
// the per-lane SOURCE address and to the fragment reads (never to the destination).
//
// gemm_tn LDS image: [128 rows][64 bf16] = 128-B rows of 8 16-B chunks, chunk' = chunk ^ (row & 7);
// fragment reads are ds_read_b128 (conflict-free: checked in tests/test_lds_model.py).
// gemm_wgrad LDS image: [64 m-rows][128 cols] = 256-B rows of 8 32-B slots, slot' = slot ^ h(row),
// h(row) = (row & 3) | ((row >> 3) & 1) << 2; fragments come from ds_read_b64_tr_b16 (the
// hardware transposing read), conflict-free for the 8 rows a half-wave touches.
//
// MFMA operand maps (16x16x32 bf16): A lane l -> A[l&15][8(l>>4)+j], B lane l -> B[8(l>>4)+j][l&15],
// D lane l reg r -> D[4(l>>4)+r][l&15].  We pass the W (resp. A-matrix) fragment as the MFMA "A"
// operand so each lane ends up with 4 CONSECUTIVE output columns of one output row: 8-B bf16 /
// 16-B fp32 stores.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256

__device__ float g_zero_page[64];   // 256 B of zeros: LDS-DMA source for rows past the end (wgrad)

enum {
    EPI_BF16 = 0,          // out_bf16 = acc (+bias)
    EPI_BF16_GELU = 1,     // pre_bf16 = acc+bias (if pre != null); out_bf16 = gelu(acc+bias)
    EPI_F32_RESID = 2,     // out_f32 = acc (+bias) (+resid_f32)
    EPI_BF16_DGELU = 3,    // out_bf16 = acc * gelu'(pre_bf16)
    EPI_BF16_ROWGROUP = 4, // out_bf16 = acc + rowgrp_f32[m / group][n]; column sum / sumsq partials
    EPI_BF16_STATS = 5,    // out_bf16 = acc (+bias); column sum / sumsq partials (BatchNorm statistics)
};

struct GemmParams {
    const __bf16* A;       // [M,K]
    const __bf16* W;       // [N,K]
    void* out;             // [M,N] bf16 or f32
    const float* bias;     // [N] or null
    const float* resid;    // [M,N] f32 or null       (EPI_F32_RESID)
    __bf16* pre;           // [M,N] bf16 or null      (EPI_BF16_GELU writes, EPI_BF16_DGELU reads)
    const float* rowgrp;   // [M/group, N] f32        (EPI_BF16_ROWGROUP)
    float* stats;          // [tiles_m][2][N] f32 partial column sums / sums of squares
    int M, N, K, group;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous run of tiles; consecutive tiles sweep N for a fixed M panel (panel of A stays in that
// XCD's L2, W is shared by everybody).  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Stage one 128x64 bf16 tile (rows row0.., k-columns k0..k0+63) of a [rows,K] matrix into LDS.
// wave w issues 4 LDS-DMA instructions of 8 rows each.
__device__ __forceinline__ void stage_tile_tn(const __bf16* __restrict__ src, int nrows, int K, int row0, int k0,
                                              char* lds_tile, int wave, int lane) {
    const int rin = lane >> 3;                       // row inside the 8-row piece
    const int chunk = (lane & 7) ^ rin;              // logical 16-B chunk this lane must fetch
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        int r = row0 + piece * 8 + rin;
        r = r < nrows ? r : nrows - 1;
        const __bf16* g = src + (size_t)r * K + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_tn_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * BM * BK * 2];   // [buf][A|W][16 KiB]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / BK;

    f32x4 acc[4][4];     // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment read offset inside a tile for k-step 0 (k-step 1 = offset ^ 64)
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);

    stage_tile_tn(p.A, p.M, p.K, m0, 0, lds, wave, lane);
    stage_tile_tn(p.W, p.N, p.K, n0, 0, lds + BM * BK * 2, wave, lane);

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                 // tile kt has landed (barrier drains LDS-DMA), buffer kt^1 is free
        char* cur = lds + (kt & 1) * (2 * BM * BK * 2);
        if (kt + 1 < nk) {
            char* nxt = lds + ((kt + 1) & 1) * (2 * BM * BK * 2);
            stage_tile_tn(p.A, p.M, p.K, m0, (kt + 1) * BK, nxt, wave, lane);
            stage_tile_tn(p.W, p.N, p.K, n0, (kt + 1) * BK, nxt + BM * BK * 2, wave, lane);
        }
        const char* la = cur + (wm * 64) * 128;
        const char* lw = cur + BM * BK * 2 + (wn * 64) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const bf16x8*>(la + i * 16 * 128 + (frag_off ^ (ks << 6)));
                fw[i] = *reinterpret_cast<const bf16x8*>(lw + i * 16 * 128 + (frag_off ^ (ks << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds, for (i,j): row m = m0+wm*64+j*16+(lane&15), cols n..n+3 ----------
    const int g = lane >> 4;
    float csum[4][4], csq[4][4];
    if (EPI == EPI_BF16_ROWGROUP || EPI == EPI_BF16_STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[i][r] = csq[i][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + g * 4;
        const bool nok = n < p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + (lane & 15);
            if (!(nok && m < p.M)) continue;
            const size_t o = (size_t)m * p.N + n;
            f32x4 v = acc[i][j] + bv;
            if (EPI == EPI_BF16) {
                bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *reinterpret_cast<bf16x4*>((__bf16*)p.out + o) = w;
            } else if (EPI == EPI_BF16_GELU) {
                if (p.pre) {
                    bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *reinterpret_cast<bf16x4*>(p.pre + o) = w;
                }
                bf16x4 w = {f2bf(gelu_f(v[0])), f2bf(gelu_f(v[1])), f2bf(gelu_f(v[2])), f2bf(gelu_f(v[3]))};
                *reinterpret_cast<bf16x4*>((__bf16*)p.out + o) = w;
            } else if (EPI == EPI_F32_RESID) {
                if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + o);
                *reinterpret_cast<f32x4*>((float*)p.out + o) = v;
            } else if (EPI == EPI_BF16_DGELU) {
                bf16x4 h = *reinterpret_cast<const bf16x4*>(p.pre + o);
                bf16x4 w = {f2bf(v[0] * dgelu_f(bf2f(h[0]))), f2bf(v[1] * dgelu_f(bf2f(h[1]))),
                            f2bf(v[2] * dgelu_f(bf2f(h[2]))), f2bf(v[3] * dgelu_f(bf2f(h[3])))};
                *reinterpret_cast<bf16x4*>((__bf16*)p.out + o) = w;
            } else {   // EPI_BF16_ROWGROUP / EPI_BF16_STATS
                if (EPI == EPI_BF16_ROWGROUP)
                    v += *reinterpret_cast<const f32x4*>(p.rowgrp + (size_t)(m / p.group) * p.N + n);
                bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *reinterpret_cast<bf16x4*>((__bf16*)p.out + o) = w;
#pragma unroll
                for (int r = 0; r < 4; ++r) { csum[i][r] += v[r]; csq[i][r] += v[r] * v[r]; }
            }
        }
    }
    if (EPI == EPI_BF16_ROWGROUP || EPI == EPI_BF16_STATS) {
        // reduce the 16 rows held across lanes (same g, lane&15 = 0..15) with DPP, then the two
        // m-halves (wm) through LDS, and write one partial row per (tile_m) -- no atomics.
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);           // [wm][2][128]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = csum[i][r], b = csq[i][r];
                a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
                b += dpp_f32<0xB1>(b); b += dpp_f32<0x4E>(b); b += dpp_f32<0x141>(b); b += dpp_f32<0x140>(b);
                if ((lane & 15) == 0) {
                    int c = wn * 64 + i * 16 + g * 4 + r;
                    red[(wm * 2 + 0) * 128 + c] = a;
                    red[(wm * 2 + 1) * 128 + c] = b;
                }
            }
        __syncthreads();
        if (t < 128 && n0 + t < p.N) {
            p.stats[((size_t)tm * 2 + 0) * p.N + n0 + t] = red[0 * 128 + t] + red[2 * 128 + t];
            p.stats[((size_t)tm * 2 + 1) * p.N + n0 + t] = red[1 * 128 + t] + red[3 * 128 + t];
        }
    }
}

template <int EPI>
static int launch_tn(const GemmParams& p, hipStream_t st) {
    int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    hipLaunchKernelGGL(gemm_tn_kernel<EPI>, dim3(tiles), dim3(GEMM_THREADS), 0, st, p);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_gemm_tn_bf16(int epi, const void* A, const void* W, void* out, const float* bias,
                                const float* resid, void* pre, const float* rowgrp, float* stats, int M, int N,
                                int K, int group, void* stream) {
    P4P_CHECK(A && W && out, "p4p_gemm_tn_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 4 && K >= BK, "p4p_gemm_tn_bf16: bad sizes M=%d N=%d K=%d", M, N, K);
    P4P_CHECK(K % BK == 0, "p4p_gemm_tn_bf16: K=%d must be a multiple of %d", K, BK);
    P4P_CHECK(N % 4 == 0, "p4p_gemm_tn_bf16: N=%d must be a multiple of 4", N);
    GemmParams p;
    p.A = (const __bf16*)A; p.W = (const __bf16*)W; p.out = out; p.bias = bias; p.resid = resid;
    p.pre = (__bf16*)pre; p.rowgrp = rowgrp; p.stats = stats; p.M = M; p.N = N; p.K = K; p.group = group;
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case EPI_BF16: return launch_tn<EPI_BF16>(p, st);
        case EPI_BF16_GELU: return launch_tn<EPI_BF16_GELU>(p, st);
        case EPI_F32_RESID: return launch_tn<EPI_F32_RESID>(p, st);
        case EPI_BF16_DGELU:
            P4P_CHECK(pre, "p4p_gemm_tn_bf16: EPI_BF16_DGELU needs pre");
            return launch_tn<EPI_BF16_DGELU>(p, st);
        case EPI_BF16_ROWGROUP:
            P4P_CHECK(rowgrp && stats && group >= 1, "p4p_gemm_tn_bf16: EPI_BF16_ROWGROUP needs rowgrp, stats, group");
            return launch_tn<EPI_BF16_ROWGROUP>(p, st);
        case EPI_BF16_STATS:
            P4P_CHECK(stats, "p4p_gemm_tn_bf16: EPI_BF16_STATS needs stats");
            return launch_tn<EPI_BF16_STATS>(p, st);
    }
    P4P_CHECK(false, "p4p_gemm_tn_bf16: unknown epilogue %d", epi);
}

extern "C" int p4p_gemm_tn_stats_rows(int M) { return ceil_div(M, BM); }

// ================================================================================================
// Weight gradient: dW[N,K] = sum_m G[m,N]^T A[m,K].  grid = (tiles_n * tiles_k, splits); each split
// reduces a contiguous run of 64-row chunks and writes its own fp32 slab; p4p_reduce_slabs sums
// them in a fixed order (bitwise reproducible, no atomics).
// ================================================================================================
#define WG_BMR 64      // m rows per LDS chunk

// Stage 64 rows x 128 columns (256-B rows) of a [M, ld] bf16 matrix; rows >= M come from the zero page.
__device__ __forceinline__ void stage_tile_wg(const __bf16* __restrict__ src, int M, int ld, int mrow0, int col0,
                                              int ncols, char* lds_tile, int wave, int lane) {
    const int rin = lane >> 4;                      // 4 rows per 1-KiB piece
    const int pc = lane & 15;                       // physical 16-B chunk in the 256-B row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        const int rl = piece * 4 + rin;             // row inside the chunk, 0..63
        const int h = (rl & 3) | (((rl >> 3) & 1) << 2);
        const int chunk = pc ^ (h << 1);            // logical chunk
        const int r = mrow0 + rl;
        int c = col0 + chunk * 8;
        const bool ok = r < M && c < ncols;
        const __bf16* g = ok ? src + (size_t)r * ld + c : reinterpret_cast<const __bf16*>(g_zero_page);
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 tr_read8(const char* p_lo, const char* p_hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

struct WgradParams {
    const __bf16* G;   // [M,N]
    const __bf16* A;   // [M,K]
    float* out;        // [splits][N][K] slabs (or the final [N,K] when splits == 1)
    int M, N, K, splits, chunks_per_split;
};

__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_wgrad_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * WG_BMR * 128 * 2];   // [buf][G|A][16 KiB]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = (p.K + 127) / 128;
    const int tile = blockIdx.x;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int split = blockIdx.y;
    const int chunk_begin = split * p.chunks_per_split;
    const int total_chunks = (p.M + WG_BMR - 1) / WG_BMR;
    int chunk_end = chunk_begin + p.chunks_per_split;
    chunk_end = chunk_end < total_chunks ? chunk_end : total_chunks;

    f32x4 acc[4][4];     // [k-tile][n-tile] -> lane: n = l&15, k = 4g + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing read: lane (group g = l>>4, i = l&15) supplies row 8g + (i>>2) (+4 for the second
    // read), columns 4(i&3).. of the 16-column block; it receives column i of those 4 rows.
    const int g = lane >> 4, li = lane & 15;
    const int trow = 8 * g + (li >> 2);
    const int tcolb = (li & 3) * 8;       // byte offset inside the 32-B slot

    if (chunk_begin < chunk_end) {
        stage_tile_wg(p.G, p.M, p.N, chunk_begin * WG_BMR, n0, p.N, lds, wave, lane);
        stage_tile_wg(p.A, p.M, p.K, chunk_begin * WG_BMR, k0, p.K, lds + 16384, wave, lane);
    }
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
        __syncthreads();
        const int bi = (ch - chunk_begin) & 1;
        char* cur = lds + bi * 32768;
        if (ch + 1 < chunk_end) {
            char* nxt = lds + (bi ^ 1) * 32768;
            stage_tile_wg(p.G, p.M, p.N, (ch + 1) * WG_BMR, n0, p.N, nxt, wave, lane);
            stage_tile_wg(p.A, p.M, p.K, (ch + 1) * WG_BMR, k0, p.K, nxt + 16384, wave, lane);
        }
        const char* lg = cur;
        const char* la = cur + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {          // 32 m-rows per MFMA k-step
            const int r_lo = ks * 32 + trow, r_hi = r_lo + 4;
            const int h_lo = (r_lo & 3) | (((r_lo >> 3) & 1) << 2);
            const int h_hi = (r_hi & 3) | (((r_hi >> 3) & 1) << 2);
            bf16x8 fg[4], fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int slot_g = wn * 4 + i, slot_a = wk * 4 + i;      // 16-column block index in the tile
                fg[i] = tr_read8(lg + r_lo * 256 + ((slot_g ^ h_lo) << 5) + tcolb,
                                 lg + r_hi * 256 + ((slot_g ^ h_hi) << 5) + tcolb);
                fa[i] = tr_read8(la + r_lo * 256 + ((slot_a ^ h_lo) << 5) + tcolb,
                                 la + r_hi * 256 + ((slot_a ^ h_hi) << 5) + tcolb);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + i * 16 + g * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + li;
            if (n < p.N && k < p.K) *reinterpret_cast<f32x4*>(out + (size_t)n * p.K + k) = acc[i][j];
        }
    }
}

// out[e] = beta * out[e] + sum_s slabs[s][e]
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                    int splits, float beta) {
    long long e = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e >= n) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + e);
    for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)k * n + e);
    if (beta != 0.f) s += beta * *reinterpret_cast<const f32x4*>(out + e);
    *reinterpret_cast<f32x4*>(out + e) = s;
}

extern "C" int p4p_wgrad_splits(int M, int N, int K) {
    int tiles = ceil_div(N, 128) * ceil_div(K, 128);
    int chunks = ceil_div(M, WG_BMR);
    int want = ceil_div(1024, tiles);             // ~4 workgroups per CU over the whole grid
    if (want < 1) want = 1;
    if (want > chunks) want = chunks;
    if (want > 64) want = 64;
    return want;
}

// workspace: splits*N*K floats (unused when splits == 1 and beta == 0)
extern "C" int p4p_gemm_wgrad_bf16(const void* G, const void* A, float* dW, float* workspace, int M, int N, int K,
                                   float beta, void* stream) {
    P4P_CHECK(G && A && dW, "p4p_gemm_wgrad_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 8 && K >= 8, "p4p_gemm_wgrad_bf16: bad sizes M=%d N=%d K=%d", M, N, K);
    P4P_CHECK(N % 8 == 0 && K % 8 == 0, "p4p_gemm_wgrad_bf16: N=%d and K=%d must be multiples of 8", N, K);
    hipStream_t st = (hipStream_t)stream;
    int splits = p4p_wgrad_splits(M, N, K);
    int chunks = ceil_div(M, WG_BMR);
    WgradParams p;
    p.G = (const __bf16*)G; p.A = (const __bf16*)A; p.M = M; p.N = N; p.K = K;
    p.chunks_per_split = ceil_div(chunks, splits);
    splits = ceil_div(chunks, p.chunks_per_split);
    p.splits = splits;
    bool direct = splits == 1 && beta == 0.f;
    P4P_CHECK(direct || workspace, "p4p_gemm_wgrad_bf16: workspace required (splits=%d)", splits);
    p.out = direct ? dW : workspace;
    dim3 grid(ceil_div(N, 128) * ceil_div(K, 128), splits);
    hipLaunchKernelGGL(gemm_wgrad_kernel, grid, dim3(GEMM_THREADS), 0, st, p);
    P4P_LAUNCH_CHECK();
    if (!direct) {
        long long n = (long long)N * K;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, st, workspace, dW, n,
                           splits, beta);
        P4P_LAUNCH_CHECK();
    }
    return 0;
}
