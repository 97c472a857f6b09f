// gemm_tn v2: C[M,N] = A[M,K] . W[N,K]^T on v_mfma_f32_16x16x32_bf16, templated on the block tile.
//
// Schedule ("mid-tile barrier"): fragments are double-buffered in REGISTERS one k-step ahead, and the
// single barrier of a K-tile sits between its two k-steps:
//
//     A) issue ds_reads of (tile t, k-step 1)          | MFMAs on (tile t, k-step 0)   [already in regs]
//     B) s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier       -> tile t+1 has landed, nobody reads buffer t&1 any more
//     C) LDS-DMA tile t+2 -> buffer t&1; issue ds_reads of (tile t+1, k-step 0) | MFMAs on (tile t, k-step 1)
//
// so when the barrier releases every wave still has a full k-step of MFMAs whose operands are already in
// registers: the matrix pipe does not drain at tile boundaries, and every LDS-DMA has a whole K-tile of
// MFMA time to land.  LDS image and swizzle as in gemm.hip (128-B rows, chunk' = chunk ^ (row & 7)).
//
// Epilogue: bf16 outputs are staged through LDS (row stride BN*2+16 B) and written as full 16-B-per-lane
// row segments; fp32 outputs are stored directly (16 B per lane).
#pragma once
#include "common.h"

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
    int ldk;               // row stride of A and W in elements (= K, except for the split-K launches of gemm_tn2_kernel)
    int band;              // gemm_tn4p_kernel: banded tile order for wide outputs (band_remap); 0 = plain XCD-contiguous order
    int epi_flags;         // p4p_set_gemm_epilogue: bit 1 = no start-time shift of the persistent kernel; bits 8.. = that shift in 1,024-cycle units
};

enum {
    EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32_RESID = 2, EPI_BF16_DGELU = 3, EPI_BF16_ROWGROUP = 4, EPI_BF16_STATS = 5,
};

// Exact-erf GELU (timm's nn.GELU) without libm's branchy erff: Phi(x) = 0.5(1+erf(x/sqrt2)) from
// Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. fp32-level; outputs are rounded to bf16 anyway).
// e = exp(-x^2/2) is shared between Phi and the density term of the derivative.
__device__ __forceinline__ void phi_and_e(float x, float& Phi, float& e) {
    // explicit FMAs (the tree is built with -ffp-contract=off for the exact-distance kernels); 0.5 is folded into
    // the coefficients, the sign select is a copysign (v_bfi)
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
    e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);                 // exp(-x^2/2)
    float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
    const float half_tail = (t * poly) * e;               // 0.5 * erfc(|x|/sqrt2)
    Phi = 0.5f + __builtin_copysignf(0.5f - half_tail, x);
}
__device__ __forceinline__ float gelu_f(float x) { float P, e; phi_and_e(x, P, e); return x * P; }
__device__ __forceinline__ float dgelu_f(float x) { float P, e; phi_and_e(x, P, e); return __builtin_fmaf(x * e, 0.3989422804014327f, P); }
// both at once for the forward epilogue: the backward epilogue then only multiplies by the stored derivative
__device__ __forceinline__ void gelu_and_dgelu(float x, float& y, float& dy) {
    float P, e;
    phi_and_e(x, P, e);
    y = x * P;
    dy = __builtin_fmaf(x * e, 0.3989422804014327f, P);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous run of tiles; consecutive tiles sweep N for a fixed M panel.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Wide outputs (tiles_n a multiple of 4, >= 8: the N = 3072 GEMMs fc1 / dfc2, 12 tiles per row panel): in the order above the 32
// workgroups of an XCD cover 2.7 row panels x ALL column tiles at any time, so every time step streams the whole weight matrix
// (12 panels x 393 KB = 4.7 MB at K = 768: it does not stay in the 4-MB L2) -- FETCH_SIZE 623 MB per fc1 launch for 106 MB of
// operands.  Inside every band of 8 row panels the tiles are therefore taken column group by column group (8 row panels x 4
// column tiles = 32 workgroups = one time step): a step shares 8 A panels + 4 W panels, consecutive steps of a band keep the A
// panels.  `t` is the XCD-contiguous index of xcd_remap; bands must tile the XCD's run (checked by the caller's `banded`).
// Measured (tools/band_ab.py, bit-identical outputs): fc1 377 -> 368 us, 723 -> 709 us at M = 131k; dfc2 unchanged -- the fabric
// traffic it removes was not what these GEMMs wait for; the column group outermost over the XCD's whole run measured the same.
__device__ __forceinline__ int band_remap(int t, int tiles_n) {
    const int band = 8 * tiles_n, r = t % band;
    const int ng = r >> 5, r2 = r & 31;
    return t - r + (r2 >> 2) * tiles_n + ng * 4 + (r2 & 3);
}

template <int BM_, int BN_, int WAVES_M, int WAVES_N, int EPI, bool PIN, int ABL = 0>   // ABL: timing-only ablations (bit0: no LDS-DMA in the loop, bit1: no fragment reads); bit2: interleaved issue order
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N * 64 * ((2 * (BM_ + BN_) * 128) > 81920 ? 1 : 2)) / 256)
void gemm_tn2_kernel(GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N, NT_ = NW * 64;
    constexpr int WM_ = BM_ / WAVES_M, WN_ = BN_ / WAVES_N;       // per-wave tile
    constexpr int MT = WM_ / 16, NTL = WN_ / 16;                 // 16x16 MFMA tiles per wave
    constexpr int STAGE = (BM_ + BN_) * 128;                     // bytes per K-tile stage (BK = 64 bf16)
    constexpr int PIECES = (BM_ + BN_) / 8;                      // 1-KiB LDS-DMA pieces per stage
    constexpr int PPW = PIECES / NW;                             // pieces per wave
    static_assert(PIECES % NW == 0, "pieces must divide evenly over waves");
    constexpr int CSTRIDE = BN_ * 2 + 16;                        // epilogue staging row stride (bytes)
    constexpr int LDS_BYTES = (2 * STAGE) > (BM_ * CSTRIDE) ? (2 * STAGE) : (BM_ * CSTRIDE);
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN_ - 1) / BN_, tiles_m = (p.M + BM_ - 1) / BM_;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int nk = p.K / 64;
    // split-K launch (gridDim.y > 1): slice y contracts columns [y*K, (y+1)*K) of the ldk-wide operands into its own
    // fp32 output plane (EPI_F32_RESID without bias / residual); splitk_finish_kernel adds the planes in a fixed order
    const size_t koff = (size_t)blockIdx.y * p.K;

    // ---- LDS-DMA staging: piece q of the stage covers 8 rows; pieces [0, BM/8) are A rows, the rest W rows
    const int rin = lane >> 3;
    const int chunk = (lane & 7) ^ rin;
    const __bf16* src[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int piece = wave * PPW + q;
        const bool isA = piece < BM_ / 8;
        int r = isA ? m0 + piece * 8 + rin : n0 + (piece - BM_ / 8) * 8 + rin;
        const int lim = isA ? p.M : p.N;
        r = r < lim ? r : lim - 1;
        src[q] = (isA ? p.A : p.W) + (size_t)r * p.ldk + koff + chunk * 8;
    }
    auto stage = [&](int kt, char* buf) {
#pragma unroll
        for (int q = 0; q < PPW; ++q)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[q] + kt * 64), LDS_PTR(buf + (wave * PPW + q) * 1024), 16, 0, 0);
    };

    // EPI_BF16_ROWGROUP: the per-group addend is the INITIAL value of the accumulators -- its loads are issued here,
    // next to the first operand stages, instead of as a dependent global read in the epilogue (K is short on this path:
    // the tokenizer's 1x1 convs; the epilogue latency chain was most of the tile time)
    f32x4 acc[NTL][MT];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_BF16_ROWGROUP) {
                const int m = m0 + wm * WM_ + j * 16 + (lane & 15), n = n0 + wn * WN_ + i * 16 + (lane >> 4) * 4;
                if (m < p.M && n < p.N) acc[i][j] = *reinterpret_cast<const f32x4*>(p.rowgrp + (size_t)(m / p.group) * p.N + n);
            }
        }

    const int frag_off = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);
    const int a_base = wm * WM_ * 128 + frag_off;
    const int w_base = BM_ * 128 + wn * WN_ * 128 + frag_off;
    bf16x8 fa0[MT], fw0[NTL], fa1[MT], fw1[NTL];
    auto read_frags = [&](const char* buf, int ks, bf16x8* fa, bf16x8* fw) {
#pragma unroll
        for (int j = 0; j < MT; ++j) fa[j] = *reinterpret_cast<const bf16x8*>(buf + ((a_base + j * 2048) ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < NTL; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(buf + ((w_base + i * 2048) ^ (ks << 6)));
    };
    auto mma = [&](const bf16x8* fa, const bf16x8* fw) {
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
    };

    stage(0, lds);
    __syncthreads();
    if (nk > 1) stage(1, lds + STAGE);
    read_frags(lds, 0, fa0, fw0);
    if (ABL & 2) read_frags(lds, 1, fa1, fw1);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = lds + (kt & 1) * STAGE;
        char* nxt = lds + ((kt + 1) & 1) * STAGE;
        if (!(ABL & 2)) read_frags(cur, 1, fa1, fw1);
        if (PIN) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); }
        mma(fa0, fw0);
        if (PIN) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
        __syncthreads();                       // tile kt+1 landed everywhere; buffer `cur` is no longer read
        if (kt + 2 < nk && !(ABL & 1)) stage(kt + 2, cur);
        if (kt + 1 < nk && !(ABL & 2)) read_frags(nxt, 0, fa0, fw0);
        if (ABL & 2) {
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(fa0[j]), "+v"(fa1[j]));
#pragma unroll
            for (int i = 0; i < NTL; ++i) asm volatile("" : "+v"(fw0[i]), "+v"(fw1[i]));
        }
        if (PIN) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); }
        mma(fa1, fw1);
        if (PIN) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
        if (ABL & 4) {
            // issue order of this phase: one LDS-DMA, then MFMAs, then a couple of fragment reads, more MFMAs ...
            // so a wave stuck in a 60-180-cycle LDS-DMA issue always has MFMAs queued in the pipe behind it
            constexpr int NMFMA = MT * NTL, NDMA = PPW, NRD = MT + NTL;
            constexpr int G = NDMA > 0 ? NDMA : 1;
#pragma unroll
            for (int q = 0; q < G; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       // 1 VMEM read (LDS-DMA)
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / G / 2, 0);           // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, (NRD + G - 1) / G, 0);       // fragment reads
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA / G - NMFMA / G / 2, 0);
            }
        }
    }

    // ---------------------------------------------------------------------------------------------
    // epilogue.  lane holds, for (i,j): row m = m0 + wm*WM + j*16 + (lane&15), cols n0 + wn*WN + i*16 + 4g .. +3
    // ---------------------------------------------------------------------------------------------
    const int g = lane >> 4, li = lane & 15;
    constexpr bool BF16_OUT = EPI != EPI_F32_RESID;
    constexpr bool STATS = EPI == EPI_BF16_ROWGROUP || EPI == EPI_BF16_STATS;
    float csum[NTL][4], csq[NTL][4];
    if (STATS) {
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[i][r] = csq[i][r] = 0.f;
    }
    if (BF16_OUT) __syncthreads();            // all waves are done with the operand buffers
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int nl = wn * WN_ + i * 16 + g * 4;           // column inside the tile
        const int n = n0 + nl;
        const bool nok = n < p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && nok) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int ml = wm * WM_ + j * 16 + li;
            const int m = m0 + ml;
            const bool ok = nok && m < p.M;
            const size_t o = (size_t)m * p.N + n;
            f32x4 v = acc[i][j] + bv;
            if (EPI == EPI_F32_RESID) {
                if (ok) {
                    if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + o);
                    if (p.rowgrp) v += *reinterpret_cast<const f32x4*>(p.rowgrp + o);           // second residual (gemm_tn3.h)
                    const size_t oo = p.group > 1 ? ((size_t)m + m / p.group + 1) * p.N + n : o;     // token-stream rows (gemm_tn3.h)
                    *reinterpret_cast<f32x4*>((float*)p.out + (size_t)blockIdx.y * p.M * p.N + oo) = v;
                }
                continue;
            }
            if (EPI == EPI_BF16_GELU) {
                // `pre` receives gelu'(h) (only the backward pass reads it, as a plain multiplier)
                f32x4 y, dy;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float yy, dd;                      // (vector elements cannot bind to float&)
                    gelu_and_dgelu(v[r], yy, dd);
                    y[r] = yy;
                    dy[r] = dd;
                }
                if (p.pre && ok) {
                    bf16x4 w = {f2bf(dy[0]), f2bf(dy[1]), f2bf(dy[2]), f2bf(dy[3])};
                    *reinterpret_cast<bf16x4*>(p.pre + o) = w;
                }
                v = y;
            } else if (EPI == EPI_BF16_DGELU) {
                if (ok) {
                    bf16x4 h = *reinterpret_cast<const bf16x4*>(p.pre + o);
                    v = f32x4{v[0] * bf2f(h[0]), v[1] * bf2f(h[1]), v[2] * bf2f(h[2]), v[3] * bf2f(h[3])};
                }
            }
            if (STATS && ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { csum[i][r] += v[r]; csq[i][r] += v[r] * v[r]; }
            }
            bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *reinterpret_cast<bf16x4*>(lds + ml * CSTRIDE + nl * 2) = w;
        }
    }
    // column statistics: 16 rows held across the lanes of one DPP row -> reduce, park per wave-row in LDS BEHIND the
    // staging tile; the same barrier publishes both, and the final sums leave together with the output stores (a
    // barrier after the stores would wait for their acknowledgement: ~2 us per tile on this short-K path)
    float* red = reinterpret_cast<float*>(lds + BM_ * CSTRIDE);       // [WAVES_M][2][BN]
    static_assert(!STATS || BM_ * CSTRIDE + WAVES_M * 2 * BN_ * 4 <= 2 * STAGE, "statistics must fit behind the staging tile");
    if (STATS) {
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = csum[i][r], b = csq[i][r];
                a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
                b += dpp_f32<0xB1>(b); b += dpp_f32<0x4E>(b); b += dpp_f32<0x141>(b); b += dpp_f32<0x140>(b);
                if (li == 0) {
                    const int c = wn * WN_ + i * 16 + g * 4 + r;
                    red[(wm * 2 + 0) * BN_ + c] = a;
                    red[(wm * 2 + 1) * BN_ + c] = b;
                }
            }
    }
    if (BF16_OUT) {
        __syncthreads();
        constexpr int CPR = BN_ / 8;                        // 16-B chunks per tile row
        for (int c = t; c < BM_ * CPR; c += NT_) {
            const int ml = c / CPR, cc = c - ml * CPR;
            const int m = m0 + ml, n = n0 + cc * 8;
            if (m < p.M && n < p.N) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + ml * CSTRIDE + cc * 16);
                *reinterpret_cast<u32x4*>((__bf16*)p.out + (size_t)m * p.N + n) = v;
            }
        }
    }
    if (STATS) {
        for (int c = t; c < BN_; c += NT_) {
            if (n0 + c < p.N) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES_M; ++w) { a += red[(w * 2 + 0) * BN_ + c]; b += red[(w * 2 + 1) * BN_ + c]; }
                p.stats[((size_t)tm * 2 + 0) * p.N + n0 + c] = a;
                p.stats[((size_t)tm * 2 + 1) * p.N + n0 + c] = b;
            }
        }
    }
}
