// wgrad v3: dW[N,K] = sum_m G[m,N]^T A[m,K] with the gemm_tn3 recipe: output tile 256 (n) x 128 (k),
// 4 waves each owning 128 x 64, 32 m-rows per step, 3-deep LDS-DMA ring, counted vmcnt + raw s_barrier,
// two workgroups per CU, M split into slabs that a second kernel reduces in a fixed order.  (No fused bias
// gradient here: its 32 extra accumulator registers would spill; the caller runs the streaming column sum.)
//
// Both operands are M-major, so every MFMA fragment is a TRANSPOSING read (ds_read_b64_tr_b16, two per
// fragment).  LDS slot per step: G rows [32][256 cols] (512-B rows) then A rows [32][128 cols] (256-B rows);
// inside every 256-B window the 32-B column slot is XORed with h(row) = (row&3) | ((row>>3)&1)<<2, which makes
// the 8 rows a half-wave touches land on 8 different slots (tools/lds_model.py).
#pragma once
#include "gemm_tn3.h"

#define W3_BN 256      // n columns of G per tile
#define W3_BK 128      // k columns of A per tile (narrow variant, 4 waves, 2 workgroups per CU); wide: 256, 8 waves

struct W3Params {
    const __bf16* G;   // [M,N]
    const __bf16* A;   // [M,K]
    float* out;        // [splits][N][K]
    int M, N, K, steps_per_split;
};

struct W3Frags {
    bf16x8 a[4];
    bf16x8 g[8];
};

// Two transposing reads (LDS byte addresses lo / hi) -> one 8-element MFMA fragment.  INLINE ASM on purpose:
// with the builtin, hipcc cannot prove the read does not alias the LDS-DMA still in flight and inserts
// s_waitcnt vmcnt(0) in front of every read, draining the ring each iteration.  Consequence (guide 5.7): hipcc
// does not count these reads, so every consumer sits behind an explicit lgkmcnt(0) + sched_barrier.
__device__ __forceinline__ bf16x8 w3_tr(int lo, int hi) {
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int BK_, bool ILV = false>
__global__ __launch_bounds__(BK_ * 2, 2) void gemm_wgrad3_kernel(W3Params p) {
    constexpr int NWK = BK_ / 64, NW = 2 * NWK;                       // waves: 2 (n) x NWK (k), each 128 x 64
    constexpr int W3_STAGE = 32 * (W3_BN + BK_) * 2;                   // 24 / 32 KiB per 32-row step
    constexpr int APIECES = BK_ / 32;                                  // 1-KiB pieces of the A tile: 4 / 8 ... per 32 rows
    constexpr int ROWS_PER_APIECE = 1024 / (BK_ * 2);                  // 4 / 2
    constexpr int W3_PPW = (16 + 32 / ROWS_PER_APIECE) / NW;           // pieces per wave: 6 / 4
    constexpr int GPW = 16 / NW;                                       // G pieces per wave: 4 / 2
    __shared__ __attribute__((aligned(1024))) char lds[3 * W3_STAGE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wn = wave / NWK, wk = wave % NWK;
    const int tiles_k = p.K / BK_;
    // 1-D grid, split-major work ids, and each XCD owns a contiguous run of ids (xcd_remap): the workgroups that
    // share an L2 are (almost) all tiles of the SAME M-range, so every G / A row panel crosses the fabric about once
    // per XCD instead of once per workgroup.  (Measured with the (tile, split) grid, whose tiles of one split are
    // dealt round-robin to the 8 XCDs: FETCH_SIZE 1.47 GB per launch for 0.4 GB of operands, 6.8 TB/s -- the kernel
    // ran at the fabric limit, not the MFMA limit.)
    const int ntiles = (p.N / W3_BN) * tiles_k;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wid / ntiles, tile = wid - split * ntiles;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * W3_BN, k0 = tk * BK_;
    (void)APIECES;
    const int total_steps = (p.M + 31) / 32;
    const int step0 = split * p.steps_per_split;
    int nk = total_steps - step0;
    nk = nk < p.steps_per_split ? nk : p.steps_per_split;          // >= 1 by construction of the grid

    // ---- staging.  LDS pieces [0,16): G, 2 rows of 512 B each; then A pieces of 1 KiB (4 rows of 256 B, or 2 rows
    // of 512 B in the wide variant).  Every wave moves GPW G pieces and W3_PPW-GPW A pieces: compile-time per q.
    // Only the last 32-row step of the matrix can be ragged, and the branch-free steady-state loop never
    // stages it, so only the prologue / tail path checks rows against M (zero page beyond the end).
    const __bf16* zero = reinterpret_cast<const __bf16*>(g_zero_page);
    u32 soff[W3_PPW];                    // byte offset of this lane's 16 bytes inside the split's first step
    const char* gbase = reinterpret_cast<const char*>(p.G) + (size_t)step0 * 32 * p.N * 2;   // wave-uniform (SGPR)
    const char* abase = reinterpret_cast<const char*>(p.A) + (size_t)step0 * 32 * p.K * 2;
#pragma unroll
    for (int q = 0; q < W3_PPW; ++q) {
        if (q < GPW) {
            const int r = (wave * GPW + q) * 2 + (lane >> 5);
            const int h = (r & 3) | (((r >> 3) & 1) << 2);
            soff[q] = ((u32)r * p.N + n0 + (((lane & 31) ^ (h << 1)) * 8)) * 2u;
        } else {
            constexpr int LPR = 64 / ROWS_PER_APIECE;                  // lanes per A row: 16 / 32
            const int r = (wave * (W3_PPW - GPW) + q - GPW) * ROWS_PER_APIECE + lane / LPR;
            const int h = (r & 3) | (((r >> 3) & 1) << 2);
            soff[q] = ((u32)r * p.K + k0 + (((lane % LPR) ^ (h << 1)) * 8)) * 2u;
        }
    }
    auto lds_piece = [&](int q) {
        return q < GPW ? (wave * GPW + q) * 1024 : (16 + wave * (W3_PPW - GPW) + q - GPW) * 1024;
    };
    auto stage_fast = [&](int kt, char* buf) {
#pragma unroll
        for (int q = 0; q < W3_PPW; ++q)
            __builtin_amdgcn_global_load_lds(GLB_PTR((q < GPW ? gbase : abase) + (size_t)kt * 64 * (q < GPW ? p.N : p.K) + soff[q]),
                                             LDS_PTR(buf + lds_piece(q)), 16, 0, 0);
    };
    auto stage = [&](int kt, char* buf) {
        const int mbase = (step0 + kt) * 32;
#pragma unroll
        for (int q = 0; q < W3_PPW; ++q) {
            const int r = q < GPW ? (wave * GPW + q) * 2 + (lane >> 5)
                                  : (wave * (W3_PPW - GPW) + q - GPW) * ROWS_PER_APIECE + lane / (64 / ROWS_PER_APIECE);
            const __bf16* g = reinterpret_cast<const __bf16*>((q < GPW ? gbase : abase) + (size_t)kt * 64 * (q < GPW ? p.N : p.K) + soff[q]);
            g = mbase + r < p.M ? g : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(buf + lds_piece(q)), 16, 0, 0);
        }
    };

    f32x4 acc[4][8];     // [k-tile][n-tile]: lane holds n = li, k = 4g + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing read: lane (g, li) supplies row 8g + (li>>2) (+4 for the second read), 8 bytes at column
    // 4(li&3) of a 16-column block; it receives column li of those rows.
    const int g = lane >> 4, li = lane & 15;
    const int r_lo = 8 * g + (li >> 2), r_hi = r_lo + 4;
    const int h_lo = (r_lo & 3) | (((r_lo >> 3) & 1) << 2), h_hi = (r_hi & 3) | (((r_hi >> 3) & 1) << 2);
    const int cb8 = (li & 3) * 8;
    const int lds_base = (int)(size_t)LDS_PTR(lds);          // LDS byte address of the ring (1 KiB aligned)
    // address of block j = base ^ (j << 5): bits 5..7 of the base hold only h (row*512, the 256-B window and
    // cb8 leave them clear), so the XOR swizzle costs one v_xor per read instead of one live VGPR per read
    const int g_lo = r_lo * 512 + wn * 256 + cb8 + (h_lo << 5);
    const int g_hi = r_hi * 512 + wn * 256 + cb8 + (h_hi << 5);
    const int a_lo = 16384 + r_lo * (BK_ * 2) + (wk >> 1) * 256 + cb8 + ((h_lo ^ ((wk & 1) * 4)) << 5);
    const int a_hi = 16384 + r_hi * (BK_ * 2) + (wk >> 1) * 256 + cb8 + ((h_hi ^ ((wk & 1) * 4)) << 5);
    auto read_frags = [&](int buf_off, W3Frags& f) {
        int bgl = lds_base + buf_off + g_lo, bgh = lds_base + buf_off + g_hi;
        int bal = lds_base + buf_off + a_lo, bah = lds_base + buf_off + a_hi;
        asm volatile("" : "+v"(bgl), "+v"(bgh), "+v"(bal), "+v"(bah));      // not loop-invariant for hipcc
#pragma unroll
        for (int j = 0; j < 8; ++j) f.g[j] = w3_tr(bgl ^ (j << 5), bgh ^ (j << 5));
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[i] = w3_tr(bal ^ (i << 5), bah ^ (i << 5));
    };
    auto mma = [&](const W3Frags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i], f.g[j], acc[i][j], 0, 0, 0);
    };

    char* s0 = lds;
    char* s1 = lds + W3_STAGE;
    char* s2 = lds + 2 * W3_STAGE;
    W3Frags f0, f1;
    stage(0, s0);
    if (nk > 1) stage(1, s1);
    if (nk > 2) stage(2, s2);
    if (nk > 2) WAIT_VM(2 * W3_PPW);
    else if (nk > 1) WAIT_VM(W3_PPW);
    else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    read_frags(0, f0);
    if (nk > 2) WAIT_VM(W3_PPW);
    else WAIT_VM(0);
    WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    char *cur = s0, *nxt = s1, *nn = s2;
#define W3_FULL(KT, FC, FN)                                                               \
    {                                                                                     \
        if (!ILV) {                                                                       \
            stage_fast((KT) + 3, cur);                                                    \
            read_frags((int)(nxt - lds), FN);                                             \
            __builtin_amdgcn_sched_barrier(0);                                            \
            mma(FC);                                                                      \
        } else {                                                                          \
            /* hand-pinned issue order, 4 groups of {LDS-DMA, 4 MFMA, 6 transposing reads, 4 MFMA} */ \
            const int bo = (int)(nxt - lds);                                              \
            int bgl = lds_base + bo + g_lo, bgh = lds_base + bo + g_hi;                    \
            int bal = lds_base + bo + a_lo, bah = lds_base + bo + a_hi;                    \
            asm volatile("" : "+v"(bgl), "+v"(bgh), "+v"(bal), "+v"(bah));                \
            _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                            \
                _Pragma("unroll") for (int q2 = qq; q2 < W3_PPW; q2 += 4)                 \
                    __builtin_amdgcn_global_load_lds(                                     \
                        GLB_PTR((q2 < GPW ? gbase : abase) + (size_t)((KT) + 3) * 64 * (q2 < GPW ? p.N : p.K) + soff[q2]), \
                        LDS_PTR(cur + lds_piece(q2)), 16, 0, 0);                          \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                             \
                    acc[qq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FC.a[qq], FC.g[j], acc[qq][j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                        \
                FN.g[2 * qq] = w3_tr(bgl ^ ((2 * qq) << 5), bgh ^ ((2 * qq) << 5));        \
                FN.g[2 * qq + 1] = w3_tr(bgl ^ ((2 * qq + 1) << 5), bgh ^ ((2 * qq + 1) << 5)); \
                FN.a[qq] = w3_tr(bal ^ (qq << 5), bah ^ (qq << 5));                        \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 4; j < 8; ++j)                             \
                    acc[qq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FC.a[qq], FC.g[j], acc[qq][j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                        \
            }                                                                             \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
        WAIT_VM(W3_PPW);                                                                  \
        WAIT_LGKM0();                                                                     \
        __builtin_amdgcn_s_barrier();                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
#define W3_ITER(FC, FN)                                                                   \
    {                                                                                     \
        const bool more3 = kt + 3 < nk;                                                   \
        if (more3) stage(kt + 3, cur);                                                    \
        if (kt + 1 < nk) read_frags((int)(nxt - lds), FN);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
        mma(FC);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kt + 1 < nk) {                                                                \
            if (more3) WAIT_VM(W3_PPW);                                                   \
            else WAIT_VM(0);                                                              \
            WAIT_LGKM0();                                                                 \
            __builtin_amdgcn_s_barrier();                                                 \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
    int kt = 0;
    // W3_FULL(kt + 1) stages step kt + 4 WITHOUT the row check: the loop must stop before that is the split's last
    // step, which in the last split may be the ragged end of the matrix (M % 32 rows).  (With `kt + 4 < nk` an odd
    // step count in the last split read up to 31 rows past the end of G and A -- found by the ViT-L / N=8192
    // reproducibility test: M = 16392, 101 steps.)
    for (; kt + 5 < nk; kt += 2) {
        W3_FULL(kt, f0, f1)
        W3_FULL(kt + 1, f1, f0)
    }
    for (; kt < nk; kt += 2) {
        W3_ITER(f0, f1)
        ++kt;
        if (kt < nk) W3_ITER(f1, f0)
        --kt;
    }
#undef W3_FULL
#undef W3_ITER

    float* out = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + i * 16 + g * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + li;
            *reinterpret_cast<f32x4*>(out + (size_t)n * p.K + k) = acc[i][j];
        }
    }
}
