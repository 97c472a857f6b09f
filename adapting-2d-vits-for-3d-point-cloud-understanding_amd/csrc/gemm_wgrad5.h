// wgrad v5 (round 5): dW[N,K] = sum_m G[m,N]^T A[m,K], output tile 256 (n) x 256 (k) by FOUR waves of 128 x 128 -- one wave per SIMD
// with the whole 512-entry register file (256 accumulator registers + two fragment sets), gemm_wgrad3.h's LDS image, ring and
// hand-interleaved issue order otherwise.
//
// Why: both operands are M-major, so every MFMA fragment costs TWO ds_read_b64_tr_b16 -- the weight gradient issues twice the LDS
// instructions per fragment of the forward GEMMs.  With 128 x 64 per wave (gemm_wgrad3.h, 8 waves) a SIMD issues, per 32-row step of
// its two waves, 64 MFMAs (1,024 cycles, each holding the vector issue port for 8 of its 16) + 48 transposing reads + 12 LDS-DMAs +
// ~48 address VALU: the issue port is ~85 % full and the loop runs at 1.13-1.22 PFLOP/s whatever its length; on the staggered schedule
// of gemm_tn4.h (gemm_wgrad4.h) the partner wave cannot even fit its 16 reads + VALU into one 256-cycle MFMA section (0.90-0.99
// PFLOP/s).  A 128 x 128 wave tile needs (8 + 8) fragments for 64 MFMAs instead of (8 + 4) for 32: a third fewer LDS instructions per
// MFMA, one wave's stream per SIMD.
//
// Layout per 32-row step (gemm_wgrad3.h's wide variant): G rows [32][256 cols] then A rows [32][256 cols], 512-B rows, the 32-B column
// slot inside every 256-B window XORed with h(row) = (row & 3) | ((row >> 3) & 1) << 2; 32 KiB per step, 3 steps in the ring.
// A wave moves 8 of the 32 one-KiB pieces of a step (4 of G, 4 of A: 2 rows each).
// Issue order of a steady-state step: 8 groups { 1 LDS-DMA | 4 MFMA | 4 transposing reads (next step's G and A fragment of this group)
// | 4 MFMA }, then vmcnt(8) (the step just staged may stay in flight), lgkmcnt(0), one barrier.
// Reference: autograd of nn.Linear.weight inside timm Block (src/models/pix4point.py:254-255 via pix4point_trainer.py:165).
#pragma once
#include "gemm_wgrad3.h"

#define W5_STAGE (32 * 512 * 2)      // 32 KiB per 32-row step
#define W5_PPW 8                     // LDS-DMA pieces per wave and step

// MFMA with the accumulator tied in place to an AGPR tuple.  Inline asm on purpose: with all 256 accumulator registers live hipcc's
// allocator resolves the loop-carried accumulators by ROTATING them (255 v_accvgpr_mov per 128 MFMAs in the builtin form of this loop).
// Hazards the compiler would otherwise pad are covered by the loop's structure: operands come from LDS reads retired by lgkmcnt(0) a whole
// step earlier, an accumulator is touched once per step, and the epilogue waits out the last MFMA (s_nop) before it reads the AGPRs.
__device__ __forceinline__ void w5_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

struct W5Frags {
    bf16x8 a[8];
    bf16x8 g[8];
};

template <bool ILV>
__global__ __launch_bounds__(256) void gemm_wgrad5_kernel(W3Params p) {
    __shared__ __attribute__((aligned(1024))) char lds[3 * W5_STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = p.K / 256;
    const int ntiles = (p.N / 256) * tiles_k;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);     // split-major ids, contiguous per XCD (gemm_wgrad3.h)
    const int split = wid / ntiles, tile = wid - split * ntiles;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int total_steps = (p.M + 31) / 32;
    const int step0 = split * p.steps_per_split;
    int nk = total_steps - step0;
    nk = nk < p.steps_per_split ? nk : p.steps_per_split;          // >= 1 by construction of the grid

    // ---- staging: pieces [0,16) of a step are G (2 rows of 512 B each), [16,32) are A; wave w moves G pieces 4w..4w+3 and A pieces
    // 4w..4w+3.  Lane l of a piece lands at byte 16 l: row l >> 5, physical chunk l & 31, which holds logical chunk (l & 31) ^ (h << 1).
    const __bf16* zero = reinterpret_cast<const __bf16*>(g_zero_page);
    u32 soff[W5_PPW];
    const char* gbase = reinterpret_cast<const char*>(p.G) + (size_t)step0 * 32 * p.N * 2;   // wave-uniform
    const char* abase = reinterpret_cast<const char*>(p.A) + (size_t)step0 * 32 * p.K * 2;
#pragma unroll
    for (int q = 0; q < W5_PPW; ++q) {
        const int r = (wave * 4 + (q & 3)) * 2 + (lane >> 5);
        const int h = (r & 3) | (((r >> 3) & 1) << 2);
        soff[q] = ((u32)r * (q < 4 ? p.N : p.K) + (q < 4 ? n0 : k0) + (((lane & 31) ^ (h << 1)) * 8)) * 2u;
    }
    auto lds_piece = [&](int q) { return ((q < 4 ? 0 : 16) + wave * 4 + (q & 3)) * 1024; };
    auto stage_one = [&](int kt, char* buf, int q) {                 // unchecked: the tile is inside the matrix
        __builtin_amdgcn_global_load_lds(GLB_PTR((q < 4 ? gbase : abase) + (size_t)kt * 64 * (q < 4 ? p.N : p.K) + soff[q]),
                                         LDS_PTR(buf + lds_piece(q)), 16, 0, 0);
    };
    auto stage = [&](int kt, char* buf) {                            // checked: rows >= M come from the zero page
        const int mbase = (step0 + kt) * 32;
#pragma unroll
        for (int q = 0; q < W5_PPW; ++q) {
            const int r = (wave * 4 + (q & 3)) * 2 + (lane >> 5);
            const __bf16* gp = reinterpret_cast<const __bf16*>((q < 4 ? gbase : abase) + (size_t)kt * 64 * (q < 4 ? p.N : p.K) + soff[q]);
            gp = mbase + r < p.M ? gp : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(buf + lds_piece(q)), 16, 0, 0);
        }
    };

    f32x4 acc[8][8];     // [k-block][n-block]: lane holds n = li, k = 4 g + r
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing read (gemm_wgrad3.h): lane (g, li) supplies row 8 g + (li >> 2) (+ 4 for the second read), 8 bytes at column
    // 4 (li & 3) of a 16-column block; it receives column li of those rows.  Block j of a 256-B window: slot j ^ h(row).
    const int g = lane >> 4, li = lane & 15;
    const int r_lo = 8 * g + (li >> 2), r_hi = r_lo + 4;
    const int h_lo = (r_lo & 3) | (((r_lo >> 3) & 1) << 2), h_hi = (r_hi & 3) | (((r_hi >> 3) & 1) << 2);
    const int cb8 = (li & 3) * 8;
    const int lds_base = (int)(size_t)LDS_PTR(lds);
    const int g_lo = r_lo * 512 + wn * 256 + cb8 + (h_lo << 5), g_hi = r_hi * 512 + wn * 256 + cb8 + (h_hi << 5);
    const int a_lo = 16384 + r_lo * 512 + wk * 256 + cb8 + (h_lo << 5), a_hi = 16384 + r_hi * 512 + wk * 256 + cb8 + (h_hi << 5);
    auto read_frags = [&](int buf_off, W5Frags& f) {
        int bgl = lds_base + buf_off + g_lo, bgh = lds_base + buf_off + g_hi;
        int bal = lds_base + buf_off + a_lo, bah = lds_base + buf_off + a_hi;
        asm volatile("" : "+v"(bgl), "+v"(bgh), "+v"(bal), "+v"(bah));      // not loop-invariant for hipcc
#pragma unroll
        for (int j = 0; j < 8; ++j) f.g[j] = w3_tr(bgl ^ (j << 5), bgh ^ (j << 5));
#pragma unroll
        for (int i = 0; i < 8; ++i) f.a[i] = w3_tr(bal ^ (i << 5), bah ^ (i << 5));
    };
    auto mma = [&](const W5Frags& f) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                w5_mfma(acc[i][j], f.a[i], f.g[j]);
    };

    char* s0 = lds;
    char* s1 = lds + W5_STAGE;
    char* s2 = lds + 2 * W5_STAGE;
    W5Frags f0, f1;
    stage(0, s0);
    if (nk > 1) stage(1, s1);
    if (nk > 2) stage(2, s2);
    if (nk > 2) WAIT_VM(2 * W5_PPW);
    else if (nk > 1) WAIT_VM(W5_PPW);
    else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    read_frags(0, f0);
    if (nk > 2) WAIT_VM(W5_PPW);
    else WAIT_VM(0);
    WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    char *cur = s0, *nxt = s1, *nn = s2;
    // steady state: step KT's fragments are in FC; stage step KT + 3 into the buffer FC came from, read step KT + 1 into FN
#define W5_FULL(KT, FC, FN)                                                               \
    {                                                                                     \
        if (!ILV) {                                                                       \
            _Pragma("unroll") for (int q2 = 0; q2 < W5_PPW; ++q2) stage_one((KT) + 3, cur, q2); \
            read_frags((int)(nxt - lds), FN);                                             \
            __builtin_amdgcn_sched_barrier(0);                                            \
            mma(FC);                                                                      \
        } else {                                                                          \
            const int bo = (int)(nxt - lds);                                              \
            int bgl = lds_base + bo + g_lo, bgh = lds_base + bo + g_hi;                    \
            int bal = lds_base + bo + a_lo, bah = lds_base + bo + a_hi;                    \
            asm volatile("" : "+v"(bgl), "+v"(bgh), "+v"(bal), "+v"(bah));                \
            _Pragma("unroll") for (int qq = 0; qq < 8; ++qq) {                            \
                stage_one((KT) + 3, cur, qq);                                             \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                             \
                    w5_mfma(acc[qq][j], FC.a[qq], FC.g[j]);                               \
                __builtin_amdgcn_sched_barrier(0);                                        \
                FN.g[qq] = w3_tr(bgl ^ (qq << 5), bgh ^ (qq << 5));                        \
                FN.a[qq] = w3_tr(bal ^ (qq << 5), bah ^ (qq << 5));                        \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 4; j < 8; ++j)                             \
                    w5_mfma(acc[qq][j], FC.a[qq], FC.g[j]);                               \
                __builtin_amdgcn_sched_barrier(0);                                        \
            }                                                                             \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
        WAIT_VM(W5_PPW);                                                                  \
        WAIT_LGKM0();                                                                     \
        __builtin_amdgcn_s_barrier();                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
#define W5_ITER(FC, FN)                                                                   \
    {                                                                                     \
        const bool more3 = kt + 3 < nk;                                                   \
        if (more3) stage(kt + 3, cur);                                                    \
        if (kt + 1 < nk) read_frags((int)(nxt - lds), FN);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
        mma(FC);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kt + 1 < nk) {                                                                \
            if (more3) WAIT_VM(W5_PPW);                                                   \
            else WAIT_VM(0);                                                              \
            WAIT_LGKM0();                                                                 \
            __builtin_amdgcn_s_barrier();                                                 \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
    int kt = 0;
    // W5_FULL(kt + 1) stages step kt + 4 WITHOUT the row check: the loop stops before that can be the split's last step, which in the
    // last split may be the ragged end of the matrix (gemm_wgrad3.h: the same bound, found there by a reproducibility test)
    for (; kt + 5 < nk; kt += 2) {
        W5_FULL(kt, f0, f1)
        W5_FULL(kt + 1, f1, f0)
    }
    for (; kt < nk; kt += 2) {
        W5_ITER(f0, f1)
        ++kt;
        if (kt < nk) W5_ITER(f1, f0)
        --kt;
    }
#undef W5_FULL
#undef W5_ITER

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs have left the matrix pipe before the AGPRs are read
    float* out = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = k0 + wk * 128 + i * 16 + g * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + li;
            *reinterpret_cast<f32x4*>(out + (size_t)n * p.K + k) = acc[i][j];
        }
    }
}
