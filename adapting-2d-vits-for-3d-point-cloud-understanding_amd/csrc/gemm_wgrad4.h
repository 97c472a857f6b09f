// wgrad v4 (round 5): dW[N,K] = sum_m G[m,N]^T A[m,K] on the STAGGERED 4-phase schedule of gemm_tn4.h -- output tile 256 (n) x 256 (k),
// 8 waves (2 x 4, 128 n x 64 k per wave), contraction tile 64 m-rows, the two wave groups (one wave of each per SIMD) one barrier
// apart, so that in every barrier interval one wave of a SIMD runs 16 MFMAs while its partner issues that phase's fragment reads
// and LDS-DMA.
//
// Why: gemm_wgrad3_kernel (one ring, all 8 waves through one barrier per 32-row step) ran its loop at 1.13-1.22 PFLOP/s whatever the
// loop length (tools/wgrad_longm.py: 8 x longer loops per workgroup gained 13 us per block), while gemm_tn4_kernel's loop sustains
// ~1.45 PFLOP/s on the long-K dgrad (dfc1: 310 GF in 209 us) -- with wgrad3 both waves of a SIMD issue their reads at the same time
// and the matrix pipe idles meanwhile, the very thing the staggered schedule was built against.
//
// Differences from gemm_tn4.h, everything else (ring of ten 16-KiB half-tiles, stage 7 half-tiles ahead, counted vmcnt only in P3 / P4,
// the WAR / RAW distances) is that file's:
//   * both operands are M-MAJOR: a half-tile is 64 m-rows x 128 columns (256-B rows, sixteen 1-KiB LDS-DMA pieces of 4 rows; wave w
//     moves pieces 2w, 2w + 1); inside a row the 32-B column slot is XORed with h(row) = (row & 3) | ((row >> 3) & 1) << 2 on the DMA's
//     SOURCE side (gemm_wgrad3.h's image: conflict-free for the 8 rows a half-wave of a transposing read touches, tools/lds_model.py);
//   * every MFMA fragment (16 columns x 32 m-rows) is TWO ds_read_b64_tr_b16 (inline asm, as in gemm_wgrad3.h: hipcc would park the
//     builtin behind vmcnt(0) while a DMA is in flight); the second 32 rows of the contraction tile are the same addresses + 8 KiB,
//     as the instruction's immediate offset;
//   * half-tile roles per 64-row tile u, consumed in the order h = 4u + s:
//         s = 0: A cols k0..+127    1: A cols +128..+255     ("B" of gemm_tn4.h: a wave reads 2 x 2 of the 16-column blocks of ITS 64 k)
//         s = 2: G cols n0..+127    3: G cols +128..+255     ("A" of gemm_tn4.h: a wave reads 2 x 4 blocks of ITS 128 n)
//   * rows past M (the ragged end of the last split) are staged from a zero page: the branch is wave-uniform (a whole tile is
//     inside the matrix or it is the one ragged tile);
//   * the M-split, the XCD-local split-major work order and the fp32 slabs are gemm_wgrad3.h's (in units of 64 rows).
// Reference: autograd of nn.Linear.weight inside timm Block (src/models/pix4point.py:254-255 via pix4point_trainer.py:165).
#pragma once
#include "gemm_tn4.h"
#include "gemm_wgrad3.h"

struct W4Params {
    const __bf16* G;   // [M,N]
    const __bf16* A;   // [M,K]
    float* out;        // [splits][N][K]
    int M, N, K, tiles_per_split;     // contraction tiles of 64 rows per M-split
};

// one MFMA fragment: transposing reads at LDS byte addresses lo / hi (+ IMM bytes)
template <int IMM>
__device__ __forceinline__ bf16x8 w4_tr(int lo, int hi) {
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4" : "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi), "i"(IMM));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

struct W4Regs {
    bf16x8 g[4][2];      // [n-block of the quadrant][m half]                       (gemm_tn4.h: a)
    bf16x8 a0x[2][2];    // [k-block][m half], k 0..31 of the wave's 64 (two copies: tiles u / u + 1)   (b0x / b0y)
    bf16x8 a0y[2][2];
    bf16x8 a1[2][2];     // k 32..63                                                 (b1)
};

__global__ __launch_bounds__(512) void gemm_wgrad4_kernel(W4Params p) {
    __shared__ __attribute__((aligned(1024))) char lds[T4_SLOTS * T4_HT];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wnh = wave >> 2, wk = wave & 3;              // group = n half (G half-tile); 64-column slice of k
    const int tiles_k = p.K / 256;
    const int ntiles = (p.N / 256) * tiles_k;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);     // split-major ids, contiguous per XCD: the tiles of one M-range share an L2
    const int split = wid / ntiles, tile = wid - split * ntiles;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int total_tiles = (p.M + 63) / 64;
    const int step0 = split * p.tiles_per_split;
    int nk = total_tiles - step0;
    nk = nk < p.tiles_per_split ? nk : p.tiles_per_split;  // >= 1 by construction of the grid

    // ---- LDS-DMA sources.  Piece 2 wave + q of a half-tile = rows 8 wave + 4 q + (lane >> 4); lane l lands at byte 16 (l & 15) of its
    // row, which must hold the LOGICAL 16-B chunk (l & 15) ^ (h(row) << 1).
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 15) * 16;
    u32 soff[4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = wave * 8 + q * 4 + (lane >> 4);
        const int h = (r & 3) | (((r >> 3) & 1) << 2);
        const int chunk = (lane & 15) ^ (h << 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ld = s < 2 ? p.K : p.N, col0 = (s < 2 ? k0 : n0) + (s & 1) * 128;
            soff[s][q] = ((u32)r * (u32)ld + (u32)(col0 + chunk * 8)) * 2u;
        }
    }
    const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)step0 * 64 * p.K * 2;
    const char* Gb = reinterpret_cast<const char*>(p.G) + (size_t)step0 * 64 * p.N * 2;
    const int row_w = wave * 8 + (lane >> 4);               // first of the lane's two rows inside a tile (the second: + 4)
    int st_slot = 0;
    auto stage = [&](int s, int us) {                       // half-tile (tile us of this split, role s): s is a literal at every call
        const char* base = (s < 2 ? Ab : Gb) + (size_t)us * 64 * (s < 2 ? p.K : p.N) * 2;
        char* dst = lds + st_slot * T4_HT + wave * 2048;
        if ((step0 + us + 1) * 64 <= p.M) {                 // wave-uniform: the whole tile is inside the matrix
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][0]), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][1]), LDS_PTR(dst + 1024), 16, 0, 0);
        } else {                                            // the ragged last tile: rows >= M contribute zeros
            const int r0 = (step0 + us) * 64 + row_w;
            const char* s0 = r0 < p.M ? base + soff[s][0] : zero;
            const char* s1 = r0 + 4 < p.M ? base + soff[s][1] : zero;
            __builtin_amdgcn_global_load_lds(GLB_PTR(s0), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(s1), LDS_PTR(dst + 1024), 16, 0, 0);
        }
        st_slot = st_slot == T4_SLOTS - 1 ? 0 : st_slot + 1;
    };

    f32x4 acc[4][8];     // [k-block][n-block]: lane holds n = li, k = 4 g + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing (gemm_wgrad3.h): lane (g, li) supplies row 8 g + (li >> 2) (+ 4 for the second read), 8 bytes at column
    // 4 (li & 3) of a 16-column block, and receives column li of those rows.  Block j of a row: slot j ^ h(row).
    const int g = lane >> 4, li = lane & 15;
    const int r_lo = 8 * g + (li >> 2), r_hi = r_lo + 4;
    const int h_lo = (r_lo & 3) | (((r_lo >> 3) & 1) << 2), h_hi = (r_hi & 3) | (((r_hi >> 3) & 1) << 2);
    const int cb8 = (li & 3) * 8;
    const int lds_base = (int)(size_t)LDS_PTR(lds);
    const int off_lo = r_lo * 256 + cb8 + (h_lo << 5), off_hi = r_hi * 256 + cb8 + (h_hi << 5);
    const int g_half = 2 + wnh, a_half = wk >> 1, a_blk = (wk & 1) * 4;     // half-tile roles; first 16-column block of the wave's 64 k
    int c_slot = 0;                                         // ring slot of role 0 of the tile being consumed
    auto slot_of = [&](int c, int d) { int s = c + d; return s >= T4_SLOTS ? s - T4_SLOTS : s; };
    W4Regs f;
    auto read_g = [&](int mq) {                             // 4 n-blocks x 2 m halves = 16 transposing reads
        const int sb = lds_base + slot_of(c_slot, g_half) * T4_HT;
        int bl = sb + off_lo, bh = sb + off_hi;
        asm volatile("" : "+v"(bl), "+v"(bh));              // not loop-invariant for hipcc (a hoisted address would be spilled)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int x = (mq * 4 + mt) << 5;
            f.g[mt][0] = w4_tr<0>(bl ^ x, bh ^ x);
            f.g[mt][1] = w4_tr<8192>(bl ^ x, bh ^ x);
        }
    };
    auto read_a = [&](bf16x8 (&b)[2][2], int c, int nq) {   // 2 k-blocks x 2 m halves = 8 transposing reads, tile whose role 0 sits in slot c
        const int sb = lds_base + slot_of(c, a_half) * T4_HT;
        int bl = sb + off_lo, bh = sb + off_hi;
        asm volatile("" : "+v"(bl), "+v"(bh));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int x = (a_blk + nq * 2 + nt) << 5;
            b[nt][0] = w4_tr<0>(bl ^ x, bh ^ x);
            b[nt][1] = w4_tr<8192>(bl ^ x, bh ^ x);
        }
    };
    auto mma16 = [&](const bf16x8 (&b)[2][2], int i0, int j0) {     // quadrant: k-blocks i0, i0 + 1; n-blocks j0 .. j0 + 3
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[i0 + nt][j0 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][mh], f.g[mt][mh], acc[i0 + nt][j0 + mt], 0, 0, 0);
    };

    // ---- prologue: half-tiles 0..6 in flight; A(0) and G(0) landed and published; a0 of tile 0 in registers
#pragma unroll
    for (int h = 0; h < 7; ++h)
        if ((h >> 2) < nk) stage(h & 3, h >> 2);
    if (nk >= 2) WAIT_VM(6);        // 7 half-tiles issued: everything but h = 4, 5, 6
    else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_a(f.a0x, c_slot, 0);
    if (wnh == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier interval behind group 0
    __builtin_amdgcn_sched_barrier(0);

#define W4_MFMA_SECTION(B, I0, J0)                 \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    WAIT_LGKM0();                                  \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_setprio(1);                 \
    mma16(B, I0, J0);                              \
    __builtin_amdgcn_s_setprio(0);                 \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0);

    // one contraction tile; BC = this tile's a0 registers, BN = the next tile's.  FAST: steady state (u + 2 < nk): every staging exists
    // and the waits are counted; otherwise stagings are guarded and the waits drain.  (gemm_tn4.h's T4_KTILE with the roles renamed.)
#define W4_TILE(BC, BN, FAST)                                                               \
    {                                                                                       \
        /* P1 */                                                                            \
        read_g(0);                                                                          \
        if (FAST || u + 1 < nk) stage(3, u + 1);                                            \
        W4_MFMA_SECTION(BC, 0, 0)                                                           \
        /* P2 */                                                                            \
        read_a(f.a1, c_slot, 1);                                                            \
        if (FAST || u + 2 < nk) stage(0, u + 2);                                            \
        W4_MFMA_SECTION(f.a1, 2, 0)                                                         \
        /* P3 */                                                                            \
        read_g(1);                                                                          \
        if (FAST || u + 2 < nk) stage(1, u + 2);                                            \
        if (FAST) WAIT_VM(8);                                                               \
        else WAIT_VM(0);                                                                    \
        W4_MFMA_SECTION(f.a1, 2, 4)                                                         \
        /* P4 */                                                                            \
        if (FAST || u + 1 < nk) read_a(BN, slot_of(c_slot, 4), 0);                          \
        if (FAST || u + 2 < nk) stage(2, u + 2);                                            \
        if (FAST) WAIT_VM(6);                                                               \
        else WAIT_VM(0);                                                                    \
        W4_MFMA_SECTION(BC, 0, 4)                                                           \
        c_slot = slot_of(c_slot, 4);                                                        \
    }

    int u = 0;
    for (; u + 3 < nk; u += 2) {                           // both tiles of the pair are in the steady state
        W4_TILE(f.a0x, f.a0y, true)
        ++u;
        W4_TILE(f.a0y, f.a0x, true)
        --u;
    }
    for (; u < nk; u += 2) {
        W4_TILE(f.a0x, f.a0y, false)
        ++u;
        if (u < nk) W4_TILE(f.a0y, f.a0x, false)
        --u;
    }
#undef W4_TILE
#undef W4_MFMA_SECTION
    if (wnh == 0) __builtin_amdgcn_s_barrier();            // group 0 catches up: every wave has executed the same count
    __builtin_amdgcn_sched_barrier(0);

    float* out = p.out + (size_t)split * p.N * p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + wk * 64 + i * 16 + g * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wnh * 128 + j * 16 + li;
            *reinterpret_cast<f32x4*>(out + (size_t)n * p.K + k) = acc[i][j];
        }
    }
}
