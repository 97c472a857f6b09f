// gemm_tn v4: C[M,N] = A[M,K] . W[N,K]^T, block tile 256 x 256, 8 waves (2 x 4, 128 x 64 per wave), K-tile 64,
// a 4-phase-per-K-tile schedule with the two wave groups (wm = 0 / 1: one wave of each group per SIMD) running
// STAGGERED by one barrier, after the "256^2 8-phase" recipe of guides/cdna_hip_programming.md section 5.
//
// Why: in gemm_tn3 all 8 waves pass the same barrier once per K-step, so both waves of a SIMD issue their LDS-DMA and
// fragment reads at the same time and the matrix pipe idles meanwhile (measured: MFMA busy ~51 %, K loop 1.1 PFLOP/s).
// Here every phase is   { load section | s_barrier | 16 MFMAs | s_barrier }   and group 1 runs one barrier behind
// group 0, so in every barrier interval one wave per SIMD is in its MFMA section (256 cycles of matrix pipe) while
// its partner issues that phase's ds_reads and LDS-DMA.
//
// Operands in LDS: half-tiles of 128 rows x 64 bf16 (16 KiB, 128-B rows, 16-B chunk' = chunk ^ (row & 7): the
// conflict-free image of gemm_tn2).  A K-tile is four half-tiles, consumed in the order h = 4u + s,
//   s = 0: W rows n0..+127 (B0)   1: W rows +128..+255 (B1)   2: A rows m0..+127 (A0)   3: A rows +128..+255 (A1),
// and half-tile h lives in ring slot h % 10 (10 x 16 KiB = all 160 KiB of LDS).  Per K-tile u and wave:
//   P1: read a <- A[wm] rows 0..63 (8 x b128)        | MFMA quadrant (m 0..63,   n 0..31)  with a, b0
//   P2: read b1 <- B[wn>>1] second 32 rows (4)       | MFMA quadrant (m 0..63,   n 32..63) with a, b1
//   P3: read a <- A[wm] rows 64..127 (8)             | MFMA quadrant (m 64..127, n 32..63) with a, b1
//   P4: read b0' <- B of K-tile u+1, first 32 rows (4) | MFMA quadrant (m 64..127, n 0..31) with a, b0
// (b0 is double-buffered in registers: the loop body is unrolled over two K-tiles).
//
// LDS-DMA: phase phi = 4u + p stages half-tile phi + 7 (2 x global_load_lds_dwordx4 per wave), i.e. the prologue puts 7
// half-tiles in flight.  Ordering rules (guide: "read a staged buffer one phase AFTER the wait that retires it";
// "restage a buffer >= 2 phases after its last ds_read" because the groups are staggered):
//   last reads : B*(u) in P2(u) = phase 4u+1, A*(u) in P3(u) = 4u+2;   first reads: B*(u) in P4(u-1), A*(u) in P1(u)
//   WAR: half-tile h overwrites h-10: (s=0,1) A0/A1(u-3), last read 4u-10 -> staged at 4u-7+s: >= 2 phases later;
//        (s=2,3) B0/B1(u-2), last read 4u-7 -> staged at 4u-5+(s-2): >= 2 phases later.
//   RAW: counted waits only in P3 and P4 (never vmcnt(0) in the steady state):
//        P3(u): B*(u+1) = h <= 4u+5 must have landed, staged so far h <= 4u+9  -> s_waitcnt vmcnt(8), read in P4(u);
//        P4(u): A*(u+1) = h <= 4u+7, staged so far h <= 4u+10                  -> s_waitcnt vmcnt(6), read in P1(u+1).
//        Every half-tile is in flight for >= 3 phases (~1500 cycles) before it is waited for.
#pragma once
#include "gemm_tn3.h"

#define T4_HT 16384
#define T4_SLOTS 10

struct T4Regs {
    bf16x8 a[4][2];      // [m-tile of the quadrant][k half]
    bf16x8 b0x[2][2];    // [n-tile][k half], n 0..31 of the wave tile (two copies: K-tiles u / u+1)
    bf16x8 b0y[2][2];
    bf16x8 b1[2][2];     // n 32..63
};

template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn4_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[T4_SLOTS * T4_HT];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n;
    const int m0 = tm * 256, n0 = (tile - tm * tiles_n) * 256;
    const int nk = p.K / 64;

    // ---- LDS-DMA sources: wave w moves the 1-KiB pieces 2w, 2w+1 (rows 16w .. 16w+15) of every half-tile.
    // 32-bit byte offsets from the (wave-uniform) matrix bases; the launcher guarantees M*K*2 < 4 GiB.
    const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
    u32 soff[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int r = (s < 2 ? n0 + s * 128 : m0 + (s - 2) * 128) + wave * 16 + q * 8 + rin;
            const int lim = s < 2 ? p.N : p.M;
            r = r < lim ? r : lim - 1;
            soff[s][q] = (u32)r * (u32)p.K * 2u + (u32)chunk * 16u;
        }
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    int st_slot = 0;                                       // ring slot of the next half-tile to stage
    auto stage = [&](int s, int us) {                      // half-tile (K-tile us, role s): s is a literal at every call
        const char* base = (s < 2 ? Wb : Ab) + (size_t)us * 128;
        char* dst = lds + st_slot * T4_HT + wave * 2048;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][0]), LDS_PTR(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][1]), LDS_PTR(dst + 1024), 16, 0, 0);
        st_slot = st_slot == T4_SLOTS - 1 ? 0 : st_slot + 1;
    };

    // ---- fragment addressing: row li of a 16-row block, 16-B chunk (4*kh + g) ^ (li & 7)
    const int li = lane & 15, g = lane >> 4;
    f32x4 acc[4][8];     // [n-tile][m-tile]; EPI_BF16_ROWGROUP: the per-group addend is the initial value (see gemm_tn2_kernel)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_BF16_ROWGROUP) {
                const int m = m0 + wm * 128 + j * 16 + li;
                if (m < p.M) acc[i][j] = *reinterpret_cast<const f32x4*>(p.rowgrp + (size_t)(m / p.group) * p.N + n0 + wn * 64 + i * 16 + g * 4);
            }
        }

    const int fo0 = li * 128 + ((g ^ (li & 7)) << 4), fo1 = fo0 ^ 64;
    const int a_half = 2 + wm, b_half = wn >> 1, b_row = (wn & 1) * 8192;     // half-tile roles, byte offset of the wave's 64 W rows
    int c_slot = 0;                                        // ring slot of B0 of the K-tile being consumed
    auto slot_of = [&](int c, int d) { int s = c + d; return s >= T4_SLOTS ? s - T4_SLOTS : s; };
    T4Regs f;
    auto read_a = [&](int mq) {                            // 8 x ds_read_b128
        const char* base = lds + slot_of(c_slot, a_half) * T4_HT + mq * 8192;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f.a[mt][0] = *reinterpret_cast<const bf16x8*>(base + fo0 + mt * 2048);
            f.a[mt][1] = *reinterpret_cast<const bf16x8*>(base + fo1 + mt * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&b)[2][2], int c, int nq) {  // 4 x ds_read_b128 from the K-tile whose B0 sits in slot c
        const char* base = lds + slot_of(c, b_half) * T4_HT + b_row + nq * 4096;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            b[nt][0] = *reinterpret_cast<const bf16x8*>(base + fo0 + nt * 2048);
            b[nt][1] = *reinterpret_cast<const bf16x8*>(base + fo1 + nt * 2048);
        }
    };
    auto mma16 = [&](const bf16x8 (&b)[2][2], int i0, int j0) {     // quadrant: n-tiles i0, i0+1; m-tiles j0 .. j0+3
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[i0 + nt][j0 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][kh], f.a[mt][kh], acc[i0 + nt][j0 + mt], 0, 0, 0);
    };

    // ---- prologue: half-tiles 0..6 in flight; B(0) and A(0) landed and published; b0 of K-tile 0 in registers
#pragma unroll
    for (int h = 0; h < 7; ++h)
        if ((h >> 2) < nk) stage(h & 3, h >> 2);
    if (nk >= 2) WAIT_VM(6);        // 7 half-tiles issued: everything but h = 4, 5, 6
    else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_b(f.b0x, c_slot, 0);
    if (wm == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier interval behind group 0
    __builtin_amdgcn_sched_barrier(0);

#define T4_MFMA_SECTION(B, I0, J0)                 \
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

    // one K-tile; BC = this K-tile's b0 registers, BN = the next K-tile's.  FAST: steady state (u + 2 < nk): every staging
    // exists and the waits are counted; otherwise stagings are guarded and the waits drain.
#define T4_KTILE(BC, BN, FAST)                                                              \
    {                                                                                       \
        /* P1 */                                                                            \
        read_a(0);                                                                          \
        if (FAST || u + 1 < nk) stage(3, u + 1);                                            \
        T4_MFMA_SECTION(BC, 0, 0)                                                           \
        /* P2 */                                                                            \
        read_b(f.b1, c_slot, 1);                                                            \
        if (FAST || u + 2 < nk) stage(0, u + 2);                                            \
        T4_MFMA_SECTION(f.b1, 2, 0)                                                         \
        /* P3 */                                                                            \
        read_a(1);                                                                          \
        if (FAST || u + 2 < nk) stage(1, u + 2);                                            \
        if (FAST) WAIT_VM(8);                                                               \
        else WAIT_VM(0);                                                                    \
        T4_MFMA_SECTION(f.b1, 2, 4)                                                         \
        /* P4 */                                                                            \
        if (FAST || u + 1 < nk) read_b(BN, slot_of(c_slot, 4), 0);                          \
        if (FAST || u + 2 < nk) stage(2, u + 2);                                            \
        if (FAST) WAIT_VM(6);                                                               \
        else WAIT_VM(0);                                                                    \
        T4_MFMA_SECTION(BC, 0, 4)                                                           \
        c_slot = slot_of(c_slot, 4);                                                        \
    }

    int u = 0;
    for (; u + 3 < nk; u += 2) {                           // both K-tiles of the pair are in the steady state
        T4_KTILE(f.b0x, f.b0y, true)
        ++u;
        T4_KTILE(f.b0y, f.b0x, true)
        --u;
    }
    for (; u < nk; u += 2) {
        T4_KTILE(f.b0x, f.b0y, false)
        ++u;
        if (u < nk) T4_KTILE(f.b0y, f.b0x, false)
        --u;
    }
#undef T4_KTILE
#undef T4_MFMA_SECTION
    if (wm == 0) __builtin_amdgcn_s_barrier();             // group 0 catches up: every wave has executed the same count
    __builtin_amdgcn_sched_barrier(0);
    tile_epilogue_128x64<EPI, true>(p, acc, lds + wave * 4608, tm, m0, n0, wm, wn, wave, lane);
}
