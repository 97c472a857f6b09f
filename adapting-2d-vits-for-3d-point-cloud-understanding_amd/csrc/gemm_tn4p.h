// gemm_tn v4p: the staggered 4-phase kernel of gemm_tn4.h as a PERSISTENT tile loop (one workgroup per CU, grid = 256).
//
// Why: with 12 K-tiles per output tile (K = 768: qkv, proj, fc1, dfc2, dproj) the per-tile fixed cost -- workgroup dispatch,
// 7 half-tiles of LDS-DMA latency before the first MFMA -- is ~10 % of the tile (K loop alone 1,100 TFLOP/s at K = 768
// against ~1,250 at K = 3072).  Here the K loop simply keeps running across the seam: the last two K-tiles of a tile stage the
// first 7 half-tiles of the workgroup's NEXT tile (the ring positions they would have had if K were longer), the epilogue runs
// while they land, and the next tile starts with its operands in LDS.  Same-box A/B against gemm_tn4.h (bit-identical outputs):
// qkv 238.6 -> 218.5 us, fc1 389.9 -> 370.0, proj 137.4 -> 133.6, dfc2 344.9 -> 338.1, dqkv 202.5 -> 195.7 at M = 65,792;
// K = 3072 (fc2, dfc1) gains nothing or loses (dfc1 249 -> 263) and stays on gemm_tn4.h; whole step -0.35 ms at N = 1024.
//
// Seam protocol (everything else is gemm_tn4.h; K/64 must be even so that the two b0 register sets keep their parity):
//   ONE K loop, always in the steady state (counted vmcnt(8) / vmcnt(6), no drain at the end of a tile): STAGE(s, us) with
//                      us >= nk goes to K-tile us - nk of the next tile.  The next tile's source offsets are this tile's per-lane
//                      offsets plus a WAVE-UNIFORM byte delta ((m0' - m0) K 2 for A, (n0' - n0) K 2 for W) folded into the scalar
//                      base pointer -- no second set of offset registers, and the loop body is the same code for every K-tile.
//                      (Two earlier forms -- next-tile offsets computed on the spot, or moved over in place, in a separate copy of
//                      the last K-tile pair -- took 230-390 spills at the 256-register limit, whose reloads each drain vmcnt(0):
//                      1.5-1.8 x SLOWER than gemm_tn4.h.  This form: 230 VGPRs, no spills.)
//   no seam          : if this tile or the next lies in the ragged last row panel (rows clamped per lane: no uniform delta), or
//                      there is no next tile, the delta is 0: the seam stages this tile's own K-tiles 0, 1 again (valid addresses,
//                      never read), and the next tile drains them and starts from a fresh prologue.
//   catch-up barrier : group 0 waits for group 1 as in gemm_tn4.h; after it nobody reads the ring.
//   epilogue         : the 7 staged half-tiles sit in slots c .. c+6 (c = slot of the next tile's B0); the per-wave slabs go to the
//                      three free slots c+7 .. c+9 (three 4.5-KiB slabs per 16-KiB slot).
//   wait before the next tile: `s_waitcnt vmcnt(6)`, the wait of gemm_tn4.h's prologue, although vmcnt now also counts the
//                      epilogue's own loads / stores, all issued AFTER the seam's 14 DMA instructions.  If vmcnt retires strictly
//                      in issue order (what hipcc assumes on this target, and the guide's "vmcnt counts stores too"), at most 6
//                      outstanding means that everything older than the newest 6 is done: a full tile's epilogue issues >= 16
//                      operations per wave, so every DMA has landed.  If only loads are ordered among loads and stores among
//                      stores, the worst case is that all stores are done and the 6 outstanding are the newest loads -- the DMAs
//                      of half-tiles 4..6 at most, exactly what the prologue of gemm_tn4.h leaves in flight; the counted waits of
//                      the first K-tiles then retire them as there.  Either way half-tiles 0..3 are in LDS, and nothing is assumed
//                      about the epilogue's operation count beyond "a seam exists only behind a full tile".
//   first K-tiles of the next tile: their counted waits see a few old stores in front of the DMAs: stricter, never weaker.
//   exit             : vmcnt(0): no DMA may outlive the workgroup's LDS allocation.
#pragma once
#include "gemm_tn4.h"

// Diagnostic build only (tools/gemm_stamps.py compiles gemm.hip with -DP4P_T4P_STAMPS into its own library): s_memtime stamps of
// every wave at the phase boundaries of every tile of the persistent loop, [workgroup][tile of the walk < 16][wave][6] u64 through a
// device pointer set by p4p_debug_set_t4p_stamps.  No stamp in the product library.
#ifdef P4P_T4P_STAMPS
__device__ unsigned long long* g_t4p_stamps;
extern "C" int p4p_debug_set_t4p_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_t4p_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 2;
}
#define T4P_STAMP(k)                                                                                                   \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0 && walk_i < 16)                                                                    \
            g_t4p_stamps[(((size_t)blockIdx.x * 16 + walk_i) * 8 + (threadIdx.x >> 6)) * 6 + (k)] = ts_;               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
#else
#define T4P_STAMP(k) do { } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn4p_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[T4_SLOTS * T4_HT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256, ntiles = tiles_m * tiles_n;
    const int nk = p.K / 64;                               // even (launcher)
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const int a_half = 2 + wm, b_half = wn >> 1, b_row = (wn & 1) * 8192;
    // lane-derived values are re-made per tile from v_mbcnt (volatile asm: not hoistable): kept live across the epilogue they
    // cost more registers than the 256 there are (315 spills in the first version of this kernel)
    int lane, rin, chunk, li, g, fo0, fo1;

    int st_slot = 0, c_slot = 0;
    auto slot_of = [&](int c, int d) { int s = c + d; return s >= T4_SLOTS ? s - T4_SLOTS : s; };

    T4Regs f;
    f32x4 acc[4][8];
    auto read_a = [&](int mq) {
        const char* base = lds + slot_of(c_slot, a_half) * T4_HT + mq * 8192;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f.a[mt][0] = *reinterpret_cast<const bf16x8*>(base + fo0 + mt * 2048);
            f.a[mt][1] = *reinterpret_cast<const bf16x8*>(base + fo1 + mt * 2048);
        }
    };
    auto read_b = [&](bf16x8 (&b)[2][2], int c, int nq) {
        const char* base = lds + slot_of(c, b_half) * T4_HT + b_row + nq * 4096;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            b[nt][0] = *reinterpret_cast<const bf16x8*>(base + fo0 + nt * 2048);
            b[nt][1] = *reinterpret_cast<const bf16x8*>(base + fo1 + nt * 2048);
        }
    };
    auto mma16 = [&](const bf16x8 (&b)[2][2], int i0, int j0) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[i0 + nt][j0 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][kh], f.a[mt][kh], acc[i0 + nt][j0 + mt], 0, 0, 0);
    };

#define T4P_MFMA_SECTION(B, I0, J0)                \
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
    // one K-tile, always in the steady state (counted waits): STAGE(s, us) with us >= nk goes to the next tile
#define T4P_KTILE(BC, BN)                                                                   \
    {                                                                                       \
        read_a(0);                                                                          \
        T4P_STAGE(3, u + 1);                                                                \
        T4P_MFMA_SECTION(BC, 0, 0)                                                          \
        read_b(f.b1, c_slot, 1);                                                            \
        T4P_STAGE(0, u + 2);                                                                \
        T4P_MFMA_SECTION(f.b1, 2, 0)                                                        \
        read_a(1);                                                                          \
        T4P_STAGE(1, u + 2);                                                                \
        WAIT_VM(8);                                                                         \
        T4P_MFMA_SECTION(f.b1, 2, 4)                                                        \
        read_b(BN, slot_of(c_slot, 4), 0);                                                  \
        T4P_STAGE(2, u + 2);                                                                \
        WAIT_VM(6);                                                                         \
        T4P_MFMA_SECTION(BC, 0, 4)                                                          \
        c_slot = slot_of(c_slot, 4);                                                        \
    }

    // every XCD's run of tiles is whole bands of 8 row panels (gemm_tn.h: band_remap)
    const bool banded = p.band && tiles_n % 4 == 0 && tiles_n >= 8 && ntiles % 8 == 0 && (ntiles / 8) % (8 * tiles_n) == 0;
    bool fresh = true;                                     // the tile's first 7 half-tiles are not in flight yet
    // Start-time shift (experiment, p4p_set_gemm_epilogue bits 8..: units of 1,024 cycles): every other workgroup of an XCD begins late,
    // so that its epilogues -- HBM-bound when all 256 CUs store and load at once (tools/gemm_stamps.py: the proj epilogue takes 47k
    // cycles on 256 CUs, 30k on 128) -- fall into the other half's K loops, which leave HBM idle.  The shift persists: the loop is persistent.
    if ((p.epi_flags >> 8) && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = __builtin_readcyclecounter(), d = (unsigned long long)(p.epi_flags >> 8) * 1024;
        while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(32);
    }
    int walk_i = -1;                                       // (stamps only)
    for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
        ++walk_i;
        T4P_STAMP(0);
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        rin = lane >> 3; chunk = (lane & 7) ^ rin; li = lane & 15; g = lane >> 4;
        fo0 = li * 128 + ((g ^ (li & 7)) << 4); fo1 = fo0 ^ 64;
        const int tile = banded ? band_remap(xcd_remap(it, ntiles), tiles_n) : xcd_remap(it, ntiles);
        const int tm = tile / tiles_n;
        const int m0 = tm * 256, n0 = (tile - tm * tiles_n) * 256;
        // the seam: the next tile's source offsets are this tile's plus a wave-uniform byte delta, unless one of the two tiles is
        // in the ragged last row panel (its rows are clamped per lane); then the seam stages this tile's K-tiles 0, 1 again (valid
        // addresses, never read) and the next tile starts from a fresh prologue
        bool seam = false;
        long long dA = 0, dW = 0;
        if (it + (int)gridDim.x < ntiles) {
            const int tl = banded ? band_remap(xcd_remap(it + gridDim.x, ntiles), tiles_n) : xcd_remap(it + gridDim.x, ntiles), tmn = tl / tiles_n;
            const int m0n = tmn * 256, n0n = (tl - tmn * tiles_n) * 256;
            seam = m0 + 256 <= p.M && m0n + 256 <= p.M;
            if (seam) {
                dA = (long long)(m0n - m0) * p.K * 2;
                dW = (long long)(n0n - n0) * p.K * 2;
            }
        }
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
        auto issue = [&](int s, long long koff) {          // s is a literal at every call; koff is wave-uniform
            const char* base = (s < 2 ? Wb : Ab) + koff;
            char* dst = lds + st_slot * T4_HT + wave * 2048;
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][0]), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(base + soff[s][1]), LDS_PTR(dst + 1024), 16, 0, 0);
            st_slot = st_slot == T4_SLOTS - 1 ? 0 : st_slot + 1;
        };
#define T4P_STAGE(s, us) issue(s, (us) < nk ? (long long)(us) * 128 : (long long)((us) - nk) * 128 + ((s) < 2 ? dW : dA))

        f32x4 bias4[4];                                    // ahead of the K loop: see tile_epilogue_128x64
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bias4[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + i * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
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

        if (fresh) {
#pragma unroll
            for (int h = 0; h < 7; ++h) issue(h & 3, (long long)(h >> 2) * 128);      // nk >= 4
            WAIT_VM(6);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_b(f.b0x, c_slot, 0);
        if (wm == 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        T4P_STAMP(1);
        for (int u = 0; u < nk; u += 2) {
            T4P_KTILE(f.b0x, f.b0y)
            ++u;
            T4P_KTILE(f.b0y, f.b0x)
            --u;
        }
        T4P_STAMP(2);
        if (wm == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        T4P_STAMP(3);
        // epilogues that LOAD (fp32 residual, gelu'): the seam's 14 LDS-DMAs, issued during the last two K-tiles, are waited for here:
        // with one of them outstanding hipcc waits vmcnt(0)-like for every plain load of the epilogue (it does not count across a
        // DMA), which serialised the residual prefetch against its own stores -- 45k of an 80k-cycle proj tile (tools/gemm_stamps.py)
        if (EPI == EPI_F32_RESID || EPI == EPI_BF16_DGELU) WAIT_VM(0);
        {
            const int sl = slot_of(c_slot, 7 + wave / 3);      // the three slots no staged half-tile occupies
            tile_epilogue_128x64<EPI, true, false>(p, acc, lds + sl * T4_HT + (wave % 3) * 4608, tm, m0, n0, wm, wn, wave, lane, bias4);
        }
        __builtin_amdgcn_sched_barrier(0);
        T4P_STAMP(4);
        if (seam) {
            WAIT_VM(6);                                    // file header: the first four half-tiles have landed under either ordering model
        } else {
            WAIT_VM(0);                                    // the placeholder half-tiles have landed: their slots can be staged again
            st_slot = c_slot;
        }
        __builtin_amdgcn_sched_barrier(0);
        T4P_STAMP(5);
        fresh = !seam;
#undef T4P_STAGE
    }
#undef T4P_KTILE
#undef T4P_MFMA_SECTION
    __builtin_amdgcn_s_waitcnt(0x0070);                    // placeholder half-tiles of the last seam: no DMA may outlive the workgroup's LDS
}
