// gemm_tn v3: C[M,N] = A[M,K] . W[N,K]^T, block tile 256 x 128 (4 waves, two workgroups per CU) or 256 x 256 (8 waves,
// one per CU), every wave owning 128 x 64, K-step 32, a 3-deep LDS-DMA ring with COUNTED vmcnt and a raw s_barrier.
// Since gemm_tn4.h (staggered 4-phase schedule) this is the fallback for shapes that kernel does not take, and the home
// of the epilogue both share (tile_epilogue_128x64 below).
//
// Why this shape (gfx950): the per-wave tile sets the LDS read bytes per MFMA (12 KiB of ds_read_b128 per
// 32 MFMAs = half of a 64x64 wave tile); the block tile keeps L2->LDS traffic at ~47 B/clk/CU at full MFMA rate.
//
// Pipeline invariant at the top of iteration t: fragments of K-step t are in registers, K-step t+1 has
// landed in LDS and is barrier-published, K-step t+2 is in flight, ring slot t%3 is free.
//     1) LDS-DMA K-step t+3 -> slot t%3
//     2) ds_read fragments of K-step t+1 (other register set)
//     3) 32 MFMAs on K-step t
//     4) s_waitcnt vmcnt(PPW)  (everything but the PPW loads issued in 1) -> K-step t+2 landed);
//        s_waitcnt lgkmcnt(0); s_barrier
// Every load has two full iterations of MFMA time to land; the barrier never drains the load queue.  The ILV variant
// hand-pins the issue order {1 LDS-DMA, 4 MFMA, 3 fragment reads, 4 MFMA} x 4 inside a step.
//
// LDS image per slot: [256 A rows | BN W rows] x 64 B (32 bf16), 16-B chunk' = chunk ^ (((row>>2)&1)<<1):
// conflict-free for the ds_read_b128 fragment reads (tools/lds_model.py).  LDS-DMA writes lane-linear, so
// the swizzle is applied to the per-lane SOURCE address.
#pragma once
#include "gemm_tn.h"

#define T3_BM 256
#define T3_BN 128                                // narrow variant (2 workgroups per CU); the wide one is 256
#define T3_THREADS 256
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8]); the BUILTIN keeps
// hipcc's own wait-count bookkeeping in sync (it cannot see through inline asm)
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

struct T3Frags {
    bf16x8 a[8];
    bf16x8 w[4];
};

// =================================================================================================
// Epilogue shared by the 128 x 64-per-wave kernels (gemm_tn3_kernel, gemm_tn4_kernel).
// lane holds, for (i,j): row ml = wm*128 + j*16 + li, cols nl = wn*64 + i*16 + 4g .. +3
//
// Streamed per WAVE in 8 slabs of 16 rows (j), with no workgroup barrier: the old epilogue ran as workgroup-wide
// phases (all math, barrier, all stores, barrier ...) that every CU entered at the same moment, so HBM idled
// during the math phases and the VALU during the store phases (K = 64 runs of the fused epilogues moved their bytes
// at 2.8-3.8 TB/s, and the epilogue time simply added to the MFMA loop).  Here the stores of slab j drain while
// slab j+1 is computed, and the loads a slab needs (residual / gelu') are issued one slab ahead.
//   fp32 residual output: LDS_FREE (gemm_tn4_kernel): (acc + bias) goes through the wave-private slab as well and leaves as
//     whole 256-B row segments, 4 rows per store instruction, the residual(s) read the same way and added on the row side
//     (same association, bit-identical; round 2: proj 150.7 -> 140.0 us, fc2 302.2 -> 285.9 at M = 65,792, 275 -> 254 and
//     575 -> 548 at M = 131,328 against the accumulator-layout form, whose instructions touch 16 rows x 64 B each);
//     otherwise registers -> global directly.  bf16 outputs: a wave-private 16 x 64 LDS slab turns the 8-B accumulator
//     pieces into 16-B stores of whole 128-B row segments.
// The caller guarantees that no wave reads the operand ring any more (the slabs live at `lds`, 4.5 KiB per wave).
// =================================================================================================
// ---- fp32-residual epilogue (round 4): a ring of D slabs of residual in flight ---------------------------------------------------
// tools/gemm_stamps.py (s_memtime stamps per wave and tile, M = 65,792): the fp32-residual epilogue of the proj GEMM took 45k cycles of
// a tile's 80k (K loop: 30k) for 256 KB read + 256 KB written -- 11.6 B/clk, half the rate of the plain bf16 epilogue.  The ISA says
// why: the residual of slab j + 1 was asked for one slab ahead, but in the persistent kernel the LDS-DMAs of the seam are in the queue
// during the epilogue, and with a DMA outstanding hipcc does not count its own loads: it waited `vmcnt(3 / 2 / 1 / 0)` in front of slab
// j's additions, i.e. for the loads of slab j + 1 it had just issued -- a full memory round trip per slab, eight per tile.
// Two changes: (i) gemm_tn4p.h drains the seam's DMAs once in front of the epilogue (they were issued two K-tiles earlier: landed or
// nearly), after which hipcc counts exactly; (ii) D slabs of residual in flight instead of one.
// (A form with the loads in inline asm and hand-counted `s_waitcnt vmcnt(n_j)` was built first: wrong results -- at 256 VGPRs hipcc
// copies a ring register between the asm load and the asm wait, i.e. before the data has landed; guides' section 5.7 item 1.)
template <int NS, int D>      // NS residual streams: 0, 1 (x1 = xin + proj(..)) or 2 (x = x1 + fc2(..) + pos); D slabs of them in flight
__device__ __forceinline__ void epi_resid_rows(const GemmParams& p, f32x4 (&acc)[4][8], char* fslab, const f32x4 (&bias4)[4], int rowbase,
                                               int colbase, int lane) {
    constexpr int FS = 272, L = 4 * NS, S = 4;
    const int li = lane & 15, g = lane >> 4, rr = lane >> 4, cc = lane & 15;
    const float* q[2] = {NS >= 1 ? p.resid + (size_t)colbase + cc * 4 : nullptr, NS == 2 ? p.rowgrp + (size_t)colbase + cc * 4 : nullptr};
    float* outp = (float*)p.out + (size_t)colbase + cc * 4;
    f32x4 ring[NS ? D : 1][NS ? NS : 1][4];
    auto issue = [&](int j, int slot) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                int m = rowbase + j * 16 + h * 4 + rr;
                m = m < p.M ? m : p.M - 1;
                ring[slot][s][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q[s] + (size_t)m * p.N));      // read once
            }
    };
    if constexpr (NS > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, d);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(fslab + li * FS + i * 64 + g * 16) = acc[i][j] + bias4[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = rowbase + j * 16 + h * 4 + rr;
            f32x4 v = *reinterpret_cast<const f32x4*>(fslab + (h * 4 + rr) * FS + cc * 16);
            if constexpr (NS == 1) v = v + ring[j % D][0][h];
            if constexpr (NS == 2) v = v + (ring[j % D][0][h] + ring[j % D][1][h]);
            if (m < p.M) {
                // group = S > 1: token-stream output, row m of the compact [B*S] rows goes to row m + m/S + 1 ([B*(S+1)] rows, one cls
                // row in front of every sequence: src/models/pix4point.py:248-252 without the torch.cat); never with a residual
                const size_t orow = (NS == 0 && p.group > 1) ? (size_t)m + m / p.group + 1 : (size_t)m;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(outp + orow * p.N));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (NS > 0) {
            if (j + D < 8) issue(j + D, j % D);
        }
    }
}

template <int EPI, bool LDS_FREE = false, bool FAST_TWO_STREAMS = true>     // FAST_TWO_STREAMS: false in the persistent kernel (never launched with a second residual; the 64 ring registers spill there)
__device__ __forceinline__ void tile_epilogue_128x64(const GemmParams& p, f32x4 (&acc)[4][8], char* wslab, int tm, int m0,
                                                     int n0, int wm, int wn, int wave, int lane,
                                                     const f32x4* bias_pre = nullptr) {      // wslab: this wave's 4.5 KiB
    const int li = lane & 15, g = lane >> 4;
    const int rowbase = m0 + wm * 128, colbase = n0 + wn * 64;
    // bias_pre: the caller loaded this wave's 4 bias vectors ahead of the K loop (gemm_tn4p.h: loaded here they are four
    // load + vmcnt(0) pairs at the head of the epilogue, and under the persistent loop each vmcnt(0) also drains the seam's DMAs)
    f32x4 bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bias4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias_pre) bias4[i] = bias_pre[i];
        else if (p.bias) bias4[i] = *reinterpret_cast<const f32x4*>(p.bias + colbase + i * 16 + g * 4);
    }

    if (EPI == EPI_F32_RESID) {
        const float* rsrc = p.resid ? p.resid + (size_t)colbase + g * 4 : nullptr;
        // optional SECOND residual (passed in the rowgrp slot): x_next = x1 + fc2(.) + pos, so that the next block's
        // LayerNorm reads one stream instead of x and pos and does not write their sum (src/models/pix4point.py:255)
        const float* rsrc2 = p.rowgrp ? p.rowgrp + (size_t)colbase + g * 4 : nullptr;
        float* dst = (float*)p.out + (size_t)colbase + g * 4;
        auto row_of = [&](int j) { int m = rowbase + j * 16 + li; return m < p.M ? m : p.M - 1; };
        if constexpr (LDS_FREE) {
            // rows through the wave's LDS slab: every global access (residual reads, output stores) is 4 whole 256-B row segments per
            // instruction instead of 16 rows x 64 B; the sums keep their association ((acc + bias) + (resid + resid2)); residual loads
            // hand-counted (epi_resid_rows).  p.epi_flags bit 0 (A/B switch) has no effect here any more: the round-3 form is gone.
            if (!p.resid) epi_resid_rows<0, 1>(p, acc, wslab, bias4, rowbase, colbase, lane);
            else if (!p.rowgrp) epi_resid_rows<1, FAST_TWO_STREAMS ? 3 : 2>(p, acc, wslab, bias4, rowbase, colbase, lane);
            else epi_resid_rows<2, 2>(p, acc, wslab, bias4, rowbase, colbase, lane);
            return;
        }
        f32x4 rnext[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rnext[i] = rsrc ? *reinterpret_cast<const f32x4*>(rsrc + (size_t)row_of(0) * p.N + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (rsrc2) rnext[i] += *reinterpret_cast<const f32x4*>(rsrc2 + (size_t)row_of(0) * p.N + i * 16);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 rcur[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rcur[i] = rnext[i];
            if (j + 1 < 8 && rsrc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rnext[i] = *reinterpret_cast<const f32x4*>(rsrc + (size_t)row_of(j + 1) * p.N + i * 16);
                    if (rsrc2) rnext[i] += *reinterpret_cast<const f32x4*>(rsrc2 + (size_t)row_of(j + 1) * p.N + i * 16);
                }
            }
            const int m = rowbase + j * 16 + li;
            if (m < p.M) {
                // group = S > 1: token-stream output, row m of the compact [B*S] rows goes to row m + m/S + 1 ([B*(S+1)]
                // rows, one cls row in front of every sequence: src/models/pix4point.py:248-252 without the torch.cat)
                const size_t orow = p.group > 1 ? (size_t)m + m / p.group + 1 : (size_t)m;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(dst + orow * p.N + i * 16) = (acc[i][j] + bias4[i]) + rcur[i];
            }
        }
        return;
    }

    // ---- bf16 outputs
    constexpr int SLAB_STRIDE = 144;                       // 128-B rows + 16: conflict-free 8-B writes, 16-B aligned reads
    constexpr int SLAB_BYTES = 16 * SLAB_STRIDE;
    char* slab = wslab;                                    // [out | pre] per wave
    const int wr_off = li * SLAB_STRIDE + g * 8;           // + i*32
    // the wave's own ds_writes and ds_reads execute in order; the fences only stop hipcc from moving them
    auto wave_lds_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // outputs are written once and are far larger than the caches: non-temporal stores (round 2, same-box A/B: fc1 +7 %, qkv +3 %,
    // fc2 / dfc2 +1-2 %; the whole step -0.8 ms at N=1024, -0.75 ms at N=2048)
    auto flush = [&](const char* sl, __bf16* dstbase, int j) {     // 16 rows x 128 B -> global, 16 B per lane
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = h2 * 8 + (lane >> 3), c = lane & 7;
            const int m = rowbase + j * 16 + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(sl + r * SLAB_STRIDE + c * 16);
            if (m < p.M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dstbase + (size_t)m * p.N + colbase + c * 8));
        }
    };
    const bool want_csum = EPI == EPI_BF16_DGELU && p.stats != nullptr;
    constexpr bool STATS = EPI == EPI_BF16_ROWGROUP || EPI == EPI_BF16_STATS;   // column sums + sums of squares (BatchNorm)
    f32x4 csum_all[4], csq_all[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) csum_all[i] = csq_all[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dGELU: gelu'(h) of the next slab is fetched one slab ahead.  LDS_FREE: as whole 128-B row segments (8 rows per load
    // instruction) that reach the accumulator layout through the idle `pre` half of the slab; otherwise straight in the
    // accumulator layout (16 rows x 32 B per instruction)
    constexpr bool HROWS = LDS_FREE && EPI == EPI_BF16_DGELU;
    bf16x4 hnext[4];
    constexpr int HD_ = 3;                                  // slabs of gelu' in flight (HROWS; round 4: one was a memory round trip per slab)
    u32x4 hring[HD_][2];
    const __bf16* hsrc = EPI == EPI_BF16_DGELU ? p.pre + (size_t)colbase + g * 4 : nullptr;
    const __bf16* hsrc_r = EPI == EPI_BF16_DGELU ? p.pre + (size_t)colbase + (lane & 7) * 8 : nullptr;
    auto hrow = [&](int j) { int m = rowbase + j * 16 + li; return m < p.M ? m : p.M - 1; };
    auto hfetch = [&](int j) {
        if constexpr (HROWS) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                int m = rowbase + j * 16 + h2 * 8 + (lane >> 3);
                m = m < p.M ? m : p.M - 1;
                hring[j % HD_][h2] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(hsrc_r + (size_t)m * p.N));   // read once (proj +3-7 %, dfc2 +1-3 % with the residual / derivative loads non-temporal)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) hnext[i] = *reinterpret_cast<const bf16x4*>(hsrc + (size_t)hrow(j) * p.N + i * 16);
        }
    };
    if (EPI == EPI_BF16_DGELU) {
        hfetch(0);
        if constexpr (HROWS) {
#pragma unroll
            for (int d = 1; d < HD_; ++d) hfetch(d);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bf16x4 hcur[4];
        if (EPI == EPI_BF16_DGELU) {
            if constexpr (HROWS) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    *reinterpret_cast<u32x4*>(slab + SLAB_BYTES + (h2 * 8 + (lane >> 3)) * SLAB_STRIDE + (lane & 7) * 16) = hring[j % HD_][h2];
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i) hcur[i] = *reinterpret_cast<const bf16x4*>(slab + SLAB_BYTES + wr_off + i * 32);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) hcur[i] = hnext[i];
            }
            if constexpr (HROWS) {
                if (j + HD_ < 8) hfetch(j + HD_);          // (slot j % HD_ has just been consumed)
            } else {
                if (j + 1 < 8) hfetch(j + 1);
            }
        }
        const bool row_ok = rowbase + j * 16 + li < p.M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = acc[i][j] + bias4[i];
            if (EPI == EPI_BF16_GELU) {
                // y = gelu(h) and gelu'(h) in one pass (`pre` receives the derivative: the backward epilogue only multiplies)
                f32x4 dy;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float yy, dd;
                    gelu_and_dgelu(v[r], yy, dd);
                    v[r] = yy;
                    dy[r] = dd;
                }
                if (p.pre) {
                    bf16x4 w2 = {f2bf(dy[0]), f2bf(dy[1]), f2bf(dy[2]), f2bf(dy[3])};
                    *reinterpret_cast<bf16x4*>(slab + SLAB_BYTES + wr_off + i * 32) = w2;
                }
            }
            if (EPI == EPI_BF16_DGELU)
                v = f32x4{v[0] * bf2f(hcur[i][0]), v[1] * bf2f(hcur[i][1]), v[2] * bf2f(hcur[i][2]), v[3] * bf2f(hcur[i][3])};
            bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *reinterpret_cast<bf16x4*>(slab + wr_off + i * 32) = w;
            if (EPI == EPI_BF16_DGELU && want_csum && row_ok)
                csum_all[i] += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
            if (STATS && row_ok) {                          // fp32 values before rounding, as in gemm_tn2_kernel
                csum_all[i] += v;
                csq_all[i] += v * v;
            }
        }
        wave_lds_sync();
        if (EPI == EPI_BF16_GELU && p.pre) flush(slab + SLAB_BYTES, p.pre, j);
        flush(slab, (__bf16*)p.out, j);
        wave_lds_sync();
    }
    // dGELU with p.stats: column sums of the (rounded) output over this wave's 128 rows -> stats[2*tm + wm][n].  Their
    // column sum is the fc1 bias gradient (autograd of nn.Linear.bias): no separate pass re-reading the 400-MB dh.
    // EPI_BF16_ROWGROUP / EPI_BF16_STATS: stats[ceil(M/128)][2][N], one row pair per 128-row block = per wave row
    if (STATS && 2 * tm + wm < (p.M + 127) / 128) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 cs = csum_all[i], cq = csq_all[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = cs[r], b = cq[r];
                a += dpp_f32<0xB1>(a); a += dpp_f32<0x4E>(a); a += dpp_f32<0x141>(a); a += dpp_f32<0x140>(a);
                b += dpp_f32<0xB1>(b); b += dpp_f32<0x4E>(b); b += dpp_f32<0x141>(b); b += dpp_f32<0x140>(b);
                cs[r] = a;
                cq[r] = b;
            }
            if (li == 0) {
                float* row = p.stats + (size_t)(2 * tm + wm) * 2 * p.N + colbase + i * 16 + g * 4;
                *reinterpret_cast<f32x4*>(row) = cs;
                *reinterpret_cast<f32x4*>(row + p.N) = cq;
            }
        }
    }
    if (EPI == EPI_BF16_DGELU && want_csum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 cs = csum_all[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // sum over the 16 rows li of this 16-lane group
                float v = cs[r];
                v += dpp_f32<0xB1>(v);
                v += dpp_f32<0x4E>(v);
                v += dpp_f32<0x141>(v);
                v += dpp_f32<0x140>(v);
                cs[r] = v;
            }
            if (li == 0) *reinterpret_cast<f32x4*>(p.stats + (size_t)(tm * 2 + wm) * p.N + colbase + i * 16 + g * 4) = cs;
        }
    }
}

template <int EPI, bool PAIR, int BN_, bool ILV = false>
__global__ __launch_bounds__(BN_ * 2, 2) void gemm_tn3_kernel(GemmParams p) {
    constexpr int NWN = BN_ / 64, NW = 2 * NWN, NTHR = NW * 64;      // waves: 2 (M) x NWN (N), each 128 x 64
    constexpr int T3_STAGE = (T3_BM + BN_) * 64;                      // 24 / 32 KiB per K-step of 32
    constexpr int T3_PPW = ((T3_BM + BN_) / 16) / NW;                 // 1-KiB LDS-DMA pieces per wave: 6 / 4
    constexpr int T3_LDS = 3 * T3_STAGE;                              // the epilogue's per-wave slabs reuse the ring
    __shared__ __attribute__((aligned(1024))) char lds[T3_LDS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int tiles_n = p.N / BN_, tiles_m = (p.M + T3_BM - 1) / T3_BM;
    int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    if (PAIR) {
        // within every run of 64 tiles of one XCD: dispatch slot c and c+32 (observed to share a CU: an XCD
        // has 32 CUs and takes 2 of these workgroups each) get neighbouring tiles 2c, 2c+1 = same A panel
        const int nwg = tiles_m * tiles_n, per = nwg >> 3;
        const int xcd_base = tile - ((blockIdx.x >> 3));
        const int loc = blockIdx.x >> 3;
        if (((loc | 63) < per)) tile = xcd_base + (loc & ~63) + (((loc & 31) << 1) | ((loc >> 5) & 1));
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * T3_BM, n0 = tn * BN_;
    const int nk = p.K / 32;

    // ---- staging: piece = 16 rows x 64 B; pieces [0,16) are A rows, [16,24) are W rows
    const int rin = lane >> 2;
    const int chunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
    const __bf16* src[T3_PPW];
#pragma unroll
    for (int q = 0; q < T3_PPW; ++q) {
        const int piece = wave * T3_PPW + q;
        const bool isA = piece < T3_BM / 16;
        int r = isA ? m0 + piece * 16 + rin : n0 + (piece - T3_BM / 16) * 16 + rin;
        const int lim = isA ? p.M : p.N;
        r = r < lim ? r : lim - 1;
        src[q] = (isA ? p.A : p.W) + (size_t)r * p.K + chunk * 8;
    }
    auto stage = [&](int kt, char* buf) {
#pragma unroll
        for (int q = 0; q < T3_PPW; ++q)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[q] + kt * 32), LDS_PTR(buf + (wave * T3_PPW + q) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][8];     // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int li = lane & 15, g = lane >> 4;
    const int frag_off = li * 64 + ((g ^ (((li >> 2) & 1) << 1)) << 4);
    const int a_off = wm * 128 * 64 + frag_off;
    const int w_off = T3_BM * 64 + wn * 64 * 64 + frag_off;
    auto read_frags = [&](const char* buf, T3Frags& f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(buf + a_off + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(buf + w_off + i * 1024);
    };
    auto mma = [&](const T3Frags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[i], f.a[j], acc[i][j], 0, 0, 0);
    };

    char* s0 = lds;
    char* s1 = lds + T3_STAGE;
    char* s2 = lds + 2 * T3_STAGE;
    T3Frags f0, f1;

    // ---- prologue: three K-steps in flight, publish K-step 0 and 1
    stage(0, s0);
    if (nk > 1) stage(1, s1);
    if (nk > 2) stage(2, s2);
    if (nk > 2) WAIT_VM(2 * T3_PPW);
    else if (nk > 1) WAIT_VM(T3_PPW);
    else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    read_frags(s0, f0);
    if (nk > 2) WAIT_VM(T3_PPW);
    else WAIT_VM(0);
    WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    // ring slots: cur holds K-step kt (already in registers: free), nxt holds kt+1, nn holds kt+2
    char *cur = s0, *nxt = s1, *nn = s2;
#define T3_ITER(FC, FN)                                                                   \
    {                                                                                     \
        const bool more3 = kt + 3 < nk;                                                   \
        if (more3) stage(kt + 3, cur);                                                    \
        if (kt + 1 < nk) read_frags(nxt, FN);                                             \
        mma(FC);                                                                          \
        if (kt + 1 < nk) {                                                                \
            if (more3) WAIT_VM(T3_PPW);      \
            else WAIT_VM(0);                         \
            WAIT_LGKM0();                            \
            __builtin_amdgcn_s_barrier();                                                 \
        }                                                                                 \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
    // steady state (K-steps kt, kt+1 with kt+4 < nk): no conditionals, so hipcc keeps the issue order
    // {LDS-DMA, fragment prefetch, 32 MFMAs, counted wait, barrier} without conservative waits
#define T3_FULL(KT, FC, FN)                                                               \
    {                                                                                     \
        if (!ILV) {                                                                       \
            stage((KT) + 3, cur);                                                         \
            read_frags(nxt, FN);                                                          \
            __builtin_amdgcn_sched_barrier(0);   /* keep the prefetch AHEAD of the MFMAs */ \
            mma(FC);                                                                      \
        } else {                                                                          \
            /* hand-pinned issue order (BN_=256: 4 groups of {1 LDS-DMA, 4 MFMA, 3 fragment reads, 4 MFMA}): an     \
               LDS-DMA issue holds the wave 60-180 cycles, so MFMAs are always queued in the pipe behind it */     \
            _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                            \
                if (qq < T3_PPW)                                                          \
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src[qq] + ((KT) + 3) * 32),  \
                                                     LDS_PTR(cur + (wave * T3_PPW + qq) * 1024), 16, 0, 0); \
                if (T3_PPW > 4 && qq < T3_PPW - 4)                                        \
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src[qq + 4] + ((KT) + 3) * 32), \
                                                     LDS_PTR(cur + (wave * T3_PPW + qq + 4) * 1024), 16, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                             \
                    acc[qq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FC.w[qq], FC.a[j], acc[qq][j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                        \
                FN.a[2 * qq] = *reinterpret_cast<const bf16x8*>(nxt + a_off + (2 * qq) * 1024);     \
                FN.a[2 * qq + 1] = *reinterpret_cast<const bf16x8*>(nxt + a_off + (2 * qq + 1) * 1024); \
                FN.w[qq] = *reinterpret_cast<const bf16x8*>(nxt + w_off + qq * 1024);    \
                __builtin_amdgcn_sched_barrier(0);                                        \
                _Pragma("unroll") for (int j = 4; j < 8; ++j)                             \
                    acc[qq][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FC.w[qq], FC.a[j], acc[qq][j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                        \
            }                                                                             \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
        WAIT_VM(T3_PPW);                                                                  \
        WAIT_LGKM0();                                                                     \
        __builtin_amdgcn_s_barrier();                                                     \
        char* tmp = cur; cur = nxt; nxt = nn; nn = tmp;                                   \
    }
    int kt = 0;
    for (; kt + 4 < nk; kt += 2) {
        T3_FULL(kt, f0, f1)
        T3_FULL(kt + 1, f1, f0)
    }
#undef T3_FULL
    for (; kt < nk; kt += 2) {                 // tail: at most 4 K-steps, f0 holds an even K-step
        T3_ITER(f0, f1)
        ++kt;
        if (kt < nk) T3_ITER(f1, f0)
        --kt;
    }
#undef T3_ITER

    __builtin_amdgcn_sched_barrier(0);
    tile_epilogue_128x64<EPI>(p, acc, lds + wave * 4608, tm, m0, n0, wm, wn, wave, lane);
}
