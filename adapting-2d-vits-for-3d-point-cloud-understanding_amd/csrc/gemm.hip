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

__device__ float g_zero_page[64];   // 256 B of zeros: LDS-DMA source for rows past the end (wgrad)

#include "gemm_tn.h"
#include "gemm_tn3.h"
#include "gemm_tn4.h"
#include "gemm_tn4p.h"
#include "gemm_wgrad3.h"
#include "gemm_wgrad4.h"
#include "gemm_wgrad5.h"

#define GEMM_THREADS 256

// tile configuration: 0 = auto, 1 = 128x128 (4 waves), 2 = 256x128 (8 waves), 3 = 256x256 (8 waves),
// 4 = 256x128 / 4 waves / K-step 32 / 3-stage ring (gemm_tn3.h); 11-13 = 1-3 with pinned scheduling
static int g_gemm_variant = 0;
static int g_tn4_persistent = 1;            // A/B switch: 0 = one workgroup per tile (gemm_tn4.h) everywhere; bit 1: no banded tile order
extern "C" int p4p_set_gemm_persistent(int v) { int o = g_tn4_persistent; g_tn4_persistent = v; return o; }
extern "C" int p4p_set_gemm_variant(int v) { int o = g_gemm_variant; g_gemm_variant = v; return o; }
static int g_epi_flags = 0;                 // A/B switch: bit 0 = epilogue loads as hipcc counts them (gemm_tn3.h: epi_resid_full off)
extern "C" int p4p_set_gemm_epilogue(int v) { int o = g_epi_flags; g_epi_flags = v; return o; }
static int g_peel_wide_rounds = 4;          // up to how many rounds of big tiles a WIDE output (> 4 tiles per row panel) has its tail row panel peeled.  Round 4 tried 32
                                            // (tools/gemm_peel_ab.py, M = 65,792, isolated: qkv 213.6 -> 207.0 us, fc1 379.6 -> 362.6, dfc2 326.3 -> 317.4, bit-identical)
                                            // and the whole step said no (tools/peel_step_ab.py, same box, interleaved: 50.43-50.49 ms whole, 50.64-50.75 peeled): 4 stays
extern "C" int p4p_set_gemm_peel_rounds(int v) { int o = g_peel_wide_rounds; g_peel_wide_rounds = v; return o; }

// ---- split-K for the short, wide tail launches (few workgroups, long K loop: latency-bound at one K-step per ~0.5 us)
// out[m][n] = sum_s part[s][m][n] + bias[n] (+ resid[m][n]), fixed order; fp32 or bf16 output
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int splits, long long plane,
                                                            const float* __restrict__ bias, const float* __restrict__ resid,
                                                            const float* __restrict__ resid2, void* __restrict__ out, int out_f32,
                                                            int N) {
    const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= plane) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + e);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(part + (size_t)s * plane + e);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + (int)(e % N));
    if (resid) v += *reinterpret_cast<const f32x4*>(resid + e);
    if (resid2) v += *reinterpret_cast<const f32x4*>(resid2 + e);
    if (out_f32) {
        *reinterpret_cast<f32x4*>((float*)out + e) = v;
    } else {
        bf16x4 w = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>((__bf16*)out + e) = w;
    }
}
// library-owned scratch for the split-K planes of the peeled tail launch (a few MB), one buffer PER DEVICE, grown on demand.
// Threading rule of this library (include/p4p_hip.h): one host thread at a time and one stream per device -- two launches of the
// same device on different streams would share this buffer (and the host-side switches above) without synchronisation.
static float* g_splitk_ws[64] = {nullptr};
static size_t g_splitk_ws_floats[64] = {0};
static int splitk_workspace(size_t need, float** out) {
    int dev = 0;
    P4P_HIP(hipGetDevice(&dev));
    P4P_CHECK(dev >= 0 && dev < 64, "p4p_gemm_tn_bf16: device index %d", dev);
    if (need > g_splitk_ws_floats[dev]) {
        if (g_splitk_ws[dev]) P4P_HIP(hipFree(g_splitk_ws[dev]));
        g_splitk_ws[dev] = nullptr;
        g_splitk_ws_floats[dev] = 0;
        P4P_HIP(hipMalloc(&g_splitk_ws[dev], need * sizeof(float)));
        g_splitk_ws_floats[dev] = need;
    }
    *out = g_splitk_ws[dev];
    return 0;
}

template <int BM_, int BN_, int WAVES_M, int WAVES_N, int EPI, bool PIN = false, int ABL = 0>
static int launch_cfg(const GemmParams& p, hipStream_t st) {
    int tiles = ceil_div(p.M, BM_) * ceil_div(p.N, BN_);
    hipLaunchKernelGGL((gemm_tn2_kernel<BM_, BN_, WAVES_M, WAVES_N, EPI, PIN, ABL>), dim3(tiles), dim3(WAVES_M * WAVES_N * 64), 0, st, p);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_colsum_partial_bf16(const void* G, float* partial, int M, int N, void* stream);

// EPI_BF16_DGELU with stats: [p4p_gemm_tn_colsum_rows(M)][N] partial column sums of the output (zero rows allowed)
extern "C" int p4p_gemm_tn_colsum_rows(int M) { return 2 * ceil_div(M, 256); }

template <int EPI>
static int launch_tn_inner(const GemmParams& p, hipStream_t st, int variant);

template <int EPI>
static int launch_tn(const GemmParams& p, hipStream_t st) {
    if constexpr (EPI == EPI_BF16_DGELU) {
        if (p.stats) P4P_HIP(hipMemsetAsync(p.stats, 0, (size_t)p4p_gemm_tn_colsum_rows(p.M) * p.N * sizeof(float), st));
    }
    return launch_tn_inner<EPI>(p, st, g_gemm_variant);
}

// variant: the tile configuration (0 = choose); the peeled launches below pass theirs down as an argument (the global switch is
// only READ, by launch_tn)
template <int EPI>
static int launch_tn_inner(const GemmParams& p, hipStream_t st, int variant) {
    constexpr bool STATS = EPI == EPI_BF16_ROWGROUP || EPI == EPI_BF16_STATS;
    int v = variant;
    const bool fits_tn4 = p.N % 256 == 0 && (long long)ceil_div(p.M, 256) * (p.N / 256) >= 256 &&
                          (unsigned long long)p.M * p.K * 2 < (1ull << 32) && (unsigned long long)p.N * p.K * 2 < (1ull << 32);
    if (STATS) {
        // statistics rows are per 128-row block (p4p_gemm_tn_stats_rows): the 128x128 kernel, or the 256x256 staggered
        // kernel whose two wave rows own one block each
        if (v == 0 || v == 10) v = fits_tn4 ? 10 : 1;
        else v = 1;
    }
    if (v == 0) {
        // measured on MI355X at the ViT-B shapes (tools/gemm_bench.py, random data):
        //   plain bf16 epilogue  : 256x256 tile wins (940-1000 TFLOP/s vs 830 / 700)
        //   GELU / fp32 residual / dGELU epilogues: the 2-workgroups-per-CU 256x128 ring overlaps the
        //   HBM / VALU-heavy epilogue with the other workgroup's MFMA loop
        const bool big = (long long)ceil_div(p.M, 256) * ceil_div(p.N, 128) >= 512;
        const bool big256 = p.N % 256 == 0 && (long long)ceil_div(p.M, 256) * (p.N / 256) >= 256;
        // 256x256 tile everywhere it fits.  Measured (tools/gemm_bench.py, TFLOP/s, ViT-B shapes): the K-step-32 ring
        // with the hand-interleaved issue order {LDS-DMA, 4 MFMA, 3 reads, 4 MFMA} (8) wins for every fused epilogue
        // (proj 438 vs 380, fc2 852 vs 782, dfc2 691 vs 671, fc1 586 vs 569) and for long K (dfc1 1052 vs 1018);
        // the mid-tile-barrier schedule (3) keeps the plain bf16 epilogue at K = 768 (qkv 911 vs 884)
        // (with the per-wave streamed epilogue of gemm_tn3.h the ring kernel also wins the plain bf16 epilogue at
        //  K = 768: qkv 921 vs 882 for the mid-tile-barrier kernel, whose epilogue still runs as workgroup phases)
        //  The staggered 4-phase schedule of gemm_tn4.h (10) beats the ring kernel on every shape: qkv 1023 vs 883,
        //  proj 512 vs 470, fc1 773 vs 695, fc2 968 vs 871, dfc2 840 vs 755, dfc1 1178 vs 1029.)
        if (big256) v = 10;
        else if (p.N % T3_BN == 0 && big) v = 4;
        else v = 1;
    }
    if constexpr (STATS) {
        if (v == 10) {
            hipLaunchKernelGGL((gemm_tn4_kernel<EPI>), dim3(ceil_div(p.M, 256) * (p.N / 256)), dim3(512), 0, st, p);
            P4P_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (!STATS) {
        // Grid quantisation: with few rounds of big tiles (M = 256*257 and N = 768 give 771 tiles of 256x256 on
        // 256 CUs = 3 rounds + 3 tiles -> a 4th round for 0.4% of the work, measured -15..20%), peel the last
        // row tile(s) off and run them with the 128x128 configuration (many small workgroups, one short round).
        const bool row_remap = EPI == EPI_F32_RESID && p.group > 1;      // token-stream output rows: no peeling / split-K
        if ((v == 3 || v == 4 || v == 5 || v == 7 || v == 8 || v == 10) && !row_remap) {
            // slots = workgroups resident at once: 256 CUs (two 256 x 128 workgroups each), or what p4p_set_reserved_cus() leaves
            // of them for the staggered / persistent 256 x 256 kernels (capi.cpp; 256 at world size 1)
            const int pgrid = v == 10 ? p4p_persistent_grid() : 0;
            const int bn = (v == 3 || v == 7 || v == 8 || v == 10) ? 256 : T3_BN, slots = (v == 3 || v == 7 || v == 8 || v == 10) ? (pgrid > 0 ? pgrid : 256) : 512;
            const int tiles_n = ceil_div(p.N, bn), tiles_m = ceil_div(p.M, 256);       // the last panel may be ragged: it is peeled first
            const long long total = (long long)tiles_m * tiles_n;
            const long long rounds = total / slots, rem = total % slots;
            const int peel_max_rounds = 12;     // narrow N (<= 4 tiles per panel): measured at N=2048 (6 rounds + 3 tiles)
            if (rounds >= 1 && rounds <= (tiles_n <= 4 ? peel_max_rounds : g_peel_wide_rounds) && rem > 0 && rem * 8 <= slots && rem % tiles_n == 0 && rem / tiles_n < tiles_m) {
                GemmParams a = p, b = p;
                a.M = (tiles_m - (int)(rem / tiles_n)) * 256;           // whole panels for the big-tile launch
                const int peel = p.M - a.M;                             // rows handled by the small-tile launch
                const size_t ro = (size_t)a.M;
                b.M = peel;
                b.A = p.A + ro * p.K;
                b.out = (char*)p.out + ro * p.N * (EPI == EPI_F32_RESID ? 4 : 2);
                if (p.resid) b.resid = p.resid + ro * p.N;
                if (EPI == EPI_F32_RESID && p.rowgrp) b.rowgrp = p.rowgrp + ro * p.N;          // second residual
                if (p.pre) b.pre = p.pre + ro * p.N;
                if (EPI == EPI_BF16_DGELU && p.stats) b.stats = p.stats + (size_t)2 * (a.M / 256) * p.N;   // rows behind the main part's
                int rc = launch_tn_inner<EPI>(a, st, v);               // main part: no further peeling (rem == 0)
                if (!rc) {                                              // tail: 64x64 tiles (variant 6), 4x more workgroups for the short launch
                    int splits = 1;
                    if constexpr (EPI == EPI_BF16 || EPI == EPI_F32_RESID) {
                        if (p.K >= 1536 && p.N % 4 == 0) {    // long K: slices of 768 (or 512) columns
                            splits = p.K % 768 == 0 ? p.K / 768 : (p.K % 512 == 0 ? p.K / 512 : 1);
                            if (splits > 8) splits = 1;
                        }
                    }
                    if (splits > 1) {
                        const size_t plane = (size_t)b.M * b.N, need = plane * splits;
                        float* ws = nullptr;
                        if (splitk_workspace(need, &ws)) return 2;
                        GemmParams c = b;
                        c.K = p.K / splits;
                        c.out = ws;
                        c.bias = nullptr;
                        c.resid = nullptr;
                        c.rowgrp = nullptr;
                        int tiles = ceil_div(c.M, 64) * ceil_div(c.N, 64);
                        hipLaunchKernelGGL((gemm_tn2_kernel<64, 64, 2, 2, EPI_F32_RESID, false, 0>), dim3(tiles, splits), dim3(256), 0, st, c);
                        P4P_LAUNCH_CHECK();
                        hipLaunchKernelGGL(splitk_finish_kernel, dim3(ceil_div((long long)plane / 4, 256)), dim3(256), 0, st,
                                           ws, splits, (long long)plane, b.bias, b.resid,
                                           EPI == EPI_F32_RESID ? b.rowgrp : nullptr, b.out,
                                           EPI == EPI_F32_RESID ? 1 : 0, b.N);
                        P4P_LAUNCH_CHECK();
                    } else {
                        rc = launch_tn_inner<EPI>(b, st, 6);
                    }
                }
                return rc;
            }
        }
        if ((v == 4 || v == 5) && p.N % T3_BN == 0) {
            int tiles = ceil_div(p.M, T3_BM) * (p.N / T3_BN);
            if (v == 5) hipLaunchKernelGGL((gemm_tn3_kernel<EPI, true, 128>), dim3(tiles), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_tn3_kernel<EPI, false, 128>), dim3(tiles), dim3(256), 0, st, p);
            P4P_LAUNCH_CHECK();
            return 0;
        }
        if (v == 10) {                                       // 256 x 256, staggered 4-phase schedule (gemm_tn4.h)
            if (p.N % 256 == 0 && (unsigned long long)p.M * p.K * 2 < (1ull << 32) && (unsigned long long)p.N * p.K * 2 < (1ull << 32)) {
                int tiles = ceil_div(p.M, 256) * (p.N / 256);
                // several tiles per CU and an even number (>= 4) of K-tiles: the persistent loop of gemm_tn4p.h hides the per-tile
                // dispatch + pipeline-fill latency behind the previous tile's epilogue
                const int pg = p4p_persistent_grid();       // CUs - reserved (capi.cpp); 0: a workgroup per tile
                if ((g_tn4_persistent & 1) && pg > 0 && tiles >= 2 * pg && p.K % 128 == 0 && p.K >= 256 && p.K <= ((g_tn4_persistent & 4) ? 4096 : 2304)) {      // bit 2 (experiment): long K too
                    GemmParams pb = p;
                    pb.band = (g_tn4_persistent & 2) ? 0 : 1;
                    // start-time shift of every other workgroup of an XCD (gemm_tn4p.h), in 1,024-cycle units: the epilogues that are
                    // HBM-bound when all CUs run them at once fall into the other half's K loops.  Same-box sweep (tools/gemm_shift_ab.py,
                    // bit-identical outputs; M = 65,792 / 131,328): fp32 residual 124.9 -> 119.8 / 229.1 -> 216.9 us at 16k, dGELU
                    // 325.4 -> 315.1 / 626 -> 617-620 at 8k, plain bf16 204.7 -> 199.4 / 397.5 -> 377.5 at 8k, GELU (VALU-bound) none
                    if (!(pb.epi_flags >> 8) && !(pb.epi_flags & 2))
                        pb.epi_flags |= (EPI == EPI_F32_RESID ? 16 : (EPI == EPI_BF16 || EPI == EPI_BF16_DGELU) ? 8 : 0) << 8;
                    hipLaunchKernelGGL((gemm_tn4p_kernel<EPI>), dim3(pg), dim3(512), 0, st, pb);
                    P4P_LAUNCH_CHECK();
                    return 0;
                }
                hipLaunchKernelGGL((gemm_tn4_kernel<EPI>), dim3(tiles), dim3(512), 0, st, p);
                P4P_LAUNCH_CHECK();
                return 0;
            }
            v = 8;
        }
        if ((v == 7 || v == 8) && p.N % 256 == 0) {          // 256 x 256, 8 waves, K-step 32 ring (1 workgroup per CU)
            int tiles = ceil_div(p.M, T3_BM) * (p.N / 256);
            if (v == 8) hipLaunchKernelGGL((gemm_tn3_kernel<EPI, false, 256, true>), dim3(tiles), dim3(512), 0, st, p);
            else hipLaunchKernelGGL((gemm_tn3_kernel<EPI, false, 256>), dim3(tiles), dim3(512), 0, st, p);
            P4P_LAUNCH_CHECK();
            return 0;
        }
        if (EPI == EPI_BF16_DGELU && p.stats) {
            // every other tile configuration: plain launch, then one streaming pass for the column sums
            GemmParams q = p;
            q.stats = nullptr;
            int rc = launch_tn_inner<EPI>(q, st, v);
            if (!rc) rc = p4p_colsum_partial_bf16(p.out, p.stats, p.M, p.N, st);
            return rc;
        }
        if (EPI == EPI_BF16) {     // timing-only ablations of the 256x256 loop (results are wrong by construction)
            if (v == 21) return launch_cfg<256, 256, 2, 4, EPI_BF16, false, 1>(p, st);
            if (v == 22) return launch_cfg<256, 256, 2, 4, EPI_BF16, false, 2>(p, st);
            if (v == 23) return launch_cfg<256, 256, 2, 4, EPI_BF16, false, 3>(p, st);
            if (v == 31) return launch_cfg<256, 256, 2, 4, EPI_BF16, false, 4>(p, st);
        }
        if (v == 6) return launch_cfg<64, 64, 2, 2, EPI>(p, st);
        if (v == 3) return launch_cfg<256, 256, 2, 4, EPI>(p, st);
        if (v == 2) return launch_cfg<256, 128, 4, 2, EPI>(p, st);
        if (v == 13) return launch_cfg<256, 256, 2, 4, EPI, true>(p, st);
        if (v == 12) return launch_cfg<256, 128, 4, 2, EPI, true>(p, st);
        if (v == 11) return launch_cfg<128, 128, 2, 2, EPI, true>(p, st);
    }
    return launch_cfg<128, 128, 2, 2, EPI>(p, st);
}

extern "C" int p4p_gemm_tn_bf16(int epi, const void* A, const void* W, void* out, const float* bias,
                                const float* resid, void* pre, const float* rowgrp, float* stats, int M, int N,
                                int K, int group, void* stream) {
    P4P_CHECK(A && W && out, "p4p_gemm_tn_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 8 && K >= 64, "p4p_gemm_tn_bf16: bad sizes M=%d N=%d K=%d", M, N, K);
    P4P_CHECK(K % 64 == 0, "p4p_gemm_tn_bf16: K=%d must be a multiple of 64", K);
    P4P_CHECK(N % 8 == 0, "p4p_gemm_tn_bf16: N=%d must be a multiple of 8", N);
    GemmParams p;
    p.A = (const __bf16*)A; p.W = (const __bf16*)W; p.out = out; p.bias = bias; p.resid = resid;
    p.pre = (__bf16*)pre; p.rowgrp = rowgrp; p.stats = stats; p.M = M; p.N = N; p.K = K; p.group = group; p.ldk = K; p.band = 0; p.epi_flags = g_epi_flags;
    hipStream_t st = (hipStream_t)stream;
    struct ProfScope {      // gemm_tn: brackets the (one or two, when the tail is peeled) launches of this call
        hipStream_t s; bool on;
        ProfScope(double fl, hipStream_t s_) : s(s_), on(p4p_prof_begin(2, fl, s_)) {}
        ~ProfScope() { if (on) p4p_prof_end(s); }
    } prof_scope(2.0 * M * N * K, st);
    switch (epi) {
        case EPI_BF16: return launch_tn<EPI_BF16>(p, st);
        case EPI_BF16_GELU: return launch_tn<EPI_BF16_GELU>(p, st);
        case EPI_F32_RESID:
            P4P_CHECK(group <= 1 || (!resid && !rowgrp), "p4p_gemm_tn_bf16: EPI_F32_RESID with group > 1 (token-stream rows) takes no residual");
            P4P_CHECK(!rowgrp || resid, "p4p_gemm_tn_bf16: EPI_F32_RESID: a second residual (rowgrp slot) needs the first");
            return launch_tn<EPI_F32_RESID>(p, st);
        case EPI_BF16_DGELU:
            P4P_CHECK(pre, "p4p_gemm_tn_bf16: EPI_BF16_DGELU needs pre");
            P4P_CHECK(!stats || N % 4 == 0, "p4p_gemm_tn_bf16: column sums need N %% 4 == 0");
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

extern "C" int p4p_gemm_tn_stats_rows(int M) { return ceil_div(M, 128); }

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
    float* bias_out;   // [splits][N] column sums of G (bias gradient) or null
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
    // bias gradient = G^T . 1: one extra MFMA per n-tile with an all-ones "A" fragment, only in the
    // workgroups of the first k-tile column and the waves with wk == 0 (each G column counted once)
    const bool do_bias = p.bias_out != nullptr && tk == 0 && wk == 0;
    f32x4 accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __bf16 one = (__bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};

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
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fg[j], accb[j], 0, 0, 0);
            }
        }
    }
    if (do_bias && g == 0) {       // D[i][n] is the same for every i: lanes 0..15, register 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + li;
            if (n < p.N) p.bias_out[(size_t)split * p.N + n] = accb[j][0];
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
    // four loads in flight per thread (one at a time the kernel sat at ~5 TB/s on latency x occupancy); the additions keep the order 0, 1, ..
    const float* q = slabs + e;
    f32x4 s = *reinterpret_cast<const f32x4*>(q);
    int k = 1;
    for (; k + 3 < splits; k += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(q + (size_t)k * n), a1 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 1) * n);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 2) * n), a3 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 3) * n);
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(q + (size_t)k * n);
    if (beta != 0.f) s += beta * *reinterpret_cast<const f32x4*>(out + e);
    *reinterpret_cast<f32x4*>(out + e) = s;
}

// (Round 4, measured and not kept: the slab sum with 8 floats per thread, every load of a thread issued up front, non-temporal loads:
// wgrad + sum 277.8 -> 291.4 us (fc1), 275.1 -> 294.4 (fc2), qkv / proj unchanged -- the slabs have just been written and are read back
// from L2 / Infinity Cache, which the streaming hint gives up.)
// out[e] = sum_s slabs[s][e] in the fixed order s = 0, 1, ..: the positional-embedding gradient as the sum of the L block-input
// gradients the encoder backward leaves in L slabs (functional._EncoderFn.backward), one pass instead of a read-modify-write per block
extern "C" int p4p_sum_slabs_f32(const float* slabs, float* out, long long n, int nslabs, void* stream) {
    P4P_CHECK(slabs && out && n >= 4 && n % 4 == 0 && nslabs >= 1, "p4p_sum_slabs_f32: bad arguments (n=%lld, nslabs=%d)", n, nslabs);
    P4P_CHECK(((reinterpret_cast<size_t>(slabs) | reinterpret_cast<size_t>(out)) & 15) == 0, "p4p_sum_slabs_f32: 16-byte alignment");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, slabs, out, n,
                       nslabs, 0.f);
    P4P_LAUNCH_CHECK();
    return 0;
}

// ---- several slab sums in ONE launch (round 5).  The weight-gradient GEMM of a ViT block leaves 64-66 MB of slabs per matrix and the
// sum above ran once per matrix: 20.7 us for 67 MB in + 7 MB out (3.6 TB/s: a launch this short never reaches the streaming rate),
// 54 launches = 1.1 ms per step.  The four matrices of a block are summed together at the end of the block's backward instead.
#define P4P_SLAB_SUM_MAX 8
struct SlabSumItem { const float* slabs; float* out; long long n; int nslabs; float beta; int block0; int pad; };
struct SlabSumBatch { SlabSumItem it[P4P_SLAB_SUM_MAX]; int count; };
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(SlabSumBatch b) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < P4P_SLAB_SUM_MAX; ++j)
        if (j < b.count && (int)blockIdx.x >= b.it[j].block0) i = j;
    const SlabSumItem& it = b.it[i];
    const long long n = it.n;
    long long e = ((long long)(blockIdx.x - it.block0) * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    const float* q = it.slabs + e;                       // the additions keep the order 0, 1, .. of reduce_slabs_kernel: bit-identical
    f32x4 s = *reinterpret_cast<const f32x4*>(q);
    const int splits = it.nslabs;
    int k = 1;
    for (; k + 3 < splits; k += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(q + (size_t)k * n), a1 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 1) * n);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 2) * n), a3 = *reinterpret_cast<const f32x4*>(q + (size_t)(k + 3) * n);
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(q + (size_t)k * n);
    if (it.beta != 0.f) s += it.beta * *reinterpret_cast<const f32x4*>(it.out + e);
    *reinterpret_cast<f32x4*>(it.out + e) = s;
}
// descs: HOST array of n records {const float* slabs; float* out; long long n; int nslabs; float beta} (32 bytes each, n <= 8)
struct P4PSlabSum { const float* slabs; float* out; long long n; int nslabs; float beta; };
extern "C" int p4p_sum_slabs_multi(const void* descs, int n, void* stream) {
    P4P_CHECK(descs && n >= 1 && n <= P4P_SLAB_SUM_MAX, "p4p_sum_slabs_multi: 1 .. %d records (n=%d)", P4P_SLAB_SUM_MAX, n);
    const P4PSlabSum* d = reinterpret_cast<const P4PSlabSum*>(descs);
    SlabSumBatch b = {};
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        P4P_CHECK(d[i].slabs && d[i].out && d[i].n >= 4 && d[i].n % 4 == 0 && d[i].nslabs >= 1, "p4p_sum_slabs_multi: record %d: bad arguments (n=%lld, nslabs=%d)", i, d[i].n, d[i].nslabs);
        P4P_CHECK(((reinterpret_cast<size_t>(d[i].slabs) | reinterpret_cast<size_t>(d[i].out)) & 15) == 0, "p4p_sum_slabs_multi: record %d: 16-byte alignment", i);
        b.it[i].slabs = d[i].slabs; b.it[i].out = d[i].out; b.it[i].n = d[i].n; b.it[i].nslabs = d[i].nslabs; b.it[i].beta = d[i].beta;
        b.it[i].block0 = (int)blocks;
        blocks += ceil_div(d[i].n / 4, 256);
        P4P_CHECK(blocks < (1ll << 31), "p4p_sum_slabs_multi: grid limit");
    }
    b.count = n;
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    P4P_LAUNCH_CHECK();
    return 0;
}

static bool wgrad_use_v3(int M, int N, int K) { return N % W3_BN == 0 && K % W3_BK == 0 && M >= 4096; }
static int g_wgrad_wide = 2;      // 0: 256x128 tiles, 1: 256x256, 2 (default): 256x256 with the hand-interleaved issue order, 3 (opt-in, measured slower: NOTES R5.6): 256x256 on the staggered 4-phase schedule (gemm_wgrad4.h) where K % 256 == 0, else as 2
extern "C" int p4p_set_wgrad_wide(int v) { int o = g_wgrad_wide; g_wgrad_wide = v; return o; }
static bool wgrad3_wide(int K) { return (g_wgrad_wide & 3) && K % 256 == 0; }      // 256x256 tile, 8 waves, 1 workgroup per CU
static bool wgrad_use_v4(int M, int N, int K) { return (g_wgrad_wide & 3) == 3 && wgrad_use_v3(M, N, K) && N % 256 == 0 && K % 256 == 0; }
static int wgrad3_steps_per_split(int M, int N, int K) {
    // one round of workgroups: as many M-splits as the resident slots allow -- the CUs p4p_set_reserved_cus() leaves (capi.cpp):
    // with r CUs held by another kernel a grid of 256 would run 256 - r workgroups and then r more, i.e. take twice as long
    const int pg = p4p_persistent_grid(), cus = pg > 0 ? pg : 256;
    const int bk = wgrad3_wide(K) ? 256 : W3_BK, slots = wgrad3_wide(K) ? cus : 2 * cus;
    int tiles = (N / W3_BN) * (K / bk);
    int steps = ceil_div(M, 32);
    int want = slots / tiles;                      // one round: never more workgroups than resident slots
    if (want < 1) want = 1;
    if (want > steps) want = steps;
    return ceil_div(steps, want);
}
static int wgrad4_tiles_per_split(int M, int N, int K) {      // the same rule in units of gemm_wgrad4.h's 64-row contraction tiles
    const int pg = p4p_persistent_grid(), cus = pg > 0 ? pg : 256;
    const int tiles = (N / 256) * (K / 256), steps = ceil_div(M, 64);
    int want = cus / tiles;
    if (want < 1) want = 1;
    if (want > steps) want = steps;
    return ceil_div(steps, want);
}
// M-splits (= slabs) of the ring kernels for this shape under the current switches
static int wgrad_ring_splits(int M, int N, int K) {
    if (wgrad_use_v4(M, N, K)) return ceil_div(ceil_div(M, 64), wgrad4_tiles_per_split(M, N, K));
    return ceil_div(ceil_div(M, 32), wgrad3_steps_per_split(M, N, K));
}

extern "C" int p4p_colsum_bf16(const void* G, float* out, float* workspace, int M, int N, float beta, void* stream);

extern "C" int p4p_wgrad_splits(int M, int N, int K) {
    // the ring kernel keeps the column-sum partials (<= 256 rows of N floats) behind its slabs: +2 slabs (K >= 128) cover them
    if (wgrad_use_v3(M, N, K)) return wgrad_ring_splits(M, N, K) + 2;
    int tiles = ceil_div(N, 128) * ceil_div(K, 128);
    int chunks = ceil_div(M, WG_BMR);
    int want = ceil_div(1024, tiles);             // ~4 workgroups per CU over the whole grid
    if (want < 1) want = 1;
    if (want > chunks) want = chunks;
    if (want > 64) want = 64;
    return want;
}

// the GEMM of p4p_gemm_wgrad_bf16 WITHOUT its slab sum (ring kernel shapes only): slabs [p4p_wgrad_slab_count(M,N,K)][N][K] are left
// in `workspace` for a later p4p_sum_slabs_multi.  p4p_wgrad_slab_count returns 0 for shapes the ring kernel does not take.
extern "C" int p4p_wgrad_slab_count(int M, int N, int K) {
    return wgrad_use_v3(M, N, K) ? wgrad_ring_splits(M, N, K) : 0;
}
static int launch_wgrad3(const void* G, const void* A, float* workspace, int M, int N, int K, hipStream_t st, int* splits_out) {
    if (wgrad_use_v4(M, N, K)) {
        W4Params q;
        q.G = (const __bf16*)G; q.A = (const __bf16*)A; q.M = M; q.N = N; q.K = K; q.out = workspace;
        q.tiles_per_split = wgrad4_tiles_per_split(M, N, K);
        const int splits4 = ceil_div(ceil_div(M, 64), q.tiles_per_split);
        P4P_CHECK((unsigned long long)64 * (N > K ? N : K) * 2 < (1ull << 31), "p4p_gemm_wgrad_bf16: row stride too large");
        const bool prof = p4p_prof_begin(1, 2.0 * M * N * K, st);
        hipLaunchKernelGGL(gemm_wgrad4_kernel, dim3((N / 256) * (K / 256) * splits4), dim3(512), 0, st, q);
        if (prof) p4p_prof_end(st);
        P4P_LAUNCH_CHECK();
        *splits_out = splits4;
        return 0;
    }
    W3Params q;
    q.G = (const __bf16*)G; q.A = (const __bf16*)A; q.M = M; q.N = N; q.K = K;
    q.steps_per_split = wgrad3_steps_per_split(M, N, K);
    const int splits3 = ceil_div(ceil_div(M, 32), q.steps_per_split);
    q.out = workspace;
    const bool prof = p4p_prof_begin(1, 2.0 * M * N * K, st);
    if (wgrad3_wide(K) && (g_wgrad_wide & 12) == 4)          // bit 2: four waves of 128 x 128 (gemm_wgrad5.h), hand-interleaved
        hipLaunchKernelGGL((gemm_wgrad5_kernel<true>), dim3((N / W3_BN) * (K / 256) * splits3), dim3(256), 0, st, q);
    else if (wgrad3_wide(K) && (g_wgrad_wide & 12) == 8)     // bit 3: the same, compiler-scheduled
        hipLaunchKernelGGL((gemm_wgrad5_kernel<false>), dim3((N / W3_BN) * (K / 256) * splits3), dim3(256), 0, st, q);
    else if (wgrad3_wide(K) && (g_wgrad_wide & 3) >= 2)
        hipLaunchKernelGGL((gemm_wgrad3_kernel<256, true>), dim3((N / W3_BN) * (K / 256) * splits3), dim3(512), 0, st, q);
    else if (wgrad3_wide(K))
        hipLaunchKernelGGL(gemm_wgrad3_kernel<256>, dim3((N / W3_BN) * (K / 256) * splits3), dim3(512), 0, st, q);
    else
        hipLaunchKernelGGL(gemm_wgrad3_kernel<128>, dim3((N / W3_BN) * (K / W3_BK) * splits3), dim3(256), 0, st, q);
    if (prof) p4p_prof_end(st);
    P4P_LAUNCH_CHECK();
    *splits_out = splits3;
    return 0;
}
extern "C" int p4p_gemm_wgrad_slabs_bf16(const void* G, const void* A, float* workspace, int M, int N, int K, void* stream) {
    P4P_CHECK(G && A && workspace, "p4p_gemm_wgrad_slabs_bf16: null pointer");
    P4P_CHECK(wgrad_use_v3(M, N, K), "p4p_gemm_wgrad_slabs_bf16: M=%d N=%d K=%d is not a ring-kernel shape (p4p_wgrad_slab_count is 0)", M, N, K);
    int splits3 = 0;
    return launch_wgrad3(G, A, workspace, M, N, K, (hipStream_t)stream, &splits3);
}

// workspace: splits*(N*K + N) floats.  dbias (may be null) = beta*dbias + column sums of G, from the same pass.
extern "C" int p4p_gemm_wgrad_bf16(const void* G, const void* A, float* dW, float* dbias, float* workspace, int M, int N,
                                   int K, float beta, void* stream) {
    P4P_CHECK(G && A && dW, "p4p_gemm_wgrad_bf16: null pointer");
    P4P_CHECK(M >= 1 && N >= 8 && K >= 8, "p4p_gemm_wgrad_bf16: bad sizes M=%d N=%d K=%d", M, N, K);
    P4P_CHECK(N % 8 == 0 && K % 8 == 0, "p4p_gemm_wgrad_bf16: N=%d and K=%d must be multiples of 8", N, K);
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_use_v3(M, N, K)) {
        P4P_CHECK(workspace, "p4p_gemm_wgrad_bf16: workspace required");
        int splits3 = 0;
        if (launch_wgrad3(G, A, workspace, M, N, K, st, &splits3)) return 2;
        long long n = (long long)N * K;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, st, workspace, dW, n, splits3,
                           beta);
        P4P_LAUNCH_CHECK();
        if (dbias) {       // streaming column sum of G; its partials live behind the slabs
            return p4p_colsum_bf16(G, dbias, workspace + (size_t)splits3 * N * K, M, N, beta, stream);
        }
        return 0;
    }
    int splits = p4p_wgrad_splits(M, N, K);
    int chunks = ceil_div(M, WG_BMR);
    WgradParams p;
    p.G = (const __bf16*)G; p.A = (const __bf16*)A; p.M = M; p.N = N; p.K = K;
    p.chunks_per_split = ceil_div(chunks, splits);
    splits = ceil_div(chunks, p.chunks_per_split);
    p.splits = splits;
    bool direct = splits == 1 && beta == 0.f;
    P4P_CHECK((direct && !dbias) || workspace, "p4p_gemm_wgrad_bf16: workspace required (splits=%d)", splits);
    p.out = direct ? dW : workspace;
    p.bias_out = dbias ? workspace + (size_t)splits * N * K : nullptr;
    dim3 grid(ceil_div(N, 128) * ceil_div(K, 128), splits);
    hipLaunchKernelGGL(gemm_wgrad_kernel, grid, dim3(GEMM_THREADS), 0, st, p);
    P4P_LAUNCH_CHECK();
    if (!direct) {
        long long n = (long long)N * K;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, st, workspace, dW, n,
                           splits, beta);
        P4P_LAUNCH_CHECK();
    }
    if (dbias) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(N / 4, 256)), dim3(256), 0, st, p.bias_out, dbias,
                           (long long)N, splits, beta);
        P4P_LAUNCH_CHECK();
    }
    return 0;
}
