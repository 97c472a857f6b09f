// Attention kernels for sequences of the form T = 64 n + 1 -- exactly what the model produces (S = N/4 point tokens plus
// the cls token: T = 257 / 513 / 2049 at N = 1024 / 2048 / 8192, src/models/pix4point.py:248-252).  With 64-row tiles the
// `+1` costs a whole extra query tile (a 17th 16-row tile with ONE valid row: 8 waves take 2 tiles each and one wave a
// third) and a whole extra key tile (one valid key, masks in every softmax).  Here the odd row -- the LAST row of the
// sequence, attention is equivariant to the row order -- is handled apart, so that everything else is whole tiles:
//   * as a KEY it is a rank-1 term of every query tile: one dot product per query (VALU, from the Q fragments the
//     wave holds anyway), one more numerator, one scaled copy of its value row added to the accumulators;
//   * as a QUERY its key loop is split over the waves (one 64-key tile each), the partial (max, sum, accumulator)
//     triples meet in LDS and one wave combines them.
// The resident forward below (T <= 257) also drops the online softmax: a wave holds the scores of its 16 queries against
// ALL keys in registers (64 n + 1 <= 257), so the row maximum is exact and taken once (two v_permlane swaps instead of
// an LDS round trip per tile), there is no accumulator rescale, and the row sums come out of the matrix pipe (one more
// MFMA per 32 keys against an all-ones A operand) instead of 16 VALU adds per tile.
//
// Arithmetic contract (tests/emulate.py: attention_emulated mirrors it):
//   main query rows:  m = c max_k s_k over ALL keys (c = scale log2 e);  p_k = exp2(fma(s_k, c, -m));
//                     o = sum_k bf16(p_k) v_k  (fp32);  l = sum_k bf16(p_k);  out = bf16(o / l);  lse2 = m + log2 l
//   the last row:     per 64-key tile t (tile 0 also holds the last key): m_t, o_t, l_t as above over the tile's keys;
//                     M = max_t m_t;  o = sum_t 2^(m_t - M) o_t;  l = sum_t 2^(m_t - M) l_t;  out, lse2 as above.
#pragma once

// v_max3_f32.  attention.hip is compiled with -fno-honor-nans (Makefile), so hipcc does not put a canonicalising
// `v_max x, x` in front of every maximum of an MFMA output (the scores are never NaN).  NOT inline asm: hipcc pads no
// hazards for an asm statement, and a VALU read of an MFMA result needs up to 12 wait states (guide 5.7 item 2).
__device__ __forceinline__ float f32_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// dot product of two fragment-layout rows (16 of the 64 dims per lane), summed over the four lanes of the column
__device__ __forceinline__ float frag_dot(const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) s = __builtin_fmaf(bf2f(a[ks][j]), bf2f(b[ks][j]), s);
    return quad_rows_sum(s);
}

#define LONE_PART_STRIDE 68      // floats per partial of the last query row: m, l, -, -, o[64]

// One 16-query tile against NT resident 64-key tiles (+ the last key as a rank-1 term when `sl` is finite).
// On return o[d][r] = sum_k bf16(p_k) v_k[16d+4g+r], l = sum_k bf16(p_k), m = the row maximum in the exp2 domain.
template <int NT>
__device__ __forceinline__ void lone_fwd_tile(const char* klds, const char* vlds, const bf16x8 (&qf)[2], float sl,
                                              const bf16x4 (&vl)[4], float c, int lane, f32x4 (&o)[4], float& l, float& m) {
    f32x4 s[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            s[kt][k16] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                s[kt][k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(klds + kt * TILE_BYTES, k16, ks, lane), qf[ks],
                                                                     s[kt][k16], 0, 0, 0);
        }
    float mx = sl;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            mx = f32_max3(mx, s[kt][k16][0], s[kt][k16][1]);
            mx = f32_max3(mx, s[kt][k16][2], s[kt][k16][3]);
        }
    m = quad_rows_max(mx) * c;
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][k16][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][k16][r], c, -m));
        const bf16x8 pb[2] = {pack_pair(s[kt][0], s[kt][1]), pack_pair(s[kt][2], s[kt][3])};
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(vlds + kt * TILE_BYTES, ss, d, lane), pb[ss], o[d], 0, 0, 0);
            lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[ss], lacc, 0, 0, 0);      // row sums of the rounded P
        }
    }
    const float pl = bf2f(f2bf(__builtin_amdgcn_exp2f(__builtin_fmaf(sl, c, -m))));         // sl = -inf: 0
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[d][r] = __builtin_fmaf(pl, bf2f(vl[d][r]), o[d][r]);
    l = lacc[0] + pl;
}

#define LONE_OSLAB (16 * 144)  // per-wave output slab: 16 rows x (128 B + 16): rows leave as whole 128-B segments
#define LONE_NW 4          // waves per workgroup: 2 workgroups per CU = 2 waves per SIMD, 256 VGPRs each (64 n + 1 <= 257 scores
                           // per lane stay in registers); the independent MFMA batches of the two passes give the overlap
// Diagnostic build only (tools/attn_stamps.py compiles attention.hip with -DP4P_LONE_STAMPS into its own library): s_memtime
// stamps per wave at the phase boundaries, written behind the real lse2 entries.  No stamp exists in the product library.
#ifdef P4P_LONE_STAMPS
#define LONE_STAMP(k)                                                                                               \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                                \
        if (lane == 0) reinterpret_cast<unsigned long long*>(p.lse2 + (size_t)p.B * p.H * p.T)[((size_t)blockIdx.x * LONE_NW + wave) * 8 + (k)] = ts_; \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#else
#define LONE_STAMP(k) do { } while (0)
#endif

template <int NT>
__global__ __launch_bounds__(LONE_NW * 64, 2) void attn_fwd_lone_res_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * NT * TILE_BYTES + NT * LONE_PART_STRIDE * 4 + LONE_NW * LONE_OSLAB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int T = p.T, D = p.D, TL = T - 1;                 // T = 64 NT + 1; TL = the last row
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const float c = p.scale * 1.4426950408889634f;
    char* klds = lds;
    char* vlds = lds + NT * TILE_BYTES;
    float* part = reinterpret_cast<float*>(lds + 2 * NT * TILE_BYTES);
    LONE_STAMP(0);

    {   // stage K and V rows 0 .. 64 NT - 1: 1-KiB pieces of 8 rows
        const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
        for (int pc = wave; pc < 2 * NT * 8; pc += LONE_NW) {
            const int isv = pc >= NT * 8, piece = isv ? pc - NT * 8 : pc;
            const int r = piece * 8 + rin;
            __builtin_amdgcn_global_load_lds(GLB_PTR(seq + (size_t)r * ld + (1 + isv) * D + h * HD + chunk * 8),
                                             LDS_PTR((isv ? vlds : klds) + piece * 1024), 16, 0, 0);
        }
    }
    // the last key / value row: fragment layout (dims 32 ks + 8 g ..) for the dot products, accumulator layout
    // (dims 16 d + 4 g ..) for the rank-1 update
    bf16x8 kl[2];
    bf16x4 vl[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kl[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)TL * ld + D + h * HD + ks * 32 + g * 8);
#pragma unroll
    for (int d = 0; d < 4; ++d) vl[d] = *reinterpret_cast<const bf16x4*>(seq + (size_t)TL * ld + 2 * D + h * HD + d * 16 + 4 * g);
    // every query fragment this wave will need (its NT tiles of 16 rows + the last row), fetched while the K / V pieces
    // are in flight: at two waves per SIMD a load at the head of each tile is an exposed ~1 us of latency per tile
    // (counters: 43 % of the wave cycles parked at waits with the loads inside the loop)
    bf16x8 qall[NT + 1][2];
#pragma unroll
    for (int sidx = 0; sidx <= NT; ++sidx) {
        const int q = sidx < NT ? (wave + LONE_NW * sidx) * 16 + i : TL;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qall[sidx][ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)q * ld + h * HD + ks * 32 + g * 8);
    }
    LONE_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0070);          // own LDS-DMA pieces landed (vmcnt(0)), then publish
    __syncthreads();
    LONE_STAMP(2);

#pragma unroll
    for (int sidx = 0; sidx < NT; ++sidx) {
        const int q = (wave + LONE_NW * sidx) * 16 + i;
        const bf16x8 (&qf)[2] = qall[sidx];
        f32x4 o[4];
        float l, m;
        lone_fwd_tile<NT>(klds, vlds, qf, frag_dot(qf, kl), vl, c, lane, o, l, m);
        const float inv = 1.0f / l;
        // the accumulator layout would store 16 rows x 32 B per instruction; through the wave's slab the tile leaves as
        // 8 rows x 128 B per instruction (this head's whole 128-B segment of an output row)
        char* oslab = lds + 2 * NT * TILE_BYTES + NT * LONE_PART_STRIDE * 4 + wave * LONE_OSLAB;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 w = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
            *reinterpret_cast<bf16x4*>(oslab + i * 144 + d * 32 + g * 8) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = h2 * 8 + (lane >> 3), cc = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(oslab + r * 144 + cc * 16);
            const int qr = (wave + LONE_NW * sidx) * 16 + r;
            *reinterpret_cast<u32x4*>(p.out + ((size_t)b * T + qr) * D + h * HD + cc * 8) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (g == 0 && p.lse2) p.lse2[((size_t)b * p.H + h) * T + q] = m + __builtin_amdgcn_logf(l);
        if (sidx == 0) LONE_STAMP(3);
    }
    LONE_STAMP(4);
    // ---- the last row as a query: wave w < NT takes key tile w (wave 0 also the last key), partials meet in LDS
    if (wave < NT) {
        const bf16x8 (&qf)[2] = qall[NT];
        const float sll = wave == 0 ? frag_dot(qf, kl) : -INFINITY;
        f32x4 o[4];
        float l, m;
        lone_fwd_tile<1>(klds + wave * TILE_BYTES, vlds + wave * TILE_BYTES, qf, sll, vl, c, lane, o, l, m);
        if (i == 0) {
            float* dst = part + wave * LONE_PART_STRIDE;
            if (g == 0) { dst[0] = m; dst[1] = l; }
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<f32x4*>(dst + 4 + d * 16 + 4 * g) = o[d];
        }
    }
    LONE_STAMP(5);
    __syncthreads();
    LONE_STAMP(6);
    if (wave == NT % LONE_NW) {                  // lane = output dim
        float M = part[0];
#pragma unroll
        for (int w = 1; w < NT; ++w) M = fmaxf(M, part[w * LONE_PART_STRIDE]);
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < NT; ++w) {
            const float f = __builtin_amdgcn_exp2f(part[w * LONE_PART_STRIDE] - M);
            o = __builtin_fmaf(f, part[w * LONE_PART_STRIDE + 4 + lane], o);
            l = __builtin_fmaf(f, part[w * LONE_PART_STRIDE + 1], l);
        }
        p.out[((size_t)b * T + TL) * D + h * HD + lane] = f2bf(o * (1.0f / l));
        if (lane == 0 && p.lse2) p.lse2[((size_t)b * p.H + h) * T + TL] = M + __builtin_amdgcn_logf(l);
    }
    LONE_STAMP(7);
}

// (Where a workgroup of the kernel above spends its life, from a diagnostic build with s_memtime stamps -- tools/attn_stamps.py,
// B=256, H=12, T=257: 33.9k cycles, of which 13.1k pass until its K / V pieces have landed (the ISSUE of its 128 KB of loads
// is throttled by the memory system: every workgroup of the chip stages at once at a 1/256 share of HBM), 17.5k in the tiles,
// 3k in the tail; the two workgroups of a CU overlap these phases only partly.  A PERSISTENT variant that fetches the next
// (b,h)'s K and V into registers during the compute phase (16 x 16 B per thread, ds_write at the seam) needs 64 more VGPRs than
// the 256 there are: 71 spills, 185 us against 121 us.  Not kept.  The way to overlap the two phases inside one workgroup is a
// second LDS buffer filled by LDS-DMA, which needs inline-asm fragment reads first (hipcc parks every builtin ds_read behind
// vmcnt(0) while a DMA is outstanding): DESIGN.md section 9.)

// (The walk of attn_bwd_fused_kernel -- a fixed number of workgroups, here two per CU, each taking every 512th (b,h) pair with the same
// body -- was measured on this kernel in round 3: 112-115 us against 99-102 us.  With two workgroups per CU the hardware's
// dispatch lets them drift apart, so that one loads while the other computes; the walk starts both at the same time and keeps them
// in step.  Not kept.  Nor was the full form of it: ONE workgroup of 8 waves per CU, K | V in two 64-KB buffers (two __shared__
// arrays), the LDS-DMAs of pair n + 1 and its query / last-row fragments in flight while pair n is computed, the body written out
// for both buffers, a raw barrier in front of the last row's combine -- bit-identical, no wait for the DMAs anywhere in the body
// (with frag_tr's reads in the one-offset form that keeps their alias-scope info: note in attention.hip), 248 VGPRs, and 118-125 us against 100-112 us: the two
// barriers per pair and the last-row tail (4 of 8 waves, then one) are paid by the whole CU, where two independent workgroups
// fill each other's gaps.)

// (A FUSED backward for these shapes -- one workgroup per (b,h): Q, dO, K and a dS^T tile buffer in LDS (138 KB), dK^T / dV^T
// accumulators in registers for the whole kernel, dS crossing LDS once for dQ, the last row's three gradient rows as
// matrix-vector MFMAs -- was built in round 2 and verified against fp32 autograd (same errors as the two-kernel path).
// It needs 5 MFMA products instead of 7 and reads q|k|v, dO once, but at one workgroup per CU its memory phase (194 KB
// per workgroup at a 1/256 share of HBM: ~180 us over the launch) and its compute phase do not overlap: 500 us against
// 433 us for attn_bwd_dq_res + attn_bwd_dkdv_res at B = 256, H = 12, T = 257.  Not kept)
