// Flash-style multi-head self-attention (head_dim 64) forward + backward for gfx950.
// Reference: timm Attention inside Block, restated in-repo at src/models/apf_utils.py:142-158
// (qkv reshape (B,T,3,H,hd), softmax(q k^T * hd^-0.5) v, head merge), called from
// src/models/pix4point.py:254-255.  The T x T score matrix never leaves registers.
//
// Layout: qkv is the fused projection output [B*T, 3*H*64] bf16 (q | k | v, head-major inside each),
// out / dout are [B*T, H*64] bf16, lse2 / delta are [B,H,T] fp32.  lse2 = log2(sum exp(score)) in the
// base-2 domain of c*score with c = scale*log2(e), so backward recomputes p = exp2(c*s - lse2).
//
// Work decomposition (all three kernels): workgroup = 4 waves; a wave owns 16 rows of the
// "lane side" (queries in fwd/dQ, keys in dK/dV), the other side streams through LDS in 64-row
// tiles filled by LDS-DMA (global_load_lds_dwordx4, double buffered, one barrier per tile).
// Every tile is [64 rows][64 bf16] with 16-B chunk' = chunk ^ (row & 7): conflict-free both for
// ds_read_b128 fragment reads and for ds_read_b64_tr_b16 transposing reads of 8 consecutive rows.
//
// MFMA plan (v_mfma_f32_16x16x32_bf16; lane l: g = l>>4, i = l&15):
//   S^T[key][q]  = K . Q^T      A = K rows (b128 from LDS), B = Q^T (registers)      -> lane i = query
//   O^T[d][q]   += V^T . P^T    A = V^T (tr read),         B = P^T = the S^T accumulators, packed
// An accumulator tile (rows 16t+4g+r on the contraction side) is used directly as the next MFMA's B
// operand; the A operand is fetched in the matching permuted k order: slot (g,j) of k-step s is
// contraction row 32s + 16(j>>2) + 4g + (j&3), which is exactly what two tr reads of rows
// 32s+4g.. and 32s+16+4g.. deliver.  No P/dS tile ever goes through LDS.
#include "common.h"

#define ATT_THREADS 256
#define HD 64
#define TILE_BYTES (64 * 128)
// The tiles are staged with LDS-DMA (global_load_lds), which is tracked by vmcnt, not by the workgroup fence inside
// __syncthreads(): every wave drains its own DMAs explicitly before the barrier, so that after the barrier all
// pieces of the tile have landed whatever the wave does next (a wave with no valid row skips the math entirely).
// The wait is vmcnt(0) AND lgkmcnt(0) (0x0070): with an explicit vmcnt-only wait in front, hipcc dropped the fence's own
// lgkmcnt(0) in attn_bwd_dkdv_kernel, and the ds_write of the next tile's lse / -delta could then still be in its SIMD's
// LDS queue when a wave of another SIMD read it right after the barrier (one 16-key dK tile off by ~1e-3 in about one
// launch in ten at B*H = 1536, T = 513; found by the bitwise-reproducibility test).
#define STAGE_FENCE()                          \
    do {                                       \
        __builtin_amdgcn_s_waitcnt(0x0070);    \
        __syncthreads();                       \
    } while (0)

// Lazy rescale of the online-softmax forward kernels (the flash-attention idiom): the reference m of a 16-query tile follows the running row
// maximum only when some query of the tile outgrew it by more than 2^ATT_LAZY_THR; numerators then reach 2^thr instead of 1 -- the same
// relative rounding in bf16, the same quotient o / l -- and the rescale of the accumulators (with 16 queries deciding together it ran on
// nearly every tile of random data: 20 multiplies per 16 scores of a lane) becomes rare.  Measured (tools/attn_fwd_thr_ab.py, same box):
// T = 513 363 -> 341 us, T = 2049 (B = 8, H = 16) 211 -> 190 us.  tests/emulate.py states it (ONLINE_THR).
#define ATT_LAZY_THR 4.0f
struct AttnParams {
    const __bf16* qkv;     // [B*T, 3D]
    const __bf16* o;       // [B*T, D]   (bwd)
    const __bf16* dout;    // [B*T, D]   (bwd)
    __bf16* out;           // fwd: O [B*T, D]; bwd: dqkv [B*T, 3D]
    float* lse2;           // [B,H,T]
    float* delta;          // [B,H,T]
    float* colsum;         // bwd, optional: [B*ceil(T/64)*4][3D] per-wave column sums of dqkv (qkv bias gradient)
    int B, T, H, D;
    float scale;
    int stagger_n, stagger_phases, stagger_cycles;   // attention_fused.h: start-time shift of the first round of workgroups (0 = off)
    float thr;             // fwd, online-softmax kernels: lazy-rescale threshold in base-2 exponent units (ATT_LAZY_THR; 0 = the running maximum exactly)
};

// Workgroup -> (64-row block, head, sequence).  The hardware deals consecutive workgroups round-robin to the 8 XCDs
// (each with its own L2); all blocks of one (b,h) re-read the same K/V (or Q/dO) rows, so they are given linear
// ids that are congruent mod 8 and adjacent in their XCD's queue: the rows cross the fabric once, the other
// blocks hit that XCD's L2 (measured: -5% on all three kernels at B=256 T=257 H=12).
__device__ __forceinline__ bool attn_block(const AttnParams& p, int& blk, int& h, int& b) {
    const int nblk = (p.T + 63) >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nblk) * 8 + xcd;
    blk = slot - (slot / nblk) * nblk;
    if (bh >= p.B * p.H) return false;
    b = bh / p.H;
    h = bh - b * p.H;
    return true;
}
static inline dim3 attn_grid(int B, int T, int H) { return dim3((unsigned)(ceil_div((long long)B * H, 8) * 8 * ceil_div(T, 64))); }

// Stage 64 rows (row r of the tile = global row min(row0 + r, T-1) of sequence b) x 128 B.
__device__ __forceinline__ void stage_rows(const __bf16* __restrict__ base, size_t ld, int row0, int T,
                                           char* tile, int wave, int lane) {
    const int rin = lane >> 3;
    const int chunk = (lane & 7) ^ rin;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int piece = wave * 2 + q;
        int r = row0 + piece * 8 + rin;
        r = r < T ? r : T - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + (size_t)r * ld + chunk * 8), LDS_PTR(tile + piece * 1024),
                                         16, 0, 0);
    }
}

// (Running source pointers instead of this clamp / multiply chain were measured in the streaming backward kernels: slower,
// 1,255 -> 1,338 us at T = 513 -- eight more VGPRs cost more than the ~40 VALU instructions per tile they save; the forward keeps
// its own running pointers.)
// b128 fragment: rows r16*16 + i, contraction columns 32ks + 8g .. +7
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int r16, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(tile + (r16 * 16 + i) * 128 + (((ks * 4 + g) ^ (i & 7)) << 4));
}

// transposed fragment: output index (lane side of the operand) = column 16*cb + i of the tile,
// contraction slots (g,j) = rows 32s + 16(j>>2) + 4g + (j&3).
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int s, int cb, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int row = 32 * s + 4 * g + (i >> 2);
    const int p = i & 3;
    const int lchunk = 2 * cb + (p >> 1);
    // (Round 3, measured and NOT adopted: both addresses as ONE offset from the tile, `tile + (off | 2048)`.  As written here hipcc
    // turns the second address into `first + 2048` -- an immediate on the same address register, free -- but a pointer derived from a
    // derived pointer, which its LDS lowering (depth-limited walk over a pointer's users) no longer traces to its array: that read
    // carries no alias-scope info, and a read without it waits for EVERY LDS-DMA in flight (vmcnt(0)), whatever array the DMA
    // writes (MIR before si-insert-waitcnts: 144 of 288 transposed reads of a kernel without `!alias.scope`; with the one-offset
    // form none).  That is the mechanism behind "hipcc drains vmcnt(0) in front of ds_reads while a DMA is outstanding".  The
    // one-offset form costs a v_or and an address register per fragment, though: attn_fwd_q_kernel<2> went from 126 VGPRs to
    // more than 128, T = 513 forward 343-349 -> 373-376 us (same box).  The kernels here keep their staging in ONE array anyway
    // (a read of it must wait for a DMA into it); a kernel with separate arrays per buffer needs the one-offset form in ITS reads.)
    const char* a0 = tile + row * 128 + ((lchunk ^ (row & 7)) << 4) + (p & 1) * 8;
    const int row1 = row + 16;
    const char* a1 = tile + row1 * 128 + ((lchunk ^ (row1 & 7)) << 4) + (p & 1) * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    bf16x8 r = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return r;
}

// Column sums over the wave's 16 rows of an accumulator set held as X^T[col 16d+4g+r][row i] -> 64 floats at dst
// (global; one partial row per WAVE: no LDS, no barrier -- these kernels are latency-bound per workgroup and two extra
// barriers in the epilogue cost more than the larger partial buffer).  The qkv bias gradient is the column sum of
// dqkv; the rows are in registers here, so the backward kernels emit partial sums instead of a separate pass
// re-reading the 300-MB dqkv.  The bf16-ROUNDED values are summed: the result equals a column sum of the stored
// tensor up to summation order (which is fixed: reproducible).
template <bool ROUND = true>     // ROUND: x holds unrounded gradients (sum what is stored); false: x already is a sum of rounded values
__device__ __forceinline__ void wave_colsum(const f32x4 (&x)[4], bool valid, float* dst, int lane) {
    // Transposing butterfly over the 16 lanes of a row: at every step a lane keeps the half of its values whose
    // column-index bit matches its own lane bit and adds the partner's copy of them (xor masks 8, 7, 2, 1 = the DPP
    // patterns row_ror:8, row_half_mirror, quad_perm[2,3,0,1], quad_perm[1,0,3,2]; they generate all 16 lanes).
    // 15 DPP adds instead of 64, and lane i ends up with the sum of column c = i (c = 4d + r).
    const int i = lane & 15, g = lane >> 4;
    const bool b3 = i & 8, b2 = i & 4, b1 = i & 2, b0 = i & 1;
    float v[16];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[d * 4 + r] = valid ? (ROUND ? bf2f(f2bf(x[d][r])) : x[d][r]) : 0.f;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int c = 0; c < 8; ++c) w8[c] = (b3 ? v[c + 8] : v[c]) + dpp_f32<0x128>(b3 ? v[c] : v[c + 8]);
#pragma unroll
    for (int c = 0; c < 4; ++c) w4[c] = (b2 ? w8[c + 4] : w8[c]) + dpp_f32<0x141>(b2 ? w8[c] : w8[c + 4]);
#pragma unroll
    for (int c = 0; c < 2; ++c) w2[c] = (b1 ? w4[c + 2] : w4[c]) + dpp_f32<0x4E>(b1 ? w4[c] : w4[c + 2]);
    const float tot = (b0 ? w2[1] : w2[0]) + dpp_f32<0xB1>(b0 ? w2[0] : w2[1]);
    dst[16 * (i >> 2) + 4 * g + (i & 3)] = tot;
}

// all-to-all over the four lanes {i, i+16, i+32, i+48} that share a query (or key) column of an MFMA accumulator: two
// half-exchanges (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second; v_permlane16_swap: odd
// 16-lane rows of the first <-> even rows of the second), VALU only -- no LDS round trip as ds_bpermute / __shfl_xor.
// INLINE ASM on purpose: with __builtin_amdgcn_permlane{16,32}_swap this hipcc (ROCm 7.2) folds the second result into
// the first when both operands are the same value (`v_permlane32_swap v1, v2; v_add v1, v1, v1`: found as a wrong softmax);
// the s_nop 1 covers the "VALU write -> v_permlane read" hazard that hipcc does not pad inside a string.
__device__ __forceinline__ void lane_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float quad_rows_max(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    a = b = __builtin_fmaxf(a, b);
    lane_swap16(a, b);
    return __builtin_fmaxf(a, b);
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    a = b = a + b;
    lane_swap16(a, b);
    return a + b;
}

// (the streaming kernels call these once per key tile: as __shfl_xor they were two ds_bpermute round trips each)
__device__ __forceinline__ float xor_lanes_max(float v) { return quad_rows_max(v); }
__device__ __forceinline__ float xor_lanes_sum(float v) { return quad_rows_sum(v); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// One K/V tile of the forward pass for a wave's 16 queries.  MASK is a compile-time flag: only the ragged last
// tile pays for the key masking (as a runtime condition hipcc if-converts it into 16 v_cndmask per tile).
// `lacc` = row sums of the bf16-ROUNDED numerators (what the P.V product sums), from the matrix pipe: one more MFMA per 32 keys
// against an all-ones A operand instead of 16 VALU adds per tile (every register of lacc holds the lane's query's sum).
// The accumulators are rescaled only when some lane's maximum grew (wave-uniform branch; alpha = 1 exactly otherwise).
template <bool MASK>
__device__ __forceinline__ void attn_fwd_tile(const char* kt_lds, const char* vt_lds, const bf16x8 (&qf)[2], f32x4 (&o)[4],
                                              float& m, f32x4& lacc, float c, float thr, int key0, int T, int lane) {
    const int g = lane >> 4;
    const int nvalid = MASK ? T - key0 : 64;            // valid keys in this tile (wave-uniform)
    f32x4 s[4];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
        s[k16] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!MASK || k16 * 16 < nvalid) {               // ragged tile: sub-tiles past the end are not computed
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                s[k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(kt_lds, k16, ks, lane), qf[ks], s[k16], 0, 0, 0);
        }
    }
    if (MASK) {
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k16 * 16 + 4 * g + r >= nvalid) s[k16][r] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
    for (int k16 = 1; k16 < 4; ++k16) mx = fmaxf(mx, fmaxf(fmaxf(s[k16][0], s[k16][1]), fmaxf(s[k16][2], s[k16][3])));
    mx = xor_lanes_max(mx) * c;
    if (__builtin_amdgcn_ballot_w64(mx > m + thr) != 0) {     // lazy rescale: below
        const float mnew = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        m = mnew;
        lacc *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] *= alpha;
    }
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[k16][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[k16][r], c, -m));
    bf16x8 pb[2] = {pack_pair(s[0], s[1]), pack_pair(s[2], s[3])};
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (MASK && ss * 32 >= nvalid) continue;          // no valid key in this 32-key contraction step
#pragma unroll
        for (int d = 0; d < 4; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(vt_lds, ss, d, lane), pb[ss], o[d], 0, 0, 0);
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[ss], lacc, 0, 0, 0);
    }
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * TILE_BYTES];     // [buf][K|V]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    int qb, h, b;
    if (!attn_block(p, qb, h, b)) return;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* kbase = seq + D + h * HD;
    const __bf16* vbase = seq + 2 * D + h * HD;
    const float c = p.scale * 1.4426950408889634f;

    const int q = qb * 64 + wave * 16 + i;
    const int qc = q < T ? q : T - 1;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)qc * ld + h * HD + ks * 32 + g * 8);

    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};

    // LDS-DMA sources of this lane's 2 K pieces and 2 V pieces for tile 0; a full tile further = +64 rows.
    // Only the ragged last tile needs row clamping (stage_rows); full tiles just advance the pointers.
    const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
    const __bf16* ksrc[2];
    const __bf16* vsrc[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const size_t off = (size_t)((wave * 2 + qq) * 8 + rin) * ld + chunk * 8;
        ksrc[qq] = kbase + off;
        vsrc[qq] = vbase + off;
    }
    const size_t tile_step = (size_t)64 * ld;
    auto stage_full = [&](int kt, char* buf) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(ksrc[qq] + kt * tile_step), LDS_PTR(buf + (wave * 2 + qq) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(vsrc[qq] + kt * tile_step), LDS_PTR(buf + TILE_BYTES + (wave * 2 + qq) * 1024), 16, 0, 0);
        }
    };
    auto stage_any = [&](int kt, char* buf) {
        if ((kt + 1) * 64 <= T) stage_full(kt, buf);
        else {
            stage_rows(kbase, ld, kt * 64, T, buf, wave, lane);
            stage_rows(vbase, ld, kt * 64, T, buf + TILE_BYTES, wave, lane);
        }
    };

    const int nkt = (T + 63) / 64;
    const int nfull = T / 64;                                 // tiles that need no masking
    stage_any(0, lds);
    const bool wave_active = qb * 64 + wave * 16 < T;     // wave-uniform: a wave with no valid query does no math
    for (int kt = 0; kt < nfull; ++kt) {
        STAGE_FENCE();
        const char* cur = lds + (kt & 1) * 2 * TILE_BYTES;
        if (kt + 1 < nkt) stage_any(kt + 1, lds + ((kt + 1) & 1) * 2 * TILE_BYTES);
        if (wave_active) attn_fwd_tile<false>(cur, cur + TILE_BYTES, qf, o, m, lacc, c, p.thr, kt * 64, T, lane);
    }
    if (nfull < nkt) {
        STAGE_FENCE();
        const char* cur = lds + (nfull & 1) * 2 * TILE_BYTES;
        if (wave_active) attn_fwd_tile<true>(cur, cur + TILE_BYTES, qf, o, m, lacc, c, p.thr, nfull * 64, T, lane);
    }
    const float l = lacc[0];
    const float inv = 1.0f / l;
    if (q < T) {
        __bf16* dst = p.out + ((size_t)b * T + q) * D + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 w = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
            *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
        }
        if (g == 0 && p.lse2) p.lse2[((size_t)b * p.H + h) * T + q] = m + __builtin_amdgcn_logf(l);
    }
}

// ---- QT query tiles per wave (workgroup = QT * 64 queries).  Every wave of the kernel above reads the whole K and V tile from
// LDS for its 16 queries: 16 KiB per wave and tile against 16 MFMAs, i.e. the LDS pipe (64 KiB per workgroup-tile) is as busy
// as the matrix pipe.  With QT tiles per wave every K / V fragment read feeds QT MFMAs: LDS bytes per FLOP drop by QT.
template <bool MASK, int QT>
__device__ __forceinline__ void attn_fwd_tile_q(const char* kt_lds, const char* vt_lds, const bf16x8 (&qf)[QT][2],
                                                f32x4 (&o)[QT][4], float (&m)[QT], f32x4 (&lacc)[QT], float c, float thr, int key0, int T,
                                                int lane) {
    const int g = lane >> 4;
    const int nvalid = MASK ? T - key0 : 64;
    f32x4 s[QT][4];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][k16] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!MASK || k16 * 16 < nvalid) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = frag_rows(kt_lds, k16, ks, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[t][k16], 0, 0, 0);
            }
        }
    }
    bf16x8 pb[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (MASK) {
#pragma unroll
            for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k16 * 16 + 4 * g + r >= nvalid) s[t][k16][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3]));
#pragma unroll
        for (int k16 = 1; k16 < 4; ++k16)
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][k16][0], s[t][k16][1]), fmaxf(s[t][k16][2], s[t][k16][3])));
        mx = xor_lanes_max(mx) * c;
        if (__builtin_amdgcn_ballot_w64(mx > m[t] + thr) != 0) {       // some lane's maximum grew (by more than thr: lazy rescale, 0 = off): rescale (see attn_fwd_tile)
            const float mnew = fmaxf(m[t], mx);
            const float alpha = __builtin_amdgcn_exp2f(m[t] - mnew);
            m[t] = mnew;
            lacc[t] *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d) o[t][d] *= alpha;
        }
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][k16][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][k16][r], c, -m[t]));
        pb[t][0] = pack_pair(s[t][0], s[t][1]);
        pb[t][1] = pack_pair(s[t][2], s[t][3]);
    }
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (MASK && ss * 32 >= nvalid) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const bf16x8 vf = frag_tr(vt_lds, ss, d, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) o[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[t][ss], o[t][d], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) lacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[t][ss], lacc[t], 0, 0, 0);
    }
}

template <int QT>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd_q_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * TILE_BYTES];     // [buf][K|V]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int T = p.T, D = p.D;
    // same XCD-aware decode as attn_block, with QT*64-query blocks
    const int nblk = (T + QT * 64 - 1) / (QT * 64);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nblk) * 8 + xcd;
    const int qb = slot - (slot / nblk) * nblk;
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* kbase = seq + D + h * HD;
    const __bf16* vbase = seq + 2 * D + h * HD;
    const float c = p.scale * 1.4426950408889634f;

    const int q0 = qb * QT * 64 + wave * QT * 16;            // this wave's QT consecutive 16-query tiles
    bf16x8 qf[QT][2];
    f32x4 o[QT][4];
    float m[QT];
    f32x4 lacc[QT];
#pragma unroll
    for (int tq = 0; tq < QT; ++tq) {
        int q = q0 + tq * 16 + i;
        q = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[tq][ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)q * ld + h * HD + ks * 32 + g * 8);
#pragma unroll
        for (int d = 0; d < 4; ++d) o[tq][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[tq] = -INFINITY;
        lacc[tq] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
    const __bf16* ksrc[2];
    const __bf16* vsrc[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const size_t off = (size_t)((wave * 2 + qq) * 8 + rin) * ld + chunk * 8;
        ksrc[qq] = kbase + off;
        vsrc[qq] = vbase + off;
    }
    const size_t tile_step = (size_t)64 * ld;
    auto stage_any = [&](int kt, char* buf) {          // tiles are staged in ascending order: the sources advance by one tile per call
        if ((kt + 1) * 64 <= T) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                __builtin_amdgcn_global_load_lds(GLB_PTR(ksrc[qq]), LDS_PTR(buf + (wave * 2 + qq) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR(vsrc[qq]), LDS_PTR(buf + TILE_BYTES + (wave * 2 + qq) * 1024), 16, 0, 0);
                ksrc[qq] += tile_step;
                vsrc[qq] += tile_step;
            }
        } else {
            stage_rows(kbase, ld, kt * 64, T, buf, wave, lane);
            stage_rows(vbase, ld, kt * 64, T, buf + TILE_BYTES, wave, lane);
        }
    };
    const int nkt = (T + 63) / 64, nfull = T / 64;
#ifdef P4P_LONE_STAMPS      // diagnostic build (tools/attn_stamps.py): cycles spent waiting at the stage fence vs in the tile math
    unsigned long long t_wait = 0, t_math = 0, t_start = __builtin_readcyclecounter();
#endif
    stage_any(0, lds);
    const bool wave_active = q0 < T;
    for (int kt = 0; kt < nfull; ++kt) {
#ifdef P4P_LONE_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ta = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#endif
        STAGE_FENCE();
#ifdef P4P_LONE_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tb = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#endif
        const char* cur = lds + (kt & 1) * 2 * TILE_BYTES;
        if (kt + 1 < nkt) stage_any(kt + 1, lds + ((kt + 1) & 1) * 2 * TILE_BYTES);
        if (wave_active) attn_fwd_tile_q<false, QT>(cur, cur + TILE_BYTES, qf, o, m, lacc, c, p.thr, kt * 64, T, lane);
#ifdef P4P_LONE_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tc = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        t_wait += tb - ta;
        t_math += tc - tb;
#endif
    }
#ifdef P4P_LONE_STAMPS
    if (lane == 0 && p.lse2) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.lse2 + (size_t)p.B * p.H * p.T) + ((size_t)blockIdx.x * 4 + wave) * 4;
        dst[0] = t_wait; dst[1] = t_math; dst[2] = __builtin_readcyclecounter() - t_start; dst[3] = wave_active;
    }
#endif
    if (nfull < nkt) {
        STAGE_FENCE();
        const char* cur = lds + (nfull & 1) * 2 * TILE_BYTES;
        if (wave_active) attn_fwd_tile_q<true, QT>(cur, cur + TILE_BYTES, qf, o, m, lacc, c, p.thr, nfull * 64, T, lane);
    }
#pragma unroll
    for (int tq = 0; tq < QT; ++tq) {
        const float l = lacc[tq][0];
        const float inv = 1.0f / l;
        const int q = q0 + tq * 16 + i;
        if (q < T) {
            __bf16* dst = p.out + ((size_t)b * T + q) * D + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 w = {f2bf(o[tq][d][0] * inv), f2bf(o[tq][d][1] * inv), f2bf(o[tq][d][2] * inv), f2bf(o[tq][d][3] * inv)};
                *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
            }
            if (g == 0 && p.lse2) p.lse2[((size_t)b * p.H + h) * T + q] = m[tq] + __builtin_amdgcn_logf(l);
        }
    }
}

// Forward for short sequences (T <= RES_ROWS): one workgroup = one (b,h), all of K and V resident in LDS (2 x 34 KiB),
// 8 waves that each take whole 16-query tiles (17 at T = 257) and run over the key tiles with NO further barrier or
// staging wait.  The streaming kernel above spends its time on per-workgroup fixed latencies and one barrier + LDS-DMA
// wait per 64-key tile (measured at B=256, T=257, H=12: 69 us with neither math nor staging in the loop, +60 us for the
// staging, +67 us for the math, and the three do not overlap); here a workgroup pays one load latency for 17 x 5 tiles
// of work and two workgroups share a CU.
#define RES_ROWS 288          // ceil(257 / 32) * 32: the PV / dK contraction steps read 32 rows at a time
#define RES_NW 8
#define RES_COLSUM_ROWS 8     // partial rows per sequence of the fused qkv bias gradient (dq: 8 waves; dk/dv: 4 waves + 4 zero rows)
__global__ __launch_bounds__(RES_NW * 64, 4) void attn_fwd_res_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * RES_ROWS * 128];          // K rows | V rows
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const float c = p.scale * 1.4426950408889634f;
    char* klds = lds;
    char* vlds = lds + RES_ROWS * 128;

    // ---- stage K and V: 1-KiB pieces of 8 rows (rows past T repeat row T-1; they are masked below)
    const int npieces = ((T + 31) >> 5) * 4;        // whole 32-row contraction steps: masked rows must still hold finite data (0 x NaN)
    const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
    for (int pc = wave; pc < 2 * npieces; pc += RES_NW) {
        const int isv = pc >= npieces, piece = isv ? pc - npieces : pc;
        int r = piece * 8 + rin;
        r = r < T ? r : T - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(seq + (size_t)r * ld + (1 + isv) * D + h * HD + chunk * 8),
                                         LDS_PTR((isv ? vlds : klds) + piece * 1024), 16, 0, 0);
    }
    // ---- this wave's query tiles (rotated per (b,h): the extra 17th tile does not always land on the same SIMD)
    const int nqt = (T + 15) >> 4;
    const int w0 = (wave + (bh >> 3)) & (RES_NW - 1);
    constexpr int MAXQ = (RES_ROWS / 16 + RES_NW - 1) / RES_NW;          // 3
    bf16x8 qf[MAXQ][2];
#pragma unroll
    for (int s = 0; s < MAXQ; ++s) {
        const int qt = w0 + s * RES_NW;
        int q = qt * 16 + i;
        q = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[s][ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)q * ld + h * HD + ks * 32 + g * 8);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);          // own LDS-DMA pieces landed (vmcnt(0)), then publish
    __syncthreads();

    const int nkt = (T + 63) / 64, nfull = T / 64;
#pragma unroll
    for (int s = 0; s < MAXQ; ++s) {
        const int qt = w0 + s * RES_NW;
        if (qt >= nqt) break;
        f32x4 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY;
        f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nfull; ++kt)
            attn_fwd_tile<false>(klds + kt * TILE_BYTES, vlds + kt * TILE_BYTES, qf[s], o, m, lacc, c, p.thr, kt * 64, T, lane);
        if (nfull < nkt)
            attn_fwd_tile<true>(klds + nfull * TILE_BYTES, vlds + nfull * TILE_BYTES, qf[s], o, m, lacc, c, p.thr, nfull * 64, T, lane);
        const float l = lacc[0];
        const float inv = 1.0f / l;
        const int q = qt * 16 + i;
        if (q < T) {
            __bf16* dst = p.out + ((size_t)b * T + q) * D + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 w = {f2bf(o[d][0] * inv), f2bf(o[d][1] * inv), f2bf(o[d][2] * inv), f2bf(o[d][3] * inv)};
                *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
            }
            if (g == 0 && p.lse2) p.lse2[((size_t)b * p.H + h) * T + q] = m + __builtin_amdgcn_logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// delta[b,h,t] = sum_d dO[t, h*64+d] * O[t, h*64+d]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int T, int H, int D) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * T) return;
    const int b = (int)(row / T), tt = (int)(row - (long long)b * T);
    for (int c0 = 0; c0 < D; c0 += 256) {
        const int cidx = c0 + lane * 4;
        float s = 0.f;
        if (cidx < D) {
            bf16x4 a = *reinterpret_cast<const bf16x4*>(o + row * D + cidx);
            bf16x4 d = *reinterpret_cast<const bf16x4*>(dout + row * D + cidx);
            s = (bf2f(a[0]) * bf2f(d[0]) + bf2f(a[1]) * bf2f(d[1])) + (bf2f(a[2]) * bf2f(d[2]) + bf2f(a[3]) * bf2f(d[3]));
        }
        s += dpp_f32<0xB1>(s); s += dpp_f32<0x4E>(s); s += dpp_f32<0x141>(s); s += dpp_f32<0x140>(s);
        if ((lane & 15) == 0 && cidx < D) delta[((size_t)b * H + cidx / HD) * T + tt] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: same decomposition as forward (query on the lane), K/V tiles stream.
//   dP^T[key][q] = V . dO^T ;  dS^T = p * (dP^T - delta[q]) * scale ;  dQ^T[d][q] += K^T . dS^T
// ------------------------------------------------------------------------------------------------
// One K/V tile of the dQ pass for a wave's 16 queries.  RAGGED (compile time) is the last tile when T % 64 != 0: only
// it pays for the key masks and the sub-tile skips.  dS is left unscaled; the kernel applies `scale` to dQ once.
template <bool RAGGED>
__device__ __forceinline__ void attn_dq_tile(const char* kt_lds, const char* vt_lds, const bf16x8 (&qf)[2],
                                             const bf16x8 (&dof)[2], f32x4 (&dq)[4], float c, float lse, float dl,
                                             int nvalid, int lane) {
    const int g = lane >> 4;
    f32x4 s[4], dp[4];
    // "row constants as the initial accumulator": the dP chains start from -delta[q], so dS = P (dP - delta) costs no
    // subtraction (16 VALU per tile); the register quad is the C operand of each chain's first MFMA
    const f32x4 ndl = {-dl, -dl, -dl, -dl};
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
        s[k16] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[k16] = ndl;
        if (!RAGGED || k16 * 16 < nvalid) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s[k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(kt_lds, k16, ks, lane), qf[ks], s[k16], 0, 0, 0);
                dp[k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(vt_lds, k16, ks, lane), dof[ks], dp[k16], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[k16][r], c, -lse));
            if (RAGGED && k16 * 16 + 4 * g + r >= nvalid) pr = 0.f;
            s[k16][r] = pr * dp[k16][r];
        }
    bf16x8 db[2] = {pack_pair(s[0], s[1]), pack_pair(s[2], s[3])};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (RAGGED && ss * 32 >= nvalid) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(kt_lds, ss, d, lane), db[ss], dq[d], 0, 0, 0);
    }
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * TILE_BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    int qb, h, b;
    if (!attn_block(p, qb, h, b)) return;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* kbase = seq + D + h * HD;
    const __bf16* vbase = seq + 2 * D + h * HD;
    const float c = p.scale * 1.4426950408889634f;

    const int q = qb * 64 + wave * 16 + i;
    const int qc = q < T ? q : T - 1;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)qc * ld + h * HD + ks * 32 + g * 8);
        dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((size_t)b * T + qc) * D + h * HD + ks * 32 + g * 8);
    }
    const size_t stat = ((size_t)b * p.H + h) * T + qc;
    const float lse = p.lse2[stat], dl = p.delta[stat];
    const bool wave_active = qb * 64 + wave * 16 < T;

    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (T + 63) / 64, nfull = T / 64;
    stage_rows(kbase, ld, 0, T, lds, wave, lane);
    stage_rows(vbase, ld, 0, T, lds + TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        STAGE_FENCE();
        const char* kt_lds = lds + (kt & 1) * 2 * TILE_BYTES;
        const char* vt_lds = kt_lds + TILE_BYTES;
        if (kt + 1 < nkt) {
            char* nxt = lds + ((kt + 1) & 1) * 2 * TILE_BYTES;
            stage_rows(kbase, ld, (kt + 1) * 64, T, nxt, wave, lane);
            stage_rows(vbase, ld, (kt + 1) * 64, T, nxt + TILE_BYTES, wave, lane);
        }
        if (!wave_active) continue;                        // no valid query in this wave: staging + barrier only
        if (kt < nfull) attn_dq_tile<false>(kt_lds, vt_lds, qf, dof, dq, c, lse, dl, 64, lane);
        else attn_dq_tile<true>(kt_lds, vt_lds, qf, dof, dq, c, lse, dl, T - kt * 64, lane);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] *= p.scale;
    if (p.colsum) wave_colsum(dq, q < T, p.colsum + (((size_t)b * ((T + 63) >> 6) + qb) * 4 + wave) * ld + h * HD, lane);
    if (q < T) {
        __bf16* dst = p.out + ((size_t)b * T + q) * ld + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 w = {f2bf(dq[d][0]), f2bf(dq[d][1]), f2bf(dq[d][2]), f2bf(dq[d][3])};
            *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
        }
    }
}

// ---- QT query tiles per wave (see attn_fwd_q_kernel): every K / V fragment read from LDS feeds QT MFMAs.
template <bool RAGGED, int QT>
__device__ __forceinline__ void attn_dq_tile_q(const char* kt_lds, const char* vt_lds, const bf16x8 (&qf)[QT][2],
                                               const bf16x8 (&dof)[QT][2], f32x4 (&dq)[QT][4], float c, const float (&lse)[QT],
                                               const float (&dl)[QT], int nvalid, int lane) {
    const int g = lane >> 4;
    f32x4 s[QT][4], dp[QT][4];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            s[t][k16] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t][k16] = f32x4{-dl[t], -dl[t], -dl[t], -dl[t]};          // dP chains start from -delta[q] (see attn_dq_tile)
        }
        if (!RAGGED || k16 * 16 < nvalid) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = frag_rows(kt_lds, k16, ks, lane), vf = frag_rows(vt_lds, k16, ks, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    s[t][k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][ks], s[t][k16], 0, 0, 0);
                    dp[t][k16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[t][ks], dp[t][k16], 0, 0, 0);
                }
            }
        }
    }
    bf16x8 db[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][k16][r], c, -lse[t]));
                if (RAGGED && k16 * 16 + 4 * g + r >= nvalid) pr = 0.f;
                s[t][k16][r] = pr * dp[t][k16][r];
            }
        db[t][0] = pack_pair(s[t][0], s[t][1]);
        db[t][1] = pack_pair(s[t][2], s[t][3]);
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (RAGGED && ss * 32 >= nvalid) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const bf16x8 kt = frag_tr(kt_lds, ss, d, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) dq[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, db[t][ss], dq[t][d], 0, 0, 0);
        }
    }
}

template <int QT>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_q_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * 2 * TILE_BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int T = p.T, D = p.D;
    const int nblk = (T + QT * 64 - 1) / (QT * 64);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nblk) * 8 + xcd;
    const int qb = slot - (slot / nblk) * nblk;
    if (bh >= p.B * p.H) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* kbase = seq + D + h * HD;
    const __bf16* vbase = seq + 2 * D + h * HD;
    const float c = p.scale * 1.4426950408889634f;

    const int q0 = qb * QT * 64 + wave * QT * 16;
    bf16x8 qf[QT][2], dof[QT][2];
    float lse[QT], dl[QT];
    f32x4 dq[QT][4];
#pragma unroll
    for (int tq = 0; tq < QT; ++tq) {
        int q = q0 + tq * 16 + i;
        q = q < T ? q : T - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[tq][ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)q * ld + h * HD + ks * 32 + g * 8);
            dof[tq][ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((size_t)b * T + q) * D + h * HD + ks * 32 + g * 8);
        }
        const size_t stat = ((size_t)b * p.H + h) * T + q;
        lse[tq] = p.lse2[stat];
        // delta[q] = dO[q] . O[q] over this head's 64 columns, from the dO fragment already in registers and the matching O
        // fragment (no separate pass over O and dO); written out for the dK/dV kernel that follows on the stream
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(p.o + ((size_t)b * T + q) * D + h * HD + ks * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += bf2f(dof[tq][ks][j]) * bf2f(of[j]);
        }
        dl[tq] = xor_lanes_sum(part);
        if (g == 0 && q0 + tq * 16 + i < T) p.delta[stat] = dl[tq];
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[tq][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool wave_active = q0 < T;
    const int nkt = (T + 63) / 64, nfull = T / 64;
    stage_rows(kbase, ld, 0, T, lds, wave, lane);
    stage_rows(vbase, ld, 0, T, lds + TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        STAGE_FENCE();
        const char* kt_lds = lds + (kt & 1) * 2 * TILE_BYTES;
        const char* vt_lds = kt_lds + TILE_BYTES;
        if (kt + 1 < nkt) {
            char* nxt = lds + ((kt + 1) & 1) * 2 * TILE_BYTES;
            stage_rows(kbase, ld, (kt + 1) * 64, T, nxt, wave, lane);
            stage_rows(vbase, ld, (kt + 1) * 64, T, nxt + TILE_BYTES, wave, lane);
        }
        if (!wave_active) continue;
        if (kt < nfull) attn_dq_tile_q<false, QT>(kt_lds, vt_lds, qf, dof, dq, c, lse, dl, 64, lane);
        else attn_dq_tile_q<true, QT>(kt_lds, vt_lds, qf, dof, dq, c, lse, dl, T - kt * 64, lane);
    }
    const int tiles16 = ((T + 63) >> 6) * 4;                 // partial rows per sequence of the colsum buffer (one per 16-query tile)
#pragma unroll
    for (int tq = 0; tq < QT; ++tq) {
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[tq][d] *= p.scale;
        const int q = q0 + tq * 16 + i;
        const int tile16 = (q0 >> 4) + tq;
        if (p.colsum && tile16 < tiles16) wave_colsum(dq[tq], q < T, p.colsum + ((size_t)b * tiles16 + tile16) * ld + h * HD, lane);
        if (q < T) {
            __bf16* dst = p.out + ((size_t)b * T + q) * ld + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 w = {f2bf(dq[tq][d][0]), f2bf(dq[tq][d][1]), f2bf(dq[tq][d][2]), f2bf(dq[tq][d][3])};
                *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: key on the lane, Q / dO tiles stream (each tile serves row reads AND transposed reads).
//   S[q][key] = Q . K^T ; dP[q][key] = dO . V^T ; P = exp2(c S - lse2[q]) ; dS = P (dP - delta[q]) scale
//   dV^T[d][key] += dO^T . P ;  dK^T[d][key] += Q^T . dS
// ------------------------------------------------------------------------------------------------
// One Q/dO tile of the dK/dV pass for a wave's 16 keys (RAGGED as in attn_dq_tile; dS unscaled, `scale` goes on dK once).
// Issue order, pinned with sched_barrier(0) between the sections (hipcc otherwise sinks every LDS read next to its MFMA to
// save registers: `2 x ds_read_b64_tr -> s_waitcnt -> v_mfma`, 16 times per tile, i.e. 16 exposed LDS latencies that two
// waves per SIMD cannot hide -- measured ~3,500 cycles per tile for 512 cycles of MFMA):
//   A  all 16 row fragments (Q and dO rows)            B  16 MFMAs (S, dP) behind counted lgkmcnt waits
//   C  all 16 transposed fragments (dO^T, Q^T) issued   D  the VALU part (P, dS) runs while they are in flight
//   E  16 MFMAs (dV, dK)
// PIN = false leaves the order to hipcc (the streaming kernel: 160 VGPRs, three workgroups per CU; pinned it needs 220 and
// drops to two, measured 1,252 -> 1,330 us at T = 513); the resident kernel has the registers and pins (-2 %).
#define PIN_BARRIER() do { if (PIN) __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool RAGGED, bool PIN = false>
__device__ __forceinline__ void attn_dkdv_tile(const char* q_lds, const char* do_lds, const float* st, const bf16x8 (&kf)[2],
                                               const bf16x8 (&vf)[2], f32x4 (&dk)[4], f32x4 (&dv)[4], float c, int nvalid,
                                               int lane) {
    if constexpr (!PIN) {
        const int g = lane >> 4;
        f32x4 s[4], dp[4];
    #pragma unroll
        for (int q16 = 0; q16 < 4; ++q16) {
            s[q16] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[q16] = *reinterpret_cast<const f32x4*>(st + 64 + q16 * 16 + 4 * g);      // -delta[q]: the chain's initial value
            if (!RAGGED || q16 * 16 < nvalid) {
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s[q16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(q_lds, q16, ks, lane), kf[ks], s[q16], 0, 0, 0);
                    dp[q16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(do_lds, q16, ks, lane), vf[ks], dp[q16], 0, 0, 0);
                }
            }
        }
        f32x4 pr[4];
    #pragma unroll
        for (int q16 = 0; q16 < 4; ++q16) {
            const f32x4 lse4 = *reinterpret_cast<const f32x4*>(st + q16 * 16 + 4 * g);
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[q16][r], c, -lse4[r]));
                if (RAGGED && q16 * 16 + 4 * g + r >= nvalid) pv = 0.f;
                pr[q16][r] = pv;
                s[q16][r] = pv * dp[q16][r];
            }
        }
        bf16x8 pb[2] = {pack_pair(pr[0], pr[1]), pack_pair(pr[2], pr[3])};
        bf16x8 db[2] = {pack_pair(s[0], s[1]), pack_pair(s[2], s[3])};
    #pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            if (RAGGED && ss * 32 >= nvalid) continue;
    #pragma unroll
            for (int d = 0; d < 4; ++d) {
                dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(do_lds, ss, d, lane), pb[ss], dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(q_lds, ss, d, lane), db[ss], dk[d], 0, 0, 0);
            }
        }
    
        return;
    }
    const int g = lane >> 4;
    f32x4 s[4], dp[4];
    bf16x8 fq[4][2], fd[4][2];
    PIN_BARRIER();
#pragma unroll
    for (int q16 = 0; q16 < 4; ++q16) {
        if (!RAGGED || q16 * 16 < nvalid) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[q16][ks] = frag_rows(q_lds, q16, ks, lane);
                fd[q16][ks] = frag_rows(do_lds, q16, ks, lane);
            }
        }
    }
    PIN_BARRIER();
#pragma unroll
    for (int q16 = 0; q16 < 4; ++q16) {
        s[q16] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[q16] = *reinterpret_cast<const f32x4*>(st + 64 + q16 * 16 + 4 * g);          // -delta[q]: the chain's initial value
        if (!RAGGED || q16 * 16 < nvalid) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s[q16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[q16][ks], kf[ks], s[q16], 0, 0, 0);
                dp[q16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[q16][ks], vf[ks], dp[q16], 0, 0, 0);
            }
        }
    }
    PIN_BARRIER();
    bf16x8 td[2][4], tq[2][4];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (RAGGED && ss * 32 >= nvalid) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            td[ss][d] = frag_tr(do_lds, ss, d, lane);
            tq[ss][d] = frag_tr(q_lds, ss, d, lane);
        }
    }
    PIN_BARRIER();
    f32x4 pr[4];
#pragma unroll
    for (int q16 = 0; q16 < 4; ++q16) {
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(st + q16 * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[q16][r], c, -lse4[r]));
            if (RAGGED && q16 * 16 + 4 * g + r >= nvalid) pv = 0.f;
            pr[q16][r] = pv;
            s[q16][r] = pv * dp[q16][r];
        }
    }
    bf16x8 pb[2] = {pack_pair(pr[0], pr[1]), pack_pair(pr[2], pr[3])};
    bf16x8 db[2] = {pack_pair(s[0], s[1]), pack_pair(s[2], s[3])};
    PIN_BARRIER();
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
        if (RAGGED && ss * 32 >= nvalid) continue;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td[ss][d], pb[ss], dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[ss][d], db[ss], dk[d], 0, 0, 0);
        }
    }
    PIN_BARRIER();
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dkdv_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * (2 * TILE_BYTES + 512)];   // [buf][Q|dO|lse,delta]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    int kb, h, b;
    if (!attn_block(p, kb, h, b)) return;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* qbase = seq + h * HD;
    const __bf16* dobase = p.dout + (size_t)b * T * D + h * HD;
    const float* lse_b = p.lse2 + ((size_t)b * p.H + h) * T;
    const float* del_b = p.delta + ((size_t)b * p.H + h) * T;
    const float c = p.scale * 1.4426950408889634f;
    const int BUF = 2 * TILE_BYTES + 512;

    const int key = kb * 64 + wave * 16 + i;
    const int kc = key < T ? key : T - 1;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)kc * ld + D + h * HD + ks * 32 + g * 8);
        vf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)kc * ld + 2 * D + h * HD + ks * 32 + g * 8);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dk[d] = dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool wave_active = kb * 64 + wave * 16 < T;    // wave-uniform: a wave with no valid key does no math

    const int nqt = (T + 63) / 64;
    // stats for tile qt live in floats [0,64) = lse2, [64,128) = -delta of the buffer
    float stat_next = 0.f;
    {
        if (t < 128) { int qq = (t & 63); qq = qq < T ? qq : T - 1; stat_next = t < 64 ? lse_b[qq] : -del_b[qq]; }
        stage_rows(qbase, ld, 0, T, lds, wave, lane);
        stage_rows(dobase, (size_t)D, 0, T, lds + TILE_BYTES, wave, lane);
        if (t < 128) reinterpret_cast<float*>(lds + 2 * TILE_BYTES)[t] = stat_next;
    }
    for (int qt = 0; qt < nqt; ++qt) {
        STAGE_FENCE();
        const char* cur = lds + (qt & 1) * BUF;
        const char* q_lds = cur;
        const char* do_lds = cur + TILE_BYTES;
        const float* st = reinterpret_cast<const float*>(cur + 2 * TILE_BYTES);
        char* nxt = lds + ((qt + 1) & 1) * BUF;
        const bool more = qt + 1 < nqt;
        if (more) {
            if (t < 128) {
                int qq = (qt + 1) * 64 + (t & 63);
                qq = qq < T ? qq : T - 1;
                stat_next = t < 64 ? lse_b[qq] : -del_b[qq];
            }
            stage_rows(qbase, ld, (qt + 1) * 64, T, nxt, wave, lane);
            stage_rows(dobase, (size_t)D, (qt + 1) * 64, T, nxt + TILE_BYTES, wave, lane);
        }
        if (wave_active) {
            if ((qt + 1) * 64 <= T) attn_dkdv_tile<false>(q_lds, do_lds, st, kf, vf, dk, dv, c, 64, lane);
            else attn_dkdv_tile<true>(q_lds, do_lds, st, kf, vf, dk, dv, c, T - qt * 64, lane);
        }
        if (more && t < 128) reinterpret_cast<float*>(nxt + 2 * TILE_BYTES)[t] = stat_next;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) dk[d] *= p.scale;
    if (p.colsum) {
        float* row = p.colsum + (((size_t)b * ((T + 63) >> 6) + kb) * 4 + wave) * ld + h * HD;
        wave_colsum(dk, key < T, row + D, lane);
        wave_colsum(dv, key < T, row + 2 * D, lane);
    }
    if (key < T) {
        __bf16* dst = p.out + ((size_t)b * T + key) * ld + h * HD;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 wk = {f2bf(dk[d][0]), f2bf(dk[d][1]), f2bf(dk[d][2]), f2bf(dk[d][3])};
            bf16x4 wv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
            *reinterpret_cast<bf16x4*>(dst + D + d * 16 + 4 * g) = wk;
            *reinterpret_cast<bf16x4*>(dst + 2 * D + d * 16 + 4 * g) = wv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward for short sequences (T <= RES_ROWS), same idea as attn_fwd_res_kernel: one workgroup per (b,h), the
// streamed operand pair resident in LDS, whole 16-row tiles per wave, one barrier in the whole kernel.
// ------------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void res_stage_pair(const __bf16* __restrict__ src0, size_t ld0, const __bf16* __restrict__ src1,
                                               size_t ld1, char* lds0, char* lds1, int T, int wave, int lane) {
    const int npieces = ((T + 31) >> 5) * 4;        // whole 32-row contraction steps (rows past T repeat row T-1: finite data)
    const int rin = lane >> 3, chunk = (lane & 7) ^ rin;
    for (int pc = wave; pc < 2 * npieces; pc += NW) {
        const int second = pc >= npieces, piece = second ? pc - npieces : pc;
        int r = piece * 8 + rin;
        r = r < T ? r : T - 1;
        const __bf16* src = second ? src1 + (size_t)r * ld1 : src0 + (size_t)r * ld0;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + chunk * 8), LDS_PTR((second ? lds1 : lds0) + piece * 1024), 16, 0, 0);
    }
}

__global__ __launch_bounds__(RES_NW * 64, 4) void attn_bwd_dq_res_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * RES_ROWS * 128];          // K rows | V rows
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const float c = p.scale * 1.4426950408889634f;
    char* klds = lds;
    char* vlds = lds + RES_ROWS * 128;
    res_stage_pair<RES_NW>(seq + D + h * HD, ld, seq + 2 * D + h * HD, ld, klds, vlds, T, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    const int nqt = (T + 15) >> 4, nkt = (T + 63) / 64, nfull = T / 64;
    const int w0 = (wave + (bh >> 3)) & (RES_NW - 1);
    f32x4 qsum[4];                                   // p.colsum: column sums of this wave's (rounded) dq rows
#pragma unroll
    for (int d = 0; d < 4; ++d) qsum[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int qt = w0; qt < nqt; qt += RES_NW) {
        const int q = qt * 16 + i;
        const int qc = q < T ? q : T - 1;
        bf16x8 qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)qc * ld + h * HD + ks * 32 + g * 8);
            dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + ((size_t)b * T + qc) * D + h * HD + ks * 32 + g * 8);
        }
        const size_t stat = ((size_t)b * p.H + h) * T + qc;
        const float lse = p.lse2[stat];
        // delta[q] = sum_d dO[q][d] * O[q][d], computed here from the dO fragments this wave holds anyway and published for
        // the dK/dV kernel (which runs after this one): no separate attn_delta_kernel pass on the resident path
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(p.o + ((size_t)b * T + qc) * D + h * HD + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += bf2f(dof[ks][e]) * bf2f(of[e]);
        }
        dl = xor_lanes_sum(dl);
        if (g == 0 && q < T) p.delta[stat] = dl;
        f32x4 dq[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nfull; ++kt)
            attn_dq_tile<false>(klds + kt * TILE_BYTES, vlds + kt * TILE_BYTES, qf, dof, dq, c, lse, dl, 64, lane);
        if (nfull < nkt)
            attn_dq_tile<true>(klds + nfull * TILE_BYTES, vlds + nfull * TILE_BYTES, qf, dof, dq, c, lse, dl, T - nfull * 64, lane);
        if (q < T) {
            __bf16* dst = p.out + ((size_t)b * T + q) * ld + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 w = {f2bf(dq[d][0] * p.scale), f2bf(dq[d][1] * p.scale), f2bf(dq[d][2] * p.scale), f2bf(dq[d][3] * p.scale)};
                *reinterpret_cast<bf16x4*>(dst + d * 16 + 4 * g) = w;
                if (p.colsum) qsum[d] += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
            }
        }
    }
    // qkv bias gradient: one partial row per (sequence, wave) -- RES_COLSUM_ROWS rows per sequence, this head's 64 columns
    if (p.colsum) wave_colsum<false>(qsum, true, p.colsum + ((size_t)b * RES_COLSUM_ROWS + wave) * ld + h * HD, lane);
}

#define RES_DKDV_NW 4         // 4 waves x ~160 VGPRs (no spills), two workgroups per CU
__global__ __launch_bounds__(RES_DKDV_NW * 64, 2) void attn_bwd_dkdv_res_kernel(AttnParams p) {
    // Q rows | dO rows | per 64-query tile: lse2[64], delta[64]
    __shared__ __attribute__((aligned(1024))) char lds[2 * RES_ROWS * 128 + (RES_ROWS / 64 + 1) * 512];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int T = p.T, D = p.D;
    const size_t ld = (size_t)3 * D;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const float c = p.scale * 1.4426950408889634f;
    char* qlds = lds;
    char* dolds = lds + RES_ROWS * 128;
    float* stats = reinterpret_cast<float*>(lds + 2 * RES_ROWS * 128);
    res_stage_pair<RES_DKDV_NW>(seq + h * HD, ld, p.dout + (size_t)b * T * D + h * HD, (size_t)D, qlds, dolds, T, wave, lane);
    const int nqt = (T + 63) / 64, nfull = T / 64;
    {
        const float* lse_b = p.lse2 + ((size_t)b * p.H + h) * T;
        const float* del_b = p.delta + ((size_t)b * p.H + h) * T;
        for (int e = t; e < nqt * 64; e += RES_DKDV_NW * 64) {
            const int qq = e < T ? e : T - 1;
            stats[(e >> 6) * 128 + (e & 63)] = lse_b[qq];
            stats[(e >> 6) * 128 + 64 + (e & 63)] = -del_b[qq];          // the dP chains start from -delta (attn_dkdv_tile)
        }
    }
    const int nkt16 = (T + 15) >> 4;
    const int w0 = (wave + (bh >> 3)) & (RES_DKDV_NW - 1);
    f32x4 ksum[4], vsum[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) ksum[d] = vsum[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's key fragments are fetched one tile AHEAD (the first ones while the Q / dO pieces are still in flight):
    // at two waves per SIMD a load at the head of each tile is ~1 us of exposed latency per tile
    auto load_kv = [&](int kt, bf16x8 (&kf)[2], bf16x8 (&vf)[2]) {
        int kc = kt * 16 + i;
        kc = kc < T ? kc : T - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)kc * ld + D + h * HD + ks * 32 + g * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(seq + (size_t)kc * ld + 2 * D + h * HD + ks * 32 + g * 8);
        }
    };
    bf16x8 kf[2], vf[2], kfn[2], vfn[2];
    load_kv(w0, kf, vf);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    for (int kt = w0; kt < nkt16; kt += RES_DKDV_NW) {
        const int key = kt * 16 + i;
        if (kt + RES_DKDV_NW < nkt16) load_kv(kt + RES_DKDV_NW, kfn, vfn);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) dk[d] = dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt < nfull; ++qt)
            attn_dkdv_tile<false, true>(qlds + qt * TILE_BYTES, dolds + qt * TILE_BYTES, stats + qt * 128, kf, vf, dk, dv, c, 64, lane);
        if (nfull < nqt)
            attn_dkdv_tile<true, true>(qlds + nfull * TILE_BYTES, dolds + nfull * TILE_BYTES, stats + nfull * 128, kf, vf, dk, dv, c,
                                       T - nfull * 64, lane);
        if (key < T) {
            __bf16* dst = p.out + ((size_t)b * T + key) * ld + h * HD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 wk = {f2bf(dk[d][0] * p.scale), f2bf(dk[d][1] * p.scale), f2bf(dk[d][2] * p.scale), f2bf(dk[d][3] * p.scale)};
                bf16x4 wv = {f2bf(dv[d][0]), f2bf(dv[d][1]), f2bf(dv[d][2]), f2bf(dv[d][3])};
                *reinterpret_cast<bf16x4*>(dst + D + d * 16 + 4 * g) = wk;
                *reinterpret_cast<bf16x4*>(dst + 2 * D + d * 16 + 4 * g) = wv;
                if (p.colsum) {
                    ksum[d] += f32x4{bf2f(wk[0]), bf2f(wk[1]), bf2f(wk[2]), bf2f(wk[3])};
                    vsum[d] += f32x4{bf2f(wv[0]), bf2f(wv[1]), bf2f(wv[2]), bf2f(wv[3])};
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { kf[ks] = kfn[ks]; vf[ks] = vfn[ks]; }
    }
    if (p.colsum) {
        float* row = p.colsum + ((size_t)b * RES_COLSUM_ROWS + wave) * ld + h * HD;
        wave_colsum<false>(ksum, true, row + D, lane);
        wave_colsum<false>(vsum, true, row + 2 * D, lane);
        if (lane < 32) {                                 // rows 4..7 of the dk / dv columns: zeros (the dq kernel has 8 waves)
            float* z = p.colsum + ((size_t)b * RES_COLSUM_ROWS + RES_DKDV_NW + wave) * ld + h * HD + (1 + (lane >> 4)) * D + (lane & 15) * 4;
            *reinterpret_cast<f32x4*>(z) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

#include "attention_lone.h"
#include "attention_fused.h"
#include "attention_fused2.h"
#include "attention_fwd2.h"

// experiment switch for tools/attn_ab.py (in-process A/B, default 0 = automatic): bit 0 = no resident forward / dQ,
// bit 1 = no resident dK/dV, bit 2 = no two-pass (64 n + 1) forward, bit 3 = no fused backward, bit 5 = no start shift of the
// fused / key-stationary backward, bit 4 = no fused backward at T = 513 (attention_fused2.h), bit 6 = no key-stationary dK/dV at T = 513, bit 7 = a workgroup per pair instead of the walk,
// bits 8-9 = start-shift phases - 1 and bits 10.. = shift in 1024-cycle units (sweeps).  Nothing in the product path sets it.
static int g_attn_variant = 0;
extern "C" int p4p_set_attention_variant(int v) { int o = g_attn_variant; g_attn_variant = v; return o; }
// forward-only switch (tools/attn_fwd2_ab.py, tools/attn_fwd_thr_ab.py): bit 3 = exact running maximum (no lazy rescale) in the online kernels; bit 0 = the K/V-resident forward at T = 513 (attention_fwd2.h: measured 380-390 us against the
// 360 us of attn_fwd_q_kernel<2>, NOTES.md A.6 -- not the default), bit 1 = a workgroup per pair instead of the walk, bit 2 = no start shift
static int g_attn_fwd_variant = 0;
extern "C" int p4p_set_attention_fwd_variant(int v) { int o = g_attn_fwd_variant; g_attn_fwd_variant = v; return o; }

static int check_attn(const char* fn, int B, int T, int H, int D) {
    P4P_CHECK(B >= 1 && T >= 1 && H >= 1, "%s: bad sizes B=%d T=%d H=%d", fn, B, T, H);
    P4P_CHECK(D == H * HD, "%s: only head_dim 64 is built (D=%d, H=%d)", fn, D, H);
    P4P_CHECK((long long)B * H * ceil_div(T, 64) < (1ll << 30), "%s: grid limit", fn);
    return 0;
}

extern "C" int p4p_attention_fwd(const void* qkv, void* out, float* lse2, int B, int T, int H, int D, float scale,
                                 void* stream) {
    P4P_CHECK(qkv && out, "p4p_attention_fwd: null pointer");
    if (check_attn("p4p_attention_fwd", B, T, H, D)) return 1;
    AttnParams p = {};
    p.qkv = (const __bf16*)qkv; p.out = (__bf16*)out; p.lse2 = lse2; p.B = B; p.T = T; p.H = H; p.D = D; p.scale = scale;
    const bool prof = p4p_prof_begin(3, 4.0 * B * H * (double)T * T * HD, (hipStream_t)stream);      // bench.py: attention forward
    const bool no_res = g_attn_variant & 1, no_lone = g_attn_variant & 4;
    p.thr = (g_attn_fwd_variant & 8) ? 0.0f : ATT_LAZY_THR;
    if (T % 64 == 1 && T >= 65 && T <= 257 && !no_lone) {                    // 64 n + 1 rows: attention_lone.h
        const dim3 grid(B * H), block(LONE_NW * 64);
        switch ((T - 1) / 64) {
            case 1: hipLaunchKernelGGL(attn_fwd_lone_res_kernel<1>, grid, block, 0, (hipStream_t)stream, p); break;
            case 2: hipLaunchKernelGGL(attn_fwd_lone_res_kernel<2>, grid, block, 0, (hipStream_t)stream, p); break;
            case 3: hipLaunchKernelGGL(attn_fwd_lone_res_kernel<3>, grid, block, 0, (hipStream_t)stream, p); break;
            default: hipLaunchKernelGGL(attn_fwd_lone_res_kernel<4>, grid, block, 0, (hipStream_t)stream, p); break;
        }
    } else if (T <= RES_ROWS - 31 && T >= 64 && !no_res)
        hipLaunchKernelGGL(attn_fwd_res_kernel, dim3(B * H), dim3(RES_NW * 64), 0, (hipStream_t)stream, p);
    else if (T == 513 && (g_attn_fwd_variant & 1)) {                        // K and V of the sequence resident: attention_fwd2.h
        const int pg = p4p_persistent_grid();
        if (pg > 0 && B * H > pg && !(g_attn_fwd_variant & 2)) {
            if (!(g_attn_fwd_variant & 4)) { p.stagger_n = pg; p.stagger_phases = 3; p.stagger_cycles = 6 * 1024; }
            hipLaunchKernelGGL((attn_fwd_kv_kernel<16, true>), dim3(pg), dim3(512), 0, (hipStream_t)stream, p);
        } else
            hipLaunchKernelGGL((attn_fwd_kv_kernel<16, false>), dim3(B * H), dim3(512), 0, (hipStream_t)stream, p);
    } else {
        // measured at B=256, H=12 (tools/attn_ab.py): T=513: 542 / 499 / 533 us for 1 / 2 / 4 tiles per wave; T=2049: 5398 / 4643 / 4501
        // round 4, with the lazy rescale (tools/attn_fwd_thr_ab.py): T=513: 337 / 406 us for 2 / 4 tiles; T=2049, B=8, H=16 (128 pairs): 175.5 / 184.3 us --
        // four tiles per wave pay only when their 256-row workgroups still fill the chip many times over (B=256: 4.50 against 4.64 ms)
        const int ncus = p4p_device_cu_count();                    // -1 if the runtime cannot say: treated as 'not many'
        const bool many = ncus > 0 && (long long)B * H * ceil_div(T, 256) >= 16ll * ncus;
        const int qt = (g_attn_fwd_variant & 48) ? ((g_attn_fwd_variant & 16) ? 4 : 2) : (T >= 1024 && many ? 4 : 2);     // bits 4 / 5: force 4 / 2 (A/B)
        if (qt == 2 && T > 128)
            hipLaunchKernelGGL(attn_fwd_q_kernel<2>, dim3((unsigned)(ceil_div((long long)B * H, 8) * 8 * ceil_div(T, 128))),
                               dim3(ATT_THREADS), 0, (hipStream_t)stream, p);
        else if (qt == 4 && T > 256)
            hipLaunchKernelGGL(attn_fwd_q_kernel<4>, dim3((unsigned)(ceil_div((long long)B * H, 8) * 8 * ceil_div(T, 256))),
                               dim3(ATT_THREADS), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(attn_fwd_kernel, attn_grid(B, T, H), dim3(ATT_THREADS), 0, (hipStream_t)stream, p);
    }
    if (prof) p4p_prof_end((hipStream_t)stream);
    P4P_LAUNCH_CHECK();
    return 0;
}

// dqkv receives dq | dk | dv in the layout of qkv.  delta is [B,H,T] workspace.
static bool attn_resident(int T) {
    return T <= RES_ROWS - 31 && T >= 64 && !(g_attn_variant & 1);
}
static bool attn_fused(int T) {           // attention_fused.h; needs the resident kernels' colsum layout (RES_COLSUM_ROWS rows per sequence)
    return !(g_attn_variant & 8) && attn_resident(T) && !(g_attn_variant & 2) && (T == 65 || T == 129 || T == 257);
}
static bool attn_resident_dkdv(int T) {
    return !(g_attn_variant & 2) && attn_resident(T);       // 4 waves x 220 VGPRs: -0.4 ms/step
}
static bool attn_fused2(int T) {          // attention_fused2.h: T = 513, walking workgroups (needs persistent kernels)
    return T == 513 && !(g_attn_variant & (64 | 16)) && p4p_persistent_grid() > 0;
}
// floats the `delta` workspace of p4p_attention_bwd must hold for this shape under the CURRENT switches (reserved CUs, variants):
// B*H*T for the kernels that keep delta in memory; the fused T = 513 kernel parks dQ^T chains there instead (per workgroup
// F2Lds<16>::ROWS x 64 floats), whichever is larger
extern "C" long long p4p_attention_bwd_workspace(int B, int T, int H) {
    long long n = (long long)B * H * T;
    if (attn_fused2(T)) {
        const int pgrid = p4p_persistent_grid();
        const long long nwg = (long long)B * H < pgrid ? (long long)B * H : pgrid;
        const long long park = nwg * (F2Lds<16>::ROWS * 64);
        if (park > n) n = park;
    }
    return n;
}
// the partial-row layouts of the streaming and the resident kernels differ: dq and dk/dv take the same path when sums are asked for
extern "C" int p4p_attention_colsum_rows(int B, int T) { return attn_resident_dkdv(T) ? B * RES_COLSUM_ROWS : B * ceil_div(T, 64) * 4; }

// colsum_partial (may be null): [p4p_attention_colsum_rows(B,T)][3D] floats; their column sum (p4p_colreduce_f32) is
// the column sum of dqkv = the gradient of the qkv bias.
extern "C" int p4p_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, float* delta,
                                 void* dqkv, float* colsum_partial, int B, int T, int H, int D, float scale,
                                 void* stream) {
    P4P_CHECK(qkv && o && dout && lse2 && delta && dqkv, "p4p_attention_bwd: null pointer");
    if (check_attn("p4p_attention_bwd", B, T, H, D)) return 1;
    hipStream_t st = (hipStream_t)stream;
    // bench.py: the whole backward (delta + dQ + dK/dV launches) as one bracket, 2.5 x the forward's FLOPs
    struct ProfScope {
        hipStream_t s; bool on;
        ProfScope(double fl, hipStream_t s_) : s(s_), on(p4p_prof_begin(4, fl, s_)) {}
        ~ProfScope() { if (on) p4p_prof_end(s); }
    } prof_scope(10.0 * B * H * (double)T * T * HD, st);
    AttnParams p = {};
    p.qkv = (const __bf16*)qkv; p.o = (const __bf16*)o; p.dout = (const __bf16*)dout; p.out = (__bf16*)dqkv;
    p.lse2 = (float*)lse2; p.delta = delta; p.colsum = colsum_partial; p.B = B; p.T = T; p.H = H; p.D = D; p.scale = scale;
    if (attn_fused(T)) {                                 // T = 32 NT + 1 <= 257: one kernel, attention_fused.h
        const int pg = p4p_persistent_grid();            // CUs of the current device - p4p_set_reserved_cus(); 0: no walking kernels
        const int ncu = pg > 0 ? pg : 256;
        const dim3 grid(B * H);
        p.stagger_n = ncu; p.stagger_phases = 3; p.stagger_cycles = 10 * 1024;      // attention_fused.h: start-time shift of the first round
        if (g_attn_variant & 32) p.stagger_phases = 1;   // bit 5: no shift
        if (g_attn_variant >> 8) {                       // experiment: bits 8-9 phases - 1, bits 10.. shift in 1024-cycle units
            p.stagger_phases = ((g_attn_variant >> 8) & 3) + 1; p.stagger_cycles = (g_attn_variant >> 10) * 1024;
        }
        switch ((T - 1) / 32) {
            case 2: hipLaunchKernelGGL(attn_bwd_fused_kernel<2>, grid, dim3(2 * 64), 0, st, p); break;
            case 4: hipLaunchKernelGGL(attn_bwd_fused_kernel<4>, grid, dim3(4 * 64), 0, st, p); break;
            default:
                // T = 257 (112 KB of LDS: one workgroup per CU anyway): ONE workgroup per CU walks its pairs (same body, a barrier
                // between pairs; same box, B = 256, H = 12: 284-288 us against 304-310 us for a workgroup per pair, results
                // bit-identical) -- no dispatch between pairs, and the start-time shift of the first round stays what it was set to.
                // Bit 7 of the variant: a workgroup per pair.
                if (!(g_attn_variant & 128) && pg > 0 && B * H > ncu)
                    hipLaunchKernelGGL((attn_bwd_fused_kernel<8, true>), dim3(ncu), dim3(8 * 64), 0, st, p);
                else hipLaunchKernelGGL(attn_bwd_fused_kernel<8>, grid, dim3(8 * 64), 0, st, p);
                break;
        }
        P4P_LAUNCH_CHECK();
        return 0;
    }
    const bool res_dkdv = attn_resident_dkdv(T);
    const bool res = attn_resident(T) && (res_dkdv || !colsum_partial);
    const bool dq_q = !res && T > 128;                   // two query tiles per wave
    if (!res && !dq_q) {                                 // the resident and the two-tile dQ kernels compute delta themselves
        hipLaunchKernelGGL(attn_delta_kernel, dim3(ceil_div((long long)B * T, 4)), dim3(256), 0, st, (const __bf16*)o,
                           (const __bf16*)dout, delta, B, T, H, D);
        P4P_LAUNCH_CHECK();
    }
    dim3 grid = attn_grid(B, T, H);
    if (attn_fused2(T)) {
        // attention_fused2.h (round 4): dQ, dK, dV in ONE kernel -- a workgroup walks its pairs and takes the two key halves of a pair
        // one after the other with Q / dO resident; half 0's dQ^T chains wait in a per-workgroup slice (139 KB) of the CALLER's
        // workspace (`delta`, sized by p4p_attention_bwd_workspace: the kernel computes delta itself and keeps it on chip).  Round 5:
        // the parking space used to be a library-owned, function-static hipMalloc shared by every stream.  Bit 4 of the variant: the
        // two kernels below.
        const int pgrid = p4p_persistent_grid();
        const int nwg = B * H < pgrid ? B * H : pgrid;
        float* const parking = delta;
        p.stagger_n = nwg; p.stagger_phases = 3; p.stagger_cycles = 12 * 1024;
        if (g_attn_variant & 32) p.stagger_phases = 1;
        if (g_attn_variant >> 8) { p.stagger_phases = ((g_attn_variant >> 8) & 3) + 1; p.stagger_cycles = (g_attn_variant >> 10) * 1024; }
        hipLaunchKernelGGL((attn_bwd_fused2_kernel<16, true>), dim3(nwg), dim3(512), 0, st, p, parking);
        P4P_LAUNCH_CHECK();
        return 0;
    }
    if (T == 513 && !(g_attn_variant & 64)) {            // streaming dQ (+ delta), then key-stationary dK / dV with Q, dO resident: attention_fused.h
        const int pg = p4p_persistent_grid();
        const int ncu = pg > 0 ? pg : 256;
        p.stagger_n = ncu; p.stagger_phases = 3; p.stagger_cycles = 12 * 1024;
        if (g_attn_variant & 32) p.stagger_phases = 1;
        if (g_attn_variant >> 8) { p.stagger_phases = ((g_attn_variant >> 8) & 3) + 1; p.stagger_cycles = (g_attn_variant >> 10) * 1024; }
        hipLaunchKernelGGL(attn_bwd_dq_q_kernel<2>, dim3((unsigned)(ceil_div((long long)B * H, 8) * 8 * ceil_div(T, 128))), dim3(ATT_THREADS), 0, st, p);
        P4P_LAUNCH_CHECK();
        if (!(g_attn_variant & 128) && pg > 0 && B * H * 2 > ncu)   // one workgroup per CU walks its work items (bit 7: a workgroup per item)
            hipLaunchKernelGGL((attn_bwd_dkdv_ks_kernel<16, true>), dim3(ncu), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(attn_bwd_dkdv_ks_kernel<16>, dim3(B * H * 2), dim3(512), 0, st, p);
        P4P_LAUNCH_CHECK();
        return 0;
    }
    if (res) hipLaunchKernelGGL(attn_bwd_dq_res_kernel, dim3(B * H), dim3(RES_NW * 64), 0, st, p);
    else if (dq_q)
        hipLaunchKernelGGL(attn_bwd_dq_q_kernel<2>, dim3((unsigned)(ceil_div((long long)B * H, 8) * 8 * ceil_div(T, 128))),
                           dim3(ATT_THREADS), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(ATT_THREADS), 0, st, p);
    P4P_LAUNCH_CHECK();
    if (res && res_dkdv) {
        hipLaunchKernelGGL(attn_bwd_dkdv_res_kernel, dim3(B * H), dim3(RES_DKDV_NW * 64), 0, st, p);
        P4P_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(ATT_THREADS), 0, st, p);
    P4P_LAUNCH_CHECK();
    return 0;
}
