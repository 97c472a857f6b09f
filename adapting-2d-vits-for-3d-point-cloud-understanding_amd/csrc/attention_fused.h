// Fused attention backward for the model's own short sequences, T = 32 NT + 1 <= 257 (N = 1024 points: 256 point tokens + cls,
// src/models/pix4point.py:248-252): ONE kernel, one workgroup per (batch, head), five MFMA products per score tile instead
// of the seven of attn_bwd_dq_res_kernel + attn_bwd_dkdv_res_kernel (which both recompute S and dP), P = exp2(.) evaluated
// once instead of twice, q | k | v | dO read once instead of twice.
//
// Decomposition (key-stationary, the structure of cdna_hip_programming.md 'Attention backward'):
//   * NT waves; wave w owns keys 32 w .. 32 w + 31 for the whole kernel: their K / V fragments (B operands) and the fp32
//     accumulators dK^T, dV^T [64 d][32 keys] live in registers (2 x 32 VGPRs) -- no sum across workgroups, no atomics;
//   * Q and dO of the whole sequence sit in LDS (one image each serves row reads and transposed reads); the loop walks the
//     NT + 1 query tiles of 32 rows (the last one holds the odd row T - 1 and 31 padding rows whose lse is +inf, so that
//     their P and dS are exactly 0);
//   * per query tile and wave, v_mfma_f32_32x32x16_bf16 with the KEY on the lane:
//         S[q][key]  = Q . K^T              dP[q][key] = dO . V^T - delta[q]   (chain starts from -delta: no subtraction)
//         P = exp2(c S - lse2[q])           dS = P dP
//         dV^T[d][key] += dO^T . P          dK^T[d][key] += Q^T . dS           (the accumulator tile IS the B operand)
//   * dQ sums over the keys, i.e. over the waves: every wave stores its dS tile (bf16, 2 KiB) to an LDS image
//     [key][32 q], ONE barrier, then the 8 output tiles dQ^T[16 d][16 q] of the query tile are shared out over the waves,
//     each a chain of v_mfma_f32_16x16x32_bf16 over all keys with A = K^T fragments held in registers for the whole kernel
//     and B = transposed reads of the dS image; the image is double buffered, so the loop has one barrier per query tile;
//   * the odd KEY (row T - 1) has no wave of its own: before the loop wave w takes query tile w against it (the same MFMAs
//     with that key alone in column 0 of the B operands), leaves dS[q, T-1] in LDS -- a rank-1 term k_{T-1} dS[q, T-1] of
//     every dQ tile -- and its share of that key's dK / dV rows, which meet in LDS.
//
// Arithmetic contract = that of the two-kernel path (tests/emulate.py): P and dS are rounded to bf16 where they enter a
// product, everything else is fp32; delta = dO . O from the bf16 tensors; `scale` is applied once to dQ and dK.
// Reference: softmax(q k^T hd^-0.5) v of timm's Attention (src/models/apf_utils.py:142-158), autograd of it.
#pragma once

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Q / dO images: 128-B rows, 16-B chunk' = chunk ^ sig(row).  sig uses row bits 1, 2 and 1^4: conflict-free for the 32-row
// ds_read_b128 operand reads AND the 4-row-block transposed reads of the 32x32x16 products (searched with tools/lds_model.py;
// the `row & 7` of the 64-row tiles above is 2-way on both).
__device__ __forceinline__ int fz_sig(int row) { return ((row >> 1) & 3) | ((((row >> 1) ^ (row >> 4)) & 1) << 2); }
// dS image: [key][32 q] bf16 = 64-B rows of eight 8-B slots, slot' = slot ^ sigd(key): conflict-free ds_write_b64 of a wave's
// tile and conflict-free transposed reads of 8 consecutive keys.
__device__ __forceinline__ int fz_sigd(int key) { return (((key >> 2) & 1) << 2) | (((key >> 3) & 1) << 1) | ((key >> 1) & 1); }

// A operand of a 32x32x16 product: rows row0 .. row0 + 31 of an image, contraction columns 16 s .. 16 s + 15
__device__ __forceinline__ bf16x8 fz_rows(const char* img, int row0, int s, int lane) {
    const int row = row0 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(img + row * 128 + (((2 * s + (lane >> 5)) ^ fz_sig(row)) << 4));
}
// A operand, transposed: output row = image column 32 dh + (lane & 31); contraction slot (h, j) = image row
// row0 + 16 sp + 8 (j >> 2) + 4 h + (j & 3) -- the k order of registers 8 sp .. 8 sp + 7 of a 32x32 accumulator (guide section 3)
__device__ __forceinline__ bf16x8 fz_tr(const char* img, int row0, int sp, int dh, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const int row = row0 + 16 * sp + 4 * (G >> 1) + (i >> 2);       // bit 3 of the row is clear: row + 8 has the same sig
    const int p = i & 3;
    const int lch = 2 * (2 * dh + (G & 1)) + (p >> 1);
    const char* a0 = img + row * 128 + ((lch ^ fz_sig(row)) << 4) + (p & 1) * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * 128));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// B operand of the 16x16x32 dQ^T product from the dS image: column = query 16 a + (lane & 15); contraction slot (g, j) = key
// 32 ks + 16 (j >> 2) + 4 g + (j & 3), the order frag_tr() gives the K^T fragments
__device__ __forceinline__ bf16x8 fz_dst(const char* dsx, int ks, int a, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int key = 32 * ks + 4 * g + (i >> 2), p = i & 3;           // bit 4 of the key is clear: key + 16 has the same sigd
    const char* a0 = dsx + key * 64 + (((4 * a + p) ^ fz_sigd(key)) << 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 64));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}
// registers 8 sp .. 8 sp + 7 of a 32x32 accumulator as the bf16 B operand of the product that sums over its rows
__device__ __forceinline__ void fz_pack(const f32x16& x, bf16x8 (&out)[2]) {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
        out[sp] = bf16x8{f2bf(x[8 * sp + 0]), f2bf(x[8 * sp + 1]), f2bf(x[8 * sp + 2]), f2bf(x[8 * sp + 3]),
                         f2bf(x[8 * sp + 4]), f2bf(x[8 * sp + 5]), f2bf(x[8 * sp + 6]), f2bf(x[8 * sp + 7])};
}
// 16 per-row constants of query tile t in the accumulator layout: register 4 jj + r <-> row 8 jj + 4 h + r
__device__ __forceinline__ f32x16 fz_rowconst(const float* arr, int row0, int lane) {
    const int hh = lane >> 5;
    f32x16 v;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(arr + row0 + 8 * jj + 4 * hh);
        v[4 * jj + 0] = q[0]; v[4 * jj + 1] = q[1]; v[4 * jj + 2] = q[2]; v[4 * jj + 3] = q[3];
    }
    return v;
}

// Diagnostic build only (tools/attn_fused_stamps.py compiles attention.hip with -DP4P_FZ_STAMPS into its own library): s_memtime
// stamps of wave 0 at the phase boundaries, written into the (otherwise unused) delta workspace.  No stamp in the product library.
#ifdef P4P_FZ_STAMPS
#define FZ_STAMP(k)                                                                                       \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                      \
        if (t == 0) reinterpret_cast<unsigned long long*>(p.delta)[(size_t)blockIdx.x * 16 + (k)] = ts_;   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#else
#define FZ_STAMP(k) do { } while (0)
#endif

// Sources of the 1-KiB LDS-DMA pieces (8 rows x 128 B of one image) as a UNIFORM base (SGPRs) + a 32-bit per-lane byte offset, so
// that a piece costs a handful of scalar instructions (global_load_lds_dwordx4 v, s[base]): written with per-lane 64-bit
// pointers the 13-17 DMAs of a wave took ~35 instructions each (64-bit multiplies, a branch per image), ~5 % of a workgroup's life.
// Row r = 8 p + rin of a piece p: fz_sig(r) = ((rin >> 1) & 3) | ((((rin >> 1) ^ (p >> 1)) & 1) << 2), i.e. two lane offsets per
// row stride, picked by the parity of p >> 1; the piece behind row L (rows past T - 1 repeat row T - 1) has no row term.
struct FzSrc {
    unsigned par[2], last;                            // lane offsets: pieces with (p >> 1) even / odd; the piece of rows L .. L + 7
};
__device__ __forceinline__ FzSrc fz_src(int lane, unsigned row_bytes) {
    const int rin = lane >> 3;
    FzSrc o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int sig = ((rin >> 1) & 3) | ((((rin >> 1) ^ e) & 1) << 2);
        o.par[e] = rin * row_bytes + (((lane & 7) ^ sig) << 4);
    }
    o.last = o.par[0] - rin * row_bytes;               // L / 8 = 4 NT: (p >> 1) even
    return o;
}
__device__ __forceinline__ void fz_dma(const char* ubase, unsigned loff, char* dst) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(ubase + loff), LDS_PTR(dst), 16, 0, 0);
}

#define FZ_SLAB_STRIDE 272          // epilogue slab row: K 128 B | V 128 B | 16 B pad (bank spread of the 8-B writes)

template <int NT>
struct FzLds {
    static constexpr int L = 32 * NT;                 // index of the odd row = number of tiled rows
    static constexpr int ROWS = L + 32;               // rows of the Q / dO images (the last tile: row L + 31 padding rows)
    static constexpr int IMG = ROWS * 128;
    static constexpr int DSX = 2 * L * 64;            // two dS images; the K image of the prologue (L x 128 B) aliases them
    static constexpr int OFF_Q = 0, OFF_DO = IMG, OFF_DSX = 2 * IMG;
    static constexpr int OFF_NLSE = OFF_DSX + DSX;    // -lse2 [ROWS], -delta [ROWS], dS[q, L] scale [ROWS]
    static constexpr int OFF_NDEL = OFF_NLSE + ROWS * 4;
    static constexpr int OFF_DSL = OFF_NDEL + ROWS * 4;
    static constexpr int OFF_PART = OFF_DSL + ROWS * 4;             // [NT + 1][2][64] partial dK / dV rows of the odd key (waves, then the pair (L, L))
    static constexpr int OFF_LONE = OFF_PART + (NT + 1) * 2 * 64 * 4;   // [2][64] its rounded dK / dV rows (column sums)
    static constexpr int TOTAL = OFF_LONE + 2 * 64 * 4;
    static_assert(2 * IMG >= NT * 32 * FZ_SLAB_STRIDE, "epilogue slabs must fit the Q / dO images");
};

template <int NT, bool PERSIST = false>
__global__ __launch_bounds__(NT * 64, 1) void attn_bwd_fused_kernel(AttnParams p) {
    using LY = FzLds<NT>;
    constexpr int L = LY::L, ROWS = LY::ROWS;
    constexpr int TPW = 8 / NT;                       // dQ^T output tiles (16 d x 16 q) per wave and query tile
    constexpr int NKS = L / 32;                       // 32-key contraction steps of the dQ product
    static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "8 output tiles must divide over the waves");
    __shared__ __attribute__((aligned(1024))) char lds[LY::TOTAL];
    const int T = p.T, D = p.D;                       // T = L + 1
    const size_t ld = (size_t)3 * D;
    const float c = p.scale * 1.4426950408889634f;
    char* qs = lds + LY::OFF_Q;
    char* dos = lds + LY::OFF_DO;
    char* dsx = lds + LY::OFF_DSX;
    float* nlse = reinterpret_cast<float*>(lds + LY::OFF_NLSE);
    float* ndel = reinterpret_cast<float*>(lds + LY::OFF_NDEL);
    float* dsl = reinterpret_cast<float*>(lds + LY::OFF_DSL);
    float* part = reinterpret_cast<float*>(lds + LY::OFF_PART);
    float* lonekv = reinterpret_cast<float*>(lds + LY::OFF_LONE);

    // With one workgroup per CU and every workgroup alike, all CUs load at the same time and compute at the same time: the memory
    // system saturates, then idles.  The first round of workgroups therefore starts in `stagger_phases` groups `stagger_cycles`
    // apart, and the rounds that follow inherit the shift (measured at B=256, H=12, T=257, same call: no shift 316-337 us, 2 groups x
    // 14k cycles 311, 3 x 10k 297, 3 x 7k 300, 4 x 5k / 4 x 8k 311-313: more groups pay more at the tail).
    if (p.stagger_phases > 1 && (int)blockIdx.x < p.stagger_n) {
        const int ph = blockIdx.x % p.stagger_phases;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)ph * p.stagger_cycles) __builtin_amdgcn_s_sleep(16);
    }
    // (PERSIST, an experiment: one workgroup per CU walks its pairs with the SAME body, a barrier between pairs)
    for (int bh = blockIdx.x;; bh += gridDim.x) {
    // the thread index passes through an empty asm in every round: nothing derived from it is loop-invariant for hipcc, which would
    // otherwise hoist the ~60 lane-constant LDS bases and offsets of the body out of the walk and keep them live across it
    // (256 VGPRs + 51 spills, two scratch reloads in every query tile -- what made the persistent forms of round 3 1.7x slower)
    int t = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);          // scalar: tile picks, LDS-DMA destinations, output-tile ids stay in SGPRs
    const int b = bh / p.H, h = bh - b * p.H;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* dob = p.dout + (size_t)b * T * D + h * HD;
    const __bf16* ob = p.o + (size_t)b * T * D + h * HD;
    FZ_STAMP(0);
    // ---- stage Q, dO (1-KiB pieces of 8 rows; rows past L repeat row L) and K (64-row tiles, `row & 7` swizzle: frag_tr)
    {
        const int rin = lane >> 3;
        constexpr int NPC = L / 8 + 1;                 // pieces per image: rows 0 .. L + 7
        const unsigned qrow = (unsigned)ld * 2, drow = (unsigned)D * 2;          // bytes per row of qkv / of dO
        const FzSrc sq = fz_src(lane, qrow), sd = fz_src(lane, drow);
        const char* qbase = reinterpret_cast<const char*>(seq + h * HD);
        const char* dbase = reinterpret_cast<const char*>(dob);
        const unsigned lk = rin * qrow + (((lane & 7) ^ rin) << 4);
        const char* kbase = reinterpret_cast<const char*>(seq + D + h * HD);
        static_assert(NPC == L / 8 + 1, "");
#pragma unroll
        for (int i = 0; i < L / 8 / NT; ++i) {          // 4 pieces of Q, of dO and of K per wave: running uniform bases, no branch
            const int pc = wave + NT * i;
            fz_dma(qbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], qs + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < L / 8 / NT; ++i) {
            const int pc = wave + NT * i;
            fz_dma(dbase + (size_t)(8 * pc) * drow, ((pc >> 1) & 1) ? sd.par[1] : sd.par[0], dos + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < L / 8 / NT; ++i) {
            const int pc = wave + NT * i;
            fz_dma(kbase + (size_t)(8 * pc) * qrow, lk, dsx + pc * 1024);
        }
        if (wave == 0) fz_dma(qbase + (size_t)L * qrow, sq.last, qs + (L / 8) * 1024);              // rows L .. L + 7: row L eight times
        if (wave == NT - 1) fz_dma(dbase + (size_t)L * drow, sd.last, dos + (L / 8) * 1024);
        // rows L + 8 .. L + 31 of both images: zeros (finite operands for the padded rows of the last query tile)
        for (int e = t; e < 2 * 24 * 8; e += NT * 64) {
            char* img = e >= 24 * 8 ? dos : qs;
            const int ee = e >= 24 * 8 ? e - 24 * 8 : e;
            *reinterpret_cast<f32x4*>(img + (L + 8) * 128 + ee * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- this wave's keys: V fragments (B operand of dP) straight from global memory; the K fragments come out of the K image
    // below.  Every byte is fetched once: the load phase is bandwidth bound (all CUs load at the same time), so its length is
    // its byte count -- K read twice and dO read twice (image + delta) made it 228 KB instead of 163.
    const int kk = lane & 31, hh = lane >> 5;
    bf16x8 kf[4], vf[4];
    {
        const __bf16* vrow = seq + (size_t)(32 * wave + kk) * ld + 2 * D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) vf[s] = *reinterpret_cast<const bf16x8*>(vrow + 16 * s);
    }
    // ---- O rows of this wave's query tile (rows 32 w ..): delta needs them after the barrier
    bf16x8 orow[4];
#pragma unroll
    for (int v8 = 0; v8 < 4; ++v8) orow[v8] = *reinterpret_cast<const bf16x8*>(ob + (size_t)(32 * wave + kk) * D + 32 * hh + 8 * v8);
    // ---- everything else the phase behind the barrier takes from global memory is asked for HERE, so that its latency runs out
    // under the wait for the images: this wave's lse2 row and the odd key's K / V row as B operands (column 0; zeros elsewhere)
    // (every lane loads the row -- one 16-B segment per half-wave -- and the lanes off column 0 are zeroed behind the barrier: a
    // predicated load would be followed by a vmcnt(0) at the merge, i.e. by a wait for every LDS-DMA issued above)
    const float lse_own = p.lse2[((size_t)b * p.H + h) * T + 32 * wave + kk];
    bf16x8 klf[4], vlf[4];
    {
        const __bf16* krow = seq + (size_t)L * ld + D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            klf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
            vlf[s] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * s);
        }
    }
    bf16x4 klq[TPW];                                                  // the odd key's 4 columns of each dQ^T output tile of this wave
#pragma unroll
    for (int k = 0; k < TPW; ++k)
        klq[k] = *reinterpret_cast<const bf16x4*>(seq + (size_t)L * ld + D + h * HD + 16 * ((wave + NT * k) & 3) + 4 * (lane >> 4));
    // ---- the odd row against itself, and its statistics: five 64-vectors, one column per lane of wave 0 (the pair (L, L) is the
    // only part of the last query tile's column of the odd key that is not padding: no MFMA tile for it)
    if (wave == 0) {
        const size_t rl = (size_t)L * ld + h * HD + lane;
        const float ql = bf2f(seq[rl]), kl = bf2f(seq[rl + D]), vl = bf2f(seq[rl + 2 * D]);
        const float dol = bf2f(dob[(size_t)L * D + lane]), ol = bf2f(ob[(size_t)L * D + lane]);
        const float lse_l = p.lse2[((size_t)b * p.H + h) * T + L];
        const float dl = wave_sum_f32(ol * dol);
        const float sl = wave_sum_f32(ql * kl);
        const float dpl = wave_sum_f32(dol * vl) - dl;
        const float pl = __builtin_amdgcn_exp2f(__builtin_fmaf(sl, c, -lse_l));
        const float pr = bf2f(f2bf(pl)), dsr = bf2f(f2bf(pl * dpl));     // as the products see them
        part[(NT * 2 + 0) * 64 + lane] = dsr * ql;                       // the pair's share of dK[L], dV[L]
        part[(NT * 2 + 1) * 64 + lane] = pr * dol;
        if (lane < 32) {                               // rows L .. L + 31: the odd row, then padding (P = exp2(-inf) = 0)
            nlse[L + lane] = lane == 0 ? -lse_l : -INFINITY;
            ndel[L + lane] = lane == 0 ? -dl : 0.f;
            dsl[L + lane] = lane == 0 ? dsr * p.scale : 0.f;
        }
    }
    FZ_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    FZ_STAMP(2);
    // ---- K fragments of this wave's keys from the K image (64-row tiles, `row & 7` swizzle)
    {
        const int key = 32 * wave + kk;
        const char* krow = dsx + (key >> 6) * TILE_BYTES + (key & 63) * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(krow + (((2 * s + hh) ^ (key & 7)) << 4));
    }
    // ---- row statistics of this wave's query tile: -lse2, and -delta[q] = -(dO[q] . O[q]) over this head's 64 columns, dO from its image
    {
        const int row = 32 * wave + kk;
        float s = 0.f;
#pragma unroll
        for (int v8 = 0; v8 < 4; ++v8) {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dos + row * 128 + (((4 * hh + v8) ^ fz_sig(row)) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) s = __builtin_fmaf(bf2f(orow[v8][j]), bf2f(d[j]), s);
        }
        float s2 = s;
        lane_swap32(s, s2);                            // lanes l and l + 32 hold the two halves of one row
        if (hh == 0) {
            ndel[row] = -(s + s2);
            nlse[row] = -lse_own;
        }
    }

    // ---- K^T fragments of this wave's dQ^T output tiles (A operands, all keys), from the K image
    bf16x8 ktf[TPW][NKS];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            ktf[k][ks] = frag_tr(dsx + (ks >> 1) * TILE_BYTES, ks & 1, (wave + NT * k) & 3, lane);

    // ---- the odd key (row L) against every query tile: column 0 of the B operands holds it, the other columns are zero
    {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 z = {};
            klf[s] = kk == 0 ? klf[s] : z;
            vlf[s] = kk == 0 ? vlf[s] : z;
        }
        f32x16 dkl[2], dvl[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) dkl[dh][r] = dvl[dh][r] = 0.f;
        {
            const int tw = wave;                       // (the pair (L, L) of the last query tile: wave 0's scalars above)
            f32x16 s, dp = fz_rowconst(ndel, 32 * tw, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows(qs, 32 * tw, ks, lane), klf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_rows(dos, 32 * tw, ks, lane), vlf[ks], dp, 0, 0, 0);
            }
            const f32x16 nl = fz_rowconst(nlse, 32 * tw, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, nl[r]));
                dp[r] = s[r] * dp[r];
            }
            bf16x8 pb[2], db[2];
            fz_pack(s, pb);
            fz_pack(dp, db);
            if (kk == 0) {                            // dS[q, L] as the dQ product sees it (bf16), times scale
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<f32x4*>(dsl + 32 * tw + 8 * jj + 4 * hh) =
                        f32x4{bf2f(f2bf(dp[4 * jj])), bf2f(f2bf(dp[4 * jj + 1])), bf2f(f2bf(dp[4 * jj + 2])), bf2f(f2bf(dp[4 * jj + 3]))} * p.scale;
            }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    dvl[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_tr(dos, 32 * tw, sp, dh, lane), pb[sp], dvl[dh], 0, 0, 0);
                    dkl[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz_tr(qs, 32 * tw, sp, dh, lane), db[sp], dkl[dh], 0, 0, 0);
                }
        }
        if (kk == 0) {
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int d = 32 * dh + 8 * jj + 4 * hh;
                    *reinterpret_cast<f32x4*>(part + (wave * 2 + 0) * 64 + d) = f32x4{dkl[dh][4 * jj], dkl[dh][4 * jj + 1], dkl[dh][4 * jj + 2], dkl[dh][4 * jj + 3]};
                    *reinterpret_cast<f32x4*>(part + (wave * 2 + 1) * 64 + d) = f32x4{dvl[dh][4 * jj], dvl[dh][4 * jj + 1], dvl[dh][4 * jj + 2], dvl[dh][4 * jj + 3]};
                }
        }
    }
    FZ_STAMP(3);
    __syncthreads();                                   // K image dead (ktf in registers), dsl / part visible
    FZ_STAMP(4);
    if (t < 128) {                                     // the odd key's dK / dV rows: fixed-order sum of the waves' partials
        const int which = t >> 6, d = t & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w <= NT; ++w) v += part[(w * 2 + which) * 64 + d];
        if (which == 0) v *= p.scale;
        const __bf16 r = f2bf(v);
        p.out[((size_t)b * T + L) * ld + (1 + which) * D + h * HD + d] = r;
        lonekv[which * 64 + d] = bf2f(r);
    }

    // ---- main loop over the query tiles.  The kernel is bound by instruction issue (~250 instructions per wave and tile in the
    // first version, a third of them LDS address arithmetic), so the loop is fully unrolled and every LDS address is a
    // lane-constant base register + an immediate: the swizzles' XORs only ever mix lane bits with lane bits, the
    // compile-time parts (tile, k-step, half) are folded into which base is used and into the immediate.
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dh][r] = dv[dh][r] = 0.f;
    f32x4 qsum[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) qsum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, g = lane >> 4;
    const int sg0 = fz_sig(kk);                                       // sig(row0 + kk) for any row0 % 32 == 0
    const char* rowb[4];                                              // A operand rows: + row0 * 128 (+ IMG for dO)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rowb[ks] = qs + kk * 128 + (((2 * ks + hh) ^ sg0) << 4);
    // transposed A operand: row = 16 sp + 4 hh + (i16 >> 2) (+ 8), chunk = (4 dh + 2 (g & 1) + (p >> 1)) ^ sig(row); bit 4 of the
    // row (sp) flips bit 2 of sig, as dh flips bit 2 of the chunk: two bases, indexed by dh ^ sp
    const char* trb[2];
    {
        const int rlo = 4 * hh + (i16 >> 2), pq = i16 & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) trb[e] = qs + rlo * 128 + ((((2 * (g & 1) + (pq >> 1)) ^ fz_sig(rlo)) ^ (4 * e)) << 4) + (pq & 1) * 8;
    }
    auto tr32 = [&](int img_off, int row0, int sp, int dh) -> bf16x8 {
        const char* a0 = trb[dh ^ sp] + img_off + row0 * 128 + sp * 2048;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    };
    // dS image: write bases (4 slots of this lane's key), read bases (one per output tile's query half a)
    const char* dswb[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dswb[jj] = dsx + (32 * wave + kk) * 64 + (((2 * jj + hh) ^ fz_sigd(kk)) << 3);
    const char* dstb[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int u = wave + NT * k, a = u >> 2, bq = u & 3;
        const int klo = 4 * g + (i16 >> 2);
        dstb[k] = dsx + klo * 64 + (((4 * a + (i16 & 3)) ^ fz_sigd(klo)) << 3);
    }
    const float* rcb = nlse + 4 * hh;                                 // row constants: + row0 + 8 jj (ndel: + ROWS)
    auto rowconst = [&](int arr_off, int row0) -> f32x16 {
        f32x16 v;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(rcb + arr_off + row0 + 8 * jj);
            v[4 * jj + 0] = q[0]; v[4 * jj + 1] = q[1]; v[4 * jj + 2] = q[2]; v[4 * jj + 3] = q[3];
        }
        return v;
    };
#pragma unroll
    for (int qt = 0; qt <= NT; ++qt) {
        const int row0 = 32 * qt, boff = (qt & 1) * (L * 64);
        f32x16 s, dp = rowconst(ROWS, row0);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(rowb[ks] + row0 * 128), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(rowb[ks] + LY::IMG + row0 * 128), vf[ks], dp, 0, 0, 0);
        }
        const f32x16 nl = rowconst(0, row0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, nl[r]));
            dp[r] = s[r] * dp[r];
        }
        bf16x8 pb[2], db[2];
        fz_pack(s, pb);
        fz_pack(dp, db);
        {                                              // dS tile -> image [key][32 q]: registers 4 jj .. 4 jj + 3 = rows 8 jj + 4 h ..
            const u32x4 w0 = __builtin_bit_cast(u32x4, db[0]), w1 = __builtin_bit_cast(u32x4, db[1]);
            const u32 words[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *reinterpret_cast<u32x2*>(const_cast<char*>(dswb[jj]) + boff) = u32x2{words[2 * jj], words[2 * jj + 1]};
        }
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                dv[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32(LY::IMG, row0, sp, dh), pb[sp], dv[dh], 0, 0, 0);
                dk[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32(0, row0, sp, dh), db[sp], dk[dh], 0, 0, 0);
            }
        __syncthreads();                               // every wave's dS tile of this query tile is in the image
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            const int u = wave + NT * k, a = u >> 2, bq = u & 3;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const char* a0 = dstb[k] + boff + ks * 2048;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                s16x8 r8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[k][ks], __builtin_bit_cast(bf16x8, r8), acc, 0, 0, 0);
            }
            const int q = row0 + 16 * a + i16;
            const float dq_l = dsl[q];                 // the odd key: k_L[d] dS[q, L] (dsl holds dS scale); d = 16 bq + 4 g + r
            if (q < T) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = f2bf(__builtin_fmaf(acc[r], p.scale, bf2f(klq[k][r]) * dq_l));
                *reinterpret_cast<bf16x4*>(p.out + ((size_t)b * T + q) * ld + h * HD + 16 * bq + 4 * g) = w;
                qsum[k] += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
            }
        }
    }
    FZ_STAMP(5);
    __syncthreads();                                   // the last tile's transposed reads of Q / dO are done: the images become slabs

    // ---- epilogue: dK, dV of this wave's 32 keys through a private LDS slab -> whole 128-B rows; column sums
    {
        char* slab = lds + wave * (32 * FZ_SLAB_STRIDE);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int d = 32 * dh + 8 * jj + 4 * hh;
                const bf16x4 wk = {f2bf(dk[dh][4 * jj] * p.scale), f2bf(dk[dh][4 * jj + 1] * p.scale), f2bf(dk[dh][4 * jj + 2] * p.scale),
                                   f2bf(dk[dh][4 * jj + 3] * p.scale)};
                const bf16x4 wv = {f2bf(dv[dh][4 * jj]), f2bf(dv[dh][4 * jj + 1]), f2bf(dv[dh][4 * jj + 2]), f2bf(dv[dh][4 * jj + 3])};
                *reinterpret_cast<bf16x4*>(slab + kk * FZ_SLAB_STRIDE + d * 2) = wk;
                *reinterpret_cast<bf16x4*>(slab + kk * FZ_SLAB_STRIDE + 128 + d * 2) = wv;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the slab is wave-private, no barrier
        {
            const int key = lane >> 1, which = lane & 1;      // one lane = one 128-B row segment (dK or dV of one key)
            const char* src = slab + key * FZ_SLAB_STRIDE + which * 128;
            __bf16* dst = p.out + ((size_t)b * T + 32 * wave + key) * ld + (1 + which) * D + h * HD;
#pragma unroll
            for (int v8 = 0; v8 < 8; ++v8)
                *reinterpret_cast<u32x4*>(dst + 8 * v8) = *reinterpret_cast<const u32x4*>(src + 16 * v8);
        }
        if (p.colsum) {
            // column sums of the STORED (rounded) dq | dk | dv over this (b, h): partial row `wave` of RES_COLSUM_ROWS
            float* row = p.colsum + ((size_t)b * RES_COLSUM_ROWS + wave) * ld + h * HD;
            float s0 = 0.f, s1 = 0.f;                  // lane l: columns 2 l, 2 l + 1 of the slab's 128 (dK 64 | dV 64)
            bf16x2 cv[32];                             // all 32 reads in flight, then the sums in key order (rolled, hipcc waited for each read)
#pragma unroll
            for (int key = 0; key < 32; ++key) cv[key] = *reinterpret_cast<const bf16x2*>(slab + key * FZ_SLAB_STRIDE + 4 * lane);
            __builtin_amdgcn_sched_barrier(0);         // (without it the scheduler sinks every read to its use again)
#pragma unroll
            for (int key = 0; key < 32; ++key) {
                s0 += bf2f(cv[key][0]);
                s1 += bf2f(cv[key][1]);
            }
            const int which = lane >> 5, d = (2 * lane) & 63;
            if (wave == 0) {                           // the odd key's rows
                s0 += lonekv[which * 64 + d];
                s1 += lonekv[which * 64 + d + 1];
            }
            *reinterpret_cast<f32x2*>(row + (1 + which) * D + d) = f32x2{s0, s1};
            // dq: sum over the 16 queries of a lane row, then lane i16 == 0 holds the 4 columns 16 bq + 4 g ..; the other
            // 48 columns of this partial row are zero (this wave did not produce them)
            f32x4 red[TPW];
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                red[k] = qsum[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = red[k][r];
                    x += dpp_f32<0xB1>(x); x += dpp_f32<0x4E>(x); x += dpp_f32<0x141>(x); x += dpp_f32<0x140>(x);
                    red[k][r] = x;
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (((wave + NT * k) & 3) == q4) z += red[k];      // wave-uniform; tiles with the same bq (a = 0, 1) add up
                if (i16 == 0) *reinterpret_cast<f32x4*>(row + 16 * q4 + 4 * g) = z;
            }
            // partial rows NT .. 7 of this (b, h): zeros (the reducer sums RES_COLSUM_ROWS rows per sequence)
            for (int e = lane; e < (RES_COLSUM_ROWS / NT - 1) * 3 * 16; e += 64) {
                const int rr = e / 48, cc = e - rr * 48;
                float* zr = p.colsum + ((size_t)b * RES_COLSUM_ROWS + wave + NT * (rr + 1)) * ld + h * HD + (cc >> 4) * D + (cc & 15) * 4;
                *reinterpret_cast<f32x4*>(zr) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    FZ_STAMP(6);
    if (!PERSIST || bh + (int)gridDim.x >= p.B * p.H) break;
    __syncthreads();                                   // the slabs were the images: nobody stages the next pair over a slab still being read
    }
    // (On top of the walk, the NEXT pair's K image was prefetched by LDS-DMA into a buffer of its own -- an array of its own, raw
    // `s_waitcnt lgkmcnt(0); s_barrier` in the query-tile loop so that nothing waits for it -- issued once the K fragments are in
    // registers: 230 VGPRs, no wait in the loop, bit-identical, and 283-288 us against 283 us without it.  A fifth of the load
    // phase's bytes moved under the loop does not shorten the phase; the rest cannot follow -- Q' / dO' could refill the tiles the
    // loop is done with, but V' / O' (66 KB) have no LDS left at 144 KB.  Not kept.)
}

// (Two wave groups HALF a query tile apart -- waves w and w + NT / 2 share a SIMD and run every phase of a tile at the same time;
// group B one half-step (scores + softmax | dV / dK + dQ) behind group A, a barrier after every half -- was built and verified:
// 307 us against 297 us, the loop 26.8k cycles instead of 23.7k.  The loop is not bound by matrix pipe and VALU taking turns but
// by the LDS: ~1,470 LDS-array cycles per tile (every wave reads the same Q / dO tile as row fragments AND as transposed
// fragments: 128 of the 208 KB a tile moves) against 1,280 MFMA cycles per SIMD; a second barrier per tile only adds.  Not kept.)
//
// (A PERSISTENT form of this kernel -- one workgroup per CU walking its pairs, the next pair's Q / dO tiles loaded into
// registers one or two query tiles ahead and written over the tiles the current pair is done with, delta / lse2 of the next pair
// computed on the way, its K image held in registers across the epilogue -- was built and verified in round 3 (git history:
// attn_bwd_fusedp_kernel).  384 us with one staging set, 411 us with two, against 355 us for this kernel at the time: the staging
// work (two loads, a dot product and 3 DPP steps per tile on half of the waves, which the barrier of every query tile makes
// everyone wait for), 60-80 spilled registers at the 256-VGPR limit and the exposed K / V fetch at the pair seam cost more than
// the overlapped load phase returns.  Touching the next round's lines into L2 / Infinity Cache from this kernel was measured too:
// 368 vs 360 us, no gain.  Not kept)
//
// (A second persistent form moved no byte through a register: every query tile of Q and of dO its OWN __shared__ array, so that
// hipcc keeps no LDS read behind the LDS-DMA of the next pair's tile j -- issued right after the barrier of iteration j -- and raw
// `s_waitcnt lgkmcnt(0); s_barrier` barriers in the loop (`__syncthreads()` drains the DMAs); the next pair's K image in a buffer of
// its own, its V fragments / O rows loaded after the last use of their registers.  Correct (dqkv bit-identical to this kernel), the
// ISA shows no vmcnt wait in the loop once the wave index is made scalar (`readfirstlane`: before that the DMAs' LDS destinations
// sat in spilled VGPRs and every reload was a vmcnt(0) behind the DMA just issued: 395 us) -- and still 367 us against 307-311 us.
// Stamps: 66k cycles per pair against 45k; the nine query tiles alone 39k against 22.5k EVEN WITH the in-loop DMAs compiled out,
// at 256 VGPRs + 54 spills against 214 and none.  Whatever makes the same loop body 1.7x slower inside the persistent kernel
// (register-limited scheduling is the suspect) was not found in the time given to it; the kernel is not in the tree.)

// ================================================================================================================================
// dK / dV for T = 32 NT + 1 with 8 < NT <= 16 (T = 513: N = 2048 points, the point north_star states its targets at), key-stationary
// like the kernel above but WITHOUT the dQ part (16 key tiles would need 16 waves of 220 registers, or a dQ hand-off between two
// workgroups; dQ stays with attn_bwd_dq_q_kernel, which also leaves delta).  One workgroup = 8 waves = one HALF of the keys of a
// (batch, head) pair: wave w owns keys 256 kh + 32 w ..; Q and dO of the whole sequence resident in LDS (139 KB at T = 513), and a
// loop over the NT + 1 query tiles with NO barrier and NO LDS write: after the prologue the waves run free.  Against
// attn_bwd_dkdv_kernel (16 keys per wave, 64-row tiles streamed behind a barrier each): every Q / dO fragment read from LDS feeds
// a 32x32x16 MFMA (twice the MACs per LDS byte), one load phase per 17 tiles.  The odd key (row L) belongs to key half 0, handled
// as above (a phase of its own before the loop; the pair (L, L) by VALU on wave 0).
template <int NT>
struct KsLds {
    static constexpr int L = 32 * NT, ROWS = L + 32, IMG = ROWS * 128;
    static constexpr int OFF_Q = 0, OFF_DO = IMG;
    static constexpr int OFF_NLSE = 2 * IMG, OFF_NDEL = OFF_NLSE + ROWS * 4;
    static constexpr int OFF_PART = OFF_NDEL + ROWS * 4;            // [8 + 1][2][64] partial dK / dV rows of the odd key
    static constexpr int OFF_LONE = OFF_PART + 9 * 2 * 64 * 4;
    static constexpr int TOTAL = OFF_LONE + 2 * 64 * 4;
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert(2 * IMG >= 8 * 32 * FZ_SLAB_STRIDE, "epilogue slabs must fit the Q / dO images");
};

template <int NT, bool PERSIST = false>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkdv_ks_kernel(AttnParams p) {
    using LY = KsLds<NT>;
    constexpr int L = LY::L, ROWS = LY::ROWS, NW = 8, KH = NT / 8;
    static_assert(NT % 8 == 0, "whole key halves of 8 waves");
    __shared__ __attribute__((aligned(1024))) char lds[LY::TOTAL];
    const int T = p.T, D = p.D;                       // T = L + 1
    const size_t ld = (size_t)3 * D;
    const float c = p.scale * 1.4426950408889634f;
    char* qs = lds + LY::OFF_Q;
    char* dos = lds + LY::OFF_DO;
    float* nlse = reinterpret_cast<float*>(lds + LY::OFF_NLSE);
    float* ndel = reinterpret_cast<float*>(lds + LY::OFF_NDEL);
    float* part = reinterpret_cast<float*>(lds + LY::OFF_PART);
    float* lonekv = reinterpret_cast<float*>(lds + LY::OFF_LONE);
    if (p.stagger_phases > 1 && (int)blockIdx.x < p.stagger_n) {       // first round in shifted groups (see attn_bwd_fused_kernel)
        const int ph = blockIdx.x % p.stagger_phases;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)ph * p.stagger_cycles) __builtin_amdgcn_s_sleep(16);
    }
    // PERSIST: one workgroup per CU walks the (pair, key half) work items; the thread index goes through an empty asm per round so
    // that nothing derived from it is hoisted out of the walk (attn_bwd_fused_kernel)
    for (int wid = blockIdx.x;; wid += gridDim.x) {
    int t = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t));
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = wid % KH, bh = wid / KH, b = bh / p.H, h = bh - b * p.H;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* dob = p.dout + (size_t)b * T * D + h * HD;
    const int kk = lane & 31, hh = lane >> 5, i16 = lane & 15, g = lane >> 4;

    // ---- stage Q, dO (1-KiB pieces of 8 rows; rows L + 1 .. L + 7 repeat row L, the rest of the last tile is zero)
    {
        constexpr int NPC = L / 8 + 1;
        const unsigned qrow = (unsigned)ld * 2, drow = (unsigned)D * 2;          // uniform base + 32-bit lane offset: fz_src
        const FzSrc sq = fz_src(lane, qrow), sd = fz_src(lane, drow);
        const char* qbase = reinterpret_cast<const char*>(seq + h * HD);
        const char* dbase = reinterpret_cast<const char*>(dob);
        static_assert(NPC == L / 8 + 1, "");
#pragma unroll
        for (int i = 0; i < L / 8 / NW; ++i) {          // 8 pieces of Q and of dO per wave
            const int pc = wave + NW * i;
            fz_dma(qbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], qs + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < L / 8 / NW; ++i) {
            const int pc = wave + NW * i;
            fz_dma(dbase + (size_t)(8 * pc) * drow, ((pc >> 1) & 1) ? sd.par[1] : sd.par[0], dos + pc * 1024);
        }
        if (wave == 0) fz_dma(qbase + (size_t)L * qrow, sq.last, qs + (L / 8) * 1024);              // rows L .. L + 7: row L eight times
        if (wave == NW - 1) fz_dma(dbase + (size_t)L * drow, sd.last, dos + (L / 8) * 1024);
        for (int e = t; e < 2 * 24 * 8; e += NW * 64) {
            char* img = e >= 24 * 8 ? dos : qs;
            const int ee = e >= 24 * 8 ? e - 24 * 8 : e;
            *reinterpret_cast<f32x4*>(img + (L + 8) * 128 + ee * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- this wave's keys: K / V fragments (B operands of S and dP)
    bf16x8 kf[4], vf[4];
    {
        const __bf16* krow = seq + (size_t)(256 * kh + 32 * wave + kk) * ld + D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
            vf[s] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * s);
        }
    }
    // ---- row statistics: -lse2 and -delta (left by the dQ kernel); padding rows of the last tile: lse2 = +inf, P = 0
    {
        const float* lse_b = p.lse2 + (size_t)bh * T;
        const float* del_b = p.delta + (size_t)bh * T;
        for (int e = t; e < ROWS; e += NW * 64) {
            nlse[e] = e < T ? -lse_b[e] : -INFINITY;
            ndel[e] = e < T ? -del_b[e] : 0.f;
        }
    }
    // ---- the odd key's K / V row as B operands (key half 0; column 0, zeros elsewhere): asked for ahead of the barrier
    bf16x8 klf[4], vlf[4];
    {
        const __bf16* krow = seq + (size_t)L * ld + D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                  // (all lanes load, the lanes off column 0 are zeroed at use: attn_bwd_fused_kernel)
            klf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
            vlf[s] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * s);
        }
    }
    // ---- the odd row against itself (key half 0, wave 0): its share of dK[L], dV[L]
    if (kh == 0 && wave == 0) {
        const size_t rl = (size_t)L * ld + h * HD + lane;
        const float ql = bf2f(seq[rl]), kl = bf2f(seq[rl + D]), vl = bf2f(seq[rl + 2 * D]);
        const float dol = bf2f(dob[(size_t)L * D + lane]);
        const float lse_l = p.lse2[(size_t)bh * T + L], dl = p.delta[(size_t)bh * T + L];
        const float sl = wave_sum_f32(ql * kl);
        const float dpl = wave_sum_f32(dol * vl) - dl;
        const float pl = __builtin_amdgcn_exp2f(__builtin_fmaf(sl, c, -lse_l));
        part[(NW * 2 + 0) * 64 + lane] = bf2f(f2bf(pl * dpl)) * ql;
        part[(NW * 2 + 1) * 64 + lane] = bf2f(f2bf(pl)) * dol;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();

    // ---- lane-constant LDS bases (attn_bwd_fused_kernel)
    const char* rowb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rowb[ks] = qs + kk * 128 + (((2 * ks + hh) ^ fz_sig(kk)) << 4);
    const char* trb[2];
    {
        const int rlo = 4 * hh + (i16 >> 2), pq = i16 & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) trb[e] = qs + rlo * 128 + ((((2 * (g & 1) + (pq >> 1)) ^ fz_sig(rlo)) ^ (4 * e)) << 4) + (pq & 1) * 8;
    }
    auto tr32 = [&](int img_off, int row0, int sp, int dh) -> bf16x8 {
        const char* a0 = trb[dh ^ sp] + img_off + row0 * 128 + sp * 2048;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    };
    const float* rcb = nlse + 4 * hh;
    auto rowconst = [&](int arr_off, int row0) -> f32x16 {
        f32x16 v;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(rcb + arr_off + row0 + 8 * jj);
            v[4 * jj + 0] = q[0]; v[4 * jj + 1] = q[1]; v[4 * jj + 2] = q[2]; v[4 * jj + 3] = q[3];
        }
        return v;
    };
    // one query tile against 32 keys: P and dS (bf16, the B operands of the products that sum over the queries)
    auto front = [&](int row0, const bf16x8 (&kfr)[4], const bf16x8 (&vfr)[4], bf16x8 (&pb)[2], bf16x8 (&db)[2]) {
        f32x16 s, dp = rowconst(ROWS, row0);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(rowb[ks] + row0 * 128), kfr[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(rowb[ks] + LY::IMG + row0 * 128), vfr[ks], dp, 0, 0, 0);
        }
        const f32x16 nl = rowconst(0, row0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, nl[r]));
            dp[r] = s[r] * dp[r];
        }
        fz_pack(s, pb);
        fz_pack(dp, db);
    };
    auto back = [&](int row0, const bf16x8 (&pb)[2], const bf16x8 (&db)[2], f32x16 (&dka)[2], f32x16 (&dva)[2]) {
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                dva[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32(LY::IMG, row0, sp, dh), pb[sp], dva[dh], 0, 0, 0);
                dka[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr32(0, row0, sp, dh), db[sp], dka[dh], 0, 0, 0);
            }
    };

    // ---- the odd key (key half 0): wave w takes query tiles w, w + 8, .. against it (column 0 of the B operands)
    if (kh == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 z = {};
            klf[s] = kk == 0 ? klf[s] : z;
            vlf[s] = kk == 0 ? vlf[s] : z;
        }
        f32x16 dkl[2], dvl[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) dkl[dh][r] = dvl[dh][r] = 0.f;
#pragma unroll
        for (int e = 0; e < NT / NW; ++e) {
            bf16x8 pb[2], db[2];
            front(32 * (wave + NW * e), klf, vlf, pb, db);
            back(32 * (wave + NW * e), pb, db, dkl, dvl);
        }
        if (kk == 0) {
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int d = 32 * dh + 8 * jj + 4 * hh;
                    *reinterpret_cast<f32x4*>(part + (wave * 2 + 0) * 64 + d) = f32x4{dkl[dh][4 * jj], dkl[dh][4 * jj + 1], dkl[dh][4 * jj + 2], dkl[dh][4 * jj + 3]};
                    *reinterpret_cast<f32x4*>(part + (wave * 2 + 1) * 64 + d) = f32x4{dvl[dh][4 * jj], dvl[dh][4 * jj + 1], dvl[dh][4 * jj + 2], dvl[dh][4 * jj + 3]};
                }
        }
        __syncthreads();                               // (block-uniform branch)
        if (t < 128) {
            const int which = t >> 6, d = t & 63;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w <= NW; ++w) v += part[(w * 2 + which) * 64 + d];
            if (which == 0) v *= p.scale;
            const __bf16 r = f2bf(v);
            p.out[((size_t)b * T + L) * ld + (1 + which) * D + h * HD + d] = r;
            lonekv[which * 64 + d] = bf2f(r);
        }
    }

    // ---- the query tiles: no barrier, no LDS write
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dh][r] = dv[dh][r] = 0.f;
#pragma unroll
    for (int qt = 0; qt <= NT; ++qt) {
        bf16x8 pb[2], db[2];
        front(32 * qt, kf, vf, pb, db);
        back(32 * qt, pb, db, dk, dv);
    }
    __syncthreads();                                   // every wave is done with the images: they become slabs

    // ---- epilogue: dK, dV of this wave's 32 keys through a private LDS slab -> whole 128-B rows; column sums in the partial-row
    // layout of the streaming kernels (one row per 16 keys: this wave's sums go to the first of its two rows, zeros to the second)
    {
        char* slab = lds + wave * (32 * FZ_SLAB_STRIDE);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int d = 32 * dh + 8 * jj + 4 * hh;
                const bf16x4 wk = {f2bf(dk[dh][4 * jj] * p.scale), f2bf(dk[dh][4 * jj + 1] * p.scale), f2bf(dk[dh][4 * jj + 2] * p.scale),
                                   f2bf(dk[dh][4 * jj + 3] * p.scale)};
                const bf16x4 wv = {f2bf(dv[dh][4 * jj]), f2bf(dv[dh][4 * jj + 1]), f2bf(dv[dh][4 * jj + 2]), f2bf(dv[dh][4 * jj + 3])};
                *reinterpret_cast<bf16x4*>(slab + kk * FZ_SLAB_STRIDE + d * 2) = wk;
                *reinterpret_cast<bf16x4*>(slab + kk * FZ_SLAB_STRIDE + 128 + d * 2) = wv;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        {
            const int key = lane >> 1, which = lane & 1;
            const char* src = slab + key * FZ_SLAB_STRIDE + which * 128;
            __bf16* dst = p.out + ((size_t)b * T + 256 * kh + 32 * wave + key) * ld + (1 + which) * D + h * HD;
#pragma unroll
            for (int v8 = 0; v8 < 8; ++v8)
                *reinterpret_cast<u32x4*>(dst + 8 * v8) = *reinterpret_cast<const u32x4*>(src + 16 * v8);
        }
        if (p.colsum) {
            const int nblk = (T + 63) >> 6, kb = 4 * kh + (wave >> 1);
            float* row = p.colsum + (((size_t)b * nblk + kb) * 4 + 2 * (wave & 1)) * ld + h * HD;
            float s0 = 0.f, s1 = 0.f;                  // lane l: columns 2 l, 2 l + 1 of the slab's 128 (dK 64 | dV 64)
            bf16x2 cv[32];                             // all 32 reads in flight, then the sums in key order (rolled, hipcc waited for each read)
#pragma unroll
            for (int key = 0; key < 32; ++key) cv[key] = *reinterpret_cast<const bf16x2*>(slab + key * FZ_SLAB_STRIDE + 4 * lane);
            __builtin_amdgcn_sched_barrier(0);         // (without it the scheduler sinks every read to its use again)
#pragma unroll
            for (int key = 0; key < 32; ++key) {
                s0 += bf2f(cv[key][0]);
                s1 += bf2f(cv[key][1]);
            }
            const int which = lane >> 5, d = (2 * lane) & 63;
            *reinterpret_cast<f32x2*>(row + (1 + which) * D + d) = f32x2{s0, s1};
            *reinterpret_cast<f32x2*>(row + ld + (1 + which) * D + d) = f32x2{0.f, 0.f};
            if (kh == 0 && wave == 0) {                // the odd key: block nblk - 1, its first row; the three rows behind it: zeros
                float* lr = p.colsum + (((size_t)b * nblk + nblk - 1) * 4) * ld + h * HD + (1 + which) * D + d;
                *reinterpret_cast<f32x2*>(lr) = f32x2{lonekv[which * 64 + d], lonekv[which * 64 + d + 1]};
#pragma unroll
                for (int z = 1; z < 4; ++z) *reinterpret_cast<f32x2*>(lr + z * ld) = f32x2{0.f, 0.f};
            }
        }
    }
    if (!PERSIST || wid + (int)gridDim.x >= p.B * p.H * KH) break;
    __syncthreads();                                   // the slabs were the images
    }
}


// (The same kernel with the load phase PIPELINED was built and verified in round 3 -- the Q / dO images in four chunks of four query
// tiles, each chunk its own __shared__ array, all LDS-DMAs issued up front in chunk order, the loop started behind chunk 0 and passing
// a counted `s_waitcnt vmcnt(13 / 9 / 5 / 0)` + raw `s_barrier` in front of tiles 4, 8, 12; the odd key's phase moved behind the loop,
// the row statistics fetched by 4-byte LDS-DMAs with their signs folded into the products (-V as the dP operand, +delta as the
// chain's start).  dqkv and the column sums bit-identical to this kernel; backward at T = 513, B = 256, H = 12 in one call:
// 1,119-1,128 us against 1,103-1,131 us.  Diagnostic builds of it: no image loaded at all 1,032-1,050 us (so the whole load phase is
// worth 80 us of this kernel's ~570), only chunk 0 loaded 1,059-1,073, all chunks loaded but waited for up front 1,119-1,148 -- the
// three later chunks cost their 55 us whether they are waited for at the start or at tiles 4 / 8 / 12.  Two things learnt on the
// way, both from the ISA: with an LDS-DMA in the queue hipcc waits for ANY register filled by a plain global load with vmcnt(0)
// (i.e. for every DMA behind it), and it puts vmcnt(0) in front of a builtin LDS store into an array that has a DMA outstanding --
// a kernel that wants DMAs in flight across its first MFMAs needs inline-asm loads / stores for those, and raw barriers.  Not in the tree.)

// (The mirror image for dQ -- query-stationary: a wave's 32 queries with their Q / dO fragments, lse2, delta and dQ^T accumulators
// in registers, K and V of the whole sequence resident in LDS, 17 key tiles without a barrier, the odd query's dQ row from
// column-0 products, 164 VGPRs -- was built and verified (same errors as attn_bwd_dq_q_kernel<2>, bitwise reproducible): backward
// 1,129-1,136 us with it against 1,109-1,122 us with the streaming dQ kernel in front of attn_bwd_dkdv_ks_kernel.  The streaming dQ
// kernel already reads every K / V fragment for two query tiles; the barrier per tile it also pays is not what it waits for.
// Not kept.)
