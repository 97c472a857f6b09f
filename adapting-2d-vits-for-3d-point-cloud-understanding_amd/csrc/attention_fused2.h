// Fused attention backward for T = 32 NT + 1 with NT = 16 (T = 513: N = 2048 points, the point north_star states its targets at):
// ONE kernel for dQ, dK, dV.  Round 3 ran this shape as attn_bwd_dq_q_kernel<2> (streaming: S, dP recomputed, then dQ) followed by
// attn_bwd_dkdv_ks_kernel<16> (key-stationary: S, dP recomputed again, then dV, dK): seven MFMA products and two exp passes per score
// tile, q | k | v | dO fetched twice.  Here five products and one exp pass, every tensor fetched once per (batch, head) pair.
//
// Why the fused kernel of attention_fused.h does not simply scale up: 16 key tiles need 16 waves of ~220 registers, or two workgroups
// per pair with a dQ hand-off between them (atomics: not reproducible; an ordered hand-off: resident-together spins).  Instead ONE
// workgroup of 8 waves takes the pair's two KEY HALVES one after the other with Q and dO of the whole sequence resident in LDS
// (139 KB) across both:
//   half kh, wave w: keys 256 kh + 32 w .. + 31 -- their K / V fragments (B operands) and fp32 dK^T / dV^T accumulators in registers,
//   the K^T fragments of the half (A operands of the dQ product, 32 VGPRs) gathered from global memory a phase ahead of their use;
//   per 32-query tile (17 of them; the last holds the odd row 512 and 31 padding rows whose lse is +inf):
//       S = Q K^T, dP = dO V^T - delta (chain starts at -delta), P = exp2(c S - lse2), dS = P dP           (32x32x16, key on the lane)
//       dV^T += dO^T P,  dK^T += Q^T dS                                                  (the accumulator tile IS the B operand)
//       dS tile (bf16) -> LDS image [256 keys][32 q]; barrier; every wave one 16 d x 16 q tile of dQ^T of this key half: a chain of
//       eight v_mfma_f32_16x16x32_bf16 over the half's keys.
//   dQ sums over both halves: after half 0 a lane parks its f32x4 of dQ^T in a per-workgroup scratch in global memory (139 KB per
//   workgroup, 36 MB for the whole walk: L2 / Infinity-Cache resident), during half 1 THE SAME LANE takes it back (asked for a tile
//   ahead), adds its half-1 chain, applies `scale`, adds the odd key's rank-1 term and stores bf16.  Fixed order, no atomics, no
//   cross-lane visibility question (a lane re-reads its own store).
// LDS: Q | dO images 2 x 69,632 B, ONE dS image of 16 KB (two barriers per tile, half a tile apart: `B` sits behind the next tile's
// scores, where no wave waits long), row constants 6.5 KB: 162,176 of 163,840 B.  The odd key's partial rows and the wave-private
// slabs of the dK / dV epilogues alias the dS image.
// The odd key (row L = 512) belongs to half 0, handled as in attention_fused.h: before the loop wave w runs query tiles w and w + 8
// against it (that key alone in column 0 of the B operands), leaves dS[q, L] * scale in LDS -- the rank-1 term k_L dS[q, L] of every
// dQ row, added in half 1 -- and its share of that key's dK / dV rows; the pair (L, L) is five 64-vectors on wave 0.
// delta = rowsum(dO . O) is computed here (O rows from global memory, dO from its image): no delta kernel, no delta workspace.
//
// Arithmetic contract = that of the kernels it replaces (tests/emulate.py): P and dS rounded to bf16 where they enter a product,
// everything else fp32, `scale` applied once to dQ and dK.  dK / dV run attn_bwd_dkdv_ks_kernel's chains on a delta summed in the order of
// attn_bwd_fused_kernel (not the streaming dQ kernel's): equal up to single bf16 flips; dQ sums its keys as (half 0 chain) + (half 1
// chain) + odd key instead of one chain over 64-key tiles: likewise.
// Reference: softmax(q k^T hd^-0.5) v of timm's Attention (src/models/apf_utils.py:142-158), autograd of it.
#pragma once

// Diagnostic build only (tools/attn_f2_stamps.py compiles attention.hip with -DP4P_F2_STAMPS into its own library): s_memtime stamps of
// thread 0 at the phase boundaries of every pair, into the delta workspace BEHIND the dQ^T parking space (the tool allocates
// p4p_attention_bwd_workspace floats + 32 per pair).  No stamp in the product library.
#ifdef P4P_F2_STAMPS
#define F2_STAMP(k)                                                                                       \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                      \
        if (t == 0) reinterpret_cast<unsigned long long*>(p.delta + (size_t)gridDim.x * (F2Lds<16>::ROWS * 64))[(size_t)bh * 16 + (k)] = ts_; \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#else
#define F2_STAMP(k) do { } while (0)
#endif

template <int NT>
struct F2Lds {
    static constexpr int L = 32 * NT, ROWS = L + 32, IMG = ROWS * 128;
    static constexpr int HK = 256;                                    // keys per half
    static constexpr int DSX = HK * 64;                               // the dS image [256 keys][32 q] bf16
    static constexpr int OFF_Q = 0, OFF_DO = IMG, OFF_DSX = 2 * IMG;
    static constexpr int OFF_NLSE = OFF_DSX + DSX;                    // -lse2 [ROWS], -delta [ROWS], dS[q, L] scale [ROWS]
    static constexpr int OFF_NDEL = OFF_NLSE + ROWS * 4;
    static constexpr int OFF_DSL = OFF_NDEL + ROWS * 4;
    static constexpr int TOTAL = OFF_DSL + ROWS * 4;
    static constexpr int PART_BYTES = 9 * 2 * 64 * 4;                 // [8 + 1][2][64] partial dK / dV rows of the odd key: aliases the dS image
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert(PART_BYTES <= DSX, "part must fit the dS image");
    static_assert(NT * 32 == 2 * HK, "two key halves of 256");
};

template <int NT, bool PERSIST = true>
__global__ __launch_bounds__(512, 1) void attn_bwd_fused2_kernel(AttnParams p, float* __restrict__ scratch) {
    using LY = F2Lds<NT>;
    constexpr int L = LY::L, ROWS = LY::ROWS, NW = 8, NKS = 8;
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
    float* part = reinterpret_cast<float*>(dsx);
    float* scr = scratch + (size_t)blockIdx.x * (ROWS * 64);          // this workgroup's dQ^T parking space: [17 tiles][8 waves][64 lanes] f32x4

    if (p.stagger_phases > 1 && (int)blockIdx.x < p.stagger_n) {       // first round in shifted groups (attn_bwd_fused_kernel)
        const int ph = blockIdx.x % p.stagger_phases;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)ph * p.stagger_cycles) __builtin_amdgcn_s_sleep(16);
    }
    for (int bh = blockIdx.x;; bh += gridDim.x) {
    int t = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t));           // nothing derived from the thread index is invariant of the walk (attention_fused.h)
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = bh / p.H, h = bh - b * p.H;
    const __bf16* seq = p.qkv + (size_t)b * T * ld;
    const __bf16* dob = p.dout + (size_t)b * T * D + h * HD;
    const __bf16* ob = p.o + (size_t)b * T * D + h * HD;
    const int kk = lane & 31, hh = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const unsigned qrow = (unsigned)ld * 2, drow = (unsigned)D * 2;   // bytes per row of qkv / of dO
    const unsigned lk = (lane >> 3) * qrow + (((lane & 7) ^ (lane >> 3)) << 4);     // K chunk pieces: 8 rows x 128 B, chunk ^ (row & 7)
    const char* kbase = reinterpret_cast<const char*>(seq + D + h * HD);
    F2_STAMP(0);

    // ---- stage Q, dO (1-KiB pieces of 8 rows; rows L + 1 .. L + 7 repeat row L, the rest of the last tile is zero)
    {
        const FzSrc sq = fz_src(lane, qrow), sd = fz_src(lane, drow);
        const char* qbase = reinterpret_cast<const char*>(seq + h * HD);
        const char* dbase = reinterpret_cast<const char*>(dob);
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
        if (wave == 0) fz_dma(qbase + (size_t)L * qrow, sq.last, qs + (L / 8) * 1024);
        if (wave == NW - 1) fz_dma(dbase + (size_t)L * drow, sd.last, dos + (L / 8) * 1024);
        for (int e = t; e < 2 * 24 * 8; e += NW * 64) {
            char* img = e >= 24 * 8 ? dos : qs;
            const int ee = e >= 24 * 8 ? e - 24 * 8 : e;
            *reinterpret_cast<f32x4*>(img + (L + 8) * 128 + ee * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- everything the phase behind the barrier takes from global memory is asked for here (unconditional loads: a predicated
    // one would be followed by vmcnt(0) at the merge): O rows and lse2 of this wave's two query tiles (w, w + 8), the odd key's K / V
    // rows as B operands, its four columns of this wave's dQ^T tile
    bf16x8 orow[2][4];
    float lse_own[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 32 * (wave + NW * e) + kk;
#pragma unroll
        for (int v8 = 0; v8 < 4; ++v8) orow[e][v8] = *reinterpret_cast<const bf16x8*>(ob + (size_t)row * D + 32 * hh + 8 * v8);
        lse_own[e] = p.lse2[(size_t)bh * T + row];
    }
    bf16x8 klf[4], vlf[4];
    {
        const __bf16* krow = seq + (size_t)L * ld + D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            klf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
            vlf[s] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * s);
        }
    }
    const bf16x4 klq = *reinterpret_cast<const bf16x4*>(seq + (size_t)L * ld + D + h * HD + 16 * (wave & 3) + 4 * g);
    // ---- the odd row against itself and its statistics (wave 0; one column per lane): the pair (L, L) is the only part of the
    // last query tile's column of the odd key that is not padding
    float pair_dk = 0.f, pair_dv = 0.f;                // wave 0: the pair's share of dK[L], dV[L] (into `part` once the dS image is free)
    if (wave == 0) {
        const size_t rl = (size_t)L * ld + h * HD + lane;
        const float ql = bf2f(seq[rl]), kl = bf2f(seq[rl + D]), vl = bf2f(seq[rl + 2 * D]);
        const float dol = bf2f(dob[(size_t)L * D + lane]), ol = bf2f(ob[(size_t)L * D + lane]);
        const float lse_l = p.lse2[(size_t)bh * T + L];
        const float dl = wave_sum_f32(ol * dol);
        const float sl = wave_sum_f32(ql * kl);
        const float dpl = wave_sum_f32(dol * vl) - dl;
        const float pl = __builtin_amdgcn_exp2f(__builtin_fmaf(sl, c, -lse_l));
        const float pr = bf2f(f2bf(pl)), dsr = bf2f(f2bf(pl * dpl));     // as the products see them
        pair_dk = dsr * ql;
        pair_dv = pr * dol;
        if (lane < 32) {                               // rows L .. L + 31: the odd row, then padding (P = exp2(-inf) = 0)
            nlse[L + lane] = lane == 0 ? -lse_l : -INFINITY;
            ndel[L + lane] = lane == 0 ? -dl : 0.f;
            dsl[L + lane] = lane == 0 ? dsr * p.scale : 0.f;
        }
    }
    F2_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    F2_STAMP(2);
    // ---- row statistics of this wave's two query tiles: -lse2, and -delta[q] = -(dO[q] . O[q]) over this head's 64 columns
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 32 * (wave + NW * e) + kk;
        float s = 0.f;
#pragma unroll
        for (int v8 = 0; v8 < 4; ++v8) {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dos + row * 128 + (((4 * hh + v8) ^ fz_sig(row)) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) s = __builtin_fmaf(bf2f(orow[e][v8][j]), bf2f(d[j]), s);
        }
        float s2 = s;
        lane_swap32(s, s2);                            // lanes l and l + 32 hold the two halves of one row
        if (hh == 0) {
            ndel[row] = -(s + s2);
            nlse[row] = -lse_own[e];
        }
    }

    // ---- lane-constant LDS bases (attn_bwd_fused_kernel): every address in the loops below is one of these + an immediate
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
    // one query tile against 32 keys: P and dS as bf16 B operands of the products that sum over the queries
    auto front = [&](int row0, const bf16x8 (&kfr)[4], const bf16x8 (&vfr)[4], bf16x8 (&pb)[2], bf16x8 (&db)[2], f32x16& dsf) {
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
        dsf = dp;
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
    // dS image: write bases (the four 8-B slots of this lane's key), read base of this wave's dQ^T tile (a = query half, bq = d block)
    const int a = wave >> 2, bq = wave & 3;
    const char* dswb[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dswb[jj] = dsx + (32 * wave + kk) * 64 + (((2 * jj + hh) ^ fz_sigd(kk)) << 3);
    const char* dstb;
    {
        const int klo = 4 * g + (i16 >> 2);
        dstb = dsx + klo * 64 + (((4 * a + (i16 & 3)) ^ fz_sigd(klo)) << 3);
    }
    f32x4 qsum = {0.f, 0.f, 0.f, 0.f};

    // ---- operands of a key half: the K / V fragments of this wave's 32 keys (B operands of S and dP) straight from global memory,
    // and the K^T fragments of ALL the half's keys for this wave's 16 d (A operands of the dQ^T product; element j of lane (i16, g) of
    // k-step ks = key 32 ks + 16 (j >> 2) + 4 g + (j & 3), the order the transposed reads of the dS image deliver).  K^T is a
    // transpose, i.e. an LDS round trip -- WAVE-PRIVATE here: the wave loads its own 32-B column block of the half's 256 K rows
    // (8 loads of 16 B per lane, asked for a whole phase ahead: half 0 in front of the odd key's phase, half 1 in front of half 0's
    // dK / dV epilogue), and later turns them 64 keys at a time through its own 2 KB of the dS image with ds_read_b64_tr_b16 -- no
    // barrier, no DMA.  (First version: a 32-KB K image through the 16-KB dS region in two chunks, each a DMA round trip between two
    // barriers: 8.1k + 8.4k of a pair's 175k cycles; second: a 2-byte column gather straight into the fragments, 64 loads per lane
    // and half: ~10-16k, bound by the texture addresser.  tools/attn_f2_stamps.py.)
    bf16x8 kf[4], vf[4], ktf[NKS];
    u32x4 kst[8];                                      // staging: rows 32 i + (lane >> 1) of the half, bytes 16 (lane & 1) .. of the wave's 32-B block
    auto load_kv = [&](int kh) {
        const __bf16* krow = seq + (size_t)(256 * kh + 32 * wave + kk) * ld + D + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
            vf[s] = *reinterpret_cast<const bf16x8*>(krow + D + 16 * s);
        }
    };
    auto load_kst = [&](int kh) {
        // a uniform base per load + ONE 32-bit lane offset, the row stride through an empty asm per call: as eight per-lane 64-bit
        // addresses they are invariants of the walk, get hoisted and spilled, and every reload is a vmcnt(0) in front of its load
        unsigned ldl = (unsigned)ld * 2;
        asm volatile("" : "+s"(ldl));
        const char* kblk = reinterpret_cast<const char*>(seq + D + h * HD + 16 * bq) + (size_t)(256 * kh) * ldl;
        unsigned koff = (lane >> 1) * ldl + 16 * (lane & 1);
        asm volatile("" : "+v"(koff));
#pragma unroll
        for (int i = 0; i < 8; ++i) kst[i] = *reinterpret_cast<const u32x4*>(kblk + (size_t)(32 * i) * ldl + koff);
    };
    auto transpose_half = [&]() {                      // kst -> ktf through this wave's 2 KB of the dS image: [64 keys][16 d] = 32-B rows
        char* slab = dsx + wave * 2048;
        // transposed read: lane 4 q + p of a 16-lane group supplies row (key) 4 g + q, columns 4 p ..; it receives column (d) i16 of the
        // group's 4 keys.  Two reads (keys + 0, + 16) make k-step ks; a 32-lane half covers 8 keys = 256 contiguous bytes: no conflict
        const char* rd = slab + (4 * g + (i16 >> 2)) * 32 + (i16 & 3) * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<u32x4*>(slab + lane * 16) = kst[2 * c];
            *reinterpret_cast<u32x4*>(slab + 1024 + lane * 16) = kst[2 * c + 1];
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rd + e * 1024));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rd + e * 1024 + 512));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ktf[2 * c + e] = __builtin_bit_cast(bf16x8, r);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);        // the reads are back before the next 64 keys overwrite the slab
        }
    };

    // ---- the odd key, ahead of the halves (its K / V operands and the 64 accumulator registers of its dK / dV rows die here): wave w
    // takes query tiles w, w + 8 against it (column 0 of the B operands)
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
#pragma unroll
        for (int e = 0; e < NT / NW; ++e) {
            const int row0 = 32 * (wave + NW * e);
            bf16x8 pb[2], db[2];
            f32x16 dsf;
            front(row0, klf, vlf, pb, db, dsf);
            if (kk == 0) {                         // dS[q, L] as the dQ product sees it (bf16), times scale
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<f32x4*>(dsl + row0 + 8 * jj + 4 * hh) =
                        f32x4{bf2f(f2bf(dsf[4 * jj])), bf2f(f2bf(dsf[4 * jj + 1])), bf2f(f2bf(dsf[4 * jj + 2])), bf2f(f2bf(dsf[4 * jj + 3]))} * p.scale;
            }
            back(row0, pb, db, dkl, dvl);
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
        if (wave == 0) {
            part[(NW * 2 + 0) * 64 + lane] = pair_dk;
            part[(NW * 2 + 1) * 64 + lane] = pair_dv;
        }
        __syncthreads();                           // part, dsl visible
        if (t < 128) {                             // the odd key's dK / dV rows: fixed-order sum of the waves' partials
            const int which = t >> 6, d = t & 63;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w <= NW; ++w) v += part[(w * 2 + which) * 64 + d];
            if (which == 0) v *= p.scale;
            const __bf16 r = f2bf(v);
            p.out[((size_t)b * T + L) * ld + (1 + which) * D + h * HD + d] = r;
            if (p.colsum) {
                // the odd key's partial rows of the qkv bias gradient (block nblk - 1 of the streaming kernels' layout): its rounded dk | dv
                // in the first row, zeros in the other three, zeros in the dq columns of all four.  Written HERE, where no store is in
                // flight: at the end of half 0 (where the key-stationary kernel does it) the five addresses came back from spill slots,
                // each reload a vmcnt(0) behind the 139 KB of parked dQ^T and the dK / dV rows -- 9k cycles on wave 0, i.e. on everyone
                // at the next barrier
                const int nblk = (T + 63) >> 6;
                float* lr = p.colsum + (((size_t)b * nblk + nblk - 1) * 4) * ld + h * HD;
                lr[(1 + which) * D + d] = bf2f(r);
#pragma unroll
                for (int z = 1; z < 4; ++z) lr[z * ld + (1 + which) * D + d] = 0.f;
                lr[(2 * which) * ld + d] = 0.f;
                lr[(2 * which + 1) * ld + d] = 0.f;
            }
        }
        __syncthreads();                           // `part` aliases the dS image: the K chunks of half 0 come next
    }


    F2_STAMP(3);
#pragma unroll 1
    for (int kh = 0; kh < 2; ++kh) {
        // the half's operands, all in this iteration (asked for earlier -- half 0's in front of the odd key's phase, half 1's in front of
        // half 0's epilogue -- the 32 staging registers ride along with 64 accumulators or across the back edge of this loop, and hipcc
        // spilled them on arrival, one exposed round trip per load: 20k cycles instead of the ~3k this wait costs)
        load_kst(kh);
        load_kv(kh);
        transpose_half();                              // (behind the barrier that ended the use of `part` / behind this wave's own epilogue slab)
        F2_STAMP(4 + 3 * kh);
        // ---- the query tiles
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[dh][r] = dv[dh][r] = 0.f;
        // this lane's corner of the dQ^T tile: query 16 a + i16 of the tile, d = 16 bq + 4 g ..  Global addresses = a uniform base
        // per tile + a 32-bit lane offset that passes through an empty asm in every half: written as 64-bit lane pointers the 17
        // scratch and 17 dq addresses of the unrolled tiles are invariants of the kh loop, and hipcc hoists all of them out of it
        // and keeps them live (68 VGPRs: 256 + 133 spills in the first build of this kernel)
        unsigned zl = 0;
        asm volatile("" : "+v"(zl));
        // scratch layout [tile][wave][lane] f32x4: the lane that wrote a value is the one that reads it, so the layout is free -- and
        // this one makes every store / load instruction 1 KB contiguous (as [q][d] rows a store was 16 x 64-B pieces of half lines:
        // the 139 KB of half 0 drained at ~12 B/clk and everything behind them in the memory queue waited: 11k cycles per pair)
        const unsigned sc_off = (wave * 64 + lane) * 16 + zl;
        const unsigned dq_off = ((16 * a + i16) * (unsigned)ld + 16 * bq + 4 * g) * 2 + zl;
        const char* dslb = reinterpret_cast<const char*>(dsl) + (16 * a + i16) * 4 + zl;
        const bool first_row = (16 * a + i16) == 0;   // the last query tile holds one real row (L = T - 1)
        int zs = 0;
        asm volatile("" : "+s"(zs));                   // (the same for the uniform bases: 17 + 17 SGPR pairs hoisted and spilled otherwise)
        char* scb = reinterpret_cast<char*>(scr) + zs;
        char* dqb = reinterpret_cast<char*>(p.out + (size_t)b * T * ld + h * HD) + zs;
#pragma unroll
        for (int qt = 0; qt <= NT; ++qt) {
            const int row0 = 32 * qt;
            unsigned sc_o = sc_off, dq_o = dq_off;     // (per tile again: the 17 per-tile lane offsets are otherwise made up front and spilled)
            asm volatile("" : "+v"(sc_o), "+v"(dq_o));
            f32x4 prev = {0.f, 0.f, 0.f, 0.f};
            if (kh == 1) prev = *reinterpret_cast<const f32x4*>(scb + (size_t)qt * 8192 + sc_o);     // half 0's chain: asked for a tile's work ahead of its use
            bf16x8 pb[2], db[2];
            f32x16 dsf;
            front(row0, kf, vf, pb, db, dsf);
#ifndef P4P_F2_NO_B                                    // (timing-only diagnostic: what the second barrier of a tile costs; racy without it)
            if (qt > 0) __syncthreads();               // [B] every wave has read the previous tile's dS image
#endif
            {                                          // dS tile -> image [key][32 q]: registers 4 jj .. 4 jj + 3 = rows 8 jj + 4 h ..
                const u32x4 w0 = __builtin_bit_cast(u32x4, db[0]), w1 = __builtin_bit_cast(u32x4, db[1]);
                const u32 words[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<u32x2*>(const_cast<char*>(dswb[jj])) = u32x2{words[2 * jj], words[2 * jj + 1]};
            }
            back(row0, pb, db, dk, dv);
            __syncthreads();                           // [A] every wave's dS tile of this query tile is in the image
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const char* a0 = dstb + ks * 2048;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                s16x8 r8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[ks], __builtin_bit_cast(bf16x8, r8), acc, 0, 0, 0);
            }
            if (kh == 0) {
#ifndef P4P_F2_NO_SCRATCH                              // (timing-only diagnostic: tools/attn_f2_stamps.py -DP4P_F2_NO_SCRATCH; results are wrong without the store)
                *reinterpret_cast<f32x4*>(scb + (size_t)qt * 8192 + sc_o) = acc;
#else
                asm volatile("" :: "v"(acc));
#endif
            } else {
                // the odd key: k_L[d] dS[q, L] (dsl holds dS scale); d = 16 bq + 4 g + r.  (lane base + immediate, and the q < T test of the
                // last tile as a lane constant: `dsl[q]` / `q < T` with q = row0 + .. gave 17 hoisted addresses and 17 hoisted masks)
                const float dq_l = *reinterpret_cast<const float*>(dslb + row0 * 4);
                if (qt < NT || first_row) {
                    bf16x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = f2bf(__builtin_fmaf(prev[r] + acc[r], p.scale, bf2f(klq[r]) * dq_l));
                    *reinterpret_cast<bf16x4*>(dqb + (size_t)row0 * ld * 2 + dq_o) = w;
                    qsum += f32x4{bf2f(w[0]), bf2f(w[1]), bf2f(w[2]), bf2f(w[3])};
                }
            }
        }

        __syncthreads();                               // every wave has read the last tile's dS image: its region becomes slabs
        F2_STAMP(5 + 3 * kh);
        // ---- dK, dV of this wave's 32 keys of the half.  The Q / dO images stay in use (half 1 / nothing else is free), so the slab
        // is this wave's 2 KB of the dS image: 8 keys x (dK 128 B | dV 128 B) per round, four rounds; 16-B chunk' = chunk ^ key:
        // conflict-free 8-B writes from the accumulator layout, 16-B reads of whole 128-B row segments, and the column sums of the
        // stored values from the same slab in key order.  (Straight from the accumulators -- 16 stores of 32 x 16-B pieces per wave --
        // this phase took 8.6k cycles per half, and 19k with a cross-lane reduction for the column sums.)
        const int nblk = (T + 63) >> 6;
        float* crow = p.colsum ? p.colsum + (((size_t)b * nblk + 4 * kh + (wave >> 1)) * 4 + 2 * (wave & 1)) * ld + h * HD : nullptr;
        {
            char* slab = dsx + wave * 2048;
            const int kl = kk & 7;                     // key of the round this lane's values belong to
            float s0 = 0.f, s1 = 0.f;                  // lane l: columns 2 l, 2 l + 1 of the slab's 128 (dK 64 | dV 64)
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                if ((kk >> 3) == rd) {
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const bf16x4 wk = {f2bf(dk[dh][4 * jj] * p.scale), f2bf(dk[dh][4 * jj + 1] * p.scale), f2bf(dk[dh][4 * jj + 2] * p.scale),
                                               f2bf(dk[dh][4 * jj + 3] * p.scale)};
                            const bf16x4 wv = {f2bf(dv[dh][4 * jj]), f2bf(dv[dh][4 * jj + 1]), f2bf(dv[dh][4 * jj + 2]), f2bf(dv[dh][4 * jj + 3])};
                            *reinterpret_cast<bf16x4*>(slab + kl * 256 + (((4 * dh + jj) ^ kl) << 4) + 8 * hh) = wk;
                            *reinterpret_cast<bf16x4*>(slab + kl * 256 + (((8 + 4 * dh + jj) ^ kl) << 4) + 8 * hh) = wv;
                        }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the slab is wave-private, no barrier
#pragma unroll
                for (int which = 0; which < 2; ++which) {     // one lane = one 16-B chunk; 8 lanes = one 128-B row segment
                    const int key = lane >> 3, ch = lane & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(slab + key * 256 + (((8 * which + ch) ^ key) << 4));
                    *reinterpret_cast<u32x4*>(p.out + ((size_t)b * T + 256 * kh + 32 * wave + 8 * rd + key) * ld + (1 + which) * D + h * HD + 8 * ch) = v;
                }
                if (p.colsum) {
                    const int cc = lane >> 2, off = (lane & 3) * 4;      // columns 2 l, 2 l + 1 = bytes 4 l .. 4 l + 3 of a key's 256
                    bf16x2 cv[8];                      // all 8 reads in flight, then the sums in key order (hipcc waits for each read otherwise)
#pragma unroll
                    for (int key = 0; key < 8; ++key) cv[key] = *reinterpret_cast<const bf16x2*>(slab + key * 256 + ((cc ^ key) << 4) + off);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int key = 0; key < 8; ++key) {
                        s0 += bf2f(cv[key][0]);
                        s1 += bf2f(cv[key][1]);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);        // the reads are back before the next round overwrites the slab
            }
            if (p.colsum) {
                // column sums of the STORED (rounded) values in the partial-row layout of the streaming kernels ([B][nblk][4 rows][3D],
                // one row pair per 32 keys: sums in the first, zeros in the second); dq columns: this wave's 16 d in half 1, zeros else
                const int which = lane >> 5, d = (2 * lane) & 63;
                *reinterpret_cast<f32x2*>(crow + (1 + which) * D + d) = f32x2{s0, s1};
                *reinterpret_cast<f32x2*>(crow + ld + (1 + which) * D + d) = f32x2{0.f, 0.f};
                f32x4 red = qsum;                       // dq: sum over the 16 queries of a lane row (half 1: all 17 tiles are in)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = red[r];
                    x += dpp_f32<0xB1>(x); x += dpp_f32<0x4E>(x); x += dpp_f32<0x141>(x); x += dpp_f32<0x140>(x);
                    red[r] = x;
                }
                // the 64 dq columns of both rows, every one written exactly once: after the butterfly all 16 lanes of a lane row hold its
                // sum, so lane (g, i16 < 4) writes column group 4 i16 + g of the first row (this wave's values where i16 == bq, zeros
                // elsewhere) and lane (g, 4 <= i16 < 8) the same group of the second row (zeros)
                if (i16 < 8) {
                    const bool mine = kh == 1 && i16 == bq;
                    *reinterpret_cast<f32x4*>(crow + (i16 >> 2) * ld + 16 * (i16 & 3) + 4 * g) = mine ? red : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        F2_STAMP(6 + 3 * kh);
    }
    if (!PERSIST || bh + (int)gridDim.x >= p.B * p.H) break;
    }
}
