// Attention forward for T = 32 NT + 1 with NT = 16 (T = 513: N = 2048 points), K and V of the whole sequence resident in LDS (139 KB), one
// workgroup of 8 waves per CU walking its (batch, head) pairs.  Round 3 ran this shape as attn_fwd_q_kernel<2>: 64-key tiles streamed
// through LDS behind a barrier each, 16x16x32 MFMAs, 16 queries per accumulator tile.  Here:
//   * v_mfma_f32_32x32x16_bf16 with the QUERY on the lane (S^T[key][q] = K . Q^T): a lane owns one query, its 32 scores of a key tile
//     are 16 registers (+ 16 in the partner lane l ^ 32), so the row maximum is 15 v_max + one v_permlane32_swap, the numerators P^T are the
//     B operand of O^T[d][q] += V^T . P^T as they stand (the accumulator tile IS the operand), and twice the MACs per LDS byte;
//   * wave w owns query tiles 2 w, 2 w + 1 (Q fragments in registers, straight from global memory) and walks the 17 key tiles of 32
//     (the last one holds the odd key 512 alone) with NO barrier in the loop: every K fragment (row read) and V^T fragment (transposed
//     read) is read once per key tile and feeds both query tiles;
//   * the odd QUERY (row 512, the cls token) has no wave: wave w runs it against key tiles 2 w, 2 w + 1 (wave 7: also tile 16) as a
//     third query tile with one valid lane, and the eight (max, sum, accumulator) partials meet in LDS.
// Arithmetic contract (tests/emulate.py, mode 'online32'): online softmax over 32-key tiles in ascending order, numerators
// p = exp2(fma(s, c, -m)) rounded to bf16 for the P.V product AND for the denominator (row sums from the matrix pipe: one MFMA against
// an all-ones operand per 16 keys), accumulators rescaled only when some lane's maximum grew, one division at the end; the odd query:
// one such chain per wave over its key tiles, the chains combined in wave order.
// Reference: softmax(q k^T hd^-0.5) v of timm's Attention (src/models/apf_utils.py:142-158).
#pragma once

// Diagnostic build only (tools/attn_fwd2_stamps.py, -DP4P_FW2_STAMPS): s_memtime stamps of thread 0 per pair, behind the lse2 rows.
#ifdef P4P_FW2_STAMPS
#define FW2_STAMP(k)                                                                                      \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                      \
        if (t == 0) reinterpret_cast<unsigned long long*>(p.lse2 + (size_t)p.B * p.H * p.T)[(size_t)bh * 16 + (k)] = ts_;   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#define FW2_STAMP_WAVES()                                                                                 \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const unsigned long long ts_ = __builtin_readcyclecounter();                                      \
        if (lane == 0) reinterpret_cast<unsigned long long*>(p.lse2 + (size_t)p.B * p.H * p.T)[(size_t)bh * 16 + 8 + wave] = ts_;   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#else
#define FW2_STAMP(k) do { } while (0)
#define FW2_STAMP_WAVES() do { } while (0)
#endif
// Diagnostic build only: knock-outs (wrong results) that price the parts of the key-tile loop.  bit 0: no exponentials, 1: no ones . P
// products, 2: no P . V products, 3: no K . Q products, 4: no row maximum, 5: no fragment reads in the loop, 6: no output stores
#ifndef P4P_FW2_KO
#define P4P_FW2_KO 0
#endif

template <int NT>
struct Fw2Lds {
    static constexpr int L = 32 * NT, ROWS = L + 32, IMG = ROWS * 128;
    static constexpr int OFF_K = 0, OFF_V = IMG, OFF_PART = 2 * IMG;   // partials of the odd query: [8 waves][2 + 64] floats
    static constexpr int TOTAL = OFF_PART + 8 * 68 * 4;
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <int NT, bool PERSIST = true>
__global__ __launch_bounds__(512, 1) void attn_fwd_kv_kernel(AttnParams p) {
    using LY = Fw2Lds<NT>;
    constexpr int L = LY::L, NW = 8, NKT = NT + 1;
    static_assert(NT == 2 * NW, "two query tiles per wave");
    __shared__ __attribute__((aligned(1024))) char lds[LY::TOTAL];
    const int T = p.T, D = p.D;                       // T = L + 1
    const size_t ld = (size_t)3 * D;
    const float c = p.scale * 1.4426950408889634f;
    char* kimg = lds + LY::OFF_K;
    char* vimg = lds + LY::OFF_V;
    float* part = reinterpret_cast<float*>(lds + LY::OFF_PART);
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
    const int kk = lane & 31, hh = lane >> 5, i16 = lane & 15, g = lane >> 4;

    FW2_STAMP(0);
    // ---- stage K and V (1-KiB pieces of 8 rows, fz_sig swizzle; rows L + 1 .. L + 7 repeat row L, the rest of the last tile is zero)
    {
        const unsigned qrow = (unsigned)ld * 2;
        const FzSrc sq = fz_src(lane, qrow);
        const char* kbase = reinterpret_cast<const char*>(seq + D + h * HD);
        const char* vbase = reinterpret_cast<const char*>(seq + 2 * D + h * HD);
#pragma unroll
        for (int i = 0; i < L / 8 / NW; ++i) {          // 8 pieces of K and of V per wave
            const int pc = wave + NW * i;
            fz_dma(kbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], kimg + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < L / 8 / NW; ++i) {
            const int pc = wave + NW * i;
            fz_dma(vbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], vimg + pc * 1024);
        }
        if (wave == 0) fz_dma(kbase + (size_t)L * qrow, sq.last, kimg + (L / 8) * 1024);
        if (wave == NW - 1) fz_dma(vbase + (size_t)L * qrow, sq.last, vimg + (L / 8) * 1024);
        for (int e = t; e < 2 * 24 * 8; e += NW * 64) {
            char* img = e >= 24 * 8 ? vimg : kimg;
            const int ee = e >= 24 * 8 ? e - 24 * 8 : e;
            *reinterpret_cast<f32x4*>(img + (L + 8) * 128 + ee * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- Q fragments (B operands of S^T = K . Q^T: lane (q = kk, hh) holds Q[q][16 s + 8 hh ..]) of query tiles 2 w, 2 w + 1
    bf16x8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const __bf16* qrowp = seq + (size_t)(32 * (2 * wave + qt) + kk) * ld + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qt][s] = *reinterpret_cast<const bf16x8*>(qrowp + 16 * s);
    }
    FW2_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    FW2_STAMP(2);

    // ---- lane-constant LDS bases (attention_fused.h): K rows as the A operand, V^T by transposed reads
    const char* rowb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rowb[ks] = kimg + kk * 128 + (((2 * ks + hh) ^ fz_sig(kk)) << 4);
    const char* trb[2];
    {
        const int rlo = 4 * hh + (i16 >> 2), pq = i16 & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) trb[e] = vimg + rlo * 128 + ((((2 * (g & 1) + (pq >> 1)) ^ fz_sig(rlo)) ^ (4 * e)) << 4) + (pq & 1) * 8;
    }
    auto vtr = [&](int toff, int sp, int dh) -> bf16x8 {          // V^T fragment: rows = d 32 dh .., contraction = keys of tile offset toff, half sp
        const char* a0 = trb[dh ^ sp] + toff + sp * 2048;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    };
    auto load_k = [&](int kt, bf16x8 (&kf)[4]) {
        if ((P4P_FW2_KO & 32) && kt > 0) return;
        const int toff = kt * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(rowb[ks] + toff);
    };
    auto load_v = [&](int kt, bf16x8 (&vf)[2][2]) {
        if ((P4P_FW2_KO & 32) && kt > 0) return;
        const int toff = kt * 4096;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) vf[sp][dh] = vtr(toff, sp, dh);
    };
    const __bf16 one = f2bf(1.f);
    bf16x8 ones = {one, one, one, one, one, one, one, one};
    asm volatile("" : "+v"(ones));                     // in vector registers for the whole walk (hipcc keeps it in SGPRs and copies per product)

    // An online-softmax chain = one query tile's state: O^T accumulators, the running denominator as a 17th accumulator row block
    // (ones . P^T: every row the same sum), the running maximum, the scores of the tile in flight and its numerators as a B operand.
    // Three steps per key tile: mma_s (matrix pipe) -> softmax (vector pipe) -> mma_pv (matrix pipe).
    struct Chain {
        f32x16 o[2], l, s;
        bf16x8 pb[2];
        float m;
    };
    auto chain_init = [&](Chain& X) {
        X.m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { X.o[0][r] = 0.f; X.o[1][r] = 0.f; X.l[r] = 0.f; }
    };
    auto mma_s = [&](Chain& X, const bf16x8 (&qq)[4], const bf16x8 (&kf)[4]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X.s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (P4P_FW2_KO & 8) { X.s[ks] += bf2f(kf[ks][0]) * bf2f(qq[ks][0]); continue; }
            X.s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qq[ks], X.s, 0, 0, 0);
        }
    };
    auto mma_pv = [&](Chain& X, const bf16x8 (&vf)[2][2], bool last) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            if (last && sp == 1) continue;              // no valid key in the second half of the last tile
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                if (P4P_FW2_KO & 4) { X.o[dh][sp] += bf2f(vf[sp][dh][0]) * bf2f(X.pb[sp][dh]); continue; }
                X.o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[sp][dh], X.pb[sp], X.o[dh], 0, 0, 0);
            }
            if (P4P_FW2_KO & 2) { X.l[0] += bf2f(X.pb[sp][0]); continue; }
            X.l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, X.pb[sp], X.l, 0, 0, 0);
        }
    };
    // The matrix block of a step: the six P . V products of chain X's finished tile and the four K . Q products of its next one, in a FIXED
    // order (pins: empty asm statements tying an accumulator to the next product's operand).  The four score products form one
    // dependent chain (a 32x32x16 product that accumulates on the one before it issues ~60 cycles after it, not 32: priced by knock-out,
    // 240 cycles per tile); the P . V products between them fill those gaps.  Left alone hipcc emits P.V x3, scores x4, P.V x3.
    auto mma_block = [&](Chain& X, const bf16x8 (&qq)[4], const bf16x8 (&kf)[4], const bf16x8 (&vf)[2][2], bool last_pv) {
#define FW2_PIN(acc, next) asm volatile("" : "+v"(acc), "+v"(next))
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        X.o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], X.pb[0], X.o[0], 0, 0, 0);
        X.o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1], X.pb[0], X.o[1], 0, 0, 0);
        X.l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, X.pb[0], X.l, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qq[0], sc, 0, 0, 0);
        if (!last_pv) {
            FW2_PIN(sc, X.o[0]);                       // a pin names the accumulator just written and the next product's: no operand is copied
            X.o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], X.pb[1], X.o[0], 0, 0, 0);
            FW2_PIN(X.o[0], sc);
        }
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qq[1], sc, 0, 0, 0);
        if (!last_pv) {
            FW2_PIN(sc, X.o[1]);
            X.o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][1], X.pb[1], X.o[1], 0, 0, 0);
            FW2_PIN(X.o[1], sc);
        }
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qq[2], sc, 0, 0, 0);
        if (!last_pv) {
            FW2_PIN(sc, X.l);
            X.l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, X.pb[1], X.l, 0, 0, 0);
            FW2_PIN(X.l, sc);
        }
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3], qq[3], sc, 0, 0, 0);
        X.s = sc;
#undef FW2_PIN
    };
    // the decision (row maximum, rescale when some lane's maximum grew): a short block of its own in front of the long one
    auto softmax_max = [&](Chain& X, bool last) {
        if (last) {                                     // the last key tile holds key L alone: register 0 of the lower lane half
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r > 0 || hh) X.s[r] = -INFINITY;
        }
        float mx = X.s[0];
#pragma unroll
        for (int r = 1; r < ((P4P_FW2_KO & 16) ? 2 : 16); ++r) mx = fmaxf(mx, X.s[r]);
        {
            float a = mx, bb = mx;
            lane_swap32(a, bb);                         // lanes l and l ^ 32 hold the two halves of a query's scores
            mx = fmaxf(a, bb) * c;
        }
        if (__builtin_amdgcn_ballot_w64(mx > X.m) != 0) {
            const float mnew = fmaxf(X.m, mx);
            const float alpha = __builtin_amdgcn_exp2f(X.m - mnew);
            X.m = mnew;
#pragma unroll
            for (int r = 0; r < 16; ++r) { X.o[0][r] *= alpha; X.o[1][r] *= alpha; X.l[r] *= alpha; }
        }
    };
    auto softmax_exp = [&](Chain& X) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X.s[r] = (P4P_FW2_KO & 1) ? __builtin_fmaf(X.s[r], c, -X.m) : __builtin_amdgcn_exp2f(__builtin_fmaf(X.s[r], c, -X.m));
        fz_pack(X.s, X.pb);
#if !(P4P_FW2_KO & 128)
        // the step's interleave, spelled out: ten matrix products, each followed by its share of the 16 fma + 16 exp2 + 8 cvt_pk
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
#endif
        // pinned here: left alone, hipcc sinks the exponentials into the block of the P.V product that uses them, one step later, beside
        // the other chain's -- and leaves this step's matrix block without vector work
        asm volatile("" : "+v"(X.pb[0]), "+v"(X.pb[1]));
    };

    // ---- the key tiles against this wave's two query tiles A, B: no barrier.  The chains run half a step apart so that every vector
    // block has a matrix block beside it that does not depend on it:
    //     step 2 kt    : softmax A(kt)   beside   P.V of B(kt - 1), scores of B(kt)        K(kt) and V(kt - 1) fragments
    //     step 2 kt + 1: softmax B(kt)   beside   P.V of A(kt),     scores of A(kt + 1)    V(kt) and K(kt + 1) fragments
    if (!(P4P_FW2_KO & 256) || wave < 4) {             // (bit 8: one wave per SIMD computes, the other four idle at the barrier)
        Chain A, Bc;
        chain_init(A);
        chain_init(Bc);
        bf16x8 kf[4], vf[2][2];
        load_k(0, kf);
        mma_s(A, qf[0], kf);
        // step 0
        softmax_max(A, false);
        softmax_exp(A);
        mma_s(Bc, qf[1], kf);
        load_v(0, vf);
#pragma unroll 1
        for (int kt = 0; kt < NT - 1; ++kt) {
            // step 2 kt + 1
            softmax_max(Bc, false);
            load_k(kt + 1, kf);
            mma_block(A, qf[0], kf, vf, false);      // (in front of the exponentials in program order: their pin is the last of the step)
            softmax_exp(Bc);
            // step 2 kt + 2
            softmax_max(A, false);
            bf16x8 vn[2][2];
            load_v(kt + 1, vn);
            mma_block(Bc, qf[1], kf, vf, false);
            softmax_exp(A);
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[e >> 1][e & 1] = vn[e >> 1][e & 1];
        }
        // step 2 NT - 1: softmax B(NT - 1) beside P.V of A(NT - 1), scores of A(NT) (the last tile: key L alone)
        softmax_max(Bc, false);
        load_k(NT, kf);
        mma_block(A, qf[0], kf, vf, false);
        softmax_exp(Bc);
        // step 2 NT
        softmax_max(A, true);
        {
            bf16x8 vn[2][2];
            load_v(NT, vn);
            mma_block(Bc, qf[1], kf, vf, false);
            softmax_exp(A);
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[e >> 1][e & 1] = vn[e >> 1][e & 1];
        }
        // step 2 NT + 1, and the drain
        softmax_max(Bc, true);
        softmax_exp(Bc);
        mma_pv(A, vf, true);
        mma_pv(Bc, vf, true);
        FW2_STAMP(3);
        FW2_STAMP_WAVES();
        // outputs: O / l as bf16, lse2 = m + log2(l)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            Chain& X = qt ? Bc : A;
            const float inv = 1.0f / X.l[0];
            const int q = 32 * (2 * wave + qt) + kk;
            __bf16* dst = p.out + ((size_t)b * T + q) * D + h * HD + 4 * hh;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const bf16x4 w = {f2bf(X.o[dh][4 * jj] * inv), f2bf(X.o[dh][4 * jj + 1] * inv), f2bf(X.o[dh][4 * jj + 2] * inv),
                                      f2bf(X.o[dh][4 * jj + 3] * inv)};
                    if (!(P4P_FW2_KO & 64) || w[0] == f2bf(12345.f)) *reinterpret_cast<bf16x4*>(dst + 32 * dh + 8 * jj) = w;
                }
            if (hh == 0 && p.lse2) p.lse2[(size_t)bh * T + q] = X.m + __builtin_amdgcn_logf(X.l[0]);
        }
    }
    FW2_STAMP(4);
    // ---- the odd query (row L) against key tiles 2 w, 2 w + 1 (wave 7: and the last): every column of the B operand is that row
    Chain O2;
    chain_init(O2);
    {
        bf16x8 ql[4];
        const __bf16* qrowp = seq + (size_t)L * ld + h * HD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) ql[s] = *reinterpret_cast<const bf16x8*>(qrowp + 16 * s);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            bf16x8 kf[4], vf[2][2];
            load_k(2 * wave + e, kf);
            load_v(2 * wave + e, vf);
            mma_s(O2, ql, kf);
            softmax_max(O2, false);
            softmax_exp(O2);
            mma_pv(O2, vf, false);
        }
        if (wave == NW - 1) {
            bf16x8 kf[4], vf[2][2];
            load_k(NT, kf);
            load_v(NT, vf);
            mma_s(O2, ql, kf);
            softmax_max(O2, true);
            softmax_exp(O2);
            mma_pv(O2, vf, true);
        }
    }
    FW2_STAMP(5);
    // ---- the odd query: this wave's partial (lane kk == 0 of both halves: d = 32 dh + 8 jj + 4 hh ..) -> LDS, combined by wave 0
    if (kk == 0) {
        float* pw = part + wave * 68;
        if (hh == 0) { pw[0] = O2.m; pw[1] = O2.l[0]; }
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *reinterpret_cast<f32x4*>(pw + 4 + 32 * dh + 8 * jj + 4 * hh) = f32x4{O2.o[dh][4 * jj], O2.o[dh][4 * jj + 1], O2.o[dh][4 * jj + 2], O2.o[dh][4 * jj + 3]};
    }
    __syncthreads();                                   // partials visible; every wave is done with the images
    if (wave == 0) {                                   // lane = column d
        float M = part[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, part[w * 68]);
        float lsum = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = __builtin_amdgcn_exp2f(part[w * 68] - M);
            lsum += part[w * 68 + 1] * f;
            acc += part[w * 68 + 4 + lane] * f;
        }
        p.out[((size_t)b * T + L) * D + h * HD + lane] = f2bf(acc * (1.0f / lsum));
        if (lane == 0 && p.lse2) p.lse2[(size_t)bh * T + L] = M + __builtin_amdgcn_logf(lsum);
    }
    FW2_STAMP(6);
    if (!PERSIST || bh + (int)gridDim.x >= p.B * p.H) break;
    __syncthreads();                                   // `part` and the images are free for the next pair
    }
}
