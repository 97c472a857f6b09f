// Attention forward for T = 32 NT + 1 with NT = 16 (T = 513: N = 2048 points): one workgroup of 8 waves per CU walks its (batch, head)
// pairs; K and V of a pair sit in LDS as two key HALVES (keys 0-255 | keys 256-512), each half's image refilled by LDS-DMA for the NEXT pair
// while the other half is being computed on, so that staging never stops the matrix pipe.  Opt-in (p4p_set_attention_fwd_variant(1));
// round 3 ran this shape as attn_fwd_q_kernel<2>: 64-key tiles streamed through LDS behind a barrier each, 16x16x32 products.  Here:
//   * v_mfma_f32_32x32x16_bf16 with the QUERY on the lane (S^T[key][q] = K . Q^T): a lane owns one query, its 32 scores of a key tile
//     are 16 registers (+ 16 in the partner lane l ^ 32), so the row maximum is 8 v_max3 + one v_permlane32_swap, the numerators P^T are the
//     B operand of O^T[d][q] += V^T . P^T as they stand (the accumulator tile IS the operand), and twice the MACs per LDS byte;
//   * wave w owns query tiles 2 w, 2 w + 1 (Q fragments in registers, straight from global memory) and walks the 17 key tiles of 32
//     (the last one holds the odd key 512 alone) as two online-softmax chains HALF A STEP APART: while the vector pipe works on one
//     chain's exponentials, the matrix pipe runs the other chain's P . V and its next scores (an order hipcc does not find: pins);
//   * two barriers per pair: E (half 1 landed, half 0 free -> refill it for the next pair) and T (half 1 free, partials written);
//   * the odd QUERY (row 512, the cls token) has no wave: wave w runs it against key tiles 2 w, 2 w + 1 (wave 7: also tile 16) as a
//     third chain with one valid lane -- waves 0-3 at the top of the pair (half 0), waves 4-7 at its end (half 1) -- and the eight
//     (max, sum, accumulator) partials meet in LDS, combined by wave 0 behind barrier T.
// What bounds it (knock-out pricing, NOTES.md A.6): at head_dim 64 a 32x32 tile is 10 matrix products (320 cycles of the pipe) against
// 16 v_exp_f32 at 16 issue cycles + 16 fma + 8 cvt_pk + the maximum: ~420 issue cycles of the vector pipe -- the loop is exp2-bound.
// Arithmetic contract (tests/emulate.py, mode 'online32'): online softmax over 32-key tiles in ascending order, numerators
// p = exp2(fma(s, c, -m)) rounded to bf16 for the P.V product AND for the denominator (the row sums are a 17th accumulator row block:
// an all-ones operand against P^T), the reference m moved (and the accumulators rescaled) only when some query of the 32 outgrew it by
// more than 2^4, one division at the end; the odd query:
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
#else
#define FW2_STAMP(k) do { } while (0)
#endif

// The running reference m of a chain follows the row maximum only when some query of the tile outgrew it by more than FW2_THR (base-2
// exponent units): numerators then reach 2^FW2_THR instead of 1 -- the same relative rounding in bf16, the same quotient o / l -- and the
// rescale of the 48 accumulator registers (three multiplies per score: with 32 queries deciding together it ran on nearly every tile)
// becomes rare.  The flash-attention 'lazy rescale'; tests/emulate.py states it ('online32', thr).
#define FW2_THR 4.0f
#define FW2_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)        // vmcnt(0) alone
#define FW2_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)      // lgkmcnt(0) alone

template <int NT>
struct Fw2Lds {
    static constexpr int L = 32 * NT, HK = L / 2;                     // keys per half (the second half: + the odd key's tile)
    static constexpr int OFF_KA = 0, OFF_VA = HK * 128;               // half 0: K | V images of keys 0 .. HK - 1
    static constexpr int OFF_KB = 2 * HK * 128, OFF_VB = OFF_KB + (HK + 32) * 128;   // half 1: keys HK .. L - 1, then the tile of key L
    static constexpr int OFF_PART = OFF_VB + (HK + 32) * 128;         // partials of the odd query: [2 pair parities][8 waves][4 + 64] floats
    static constexpr int TOTAL = OFF_PART + 2 * 8 * 68 * 4;
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <int NT, bool PERSIST = true>
__global__ __launch_bounds__(512, 1) void attn_fwd_kv_kernel(AttnParams p) {
    using LY = Fw2Lds<NT>;
    constexpr int L = LY::L, NW = 8, HT = NT / 2;     // HT key tiles per half
    static_assert(NT == 2 * NW, "two query tiles per wave");
    __shared__ __attribute__((aligned(1024))) char lds[LY::TOTAL];
    const int T = p.T, D = p.D;                       // T = L + 1
    const size_t ld = (size_t)3 * D;
    const float c = p.scale * 1.4426950408889634f;
    const unsigned qrow = (unsigned)ld * 2;           // bytes per row of qkv
    const int npairs = p.B * p.H;
    if (p.stagger_phases > 1 && (int)blockIdx.x < p.stagger_n) {       // first round in shifted groups (attn_bwd_fused_kernel)
        const int ph = blockIdx.x % p.stagger_phases;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)ph * p.stagger_cycles) __builtin_amdgcn_s_sleep(16);
    }
    // LDS offset of key tile kt's K / V image (tiles 0 .. HT - 1 in half 0, HT .. NT in half 1)
    auto kofs = [](int kt) { return kt * 4096 + (kt >= HT ? LY::OFF_KB - HT * 4096 : LY::OFF_KA); };
    auto vofs = [](int kt) { return kt * 4096 + (kt >= HT ? LY::OFF_VB - HT * 4096 : LY::OFF_VA); };
    // the LDS-DMAs of one half of pair (b, h): 1-KiB pieces of 8 rows, fz_sig swizzle, 4 pieces of K and 4 of V per wave; half 1: the
    // piece of rows L .. L + 7 (all of them row L) by wave 0 (K) and wave 7 (V); rows L + 8 .. L + 31 of that tile stay zero
    auto stage_half = [&](int bh_, int half, int wave, int lane) {
        const int b_ = bh_ / p.H, h_ = bh_ - b_ * p.H;
        const __bf16* seq_ = p.qkv + (size_t)b_ * T * ld;
        const FzSrc sq = fz_src(lane, qrow);
        const char* kbase = reinterpret_cast<const char*>(seq_ + D + h_ * HD) + (size_t)(half * LY::HK) * qrow;
        const char* vbase = kbase + (size_t)D * 2;
        char* kdst = lds + (half ? LY::OFF_KB : LY::OFF_KA);
        char* vdst = lds + (half ? LY::OFF_VB : LY::OFF_VA);
#pragma unroll
        for (int i = 0; i < LY::HK / 8 / NW; ++i) {
            const int pc = wave + NW * i;
            fz_dma(kbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], kdst + pc * 1024);
        }
#pragma unroll
        for (int i = 0; i < LY::HK / 8 / NW; ++i) {
            const int pc = wave + NW * i;
            fz_dma(vbase + (size_t)(8 * pc) * qrow, ((pc >> 1) & 1) ? sq.par[1] : sq.par[0], vdst + pc * 1024);
        }
        if (half) {
            if (wave == 0) fz_dma(kbase + (size_t)LY::HK * qrow, sq.last, kdst + (LY::HK / 8) * 1024);
            if (wave == NW - 1) fz_dma(vbase + (size_t)LY::HK * qrow, sq.last, vdst + (LY::HK / 8) * 1024);
        }
    };
    bf16x8 qf[2][4], ql[4];                            // Q fragments of this wave's two query tiles; of the odd query (row L)
    // Q fragments (B operands of S^T = K . Q^T: lane (q = kk, hh) holds Q[q][16 s + 8 hh ..]) of pair bh_; the odd row of pair bl_
    auto load_q = [&](int bh_, int bl_, int wave, int lane) {
        const int kk_ = lane & 31, hh_ = lane >> 5;
        const int b_ = bh_ / p.H, h_ = bh_ - b_ * p.H;
        const __bf16* seq_ = p.qkv + (size_t)b_ * T * ld;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const __bf16* qrowp = seq_ + (size_t)(32 * (2 * wave + qt) + kk_) * ld + h_ * HD + 8 * hh_;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[qt][s] = *reinterpret_cast<const bf16x8*>(qrowp + 16 * s);
        }
        const int bl = bl_ / p.H, hl = bl_ - bl * p.H;
        const __bf16* qrowp = p.qkv + ((size_t)bl * T + L) * ld + hl * HD + 8 * hh_;
#pragma unroll
        for (int s = 0; s < 4; ++s) ql[s] = *reinterpret_cast<const bf16x8*>(qrowp + 16 * s);
    };
    auto pin_q = [&]() {                               // the fragments have arrived (the compiler's wait sits here, in front of the DMAs that follow)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[0][s]), "+v"(qf[1][s]), "+v"(ql[s]));
    };

    // ---- before the walk: the zero rows of the odd key's tile, half 0 and the Q fragments of the first pair
    {
        const int t0 = threadIdx.x;
        for (int e = t0; e < 2 * 24 * 8; e += NW * 64) {
            char* img = lds + (e >= 24 * 8 ? LY::OFF_VB : LY::OFF_KB);
            const int ee = e >= 24 * 8 ? e - 24 * 8 : e;
            *reinterpret_cast<f32x4*>(img + (LY::HK + 8) * 128 + ee * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int lane0 = t0 & 63, wave0 = __builtin_amdgcn_readfirstlane(t0 >> 6);
        stage_half(blockIdx.x, 0, wave0, lane0);
        load_q(blockIdx.x, blockIdx.x, wave0, lane0);
        FW2_WAIT_VM0();
        FW2_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        pin_q();
        stage_half(blockIdx.x, 1, wave0, lane0);
    }
    int par = 0;
    for (int bh = blockIdx.x;; bh += gridDim.x) {
    int t = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(t));           // nothing derived from the thread index is invariant of the walk (attention_fused.h)
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = bh / p.H, h = bh - b * p.H;
    const int kk = lane & 31, hh = lane >> 5, i16 = lane & 15, g = lane >> 4;
    const bool has_next = PERSIST && bh + (int)gridDim.x < npairs;
    const int bh_next = has_next ? bh + (int)gridDim.x : bh;
    float* part = reinterpret_cast<float*>(lds + LY::OFF_PART) + par * (NW * 68);
    FW2_STAMP(0);

    // ---- lane-constant LDS bases (attention_fused.h): K rows as the A operand, V^T by transposed reads; + kofs / vofs of the tile
    const char* rowb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rowb[ks] = lds + kk * 128 + (((2 * ks + hh) ^ fz_sig(kk)) << 4);
    const char* trb[2];
    {
        const int rlo = 4 * hh + (i16 >> 2), pq = i16 & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) trb[e] = lds + rlo * 128 + ((((2 * (g & 1) + (pq >> 1)) ^ fz_sig(rlo)) ^ (4 * e)) << 4) + (pq & 1) * 8;
    }
    auto vtr = [&](int toff, int sp, int dh) -> bf16x8 {          // V^T fragment: rows = d 32 dh .., contraction = keys of the tile at toff, half sp
        const char* a0 = trb[dh ^ sp] + toff + sp * 2048;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, r);
    };
    auto load_k = [&](int kt, bf16x8 (&kf)[4]) {
        const int toff = kofs(kt);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(rowb[ks] + toff);
    };
    auto load_v = [&](int kt, bf16x8 (&vf)[2][2]) {
        const int toff = vofs(kt);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) vf[sp][dh] = vtr(toff, sp, dh);
    };
    // the all-ones A operand, made inside the walk (a value made in front of it is spilled around the key-tile loops and its reload at the top
    // of a pair waits with vmcnt(0) -- for the LDS-DMAs just issued)
    bf16x8 ones;
    {
        unsigned one2;
        asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(one2));
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 o4 = {one2, one2, one2, one2};
        ones = __builtin_bit_cast(bf16x8, o4);
    }

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
            X.s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qq[ks], X.s, 0, 0, 0);
        }
    };
    auto mma_pv = [&](Chain& X, const bf16x8 (&vf)[2][2], bool last) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            if (last && sp == 1) continue;              // no valid key in the second half of the last tile
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                X.o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[sp][dh], X.pb[sp], X.o[dh], 0, 0, 0);
            }
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
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, X.s[r]);
        {
            float a = mx, bb = mx;
            lane_swap32(a, bb);                         // lanes l and l ^ 32 hold the two halves of a query's scores
            mx = fmaxf(a, bb) * c;
        }
        if (__builtin_amdgcn_ballot_w64(mx > X.m + FW2_THR) != 0) {
            const float mnew = fmaxf(X.m, mx);
            const float alpha = __builtin_amdgcn_exp2f(X.m - mnew);
            X.m = mnew;
#pragma unroll
            for (int r = 0; r < 16; ++r) { X.o[0][r] *= alpha; X.o[1][r] *= alpha; X.l[r] *= alpha; }
        }
    };
    auto softmax_exp = [&](Chain& X) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X.s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(X.s[r], c, -X.m));
        fz_pack(X.s, X.pb);
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
        // pinned here: left alone, hipcc sinks the exponentials into the block of the P.V product that uses them, one step later, beside
        // the other chain's -- and leaves this step's matrix block without vector work
        asm volatile("" : "+v"(X.pb[0]), "+v"(X.pb[1]));
    };

    // One step of the two-chain schedule: chain X's softmax beside chain Y's matrix block.  The numerators are taken against the maximum SO
    // FAR (no wait for this tile's maximum: the vector block starts at once), the row maximum runs beside them, and only when some
    // lane's maximum grew (the first tiles of a chain, a late dominant key) are the accumulators rescaled and the numerators taken
    // again -- the same values as deciding first, by another route.
    auto step = [&](Chain& X, bool last_x, Chain& Y, const bf16x8 (&qy)[4], const bf16x8 (&kf)[4], const bf16x8 (&vf)[2][2]) {
        if (last_x) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r > 0 || hh) X.s[r] = -INFINITY;
        }
        {
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(X.s[r], c, -X.m));
            fz_pack(e, X.pb);
        }
        mma_block(Y, qy, kf, vf, false);
        float mx = X.s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, X.s[r]);
        {
            float a = mx, bb = mx;
            lane_swap32(a, bb);
            mx = fmaxf(a, bb) * c;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        asm volatile("" : "+v"(X.pb[0]), "+v"(X.pb[1]));
        if (__builtin_amdgcn_ballot_w64(mx > X.m + FW2_THR) != 0) {
            const float mnew = fmaxf(X.m, mx);
            const float alpha = __builtin_amdgcn_exp2f(X.m - mnew);
            X.m = mnew;
#pragma unroll
            for (int r = 0; r < 16; ++r) { X.o[0][r] *= alpha; X.o[1][r] *= alpha; X.l[r] *= alpha; }
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(X.s[r], c, -X.m));
            fz_pack(e, X.pb);
        }
    };
    // the odd query against one key tile (a chain of its own, one step after the other), and its partial into LDS
    auto odd_tile = [&](Chain& X, int kt, bool last) {
        bf16x8 kf[4], vf[2][2];
        load_k(kt, kf);
        load_v(kt, vf);
        mma_s(X, ql, kf);
        softmax_max(X, last);
        softmax_exp(X);
        mma_pv(X, vf, last);
    };
    auto odd_store = [&](Chain& X) {                   // lane kk == 0 of both halves holds d = 32 dh + 8 jj + 4 hh ..
        if (kk == 0) {
            float* pw = part + wave * 68;
            if (hh == 0) { pw[0] = X.m; pw[1] = X.l[0]; }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *reinterpret_cast<f32x4*>(pw + 4 + 32 * dh + 8 * jj + 4 * hh) = f32x4{X.o[dh][4 * jj], X.o[dh][4 * jj + 1], X.o[dh][4 * jj + 2], X.o[dh][4 * jj + 3]};
        }
    };
    // ---- waves 0-3: the odd query against key tiles 2 w, 2 w + 1 of half 0 (every column of the B operand is that row)
    if (wave < NW / 2) {
        Chain O2;
        chain_init(O2);
        odd_tile(O2, 2 * wave, false);
        odd_tile(O2, 2 * wave + 1, false);
        odd_store(O2);
    }
    FW2_STAMP(1);

    // ---- the key tiles against this wave's two query tiles A, B.  The chains run half a step apart so that every vector block has a
    // matrix block beside it that does not depend on it:
    //     step 2 kt    : softmax A(kt)   beside   P.V of B(kt - 1), scores of B(kt)        K(kt) and V(kt - 1) fragments
    //     step 2 kt + 1: softmax B(kt)   beside   P.V of A(kt),     scores of A(kt + 1)    V(kt) and K(kt + 1) fragments
    // Half 0 is last read in step 2 HT - 2 (the V fragments of tile HT - 1), half 1 first behind it (the K fragments of tile HT): barrier E between them.
    Chain A, Bc;
    chain_init(A);
    chain_init(Bc);
    {
        bf16x8 kf[4], vf[2][2];
        load_k(0, kf);
        mma_s(A, qf[0], kf);
        softmax_max(A, false);                         // step 0
        softmax_exp(A);
        mma_s(Bc, qf[1], kf);
        load_v(0, vf);
        load_k(1, kf);
        auto two_steps = [&](int kt, bool next_k) {
            step(Bc, false, A, qf[0], kf, vf);          // step 2 kt + 1 (K(kt + 1) in kf, V(kt) in vf)
            bf16x8 vn[2][2];
            load_v(kt + 1, vn);
            step(A, false, Bc, qf[1], kf, vf);          // step 2 kt + 2
            if (next_k) load_k(kt + 2, kf);             // behind the last product that reads K(kt + 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[e >> 1][e & 1] = vn[e >> 1][e & 1];
        };
#pragma unroll 1
        for (int kt = 0; kt < HT - 2; ++kt) two_steps(kt, true);
        two_steps(HT - 2, false);                      // (K(HT) is in half 1: behind the barrier)
        // ---- barrier E: half 1 of this pair has landed (issued behind barrier T of the pair before); nobody reads half 0 any more
        FW2_STAMP(2);
        FW2_WAIT_VM0();
        FW2_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (has_next) stage_half(bh_next, 0, wave, lane);
        FW2_STAMP(3);
        load_k(HT, kf);
#pragma unroll 1
        for (int kt = HT - 1; kt < NT - 1; ++kt) two_steps(kt, true);
        // step 2 NT - 1: softmax B(NT - 1) beside P.V of A(NT - 1), scores of A(NT) (the last tile: key L alone; K(NT) is in kf)
        step(Bc, false, A, qf[0], kf, vf);
        {
            bf16x8 vn[2][2];
            load_v(NT, vn);
            step(A, true, Bc, qf[1], kf, vf);           // step 2 NT
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[e >> 1][e & 1] = vn[e >> 1][e & 1];
        }
        softmax_max(Bc, true);                         // step 2 NT + 1, and the drain
        softmax_exp(Bc);
        mma_pv(A, vf, true);
        mma_pv(Bc, vf, true);
    }
    FW2_STAMP(4);
    // ---- half 0 of the next pair has landed (issued behind barrier E, a half ago); its Q fragments are asked for before the stores below,
    // so that waiting for them does not wait for the stores (vmcnt retires in order)
    FW2_WAIT_VM0();
    if (has_next || wave >= NW / 2) load_q(bh_next, wave < NW / 2 ? bh_next : bh, wave, lane);
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
                *reinterpret_cast<bf16x4*>(dst + 32 * dh + 8 * jj) = w;
            }
        if (hh == 0 && p.lse2) p.lse2[(size_t)bh * T + q] = X.m + __builtin_amdgcn_logf(X.l[0]);
    }
    // ---- waves 4-7: the odd query against key tiles 2 w, 2 w + 1 of half 1 (wave 7: and the tile of the odd key)
    if (wave >= NW / 2) {
        Chain O2;
        chain_init(O2);
        odd_tile(O2, 2 * wave, false);
        odd_tile(O2, 2 * wave + 1, false);
        if (wave == NW - 1) odd_tile(O2, NT, true);
        odd_store(O2);
    }
    FW2_STAMP(5);
    // ---- barrier T: the partials are in LDS, nobody reads half 1 any more
    FW2_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if (has_next) {
        pin_q();
        stage_half(bh_next, 1, wave, lane);
    }
    if (wave == 0) {                                   // the odd query's eight partials, combined in wave order; lane = column d
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
    if (!has_next) break;
    par ^= 1;
    }
}
