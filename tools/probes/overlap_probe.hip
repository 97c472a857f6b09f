// How much of a GELU epilogue can hide under the K loop of ANOTHER tile when two 4-wave groups of one workgroup take turns (one wave of
// each group per SIMD)?  Synthetic: per wave and tile 128 x 64 outputs = 128 accumulator registers; "K loop" = the MFMAs of K = 768 on
// register operands (no LDS, no HBM: the matrix pipe and the issue port alone), "epilogue" = gelu_and_dgelu of gemm_tn.h on the 128
// accumulators + cvt_pk.  Modes: 0 serial (all 8 waves: K loop, then epilogue), 1 ping-pong (groups half a period apart, a barrier
// per phase), 2 K loops only, 3 epilogues only.  SHAPE 0: v_mfma_f32_16x16x32_bf16 (holds the vector issue 8 of 16 cycles),
// 1: v_mfma_f32_32x32x16_bf16 (8 of 32).        tools/overlap_probe.py builds and runs it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void phi_and_e(float x, float& Phi, float& e) {
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));
    e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);
    float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
    const float half_tail = (t * poly) * e;
    Phi = 0.5f + __builtin_copysignf(0.5f - half_tail, x);
}

template <int SHAPE>
struct Acc {
    f32x4 a4[SHAPE == 0 ? 32 : 1];
    f32x16 a16[SHAPE == 1 ? 8 : 1];
};

template <int SHAPE>
__device__ __forceinline__ void kloop(Acc<SHAPE>& acc, const bf16x8 (&a)[8], const bf16x8 (&b)[4]) {
    if (SHAPE == 0) {
#pragma unroll 1
        for (int kt = 0; kt < 24; ++kt) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc.a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[i >> 3], acc.a4[i], 0, 0, 0);
        }
    } else {
#pragma unroll 1
        for (int kt = 0; kt < 48; ++kt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc.a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i & 3], acc.a16[i], 0, 0, 0);
        }
    }
}
template <int SHAPE>
__device__ __forceinline__ unsigned epilogue(Acc<SHAPE>& acc) {
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < 128; i += 2) {
        float x0 = SHAPE == 0 ? acc.a4[i >> 2][i & 3] : acc.a16[i >> 4][i & 15];
        float x1 = SHAPE == 0 ? acc.a4[(i + 1) >> 2][(i + 1) & 3] : acc.a16[(i + 1) >> 4][(i + 1) & 15];
        float P0, e0, P1, e1;
        phi_and_e(x0, P0, e0);
        phi_and_e(x1, P1, e1);
        const float y0 = x0 * P0, y1 = x1 * P1;
        const float d0 = __builtin_fmaf(x0 * e0, 0.3989422804014327f, P0), d1 = __builtin_fmaf(x1 * e1, 0.3989422804014327f, P1);
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 py = {(__bf16)y0, (__bf16)y1}, pd = {(__bf16)d0, (__bf16)d1};
        sum ^= __builtin_bit_cast(unsigned, py) + __builtin_bit_cast(unsigned, pd);
        if (SHAPE == 0) { acc.a4[i >> 2][i & 3] = y0 * 1e-3f; acc.a4[(i + 1) >> 2][(i + 1) & 3] = y1 * 1e-3f; }
        else { acc.a16[i >> 4][i & 15] = y0 * 1e-3f; acc.a16[(i + 1) >> 4][(i + 1) & 15] = y1 * 1e-3f; }
    }
    return sum;
}

// MODE 4: every wave holds TWO half-size accumulator sets (2 x 64 registers: a 256 x 128 workgroup tile each); the K loop of one set carries the
// epilogue of the other in slices (8 elements behind every 3 K tiles), in the wave's OWN instruction stream.
__device__ __forceinline__ unsigned epi8(f32x4 (&acc)[16], int s) {
    unsigned sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float x0 = acc[2 * s + (j >> 2)][j & 3], x1 = acc[2 * s + ((j + 1) >> 2)][(j + 1) & 3];
        float P0, e0, P1, e1;
        phi_and_e(x0, P0, e0);
        phi_and_e(x1, P1, e1);
        const float y0 = x0 * P0, y1 = x1 * P1;
        const float d0 = __builtin_fmaf(x0 * e0, 0.3989422804014327f, P0), d1 = __builtin_fmaf(x1 * e1, 0.3989422804014327f, P1);
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 py = {(__bf16)y0, (__bf16)y1}, pd = {(__bf16)d0, (__bf16)d1};
        sum ^= __builtin_bit_cast(unsigned, py) + __builtin_bit_cast(unsigned, pd);
        acc[2 * s + (j >> 2)][j & 3] = y0 * 1e-3f;
        acc[2 * s + ((j + 1) >> 2)][(j + 1) & 3] = y1 * 1e-3f;
    }
    return sum;
}
template <bool WITH_EPI>
__device__ __forceinline__ unsigned kloop_with_epilogue(f32x4 (&cur)[16], f32x4 (&prev)[16], const bf16x8 (&a)[8], const bf16x8 (&b)[4]) {
    unsigned sum = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) cur[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[i >> 2], cur[i], 0, 0, 0);
        if (WITH_EPI) sum ^= epi8(prev, s);
    }
    return sum;
}
template <int VARIANT>                                   // 0: interleaved, 1: the same work one after the other (K loop, then the 64-element epilogue)
__global__ __launch_bounds__(512, 1) void probe4_kernel(unsigned* out, int tiles) {
    const int t = threadIdx.x;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.01f * ((t + 3 * i + j) % 7 - 3));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.02f * ((t + 5 * i + j) % 5 - 2));
    f32x4 A[16], B[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { A[i] = f32x4{0.f, 0.f, 0.f, 0.f}; B[i] = A[i]; asm volatile("" : "+v"(A[i]), "+v"(B[i])); }
    unsigned sum = 0;
    for (int i = 0; i < tiles; ++i) {                    // per iteration: two half-size tiles = one 256 x 256 tile of the workgroup
        if (VARIANT == 0) {
            sum ^= kloop_with_epilogue<true>(A, B, a, b);
            sum ^= kloop_with_epilogue<true>(B, A, a, b);
        } else {
            sum ^= kloop_with_epilogue<false>(A, B, a, b);
#pragma unroll
            for (int s = 0; s < 8; ++s) sum ^= epi8(A, s);
            sum ^= kloop_with_epilogue<false>(B, A, a, b);
#pragma unroll
            for (int s = 0; s < 8; ++s) sum ^= epi8(B, s);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(A[k]), "+v"(B[k]));
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * 512 + t] = sum;
}

template <int SHAPE, int MODE>
__global__ __launch_bounds__(512, 1) void probe_kernel(unsigned* out, int tiles) {
    const int t = threadIdx.x, wave = t >> 6;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.01f * ((t + 3 * i + j) % 7 - 3));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.02f * ((t + 5 * i + j) % 5 - 2));
    Acc<SHAPE> acc;
#pragma unroll
    for (int i = 0; i < (SHAPE == 0 ? 32 : 1); ++i) acc.a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (SHAPE == 1 ? 8 : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.a16[i][r] = 0.f;
    unsigned sum = 0;
    auto launder = [&]() {                              // the accumulators are opaque to the optimiser
#pragma unroll
        for (int i = 0; i < (SHAPE == 0 ? 32 : 1); ++i) asm volatile("" : "+v"(acc.a4[i]));
#pragma unroll
        for (int i = 0; i < (SHAPE == 1 ? 8 : 1); ++i) asm volatile("" : "+v"(acc.a16[i]));
    };
    launder();
    if (MODE == 0 || MODE == 1) {                      // one code path; MODE 1: group 1 runs one phase (= one barrier) behind group 0
        if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_barrier();
        for (int i = 0; i < tiles; ++i) {
            kloop<SHAPE>(acc, a, b);
            if (MODE == 1) __builtin_amdgcn_s_barrier();
            sum ^= epilogue<SHAPE>(acc);
            launder();
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 1 && wave < 4) __builtin_amdgcn_s_barrier();
    } else if (MODE == 2) {
        for (int i = 0; i < tiles; ++i) { kloop<SHAPE>(acc, a, b); launder(); __builtin_amdgcn_s_barrier(); }
        sum ^= epilogue<SHAPE>(acc);
    } else {
        for (int i = 0; i < tiles; ++i) {
            launder();
            sum ^= epilogue<SHAPE>(acc);
            __builtin_amdgcn_s_barrier();
        }
    }
    out[blockIdx.x * 512 + t] = sum;
}

template <int SHAPE, int MODE>
static float run_one(unsigned* out, int tiles, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_kernel<SHAPE, MODE>), dim3(grid), dim3(512), 0, 0, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_kernel<SHAPE, MODE>), dim3(grid), dim3(512), 0, 0, out, tiles);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VARIANT>
static float run_four(unsigned* out, int tiles, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe4_kernel<VARIANT>), dim3(grid), dim3(512), 0, 0, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe4_kernel<VARIANT>), dim3(grid), dim3(512), 0, 0, out, tiles);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
extern "C" int overlap_probe4(int tiles, float* ms /* [interleaved, one after the other] */) {
    unsigned* out = nullptr;
    if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    ms[0] = run_four<0>(out, tiles, 256); ms[1] = run_four<1>(out, tiles, 256);
    hipFree(out);
    return hipGetLastError() != hipSuccess;
}

extern "C" int overlap_probe(int tiles, float* ms /* [2 shapes][4 modes] */) {
    unsigned* out = nullptr;
    if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    ms[0] = run_one<0, 0>(out, tiles, 256); ms[1] = run_one<0, 1>(out, tiles, 256); ms[2] = run_one<0, 2>(out, tiles, 256); ms[3] = run_one<0, 3>(out, tiles, 256);
    ms[4] = run_one<1, 0>(out, tiles, 256); ms[5] = run_one<1, 1>(out, tiles, 256); ms[6] = run_one<1, 2>(out, tiles, 256); ms[7] = run_one<1, 3>(out, tiles, 256);
    hipFree(out);
    return hipGetLastError() != hipSuccess;
}
