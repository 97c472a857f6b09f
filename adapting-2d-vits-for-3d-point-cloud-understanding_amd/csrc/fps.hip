// Farthest point sampling for gfx950.  Replaces farthest_point_sampling,
// reference src/models/pix4point.py:8-53 (one Python loop round = ~8 small torch launches that
// re-read (B,N,3) from memory each time).
//
// One workgroup per cloud.  The cloud is read from HBM exactly once (coalesced float stream) into
// an LDS image; every thread keeps its P points and their running minimum distance in registers.
// Each of the S strictly dependent rounds is: broadcast-read the centroid from LDS, P exact
// distance updates, a DPP arg-max inside each wave, one LDS slot per wave and ONE barrier.
//
// Bit-exactness contract (oracle/pix4point_oracle.py: fps_explicit_numpy):
//   d = (dx*dx + dy*dy) + dz*dz  in fp32, no FMA contraction;  running = min(running, d);
//   next = arg-max(running) with the LOWEST index winning ties;  running starts at 1e10f.
// The arg-max is a u64 max over  (float_bits(running) << 32) | (0xFFFFFFFF - index):  running is a
// non-negative float so its bit pattern orders like the value, and the inverted index makes the
// lowest index the largest key.
#include "common.h"

#define FPS_THREADS 256
#define FPS_WAVES (FPS_THREADS / 64)

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ points,
                                                           const int64_t* __restrict__ start_idx,
                                                           int64_t* __restrict__ out_idx, int N, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* slot = reinterpret_cast<u64*>(smem);                 // [2][FPS_WAVES]
    float* raw = reinterpret_cast<float*>(smem + 2 * FPS_WAVES * sizeof(u64));   // [3N] xyz AoS

    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* src = points + (size_t)b * N * 3;
    for (int i = t; i < 3 * N; i += FPS_THREADS) raw[i] = src[i];
    __syncthreads();

    float x[P], y[P], z[P], md[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        int i = j * FPS_THREADS + t;
        int ic = i < N ? i : N - 1;
        x[j] = raw[3 * ic]; y[j] = raw[3 * ic + 1]; z[j] = raw[3 * ic + 2];
        md[j] = 1e10f;
    }

    int far = (int)start_idx[b];
    far = far < 0 ? 0 : (far >= N ? N - 1 : far);
    int64_t* out = out_idx + (size_t)b * S;
    if (t == 0) out[0] = far;

    for (int it = 1; it < S; ++it) {
        const float cx = raw[3 * far], cy = raw[3 * far + 1], cz = raw[3 * far + 2];
        u64 best = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            int i = j * FPS_THREADS + t;
            float d = sqdist3(x[j], y[j], z[j], cx, cy, cz);
            md[j] = d < md[j] ? d : md[j];
            u64 key = ((u64)__builtin_bit_cast(u32, md[j]) << 32) | (u64)(0xFFFFFFFFu - (u32)i);
            key = i < N ? key : 0ull;
            best = key > best ? key : best;
        }
        best = wave_max_u64(best);
        u64* s = slot + (it & 1) * FPS_WAVES;
        if (lane == 0) s[wave] = best;
        __syncthreads();
        u64 m = s[0];
#pragma unroll
        for (int w = 1; w < FPS_WAVES; ++w) { u64 o = s[w]; m = o > m ? o : m; }
        far = (int)(0xFFFFFFFFu - (u32)m);
        if (t == 0) out[it] = far;
    }
}

template <int P>
static int launch_fps(const float* points, const int64_t* start, int64_t* out, int B, int N, int S,
                      hipStream_t stream) {
    size_t lds = 2 * FPS_WAVES * sizeof(u64) + (size_t)3 * N * sizeof(float);
    P4P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<P>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fps_kernel<P>, dim3(B), dim3(FPS_THREADS), lds, stream, points, start, out, N, S);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_fps_f32(const float* points, const int64_t* start_idx, int64_t* out_idx, int B, int N,
                           int S, void* stream) {
    P4P_CHECK(points && start_idx && out_idx, "p4p_fps_f32: null pointer");
    P4P_CHECK(B >= 0 && N >= 1 && S >= 0, "p4p_fps_f32: bad sizes B=%d N=%d S=%d", B, N, S);
    P4P_CHECK(S <= N, "p4p_fps_f32: n_samples S=%d exceeds N=%d (the caller clamps, pix4point.py:23)", S, N);
    P4P_CHECK(N <= 12288, "p4p_fps_f32: N=%d exceeds the LDS-resident limit 12288", N);
    if (B == 0 || S == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int p = ceil_div(N, FPS_THREADS);
    if (p <= 1) return launch_fps<1>(points, start_idx, out_idx, B, N, S, st);
    if (p <= 2) return launch_fps<2>(points, start_idx, out_idx, B, N, S, st);
    if (p <= 4) return launch_fps<4>(points, start_idx, out_idx, B, N, S, st);
    if (p <= 8) return launch_fps<8>(points, start_idx, out_idx, B, N, S, st);
    if (p <= 16) return launch_fps<16>(points, start_idx, out_idx, B, N, S, st);
    if (p <= 32) return launch_fps<32>(points, start_idx, out_idx, B, N, S, st);
    return launch_fps<48>(points, start_idx, out_idx, B, N, S, st);
}
