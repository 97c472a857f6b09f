// Batched point-cloud augmentation for gfx950 (the reference augments one cloud at a time in
// numpy inside DataLoader workers, src/data/scanobjectnn.py:186-205 calling src/data/augment.py:4-191, which starves 8 GPUs).
// One workgroup per cloud, the cloud lives in LDS for the whole chain, every random draw is an INPUT (made by the caller with a
// torch.Generator), so the kernel is a pure function and parity with the numpy chain is testable draw for draw:
//   1. normalize_point_cloud      (augment.py:4-18)    p -= mean(p);  p /= max ||p||   (if the maximum is > 0)
//   2. rotations / scalings / shift, composed by the host into one affine map  p = p W + t
//                                 (random_rotate_point_cloud :158-191, rotate_point_cloud_y/z :76-108, random_scale_point_cloud
//                                  :35-47, random_shift_point_cloud :49-60, scale_point_cloud :193-226, rotate_point_cloud :261-307)
//   3. random_jitter_point_cloud  (:62-74)             p += clip(jitter, -clip, clip)          jitter = sigma * randn, given
//   4. drop_and_replace_with_noise(:110-156)           dropped points <- min + u (max - min) + gauss over the CURRENT bounding box
//   5. random_point_dropout       (:20-33)             dropped points <- point 0
// Per-cloud switches come in `flags` (bit 0 normalise, 1 affine, 2 jitter, 3 drop-and-replace, 4 point dropout).
// HBM traffic: the cloud once in, once out, plus the random tensors that are used.
#include "common.h"

#define AUG_THREADS 256
#define AUG_NORMALIZE 1
#define AUG_AFFINE 2
#define AUG_JITTER 4
#define AUG_REPLACE 8
#define AUG_DROPOUT 16

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red, int t) {
    const int lane = t & 63, wave = t >> 6;
    v = is_max ? wave_max_f32(v) : wave_sum_f32(v);
    __syncthreads();                        // red may still be read from the previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < AUG_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

__global__ __launch_bounds__(AUG_THREADS) void augment_kernel(const float* __restrict__ points, float* __restrict__ out,
                                                              const int32_t* __restrict__ flags,
                                                              const float* __restrict__ affine,      // [B][12]: W row-major (p' = p W), then t
                                                              const float* __restrict__ jitter, float clip,
                                                              const uint8_t* __restrict__ replace_mask,
                                                              const float* __restrict__ replace_u,
                                                              const float* __restrict__ replace_g,
                                                              const uint8_t* __restrict__ dropout_mask, int N, int enabled) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p = reinterpret_cast<float*>(smem);                 // [N][3]
    __shared__ float red[AUG_THREADS / 64];
    const int b = blockIdx.x, t = threadIdx.x;
    const int f = flags[b] & enabled;          // `enabled`: the steps whose input tensors were passed
    const size_t base = (size_t)b * N * 3;
    for (int i = t; i < 3 * N; i += AUG_THREADS) p[i] = points[base + i];
    __syncthreads();

    if (f & AUG_NORMALIZE) {
        float c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s = 0.f;
            for (int n = t; n < N; n += AUG_THREADS) s += p[3 * n + a];
            c[a] = block_reduce(s, false, red, t) / (float)N;
        }
        float m = 0.f;
        for (int n = t; n < N; n += AUG_THREADS) {
            const float x = p[3 * n] - c[0], y = p[3 * n + 1] - c[1], z = p[3 * n + 2] - c[2];
            p[3 * n] = x; p[3 * n + 1] = y; p[3 * n + 2] = z;
            m = fmaxf(m, sqrtf(x * x + y * y + z * z));
        }
        m = block_reduce(m, true, red, t);
        if (m > 0.f) {
            for (int n = t; n < N; n += AUG_THREADS) {
                p[3 * n] /= m; p[3 * n + 1] /= m; p[3 * n + 2] /= m;
            }
        }
    }
    __syncthreads();
    if (f & AUG_AFFINE) {
        float W[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) W[i] = affine[(size_t)b * 12 + i];
        for (int n = t; n < N; n += AUG_THREADS) {
            const float x = p[3 * n], y = p[3 * n + 1], z = p[3 * n + 2];
            p[3 * n] = x * W[0] + y * W[3] + z * W[6] + W[9];
            p[3 * n + 1] = x * W[1] + y * W[4] + z * W[7] + W[10];
            p[3 * n + 2] = x * W[2] + y * W[5] + z * W[8] + W[11];
        }
    }
    __syncthreads();                  // the phases index the cloud differently (per point / per float): fence between them
    if (f & AUG_JITTER) {
        for (int i = t; i < 3 * N; i += AUG_THREADS) p[i] += fminf(fmaxf(jitter[base + i], -clip), clip);
    }
    __syncthreads();
    if (f & AUG_REPLACE) {
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float mn = INFINITY, mx = -INFINITY;
            for (int n = t; n < N; n += AUG_THREADS) { mn = fminf(mn, p[3 * n + a]); mx = fmaxf(mx, p[3 * n + a]); }
            lo[a] = -block_reduce(-mn, true, red, t);
            hi[a] = block_reduce(mx, true, red, t);
        }
        for (int n = t; n < N; n += AUG_THREADS) {
            if (replace_mask[(size_t)b * N + n]) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    p[3 * n + a] = lo[a] + (hi[a] - lo[a]) * replace_u[base + 3 * n + a] + replace_g[base + 3 * n + a];
            }
        }
    }
    if (f & AUG_DROPOUT) {
        __syncthreads();
        const float x0 = p[0], y0 = p[1], z0 = p[2];
        __syncthreads();
        for (int n = t; n < N; n += AUG_THREADS) {
            if (dropout_mask[(size_t)b * N + n]) { p[3 * n] = x0; p[3 * n + 1] = y0; p[3 * n + 2] = z0; }
        }
    }
    __syncthreads();
    for (int i = t; i < 3 * N; i += AUG_THREADS) out[base + i] = p[i];
}

extern "C" int p4p_augment_f32(const float* points, float* out, const int32_t* flags, const float* affine,
                               const float* jitter, float clip, const uint8_t* replace_mask, const float* replace_u,
                               const float* replace_g, const uint8_t* dropout_mask, int B, int N, void* stream) {
    P4P_CHECK(points && out && flags, "p4p_augment_f32: null pointer");
    P4P_CHECK((!replace_mask) == (!replace_u) && (!replace_mask) == (!replace_g),
              "p4p_augment_f32: replace_mask, replace_u and replace_g go together");
    const int enabled = AUG_NORMALIZE | (affine ? AUG_AFFINE : 0) | (jitter ? AUG_JITTER : 0) | (replace_mask ? AUG_REPLACE : 0) |
                        (dropout_mask ? AUG_DROPOUT : 0);
    P4P_CHECK(B >= 0 && N >= 1 && N <= 12288, "p4p_augment_f32: bad sizes B=%d N=%d (N <= 12288: the cloud lives in LDS)", B, N);
    P4P_CHECK(clip >= 0.f, "p4p_augment_f32: clip must be >= 0");
    if (B == 0) return 0;
    const size_t lds = (size_t)3 * N * sizeof(float);
    P4P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(augment_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(AUG_THREADS), lds, (hipStream_t)stream, points, out, flags, affine, jitter,
                       clip, replace_mask, replace_u, replace_g, dropout_mask, N, enabled);
    P4P_LAUNCH_CHECK();
    return 0;
}
