// Train-step tail (src/train/pix4point_trainer.py:164-167): clip_grad_norm_(max_norm) followed by AdamW.step(), as three
// launches over ALL parameter tensors instead of torch's ~20 multi-tensor launches, with no host synchronisation:
//   1) p4p_grad_sumsq   : per-chunk sums of squares of every gradient (fixed chunking: reproducible)
//   2) p4p_clip_coef    : total norm = sqrt(sum), coef = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
//   3) p4p_adamw_step   : g' = coef * g;  p *= 1 - lr*wd;  m += (g' - m)(1 - b1);  v = b2 v + (1 - b2) g'^2;
//                         p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)          (torch.optim.AdamW, amsgrad=False)
// The gradients themselves are left unscaled in memory (torch's clip rewrites them; nothing on the path reads them
// afterwards).  Tensors are described by a device table; chunks of 16384 elements never straddle tensors.
// HBM-bound: 16 B read + 12 B written per parameter in (3), 4 B read in (1).
#include "common.h"

struct P4POptTensor {          // 48 bytes, built by the host (optim.py) as 6 x int64
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    float lr, wd;
};
struct P4POptChunk {           // 8 bytes
    int tensor;
    int offset;                // in units of OPT_CHUNK elements
};
#define OPT_CHUNK 16384

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const P4POptTensor* __restrict__ tens, const P4POptChunk* __restrict__ chunks,
                                                         float* __restrict__ partial) {
    const P4POptChunk c = chunks[blockIdx.x];
    const P4POptTensor T = tens[c.tensor];
    const long long base = (long long)c.offset * OPT_CHUNK;
    long long n = T.n - base;
    n = n < OPT_CHUNK ? n : OPT_CHUNK;
    float s = 0.f;
    const float* g = T.g + base;
    if ((reinterpret_cast<size_t>(g) & 15) == 0) {
        for (long long e = (long long)threadIdx.x * 4; e + 3 < n; e += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + e);
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        for (long long e = (n & ~3ll) + threadIdx.x; e < n; e += 256) s += g[e] * g[e];
    } else {
        for (long long e = threadIdx.x; e < n; e += 256) s += g[e] * g[e];
    }
    __shared__ float red[4];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = gradient factor = grad_scale * clip coefficient, out[1] = total norm of the SCALED gradients.  grad_scale is the
// data-parallel 1/world (the all-reduce sums; the mean is never written back to the 345-MB gradient buffer).  A NaN norm
// gives a NaN factor, as torch.nn.utils.clip_grad_norm_ does (clamp(nan) = nan): the blow-up stays visible.
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm, float grad_scale,
                                                        float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    __shared__ float red[4];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3])) * grad_scale;
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = ((coef < 1.f || coef != coef) ? coef : 1.f) * grad_scale;
        out[1] = norm;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(const P4POptTensor* __restrict__ tens, const P4POptChunk* __restrict__ chunks,
                                                    const float* __restrict__ coef_ptr, float beta1, float beta2, float eps,
                                                    float bc1, float rsqrt_bc2) {
    const P4POptChunk c = chunks[blockIdx.x];
    const P4POptTensor T = tens[c.tensor];
    const long long base = (long long)c.offset * OPT_CHUNK;
    long long n = T.n - base;
    n = n < OPT_CHUNK ? n : OPT_CHUNK;
    const float coef = coef_ptr ? coef_ptr[0] : 1.f;
    const float decay = 1.f - T.lr * T.wd, step_size = T.lr / bc1, omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    float* p = T.p + base;
    const float* g = T.g + base;
    float* m = T.m + base;
    float* v = T.v + base;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        pp *= decay;
        mm += (gg - mm) * omb1;
        vv = vv * beta2 + omb2 * gg * gg;
        const float denom = sqrtf(vv) * rsqrt_bc2 + eps;
        pp -= step_size * (mm / denom);
    };
    const bool aligned = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) |
                           reinterpret_cast<size_t>(v)) & 15) == 0;
    if (aligned) {
        for (long long e = (long long)threadIdx.x * 4; e + 3 < n; e += 1024) {
            f32x4 pp = *reinterpret_cast<const f32x4*>(p + e), gg = *reinterpret_cast<const f32x4*>(g + e);
            f32x4 mm = *reinterpret_cast<const f32x4*>(m + e), vv = *reinterpret_cast<const f32x4*>(v + e);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = pp[r], b = mm[r], c2 = vv[r];
                upd(a, gg[r], b, c2);
                pp[r] = a; mm[r] = b; vv[r] = c2;
            }
            *reinterpret_cast<f32x4*>(p + e) = pp;
            *reinterpret_cast<f32x4*>(m + e) = mm;
            *reinterpret_cast<f32x4*>(v + e) = vv;
        }
        for (long long e = (n & ~3ll) + threadIdx.x; e < n; e += 256) upd(p[e], g[e], m[e], v[e]);
    } else {
        for (long long e = threadIdx.x; e < n; e += 256) upd(p[e], g[e], m[e], v[e]);
    }
}

extern "C" int p4p_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int p4p_grad_sumsq(const void* tensors, const void* chunks, int nchunks, float* partial, void* stream) {
    P4P_CHECK(tensors && chunks && partial && nchunks >= 1, "p4p_grad_sumsq: bad arguments");
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const P4POptTensor*)tensors,
                       (const P4POptChunk*)chunks, partial);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_clip_coef(const float* partial, int nchunks, float max_norm, float grad_scale, float* coef_and_norm,
                             void* stream) {
    P4P_CHECK(partial && coef_and_norm && nchunks >= 1 && max_norm > 0.f && grad_scale > 0.f, "p4p_clip_coef: bad arguments");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nchunks, max_norm, grad_scale,
                       coef_and_norm);
    P4P_LAUNCH_CHECK();
    return 0;
}

// coef (may be NULL = 1): device pointer to the clip coefficient.  bc1 = 1 - beta1^t, bc2 = 1 - beta2^t for step count t.
extern "C" int p4p_adamw_step(const void* tensors, const void* chunks, int nchunks, const float* coef, float beta1,
                              float beta2, float eps, float bc1, float bc2, void* stream) {
    P4P_CHECK(tensors && chunks && nchunks >= 1, "p4p_adamw_step: bad arguments");
    P4P_CHECK(bc1 > 0.f && bc2 > 0.f, "p4p_adamw_step: bias corrections must be positive");
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const P4POptTensor*)tensors,
                       (const P4POptChunk*)chunks, coef, beta1, beta2, eps, bc1, 1.0f / sqrtf(bc2));
    P4P_LAUNCH_CHECK();
    return 0;
}
