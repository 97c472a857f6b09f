// k-nearest-neighbour grouping for gfx950.  Replaces the `knn` closure and the two gathers of
// group_knn, reference src/models/pix4point.py:56-102 (cdist materialises a (B,S,N) fp32 matrix,
// topk re-reads it).  Here the S x N distances never leave registers.
//
// Workgroup = 4 waves on one cloud; the cloud is staged once into LDS as SoA x[N] y[N] z[N]
// (coalesced HBM read, conflict-free LDS reads).  Each wave owns one centroid at a time:
//   * every lane computes the exact squared distance to 16 points of the current 1024-point chunk
//     and keeps them as u64 keys (float_bits(d) << 32) | index  -- d >= 0, so u64 order == (d, index)
//     lexicographic order, which IS the tie rule (lowest index first);
//   * the k best keys seen so far live one per lane (lanes 0..k-1); `tau` is the largest of them;
//   * selection: wave-min of the chunk (DPP), stop the chunk as soon as min >= tau, otherwise the
//     owning lane retires that key (ballot picks the owner) and it replaces the lane holding tau.
//     The first chunk costs k+1 rounds; later chunks stop after the few keys that beat tau.
//   * the k survivors are ranked by counting smaller keys (k readlanes) and written sorted.
//
// Order contract (oracle/pix4point_oracle.py: knn_indices): ascending ((dx*dx+dy*dy)+dz*dz) in
// unfused fp32, ties to the lowest point index, int32 output (pix4point.py:89).
#include "common.h"

#define KNN_THREADS 256
#define KNN_WAVES 4
#define KNN_CHUNK_P 16                    // keys per lane per chunk -> 1024 points per chunk

__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ points,
                                                           const float* __restrict__ centroids,
                                                           int32_t* __restrict__ out_idx, int N, int S, int k,
                                                           int cpb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sx = reinterpret_cast<float*>(smem);
    float* sy = sx + N;
    float* sz = sy + N;

    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* src = points + (size_t)b * N * 3;
    for (int i = t; i < 3 * N; i += KNN_THREADS) {
        float v = src[i];
        int p = i / 3, c = i - 3 * p;
        (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
    }
    __syncthreads();

    const int s_begin = blockIdx.x * cpb;
    const int s_end = min(S, s_begin + cpb);
    const u64 INF = ~0ull;

    for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
        const float* c = centroids + ((size_t)b * S + s) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        u64 best = INF;          // lanes [0,k): one of the k best keys so far
        u64 tau = INF;           // max over best[0..k)
        for (int base = 0; base < N; base += 64 * KNN_CHUNK_P) {
            u64 key[KNN_CHUNK_P];
            u64 lmin = INF;
#pragma unroll
            for (int j = 0; j < KNN_CHUNK_P; ++j) {
                int i = base + j * 64 + lane;
                int ic = i < N ? i : N - 1;
                float d = sqdist3(sx[ic], sy[ic], sz[ic], cx, cy, cz);
                u64 kk = ((u64)__builtin_bit_cast(u32, d) << 32) | (u32)i;
                key[j] = i < N ? kk : INF;
                lmin = key[j] < lmin ? key[j] : lmin;
            }
            // at most k+1 rounds in the first chunk, far fewer later; bounded for safety
            for (int round = 0; round < 64 * KNN_CHUNK_P; ++round) {
                u64 m = wave_min_u64(lmin);
                if (m >= tau) break;                           // wave-uniform (also covers m == INF)
                if (lmin == m) {                               // exactly one lane: keys are unique
                    lmin = INF;
#pragma unroll
                    for (int j = 0; j < KNN_CHUNK_P; ++j) {
                        key[j] = key[j] == m ? INF : key[j];
                        lmin = key[j] < lmin ? key[j] : lmin;
                    }
                }
                // replace the first lane (< k) whose best equals tau
                u64 holders = __ballot(lane < k && best == tau);
                int victim = __builtin_ctzll(holders);
                if (lane == victim) best = m;
                tau = wave_max_u64(lane < k ? best : 0ull);
            }
        }
        // rank the k survivors; keys are distinct so ranks are a permutation of 0..k-1
        int rank = 0;
        for (int q = 0; q < k; ++q) {
            u64 o = readlane_u64(best, q);
            rank += o < best ? 1 : 0;
        }
        if (lane < k) out_idx[((size_t)b * S + s) * k + rank] = (int32_t)(u32)best;
    }
}

// Gather rows by index: out[b,s,j,:] = src[b, idx[b,s,j], :]   (pix4point.py:94-100)
__global__ void group_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                    float* __restrict__ out, int N, long long total, int SK, int C) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long long row = e / C;
    int c = (int)(e - row * C);
    long long b = row / SK;
    out[e] = src[(b * N + idx[row]) * C + c];
}

extern "C" int p4p_knn_f32(const float* points, const float* centroids, int32_t* out_idx, int B, int N, int S,
                           int k, void* stream) {
    P4P_CHECK(points && centroids && out_idx, "p4p_knn_f32: null pointer");
    P4P_CHECK(B >= 0 && N >= 1 && S >= 0, "p4p_knn_f32: bad sizes B=%d N=%d S=%d", B, N, S);
    P4P_CHECK(k >= 1 && k <= 64, "p4p_knn_f32: k=%d outside [1,64]", k);
    P4P_CHECK(k <= N, "p4p_knn_f32: k=%d exceeds N=%d (topk would raise, pix4point.py:88)", k, N);
    P4P_CHECK(N <= 12288, "p4p_knn_f32: N=%d exceeds the LDS-resident limit 12288", N);
    if (B == 0 || S == 0) return 0;
    size_t lds = (size_t)3 * N * sizeof(float);
    P4P_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // centroids per block: enough blocks to fill 256 CUs several times, few enough to amortise staging
    int cpb = 32;
    while (cpb > 4 && (long long)B * ceil_div(S, cpb) < 2048) cpb >>= 1;
    dim3 grid(ceil_div(S, cpb), B);
    hipLaunchKernelGGL(knn_kernel, grid, dim3(KNN_THREADS), lds, (hipStream_t)stream, points, centroids,
                       out_idx, N, S, k, cpb);
    P4P_LAUNCH_CHECK();
    return 0;
}

extern "C" int p4p_group_gather_f32(const float* src, const int32_t* idx, float* out, int B, int N, int S, int k,
                                    int C, void* stream) {
    P4P_CHECK(src && idx && out, "p4p_group_gather_f32: null pointer");
    P4P_CHECK(B >= 0 && N >= 1 && S >= 0 && k >= 1 && C >= 1, "p4p_group_gather_f32: bad sizes");
    long long total = (long long)B * S * k * C;
    if (total == 0) return 0;
    hipLaunchKernelGGL(group_gather_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       idx, out, N, total, S * k, C);
    P4P_LAUNCH_CHECK();
    return 0;
}
