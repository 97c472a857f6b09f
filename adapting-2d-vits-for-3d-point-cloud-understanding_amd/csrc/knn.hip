// k-nearest-neighbour grouping for gfx950.  Replaces the `knn` closure and the two gathers of
// group_knn, reference src/models/pix4point.py:56-102 (cdist materialises a (B,S,N) fp32 matrix,
// topk re-reads it).  Here the S x N distances never leave registers.
//
// Workgroup = 4 waves on one cloud; the cloud is staged once into LDS as SoA x[N] y[N] z[N]
// (coalesced HBM read, conflict-free LDS reads).  Each wave owns one centroid at a time:
//   * every lane computes the exact squared distance to 16 points of the current 1024-point chunk
//     and keeps them as u64 keys (float_bits(d) << 32) | index  -- d >= 0, so u64 order == (d, index)
//     lexicographic order, which IS the tie rule (lowest index first); while it does so it tracks its two
//     smallest keys (min1, min2);
//   * the k best keys so far live SORTED in lanes 0..k-1; tau = the k-th of them (+inf until there are k);
//   * selection = repeated extract-min: wave-min of the lanes' min1 (two 32-bit DPP reductions: distance bits,
//     then index among the ties), stop the chunk as soon as it is >= tau; the winning key is inserted into the
//     sorted list (a ballot count gives the position, one wave_shr DPP shifts the tail) and its lane advances to
//     min2.  A lane that loses both tracked keys rescans its 16 (it retires keys in ascending order, so the
//     survivors are simply the keys above the last one it gave away) -- rare: ~2 lanes per centroid.
//     Round 1 re-scanned all 16 keys of every lane and re-derived tau with a second wave reduction in EVERY
//     round (~150 instructions per round, 2,700 per centroid at N = 1024, k = 16); this form needs ~50 per round
//     and no ranking pass at the end: the list is already in output order.
//
// Order contract (oracle/pix4point_oracle.py: knn_indices): ascending ((dx*dx+dy*dy)+dz*dz) in
// unfused fp32, ties to the lowest point index, int32 output (pix4point.py:89).
#include "common.h"

#define KNN_THREADS 256
#define KNN_WAVES 4
#define KNN_CHUNK_P 16                    // keys per lane per chunk -> 1024 points per chunk

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    u32 o;
    o = dpp_u32<0xB1>(v);  v = o < v ? o : v;
    o = dpp_u32<0x4E>(v);  v = o < v ? o : v;
    o = dpp_u32<0x141>(v); v = o < v ? o : v;
    o = dpp_u32<0x140>(v); v = o < v ? o : v;
    const u32 a = (u32)__builtin_amdgcn_readlane((int)v, 0), b = (u32)__builtin_amdgcn_readlane((int)v, 16);
    const u32 c = (u32)__builtin_amdgcn_readlane((int)v, 32), d = (u32)__builtin_amdgcn_readlane((int)v, 48);
    const u32 ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// minimum of the lanes' keys in (distance bits, index) order: the index only among the lanes that hold the minimum distance
__device__ __forceinline__ u64 wave_min_key(u64 v) {
    const u32 hi = (u32)(v >> 32);
    const u32 mh = wave_min_u32(hi);
    const u32 ml = wave_min_u32(hi == mh ? (u32)v : 0xFFFFFFFFu);
    return ((u64)mh << 32) | ml;
}
// lane l receives lane l-1's value (lane 0 keeps its own): DPP wave_shr:1
__device__ __forceinline__ u64 wave_shr1_u64(u64 v) {
    const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)v, (int)(u32)v, 0x138, 0xF, 0xF, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(v >> 32), (int)(u32)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}

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
    const u64 INF = ~0ull;          // no key
    const u64 UNK = ~0ull - 1;      // "second smallest not tracked any more": a real key is always below (index < 2^32 - 2)

    for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
        const float* c = centroids + ((size_t)b * S + s) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        u64 best = INF;          // lanes [0,k): the k best keys so far, ascending; INF = empty slot (and every lane >= k)
        u64 tau = INF;           // the k-th best once k keys are held
        int cnt = 0;             // keys held (wave-uniform)
        for (int base = 0; base < N; base += 64 * KNN_CHUNK_P) {
            u64 key[KNN_CHUNK_P];
            u64 min1 = INF, min2 = INF;
#pragma unroll
            for (int j = 0; j < KNN_CHUNK_P; ++j) {
                int i = base + j * 64 + lane;
                int ic = i < N ? i : N - 1;
                float d = sqdist3(sx[ic], sy[ic], sz[ic], cx, cy, cz);
                u64 kk = ((u64)__builtin_bit_cast(u32, d) << 32) | (u32)i;
                kk = i < N ? kk : INF;
                key[j] = kk;
                const u64 hi2 = kk < min1 ? min1 : kk;          // the larger of (kk, min1)
                min1 = kk < min1 ? kk : min1;
                min2 = hi2 < min2 ? hi2 : min2;
            }
            u64 floor_key = 0;       // this lane has given away every key below floor_key
            // at most k rounds in the first chunk, far fewer later; bounded for safety
            for (int round = 0; round < 64 * KNN_CHUNK_P; ++round) {
                const u64 m = wave_min_key(min1);
                if (m >= tau) break;                           // wave-uniform (also covers m == INF)
                // ---- insert m into the sorted list: position = number of held keys below it
                const int pos = __builtin_popcountll(__ballot(best < m));
                const u64 shifted = wave_shr1_u64(best);
                best = lane < pos ? best : (lane == pos ? m : shifted);
                best = lane < k ? best : INF;
                cnt = cnt < k ? cnt + 1 : k;
                tau = cnt == k ? readlane_u64(best, k - 1) : INF;
                // ---- the owning lane (keys are unique) moves on to its next key
                const bool owner = min1 == m;
                bool rescan = false;
                if (owner) {
                    floor_key = m + 1;
                    if (min2 == UNK) rescan = true;
                    else { min1 = min2; min2 = min2 == INF ? INF : UNK; }
                }
                if (__any(rescan)) {                            // rare: the lane lost both tracked keys
                    u64 n1 = INF, n2 = INF;
#pragma unroll
                    for (int j = 0; j < KNN_CHUNK_P; ++j) {
                        const u64 kk = key[j] >= floor_key ? key[j] : INF;
                        const u64 hi2 = kk < n1 ? n1 : kk;
                        n1 = kk < n1 ? kk : n1;
                        n2 = hi2 < n2 ? hi2 : n2;
                    }
                    if (rescan) { min1 = n1; min2 = n2; }
                }
            }
        }
        if (lane < k) out_idx[((size_t)b * S + s) * k + lane] = (int32_t)(u32)best;
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
