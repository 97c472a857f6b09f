/* C ABI of libp4p_hip.so: the MI355X (gfx950) implementation of the Pix4Point hot path.
 *
 * The reference (Irish-77/adapting-2D-ViTs-for-3D-point-cloud-understanding) has no FFI: its hot
 * path is stock torch ops called from Python.  Each entry point below names the reference lines
 * whose arithmetic it replaces; INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.  Conventions:
 *   - every pointer is a DEVICE pointer unless said otherwise; tensors are dense row-major;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, never synchronise,
 *     never allocate: workspaces are caller-provided;
 *   - return 0 on success; non-zero = invalid argument (1) or HIP error (2), message from
 *     p4p_last_error() (thread local).  Nothing is launched when an argument check fails.
 *   - bf16 buffers are passed as void*.
 */
#ifndef P4P_HIP_H
#define P4P_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int p4p_abi_version(void);
const char* p4p_last_error(void);

/* ---- tokenizer index ops --------------------------------------------------------------------- */

/* farthest_point_sampling, src/models/pix4point.py:8-53.  points [B,N,3] f32, start_idx [B] i64 (the
 * draw of :30 made explicit), out_idx [B,S] i64, S <= N <= 12288.  Bit-exact: d=(dx*dx+dy*dy)+dz*dz
 * unfused fp32, lowest index wins ties. */
int p4p_fps_f32(const float* points, const int64_t* start_idx, int64_t* out_idx, int B, int N, int S, void* stream);

/* the `knn` closure of group_knn, src/models/pix4point.py:79-89.  centroids [B,S,3] f32,
 * out_idx [B,S,k] i32 sorted by exact squared distance, ties to the lowest index. 1 <= k <= 64. */
int p4p_knn_f32(const float* points, const float* centroids, int32_t* out_idx, int B, int N, int S, int k,
                void* stream);

/* the two gathers of group_knn, src/models/pix4point.py:94-100: out[b,s,j,:] = src[b, idx[b,s,j], :] */
int p4p_group_gather_f32(const float* src, const int32_t* idx, float* out, int B, int N, int S, int k, int C,
                         void* stream);

/* ---- bf16 MFMA GEMMs (every Linear / 1x1 conv of the path) ---------------------------------------- */

enum p4p_epilogue {
    P4P_EPI_BF16 = 0,          /* out bf16 = A W^T (+bias) */
    P4P_EPI_BF16_GELU = 1,     /* pre bf16 = A W^T + bias (if pre); out bf16 = gelu(pre)   (timm Mlp fc1+act) */
    P4P_EPI_F32_RESID = 2,     /* out f32 = A W^T (+bias) (+resid f32)                      (proj / fc2 + residual) */
    P4P_EPI_BF16_DGELU = 3,    /* out bf16 = (A W^T) * gelu'(pre)                            (backward of fc1's act) */
    P4P_EPI_BF16_ROWGROUP = 4, /* out bf16 = A W^T + rowgrp[m/group]; column sum/sumsq partials (P3Embed conv2a) */
    P4P_EPI_BF16_STATS = 5     /* out bf16 = A W^T (+bias); column sum/sumsq partials          (P3Embed conv2b) */
};

/* C[M,N] = A[M,K] W[N,K]^T with a fused epilogue.  K % 64 == 0, N % 4 == 0.  stats (epilogues 4,5):
 * [p4p_gemm_tn_stats_rows(M)][2][N] f32 partial column sums and sums of squares. */
int p4p_gemm_tn_bf16(int epilogue, const void* A, const void* W, void* out, const float* bias, const float* resid,
                     void* pre, const float* rowgrp, float* stats, int M, int N, int K, int group, void* stream);
int p4p_gemm_tn_stats_rows(int M);

/* dW[N,K] (f32) = beta*dW + G[M,N]^T A[M,K]  (autograd of nn.Linear.weight).  N,K % 8 == 0.
 * workspace: p4p_wgrad_splits(M,N,K) * N * K floats (deterministic split reduction, no atomics). */
int p4p_gemm_wgrad_bf16(const void* G, const void* A, float* dW, float* workspace, int M, int N, int K, float beta,
                        void* stream);
int p4p_wgrad_splits(int M, int N, int K);

/* ---- LayerNorm (timm Block.norm1/norm2, VisionTransformer.norm; eps 1e-6) ------------------------- */

/* y = LN(x + pos) * gamma + beta.  x,pos [M,D] f32 (pos may be NULL); xsum (may be NULL) receives x+pos
 * (the `feats + pos_embed` of src/models/pix4point.py:255); y bf16 or f32; mean/rstd [M] (may be NULL). */
int p4p_layernorm_fwd(const float* x, const float* pos, const float* gamma, const float* beta, float eps,
                      float* xsum, void* y, int y_is_f32, float* mean, float* rstd, int M, int D, void* stream);
/* dx = (accumulate_dx ? dx : 0) + LN'(dy); dgamma/dbeta = beta_acc*old + sums (either may be NULL).
 * partial: [p4p_layernorm_bwd_partial_rows()][2][D] f32 workspace. */
int p4p_layernorm_bwd(const void* dy, int dy_is_f32, const float* xin, const float* mean, const float* rstd,
                      const float* gamma, float* dx, int accumulate_dx, float* partial, float* dgamma, float* dbeta,
                      float beta_acc, int M, int D, void* stream);
int p4p_layernorm_bwd_partial_rows(void);
int p4p_colreduce_f32(const float* part, float* out, int nrows, long long stride, int ncols, float beta, void* stream);

/* ---- multi-head self-attention, head_dim 64 (timm Attention == src/models/apf_utils.py:142-158) ---- */

/* qkv [B*T,3D] bf16 (q|k|v, heads inside), out [B*T,D] bf16, lse2 [B,H,T] f32 (may be NULL for inference). */
int p4p_attention_fwd(const void* qkv, void* out, float* lse2, int B, int T, int H, int D, float scale, void* stream);
/* dqkv [B*T,3D] bf16 <- gradients; delta [B,H,T] f32 workspace. */
int p4p_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, float* delta, void* dqkv,
                      int B, int T, int H, int D, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
