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
 *   - threading: ONE host thread at a time per process and ONE stream per device.  The switches (p4p_set_*) and the per-device
 *     split-K scratch of p4p_gemm_tn_bf16 (a few MB, grown with hipMalloc on first use of a larger shape: the only allocation the
 *     library makes) are unsynchronised process state; forward and backward of one model never run concurrently in the trainer
 *     (autograd's backward thread takes over after the forward has returned), which is the case this library is written for.
 */
#ifndef P4P_HIP_H
#define P4P_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on every incompatible change of a signature below (2: p4p_clip_coef gained grad_scale, p4p_apf_group gained C;
 * 3: the version check itself; 4: p4p_layernorm_bwd gained dx_in; 5: p4p_set_reserved_cus / p4p_persistent_grid_size;
 * 6: p4p_attention_bwd's workspace is sized by p4p_attention_bwd_workspace and no library-owned scratch remains on that path;
 * additions since (no bump): p4p_wgrad_slab_count, p4p_gemm_wgrad_slabs_bf16, p4p_sum_slabs_multi).  libp4p_hip.so is built separately from the Python package: the binding compares p4p_abi_version() with
 * this constant when it loads the library and refuses a stale build. */
#define P4P_ABI_VERSION 6
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
    P4P_EPI_BF16_GELU = 1,     /* h = A W^T + bias; out bf16 = gelu(h); pre bf16 = gelu'(h) if pre (timm Mlp fc1+act) */
    P4P_EPI_F32_RESID = 2,     /* out f32 = A W^T (+bias) (+resid f32) (+ a second f32 residual passed as `rowgrp`: the next
                                  block's x + pos_embed, pix4point.py:255)                 (proj / fc2 + residual);
                                  group = S > 1 (no resid): row m is written to row m + m/S + 1 of a [B*(S+1), N]
                                  token stream, i.e. behind one cls row per sequence (pix4point.py:248-252) */
    P4P_EPI_BF16_DGELU = 3,    /* out bf16 = (A W^T) * pre   (pre = the gelu'(h) saved by epilogue 1)  (backward of fc1's act) */
    P4P_EPI_BF16_ROWGROUP = 4, /* out bf16 = A W^T + rowgrp[m/group]; column sum/sumsq partials (P3Embed conv2a) */
    P4P_EPI_BF16_STATS = 5     /* out bf16 = A W^T (+bias); column sum/sumsq partials          (P3Embed conv2b) */
};

/* C[M,N] = A[M,K] W[N,K]^T with a fused epilogue.  K % 64 == 0, N % 4 == 0.  stats (epilogues 4,5):
 * [p4p_gemm_tn_stats_rows(M)][2][N] f32 partial column sums and sums of squares. */
int p4p_gemm_tn_bf16(int epilogue, const void* A, const void* W, void* out, const float* bias, const float* resid,
                     void* pre, const float* rowgrp, float* stats, int M, int N, int K, int group, void* stream);
int p4p_gemm_tn_stats_rows(int M);
/* EPI_BF16_DGELU accepts stats = [p4p_gemm_tn_colsum_rows(M)][N] f32: partial column sums of the output written by the
   same epilogue (zero rows allowed); reduced with p4p_colreduce_f32 they are the bias gradient of the Linear that
   produced `pre` (autograd of timm Mlp.fc1.bias). */
int p4p_gemm_tn_colsum_rows(int M);

/* dW[N,K] (f32) = beta*dW + G[M,N]^T A[M,K]  (autograd of nn.Linear.weight).  N,K % 8 == 0.
 * dbias[N] (may be NULL) = beta*dbias + column sums of G (autograd of nn.Linear.bias), from the same pass.
 * workspace: p4p_wgrad_splits(M,N,K) * (N*K + N) floats (deterministic split reduction, no atomics). */
int p4p_gemm_wgrad_bf16(const void* G, const void* A, float* dW, float* dbias, float* workspace, int M, int N, int K,
                        float beta, void* stream);
int p4p_wgrad_splits(int M, int N, int K);
/* The same GEMM with its slab sum deferred (round 5): p4p_gemm_wgrad_slabs_bf16 leaves [p4p_wgrad_slab_count(M,N,K)][N][K] f32 slabs
 * in `workspace` (count 0: the shape is not taken, use p4p_gemm_wgrad_bf16), p4p_sum_slabs_multi sums the slabs of up to 8 matrices
 * in ONE launch: out_i[e] = beta_i*out_i[e] + slabs_i[0][e] + slabs_i[1][e] + .. (the order of p4p_gemm_wgrad_bf16: bit-identical).
 * descs: HOST array of n records {const float* slabs; float* out; long long n; int nslabs; float beta} (32 bytes each; n_i % 4 == 0).
 * The four weight gradients of a ViT block (autograd of src/models/pix4point.py:254-255) are summed by one call per block. */
int p4p_wgrad_slab_count(int M, int N, int K);
int p4p_gemm_wgrad_slabs_bf16(const void* G, const void* A, float* workspace, int M, int N, int K, void* stream);
int p4p_sum_slabs_multi(const void* descs, int n, void* stream);

/* ---- LayerNorm (timm Block.norm1/norm2, VisionTransformer.norm; eps 1e-6) ------------------------- */

/* y = LN(x + pos) * gamma + beta.  x,pos [M,D] f32 (pos may be NULL); xsum (may be NULL) receives x+pos
 * (the `feats + pos_embed` of src/models/pix4point.py:255); y bf16 or f32; mean/rstd [M] (may be NULL). */
int p4p_layernorm_fwd(const float* x, const float* pos, const float* gamma, const float* beta, float eps,
                      float* xsum, void* y, int y_is_f32, float* mean, float* rstd, int M, int D, void* stream);
/* dx = accumulate_dx * dx_in + LN'(dy) with accumulate_dx in {0, 1, 2} (2: the residual stream enters its block twice,
 * src/models/apf_utils.py:232,292); dx_in NULL = dx (in place); dgamma/dbeta = beta_acc*old + sums (either may be NULL);
 * dxsum (may be NULL) = beta_acc*old + column sums of the resulting dx: the bias gradient of the Linear that wrote
 * this residual stream (proj.bias / fc2.bias), free of an extra pass.
 * dx_bf16 (may be NULL) receives a bf16 copy of dx (next GEMM operand); dacc (may be NULL) accumulates dx
 * (gradient of the positional embedding that is re-added before every block, pix4point.py:255).
 * partial: [p4p_layernorm_bwd_partial_rows()][3][D] f32 workspace. */
int p4p_layernorm_bwd(const void* dy, int dy_is_f32, const float* xin, const float* mean, const float* rstd,
                      const float* gamma, float* dx, const float* dx_in, int accumulate_dx, void* dx_bf16, float* dacc,
                      int dacc_init, float* partial, float* dgamma, float* dbeta, float* dxsum, float beta_acc, int M, int D,
                      void* stream);
/* out[n] = slabs[0][n] + slabs[1][n] + .. in that order (n % 4 == 0): the gradient of the positional embedding that is re-added
 * before every block (pix4point.py:255) as the sum of the block-input gradients kept in nslabs slabs. */
int p4p_sum_slabs_f32(const float* slabs, float* out, long long n, int nslabs, void* stream);
int p4p_layernorm_bwd_partial_rows(void);
int p4p_colreduce_f32(const float* part, float* out, int nrows, long long stride, int ncols, float beta, void* stream);

/* ---- multi-head self-attention, head_dim 64 (timm Attention == src/models/apf_utils.py:142-158) ---- */

/* qkv [B*T,3D] bf16 (q|k|v, heads inside), out [B*T,D] bf16, lse2 [B,H,T] f32 (may be NULL for inference). */
int p4p_attention_fwd(const void* qkv, void* out, float* lse2, int B, int T, int H, int D, float scale, void* stream);
/* dqkv [B*T,3D] bf16 <- gradients; delta: f32 workspace of p4p_attention_bwd_workspace(B,T,H) floats (>= B*H*T; the one-kernel
   backward at T = 513, csrc/attention_fused2.h, parks its half-finished dQ^T chains there, 139 KB per workgroup; contents after the
   call are unspecified).  The workspace belongs to the call: two calls in flight on different streams need two workspaces.  Same arithmetic contract for
   every T (P and dS rounded to bf16 where they enter a product, fp32 accumulation, no atomics: bitwise reproducible); which
   kernels run depends on T only.  colsum_partial (may be NULL):
   [p4p_attention_colsum_rows(B,T)][3D] f32 per-wave column sums of dqkv, written by the same kernels; reduced with
   p4p_colreduce_f32 they are the gradient of the qkv bias (autograd of nn.Linear.bias in timm Attention.qkv). */
int p4p_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, float* delta, void* dqkv,
                      float* colsum_partial, int B, int T, int H, int D, float scale, void* stream);
int p4p_attention_colsum_rows(int B, int T);
long long p4p_attention_bwd_workspace(int B, int T, int H);

/* ---- streaming helpers ----------------------------------------------------------------------------- */

/* fp32 master weight [R,C] (row stride ld_src) -> bf16 copies dst [R,C] and dstT [C,R] (either may be NULL). */
int p4p_cast_transpose_f32_bf16(const float* src, long long ld_src, void* dst, void* dstT, int R, int C, void* stream);
/* the same for n weights in ONE launch (the per-step refresh of the operand copies after the optimizer step): descs is a DEVICE
   array of n records {const float* src; bf16* dst; bf16* dstT; long long ld; int R, C; int tile0, tiles_c} (48 B each; tile0 =
   64 x 64 tiles of the records before it, tiles_c = ceil(C / 64)); every record: R, C, ld % 4 == 0, src 16-B aligned */
int p4p_cast_transpose_multi(const void* descs, int n, int total_tiles, void* stream);
int p4p_cast_f32_bf16(const float* src, void* dst, long long n, void* stream);
/* out[N] = beta*out + column sums of G[M,N] bf16 (autograd of nn.Linear.bias). workspace: p4p_colsum_splits(M)*N f32 */
int p4p_colsum_bf16(const void* G, float* out, float* workspace, int M, int N, float beta, void* stream);
int p4p_colsum_splits(int M);
/* partial[r][N] f32, r < ceil(M/256): column sums of rows [256r, 256r+256) of G[M,N] bf16 (no final reduction). */
int p4p_colsum_partial_bf16(const void* G, float* partial, int M, int N, void* stream);
/* token stream [B,T=S+1,D] f32 <-> compact patch rows [B*S,D] (cls row first, pix4point.py:248-252).
 * mode 0: compact(bf16) = stream rows; mode 1: stream rows = compact(f32). */
int p4p_token_rows(float* stream_f32, void* compact, int B, int S, int D, int mode, void* stream);
/* mode 1: stream[b,0,:] = vec; mode 0: vec = sum_b stream[b,0,:] */
int p4p_cls_rows(float* stream_f32, float* vec, int B, int T, int D, int mode, void* stream);

/* ---- P3Embed local MLP (src/models/pix4point.py:137-156,178-188); see csrc/embed.hip for the algebra ---- */

int p4p_embed_partial_rows(void);
/* partial [p4p_embed_partial_rows()][27]: sums and upper-triangular second moments of x6 = [xyz, feats][idx] */
int p4p_embed_moments(const float* points, const float* feats, const int32_t* idx, float* partial, int B, int N, int S,
                      int k, void* stream);
/* folds convs[0][0], convs[0][1], BatchNorm2d into f1 = relu(scale * (aeff . x6) + shift); updates the running
 * statistics in training mode.  aeff [C1,8], mom [42]. */
int p4p_embed_fold(const float* W1, const float* W2, const float* b2, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, const float* mom_partial, float* aeff, float* scale,
                   float* shift, float* save_mean, float* save_invstd, float* mom, long long Np, float eps,
                   float momentum, int training, int C1, void* stream);
/* F1 [B*S*k,C1] bf16, pooled [B*S,C1] bf16 (= max_k, pix4point.py:163,185), arg [B*S,C1] u8 */
int p4p_embed_stage1(const float* points, const float* feats, const int32_t* idx, const float* aeff,
                     const float* scale, const float* shift, void* F1, void* pooled, uint8_t* arg, int B, int N, int S,
                     int k, int C1, void* stream);
/* BatchNorm scale/shift from column sums [2][C] (training) or running statistics (eval) */
int p4p_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float* scale, float* shift, float* save_mean, float* save_invstd, long long Np, float eps,
                    float momentum, int training, int C, void* stream);
int p4p_bn_relu_bf16(const void* x, void* y, const float* scale, const float* shift, long long rows, int C, void* stream);
/* tokens[g,c] = max_j relu(scale*H3[g*k+j,c]+shift) (pix4point.py:188) */
int p4p_bn_relu_maxpool(const void* H3, const float* scale, const float* shift, void* tok_bf16, float* tok_f32,
                        uint8_t* arg, long long G, int k, int C, void* stream);
/* backward of max-pool + ReLU + BatchNorm: dH3 [G*k,C] bf16, sums[2][C] = (sum dy, sum dy*xhat) = (dbeta, dgamma) */
int p4p_pool_bn_bwd(const void* dtok, int dtok_is_f32, const void* H3, const uint8_t* arg, const float* scale,
                    const float* shift, const float* mean, const float* invstd, float* partial, float* sums, void* dH3,
                    long long G, int k, int C, void* stream);
/* backward of ReLU + BatchNorm on a dense gradient, in place over dY; sums as above */
int p4p_relu_bn_bwd(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                    const float* invstd, float* partial, float* sums, long long rows, int C, void* stream);
/* the same, and group_out [rows / k, C] bf16 = p4p_group_sum_bf16 of the result (bit for bit) out of the same pass (round 5) */
int p4p_relu_bn_bwd_gsum(void* dY, const void* H, const float* scale, const float* shift, const float* mean,
                         const float* invstd, float* partial, float* sums, void* group_out, int k, long long rows, int C,
                         void* stream);
int p4p_group_sum_bf16(const void* X, void* out, long long G, int k, int C, void* stream);
/* backward of stage 1 (ReLU, BatchNorm, both leading convs) in closed form from sum(dy), sum(dy (x) x6) */
int p4p_embed_stage1_bwd(const float* points, const float* feats, const int32_t* idx, const void* F1, const void* dF1,
                         const void* dPooled, const uint8_t* arg, float* partial, float* sums, const float* mom,
                         const float* aeff, const float* W1, const float* W2, const float* gamma,
                         const float* save_invstd, float* dW1, float* dW2, float* db2, float* dgamma, float* dbeta,
                         float beta_acc, int B, int N, int S, int k, int C1, void* stream);

/* ---- measurement aid (bench.py roofline): HIP events on the launch stream around exactly the main kernel of
 * kind 1 (weight-gradient GEMM), 2 (gemm_tn), 3 (attention forward), 4 (attention backward: its launches in one bracket).
 * collect(): out[0] = total ms, out[1] = launches, out[2] = algorithmic FLOPs (2MNK; attention 4 B H T^2 64, x2.5 backward). */
int p4p_prof_enable(int kind);
int p4p_prof_collect(double* out);
/* ---- CU-masked streams (hipExtStreamCreateWithCUMask): kernels of such a stream only occupy the compute units whose bit is
 * set in mask[words]; disjoint masks partition the chip between a memory-bound and an MFMA-bound stream.  The reference has
 * no counterpart (single stream, src/train/pix4point_trainer.py:153-176). */
int p4p_stream_create_cu_mask(const uint32_t* mask, int words, void** out_stream);
int p4p_stream_destroy(void* stream);
int p4p_device_cu_count(void);
/* experiment switches (tools/gemm_bench.py): tile configuration overrides, 0 = automatic */
int p4p_set_gemm_variant(int v);
int p4p_set_wgrad_wide(int v);             /* weight-gradient GEMM where K % 256 == 0: 2 (default) gemm_wgrad3.h, 8 waves of 128 x 64, hand-interleaved; 3 gemm_wgrad4.h, the staggered 4-phase schedule (12 % slower); 6 / 10 gemm_wgrad5.h, 4 waves of 128 x 128 with AGPR accumulators, hand-interleaved / compiler-scheduled (bit-identical to 2; +1-3 % per kernel, nothing per step); 0 / 1: 256 x 128 tiles / 256 x 256 compiler-scheduled */
int p4p_set_gemm_persistent(int v);        /* 1 (default): short-K GEMMs with several tiles per CU run the persistent tile loop (gemm_tn4p.h); 0: one workgroup per tile; 3: persistent without the banded tile order of wide outputs */
int p4p_set_gemm_epilogue(int v);          /* bit 1: no start-time shift in the persistent GEMM; bits 8..: that shift in 1,024-cycle units (0 = the per-epilogue default); bit 0: unused */
int p4p_set_gemm_peel_rounds(int v);       /* wide outputs (> 4 tiles of 256 columns): peel the tail row panel(s) into a small-tile launch when the big tiles run at most v rounds (default 4; 32 was measured faster per kernel and slower per step: NOTES A.8) */
int p4p_set_attention_variant(int v);      /* bit 0: no resident forward / dQ, bit 1: no resident dK/dV, bit 2: no two-pass (64n+1) forward, bit 3: no fused backward (T = 32n+1 <= 257), bit 4: no fused backward at T = 513 (the streaming dQ + key-stationary dK/dV pair instead), bit 5: no start-time shift of the first round of the fused / key-stationary backward, bit 6: no key-stationary dK/dV at T = 513, bit 7: a workgroup per (b,h) pair instead of the one-workgroup-per-CU walk, bits 8-9: shift phases - 1 and bits 10..: shift in 1024-cycle units (sweeps only) */
int p4p_set_attention_fwd_variant(int v);  /* forward only, experiments: bit 0: the K/V-resident forward at T = 513 (attention_fwd2.h, another tile order: tests/emulate.py 'online32') instead of attn_fwd_q_kernel<2>, bit 1: a workgroup per (b,h) pair instead of the walk, bit 2: no start-time shift, bit 3: the online-softmax kernels follow the running maximum exactly (no lazy rescale: the contract of rounds 1-3), bits 4 / 5: four / two 16-query tiles per wave in the streaming kernel whatever the grid */
/* ---- persistent kernels next to a concurrent long-lived kernel (the RCCL all-reduce of a data-parallel backward; the reference has no
 * counterpart: src/train/pix4point_trainer.py:4-6 imports DistributedDataParallel and never uses it).  The persistent tile loop of the
 * short-K GEMMs and the walking attention backward launch one workgroup per CU with a static share of the work; r CUs held by another
 * kernel would make r of them start late with their whole share to do.  r = 0 (default): all CUs; r > 0: CUs - r workgroups (r rounded
 * up to a multiple of 8), GEMM / attention outputs bit-identical, the weight-gradient GEMM sizes its M-splits to one round of the same CUs - r workgroups (another fixed summation order); r < 0: a workgroup per tile / pair (no persistent kernels).  Returns the previous
 * value.  p4p_persistent_grid_size(): workgroups a persistent launch takes on the current device now (0: none are launched). */
int p4p_set_reserved_cus(int r);
int p4p_persistent_grid_size(void);
/* rehearsal aid (tools/contention_rehearsal.py): `workgroups` one-wave workgroups, each holding lds_bytes of LDS (> 80 KiB: one per
 * CU, excluding every 160-KiB workgroup of this library from that CU) for `microseconds` (bounded: 1 .. 5,000,000), on `stream`. */
int p4p_debug_occupy_cus(int workgroups, int lds_bytes, long long microseconds, void* stream);

/* ---- global feature of PointViT.forward_cls_feat (src/models/pix4point.py:260-271) ----
   fwd: out_max[B,D] = max over tokens 1..T-1 of x[B,T,D] (arg[B,D] = token index of the first maximum), out_cls[B,D] = x[:,0]
   (either output may be NULL); bwd: dx[B,T,D] written in one pass (dmax scattered to arg, dcls to token 0, zero elsewhere). */
int p4p_global_feat_fwd(const float* x, float* out_max, int* arg, float* out_cls, int B, int T, int D, void* stream);
int p4p_global_feat_bwd(float* dx, const float* dmax, const int* arg, const float* dcls, int B, int T, int D, void* stream);

/* ---- first layer of PointViT.pos_embed: h[M,C] = GELU(x[M,3] W[C,3]^T + b)  (src/models/pix4point.py:214-218,246) ----
   bwd: partial [p4p_pos_mlp0_bwd_rows(M)][C][4] f32; its column sums (p4p_colreduce_f32) are (dW[j][0..2], db[j]). */
int p4p_pos_mlp0_fwd(const float* x, const float* W, const float* b, void* h, int h_is_bf16, long long M, int C, void* stream);
int p4p_pos_mlp0_bwd_rows(long long M);
int p4p_pos_mlp0_bwd(const float* x, const float* W, const float* b, const float* dh, float* partial, long long M, int C,
                     void* stream);

/* ---- train-step tail: clip_grad_norm_ + AdamW over all parameter tensors (src/train/pix4point_trainer.py:164-167) ----
   tensors: device array of {float* p; const float* g; float* m; float* v; long long n; float lr, wd} (48 B each);
   chunks: device array of {int tensor; int offset} in units of p4p_opt_chunk_elems() elements, never straddling tensors. */
int p4p_opt_chunk_elems(void);
/* partial[c] = sum of squares of gradient chunk c */
int p4p_grad_sumsq(const void* tensors, const void* chunks, int nchunks, float* partial, void* stream);
/* norm = grad_scale * sqrt(sum partial); coef_and_norm[0] = grad_scale * min(1, max_norm / (norm + 1e-6)), [1] = norm
   (torch.nn.utils.clip_grad_norm_ on gradients scaled by grad_scale: the data-parallel 1/world folded into the factor the
   update kernel applies anyway; max_norm = +inf disables the clip; a NaN norm gives a NaN factor, as torch's clamp does) */
int p4p_clip_coef(const float* partial, int nchunks, float max_norm, float grad_scale, float* coef_and_norm, void* stream);
/* torch.optim.AdamW update (amsgrad=False) with g scaled by *coef (NULL = 1); bc1 = 1 - beta1^t, bc2 = 1 - beta2^t */
int p4p_adamw_step(const void* tensors, const void* chunks, int nchunks, const float* coef, float beta1, float beta2,
                   float eps, float bc1, float bc2, void* stream);

/* ---- AdaptPointFormer tokenizer glue (FPS and kNN are p4p_fps_f32 / p4p_knn_f32 above: the APF
   copies src/data/sampler.py:4-30,47-75 are the same algorithms; kNN neighbours come out sorted, the reference's
   topk(sorted=False) order is unspecified and nothing downstream depends on it) ---- */
/* MortonEncoder.points_to_morton (src/models/apf_utils.py:67-104): order[b][r] = index of the centre with the r-th smallest
   30-bit Morton code (equal codes: lowest index first); codes (may be NULL) [B,G] int32.  centers [B,G,3] f32, G <= 4096. */
int p4p_morton_order_f32(const float* centers, int32_t* order, int32_t* codes, int B, int G, int resolution, void* stream);
/* Group.forward after the index ops (src/models/apf.py:70-110): x [B,N,C] f32 with C = 3 (xyz) or 4 (xyz + height: the
   reference's shipped configuration, train_apf.py:5,29), fps_idx [B,G] i64, knn_idx [B,G,gs] i32, order [B,G] i32.
   Outputs, each may be NULL: nb_f32 [B,G,gs,2C] = [x[nbr]-x[centre], x[centre]] (apf.py:83-95), x0_bf16 [B*G*gs,64] the
   hi/lo-split bf16 operand of the first 1x1 conv ([hi (2C) | lo (2C) | hi (2C) | 0], csrc/apf.hip), center [B,G,3]
   (the xyz of the group centres); all in Morton order. */
int p4p_apf_group(const float* x, const int64_t* fps_idx, const int32_t* knn_idx, const int32_t* order, float* nb_f32,
                  void* x0_bf16, float* center, int B, int N, int C, int G, int gs, void* stream);
/* torch.max over the points of a group (src/models/apf.py:160,167): H [G*k,C] bf16 -> tok [G,C] (+ bias[c] if given) as f32
   and/or bf16 (either may be NULL), arg [G,C] u8 (may be NULL) = first maximum.  bwd: dH [G*k,C] bf16 = scatter of
   dtok [G,C] (f32 or bf16) to the arg-max rows; accumulate != 0 adds into dH instead of overwriting it. */
int p4p_group_max_bf16(const void* H, const float* bias, float* tok_f32, void* tok_bf16, uint8_t* arg, long long G, int k,
                       int C, void* stream);
int p4p_group_max_bwd(const void* dtok, int dtok_is_f32, const uint8_t* arg, void* dH, int accumulate, long long G, int k,
                      int C, void* stream);
/* DropPath of APFViTLayer (src/models/apf_utils.py:258,280,290; timm DropPath: x * Bernoulli(keep)/keep per sample) without an
   elementwise pass: the branch's last GEMM runs with weights scaled by 1/keep and adds the residual for every sample, then
   p4p_rows_restore_f32 puts dst[b] = src[b] back for the dropped samples (keep[b] == 0; keep is [B] u8); in the backward
   p4p_rows_zero_bf16 clears the branch gradient rows of the same samples.  dst/src [B, per_sample] f32, x [B, per_sample] bf16. */
int p4p_rows_restore_f32(float* dst, const float* src, const uint8_t* keep, int B, long long per_sample, void* stream);
int p4p_rows_zero_bf16(void* x, const uint8_t* keep, int B, long long per_sample, void* stream);
/* AdapterLayer bottleneck between its two GEMMs (src/models/apf_utils.py:222-226): down bf16 = dropout_p(relu(pre)) over n
   elements; u (NULL when p == 0) holds uniform [0,1) draws, an element is kept where u >= p and scaled by 1/(1-p);
   gate f32 = relu'(pre) * dropout factor.  bwd: dpre bf16 = ddown * gate. */
int p4p_adapter_mid_fwd(const float* pre, const float* u, float p, void* down_bf16, float* gate, long long n, void* stream);
int p4p_adapter_mid_bwd(const float* ddown, const float* gate, void* dpre_bf16, long long n, void* stream);

/* ---- batched augmentation: the per-cloud numpy chain of src/data/scanobjectnn.py:186-205 /
   src/data/augment.py:4-191 for a whole batch resident in HBM, one workgroup per cloud.  Every random draw is an input.
   flags [B] i32 per cloud: 1 normalize_point_cloud (:4-18), 2 affine p = p W + t (rotations :76-108,158-191,261-307, scalings
   :35-47,193-226, shift :49-60 composed by the caller; affine [B][12] = W row-major then t), 4 jitter p += clip(jitter, +-clip)
   (:62-74; jitter [B,N,3] = sigma * randn), 8 drop_and_replace_with_noise (:110-156; replace_mask [B,N] u8, replace_u uniform
   [0,1) and replace_g = noise_std * randn, both [B,N,3]; the box is the cloud's CURRENT bounding box), 16 random_point_dropout
   (:20-33; dropout_mask [B,N] u8: those points become point 0).  Steps run in that order; a step whose tensors are NULL is
   skipped whatever the flag says.  points, out [B,N,3] f32 (may alias), N <= 12288. */
int p4p_augment_f32(const float* points, float* out, const int32_t* flags, const float* affine, const float* jitter, float clip,
                    const uint8_t* replace_mask, const float* replace_u, const float* replace_g, const uint8_t* dropout_mask,
                    int B, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif
