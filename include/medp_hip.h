/* libmedp_hip — C ABI of the MI355X (gfx950) hot path of lastdancewithyou/multimodal_edema_prediction.
 *
 * The reference is 100 % Python and has NO plugin / FFI interface of its own: every
 * kernel it runs is dispatched by PyTorch/ATen or by the third-party x_transformers package.  The drop-in
 * boundary is therefore the Python nn.Module protocol between training_duett/{engine,trainer,evaluator}.py
 * and models/main_architecture_duett.py + loss/losses_duett.py; this library sits BEHIND the build's
 * mirror of those classes.  Each entry point cites the reference arithmetic it replaces
 * (paths relative to /root/reference; "model" = models/main_architecture_duett.py).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = invalid argument, >0 = hipError_t; the message is
 *     available from medp_last_error() (thread-local).  No exception crosses the ABI.
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless named host_*; `stream` is a
 *     hipStream_t passed as void*.  Functions only ENQUEUE: no allocation, no synchronisation, no hidden
 *     state — callers own every buffer, including workspaces (query *_workspace_bytes first).
 *   - "bf16" buffers hold raw bfloat16 bits (uint16); matrices are row-major with an explicit leading
 *     dimension in ELEMENTS.
 */
#ifndef MEDP_HIP_H
#define MEDP_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------------ */
const char* medp_last_error(void);
int medp_version(void);            /* 100*major + minor */
const char* medp_arch(void);       /* "gfx950" */

/* ---- GEMM: C[M,N] = epi(A[M,K] · W[N,K]^T), bf16 MFMA, fp32 accumulate ---------------------------
 * Replaces every nn.Linear / Conv2d-as-GEMM on the path: Dinov2 query/key/value/dense/fc1/fc2 and patch
 * projection (transformers modeling_dinov2.py:139,199-201,246,286,291), x_transformers to_q/k/v/out + ff
 * (duett/duett.py:95-105), img_proj / ts_proj / perceiver in_proj, out_proj, ff (model :566,:749-757,:1027).
 * epi: (+bias[n]) -> (act==1: GELU erf) -> (*scale[n], Dinov2 LayerScale :278) -> (+residual[m,n] fp32).
 * (GELU: erf to 1.5e-7 for an fp32 result; the large-tile kernels use a degree-17 odd polynomial, |GELU error| <= 6.3e-5,
 * where the result is rounded to bf16 anyway.)
 * Requirements: K, lda, ldw multiples of 8; N, ldc, ldr multiples of 4; 16-B aligned bases.  M, N, K ragged OK. */
int medp_gemm_bf16_nt(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                      const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                      void* stream);

/* The same GEMM with a caller-owned workspace: small grids (DuETT's skinny GEMMs, duett/duett.py:95-105 at M = B(V+1) / B(T+1) rows:
 * 25-100 tiles for 256 CUs) are split along K into slices that fill the chip; the slices' fp32 partial sums go to the workspace
 * and a second pass sums them in a fixed order and applies the epilogue (deterministic, no atomics).
 * medp_gemm_nt_workspace_bytes returns 0 where the one-pass kernel is used anyway. */
size_t medp_gemm_nt_workspace_bytes(int M, int N, int K);
int medp_gemm_bf16_nt_ws(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc, const float* bias,
                         const float* scale, const float* residual, int ldr, int act, int out_bf16, float* workspace,
                         size_t workspace_bytes, void* stream);

/* FP32 KERNEL MODE (SURVEY.md 7 "always keep an fp32 kernel mode for tight checks"; 8(d) fp32 tolerances): the same two GEMMs with
 * fp32 operands and fp32 FMA accumulation on the vector ALUs (no matrix cores): a parity instrument, selected explicitly by the
 * host layer (functional.set_precision("fp32")), never by tensor dtype.  No alignment requirements. */
int medp_gemm_f32_nt(const float* A, const float* W, float* C, int M, int N, int K, int lda, int ldw, int ldc, const float* bias,
                     const float* scale, const float* residual, int ldr, int act, void* stream);
int medp_gemm_f32_tn(const float* dY, const float* X, float* C, int M, int N, int K, int lddy, int ldx, void* stream);

/* Weight gradient C[N,K] = sum_m dY[m,n] X[m,k] (fp32 out, bf16 row-major operands, transposing LDS reads, split over m with
 * a deterministic slab reduction): the dW of every trainable Linear (autograd of model :566,:749-757,:1027, duett.py:95-105). */
size_t medp_gemm_tn_workspace_bytes(int M, int N, int K);
int medp_gemm_bf16_tn(const void* dY, const void* X, float* C, int M, int N, int K, int lddy, int ldx, float* workspace, void* stream);

/* How many CUs the persistent 256 x 256 GEMM (multi-round grids: the CXR encoder's qkv / fc1) may hold for a whole launch:
 * launches issued or CAPTURED from now on start at most `cap` workgroups (a multiple of 8 in 8..256 is used; 0 = the default,
 * 256 or MEDP_V7_WGS) and, below the cap, the fewest that still finish in the same number of rounds.  A step whose OTHER branch
 * is the long one (graph_step.GraphedStudentStep: the student's DuETT forward/backward beside the frozen teacher) leaves it
 * more of the chip this way.  Returns the previous cap. */
int medp_gemm_persistent_cap(int cap);

/* Live timing of the step's dominant kernel (the CXR-encoder block GEMMs launched by medp_vit_forward), bench.py's roofline
 * leg.  mode 1: HIP events bracket every such launch on its own stream (eager launches).  mode 2: every such launch issued
 * or CAPTURED while the mode is on carries an in-kernel launch clock (first workgroup in / last workgroup out stamp the
 * 100-MHz wall clock into a private device slot), so the begin / end of the LAST execution of each launch can be read after
 * replaying a hipGraph - the in-step figure.  mode 0 stops arming and keeps what was gathered.  collect() (device
 * synchronised by the caller in mode 2) returns the summed kernel time, the launch count and the algorithmic FLOPs
 * (2*M*N*K per launch). */
int medp_gemm_profile_enable(int mode);
int medp_gemm_profile_collect(double* host_total_ms, long long* host_n_launches, double* host_total_flops);

/* ---- attention ---------------------------------------------------------------------------------- */
/* Dense softmax(QK^T*scale)V, head dim 64, bf16 in/out: Dinov2 eager_attention_forward (modeling_dinov2.py:153-178).
 * q/k/v: [B*S, ...] rows with strides ld*, head h at column h*64.  o: [B*S, H*64]. */
int medp_attn_fwd_dh64(const void* q, const void* k, const void* v, void* o, int B, int S, int H, int ldq, int ldk,
                       int ldv, int ldo, float scale, void* stream);
/* Dense self-attention for small head dims (dh <= 16, dh % 4 == 0, N <= 272) on the matrix cores: DuETT's event / time axis
 * encoders (x_transformers Encoder, duett/duett.py:95-105: 2 heads of dim 12 over V+1 / T+1 tokens, no mask), inference form.
 * qkv: fp32 rows [B*N, ld] holding q | k | v column blocks of H*dh each (the fused QKV GEMM's output); o: bf16 [B*N, ldo].
 * Returns -2 (nothing launched) for shapes it is not built for: the caller then uses medp_attn_small_fwd. */
int medp_attn_dh16_fwd(const float* qkv, int ld, void* o_bf16, int ldo, int B, int N, int H, int dh, float scale, void* stream);
/* The same attention in TRAINING form (the student-KD step: dropout on the probabilities, gradients to q, k, v; autograd of
 * duett/duett.py:95-105): forward writes o [B*N][ldo] and the rows' log2-sum-exp lse [B*H*N]; backward writes dQ | dK | dV into
 * the column blocks of dqkv [B*N][lddqkv] (delta_ws: B*H*N floats of scratch).  Three MFMA kernels, a wave per 16-row tile, no LDS,
 * nothing added into memory (bitwise reproducible); same dropout stream as medp_attn_small_*.  Both return -2 (nothing launched)
 * for unsupported shapes. */
/* io_bf16 = 0: qkv, dout fp32 in, o, dqkv fp32 out (rounded to bf16 MFMA operands inside).  io_bf16 = 1: all four are bf16 — the
 * 16-bit hand-over between the projection GEMMs and the attention of a training step; the same operand bits, so the same results. */
int medp_attn_dh16_train_supported(int B, int N, int H, int dh, int ld, int ldo);   /* 1: the shapes these kernels take (16-byte aligned buffers assumed) */
int medp_attn_dh16_train_fwd(const void* qkv, int ld, void* o, int ldo, float* lse, int io_bf16, int B, int N, int H, int dh, float scale,
                             float dropout_p, unsigned seed, unsigned stream_id, void* stream);
int medp_attn_dh16_train_bwd(const void* dout, int lddo, const void* qkv, int ld, const float* lse, float* delta_ws, void* dqkv,
                             int lddqkv, int io_bf16, int B, int N, int H, int dh, float scale, float dropout_p, unsigned seed,
                             unsigned stream_id, void* stream);
/* Training form (--unfreeze_cxr, run.py:184-187): the same forward that also writes lse[B,H,S], the log2-domain logsumexp of the
 * scaled scores, and the flash backward that consumes it.  prep: dsum[b,h,s] = <dout, o> and the bf16 copy of dout.
 * bwd: dq/dk/dv fp32 with row stride ldd (e.g. the three column blocks of one [B*S, 3*H*64] buffer); q/k/v/dout_bf16 bf16. */
int medp_attn_fwd_dh64_lse(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H, int ldq,
                           int ldk, int ldv, int ldo, float scale, void* stream);
int medp_attn_bwd_dh64_prep(const float* dout, int lddout, const void* o_bf16, int ldo, void* dout_bf16, int lddob,
                            float* dsum, int B, int S, int H, void* stream);
int medp_attn_bwd_dh64(const void* q, const void* k, const void* v, int ldqkv, const void* dout_bf16, int lddo,
                       const float* lse, const float* dsum, float* dq, float* dk, float* dv, int ldd, int B, int S, int H,
                       float scale, void* stream);
/* Small fp32 attention (head dim <= 64, Lk <= 1536), fwd/bwd with optional dropout on the probabilities and
 * optional head-averaged weights (pre-zeroed [B,Lq,Lk]): x_transformers Attention inside the DuETT encoders
 * (model :81,:91) and nn.MultiheadAttention inside _PerceiverBlock (model :759-762, need_weights/average). */
int medp_attn_small_fwd(const float* q, int ldq, long long q_batch_stride, const float* k, const float* v, int ldkv,
                        long long kv_batch_stride, void* o, int ldo, int o_bf16, float* attn_avg, int B, int Lq, int Lk,
                        int H, int dh, float scale, float dropout_p, unsigned seed, unsigned stream_id, void* stream);
int medp_attn_small_bwd(const float* dout, int lddo, const float* q, int ldq, long long q_batch_stride, const float* k,
                        const float* v, int ldkv, long long kv_batch_stride, float* dq, int lddq, float* dk, int lddkv,
                        float* dv, int reserved, long long dkv_batch_stride, int B, int Lq, int Lk, int H, int dh, float scale,
                        float dropout_p, unsigned seed, unsigned stream_id, void* stream);

/* ---- normalisation -------------------------------------------------------------------------------- */
/* nn.LayerNorm (modeling_dinov2.py:348,353 eps 1e-6; model :750-753 eps 1e-5).  y is bf16 (feeds a GEMM) or fp32. */
int medp_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int y_bf16, float* mean,
                       float* rstd, int rows, int D, float eps, void* stream);
size_t medp_colsum_workspace_bytes(int rows, int D);
int medp_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* w, const float* mean,
                       const float* rstd, float* dx, int lddx, int accumulate_dx, float* dw, float* db, float* workspace,
                       int rows, int D, void* stream);
/* out[c] = sum_r x[r][c] (bias gradients; deterministic two-stage reduction) */
int medp_colsum_f32(const float* x, int ldx, float* out, float* workspace, int rows, int D, void* stream);
/* x_transformers ScaleNorm: y = x / max(||x||, eps) * sqrt(D) * g  (duett/duett.py:95-105 use_scalenorm=True) */
int medp_scalenorm_fwd(const float* x, int ldx, const float* g, void* y, int ldy, int y_bf16, float* rnorm, int rows, int D,
                       float eps, void* stream);
int medp_scalenorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* rnorm, float* dx,
                       int lddx, int accumulate_dx, float* dg, float* workspace_rows, int rows, int D, void* stream);
/* dx = add + d ScaleNorm(x) / dx applied to dy: the residual join of a pre-norm block (x feeds the norm AND the residual) in the backward,
 * out of place — `add` (the gradient that arrived through the residual) is only read.  dg as above. */
int medp_scalenorm_bwd_add(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* rnorm, const float* add,
                           int ldadd, float* dx, int lddx, float* dg, float* workspace_rows, int rows, int D, void* stream);

/* ---- layout / pointwise ------------------------------------------------------------------------------ */
int medp_cast_f32_bf16(const float* x, int ldx, void* y, int ldy, int rows, int cols, void* stream);
int medp_transpose_to_bf16(const void* x, int x_is_bf16, int ldx, void* y, int ldy, int rows, int cols, void* stream);
/* The bf16 GEMM operands of MANY fp32 weight matrices in one launch (the per-step casts of a training step's trainable weights: each
 * nn.Linear of the reference needs W as bf16 [N,K] for y = x W^T and W^T as bf16 [K, Npad8] for dX = dY W).  Job: src fp32 [rows, ld_src];
 * dst_plain bf16 [rows, ld_plain] and / or dst_t bf16 [cols, ld_t] (either may be NULL; pad columns of dst_t are the caller's zeros).
 * Block b handles the 64x64 tile dev_block_tile[b] (row-major over ceil(rows/64) x ceil(cols/64)) of job dev_block_job[b].
 * Same rounding as medp_cast_f32_bf16 / medp_transpose_to_bf16. */
typedef struct {
    const void* src;
    void* dst_plain;
    void* dst_t;
    int rows, cols, ld_src, ld_plain, ld_t, reserved_;
} MedpOperandJob;
int medp_weight_operands_multi(const MedpOperandJob* dev_jobs, const int* dev_block_job, const int* dev_block_tile, int n_blocks,
                               void* stream);
int medp_gelu_bwd(const float* dy, const float* pre, float* dx, long long n, void* stream);
/* bf16 forms (trainable CXR encoder, fused blocks): out = gelu(pre); dx = dy * gelu'(pre); all three bf16, n % 8 == 0 */
int medp_gelu_bf16_fwd(const void* pre, void* out, long long n, void* stream);
int medp_gelu_bf16_bwd(const void* dy, const void* pre, void* dx, long long n, void* stream);
/* Dinov2PatchEmbeddings conv14/s14 as im2col (modeling_dinov2.py:139,148) */
int medp_im2col_patch(const float* pix, void* A, int B, int C, int H, int W, int patch, int kpad, void* stream);
/* cls token + position embeddings (modeling_dinov2.py:108-112) */
int medp_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int P, int D, void* stream);
/* interpolate_pos_encoding, bicubic align_corners=False (modeling_dinov2.py:57-95) */
int medp_pos_embed_bicubic(const float* pos, float* out, int src_side, int gh, int gw, int D, void* stream);
/* its transpose (the trainable encoder): dout [1 + gh*gw, D] -> dpos [1 + src_side^2, D]; a gather, deterministic */
int medp_pos_embed_bicubic_bwd(const float* dout, float* dpos, int src_side, int gh, int gw, int D, void* stream);

/* ---- whole-module forward of the frozen CXR encoder: CXREncoder.forward (model :152-158) ---------- */
typedef struct {
    const float *ln1_w, *ln1_b;
    const void* qkv_w;            /* bf16 [3*hidden, hidden]: query | key | value rows */
    const float* qkv_b;           /* [3*hidden] */
    const void* proj_w;           /* bf16 [hidden, hidden]  attention.output.dense */
    const float *proj_b, *ls1;    /* layer_scale1.lambda1 */
    const float *ln2_w, *ln2_b;
    const void* fc1_w;            /* bf16 [mlp, hidden] */
    const float* fc1_b;
    const void* fc2_w;            /* bf16 [hidden, mlp] */
    const float *fc2_b, *ls2;
    /* LayerNorm folded into the qkv / fc1 GEMMs (all six non-NULL, or the LayerNorm launches stay): LN(x) W^T + b =
     * rstd (x (W g)^T) - rstd mean colsum(W g) + (b + W beta).  *_wg = bf16(W * ln_w) [rows as *_w], *_cs = fp32 row sums of the bf16
     * *_wg [3*hidden / mlp], *_b2 = b + W ln_b (fp32).  Used for batches whose block GEMMs take the 256-tile kernels. */
    const void *qkv_wg, *fc1_wg;
    const float *qkv_cs, *qkv_b2, *fc1_cs, *fc1_b2;
} MedpVitLayer;

typedef struct {
    int hidden, n_layers, n_heads, mlp_hidden, patch, pos_side, patch_kpad;
    float ln_eps;
    const void* patch_w;          /* bf16 [hidden, patch_kpad] (conv weight flattened c,i,j; zero padded) */
    const float *patch_b, *cls, *pos;   /* pos: fp32 [pos_side^2 + 1, hidden] */
    const float *final_ln_w, *final_ln_b;
    const MedpVitLayer* layers;   /* HOST array of n_layers entries */
} MedpVitWeights;

size_t medp_vit_workspace_bytes(const MedpVitWeights* host_w, int B, int H, int W);
/* pixels fp32 [B,3,H,W] -> tokens_f32 [B, P+1, hidden] (may be NULL) and/or tokens_bf16 (may be NULL) after the final LN */
int medp_vit_forward(const MedpVitWeights* host_w, const float* pixels, int B, int H, int W, float* tokens_f32,
                     void* tokens_bf16, void* workspace, size_t workspace_bytes, void* stream);
/* The same forward in pieces: the embedding stage runs when first_layer == 0, encoder layers [first_layer, last_layer), the final
 * LayerNorm (and the outputs) when last_layer == n_layers.  Between calls the fp32 token stream lives in `workspace`, which
 * the caller must leave untouched.  (graph_step.py splits the frozen encoder of the NEXT batch across the two captured graphs
 * of a multi-GPU step, so that the gradient all-reduce and the optimiser run beside encoder layers instead of after them.) */
int medp_vit_forward_part(const MedpVitWeights* w, const float* pixels, int B, int H, int W, float* tokens_f32, void* tokens_bf16,
                          void* workspace, size_t workspace_bytes, int first_layer, int last_layer, void* stream);

/* ---- whole-module forward of the DuETT backbone in inference form: DuettFeatureExtractor.encode (model :31-94) -----
 * BatchNorm layers are folded by the caller into (scale, shift) = (w/sqrt(var+eps), b - mean*scale): eval-mode
 * BatchNormLastDim (duett/duett.py:11-22).  Encoder = x_transformers.Encoder(depth=1) (duett/duett.py:95-105),
 * restated in oracle/xt_encoder.py (parity unpinned at that boundary). */
typedef struct {
    const float* g_attn;          /* ScaleNorm gain before attention [1] */
    const void* qkv_w;            /* bf16 [3*E, D]: to_q | to_k | to_v rows (no bias) */
    const void* out_w;            /* bf16 [D, E]   to_out (no bias) */
    const float* g_ff;            /* ScaleNorm gain before the feed-forward [1] */
    const void* ff1_w;            /* bf16 [d_ff, D] */
    const float* ff1_b;
    const void* ff2_w;            /* bf16 [D, d_ff] */
    const float* ff2_b;
    const float* g_final;         /* final ScaleNorm gain [1] (used when final_norm != 0) */
} MedpEncoderWeights;

typedef struct {
    int n_vars, n_static, d_embedding, n_heads, n_layers, d_ff, d_hidden_embed, d_hidden_tab, d_hidden_time, n_obs_rows,
        final_norm;
    float norm_eps;
    /* per-variable embedding MLPs stacked over V: Linear(2,64) -> ReLU -> BN -> Linear(64,E)  (duett.py:84-86) */
    const void *emb_w0, *emb_b0, *emb_bn_scale, *emb_bn_shift, *emb_w4, *emb_b4;   /* fp32 [V,64,2] [V,64] [V,64] [V,64] [V,E,64] [V,E] */
    /* the same per-variable MLPs in the layout the fused embed kernel reads with SCALAR loads (weights are uniform over a
     * workgroup): emb_l0 fp32 [V,64,8] = (w0[j][0], w0[j][1], b0[j], bn_scale[j], bn_shift[j], 0, 0, 0); emb_w4t fp32 [V,64,E] = w4 transposed */
    const void *emb_l0, *emb_w4t;
    const void* n_obs_table;      /* fp32 [n_obs_rows] (n_obs_embedding.weight[:,0]) */
    const void *tab_w0, *tab_b0, *tab_bn_scale, *tab_bn_shift, *tab_w4, *tab_b4;   /* tab_encoder (duett.py:124-125) */
    const void* special;          /* fp32 [8, E] special_embeddings */
    const void *time_w0, *time_b0, *time_bn_scale, *time_bn_shift, *time_w3t, *time_b3;   /* full_time_embedding = cve (duett.py:151-157); time_w3t = weight of the last Linear TRANSPOSED, fp32 [h, tt] */
    const void* rep_embedding;    /* fp32 [tt_dim]  full_rep_embedding.weight[:,0] */
    const void* event_embedding;  /* fp32 [V+1, et_dim] full_event_embedding.weight */
    const MedpEncoderWeights* event_enc;   /* HOST arrays of n_layers entries */
    const MedpEncoderWeights* time_enc;
} MedpDuettWeights;

/* The embedding stage of medp_duett_encode on its own (model :41-69; kernel-level parity and the per-kernel HBM table of bench.py).
 * stages bit 0: static encoder + FUSED psi build: psi (model :45-66) is produced directly in the EVENT view with the event
 *   embedding added (model :80) and the event encoder's first ScaleNorm applied: xe_out fp32 [B, V+1, (T+1)E], h_out bf16 (same
 *   shape), psi0_out (optional) psi in the time view [B, T+1, V+1, E] before the add; tab_workspace: unused (may be NULL).
 * stages bit 1: time embedding cve(xs_times) with the REP row appended (model :67-69): temb_out fp32 [B, T+1, (V+1)E]. */
int medp_duett_embed_fwd(const MedpDuettWeights* host_w, const float* xs_static, const float* xs_ts, const float* xs_times, int B, int T,
                         float* xe_out, void* h_out_bf16, float* temb_out, float* psi0_out, float* tab_workspace, int stages, void* stream);
/* Axis swap + positional add + the next encoder's first ScaleNorm in one pass (model :80 / :90 + x_transformers pre-norm):
 * x_out[b][a2][a1][:] = in[b][a1][a2][:] * rowscale(b,a1) + add, h_out = ScaleNorm(x_out) as bf16.  rowscale applies the previous
 * encoder's pending final ScaleNorm (rnorm [B*A1] from medp_scalenorm_fwd, gain g_prev) or is 1 when rnorm is NULL.
 * add: [A2, A1, E] (add_batch_stride 0) or [B, A2, A1, E] (stride A2*A1*E). */
int medp_duett_swap_add_norm(const float* in, const float* rnorm, const float* g_prev, const float* add, long long add_batch_stride,
                             const float* g_norm, float norm_eps, float* x_out, void* h_out_bf16, int B, int A1, int A2, int E,
                             void* stream);

/* ---- device-side batch assembly -------------------------------------------------------------------------
 * medp_feats_to_input replaces the host loop of Model.feats_to_input (duett/duett.py:159-187): sample b is a ragged series
 * [T_b, 2V] (values | observation counts) with bin times [T_b], given EITHER by device pointer tables (ts_ptrs / time_ptrs,
 * [B] device arrays of device pointers) OR as rows of one stacked buffer (ts_base + b*ts_stride, time_base + b*time_stride,
 * strides in elements); lengths [B] int32 device array of T_b (NULL: every sample has T_uniform steps).
 * Output: xs_ts [B, Tpad, 2V+1] (last max_len steps, zero mask column appended, zero padded; Tpad = the longest kept length,
 * computed by the caller from the shapes), xs_times [B, Tpad], xs_static [B, Ds].
 * Training augmentation (duett.py:169-175,184-185): values += aug_noise * N(0,1) * count; timesteps dropped with probability
 * aug_mask (row := 0, mask column := 1); static += aug_noise * N(0,1); draws from the counter-based hash (seed, stream_id,
 * element, RNG epoch) - with both rates 0 the result is bit-exact. */
int medp_feats_to_input(const float* const* ts_ptrs, const float* ts_base, long long ts_stride, const float* const* time_ptrs,
                        const float* time_base, long long time_stride, const int* lengths, int T_uniform, const float* static_in,
                        float* xs_ts, float* xs_times, float* xs_static, int B, int V, int Ds, int max_len, int Tpad, float aug_noise,
                        float aug_mask, unsigned seed, unsigned stream_id, void* stream);
/* Masking half of Model.pretrain_prep_batch (duett.py:189-237) for pretrain_masked_steps == 1: mask_t [B] int32 = masked
 * timestep, event_idx [B] int32 = masked variable (NULL: predict_events off), keep [B,V] uint8 = the variable-dropout draw
 * (NULL: pretrain_dropout 0) - the HOST draws them (numpy Generator, reference order).  Writes the clipped input
 * [B,T,2V+1], y_ts / y_masks [B,V], y_events / y_events_mask [B,T]. */
int medp_ssl_mask_batch(const float* xs_ts, const int* mask_t, const int* event_idx, const unsigned char* keep, float* clipped,
                        float* y_ts, float* y_masks, float* y_events, float* y_events_mask, int B, int T, int V, void* stream);

size_t medp_duett_workspace_bytes(const MedpDuettWeights* host_w, int B, int T);
/* xs_static [B,Ds], xs_ts [B,T,2V+1], xs_times [B,T] fp32 (outputs of feats_to_input) -> tokens [B, T+1, E*(V+1)];
 * psi0_out (optional, [B,T+1,V+1,E]) receives psi after the embedding stage for parity checks. */
int medp_duett_encode(const MedpDuettWeights* host_w, const float* xs_static, const float* xs_ts, const float* xs_times, int B,
                      int T, float* tokens_f32, void* tokens_bf16, float* psi0_out, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- fusion head pointwise ops (fp32), dropout masks regenerated from (seed, stream_id, element index) --------- */
int medp_gelu_dropout_fwd(const float* x, float* y, long long n, float p, unsigned seed, unsigned stream_id, void* stream);
int medp_gelu_dropout_bwd(const float* dy, const float* x, float* dx, long long n, float p, unsigned seed, unsigned stream_id,
                          void* stream);
/* 16-bit hand-over forms for a fused feed-forward node (ScaleNorm -> Linear -> GELU -> Dropout -> Linear of x_transformers' FeedForward):
 * the forward writes the next Linear's bf16 operand directly; the backward writes dx in fp32 (bias gradient) AND its bf16 copy (operand of
 * the previous Linear's gradient GEMMs).  Same rounding as medp_cast_f32_bf16 on the fp32 result. */
int medp_gelu_dropout_fwd_bf16(const float* x, void* y_bf16, long long n, float p, unsigned seed, unsigned stream_id, void* stream);
int medp_gelu_dropout_bwd_bf16(const float* dy, const float* x, float* dx, void* dx_bf16, long long n, float p, unsigned seed,
                               unsigned stream_id, void* stream);
/* out = residual + dropout(y) (residual may be NULL; backward = same call on the incoming gradient with residual NULL) */
int medp_dropout_add(const float* y, const float* residual, float* out, long long n, float p, unsigned seed, unsigned stream_id,
                     void* stream);
/* Linear(D,1) output layer of the pathology heads (model :574,:590) */
int medp_rowdot_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int rows, int D, void* stream);
int medp_rowdot_bwd(const float* dy, const float* x, int ldx, const float* w, float* dx, float* dw, float* db, int rows, int D,
                    void* stream);
/* per-label biases + residual fusion  fusion = img.detach() + beta*correction  (model :634-639) */
int medp_fusion_logits_fwd(const float* hi, const float* ht, const float* hc, const float* img_bias, const float* ts_bias,
                           const float* beta, float* img, float* ts, float* scaled, float* fus, int B, int K, void* stream);
int medp_fusion_logits_bwd(const float* d_img, const float* d_ts, const float* d_scaled, const float* d_fus, const float* hc,
                           const float* beta, float* d_hi, float* d_ht, float* d_hc, float* d_img_bias, float* d_ts_bias,
                           float* d_beta, int B, int K, void* stream);
/* StudentModel pool="mean" over the T hourly tokens of [B, T1, D] (model :1231) */
int medp_meanpool_fwd(const float* x, float* y, int B, int T, int T1, int D, void* stream);
int medp_meanpool_bwd(const float* dy, float* dx, int B, int T, int T1, int D, void* stream);

/* ---- losses: value + gradient in one launch ------------------------------------------------------------------------
 * DualPathologyLoss (loss/losses_duett.py:131-194).  out: [0] total, [1] img_total, [2] ts_total, [3] fus_total,
 * [4..4+3K) img_per | ts_per | fus_per.  g_*: d total / d logits (any may be NULL).  pos_weight may be NULL. */
int medp_dual_pathology_loss(const float* img, const float* ts, const float* fus, const float* y, const float* mask,
                             const float* label_weights, const float* pos_weight, float alpha_img, float alpha_ts,
                             float alpha_fus, float eps, float* out, float* g_img, float* g_ts, float* g_fus, int B, int K,
                             void* stream);
/* StudentKDLoss + VanillaKLKD (loss/losses_duett.py:8-57).  out: [0] total [1] bce [2] kd.  pos_weight = 1 for none. */
int medp_student_kd_loss(const float* z_s, const float* z_t, const float* y, float T, float alpha, float pos_weight, float* out,
                         float* g_zs, int B, void* stream);

/* engine extras: aux residual KL with label smoothing (training_duett/engine.py:149-165); LP regularisers
 * coef*mean(x^2) (engine.py:217-223); the linear probe's global masked BCE (cxr_linear_training.ipynb:426-437).
 * Each writes the scalar to out[0] and, when g != NULL, the gradient w.r.t. its differentiable input. */
int medp_aux_residual_kl(const float* img_logits, const float* scaled_correction, const float* y, const float* mask,
                         float label_smoothing, float* out, float* g_scaled, int n, void* stream);
int medp_sq_mean(const float* x, float coef, float* out, float* g, int n, void* stream);
int medp_masked_bce_global(const float* logits, const float* y, const float* mask, float* out, float* g, int n, void* stream);

/* DuETT-only step losses (duett/duett.py:337-365): mean(((a-b)*mask)^2) and mean(weight * BCEWithLogits); mask/weight may be NULL */
int medp_masked_mse(const float* a, const float* b, const float* mask, float* out, float* g_a, int n, void* stream);
int medp_bce_mean(const float* logits, const float* y, const float* weight, float* out, float* g, int n, void* stream);

/* ---- optimiser: torch.optim.AdamW semantics over a device table of tensors (trainer.py:77-116,383) -------------- */
typedef struct {
    void* param;            /* fp32, updated in place */
    const void* grad;       /* fp32 */
    void* exp_avg;          /* fp32 */
    void* exp_avg_sq;       /* fp32 */
    long long numel;
    float lr, weight_decay;
} MedpAdamTensor;
int medp_adamw_chunk_elems(void);   /* elements one workgroup updates; block b handles chunk dev_block_chunk[b] of tensor dev_block_tensor[b] */
/* step: host step count (>= 1), used when dev_step == NULL; dev_step: device counter holding the step (graph replay) */
int medp_adamw_multi(const MedpAdamTensor* dev_descs, const int* dev_block_tensor, const int* dev_block_chunk, int n_blocks,
                     float beta1, float beta2, float eps, int step, const unsigned* dev_step, float grad_scale, void* stream);

/* ---- HIP-graph support: state that must change between replays lives in device memory --------------------------------
 * medp_rng_set_epoch_ptr: device uint32 mixed into every dropout seed (NULL = off); medp_counter_advance: *c += 1 */
int medp_rng_set_epoch_ptr(const unsigned* dev_ptr);
int medp_counter_advance(unsigned* dev_counter, void* stream);

/* ---- DuETT embedding stage in TRAINING form (student KD path; BatchNorm batch statistics, gradients everywhere) ------
 * grouped tiny layers, group = variable: x [G,R,K], W [G,N,K], b [G,N]   (duett/duett.py:24-39,84-86,124-125,151-157) */
int medp_glinear_fwd(const float* x, const float* W, const float* b, float* y, int G, int R, int K, int N, void* stream);
size_t medp_glinear_bwd_workspace_bytes(int G, int R, int K, int N);
int medp_glinear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, float* workspace, int G, int R,
                     int K, int N, void* stream);
/* The per-variable embedding MLP Linear(KIN, C) -> ReLU -> BatchNormLastDim(C) -> Linear(C, E) (duett/duett.py:24-39 `simple_mlp` with
 * hidden_batch_norm, model file :45-55) of ALL variables as fused kernels that never store the hidden activations (csrc/duett_embed_train.hip):
 * x [G][R][KIN], W0 [G][C][KIN], b0 / bn_w / bn_b / running_* / save_* [G][C], W1 [G][E][C], b1 [G][E], out [G][R][E].  batch_stats = 1:
 * train() (batch statistics normalise, running statistics updated in place), 0: eval().  Built for (KIN, C, E) = (2, 64, 24) — the
 * reference's defaults; medp_gmlp_supported says so, the entry points return -2 otherwise (use medp_glinear_* / medp_gbn_* / medp_act_*).
 * bwd: dx may be NULL; every other gradient is written (dbn_w / dbn_b: BatchNorm weight / bias).  Bitwise reproducible. */
int medp_gmlp_supported(int KIN, int C, int E);
size_t medp_gmlp_workspace_bytes(int G, int R, int KIN, int C, int E);
int medp_gmlp_fwd(const float* x, const float* W0, const float* b0, const float* bn_w, const float* bn_b, float* running_mean,
                  float* running_var, const float* W1, const float* b1, float* out, float* save_mean, float* save_var, int G, int R,
                  int KIN, int C, int E, float eps, float momentum, int batch_stats, float* workspace, void* stream);
int medp_gmlp_bwd(const float* dout, const float* x, const float* W0, const float* b0, const float* bn_w, const float* bn_b,
                  const float* save_mean, const float* save_var, const float* W1, float* dx, float* dW0, float* db0, float* dbn_w,
                  float* dbn_b, float* dW1, float* db1, int G, int R, int KIN, int C, int E, float eps, int batch_stats,
                  float* workspace, void* stream);
/* BatchNormLastDim over the R rows of every group (duett/duett.py:11-22): batch_stats=1 train (biased var normalises,
 * unbiased updates running), 0 eval.  save_mean/save_var [G,C] feed the backward. */
size_t medp_gbn_workspace_bytes(int G, int R, int C);   /* per-chunk partial sums of the two-stage (deterministic) row reductions */
int medp_gbn_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var, float* y, float* save_mean,
                 float* save_var, int G, int R, int C, float eps, float momentum, int batch_stats, float* workspace, void* stream);
int medp_gbn_bwd(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_var, float* dx, float* dw,
                 float* db, int G, int R, int C, float eps, int batch_stats, float* workspace, void* stream);
int medp_act_fwd(const float* x, float* y, long long n, int mode /*0 relu, 1 tanh*/, void* stream);
int medp_act_bwd(const float* dy, const float* y, float* dx, long long n, int mode, void* stream);
/* (value, n_obs_embedding[clip(int(count),0,15)]) pairs per variable, zero-padded to KP columns (model :41-52) */
int medp_embed_inputs_fwd(const float* xs_ts, const float* n_obs_table, int table_rows, float* xin, int B, int T, int V, int KP,
                          void* stream);
int medp_embed_inputs_bwd_blocks(int B, int T, int V);
int medp_embed_inputs_bwd(const float* xs_ts, const float* d_xin, float* partial, int table_rows, int B, int T, int V, int KP,
                          void* stream);
/* psi assembly with special-token overrides (model :53-66) and its backward */
int medp_psi_assemble_fwd(const float* xs_ts, const float* var_out, const float* tab_out, const float* special, float* psi, int B, int T,
                          int V, int E, void* stream);
/* backward: d_var_out [V][B*T][E] (zero where a cell was overridden) and, per batch element and SLICE of its cells (S =
 * medp_psi_assemble_bwd_slices), the partial sums d_tab_partial [B][S][E] (static column) and d_special_partial [B][S][2][E]
 * (masked / REP cells): the caller adds the slices (a fixed order: deterministic) */
int medp_psi_assemble_bwd_slices(int B, int T, int V);
int medp_psi_assemble_bwd(const float* xs_ts, const float* dpsi, float* d_var_out, float* d_tab_partial, float* d_special_partial, int B,
                          int T, int V, int E, void* stream);
int medp_axis_swap(const float* in, float* out, int B, int A1, int A2, int E, void* stream);
/* the axis swap with the new leading axis' positional embedding added on the way: mode 1: add [A2][A1][E] shared by the batch (model :80-81);
 * mode 3: add [B*(A2-1)][A1][E] per sample for a2 < A2-1 and add_last [A1][E] for the last row (model :90 without the concatenation) */
int medp_axis_swap_add(const float* in, const float* add, const float* add_last, float* out, int B, int A1, int A2, int E, int mode,
                       void* stream);
int medp_add_bcast(const float* a, const float* b, float* out, long long per_batch, int B, int broadcast_b, void* stream);

/* ---- LocalTrajectoryEncoder (models/main_architecture_duett.py:1242-1391) -------------------------------
 * The Linear / LayerNorm / GELU stages of the module are the kernels above; these are the rest.
 * medp_traj_features: x [B,T,2V] fp32 (values | counts) -> out [B*V, T, 8] fp32 = {value (0 where unobserved), observed,
 *   log1p(count)/ln 16, steps since the previous observation / T, (T - t) / T, 0, 0, 0}  (:1316-1330, :1342-1358; the five
 *   features padded to 8 so that Linear(5, d) runs as a K = 8 GEMM).
 * medp_gru_fwd: one-layer batch-first nn.GRU, h0 = 0 (:1297-1303, :1366), hidden size d = 128 only (anything else: rc < 0).
 *   gi [S,T,3d] fp32 = x_t W_ih^T + b_ih (gate order r | z | n), whh_bf16 [3d,d], bhh [3d] -> hseq [S,T,d]; gates [S,T,3d]
 *   (r | z | n) and hn [S,T,d] (= W_hn h + b_hn) are saved for the backward (both null: inference).
 * medp_gru_bwd: dh [S,T,d] (gradient w.r.t. every h_t) -> dgi [S,T,3d], dghn [S,T,d] (gradient w.r.t. W_hn h + b_hn) and
 *   dgh_bf16 [S,T,3d] (r | z | n parts of the gradient w.r.t. h W_hh^T + b_hh; dW_hh = dgh^T h_prev is a medp_gemm_bf16_tn).
 *   whh_t_bf16 [d,3d] = W_hh transposed. */
int medp_traj_features(const float* x, float* out, int B, int T, int V, void* stream);
int medp_gru_fwd(const float* gi, const void* whh_bf16, const float* bhh, float* hseq, float* gates, float* hn, int S, int T, int d,
                 void* stream);
int medp_gru_bwd(const float* dh, const float* gates, const float* hn, const float* hseq, const void* whh_t_bf16, float* dgi,
                 float* dghn, void* dgh_bf16, int S, int T, int d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDP_HIP_H */
