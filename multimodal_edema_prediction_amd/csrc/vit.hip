// Whole-module forward of the frozen CXR encoder (ViT-B/14, Dinov2 layout) as ONE C call: every kernel of the
// 12 blocks is enqueued from C++ on the caller's stream (no Python between launches, graph-capturable).
//   tokens = LN_f( blocks( [cls; conv14(pixels)] + pos ) )        reference: model :152-158 -> modeling_dinov2.py
// Data layout in HBM: residual stream x fp32 [B*S, hidden]; every GEMM input is bf16 written by the producing
// kernel (LayerNorm / GELU epilogue / attention); LayerScale and the residual add are fused in GEMM epilogues.
//
// LayerNorm fold (MEDP_VIT_LNFOLD=1; OFF by default — measured, see below; for batches whose block GEMMs take the 256 x 256 tile
// kernels): the 24 LayerNorm launches of the block loop (a 76-MB pass each: 0.37 ms of a 4.4-ms encoder at B = 64) are folded into the
// GEMMs on both sides of them.  proj / fc2 — which write the fp32 token stream x anyway — also write bf16(x) and per row and 256-column tile the
// (sum, sum of squares) of x; qkv / fc1 multiply bf16(x) by W g and their epilogue applies
//     LN(x) W^T + b = rstd (x (W g)^T) - rstd mean colsum(W g) + (b + W beta)
// with mean / rstd from the three partial sums of the row (gemm_variants.h).  x stays fp32; what changes is WHICH bf16 rounding the
// GEMM operand carries (x instead of LN(x): the same relative precision per element).  The first block's statistics come from one
// extra pass over x (rowstats_cast_kernel), the final LayerNorm stays a launch.
// Measured (round 3, tools/bench_gemm_fold.py, profiles/r03_ab_experiments.txt): alone, the four GEMMs of a block cost +21 us with the
// fold epilogues (qkv +3.2, proj +6.2, fc1 +8.4: its GELU epilogue is VALU-bound already, fc2 +3.5) against 2 x 13.7 us of LayerNorm +
// their launch gaps — a gain of ~10 us per block; INSIDE the step the same epilogues cost +38 us per block (the side branch competes
// for exactly the VALU / HBM time they add) and the step is 0.4 % SLOWER (teacher 5.351 / 5.356 ms with, 5.327 / 5.331 without; student
// 7.639 vs 7.601).  Correct and tested (tests/test_gpu_vit_lnfold.py), not faster: off.
#include <stdlib.h>

#include "common.h"
#include "gemm_variants.h"
#include "medp_hip.h"

namespace {
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct VitWs {
    size_t a0, patch, pos, x, h, qkv, att, f, xb, stats, total;
};

// fp32 rows [M, D] -> bf16 copy + per row and 256-column tile (sum, sum of squares): a wave per row, the row in registers (the lane /
// chunk order of layernorm_fwd_reg_kernel: lane i holds float4 chunks i, i + 64, ...; chunk group k IS column tile k).  Rows M .. Mpad - 1
// of `stats` are zeroed (the consumer GEMM stages whole 256-row tiles of it).
template <int NT>
__global__ __launch_bounds__(256) void rowstats_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb, float* __restrict__ stats,
                                                            int M, int Mpad, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Mpad) return;
    float s1[NT], s2[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        s1[k] = s2[k] = 0.f;
        if (row < M) {
            const float4 v = *(const float4*)(x + (size_t)row * D + 4 * (lane + 64 * k));
            uint2 o;
            o.x = pack_bf2(v.x, v.y);
            o.y = pack_bf2(v.z, v.w);
            *(uint2*)(xb + (size_t)row * D + 4 * (lane + 64 * k)) = o;
            s1[k] = (v.x + v.y) + (v.z + v.w);
            s2[k] = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        s1[k] = wave_sum(s1[k]);
        s2[k] = wave_sum(s2[k]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) *(float2*)(stats + ((size_t)row * NT + k) * 2) = make_float2(s1[k], s2[k]);
    }
}
VitWs plan(const MedpVitWeights* w, int B, int H, int W) {
    const size_t P = (size_t)(H / w->patch) * (W / w->patch), S = P + 1, M = (size_t)B * S, D = w->hidden;
    VitWs s{};
    size_t off = 0;
    s.a0 = off;    off += al((size_t)B * P * w->patch_kpad * 2);
    s.patch = off; off += al((size_t)B * P * D * 4);
    s.pos = off;   off += al(S * D * 4);
    s.x = off;     off += al(M * D * 4);
    // The block loop's activations are aliased so that its whole working set (x 50 MB + h 25 MB + max(qkv, f) 101 MB at
    // B = 64, 224 px) stays inside the 256-MB Infinity Cache: att (attention output) takes the place of h (the LN1 output,
    // dead once qkv is computed; LN2 rewrites it after proj has consumed att), and f (the MLP hidden) overlays qkv (dead
    // after attention).  With separate buffers (277 MB) every GEMM epilogue burst went to HBM.
    s.h = off;     s.att = off;  off += al(M * D * 2);
    s.qkv = off;   s.f = off;    off += al(M * (size_t)(3 * D > (size_t)w->mlp_hidden ? 3 * D : (size_t)w->mlp_hidden) * 2);
    // LayerNorm fold: bf16(x) (NOT aliased with att: proj reads att as its A operand while its epilogue writes bf16(x)) + row statistics;
    // 176 + 25 MB at B = 64 still sits inside the Infinity Cache
    s.xb = off;    off += al(M * D * 2);
    s.stats = off; off += al(((M + 255) / 256 * 256) * (D / 256 + 1) * 2 * 4);
    s.total = off;
    return s;
}

int g_vit_lnfold = -1;          // medp_dbg_vit_lnfold (tests): 1 / 0 force the fold on / off, -1 the environment's choice
bool fold_wanted(const MedpVitWeights* w, int M) {
    static const int env_on = [] { const char* e = getenv("MEDP_VIT_LNFOLD"); return e ? atoi(e) : 0; }();
    const int on = g_vit_lnfold >= 0 ? g_vit_lnfold : env_on;
    const int D = w->hidden;
    if (!on || D % 256 != 0 || D / 256 > 4 || w->n_layers <= 0) return false;
    for (int l = 0; l < w->n_layers; ++l) {
        const MedpVitLayer& L = w->layers[l];
        if (!L.qkv_wg || !L.fc1_wg || !L.qkv_cs || !L.qkv_b2 || !L.fc1_cs || !L.fc1_b2) return false;
    }
    return medp_gemm_fold_eligible(M, 3 * D, D) && medp_gemm_fold_eligible(M, w->mlp_hidden, D) && medp_gemm_fold_eligible(M, D, D) &&
           medp_gemm_fold_eligible(M, D, w->mlp_hidden);
}
}  // namespace

extern "C" size_t medp_vit_workspace_bytes(const MedpVitWeights* w, int B, int H, int W) {
    if (!w || B <= 0 || H <= 0 || W <= 0 || w->patch <= 0) return 0;
    return plan(w, B, H, W).total;
}

extern "C" int medp_vit_forward(const MedpVitWeights* w, const float* pixels, int B, int H, int W, float* tokens_f32,
                                void* tokens_bf16, void* workspace, size_t workspace_bytes, void* stream) {
    MEDP_CHECK_ARG(w, "vit_forward: null argument");
    return medp_vit_forward_part(w, pixels, B, H, W, tokens_f32, tokens_bf16, workspace, workspace_bytes, 0, w->n_layers, stream);
}

extern "C" int medp_vit_forward_part(const MedpVitWeights* w, const float* pixels, int B, int H, int W, float* tokens_f32,
                                     void* tokens_bf16, void* workspace, size_t workspace_bytes, int first_layer, int last_layer,
                                     void* stream) {
    MEDP_CHECK_ARG(w && pixels && workspace, "vit_forward: null argument");
    MEDP_CHECK_ARG(0 <= first_layer && first_layer <= last_layer && last_layer <= w->n_layers,
                   "vit_forward: bad layer range [%d, %d) of %d", first_layer, last_layer, w->n_layers);
    const bool embed = first_layer == 0, finish = last_layer == w->n_layers;
    MEDP_CHECK_ARG(!finish || tokens_f32 || tokens_bf16, "vit_forward: no output requested");
    MEDP_CHECK_ARG(w->hidden == w->n_heads * 64, "vit_forward: head dim must be 64 (hidden %d, heads %d)", w->hidden, w->n_heads);
    MEDP_CHECK_ARG(H >= w->patch && W >= w->patch, "vit_forward: image %dx%d smaller than one patch (%d)", H, W, w->patch);
    const VitWs ws = plan(w, B, H, W);
    MEDP_CHECK_ARG(workspace_bytes >= ws.total, "vit_forward: workspace %zu < required %zu", workspace_bytes, ws.total);
    char* base = (char*)workspace;
    const int gh = H / w->patch, gw = W / w->patch, P = gh * gw, S = P + 1, M = B * S, D = w->hidden;
    void* a0 = base + ws.a0;
    float* patch = (float*)(base + ws.patch);
    float* pos = (float*)(base + ws.pos);
    float* x = (float*)(base + ws.x);
    void* h = base + ws.h;
    void* qkv = base + ws.qkv;
    void* att = base + ws.att;
    void* f = base + ws.f;

    if (embed) {
        MEDP_TRY(medp_im2col_patch(pixels, a0, B, 3, H, W, w->patch, w->patch_kpad, stream));
        MEDP_TRY(medp_gemm_bf16_nt(a0, w->patch_w, patch, B * P, D, w->patch_kpad, w->patch_kpad, w->patch_kpad, D, w->patch_b,
                                   nullptr, nullptr, 0, 0, 0, stream));
        const float* pos_used = w->pos;
        if (!(gh == w->pos_side && gw == w->pos_side)) {
            MEDP_TRY(medp_pos_embed_bicubic(w->pos, pos, w->pos_side, gh, gw, D, stream));
            pos_used = pos;
        }
        MEDP_TRY(medp_vit_assemble(patch, w->cls, pos_used, x, B, P, D, stream));
    }
    const float scale = 0.125f;   // 64^-0.5
    if (fold_wanted(w, M) && first_layer < last_layer) {
        bf16_t* xb = (bf16_t*)(base + ws.xb);
        float* stats = (float*)(base + ws.stats);
        const int NT = D / 256, Mpad = (M + 255) / 256 * 256;
        hipStream_t st = (hipStream_t)stream;
        // the first block of this call: statistics and bf16 copy of x as it stands (embedding stage, or the previous call's last block)
        if (NT == 1) rowstats_cast_kernel<1><<<(Mpad + 3) / 4, 256, 0, st>>>(x, xb, stats, M, Mpad, D);
        else if (NT == 2) rowstats_cast_kernel<2><<<(Mpad + 3) / 4, 256, 0, st>>>(x, xb, stats, M, Mpad, D);
        else if (NT == 3) rowstats_cast_kernel<3><<<(Mpad + 3) / 4, 256, 0, st>>>(x, xb, stats, M, Mpad, D);
        else rowstats_cast_kernel<4><<<(Mpad + 3) / 4, 256, 0, st>>>(x, xb, stats, M, Mpad, D);
        MEDP_LAUNCH_CHECK("vit rowstats_cast");
        MedpGemmFold cons{}, prod{};
        cons.stats_in = stats; cons.stats_tiles = NT; cons.ln_eps = w->ln_eps; cons.ln_dim = D;
        prod.c2 = xb; prod.ldc2 = D; prod.stats_out = stats;
        for (int l = first_layer; l < last_layer; ++l) {
            const MedpVitLayer& L = w->layers[l];
            cons.colsum = L.qkv_cs;
            MEDP_TRY(medp_gemm_bf16_nt_fold(xb, L.qkv_wg, qkv, M, 3 * D, D, D, D, 3 * D, L.qkv_b2, nullptr, nullptr, 0, 0, 1, cons, stream));
            MEDP_TRY(medp_attn_fwd_dh64(qkv, (const bf16_t*)qkv + D, (const bf16_t*)qkv + 2 * D, att, B, S, w->n_heads, 3 * D, 3 * D,
                                        3 * D, D, scale, stream));
            MEDP_TRY(medp_gemm_bf16_nt_fold(att, L.proj_w, x, M, D, D, D, D, D, L.proj_b, L.ls1, x, D, 0, 0, prod, stream));
            cons.colsum = L.fc1_cs;
            MEDP_TRY(medp_gemm_bf16_nt_fold(xb, L.fc1_wg, f, M, w->mlp_hidden, D, D, D, w->mlp_hidden, L.fc1_b2, nullptr, nullptr, 0, 1, 1, cons, stream));
            MEDP_TRY(medp_gemm_bf16_nt_fold(f, L.fc2_w, x, M, D, w->mlp_hidden, w->mlp_hidden, w->mlp_hidden, D, L.fc2_b, L.ls2, x, D, 0, 0, prod, stream));
        }
    } else {
        for (int l = first_layer; l < last_layer; ++l) {
            const MedpVitLayer& L = w->layers[l];
            MEDP_TRY(medp_layernorm_fwd(x, D, L.ln1_w, L.ln1_b, h, D, 1, nullptr, nullptr, M, D, w->ln_eps, stream));
            MEDP_TRY(medp_gemm_bf16_nt_tagged(1, h, L.qkv_w, qkv, M, 3 * D, D, D, D, 3 * D, L.qkv_b, nullptr, nullptr, 0, 0, 1, stream));
            MEDP_TRY(medp_attn_fwd_dh64(qkv, (const bf16_t*)qkv + D, (const bf16_t*)qkv + 2 * D, att, B, S, w->n_heads, 3 * D, 3 * D,
                                        3 * D, D, scale, stream));
            MEDP_TRY(medp_gemm_bf16_nt_tagged(1, att, L.proj_w, x, M, D, D, D, D, D, L.proj_b, L.ls1, x, D, 0, 0, stream));
            MEDP_TRY(medp_layernorm_fwd(x, D, L.ln2_w, L.ln2_b, h, D, 1, nullptr, nullptr, M, D, w->ln_eps, stream));
            MEDP_TRY(medp_gemm_bf16_nt_tagged(1, h, L.fc1_w, f, M, w->mlp_hidden, D, D, D, w->mlp_hidden, L.fc1_b, nullptr, nullptr, 0, 1, 1, stream));
            MEDP_TRY(medp_gemm_bf16_nt_tagged(1, f, L.fc2_w, x, M, D, w->mlp_hidden, w->mlp_hidden, w->mlp_hidden, D, L.fc2_b, L.ls2, x, D, 0, 0, stream));
        }
    }
    if (!finish) return 0;                 // the fp32 token stream stays in the workspace for the call that continues
    if (tokens_f32)
        MEDP_TRY(medp_layernorm_fwd(x, D, w->final_ln_w, w->final_ln_b, tokens_f32, D, 0, nullptr, nullptr, M, D, w->ln_eps, stream));
    if (tokens_bf16)
        MEDP_TRY(medp_layernorm_fwd(x, D, w->final_ln_w, w->final_ln_b, tokens_bf16, D, 1, nullptr, nullptr, M, D, w->ln_eps, stream));
    return 0;
}

// Debug hooks (NOT part of the C ABI in include/medp_hip.h; tests/test_gpu_vit_lnfold.py).
//   medp_dbg_vit_lnfold(1 / 0 / -1): force the LayerNorm fold on / off / back to MEDP_VIT_LNFOLD; returns the previous setting.
//   medp_dbg_gemm_fold: one block GEMM with the fold's producer (c2 + stats_out) or consumer (stats_in + colsum) epilogue.
extern "C" int medp_dbg_vit_lnfold(int on) {
    const int prev = g_vit_lnfold;
    g_vit_lnfold = on < 0 ? -1 : (on != 0);
    return prev;
}

extern "C" int medp_dbg_gemm_fold(const void* A, const void* W, void* C, int M, int N, int K, const float* bias, const float* scale,
                                  const float* residual, int act, int out_bf16, void* c2, float* stats_out, const float* stats_in,
                                  int stats_tiles, const float* colsum, float ln_eps, int ln_dim, void* stream) {
    MedpGemmFold f{};
    f.c2 = c2; f.ldc2 = N; f.stats_out = stats_out;
    f.stats_in = stats_in; f.stats_tiles = stats_tiles; f.colsum = colsum; f.ln_eps = ln_eps; f.ln_dim = ln_dim;
    return medp_gemm_bf16_nt_fold(A, W, C, M, N, K, K, K, N, bias, scale, residual, N, act, out_bf16, f, stream);
}
