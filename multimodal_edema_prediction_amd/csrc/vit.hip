// Whole-module forward of the frozen CXR encoder (ViT-B/14, Dinov2 layout) as ONE C call: every kernel of the
// 12 blocks is enqueued from C++ on the caller's stream (no Python between launches, graph-capturable).
//   tokens = LN_f( blocks( [cls; conv14(pixels)] + pos ) )        reference: model :152-158 -> modeling_dinov2.py
// Data layout in HBM: residual stream x fp32 [B*S, hidden]; every GEMM input is bf16 written by the producing
// kernel (LayerNorm / GELU epilogue / attention); LayerScale and the residual add are fused in GEMM epilogues.
#include "common.h"
#include "medp_hip.h"

namespace {
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct VitWs {
    size_t a0, patch, pos, x, h, qkv, att, f, total;
};
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
    s.total = off;
    return s;
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
    if (!finish) return 0;                 // the fp32 token stream stays in the workspace for the call that continues
    if (tokens_f32)
        MEDP_TRY(medp_layernorm_fwd(x, D, w->final_ln_w, w->final_ln_b, tokens_f32, D, 0, nullptr, nullptr, M, D, w->ln_eps, stream));
    if (tokens_bf16)
        MEDP_TRY(medp_layernorm_fwd(x, D, w->final_ln_w, w->final_ln_b, tokens_bf16, D, 1, nullptr, nullptr, M, D, w->ln_eps, stream));
    return 0;
}
