// Pointwise / tiny-reduction kernels of the cross-modal fusion head and of the losses (gfx950).
// Everything here is launch-bound ([B,7] logits, [B*7,256] latents): the point is ONE launch per logical op, fp32
// math identical to the reference, and dropout masks that are regenerated (not stored) in backward.
#include "common.h"
#include "medp_hip.h"

namespace {

inline int grid_for(size_t work_items) { return (int)min((size_t)2048, max((size_t)1, (work_items + 255) / 256)); }

// y = dropout(gelu(x))                        (nn.GELU -> nn.Dropout, model :755, :573)
template <bool Y16>
__global__ __launch_bounds__(256) void gelu_dropout_fwd_kernel(const float* __restrict__ x, void* __restrict__ y, size_t n, float p,
                                                               float inv_keep, uint32_t seed0, uint32_t sid, const uint32_t* __restrict__ epoch) {
    const uint32_t seed = medp_mix_epoch(seed0, epoch);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = gelu_erf(x[i]);
        if (p > 0.f) v *= dropout_scale(seed, sid, (uint32_t)i, p, inv_keep);
        if constexpr (Y16) ((bf16_t*)y)[i] = f2bf(v);          // the next Linear's operand directly (same rounding as the cast kernel)
        else ((float*)y)[i] = v;
    }
}
// dx = dy * mask * gelu'(x); DX16: also the bf16 copy the previous Linear's two gradient GEMMs take as their operand
template <bool DX16>
__global__ __launch_bounds__(256) void gelu_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dx, bf16_t* __restrict__ dx16, size_t n, float p, float inv_keep,
                                                               uint32_t seed0, uint32_t sid, const uint32_t* __restrict__ epoch) {
    const uint32_t seed = medp_mix_epoch(seed0, epoch);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float g = dy[i] * gelu_erf_grad(x[i]);
        if (p > 0.f) g *= dropout_scale(seed, sid, (uint32_t)i, p, inv_keep);
        dx[i] = g;
        if constexpr (DX16) dx16[i] = f2bf(g);
    }
}
// out = res + dropout(y)  (fwd, res may be null)   |   dy = dout * mask  (bwd: res == nullptr, y = dout)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                          float* __restrict__ out, size_t n, float p, float inv_keep, uint32_t seed0,
                                                          uint32_t sid, const uint32_t* __restrict__ epoch) {
    const uint32_t seed = medp_mix_epoch(seed0, epoch);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = y[i];
        if (p > 0.f) v *= dropout_scale(seed, sid, (uint32_t)i, p, inv_keep);
        out[i] = res ? res[i] + v : v;
    }
}

// y[m] = <x[m,:], w> (+ b)        — the 64 -> 1 output layer of the pathology heads (model :574, :590); wave per row
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int rows, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += x[(size_t)row * ldx + c] * w[c];
    a = wave_sum(a);
    if (lane == 0) y[row] = a + (b ? b[0] : 0.f);
}
// dx[m,c] = dy[m]*w[c] ; dw[c] = sum_m dy[m]*x[m,c] ; db = sum_m dy[m]
// grid = ceil(D/64) blocks of 64 columns x 16 row-lanes (block 0 also writes db).  The heads have D = 64 and rows = B*K = 448,
// i.e. ONE block: 16 row-lanes (1024 threads) keep the per-thread chain at 28 dependent iterations instead of 112 (48 -> ~12 us,
// on the serial tail of the step behind the CXR encoder).
__global__ __launch_bounds__(1024) void rowdot_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ dw,
                                                          float* __restrict__ db, int rows, int D) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float a = 0.f;
    if (c < D) {
        const float wc = w[c];
#pragma unroll 4
        for (int m = rl; m < rows; m += 16) {
            const float g = dy[m];
            dx[(size_t)m * D + c] = g * wc;
            a += g * x[(size_t)m * ldx + c];
        }
    }
    red[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        dw[c] = t;
    }
    if (db && blockIdx.x == 0) {
        __syncthreads();
        float bs = 0.f;
        for (int m = threadIdx.x; m < rows; m += 1024) bs += dy[m];
        bs = wave_sum(bs);
        if (cl == 0) red[0][rl] = bs;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[0][k];
            db[0] = t;
        }
    }
}

// residual-fusion logit assembly (model :634-639):
//   img = hi + image_label_bias ; ts = ht + temporal_label_bias ; corr = hc ; scaled = beta*corr ; fus = img.detach() + scaled
__global__ __launch_bounds__(256) void fusion_logits_fwd_kernel(const float* __restrict__ hi, const float* __restrict__ ht,
                                                                const float* __restrict__ hc, const float* __restrict__ ib,
                                                                const float* __restrict__ tb, const float* __restrict__ beta,
                                                                float* __restrict__ img, float* __restrict__ ts, float* __restrict__ scaled,
                                                                float* __restrict__ fus, int B, int K) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * K; i += gridDim.x * 256) {
        const int k = i % K;
        const float a = hi[i] + ib[k], s = beta[k] * hc[i];
        img[i] = a;
        ts[i] = ht[i] + tb[k];
        scaled[i] = s;
        fus[i] = a + s;
    }
}
// grads: d_hi = d_img (fus is detached from img); d_ht = d_ts; d_hc = beta*(d_scaled + d_fus); per-label sums for the biases / beta
__global__ __launch_bounds__(64) void fusion_logits_bwd_kernel(const float* __restrict__ d_img, const float* __restrict__ d_ts,
                                                               const float* __restrict__ d_scaled, const float* __restrict__ d_fus,
                                                               const float* __restrict__ hc, const float* __restrict__ beta,
                                                               float* __restrict__ d_hi, float* __restrict__ d_ht, float* __restrict__ d_hc,
                                                               float* __restrict__ d_ib, float* __restrict__ d_tb, float* __restrict__ d_beta,
                                                               int B, int K) {
    const int k = threadIdx.x;
    if (k >= K) return;
    float sib = 0.f, stb = 0.f, sbe = 0.f;
    for (int b = 0; b < B; ++b) {
        const int i = b * K + k;
        const float gi = d_img ? d_img[i] : 0.f, gt = d_ts ? d_ts[i] : 0.f;
        const float gs = (d_scaled ? d_scaled[i] : 0.f) + (d_fus ? d_fus[i] : 0.f);
        d_hi[i] = gi;
        d_ht[i] = gt;
        d_hc[i] = beta[k] * gs;
        sib += gi;
        stb += gt;
        sbe += gs * hc[i];
    }
    d_ib[k] = sib;
    d_tb[k] = stb;
    d_beta[k] = sbe;
}

// ---- DualPathologyLoss (loss/losses_duett.py:131-194) forward + gradient in ONE single-block launch -------------------
// per branch r, label k:  per[r][k] = sum_b bce(l_bk, y_bk [,pos_weight_k]) * m_bk / (sum_b m_bk + eps)
// branch_total[r] = sum_k w_k per[r][k];  total = sum_r alpha_r branch_total[r]
// grad[r][b][k] = alpha_r * w_k * m_bk / (sum_b m_bk + eps) * dbce/dl
// out: [0] total, [1..3] branch totals, [4 .. 4+3K) per-label losses (img | ts | fus)
__device__ __forceinline__ float softplus_neg(float l) { return fmaxf(-l, 0.f) + log1pf(__expf(-fabsf(l))); }   // log(1+exp(-l))
__global__ __launch_bounds__(256) void dual_loss_kernel(const float* __restrict__ img, const float* __restrict__ ts,
                                                        const float* __restrict__ fus, const float* __restrict__ y,
                                                        const float* __restrict__ mask, const float* __restrict__ lw,
                                                        const float* __restrict__ pw, float a_img, float a_ts, float a_fus, float eps,
                                                        float* __restrict__ out, float* __restrict__ g_img, float* __restrict__ g_ts,
                                                        float* __restrict__ g_fus, int B, int K) {
    __shared__ float s_per[3][32];
    __shared__ float s_den[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* L[3] = {img, ts, fus};
    // wave w handles labels w, w+4, ...
    for (int k = wave; k < K; k += 4) {
        float den = 0.f, acc[3] = {0.f, 0.f, 0.f};
        const float pwk = pw ? pw[k] : 1.f;
        for (int b = lane; b < B; b += 64) {
            const float m = mask[b * K + k], yy = y[b * K + k];
            den += m;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float l = L[r][b * K + k];
                // BCEWithLogits with pos_weight: (1-y)*l + (1 + (pw-1)*y) * log(1+exp(-l))
                acc[r] += ((1.f - yy) * l + (1.f + (pwk - 1.f) * yy) * softplus_neg(l)) * m;
            }
        }
        den = wave_sum(den);
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            s_den[k] = den + eps;
#pragma unroll
            for (int r = 0; r < 3; ++r) s_per[r][k] = acc[r] / (den + eps);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[3] = {0.f, 0.f, 0.f};
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < K; ++k) {
                tot[r] += lw[k] * s_per[r][k];
                out[4 + r * K + k] = s_per[r][k];
            }
        out[0] = a_img * tot[0] + a_ts * tot[1] + a_fus * tot[2];
        out[1] = tot[0];
        out[2] = tot[1];
        out[3] = tot[2];
    }
    float* G[3] = {g_img, g_ts, g_fus};
    const float A[3] = {a_img, a_ts, a_fus};
    for (int i = threadIdx.x; i < B * K; i += 256) {
        const int k = i % K;
        const float m = mask[i], yy = y[i], pwk = pw ? pw[k] : 1.f;
        const float c = lw[k] * m / s_den[k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (!G[r]) continue;
            const float l = L[r][i];
            const float sg = 1.f / (1.f + __expf(-l));
            // d/dl [(1-y) l + (1+(pw-1)y) softplus(-l)] = (1-y) - (1+(pw-1)y)(1-sigmoid(l))
            G[r][i] = A[r] * c * ((1.f - yy) - (1.f + (pwk - 1.f) * yy) * (1.f - sg));
        }
    }
}

// ---- StudentKDLoss (losses_duett.py:8-57): total = alpha*BCE(z_s,y) + (1-alpha)*T^2*mean KL(sig(z_t/T) || sig(z_s/T)) ----
// out: [0] total [1] bce [2] kd ; grad wrt z_s
__global__ __launch_bounds__(256) void kd_loss_kernel(const float* __restrict__ zs, const float* __restrict__ zt, const float* __restrict__ y,
                                                      float T, float alpha, float pos_weight, float eps, float* __restrict__ out,
                                                      float* __restrict__ g, int B) {
    __shared__ float red[2][4];
    float bce = 0.f, kl = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float l = zs[b], yy = y[b];
        bce += (1.f - yy) * l + (1.f + (pos_weight - 1.f) * yy) * softplus_neg(l);
        float pt = 1.f / (1.f + __expf(-zt[b] / T)), ps = 1.f / (1.f + __expf(-l / T));
        const float psc = fminf(fmaxf(ps, eps), 1.f - eps);
        pt = fminf(fmaxf(pt, eps), 1.f - eps);
        kl += pt * (logf(pt) - logf(psc)) + (1.f - pt) * (logf(1.f - pt) - logf(1.f - psc));
        if (g) {
            const float sg = 1.f / (1.f + __expf(-l));
            const float dbce = (1.f - yy) - (1.f + (pos_weight - 1.f) * yy) * (1.f - sg);
            // d kl / d ps = -pt/ps + (1-pt)/(1-ps) (zero where the clamp is active); d ps / d l = ps(1-ps)/T
            const float inside = (ps > eps && ps < 1.f - eps) ? 1.f : 0.f;
            const float dkl = inside * (-pt / psc + (1.f - pt) / (1.f - psc)) * ps * (1.f - ps) / T;
            g[b] = (alpha * dbce + (1.f - alpha) * T * T * dkl) / (float)B;
        }
    }
    bce = wave_sum(bce);
    kl = wave_sum(kl);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = bce;
        red[1][threadIdx.x >> 6] = kl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b_ = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)B;
        const float k_ = T * T * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)B;
        out[0] = alpha * b_ + (1.f - alpha) * k_;
        out[1] = b_;
        out[2] = k_;
    }
}

// mean over the first T tokens of [B, T+1, D] (StudentModel pool="mean", model :1231) and its backward
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, int T1, int D) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * D; i += gridDim.x * 256) {
        const int b = i / D, d = i % D;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += x[((size_t)b * T1 + t) * D + d];
        y[i] = a / (float)T;
    }
}
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int T, int T1, int D) {
    const size_t n = (size_t)B * T1 * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % D), t = (int)((i / D) % T1), b = (int)(i / ((size_t)D * T1));
        dx[i] = t < T ? dy[(size_t)b * D + d] / (float)T : 0.f;
    }
}

// ---- engine extras (training_duett/engine.py:149-165, 217-223) and the linear-probe loss (cxr_linear_training.ipynb:426-437) ----
// aux residual KL: KL(Bern(y_smooth) || Bern(sigmoid(img.detach() + scaled))) masked mean; grad wrt `scaled` only.
__global__ __launch_bounds__(256) void aux_kl_kernel(const float* __restrict__ img, const float* __restrict__ scaled, const float* __restrict__ y,
                                                     const float* __restrict__ mask, float smooth, float* __restrict__ out,
                                                     float* __restrict__ g, int n) {
    __shared__ float red[2][4];
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float ys = y[i] * (1.f - smooth) + (1.f - y[i]) * smooth;
        const float pr = 1.f / (1.f + __expf(-(img[i] + scaled[i])));
        const float p = fminf(fmaxf(pr, 1e-6f), 1.f - 1e-6f);
        num += (ys * (logf(ys) - logf(p)) + (1.f - ys) * (logf(1.f - ys) - logf(1.f - p))) * mask[i];
        den += mask[i];
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = num; red[1][threadIdx.x >> 6] = den; }
    __syncthreads();
    const float D = fmaxf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), 1.f);
    if (threadIdx.x == 0) out[0] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / D;
    if (g) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float ys = y[i] * (1.f - smooth) + (1.f - y[i]) * smooth;
            const float pr = 1.f / (1.f + __expf(-(img[i] + scaled[i])));
            const float inside = (pr > 1e-6f && pr < 1.f - 1e-6f) ? 1.f : 0.f;
            const float p = fminf(fmaxf(pr, 1e-6f), 1.f - 1e-6f);
            g[i] = inside * (-ys / p + (1.f - ys) / (1.f - p)) * pr * (1.f - pr) * mask[i] / D;
        }
    }
}
// out = coef * mean(x^2), g = 2*coef*x/n        (LP regularisers beta_l2*mean(beta^2), corr_l2*mean(scaled^2))
__global__ __launch_bounds__(256) void sq_mean_kernel(const float* __restrict__ x, float coef, float* __restrict__ out, float* __restrict__ g, int n) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += x[i] * x[i];
        if (g) g[i] = 2.f * coef * x[i] / (float)n;
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = coef * ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}
// one global masked mean: sum(bce*m)/max(sum(m),1)   (config 2 linear probe)
__global__ __launch_bounds__(256) void masked_bce_kernel(const float* __restrict__ l, const float* __restrict__ y, const float* __restrict__ mask,
                                                         float* __restrict__ out, float* __restrict__ g, int n) {
    __shared__ float red[2][4];
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        num += ((1.f - y[i]) * l[i] + softplus_neg(l[i])) * mask[i];
        den += mask[i];
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = num; red[1][threadIdx.x >> 6] = den; }
    __syncthreads();
    const float D = fmaxf((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), 1.f);
    if (threadIdx.x == 0) out[0] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / D;
    if (g)
        for (int i = threadIdx.x; i < n; i += 256) g[i] = (1.f / (1.f + __expf(-l[i])) - y[i]) * mask[i] / D;
}

}  // namespace

extern "C" int medp_gelu_dropout_fwd(const float* x, float* y, long long n, float p, unsigned seed, unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f, "gelu_dropout_fwd: bad argument");
    gelu_dropout_fwd_kernel<false><<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(x, y, (size_t)n, p, 1.f / (1.f - p), seed, stream_id, medp_rng_epoch_ptr());
    MEDP_LAUNCH_CHECK("medp_gelu_dropout_fwd");
    return 0;
}
extern "C" int medp_gelu_dropout_fwd_bf16(const float* x, void* y_bf16, long long n, float p, unsigned seed, unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(x && y_bf16 && n > 0 && p >= 0.f && p < 1.f, "gelu_dropout_fwd_bf16: bad argument");
    gelu_dropout_fwd_kernel<true><<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(x, y_bf16, (size_t)n, p, 1.f / (1.f - p), seed, stream_id, medp_rng_epoch_ptr());
    MEDP_LAUNCH_CHECK("medp_gelu_dropout_fwd_bf16");
    return 0;
}
extern "C" int medp_gelu_dropout_bwd(const float* dy, const float* x, float* dx, long long n, float p, unsigned seed,
                                     unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(dy && x && dx && n > 0 && p >= 0.f && p < 1.f, "gelu_dropout_bwd: bad argument");
    gelu_dropout_bwd_kernel<false><<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(dy, x, dx, nullptr, (size_t)n, p, 1.f / (1.f - p), seed, stream_id, medp_rng_epoch_ptr());
    MEDP_LAUNCH_CHECK("medp_gelu_dropout_bwd");
    return 0;
}
extern "C" int medp_gelu_dropout_bwd_bf16(const float* dy, const float* x, float* dx, void* dx_bf16, long long n, float p, unsigned seed,
                                          unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(dy && x && dx && dx_bf16 && n > 0 && p >= 0.f && p < 1.f, "gelu_dropout_bwd_bf16: bad argument");
    gelu_dropout_bwd_kernel<true><<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(dy, x, dx, (bf16_t*)dx_bf16, (size_t)n, p, 1.f / (1.f - p), seed, stream_id,
                                                                                       medp_rng_epoch_ptr());
    MEDP_LAUNCH_CHECK("medp_gelu_dropout_bwd_bf16");
    return 0;
}
extern "C" int medp_dropout_add(const float* y, const float* residual, float* out, long long n, float p, unsigned seed,
                                unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(y && out && n > 0 && p >= 0.f && p < 1.f, "dropout_add: bad argument");
    dropout_add_kernel<<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(y, residual, out, (size_t)n, p, 1.f / (1.f - p), seed, stream_id, medp_rng_epoch_ptr());
    MEDP_LAUNCH_CHECK("medp_dropout_add");
    return 0;
}
extern "C" int medp_rowdot_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int rows, int D, void* stream) {
    MEDP_CHECK_ARG(x && w && y && rows > 0 && D > 0, "rowdot_fwd: bad argument");
    rowdot_fwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ldx, w, b, y, rows, D);
    MEDP_LAUNCH_CHECK("medp_rowdot_fwd");
    return 0;
}
extern "C" int medp_rowdot_bwd(const float* dy, const float* x, int ldx, const float* w, float* dx, float* dw, float* db, int rows,
                               int D, void* stream) {
    MEDP_CHECK_ARG(dy && x && w && dx && dw && rows > 0 && D > 0, "rowdot_bwd: bad argument");
    rowdot_bwd_kernel<<<(D + 63) / 64, 1024, 0, (hipStream_t)stream>>>(dy, x, ldx, w, dx, dw, db, rows, D);
    MEDP_LAUNCH_CHECK("medp_rowdot_bwd");
    return 0;
}
extern "C" int medp_fusion_logits_fwd(const float* hi, const float* ht, const float* hc, const float* img_bias, const float* ts_bias,
                                      const float* beta, float* img, float* ts, float* scaled, float* fus, int B, int K, void* stream) {
    MEDP_CHECK_ARG(hi && ht && hc && img_bias && ts_bias && beta && img && ts && scaled && fus && B > 0 && K > 0, "fusion_logits_fwd: bad argument");
    fusion_logits_fwd_kernel<<<grid_for((size_t)B * K), 256, 0, (hipStream_t)stream>>>(hi, ht, hc, img_bias, ts_bias, beta, img, ts, scaled, fus, B, K);
    MEDP_LAUNCH_CHECK("medp_fusion_logits_fwd");
    return 0;
}
extern "C" int medp_fusion_logits_bwd(const float* d_img, const float* d_ts, const float* d_scaled, const float* d_fus, const float* hc,
                                      const float* beta, float* d_hi, float* d_ht, float* d_hc, float* d_img_bias, float* d_ts_bias,
                                      float* d_beta, int B, int K, void* stream) {
    MEDP_CHECK_ARG(hc && beta && d_hi && d_ht && d_hc && d_img_bias && d_ts_bias && d_beta && B > 0 && K > 0 && K <= 64, "fusion_logits_bwd: bad argument");
    fusion_logits_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(d_img, d_ts, d_scaled, d_fus, hc, beta, d_hi, d_ht, d_hc, d_img_bias, d_ts_bias, d_beta, B, K);
    MEDP_LAUNCH_CHECK("medp_fusion_logits_bwd");
    return 0;
}
extern "C" int medp_dual_pathology_loss(const float* img, const float* ts, const float* fus, const float* y, const float* mask,
                                        const float* label_weights, const float* pos_weight, float alpha_img, float alpha_ts,
                                        float alpha_fus, float eps, float* out, float* g_img, float* g_ts, float* g_fus, int B, int K,
                                        void* stream) {
    MEDP_CHECK_ARG(img && ts && fus && y && mask && label_weights && out, "dual_pathology_loss: null argument");
    MEDP_CHECK_ARG(B > 0 && K > 0 && K <= 32, "dual_pathology_loss: need 0 < K <= 32 (got %d)", K);
    dual_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(img, ts, fus, y, mask, label_weights, pos_weight, alpha_img, alpha_ts, alpha_fus, eps,
                                                         out, g_img, g_ts, g_fus, B, K);
    MEDP_LAUNCH_CHECK("medp_dual_pathology_loss");
    return 0;
}
extern "C" int medp_student_kd_loss(const float* z_s, const float* z_t, const float* y, float T, float alpha, float pos_weight,
                                    float* out, float* g_zs, int B, void* stream) {
    MEDP_CHECK_ARG(z_s && z_t && y && out && B > 0 && T > 0.f, "student_kd_loss: bad argument");
    kd_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(z_s, z_t, y, T, alpha, pos_weight, 1e-7f, out, g_zs, B);
    MEDP_LAUNCH_CHECK("medp_student_kd_loss");
    return 0;
}
extern "C" int medp_meanpool_fwd(const float* x, float* y, int B, int T, int T1, int D, void* stream) {
    MEDP_CHECK_ARG(x && y && B > 0 && T > 0 && T1 >= T && D > 0, "meanpool_fwd: bad argument");
    meanpool_fwd_kernel<<<grid_for((size_t)B * D), 256, 0, (hipStream_t)stream>>>(x, y, B, T, T1, D);
    MEDP_LAUNCH_CHECK("medp_meanpool_fwd");
    return 0;
}
extern "C" int medp_meanpool_bwd(const float* dy, float* dx, int B, int T, int T1, int D, void* stream) {
    MEDP_CHECK_ARG(dy && dx && B > 0 && T > 0 && T1 >= T && D > 0, "meanpool_bwd: bad argument");
    meanpool_bwd_kernel<<<grid_for((size_t)B * T1 * D), 256, 0, (hipStream_t)stream>>>(dy, dx, B, T, T1, D);
    MEDP_LAUNCH_CHECK("medp_meanpool_bwd");
    return 0;
}

extern "C" int medp_aux_residual_kl(const float* img_logits, const float* scaled_correction, const float* y, const float* mask,
                                    float label_smoothing, float* out, float* g_scaled, int n, void* stream) {
    MEDP_CHECK_ARG(img_logits && scaled_correction && y && mask && out && n > 0, "aux_residual_kl: bad argument");
    aux_kl_kernel<<<1, 256, 0, (hipStream_t)stream>>>(img_logits, scaled_correction, y, mask, label_smoothing, out, g_scaled, n);
    MEDP_LAUNCH_CHECK("medp_aux_residual_kl");
    return 0;
}
extern "C" int medp_sq_mean(const float* x, float coef, float* out, float* g, int n, void* stream) {
    MEDP_CHECK_ARG(x && out && n > 0, "sq_mean: bad argument");
    sq_mean_kernel<<<1, 256, 0, (hipStream_t)stream>>>(x, coef, out, g, n);
    MEDP_LAUNCH_CHECK("medp_sq_mean");
    return 0;
}
extern "C" int medp_masked_bce_global(const float* logits, const float* y, const float* mask, float* out, float* g, int n, void* stream) {
    MEDP_CHECK_ARG(logits && y && mask && out && n > 0, "masked_bce_global: bad argument");
    masked_bce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, y, mask, out, g, n);
    MEDP_LAUNCH_CHECK("medp_masked_bce_global");
    return 0;
}

namespace {
__global__ void counter_advance_kernel(unsigned* c) { c[0] += 1u; }
}
extern "C" int medp_counter_advance(unsigned* dev_counter, void* stream) {
    MEDP_CHECK_ARG(dev_counter, "counter_advance: null pointer");
    counter_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dev_counter);
    MEDP_LAUNCH_CHECK("medp_counter_advance");
    return 0;
}
