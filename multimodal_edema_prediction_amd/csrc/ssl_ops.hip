// Losses of the DuETT-only step (BASELINE.json configs[0]; reference duett/duett.py:337-365): masked MSE of the value
// read-outs and (optionally weighted) mean BCE-with-logits; value + gradient in one single-block launch each.
#include "common.h"
#include "medp_hip.h"

namespace {
__device__ __forceinline__ float softplus_neg2(float l) { return fmaxf(-l, 0.f) + log1pf(__expf(-fabsf(l))); }

// out = mean(((a - b) * m)^2) ; g = 2 (a - b) m^2 / n
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m,
                                                         float* __restrict__ out, float* __restrict__ g, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float mm = m ? m[i] : 1.f, d = (a[i] - b[i]) * mm;
        s += d * d;
        if (g) g[i] = 2.f * d * mm / (float)n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}
// out = mean(w * bce(l, y)) ; g = w (sigmoid(l) - y) / n          (w may be NULL)
__global__ __launch_bounds__(256) void bce_mean_kernel(const float* __restrict__ l, const float* __restrict__ y, const float* __restrict__ w,
                                                       float* __restrict__ out, float* __restrict__ g, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float ww = w ? w[i] : 1.f;
        s += ww * ((1.f - y[i]) * l[i] + softplus_neg2(l[i]));
        if (g) g[i] = ww * (1.f / (1.f + __expf(-l[i])) - y[i]) / (float)n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}
}  // namespace

extern "C" int medp_masked_mse(const float* a, const float* b, const float* mask, float* out, float* g_a, int n, void* stream) {
    MEDP_CHECK_ARG(a && b && out && n > 0, "masked_mse: bad argument");
    masked_mse_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, b, mask, out, g_a, n);
    MEDP_LAUNCH_CHECK("medp_masked_mse");
    return 0;
}
extern "C" int medp_bce_mean(const float* logits, const float* y, const float* weight, float* out, float* g, int n, void* stream) {
    MEDP_CHECK_ARG(logits && y && out && n > 0, "bce_mean: bad argument");
    bce_mean_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, y, weight, out, g, n);
    MEDP_LAUNCH_CHECK("medp_bce_mean");
    return 0;
}
