// Row normalisations for gfx950: LayerNorm (ViT, perceiver) and ScaleNorm (DuETT encoders), forward + backward.
// HBM-bound: one 64-lane wave per row, float4 loads, wavefront shuffle reductions, statistics in fp32.
// The row is read from HBM once (the second/third sweep of the same wave hits L1/L2); the output is written
// as bf16 when it feeds an MFMA GEMM, as fp32 when it is a residual stream.
#include <stdlib.h>

#include "common.h"
#include "medp_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm fwd
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ b, void* __restrict__ y, int ldy,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const int D4 = D >> 2;
    float s = 0.f;
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i);
        const float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
        ss += (a * a + c * c) + (d * d + e * e);
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i);
        const float4 ww = *(const float4*)(w + 4 * i);
        const float4 bb = *(const float4*)(b + 4 * i);
        const float o0 = (v.x - mean) * rstd * ww.x + bb.x, o1 = (v.y - mean) * rstd * ww.y + bb.y;
        const float o2 = (v.z - mean) * rstd * ww.z + bb.z, o3 = (v.w - mean) * rstd * ww.w + bb.w;
        if (OUT_BF16) {
            uint2 o;
            o.x = pack_bf2(o0, o1);
            o.y = pack_bf2(o2, o3);
            *(uint2*)((bf16_t*)y + (size_t)row * ldy + 4 * i) = o;
        } else {
            *(float4*)((float*)y + (size_t)row * ldy + 4 * i) = make_float4(o0, o1, o2, o3);
        }
    }
}

// Register-resident variant for D == 256*NV (ViT: NV=3, perceiver: NV=1): each lane keeps its NV float4 of the row, so the
// row crosses the memory pipeline exactly once.  Same per-lane summation order as the generic kernel (bit-identical).
template <bool OUT_BF16, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_reg_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                const float* __restrict__ b, void* __restrict__ y, int ldy,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                int rows, float eps) {
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *(const float4*)(xr + 4 * (lane + 64 * j));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float a = v[j].x - mean, c = v[j].y - mean, d = v[j].z - mean, e = v[j].w - mean;
        ss += (a * a + c * c) + (d * d + e * e);
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        const float4 ww = *(const float4*)(w + 4 * i);
        const float4 bb = *(const float4*)(b + 4 * i);
        const float o0 = (v[j].x - mean) * rstd * ww.x + bb.x, o1 = (v[j].y - mean) * rstd * ww.y + bb.y;
        const float o2 = (v[j].z - mean) * rstd * ww.z + bb.z, o3 = (v[j].w - mean) * rstd * ww.w + bb.w;
        if (OUT_BF16) {
            uint2 o;
            o.x = pack_bf2(o0, o1);
            o.y = pack_bf2(o2, o3);
            *(uint2*)((bf16_t*)y + (size_t)row * ldy + 4 * i) = o;
        } else {
            *(float4*)((float*)y + (size_t)row * ldy + 4 * i) = make_float4(o0, o1, o2, o3);
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm bwd (dx)
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                               int ldx, const float* __restrict__ w, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx, int lddx,
                                                               int rows, int D, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const float* gr = dy + (size_t)row * lddy;
    const float mu = mean[row], rs = rstd[row];
    const int D4 = D >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i), g = *(const float4*)(gr + 4 * i), ww = *(const float4*)(w + 4 * i);
        const float g0 = g.x * ww.x, g1 = g.y * ww.y, g2 = g.z * ww.z, g3 = g.w * ww.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += (g0 * (v.x - mu) + g1 * (v.y - mu)) + (g2 * (v.z - mu) + g3 * (v.w - mu));
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) * rs / (float)D;   // mean(g * xhat)
    float* dr = dx + (size_t)row * lddx;
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i), g = *(const float4*)(gr + 4 * i), ww = *(const float4*)(w + 4 * i);
        float4 o;
        o.x = rs * (g.x * ww.x - s1 - (v.x - mu) * rs * s2);
        o.y = rs * (g.y * ww.y - s1 - (v.y - mu) * rs * s2);
        o.z = rs * (g.z * ww.z - s1 - (v.z - mu) * rs * s2);
        o.w = rs * (g.w * ww.w - s1 - (v.w - mu) * rs * s2);
        if (accumulate) {
            const float4 old = *(const float4*)(dr + 4 * i);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *(float4*)(dr + 4 * i) = o;
    }
}

// ------------------------------------------------------------------------------------------------ column reductions
// partial[chunk][c]        = sum_{r in chunk} dy[r][c] * (x[r][c]-mean[r])*rstd[r]     (MODE 1: LayerNorm dweight)
// partial[nchunk+chunk][c] = sum_{r in chunk} dy[r][c]                                (bias grad / plain column sum)
// Deterministic two-stage reduction (no float atomics): results are bitwise reproducible.
template <int MODE>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ partial, int rows, int D, int rows_per_chunk) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y, nchunk = gridDim.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float a = 0.f, bsum = 0.f;
    if (c < D) {
        for (int r = r0 + rl; r < r1; r += 4) {
            const float g = dy[(size_t)r * lddy + c];
            bsum += g;
            if (MODE == 1) a += g * (x[(size_t)r * ldx + c] - mean[r]) * rstd[r];
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = bsum;
    __syncthreads();
    if (rl == 0 && c < D) {
        if (MODE == 1) partial[(size_t)chunk * D + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        partial[(size_t)(nchunk + chunk) * D + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out_a,
                                                           float* __restrict__ out_b, int nchunk, int D) {
    // 64 columns x 4 chunk lanes per workgroup: the chunk loop is 4x shorter and its loads are independent
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < D) {
#pragma unroll 4
        for (int k = kl; k < nchunk; k += 4) {
            if (out_a) a += partial[(size_t)k * D + c];
            b += partial[(size_t)(nchunk + k) * D + c];
        }
    }
    red[0][kl][cl] = a;
    red[1][kl][cl] = b;
    __syncthreads();
    if (kl == 0 && c < D) {
        if (out_a) out_a[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        if (out_b) out_b[c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

// ------------------------------------------------------------------------------------------------ ScaleNorm
// y = x / max(||x||_2, eps) * sqrt(D) * g        (x_transformers ScaleNorm; g is a 1-element parameter)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void scalenorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                            void* __restrict__ y, int ldy, float* __restrict__ rnorm_out, int rows,
                                                            int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const int D4 = D >> 2;
    float ss = 0.f;
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float rn = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
    if (lane == 0 && rnorm_out) rnorm_out[row] = rn;
    const float sc = rn * sqrtf((float)D) * g[0];
    for (int i = lane; i < D4; i += 64) {
        const float4 v = *(const float4*)(xr + 4 * i);
        if (OUT_BF16) {
            uint2 o;
            o.x = pack_bf2(v.x * sc, v.y * sc);
            o.y = pack_bf2(v.z * sc, v.w * sc);
            *(uint2*)((bf16_t*)y + (size_t)row * ldy + 4 * i) = o;
        } else {
            *(float4*)((float*)y + (size_t)row * ldy + 4 * i) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
    }
}

// The same with the row held in registers (D <= 4 * 64 * NV): ONE pass over memory and NV independent 16-B loads in flight per
// lane.  The two-pass kernel above keeps one load per wave in flight: at D = 2328 (DuETT's event-axis tokens, 29 MB per call)
// it ran at 1.9 TB/s, this one is bound by HBM.  Same per-lane summation order: the results are bit-identical.
template <bool OUT_BF16, int NV>
__global__ __launch_bounds__(256) void scalenorm_fwd_reg_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                                void* __restrict__ y, int ldy, float* __restrict__ rnorm_out,
                                                                int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const int D4 = D >> 2;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < D4 ? *(const float4*)(xr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < D4) ss += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    const float rn = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
    if (lane == 0 && rnorm_out) rnorm_out[row] = rn;
    const float sc = rn * sqrtf((float)D) * g[0];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < D4) {
            if (OUT_BF16) {
                uint2 o;
                o.x = pack_bf2(v[k].x * sc, v[k].y * sc);
                o.y = pack_bf2(v[k].z * sc, v[k].w * sc);
                *(uint2*)((bf16_t*)y + (size_t)row * ldy + 4 * i) = o;
            } else {
                *(float4*)((float*)y + (size_t)row * ldy + 4 * i) = make_float4(v[k].x * sc, v[k].y * sc, v[k].z * sc, v[k].w * sc);
            }
        }
    }
}

// dx = [acc_src +] s*rn*(dy - x*rn^2*<dy,x>),  s = sqrt(D)*g ;  dg_row = sqrt(D)*rn*<dy,x>   (summed over rows by sum_all_kernel).
// acc_src may alias dx (the in-place accumulate form) or be a third tensor: dx = acc_src + ..., the residual join of a pre-norm block
// without touching the incoming gradient.  (Measured and rejected: the LAST workgroup finishing the dg sum behind a ticket — the
// agent-scope fence every workgroup then needs writes back L2 1500 times per launch: student step 7.36 ms against 6.71,
// profiles/r03_ab_experiments.txt section 12.)
__global__ __launch_bounds__(256) void scalenorm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ g, const float* __restrict__ rnorm, const float* acc_src,
                                                            int ldacc, float* dx, int lddx, float* dg_rows, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    {
        const float* xr = x + (size_t)row * ldx;
        const float* gr = dy + (size_t)row * lddy;
        const int D4 = D >> 2;
        float dot = 0.f;
        for (int i = lane; i < D4; i += 64) {
            const float4 v = *(const float4*)(xr + 4 * i), d = *(const float4*)(gr + 4 * i);
            dot += (v.x * d.x + v.y * d.y) + (v.z * d.z + v.w * d.w);
        }
        dot = wave_sum(dot);
        const float rn = rnorm[row], sq = sqrtf((float)D);
        if (lane == 0 && dg_rows) dg_rows[row] = sq * rn * dot;
        const float s = sq * g[0] * rn, k = rn * rn * dot;
        float* dr = dx + (size_t)row * lddx;
        for (int i = lane; i < D4; i += 64) {
            const float4 v = *(const float4*)(xr + 4 * i), d = *(const float4*)(gr + 4 * i);
            float4 o = make_float4(s * (d.x - v.x * k), s * (d.y - v.y * k), s * (d.z - v.z * k), s * (d.w - v.w * k));
            if (acc_src) {
                const float4 old = *(const float4*)(acc_src + (size_t)row * ldacc + 4 * i);
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *(float4*)(dr + 4 * i) = o;
        }
    }
}

// The same with the row (x and dy) held in registers: one pass over the operands instead of two (D <= 256 NV floats).  The student
// step runs this 13 times over 29-MB rowsets: 23 us with the two-pass kernel.
template <int NV>
__global__ __launch_bounds__(256) void scalenorm_bwd_reg_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ g, const float* __restrict__ rnorm,
                                                                const float* acc_src, int ldacc, float* dx, int lddx, float* dg_rows, int rows,
                                                                int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    {
        const float* xr = x + (size_t)row * ldx;
        const float* gr = dy + (size_t)row * lddy;
        const int D4 = D >> 2;
        float4 v[NV], d[NV];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            const bool ok = i < D4;
            v[j] = ok ? *(const float4*)(xr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[j] = ok ? *(const float4*)(gr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) dot += (v[j].x * d[j].x + v[j].y * d[j].y) + (v[j].z * d[j].z + v[j].w * d[j].w);
        dot = wave_sum(dot);
        const float rn = rnorm[row], sq = sqrtf((float)D);
        if (lane == 0 && dg_rows) dg_rows[row] = sq * rn * dot;
        const float s = sq * g[0] * rn, k = rn * rn * dot;
        float* dr = dx + (size_t)row * lddx;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < D4) {
                float4 o = make_float4(s * (d[j].x - v[j].x * k), s * (d[j].y - v[j].y * k), s * (d[j].z - v[j].z * k), s * (d[j].w - v[j].w * k));
                if (acc_src) {
                    const float4 old = *(const float4*)(acc_src + (size_t)row * ldacc + 4 * i);
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *(float4*)(dr + 4 * i) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sum_all_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

int colsum_chunks(int rows) { return max(1, min(128, rows / 64)); }

// Few rows (the perceiver's B*7 = 448 latent rows): ONE launch, 64 columns x 16 row-lanes per workgroup, instead of the
// partial + final pair — these sums sit on the dependent backward chain, where a launch costs more than the arithmetic.
constexpr int COLSUM_DIRECT_MAX_ROWS = 2048;
template <int MODE>
__global__ __launch_bounds__(1024) void colsum_direct_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ out_a, float* __restrict__ out_b, int rows, int D) {
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, bsum = 0.f;
    if (c < D) {
#pragma unroll 4
        for (int r = rl; r < rows; r += 16) {
            const float g = dy[(size_t)r * lddy + c];
            bsum += g;
            if (MODE == 1) a += g * (x[(size_t)r * ldx + c] - mean[r]) * rstd[r];
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = bsum;
    __syncthreads();
    if (rl == 0 && c < D) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            sa += red[0][k][cl];
            sb += red[1][k][cl];
        }
        if (MODE == 1 && out_a) out_a[c] = sa;
        if (out_b) out_b[c] = sb;
    }
}

}  // namespace

extern "C" int medp_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, void* y, int ldy, int y_bf16,
                                  float* mean, float* rstd, int rows, int D, float eps, void* stream) {
    MEDP_CHECK_ARG(x && w && b && y, "layernorm_fwd: null operand");
    MEDP_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm_fwd: D, ldx, ldy must be multiples of 4");
    dim3 grid((rows + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (D == 768 && y_bf16)
        layernorm_fwd_reg_kernel<true, 3><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, eps);
    else if (D == 768)
        layernorm_fwd_reg_kernel<false, 3><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, eps);
    else if (D == 256 && y_bf16)
        layernorm_fwd_reg_kernel<true, 1><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, eps);
    else if (D == 256)
        layernorm_fwd_reg_kernel<false, 1><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, eps);
    else if (y_bf16)
        layernorm_fwd_kernel<true><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, D, eps);
    else
        layernorm_fwd_kernel<false><<<grid, 256, 0, s>>>(x, ldx, w, b, y, ldy, mean, rstd, rows, D, eps);
    MEDP_LAUNCH_CHECK("medp_layernorm_fwd");
    return 0;
}

extern "C" size_t medp_colsum_workspace_bytes(int rows, int D) { return (size_t)2 * colsum_chunks(rows) * D * sizeof(float); }

extern "C" int medp_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* w, const float* mean,
                                  const float* rstd, float* dx, int lddx, int accumulate_dx, float* dw, float* db,
                                  float* workspace, int rows, int D, void* stream) {
    MEDP_CHECK_ARG(dy && x && w && mean && rstd, "layernorm_bwd: null operand");
    MEDP_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "layernorm_bwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        layernorm_bwd_dx_kernel<<<(rows + 3) / 4, 256, 0, s>>>(dy, lddy, x, ldx, w, mean, rstd, dx, lddx, rows, D, accumulate_dx);
        MEDP_LAUNCH_CHECK("medp_layernorm_bwd(dx)");
    }
    if (dw || db) {
        MEDP_CHECK_ARG(workspace, "layernorm_bwd: workspace required for dw/db (medp_colsum_workspace_bytes)");
        if (rows <= COLSUM_DIRECT_MAX_ROWS) {
            colsum_direct_kernel<1><<<(D + 63) / 64, 1024, 0, s>>>(dy, lddy, x, ldx, mean, rstd, dw, db, rows, D);
            MEDP_LAUNCH_CHECK("medp_layernorm_bwd(direct)");
            return 0;
        }
        const int nchunk = colsum_chunks(rows), rpc = (rows + nchunk - 1) / nchunk;
        colsum_partial_kernel<1><<<dim3((D + 63) / 64, nchunk), 256, 0, s>>>(dy, lddy, x, ldx, mean, rstd, workspace, rows, D, rpc);
        MEDP_LAUNCH_CHECK("medp_layernorm_bwd(partial)");
        colsum_final_kernel<<<(D + 63) / 64, 256, 0, s>>>(workspace, dw, db, nchunk, D);
        MEDP_LAUNCH_CHECK("medp_layernorm_bwd(final)");
    }
    return 0;
}

extern "C" int medp_colsum_f32(const float* x, int ldx, float* out, float* workspace, int rows, int D, void* stream) {
    MEDP_CHECK_ARG(x && out && workspace && rows > 0 && D > 0, "colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (rows <= COLSUM_DIRECT_MAX_ROWS) {
        colsum_direct_kernel<0><<<(D + 63) / 64, 1024, 0, s>>>(x, ldx, nullptr, 0, nullptr, nullptr, nullptr, out, rows, D);
        MEDP_LAUNCH_CHECK("medp_colsum_f32(direct)");
        return 0;
    }
    const int nchunk = colsum_chunks(rows), rpc = (rows + nchunk - 1) / nchunk;
    colsum_partial_kernel<0><<<dim3((D + 63) / 64, nchunk), 256, 0, s>>>(x, ldx, nullptr, 0, nullptr, nullptr, workspace, rows, D, rpc);
    MEDP_LAUNCH_CHECK("medp_colsum_f32(partial)");
    colsum_final_kernel<<<(D + 63) / 64, 256, 0, s>>>(workspace, nullptr, out, nchunk, D);
    MEDP_LAUNCH_CHECK("medp_colsum_f32(final)");
    return 0;
}

extern "C" int medp_scalenorm_fwd(const float* x, int ldx, const float* g, void* y, int ldy, int y_bf16, float* rnorm, int rows,
                                  int D, float eps, void* stream) {
    MEDP_CHECK_ARG(x && g && y, "scalenorm_fwd: null operand");
    MEDP_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "scalenorm_fwd: D, ldx, ldy must be multiples of 4");
    dim3 grid((rows + 3) / 4);
    const int nv = (D / 4 + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define MEDP_SN_REG(NV)                                                                                             \
    do {                                                                                                            \
        if (y_bf16) scalenorm_fwd_reg_kernel<true, NV><<<grid, 256, 0, st>>>(x, ldx, g, y, ldy, rnorm, rows, D, eps);  \
        else scalenorm_fwd_reg_kernel<false, NV><<<grid, 256, 0, st>>>(x, ldx, g, y, ldy, rnorm, rows, D, eps);       \
    } while (0)
    if (nv <= 2) MEDP_SN_REG(2);
    else if (nv <= 5) MEDP_SN_REG(5);
    else if (nv <= 10) MEDP_SN_REG(10);
    else if (nv <= 16) MEDP_SN_REG(16);
    else if (y_bf16)
        scalenorm_fwd_kernel<true><<<grid, 256, 0, st>>>(x, ldx, g, y, ldy, rnorm, rows, D, eps);
    else
        scalenorm_fwd_kernel<false><<<grid, 256, 0, st>>>(x, ldx, g, y, ldy, rnorm, rows, D, eps);
#undef MEDP_SN_REG
    MEDP_LAUNCH_CHECK("medp_scalenorm_fwd");
    return 0;
}

static int scalenorm_bwd_launch(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* rnorm, const float* acc_src,
                                int ldacc, float* dx, int lddx, float* dg, float* workspace_rows, int rows, int D, void* stream) {
    MEDP_CHECK_ARG(dy && x && g && rnorm && dx, "scalenorm_bwd: null operand");
    MEDP_CHECK_ARG(rows > 0 && D % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldacc % 4 == 0, "scalenorm_bwd: alignment");
    MEDP_CHECK_ARG(!dg || workspace_rows, "scalenorm_bwd: dg needs a rows-float workspace");
    hipStream_t s = (hipStream_t)stream;
    const int nv = (D / 4 + 63) / 64;
    float* dgr = dg ? workspace_rows : nullptr;
    const int grid = (rows + 3) / 4;
    if (nv <= 5) scalenorm_bwd_reg_kernel<5><<<grid, 256, 0, s>>>(dy, lddy, x, ldx, g, rnorm, acc_src, ldacc, dx, lddx, dgr, rows, D);
    else if (nv <= 10) scalenorm_bwd_reg_kernel<10><<<grid, 256, 0, s>>>(dy, lddy, x, ldx, g, rnorm, acc_src, ldacc, dx, lddx, dgr, rows, D);
    else scalenorm_bwd_kernel<<<grid, 256, 0, s>>>(dy, lddy, x, ldx, g, rnorm, acc_src, ldacc, dx, lddx, dgr, rows, D);
    MEDP_LAUNCH_CHECK("medp_scalenorm_bwd");
    if (dg) {
        sum_all_kernel<<<1, 256, 0, s>>>(workspace_rows, dg, rows);
        MEDP_LAUNCH_CHECK("medp_scalenorm_bwd(dg)");
    }
    return 0;
}

extern "C" int medp_scalenorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* rnorm, float* dx,
                                  int lddx, int accumulate_dx, float* dg, float* workspace_rows, int rows, int D, void* stream) {
    return scalenorm_bwd_launch(dy, lddy, x, ldx, g, rnorm, accumulate_dx ? dx : nullptr, lddx, dx, lddx, dg, workspace_rows, rows, D, stream);
}

extern "C" int medp_scalenorm_bwd_add(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* rnorm, const float* add,
                                      int ldadd, float* dx, int lddx, float* dg, float* workspace_rows, int rows, int D, void* stream) {
    MEDP_CHECK_ARG(add, "scalenorm_bwd_add: null operand");
    return scalenorm_bwd_launch(dy, lddy, x, ldx, g, rnorm, add, ldadd, dx, lddx, dg, workspace_rows, rows, D, stream);
}
