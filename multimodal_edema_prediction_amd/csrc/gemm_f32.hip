// fp32 GEMMs for the library's FP32 KERNEL MODE (SURVEY.md §7 "always keep an fp32 kernel mode for tight checks", §8(d): logits
// <= 1e-4, loss <= 1e-5 against the CPU restatement).  Same operator contract as medp_gemm_bf16_nt / medp_gemm_bf16_tn, fp32
// operands, fp32 FMA accumulation on the vector ALUs — a parity instrument, not a throughput path: 64 x 64 x 16 LDS tiles,
// 4 x 4 outputs per thread, no matrix cores (v_mfma_f32_*_f32 would round the same way; the VALU form keeps one code path for
// both operand layouts).
//   NT:  C[m][n] = epi( sum_k A[m][k] W[n][k] )            every nn.Linear forward and its dX
//   TN:  C[n][k] = sum_m dY[m][n] X[m][k]                  every weight gradient
// Products are summed in ascending k within a thread (one chain per output), deterministic.
#include "common.h"
#include "medp_hip.h"

namespace {

struct F32GemmParams {
    const float *A, *B;     // operand element (i, r) = A[i * sai + r * sar] ;  (j, r) = B[j * sbj + r * sbr]
    long long sai, sar, sbj, sbr;
    float* C;
    int I, J, R, ldc;
    const float *bias, *scale, *residual;
    int ldr, act;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const F32GemmParams p) {
    __shared__ float sa[16][65], sb[16][65];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;          // 16 x 16 threads, 4 x 4 outputs each
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int r0 = 0; r0 < p.R; r0 += 16) {
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            // pick the index that is contiguous in memory as the fast one of the load
            int ii, rr;
            if (p.sar == 1) { rr = idx & 15; ii = idx >> 4; } else { ii = idx & 63; rr = idx >> 6; }
            const int gi = i0 + ii, gr = r0 + rr;
            sa[rr][ii] = (gi < p.I && gr < p.R) ? p.A[(long long)gi * p.sai + (long long)gr * p.sar] : 0.f;
            int jj, r2;
            if (p.sbr == 1) { r2 = idx & 15; jj = idx >> 4; } else { jj = idx & 63; r2 = idx >> 6; }
            const int gj = j0 + jj, gr2 = r0 + r2;
            sb[r2][jj] = (gj < p.J && gr2 < p.R) ? p.B[(long long)gj * p.sbj + (long long)gr2 * p.sbr] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = sa[r][ti * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = sb[r][tj * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ti * 4 + a;
        if (i >= p.I) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tj * 4 + b;
            if (j >= p.J) continue;
            float v = acc[a][b];
            if (p.bias) v += p.bias[j];
            if (p.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));      // exact-form GELU, libm erf
            if (p.scale) v *= p.scale[j];
            if (p.residual) v += p.residual[(size_t)i * p.ldr + j];
            p.C[(size_t)i * p.ldc + j] = v;
        }
    }
}

}  // namespace

extern "C" int medp_gemm_f32_nt(const float* A, const float* W, float* C, int M, int N, int K, int lda, int ldw, int ldc, const float* bias,
                                const float* scale, const float* residual, int ldr, int act, void* stream) {
    MEDP_CHECK_ARG(A && W && C, "gemm_f32_nt: null operand");
    MEDP_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "gemm_f32_nt: bad shape M=%d N=%d K=%d", M, N, K);
    MEDP_CHECK_ARG(act == 0 || act == 1, "gemm_f32_nt: act must be 0 (none) or 1 (gelu)");
    F32GemmParams p{A, W, lda, 1, ldw, 1, C, M, N, K, ldc, bias, scale, residual, ldr, act};
    gemm_f32_kernel<<<dim3((N + 63) / 64, (M + 63) / 64), 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_f32_nt");
    return 0;
}

extern "C" int medp_gemm_f32_tn(const float* dY, const float* X, float* C, int M, int N, int K, int lddy, int ldx, void* stream) {
    MEDP_CHECK_ARG(dY && X && C, "gemm_f32_tn: null operand");
    MEDP_CHECK_ARG(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K, "gemm_f32_tn: bad shape M=%d N=%d K=%d", M, N, K);
    F32GemmParams p{dY, X, 1, lddy, 1, ldx, C, N, K, M, K, nullptr, nullptr, nullptr, 0, 0};
    gemm_f32_kernel<<<dim3((K + 63) / 64, (N + 63) / 64), 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_f32_tn");
    return 0;
}
