// Internal interface of the persistent GEMM variant (gemm_bf16_v4.hip); not part of the C ABI.
#pragma once
struct MedpGemmArgs {
    const void* A;
    const void* W;
    void* C;
    int M, N, K, lda, ldw, ldc;
    const float* bias;
    const float* scale;
    const float* residual;
    int ldr, act, out_bf16;
};
int medp_gemm_v4_launch(const MedpGemmArgs& a, int tag, void* stream);
int medp_gemm_v5_launch(const MedpGemmArgs& a, int tag, void* stream);
int medp_gemm_v6_launch(const MedpGemmArgs& a, int tag, void* stream);
