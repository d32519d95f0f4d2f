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
// v7 = v6 made persistent for grids of more than 256 tiles (gemm_bf16_v7.hip); launch returns -1 when it has no private
// ticket block left, and the caller launches v6 instead
bool medp_gemm_v7_eligible(const MedpGemmArgs& a);
int medp_gemm_v7_launch(const MedpGemmArgs& a, int tag, void* stream);
