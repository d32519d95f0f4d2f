// Internal interface between the GEMM dispatcher (gemm_bf16.hip) and the 256x256x64 kernels (gemm_bf16_v6.hip, gemm_bf16_v7.hip);
// not part of the C ABI.
#pragma once
// LayerNorm folded into the block GEMMs of the CXR encoder (vit.hip): LN(x) W^T + b = rstd (x (W g)^T) - rstd mean colsum(W g) + (b + W beta).
//   PRODUCER (proj / fc2: fp32 residual output x): also writes c2 = bf16(x) and, per row and 256-column tile, the (sum, sum of squares)
//     of the fp32 values into stats_out [rows padded to 256][tiles_n][2].
//   CONSUMER (qkv / fc1: A = bf16(x), W = bf16(W g)): C = rstd[m] acc - rstd[m] mean[m] colsum[n] + bias[n] (then GELU), mean / rstd from
//     stats_in [rows padded to 256][stats_tiles][2] over ln_dim columns.
struct MedpGemmFold {
    void* c2 = nullptr;                 // producer: bf16 copy of C
    int ldc2 = 0;
    float* stats_out = nullptr;
    const float* stats_in = nullptr;    // consumer
    int stats_tiles = 0;
    const float* colsum = nullptr;
    float ln_eps = 0.f;
    int ln_dim = 0;
};
struct MedpGemmArgs {
    const void* A;
    const void* W;
    void* C;
    int M, N, K, lda, ldw, ldc;
    const float* bias;
    const float* scale;
    const float* residual;
    int ldr, act, out_bf16;
    // optional in-kernel launch clock (medp_gemm_profile_enable(2)): 4 x u64 {t_first_wg_in, t_last_wg_out, arrivals, departures}
    unsigned long long* prof;
    int prof_flags;      // 1: do not stamp the arrival, 2: do not stamp the departure (a GEMM issued as two launches shares one clock)
    MedpGemmFold fold;   // all null: a plain GEMM
};
// kernel-side halves of the launch clock: the first workgroup to arrive stamps the 100-MHz wall clock, the last to leave
// stamps it again (two agent-scope atomics per workgroup; works inside a replayed hipGraph, where HIP events cannot be read)
#define MEDP_PROF_ENTER(prof, flags)                                                                                       \
    do {                                                                                                                   \
        if ((prof) && !((flags) & 1) && threadIdx.x == 0) {                                                                                \
            const unsigned long long n__ = __hip_atomic_fetch_add((prof) + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            if (n__ % gridDim.x == 0) __hip_atomic_store((prof), wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   \
        }                                                                                                                  \
    } while (0)
#define MEDP_PROF_LEAVE(prof, flags)                                                                                       \
    do {                                                                                                                   \
        if ((prof) && !((flags) & 2)) {                                                                                    \
            __syncthreads();                                                                                               \
            if (threadIdx.x == 0) {                                                                                        \
                const unsigned long long n__ = __hip_atomic_fetch_add((prof) + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                if (n__ % gridDim.x == gridDim.x - 1)                                                                      \
                    __hip_atomic_store((prof) + 1, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
int medp_gemm_v6_launch(const MedpGemmArgs& a, int tag, void* stream);
// rows [m_begin, a.M) (at most 128 of them) as a skinny launch of their own (gemm_ragged_rows.hip): same bits as the tile kernels
int medp_gemm_ragged_rows_launch(const MedpGemmArgs& a, int m_begin, void* stream);
// v7 = v6 made persistent for grids of more than 256 tiles (gemm_bf16_v7.hip); launch returns -1 when it has no private
// ticket block left, and the caller launches v6 instead
bool medp_gemm_v7_eligible(const MedpGemmArgs& a);
// a private block of per-XCD ticket counters for ONE launch on `stream` (gemm_bf16_v7.hip; nullptr: none left)
unsigned* medp_gemm_ticket_block(void* stream);
int medp_gemm_v7_launch(const MedpGemmArgs& a, int tag, void* stream);
// the CXR-encoder block GEMMs with the LayerNorm fold (tag 1: they carry the launch clock like medp_gemm_bf16_nt_tagged(1, ...)); only
// shapes that dispatch to the 256-tile kernels support it: ask medp_gemm_fold_eligible first
bool medp_gemm_fold_eligible(int M, int N, int K);
int medp_gemm_bf16_nt_fold(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc, const float* bias,
                           const float* scale, const float* residual, int ldr, int act, int out_bf16, const MedpGemmFold& fold, void* stream);
