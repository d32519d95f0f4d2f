// Internal interface between the GEMM dispatcher (gemm_bf16.hip) and the 256x256x64 kernels (gemm_bf16_v6.hip, gemm_bf16_v7.hip);
// not part of the C ABI.
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
    // optional in-kernel launch clock (medp_gemm_profile_enable(2)): 4 x u64 {t_first_wg_in, t_last_wg_out, arrivals, departures}
    unsigned long long* prof;
    int prof_flags;      // 1: do not stamp the arrival, 2: do not stamp the departure (a GEMM issued as two launches shares one clock)
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
