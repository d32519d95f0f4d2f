// The ragged last rows of a 256-row-tiled GEMM as their own small launch.
//
// M = 64 * 257 token rows leave 64 rows past the last full 256-row tile.  In the 256 x 256 kernels those rows cost a whole extra
// tile per column (fc1: 780 tiles instead of 768 = a FOURTH round of workgroups on 256 CUs for 12 tiles that are three quarters
// empty: 104 us against 90 us for M = 16384, tools/bench_gemm_instep.py).  Here they are a skinny GEMM, C[m_begin:M, :] only:
// one workgroup per 16 output columns, wave w owns rows m_begin + 16 w .. + 15 (<= 8 waves: <= 128 ragged rows), operand
// fragments straight from global memory into registers (the 64 x K row band is read by every workgroup: L2 hits; no LDS, no
// barrier), the same MFMA (16 x 16 x 32, W as the first operand), the same K order (ascending 32-deep steps from a zero
// accumulator) and the same epilogue arithmetic as gemm_bf16_v6.hip / gemm_bf16_v7.hip — so a row computes to the same bits
// whichever kernel owns it (tests/test_gpu_gemm_persistent.py).
#include "common.h"
#include "gemm_variants.h"

namespace {

constexpr int KU = 8;      // k-steps in flight per wave: 8 x (16 B of A + 16 B of W) per lane

__global__ __launch_bounds__(512) void gemm_ragged_rows_kernel(const MedpGemmArgs p, const int m_begin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, kq = lane >> 4;
    MEDP_PROF_ENTER(p.prof, p.prof_flags);
    const int m = m_begin + wave * 16 + fr;                 // this lane's A row (= its output row)
    const int nw = blockIdx.x * 16 + fr;                    // this lane's W row (operand fragment)
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const bool a_ok = m < p.M, w_ok = nw < p.N;
    const bf16x8* ap = (const bf16x8*)(A + (size_t)(a_ok ? m : m_begin) * p.lda + kq * 8);
    const bf16x8* wp = (const bf16x8*)(W + (size_t)(w_ok ? nw : 0) * p.ldw + kq * 8);
    const bf16x8 zero8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = p.K >> 5;                               // host: K % 32 == 0
    for (int k0 = 0; k0 < nks; k0 += KU) {
        bf16x8 fa[KU], fw[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const bool kin = k0 + u < nks;
            fa[u] = (kin && a_ok) ? ap[(k0 + u) * 4] : zero8;        // 32 elements = 4 x bf16x8 per k-step
            fw[u] = (kin && w_ok) ? wp[(k0 + u) * 4] : zero8;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[u], fa[u], acc, 0, 0, 0);
    }
    // the lane holds columns n .. n + 3 of row m (accumulator layout of the tile kernels); epilogue in their order
    const int n = blockIdx.x * 16 + kq * 4;
    if (a_ok && n < p.N) {                                  // host: N % 4 == 0
        const f32x4 bias4 = p.bias ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 scale4 = p.scale ? *(const f32x4*)(p.scale + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
        f32x4 v = acc + bias4;                              // (unconditional, as in the tile kernels: -0 + 0 = +0)
        if (p.act == 1) v = p.out_bf16 ? gelu_bf16_4(v) : gelu_erf4(v);
        v *= scale4;
        if (p.residual) v += *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
        if (p.out_bf16) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            *(u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        } else {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
        }
    }
    MEDP_PROF_LEAVE(p.prof, p.prof_flags);
}

}  // namespace

// rows [m_begin, a.M) of the GEMM `a`; a.M - m_begin <= 128, a.K % 32 == 0
int medp_gemm_ragged_rows_launch(const MedpGemmArgs& a, int m_begin, void* stream) {
    const int rows = a.M - m_begin;
    if (rows <= 0 || rows > 128 || a.K % 32 != 0) {
        medp_set_error("gemm(ragged rows): %d rows, K %d not supported", rows, a.K);
        return -1;
    }
    const int waves = (rows + 15) / 16;
    gemm_ragged_rows_kernel<<<(a.N + 15) / 16, waves * 64, 0, (hipStream_t)stream>>>(a, m_begin);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(ragged rows)");
    return 0;
}
