// Weight-gradient GEMM for gfx950:  C[N,K] (fp32) = sum_m dY[m,n] * X[m,k]     (both operands row-major over m, bf16)
//
// The contraction index m is the ROW index of both operands, so an MFMA fragment (8 consecutive m of one column) is a
// transposed access.  Instead of materialising dY^T and X^T (two extra HBM passes + two launches per Linear backward) the
// tiles are staged row-major by LDS-DMA and the fragments are read with ds_read_b64_tr_b16, the hardware transposing read.
//   block = 4 waves (2 x 2), output tile 128 (n) x 128 (k), 32 rows of m per step, double-buffered LDS (2 x 16 KiB).
//   MFMA roles: A-operand := X columns (k_out), B-operand := dY columns (n)  ->  a lane holds 4 consecutive k_out of one n:
//   16-B stores into C[n][k..k+3].
//   split-m: blockIdx.y walks `splits` slices of the m range and writes its own fp32 slab; a second launch sums the slabs
//   (deterministic, no float atomics).  With 16 448 rows and a 256 x 768 output the unsplit grid would be 12 workgroups.
#include "common.h"
#include "medp_hip.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_tn[4] = {0, 0, 0, 0};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ bf16x4 lds_tr16(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}

struct TnParams {
    const bf16_t* dY;   // [M, N] ld = lddy
    const bf16_t* X;    // [M, K] ld = ldx
    float* C;           // [N, K] ldc  (or slabs [splits][N][ldc])
    int M, N, K, lddy, ldx, ldc, rows_per_split;
    long long slab_stride;
};

constexpr int TILE_BYTES = 32 * 256;   // 32 rows x 128 bf16

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][dY tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int il = lane & 15, kq = lane >> 4;
    const int tiles_k = (p.K + 127) / 128;
    const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
    const int m_begin = blockIdx.y * p.rows_per_split, m_end = min(p.M, m_begin + p.rows_per_split);
    const int nsteps = (m_end - m_begin + 31) >> 5;
    const bf16_t* zero = (const bf16_t*)g_zero16_tn;

    auto stage = [&](int buf, int step) {
        char* sy = smem + buf * 2 * TILE_BYTES;
        char* sx = sy + TILE_BYTES;
        const int mb = m_begin + step * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qd = i * 256 + tid;
            const int row = qd >> 4, c = qd & 15;
            const int m = mb + row;
            const bool mok = m < m_end;
            const bf16_t* s1 = (mok && n0 + c * 8 < p.N) ? p.dY + (size_t)m * p.lddy + n0 + c * 8 : zero;
            const bf16_t* s2 = (mok && k0 + c * 8 < p.K) ? p.X + (size_t)m * p.ldx + k0 + c * 8 : zero;
            glds16(s1, sy + (i * 256 + wave * 64) * 16);
            glds16(s2, sx + (i * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) stage(0, 0);
    MEDP_WAIT_LDS_DMA();
    __syncthreads();
    // transposing read: lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3; group kq covers rows 8kq..8kq+7
    const int tq = il >> 2, tp = il & 3;
    const int row_a = (kq * 8 + tq) * 256, row_b = row_a + 4 * 256;
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) stage(cur ^ 1, st + 1);
        const char* sy = smem + cur * 2 * TILE_BYTES;
        const char* sx = sy + TILE_BYTES;
        bf16x8 fy[4], fx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int colb = (wr * 64 + i * 16 + tp * 4) * 2;
            const bf16x4 a = lds_tr16(sy + row_a + colb), b = lds_tr16(sy + row_b + colb);
            fy[i] = (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int colb = (wc * 64 + j * 16 + tp * 4) * 2;
            const bf16x4 a = lds_tr16(sx + row_a + colb), b = lds_tr16(sx + row_b + colb);
            fx[j] = (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[j], fy[i], acc[i][j], 0, 0, 0);
        MEDP_WAIT_LDS_DMA();   // slab st+1 (issued at the top of this iteration) has landed
        __syncthreads();
    }
    float* C = p.C + (size_t)blockIdx.y * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wr * 64 + i * 16 + il;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wc * 64 + j * 16 + kq * 4;
            if (k >= p.K) continue;
            *(f32x4*)(C + (size_t)n * p.ldc + k) = acc[i][j];
        }
    }
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long slab_stride, int splits,
                                                        long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 a = *(const float4*)(slabs + i * 4);
        for (int s = 1; s < splits; ++s) {
            const float4 b = *(const float4*)(slabs + s * slab_stride + i * 4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *(float4*)(out + i * 4) = a;
    }
}

// Aim at ~768 workgroups (256 CUs x 2 resident x 1.5): the CXR-encoder weight gradients (M = 16 448, 108..144 output tiles) ran
// as 144 workgroups of 514 m-steps each before — 400+ us per launch, 37 % of the trainable encoder's step.
int choose_splits(int M, int tiles) {
    if (tiles >= 768 || M < 1024) return 1;
    int s = min(64, max(1, (768 + tiles - 1) / tiles));
    s = min(s, M / 512);
    return max(s, 1);
}

}  // namespace

extern "C" size_t medp_gemm_tn_workspace_bytes(int M, int N, int K) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    const int s = choose_splits(M, tiles);
    return s > 1 ? (size_t)s * N * K * sizeof(float) : 0;
}

extern "C" int medp_gemm_bf16_tn(const void* dY, const void* X, float* C, int M, int N, int K, int lddy, int ldx, float* workspace,
                                 void* stream) {
    MEDP_CHECK_ARG(dY && X && C && M > 0 && N > 0 && K > 0, "gemm_tn: bad argument");
    MEDP_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0, "gemm_tn: N, K, lddy, ldx must be multiples of 8");
    MEDP_CHECK_ARG(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm_tn: 16-byte alignment");
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    const int splits = choose_splits(M, tiles);
    MEDP_CHECK_ARG(splits == 1 || workspace, "gemm_tn: workspace required (medp_gemm_tn_workspace_bytes)");
    int rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    TnParams p{(const bf16_t*)dY, (const bf16_t*)X, splits > 1 ? workspace : C, M, N, K, lddy, ldx, K, rps, (long long)N * K};
    hipStream_t s = (hipStream_t)stream;
    const int nsplit = (M + rps - 1) / rps;
    gemm_bf16_tn_kernel<<<dim3(tiles, nsplit), 256, 4 * TILE_BYTES, s>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_tn");
    if (splits > 1) {
        const long long n4 = (long long)N * K / 4;
        sum_slabs_kernel<<<(int)min((long long)2048, (n4 + 255) / 256), 256, 0, s>>>(workspace, C, (long long)N * K, nsplit, n4);
        MEDP_LAUNCH_CHECK("medp_gemm_bf16_tn(reduce)");
    }
    return 0;
}
