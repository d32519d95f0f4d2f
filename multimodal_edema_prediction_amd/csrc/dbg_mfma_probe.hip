// Debug probe (NOT part of the C ABI in include/medp_hip.h; tools/probe_mfma_dma.py): what ONE wave per SIMD pays for LDS-DMA pieces and
// fragment reads issued inside its own MFMA stream.  4 waves per workgroup, one workgroup per CU; every iteration is 64 independent
// v_mfma_f32_16x16x32_bf16 (a 128 x 128 wave tile's K-step) with `pieces` global_load_lds (16 B per lane) and `reads` ds_read_b128
// spread between them.  Output: wall-clock ticks (100 MHz) of the loop per workgroup.
#include <stdlib.h>

#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int PIECES, int READS>
__global__ __launch_bounds__(256) void mfma_dma_probe_kernel(const char* __restrict__ src, size_t src_bytes, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 128 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[i] = (bf16x8_t){(short)(lane + i), 1, 2, 3, 4, 5, 6, 7};
        fb[i] = (bf16x8_t){(short)(lane * 3 + i), 7, 6, 5, 4, 3, 2, 1};
    }
    const char* gp = src + ((size_t)blockIdx.x * 65536 + wave * 16384 + lane * 16) % (src_bytes - (1 << 20));
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const char* g = gp + (size_t)(it & 15) * 4096;
        char* l = smem + ((it & 1) * 65536) + wave * 16384;
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[m & 7], fa[m >> 3], acc[m], 0, 0, 0);
            if (PIECES > 0 && (m % (64 / (PIECES > 0 ? PIECES : 1))) == 0 && m / (64 / (PIECES > 0 ? PIECES : 1)) < PIECES)
                glds16(g + (m / (64 / (PIECES > 0 ? PIECES : 1))) * 1024, l + (m / (64 / (PIECES > 0 ? PIECES : 1))) * 1024);
            if (READS > 0 && (m % (64 / (READS > 0 ? READS : 1))) == 1 && m / (64 / (READS > 0 ? READS : 1)) < READS) {
                const int r = m / (64 / (READS > 0 ? READS : 1));
                const bf16x8_t v = *(const bf16x8_t*)(smem + ((it + 1) & 1) * 65536 + wave * 16384 + r * 1024 + lane * 16);
                if (r < 8) fa[r & 7] = v; else fb[r & 7] = v;            // consumed by the NEXT iteration's MFMAs
            }
        }
        if (PIECES > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int P, int R>
int run(const char* src, size_t bytes, int iters, unsigned long long* out, float* sink, hipStream_t s, int grid) {
    (void)hipFuncSetAttribute((const void*)mfma_dma_probe_kernel<P, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    mfma_dma_probe_kernel<P, R><<<grid, 256, 131072, s>>>(src, bytes, iters, out, sink);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int medp_dbg_mfma_dma_probe(int pieces, int reads, const void* src, size_t src_bytes, int iters, unsigned long long* out, float* sink,
                                       void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const char* p = (const char*)src;
    static const int grid = [] { const char* e = getenv("MEDP_PROBE_GRID"); return e ? atoi(e) : 256; }();
#define CASE(P, R) if (pieces == P && reads == R) return run<P, R>(p, src_bytes, iters, out, sink, s, grid)
    CASE(0, 0); CASE(8, 0); CASE(16, 0); CASE(0, 16); CASE(8, 16); CASE(16, 16); CASE(4, 16); CASE(16, 8);
#undef CASE
    return -1;
}
