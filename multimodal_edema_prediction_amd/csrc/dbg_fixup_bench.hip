// Debug hook (NOT part of the C ABI in include/medp_hip.h; tools/bench_splitk_fixup.py): what a stream-K / split-K FIX-UP of the
// 256 x 256 block GEMMs would move, as two plain kernels, so that its cost can be MEASURED next to the time tile balancing could
// save (profiles/r03_gemm_balance.txt).  A workgroup that ends its K range inside a tile leaves a 256 x 256 fp32 partial (256 KiB)
// in a workspace; the tile's owner adds the partials of the other workgroups to its accumulators before the epilogue.
//   kernel 1: `writers` workgroups (512 threads, the GEMM's geometry) each store one 256-KiB slab from registers (float4 per lane);
//   kernel 2: `owners` workgroups each read `per_owner` slabs in a fixed order, sum them and write a bf16 tile (the epilogue's bytes).
// An in-launch fix-up saves at most the boundary between the two (~1.5 us) and adds its release / acquire.
#include "common.h"

namespace {
constexpr int SLAB_F4 = 256 * 256 / 4;      // float4 per slab

__global__ __launch_bounds__(512) void fixup_write_kernel(float* ws, float seed) {
    f32x4* slab = (f32x4*)ws + (size_t)blockIdx.x * SLAB_F4;
    const f32x4 v = (f32x4){seed + threadIdx.x, seed, seed * 2.f, 1.f};
#pragma unroll 8
    for (int i = threadIdx.x; i < SLAB_F4; i += 512) slab[i] = v;
}

__global__ __launch_bounds__(512) void fixup_reduce_kernel(const float* ws, bf16_t* out, int per_owner, int writers) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    u32x2* tile = (u32x2*)out + (size_t)blockIdx.x * SLAB_F4;
    for (int i = threadIdx.x; i < SLAB_F4; i += 512) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < per_owner; ++s) {
            const int slab = (blockIdx.x * per_owner + s) % writers;
            acc += ((const f32x4*)ws + (size_t)slab * SLAB_F4)[i];
        }
        tile[i] = (u32x2){pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
    }
}
}  // namespace

extern "C" int medp_dbg_splitk_fixup(float* ws, void* out, int writers, int owners, int per_owner, void* stream) {
    MEDP_CHECK_ARG(ws && out && writers > 0 && owners > 0 && per_owner > 0, "dbg_splitk_fixup: bad argument");
    fixup_write_kernel<<<writers, 512, 0, (hipStream_t)stream>>>(ws, 1.0f);
    fixup_reduce_kernel<<<owners, 512, 0, (hipStream_t)stream>>>(ws, (bf16_t*)out, per_owner, writers);
    MEDP_LAUNCH_CHECK("medp_dbg_splitk_fixup");
    return 0;
}
