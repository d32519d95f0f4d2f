// v5: 256 x 256 x 32 tiles, 4 waves (2 x 2), 128 x 128 PER WAVE.
//
// Why: in v3 a wave multiplies 128 x 64 per K-step and reads 12 fragments (12 KiB) from LDS for 32 MFMAs; two workgroups
// per CU make that 96 KiB of ds_read per 1024 matrix-pipe clocks, i.e. ~75 % of the CU's 128 B/clk LDS port before the
// LDS-DMA fills and any bank conflict are counted — the SQ counters show the MFMA pipe ~48 % busy in steady state.
// A 128 x 128 wave tile needs 16 fragments (16 KiB) per 64 MFMAs: 64 KiB per 1024 clocks for the CU's four waves, half the
// port.  The price is 256 accumulator registers per lane, so the kernel runs ONE wave per SIMD (512-register budget:
// accumulators in AGPRs) and relies on its own software pipeline (LDS-DMA three K-tiles ahead, fragments of the next
// K-tile read into a second register set during the MFMAs, counted vmcnt, one s_barrier per K-tile) instead of a
// co-resident workgroup to hide latency.
// The MFMAs are issued through inline asm with "+a" operands: with the builtin the register allocator shuffles the 64
// accumulator tuples between AGPRs and VGPRs (hundreds of v_accvgpr_mov per K-step, or spills); pinned, the K-step is
// exactly 64 MFMA + 16 ds_read_b128 + 8 LDS-DMA + address arithmetic.
//   LDS: 3 slots x (A 256 rows + W 256 rows) x 64 B = 96 KiB.  64-B rows, 16-B chunk c stored at c ^ (((row>>2)&1)<<1).
//   Epilogue: per-wave [64][68] fp32 patch in LDS, four 64 x 64 passes -> whole-row coalesced stores (as v3).
#include <stdlib.h>

#include "common.h"
#include "gemm_v4.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_v5[4] = {0, 0, 0, 0};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM = 256, BN = 256, A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 32 KiB / slot

template <int TAG>
__global__ __launch_bounds__(256, 1) void gemm_bf16_nt_v5_kernel(const MedpGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)g_zero16_v5;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // band of 8 row-tiles x super-column of 4 column-tiles per XCD run (see v3): the 32 tiles an XCD runs at once share
    // 8 A panels and 4 W panels
    constexpr int MB = 8, SN = 4;
    const int band = wg / (MB * tiles_n), rb = wg % (MB * tiles_n);
    const int mb = min(MB, tiles_m - band * MB);
    const int sc = rb / (mb * SN), r2 = rb % (mb * SN);
    const int sn = min(SN, tiles_n - sc * SN);
    const int m0 = (band * MB + r2 / sn) * BM, n0 = (sc * SN + r2 % sn) * BN;
    const int nkt = (p.K + 31) >> 5;

    // one 4-KiB LDS-DMA piece (1 KiB per wave) per call: pieces 0..3 = A rows, 4..7 = W rows of K-tile kt
    auto stage_piece = [&](int slot, int kt, int piece) {
        char* sa = smem + slot * STAGE;
        const int k0 = kt << 5;
        const int qd = (piece & 3) * 256 + tid;
        const int row = qd >> 2, c = (qd & 3) ^ (((row >> 2) & 1) << 1);
        const int gk = k0 + c * 8;
        if (piece < 4) {
            const int gr = m0 + row;
            const bf16_t* src = (gr < p.M && gk < p.K) ? A + (size_t)gr * p.lda + gk : zero;
            glds16(src, sa + (piece * 256 + wave * 64) * 16);
        } else {
            const int gr = n0 + row;
            const bf16_t* src = (gr < p.N && gk < p.K) ? W + (size_t)gr * p.ldw + gk : zero;
            glds16(src, sa + A_BYTES + ((piece - 4) * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int pc = 0; pc < 8; ++pc) stage_piece(0, 0, pc);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) stage_piece(1, 1, pc);   // past-K tiles are zero-sourced: no branch, fixed vmcnt counts
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) stage_piece(2, 2, pc);

    const int coff = (kq ^ (((fr >> 2) & 1) << 1)) << 4;
    const int a_off = (wm * 128 + fr) * 64 + coff, b_off = (wn * 128 + fr) * 64 + coff;

    // Register double buffer for the fragments: while K-tile kt is multiplied out of `cur`, the 16 ds_read_b128 of K-tile
    // kt+1 land in `nxt` — with one wave per SIMD nobody else would cover the LDS latency.
    //   top of step kt:  vmcnt(8)   -> K-tile kt+1 is in LDS (only tile kt+2's eight pieces may still be in flight)
    //                    lgkmcnt(0) -> this wave holds tile kt's fragments, i.e. it is done reading slot kt % 3
    //                    s_barrier  -> both hold for all four waves: slot (kt+1)%3 readable, slot kt%3 reusable
    //   body:            ds_read tile kt+1 -> nxt ; LDS-DMA tile kt+3 -> slot kt%3 ; 64 MFMA on cur
    bf16x8 fa[2][8], fb[2][8];
    auto read_frags = [&](int slot, bf16x8* xa, bf16x8* wb) {
        const char* sa = smem + slot * STAGE + a_off;
        const char* sb = smem + slot * STAGE + A_BYTES + b_off;
#pragma unroll
        for (int j = 0; j < 8; ++j) wb[j] = *(const bf16x8*)(sb + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = *(const bf16x8*)(sa + i * 16 * 64);
    };
    auto step = [&](int kt, int slot, const bf16x8* xa, const bf16x8* wb, bf16x8* nxa, bf16x8* nwb) {
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): a real S_WAITCNT, so the compiler's own counter model sees it
        __builtin_amdgcn_s_barrier();
        read_frags(slot == 2 ? 0 : slot + 1, nxa, nwb);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wb[j]), "v"(xa[i]));
            stage_piece(slot, kt + 3, i);
        }
    };

    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, fa[0], fb[0]);
    int slot = 0;
    for (int kt = 0; kt < nkt; kt += 2) {
        step(kt, slot, fa[0], fb[0], fa[1], fb[1]);
        slot = slot == 2 ? 0 : slot + 1;
        if (kt + 1 < nkt) {
            step(kt + 1, slot, fa[1], fb[1], fa[0], fb[0]);
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    // ---- epilogue through LDS (whole-row coalesced global accesses) -------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // trailing zero-sourced prefetches / fragment reads
    // the MFMAs are inline asm (accumulators pinned in AGPRs, D == C in place), so the compiler's hazard recogniser does not
    // know an XDL write precedes the accumulator reads below: pay the worst-case wait states by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    float* wl = (float*)(smem + wave * (64 * 68 * 4));
    const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int n = n0 + wn * 128 + ch * 64 + ec;
        f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f}, scale4 = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (n < p.N) {
            if (p.bias) bias4 = *(const f32x4*)(p.bias + n);
            if (p.scale) scale4 = *(const f32x4*)(p.scale + n);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(wl + (i4 * 16 + fr) * 68 + j * 16 + kq * 4) = acc[half * 4 + i4][ch * 4 + j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rr = it * 4 + er;
                const int m = m0 + wm * 128 + half * 64 + rr;
                f32x4 v = *(const f32x4*)(wl + rr * 68 + ec);
                if (m < p.M && n < p.N) {
                    v += bias4;
                    if (p.act == 1) {
                    v = gelu_erf4(v);
                    }
                    v *= scale4;
                    if (p.residual) v += *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
                    if (p.out_bf16) {
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
                    } else {
                        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

template <int TAG>
int launch_v5(const MedpGemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    constexpr int LDS = 3 * STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bf16_nt_v5_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    gemm_bf16_nt_v5_kernel<TAG><<<tiles, 256, LDS, stream>>>(a);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v5)");
    return 0;
}

}  // namespace

int medp_gemm_v5_launch(const MedpGemmArgs& a, int tag, void* stream) {
    return tag == 1 ? launch_v5<1>(a, (hipStream_t)stream) : launch_v5<0>(a, (hipStream_t)stream);
}
