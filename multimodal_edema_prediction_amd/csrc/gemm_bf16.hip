// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] · W[N,K]^T )
//
// Both operands are K-contiguous (torch `nn.Linear` layout), which is the natural operand layout
// of v_mfma_f32_16x16x32_bf16: a lane's fragment is 8 consecutive k of one row (16 B).
//
// Structure (one 256-thread workgroup = 4 waves in 2x2, tile BM x BN x 64):
//   * global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, no VGPR round trip),
//     double-buffered: the loads of K-tile t+1 are in flight while tile t is multiplied.
//   * LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with (row & 7) so that the
//     ds_read_b128 fragment reads of 16 different rows are bank-conflict free.  LDS-DMA writes LDS
//     linearly (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE address and
//     again on the read (both-sides-or-neither).
//   * MFMA operand roles are swapped (A-operand := W rows, B-operand := activation rows) so each
//     lane ends up with 4 consecutive output columns of one output row -> 16-B vector epilogue.
//   * Ragged M / N / K: out-of-range 16-B chunks are sourced from a zero buffer (the LDS-DMA source
//     address is per lane), so no padding convention is imposed on callers (K % 8 == 0 only).
//   * Fused epilogue: + bias[n] -> GELU(erf) -> * scale[n] (LayerScale) -> + residual[m,n] -> fp32 or bf16.
//   * blockIdx -> tile map is XCD-aware (bijective): the 8 XCDs each get a contiguous run of tiles that
//     walks N fastest, so one XCD's L2 sees one A row-panel and the whole W.
#include <stdlib.h>

#include <vector>

#include <algorithm>

#include "common.h"
#include "gemm_variants.h"
#include "medp_hip.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0, 0, 0, 0};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    void* C;
    int M, N, K, lda, ldw, ldc;
    const float* bias;
    const float* scale;
    const float* residual;
    int ldr;
    int act;
    int out_bf16;
    float* partial;     // split-K (small grids): blockIdx.y = K slice, raw fp32 partial sums go to partial[slice][M][N]
    int nsplit;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// TAG only gives the CXR-encoder launches their own kernel symbol, so a kernel trace separates the ViT GEMMs (the step's
// dominant kernel, 4 shapes x 12 layers) from the DuETT / fusion-head launches of the same code.
template <int BM, int BN, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / 32, TN = BN / 32;   // 16x16 tiles per wave along m / n (wave tile = BM/2 x BN/2)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- XCD-aware, bijective block -> tile map -----------------------------------------------
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

    // split-K: slice blockIdx.y owns K-tiles [kt_begin, kt_begin + nkt)
    const int nkt_all = (p.K + 63) >> 6;
    const int per_split = (nkt_all + p.nsplit - 1) / p.nsplit;
    const int kt_begin = blockIdx.y * per_split;
    const int nkt = max(0, min(nkt_all, kt_begin + per_split) - kt_begin);
    const bf16_t* zero = (const bf16_t*)g_zero16;

    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const int k0 = (kt_begin + kt) << 6;
#pragma unroll
        for (int i = 0; i < BM * 8 / 256; ++i) {
            const int qd = i * 256 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const int gr = m0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.M && gk < p.K) ? p.A + (size_t)gr * p.lda + gk : zero;
            glds16(src, sa + (i * 256 + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < BN * 8 / 256; ++i) {
            const int qd = i * 256 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const int gr = n0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.N && gk < p.K) ? p.W + (size_t)gr * p.ldw + gk : zero;
            glds16(src, sb + (i * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nkt > 0) stage(0, 0);
    MEDP_WAIT_LDS_DMA();
    __syncthreads();   // tile 0 landed (explicit wait) before any wave reads it

    const int a_row0 = wr * (BM / 2) + fr, b_row0 = wc * (BN / 2) + fr;
    const int sw = fr & 7;   // (row & 7): every row base is a multiple of 16

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = ((s * 4 + kq) ^ sw) << 4;
            bf16x8 xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const bf16x8*)(sa + (a_row0 + i * 16) * 128 + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = *(const bf16x8*)(sb + (b_row0 + j * 16) * 128 + coff);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
        MEDP_WAIT_LDS_DMA();   // tile kt+1 (issued at the top of this iteration) has landed
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3],  m = tile row (lane&15),  n = 4*(lane>>4) ----------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wr * (BM / 2) + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 16 + kq * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.partial) {          // split-K: raw partial sums; bias / activation / residual happen in splitk_finish_kernel
                *(f32x4*)(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n) = v;
                continue;
            }
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
                v += b;
            }
            if (p.act == 1) {
                    v = gelu_erf4(v);
            }
            if (p.scale) {
                const f32x4 sc = *(const f32x4*)(p.scale + n);
                v *= sc;
            }
            if (p.residual) {
                const f32x4 rs = *(const f32x4*)(p.residual + (size_t)m * p.ldr + n);
                v += rs;
            }
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
}


// ------------------------------------------------------------------------------------------------------------------
// v2: 256 x 128 x 64 tiles, 8 waves (4 x 2, 64 x 64 per wave), THREE LDS slots filled by LDS-DMA two K-tiles ahead.
// One raw s_barrier per K-tile and a COUNTED s_waitcnt vmcnt (never 0 in the loop), so the loads of tiles t+1 / t+2 stay
// in flight across the barrier while tile t is multiplied (cdna guide §5 "Pipelining across barriers").
//   iteration t:  vmcnt(6)  -> tile t landed (this wave's 6 younger loads = tile t+1 may still fly)
//                 s_barrier -> every wave's tile-t loads landed AND every wave finished reading slot (t-1)%3
//                 issue tile t+2 into slot (t+2)%3 == (t-1)%3
//                 ds_read + 32 MFMA on slot t%3
template <int TAG>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 128;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 48 KiB, x3 = 144 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int nkt = (p.K + 63) >> 6;
    const bf16_t* zero = (const bf16_t*)g_zero16;

    // per-thread source rows are loop-invariant: precompute row base pointers (nullptr -> zero chunk)
    auto stage = [&](int slot, int kt) {
        char* sa = smem + slot * STAGE;
        char* sb = sa + A_BYTES;
        const int k0 = kt << 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = i * 512 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const int gr = m0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.M && gk < p.K) ? p.A + (size_t)gr * p.lda + gk : zero;
            glds16(src, sa + (i * 512 + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qd = i * 512 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const int gr = n0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.N && gk < p.K) ? p.W + (size_t)gr * p.ldw + gk : zero;
            glds16(src, sb + (i * 512 + wave * 64) * 16);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    if (nkt > 1) stage(1, 1);

    const int a_row0 = wr * 64 + fr, b_row0 = wc * 64 + fr;
    const int sw = fr & 7;
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) stage(slot == 0 ? 2 : slot - 1, kt + 2);      // (kt+2)%3 == (slot+2)%3
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = ((s * 4 + kq) ^ sw) << 4;
            bf16x8 xa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *(const bf16x8*)(sa + (a_row0 + i * 16) * 128 + coff);
#pragma unroll
            for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8*)(sb + (b_row0 + j * 16) * 128 + coff);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        slot = slot == 2 ? 0 : slot + 1;
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + kq * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (p.act == 1) {
                    v = gelu_erf4(v);
            }
            if (p.scale) v *= *(const f32x4*)(p.scale + n);
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
}

template <int TAG>
int launch_v2(const GemmParams& p, hipStream_t stream) {
    const int tiles = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    constexpr int LDS = 3 * (256 + 128) * 128;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)gemm_bf16_nt_v2_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    gemm_bf16_nt_v2_kernel<TAG><<<tiles, 512, LDS, stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v2)");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// v3: the LDS-traffic fix.  v1/v2 give each wave a 64x64 output tile: 32 flop per LDS byte read, and the LDS (fragment
// reads + LDS-DMA fills) is as busy as the matrix pipe.  v3 gives each wave 128 x 64 (8 x 4 MFMA tiles, 128 accumulator
// VGPRs): 43 flop per LDS byte, 12 ds_read_b128 per 32 MFMAs.
//   block 256 x 128, 4 waves (2 x 2), BK = 32 (64-B LDS rows), THREE 24-KiB slots (72 KiB -> 2 blocks per CU),
//   LDS-DMA two K-tiles ahead, counted vmcnt + one raw s_barrier per K-tile (same protocol as v2).
//   64-B rows: chunk' = chunk ^ (((row >> 2) & 1) << 1) makes every ds_read_b128 lane group hit 16 distinct 16-B slots.
template <int TAG>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_v3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 128;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 24 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // L2-aware linearisation: an XCD's contiguous run of ~nwg/8 ids walks one BAND of 8 row-tiles in SUPER-COLUMNS of 8
    // column-tiles, so the 64 tiles it runs concurrently (2 per CU x 32 CUs) share 8 A panels and 8 W panels (~4.7 MB at
    // K = 768) instead of 3 A panels and ALL of W (PMC: 2.4x the algorithmic fetch with the row-major order).
    constexpr int MB = 8, SN = 8;
    const int band = wg / (MB * tiles_n), rb = wg % (MB * tiles_n);
    const int mb = min(MB, tiles_m - band * MB);
    const int sc = rb / (mb * SN), r2 = rb % (mb * SN);
    const int sn = min(SN, tiles_n - sc * SN);
    const int m0 = (band * MB + r2 / sn) * BM, n0 = (sc * SN + r2 % sn) * BN;
    const int nkt = (p.K + 31) >> 5;
    const bf16_t* zero = (const bf16_t*)g_zero16;

    // one 1-KiB LDS-DMA piece per call: pieces 0..3 = A rows, 4..5 = W rows of K-tile kt
    auto stage_piece = [&](int slot, int kt, int piece) {
        char* sa = smem + slot * STAGE;
        const int k0 = kt << 5;
        if (piece < 4) {
            const int qd = piece * 256 + tid;
            const int row = qd >> 2, c = (qd & 3) ^ (((row >> 2) & 1) << 1);
            const int gr = m0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.M && gk < p.K) ? p.A + (size_t)gr * p.lda + gk : zero;
            glds16(src, sa + (piece * 256 + wave * 64) * 16);
        } else {
            const int qd = (piece - 4) * 256 + tid;
            const int row = qd >> 2, c = (qd & 3) ^ (((row >> 2) & 1) << 1);
            const int gr = n0 + row, gk = k0 + c * 8;
            const bf16_t* src = (gr < p.N && gk < p.K) ? p.W + (size_t)gr * p.ldw + gk : zero;
            glds16(src, sa + A_BYTES + ((piece - 4) * 256 + wave * 64) * 16);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int pc = 0; pc < 6; ++pc) stage_piece(0, 0, pc);
    // tiles past the end of K are sourced from the zero chunk (bounds check in stage_piece), so the prefetch is issued
    // unconditionally: no branch in the loop body, one basic block, and the wait is always vmcnt(6)
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) stage_piece(1, 1, pc);

    const int coff = (kq ^ (((fr >> 2) & 1) << 1)) << 4;
    const int a_off = (wm * 128 + fr) * 64 + coff, b_off = (wn * 64 + fr) * 64 + coff;
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* sa = smem + slot * STAGE + a_off;
        const char* sb = smem + slot * STAGE + A_BYTES + b_off;
        const int nslot = slot == 0 ? 2 : slot - 1;
        bf16x8 xa[8], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8*)(sb + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = *(const bf16x8*)(sa + i * 16 * 64);
        // MFMA rows in fragment-arrival order; the next-next tile's six LDS-DMA pieces are issued between the MFMA groups
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            if (i < 6) stage_piece(nslot, kt + 2, i);
        }
        // shape the schedule: 6 fragment reads up front, then one more read behind every group of 4 MFMAs, so each MFMA group
        // waits (counted lgkmcnt) only for the fragments it consumes instead of for all 12 reads
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        slot = slot == 2 ? 0 : slot + 1;
    }

    // ---- epilogue through LDS: every global access is a whole 256-B (fp32) / 128-B (bf16) row segment ----------------
    // Each wave owns a private [64][68] fp32 patch (17 KiB); two passes cover its 128 rows.  Accumulator layout (lane =
    // output row, 4 consecutive columns) would store 32-B pieces of 16 different rows per instruction; after the LDS
    // transpose a wave instruction covers 4 full rows, and bias / LayerScale / residual loads are coalesced too.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (zero-sourced) prefetches must land before LDS is reused
    __syncthreads();                                   // last slot fully consumed by every wave, no LDS-DMA in flight
    float* wl = (float*)(smem + wave * (64 * 68 * 4));
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int n = n0 + wn * 64 + ec;
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
            for (int j = 0; j < 4; ++j) *(f32x4*)(wl + (i4 * 16 + fr) * 68 + j * 16 + kq * 4) = acc[half * 4 + i4][j];
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

// split-K second pass: C = epi(sum_s partial[s]) — the slices are summed in ascending order (deterministic), then the same
// epilogue as the one-pass kernel
__global__ __launch_bounds__(256) void splitk_finish_kernel(const GemmParams p) {
    const int n4 = p.N >> 2;
    const size_t total = (size_t)p.M * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        f32x4 v = *(const f32x4*)(p.partial + (size_t)m * p.N + n);
        for (int s = 1; s < p.nsplit; ++s) v += *(const f32x4*)(p.partial + ((size_t)s * p.M + m) * p.N + n);
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (p.act == 1) v = gelu_erf4(v);
        if (p.scale) v *= *(const f32x4*)(p.scale + n);
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

template <int TAG>
int launch_v3(const GemmParams& p, hipStream_t stream) {
    const int tiles = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    constexpr int LDS = 3 * (256 + 128) * 64;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)gemm_bf16_nt_v3_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    gemm_bf16_nt_v3_kernel<TAG><<<tiles, 256, LDS, stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v3)");
    return 0;
}

template <int BM, int BN, int TAG>
int launch(const GemmParams& p, hipStream_t stream) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr int LDS = 2 * (BM + BN) * 128;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<BM, BN, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    gemm_bf16_nt_kernel<BM, BN, TAG><<<dim3(tiles, p.nsplit), 256, LDS, stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt");
    if (p.partial) {
        const size_t items = (size_t)p.M * (p.N >> 2);
        splitk_finish_kernel<<<(int)min((size_t)2048, (items + 255) / 256), 256, 0, stream>>>(p);
        MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(split-K finish)");
    }
    return 0;
}

}  // namespace

// ---- live timing of the dominant kernel (bench.py roofline leg) ----------------------------------------------------------
// mode 1: HIP events around every tag-1 launch, recorded on the stream the kernel is launched on (eager launches only:
//         events cannot be read back out of a replayed hipGraph on this ROCm);
// mode 2: an in-kernel launch clock — every tag-1 launch issued (or CAPTURED) while the mode is on gets a private slot of
//         four u64 in device memory; the first workgroup in and the last workgroup out stamp the 100-MHz wall clock, so a
//         slot always holds the begin / end of the LAST execution of its launch, replays of a captured graph included.
namespace {
struct GemmProfile {
    int mode = 0;
    int collect_mode = 0;            // the last mode ARMED (1 or 2): what collect() reads — mode 0 only stops the gathering
    std::vector<hipEvent_t> ev;      // mode 1: pairs (start, stop)
    size_t used = 0;
    double flops = 0.0;
    unsigned long long* slots = nullptr;     // mode 2: device [MAX_SLOTS][4]
    std::vector<double> slot_flops;
    int slot_dev = -1;
};
constexpr size_t MAX_PROF_SLOTS = 4096;
GemmProfile g_prof;
}  // namespace

extern "C" int medp_gemm_profile_enable(int mode) {
    MEDP_CHECK_ARG(mode >= 0 && mode <= 2, "gemm_profile_enable: mode must be 0 (off), 1 (HIP events) or 2 (in-kernel clock)");
    if (mode == 1) {
        g_prof.used = 0;
        g_prof.flops = 0.0;
    }
    if (mode == 2) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!g_prof.slots || g_prof.slot_dev != dev) {
            hipError_t e = hipMalloc((void**)&g_prof.slots, MAX_PROF_SLOTS * 4 * sizeof(unsigned long long));
            if (e != hipSuccess) { medp_set_error("gemm_profile_enable: %s", hipGetErrorString(e)); return (int)e; }
            g_prof.slot_dev = dev;
        }
        hipError_t e = hipMemset(g_prof.slots, 0, MAX_PROF_SLOTS * 4 * sizeof(unsigned long long));
        if (e != hipSuccess) { medp_set_error("gemm_profile_enable: %s", hipGetErrorString(e)); return (int)e; }
        g_prof.slot_flops.clear();
    }
    g_prof.mode = mode;      // mode 0 keeps what was gathered for collect()
    if (mode != 0) g_prof.collect_mode = mode;
    return 0;
}

extern "C" int medp_gemm_profile_collect(double* total_ms, long long* n_launches, double* total_flops) {
    MEDP_CHECK_ARG(total_ms && n_launches && total_flops, "gemm_profile_collect: null argument");
    double ms = 0.0;
    // Dispatch on the mode that was armed last, NOT on "slots exist": slots armed by an earlier mode-2 capture stay alive (their
    // graph may still be replayed) while a later mode-1 measurement of eager launches must report ITS events.
    if (g_prof.collect_mode == 2) {          // in-kernel clocks: the caller has synchronised the device
        const size_t n = g_prof.slot_flops.size();
        std::vector<unsigned long long> h(n * 4);
        hipError_t e = hipMemcpy(h.data(), g_prof.slots, n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { medp_set_error("gemm_profile_collect: %s", hipGetErrorString(e)); return (int)e; }
        int khz = 100000;
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, g_prof.slot_dev);
        double fl = 0.0;
        long long cnt = 0;
        for (size_t i = 0; i < n; ++i) {
            if (h[4 * i + 1] <= h[4 * i]) continue;      // never executed
            ms += (double)(h[4 * i + 1] - h[4 * i]) / (double)khz;
            fl += g_prof.slot_flops[i];
            ++cnt;
        }
        *total_ms = ms;
        *n_launches = cnt;
        *total_flops = fl;
        return 0;
    }
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        hipError_t e = hipEventSynchronize(g_prof.ev[i + 1]);
        if (e != hipSuccess) { medp_set_error("gemm_profile_collect: %s", hipGetErrorString(e)); return (int)e; }
        float t = 0.f;
        e = hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
        if (e != hipSuccess) { medp_set_error("gemm_profile_collect: %s", hipGetErrorString(e)); return (int)e; }
        ms += t;
    }
    *total_ms = ms;
    *n_launches = (long long)(g_prof.used / 2);
    *total_flops = g_prof.flops;
    g_prof.used = 0;
    g_prof.flops = 0.0;
    return 0;
}

// Debug hook (NOT part of the C ABI in include/medp_hip.h; tools/time_branches.py): the raw mode-2 stamps, slot i ->
// out[2 i] = first workgroup in, out[2 i + 1] = last workgroup out (ticks of the wall clock, *khz per ms), in launch order.
extern "C" int medp_dbg_gemm_profile_raw(unsigned long long* out, int max_slots, int* n_slots, int* khz) {
    const int n = (int)std::min(g_prof.slot_flops.size(), (size_t)std::max(max_slots, 0));
    std::vector<unsigned long long> h((size_t)n * 4);
    if (n > 0 && hipMemcpy(h.data(), g_prof.slots, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) { out[2 * i] = h[4 * i]; out[2 * i + 1] = h[4 * i + 1]; }
    *n_slots = n;
    *khz = 100000;
    (void)hipDeviceGetAttribute(khz, hipDeviceAttributeWallClockRate, g_prof.slot_dev);
    return 0;
}

// Split-K plan for SMALL grids (DuETT's skinny GEMMs: M = 3136 / 6208 rows, N = 72 / 512, K = 1176 / 2328 give 25-100 tiles
// on 256 CUs with a 73-step K-loop each): K is cut into S slices so that tiles x S fills the chip; S = 1: no split.
static int splitk_slices(int M, int N, int K) {
    static const int force = [] { const char* e = getenv("MEDP_GEMM_SPLITK"); return e ? atoi(e) : -1; }();
    const int tiles = ((M + 127) / 128) * (N <= 64 ? (N + 63) / 64 : (N + 127) / 128);
    const int nkt = (K + 63) / 64;
    if (force >= 0) return max(1, min(force, nkt));
    if (tiles >= 128 || nkt < 8) return 1;
    // Round-3 sweep (tools/bench_gemm_splitk.py, profiles/r03_ab_experiments.txt): below 64 tiles the slices should fill the chip
    // (event / time qkv: 33.5 -> 15.8 us, 19.4 -> 15.3); between 64 and 127 tiles a split pays only for a long K (event ff1, 100 tiles x
    // 37 K-tiles: 36.3 / 28.1 / 25.6 / 25.6 / 29.6 us at 1 / 2 / 3 / 4 / 6 slices) and costs for a short one (ts_proj, 96 tiles x 18
    // K-tiles: 19.5 us unsplit, 22.3 with 2 slices).
    if (tiles < 64) return max(1, min(min(256 / tiles, nkt / 4), 16));
    return nkt >= 32 ? max(1, min(384 / tiles, nkt / 8)) : 1;
}

extern "C" size_t medp_gemm_nt_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int S = splitk_slices(M, N, K);
    return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

static int gemm_dispatch(int tag, const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                         const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                         float* workspace, size_t workspace_bytes, const MedpGemmFold* fold, void* stream);

int medp_gemm_bf16_nt_tagged_ws(int tag, const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                                const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                                float* workspace, size_t workspace_bytes, void* stream) {
    return gemm_dispatch(tag, A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, workspace, workspace_bytes, nullptr, stream);
}

// the conditions under which gemm_dispatch takes the 256 x 256 tile kernels (v6 / v7): the only ones that carry the LayerNorm fold
static bool uses_tile256(int M, int N) {
    static const int force = [] { const char* e = getenv("MEDP_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
    const int tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    return M >= 2048 && N >= 256 && (force == 6 || (force == 0 && tiles256 >= 160));
}

bool medp_gemm_fold_eligible(int M, int N, int K) {
    static const int ragged_on = [] { const char* e = getenv("MEDP_GEMM_RAGGED"); return e ? atoi(e) : 0; }();
    return uses_tile256(M, N) && !ragged_on && K % 64 == 0 && N % 256 == 0;
}

int medp_gemm_bf16_nt_fold(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc, const float* bias,
                           const float* scale, const float* residual, int ldr, int act, int out_bf16, const MedpGemmFold& fold, void* stream) {
    MEDP_CHECK_ARG(medp_gemm_fold_eligible(M, N, K), "gemm(fold): M=%d N=%d K=%d does not take the 256-tile kernels", M, N, K);
    MEDP_CHECK_ARG(!fold.stats_in || (fold.colsum && fold.ln_dim > 0 && fold.stats_tiles > 0 && fold.stats_tiles <= 4 && out_bf16 && !residual && !scale),
                   "gemm(fold): a consumer needs colsum, ln_dim, 1..4 stats tiles, a bf16 result and no residual / scale");
    MEDP_CHECK_ARG(!fold.c2 || (fold.stats_out && !out_bf16 && fold.ldc2 % 4 == 0), "gemm(fold): a producer writes fp32 C, c2 (ld % 4 == 0) and stats_out");
    return gemm_dispatch(1, A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, nullptr, 0, &fold, stream);
}

int medp_gemm_bf16_nt_tagged(int tag, const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                             const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                             void* stream) {
    return medp_gemm_bf16_nt_tagged_ws(tag, A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, nullptr, 0, stream);
}

static int gemm_dispatch(int tag, const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                         const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                         float* workspace, size_t workspace_bytes, const MedpGemmFold* fold, void* stream) {
    MEDP_CHECK_ARG(A && W && C, "gemm: null operand");
    MEDP_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    MEDP_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0, "gemm: K, lda, ldw must be multiples of 8 (16-B chunks)");
    MEDP_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0, "gemm: N and ldc must be multiples of 4 (vector epilogue)");
    MEDP_CHECK_ARG(lda >= K && ldw >= K && ldc >= N, "gemm: leading dimension smaller than the row");
    MEDP_CHECK_ARG(!residual || ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    MEDP_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0,
                   "gemm: operands must be 16-byte aligned");
    MEDP_CHECK_ARG(act == 0 || act == 1, "gemm: act must be 0 (none) or 1 (gelu)");
    GemmParams p{(const bf16_t*)A, (const bf16_t*)W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, nullptr, 1};
    hipStream_t s = (hipStream_t)stream;
    if (workspace && tag != 1) {
        const int S = splitk_slices(M, N, K);
        if (S > 1) {
            MEDP_CHECK_ARG(workspace_bytes >= (size_t)S * M * N * sizeof(float), "gemm: split-K workspace too small (medp_gemm_nt_workspace_bytes)");
            p.partial = workspace;
            p.nsplit = S;
            return N <= 64 ? launch<128, 64, 0>(p, s) : launch<128, 128, 0>(p, s);
        }
    }
    static const int force = [] { const char* e = getenv("MEDP_GEMM_VARIANT"); return e ? atoi(e) : 0; }();   // 1 = v1, 2 = v2 (A/B tests)
    const bool use_v2 = force == 2;
    // v3 (256 x 128 tiles) only where its grid covers most of the chip: below that the 128 x 128 kernel has twice the tiles
    // (img_proj: 130 -> 258, DuETT ff1 on the time axis: 100 -> 196) and wins on CU fill what it loses per tile
    static const int v3_min_tiles = [] { const char* e = getenv("MEDP_GEMM_V3_MIN_TILES"); return e ? atoi(e) : 192; }();
    const int tiles_v3 = ((M + 255) / 256) * ((N + 127) / 128);
    const bool use_v3 = force == 3 || (force == 0 && M >= 2048 && N >= 256 && tiles_v3 >= v3_min_tiles);
    MedpGemmArgs a4{A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, nullptr, 0};
    if (fold) a4.fold = *fold;
    if (N <= 64) return launch<128, 64, 0>(p, s);
    // v6 (256 x 256 x 64, 8 waves ping-pong, gemm_bf16_v6.hip) runs ONE workgroup per CU: default once there are enough
    // 256^2 tiles to occupy most of the chip (the CXR-encoder shapes: 195 / 585 / 780 tiles); v3 keeps the smaller grids
    const bool use_v6 = uses_tile256(M, N);
    MEDP_CHECK_ARG(!fold || use_v6, "gemm(fold): shape does not take the 256-tile kernels");
    // v7: the same K-loop, persistent over the tile list, where the grid is more than one round of workgroups (qkv, fc1)
    static const int v7_on = [] { const char* e = getenv("MEDP_GEMM_V7"); return e ? atoi(e) : 1; }();
    const bool use_v7 = use_v6 && force != 6 && v7_on && medp_gemm_v7_eligible(a4);
    // The ragged last rows (M = 64 * 257: 64 rows past the last full 256-row tile) go to a skinny launch of their own where that
    // saves the persistent kernel a round of workgroups (fc1: 780 tiles = 4 rounds on 256 CUs, 768 = 3; qkv: 585 and 576 are both
    // 3 rounds and stay one launch).  Same bits either way (gemm_ragged_rows.hip).  OFF by default (MEDP_GEMM_RAGGED=1 turns it
    // on): alone fc1 gains (102 -> 94 us), but inside the teacher step the 3 full rounds take all 256 CUs where the 4 rounds of
    // 200 workgroups leave 56 to the other branch for the same CU-time — in-box A/B 5.17 ms with, 5.11-5.13 ms without.
    static const int ragged_on = [] { const char* e = getenv("MEDP_GEMM_RAGGED"); return e ? atoi(e) : 0; }();
    const int rag_rows = M % 256, tiles_n256 = (N + 255) / 256, nfull256 = (M / 256) * tiles_n256;
    const bool split_ragged = use_v7 && ragged_on && rag_rows > 0 && rag_rows <= 128 && nfull256 > 256 &&
                              (nfull256 + tiles_n256 + 255) / 256 > (nfull256 + 255) / 256;
    auto launch_v67 = [&](int tg) {
        if (split_ragged) {
            MedpGemmArgs ar = a4, am = a4;
            ar.prof_flags = 2;                 // one clock over both launches: the skinny one stamps the arrival ...
            am.prof_flags = 1;                 // ... the tile kernel the departure
            am.M = M - rag_rows;
            if (medp_gemm_v7_eligible(am)) {
                int rc = medp_gemm_ragged_rows_launch(ar, M - rag_rows, stream);
                if (rc != 0) return rc;
                rc = medp_gemm_v7_launch(am, tg, stream);
                return rc != -1 ? rc : medp_gemm_v6_launch(am, tg, stream);
            }
        }
        if (use_v7) {
            const int rc = medp_gemm_v7_launch(a4, tg, stream);
            if (rc != -1) return rc;
        }
        return medp_gemm_v6_launch(a4, tg, stream);
    };
    if (use_v6 && tag != 1) return launch_v67(0);
    if (use_v3 && tag != 1) return launch_v3<0>(p, s);
    if (use_v2 && tag != 1) return launch_v2<0>(p, s);
    if (tag == 1) {
        const bool prof = g_prof.mode == 1;
        if (g_prof.mode == 2 && use_v6 && g_prof.slot_flops.size() < MAX_PROF_SLOTS) {
            a4.prof = g_prof.slots + 4 * g_prof.slot_flops.size();
            g_prof.slot_flops.push_back(2.0 * (double)M * (double)N * (double)K);
        }
        if (prof) {
            if (g_prof.used + 2 > g_prof.ev.size()) {
                for (int i = 0; i < 2; ++i) {
                    hipEvent_t e;
                    if (hipEventCreate(&e) != hipSuccess) { medp_set_error("gemm profile: hipEventCreate failed"); return 1; }
                    g_prof.ev.push_back(e);
                }
            }
            hipEventRecord(g_prof.ev[g_prof.used], s);
        }
        const int rc = use_v6 ? launch_v67(1) : (use_v3 ? launch_v3<1>(p, s) : (use_v2 ? launch_v2<1>(p, s) : launch<128, 128, 1>(p, s)));
        if (prof) {
            hipEventRecord(g_prof.ev[g_prof.used + 1], s);
            g_prof.used += 2;
            g_prof.flops += 2.0 * (double)M * (double)N * (double)K;
        }
        return rc;
    }
    return launch<128, 128, 0>(p, s);
}

extern "C" int medp_gemm_bf16_nt(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                                 const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                                 void* stream) {
    return medp_gemm_bf16_nt_tagged(0, A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, stream);
}

extern "C" int medp_gemm_bf16_nt_ws(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                                    const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                                    float* workspace, size_t workspace_bytes, void* stream) {
    return medp_gemm_bf16_nt_tagged_ws(0, A, W, C, M, N, K, lda, ldw, ldc, bias, scale, residual, ldr, act, out_bf16, workspace,
                                       workspace_bytes, stream);
}
