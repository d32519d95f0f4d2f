// v4: PERSISTENT form of the v3 GEMM (256 x 128 x 32 tiles, 4 waves, 128 x 64 per wave, three 24-KiB LDS slots by LDS-DMA).
// v3 launches one workgroup per output tile: with K = 768 a tile is only 24 K-steps, so the two-K-tile prologue (a full
// memory latency with nothing to multiply) and the epilogue are ~25 % of a tile's life, hidden only by the co-resident block.
// Here 2 workgroups per CU stay resident and walk their share of the tiles; the LDS ring runs CONTINUOUSLY across tile
// boundaries (the stream of K-tiles of tile j+1 follows that of tile j), so the next tile's first K-tiles are already
// landing while the last K-steps of the current tile are multiplied, and a wave's epilogue (straight from the accumulators,
// no LDS, no barrier) overlaps the other waves' MFMAs.
//   iteration g:  vmcnt(N) -> K-tile g landed ; s_barrier ; ds_read fragments ; 32 MFMA with the 6 LDS-DMA pieces of
//                 K-tile g+2 issued between the MFMA groups ; if g closes a tile: epilogue, clear accumulators.
// Tile ownership: XCD group x = blockIdx & 7 owns a contiguous run of tile ids (band / super-column order, see v3), its
// blocks take ids li, li + bpx, li + 2 bpx, ...  — placement affects speed only.
#include <stdlib.h>

#include "common.h"
#include "gemm_v4.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_v4[4] = {0, 0, 0, 0};
// sink for the epilogue stores of out-of-range lanes: every wave then issues EXACTLY 32 store instructions per tile, which
// lets the K-loop wait with a counted vmcnt that skips over them instead of waiting for HBM write latency (see the loop)
__device__ __attribute__((aligned(16))) uint32_t g_store_sink_v4[64 * 4];

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM = 256, BN = 128, A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;

template <int TAG>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_v4_kernel(const MedpGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)g_zero16_v4;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int nkt = (p.K + 31) >> 5;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int count_x = q + (xcd < r ? 1 : 0);
    const int base_x = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int my_tiles = count_x > li ? (count_x - li + bpx - 1) / bpx : 0;
    if (my_tiles == 0) return;
    const int total = my_tiles * nkt;

    auto decode = [&](int j, int& m0, int& n0) {          // j-th tile of this block -> tile origin (band / super-column order)
        constexpr int MB = 8, SN = 8;
        const int wg = base_x + li + j * bpx;
        const int band = wg / (MB * tiles_n), rb = wg % (MB * tiles_n);
        const int mb = min(MB, tiles_m - band * MB);
        const int sc = rb / (mb * SN), r2 = rb % (mb * SN);
        const int sn = min(SN, tiles_n - sc * SN);
        m0 = (band * MB + r2 / sn) * BM;
        n0 = (sc * SN + r2 % sn) * BN;
    };

    // ---- prefetch stream state ---------------------------------------------------------------------------------
    int pf_j = 0, pf_kt = 0, pf_m0, pf_n0;
    bool pf_valid = true;
    decode(0, pf_m0, pf_n0);
    auto stage_piece = [&](int slot, int piece) {
        char* sa = smem + slot * STAGE;
        const int k0 = pf_kt << 5;
        if (piece < 4) {
            const int qd = piece * 256 + tid;
            const int row = qd >> 2, c = (qd & 3) ^ (((row >> 2) & 1) << 1);
            const int gr = pf_m0 + row, gk = k0 + c * 8;
            const bf16_t* src = (pf_valid && gr < p.M && gk < p.K) ? A + (size_t)gr * p.lda + gk : zero;
            glds16(src, sa + (piece * 256 + wave * 64) * 16);
        } else {
            const int qd = (piece - 4) * 256 + tid;
            const int row = qd >> 2, c = (qd & 3) ^ (((row >> 2) & 1) << 1);
            const int gr = pf_n0 + row, gk = k0 + c * 8;
            const bf16_t* src = (pf_valid && gr < p.N && gk < p.K) ? W + (size_t)gr * p.ldw + gk : zero;
            glds16(src, sa + A_BYTES + ((piece - 4) * 256 + wave * 64) * 16);
        }
    };
    auto pf_advance = [&]() {
        if (++pf_kt == nkt) {
            pf_kt = 0;
            if (++pf_j < my_tiles) decode(pf_j, pf_m0, pf_n0);
            else pf_valid = false;
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int pc = 0; pc < 6; ++pc) stage_piece(0, pc);
    pf_advance();
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) stage_piece(1, pc);
    pf_advance();

    const int coff = (kq ^ (((fr >> 2) & 1) << 1)) << 4;
    const int a_off = (wm * 128 + fr) * 64 + coff, b_off = (wn * 64 + fr) * 64 + coff;
    int slot = 0, c_kt = 0, c_j = 0, c_m0, c_n0;
    decode(0, c_m0, c_n0);

    int store_grace = 0;      // iterations during which the previous tile's 32 stores may still be in flight
    for (int g = 0; g < total; ++g) {
        // K-tile g landed when at most the 6 pieces of K-tile g+1 are still in flight.  vmcnt retires in issue order and counts
        // stores too: for two iterations after an epilogue the 32 output stores sit between the DMA pieces in the queue, so the
        // wait allows 32 more outstanding operations; by the third iteration the stores have long completed (measured: waiting
        // for them right away cost 16-20 % of a K = 768 tile).
        if (store_grace > 0) {
            asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            --store_grace;
        } else {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* sa = smem + slot * STAGE + a_off;
        const char* sb = smem + slot * STAGE + A_BYTES + b_off;
        const int nslot = slot == 0 ? 2 : slot - 1;
        bf16x8 xa[8], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = *(const bf16x8*)(sb + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = *(const bf16x8*)(sa + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            if (i < 6) stage_piece(nslot, i);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        pf_advance();
        slot = slot == 2 ? 0 : slot + 1;

        if (++c_kt == nkt) {
            // ---- epilogue of tile c_j straight from the accumulators: lane = output row, 4 consecutive columns ----------
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = c_m0 + wm * 128 + i * 16 + fr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = c_n0 + wn * 64 + j * 16 + kq * 4;
                    f32x4 v = acc[i][j];
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const bool ok = m < p.M && n < p.N;
                    const int nn = ok ? n : 0;
                    const size_t mm = ok ? (size_t)m : 0;
                    if (p.bias) v += *(const f32x4*)(p.bias + nn);
                    if (p.act == 1) {
                    v = gelu_erf4(v);
                    }
                    if (p.scale) v *= *(const f32x4*)(p.scale + nn);
                    if (p.residual) v += *(const f32x4*)(p.residual + mm * p.ldr + nn);
                    if (TAG == 7) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }   // diagnostic build: no stores
                    char* sink = (char*)g_store_sink_v4 + lane * 16;
                    if (p.out_bf16) {
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        uint2* dst = ok ? (uint2*)((bf16_t*)p.C + mm * p.ldc + nn) : (uint2*)sink;
                        *dst = o;
                    } else {
                        f32x4* dst = ok ? (f32x4*)((float*)p.C + mm * p.ldc + nn) : (f32x4*)sink;
                        *dst = v;
                    }
                }
            }
            c_kt = 0;
            store_grace = 2;
            if (++c_j < my_tiles) decode(c_j, c_m0, c_n0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing zero-sourced prefetches must land before the LDS is released
}

template <int TAG>
int launch_v4(const MedpGemmArgs& p, hipStream_t stream) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr int LDS = 3 * STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_bf16_nt_v4_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    // 2 resident workgroups per CU (72 KiB LDS each), grid a multiple of 8 so every XCD group has the same block count
    const int grid = min(512, ((tiles + 7) / 8) * 8);
    gemm_bf16_nt_v4_kernel<TAG><<<grid, 256, LDS, stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v4)");
    return 0;
}

}  // namespace

int medp_gemm_v4_launch(const MedpGemmArgs& a, int tag, void* stream) {
    static const bool nostore = getenv("MEDP_GEMM_NOSTORE") != nullptr;     // timing-only diagnostic, outputs are not written
    if (nostore) return launch_v4<7>(a, (hipStream_t)stream);
    return tag == 1 ? launch_v4<1>(a, (hipStream_t)stream) : launch_v4<0>(a, (hipStream_t)stream);
}
