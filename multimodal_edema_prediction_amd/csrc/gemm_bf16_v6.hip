// v6: 256 x 256 x 64 tiles, EIGHT waves (2 x 4, 128 x 64 per wave), two waves per SIMD in PING-PONG.
//
// v3/v5 measurements: a wave that owns its SIMD alone pays for every LDS-DMA issue (~100+ cycles inside a K-step that also
// carries the fragment reads) and for every barrier with an idle matrix pipe.  Here each SIMD holds one wave of group 0
// (rows 0..127 of the tile) and one of group 1 (rows 128..255).  A K-tile (64 deep) is four PHASES of 16 MFMAs (one
// 64 x 32 quadrant of the wave's 128 x 64 output, both k-halves); every phase is
//     [load section: ds_read fragments, 2 LDS-DMA pieces, lgkmcnt(0)]  s_barrier  [16 MFMA at raised priority]  s_barrier
// and group 1 runs ONE barrier behind group 0, so between any two consecutive barriers one group multiplies while the
// other issues its loads: the matrix pipe always has a wave whose operands are already in registers.
//
// LDS: two K-tile buffers of 64 KiB = four 16-KiB half-tiles each (A rows 0-127 | A rows 128-255 | W rows 0-127 |
// W rows 128-255), 128-B rows, 16-B chunk c stored at c ^ (row & 7) (conflict-free ds_read_b128 lane groups).
// Fragment schedule of K-tile t (per wave): P1 reads A0 (8) + W0 (4), P2 W1 (4), P3 A1 (8), P4 nothing;
// MFMA quadrants: P1 (A0,W0)  P2 (A0,W1)  P3 (A1,W1)  P4 (A1,W0).
// Phase plan of K-tile t (per wave; in-kernel phase clocks, tools/trace_gemm_v7.py --phases, decided it: a load section with
// 12 ds_read_b128 took 670 ticks against ~330 for the 16 MFMAs it has to hide behind, one with 8 or 4 reads 300-360):
//     P1: read A rows 0-63 (8),   DMA A rows 64-127 (t+1)   MFMA (A0, W0)
//     P2: read W1 (4),            DMA A rows 0-63 (t+2)     MFMA (A0, W1)    wait vmcnt(10): A rows 64-127 of K-tile t
//     P3: read A rows 64-127 (8), DMA W half 0 (t+2)        MFMA (A1, W1)    wait vmcnt(6):  W of K-tile t+1
//     P4: read W0 of K-TILE t+1 (4) into the other W0 register set,
//                                 DMA W half 1 (t+2)        MFMA (A1, W0)
// i.e. no load section carries more than 8 fragment reads or more than 2 DMA pieces (4 pieces + 4 reads in one section cost
// 560-700 ticks).  Every LDS region is refilled (same buffer) one or two phases after its last read.  The counts of the
// waits are "everything but the pieces issued after the one needed" (2 pieces per phase, in the order above).
// RAW: a wait sits in a load section, before a barrier every wave passes, and the data is first read one phase later (group 1
// runs one barrier behind: a wait placed after the MFMAs would not yet have been executed by it).  WAR: every ds_read is
// retired (lgkmcnt(0)) before the barrier that ends its load section; the refill is issued one phase later.
#include <stdlib.h>

#include "common.h"
#include "gemm_variants.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_v6[4] = {0, 0, 0, 0};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM = 256, BN = 256, HALF = 128 * 128, KBUF = 4 * HALF;   // 16 KiB half-tile, 64 KiB K-tile buffer
constexpr int LDS_MAIN = 2 * KBUF, LDS_EPI = 8 * 64 * 68 * 4;
constexpr int LDS_STATS = LDS_EPI;                         // LayerNorm-fold producer: [256 rows][4 column waves][2] floats behind the epilogue patches
constexpr int LDS_BYTES = LDS_STATS + 256 * 4 * 2 * 4;
static_assert(LDS_BYTES >= LDS_MAIN && LDS_BYTES <= 160 * 1024, "v6 LDS plan");

// patch write -> read (and read -> next write) inside ONE wave: the LDS executes a wave's operations in order, only the
// compiler must not reorder them.  (A workgroup-scope fence here also emits vmcnt(0): the second half of the epilogue would
// wait for the global stores of the first.)
#define MEDP_WAVE_LDS_SYNC()                        \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_wave_barrier();            \
        asm volatile("" ::: "memory");              \
    } while (0)

#define MEDP_BAR()                                  \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        __builtin_amdgcn_s_barrier();               \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

template <int TAG>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v6_kernel(const MedpGemmArgs p, unsigned* slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;          // wm = ping-pong group
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)g_zero16_v6;
    MEDP_PROF_ENTER(p.prof, p.prof_flags);

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // Tile index.  Static: workgroup b owns tile b.  With a ticket block (`slot`, one-round grids inside a multi-stream step: proj /
    // fc2, 195 tiles) MORE workgroups than tiles are launched and each draws its tile from the queue of its XCD (ticket t of XCD x
    // is tile 8 t + x: the tile the static map gives that XCD's t-th workgroup, so the L2 locality of the map below holds): the
    // tiles go to whichever CUs are free FIRST, a workgroup that becomes resident late — its CU still held short kernels of the
    // step's other branch — finds the queue empty and leaves, instead of starting its 30-us tile late.  (An experiment, off by
    // default: see launch_v6.)
    int bid = blockIdx.x;
    if (slot) {
        const int xcd = blockIdx.x & 7;
        volatile unsigned* box = (volatile unsigned*)smem;
        if (tid == 0) box[0] = __hip_atomic_fetch_add(slot + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        bid = 8 * (int)__builtin_amdgcn_readfirstlane(box[0]) + xcd;
        __syncthreads();                                   // everyone has read the ticket before the first LDS-DMA piece may land on it
        if (bid >= tiles_m * tiles_n) {
            if (tid == 0) {                                // the last workgroup to leave re-arms the ticket block (as in gemm_bf16_v7.hip)
                const unsigned left = __hip_atomic_fetch_add(slot + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (left == gridDim.x - 1)
                    for (int i = 0; i < 9; ++i) __hip_atomic_store(slot + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            MEDP_PROF_LEAVE(p.prof, p.prof_flags);
            return;
        }
    }
    // Block -> tile.  Workgroups are dispatched in index order, block b to XCD b % 8, one per CU (32 CUs per XCD).  The
    // FULL row-tiles come first: XCD x gets a contiguous run of them (band x super-column order inside, see v3) so its L2 sees
    // few panels; the cheap tiles of a ragged last row (M = 64 * 257: 64 live rows, three quarters of their MFMAs skipped)
    // take the highest indices, i.e. they are dispatched LAST.  fc1 (768 full + 12 ragged tiles) is then 96 full tiles =
    // exactly 3 rounds per XCD plus a short ragged tail, instead of a fourth round that holds one full tile per XCD.
    const int rag = (p.M % BM) ? 1 : 0;
    const int tm_full = tiles_m - rag;
    const int nfull = tm_full * tiles_n;
    const int full8 = nfull & ~7;                      // full tiles dealt in runs of nfull/8 per XCD; the rest by index
    int m0, n0;
    if (bid < nfull) {
        const int wg = bid < full8 ? (bid & 7) * (full8 >> 3) + (bid >> 3) : bid;   // the < 8 leftover full tiles keep their index
        constexpr int MB = 8, SN = 4;
        const int band = wg / (MB * tiles_n), rb = wg % (MB * tiles_n);
        const int mb = min(MB, tm_full - band * MB);
        const int sc = rb / (mb * SN), r2 = rb % (mb * SN);
        const int sn = min(SN, tiles_n - sc * SN);
        m0 = (band * MB + r2 / sn) * BM;
        n0 = (sc * SN + r2 % sn) * BN;
    } else {                                           // the ragged row, one tile per column
        m0 = tm_full * BM;
        n0 = (bid - nfull) * BN;
    }
    const int nkt = (p.K + 63) >> 6;

    // ---- LDS-DMA staging: half-tile = 128 rows x 8 chunks; lane's two pieces are rows (tid>>3) and (tid>>3)+64 ------
    const int srow = tid >> 3, schunk = (tid & 7) ^ (srow & 7);       // source chunk for LDS position (tid & 7)
    const bf16_t* a_src[2][2];                                        // [half][piece] row base (k = 0) or nullptr
    const bf16_t* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ra = m0 + h * 128 + j * 64 + srow, rw = n0 + h * 128 + j * 64 + srow;
            a_src[h][j] = ra < p.M ? A + (size_t)ra * p.lda + schunk * 8 : nullptr;
            w_src[h][j] = rw < p.N ? W + (size_t)rw * p.ldw + schunk * 8 : nullptr;
        }
    // piece j (rows 64 j .. 64 j + 63) of half-tile `which` (0: A rows 0-127, 1: A rows 128-255, 2: W rows 0-127, 3: W rows
    // 128-255) of K-tile kt -> buffer kt & 1; past K (the tail of the stream) the source is the zero chunk, counts stay uniform
    auto stage_piece = [&](int kt, int which, int j) {
        char* dst = smem + (kt & 1) * KBUF + which * HALF + wave * 1024 + j * 8192;
        const int k0 = kt * 64;
        const bool kin = k0 + schunk * 8 < p.K;
        const bf16_t* base = which < 2 ? a_src[which & 1][j] : w_src[which & 1][j];
        glds16((base != nullptr && kin) ? base + k0 : zero, dst);
    };
    auto stage_a = [&](int kt, int j) { stage_piece(kt, 0, j); stage_piece(kt, 1, j); };
    auto stage_w = [&](int kt, int h) { stage_piece(kt, 2 + h, 0); stage_piece(kt, 2 + h, 1); };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: K-tiles 0 and 1, in the order of the steady-state stream
    stage_a(0, 0); stage_w(0, 0); stage_w(0, 1); stage_a(0, 1);
    stage_a(1, 0); stage_w(1, 0); stage_w(1, 1);           // (A rows 64-127 of K-tile 1 follow in P1 of K-tile 0)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // A rows 0-63 and W of K-tile 0

    // fragment read addresses: row = base + 16*i + fr, chunk (kh*4 + kq) ^ (row & 7); row & 7 == fr & 7 (bases are multiples of 16)
    const int sw = fr & 7;
    const int fa_off = wm * HALF + fr * 128;                          // + a*64 rows*128 + i*16*128 ; chunk term added per kh
    const int fw_off = 2 * HALF + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;
    const int ch0 = ((0 + kq) ^ sw) << 4, ch1 = ((4 + kq) ^ sw) << 4;

    bf16x8 fa[4][2], fw0a[2][2], fw0b[2][2], fw1[2][2];      // two W0 sets: K-tile t+1's is read while K-tile t's still multiplies
    auto read_a = [&](const char* buf, int a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* rp = buf + fa_off + (a * 64 + i * 16) * 128;
            fa[i][0] = *(const bf16x8*)(rp + ch0);
            fa[i][1] = *(const bf16x8*)(rp + ch1);
        }
    };
    auto read_w = [&](const char* buf, int b, bf16x8 (*fw)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* rp = buf + fw_off + (b * 32 + j * 16) * 128;
            fw[j][0] = *(const bf16x8*)(rp + ch0);
            fw[j][1] = *(const bf16x8*)(rp + ch1);
        }
    };
    // ragged last row-tile (M = 64 * 257 leaves 64 valid rows of 256): a 64-row quadrant entirely past M keeps its zero
    // accumulators and skips its MFMAs (wave-uniform), so that tile costs its loads and barriers only
    const bool rows_live[2] = {m0 + wm * 128 < p.M, m0 + wm * 128 + 64 < p.M};
    auto mma = [&](int a, int b, const bf16x8 (*fw)[2]) {
        if (!rows_live[a]) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[a * 4 + i][b * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][kh], fa[i][kh], acc[a * 4 + i][b * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    MEDP_BAR();                    // K-tile 0 (A rows 0-63, W) visible to everyone
    read_w(smem, 0, fw0a);
    if (wm == 1) MEDP_BAR();       // group 1 runs one barrier behind (group 0 pays its extra barrier after the loop)

    auto ktile = [&](int kt, const bf16x8 (*fw0)[2], bf16x8 (*fw0n)[2]) {
        const char* buf = smem + (kt & 1) * KBUF;
        // ---- P1
        read_a(buf, 0);
        stage_a(kt + 1, 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), vmcnt/expcnt untouched
        MEDP_BAR();
        mma(0, 0, fw0);
        MEDP_BAR();
        // ---- P2
        read_w(buf, 1, fw1);
        stage_a(kt + 2, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // A rows 64-127 of K-tile kt (issued in P1(kt-1)) have landed
        MEDP_BAR();
        mma(0, 1, fw1);
        MEDP_BAR();
        // ---- P3
        read_a(buf, 1);
        stage_w(kt + 2, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // W (and A rows 0-63) of K-tile kt+1 (issued in P2..P4(kt-1)) have landed
        MEDP_BAR();
        mma(1, 1, fw1);
        MEDP_BAR();
        // ---- P4
        read_w(smem + ((kt + 1) & 1) * KBUF, 0, fw0n);
        stage_w(kt + 2, 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        MEDP_BAR();
        mma(1, 0, fw0);
        MEDP_BAR();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        ktile(kt, fw0a, fw0b);
        if (kt + 1 < nkt) ktile(kt + 1, fw0b, fw0a);
    }
    if (wm == 0) MEDP_BAR();

    // ---- epilogue through LDS (whole-row coalesced global accesses), as v3 ------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* wl = (float*)(smem + wave * (64 * 68 * 4));
    const int er = lane >> 4, ec = (lane & 15) * 4;
    const int n = n0 + wn * 64 + ec;
    f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f}, scale4 = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (n < p.N) {
        if (p.bias) bias4 = *(const f32x4*)(p.bias + n);
        if (p.scale) scale4 = *(const f32x4*)(p.scale + n);
    }
    // The fp32 residual tile (proj / fc2: 256 KiB per workgroup) is fetched into registers BEFORE the accumulators go through
    // LDS — 16 independent 16-B loads per lane and half, all in flight at once (the fragment registers are free now).  Loading
    // it inside the store loop made the epilogue a chain of exposed HBM round trips: 40 us of the 61-us proj GEMM.
    f32x4 res[2][16];
    auto load_res = [&](int half) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + wm * 128 + half * 64 + it * 4 + er;
            res[half][it] = (m < p.M && n < p.N) ? *(const f32x4*)(p.residual + (size_t)m * p.ldr + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const bool has_res = p.residual != nullptr;
    if (has_res) load_res(0);
    // LayerNorm fold (gemm_variants.h).  Consumer: the row's mean / rstd from the producer's per-tile partial sums, summed in tile order;
    // C = rstd acc - rstd mean colsum + bias.  Producer: bf16 copy of the fp32 result + this tile's (sum, sum of squares) per row.
    const MedpGemmFold& fo = p.fold;
    const bool consumer = fo.stats_in != nullptr, producer = fo.c2 != nullptr;
    f32x4 cs4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (consumer && n < p.N) cs4 = *(const f32x4*)(fo.colsum + n);
    const float inv_dim = consumer ? 1.0f / (float)fo.ln_dim : 0.f;
    float* sst = (float*)(smem + LDS_STATS);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(wl + (i4 * 16 + fr) * 68 + j * 16 + kq * 4) = acc[half * 4 + i4][j];
        MEDP_WAVE_LDS_SYNC();
        if (half == 0 && has_res) load_res(1);      // second half's residual flies while the first half is stored
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rr = it * 4 + er;
            const int m = m0 + wm * 128 + half * 64 + rr;
            f32x4 v = *(const f32x4*)(wl + rr * 68 + ec);
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M && n < p.N) {
                if (consumer) {
                    float a1 = 0.f, a2 = 0.f;
                    for (int t = 0; t < fo.stats_tiles; ++t) {
                        const float2 st = *(const float2*)(fo.stats_in + ((size_t)m * fo.stats_tiles + t) * 2);
                        a1 += st.x;
                        a2 += st.y;
                    }
                    const float mean = a1 * inv_dim;
                    const float rstd = rsqrtf(fmaxf(a2 * inv_dim - mean * mean, 0.f) + fo.ln_eps);
                    v = v * rstd - (mean * rstd) * cs4 + bias4;
                } else {
                    v += bias4;
                }
                if (p.act == 1) v = p.out_bf16 ? gelu_bf16_4(v) : gelu_erf4(v);
                v *= scale4;
                if (has_res) v += res[half][it];
                if (p.out_bf16) {
                    uint2 o;
                    o.x = pack_bf2(v[0], v[1]);
                    o.y = pack_bf2(v[2], v[3]);
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    // non-temporal: the tile is written once and all 256 workgroups flush together (-2 % step time measured)
                    __builtin_nontemporal_store((u32x2){o.x, o.y}, (u32x2*)((bf16_t*)p.C + (size_t)m * p.ldc + n));
                } else {
                    __builtin_nontemporal_store(v, (f32x4*)((float*)p.C + (size_t)m * p.ldc + n));
                }
                if (producer) {
                    uint2 o;
                    o.x = pack_bf2(v[0], v[1]);
                    o.y = pack_bf2(v[2], v[3]);
                    *(uint2*)((bf16_t*)fo.c2 + (size_t)m * fo.ldc2 + n) = o;          // read next by the consumer GEMM: default cache policy
                    s1 = (v[0] + v[1]) + (v[2] + v[3]);
                    s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
            if (producer) {                         // the row's 64 columns of this wave: butterfly over the 16 lanes (one DPP row) that share it
                // DPP operands (VALU rate), not __shfl_xor (ds_bpermute: an LDS round trip per step): quad xor 1, quad xor 2, then the two
                // mirrors — after the quad steps all four lanes of a quad agree, so mirroring pairs quads / halves
                auto row16_sum = [](float x) {
                    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
                    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
                    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));    // row_half_mirror
                    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));    // row_mirror
                    return x;
                };
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if ((lane & 15) == 0) *(float2*)(sst + ((wm * 128 + half * 64 + rr) * 4 + wn) * 2) = make_float2(s1, s2);
            }
        }
        MEDP_WAVE_LDS_SYNC();
    }
    if (producer) {                                 // the four column waves' partials of a row, added in wave order: one (sum, sum sq) per row and tile
        __syncthreads();
        if (tid < 256 && m0 + tid < ((p.M + 255) & ~255)) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 st = *(const float2*)(sst + (tid * 4 + q) * 2);
                a1 += st.x;
                a2 += st.y;
            }
            *(float2*)(fo.stats_out + ((size_t)(m0 + tid) * tiles_n + n0 / BN) * 2) = make_float2(a1, a2);
        }
    }
    if (slot && tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(slot + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gridDim.x - 1)
            for (int i = 0; i < 9; ++i) __hip_atomic_store(slot + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    MEDP_PROF_LEAVE(p.prof, p.prof_flags);
}

template <int TAG>
int launch_v6(const MedpGemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)gemm_bf16_nt_v6_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    // MEDP_V6_TICKETS=1: one-round grids launch 256 workgroups that queue for the tiles (see the kernel).  OFF by default: the
    // hypothesis it tests — proj / fc2 run 30-40 % slower inside the step because workgroups start late on CUs still holding the
    // other branch's short kernels — did not hold (in-box A/B: proj 41.9 us static, 44.2 us queued; step 5.03 vs 5.12 ms; alone the
    // ticket costs 1.2 us): what the other branch takes from these GEMMs is cache and memory bandwidth, not CU slots.
    static const int tickets_on = [] { const char* e = getenv("MEDP_V6_TICKETS"); return e ? atoi(e) : 0; }();
    unsigned* slot = (tickets_on && tiles >= 64 && tiles < 256) ? medp_gemm_ticket_block(stream) : nullptr;
    gemm_bf16_nt_v6_kernel<TAG><<<slot ? 256 : tiles, 512, LDS_BYTES, stream>>>(a, slot);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v6)");
    return 0;
}

}  // namespace

int medp_gemm_v6_launch(const MedpGemmArgs& a, int tag, void* stream) {
    return tag == 1 ? launch_v6<1>(a, (hipStream_t)stream) : launch_v6<0>(a, (hipStream_t)stream);
}
