// v7: the v6 K-loop (256 x 256 x 64 tiles, 8 waves, two ping-pong groups, see gemm_bf16_v6.hip) made PERSISTENT for grids of
// more than one round of workgroups with a bf16 result and no residual (qkv: 585 tiles, fc1: 780 tiles on 256 CUs).
//
// 256 workgroups (MEDP_V7_WGS) stay resident and walk the tile list:
//   * the LDS-DMA stream never stops: from K-tile nkt-2 on it carries K-tiles 0 and 1 of the NEXT tile (all but the last A
//     piece, which P1 of the next K-tile 0 issues as in the steady state) — the next K-loop starts two K-tiles deep;
//   * the epilogue needs no K buffer: bias / GELU / bf16 pack happen on the accumulator layout, the transposition to whole
//     128-B rows goes through a 16-row bf16 patch per wave behind the K buffers, the bias slice arrives by LDS-DMA;
//   * nothing waits for the output stores until P2 of the next K-tile 1 (vmcnt is in order: the wait that ends the epilogue is
//     counted so that it stops just before the 16 stores of a full tile, and there is no register-returning load anywhere in the
//     steady state whose wait the compiler would have to place).
// What it bought (tools/trace_gemm_v7.py): the tile walk itself ~2 % on qkv / fc1 and 24 % fewer L2 misses;
// dispatch, first-tile latency and the store drain turned out NOT to be what a round of workgroups pays for — the K-loop is
// bound by L2 -> LDS delivery (L2 channels 79 % busy).
// Tiles: workgroup b takes tile b first (static, so a workgroup that becomes resident late still has work that nobody else
// does) and then draws tickets from the queue of ITS XCD (b % 8): ticket i is tile 256 + 8 i + b % 8, i.e. exactly the tile
// the hardware dispatcher would have sent to that XCD in v6, so the band x super-column L2 locality of the v6 map holds.
// A workgroup that draws a ticket past the end leaves; the last one to leave zeroes the tickets for the next launch.
// The ticket block is private to one launch (host side: a ring for eager launches, never-reused slots under stream capture).
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.h"
#include "gemm_variants.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_v7[4] = {0, 0, 0, 0};

constexpr int SLOT_WORDS = 16;                    // 8 per-XCD ticket counters, 1 exit counter, padding (64 B)
constexpr int RING_SLOTS = 1024, CAPTURE_SLOTS = 15360;
__device__ unsigned g_v7_slots[(RING_SLOTS + CAPTURE_SLOTS) * SLOT_WORDS];   // zero at module load, self-resetting

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int BM = 256, BN = 256, HALF = 128 * 128, KBUF = 4 * HALF;   // 16 KiB half-tile, 64 KiB K-tile buffer
constexpr int PROW = 144;                                              // bytes per patch row: 64 bf16 + 16 B pad (b128 reads stay aligned)
constexpr int PATCH = 16 * PROW;                                       // one wave's 16 x 64 bf16 patch
constexpr int MAILBOX = 2 * KBUF + 8 * PATCH;                          // the next tile index, written by wave 0
constexpr int BIASBUF = MAILBOX + 256;                                 // 2 x 256 floats: bias slice of this / the next tile
constexpr int TRACEBUF = BIASBUF + 2048;                               // debug timeline (medp_dbg_gemm_v7_trace): 8 tiles x 5 x u64
constexpr int STATBUF = TRACEBUF + 512;                                // LayerNorm fold (consumer): this tile's [256 rows][<= 4 tiles][2] raw row sums
constexpr int CSBUF = STATBUF + 256 * 4 * 2 * 4;                       // 2 x 256 floats: colsum(W g) slice of this / the next tile
constexpr int LDS_BYTES = CSBUF + 2048;
static_assert(LDS_BYTES <= 160 * 1024, "K buffers + patches + mailbox exceed the LDS");
constexpr int NWG = 256;                                               // resident workgroups = CUs of an MI355X

// patch write -> read (and read -> next write) inside ONE wave: the LDS executes a wave's operations in order, only the
// compiler must not reorder them.  (A workgroup-scope fence here also emits vmcnt(0): every pass would wait for the global
// stores of the pass before.)
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

// -DMEDP_V7_PHASE_TRACE (tools/build_trace_lib.sh, a separate library): every wave sums, per K-tile phase, the clock ticks
// of {load section, wait at its barrier, MFMA issue, wait at its barrier}; dumped behind the tile timeline.
#ifdef MEDP_V7_PHASE_TRACE
#define PT(k)                                                          \
    do {                                                               \
        const unsigned now_ = (unsigned)__builtin_readcyclecounter();  \
        pt_acc[k] += now_ - pt_last;                                   \
        pt_last = now_;                                                \
    } while (0)
#else
#define PT(k) do { } while (0)
#endif

// FOLD: the LayerNorm-fold CONSUMER epilogue (gemm_variants.h) — its own instantiation, so that the plain kernel's register
// allocation (251 of 256 VGPRs, no spill) is untouched by it
template <int TAG, bool FOLD = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_v7_kernel(const MedpGemmArgs p, unsigned* slot, unsigned long long* trace, const int MB, const int SN) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;          // wm = ping-pong group
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* zero = (const bf16_t*)g_zero16_v7;
    MEDP_PROF_ENTER(p.prof, p.prof_flags);

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int rag = (p.M % BM) ? 1 : 0;
    const int tm_full = tiles_m - rag;
    const int nfull = tm_full * tiles_n;
    const int full8 = nfull & ~7;
    // tile index -> origin: the v6 map (full row-tiles in XCD-contiguous band x super-column order, the ragged row last)
    auto origin = [&](int bid, int& m0, int& n0) {
        if (bid < nfull) {
            const int wg = bid < full8 ? (bid & 7) * (full8 >> 3) + (bid >> 3) : bid;
            const int band = wg / (MB * tiles_n), rb = wg % (MB * tiles_n);
            const int mb = min(MB, tm_full - band * MB);
            const int sc = rb / (mb * SN), r2 = rb % (mb * SN);
            const int sn = min(SN, tiles_n - sc * SN);
            m0 = (band * MB + r2 / sn) * BM;
            n0 = (sc * SN + r2 % sn) * BN;
        } else {
            m0 = tm_full * BM;
            n0 = (bid - nfull) * BN;
        }
    };
    const int nkt = p.K >> 6;                          // host guarantees K % 128 == 0 and K >= 256: nkt even, >= 4
    const int xcd = blockIdx.x & 7;

    int m0, n0;
    origin(blockIdx.x, m0, n0);

    // ---- LDS-DMA staging: half-tile = 128 rows x 8 chunks; lane's two pieces are rows (tid>>3) and (tid>>3)+64 ------
    const int srow = tid >> 3, schunk = (tid & 7) ^ (srow & 7);       // source chunk for LDS position (tid & 7)
    // Source state: the tile origin the A / W stream currently reads from (scalars, switched to the next tile near the end of
    // a K-loop; NOBODY = past M / N: zero source) and ONE 32-bit element offset per operand for this lane's first piece —
    // eight row pointers in VGPRs did not fit next to 128 accumulators once the epilogue sits inside the tile loop.
    // The other pieces are scalar multiples of 64 rows away.  Host checks M*lda, N*ldw < 2^31.
    const int NOBODY = 0x3fffff00;
    int m_src, n_src;
    unsigned a_off, w_off;
    const unsigned lda64 = 64u * (unsigned)p.lda, ldw64 = 64u * (unsigned)p.ldw;
    auto set_m = [&](int mm) { m_src = mm; a_off = (unsigned)(mm + srow) * (unsigned)p.lda + (unsigned)(schunk * 8); };
    auto set_n = [&](int nn) { n_src = nn; w_off = (unsigned)(nn + srow) * (unsigned)p.ldw + (unsigned)(schunk * 8); };
    set_m(m0);
    set_n(n0);
#ifdef MEDP_V7_ABLATE_LOADS
    bool ablate_on = false;          // the prologue still loads (the first tile's waits count what it issued)
#endif
    // piece j (rows 64 j .. 64 j + 63) of half-tile `which` (0: A rows 0-127, 1: A rows 128-255, 2: W rows 0-127, 3: W rows
    // 128-255) of the K-tile at element offset k0 of the CURRENT source tile -> K buffer `b`
    auto stage_piece = [&](int k0, int b, int which, int j) {
#ifdef MEDP_V7_ABLATE_LOADS      // timing-only build (tools/ablate_gemm_v7.py): the K-loop without its staging stream (wrong results)
        if (ablate_on) return;
#endif
        char* dst = smem + b * KBUF + which * HALF + wave * 1024 + j * 8192;
        const bool kin = k0 + schunk * 8 < p.K;
        const int r = (which < 2 ? m_src : n_src) + (which & 1) * 128 + j * 64 + srow;
        const bool ok = kin && r < (which < 2 ? p.M : p.N);
        const unsigned off = (which < 2 ? a_off + ((which & 1) * 2 + j) * lda64 : w_off + ((which & 1) * 2 + j) * ldw64) + (unsigned)k0;
        const bf16_t* live = (which < 2 ? A : W) + off;
        glds16(ok ? live : zero, dst);
    };
    auto stage_a = [&](int k0, int b, int j) { stage_piece(k0, b, 0, j); stage_piece(k0, b, 1, j); };
    auto stage_w = [&](int k0, int b, int h) { stage_piece(k0, b, 2 + h, 0); stage_piece(k0, b, 2 + h, 1); };
    typedef __attribute__((address_space(3))) volatile int lds_int;
    lds_int* mailbox = (lds_int*)(__attribute__((address_space(3))) char*)(smem + MAILBOX);
    // debug timeline: wave 0 stamps the 100-MHz wall clock into LDS at four points of every tile; dumped at exit
    typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64;
    lds_u64* tbuf = (lds_u64*)(__attribute__((address_space(3))) char*)(smem + TRACEBUF);
    int tcount = 0;
    auto stamp = [&](int k, long long tile) {
        if (trace && wave == 0 && tcount < 8) {
            if (k == 0) tbuf[tcount * 5] = (unsigned long long)tile;
            tbuf[tcount * 5 + 1 + k] = wall_clock64();
        }
    };

    f32x4 acc[8][4];

    // fragment read addresses: row = base + 16*i + fr, chunk (kh*4 + kq) ^ (row & 7); row & 7 == fr & 7 (bases are multiples of 16)
    const int sw = fr & 7;
    const int fa_off = wm * HALF + fr * 128;
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
    bool rows_live[2];
    auto mma = [&](int a, int b, const bf16x8 (*fw)[2]) {
        if (!rows_live[a]) return;
#ifdef MEDP_V7_ABLATE_MFMA       // timing-only build: fragment reads kept alive, no matrix instructions (wrong results)
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(fa[i][0]), "v"(fa[i][1])); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { asm volatile("" ::"v"(fw[j][0]), "v"(fw[j][1])); }
        return;
#endif
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

    // ticket for the tile after the one in hand; the result is parked in LDS (wave 0) and read at K-tile nkt-2 of the next
    // K-loop.  Inline asm: the compiler would wait for the returned value where it is drawn.
    unsigned ticket = 0;
    auto draw_ticket = [&]() {
        if (tid == 0) {
            const unsigned one = 1;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(ticket) : "v"(slot + xcd), "v"(one) : "memory");
        }
    };
    auto post_ticket = [&]() {
        if (wave == 0) {
            const int tn = (int)gridDim.x + 8 * (int)ticket + xcd;      // (the grid is a multiple of 8)
            mailbox[lane] = tn < ntiles ? tn : -1;     // lane 0 holds the ticket
        }
    };

    // The bias slice of a tile (256 floats) travels by LDS-DMA as well (one 16-B piece per lane of wave 0, double-buffered,
    // issued between the K-loops for the NEXT tile).  A register-returning load would either be pending at a K-loop back edge
    // (the compiler then retires the whole DMA stream at the top of every K-tile) or be waited for where it is used — and
    // vmcnt is in order: that wait would cover every output store issued before it.
    auto stage_bias = [&](int nn, bool live, int which) {
        if (wave == 0) {
            const int nj = nn + lane * 4;
            const float* src = (live && p.bias && nj < p.N) ? p.bias + nj : (const float*)zero;
            glds16(src, smem + BIASBUF + which * 1024);
        }
    };
    int bias_cur = 0;
    // LayerNorm fold, consumer side (gemm_variants.h): the column sums of W g travel like the bias (wave 1); the producer's raw row sums of
    // the tile's 256 rows (stats_in is padded to whole tiles) are one flat copy of 256 x tiles x 8 bytes, 1 KB per wave.  Single-buffered:
    // issued right behind a tile-top barrier — every wave has then left the previous epilogue, the only reader — and retired long before
    // this tile's epilogue (the K-loop's first counted wait, P2 of K-tile 1, leaves only the 10 youngest pieces in flight: 12 follow it).
    const MedpGemmFold& fo = p.fold;
    constexpr bool consumer = FOLD;
    // (`opaque`: lane-constant address parts computed where they are used — hoisted out of the tile loop they stay live through the
    //  K-loop, which has 5 VGPRs to spare, and are spilled)
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    auto stage_cs = [&](int nn, bool live, int which) {
        if (consumer && wave == 1) {
            const int nj = nn + opaque(lane) * 4;
            const float* src = (live && nj < p.N) ? fo.colsum + nj : (const float*)zero;
            glds16(src, smem + CSBUF + which * 1024);
        }
    };
    auto stage_stats = [&](int mm) {
        if (consumer && wave * 1024 < 256 * fo.stats_tiles * 8)
            glds16((const char*)fo.stats_in + (size_t)mm * fo.stats_tiles * 8 + wave * 1024 + opaque(lane) * 16, smem + STATBUF + wave * 1024);
    };

    // prologue of the FIRST tile: K-tile 0 and K-tile 1 complete — the state every later tile starts from
    stage_bias(n0, true, 0);
    stage_cs(n0, true, 0);
    stage_stats(m0);
    draw_ticket();
    stage_a(0, 0, 0); stage_w(0, 0, 0); stage_w(0, 0, 1); stage_a(0, 0, 1);        // the order of the steady-state stream
    stage_a(64, 1, 0); stage_w(64, 1, 0); stage_w(64, 1, 1);                        // (A rows 64-127 of K-tile 1: P1 of K-tile 0)
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    asm volatile("" : "+v"(ticket)::"memory");
    post_ticket();
    bool first = true;
#ifdef MEDP_V7_ABLATE_LOADS
    ablate_on = true;
#endif

#ifdef MEDP_V7_PHASE_TRACE
    unsigned pt_acc[17], pt_last = 0;
#pragma unroll
    for (int i = 0; i < 17; ++i) pt_acc[i] = 0;
#endif
    for (;;) {
        // ---- tile top: K-tiles 0 and 1 of this tile have landed in every wave; the barrier makes them (and the mailbox) visible
        MEDP_BAR();
        stamp(0, ((long long)m0 << 32) | (unsigned)n0);
        rows_live[0] = m0 + wm * 128 < p.M;
        rows_live[1] = m0 + wm * 128 + 64 < p.M;
        float z;                       // an opaque zero: a known-zero accumulator makes the compiler peel K-tile 0 (code x6, +30 VGPRs)
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){z, z, z, z};
        if (first) read_w(smem, 0, fw0a);      // later tiles: read in P4 of the previous tile's last K-tile
        else stage_stats(m0);                  // (the first tile's were staged by the prologue)
        first = false;
        if (wm == 1) MEDP_BAR();       // group 1 runs one barrier behind (group 0 pays its extra barrier after the loop)

#ifdef MEDP_V7_PHASE_TRACE
        pt_last = (unsigned)__builtin_readcyclecounter();
#endif
        int t_next = -1, m0n = 0, n0n = 0;
        // one K-tile; the phase plan and the waits are those of gemm_bf16_v6.hip.  The stream runs two K-tiles ahead and from
        // K-tile nkt-2 on belongs to the NEXT tile (its K-tiles 0 and 1; nobody: zero source): when the loop ends all of them but
        // the last A piece are issued, and the W0 fragments of the next K-tile 0 are in registers.  Tiles start with those
        // landed, so the waits of K-tile 0 are skipped: the first wait that covers the previous tile's output stores is P2 of
        // K-tile 1.
        auto ktile = [&](int kt, const bf16x8 (*fw0)[2], bf16x8 (*fw0n)[2]) {
            const char* buf = smem + (kt & 1) * KBUF;
            const int k2 = kt + 2 < nkt ? (kt + 2) * 64 : (kt + 2 - nkt) * 64;
            const int k1 = kt + 1 < nkt ? (kt + 1) * 64 : 0;
            const int b2 = kt & 1;
            // ---- P1
            read_a(buf, 0);
            stage_a(k1, b2 ^ 1, 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), vmcnt/expcnt untouched
            PT(0);
            MEDP_BAR();
            PT(1);
            mma(0, 0, fw0);
            PT(2);
            MEDP_BAR();
            PT(3);
            // ---- P2
            if (kt == nkt - 2) {       // from here on the stream reads the next tile (P1 above still staged this tile's last A piece)
                t_next = __builtin_amdgcn_readfirstlane(mailbox[0]);
                if (t_next >= 0) origin(t_next, m0n, n0n);
                set_m(t_next >= 0 ? m0n : NOBODY);
                set_n(t_next >= 0 ? n0n : NOBODY);
            }
            read_w(buf, 1, fw1);
            stage_a(k2, b2, 0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (kt >= 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // A rows 64-127 of K-tile kt (issued in P1(kt-1)) have landed
            PT(4);
            MEDP_BAR();
            PT(5);
            mma(0, 1, fw1);
            PT(6);
            MEDP_BAR();
            PT(7);
            // ---- P3
            read_a(buf, 1);
            stage_w(k2, b2, 0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (kt >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // W (and A rows 0-63) of K-tile kt+1 have landed
            PT(8);
            MEDP_BAR();
            PT(9);
            mma(1, 1, fw1);
            PT(10);
            MEDP_BAR();
            PT(11);
            // ---- P4
            read_w(smem + ((kt + 1) & 1) * KBUF, 0, fw0n);
            stage_w(k2, b2, 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            PT(12);
            MEDP_BAR();
            PT(13);
            mma(1, 0, fw0);
            PT(14);
            MEDP_BAR();
            PT(15);
        };
#pragma clang loop unroll(disable)
        for (int kt = 0; kt < nkt; kt += 2) {
            ktile(kt, fw0a, fw0b);
            ktile(kt + 1, fw0b, fw0a);
        }
        if (wm == 0) MEDP_BAR();

        // ---- between K-loops: K-tiles 0 and 1 of the next tile are in flight; its bias slice and the ticket after it join them
        stamp(1, 0);
        draw_ticket();
        stage_bias(n0n, t_next >= 0, bias_cur ^ 1);
        stage_cs(n0n, t_next >= 0, bias_cur ^ 1);

        // ---- epilogue: bias / GELU / bf16 pack on the accumulator layout (a lane holds 4 consecutive columns of one row), then a
        // 16-row x 64-column bf16 patch per wave through LDS so that every global store instruction writes whole 128-B rows
        f32x4 bj[4];
        if constexpr (!FOLD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bj[j] = *(const f32x4*)(smem + BIASBUF + bias_cur * 1024 + (wn * 64 + j * 16 + kq * 4) * 4);
        }
        char* wl = smem + 2 * KBUF + wave * PATCH;
        const bool full_tile = m0 + BM <= p.M && n0 + BN <= p.N;
        const int prow = lane >> 3, pchunk = lane & 7;
        const int ncol = n0 + wn * 64 + pchunk * 8;
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const float inv_dim = consumer ? 1.0f / (float)fo.ln_dim : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float rstd = 1.f, mr = 0.f;
            if constexpr (consumer) {          // the row's mean / rstd from its per-tile (sum, sum of squares), added in tile order
                const float* sr = (const float*)(smem + STATBUF) + (wm * 128 + i * 16 + opaque(fr)) * fo.stats_tiles * 2;
                float a1 = 0.f, a2 = 0.f;
                for (int t = 0; t < fo.stats_tiles; ++t) {
                    const float2 st = *(const float2*)(sr + 2 * t);
                    a1 += st.x;
                    a2 += st.y;
                }
                const float mean = a1 * inv_dim;
                rstd = rsqrtf(fmaxf(a2 * inv_dim - mean * mean, 0.f) + fo.ln_eps);
                mr = mean * rstd;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
                if constexpr (consumer) {      // (bias and column sums are re-read per element group: 32 registers less than holding them across i)
                    const f32x4 cj = *(const f32x4*)(smem + CSBUF + bias_cur * 1024 + (wn * 64 + j * 16 + kq * 4) * 4);
                    const f32x4 bb = *(const f32x4*)(smem + BIASBUF + bias_cur * 1024 + (wn * 64 + j * 16 + kq * 4) * 4);
                    v = acc[i][j] * rstd - mr * cj + bb;
                } else {
                    v = acc[i][j] + bj[j];
                }
                if (p.act == 1) v = gelu_bf16_4(v);
                *(u32x2*)(wl + fr * PROW + j * 32 + kq * 8) = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
            MEDP_WAVE_LDS_SYNC();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = h * 8 + prow;
                const int m = m0 + wm * 128 + i * 16 + row;
                const u32x4 o = *(const u32x4*)(wl + row * PROW + pchunk * 16);
                if (m < p.M && ncol < p.N) __builtin_nontemporal_store(o, (u32x4*)((bf16_t*)p.C + (size_t)m * p.ldc + ncol));
            }
            MEDP_WAVE_LDS_SYNC();
        }
        // K-tiles 0 and 1 of the next tile (and the ticket) have landed once everything older than the 16 output stores of a
        // full tile has; a clipped tile issues an unknown number of stores and waits for all of them
        stamp(2, 0);
        if (full_tile) __builtin_amdgcn_s_waitcnt(0x4f70);     // vmcnt(16)
        else __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
        asm volatile("" : "+v"(ticket)::"memory");
        stamp(3, 0);
        ++tcount;
        post_ticket();
        bias_cur ^= 1;
        if (t_next < 0) break;
        m0 = m0n;
        n0 = n0n;
    }
    if (trace && wave == 0) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane < 40) trace[(size_t)blockIdx.x * 40 + lane] = lane < tcount * 5 ? tbuf[lane] : 0ull;
    }
#ifdef MEDP_V7_PHASE_TRACE
    if (trace && lane < 16) {
        unsigned v = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v = lane == i ? pt_acc[i] : v;
        trace[256 * 40 + ((size_t)blockIdx.x * 8 + wave) * 16 + lane] = v;
    }
#endif
    // the last workgroup to leave re-arms the ticket block (the next launch on this slot starts after this kernel ends)
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(slot + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(slot + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    MEDP_PROF_LEAVE(p.prof, p.prof_flags);
}

std::atomic<unsigned> g_ring_next{0}, g_capture_next{0};
std::atomic<int> g_api_cap{0};           // medp_gemm_persistent_cap: 0 = default
unsigned long long* g_trace = nullptr;   // debug hook, see medp_dbg_gemm_v7_trace

// A ticket block nobody else is using while the launch it is handed to runs: launches being CAPTURED (replayed for the life of
// their graph) never share one, eager launches take the next of a ring.  nullptr: none left (the caller launches without tickets).
unsigned* ticket_block(hipStream_t stream) {
    static unsigned* slots_of[MEDP_MAX_DEVICES] = {};      // the ticket blocks are a __device__ symbol: one copy per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    MEDP_ONCE_PER_DEVICE({
        unsigned* sp = nullptr;
        if (hipGetSymbolAddress((void**)&sp, HIP_SYMBOL(g_v7_slots)) == hipSuccess) slots_of[dev % MEDP_MAX_DEVICES] = sp;
    });
    unsigned* slots = slots_of[dev % MEDP_MAX_DEVICES];
    if (!slots) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(stream, &cs);
    unsigned s;
    if (cs == hipStreamCaptureStatusActive) {
        s = g_capture_next.fetch_add(1);
        if (s >= (unsigned)CAPTURE_SLOTS) return nullptr;
        s += RING_SLOTS;
    } else {
        s = g_ring_next.fetch_add(1) % RING_SLOTS;
    }
    return slots + (size_t)s * SLOT_WORDS;
}

template <int TAG, bool FOLD>
int launch_v7(const MedpGemmArgs& a, hipStream_t stream) {
    MEDP_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_v7_kernel<TAG, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    });
    unsigned* slot = ticket_block(stream);
    if (!slot) return -1;                                  // out of private blocks: the caller falls back to v6
    // Resident workgroups: the FEWEST (a multiple of 8) that still finish in the same number of rounds as 256 would — qkv (585
    // tiles) and fc1 (780) need 3 and 4 rounds on 256 CUs and equally on 200; the 56 CUs left alone serve the other branches of
    // the step for the whole launch (teacher step 5.39 -> 5.19 ms, and the GEMMs themselves run 4 % faster: fewer L2 clients).
    // MEDP_V7_WGS (a multiple of 8, <= 256) caps the count: the rounds are then counted against the cap.
    static const int env_cap = [] { const char* e = getenv("MEDP_V7_WGS"); const int v = e ? atoi(e) : NWG; return (v >= 8 && v <= NWG) ? (v & ~7) : NWG; }();
    const int api_cap = g_api_cap.load();
    const int cap = api_cap > 0 ? api_cap : env_cap;
    const int ntiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int rounds = (ntiles + cap - 1) / cap;
    static const int fewest = [] { const char* e = getenv("MEDP_V7_FEWEST_WGS"); return e ? atoi(e) : 1; }();      // 0: always `cap` workgroups (A/B)
    const int nwg = fewest ? min(cap, ((ntiles + rounds - 1) / rounds + 7) & ~7) : cap;
    // Tile walk inside an XCD: blocks of MB row-tiles x SN column-tiles (8 x 4 = one round of an XCD's 32 CUs; MEDP_V7_BLOCK=MBxSN
    // overrides).  Measured (FETCH_SIZE per launch, mean of qkv / fc1; in-box step time): 256 workgroups 8x4: 51.1 k KiB, 5.26 ms;
    // 200 workgroups 8x4: 69.2 k, 5.03-5.11 ms (the rounds of 25 per XCD straddle the blocks, the K-loops drift out of step and
    // re-fetch panels); 200 with 5x5: 71.1 k, same time; 192 workgroups 8x3 + the ragged rows as their own launch (every round
    // exactly one block): 48.0 k but 5.25 ms.  HBM traffic is not what binds these GEMMs (142.5 MB algorithmic = 18 us at 8 TB/s
    // against 65-100 us): the fastest arrangement is kept, the extra re-fetch is reported (profiles/traffic.json).
    static const int env_mb = [] { const char* e = getenv("MEDP_V7_BLOCK"); return e ? atoi(e) : 0; }();
    static const int env_sn = [] { const char* e = getenv("MEDP_V7_BLOCK"); const char* x = e ? strchr(e, 'x') : nullptr; return x ? atoi(x + 1) : 0; }();
    int mb = 8, sn = 4;
    if (env_mb > 0 && env_sn > 0) { mb = env_mb; sn = env_sn; }
    gemm_bf16_nt_v7_kernel<TAG, FOLD><<<nwg, 512, LDS_BYTES, stream>>>(a, slot, g_trace, mb, sn);
    MEDP_LAUNCH_CHECK("medp_gemm_bf16_nt(v7)");
    return 0;
}

}  // namespace

bool medp_gemm_v7_eligible(const MedpGemmArgs& a) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    return tiles > NWG && a.out_bf16 && a.residual == nullptr && a.scale == nullptr && a.N % 8 == 0 && a.ldc % 8 == 0 &&
           a.K % 128 == 0 && a.K >= 256 && (long long)a.M * a.lda < (1ll << 31) && (long long)a.N * a.ldw < (1ll << 31);
}

// returns -1 when no private ticket block is left (caller launches v6 instead)
int medp_gemm_v7_launch(const MedpGemmArgs& a, int tag, void* stream) {
    if (a.fold.stats_in) return launch_v7<1, true>(a, (hipStream_t)stream);        // (the fold exists for the tagged encoder GEMMs only)
    return tag == 1 ? launch_v7<1, false>(a, (hipStream_t)stream) : launch_v7<0, false>(a, (hipStream_t)stream);
}

unsigned* medp_gemm_ticket_block(void* stream) { return ticket_block((hipStream_t)stream); }

extern "C" int medp_gemm_persistent_cap(int cap) {
    const int v = cap <= 0 ? 0 : (cap < 8 ? 8 : (cap > NWG ? NWG : (cap & ~7)));
    return g_api_cap.exchange(v);
}

// Debug hook (NOT part of the C ABI in include/medp_hip.h; tools/trace_gemm_v7.py): while `buf` (device memory, 256 x 40 x u64)
// is set, every v7 launch dumps per workgroup and tile {origin, t(tile top), t(K-loop done), t(epilogue issued), t(next tile
// landed)} in 10-ns ticks.  Pass nullptr to switch it off.
extern "C" int medp_dbg_gemm_v7_trace(void* buf) {
    g_trace = (unsigned long long*)buf;
    return 0;
}
