// Flash-style attention forward for head dim 64 on gfx950 (the ViT-B/14 CXR encoder: 12 heads, 257 / 1297 tokens).
//   O[b,s,h,:] = softmax(Q K^T * scale) V,  dense, no mask, bf16 in / bf16 out, fp32 softmax.
//
// One 256-thread workgroup = 4 waves of one (batch, head).  The ceil(S/16) 16-query subtiles are dealt evenly over the
// 4*gridDim.x waves (1..3 subtiles per wave, one pass each): for S = 257 that is two workgroups per (b, h) with one wave
// carrying the odd CLS subtile, instead of a third workgroup that stages all of K/V for a single query.
// K and V for a chunk of up to 320 keys sit in LDS (row-major [key][64], 128-B rows, XOR-swizzled 16-B
// chunks, filled by LDS-DMA) and are shared by the 4 waves.  Per 64-key block a wave computes
//   S^T = K Q^T   (MFMA 16x16x32, A := K rows, B := Q rows -> key on the accumulator rows, query on the lane)
// so a query's scores live on ONE lane column: the online-softmax statistics are per lane, the running
// rescale multiplies whole registers, and the exponentiated tile is ALREADY the B operand of
//   O^T = V^T P^T (A := V^T read with ds_read_b64_tr_b16 from the row-major V image, k order permuted
//                  identically on both operands), so P never touches LDS.
// The block body is straight-line code specialised on the wave's subtile count (a wave-uniform switch) and on
// "full block" vs "ragged tail"; the softmax is max / fma / v_exp_f32 / add per score (scale folded into the fma, bare
// hardware exp2, scalar f32 ops — see key_block), and the 4-lane max butterfly uses v_permlane16/32_swap (VALU) instead
// of ds_bpermute round trips.
#include <stdlib.h>

#include "common.h"
#include "medp_hip.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_attn[4] = {0, 0, 0, 0};

constexpr int KC = 320;   // most keys per LDS chunk (5 blocks of 64)
constexpr int NQ = 3;     // most 16-query subtiles one wave carries

struct AttnParams {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int B, S, H;
    int ldq, ldk, ldv, ldo;
    float scale_log2e;
    int crows;   // LDS rows per K / V image
    float* lse;  // optional [B, H, S]: log2-domain logsumexp of the scaled scores (training: consumed by the backward)
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 lds_tr16(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}

// max over the four lanes {fr, fr+16, fr+32, fr+48} that share a query column
__device__ __forceinline__ float colmax4(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int NQW>
struct WaveState {
    bf16x8 qf[NQW][2];
    f32x4 o[NQW][4];
    float m_run[NQW], l_run[NQW];   // m_run in raw-score units (before the scale)
};

// One block of up to 64 keys (rows kb*64.. of the LDS chunk) against NQW query subtiles.  nvalid = real keys in the block
// (64 unless MASKED).
// DEFER: deferred rescale (the running max moves only when some query of the wave sees a block maximum more than DEFER_LOG2
// above it in exp2 units: p <= 2^DEFER_LOG2 instead of <= 1 — bf16 / fp32 are floating point, the relative precision of P, of its
// row sum and of O is unchanged; the 16 output multiplies and the row-sum multiply of a block are skipped when nothing moved).
constexpr float DEFER_LOG2 = 5.0f;
template <int NQW, bool MASKED, bool DEFER = false>
__device__ __forceinline__ void key_block(WaveState<NQW>& w, const char* sK, const char* sV, int kb, int nvalid, float c, int fr,
                                          int kq) {
    const int ktv = MASKED ? (nvalid + 15) >> 4 : 4;   // 16-key tiles holding a real key (wave-uniform)
    f32x4 st[NQW][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (MASKED && kt >= ktv) {
#pragma unroll
            for (int qs = 0; qs < NQW; ++qs) st[qs][kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            continue;
        }
        const int krow = kb * 64 + kt * 16 + fr;
        const char* base = sK + krow * 128;
        const bf16x8 k0 = *(const bf16x8*)(base + (((0 + kq) ^ (krow & 7)) << 4));
        const bf16x8 k1 = *(const bf16x8*)(base + (((4 + kq) ^ (krow & 7)) << 4));
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) {
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, w.qf[qs][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, w.qf[qs][1], a, 0, 0, 0);
            st[qs][kt] = a;
        }
    }
    // ---- online softmax: e = exp2(s*c - m*c) -----------------------------------------------------------------------
#pragma unroll
    for (int qs = 0; qs < NQW; ++qs) {
        if (MASKED) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + kq * 4 + r >= nvalid) st[qs][kt][r] = -INFINITY;
        }
        // the softmax is the VALU-bound part of a dh = 64 head (about 2x the MFMA time): chained max -> v_max3_f32 (two
        // scores per instruction), scale folded into one fma per score, bare v_exp_f32
        float mx = fmaxf(st[qs][0][0], st[qs][0][1]);
        mx = fmaxf(fmaxf(mx, st[qs][0][2]), st[qs][0][3]);
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) {
            mx = fmaxf(fmaxf(mx, st[qs][kt][0]), st[qs][kt][1]);
            mx = fmaxf(fmaxf(mx, st[qs][kt][2]), st[qs][kt][3]);
        }
        mx = colmax4(mx);
        bool moved = true;
        if (DEFER) moved = __any((mx - w.m_run[qs]) * c > DEFER_LOG2) != 0;      // wave-uniform
        float mc, alpha = 1.0f;
        if (moved) {
            const float m_new = fmaxf(w.m_run[qs], mx);
            mc = m_new * c;
            alpha = __builtin_amdgcn_exp2f(w.m_run[qs] * c - mc);
            w.m_run[qs] = m_new;
        } else {
            mc = w.m_run[qs] * c;
        }
        // Plain scalar fma / exp / add on purpose.  Written as packed pairs (v_pk_fma_f32 on the MFMA results, v_pk_add_f32
        // on the fresh v_exp_f32 results) the same arithmetic was NOT bit-stable once other kernels shared the SIMD: about
        // 1 % of launches returned a 16-query subtile off by ~1e-2 under the two-stream training step, none when the
        // kernel ran alone.  The ISA of both builds was audited (DESIGN.md "Bit stability", tools/isa_hazard_audit.py): the
        // LDS-DMA wait precedes every staging barrier in BOTH, every software-managed dependency distance is the same in both
        // except the packed build's own pairs (MFMA -> v_pk_fma 31 wait states, v_pk_fma -> v_exp 4), all beyond the ISA's
        // stated minima — so the cause is not a missing wait the source can add; the scalar form stays (it costs no time).
        // tests/test_gpu_bit_stability.py is the screen for it.
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(st[qs][kt][r], c, -mc));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(st[qs][kt][r + 1], c, -mc));
                st[qs][kt][r] = e0;
                st[qs][kt][r + 1] = e1;
                ls += e0;
                ls += e1;
            }
        if (moved) {
            w.l_run[qs] = w.l_run[qs] * alpha + ls;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) w.o[qs][dt] *= alpha;
        } else {
            w.l_run[qs] += ls;
        }
    }
    // ---- O^T += V^T P^T ----------------------------------------------------------------------------------------------
    const int tr_q = fr >> 2, tr_p = fr & 3;   // tr-read lane geometry: lane 4*qq+pp of a 16-lane group -> row qq, columns 4pp..4pp+3
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (MASKED && 2 * ks >= ktv) continue;
        bf16x8 pf[NQW];
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) {
            const f32x4 a = st[qs][2 * ks], b = st[qs][2 * ks + 1];
            union { bf16x8 v; uint32_t u[4]; } pk;
            pk.u[0] = pack_bf2(a[0], a[1]);
            pk.u[1] = pack_bf2(a[2], a[3]);
            pk.u[2] = pack_bf2(b[0], b[1]);
            pk.u[3] = pack_bf2(b[2], b[3]);
            pf[qs] = pk.v;
        }
        // k slot j of lane group kq: j<4 -> key base0 + 4kq + j ; j>=4 -> key base1 + 4kq + (j-4)
        const int key0 = kb * 64 + (2 * ks) * 16 + kq * 4 + tr_q;
        const int key1 = key0 + 16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int chunk = dt * 2 + (tr_p >> 1), off = (tr_p & 1) * 8;
            const bf16x4 v0 = lds_tr16(sV + key0 * 128 + ((chunk ^ (key0 & 7)) << 4) + off);
            const bf16x4 v1 = lds_tr16(sV + key1 * 128 + ((chunk ^ (key1 & 7)) << 4) + off);
            const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int qs = 0; qs < NQW; ++qs) w.o[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qs], w.o[qs][dt], 0, 0, 0);
        }
    }
}

// Everything one wave does, specialised on its subtile count: the four waves of a workgroup may run different
// instantiations (a wave-uniform switch in the kernel), but each executes the same sequence of workgroup barriers.
template <int NQW, int NT = 256, bool DEFER = false>
__device__ __forceinline__ void wave_body(const AttnParams& p, char* sK, char* sV, int q0, int b, int h, int tid, int wave) {
    const int lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int niter = (p.crows * 8 + NT - 1) / NT;              // 16-B pieces per thread and operand: crows rows x 8 chunks
    const bf16_t* zero = (const bf16_t*)g_zero16_attn;
    const size_t row0 = (size_t)b * p.S;

    WaveState<(NQW > 0 ? NQW : 1)> w;
    if constexpr (NQW > 0) {
        // ---- Q fragments (B operand: lane = query, 8 consecutive d) --------------------------------
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) {
            const int qi = q0 + qs * 16 + fr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16_t* src = qi < p.S ? p.q + (row0 + qi) * p.ldq + h * 64 + s * 32 + kq * 8 : zero;
                w.qf[qs][s] = *(const bf16x8*)src;
            }
            w.m_run[qs] = -INFINITY;
            w.l_run[qs] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) w.o[qs][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }

    for (int c0 = 0; c0 < p.S; c0 += KC) {
        const int nkeys = min(KC, p.S - c0);
        if (c0 > 0) __syncthreads();   // previous chunk fully consumed
        // ---- stage K, V chunk (LDS-DMA, swizzle on the source chunk index) -----------------
        for (int i = 0; i < niter; ++i) {
            const int qd = i * NT + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            if (NT > 256 && row >= p.crows) break;                // (wave-uniform: a wave's 64 pieces are 8 whole rows, crows % 32 == 0)
            const bool ok = row < nkeys;
            const size_t grow = row0 + c0 + row;
            glds16(ok ? p.k + grow * p.ldk + h * 64 + c * 8 : zero, sK + (i * NT + wave * 64) * 16);
            glds16(ok ? p.v + grow * p.ldv + h * 64 + c * 8 : zero, sV + (i * NT + wave * 64) * 16);
        }
        MEDP_WAIT_LDS_DMA();           // the chunk's LDS-DMA has landed ...
        __syncthreads();               // ... before any wave reads it
        if constexpr (NQW > 0) {
            const int nfull = nkeys >> 6;
            for (int kb = 0; kb < nfull; ++kb) key_block<NQW, false, DEFER>(w, sK, sV, kb, 64, p.scale_log2e, fr, kq);
            if (nkeys & 63) key_block<NQW, true, DEFER>(w, sK, sV, nfull, nkeys & 63, p.scale_log2e, fr, kq);
        }
    }

    // ---- normalise and store: lane holds O[q = fr][d = dt*16 + kq*4 .. +3] --------------------------
    if constexpr (NQW > 0) {
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) {
            float l = w.l_run[qs];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            const int qi = q0 + qs * 16 + fr;
            if (p.lse && kq == 0 && qi < p.S) p.lse[((size_t)b * p.H + h) * p.S + qi] = w.m_run[qs] * p.scale_log2e + __log2f(l);
            if (qi < p.S) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    uint2 v;
                    v.x = pack_bf2(w.o[qs][dt][0] * inv, w.o[qs][dt][1] * inv);
                    v.y = pack_bf2(w.o[qs][dt][2] * inv, w.o[qs][dt][3] * inv);
                    *(uint2*)(p.o + (row0 + qi) * p.ldo + h * 64 + dt * 16 + kq * 4) = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_dh64_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + p.crows * 128;   // crows = min(KC, S rounded up to 32): 288 rows at S = 257, so TWO workgroups fit a CU
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int ntile = (p.S + 15) >> 4, nwave = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int tbase = ntile / nwave, trem = ntile % nwave;
    const int nq = tbase + (gw < trem ? 1 : 0);                 // subtiles of this wave (wave-uniform, <= NQ)
    const int q0 = (gw * tbase + min(gw, trem)) * 16;
    if (nq == 3) wave_body<3>(p, sK, sV, q0, b, h, tid, wave);
    else if (nq == 2) wave_body<2>(p, sK, sV, q0, b, h, tid, wave);
    else if (nq == 1) wave_body<1>(p, sK, sV, q0, b, h, tid, wave);
    else wave_body<0>(p, sK, sV, q0, b, h, tid, wave);          // idle wave: staging share and barriers only
}

// Long sequences (S >= 512: 512 x 512 images give 1370 tokens): the same body in 512-thread workgroups, at most two subtiles per wave.
// The 4-wave kernel cuts a (b, h) of 86 subtiles into 10 workgroups that EACH stage all of K / V (350 KB) for 8.6 subtiles, and leaves a
// CU 8 waves to hide the softmax chains behind; here 6 workgroups of 8 waves stage it for 14.3 subtiles each and a CU holds 16 waves
// (capped at 128 VGPRs: 10 spilled registers cost less than the second workgroup per CU gains — 268.8 us against 325.1 uncapped at
// B 32, S 1370, H 12; the 4-wave kernel: 292.5; S 1025: 174.5 / 208.7; S 577: 75.9 / 82.4).
__device__ __forceinline__ void w8_body(const AttnParams& p);
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_dh64_w8_kernel(const AttnParams p) { w8_body(p); }
__device__ __forceinline__ void w8_body(const AttnParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + p.crows * 128;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int ntile = (p.S + 15) >> 4, nwave = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
    const int tbase = ntile / nwave, trem = ntile % nwave;
    const int nq = tbase + (gw < trem ? 1 : 0);                 // subtiles of this wave (wave-uniform, <= 2)
    const int q0 = (gw * tbase + min(gw, trem)) * 16;
    if (nq == 2) wave_body<2, 512, true>(p, sK, sV, q0, b, h, tid, wave);
    else if (nq == 1) wave_body<1, 512, true>(p, sK, sV, q0, b, h, tid, wave);
    else wave_body<0, 512, true>(p, sK, sV, q0, b, h, tid, wave);
}

// ---- S = 257 (ViT-B/14 at 224 x 224: the class token + 256 patches): one 512-thread workgroup per (batch, head) ----------------------
// The general kernel above cuts a (b, h) into two workgroups that EACH stage all of K / V, deals 17 query subtiles over 8 waves
// (one wave carries 3, its workgroup waits for it: 17 / 24 of the wave-time used) and pays a masked fifth key block for ONE key.
// Here: K / V are staged once (288-row images, 2 workgroups = 16 waves per CU); the 256 patch queries are 16 subtiles = exactly two
// per wave; the keys are four FULL blocks of 64 (keys 0..255) and key 256 INITIALISES the online softmax (m = its score, l = 1,
// O = its V row: p = exp2(0) = 1 exactly), so no masked block exists; the class query (row 0) is one MFMA column whose 257 keys are
// split over the 8 waves (32 keys each, wave 0 also key 256), partial (m, l, O) combined through 2 KB of LDS by wave 7.
constexpr int S257 = 257, CR257 = 288;
constexpr int SCR_LD = 68;                                   // floats per wave in the class-query scratch: 64 O + m + l (+ pad)
constexpr int LDS_257 = 2 * CR257 * 128 + 8 * SCR_LD * 4;

// One work unit of the S = 257 kernel: the patch-query subtiles t0 .. t0 + NQW - 1 of (b, h) for this wave (NQW = 2: a whole (b, h) per
// workgroup, wave w takes subtiles 2w, 2w+1; NQW = 1: HALF a (b, h), wave w takes subtile 8 half + w) and, when `do_cls`, the class query.
template <int NQW>
__device__ __forceinline__ void s257_unit(const AttnParams& p, char* sK, char* sV, float* scr, int b, int h, int t0, bool do_cls, int tid,
                                          int wave) {
    const int lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const bf16_t* zero = (const bf16_t*)g_zero16_attn;
    const size_t row0 = (size_t)b * S257;
    const float c = p.scale_log2e;

    // ---- loads, in the order they are needed, every wave the SAME number of vector-memory operations (vmcnt retires in order):
    //   2 LDS-DMA pieces: K / V rows 256..287 (key 256 initialises the online softmax; waves 4-7 rewrite the pieces of waves 0-3 with the
    //                     same bytes, so that the counts below hold for every wave)
    //   2 NQW + 2 register loads: the Q fragments (inline asm: beside LDS-DMA hipcc would retire the WHOLE queue at the first use of an
    //                     ordinary load's result — cdna guide §5 trap (b) — and the key blocks below could not start before the last row landed)
    //   8 LDS-DMA pieces: K / V rows 64 i .. 64 i + 63 = key block i, i = 0..3
    // Key block i waits for "all but the 2 (3 - i) youngest" and a raw barrier (every wave has then passed ITS wait): the first block runs
    // while three quarters of K / V are still in flight — the launch reads 76 MB and writes 25 MB, HBM time is what bounds it.
    auto stage_pass = [&](int i, int wsrc) {
        const int row = i * 64 + (wsrc * 8 + (lane >> 3)), ch = (lane & 7) ^ (row & 7);
        const bool ok = row < S257;
        const size_t grow = row0 + row;
        glds16(ok ? p.k + grow * p.ldk + h * 64 + ch * 8 : zero, sK + (i * 512 + wsrc * 64) * 16);
        glds16(ok ? p.v + grow * p.ldv + h * 64 + ch * 8 : zero, sV + (i * 512 + wsrc * 64) * 16);
    };
    stage_pass(4, wave & 3);
    WaveState<NQW> w;
    bf16x8 qc[2];
    auto ld16 = [&](bf16x8& dst, const bf16_t* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) ld16(w.qf[qs][s2], p.q + (row0 + 1 + (t0 + qs) * 16 + fr) * p.ldq + h * 64 + s2 * 32 + kq * 8);
        ld16(qc[s2], (do_cls && fr == 0) ? p.q + row0 * p.ldq + h * 64 + s2 * 32 + kq * 8 : zero);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_pass(i, wave);
    // rows 256..287 and the Q fragments have landed (8 pieces may still fly); the asm names the fragments so that no use moves above it
    if constexpr (NQW == 2)
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(w.qf[0][0]), "+v"(w.qf[0][1]), "+v"(w.qf[1][0]), "+v"(w.qf[1][1]), "+v"(qc[0]), "+v"(qc[1])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(w.qf[0][0]), "+v"(w.qf[0][1]), "+v"(qc[0]), "+v"(qc[1])::"memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const int tr_q = fr >> 2, tr_p = fr & 3;
    auto k_frag = [&](int krow, bf16x8& k0, bf16x8& k1) {
        const char* base = sK + krow * 128;
        k0 = *(const bf16x8*)(base + (((0 + kq) ^ (krow & 7)) << 4));
        k1 = *(const bf16x8*)(base + (((4 + kq) ^ (krow & 7)) << 4));
    };
    // ---- patch queries: the online softmax starts from key 256 (tile 16, row 0 of it) ------------------------------------------------
    {
        bf16x8 k0, k1;
        k_frag(256 + fr, k0, k1);
        const int d0 = kq * 4;                                // V row 256: features 16 dt + 4 kq .. + 3 -> chunk 2 dt + (kq >> 1), half (kq & 1)
#pragma unroll
        for (int qs = 0; qs < NQW; ++qs) {
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, w.qf[qs][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, w.qf[qs][1], a, 0, 0, 0);
            w.m_run[qs] = colmax4(kq == 0 ? a[0] : -INFINITY);
            w.l_run[qs] = kq == 0 ? 1.0f : 0.0f;              // per-lane partial sums, reduced over the 4 lanes of a query at the end
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // (read as a bf16 vector like every other read of the staged images: a `uint2`-typed read made hipcc put s_waitcnt vmcnt(0)
                //  in front of it — it may alias the LDS-DMA in flight — and the key blocks waited for the whole of K / V again)
                const bf16x4 vq = *(const bf16x4*)(sV + 256 * 128 + dt * 32 + d0 * 2);       // row 256: 256 & 7 == 0, no swizzle
                const uint2 vv = __builtin_bit_cast(uint2, vq);
                w.o[qs][dt] = (f32x4){__uint_as_float(vv.x << 16), __uint_as_float(vv.x & 0xffff0000u),
                                      __uint_as_float(vv.y << 16), __uint_as_float(vv.y & 0xffff0000u)};
            }
        }
    }
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {                          // (not unrolled: four copies of the block body cost 40 VGPRs and a wave per SIMD)
        if (kb == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (kb == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kb == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave's pieces of rows 64 kb .. 64 kb + 63 have landed
        __builtin_amdgcn_sched_barrier(0);
        key_block<NQW, false, true>(w, sK, sV, kb, 64, c, fr, kq);
    }

    // ---- normalise and store: lane holds O[q][d = 16 dt + 4 kq .. + 3] ------------------------------------------------------------------
#pragma unroll
    for (int qs = 0; qs < NQW; ++qs) {
        float l = w.l_run[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int qi = 1 + (t0 + qs) * 16 + fr;
        if (p.lse && kq == 0) p.lse[((size_t)b * p.H + h) * S257 + qi] = w.m_run[qs] * c + __log2f(l);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 v;
            v.x = pack_bf2(w.o[qs][dt][0] * inv, w.o[qs][dt][1] * inv);
            v.y = pack_bf2(w.o[qs][dt][2] * inv, w.o[qs][dt][3] * inv);
            *(uint2*)(p.o + (row0 + qi) * p.ldo + h * 64 + dt * 16 + kq * 4) = v;
        }
    }
    // ---- class query (all of K / V has landed by now): this wave's keys 32 wave .. 32 wave + 31 (key tiles 2 wave, 2 wave + 1); wave 0 also key 256 (tile 16) -------
    if (do_cls) {                                             // (workgroup-uniform)
        f32x4 st[3];
        bf16x8 k0, k1;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int kt = t < 2 ? 2 * wave + t : 16;
            if (t == 2 && wave != 0) { st[2] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}; continue; }
            k_frag(kt * 16 + fr, k0, k1);
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qc[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qc[1], a, 0, 0, 0);
            if (t == 2) a = (f32x4){kq == 0 ? a[0] : -INFINITY, -INFINITY, -INFINITY, -INFINITY};      // only key 256 is real
            st[t] = a;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t) mx = fmaxf(fmaxf(fmaxf(mx, st[t][0]), st[t][1]), fmaxf(st[t][2], st[t][3]));
        mx = colmax4(mx);
        const float mc = mx * c;
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c, -mc));      // exp2(-inf) = 0 for the masked slots
                st[t][r] = e;
                ls += e;
            }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        f32x4 oc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1 && wave != 0) break;
            const f32x4 a = st[2 * ks], b2 = ks == 0 ? st[1] : (f32x4){0.f, 0.f, 0.f, 0.f};
            union { bf16x8 v; uint32_t u[4]; } pk;
            pk.u[0] = pack_bf2(a[0], a[1]);
            pk.u[1] = pack_bf2(a[2], a[3]);
            pk.u[2] = pack_bf2(b2[0], b2[1]);
            pk.u[3] = pack_bf2(b2[2], b2[3]);
            const int key0 = (ks == 0 ? 2 * wave : 16) * 16 + kq * 4 + tr_q, key1 = key0 + 16;      // (tile 17 = rows 272..287: zeros)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int chunk = dt * 2 + (tr_p >> 1), off = (tr_p & 1) * 8;
                const bf16x4 v0 = lds_tr16(sV + key0 * 128 + ((chunk ^ (key0 & 7)) << 4) + off);
                const bf16x4 v1 = lds_tr16(sV + key1 * 128 + ((chunk ^ (key1 & 7)) << 4) + off);
                const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                oc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pk.v, oc[dt], 0, 0, 0);
            }
        }
        if (fr == 0) {                                        // column 0 = the class query: O[d = 16 dt + 4 kq + r]
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(scr + wave * SCR_LD + dt * 16 + kq * 4) = oc[dt];
            if (kq == 0) { scr[wave * SCR_LD + 64] = mx; scr[wave * SCR_LD + 65] = ls; }
        }
        __syncthreads();
        if (wave == 7) {                                      // combine the 8 partials: lane = output feature d
            float M = -INFINITY;
#pragma unroll
            for (int i = 0; i < 8; ++i) M = fmaxf(M, scr[i * SCR_LD + 64]);
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float f = __builtin_amdgcn_exp2f((scr[i * SCR_LD + 64] - M) * c);
                L = fmaf(scr[i * SCR_LD + 65], f, L);
                O = fmaf(scr[i * SCR_LD + lane], f, O);
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
            const bf2_t ob = __builtin_convertvector((f32x2){O / L, 0.f}, bf2_t);
            p.o[row0 * p.ldo + h * 64 + lane] = __builtin_bit_cast(uint32_t, ob) & 0xffffu;
            if (p.lse && lane == 0) p.lse[((size_t)b * p.H + h) * S257] = M * c + __log2f(L);
        }
    }

}

// NQW = 2: workgroup i does the whole (b, h) = item0 + i (two subtiles per wave).  NQW = 1: workgroup i does HALF of (b, h) = item0 + i / 2
// (8 query subtiles, one per wave; the even half also the class query), staging K / V for it.  The launcher runs the (b, h) that fill
// whole rounds of the 2-per-CU resident slots as <2> and a remainder of at most half a round as <1> halves: B H = 768 on 512 slots is one
// round of whole units + one round of half units (1.55 unit-times) instead of a second round that leaves every second slot idle (2).
template <int NQW>
__global__ __launch_bounds__(512, 2) void attn_fwd_dh64_s257_kernel(const AttnParams p, const int item0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + CR257 * 128;
    float* scr = (float*)(smem + 2 * CR257 * 128);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = item0 + (NQW == 2 ? (int)blockIdx.x : (int)(blockIdx.x >> 1));
    const int half = NQW == 2 ? 0 : (int)(blockIdx.x & 1);
    const int b = item / p.H, h = item - b * p.H;
    s257_unit<NQW>(p, sK, sV, scr, b, h, NQW == 2 ? 2 * wave : 8 * half + wave, half == 0, tid, wave);
}

// Measured and removed (round 3, git history: "attention: persistent 16-wave S=257 kernel"): ONE 1024-thread workgroup per CU walking its
// three (b, h) with K / V double-buffered and one subtile per wave — 37.8-39.1 us against 32.6 for the kernels above.  A wave's chain
// (init, four key blocks, class-query share, two barriers) is latency-bound; two subtiles per wave give it two independent chains to
// interleave, one subtile per wave does not, and sixteen such waves per CU do not make up for it.

}  // namespace

static int g_s257_slots_override = 0;       // medp_dbg_attn_s257_slots (tests): pretend the device has this many resident slots

static int attn_fwd_launch(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H, int ldq, int ldk,
                           int ldv, int ldo, float scale, void* stream) {
    MEDP_CHECK_ARG(q && k && v && o, "attn_fwd_dh64: null operand");
    MEDP_CHECK_ARG(B > 0 && S > 0 && H > 0, "attn_fwd_dh64: bad shape B=%d S=%d H=%d", B, S, H);
    MEDP_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd_dh64: row strides must keep 16-B alignment");
    MEDP_CHECK_ARG(B <= 65535 && H <= 65535, "attn_fwd_dh64: grid limit");
    MEDP_CHECK_ARG(scale > 0.f, "attn_fwd_dh64: scale must be positive (it is folded into the running max)");
    AttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, B, S, H, ldq, ldk, ldv, ldo,
                 scale * 1.4426950408889634f, 0, lse};
    // S = 257 (224 x 224 images): the one-workgroup-per-(batch, head) kernel (MEDP_ATTN_S257=0: the general kernel, for A/B runs)
    static const int s257_on = [] { const char* e = getenv("MEDP_ATTN_S257"); return e ? atoi(e) : 1; }();
    if (S == S257 && s257_on) {
        MEDP_ONCE_PER_DEVICE({
            hipFuncSetAttribute((const void*)attn_fwd_dh64_s257_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_257);
            hipFuncSetAttribute((const void*)attn_fwd_dh64_s257_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_257);
        });
        // resident slots: 2 workgroups per CU (LDS).  Whole rounds of (b, h) as <2>; a remainder of at most half a round as <1> halves
        // (MEDP_ATTN_S257_BALANCE=1 or medp_dbg_attn_s257_slots: the half-unit arrangement, kept for A/B runs and its test)
        static const int dev_slots = [] { int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); return 2 * cus; }();
        const int slots = g_s257_slots_override > 0 ? g_s257_slots_override : dev_slots;
        // Measured (B 64, H 12: 768 (b, h) on 512 slots): whole units only 32.2 us, whole + half units 37.8 us — a (b, h)'s cost is its 99 KB of
        // loads, not its 16 subtiles of arithmetic, and a half unit loads all of K / V for half the queries.  Default: whole units.
        static const int bal = [] { const char* e = getenv("MEDP_ATTN_S257_BALANCE"); return e ? atoi(e) : 0; }();
        const int n = B * H, rem = n % slots;
        const bool halves = bal || g_s257_slots_override > 0;        // (the test hook plans halves on its pretended slots)
        const int n_half = (halves && n > slots && rem > 0 && 2 * rem <= slots) ? rem : 0, n_whole = n - n_half;
        attn_fwd_dh64_s257_kernel<2><<<n_whole, 512, LDS_257, (hipStream_t)stream>>>(p, 0);
        if (n_half) attn_fwd_dh64_s257_kernel<1><<<2 * n_half, 512, LDS_257, (hipStream_t)stream>>>(p, n_whole);
        MEDP_LAUNCH_CHECK("medp_attn_fwd_dh64(S=257)");
        return 0;
    }
    p.crows = min(KC, (S + 31) / 32 * 32);
    constexpr int LDS_MAX = 2 * KC * 128;
    const int LDS = 2 * p.crows * 128;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)attn_fwd_dh64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    });
    const int ntile = (S + 15) / 16;
    static const int w8_min_s = [] { const char* e = getenv("MEDP_ATTN_W8_MIN_S"); return e ? atoi(e) : 512; }();      // 0 / huge: never (A/B runs)
    if (w8_min_s > 0 && S >= w8_min_s) {
        MEDP_ONCE_PER_DEVICE({
            hipFuncSetAttribute((const void*)attn_fwd_dh64_w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        });
        dim3 grid8((ntile + 15) / 16, H, B);                   // ceil: every wave gets 1..2 subtiles (or none in the last workgroup)
        attn_fwd_dh64_w8_kernel<<<grid8, 512, LDS, (hipStream_t)stream>>>(p);
        MEDP_LAUNCH_CHECK("medp_attn_fwd_dh64(8 waves)");
        return 0;
    }
    // floor(ntile/8) workgroups: every wave gets 1..3 subtiles (ntile < 8*(nb+1) <= 12*nb)
    dim3 grid(ntile >= 8 ? ntile / 8 : 1, H, B);
    attn_fwd_dh64_kernel<<<grid, 256, LDS, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_attn_fwd_dh64");
    return 0;
}

extern "C" int medp_attn_fwd_dh64(const void* q, const void* k, const void* v, void* o, int B, int S, int H, int ldq,
                                  int ldk, int ldv, int ldo, float scale, void* stream) {
    return attn_fwd_launch(q, k, v, o, nullptr, B, S, H, ldq, ldk, ldv, ldo, scale, stream);
}

extern "C" int medp_attn_fwd_dh64_lse(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H,
                                      int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
    MEDP_CHECK_ARG(lse, "attn_fwd_dh64_lse: null lse");
    return attn_fwd_launch(q, k, v, o, lse, B, S, H, ldq, ldk, ldv, ldo, scale, stream);
}

// Debug hook (NOT part of the C ABI in include/medp_hip.h; tests/test_gpu_kernels.py): resident-slot count the S = 257 launcher plans
// with (0 = the device's own 2 x CUs), so that the half-unit kernel can be exercised at small B x H.  Returns the previous value.
extern "C" int medp_dbg_attn_s257_slots(int slots) {
    const int prev = g_s257_slots_override;
    g_s257_slots_override = slots > 0 ? slots : 0;
    return prev;
}
