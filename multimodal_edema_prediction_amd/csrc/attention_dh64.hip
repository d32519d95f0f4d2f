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
template <int NQW, bool MASKED>
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
        const float m_new = fmaxf(w.m_run[qs], mx);
        const float mc = m_new * c;
        const float alpha = __builtin_amdgcn_exp2f(w.m_run[qs] * c - mc);
        w.m_run[qs] = m_new;
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
        w.l_run[qs] = w.l_run[qs] * alpha + ls;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) w.o[qs][dt] *= alpha;
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
template <int NQW>
__device__ __forceinline__ void wave_body(const AttnParams& p, char* sK, char* sV, int q0, int b, int h, int tid, int wave) {
    const int lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int niter = p.crows >> 5;
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
            const int qd = i * 256 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const bool ok = row < nkeys;
            const size_t grow = row0 + c0 + row;
            glds16(ok ? p.k + grow * p.ldk + h * 64 + c * 8 : zero, sK + (i * 256 + wave * 64) * 16);
            glds16(ok ? p.v + grow * p.ldv + h * 64 + c * 8 : zero, sV + (i * 256 + wave * 64) * 16);
        }
        MEDP_WAIT_LDS_DMA();           // the chunk's LDS-DMA has landed ...
        __syncthreads();               // ... before any wave reads it
        if constexpr (NQW > 0) {
            const int nfull = nkeys >> 6;
            for (int kb = 0; kb < nfull; ++kb) key_block<NQW, false>(w, sK, sV, kb, 64, p.scale_log2e, fr, kq);
            if (nkeys & 63) key_block<NQW, true>(w, sK, sV, nfull, nkeys & 63, p.scale_log2e, fr, kq);
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

}  // namespace

static int attn_fwd_launch(const void* q, const void* k, const void* v, void* o, float* lse, int B, int S, int H, int ldq, int ldk,
                           int ldv, int ldo, float scale, void* stream) {
    MEDP_CHECK_ARG(q && k && v && o, "attn_fwd_dh64: null operand");
    MEDP_CHECK_ARG(B > 0 && S > 0 && H > 0, "attn_fwd_dh64: bad shape B=%d S=%d H=%d", B, S, H);
    MEDP_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd_dh64: row strides must keep 16-B alignment");
    MEDP_CHECK_ARG(B <= 65535 && H <= 65535, "attn_fwd_dh64: grid limit");
    MEDP_CHECK_ARG(scale > 0.f, "attn_fwd_dh64: scale must be positive (it is folded into the running max)");
    AttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, B, S, H, ldq, ldk, ldv, ldo,
                 scale * 1.4426950408889634f, 0, lse};
    p.crows = min(KC, (S + 31) / 32 * 32);
    constexpr int LDS_MAX = 2 * KC * 128;
    const int LDS = 2 * p.crows * 128;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)attn_fwd_dh64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    });
    // floor(ntile/8) workgroups: every wave gets 1..3 subtiles (ntile < 8*(nb+1) <= 12*nb)
    const int ntile = (S + 15) / 16;
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
