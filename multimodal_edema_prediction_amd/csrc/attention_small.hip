// Small dense attention, forward + backward, fp32 math (gfx950).
// Used where the attention core is a negligible share of the work and the shapes are MFMA-hostile:
//   * DuETT event/time encoders: 2 heads, head dim 12, 49 / 97 tokens (x_transformers Encoder, no mask)
//   * perceiver blocks: 4 heads, head dim 64, 7 pathology queries over 256 patches / 96 hours / 7 latents
// One workgroup per (batch, head); a wave per query row; keys on lanes for QK^T and softmax (wavefront
// reductions), head-dim on lanes for PV.  Backward recomputes the probabilities, writes dQ per query and
// accumulates dK/dV in the block's own (batch, head) slice — no cross-block atomics, bitwise reproducible.
// Dropout on the probabilities uses the counter hash of common.h, regenerated in backward.
#include "common.h"
#include "medp_hip.h"

namespace {

constexpr int MAXK_PER_LANE = 24;   // Lk <= 1536

// LDS hand-off between lanes of ONE wave (waves run different trip counts, so no block barrier here)
#define WAVE_LDS_SYNC()                                   \
    do {                                                  \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                  \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

struct SmallAttnParams {
    const float *q, *k, *v;
    int ldq, ldk, ldv;
    long long q_bs, kv_bs;   // batch strides in elements (rows of one batch need not abut the next batch's)
    int B, Lq, Lk, H, dh;
    float scale, drop_p, inv_keep;
    uint32_t seed, stream_id;
    const uint32_t* epoch;
};

// <a, b> over dh floats: a in LDS (same address on every lane), b = this lane's own K / V row in global memory.  Rows are
// contiguous, so the vector form reads them as float4 (4x fewer load instructions on what is a latency-bound kernel).
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const float* __restrict__ b, int dh, bool vec) {
    float acc = 0.f;
    if (vec) {
        for (int d = 0; d < dh; d += 4) {
            const float4 x = *(const float4*)(a + d), y = *(const float4*)(b + d);
            acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
    } else {
        for (int d = 0; d < dh; ++d) acc += a[d] * b[d];
    }
    return acc;
}

// acc = sum_j w[j] * col[j * ld]  for j = 0 .. n-1, summed in ascending j (one chain: same rounding as the plain loop), with the
// global loads issued EIGHT at a time.  The plain loop kept one load in flight per wave: 257 keys x an L2 round trip each was
// most of the 149-us perceiver backward and of the 105-us DuETT forward.
__device__ __forceinline__ float weighted_col_sum(const float* __restrict__ w, const float* __restrict__ col, size_t ld, int n) {
    float acc = 0.f;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(j + u) * ld];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[j + u] * v[u];
    }
    for (; j < n; ++j) acc += w[j] * col[(size_t)j * ld];
    return acc;
}

// scores + softmax for one query row; returns p (post-softmax, pre-dropout) per owned key in pj[], writes nothing
template <int NPER>
__device__ __forceinline__ void row_softmax(const SmallAttnParams& p, const float* qrow, const float* kbase, int lane,
                                            float (&pj)[NPER], bool vec) {
    constexpr int nper = NPER;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < nper; ++i) {
        const int j = lane + 64 * i;
        float s = -INFINITY;
        if (j < p.Lk) {
            s = dot_row(qrow, kbase + (size_t)j * p.ldk, p.dh, vec) * p.scale;
        }
        pj[i] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < nper; ++i) {
        const float e = (lane + 64 * i < p.Lk) ? __expf(pj[i] - mx) : 0.f;
        pj[i] = e;
        sum += e;
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < nper; ++i) pj[i] *= inv;
}

constexpr int FWD_QCH = 8;     // queries per workgroup of the forward kernel (two per wave)

template <int NPER>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const SmallAttnParams p, void* __restrict__ o, int ldo, int o_bf16,
                                                             float* __restrict__ attn_avg) {
    extern __shared__ float sm[];   // [4][Lk] probabilities, [4][dh] query row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    float* sp = sm + wave * p.Lk;
    float* sq = sm + 4 * p.Lk + wave * p.dh;
    const float* kbase = p.k + (size_t)b * p.kv_bs + h * p.dh;
    const float* vbase = p.v + (size_t)b * p.kv_bs + h * p.dh;
    const bool vec = (((p.dh | p.ldk | p.ldv) & 3) == 0) && ((((uintptr_t)kbase | (uintptr_t)vbase) & 15) == 0);
    constexpr int nper = NPER;
    float pj[NPER];
    // a workgroup takes FWD_QCH queries (grid.y chunks): with all Lq queries in one workgroup a DuETT call (97 queries,
    // 128 (batch, head) pairs) kept 128 CUs busy for 91 us on four waves each
    const int qend = min(p.Lq, ((int)blockIdx.y + 1) * FWD_QCH);
    for (int qi = blockIdx.y * FWD_QCH + wave; qi < qend; qi += 4) {
        const float* qr = p.q + (size_t)b * p.q_bs + (size_t)qi * p.ldq + h * p.dh;
        if (lane < p.dh) sq[lane] = qr[lane];
        WAVE_LDS_SYNC();
        row_softmax<NPER>(p, sq, kbase, lane, pj, vec);
#pragma unroll
        for (int i = 0; i < nper; ++i) {
            const int j = lane + 64 * i;
            if (j < p.Lk) {
                float w = pj[i];
                if (p.drop_p > 0.f) w *= dropout_scale(medp_mix_epoch(p.seed, p.epoch), p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + qi) * p.Lk + j, p.drop_p, p.inv_keep);
                sp[j] = w;
                if (attn_avg) atomicAdd(attn_avg + ((size_t)b * p.Lq + qi) * p.Lk + j, w / (float)p.H);
            }
        }
        WAVE_LDS_SYNC();
        if (lane < p.dh) {
            const float acc = weighted_col_sum(sp, vbase + lane, p.ldv, p.Lk);
            const size_t oi = ((size_t)b * p.Lq + qi) * ldo + h * p.dh + lane;
            if (o_bf16) ((bf16_t*)o)[oi] = f2bf(acc); else ((float*)o)[oi] = acc;
        }
        WAVE_LDS_SYNC();
    }
}

// Backward.  Per chunk of QCH queries: phase 1 (wave per query) recomputes P, forms dS, writes both to LDS and dQ to
// global; phase 2 (thread per (key, d)) adds the chunk's contribution to dK, dV of this (batch, head).
constexpr int QCH = 8;

template <int NPER>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const SmallAttnParams p, const float* __restrict__ dout, int lddo,
                                                             float* __restrict__ dq, int lddq, float* __restrict__ dk, int lddk,
                                                             float* __restrict__ dv, int lddv, long long dkv_bs) {
    extern __shared__ float sm[];
    // layout: P[QCH][Lk], dS[QCH][Lk], qrows[QCH][dh], dorows[QCH][dh]
    float* sP = sm;
    float* sS = sP + QCH * p.Lk;
    float* sQ = sS + QCH * p.Lk;
    float* sDO = sQ + QCH * p.dh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const float* kbase = p.k + (size_t)b * p.kv_bs + h * p.dh;
    const float* vbase = p.v + (size_t)b * p.kv_bs + h * p.dh;
    float* dkbase = dk + (size_t)b * dkv_bs + h * p.dh;
    float* dvbase = dv + (size_t)b * dkv_bs + h * p.dh;
    const bool vec = (((p.dh | p.ldk | p.ldv) & 3) == 0) && ((((uintptr_t)kbase | (uintptr_t)vbase) & 15) == 0);
    constexpr int nper = NPER;
    float pj[NPER];

    for (int c0 = 0; c0 < p.Lq; c0 += QCH) {
        const int nq = min(QCH, p.Lq - c0);
        for (int t = threadIdx.x; t < nq * p.dh; t += 256) {
            const int qi = t / p.dh, d = t % p.dh;
            sQ[qi * p.dh + d] = p.q[(size_t)b * p.q_bs + (size_t)(c0 + qi) * p.ldq + h * p.dh + d];
            sDO[qi * p.dh + d] = dout[((size_t)b * p.Lq + c0 + qi) * lddo + h * p.dh + d];
        }
        __syncthreads();
        for (int ql = wave; ql < nq; ql += 4) {
            const int qi = c0 + ql;
            const float* qr = sQ + ql * p.dh;
            const float* dor = sDO + ql * p.dh;
            row_softmax<NPER>(p, qr, kbase, lane, pj, vec);
            // dP_j = <dO, V_j> * dropmask_j ; delta = sum_j P_j*dropmask_j*... (softmax bwd on the pre-dropout p)
            float dpj[NPER];
            float delta = 0.f;
#pragma unroll
            for (int i = 0; i < nper; ++i) {
                const int j = lane + 64 * i;
                float dp = 0.f, msk = 1.f;
                if (j < p.Lk) {
                    dp = dot_row(dor, vbase + (size_t)j * p.ldv, p.dh, vec);
                    if (p.drop_p > 0.f) msk = dropout_scale(medp_mix_epoch(p.seed, p.epoch), p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + qi) * p.Lk + j, p.drop_p, p.inv_keep);
                    dp *= msk;
                    sP[ql * p.Lk + j] = pj[i] * msk;          // dropped-out weights multiply V in forward
                }
                dpj[i] = dp;
                delta += pj[i] * dp;
            }
            delta = wave_sum(delta);
#pragma unroll
            for (int i = 0; i < nper; ++i) {
                const int j = lane + 64 * i;
                if (j < p.Lk) sS[ql * p.Lk + j] = pj[i] * (dpj[i] - delta) * p.scale;   // dS * scale
            }
            WAVE_LDS_SYNC();
            if (lane < p.dh) {
                dq[((size_t)b * p.Lq + qi) * lddq + h * p.dh + lane] = weighted_col_sum(sS + ql * p.Lk, kbase + lane, p.ldk, p.Lk);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < p.Lk * p.dh; t += 256) {
            const int j = t / p.dh, d = t % p.dh;
            float ak = 0.f, av = 0.f;
            for (int ql = 0; ql < nq; ++ql) {
                ak += sS[ql * p.Lk + j] * sQ[ql * p.dh + d];
                av += sP[ql * p.Lk + j] * sDO[ql * p.dh + d];
            }
            float* pk = dkbase + (size_t)j * lddk + d;
            float* pv = dvbase + (size_t)j * lddv + d;
            if (c0 == 0) { *pk = ak; *pv = av; } else { *pk += ak; *pv += av; }
        }
        __syncthreads();
    }
}

int check(const SmallAttnParams& p) {
    MEDP_CHECK_ARG(p.q && p.k && p.v, "attn_small: null operand");
    MEDP_CHECK_ARG(p.B > 0 && p.Lq > 0 && p.Lk > 0 && p.H > 0 && p.dh > 0, "attn_small: bad shape");
    MEDP_CHECK_ARG(p.dh <= 64, "attn_small: head dim %d > 64", p.dh);
    MEDP_CHECK_ARG(p.Lk <= 64 * MAXK_PER_LANE, "attn_small: Lk %d > %d", p.Lk, 64 * MAXK_PER_LANE);
    MEDP_CHECK_ARG(p.drop_p >= 0.f && p.drop_p < 1.f, "attn_small: dropout p out of range");
    return 0;
}

}  // namespace

extern "C" int medp_attn_small_fwd(const float* q, int ldq, long long q_batch_stride, const float* k, const float* v, int ldkv,
                                   long long kv_batch_stride, void* o, int ldo, int o_bf16, float* attn_avg, int B, int Lq, int Lk,
                                   int H, int dh, float scale, float dropout_p, unsigned seed, unsigned stream_id, void* stream) {
    SmallAttnParams p{q, k, v, ldq, ldkv, ldkv, q_batch_stride, kv_batch_stride, B, Lq, Lk, H, dh, scale, dropout_p, 1.0f / (1.0f - dropout_p), seed, stream_id, medp_rng_epoch_ptr()};
    MEDP_TRY(check(p));
    MEDP_CHECK_ARG(o, "attn_small_fwd: null output");
    const size_t lds = (size_t)(4 * Lk + 4 * dh) * sizeof(float);
    const int nper = (Lk + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * H, (Lq + FWD_QCH - 1) / FWD_QCH);
    if (nper <= 1) attn_small_fwd_kernel<1><<<grid, 256, lds, st>>>(p, o, ldo, o_bf16, attn_avg);
    else if (nper <= 2) attn_small_fwd_kernel<2><<<grid, 256, lds, st>>>(p, o, ldo, o_bf16, attn_avg);
    else if (nper <= 4) attn_small_fwd_kernel<4><<<grid, 256, lds, st>>>(p, o, ldo, o_bf16, attn_avg);
    else if (nper <= 8) attn_small_fwd_kernel<8><<<grid, 256, lds, st>>>(p, o, ldo, o_bf16, attn_avg);
    else attn_small_fwd_kernel<MAXK_PER_LANE><<<grid, 256, lds, st>>>(p, o, ldo, o_bf16, attn_avg);
    MEDP_LAUNCH_CHECK("medp_attn_small_fwd");
    return 0;
}

extern "C" int medp_attn_small_bwd(const float* dout, int lddo, const float* q, int ldq, long long q_batch_stride, const float* k,
                                   const float* v, int ldkv, long long kv_batch_stride, float* dq, int lddq, float* dk, int lddk,
                                   float* dv, int lddkv_unused, long long dkv_batch_stride, int B, int Lq, int Lk, int H, int dh,
                                   float scale, float dropout_p, unsigned seed, unsigned stream_id, void* stream) {
    const int lddv = lddk;
    (void)lddkv_unused;
    SmallAttnParams p{q, k, v, ldq, ldkv, ldkv, q_batch_stride, kv_batch_stride, B, Lq, Lk, H, dh, scale, dropout_p, 1.0f / (1.0f - dropout_p), seed, stream_id, medp_rng_epoch_ptr()};
    MEDP_TRY(check(p));
    MEDP_CHECK_ARG(dout && dq && dk && dv, "attn_small_bwd: null gradient buffer");
    const size_t lds = (size_t)(2 * QCH * Lk + 2 * QCH * dh) * sizeof(float);
    MEDP_CHECK_ARG(lds <= 160 * 1024, "attn_small_bwd: Lk too large for LDS");
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)attn_small_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_small_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_small_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_small_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_small_bwd_kernel<MAXK_PER_LANE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const int nper = (Lk + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
#define MEDP_BWD_ARGS p, dout, lddo, dq, lddq, dk, lddk, dv, lddv, dkv_batch_stride
    if (nper <= 1) attn_small_bwd_kernel<1><<<B * H, 256, lds, st>>>(MEDP_BWD_ARGS);
    else if (nper <= 2) attn_small_bwd_kernel<2><<<B * H, 256, lds, st>>>(MEDP_BWD_ARGS);
    else if (nper <= 4) attn_small_bwd_kernel<4><<<B * H, 256, lds, st>>>(MEDP_BWD_ARGS);
    else if (nper <= 8) attn_small_bwd_kernel<8><<<B * H, 256, lds, st>>>(MEDP_BWD_ARGS);
    else attn_small_bwd_kernel<MAXK_PER_LANE><<<B * H, 256, lds, st>>>(MEDP_BWD_ARGS);
#undef MEDP_BWD_ARGS
    MEDP_LAUNCH_CHECK("medp_attn_small_bwd");
    return 0;
}
