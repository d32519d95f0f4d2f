// Small dense attention, forward + backward, fp32 math (gfx950).
// Used where the attention core is a negligible share of the work and the shapes are MFMA-hostile:
//   * DuETT event/time encoders: 2 heads, head dim 12, 49 / 97 tokens (x_transformers Encoder, no mask)
//   * perceiver blocks: 4 heads, head dim 64, 7 pathology queries over 256 patches / 96 hours / 7 latents
// One workgroup per (batch, head); a wave per query row; keys on lanes for QK^T and softmax (wavefront
// reductions), head-dim on lanes for PV.  Backward recomputes the probabilities, writes dQ per query and
// accumulates dK/dV in the block's own (batch, head) slice — no cross-block atomics, bitwise reproducible.
// Dropout on the probabilities uses the counter hash of common.h, regenerated in backward.
#include <stdlib.h>

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
    // (once per thread: inside the loops the epoch word was a global load per probability — 1132 us instead of ~110 for the
    // 3136 x 97 x 97 forward of DuETT's event axis with dropout on)
    const uint32_t mixed_seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
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
                if (p.drop_p > 0.f) w *= dropout_scale(mixed_seed, p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + qi) * p.Lk + j, p.drop_p, p.inv_keep);
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
    const uint32_t mixed_seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
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
                    if (p.drop_p > 0.f) msk = dropout_scale(mixed_seed, p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + qi) * p.Lk + j, p.drop_p, p.inv_keep);
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

// ---- FEW queries over many keys (the perceiver: 7 pathology queries over 256 patches / 96 hours / 7 latents, 4 heads of 64) -----
// The wave-per-query kernels above walk the keys once per query with one wave: 257 dependent-ish steps for 7 rows of output (40 us
// forward, 91 us backward for 33 MB of K / V — a tenth of what the bytes cost).  Here a workgroup owns one (batch, head) and a
// THREAD owns a key: its K (and in the backward its V) row sits in registers, the <= 8 query rows in LDS (broadcast reads), so all
// scores of a key are 8 x 64 FMAs on one lane; the softmax statistics are two block reductions of 8 values; what needs a sum over
// KEYS (P V in the forward, dS K in the backward) is a second phase with (16-B column piece, key slice) on the threads — whole
// 256-B rows per 16 lanes — and a 16-way reduction through LDS.  fp32 throughout; K and V are read once (the backward reads K
// twice, the second time out of L2), dK / dV rows are written by their key's thread.  Same dropout stream as the kernels above.
constexpr int FQ = 8;            // most queries
constexpr int FQ_T = 256;        // threads = keys per chunk
constexpr int FQ_MAXCH = 4;      // Lk <= 1024

// The compiler would hoist all 8 x 16 broadcast LDS reads of a fully unrolled (d, q) loop nest to the top (512 VGPRs: the backward
// spilled 8 KB per lane); a compiler + scheduler barrier per d step keeps each step's 8 reads next to their 32 FMAs.
#define FQ_KEEP_IN_STEP()                    \
    do {                                     \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// an LDS address the compiler cannot see through: loads from it can neither be hoisted above this point nor merged with
// earlier loads of the same bytes (kept in 512 VGPRs from the score phase to the dK phase otherwise)
__device__ __forceinline__ const float* FQ_OPAQUE(const float* p) {
    asm volatile("" : "+v"(p));
    return p;
}

__device__ __forceinline__ void fq_block_reduce(float (&v)[FQ], float* red, int lane, int wave, bool is_max) {
#pragma unroll
    for (int q = 0; q < FQ; ++q) v[q] = is_max ? wave_max(v[q]) : wave_sum(v[q]);
    __syncthreads();                                   // `red` may still be read from the previous reduction
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < FQ; ++q) red[wave * FQ + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
        const float a = red[q], b = red[FQ + q], c = red[2 * FQ + q], d = red[3 * FQ + q];
        v[q] = is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
    }
}

// scores of this thread's keys (one per chunk) against all queries; returns the softmax probabilities in pr[ch][q]
template <int NCH>
__device__ __forceinline__ void fq_probs(const SmallAttnParams& p, const float* sQ, const float* kbase, float* red, int tid, float (&pr)[NCH][FQ]) {
    const int lane = tid & 63, wave = tid >> 6;
    float mx[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) mx[q] = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch * FQ_T + tid;
        float s[FQ];
#pragma unroll
        for (int q = 0; q < FQ; ++q) s[q] = 0.f;
        if (j < p.Lk) {
            const float4* kr = (const float4*)(kbase + (size_t)j * p.ldk);
            float4 kv[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) kv[d] = kr[d];
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const float* sq = FQ_OPAQUE(sQ + d * 4);
#pragma unroll
                for (int q = 0; q < FQ; ++q) {
                    const float4 x = *(const float4*)(sq + q * 64);
                    s[q] += (x.x * kv[d].x + x.y * kv[d].y) + (x.z * kv[d].z + x.w * kv[d].w);
                }
                FQ_KEEP_IN_STEP();
            }
        }
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            pr[ch][q] = j < p.Lk ? s[q] * p.scale : -INFINITY;
            mx[q] = fmaxf(mx[q], pr[ch][q]);
        }
    }
    fq_block_reduce(mx, red, lane, wave, true);
    float sum[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) sum[q] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            const float e = (ch * FQ_T + tid < p.Lk) ? __expf(pr[ch][q] - mx[q]) : 0.f;
            pr[ch][q] = e;
            sum[q] += e;
        }
    fq_block_reduce(sum, red, lane, wave, false);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int q = 0; q < FQ; ++q) pr[ch][q] *= 1.0f / sum[q];
}

// out[q][0..63] = sum_j w[q][j] * rows[j][0..63]: thread = (16-B piece c of the row, key slice ks of 16); partial sums meet in LDS
__device__ __forceinline__ void fq_weighted_rows(const float* sW, int ldw, const float* rows, size_t ld, int Lk, int Lq, float* sR, int tid,
                                                 float* out, size_t ldo) {
    const int c = tid & 15, ks = tid >> 4;
    float4 acc[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = ks; j0 < Lk; j0 += 16 * 4) {
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = (j0 + 16 * u < Lk) ? *(const float4*)(rows + (size_t)(j0 + 16 * u) * ld + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 16 * u, Lk - 1);            // (past the end the row is zero)
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                const float w = sW[q * ldw + j];
                acc[q].x += w * r[u].x; acc[q].y += w * r[u].y; acc[q].z += w * r[u].z; acc[q].w += w * r[u].w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < FQ; ++q) *(float4*)(sR + ((ks * FQ + q) * 64 + c * 4)) = acc[q];
    __syncthreads();
    for (int t = tid; t < Lq * 64; t += FQ_T) {
        const int q = t >> 6, d = t & 63;
        float a = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) a += sR[(k2 * FQ + q) * 64 + d];
        out[(size_t)q * ldo + d] = a;
    }
}

template <int NCH>
__global__ __launch_bounds__(FQ_T) void attn_fq_fwd_kernel(const SmallAttnParams p, void* __restrict__ o, int ldo, int o_bf16) {
    extern __shared__ float sm[];
    float* sQ = sm;                          // [FQ][64]
    float* red = sQ + FQ * 64;               // [4][FQ]
    float* sO = red + 4 * FQ;                // [FQ][64] fp32 result before the store
    float* sR = sO + FQ * 64;                // [16][FQ][64]
    float* sP = sR + 16 * FQ * 64;           // [FQ][LkP]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int LkP = NCH * FQ_T;
    const float* kbase = p.k + (size_t)b * p.kv_bs + h * 64;
    const float* vbase = p.v + (size_t)b * p.kv_bs + h * 64;
    for (int t = tid; t < FQ * 64; t += FQ_T) {
        const int q = t >> 6, d = t & 63;
        sQ[t] = q < p.Lq ? p.q[(size_t)b * p.q_bs + (size_t)q * p.ldq + h * 64 + d] : 0.f;
    }
    __syncthreads();
    float pr[NCH][FQ];
    fq_probs<NCH>(p, sQ, kbase, red, tid, pr);
    const uint32_t seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch * FQ_T + tid;
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            float w = pr[ch][q];
            if (p.drop_p > 0.f && q < p.Lq && j < p.Lk)
                w *= dropout_scale(seed, p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + q) * p.Lk + j, p.drop_p, p.inv_keep);
            sP[q * LkP + j] = (q < p.Lq && j < p.Lk) ? w : 0.f;
        }
    }
    __syncthreads();
    fq_weighted_rows(sP, LkP, vbase, p.ldv, p.Lk, p.Lq, sR, tid, sO, 64);
    __syncthreads();
    for (int t = tid; t < p.Lq * 64; t += FQ_T) {
        const int q = t >> 6, d = t & 63;
        const size_t oi = ((size_t)b * p.Lq + q) * ldo + h * 64 + d;
        if (o_bf16) ((bf16_t*)o)[oi] = f2bf(sO[t]); else ((float*)o)[oi] = sO[t];
    }
}

template <int NCH>
__global__ __launch_bounds__(FQ_T) void attn_fq_bwd_kernel(const SmallAttnParams p, const float* __restrict__ dout, int lddo,
                                                           float* __restrict__ dq, int lddq, float* __restrict__ dk, int lddk,
                                                           float* __restrict__ dv, int lddv, long long dkv_bs) {
    extern __shared__ float sm[];
    float* sQ = sm;                          // [FQ][64]
    float* sDO = sQ + FQ * 64;               // [FQ][64]
    float* red = sDO + FQ * 64;              // [4][FQ]
    float* sR = red + 4 * FQ;                // [16][FQ][64]
    float* sS = sR + 16 * FQ * 64;           // [FQ][LkP]  dS * scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int LkP = NCH * FQ_T;
    const float* kbase = p.k + (size_t)b * p.kv_bs + h * 64;
    const float* vbase = p.v + (size_t)b * p.kv_bs + h * 64;
    float* dkbase = dk + (size_t)b * dkv_bs + h * 64;
    float* dvbase = dv + (size_t)b * dkv_bs + h * 64;
    for (int t = tid; t < FQ * 64; t += FQ_T) {
        const int q = t >> 6, d = t & 63;
        sQ[t] = q < p.Lq ? p.q[(size_t)b * p.q_bs + (size_t)q * p.ldq + h * 64 + d] : 0.f;
        sDO[t] = q < p.Lq ? dout[((size_t)b * p.Lq + q) * lddo + h * 64 + d] : 0.f;
    }
    __syncthreads();
    float pr[NCH][FQ];
    fq_probs<NCH>(p, sQ, kbase, red, tid, pr);
    const uint32_t seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
    // dP = <dO, V_j> (times the dropout mask), delta = sum_j P dP, and this key's dV row = sum_q (P mask) dO_q
    float dp[NCH][FQ], delta[FQ];
#pragma unroll
    for (int q = 0; q < FQ; ++q) delta[q] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch * FQ_T + tid;
        const bool live = j < p.Lk;
        float pm[FQ], a[FQ];
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            pm[q] = 0.f;                                       // the dropout mask (0 for padding)
            a[q] = 0.f;
        }
        if (live) {                                            // one branch around the whole row: no barrier inside
            float4 vv[16];
            const float4* vr = (const float4*)(vbase + (size_t)j * p.ldv);
#pragma unroll
            for (int d = 0; d < 16; ++d) vv[d] = vr[d];
#pragma unroll
            for (int q = 0; q < FQ; ++q) {
                float msk = 1.f;
                if (p.drop_p > 0.f && q < p.Lq)
                    msk = dropout_scale(seed, p.stream_id, ((uint32_t)(b * p.H + h) * p.Lq + q) * p.Lk + j, p.drop_p, p.inv_keep);
                pm[q] = q < p.Lq ? msk : 0.f;
            }
            float4* dvr = (float4*)(dvbase + (size_t)j * lddv);
#pragma unroll
            for (int d = 0; d < 16; ++d) {                     // one pass over dO's 16-B pieces serves dP (dot with V) and dV (sum over q)
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* sd = FQ_OPAQUE(sDO + d * 4);
#pragma unroll
                for (int q = 0; q < FQ; ++q) {
                    const float4 x = *(const float4*)(sd + q * 64);
                    a[q] += (x.x * vv[d].x + x.y * vv[d].y) + (x.z * vv[d].z + x.w * vv[d].w);
                    const float w = pr[ch][q] * pm[q];
                    acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
                }
                dvr[d] = acc;
                FQ_KEEP_IN_STEP();
            }
        }
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            dp[ch][q] = a[q] * pm[q];
            delta[q] += pr[ch][q] * dp[ch][q];
        }
    }
    fq_block_reduce(delta, red, lane, wave, false);
    // dS (times the scale) -> LDS for the dQ phase; this key's dK row = sum_q dS_q Q_q
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int j = ch * FQ_T + tid;
        const bool live = j < p.Lk;
        float ds[FQ];
#pragma unroll
        for (int q = 0; q < FQ; ++q) {
            ds[q] = (live && q < p.Lq) ? pr[ch][q] * (dp[ch][q] - delta[q]) * p.scale : 0.f;
            sS[q * LkP + j] = ds[q];
        }
        if (live) {
            float4* dkr = (float4*)(dkbase + (size_t)j * lddk);
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* sq = FQ_OPAQUE(sQ + d * 4);
#pragma unroll
                for (int q = 0; q < FQ; ++q) {
                    const float4 x = *(const float4*)(sq + q * 64);
                    a.x += ds[q] * x.x; a.y += ds[q] * x.y; a.z += ds[q] * x.z; a.w += ds[q] * x.w;
                }
                dkr[d] = a;
                FQ_KEEP_IN_STEP();
            }
        }
    }
    __syncthreads();
    fq_weighted_rows(sS, LkP, kbase, p.ldk, p.Lk, p.Lq, sR, tid, dq + (size_t)b * p.Lq * lddq + h * 64, lddq);
}

// shapes the few-query kernels take: <= 8 queries, head dim 64, <= 1024 keys, 16-B aligned rows
bool fq_eligible(const SmallAttnParams& p, const void* a, const void* b2, int lda, int ldb) {
    static const int on = [] { const char* e = getenv("MEDP_ATTN_FEWQ"); return e ? atoi(e) : 1; }();
    return on && p.Lq <= FQ && p.dh == 64 && p.Lk <= FQ_MAXCH * FQ_T && ((p.ldk | p.ldv | lda | ldb) & 3) == 0 && (p.kv_bs & 3) == 0 &&
           ((((uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)a | (uintptr_t)b2) & 15) == 0);
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
    hipStream_t st = (hipStream_t)stream;
    if (!attn_avg && fq_eligible(p, k, v, 4, 4)) {
        const int nch = (Lk + FQ_T - 1) / FQ_T;
        const int nchp = nch <= 1 ? 1 : (nch <= 2 ? 2 : 4);
        const size_t fl = (size_t)(2 * FQ * 64 + 4 * FQ + 16 * FQ * 64 + FQ * nchp * FQ_T) * sizeof(float);
        MEDP_ONCE_PER_DEVICE({
            hipFuncSetAttribute((const void*)attn_fq_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            hipFuncSetAttribute((const void*)attn_fq_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            hipFuncSetAttribute((const void*)attn_fq_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        });
        if (nchp == 1) attn_fq_fwd_kernel<1><<<B * H, FQ_T, fl, st>>>(p, o, ldo, o_bf16);
        else if (nchp == 2) attn_fq_fwd_kernel<2><<<B * H, FQ_T, fl, st>>>(p, o, ldo, o_bf16);
        else attn_fq_fwd_kernel<4><<<B * H, FQ_T, fl, st>>>(p, o, ldo, o_bf16);
        MEDP_LAUNCH_CHECK("medp_attn_small_fwd(few queries)");
        return 0;
    }
    const size_t lds = (size_t)(4 * Lk + 4 * dh) * sizeof(float);
    const int nper = (Lk + 63) / 64;
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
    if (fq_eligible(p, dk, dv, lddk, lddv) && (dkv_batch_stride & 3) == 0) {     // both gradient row strides: the kernel stores float4 rows of dK AND dV
        const int nch = (Lk + FQ_T - 1) / FQ_T;
        const int nchp = nch <= 1 ? 1 : (nch <= 2 ? 2 : 4);
        const size_t fl = (size_t)(2 * FQ * 64 + 4 * FQ + 16 * FQ * 64 + FQ * nchp * FQ_T) * sizeof(float);
        MEDP_ONCE_PER_DEVICE({
            hipFuncSetAttribute((const void*)attn_fq_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            hipFuncSetAttribute((const void*)attn_fq_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            hipFuncSetAttribute((const void*)attn_fq_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        });
        hipStream_t fs = (hipStream_t)stream;
        if (nchp == 1) attn_fq_bwd_kernel<1><<<B * H, FQ_T, fl, fs>>>(p, dout, lddo, dq, lddq, dk, lddk, dv, lddv, dkv_batch_stride);
        else if (nchp == 2) attn_fq_bwd_kernel<2><<<B * H, FQ_T, fl, fs>>>(p, dout, lddo, dq, lddq, dk, lddk, dv, lddv, dkv_batch_stride);
        else attn_fq_bwd_kernel<4><<<B * H, FQ_T, fl, fs>>>(p, dout, lddo, dq, lddq, dk, lddk, dv, lddv, dkv_batch_stride);
        MEDP_LAUNCH_CHECK("medp_attn_small_bwd(few queries)");
        return 0;
    }
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
