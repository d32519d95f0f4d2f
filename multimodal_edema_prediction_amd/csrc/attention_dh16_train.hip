// MFMA attention for small head dims (dh <= 16), TRAINING form: DuETT's event / time axis encoders inside the student-KD step
// (2 heads of dim 12 over 49 / 97 tokens; reference duett/duett.py:95-105 through x_transformers' Attention, dropout on the
// probabilities).  The fp32 VALU kernels it replaces (attention_small.hip, a wave per query row) took 143 us per backward and
// 20 us per forward on the time axis.  Same design as attention_dh16.hip — one WAVE per 16-row tile, no LDS, no barrier, operands
// rounded to bf16, softmax / dropout / all sums in fp32 — in three kernels, none of which adds into memory (bitwise reproducible):
//   forward  (wave = 16 queries):  S^T = K Q^T, softmax over the lane's column, dropout, O^T = V^T P^T; keeps the row's
//                                  log2-sum-exp for the backward
//   backward dQ (wave = 16 queries): P^T from the saved log-sum-exp, dP^T = V dO^T, delta = sum_j P dP (also stored),
//                                  dS^T = P (dP - delta) scale, dQ^T = K^T dS^T
//   backward dK, dV (wave = 16 keys): the transposed orientation S = Q K^T per query tile, P and dS from the saved statistics,
//                                  dV += P^T dO, dK += dS^T Q accumulated over the query tiles IN REGISTERS
// MFMA v_mfma_f32_16x16x16_bf16: A lane (row = lane & 15, k = 4 (lane >> 4) ..+3), B lane (col = lane & 15, same k),
// accumulator lane (rows 4 (lane >> 4) + r, col = lane & 15) — an accumulator tile, packed to bf16, is at once a B operand
// indexed (k = its rows, col) and an A operand indexed (row = its col, k = its rows): no transposition through LDS anywhere.
// Dropout: the counter hash of common.h on ((b H + h) N + query) N + key, as attention_small.hip.
#include "common.h"
#include "medp_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4_t;

struct TrainParams {
    const void* qkv;      // rows [B*N][ld]: q | k | v column blocks of H*dh each; fp32, or bf16 in the 16-bit hand-over form (T = bf16_t)
    int ld;
    void* o;              // forward: [B*N][ldo], same element type as qkv
    int ldo;
    float* lse;           // [B*H*N] log2-domain log-sum-exp of the scaled scores
    const void* dout;     // backward: [B*N][lddo], same element type as qkv
    int lddo;
    float* delta;         // [B*H*N] sum_j P dP
    void* dqkv;           // [B*N][lddqkv], same column blocks, same element type
    int lddqkv;
    int B, N, H, dh;
    float scale_log2e, scale, drop_p, inv_keep;
    uint32_t seed, stream_id;
    const uint32_t* epoch;
};

__device__ __forceinline__ bf16x4_t pack4(float a, float b, float c, float d) {
    union { bf16x4_t v; uint32_t u[2]; } p;
    p.u[0] = pack_bf2(a, b);
    p.u[1] = pack_bf2(c, d);
    return p.v;
}
__device__ __forceinline__ bf16x4_t pack4(const f32x4& v) { return pack4(v[0], v[1], v[2], v[3]); }

// row `row` of a [N][ld] block, elements 4 g4 ..+3 (an operand indexed (row | col = lane & 15, k = head dim)).  fp32 storage is rounded to
// bf16 here; bf16 storage (the hand-over form: the producing GEMM rounded with the same f2bf) is loaded as it is — the same operand bits
__device__ __forceinline__ bf16x4_t row_frag(const float* base, int ld, int row, int N, int g4, bool dvalid) {
    if (row >= N || !dvalid) return (bf16x4_t){0, 0, 0, 0};
    const float4 x = *(const float4*)(base + (size_t)row * ld + 4 * g4);
    return pack4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ bf16x4_t row_frag(const bf16_t* base, int ld, int row, int N, int g4, bool dvalid) {
    if (row >= N || !dvalid) return (bf16x4_t){0, 0, 0, 0};
    return *(const bf16x4_t*)(base + (size_t)row * ld + 4 * g4);
}
// column `col` of rows r0 ..+3 (an operand indexed (row | col = head dim lane & 15, k = token))
__device__ __forceinline__ bf16x4_t col_frag(const float* base, int ld, int r0, int N, int col, int dh) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < dh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < N) v[i] = base[(size_t)(r0 + i) * ld + col];
    }
    return pack4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ bf16x4_t col_frag(const bf16_t* base, int ld, int r0, int N, int col, int dh) {
    bf16x4_t v = {0, 0, 0, 0};
    if (col < dh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (r0 + i < N) v[i] = (short)base[(size_t)(r0 + i) * ld + col];
    }
    return v;
}
__device__ __forceinline__ void store4(float* dst, float a, float b, float c, float d) { *(float4*)dst = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4(bf16_t* dst, float a, float b, float c, float d) { *(bf16x4_t*)dst = pack4(a, b, c, d); }
__device__ __forceinline__ void store1(float* dst, float a) { *dst = a; }
__device__ __forceinline__ void store1(bf16_t* dst, float a) { *dst = f2bf(a); }

__device__ __forceinline__ float keep_scale(const TrainParams& p, uint32_t seed, int bh, int q, int key) {
    return dropout_scale(seed, p.stream_id, ((uint32_t)bh * p.N + q) * p.N + key, p.drop_p, p.inv_keep);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)

template <int NT, typename T>
__global__ __launch_bounds__(256) void dh16_train_fwd_kernel(const TrainParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qt = blockIdx.x * 4 + wave;
    if (qt * 16 >= p.N) return;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int c16 = lane & 15, g4 = lane >> 4, D = p.H * p.dh;
    const T* base = (const T*)p.qkv + (size_t)b * p.N * p.ld + h * p.dh;
    const bool dvalid = 4 * g4 < p.dh;
    const int q = qt * 16 + c16;
    const uint32_t seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
    const bf16x4_t qf = row_frag(base, p.ld, q, p.N, g4, dvalid);
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        st[kt] = MFMA16(row_frag(base + D, p.ld, kt * 16 + c16, p.N, g4, dvalid), qf, ((f32x4){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kt * 16 + 4 * g4 + r >= p.N) st[kt][r] = -INFINITY;
            mx = fmaxf(mx, st[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * p.scale_log2e;
    float sum = 0.f;
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(fmaf(st[kt][r], p.scale_log2e, -mc));
            sum += e[r];                                                           // the softmax normalises BEFORE the dropout
            const int key = kt * 16 + 4 * g4 + r;
            if (p.drop_p > 0.f && q < p.N && key < p.N) e[r] *= keep_scale(p, seed, bh, q, key);
        }
        ot = MFMA16(col_frag(base + 2 * D, p.ld, kt * 16 + 4 * g4, p.N, c16, p.dh), pack4(e[0], e[1], e[2], e[3]), ot);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (q < p.N) {
        if (g4 == 0) p.lse[(size_t)bh * p.N + q] = mc + __log2f(sum);
        if (dvalid) {
            const float inv = 1.0f / sum;
            store4((T*)p.o + ((size_t)b * p.N + q) * p.ldo + h * p.dh + 4 * g4, ot[0] * inv, ot[1] * inv, ot[2] * inv, ot[3] * inv);
        }
    }
}

template <int NT, typename T>
__global__ __launch_bounds__(256) void dh16_train_bwd_dq_kernel(const TrainParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qt = blockIdx.x * 4 + wave;
    if (qt * 16 >= p.N) return;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int c16 = lane & 15, g4 = lane >> 4, D = p.H * p.dh;
    const T* base = (const T*)p.qkv + (size_t)b * p.N * p.ld + h * p.dh;
    const T* dob = (const T*)p.dout + (size_t)b * p.N * p.lddo + h * p.dh;
    const bool dvalid = 4 * g4 < p.dh;
    const int q = qt * 16 + c16;
    const uint32_t seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
    const bf16x4_t qf = row_frag(base, p.ld, q, p.N, g4, dvalid), dof = row_frag(dob, p.lddo, q, p.N, g4, dvalid);
    const float lse = q < p.N ? p.lse[(size_t)bh * p.N + q] : 0.f;
    f32x4 pt[NT], dpt[NT];
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const f32x4 s = MFMA16(row_frag(base + D, p.ld, kt * 16 + c16, p.N, g4, dvalid), qf, ((f32x4){0.f, 0.f, 0.f, 0.f}));
        dpt[kt] = MFMA16(row_frag(base + 2 * D, p.ld, kt * 16 + c16, p.N, g4, dvalid), dof, ((f32x4){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g4 + r;
            const bool live = q < p.N && key < p.N;
            pt[kt][r] = live ? __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -lse)) : 0.f;
            if (p.drop_p > 0.f && live) dpt[kt][r] *= keep_scale(p, seed, bh, q, key);
            delta += pt[kt][r] * dpt[kt][r];
        }
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = pt[kt][r] * (dpt[kt][r] - delta) * p.scale;
        dq = MFMA16(col_frag(base + D, p.ld, kt * 16 + 4 * g4, p.N, c16, p.dh), pack4(ds), dq);       // dQ^T += K^T dS^T
    }
    if (q < p.N) {
        if (g4 == 0) p.delta[(size_t)bh * p.N + q] = delta;
        if (dvalid) store4((T*)p.dqkv + ((size_t)b * p.N + q) * p.lddqkv + h * p.dh + 4 * g4, dq[0], dq[1], dq[2], dq[3]);
    }
}

template <int NT, typename T>
__global__ __launch_bounds__(256) void dh16_train_bwd_dkv_kernel(const TrainParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kt = blockIdx.x * 4 + wave;
    if (kt * 16 >= p.N) return;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int c16 = lane & 15, g4 = lane >> 4, D = p.H * p.dh;
    const T* base = (const T*)p.qkv + (size_t)b * p.N * p.ld + h * p.dh;
    const T* dob = (const T*)p.dout + (size_t)b * p.N * p.lddo + h * p.dh;
    const bool dvalid = 4 * g4 < p.dh;
    const int key = kt * 16 + c16;
    const uint32_t seed = p.drop_p > 0.f ? medp_mix_epoch(p.seed, p.epoch) : 0u;
    const bf16x4_t kb = row_frag(base + D, p.ld, key, p.N, g4, dvalid), vb = row_frag(base + 2 * D, p.ld, key, p.N, g4, dvalid);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        if (qt * 16 >= p.N) break;
        const int qrow = qt * 16 + c16;                                             // operand row of this lane
        const f32x4 s = MFMA16(row_frag(base, p.ld, qrow, p.N, g4, dvalid), kb, ((f32x4){0.f, 0.f, 0.f, 0.f}));       // S = Q K^T
        const f32x4 dp = MFMA16(row_frag(dob, p.lddo, qrow, p.N, g4, dvalid), vb, ((f32x4){0.f, 0.f, 0.f, 0.f}));      // dP = dO V^T
        f32x4 pm, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * g4 + r;                                     // accumulator row of this lane
            const bool live = q < p.N && key < p.N;
            const float pr = live ? __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -p.lse[(size_t)bh * p.N + q])) : 0.f;
            const float msk = (p.drop_p > 0.f && live) ? keep_scale(p, seed, bh, q, key) : 1.f;
            pm[r] = pr * msk;
            ds[r] = live ? pr * (dp[r] * msk - p.delta[(size_t)bh * p.N + q]) * p.scale : 0.f;
        }
        dv = MFMA16(pack4(pm), col_frag(dob, p.lddo, qt * 16 + 4 * g4, p.N, c16, p.dh), dv);          // dV += P^T dO
        dk = MFMA16(pack4(ds), col_frag(base, p.ld, qt * 16 + 4 * g4, p.N, c16, p.dh), dk);           // dK += dS^T Q
    }
    // the lane holds keys kt 16 + 4 g4 + r of head dim c16
    if (c16 < p.dh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + 4 * g4 + r;
            if (kk < p.N) {
                T* row = (T*)p.dqkv + ((size_t)b * p.N + kk) * p.lddqkv + h * p.dh + c16;
                store1(row + D, dk[r]);
                store1(row + 2 * D, dv[r]);
            }
        }
    }
}

bool supported(int B, int N, int H, int dh, int ld, int ld2, const void* a, const void* b2) {
    return dh <= 16 && dh % 4 == 0 && N <= 272 && ld % 4 == 0 && ld2 % 4 == 0 && (H * dh) % 4 == 0 && !((uintptr_t)a & 15) && !((uintptr_t)b2 & 15) &&
           (long long)B * H <= 65535 && (long long)B * H * N * N < (1ll << 32);
}

#define DH16_DISPATCH_T(kernel, T, nt, grid, s, p)                          \
    do {                                                                   \
        if (nt <= 2) kernel<2, T><<<grid, 256, 0, s>>>(p);                  \
        else if (nt <= 4) kernel<4, T><<<grid, 256, 0, s>>>(p);             \
        else if (nt <= 7) kernel<7, T><<<grid, 256, 0, s>>>(p);             \
        else if (nt <= 10) kernel<10, T><<<grid, 256, 0, s>>>(p);           \
        else kernel<17, T><<<grid, 256, 0, s>>>(p);                         \
    } while (0)
#define DH16_DISPATCH(kernel, io16, nt, grid, s, p)                        \
    do {                                                                   \
        if (io16) DH16_DISPATCH_T(kernel, bf16_t, nt, grid, s, p);         \
        else DH16_DISPATCH_T(kernel, float, nt, grid, s, p);               \
    } while (0)

}  // namespace

extern "C" int medp_attn_dh16_train_supported(int B, int N, int H, int dh, int ld, int ldo) {
    return B > 0 && N > 0 && H > 0 && dh > 0 && ld >= 3 * H * dh && ldo >= H * dh && supported(B, N, H, dh, ld, ldo, nullptr, nullptr) ? 1 : 0;
}

// returns -2 (nothing launched) for shapes these kernels are not built for: the caller then uses medp_attn_small_fwd / _bwd
extern "C" int medp_attn_dh16_train_fwd(const void* qkv, int ld, void* o, int ldo, float* lse, int io_bf16, int B, int N, int H, int dh,
                                        float scale, float dropout_p, unsigned seed, unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(qkv && o && lse && B > 0 && N > 0 && H > 0 && dh > 0, "attn_dh16_train_fwd: bad argument");
    MEDP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && scale > 0.f, "attn_dh16_train_fwd: dropout p / scale out of range");
    if (!supported(B, N, H, dh, ld, ldo, qkv, o)) return -2;
    MEDP_CHECK_ARG(ld >= 3 * H * dh && ldo >= H * dh, "attn_dh16_train_fwd: bad leading dimension");
    TrainParams p{qkv, ld, o, ldo, lse, nullptr, 0, nullptr, nullptr, 0, B, N, H, dh, scale * 1.4426950408889634f, scale, dropout_p,
                  1.0f / (1.0f - dropout_p), seed, stream_id, medp_rng_epoch_ptr()};
    const int nt = (N + 15) / 16;
    const dim3 grid((nt + 3) / 4, B * H);
    hipStream_t s = (hipStream_t)stream;
    DH16_DISPATCH(dh16_train_fwd_kernel, io_bf16, nt, grid, s, p);
    MEDP_LAUNCH_CHECK("medp_attn_dh16_train_fwd");
    return 0;
}

// dqkv [B*N][lddqkv] receives dQ | dK | dV in the column blocks of qkv; delta_ws: B*H*N floats of scratch
extern "C" int medp_attn_dh16_train_bwd(const void* dout, int lddo, const void* qkv, int ld, const float* lse, float* delta_ws, void* dqkv,
                                        int lddqkv, int io_bf16, int B, int N, int H, int dh, float scale, float dropout_p, unsigned seed,
                                        unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG(dout && qkv && lse && delta_ws && dqkv && B > 0 && N > 0 && H > 0 && dh > 0, "attn_dh16_train_bwd: bad argument");
    MEDP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && scale > 0.f, "attn_dh16_train_bwd: dropout p / scale out of range");
    if (!supported(B, N, H, dh, ld, lddo, qkv, dout) || lddqkv % 4 != 0 || ((uintptr_t)dqkv & 15)) return -2;
    MEDP_CHECK_ARG(ld >= 3 * H * dh && lddqkv >= 3 * H * dh && lddo >= H * dh, "attn_dh16_train_bwd: bad leading dimension");
    TrainParams p{qkv, ld, nullptr, 0, (float*)lse, dout, lddo, delta_ws, dqkv, lddqkv, B, N, H, dh, scale * 1.4426950408889634f, scale,
                  dropout_p, 1.0f / (1.0f - dropout_p), seed, stream_id, medp_rng_epoch_ptr()};
    const int nt = (N + 15) / 16;
    const dim3 grid((nt + 3) / 4, B * H);
    hipStream_t s = (hipStream_t)stream;
    DH16_DISPATCH(dh16_train_bwd_dq_kernel, io_bf16, nt, grid, s, p);
    MEDP_LAUNCH_CHECK("medp_attn_dh16_train_bwd(dq)");
    DH16_DISPATCH(dh16_train_bwd_dkv_kernel, io_bf16, nt, grid, s, p);
    MEDP_LAUNCH_CHECK("medp_attn_dh16_train_bwd(dk, dv)");
    return 0;
}
