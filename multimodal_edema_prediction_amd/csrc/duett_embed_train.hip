// The per-variable embedding MLP of DuETT in TRAINING form as fused kernels (student KD path; reference duett/duett.py:11-39 `simple_mlp`
// = Linear(2, C) -> ReLU -> BatchNormLastDim(C) -> Linear(C, E), one per variable, model file :45-55), group = variable:
//     x [G][R][KIN]  ->  a = relu(W0 x + b0)  ->  hb = BN_train(a)  ->  out = W1 hb + b1   [G][R][E]
// The grouped-layer kernels of duett_train.hip materialise a and hb ([48][6144][64] fp32 = 75 MB each at cfg3) and walk them with seven
// forward and ~twenty backward launches (0.24 + 0.43 ms per student step, profiles/r03_kerneltrace_bench_student.txt).  With KIN = 2 the
// hidden row costs two FMAs per channel to recompute, so here NOTHING of size C is ever stored: the forward is a statistics pass and an
// output pass over x (2.4 MB), the backward two passes over (x, dout) that recompute a / xhat / dhb in registers.
//   lane = hidden channel c (C = 64 = one wave), a wave walks rows; row-uniform values (x[r][:], dout[r][:]) come from one coalesced
//   load + v_readlane, so the FMAs take them as scalar operands.  Sums over rows are per-lane accumulators, combined across waves and
//   row chunks in a fixed order: bitwise reproducible, no atomics.
#include "common.h"
#include "medp_hip.h"

namespace {

constexpr int C = 64;          // hidden width = lanes of a wave
constexpr int ST_RPC = 128;    // rows per workgroup of the statistics pass (same chunking as gbn_stats_partial_kernel)
constexpr int BW_RPC = 128;    // rows per workgroup of the two backward passes (4 waves x 32 rows, staged in LDS; 256 measured slower: 73 vs 46 us)
inline int st_chunks(int R) { return (R + ST_RPC - 1) / ST_RPC; }
inline int bw_chunks(int R) { return (R + BW_RPC - 1) / BW_RPC; }

template <int KIN>
struct HiddenPar {             // what a lane (= channel c of group g) needs to rebuild its hidden value
    float w0[KIN], b0, mu, rs, bw, bb;
};
template <int KIN>
__device__ __forceinline__ float pre_act(const HiddenPar<KIN>& p, const float* xr) {
    float a = p.b0;
#pragma unroll
    for (int k = 0; k < KIN; ++k) a += p.w0[k] * xr[k];       // bias first, ascending k: the order of glinear_fwd_kernel
    return a;
}

// ---- forward, pass 1: shifted sums of a = relu(W0 x + b0) per (g, c) and row chunk (pivot = the group's first row) --------------
template <int KIN>
__global__ __launch_bounds__(256) void gmlp_stats_partial_kernel(const float* __restrict__ x, const float* __restrict__ W0,
                                                                 const float* __restrict__ b0, float* __restrict__ part, int R) {
    __shared__ float red[2][4][C];
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    HiddenPar<KIN> p;
#pragma unroll
    for (int k = 0; k < KIN; ++k) p.w0[k] = W0[((size_t)g * C + c) * KIN + k];
    p.b0 = b0[(size_t)g * C + c];
    const float* xg = x + (size_t)g * R * KIN;
    const float pivot = fmaxf(pre_act<KIN>(p, xg), 0.f);
    const int r0 = chunk * ST_RPC, r1 = min(R, r0 + ST_RPC);
    float s1 = 0.f, s2 = 0.f;
    for (int r = r0 + rl; r < r1; r += 4) {
        const float d = fmaxf(pre_act<KIN>(p, xg + (size_t)r * KIN), 0.f) - pivot;
        s1 += d;
        s2 += d * d;
    }
    red[0][rl][c] = s1;
    red[1][rl][c] = s2;
    __syncthreads();
    if (rl == 0) {
        float* o = part + (((size_t)g * nchunk + chunk) * 2) * C;
        o[c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        o[C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}
// chunks added in order; mean / biased variance saved for the backward; running statistics updated as gbn_running_kernel does
template <int KIN>
__global__ __launch_bounds__(256) void gmlp_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ W0, const float* __restrict__ b0,
                                                               const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ var,
                                                               float* __restrict__ rmean, float* __restrict__ rvar, int G, int R, int nchunk,
                                                               float momentum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    HiddenPar<KIN> p;
#pragma unroll
    for (int k = 0; k < KIN; ++k) p.w0[k] = W0[(size_t)i * KIN + k];
    p.b0 = b0[i];
    const float pivot = fmaxf(pre_act<KIN>(p, x + (size_t)g * R * KIN), 0.f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) {
        const float* o = part + (((size_t)g * nchunk + k) * 2) * C;
        s1 += o[c];
        s2 += o[C + c];
    }
    const float m1 = s1 / (float)R;
    const float mu = pivot + m1, v = fmaxf(s2 / (float)R - m1 * m1, 0.f);
    mean[i] = mu;
    var[i] = v;
    if (rmean && rvar) {
        rmean[i] = (1.f - momentum) * rmean[i] + momentum * mu;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * v * ((float)R / (float)max(R - 1, 1));
    }
}

// ---- forward, pass 2: out[r][:] = b1 + W1 BN(relu(W0 x[r] + b0)).  The group's parameters are broadcast out of LDS; a broadcast read
// costs the LDS its full 8 clocks per 16 bytes and lane, so a thread owns OUT_RPT rows and every read feeds that many rows (one row per
// thread was LDS-bound: 30 us at cfg3) ---------------------------------------------------------------------------------------------
constexpr int OUT_RPT = 4, OUT_THREADS = 64;
template <int KIN, int E>
__global__ __launch_bounds__(OUT_THREADS) void gmlp_out_kernel(const float* __restrict__ x, const float* __restrict__ W0, const float* __restrict__ b0,
                                                               const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                               const float* __restrict__ mean, const float* __restrict__ var,
                                                               const float* __restrict__ W1, const float* __restrict__ b1, float* __restrict__ out,
                                                               int R, float eps) {
    static_assert(E % 4 == 0 && KIN <= 3, "layout");
    __shared__ __attribute__((aligned(16))) float s_par[C][8];      // w0[KIN] | b0 | mu | rs | w | b  (KIN + 5 <= 8)
    __shared__ __attribute__((aligned(16))) float s_w1[C][E];       // W1 transposed: [c][n]
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += OUT_THREADS) {
        const size_t gc = (size_t)g * C + i;
#pragma unroll
        for (int k = 0; k < KIN; ++k) s_par[i][k] = W0[gc * KIN + k];
        s_par[i][KIN] = b0[gc];
        s_par[i][KIN + 1] = mean[gc];
        s_par[i][KIN + 2] = rsqrtf(var[gc] + eps);
        s_par[i][KIN + 3] = bn_w[gc];
        s_par[i][KIN + 4] = bn_b[gc];
    }
    for (int i = threadIdx.x; i < C * E; i += OUT_THREADS) {
        const int n = i / C, c = i - n * C;                         // coalesced read of W1 [E][C]
        s_w1[c][n] = W1[(size_t)g * E * C + i];
    }
    __syncthreads();
    // rows of this thread: blockIdx.x * (OUT_THREADS * OUT_RPT) + j * OUT_THREADS + threadIdx.x  (consecutive lanes = consecutive rows)
    const int rb = blockIdx.x * (OUT_THREADS * OUT_RPT) + threadIdx.x;
    float xr[OUT_RPT][KIN], acc[OUT_RPT][E];
#pragma unroll
    for (int j = 0; j < OUT_RPT; ++j) {
        const int r = min(rb + j * OUT_THREADS, R - 1);
#pragma unroll
        for (int k = 0; k < KIN; ++k) xr[j][k] = x[((size_t)g * R + r) * KIN + k];
#pragma unroll
        for (int n = 0; n < E; ++n) acc[j][n] = b1[(size_t)g * E + n];
    }
#pragma unroll 2
    for (int c = 0; c < C; ++c) {
        const float4 p0 = *(const float4*)&s_par[c][0], p1 = *(const float4*)&s_par[c][4];
        const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float hb[OUT_RPT];
#pragma unroll
        for (int j = 0; j < OUT_RPT; ++j) {
            float a = pv[KIN];
#pragma unroll
            for (int k = 0; k < KIN; ++k) a += pv[k] * xr[j][k];
            a = fmaxf(a, 0.f);
            hb[j] = (a - pv[KIN + 1]) * pv[KIN + 2] * pv[KIN + 3] + pv[KIN + 4];            // the expression of gbn_apply_kernel
        }
#pragma unroll
        for (int n4 = 0; n4 < E / 4; ++n4) {
            const float4 w = *(const float4*)&s_w1[c][4 * n4];
#pragma unroll
            for (int j = 0; j < OUT_RPT; ++j) {
                acc[j][4 * n4 + 0] += w.x * hb[j];                                          // ascending c from the bias: glinear_fwd_kernel's order
                acc[j][4 * n4 + 1] += w.y * hb[j];
                acc[j][4 * n4 + 2] += w.z * hb[j];
                acc[j][4 * n4 + 3] += w.w * hb[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < OUT_RPT; ++j) {
        const int r = rb + j * OUT_THREADS;
        if (r < R) {
            float* o = out + ((size_t)g * R + r) * E;
#pragma unroll
            for (int n4 = 0; n4 < E / 4; ++n4)
                *(float4*)(o + 4 * n4) = make_float4(acc[j][4 * n4], acc[j][4 * n4 + 1], acc[j][4 * n4 + 2], acc[j][4 * n4 + 3]);
        }
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------------------
// Both passes: workgroup = (row chunk of BW_RPC rows, group), 4 waves x BW_RPC/4 rows each, lane = channel c.
template <int KIN, int E>
struct LaneState {
    HiddenPar<KIN> p;
    float w1c[E];              // W1[n][c] for every n
};
template <int KIN, int E>
__device__ __forceinline__ void load_lane_state(LaneState<KIN, E>& st, int g, int c, const float* W0, const float* b0, const float* bn_w,
                                                const float* bn_b, const float* mean, const float* var, const float* W1, float eps) {
    const size_t gc = (size_t)g * C + c;
#pragma unroll
    for (int k = 0; k < KIN; ++k) st.p.w0[k] = W0[gc * KIN + k];
    st.p.b0 = b0[gc];
    st.p.mu = mean[gc];
    st.p.rs = rsqrtf(var[gc] + eps);
    st.p.bw = bn_w[gc];
    st.p.bb = bn_b[gc];
#pragma unroll
    for (int n = 0; n < E; ++n) st.w1c[n] = W1[((size_t)g * E + n) * C + c];
}
// The chunk's dout and x rows are staged in LDS once (coalesced 16-byte loads, all in flight together); the row loop then reads a row's
// E + KIN values with wave-uniform addresses (LDS broadcast, in-order returns the compiler can pipeline).  Reading them per row from
// global or scalar memory made the loop a chain of exposed load latencies (82 / 106 us per pass at cfg3).
template <int KIN, int E>
struct RowStage {
    static constexpr int ROWF = E + 4;                         // dout[E] | x[KIN] | pad: 16-byte aligned rows
    static_assert(E % 4 == 0 && KIN <= 4, "row layout");
    __device__ static __forceinline__ void fill(float* stage, const float* __restrict__ dg, const float* __restrict__ xg, int rbase, int nr) {
        for (int i = threadIdx.x; i < nr * (E / 4); i += 256) {
            const int row = i / (E / 4), q = i - row * (E / 4);
            *(float4*)(stage + row * ROWF + 4 * q) = *(const float4*)(dg + ((size_t)rbase + row) * E + 4 * q);
        }
        for (int i = threadIdx.x; i < nr * KIN; i += 256) stage[(i / KIN) * ROWF + E + i % KIN] = xg[(size_t)rbase * KIN + i];
    }
    __device__ static __forceinline__ void row(const float* stage, int rl, float* xr, float* dn) {
        const float* p = stage + rl * ROWF;
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const float4 v = *(const float4*)(p + 4 * q);
            dn[4 * q] = v.x; dn[4 * q + 1] = v.y; dn[4 * q + 2] = v.z; dn[4 * q + 3] = v.w;
        }
        const float4 xv = *(const float4*)(p + E);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < KIN; ++k) xr[k] = xa[k];
    }
};
// sum over the 64 lanes in a fixed order: DPP butterfly inside each 16-lane row (VALU rate, no LDS round trips), then the four rows
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));    // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));    // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
    return (r0 + r1) + (r2 + r3);
}

// pass 1: per chunk  dW1[n][c] = sum_r dout[r][n] hb[r][c],  s1[c] = sum_r dhb,  s2[c] = sum_r dhb xhat,  db1[n] = sum_r dout[r][n]
// partial layout per (g, chunk): E*C (dW1, n-major) | C (s1) | C (s2) | E (db1)
template <int KIN, int E>
__global__ __launch_bounds__(256) void gmlp_bwd_sums_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ W0,
                                                            const float* __restrict__ b0, const float* __restrict__ bn_w,
                                                            const float* __restrict__ bn_b, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ W1, float* __restrict__ part,
                                                            int R, float eps) {
    using RS = RowStage<KIN, E>;
    constexpr int RED_F = 4 * (E + 3) * C, STAGE_F = BW_RPC * RS::ROWF;
    __shared__ __attribute__((aligned(16))) float smem[RED_F > STAGE_F ? RED_F : STAGE_F];          // the staged rows, then the cross-wave sums
    float (*red)[E + 3][C] = (float (*)[E + 3][C])smem;
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LaneState<KIN, E> st;
    load_lane_state<KIN, E>(st, g, lane, W0, b0, bn_w, bn_b, mean, var, W1, eps);
    const int rbase = chunk * BW_RPC, nr = min(R - rbase, BW_RPC);
    RS::fill(smem, dout + (size_t)g * R * E, x + (size_t)g * R * KIN, rbase, nr);
    __syncthreads();
    const int l0 = wave * (BW_RPC / 4), l1 = min(nr, l0 + BW_RPC / 4);
    float dw1[E], s1 = 0.f, s2 = 0.f, db1 = 0.f;
#pragma unroll
    for (int n = 0; n < E; ++n) dw1[n] = 0.f;
    if (lane < E)                                              // db1: lane n adds column n of the wave's rows
        for (int rl = l0; rl < l1; ++rl) db1 += smem[rl * RS::ROWF + lane];
#pragma unroll 2
    for (int rl = l0; rl < l1; ++rl) {
        float xr[KIN], dn[E];
        RS::row(smem, rl, xr, dn);
        const float a = fmaxf(pre_act<KIN>(st.p, xr), 0.f);
        const float xhat = (a - st.p.mu) * st.p.rs;
        const float hb = xhat * st.p.bw + st.p.bb;
        float dhb = 0.f;
#pragma unroll
        for (int n = 0; n < E; ++n) {
            dhb += dn[n] * st.w1c[n];                          // ascending n: glinear_bwd_dx_kernel's order
            dw1[n] += dn[n] * hb;
        }
        s1 += dhb;
        s2 += dhb * xhat;
    }
    __syncthreads();                                           // every wave is done with the staged rows: their LDS becomes `red`
#pragma unroll
    for (int n = 0; n < E; ++n) red[wave][n][lane] = dw1[n];
    red[wave][E][lane] = s1;
    red[wave][E + 1][lane] = s2;
    red[wave][E + 2][lane] = db1;
    __syncthreads();
    float* o = part + ((size_t)g * nchunk + chunk) * (E * C + 2 * C + E);
    for (int i = threadIdx.x; i < (E + 3) * C; i += 256) {
        const int q = i / C, c = i - q * C;
        const float v = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
        if (q < E + 2) o[i] = v;                               // dW1 rows, s1, s2 keep the [q][c] layout
        else if (c < E) o[(E + 2) * C + c] = v;                // db1: lanes 0..E-1 carry it
    }
}
// chunks added in order, each block of the sum written where it belongs: dW1 [G][E][C] | d(bn bias) = s1 [G][C] | d(bn weight) = s2 [G][C] | db1 [G][E]
template <int E>
__global__ __launch_bounds__(256) void gmlp_bwd_sums_final_kernel(const float* __restrict__ part, float* __restrict__ dW1, float* __restrict__ dbn_b,
                                                                  float* __restrict__ dbn_w, float* __restrict__ db1, int nchunk) {
    constexpr int D = E * C + 2 * C + E;
    const int g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) a += part[((size_t)g * nchunk + k) * D + i];
    if (i < E * C) dW1[(size_t)g * E * C + i] = a;
    else if (i < E * C + C) dbn_b[(size_t)g * C + (i - E * C)] = a;
    else if (i < E * C + 2 * C) dbn_w[(size_t)g * C + (i - E * C - C)] = a;
    else db1[(size_t)g * E + (i - E * C - 2 * C)] = a;
}
template <int KIN>
__global__ __launch_bounds__(256) void gmlp_bwd_dx_final_kernel(const float* __restrict__ part, float* __restrict__ dW0, float* __restrict__ db0, int nchunk) {
    constexpr int D = C * KIN + C;
    const int g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < nchunk; ++k) a += part[((size_t)g * nchunk + k) * D + i];
    if (i < C * KIN) dW0[(size_t)g * C * KIN + i] = a;
    else db0[(size_t)g * C + (i - C * KIN)] = a;
}

// pass 2: dpre = relu'(a) BN'(dhb);  dx[r][k] = sum_c dpre W0[c][k] (wave reduction);  per chunk  dW0[c][k] = sum_r dpre x[r][k],  db0[c] = sum_r dpre
// s1 / s2 [G][C]: the finished BatchNorm sums of pass 1 (= the gradients of its bias / weight).  partial layout per (g, chunk): C*KIN (dW0) | C (db0)
template <int KIN, int E>
__global__ __launch_bounds__(256) void gmlp_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ W0,
                                                          const float* __restrict__ b0, const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                          const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ W1,
                                                          const float* __restrict__ s1g, const float* __restrict__ s2g, float* __restrict__ dx,
                                                          float* __restrict__ part, int R, float eps, int batch_stats) {
    using RS = RowStage<KIN, E>;
    __shared__ __attribute__((aligned(16))) float stage[BW_RPC * RS::ROWF];
    __shared__ float red[4][KIN + 1][C];
    __shared__ float sdx[BW_RPC][KIN];                          // dx of the chunk's rows, written out coalesced at the end
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LaneState<KIN, E> st;
    load_lane_state<KIN, E>(st, g, lane, W0, b0, bn_w, bn_b, mean, var, W1, eps);
    const float s1 = s1g[(size_t)g * C + lane], s2 = s2g[(size_t)g * C + lane];
    const int rbase = chunk * BW_RPC, nr = min(R - rbase, BW_RPC);
    RS::fill(stage, dout + (size_t)g * R * E, x + (size_t)g * R * KIN, rbase, nr);
    __syncthreads();
    const int l0 = wave * (BW_RPC / 4), l1 = min(nr, l0 + BW_RPC / 4);
    float dw0[KIN], db0 = 0.f;
#pragma unroll
    for (int k = 0; k < KIN; ++k) dw0[k] = 0.f;
    for (int rl = l0; rl < l1; ++rl) {
        float xr[KIN], dn[E];
        RS::row(stage, rl, xr, dn);
        const float pre = pre_act<KIN>(st.p, xr);
        const float a = fmaxf(pre, 0.f);
        float dhb = 0.f;
#pragma unroll
        for (int n = 0; n < E; ++n) dhb += dn[n] * st.w1c[n];
        float v = dhb;
        if (batch_stats) v -= s1 / (float)R + (a - st.p.mu) * st.p.rs * s2 / (float)R;      // the expression of gbn_bwd_dx_kernel
        const float da = st.p.bw * st.p.rs * v;
        const float dpre = a > 0.f ? da : 0.f;
        db0 += dpre;
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            dw0[k] += dpre * xr[k];
            const float t = wave_sum_dpp(dpre * st.p.w0[k]);
            if (lane == k) sdx[rl][k] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < KIN; ++k) red[wave][k][lane] = dw0[k];
    red[wave][KIN][lane] = db0;
    __syncthreads();
    if (dx)
        for (int i = threadIdx.x; i < nr * KIN; i += 256) dx[((size_t)g * R + rbase) * KIN + i] = (&sdx[0][0])[i];
    float* o = part + ((size_t)g * nchunk + chunk) * (C * KIN + C);
    for (int i = threadIdx.x; i < (KIN + 1) * C; i += 256) {
        const int q = i / C, c = i - q * C;
        const float v = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
        if (q < KIN) o[c * KIN + q] = v;                       // dW0 [C][KIN]
        else o[C * KIN + c] = v;                               // db0
    }
}

bool gmlp_shape_ok(int KIN, int Cc, int E) { return KIN == 2 && Cc == C && E == 24; }

}  // namespace

extern "C" int medp_gmlp_supported(int KIN, int Ch, int E) { return gmlp_shape_ok(KIN, Ch, E) ? 1 : 0; }

extern "C" size_t medp_gmlp_workspace_bytes(int G, int R, int KIN, int Ch, int E) {
    if (G <= 0 || R <= 0 || !gmlp_shape_ok(KIN, Ch, E)) return 0;
    const size_t fwd = (size_t)G * st_chunks(R) * 2 * C;
    const size_t bwd = (size_t)G * bw_chunks(R) * ((size_t)E * C + 2 * C + E) + (size_t)G * bw_chunks(R) * ((size_t)C * KIN + C);
    return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

extern "C" int medp_gmlp_fwd(const float* x, const float* W0, const float* b0, const float* bn_w, const float* bn_b, float* running_mean,
                             float* running_var, const float* W1, const float* b1, float* out, float* save_mean, float* save_var, int G, int R,
                             int KIN, int Ch, int E, float eps, float momentum, int batch_stats, float* workspace, void* stream) {
    MEDP_CHECK_ARG(x && W0 && b0 && bn_w && bn_b && W1 && b1 && out && save_mean && save_var && G > 0 && R > 0, "gmlp_fwd: bad argument");
    if (!gmlp_shape_ok(KIN, Ch, E)) return -2;
    MEDP_CHECK_ARG(G <= 65535, "gmlp_fwd: at most 65535 groups");
    hipStream_t s = (hipStream_t)stream;
    if (batch_stats) {
        MEDP_CHECK_ARG(workspace, "gmlp_fwd: batch statistics need a workspace (medp_gmlp_workspace_bytes)");
        const int nc = st_chunks(R);
        gmlp_stats_partial_kernel<2><<<dim3(nc, G), 256, 0, s>>>(x, W0, b0, workspace, R);
        gmlp_stats_final_kernel<2><<<(G * C + 255) / 256, 256, 0, s>>>(x, W0, b0, workspace, save_mean, save_var, running_mean, running_var, G, R, nc,
                                                                         momentum);
        MEDP_LAUNCH_CHECK("medp_gmlp_fwd(stats)");
    } else {
        MEDP_CHECK_ARG(running_mean && running_var, "gmlp_fwd: eval mode needs running statistics");
        (void)hipMemcpyAsync(save_mean, running_mean, (size_t)G * C * 4, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(save_var, running_var, (size_t)G * C * 4, hipMemcpyDeviceToDevice, s);
    }
    gmlp_out_kernel<2, 24><<<dim3((R + OUT_THREADS * OUT_RPT - 1) / (OUT_THREADS * OUT_RPT), G), OUT_THREADS, 0, s>>>(x, W0, b0, bn_w, bn_b, save_mean, save_var, W1, b1,
                                                                                                                out, R, eps);
    MEDP_LAUNCH_CHECK("medp_gmlp_fwd(out)");
    return 0;
}

extern "C" int medp_gmlp_bwd(const float* dout, const float* x, const float* W0, const float* b0, const float* bn_w, const float* bn_b,
                             const float* save_mean, const float* save_var, const float* W1, float* dx, float* dW0, float* db0, float* dbn_w,
                             float* dbn_b, float* dW1, float* db1, int G, int R, int KIN, int Ch, int E, float eps, int batch_stats,
                             float* workspace, void* stream) {
    MEDP_CHECK_ARG(dout && x && W0 && b0 && bn_w && bn_b && save_mean && save_var && W1 && dW0 && db0 && dbn_w && dbn_b && dW1 && db1 &&
                       workspace && G > 0 && R > 0, "gmlp_bwd: bad argument");
    if (!gmlp_shape_ok(KIN, Ch, E)) return -2;
    MEDP_CHECK_ARG(G <= 65535, "gmlp_bwd: at most 65535 groups");
    hipStream_t s = (hipStream_t)stream;
    const int nc = bw_chunks(R);
    const int D1 = E * C + 2 * C + E, D0 = C * KIN + C;
    float* part1 = workspace;
    float* part0 = part1 + (size_t)G * nc * D1;
    gmlp_bwd_sums_kernel<2, 24><<<dim3(nc, G), 256, 0, s>>>(dout, x, W0, b0, bn_w, bn_b, save_mean, save_var, W1, part1, R, eps);
    gmlp_bwd_sums_final_kernel<24><<<dim3((D1 + 255) / 256, G), 256, 0, s>>>(part1, dW1, dbn_b, dbn_w, db1, nc);
    MEDP_LAUNCH_CHECK("medp_gmlp_bwd(sums)");
    gmlp_bwd_dx_kernel<2, 24><<<dim3(nc, G), 256, 0, s>>>(dout, x, W0, b0, bn_w, bn_b, save_mean, save_var, W1, dbn_b, dbn_w, dx, part0, R, eps,
                                                          batch_stats);
    gmlp_bwd_dx_final_kernel<2><<<dim3((D0 + 255) / 256, G), 256, 0, s>>>(part0, dW0, db0, nc);
    MEDP_LAUNCH_CHECK("medp_gmlp_bwd(dx)");
    return 0;
}
