// Training-form kernels of the DuETT embedding stage (student KD path: BatchNorm in TRAIN mode, gradients to every
// DuETT parameter; reference duett/duett.py:11-39,84-88,124-125,151-157 + model file :41-69).
// The per-variable MLPs are run as GROUPED tiny layers (group = variable): hidden activations are materialised
// [G, R, C] so BatchNorm batch statistics and every weight gradient are plain column reductions — deterministic
// two-stage sums, no float atomics.  All fp32; FLOPs are negligible, the cost is HBM traffic and launch count.
#include "common.h"
#include "medp_hip.h"

namespace {

inline int grid_for(size_t work_items) { return (int)min((size_t)4096, max((size_t)1, (work_items + 255) / 256)); }

// ---- grouped tiny linear: y[g][r][n] = b[g][n] + sum_k W[g][n][k] x[g][r][k]      (N*K <= 8192) ---------------------
constexpr int GLINEAR_MAX_FLOATS = 36000;     // weights (+ bias) of one group held in LDS: 144 KB of the CU's 160 KB
// y[g][r][n] = b[g][n] + sum_k W[g][n][k] x[g][r][k].  ONE OUTPUT ELEMENT PER THREAD, consecutive threads = consecutive n of a row,
// so every store instruction writes one contiguous span of y (the first form gave a thread a whole row: its 64 stores each
// scattered 64 lanes 256 B apart — 193 us for a 75-MB result at cfg3).  The group's weights sit in LDS with rows padded to
// K + 1 floats (lanes differ in n: stride K would put them all in one bank); x is read through the cache (the N threads of a
// row read the same K floats).  Products are summed in ascending k from the bias, as before: same values.
__global__ __launch_bounds__(256) void glinear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ y, int R, int K, int N) {
    extern __shared__ float sw[];            // N*(K+1) weights + N bias
    const int g = blockIdx.y, KP = K + 1;
    for (int i = threadIdx.x; i < N * K; i += 256) sw[(i / K) * KP + i % K] = W[(size_t)g * N * K + i];
    for (int i = threadIdx.x; i < N; i += 256) sw[N * KP + i] = b ? b[(size_t)g * N + i] : 0.f;
    __syncthreads();
    const size_t total = (size_t)R * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / N), n = (int)(idx - (size_t)r * N);
        const float* xr = x + ((size_t)g * R + r) * K;
        const float* wr = sw + n * KP;
        float a = sw[N * KP + n];
        for (int k = 0; k < K; ++k) a += wr[k] * xr[k];
        y[(size_t)g * total + idx] = a;
    }
}
// dx[g][r][k] = sum_n dy[g][r][n] W[g][n][k]: one output element per thread, consecutive threads = consecutive k (LDS reads
// conflict-free, stores contiguous); ascending n as before.
__global__ __launch_bounds__(256) void glinear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W, float* __restrict__ dx,
                                                             int R, int K, int N) {
    extern __shared__ float sw[];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < N * K; i += 256) sw[i] = W[(size_t)g * N * K + i];
    __syncthreads();
    const size_t total = (size_t)R * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / K), k = (int)(idx - (size_t)r * K);
        const float* dr = dy + ((size_t)g * R + r) * N;
        float a = 0.f;
        for (int n = 0; n < N; ++n) a += dr[n] * sw[n * K + k];
        dx[(size_t)g * total + idx] = a;
    }
}
// partial[g][chunk][n*K+k] = sum_{r in chunk} dy[r][n] x[r][k] ;  partial_b[g][chunk][n] = sum dy[r][n]
// The chunk's dy and x rows are staged in LDS with coalesced loads (they are contiguous spans), then every thread walks the rows
// for its (n, k) outputs out of LDS — the first form read both operands straight from global memory inside the row loop, two
// dependent loads per row and output (126 us per launch at cfg3).  Same summation order (ascending r): same values.
__global__ __launch_bounds__(256) void glinear_bwd_dw_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                     float* __restrict__ pw, float* __restrict__ pb, int R, int K, int N,
                                                                     int rows_per_chunk) {
    extern __shared__ float sm[];            // [rows][N] dy, then [rows][K] x
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int r0 = chunk * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk), nr = max(r1 - r0, 0);
    float* sdy = sm;
    float* sx = sm + (size_t)rows_per_chunk * N;
    const float* dyg = dy + ((size_t)g * R + r0) * N;
    const float* xg = x + ((size_t)g * R + r0) * K;
    for (int i = threadIdx.x; i < nr * N; i += 256) sdy[i] = dyg[i];
    for (int i = threadIdx.x; i < nr * K; i += 256) sx[i] = xg[i];
    __syncthreads();
    for (int i = threadIdx.x; i < N * K; i += 256) {
        const int n = i / K, k = i % K;
        float a = 0.f;
        for (int r = 0; r < nr; ++r) a += sdy[r * N + n] * sx[r * K + k];
        pw[((size_t)g * nchunk + chunk) * N * K + i] = a;
    }
    for (int n = threadIdx.x; n < N; n += 256) {
        float a = 0.f;
        for (int r = 0; r < nr; ++r) a += sdy[r * N + n];
        pb[((size_t)g * nchunk + chunk) * N + n] = a;
    }
}
// out[g][i] = sum_chunk partial[g][chunk][i]
__global__ __launch_bounds__(256) void sum_chunks_kernel(const float* __restrict__ partial, float* __restrict__ out, int nchunk, int D) {
    const int g = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float a = 0.f;
    for (int c = 0; c < nchunk; ++c) a += partial[((size_t)g * nchunk + c) * D + i];
    out[(size_t)g * D + i] = a;
}

// ---- grouped BatchNorm over rows: x [G][R][C], statistics per (g, c) ------------------------------------------------
// Two stages, deterministic: (1) workgroups of 64 columns x 4 row-lanes each take a CHUNK of GBN_RPC rows of one group and
// write the chunk's shifted sums  S1 = sum (x - p),  S2 = sum (x - p)^2  (pivot p = the group's first row: no cancellation
// in  var = S2/R - (S1/R)^2  for activations whose mean dwarfs their spread); (2) one thread per (g, c) adds the chunks in
// order.  The first form ran ONE workgroup per group over all R rows (48 workgroups on 256 CUs, two dependent sweeps): 423 us
// for a 75-MB operand at cfg3; the backward sums kernel had the same shape (427 us).
constexpr int GBN_RPC = 128;
inline int gbn_chunks(int R) { return (R + GBN_RPC - 1) / GBN_RPC; }
__global__ __launch_bounds__(256) void gbn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int R, int C) {
    __shared__ float red[2][4][64];
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, cb = blockIdx.z * 64;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = cb + cl;
    const float* xg = x + (size_t)g * R * C;
    const int r0 = chunk * GBN_RPC, r1 = min(R, r0 + GBN_RPC);
    float a = 0.f, q = 0.f;
    if (c < C) {
        const float p = xg[c];
        for (int r = r0 + rl; r < r1; r += 4) {
            const float d = xg[(size_t)r * C + c] - p;
            a += d;
            q += d * d;
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = q;
    __syncthreads();
    if (rl == 0 && c < C) {
        float* o = part + (((size_t)g * nchunk + chunk) * 2) * C;
        o[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        o[C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}
__global__ __launch_bounds__(256) void gbn_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ var, int G, int R, int C, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        const float* o = part + (((size_t)g * nchunk + k) * 2) * C;
        s1 += o[c];
        s2 += o[C + c];
    }
    const float m1 = s1 / (float)R;
    mean[i] = x[(size_t)g * R * C + c] + m1;
    var[i] = fmaxf(s2 / (float)R - m1 * m1, 0.f);             // biased
}
// running = (1-m)*running + m*batch  (unbiased variance), num_batches_tracked handled by the caller
__global__ __launch_bounds__(256) void gbn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, int n, int R, float momentum) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    rmean[i] = (1.f - momentum) * rmean[i] + momentum * mean[i];
    rvar[i] = (1.f - momentum) * rvar[i] + momentum * var[i] * ((float)R / (float)max(R - 1, 1));
}
__global__ __launch_bounds__(256) void gbn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                        int G, int R, int C, float eps) {
    const size_t n = (size_t)G * R * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C), g = (int)(i / ((size_t)R * C));
        const int gc = g * C + c;
        y[i] = (x[i] - mean[gc]) * rsqrtf(var[gc] + eps) * w[gc] + b[gc];
    }
}
// sums for backward: s1[g][c] = sum_r dy ; s2[g][c] = sum_r dy * xhat — per chunk of rows, then added in order (see above)
__global__ __launch_bounds__(256) void gbn_bwd_sums_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ var,
                                                                   float* __restrict__ part, int R, int C, float eps) {
    __shared__ float red[2][4][64];
    const int g = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, cb = blockIdx.z * 64;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = cb + cl;
    const int r0 = chunk * GBN_RPC, r1 = min(R, r0 + GBN_RPC);
    float a = 0.f, q = 0.f;
    if (c < C) {
        const float mu = mean[(size_t)g * C + c], rs = rsqrtf(var[(size_t)g * C + c] + eps);
        for (int r = r0 + rl; r < r1; r += 4) {
            const size_t i = ((size_t)g * R + r) * C + c;
            a += dy[i];
            q += dy[i] * (x[i] - mu) * rs;
        }
    }
    red[0][rl][cl] = a;
    red[1][rl][cl] = q;
    __syncthreads();
    if (rl == 0 && c < C) {
        float* o = part + (((size_t)g * nchunk + chunk) * 2) * C;
        o[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        o[C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}
__global__ __launch_bounds__(256) void gbn_bwd_sums_final_kernel(const float* __restrict__ part, float* __restrict__ s1, float* __restrict__ s2,
                                                                 int G, int C, int nchunk) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    float a = 0.f, q = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        const float* o = part + (((size_t)g * nchunk + k) * 2) * C;
        a += o[c];
        q += o[C + c];
    }
    s1[i] = a;
    s2[i] = q;
}
// train: dx = w*rstd*(dy - s1/R - xhat*s2/R) ; eval (batch_stats == 0): dx = w*rstd*dy
__global__ __launch_bounds__(256) void gbn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ var, const float* __restrict__ w, const float* __restrict__ s1,
                                                         const float* __restrict__ s2, float* __restrict__ dx, int G, int R, int C, float eps,
                                                         int batch_stats) {
    const size_t n = (size_t)G * R * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C), g = (int)(i / ((size_t)R * C));
        const int gc = g * C + c;
        const float rs = rsqrtf(var[gc] + eps);
        float v = dy[i];
        if (batch_stats) v -= s1[gc] / (float)R + (x[i] - mean[gc]) * rs * s2[gc] / (float)R;
        dx[i] = w[gc] * rs * v;
    }
}

// ---- activations: mode 0 = ReLU, 1 = tanh --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int mode) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = mode == 0 ? fmaxf(x[i], 0.f) : tanhf(x[i]);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n, int mode) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dx[i] = mode == 0 ? (y[i] > 0.f ? dy[i] : 0.f) : dy[i] * (1.f - y[i] * y[i]);
}

// ---- embedding inputs: xin[v][b*T+t][0] = value, [1] = n_obs_table[clip(int(count))], [2..KP) = 0   (model :41-52) -----
__global__ __launch_bounds__(256) void embed_inputs_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ table, int nrows_table,
                                                               float* __restrict__ xin, int B, int T, int V, int KP) {
    const size_t n = (size_t)V * B * T;
    const int F = 2 * V + 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i % ((size_t)B * T)), v = (int)(i / ((size_t)B * T));
        const float* row = xs + (size_t)r * F;
        const int idx = min(max((int)row[V + v], 0), nrows_table - 1);
        float* o = xin + i * KP;
        o[0] = row[v];
        o[1] = table[idx];
        for (int k = 2; k < KP; ++k) o[k] = 0.f;
    }
}
// d_table partials: every thread keeps 16 private bins over ITS elements (a fixed assignment), the bins are then summed over the
// wave with the fixed shuffle tree and over the four waves in a fixed order — bitwise reproducible (LDS float atomics, the
// first form of this kernel, are not: their order across waves varies from launch to launch, and AdamW turns a last-bit
// difference of a near-zero gradient into a full-size step).  The cross-block sum is a deterministic column sum.
__global__ __launch_bounds__(256) void embed_inputs_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ dxin, float* __restrict__ partial,
                                                               int nrows_table, int B, int T, int V, int KP) {
    constexpr int NB = 16;                       // n_obs_embedding has 16 rows (duett.py:88); checked by the launcher
    __shared__ float red[4][NB];
    float bins[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bins[k] = 0.f;
    const size_t n = (size_t)V * B * T;
    const int F = 2 * V + 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i % ((size_t)B * T)), v = (int)(i / ((size_t)B * T));
        const int idx = min(max((int)xs[(size_t)r * F + V + v], 0), nrows_table - 1);
        const float g = dxin[i * KP + 1];
#pragma unroll
        for (int k = 0; k < NB; ++k) bins[k] += idx == k ? g : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float w = wave_sum(bins[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < nrows_table)
        partial[(size_t)blockIdx.x * nrows_table + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- psi assembly (model :53-66) and its backward -------------------------------------------------------------------------
// psi[b][t][v][:] = var_out[v][b*T+t][:] | tab_out[b][:] (v == V) | special[0] (masked timestep / masked event) | special[1] (t == T)
__device__ __forceinline__ int psi_cell_kind(const float* __restrict__ xs, int b, int t, int v, int T, int V) {
    if (t == T) return 3;                                   // REP row
    const float* row = xs + ((size_t)b * T + t) * (2 * V + 1);
    if (row[2 * V] == 1.0f) return 2;                       // masked timestep
    if (v == V) return 1;                                   // static column
    if (row[V + v] == -1.0f) return 2;                      // masked event
    return 0;                                               // variable MLP output
}
__global__ __launch_bounds__(256) void psi_assemble_fwd_kernel(const float* __restrict__ xs, const float* __restrict__ var_out,
                                                               const float* __restrict__ tab_out, const float* __restrict__ special,
                                                               float* __restrict__ psi, int B, int T, int V, int E) {
    const int E4 = E / 4;
    const size_t n = (size_t)B * (T + 1) * (V + 1) * E4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int e4 = (int)(i % E4);
        size_t c = i / E4;
        const int v = (int)(c % (V + 1));
        c /= (V + 1);
        const int t = (int)(c % (T + 1)), b = (int)(c / (T + 1));
        const int kind = psi_cell_kind(xs, b, t, v, T, V);
        const float* src = kind == 0 ? var_out + (((size_t)v * B + b) * T + t) * E : kind == 1 ? tab_out + (size_t)b * E
                         : kind == 2 ? special : special + E;
        *(float4*)(psi + i * 4) = *(const float4*)(src + e4 * 4);
    }
}
// d_var_out (zero where overridden); per-batch partials of d_tab [B][E] and d_special [B][2][E] (summed over B by colsum)
__global__ __launch_bounds__(256) void psi_assemble_bwd_kernel(const float* __restrict__ xs, const float* __restrict__ dpsi, float* __restrict__ d_var,
                                                               float* __restrict__ d_tab, float* __restrict__ d_special_part, int B, int T, int V,
                                                               int E) {
    // Deterministic and coalesced: a thread = (cell lane, 16-B piece of the cell); it walks every NL-th cell of this batch element,
    // scatters d_var and keeps three private float4 sums (static column / masked / REP cells); the NL cell lanes are then added in
    // a fixed order through LDS.  (LDS float atomics, the first form, made the sums depend on the waves' timing; a one-thread-per-
    // sum walk over all cells was deterministic but took 264 us at cfg3.)
    // One workgroup per (batch element, SLICE of its cells): 64 workgroups walking 4753 cells each left three quarters of the chip
    // idle for 142 us; the slices' sums are separate output rows, added by the caller in a fixed order.
    extern __shared__ __attribute__((aligned(16))) float red[];          // [NL][3][E]
    const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y, E4 = E >> 2;
    const int NL = 256 / E4;                                              // cell lanes (42 at E = 24)
    const int cl = threadIdx.x / E4, e4 = threadIdx.x - cl * E4;
    const int cells = (T + 1) * (V + 1);
    const int per = (cells + S - 1) / S, c_lo = sl * per, c_hi = min(cells, c_lo + per);
    float4 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cl < NL) {
        for (int cidx = c_lo + cl; cidx < c_hi; cidx += NL) {
            const int t = cidx / (V + 1), v = cidx - t * (V + 1);
            const int kind = psi_cell_kind(xs, b, t, v, T, V);
            const float4 g = *(const float4*)(dpsi + ((size_t)b * cells + cidx) * E + e4 * 4);
            if (v < V && t < T)
                *(float4*)(d_var + (((size_t)v * B + b) * T + t) * E + e4 * 4) = kind == 0 ? g : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (kind == k + 1) acc[k] = make_float4(acc[k].x + g.x, acc[k].y + g.y, acc[k].z + g.z, acc[k].w + g.w);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) *(float4*)(red + ((size_t)cl * 3 + k) * E + e4 * 4) = acc[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * E) {
        const int k = threadIdx.x / E, e = threadIdx.x - k * E;
        float tot = 0.f;
        for (int l = 0; l < NL; ++l) tot += red[((size_t)l * 3 + k) * E + e];
        if (k == 0) d_tab[((size_t)b * S + sl) * E + e] = tot;
        else d_special_part[(((size_t)b * S + sl) * 2 + (k - 1)) * E + e] = tot;
    }
}

// plain axis swap of E-float cells: out[b][a2][a1][:] = in[b][a1][a2][:]
__global__ __launch_bounds__(256) void axis_swap_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int A1, int A2, int E4) {
    const size_t total = (size_t)B * A1 * A2 * E4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e4 = (int)(i % E4);
        size_t r = i / E4;
        const int a1 = (int)(r % A1);
        r /= A1;
        const int a2 = (int)(r % A2), b = (int)(r / A2);
        *(float4*)(out + i * 4) = *(const float4*)(in + ((((size_t)b * A1 + a1) * A2 + a2) * E4 + e4) * 4);
    }
}
// the swap with the positional embedding of the new leading axis added on the way (model :80-81, :90): out[b][a2][a1][:] = in[b][a1][a2][:] + emb,
//   MODE 1: emb = add[a2][a1][:]  (one table for every batch element: `full_event_embedding`)
//   MODE 3: emb = a2 < A2 - 1 ? add[b * (A2 - 1) + a2][a1][:] : add_last[a1][:]  (the per-sample time embedding rows and the REP row — the
//           reference concatenates them into [B, T + 1, tt] first; here that tensor is never built)
template <int MODE>
__global__ __launch_bounds__(256) void axis_swap_add_kernel(const float* __restrict__ in, const float* __restrict__ add, const float* __restrict__ add_last,
                                                            float* __restrict__ out, int B, int A1, int A2, int E4) {
    const size_t total = (size_t)B * A1 * A2 * E4;
    const size_t row4 = (size_t)A1 * E4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e4 = (int)(i % E4);
        size_t r = i / E4;
        const int a1 = (int)(r % A1);
        r /= A1;
        const int a2 = (int)(r % A2), b = (int)(r / A2);
        const float4 x = *(const float4*)(in + ((((size_t)b * A1 + a1) * A2 + a2) * E4 + e4) * 4);
        const size_t rest = (size_t)a1 * E4 + e4;
        const float* ep;
        if (MODE == 1) ep = add + ((size_t)a2 * row4 + rest) * 4;
        else ep = a2 < A2 - 1 ? add + (((size_t)b * (A2 - 1) + a2) * row4 + rest) * 4 : add_last + rest * 4;
        const float4 y = *(const float4*)ep;
        *(float4*)(out + i * 4) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
// out = a + b  (b broadcast over the leading batch dim when b_bs == 0)
__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t per_batch,
                                                        int B, int bcast) {
    const size_t n = per_batch * B;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        const float4 x = *(const float4*)(a + i);
        const float4 y = *(const float4*)(b + (bcast ? i % per_batch : i));
        *(float4*)(out + i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

int dw_chunks(int R) { return max(1, min(64, R / 96)); }

}  // namespace

extern "C" int medp_glinear_fwd(const float* x, const float* W, const float* b, float* y, int G, int R, int K, int N, void* stream) {
    MEDP_CHECK_ARG(x && W && y && G > 0 && R > 0 && K > 0 && N > 0 && (size_t)N * (K + 1) + N <= GLINEAR_MAX_FLOATS, "glinear_fwd: bad argument");
    MEDP_ONCE_PER_DEVICE({   // the per-pathology heads (256 -> 64) need 64.3 KB of weights in LDS: above the 64-KB default limit
        (void)hipFuncSetAttribute((const void*)glinear_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GLINEAR_MAX_FLOATS * 4);
    });
    glinear_fwd_kernel<<<dim3((int)min((size_t)1024, ((size_t)R * N + 255) / 256), G), 256, ((size_t)N * (K + 1) + N) * 4, (hipStream_t)stream>>>(x, W, b, y, R, K, N);
    MEDP_LAUNCH_CHECK("medp_glinear_fwd");
    return 0;
}
extern "C" size_t medp_glinear_bwd_workspace_bytes(int G, int R, int K, int N) { return (size_t)G * dw_chunks(R) * ((size_t)N * K + N) * 4; }
extern "C" int medp_glinear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, float* workspace, int G,
                                int R, int K, int N, void* stream) {
    MEDP_CHECK_ARG(dy && x && W && G > 0 && R > 0 && K > 0 && N > 0 && (size_t)N * K <= GLINEAR_MAX_FLOATS, "glinear_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        MEDP_ONCE_PER_DEVICE({
            (void)hipFuncSetAttribute((const void*)glinear_bwd_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GLINEAR_MAX_FLOATS * 4);
        });
        glinear_bwd_dx_kernel<<<dim3((int)min((size_t)1024, ((size_t)R * K + 255) / 256), G), 256, (size_t)N * K * 4, s>>>(dy, W, dx, R, K, N);
        MEDP_LAUNCH_CHECK("medp_glinear_bwd(dx)");
    }
    if (dW) {
        MEDP_CHECK_ARG(workspace && db, "glinear_bwd: dW needs db and a workspace");
        const int nc = dw_chunks(R), rpc = (R + nc - 1) / nc;
        float* pw = workspace;
        float* pb = workspace + (size_t)G * nc * N * K;
        MEDP_CHECK_ARG((size_t)rpc * (N + K) * 4 <= 150 * 1024, "glinear_bwd: rows-per-chunk x (N + K) floats must fit LDS");
        MEDP_ONCE_PER_DEVICE({
            (void)hipFuncSetAttribute((const void*)glinear_bwd_dw_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        });
        glinear_bwd_dw_partial_kernel<<<dim3(nc, G), 256, (size_t)rpc * (N + K) * 4, s>>>(dy, x, pw, pb, R, K, N, rpc);
        MEDP_LAUNCH_CHECK("medp_glinear_bwd(partial)");
        sum_chunks_kernel<<<dim3((N * K + 255) / 256, G), 256, 0, s>>>(pw, dW, nc, N * K);
        sum_chunks_kernel<<<dim3((N + 255) / 256, G), 256, 0, s>>>(pb, db, nc, N);
        MEDP_LAUNCH_CHECK("medp_glinear_bwd(final)");
    }
    return 0;
}
extern "C" size_t medp_gbn_workspace_bytes(int G, int R, int C) {
    if (G <= 0 || R <= 0 || C <= 0) return 0;
    return (size_t)G * gbn_chunks(R) * 2 * C * sizeof(float);
}
extern "C" int medp_gbn_fwd(const float* x, const float* w, const float* b, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_var, int G, int R, int C, float eps, float momentum, int batch_stats,
                            float* workspace, void* stream) {
    MEDP_CHECK_ARG(x && w && b && y && save_mean && save_var && G > 0 && R > 0 && C > 0, "gbn_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (batch_stats) {
        MEDP_CHECK_ARG(workspace, "gbn_fwd: batch statistics need a workspace (medp_gbn_workspace_bytes)");
        const int nc = gbn_chunks(R);
        gbn_stats_partial_kernel<<<dim3(nc, G, (C + 63) / 64), 256, 0, s>>>(x, workspace, R, C);
        gbn_stats_final_kernel<<<(G * C + 255) / 256, 256, 0, s>>>(x, workspace, save_mean, save_var, G, R, C, nc);
        MEDP_LAUNCH_CHECK("medp_gbn_fwd(stats)");
        if (running_mean && running_var) {
            gbn_running_kernel<<<(G * C + 255) / 256, 256, 0, s>>>(save_mean, save_var, running_mean, running_var, G * C, R, momentum);
            MEDP_LAUNCH_CHECK("medp_gbn_fwd(running)");
        }
    } else {
        MEDP_CHECK_ARG(running_mean && running_var, "gbn_fwd: eval mode needs running statistics");
        hipMemcpyAsync(save_mean, running_mean, (size_t)G * C * 4, hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(save_var, running_var, (size_t)G * C * 4, hipMemcpyDeviceToDevice, s);
    }
    gbn_apply_kernel<<<grid_for((size_t)G * R * C), 256, 0, s>>>(x, save_mean, save_var, w, b, y, G, R, C, eps);
    MEDP_LAUNCH_CHECK("medp_gbn_fwd(apply)");
    return 0;
}
extern "C" int medp_gbn_bwd(const float* dy, const float* x, const float* w, const float* save_mean, const float* save_var, float* dx,
                            float* dw, float* db, int G, int R, int C, float eps, int batch_stats, float* workspace, void* stream) {
    MEDP_CHECK_ARG(dy && x && w && save_mean && save_var && dx && dw && db && workspace && G > 0 && R > 0 && C > 0, "gbn_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nc = gbn_chunks(R);
    gbn_bwd_sums_partial_kernel<<<dim3(nc, G, (C + 63) / 64), 256, 0, s>>>(dy, x, save_mean, save_var, workspace, R, C, eps);
    gbn_bwd_sums_final_kernel<<<(G * C + 255) / 256, 256, 0, s>>>(workspace, db, dw, G, C, nc);
    MEDP_LAUNCH_CHECK("medp_gbn_bwd(sums)");
    gbn_bwd_dx_kernel<<<grid_for((size_t)G * R * C), 256, 0, s>>>(dy, x, save_mean, save_var, w, db, dw, dx, G, R, C, eps, batch_stats);
    MEDP_LAUNCH_CHECK("medp_gbn_bwd(dx)");
    return 0;
}
extern "C" int medp_act_fwd(const float* x, float* y, long long n, int mode, void* stream) {
    MEDP_CHECK_ARG(x && y && n > 0 && (mode == 0 || mode == 1), "act_fwd: bad argument");
    act_fwd_kernel<<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(x, y, (size_t)n, mode);
    MEDP_LAUNCH_CHECK("medp_act_fwd");
    return 0;
}
extern "C" int medp_act_bwd(const float* dy, const float* y, float* dx, long long n, int mode, void* stream) {
    MEDP_CHECK_ARG(dy && y && dx && n > 0 && (mode == 0 || mode == 1), "act_bwd: bad argument");
    act_bwd_kernel<<<grid_for((size_t)n), 256, 0, (hipStream_t)stream>>>(dy, y, dx, (size_t)n, mode);
    MEDP_LAUNCH_CHECK("medp_act_bwd");
    return 0;
}
extern "C" int medp_embed_inputs_fwd(const float* xs_ts, const float* n_obs_table, int table_rows, float* xin, int B, int T, int V, int KP, void* stream) {
    MEDP_CHECK_ARG(xs_ts && n_obs_table && xin && B > 0 && T > 0 && V > 0 && KP >= 2 && table_rows > 0 && table_rows <= 64, "embed_inputs_fwd: bad argument");
    embed_inputs_fwd_kernel<<<grid_for((size_t)V * B * T), 256, 0, (hipStream_t)stream>>>(xs_ts, n_obs_table, table_rows, xin, B, T, V, KP);
    MEDP_LAUNCH_CHECK("medp_embed_inputs_fwd");
    return 0;
}
extern "C" int medp_embed_inputs_bwd_blocks(int B, int T, int V) { return min(256, grid_for((size_t)V * B * T)); }
extern "C" int medp_embed_inputs_bwd(const float* xs_ts, const float* d_xin, float* partial /*[blocks][table_rows]*/, int table_rows, int B, int T,
                                     int V, int KP, void* stream) {
    MEDP_CHECK_ARG(xs_ts && d_xin && partial && table_rows > 0 && table_rows <= 16, "embed_inputs_bwd: bad argument (the n_obs table has at most 16 rows)");
    embed_inputs_bwd_kernel<<<medp_embed_inputs_bwd_blocks(B, T, V), 256, 0, (hipStream_t)stream>>>(xs_ts, d_xin, partial, table_rows, B, T, V, KP);
    MEDP_LAUNCH_CHECK("medp_embed_inputs_bwd");
    return 0;
}
extern "C" int medp_psi_assemble_fwd(const float* xs_ts, const float* var_out, const float* tab_out, const float* special, float* psi, int B,
                                     int T, int V, int E, void* stream) {
    MEDP_CHECK_ARG(xs_ts && var_out && tab_out && special && psi && E % 4 == 0, "psi_assemble_fwd: bad argument");
    psi_assemble_fwd_kernel<<<grid_for((size_t)B * (T + 1) * (V + 1) * E / 4), 256, 0, (hipStream_t)stream>>>(xs_ts, var_out, tab_out, special, psi, B, T, V, E);
    MEDP_LAUNCH_CHECK("medp_psi_assemble_fwd");
    return 0;
}
extern "C" int medp_psi_assemble_bwd_slices(int B, int T, int V) {
    (void)T; (void)V;
    return max(1, min(16, 512 / max(B, 1)));          // ~512 workgroups
}
extern "C" int medp_psi_assemble_bwd(const float* xs_ts, const float* dpsi, float* d_var_out, float* d_tab_partial /*[B][S][E]*/,
                                     float* d_special_partial /*[B][S][2][E]*/, int B, int T, int V, int E, void* stream) {
    MEDP_CHECK_ARG(xs_ts && dpsi && d_var_out && d_tab_partial && d_special_partial && E > 0 && E % 4 == 0 && 3 * E <= 256,
                   "psi_assemble_bwd: bad argument");
    psi_assemble_bwd_kernel<<<dim3(B, medp_psi_assemble_bwd_slices(B, T, V)), 256, (size_t)(256 / (E / 4)) * 3 * E * sizeof(float), (hipStream_t)stream>>>(
        xs_ts, dpsi, d_var_out, d_tab_partial, d_special_partial, B, T, V, E);
    MEDP_LAUNCH_CHECK("medp_psi_assemble_bwd");
    return 0;
}
extern "C" int medp_axis_swap(const float* in, float* out, int B, int A1, int A2, int E, void* stream) {
    MEDP_CHECK_ARG(in && out && B > 0 && A1 > 0 && A2 > 0 && E % 4 == 0, "axis_swap: bad argument");
    axis_swap_kernel<<<grid_for((size_t)B * A1 * A2 * E / 4), 256, 0, (hipStream_t)stream>>>(in, out, B, A1, A2, E / 4);
    MEDP_LAUNCH_CHECK("medp_axis_swap");
    return 0;
}
extern "C" int medp_axis_swap_add(const float* in, const float* add, const float* add_last, float* out, int B, int A1, int A2, int E, int mode,
                                  void* stream) {
    MEDP_CHECK_ARG(in && add && out && B > 0 && A1 > 0 && A2 > 0 && E % 4 == 0 && (mode == 1 || (mode == 3 && add_last && A2 > 1)),
                   "axis_swap_add: bad argument");
    const int grid = grid_for((size_t)B * A1 * A2 * E / 4);
    if (mode == 1) axis_swap_add_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(in, add, add_last, out, B, A1, A2, E / 4);
    else axis_swap_add_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(in, add, add_last, out, B, A1, A2, E / 4);
    MEDP_LAUNCH_CHECK("medp_axis_swap_add");
    return 0;
}
extern "C" int medp_add_bcast(const float* a, const float* b, float* out, long long per_batch, int B, int broadcast_b, void* stream) {
    MEDP_CHECK_ARG(a && b && out && per_batch > 0 && per_batch % 4 == 0 && B > 0, "add_bcast: bad argument");
    add_bcast_kernel<<<grid_for((size_t)per_batch * B / 4), 256, 0, (hipStream_t)stream>>>(a, b, out, (size_t)per_batch, B, broadcast_b);
    MEDP_LAUNCH_CHECK("medp_add_bcast");
    return 0;
}
