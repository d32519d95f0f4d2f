// DuETT (dual-axis event x time transformer) encode path for gfx950, inference form (BatchNorm folded to an affine):
//   DuettFeatureExtractor.encode, reference models/main_architecture_duett.py:31-94 (== duett/duett.py:245-280).
//
// psi [B, T+1, V+1, E] fp32 is built by ONE launch (the reference runs V sequential 5-op MLPs, ~250 eager launches):
// a workgroup handles 256 (batch, time) cells of one variable with that variable's MLP weights broadcast from LDS.
// The layer loop alternates the two token views of psi
//     event view [B, V+1, (T+1)E]  <->  time view [B, T+1, (V+1)E]
// through "axis swap + add" kernels that move whole E-float (96 B) cells with 16-B accesses and fuse the positional
// add and the preceding encoder's final ScaleNorm scaling (per-row 1/||x||), so psi makes one HBM round trip per swap.
// Each encoder (x_transformers Encoder(depth=1), restated in oracle/xt_encoder.py) is
//     x += to_out(attn(ScaleNorm(x)));  x += ff2(gelu(ff1(ScaleNorm(x))));  [x = ScaleNorm(x)]
// with ScaleNorm emitting bf16 straight into the MFMA GEMMs and both residual adds fused in GEMM epilogues.
#include "common.h"
#include "medp_hip.h"

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline int grid_for(size_t work_items) { return (int)min((size_t)4096, max((size_t)1, (work_items + 255) / 256)); }

// ---- static (tabular) encoder: Linear(Ds,128) -> ReLU -> BN(eval affine) -> Linear(128,E);  one block per sample ----
__global__ __launch_bounds__(128) void tab_encoder_kernel(const float* __restrict__ xs, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, const float* __restrict__ s,
                                                          const float* __restrict__ sh, const float* __restrict__ w4,
                                                          const float* __restrict__ b4, float* __restrict__ out, int Ds, int Hd, int E) {
    extern __shared__ float hid[];
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        float a = b0[j];
        for (int i = 0; i < Ds; ++i) a += w0[j * Ds + i] * xs[(size_t)b * Ds + i];
        hid[j] = fmaxf(a, 0.f) * s[j] + sh[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float a = b4[e];
        for (int j = 0; j < Hd; ++j) a += w4[e * Hd + j] * hid[j];
        out[(size_t)b * E + e] = a;
    }
}

// ---- fused psi build (K2-K5 of SURVEY.md §2.2) -----------------------------------------------------------------------
// grid (ceil(B*(T+1)/256), V+1); thread = one (b, t) cell of variable slot v = blockIdx.y.  E <= 32, hidden <= 64.
template <int E, int HD>
__global__ __launch_bounds__(256) void psi_embed_kernel(const float* __restrict__ xs_ts, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ bs,
                                                        const float* __restrict__ bsh, const float* __restrict__ w4,
                                                        const float* __restrict__ b4, const float* __restrict__ nobs_table,
                                                        int nobs_rows, const float* __restrict__ tab_out,
                                                        const float* __restrict__ special, float* __restrict__ psi, int B, int T, int V) {
    __shared__ float sw0[HD * 2], sb0[HD], ss[HD], ssh[HD], sw4[E * HD], sb4[E];
    const int v = blockIdx.y;
    if (v < V) {
        for (int i = threadIdx.x; i < HD * 2; i += 256) sw0[i] = w0[(size_t)v * HD * 2 + i];
        for (int i = threadIdx.x; i < HD; i += 256) {
            sb0[i] = b0[(size_t)v * HD + i];
            ss[i] = bs[(size_t)v * HD + i];
            ssh[i] = bsh[(size_t)v * HD + i];
        }
        for (int i = threadIdx.x; i < E * HD; i += 256) sw4[i] = w4[(size_t)v * E * HD + i];
        for (int i = threadIdx.x; i < E; i += 256) sb4[i] = b4[(size_t)v * E + i];
    }
    __syncthreads();
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= B * (T + 1)) return;
    const int b = cell / (T + 1), t = cell % (T + 1);
    const int F = 2 * V + 1;
    float out[E];
    const float* src = nullptr;   // whole-vector overrides
    if (t == T) {
        src = special + E;                                   // REP row: special_embeddings[1]         (model :58-60)
    } else {
        const float* row = xs_ts + ((size_t)b * T + t) * F;
        if (row[2 * V] == 1.0f) {
            src = special;                                   // masked timestep: special_embeddings[0] (model :61-64)
        } else if (v == V) {
            src = tab_out + (size_t)b * E;                   // static column                          (model :57)
        } else {
            const float cnt = row[V + v];
            if (cnt == -1.0f) {
                src = special;                               // masked event (SSL only)                (model :65-66)
            } else {
                const int idx = min(max((int)cnt, 0), nobs_rows - 1);          // .to(int).clip(0, 15) (model :41)
                const float val = row[v], nob = nobs_table[idx];
#pragma unroll
                for (int e = 0; e < E; ++e) out[e] = sb4[e];
#pragma unroll 8
                for (int j = 0; j < HD; ++j) {
                    const float h = fmaxf(sw0[2 * j] * val + sw0[2 * j + 1] * nob + sb0[j], 0.f) * ss[j] + ssh[j];
#pragma unroll
                    for (int e = 0; e < E; ++e) out[e] += sw4[e * HD + j] * h;
                }
            }
        }
    }
    float* dst = psi + (((size_t)b * (T + 1) + t) * (V + 1) + v) * E;
    if (src) {
#pragma unroll
        for (int e = 0; e < E; e += 4) *(float4*)(dst + e) = *(const float4*)(src + e);
    } else {
#pragma unroll
        for (int e = 0; e < E; e += 4) *(float4*)(dst + e) = make_float4(out[e], out[e + 1], out[e + 2], out[e + 3]);
    }
}

// ---- time embedding (K6): cve(batch_norm) = Linear(1,h) -> tanh -> BN affine -> Linear(h, tt) ; REP row appended ------
// 16 rows (b, t) per workgroup and one output column per thread: a column of the second Linear is read once per 16 rows and the
// hidden activations come from LDS by broadcast.  (One row per workgroup re-read the whole 34 x 1176 weight for every row:
// 127 us for a 29-MB result.)  Per output the products are still summed in ascending j: results unchanged.
constexpr int TE_ROWS = 16;
__global__ __launch_bounds__(256) void time_embed_kernel(const float* __restrict__ times, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ s,
                                                         const float* __restrict__ sh, const float* __restrict__ w3,
                                                         const float* __restrict__ b3, const float* __restrict__ rep,
                                                         float* __restrict__ out, int B, int T, int Hd, int tt) {
    extern __shared__ __attribute__((aligned(16))) float hid[];              // [Hd][TE_ROWS]
    const int rows = B * (T + 1);
    const int r0 = blockIdx.x * TE_ROWS;
    for (int idx = threadIdx.x; idx < TE_ROWS * Hd; idx += 256) {
        const int rr = idx / Hd, j = idx - rr * Hd, row = r0 + rr;
        float h = 0.f;
        if (row < rows) {
            const int b = row / (T + 1), t = row - b * (T + 1);
            if (t < T) h = fmaf(tanhf(fmaf(w0[j], times[(size_t)b * T + t], b0[j])), s[j], sh[j]);      // (explicit fmaf: the same in time_embed_mfma_kernel)
        }
        hid[j * TE_ROWS + rr] = h;             // [Hd][TE_ROWS]: the 16 rows of one hidden unit are one 64-B broadcast read
    }
    __syncthreads();
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= tt) return;
    float acc[TE_ROWS];
    const float bias = b3[c];
#pragma unroll
    for (int rr = 0; rr < TE_ROWS; ++rr) acc[rr] = bias;
    // w3t is the TRANSPOSED weight [Hd][tt]: consecutive threads read consecutive addresses
    // (eight weight loads in flight: one per iteration made the loop a chain of L2 round trips)
    int j = 0;
    for (; j + 8 <= Hd; j += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = w3[(size_t)(j + u) * tt + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4* hj = (const float4*)(hid + (j + u) * TE_ROWS);
#pragma unroll
            for (int q = 0; q < TE_ROWS / 4; ++q) {
                const float4 h4 = hj[q];
                acc[4 * q] = fmaf(w[u], h4.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w[u], h4.y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(w[u], h4.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w[u], h4.w, acc[4 * q + 3]);
            }
        }
    }
    for (; j < Hd; ++j) {
        const float w = w3[(size_t)j * tt + c];
#pragma unroll
        for (int rr = 0; rr < TE_ROWS; ++rr) acc[rr] = fmaf(w, hid[j * TE_ROWS + rr], acc[rr]);
    }
    const float repc = rep[c];
#pragma unroll
    for (int rr = 0; rr < TE_ROWS; ++rr) {
        const int row = r0 + rr;
        if (row < rows) out[(size_t)row * tt + c] = (row % (T + 1) == T) ? repc : acc[rr];     // REP row appended
    }
}

// ---- time embedding on the matrix cores, in exact fp32 --------------------------------------------------------------------------------
// out[row][c] = b3[c] + sum_j w3t[j][c] * hid[row][j] is a [B(T+1) x Hd] x [Hd x tt] product (6208 x 34 x 1176 at cfg3): 0.5 GFLOP behind a
// 29-MB store.  v_mfma_f32_16x16x4_f32 takes fp32 operands and its result is bit for bit the k-ordered fmaf chain the VALU kernel above
// computes (C-in = the bias, products added in ascending j; the rows of k past Hd are zeros: fma(0, 0, acc) = acc), so the fp32 kernel
// mode needs no second form.  A wave owns 16 rows: each lane computes 9 of their 16 x 36 hidden activations (the A fragments, kept in
// registers for the whole launch) and walks its share of the 16-column tiles; the B fragments come straight from the L2-resident weight.
constexpr int TEM_KS = 9;                                    // k-steps of 4: hidden widths up to 36
constexpr int TEM_CW = 208;                                  // columns per workgroup (13 tiles of 16); 208 floats = 6.5 x 32 banks: the two
                                                             // 16-lane k-groups of a 32-lane half read disjoint bank sets (conflict-free ds_read_b32)
__global__ __launch_bounds__(256) void time_embed_mfma_kernel(const float* __restrict__ times, const float* __restrict__ w0,
                                                              const float* __restrict__ b0, const float* __restrict__ s,
                                                              const float* __restrict__ sh, const float* __restrict__ w3t,
                                                              const float* __restrict__ b3, const float* __restrict__ rep,
                                                              float* __restrict__ out, int B, int T, int Hd, int tt) {
    // the workgroup's slice of the second Linear, [4 TEM_KS][TEM_CW] + bias + REP rows, staged once (coalesced) and read back as B
    // fragments by all four waves for every column tile: the first form loaded them from L2 tile by tile — 13 dependent ~1-us round
    // trips per wave, 18.7 us for a launch whose matrix work is 3.5 us
    __shared__ __attribute__((aligned(16))) float sw[(4 * TEM_KS + 2) * TEM_CW];
    const int lane = threadIdx.x & 63, rr = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rows = B * (T + 1);
    const int c0 = blockIdx.y * TEM_CW, ncol = min(TEM_CW, tt - c0);
    {   // 38 rows of 52 float4 (the tail rows: zeros, bias, REP); every thread's 8 loads are independent and issued together — one load
        // per loop iteration made the staging a chain of 31 L2 round trips (23 us for the launch)
        constexpr int C4 = TEM_CW / 4, N4 = (4 * TEM_KS + 2) * C4;
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i4 = threadIdx.x + 256 * u, j = i4 / C4, cc = (i4 - j * C4) * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i4 < N4 && cc < ncol) {                    // (tt and the chunk width are multiples of 4: a float4 is all in or all out)
                const float* src = j < Hd ? w3t + (size_t)j * tt : (j == 4 * TEM_KS ? b3 : (j == 4 * TEM_KS + 1 ? rep : nullptr));
                if (src) v[u] = *(const float4*)(src + c0 + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i4 = threadIdx.x + 256 * u;
            if (i4 < N4) *(float4*)(sw + 4 * i4) = v[u];
        }
    }
    const int r0 = (blockIdx.x * 4 + wave) * 16;
    float a[TEM_KS];
    {
        const int row = r0 + rr;
        const int b = row / (T + 1), t = row - b * (T + 1);
        const bool live = row < rows && t < T;
        const float tv = live ? times[(size_t)b * T + t] : 0.f;
#pragma unroll
        for (int ks = 0; ks < TEM_KS; ++ks) {
            const int j = 4 * ks + kk;
            a[ks] = (live && j < Hd) ? fmaf(tanhf(fmaf(w0[j], tv, b0[j])), s[j], sh[j]) : 0.f;
        }
    }
    __syncthreads();
    if (r0 >= rows) return;
    bool is_rep[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) is_rep[r] = (r0 + kk * 4 + r) % (T + 1) == T;
    const int ntile = (ncol + 15) >> 4;
    for (int ct = 0; ct < ntile; ++ct) {
        const int cc = ct * 16 + rr;
        const float bias = sw[4 * TEM_KS * TEM_CW + cc], repc = sw[(4 * TEM_KS + 1) * TEM_CW + cc];
        f32x4 acc = (f32x4){bias, bias, bias, bias};
#pragma unroll
        for (int ks = 0; ks < TEM_KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], sw[(4 * ks + kk) * TEM_CW + cc], acc, 0, 0, 0);
        if (cc < ncol) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + kk * 4 + r;
                if (row < rows) out[(size_t)row * tt + c0 + cc] = is_rep[r] ? repc : acc[r];          // REP row appended
            }
        }
    }
}

// MEDP_EMBED_MFMA=0 (or medp_dbg_embed_mfma(0), tests): the VALU forms of the two embedding kernels, for A/B runs and bit comparisons
int g_embed_mfma = -1;
static int embed_mfma_on() {
    if (g_embed_mfma < 0) { const char* e = getenv("MEDP_EMBED_MFMA"); g_embed_mfma = e ? (atoi(e) != 0) : 1; }
    return g_embed_mfma;
}

static int launch_time_embed(const MedpDuettWeights* w, const float* xs_times, float* temb, int B, int T, hipStream_t s) {
    const int T1 = T + 1, tt = w->d_embedding * (w->n_vars + 1);
    const int mfma_on = embed_mfma_on();
    if (mfma_on && w->d_hidden_time <= 4 * TEM_KS) {
        time_embed_mfma_kernel<<<dim3((B * T1 + 63) / 64, (tt + TEM_CW - 1) / TEM_CW), 256, 0, s>>>(
            xs_times, (const float*)w->time_w0, (const float*)w->time_b0, (const float*)w->time_bn_scale, (const float*)w->time_bn_shift,
            (const float*)w->time_w3t, (const float*)w->time_b3, (const float*)w->rep_embedding, temb, B, T, w->d_hidden_time, tt);
    } else {
        time_embed_kernel<<<dim3((B * T1 + TE_ROWS - 1) / TE_ROWS, (tt + 255) / 256), 256, TE_ROWS * w->d_hidden_time * sizeof(float), s>>>(
            xs_times, (const float*)w->time_w0, (const float*)w->time_b0, (const float*)w->time_bn_scale, (const float*)w->time_bn_shift,
            (const float*)w->time_w3t, (const float*)w->time_b3, (const float*)w->rep_embedding, temb, B, T, w->d_hidden_time, tt);
    }
    MEDP_LAUNCH_CHECK("duett time_embed");
    return 0;
}

// ---- axis swaps (K7): cells of E floats move between [B, A1, A2, E] and [B, A2, A1, E] ------------------------------------
// out[b][a2][a1][:] = in[b][a1][a2][:] * rowscale(b, a1) + add        add: either per-(a2,a1,e) table (event embedding,
// batch-invariant, add_bs = 0) or per-(b,a2,a1,e) tensor (time embedding, add_bs = A2*A1*E).  rowscale = rnorm * gain
// applies the producing encoder's final ScaleNorm (row = one (b, a1) token of A2*E features); rnorm == nullptr -> 1.
__global__ __launch_bounds__(256) void axis_swap_add_kernel(const float* __restrict__ in, const float* __restrict__ rnorm,
                                                            const float* __restrict__ g, float gain_sqrt_dim,
                                                            const float* __restrict__ add, long long add_bs,
                                                            float* __restrict__ out, int B, int A1, int A2, int E4) {
    // 32-bit index arithmetic (the launcher checks B*A1*A2*E4 < 2^31): the 64-bit divisions of the first version cost more than
    // the memory traffic (33 us for 87 MB)
    const unsigned total = (unsigned)B * A1 * A2 * E4;
    const float gain = rnorm ? gain_sqrt_dim * g[0] : 1.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        // iterate in OUTPUT order (coalesced writes): i = ((b*A2 + a2)*A1 + a1)*E4 + e4
        const int e4 = (int)(i % (unsigned)E4);
        unsigned r = i / (unsigned)E4;
        const int a1 = (int)(r % (unsigned)A1);
        r /= (unsigned)A1;
        const int a2 = (int)(r % (unsigned)A2);
        const int b = (int)(r / (unsigned)A2);
        const float4 v = *(const float4*)(in + ((((size_t)b * A1 + a1) * A2 + a2) * E4 + e4) * 4);
        const float sc = rnorm ? rnorm[(size_t)b * A1 + a1] * gain : 1.f;
        const float4 ad = *(const float4*)(add + (size_t)b * add_bs + (((size_t)a2 * A1 + a1) * E4 + e4) * 4);
        *(float4*)(out + (size_t)i * 4) = make_float4(v.x * sc + ad.x, v.y * sc + ad.y, v.z * sc + ad.z, v.w * sc + ad.w);
    }
}

// ---- layer-0 fusion: psi build + (time view -> event view) + event embedding + the event encoder's first ScaleNorm -----------
// One workgroup per event-view row (b, v): its T+1 cells are CONTIGUOUS in the event view ((T+1)*E floats), so the row is
// assembled in LDS and leaves in whole-row coalesced 16-B stores — the time-view psi kernel above writes 96-B cells 4.7 KB apart.
// It replaces psi_embed + axis_swap_add + scalenorm of layer 0 (psi0 is never written; reads 2.4 MB, writes |psi| fp32 + bf16).
// The row's sum of squares is taken by wave 0 in the lane / chunk order of scalenorm_fwd_reg_kernel, so `h` is bit-identical to
// what the separate ScaleNorm launch produced.
struct TabWeights {
    const float *xs, *w0, *b0, *s, *sh, *w4, *b4;
    int Ds, Hd;
};
// PE_NB = batch elements (event-view rows of one variable) per workgroup of the fused embed kernel = cells per lane
template <int E, int HD, int PE_NB>
__global__ __launch_bounds__(128) void psi_embed_event_kernel(const float* __restrict__ xs_ts, const float* __restrict__ l0,
                                                              const float* __restrict__ w4t, const float* __restrict__ b4,
                                                              const float* __restrict__ nobs_table, int nobs_rows, const TabWeights tw,
                                                              const float* __restrict__ special,
                                                              const float* __restrict__ event_emb, const float* __restrict__ g_norm,
                                                              float norm_eps, float* __restrict__ xe, bf16_t* __restrict__ h,
                                                              float* __restrict__ psi0_out, int B, int T, int V, int use_mfma) {
    // A workgroup = one variable v x PE_NB consecutive batch elements (PE_NB cells per lane, one per batch element); the
    // variable's MLP (7.5 KB) is staged in LDS once per workgroup and read back by broadcast 16-B reads.  What bounds it: every
    // lane needs the same 29 floats per hidden unit, and a wave-wide LDS read returns 1 KB through the CU's 128-B/clk LDS
    // return path whether or not the lanes share an address: 64 hidden units x 8 reads x 8 clocks per wave = 38 us of LDS time
    // per CU at cfg3 (measured 44 us with PE_NB = 1).  Tried and measured: weights by scalar loads into SGPRs (48 us, 79 % of
    // the wave cycles parked in s_waitcnt: SMEM returns out of order, so only lgkmcnt(0) is available and the loads of a hidden
    // unit cannot be overlapped with the previous one's FMAs); 2 / 4 cells per lane to amortise the reads (53 / 77 us: the
    // launch then has 3 / 1.5 waves per SIMD and the long waves no longer hide each other's latency).
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [PE_NB][(T+1)*E] the rows
    __shared__ __attribute__((aligned(16))) float sl0[HD * 8], sw4[HD * E], sb4[E];
    __shared__ float s_rn[PE_NB], s_tab[PE_NB][E], s_hid[256], s_part[2][E], s_nobs[64];
    const int v = blockIdx.y, b0 = blockIdx.x * PE_NB;
    const int nb = min(PE_NB, B - b0);
    const int T1 = T + 1, F = 2 * V + 1, D = T1 * E, D4 = D >> 2;
    if (threadIdx.x < 64) s_nobs[threadIdx.x] = nobs_table[min((int)threadIdx.x, nobs_rows - 1)];
    if (v < V) {
        for (int i = threadIdx.x; i < HD * 8 / 4; i += 128) ((float4*)sl0)[i] = ((const float4*)(l0 + (size_t)v * HD * 8))[i];
        for (int i = threadIdx.x; i < HD * E / 4; i += 128) ((float4*)sw4)[i] = ((const float4*)(w4t + (size_t)v * HD * E))[i];
        if (threadIdx.x < E) sb4[threadIdx.x] = b4[(size_t)v * E + threadIdx.x];
    } else {
        // the static column's rows: each sample's tab_encoder output (Linear(Ds,Hd) -> ReLU -> BN affine -> Linear(Hd,E), model :57,
        // duett.py:124-125) computed here, one hidden unit per thread — no separate launch in front of the psi build
        for (int bb = 0; bb < nb; ++bb) {
            for (int j = threadIdx.x; j < tw.Hd; j += 128) {
                float a = tw.b0[j];
                for (int i = 0; i < tw.Ds; ++i) a += tw.w0[j * tw.Ds + i] * tw.xs[(size_t)(b0 + bb) * tw.Ds + i];
                s_hid[j] = fmaxf(a, 0.f) * tw.s[j] + tw.sh[j];
            }
            __syncthreads();
            // second Linear: lane j multiplies hidden unit j (coalesced weight reads), wave tree + the two waves' halves per output
            // (a thread per output walked its 128 weights 512 B apart: 12 us per sample, the long pole of the whole launch)
            for (int e = 0; e < E; ++e) {
                float pa = 0.f;
                for (int j = threadIdx.x; j < tw.Hd; j += 128) pa += tw.w4[e * tw.Hd + j] * s_hid[j];
                pa = wave_sum(pa);
                if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][e] = pa;
            }
            __syncthreads();
            if (threadIdx.x < E) s_tab[bb][threadIdx.x] = tw.b4[threadIdx.x] + s_part[0][threadIdx.x] + s_part[1][threadIdx.x];
            __syncthreads();
        }
    }
    __syncthreads();
    // (whole waves enter: the matrix-core path below needs all 64 lanes; t >= T1 is "no cell" and is never stored)
    for (int t = threadIdx.x; t < ((T1 + 63) & ~63); t += 128) {
        float out[PE_NB][E];
        float val[PE_NB], nob[PE_NB];
        const float* src[PE_NB];
        bool any_mlp = false;
        // every input this lane needs, loaded up front and unconditionally (12 independent loads in flight): behind the
        // data-dependent branches below they formed a chain of three L2 round trips per cell, four cells in a row
        float msk[PE_NB], cnt[PE_NB];
#pragma unroll
        for (int bb = 0; bb < PE_NB; ++bb) {
            msk[bb] = 1.f; cnt[bb] = -1.f; val[bb] = 0.f;
            if (bb < nb && t < T) {
                const float* row = xs_ts + ((size_t)(b0 + bb) * T + t) * F;
                msk[bb] = row[2 * V];
                if (v < V) { cnt[bb] = row[V + v]; val[bb] = row[v]; }
            }
        }
#pragma unroll
        for (int bb = 0; bb < PE_NB; ++bb) {
            src[bb] = nullptr;
            nob[bb] = 0.f;
            if (bb >= nb || t >= T1) { src[bb] = special; continue; }  // no such row / cell: never stored
            if (t == T) {
                src[bb] = special + E;                                   // REP row                                 (model :58-60)
            } else if (msk[bb] == 1.0f) {
                src[bb] = special;                                       // masked timestep                         (model :61-64)
            } else if (v == V) {
                src[bb] = s_tab[bb];                                     // static column                           (model :57)
            } else if (cnt[bb] == -1.0f) {
                src[bb] = special;                                       // masked event (SSL only)                 (model :65-66)
            } else {
                nob[bb] = s_nobs[min(max((int)cnt[bb], 0), nobs_rows - 1)];            // .to(int).clip(0, 15) (model :41)
                any_mlp = true;
            }
        }
        // ---- the variable's MLP on the matrix cores, in exact fp32 (PE_NB = 1): out[cell][e] = b4[e] + sum_j w4[e][j] h_j(cell) is a
        // [cells x 64] x [64 x 24] product per variable.  v_mfma_f32_16x16x4_f32 is bit for bit the ascending-j fmaf chain of the VALU
        // form below (C-in = b4), so results are unchanged and the fp32 kernel mode needs no second form.  A wave takes the cells of ITS
        // lanes in groups of 16 (t = 16 g .. 16 g + 15); a lane computes 16 of the group's 16 x 64 hidden activations (2 FMA + max + FMA
        // each: the A fragments), the second Linear's weights are the B fragments (read once per wave: 32 registers), and the VALU form's
        // 8 broadcast LDS reads per hidden unit and cell — what bound it (44 us) — are gone.
        bool mfma_done = false;
        if constexpr (PE_NB == 1) {
            if (use_mfma && v < V) {
                mfma_done = true;
                const int lane = threadIdx.x & 63, rr = lane & 15, kk = lane >> 4;
                const int tbase = t & ~63;                                    // first cell of this wave in this pass
                const unsigned long long need = __ballot(any_mlp);
                float bw0[16], bw1[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    bw0[ks] = sw4[(4 * ks + kk) * E + rr];
                    bw1[ks] = rr < E - 16 ? sw4[(4 * ks + kk) * E + 16 + rr] : 0.f;
                }
                const float c0 = sb4[rr], c1 = rr < E - 16 ? sb4[16 + rr] : 0.f;
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    if (tbase + 16 * g >= T) break;                            // no time step in this group (wave-uniform)
                    if (((need >> (16 * g)) & 0xffffull) == 0) continue;       // every cell of the group is an override row
                    const float vg = __shfl(val[0], 16 * g + rr, 64), ng = __shfl(nob[0], 16 * g + rr, 64);
                    f32x4 a0 = (f32x4){c0, c0, c0, c0}, a1 = (f32x4){c1, c1, c1, c1};
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const int j = 4 * ks + kk;
                        const float4 la = *(const float4*)(sl0 + j * 8);
                        const float hj = fmaf(fmaxf(fmaf(la.x, vg, fmaf(la.y, ng, la.z)), 0.f), la.w, sl0[j * 8 + 4]);   // (as the VALU form below)
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hj, bw0[ks], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hj, bw1[ks], a1, 0, 0, 0);
                    }
                    // C layout: column e = rr (+16), rows = cells 4 kk + r of the group
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int tc = tbase + 16 * g + 4 * kk + r;
                        if (tc < T1) {
                            tile[tc * E + rr] = a0[r];
                            if (rr < E - 16) tile[tc * E + 16 + rr] = a1[r];
                        }
                    }
                }
                // the wave's own LDS operations execute in order: the override rows below (same wave, same cells) land after these
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
            }
        }
        if (!mfma_done && v < V && __any(any_mlp)) {
#pragma unroll
            for (int bb = 0; bb < PE_NB; ++bb)
#pragma unroll
                for (int e = 0; e < E; ++e) out[bb][e] = sb4[e];
#pragma unroll 2
            for (int j = 0; j < HD; ++j) {
                const float4 la = *(const float4*)(sl0 + j * 8);        // w0[j][0], w0[j][1], b0[j], bn scale
                const float lsh = sl0[j * 8 + 4];                        // bn shift
                float hj[PE_NB];
#pragma unroll
                for (int bb = 0; bb < PE_NB; ++bb) hj[bb] = fmaf(fmaxf(fmaf(la.x, val[bb], fmaf(la.y, nob[bb], la.z)), 0.f), la.w, lsh);   // explicit fmaf: both forms round alike
                const float4* wj = (const float4*)(sw4 + j * E);
#pragma unroll
                for (int e4 = 0; e4 < E / 4; ++e4) {
                    const float4 w = wj[e4];
#pragma unroll
                    for (int bb = 0; bb < PE_NB; ++bb) {
                        out[bb][4 * e4] = fmaf(w.x, hj[bb], out[bb][4 * e4]); out[bb][4 * e4 + 1] = fmaf(w.y, hj[bb], out[bb][4 * e4 + 1]);
                        out[bb][4 * e4 + 2] = fmaf(w.z, hj[bb], out[bb][4 * e4 + 2]); out[bb][4 * e4 + 3] = fmaf(w.w, hj[bb], out[bb][4 * e4 + 3]);
                    }
                }
            }
        }
#pragma unroll
        for (int bb = 0; bb < PE_NB; ++bb) {
            if (bb >= nb || t >= T1) continue;
            if (src[bb]) {
#pragma unroll
                for (int e = 0; e < E; ++e) out[bb][e] = src[bb][e];
            } else if (mfma_done) {                                 // the cell's row is in the tile already (written by this wave): take it back
#pragma unroll                                                        // for the psi0 copy, and rewrite it unchanged below
                for (int e = 0; e < E; ++e) out[bb][e] = tile[(size_t)bb * D + t * E + e];
            }
            if (psi0_out) {                                          // parity checks only: psi0 in the time view, before the add
                float* d0 = psi0_out + (((size_t)(b0 + bb) * T1 + t) * (V + 1) + v) * E;
#pragma unroll
                for (int e = 0; e < E; e += 4) *(float4*)(d0 + e) = make_float4(out[bb][e], out[bb][e + 1], out[bb][e + 2], out[bb][e + 3]);
            }
#pragma unroll
            for (int e = 0; e < E; e += 4)
                *(float4*)(tile + (size_t)bb * D + t * E + e) = make_float4(out[bb][e], out[bb][e + 1], out[bb][e + 2], out[bb][e + 3]);
        }
    }
    __syncthreads();
    // + event embedding (row v of [V+1, (T+1)E]) ; fp32 rows out, coalesced
    for (int bb = 0; bb < nb; ++bb) {
        const size_t rbase = ((size_t)(b0 + bb) * (V + 1) + v) * D;
        float* tb = tile + (size_t)bb * D;
        for (int i = threadIdx.x; i < D4; i += 128) {
            float4 x = *(float4*)(tb + 4 * i);
            const float4 a = *(const float4*)(event_emb + (size_t)v * D + 4 * i);
            x = make_float4(x.x + a.x, x.y + a.y, x.z + a.z, x.w + a.w);
            *(float4*)(tb + 4 * i) = x;
            *(float4*)(xe + rbase + 4 * i) = x;
        }
    }
    __syncthreads();
    {   // row norms: wave w takes rows w, w+2 in the canonical order (lane + 64 k, then the wave tree) of scalenorm_fwd_reg_kernel
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int bb = wave; bb < nb; bb += 2) {
            const float* tb = tile + (size_t)bb * D;
            float ss = 0.f;
            for (int i = lane; i < D4; i += 64) {
                const float4 x = *(const float4*)(tb + 4 * i);
                ss += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
            }
            const float rn = 1.0f / fmaxf(sqrtf(wave_sum(ss)), norm_eps);
            if (lane == 0) s_rn[bb] = rn;
        }
    }
    __syncthreads();
    for (int bb = 0; bb < nb; ++bb) {
        const size_t rbase = ((size_t)(b0 + bb) * (V + 1) + v) * D;
        const float* tb = tile + (size_t)bb * D;
        const float sc = s_rn[bb] * sqrtf((float)D) * g_norm[0];
        for (int i = threadIdx.x; i < D4; i += 128) {
            const float4 x = *(const float4*)(tb + 4 * i);
            uint2 o;
            o.x = pack_bf2(x.x * sc, x.y * sc);
            o.y = pack_bf2(x.z * sc, x.w * sc);
            *(uint2*)(h + rbase + 4 * i) = o;
        }
    }
}

// ---- axis swap + positional add + the NEXT encoder's first ScaleNorm in one pass (wave per output row, row in registers) --------
// x[b][a2][a1][:] = in[b][a1][a2][:] * rowscale(b, a1) + add ;  h = ScaleNorm(x) as bf16.  One read of psi, one fp32 + one bf16
// write, instead of swap (read + write) then ScaleNorm (read + write).  Lane / chunk order of scalenorm_fwd_reg_kernel: bit-identical.
template <int NV>
__global__ __launch_bounds__(256) void swap_add_norm_kernel(const float* __restrict__ in, const float* __restrict__ rnorm,
                                                            const float* __restrict__ g_prev, float gain_sqrt_dim,
                                                            const float* __restrict__ add, long long add_bs,
                                                            const float* __restrict__ g_norm, float norm_eps, float* __restrict__ x_out,
                                                            bf16_t* __restrict__ h_out, int B, int A1, int A2, int E4) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * A2) return;
    const int b = row / A2, a2 = row - b * A2;
    const int D4 = A1 * E4, D = D4 * 4;
    const float gain = rnorm ? gain_sqrt_dim * g_prev[0] : 1.f;
    const float* addr = add + (size_t)b * add_bs + (size_t)a2 * D;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < D4) {
            const int a1 = i / E4, e4 = i - a1 * E4;
            const float4 s = *(const float4*)(in + ((((size_t)b * A1 + a1) * A2 + a2) * E4 + e4) * 4);
            const float sc = rnorm ? rnorm[(size_t)b * A1 + a1] * gain : 1.f;
            const float4 ad = *(const float4*)(addr + 4 * i);
            v[k] = make_float4(s.x * sc + ad.x, s.y * sc + ad.y, s.z * sc + ad.z, s.w * sc + ad.w);
            *(float4*)(x_out + (size_t)row * D + 4 * i) = v[k];
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < D4) ss += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    const float rn = 1.0f / fmaxf(sqrtf(wave_sum(ss)), norm_eps);
    const float sc = rn * sqrtf((float)D) * g_norm[0];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < D4) {
            uint2 o;
            o.x = pack_bf2(v[k].x * sc, v[k].y * sc);
            o.y = pack_bf2(v[k].z * sc, v[k].w * sc);
            *(uint2*)(h_out + (size_t)row * D + 4 * i) = o;
        }
    }
}

int launch_swap_add_norm(const float* in, const float* rnorm, const float* g_prev, float gain_sqrt_dim, const float* add, long long add_bs,
                         const float* g_norm, float norm_eps, float* x_out, void* h_out, int B, int A1, int A2, int E4, hipStream_t s) {
    const int nv = (A1 * E4 + 63) / 64, grid = (B * A2 + 3) / 4;
#define MEDP_SAN(NV) swap_add_norm_kernel<NV><<<grid, 256, 0, s>>>(in, rnorm, g_prev, gain_sqrt_dim, add, add_bs, g_norm, norm_eps, x_out, (bf16_t*)h_out, B, A1, A2, E4)
    if (nv <= 5) MEDP_SAN(5);
    else if (nv <= 10) MEDP_SAN(10);
    else if (nv <= 16) MEDP_SAN(16);
    else if (nv <= 25) MEDP_SAN(25);
    else return 1;          // wider rows: the caller takes the two-launch path
#undef MEDP_SAN
    MEDP_LAUNCH_CHECK("duett swap+add+norm");
    return 0;
}

template <int NB>
int launch_psi_embed_event_nb(const MedpDuettWeights* w, const float* xs_static, const float* xs_ts, float* xe, void* h, float* psi0_out, int B,
                              int T, hipStream_t s) {
    const int V = w->n_vars, V1 = V + 1, T1 = T + 1, E = w->d_embedding;
    const size_t lds = (size_t)NB * T1 * E * sizeof(float);
    if (lds > 120 * 1024) return -2;
    MEDP_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)psi_embed_event_kernel<24, 64, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    });
    const TabWeights tw{xs_static, (const float*)w->tab_w0, (const float*)w->tab_b0, (const float*)w->tab_bn_scale, (const float*)w->tab_bn_shift,
                        (const float*)w->tab_w4, (const float*)w->tab_b4, w->n_static, w->d_hidden_tab};
    psi_embed_event_kernel<24, 64, NB><<<dim3((B + NB - 1) / NB, V1), 128, lds, s>>>(
        xs_ts, (const float*)w->emb_l0, (const float*)w->emb_w4t, (const float*)w->emb_b4, (const float*)w->n_obs_table, w->n_obs_rows, tw,
        (const float*)w->special, (const float*)w->event_embedding, (const float*)w->event_enc[0].g_attn, w->norm_eps, xe, (bf16_t*)h, psi0_out,
        B, T, V, NB == 1 ? embed_mfma_on() : 0);
    MEDP_LAUNCH_CHECK("duett psi_embed_event");
    return 0;
}

int launch_psi_embed_event(const MedpDuettWeights* w, const float* xs_static, const float* xs_ts, float* xe, void* h, float* psi0_out, int B,
                           int T, hipStream_t s) {
    MEDP_CHECK_ARG(w->emb_l0 && w->emb_w4t, "duett: emb_l0 / emb_w4t (transposed weight layout) missing");
    MEDP_CHECK_ARG(w->d_hidden_tab <= 256 && w->n_obs_rows >= 1 && w->n_obs_rows <= 64, "duett: tab encoder hidden size above 256 / n_obs table above 64 rows");
    // rows (= cells per lane) per workgroup.  Measured at cfg3 (tools/time_embed_variants.py): 1 -> 44 us, 2 -> 53 us, 4 -> 77 us:
    // fewer, longer waves lose more than the amortised weight reads win; 1 is the default.
    static const int nb = [] { const char* e = getenv("MEDP_PSI_NB"); return e ? atoi(e) : 1; }();
    int rc = nb >= 4 ? launch_psi_embed_event_nb<4>(w, xs_static, xs_ts, xe, h, psi0_out, B, T, s)
           : nb >= 2 ? launch_psi_embed_event_nb<2>(w, xs_static, xs_ts, xe, h, psi0_out, B, T, s) : -2;
    if (rc == -2) rc = launch_psi_embed_event_nb<1>(w, xs_static, xs_ts, xe, h, psi0_out, B, T, s);   // long rows: one per workgroup
    MEDP_CHECK_ARG(rc != -2, "duett: a (T+1)*E row does not fit the embed kernel's LDS tile");
    return rc;
}

struct DuettWs {
    size_t tab, psi, xe, xt, temb, h, qkv, o, f, rn, split, split_bytes, total;
};
DuettWs plan(const MedpDuettWeights* w, int B, int T) {
    const size_t V1 = w->n_vars + 1, T1 = T + 1, E = w->d_embedding;
    const size_t cells = (size_t)B * T1 * V1 * E;
    const size_t rows = (size_t)B * max(T1, V1);
    DuettWs s{};
    size_t off = 0;
    s.tab = off;  off += al((size_t)B * E * 4);
    s.psi = off;  off += al(cells * 4);
    s.xe = off;   off += al(cells * 4);
    s.xt = off;   off += al(cells * 4);
    s.temb = off; off += al(cells * 4);
    s.h = off;    off += al(cells * 2);
    s.qkv = off;  off += al(rows * 3 * E * 4);
    s.o = off;    off += al(rows * E * 2);
    s.f = off;    off += al(rows * (size_t)w->d_ff * 2);
    s.rn = off;   off += al(rows * 4);
    // split-K partial sums of the skinny GEMMs (medp_gemm_bf16_nt_ws): the largest need over both axes' four shapes
    size_t sp = 0;
    for (int axis = 0; axis < 2; ++axis) {
        const int M = (int)(B * (axis ? T1 : V1)), D = (int)(E * (axis ? V1 : T1));
        sp = max(sp, medp_gemm_nt_workspace_bytes(M, 3 * (int)E, D));
        sp = max(sp, medp_gemm_nt_workspace_bytes(M, D, (int)E));
        sp = max(sp, medp_gemm_nt_workspace_bytes(M, w->d_ff, D));
        sp = max(sp, medp_gemm_nt_workspace_bytes(M, D, w->d_ff));
    }
    s.split = off; s.split_bytes = sp; off += al(sp);
    s.total = off;
    return s;
}

// one x_transformers encoder block on x [M = B*N tokens, D] (in place); leaves the FINAL ScaleNorm to the caller
int encoder_forward(const MedpEncoderWeights& e, const MedpDuettWeights* w, float* x, int B, int N, int D, char* base,
                    const DuettWs& ws, void* stream, bool h_ready = false) {
    const int M = B * N, E = w->d_embedding, H = w->n_heads, dh = E / H;
    void* h = base + ws.h;
    float* qkv = (float*)(base + ws.qkv);
    void* o = base + ws.o;
    void* f = base + ws.f;
    float* sp = (float*)(base + ws.split);
    if (!h_ready)      // else: the producer of x (embed / swap kernel) has written ScaleNorm(x) to ws.h already
        MEDP_TRY(medp_scalenorm_fwd(x, D, e.g_attn, h, D, 1, nullptr, M, D, w->norm_eps, stream));
    MEDP_TRY(medp_gemm_bf16_nt_ws(h, e.qkv_w, qkv, M, 3 * E, D, D, D, 3 * E, nullptr, nullptr, nullptr, 0, 0, 0, sp, ws.split_bytes, stream));
    // both axes' attention on the matrix cores (attention_dh16.hip); shapes it is not built for take the fp32 VALU kernel
    static const bool mfma_attn = [] { const char* e2 = getenv("MEDP_DUETT_MFMA_ATTN"); return !e2 || atoi(e2) != 0; }();
    int arc = mfma_attn ? medp_attn_dh16_fwd(qkv, 3 * E, o, E, B, N, H, dh, 1.0f / sqrtf((float)dh), stream) : -2;
    if (arc == -2)
        arc = medp_attn_small_fwd(qkv, 3 * E, (long long)N * 3 * E, qkv + E, qkv + 2 * E, 3 * E, (long long)N * 3 * E, o, E, 1, nullptr, B, N,
                                  N, H, dh, 1.0f / sqrtf((float)dh), 0.f, 0u, 0u, stream);
    MEDP_TRY(arc);
    MEDP_TRY(medp_gemm_bf16_nt_ws(o, e.out_w, x, M, D, E, E, E, D, nullptr, nullptr, x, D, 0, 0, sp, ws.split_bytes, stream));
    MEDP_TRY(medp_scalenorm_fwd(x, D, e.g_ff, h, D, 1, nullptr, M, D, w->norm_eps, stream));
    MEDP_TRY(medp_gemm_bf16_nt_ws(h, e.ff1_w, f, M, w->d_ff, D, D, D, w->d_ff, e.ff1_b, nullptr, nullptr, 0, 1, 1, sp, ws.split_bytes, stream));
    MEDP_TRY(medp_gemm_bf16_nt_ws(f, e.ff2_w, x, M, D, w->d_ff, w->d_ff, w->d_ff, D, e.ff2_b, nullptr, x, D, 0, 0, sp, ws.split_bytes, stream));
    return 0;
}

}  // namespace

extern "C" size_t medp_duett_workspace_bytes(const MedpDuettWeights* w, int B, int T) {
    if (!w || B <= 0 || T <= 0) return 0;
    return plan(w, B, T).total;
}

extern "C" int medp_duett_encode(const MedpDuettWeights* w, const float* xs_static, const float* xs_ts, const float* xs_times,
                                 int B, int T, float* tokens_f32, void* tokens_bf16, float* psi0_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    MEDP_CHECK_ARG(w && xs_static && xs_ts && xs_times && workspace, "duett_encode: null argument");
    MEDP_CHECK_ARG(tokens_f32, "duett_encode: tokens_f32 output is required");
    MEDP_CHECK_ARG(B > 0 && T > 0, "duett_encode: bad shape B=%d T=%d", B, T);
    MEDP_CHECK_ARG(w->d_embedding == 24 && w->d_hidden_embed == 64, "duett_encode: built for d_embedding 24 / hidden 64 (duett.py:50)");
    MEDP_CHECK_ARG(w->d_embedding % w->n_heads == 0 && (3 * w->d_embedding) % 4 == 0, "duett_encode: bad head split");
    const DuettWs ws = plan(w, B, T);
    MEDP_CHECK_ARG(workspace_bytes >= ws.total, "duett_encode: workspace %zu < required %zu", workspace_bytes, ws.total);
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    const int V = w->n_vars, V1 = V + 1, T1 = T + 1, E = w->d_embedding;
    const int et = E * T1, tt = E * V1;
    float* tab = (float*)(base + ws.tab);
    float* psi = (float*)(base + ws.psi);
    float* xe = (float*)(base + ws.xe);
    float* xt = (float*)(base + ws.xt);
    float* temb = (float*)(base + ws.temb);
    float* rn = (float*)(base + ws.rn);

    const int E4 = E / 4;
    MEDP_CHECK_ARG((size_t)B * T1 * V1 * E4 < (1ull << 31), "duett_encode: B*(T+1)*(V+1)*E/4 must stay below 2^31");
    // Layer 0, event view, in ONE launch: psi build + axis swap + event embedding + the event encoder's first ScaleNorm.
    // MEDP_DUETT_FUSED=0 keeps the separate launches (psi_embed -> swap -> scalenorm ...) for A/B runs; same values.
    static const bool fused = [] { const char* e = getenv("MEDP_DUETT_FUSED"); return !e || atoi(e) != 0; }();
    const int swap_grid = grid_for((size_t)B * T1 * V1 * E4);
    if (fused) {
        MEDP_TRY(launch_psi_embed_event(w, xs_static, xs_ts, xe, base + ws.h, psi0_out, B, T, s));
    } else {
        tab_encoder_kernel<<<B, 128, w->d_hidden_tab * sizeof(float), s>>>(xs_static, (const float*)w->tab_w0, (const float*)w->tab_b0,
                                                                           (const float*)w->tab_bn_scale, (const float*)w->tab_bn_shift,
                                                                           (const float*)w->tab_w4, (const float*)w->tab_b4, tab,
                                                                           w->n_static, w->d_hidden_tab, E);
        MEDP_LAUNCH_CHECK("duett tab_encoder");
        psi_embed_kernel<24, 64><<<dim3((B * T1 + 255) / 256, V1), 256, 0, s>>>(
            xs_ts, (const float*)w->emb_w0, (const float*)w->emb_b0, (const float*)w->emb_bn_scale, (const float*)w->emb_bn_shift,
            (const float*)w->emb_w4, (const float*)w->emb_b4, (const float*)w->n_obs_table, w->n_obs_rows, tab, (const float*)w->special, psi,
            B, T, V);
        MEDP_LAUNCH_CHECK("duett psi_embed");
        if (psi0_out) {
            hipError_t e = hipMemcpyAsync(psi0_out, psi, (size_t)B * T1 * V1 * E * 4, hipMemcpyDeviceToDevice, s);
            MEDP_CHECK_ARG(e == hipSuccess, "duett_encode: psi0 copy failed");
        }
    }
    MEDP_TRY(launch_time_embed(w, xs_times, temb, B, T, s));

    const float* cur = psi;          // time view [B, T1, V1, E]; rows (b,t) of tt features
    const float* cur_rn = nullptr;   // pending final-ScaleNorm row scales of `cur`
    const float* cur_g = nullptr;
    for (int l = 0; l < w->n_layers; ++l) {
        // time view -> event view (+ event embedding), applying the previous time encoder's final norm      (model :80)
        bool h_ready = fused && l == 0;          // layer 0: the embed kernel wrote xe and ScaleNorm(xe) already
        if (!h_ready) {
            if (fused && launch_swap_add_norm(cur, cur_rn, cur_g, sqrtf((float)tt), (const float*)w->event_embedding, 0,
                                              (const float*)w->event_enc[l].g_attn, w->norm_eps, xe, base + ws.h, B, T1, V1, E4, s) == 0) {
                h_ready = true;
            } else {
                axis_swap_add_kernel<<<swap_grid, 256, 0, s>>>(cur, cur_rn, cur_g, sqrtf((float)tt), (const float*)w->event_embedding, 0, xe, B,
                                                               T1, V1, E4);
                MEDP_LAUNCH_CHECK("duett swap t->e");
            }
        }
        MEDP_TRY(encoder_forward(w->event_enc[l], w, xe, B, V1, et, base, ws, stream, h_ready));              // (model :81)
        const float* e_rn = nullptr;
        if (w->final_norm) {
            // final ScaleNorm of the event encoder: statistics now, scaling fused into the swap below
            MEDP_TRY(medp_scalenorm_fwd(xe, et, w->event_enc[l].g_final, base + ws.h, et, 1, rn, B * V1, et, w->norm_eps, stream));
            e_rn = rn;
        }
        // event view -> time view (+ time embedding) (+ the time encoder's first ScaleNorm)                   (model :81,:90)
        h_ready = fused && launch_swap_add_norm(xe, e_rn, (const float*)w->event_enc[l].g_final, sqrtf((float)et), temb,
                                                (long long)T1 * V1 * E, (const float*)w->time_enc[l].g_attn, w->norm_eps, xt, base + ws.h, B,
                                                V1, T1, E4, s) == 0;
        if (!h_ready) {
            axis_swap_add_kernel<<<swap_grid, 256, 0, s>>>(xe, e_rn, (const float*)w->event_enc[l].g_final, sqrtf((float)et), temb,
                                                           (long long)T1 * V1 * E, xt, B, V1, T1, E4);
            MEDP_LAUNCH_CHECK("duett swap e->t");
        }
        MEDP_TRY(encoder_forward(w->time_enc[l], w, xt, B, T1, tt, base, ws, stream, h_ready));               // (model :91)
        if (l + 1 < w->n_layers) {
            if (w->final_norm) {
                MEDP_TRY(medp_scalenorm_fwd(xt, tt, w->time_enc[l].g_final, base + ws.h, tt, 1, rn, B * T1, tt, w->norm_eps, stream));
                cur_rn = rn;
                cur_g = (const float*)w->time_enc[l].g_final;
            }
            // ping-pong: the next swap reads xt and writes xe
            cur = xt;
        } else {
            if (w->final_norm) {
                MEDP_TRY(medp_scalenorm_fwd(xt, tt, w->time_enc[l].g_final, tokens_f32, tt, 0, nullptr, B * T1, tt, w->norm_eps, stream));
            } else {
                hipError_t e = hipMemcpyAsync(tokens_f32, xt, (size_t)B * T1 * tt * 4, hipMemcpyDeviceToDevice, s);
                MEDP_CHECK_ARG(e == hipSuccess, "duett_encode: output copy failed");
            }
        }
    }
    if (tokens_bf16) MEDP_TRY(medp_cast_f32_bf16(tokens_f32, tt, tokens_bf16, tt, B * T1, tt, stream));
    return 0;
}

// ---- the embedding stage and the fused swap on their own (kernel-level parity checks and bench.py's per-kernel HBM table) ----------
// stages: bit 0 = static encoder + fused psi build (writes xe_out fp32 [B,V+1,(T+1)E], h_out bf16 same shape, psi0_out optional);
//         bit 1 = time embedding (writes temb_out fp32 [B,T+1,(V+1)E]).
extern "C" int medp_duett_embed_fwd(const MedpDuettWeights* w, const float* xs_static, const float* xs_ts, const float* xs_times, int B, int T,
                                    float* xe_out, void* h_out, float* temb_out, float* psi0_out, float* tab_workspace, int stages,
                                    void* stream) {
    MEDP_CHECK_ARG(w && xs_static && xs_ts && xs_times && B > 0 && T > 0, "duett_embed_fwd: bad argument");
    MEDP_CHECK_ARG(w->d_embedding == 24 && w->d_hidden_embed == 64, "duett_embed_fwd: built for d_embedding 24 / hidden 64 (duett.py:50)");
    hipStream_t s = (hipStream_t)stream;
    const int V = w->n_vars, V1 = V + 1, T1 = T + 1, E = w->d_embedding, tt = E * V1;
    if (stages & 1) {
        MEDP_CHECK_ARG(xe_out && h_out, "duett_embed_fwd: stage 1 needs xe_out and h_out");
        (void)tab_workspace;          // kept in the signature; the static encoder runs inside the fused kernel
        MEDP_TRY(launch_psi_embed_event(w, xs_static, xs_ts, xe_out, h_out, psi0_out, B, T, s));
    }
    if (stages & 2) {
        MEDP_CHECK_ARG(temb_out, "duett_embed_fwd: stage 2 needs temb_out");
        MEDP_TRY(launch_time_embed(w, xs_times, temb_out, B, T, s));
    }
    return 0;
}

// x_out[b][a2][a1][:] = in[b][a1][a2][:] * (rnorm ? rnorm[b][a1] * sqrt(A2*E) * g_prev : 1) + add ;  h_out = ScaleNorm(x_out) (bf16).
// add: [A2, A1, E] table (add_batch_stride 0) or [B, A2, A1, E].
extern "C" int medp_duett_swap_add_norm(const float* in, const float* rnorm, const float* g_prev, const float* add, long long add_batch_stride,
                                        const float* g_norm, float norm_eps, float* x_out, void* h_out, int B, int A1, int A2, int E,
                                        void* stream) {
    MEDP_CHECK_ARG(in && add && g_norm && x_out && h_out && B > 0 && A1 > 0 && A2 > 0 && E > 0 && E % 4 == 0, "duett_swap_add_norm: bad argument");
    MEDP_CHECK_ARG(!rnorm || g_prev, "duett_swap_add_norm: rnorm needs g_prev");
    MEDP_CHECK_ARG((size_t)B * A1 * A2 * (E / 4) < (1ull << 31), "duett_swap_add_norm: too large");
    const int rc = launch_swap_add_norm(in, rnorm, g_prev, sqrtf((float)A2 * E), add, add_batch_stride, g_norm, norm_eps, x_out, h_out, B, A1, A2,
                                        E / 4, (hipStream_t)stream);
    MEDP_CHECK_ARG(rc != 1, "duett_swap_add_norm: rows of more than 6400 floats are not built");
    return rc;
}

// Debug hook (NOT part of the C ABI in include/medp_hip.h; tests/test_gpu_duett_fused.py): 1 / 0 = the fp32-MFMA / VALU forms of the
// time-embedding and psi-embedding kernels, -1 = back to the environment's choice.  Returns the previous setting.
extern "C" int medp_dbg_embed_mfma(int on) {
    const int prev = g_embed_mfma;
    g_embed_mfma = on < 0 ? -1 : (on != 0);
    return prev;
}
