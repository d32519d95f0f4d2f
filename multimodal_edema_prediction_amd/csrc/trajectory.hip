// LocalTrajectoryEncoder (reference models/main_architecture_duett.py:1242-1391): the parts that are not a
// Linear / LayerNorm — the per-variable local features with their sequential "time since last observation" scan, and the GRU
// recurrence over the B*V independent sequences, forward and backward (BPTT).
//
// GRU (hidden 128, the module's default d_model): the input products x_t W_ih^T are ONE bf16 MFMA GEMM over all (sequence,
// step) rows outside these kernels; what is sequential is h_{t-1} W_hh^T.  A workgroup of 4 waves owns 16 sequences for all T
// steps; wave w owns hidden units [32 w, 32 w + 32) of all three gates, so that r, z and n of one (sequence, unit) meet in one
// lane.  Its 24 W_hh fragments (3 gates x 2 column blocks x 4 k-steps of v_mfma_f32_16x16x32_bf16) stay in REGISTERS for the
// whole kernel (96 VGPRs); per step only the new hidden state crosses the waves, as a 16 x 128 bf16 tile in LDS (double
// buffered: one barrier per step).  The hidden state itself is carried in fp32 registers; bf16 only feeds the MFMA.
// Backward walks the steps in reverse with the same ownership: the gate gradients of a step go through LDS as a 16 x 384 bf16
// tile and dh_{t-1} += dgh W_hh uses the TRANSPOSED weight fragments (again register resident).  dW_hh and db_hh are a
// transposed GEMM / column sums over the stored gate gradients afterwards (host side of the autograd node).
#include "common.h"

namespace {

constexpr int D = 128;          // hidden size the register plan is written for
constexpr int SB = 16;          // sequences per workgroup
constexpr int HROW = 2 * D + 16;          // bytes per row of the bf16 hidden tile (padded: conflict-free 16-B fragment reads)
constexpr int GROW = 2 * 3 * D + 16;      // bytes per row of the bf16 gate-gradient tile

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * sigmoidf_(2.0f * x) - 1.0f; }

// one thread per (sample, variable): walks the T steps, keeps the elapsed-steps counter (reference :1316-1330) and writes the
// five features of :1351-1356 padded to 8 floats (the Linear(5, d) then runs as a K = 8 GEMM)
__global__ void traj_features_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V, v = i % V;
    const float inv_t = 1.0f / (float)T, inv_log16 = 0.36067376022224085f;     // 1 / ln 16
    float elapsed = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float* row = x + ((size_t)b * T + t) * 2 * V;
        const float cnt = fmaxf(row[V + v], 0.0f);
        const bool obs = cnt > 0.0f;
        elapsed += 1.0f;
        f32x4 lo = (f32x4){obs ? row[v] : 0.0f, obs ? 1.0f : 0.0f, log1pf(cnt) * inv_log16, elapsed * inv_t};
        f32x4 hi = (f32x4){(float)(T - t) * inv_t, 0.0f, 0.0f, 0.0f};
        float* o = out + ((size_t)i * T + t) * 8;
        *(f32x4*)o = lo;
        *(f32x4*)(o + 4) = hi;
        if (obs) elapsed = 0.0f;
    }
}

struct GruFwd {
    const float* gi;            // [S, T, 3D]  x_t W_ih^T + b_ih
    const bf16_t* whh;          // [3D, D] bf16
    const float* bhh;           // [3D]
    float* hseq;                // [S, T, D]
    float* gates;               // [S, T, 3D]  r | z | n   (saved for backward; may be null)
    float* hn;                  // [S, T, D]   W_hn h + b_hn (saved for backward; may be null)
    int S, T;
};

__global__ __launch_bounds__(256) void gru_fwd_kernel(const GruFwd p) {
    __shared__ __attribute__((aligned(16))) char hbuf[2][SB * HROW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, kq = lane >> 4;
    const int seq = blockIdx.x * SB + fr;
    const bool ok = seq < p.S;
    // W_hh fragments: a-operand rows = gate columns g*128 + 32w + jb*16 + fr, k = hidden index
    bf16x8 wf[3][2][4];
    f32x4 bh[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int col = g * D + 32 * w + jb * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[g][jb][ks] = *(const bf16x8*)(p.whh + (size_t)(col + fr) * D + ks * 32 + kq * 8);
            bh[g][jb] = *(const f32x4*)(p.bhh + col + kq * 4);
        }
    f32x4 h[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    // h_0 = 0: this lane's slice of tile 0
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) *(uint2*)(hbuf[0] + fr * HROW + (32 * w + jb * 16 + kq * 4) * 2) = make_uint2(0u, 0u);
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
        const int cur = t & 1;
        f32x4 gi[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
                gi[g][jb] = ok ? *(const f32x4*)(p.gi + ((size_t)seq * p.T + t) * 3 * D + g * D + 32 * w + jb * 16 + kq * 4)
                               : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 acc[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) acc[g][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 hb = *(const bf16x8*)(hbuf[cur] + fr * HROW + (ks * 32 + kq * 8) * 2);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) acc[g][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][jb][ks], hb, acc[g][jb], 0, 0, 0);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x4 r, z, n, ghn, hnew;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r[c] = sigmoidf_(gi[0][jb][c] + acc[0][jb][c] + bh[0][jb][c]);
                z[c] = sigmoidf_(gi[1][jb][c] + acc[1][jb][c] + bh[1][jb][c]);
                ghn[c] = acc[2][jb][c] + bh[2][jb][c];
                n[c] = tanhf_(gi[2][jb][c] + r[c] * ghn[c]);
                hnew[c] = (1.0f - z[c]) * n[c] + z[c] * h[jb][c];
            }
            h[jb] = hnew;
            const int u = 32 * w + jb * 16 + kq * 4;
            *(uint2*)(hbuf[cur ^ 1] + fr * HROW + u * 2) = make_uint2(pack_bf2(hnew[0], hnew[1]), pack_bf2(hnew[2], hnew[3]));
            if (ok) {
                const size_t row = (size_t)seq * p.T + t;
                *(f32x4*)(p.hseq + row * D + u) = hnew;
                if (p.gates) {
                    *(f32x4*)(p.gates + row * 3 * D + u) = r;
                    *(f32x4*)(p.gates + row * 3 * D + D + u) = z;
                    *(f32x4*)(p.gates + row * 3 * D + 2 * D + u) = n;
                    *(f32x4*)(p.hn + row * D + u) = ghn;
                }
            }
        }
        __syncthreads();
    }
}

struct GruBwd {
    const float* dh;            // [S, T, D]   gradient w.r.t. every hidden state that left the GRU
    const float* gates;         // [S, T, 3D]
    const float* hn;            // [S, T, D]
    const float* hseq;          // [S, T, D]
    const bf16_t* whh_t;        // [D, 3D] bf16: W_hh transposed
    float* dgi;                 // [S, T, 3D]  gradient w.r.t. x_t W_ih^T + b_ih   (= dgh for r, z)
    float* dghn;                // [S, T, D]   gradient w.r.t. W_hn h + b_hn
    bf16_t* dgh16;              // [S, T, 3D]  bf16 copy of dgh (operand of the dW_hh GEMM)
    int S, T;
};

__global__ __launch_bounds__(256) void gru_bwd_kernel(const GruBwd p) {
    __shared__ __attribute__((aligned(16))) char gbuf[2][SB * GROW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, kq = lane >> 4;
    const int seq = blockIdx.x * SB + fr;
    const bool ok = seq < p.S;
    // W_hh^T fragments: a-operand rows = hidden units 32w + ub*16 + fr, k = gate column (0..383)
    bf16x8 wt[2][12];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) wt[ub][ks] = *(const bf16x8*)(p.whh_t + (size_t)(32 * w + ub * 16 + fr) * 3 * D + ks * 32 + kq * 8);
    f32x4 carry[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    for (int t = p.T - 1; t >= 0; --t) {
        const int cur = t & 1;
        f32x4 direct[2];
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            const int u = 32 * w + ub * 16 + kq * 4;
            const size_t row = (size_t)(ok ? seq : 0) * p.T + t;
            const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 dht = (ok ? *(const f32x4*)(p.dh + row * D + u) : zero4) + carry[ub];
            const f32x4 r = ok ? *(const f32x4*)(p.gates + row * 3 * D + u) : zero4;
            const f32x4 z = ok ? *(const f32x4*)(p.gates + row * 3 * D + D + u) : zero4;
            const f32x4 n = ok ? *(const f32x4*)(p.gates + row * 3 * D + 2 * D + u) : zero4;
            const f32x4 hn = ok ? *(const f32x4*)(p.hn + row * D + u) : zero4;
            const f32x4 hp = (ok && t > 0) ? *(const f32x4*)(p.hseq + (row - 1) * D + u) : zero4;
            f32x4 drp, dzp, dnp, dgn;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float dn = dht[c] * (1.0f - z[c]);
                const float dz = dht[c] * (hp[c] - n[c]);
                dnp[c] = dn * (1.0f - n[c] * n[c]);
                dzp[c] = dz * z[c] * (1.0f - z[c]);
                drp[c] = dnp[c] * hn[c] * r[c] * (1.0f - r[c]);
                dgn[c] = dnp[c] * r[c];
                direct[ub][c] = dht[c] * z[c];
            }
            char* g = gbuf[cur] + fr * GROW;
            const uint2 pr = make_uint2(pack_bf2(drp[0], drp[1]), pack_bf2(drp[2], drp[3]));
            const uint2 pz = make_uint2(pack_bf2(dzp[0], dzp[1]), pack_bf2(dzp[2], dzp[3]));
            const uint2 pn = make_uint2(pack_bf2(dgn[0], dgn[1]), pack_bf2(dgn[2], dgn[3]));
            *(uint2*)(g + u * 2) = pr;
            *(uint2*)(g + (D + u) * 2) = pz;
            *(uint2*)(g + (2 * D + u) * 2) = pn;
            if (ok) {
                *(f32x4*)(p.dgi + row * 3 * D + u) = drp;
                *(f32x4*)(p.dgi + row * 3 * D + D + u) = dzp;
                *(f32x4*)(p.dgi + row * 3 * D + 2 * D + u) = dnp;
                *(f32x4*)(p.dghn + row * D + u) = dgn;
                *(uint2*)(p.dgh16 + row * 3 * D + u) = pr;
                *(uint2*)(p.dgh16 + row * 3 * D + D + u) = pz;
                *(uint2*)(p.dgh16 + row * 3 * D + 2 * D + u) = pn;
            }
        }
        __syncthreads();
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            const bf16x8 gb = *(const bf16x8*)(gbuf[cur] + fr * GROW + (ks * 32 + kq * 8) * 2);
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) acc[ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ub][ks], gb, acc[ub], 0, 0, 0);
        }
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) carry[ub] = direct[ub] + acc[ub];
    }
}

}  // namespace

extern "C" int medp_traj_features(const float* x, float* out, int B, int T, int V, void* stream) {
    MEDP_CHECK_ARG(x && out, "traj_features: null argument");
    MEDP_CHECK_ARG(B > 0 && T > 0 && V > 0, "traj_features: bad shape B=%d T=%d V=%d", B, T, V);
    const int n = B * V;
    traj_features_kernel<<<(n + 127) / 128, 128, 0, (hipStream_t)stream>>>(x, out, B, T, V);
    MEDP_LAUNCH_CHECK("medp_traj_features");
    return 0;
}

extern "C" int medp_gru_fwd(const float* gi, const void* whh_bf16, const float* bhh, float* hseq, float* gates, float* hn, int S,
                            int T, int d, void* stream) {
    MEDP_CHECK_ARG(gi && whh_bf16 && bhh && hseq, "gru_fwd: null argument");
    MEDP_CHECK_ARG((gates == nullptr) == (hn == nullptr), "gru_fwd: gates and hn are saved together or not at all");
    MEDP_CHECK_ARG(d == D, "gru_fwd: hidden size %d is not built (the register plan is written for %d)", d, D);
    MEDP_CHECK_ARG(S > 0 && T > 0, "gru_fwd: bad shape S=%d T=%d", S, T);
    const GruFwd p{gi, (const bf16_t*)whh_bf16, bhh, hseq, gates, hn, S, T};
    gru_fwd_kernel<<<(S + SB - 1) / SB, 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gru_fwd");
    return 0;
}

extern "C" int medp_gru_bwd(const float* dh, const float* gates, const float* hn, const float* hseq, const void* whh_t_bf16,
                            float* dgi, float* dghn, void* dgh_bf16, int S, int T, int d, void* stream) {
    MEDP_CHECK_ARG(dh && gates && hn && hseq && whh_t_bf16 && dgi && dghn && dgh_bf16, "gru_bwd: null argument");
    MEDP_CHECK_ARG(d == D, "gru_bwd: hidden size %d is not built (the register plan is written for %d)", d, D);
    MEDP_CHECK_ARG(S > 0 && T > 0, "gru_bwd: bad shape S=%d T=%d", S, T);
    const GruBwd p{dh, gates, hn, hseq, (const bf16_t*)whh_t_bf16, dgi, dghn, (bf16_t*)dgh_bf16, S, T};
    gru_bwd_kernel<<<(S + SB - 1) / SB, 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_gru_bwd");
    return 0;
}
