// Flash-attention BACKWARD for head dim 64 on gfx950 (the trainable CXR encoder, --unfreeze_cxr).
//
// Given Q, K, V (bf16), dO (bf16), the forward's log2-domain logsumexp L[q] and D[q] = <dO[q], O[q]>:
//     P = exp2(S c - L),  S = Q K^T           dP = dO V^T          dS = P o (dP - D) * scale
//     dV = P^T dO         dK = dS^T Q         dQ = dS K
// Two launches of ONE templated kernel, each shaped like the forward (attention_dh64.hip): a 256-thread workgroup of one
// (batch, head) stages two [row][64] bf16 images in LDS by LDS-DMA and every wave owns 16-row subtiles of the OTHER
// sequence dimension, whose fragments stay in registers:
//   DKV = false (dQ):      images K, V (keys);   own = queries (Q, dO fragments);  S^T = K Q^T, dP^T = V dO^T  ->
//                          dS^T (keys on accumulator rows, query on the lane) is directly the B operand of dQ^T = K^T dS^T
//   DKV = true  (dK, dV):  images Q, dO (queries); own = keys (K, V fragments);    S = Q K^T,  dP = dO V^T     ->
//                          P and dS (queries on rows, key on the lane) are the B operands of dV^T = dO^T P, dK^T = Q^T dS
// The transposed A operands (K^T, dO^T, Q^T) come from the row-major images through ds_read_b64_tr_b16, exactly as V^T does
// in the forward, so nothing is ever transposed in memory and P / dS never leave registers.
#include "common.h"
#include "medp_hip.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_attnb[4] = {0, 0, 0, 0};

constexpr int KC = 320;   // most rows per LDS chunk
constexpr int NW = 2;     // most 16-row subtiles one wave owns

struct BwdParams {
    const bf16_t *q, *k, *v, *dob;
    const float *lse, *dsum;      // [B, H, S]
    float *dq, *dk, *dv;          // fp32, row stride ldd, head h at column h*64
    int B, S, H, ldqkv, lddo, ldd;
    float scale, scale_log2e;
    int crows;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ bf16x4 lds_tr16(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}
__device__ __forceinline__ bf16x8 pack8(const f32x4 a, const f32x4 b) {
    union { bf16x8 v; uint32_t u[4]; } pk;
    pk.u[0] = pack_bf2(a[0], a[1]);
    pk.u[1] = pack_bf2(a[2], a[3]);
    pk.u[2] = pack_bf2(b[0], b[1]);
    pk.u[3] = pack_bf2(b[2], b[3]);
    return pk.v;
}

template <bool DKV, int NWW>
__device__ __forceinline__ void wave_body(const BwdParams& p, char* smem, int own0, int b, int h, int tid, int wave) {
    const int lane = tid & 63, fr = lane & 15, kq = lane >> 4;
    const int niter = p.crows >> 5;
    char* sX = smem;                              // dQ: K   | dK/dV: Q
    char* sY = smem + p.crows * 128;              // dQ: V   | dK/dV: dO
    float* sL = (float*)(smem + 2 * p.crows * 128);
    float* sD = sL + p.crows;
    const bf16_t* zero = (const bf16_t*)g_zero16_attnb;
    const size_t row0 = (size_t)b * p.S;
    const size_t st0 = ((size_t)b * p.H + h) * p.S;
    const bf16_t* gX = DKV ? p.q : p.k;           // image sources
    const bf16_t* gY = DKV ? p.dob : p.v;
    const int ldX = p.ldqkv, ldY = DKV ? p.lddo : p.ldqkv;
    const bf16_t* gx = DKV ? p.k : p.q;           // own-fragment sources
    const bf16_t* gy = DKV ? p.v : p.dob;
    const int ldx = p.ldqkv, ldy = DKV ? p.ldqkv : p.lddo;

    constexpr int NA = NWW > 0 ? NWW : 1;
    bf16x8 xf[NA][2], yf[NA][2];
    f32x4 acc1[NA][4], acc2[NA][4];
    float lse_own[NA], d_own[NA];
    if constexpr (NWW > 0) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) {
            const int oi = own0 + w * 16 + fr;
            const bool ok = oi < p.S;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                xf[w][s] = *(const bf16x8*)(ok ? gx + (row0 + oi) * ldx + h * 64 + s * 32 + kq * 8 : zero);
                yf[w][s] = *(const bf16x8*)(ok ? gy + (row0 + oi) * ldy + h * 64 + s * 32 + kq * 8 : zero);
            }
            lse_own[w] = (!DKV && ok) ? p.lse[st0 + oi] : 0.f;
            d_own[w] = (!DKV && ok) ? p.dsum[st0 + oi] : 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                acc1[w][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc2[w][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    const int tr_q = fr >> 2, tr_p = fr & 3;
    const float c = p.scale_log2e;

    for (int c0 = 0; c0 < p.S; c0 += KC) {
        const int nrows = min(KC, p.S - c0);
        if (c0 > 0) __syncthreads();
        for (int i = 0; i < niter; ++i) {
            const int qd = i * 256 + tid;
            const int row = qd >> 3, ch = (qd & 7) ^ (row & 7);
            const bool ok = row < nrows;
            const size_t grow = row0 + c0 + row;
            glds16(ok ? gX + grow * ldX + h * 64 + ch * 8 : zero, sX + (i * 256 + wave * 64) * 16);
            glds16(ok ? gY + grow * ldY + h * 64 + ch * 8 : zero, sY + (i * 256 + wave * 64) * 16);
        }
        if (DKV) {
            for (int i = tid; i < p.crows; i += 256) {
                const bool ok = i < nrows;
                sL[i] = ok ? p.lse[st0 + c0 + i] : 0.f;
                sD[i] = ok ? p.dsum[st0 + c0 + i] : 0.f;
            }
        }
        MEDP_WAIT_LDS_DMA();
        __syncthreads();               // LDS-DMA landed (explicit wait above) before any wave reads the chunk
        if constexpr (NWW > 0) {
            const int nblk = (nrows + 63) >> 6;
            for (int kb = 0; kb < nblk; ++kb) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (kb * 64 + ks * 32 >= nrows) continue;             // wave-uniform: nothing real in this pair of row tiles
                    f32x4 sc[NWW][2], dp[NWW][2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int r = kb * 64 + (2 * ks + t) * 16 + fr;
                        const char* bx = sX + r * 128;
                        const char* by = sY + r * 128;
                        const bf16x8 x0 = *(const bf16x8*)(bx + (((0 + kq) ^ (r & 7)) << 4));
                        const bf16x8 x1 = *(const bf16x8*)(bx + (((4 + kq) ^ (r & 7)) << 4));
                        const bf16x8 y0 = *(const bf16x8*)(by + (((0 + kq) ^ (r & 7)) << 4));
                        const bf16x8 y1 = *(const bf16x8*)(by + (((4 + kq) ^ (r & 7)) << 4));
#pragma unroll
                        for (int w = 0; w < NWW; ++w) {
                            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, d = (f32x4){0.f, 0.f, 0.f, 0.f};
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, xf[w][0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, xf[w][1], a, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, yf[w][0], d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, yf[w][1], d, 0, 0, 0);
                            sc[w][t] = a;
                            dp[w][t] = d;
                        }
                    }
                    // ---- P and dS for the 32 image rows x 16 own columns of every owned subtile (scalar f32 math) ------
                    bf16x8 pP[NWW], pS[NWW];
#pragma unroll
                    for (int w = 0; w < NWW; ++w) {
                        f32x4 pv[2], dsv[2];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int rbase = kb * 64 + (2 * ks + t) * 16 + kq * 4;
                            f32x4 l4, d4;
                            if (DKV) {
                                l4 = *(const f32x4*)(sL + rbase);
                                d4 = *(const f32x4*)(sD + rbase);
                            } else {
                                l4 = (f32x4){lse_own[w], lse_own[w], lse_own[w], lse_own[w]};
                                d4 = (f32x4){d_own[w], d_own[w], d_own[w], d_own[w]};
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float pe = __builtin_amdgcn_exp2f(fmaf(sc[w][t][e], c, -l4[e]));
                                if (!DKV && rbase + e >= nrows) pe = 0.f;             // keys past the end of the sequence
                                pv[t][e] = pe;
                                dsv[t][e] = pe * (dp[w][t][e] - d4[e]) * p.scale;
                            }
                        }
                        pP[w] = pack8(pv[0], pv[1]);
                        pS[w] = pack8(dsv[0], dsv[1]);
                    }
                    // ---- transposed products: A operand = image^T through transposing LDS reads ------------------------
                    const int key0 = kb * 64 + (2 * ks) * 16 + kq * 4 + tr_q;
                    const int key1 = key0 + 16;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int chunk = dt * 2 + (tr_p >> 1), off = (tr_p & 1) * 8;
                        const int o0 = key0 * 128 + ((chunk ^ (key0 & 7)) << 4) + off;
                        const int o1 = key1 * 128 + ((chunk ^ (key1 & 7)) << 4) + off;
                        const bf16x8 xt = __builtin_shufflevector(lds_tr16(sX + o0), lds_tr16(sX + o1), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int w = 0; w < NWW; ++w) acc1[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xt, pS[w], acc1[w][dt], 0, 0, 0);
                        if (DKV) {
                            const bf16x8 yt = __builtin_shufflevector(lds_tr16(sY + o0), lds_tr16(sY + o1), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                            for (int w = 0; w < NWW; ++w) acc2[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yt, pP[w], acc2[w][dt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- store: lane holds out[own = fr][d = dt*16 + kq*4 .. +3] -------------------------------------------------------
    if constexpr (NWW > 0) {
#pragma unroll
        for (int w = 0; w < NWW; ++w) {
            const int oi = own0 + w * 16 + fr;
            if (oi >= p.S) continue;
            const size_t base = (row0 + oi) * p.ldd + h * 64 + kq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (DKV) {
                    *(f32x4*)(p.dk + base + dt * 16) = acc1[w][dt];
                    *(f32x4*)(p.dv + base + dt * 16) = acc2[w][dt];
                } else {
                    *(f32x4*)(p.dq + base + dt * 16) = acc1[w][dt];
                }
            }
        }
    }
}

template <bool DKV>
__global__ __launch_bounds__(256, 2) void attn_bwd_dh64_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int ntile = (p.S + 15) >> 4, nwave = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int tbase = ntile / nwave, trem = ntile % nwave;
    const int nw = tbase + (gw < trem ? 1 : 0);                 // subtiles of this wave (<= NW by the grid choice)
    const int own0 = (gw * tbase + min(gw, trem)) * 16;
    if (nw == 2) wave_body<DKV, 2>(p, smem, own0, b, h, tid, wave);
    else if (nw == 1) wave_body<DKV, 1>(p, smem, own0, b, h, tid, wave);
    else wave_body<DKV, 0>(p, smem, own0, b, h, tid, wave);     // idle wave: staging share and barriers only
}

// D[b,h,s] = <dO[b,s,h,:], O[b,s,h,:]> and the bf16 copy of dO the MFMAs read: 16 lanes per (row, head), float4 each
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ dof, int lddof, const bf16_t* __restrict__ o, int ldo,
                                                            bf16_t* __restrict__ dob, int lddob, float* __restrict__ dsum, int B, int S,
                                                            int H) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int unit = gid >> 4, l16 = gid & 15;            // unit = (row m, head h)
    const int total = B * S * H;
    if (unit >= total) return;                             // whole 16-lane groups leave together
    const int m = unit / H, hh = unit % H;
    const float4 g = *(const float4*)(dof + (size_t)m * lddof + hh * 64 + l16 * 4);
    const uint2 ob = *(const uint2*)(o + (size_t)m * ldo + hh * 64 + l16 * 4);
    const float o0 = __uint_as_float(ob.x << 16), o1 = __uint_as_float(ob.x & 0xffff0000u);
    const float o2 = __uint_as_float(ob.y << 16), o3 = __uint_as_float(ob.y & 0xffff0000u);
    uint2 w;
    w.x = pack_bf2(g.x, g.y);
    w.y = pack_bf2(g.z, g.w);
    *(uint2*)(dob + (size_t)m * lddob + hh * 64 + l16 * 4) = w;
    float s = (g.x * o0 + g.y * o1) + (g.z * o2 + g.w * o3);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (l16 == 0) {
        const int bb = m / S, ss = m % S;
        dsum[((size_t)bb * H + hh) * S + ss] = s;
    }
}

}  // namespace

extern "C" int medp_attn_bwd_dh64_prep(const float* dout, int lddout, const void* o_bf16, int ldo, void* dout_bf16, int lddob,
                                       float* dsum, int B, int S, int H, void* stream) {
    MEDP_CHECK_ARG(dout && o_bf16 && dout_bf16 && dsum, "attn_bwd_dh64_prep: null operand");
    MEDP_CHECK_ARG(B > 0 && S > 0 && H > 0 && lddout % 4 == 0 && ldo % 4 == 0 && lddob % 4 == 0, "attn_bwd_dh64_prep: bad shape / stride");
    const long long lanes = (long long)B * S * H * 16;
    attn_bwd_prep_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, (hipStream_t)stream>>>(dout, lddout, (const bf16_t*)o_bf16, ldo,
                                                                                          (bf16_t*)dout_bf16, lddob, dsum, B, S, H);
    MEDP_LAUNCH_CHECK("medp_attn_bwd_dh64_prep");
    return 0;
}

extern "C" int medp_attn_bwd_dh64(const void* q, const void* k, const void* v, int ldqkv, const void* dout_bf16, int lddo,
                                  const float* lse, const float* dsum, float* dq, float* dk, float* dv, int ldd, int B, int S, int H,
                                  float scale, void* stream) {
    MEDP_CHECK_ARG(q && k && v && dout_bf16 && lse && dsum && dq && dk && dv, "attn_bwd_dh64: null operand");
    MEDP_CHECK_ARG(B > 0 && S > 0 && H > 0 && B <= 65535 && H <= 65535, "attn_bwd_dh64: bad shape B=%d S=%d H=%d", B, S, H);
    MEDP_CHECK_ARG(ldqkv % 8 == 0 && lddo % 8 == 0 && ldd % 4 == 0, "attn_bwd_dh64: row strides must keep 16-B alignment");
    MEDP_CHECK_ARG(scale > 0.f, "attn_bwd_dh64: scale must be positive");
    BwdParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout_bf16, lse, dsum, dq, dk, dv,
                B, S, H, ldqkv, lddo, ldd, scale, scale * 1.4426950408889634f, 0};
    p.crows = min(KC, (S + 31) / 32 * 32);
    constexpr int LDS_MAX = 2 * KC * 128 + 2 * KC * 4;
    const int LDS = 2 * p.crows * 128 + 2 * p.crows * 4;
    MEDP_ONCE_PER_DEVICE({
        hipFuncSetAttribute((const void*)attn_bwd_dh64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        hipFuncSetAttribute((const void*)attn_bwd_dh64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    });
    const int ntile = (S + 15) / 16;
    dim3 grid((ntile + 7) / 8, H, B);              // every wave owns 0..2 subtiles
    attn_bwd_dh64_kernel<false><<<grid, 256, LDS, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_attn_bwd_dh64(dq)");
    attn_bwd_dh64_kernel<true><<<grid, 256, LDS, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_attn_bwd_dh64(dkv)");
    return 0;
}
