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
#include "common.h"
#include "medp_hip.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_attn[4] = {0, 0, 0, 0};

constexpr int KC = 320;   // keys per LDS chunk (5 blocks of 64)
constexpr int NQ = 3;     // most 16-query subtiles one wave carries

struct AttnParams {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int B, S, H;
    int ldq, ldk, ldv, ldo;
    float scale_log2e;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 lds_tr16(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(addr));
}

__global__ __launch_bounds__(256, 2) void attn_fwd_dh64_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + KC * 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int ntile = (p.S + 15) >> 4, nwave = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int tbase = ntile / nwave, trem = ntile % nwave;
    const int nq = tbase + (gw < trem ? 1 : 0);                 // subtiles of this wave (wave-uniform, <= NQ)
    const int q0 = (gw * tbase + min(gw, trem)) * 16;
    const bf16_t* zero = (const bf16_t*)g_zero16_attn;
    const size_t row0 = (size_t)b * p.S;

    // ---- Q fragments (B operand: lane = query, 8 consecutive d) --------------------------------
    bf16x8 qf[NQ][2];
#pragma unroll
    for (int qs = 0; qs < NQ; ++qs) {
        const int qi = q0 + qs * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16_t* src = (qs < nq && qi < p.S) ? p.q + (row0 + qi) * p.ldq + h * 64 + s * 32 + kq * 8 : zero;
            qf[qs][s] = *(const bf16x8*)src;
        }
    }

    f32x4 o[NQ][4];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int qs = 0; qs < NQ; ++qs) {
        m_run[qs] = -INFINITY;
        l_run[qs] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qs][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // tr-read lane geometry: lane 4*qq+pp of a 16-lane group addresses row qq, columns 4pp..4pp+3
    const int tr_q = fr >> 2, tr_p = fr & 3;

    for (int c0 = 0; c0 < p.S; c0 += KC) {
        const int nkeys = min(KC, p.S - c0);
        if (c0 > 0) __syncthreads();   // previous chunk fully consumed
        // ---- stage K, V chunk (LDS-DMA, swizzle on the source chunk index) -----------------
#pragma unroll
        for (int i = 0; i < KC * 8 / 256; ++i) {
            const int qd = i * 256 + tid;
            const int row = qd >> 3, c = (qd & 7) ^ (row & 7);
            const bool ok = row < nkeys;
            const size_t grow = row0 + c0 + row;
            glds16(ok ? p.k + grow * p.ldk + h * 64 + c * 8 : zero, sK + (i * 256 + wave * 64) * 16);
            glds16(ok ? p.v + grow * p.ldv + h * 64 + c * 8 : zero, sV + (i * 256 + wave * 64) * 16);
        }
        __syncthreads();

        const int nblk = (nq > 0) ? (nkeys + 63) >> 6 : 0;
        for (int kb = 0; kb < nblk; ++kb) {
            // 16-key tiles of this block that hold at least one real key (wave-uniform; < 4 only on the ragged tail)
            const int ktv = min(4, (nkeys - kb * 64 + 15) >> 4);
            // ---- S^T = K Q^T for 64 keys x 16*nq queries ----------------------------------------
            f32x4 st[NQ][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= ktv) {
#pragma unroll
                    for (int qs = 0; qs < NQ; ++qs) st[qs][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                const int krow = kb * 64 + kt * 16 + fr;
                const char* base = sK + krow * 128;
                const bf16x8 k0 = *(const bf16x8*)(base + (((0 + kq) ^ (krow & 7)) << 4));
                const bf16x8 k1 = *(const bf16x8*)(base + (((4 + kq) ^ (krow & 7)) << 4));
#pragma unroll
                for (int qs = 0; qs < NQ; ++qs) {
                    if (qs >= nq) continue;
                    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qs][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qs][1], a, 0, 0, 0);
                    st[qs][kt] = a;
                }
            }
            // ---- scale, mask the ragged tail, online softmax ------------------------------------
            const bool tail = (kb * 64 + 64 > nkeys);
#pragma unroll
            for (int qs = 0; qs < NQ; ++qs) {
                if (qs >= nq) continue;
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kt >= ktv) continue;
                        float s = st[qs][kt][r] * p.scale_log2e;
                        if (tail && (kb * 64 + kt * 16 + kq * 4 + r >= nkeys)) s = -INFINITY;
                        st[qs][kt][r] = s;
                        mx = fmaxf(mx, s);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qs], mx);
                const float alpha = exp2f(m_run[qs] - m_new);
                m_run[qs] = m_new;
                float ls = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (kt >= ktv) continue;     // st stays 0: no weight on absent keys
                        const float e = exp2f(st[qs][kt][r] - m_new);
                        st[qs][kt][r] = e;
                        ls += e;
                    }
                l_run[qs] = l_run[qs] * alpha + ls;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[qs][dt] *= alpha;
            }
            // ---- O^T += V^T P^T ---------------------------------------------------------------------
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (2 * ks >= ktv) continue;
                bf16x8 pf[NQ];
#pragma unroll
                for (int qs = 0; qs < NQ; ++qs) {
                    const f32x4 a = st[qs][2 * ks], c = st[qs][2 * ks + 1];
                    union { bf16x8 v; uint32_t u[4]; } pk;
                    pk.u[0] = pack_bf2(a[0], a[1]);
                    pk.u[1] = pack_bf2(a[2], a[3]);
                    pk.u[2] = pack_bf2(c[0], c[1]);
                    pk.u[3] = pack_bf2(c[2], c[3]);
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
                    bf16x8 vf;
                    vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                    vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
#pragma unroll
                    for (int qs = 0; qs < NQ; ++qs)
                        if (qs < nq) o[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qs], o[qs][dt], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store: lane holds O[q = fr][d = dt*16 + kq*4 .. +3] --------------------------
#pragma unroll
    for (int qs = 0; qs < NQ; ++qs) {
        if (qs >= nq) continue;
        float l = l_run[qs];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int qi = q0 + qs * 16 + fr;
        if (qi < p.S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 w;
                w.x = pack_bf2(o[qs][dt][0] * inv, o[qs][dt][1] * inv);
                w.y = pack_bf2(o[qs][dt][2] * inv, o[qs][dt][3] * inv);
                *(uint2*)(p.o + (row0 + qi) * p.ldo + h * 64 + dt * 16 + kq * 4) = w;
            }
        }
    }
}

}  // namespace

extern "C" int medp_attn_fwd_dh64(const void* q, const void* k, const void* v, void* o, int B, int S, int H, int ldq,
                                  int ldk, int ldv, int ldo, float scale, void* stream) {
    MEDP_CHECK_ARG(q && k && v && o, "attn_fwd_dh64: null operand");
    MEDP_CHECK_ARG(B > 0 && S > 0 && H > 0, "attn_fwd_dh64: bad shape B=%d S=%d H=%d", B, S, H);
    MEDP_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd_dh64: row strides must keep 16-B alignment");
    MEDP_CHECK_ARG(B <= 65535 && H <= 65535, "attn_fwd_dh64: grid limit");
    AttnParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, B, S, H, ldq, ldk, ldv, ldo,
                 scale * 1.4426950408889634f};
    constexpr int LDS = 2 * KC * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)attn_fwd_dh64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    // floor(ntile/8) workgroups: every wave gets 1..3 subtiles (ntile < 8*(nb+1) <= 12*nb)
    const int ntile = (S + 15) / 16;
    dim3 grid(ntile >= 8 ? ntile / 8 : 1, H, B);
    attn_fwd_dh64_kernel<<<grid, 256, LDS, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_attn_fwd_dh64");
    return 0;
}
