// MFMA attention for SMALL head dims (dh <= 16) on gfx950: DuETT's event / time axis encoders — 2 heads of dim 12 over
// 49 / 97 tokens (257 at the stress shapes), dense, no mask (x_transformers Encoder as built at reference duett/duett.py:95-105,
// invoked at models/main_architecture_duett.py:81,91).  Inference form: fp32 q|k|v rows (the output of the fused QKV GEMM) in,
// bf16 out; the operands are rounded to bf16 for the matrix cores like every other GEMM operand of the path, softmax in fp32.
//
// One WAVE per (batch, head, 16-query tile), no LDS, no barrier:
//   S^T = K Q^T   v_mfma_f32_16x16x16_bf16 per 16-key tile (A := K rows, B := Q rows; dh padded to 16 with zeros)
//                 -> keys on the accumulator rows, the query on the lane column: a query's scores sit on the 4 lanes l, l+16, l+32, l+48
//   softmax       per-lane max / sum over its 4 x NT scores + a 2-step butterfly over those 4 lanes
//   O^T = V^T P^T the exponentiated accumulators, packed to bf16, ARE the B operand (same register layout); A := V^T
//                 -> lane holds 4 consecutive head-dim outputs of its query: one 8-byte store
// The fp32 VALU kernel it replaces (attention_small.hip) ran a 97-step dependent-load chain on 12 of 64 lanes for P V and took
// 73 us at B=64, N=97 for 58 MFLOP.
#include "common.h"
#include "medp_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4_t;

struct Dh16Params {
    const float* qkv;     // rows [B*N][ld]: q | k | v column blocks of H*dh each
    bf16_t* o;            // [B*N][ldo]
    int B, N, H, dh, ld, ldo;
    float scale_log2e;
};

__device__ __forceinline__ bf16x4_t pack4(float a, float b, float c, float d) {
    union { bf16x4_t v; uint32_t u[2]; } p;
    p.u[0] = pack_bf2(a, b);
    p.u[1] = pack_bf2(c, d);
    return p.v;
}

template <int NT>
__global__ __launch_bounds__(256) void attn_dh16_fwd_kernel(const Dh16Params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qt = blockIdx.x * 4 + wave;                    // 16-query tile of this wave
    if (qt * 16 >= p.N) return;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int c16 = lane & 15, g4 = lane >> 4;               // column / 4-row group of the MFMA layouts
    const int D = p.H * p.dh;
    const float* base = p.qkv + (size_t)b * p.N * p.ld + h * p.dh;
    const bool dvalid = 4 * g4 < p.dh;                       // this lane's 4 head-dim slots exist (dh is a multiple of 4)

    // Q fragment (B operand of S^T): query c16, head dims 4 g4 .. 4 g4 + 3
    const int q = qt * 16 + c16;
    bf16x4_t qf = {0, 0, 0, 0};
    if (q < p.N && dvalid) {
        const float4 x = *(const float4*)(base + (size_t)q * p.ld + 4 * g4);
        qf = pack4(x.x, x.y, x.z, x.w);
    }
    // K fragments (A operand): key kt*16 + c16, head dims 4 g4 ..; V^T fragments (A operand of O^T): head dim c16, keys kt*16 + 4 g4 ..
    bf16x4_t kf[NT], vf[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int key = kt * 16 + c16;
        kf[kt] = (bf16x4_t){0, 0, 0, 0};
        if (key < p.N && dvalid) {
            const float4 x = *(const float4*)(base + D + (size_t)key * p.ld + 4 * g4);
            kf[kt] = pack4(x.x, x.y, x.z, x.w);
        }
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        if (c16 < p.dh) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = kt * 16 + 4 * g4 + i;
                if (kk < p.N) v4[i] = base[2 * D + (size_t)kk * p.ld + c16];
            }
        }
        vf[kt] = pack4(v4[0], v4[1], v4[2], v4[3]);
    }
    // ---- S^T tiles ------------------------------------------------------------------------------------------------------
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf[kt], qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kt * 16 + 4 * g4 + r >= p.N) st[kt][r] = -INFINITY;      // padded keys
            mx = fmaxf(mx, st[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * p.scale_log2e;
    float sum = 0.f;
    bf16x4_t pf[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(fmaf(st[kt][r], p.scale_log2e, -mc));      // exp2(-inf) = 0 for the padded keys
            sum += e[r];
        }
        pf[kt] = pack4(e[0], e[1], e[2], e[3]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------------
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) ot = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[kt], pf[kt], ot, 0, 0, 0);
    if (q < p.N && dvalid) {
        const float inv = 1.0f / sum;
        uint2 o;
        o.x = pack_bf2(ot[0] * inv, ot[1] * inv);
        o.y = pack_bf2(ot[2] * inv, ot[3] * inv);
        *(uint2*)(p.o + ((size_t)b * p.N + q) * p.ldo + h * p.dh + 4 * g4) = o;
    }
}

}  // namespace

// returns -2 when the shape is outside what this kernel is built for (the caller then takes medp_attn_small_fwd)
extern "C" int medp_attn_dh16_fwd(const float* qkv, int ld, void* o_bf16, int ldo, int B, int N, int H, int dh, float scale, void* stream) {
    MEDP_CHECK_ARG(qkv && o_bf16 && B > 0 && N > 0 && H > 0 && dh > 0, "attn_dh16_fwd: bad argument");
    if (dh > 16 || dh % 4 != 0 || N > 272 || ld % 4 != 0 || ldo % 4 != 0 || (H * dh) % 4 != 0 || ((uintptr_t)qkv & 15) || ((uintptr_t)o_bf16 & 7) ||
        (long long)B * H > 65535)
        return -2;
    MEDP_CHECK_ARG(ld >= 3 * H * dh && ldo >= H * dh && scale > 0.f, "attn_dh16_fwd: bad leading dimension / scale");
    Dh16Params p{qkv, (bf16_t*)o_bf16, B, N, H, dh, ld, ldo, scale * 1.4426950408889634f};
    const int nt = (N + 15) / 16;
    dim3 grid((nt + 3) / 4, B * H);
    hipStream_t s = (hipStream_t)stream;
    if (nt <= 2) attn_dh16_fwd_kernel<2><<<grid, 256, 0, s>>>(p);
    else if (nt <= 4) attn_dh16_fwd_kernel<4><<<grid, 256, 0, s>>>(p);
    else if (nt <= 7) attn_dh16_fwd_kernel<7><<<grid, 256, 0, s>>>(p);
    else if (nt <= 10) attn_dh16_fwd_kernel<10><<<grid, 256, 0, s>>>(p);
    else attn_dh16_fwd_kernel<17><<<grid, 256, 0, s>>>(p);
    MEDP_LAUNCH_CHECK("medp_attn_dh16_fwd");
    return 0;
}
