// Shared device/host helpers for libmedp_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

typedef uint16_t bf16_t;                                            // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;           // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MEDP_WAVE 64

// ---- error plumbing (C ABI: 0 ok, <0 invalid argument, >0 hipError_t) --------------------------
void medp_set_error(const char* fmt, ...);
#define MEDP_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            medp_set_error(__VA_ARGS__);     \
            return -1;                       \
        }                                    \
    } while (0)
#define MEDP_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            medp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)
// Run `body` once per HIP device (thread-safe: the autograd thread and the caller's thread may race to the first launch;
// hipFuncSetAttribute applies to the CURRENT device only).  `body` is a brace block.
#define MEDP_MAX_DEVICES 16
#define MEDP_ONCE_PER_DEVICE(...)                                              \
    do {                                                                       \
        static std::once_flag once__[MEDP_MAX_DEVICES];                        \
        int dev__ = 0;                                                         \
        (void)hipGetDevice(&dev__);                                            \
        std::call_once(once__[dev__ % MEDP_MAX_DEVICES], [&]() __VA_ARGS__);   \
    } while (0)
// LDS-DMA (`global_load_lds`) is tracked by vmcnt: the data must have landed before the workgroup barrier that publishes
// the tile.  hipcc emits this wait in front of __syncthreads() today (checked in the ISA, tools/isa_hazard_audit.py); it is
// written out so that the guarantee does not hang on the compiler version.
#define MEDP_WAIT_LDS_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MEDP_TRY(expr)               \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != 0) return rc__;  \
    } while (0)

// internal (not part of the C ABI): GEMM launcher with a kernel-symbol tag; tag 1 = CXR-encoder block GEMMs
int medp_gemm_bf16_nt_tagged(int tag, const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                             const float* bias, const float* scale, const float* residual, int ldr, int act, int out_bf16,
                             void* stream);

// device pointer (may be null) to a uint32 "RNG epoch" mixed into every dropout seed: lets a captured HIP graph draw a
// fresh mask on every replay although the per-call seeds are baked into the graph (set by medp_rng_set_epoch_ptr)
const uint32_t* medp_rng_epoch_ptr();
__device__ __forceinline__ uint32_t medp_mix_epoch(uint32_t seed, const uint32_t* epoch) { return epoch ? seed + epoch[0] * 0x9E3779B9u : seed; }

// ---- bf16 <-> f32 --------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two floats -> one dword of two bf16 with ONE v_cvt_pk_bf16_f32 (the scalar casts + shift + or cost four VALU ops)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const bf2_t b = __builtin_convertvector((f32x2){lo, hi}, bf2_t);
    return __builtin_bit_cast(uint32_t, b);
}

// ---- wave reductions (64 lanes) --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level): branch-free, one v_rcp + one v_exp
// + 7 FMAs.  libm's erff costs ~4x as many VALU cycles, which showed up as ~26 % of the fc1 GEMM (GELU fused in its epilogue).
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
// exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) for a PAIR of values, written so the compiler emits packed f32 ops
// (v_pk_mul / v_pk_fma): per element 6 packed slots + v_exp + v_rcp instead of ~18 scalar VALU ops — the fused GELU was
// 42 us of the 147-us fc1 GEMM.  With r = poly(t) * exp(-x^2/2) (A&S 7.1.26):  x > 0: x - x r / 2 ;  x <= 0: x r / 2, which
// also avoids the 1 - (1 - r) cancellation for large |x|.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = (f32x2){fabsf(x[0]), fabsf(x[1])};
    const f32x2 u = x * 0.84932180028801904272f;                    // sqrt(log2(e) / 2):  exp(-x^2/2) = exp2(-u^2)
    const f32x2 w = -(u * u);
    const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    const f32x2 d = ax * 0.23164189224774112f + 1.0f;               // 1 + 0.3275911 |x| / sqrt 2
    const f32x2 t = (f32x2){__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2 h = (x * 0.5f) * (p * t * e);
    return (f32x2){x[0] > 0.f ? x[0] - h[0] : h[0], x[1] > 0.f ? x[1] - h[1] : h[1]};
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2 a = gelu_erf2((f32x2){v[0], v[1]}), b = gelu_erf2((f32x2){v[2], v[3]});
    return (f32x4){a[0], a[1], b[0], b[1]};
}
// GELU for a result that is ROUNDED TO bf16 (the fc1 epilogue of the CXR encoder): erf(z) = z P(z^2) on |z| <= 3, P of degree 8
// (weighted minimax fit, |error| <= 3e-5, constrained to 3 P(9) = 1 so that the clamp saturates to exactly +-1), no
// transcendental: 15 VALU slots per pair against ~19 + 2 v_exp + 2 v_rcp (quarter rate) for the A&S form — the fused GELU was
// 4.7 us of a 33-us fc1 tile.  |GELU error| <= 6.3e-5 (at x = 4.2, relative 1.5e-5; 3e-5 at |x| < 1): below half a bf16 ulp
// wherever |GELU(x)| >= 0.016, and an absolute 5e-5 on the vanishing negative tail.
__device__ __forceinline__ f32x2 gelu_bf16_2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 zc = (f32x2){__builtin_amdgcn_fmed3f(z[0], -3.0f, 3.0f), __builtin_amdgcn_fmed3f(z[1], -3.0f, 3.0f)};
    const f32x2 u = zc * zc;
    f32x2 p = u * 4.4700623647031534e-08f + -2.102196731357253e-06f;
    p = p * u + 4.3658408685587347e-05f;
    p = p * u + -0.0005340541829355061f;
    p = p * u + 0.00435347855091095f;
    p = p * u + -0.025454800575971603f;
    p = p * u + 0.11165805906057358f;
    p = p * u + -0.375773549079895f;
    p = p * u + 1.1283923387527466f;
    const f32x2 hx = x * 0.5f;
    return hx * (p * zc) + hx;
}
__device__ __forceinline__ f32x4 gelu_bf16_4(f32x4 v) {
    const f32x2 a = gelu_bf16_2((f32x2){v[0], v[1]}), b = gelu_bf16_2((f32x2){v[2], v[3]});
    return (f32x4){a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f32x2){x, x})[0]; }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// counter-based RNG for dropout: one 32-bit hash per (seed, stream, element) — regenerated in backward
__device__ __forceinline__ uint32_t medp_hash(uint32_t seed, uint32_t stream, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u ^ (seed + 0x7F4A7C15u * (stream + 1u));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    x += seed * 0x27D4EB2Fu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}
// keep-mask: returns scale 1/(1-p) if kept, 0 if dropped
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t stream, uint32_t idx, float p, float inv_keep) {
    const uint32_t h = medp_hash(seed, stream, idx);
    return ((float)(h >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.0f;
}
