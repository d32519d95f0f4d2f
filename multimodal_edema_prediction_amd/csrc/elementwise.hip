// Layout / cast / pointwise kernels (gfx950).  All HBM-bound: 16-B vector accesses, grid-stride, >= 2048 blocks.
#include "common.h"
#include "medp_hip.h"

namespace {

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                            int rows, int cols) {
    const int c4 = cols >> 2;
    const size_t total = (size_t)rows * c4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / c4), c = (int)(i % c4) * 4;
        const float4 v = *(const float4*)(x + (size_t)r * ldx + c);
        uint2 o;
        o.x = pack_bf2(v.x, v.y);
        o.y = pack_bf2(v.z, v.w);
        *(uint2*)(y + (size_t)r * ldy + c) = o;
    }
}

// y[c][r] = (bf16) x[r][c]   — 64x64 tiles through LDS (+1 pad), coalesced on both sides
template <typename TIN>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const TIN* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                                int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            if constexpr (sizeof(TIN) == 2) v = bf2f(((const bf16_t*)x)[(size_t)r * ldx + c]);
            else v = ((const float*)x)[(size_t)r * ldx + c];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) y[(size_t)c * ldy + r] = f2bf(tile[tx][i]);
    }
}

// Every trainable weight's GEMM operands in ONE launch (graph_step's operand pool): block b converts the 64x64 tile blk_tile[b] of job
// blk_job[b]: plain bf16 copy (forward operand, [rows, ld_plain]) and / or transposed bf16 copy (dX operand, [cols, ld_t]) of one fp32
// matrix.  Same rounding as cast_f32_bf16_kernel / transpose_to_bf16_kernel (f2bf), so a pooled operand equals the one the
// per-tensor kernels make, bit for bit.
__global__ __launch_bounds__(256) void weight_operands_multi_kernel(const MedpOperandJob* __restrict__ jobs, const int* __restrict__ blk_job,
                                                                    const int* __restrict__ blk_tile) {
    __shared__ float tile[64][65];
    const MedpOperandJob j = jobs[blk_job[blockIdx.x]];
    const int tiles_c = (j.cols + 63) / 64;
    const int t = blk_tile[blockIdx.x];
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* x = (const float*)j.src;
    bf16_t* yp = (bf16_t*)j.dst_plain;
    bf16_t* yt = (bf16_t*)j.dst_t;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < j.rows && c < j.cols) {
            v = x[(size_t)r * j.ld_src + c];
            if (yp) yp[(size_t)r * j.ld_plain + c] = f2bf(v);
        }
        tile[i][tx] = v;
    }
    if (!yt) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < j.cols && r < j.rows) yt[(size_t)c * j.ld_t + r] = f2bf(tile[tx][i]);
    }
}

// out = dy * gelu'(pre)     (pre-activation saved in fp32 or bf16)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx,
                                                       size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        const float4 g = *(const float4*)(dy + i), p = *(const float4*)(pre + i);
        *(float4*)(dx + i) = make_float4(g.x * gelu_erf_grad(p.x), g.y * gelu_erf_grad(p.y), g.z * gelu_erf_grad(p.z),
                                         g.w * gelu_erf_grad(p.w));
    }
}

// bf16 forms for the fused blocks of the trainable CXR encoder (cxr_train.py): the fc1 pre-activation is kept in bf16
// (it is the only copy the backward needs), f = gelu(pre) feeds fc2 in bf16, and the backward writes d(pre) in bf16 — the operand
// of both fc1 gradient GEMMs.  8 values (16 B) per thread; arithmetic in fp32.
__global__ __launch_bounds__(256) void gelu_bf16_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 p = *(const uint4*)(pre + i * 8);
        const uint32_t w[4] = {p.x, p.y, p.z, p.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 g = gelu_erf2((f32x2){__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)});
            o[k] = pack_bf2(g[0], g[1]);
        }
        *(uint4*)(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
__global__ __launch_bounds__(256) void gelu_bf16_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre,
                                                            bf16_t* __restrict__ dx, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 g = *(const uint4*)(dy + i * 8), p = *(const uint4*)(pre + i * 8);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, pw[4] = {p.x, p.y, p.z, p.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = __uint_as_float(gw[k] << 16) * gelu_erf_grad(__uint_as_float(pw[k] << 16));
            const float b = __uint_as_float(gw[k] & 0xffff0000u) * gelu_erf_grad(__uint_as_float(pw[k] & 0xffff0000u));
            o[k] = pack_bf2(a, b);
        }
        *(uint4*)(dx + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- ViT front end ---------------------------------------------------------------------------------------
// im2col for the 14x14 / stride-14 patch embedding: A[b*P + py*gw + px][c*196 + i*14 + j] = pix[b][c][py*14+i][px*14+j]
// written as bf16 with the row padded to `kpad` (zeros) so the GEMM sees 16-B aligned rows.
// One thread writes EIGHT consecutive k (one 16-B store) of one patch row; 32-bit index arithmetic (the launcher checks the
// sizes).  The first version wrote one bf16 per thread behind four 64-bit divisions: 43 us for a 20-MB result.
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ pix, bf16_t* __restrict__ A, int B, int C, int H,
                                                           int W, int ps, int kpad) {
    const unsigned gh = H / ps, gw = W / ps, K = C * ps * ps, pp = ps * ps, k8 = kpad >> 3;
    const unsigned total = (unsigned)B * gh * gw * k8;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned kc = i % k8, row = i / k8;
        const unsigned px = row % gw, r2 = row / gw, py = r2 % gh, b = r2 / gh;
        const float* img = pix + (size_t)b * C * H * W + (size_t)(py * ps) * W + px * ps;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned kk = kc * 8 + u;
            float x = 0.f;
            if (kk < K) {
                const unsigned c = kk / pp, rem = kk - c * pp, ii = rem / ps, jj = rem - ii * ps;
                x = img[((size_t)c * H + ii) * W + jj];
            }
            v[u] = x;
        }
        uint4 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        o.z = pack_bf2(v[4], v[5]);
        o.w = pack_bf2(v[6], v[7]);
        *(uint4*)(A + (size_t)row * kpad + kc * 8) = o;
    }
}

// x[b][0][:] = cls + pos[0] ;  x[b][1+p][:] = patch[b*P+p][:] + pos[1+p]
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int B, int P, int D) {
    const int d4 = D >> 2;
    const size_t total = (size_t)B * (P + 1) * d4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % d4) * 4;
        const size_t tokrow = i / d4;
        const int t = (int)(tokrow % (P + 1)), b = (int)(tokrow / (P + 1));
        const float4 pe = *(const float4*)(pos + (size_t)t * D + c);
        const float4 v = (t == 0) ? *(const float4*)(cls + c) : *(const float4*)(patch + ((size_t)b * P + t - 1) * D + c);
        *(float4*)(x + tokrow * D + c) = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
    }
}

// bicubic (A = -0.75, align_corners = False, no antialias) resize of the [s,s,D] position grid to [gh,gw,D];
// token 0 (class position) is copied.  Matches torch.nn.functional.interpolate(mode="bicubic").
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ __launch_bounds__(256) void pos_bicubic_kernel(const float* __restrict__ pos, float* __restrict__ out, int s, int gh, int gw,
                                                          int D) {
    const size_t total = (size_t)(gh * gw + 1) * D;
    const float A = -0.75f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % D), t = (int)(i / D);
        if (t == 0) { out[i] = pos[d]; continue; }
        const int oy = (t - 1) / gw, ox = (t - 1) % gw;
        const float sy = (float)s / (float)gh, sx = (float)s / (float)gw;
        const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float ty = fy - iy, tx = fx - ix;
        const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), s - 1);
            float rowacc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = min(max(ix - 1 + c, 0), s - 1);
                rowacc += wx[c] * pos[(size_t)(1 + yy * s + xx) * D + d];
            }
            acc += wy[a] * rowacc;
        }
        out[i] = acc;
    }
}

// Backward of pos_bicubic_kernel (the trainable CXR encoder, --unfreeze_cxr): dpos[src] = sum over the destination cells whose
// 4 x 4 footprint (with border clamping) contains src of wy * wx * dout[dst].  Written as a GATHER — one thread per (source
// cell, 4 channels) walks the destination grid in a fixed order — so it is deterministic (no float atomics).  The footprint is a
// product set, so the weight of (src, dst) factorises into (sum of the y taps that clamp onto src's row) x (the same in x).
__device__ __forceinline__ float bicubic_tap_weight(int src, int dst, int s, int g) {
    const float A = -0.75f;
    const float sc = (float)s / (float)g;
    const float f = (dst + 0.5f) * sc - 0.5f;
    const int i0 = (int)floorf(f);
    const float t = f - i0;
    const float w[4] = {cubic2(t + 1.f, A), cubic1(t, A), cubic1(1.f - t, A), cubic2(2.f - t, A)};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (min(max(i0 - 1 + a, 0), s - 1) == src) acc += w[a];
    return acc;
}
__global__ __launch_bounds__(256) void pos_bicubic_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dpos, int s, int gh,
                                                              int gw, int D) {
    const int d4n = D >> 2;
    const int total = (s * s + 1) * d4n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int d4 = i % d4n, cell = i / d4n;
        if (cell == 0) {                                      // class position: copied in the forward
            *(float4*)(dpos + 4 * d4) = *(const float4*)(dout + 4 * d4);
            continue;
        }
        const int sy = (cell - 1) / s, sx = (cell - 1) % s;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = 0; oy < gh; ++oy) {
            const float wy = bicubic_tap_weight(sy, oy, s, gh);
            if (wy == 0.f) continue;
            for (int ox = 0; ox < gw; ++ox) {
                const float wx = bicubic_tap_weight(sx, ox, s, gw);
                if (wx == 0.f) continue;
                const float w = wy * wx;
                const float4 g = *(const float4*)(dout + (size_t)(1 + oy * gw + ox) * D + 4 * d4);
                acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
            }
        }
        *(float4*)(dpos + (size_t)cell * D + 4 * d4) = acc;
    }
}

inline int grid_for(size_t work_items) { return (int)min((size_t)4096, max((size_t)1, (work_items + 255) / 256)); }

}  // namespace

extern "C" int medp_cast_f32_bf16(const float* x, int ldx, void* y, int ldy, int rows, int cols, void* stream) {
    MEDP_CHECK_ARG(x && y && rows > 0 && cols > 0, "cast: bad argument");
    MEDP_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "cast: cols, ldx, ldy must be multiples of 4");
    cast_f32_bf16_kernel<<<grid_for((size_t)rows * cols / 4), 256, 0, (hipStream_t)stream>>>(x, ldx, (bf16_t*)y, ldy, rows, cols);
    MEDP_LAUNCH_CHECK("medp_cast_f32_bf16");
    return 0;
}

extern "C" int medp_transpose_to_bf16(const void* x, int x_is_bf16, int ldx, void* y, int ldy, int rows, int cols, void* stream) {
    MEDP_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= rows, "transpose: bad argument");
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (x_is_bf16)
        transpose_to_bf16_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, cols);
    else
        transpose_to_bf16_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, ldx, (bf16_t*)y, ldy, rows, cols);
    MEDP_LAUNCH_CHECK("medp_transpose_to_bf16");
    return 0;
}

extern "C" int medp_weight_operands_multi(const MedpOperandJob* dev_jobs, const int* dev_block_job, const int* dev_block_tile, int n_blocks,
                                          void* stream) {
    MEDP_CHECK_ARG(dev_jobs && dev_block_job && dev_block_tile && n_blocks > 0, "weight_operands_multi: bad argument");
    weight_operands_multi_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>(dev_jobs, dev_block_job, dev_block_tile);
    MEDP_LAUNCH_CHECK("medp_weight_operands_multi");
    return 0;
}

extern "C" int medp_gelu_bwd(const float* dy, const float* pre, float* dx, long long n, void* stream) {
    MEDP_CHECK_ARG(dy && pre && dx && n > 0 && n % 4 == 0, "gelu_bwd: bad argument (n must be a multiple of 4)");
    gelu_bwd_kernel<<<grid_for((size_t)n / 4), 256, 0, (hipStream_t)stream>>>(dy, pre, dx, (size_t)n);
    MEDP_LAUNCH_CHECK("medp_gelu_bwd");
    return 0;
}

extern "C" int medp_gelu_bf16_fwd(const void* pre, void* out, long long n, void* stream) {
    MEDP_CHECK_ARG(pre && out && n > 0 && n % 8 == 0, "gelu_bf16_fwd: bad argument (n must be a multiple of 8)");
    gelu_bf16_fwd_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)pre, (bf16_t*)out, (size_t)n / 8);
    MEDP_LAUNCH_CHECK("medp_gelu_bf16_fwd");
    return 0;
}

extern "C" int medp_gelu_bf16_bwd(const void* dy, const void* pre, void* dx, long long n, void* stream) {
    MEDP_CHECK_ARG(dy && pre && dx && n > 0 && n % 8 == 0, "gelu_bf16_bwd: bad argument (n must be a multiple of 8)");
    gelu_bf16_bwd_kernel<<<grid_for((size_t)n / 8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)dx,
                                                                                  (size_t)n / 8);
    MEDP_LAUNCH_CHECK("medp_gelu_bf16_bwd");
    return 0;
}

extern "C" int medp_im2col_patch(const float* pix, void* A, int B, int C, int H, int W, int patch, int kpad, void* stream) {
    MEDP_CHECK_ARG(pix && A && B > 0 && C > 0 && patch > 0, "im2col: bad argument");
    MEDP_CHECK_ARG(H >= patch && W >= patch, "im2col: image %dx%d smaller than the patch size %d", H, W, patch);   /* conv stride semantics: the remainder rows/cols are ignored */
    MEDP_CHECK_ARG(kpad >= C * patch * patch && kpad % 8 == 0, "im2col: kpad must be >= C*p*p and a multiple of 8");
    const size_t total = (size_t)B * (H / patch) * (W / patch) * (kpad / 8);      // one thread per 8 columns
    MEDP_CHECK_ARG(total < (1ull << 31), "im2col: B * patches * kpad / 8 must stay below 2^31");
    im2col_patch_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(pix, (bf16_t*)A, B, C, H, W, patch, kpad);
    MEDP_LAUNCH_CHECK("medp_im2col_patch");
    return 0;
}

extern "C" int medp_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int P, int D, void* stream) {
    MEDP_CHECK_ARG(patch && cls && pos && x && B > 0 && P > 0 && D % 4 == 0, "vit_assemble: bad argument");
    vit_assemble_kernel<<<grid_for((size_t)B * (P + 1) * D / 4), 256, 0, (hipStream_t)stream>>>(patch, cls, pos, x, B, P, D);
    MEDP_LAUNCH_CHECK("medp_vit_assemble");
    return 0;
}

extern "C" int medp_pos_embed_bicubic_bwd(const float* dout, float* dpos, int src_side, int gh, int gw, int D, void* stream) {
    MEDP_CHECK_ARG(dout && dpos && src_side > 0 && gh > 0 && gw > 0 && D > 0 && D % 4 == 0, "pos_embed_bicubic_bwd: bad argument");
    pos_bicubic_bwd_kernel<<<grid_for((size_t)(src_side * src_side + 1) * (D / 4)), 256, 0, (hipStream_t)stream>>>(dout, dpos, src_side, gh, gw, D);
    MEDP_LAUNCH_CHECK("medp_pos_embed_bicubic_bwd");
    return 0;
}

extern "C" int medp_pos_embed_bicubic(const float* pos, float* out, int src_side, int gh, int gw, int D, void* stream) {
    MEDP_CHECK_ARG(pos && out && src_side > 0 && gh > 0 && gw > 0 && D > 0, "pos_embed_bicubic: bad argument");
    pos_bicubic_kernel<<<grid_for((size_t)(gh * gw + 1) * D), 256, 0, (hipStream_t)stream>>>(pos, out, src_side, gh, gw, D);
    MEDP_LAUNCH_CHECK("medp_pos_embed_bicubic");
    return 0;
}
