// Multi-tensor AdamW for gfx950: ONE launch updates every trainable tensor (torch.optim.AdamW semantics, decoupled
// weight decay; reference trainer.py:383 with per-group lr from _make_param_groups :77-116).
//   p -= lr*wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// A device table of tensor descriptors + a block->(tensor, chunk) map replace the per-tensor launches of the eager
// foreach path; HBM-bound: 16-B accesses, each element read/written once (p, m, v) + one read of g.
#include "common.h"
#include "medp_hip.h"

namespace {
constexpr int CHUNK = 4096;   // elements per block (256 threads x 4 float4)

__global__ __launch_bounds__(256) void adamw_multi_kernel(const MedpAdamTensor* __restrict__ descs, const int* __restrict__ blk_tensor,
                                                          const int* __restrict__ blk_chunk, float beta1, float beta2, float eps,
                                                          float bc1_host, float bc2_sqrt_host, float grad_scale,
                                                          const unsigned* __restrict__ dev_step) {
    // bias corrections from a DEVICE step counter when given (graph replay: the host step count is frozen in the graph)
    float bc1 = bc1_host, bc2_sqrt = bc2_sqrt_host;
    if (dev_step) {
        const float st = (float)dev_step[0];
        bc1 = 1.f - powf(beta1, st);
        bc2_sqrt = sqrtf(1.f - powf(beta2, st));
    }
    const MedpAdamTensor d = descs[blk_tensor[blockIdx.x]];
    const long long base = (long long)blk_chunk[blockIdx.x] * CHUNK;
    const float lr = d.lr, decay = 1.f - d.lr * d.weight_decay, step_size = d.lr / bc1;
    float* p = (float*)d.param;
    const float* g = (const float*)d.grad;
    float* m = (float*)d.exp_avg;
    float* v = (float*)d.exp_avg_sq;
    (void)lr;
    const bool vec = ((d.numel & 3) == 0);
    if (vec) {
        for (long long i = base + threadIdx.x * 4; i < min(base + CHUNK, d.numel); i += 1024) {
            float4 pp = *(float4*)(p + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
            const float4 gg = *(const float4*)(g + i);
            float* pa = (float*)&pp; float* ma = (float*)&mm; float* va = (float*)&vv; const float* ga = (const float*)&gg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = ga[e] * grad_scale;
                pa[e] *= decay;
                ma[e] = beta1 * ma[e] + (1.f - beta1) * gr;
                va[e] = beta2 * va[e] + (1.f - beta2) * gr * gr;
                pa[e] -= step_size * ma[e] / (sqrtf(va[e]) / bc2_sqrt + eps);
            }
            *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
        }
    } else {
        for (long long i = base + threadIdx.x; i < min(base + CHUNK, d.numel); i += 256) {
            const float gr = g[i] * grad_scale;
            float pp = p[i] * decay;
            const float mm = beta1 * m[i] + (1.f - beta1) * gr;
            const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
            pp -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}
}  // namespace

extern "C" int medp_adamw_chunk_elems(void) { return CHUNK; }

extern "C" int medp_adamw_multi(const MedpAdamTensor* dev_descs, const int* dev_block_tensor, const int* dev_block_chunk, int n_blocks,
                                float beta1, float beta2, float eps, int step, const unsigned* dev_step, float grad_scale, void* stream) {
    MEDP_CHECK_ARG(dev_descs && dev_block_tensor && dev_block_chunk && n_blocks > 0 && (step >= 1 || dev_step), "adamw_multi: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    adamw_multi_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>(dev_descs, dev_block_tensor, dev_block_chunk, beta1, beta2, eps, bc1, bc2s, grad_scale, dev_step);
    MEDP_LAUNCH_CHECK("medp_adamw_multi");
    return 0;
}
