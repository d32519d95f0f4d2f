// Device-side batch assembly: what the reference does in a Python loop over the batch on the host —
// `Model.feats_to_input` (duett/duett.py:159-187) and the masking half of `Model.pretrain_prep_batch` (:189-237) — as one
// launch each over data that is already in HBM (the per-sample tensors `_move_lists` produced, or one pinned-buffer upload).
//
//   feats_to_input : B ragged series [T_i, 2V] (+ their bin times [T_i]) -> xs_ts [B, Tpad, 2V+1], xs_times [B, Tpad]:
//                    keep the LAST max_len steps, append the zero "mask" column, zero-pad to the longest kept length;
//                    training augmentation: values += aug_noise * N(0,1) * count column, whole timesteps dropped with
//                    probability aug_mask (row := 0, mask column := 1), static features += aug_noise * N(0,1).
//                    The normals/uniforms come from the library's counter-based hash (seed, stream id, element index,
//                    RNG epoch for graph replays) — not torch's generator: with augmentation off the result is bit-exact.
//   ssl_mask_batch : given the host's draws (masked timestep, masked event, keep table — the reference draws them from a
//                    numpy Generator, and so does the host mirror, in the same order) build the clipped input and the four
//                    target tensors in one pass; bit-exact against the reference's index_put / multiply sequence.
//
// Both are pure data movement over a few MB: HBM/latency-bound, one element per thread, coalesced on the OUTPUT side.
#include "common.h"
#include "medp_hip.h"

namespace {

struct FeatsParams {
    const float* const* ts_ptrs;      // [B] device pointers to [T_i, 2V] (or null: ts_base + i * ts_stride)
    const float* ts_base;
    long long ts_stride;
    const float* const* time_ptrs;    // [B] device pointers to [T_i]      (or null: time_base + i * time_stride)
    const float* time_base;
    long long time_stride;
    const int* lengths;               // [B] original T_i (or null: all T_uniform)
    int T_uniform;
    const float* static_in;           // [B, Ds]
    float* xs_ts;                     // [B, Tpad, 2V+1]
    float* xs_times;                  // [B, Tpad]
    float* xs_static;                 // [B, Ds]
    int B, V, Ds, max_len, Tpad;
    float aug_noise, aug_mask;
    uint32_t seed, stream_id;
    const uint32_t* epoch;
};

__device__ __forceinline__ float u01(uint32_t h) { return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)
// one standard normal per (stream, idx): Box-Muller on two hashes
__device__ __forceinline__ float normal_at(uint32_t seed, uint32_t stream, uint32_t idx) {
    const float u1 = u01(medp_hash(seed, stream, 2u * idx)), u2 = u01(medp_hash(seed, stream, 2u * idx + 1u));
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

__global__ __launch_bounds__(256) void feats_to_input_kernel(const FeatsParams p) {
    const int F = 2 * p.V + 1;
    const long long n_ts = (long long)p.B * p.Tpad * F, n_tm = (long long)p.B * p.Tpad, n_st = (long long)p.B * p.Ds;
    const uint32_t seed = medp_mix_epoch(p.seed, p.epoch);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_ts + n_tm + n_st; i += (long long)gridDim.x * 256) {
        if (i < n_ts) {
            const int c = (int)(i % F);
            const long long bt = i / F;
            const int t = (int)(bt % p.Tpad), b = (int)(bt / p.Tpad);
            const int Ti = p.lengths ? p.lengths[b] : p.T_uniform;
            const int n = min(Ti, p.max_len), off = Ti - n;             // the last max_len steps
            float v = 0.f;
            if (t < n) {
                const bool dropped = p.aug_mask > 0.f && u01(medp_hash(seed, p.stream_id + 1u, (uint32_t)(b * p.Tpad + t))) < p.aug_mask;
                if (dropped) {
                    v = c == F - 1 ? 1.f : 0.f;
                } else if (c < F - 1) {
                    const float* src = p.ts_ptrs ? p.ts_ptrs[b] : p.ts_base + (long long)b * p.ts_stride;
                    const float* row = src + (long long)(off + t) * (2 * p.V);
                    v = row[c];
                    if (p.aug_noise > 0.f && c < p.V) v += p.aug_noise * normal_at(seed, p.stream_id, (uint32_t)i) * row[c + p.V];
                }
            }
            p.xs_ts[i] = v;
        } else if (i < n_ts + n_tm) {
            const long long j = i - n_ts;
            const int t = (int)(j % p.Tpad), b = (int)(j / p.Tpad);
            const int Ti = p.lengths ? p.lengths[b] : p.T_uniform;
            const int n = min(Ti, p.max_len), off = Ti - n;
            const float* src = p.time_ptrs ? p.time_ptrs[b] : p.time_base + (long long)b * p.time_stride;
            p.xs_times[j] = t < n ? src[off + t] : 0.f;
        } else {
            const long long j = i - n_ts - n_tm;
            float v = p.static_in[j];
            if (p.aug_noise > 0.f) v += p.aug_noise * normal_at(seed, p.stream_id + 2u, (uint32_t)j);
            p.xs_static[j] = v;
        }
    }
}

struct SslParams {
    const float* xs;            // [B, T, 2V+1]
    const int* mask_t;          // [B]
    const int* event_idx;       // [B] or null
    const unsigned char* keep;  // [B, V] or null
    float *clipped, *y_ts, *y_masks, *y_events, *y_events_mask;
    int B, T, V;
};

__global__ __launch_bounds__(256) void ssl_mask_batch_kernel(const SslParams p) {
    const int F = 2 * p.V + 1;
    const long long n_x = (long long)p.B * p.T * F, n_y = (long long)p.B * p.V, n_e = p.event_idx ? (long long)p.B * p.T : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_x + n_y + n_e; i += (long long)gridDim.x * 256) {
        if (i < n_x) {
            const int c = (int)(i % F);
            const long long bt = i / F;
            const int t = (int)(bt % p.T), b = (int)(bt / p.T);
            const int mt = p.mask_t[b];
            float v = p.xs[i];
            if (t == mt) v = c == F - 1 ? 1.f : 0.f;                                    // duett.py:209-210
            if (p.event_idx) {
                const int ev = p.event_idx[b];
                if (c == ev) v = 0.f;                                                    // :216
                else if (c == ev + p.V) v = -1.f;                                        // :217
            }
            if (p.keep) {                                                                // :226-234
                bool k = true;
                if (c < F - 1) {
                    const int vv = c % p.V;
                    const float n_obs = mt < p.T ? p.xs[((long long)b * p.T + mt) * F + p.V + vv] : 0.f;
                    const float ym = fminf(fmaxf(n_obs, 0.f), 1.f);
                    k = p.keep[b * p.V + vv] != 0 || (1.f - ym) != 0.f;
                }
                v = v * ((k || v == -1.f) ? 1.f : 0.f);
            }
            p.clipped[i] = v;
        } else if (i < n_x + n_y) {
            const long long j = i - n_x;
            const int vv = (int)(j % p.V), b = (int)(j / p.V);
            const float* row = p.xs + ((long long)b * p.T + p.mask_t[b]) * F;
            p.y_ts[j] = row[vv];
            p.y_masks[j] = fminf(fmaxf(row[p.V + vv], 0.f), 1.f);
        } else {
            const long long j = i - n_x - n_y;
            const int t = (int)(j % p.T), b = (int)(j / p.T);
            const float* row = p.xs + ((long long)b * p.T + t) * F;
            const int ev = p.event_idx[b];
            p.y_events[j] = row[ev];
            p.y_events_mask[j] = fminf(fmaxf(row[ev + p.V], 0.f), 1.f);
        }
    }
}

}  // namespace

extern "C" int medp_feats_to_input(const float* const* ts_ptrs, const float* ts_base, long long ts_stride, const float* const* time_ptrs,
                                   const float* time_base, long long time_stride, const int* lengths, int T_uniform,
                                   const float* static_in, float* xs_ts, float* xs_times, float* xs_static, int B, int V, int Ds,
                                   int max_len, int Tpad, float aug_noise, float aug_mask, unsigned seed, unsigned stream_id, void* stream) {
    MEDP_CHECK_ARG((ts_ptrs || ts_base) && (time_ptrs || time_base) && static_in && xs_ts && xs_times && xs_static,
                   "feats_to_input: null operand");
    MEDP_CHECK_ARG(B > 0 && V > 0 && Ds > 0 && max_len > 0 && Tpad > 0 && Tpad <= max_len, "feats_to_input: bad shape B=%d V=%d Ds=%d max_len=%d Tpad=%d",
                   B, V, Ds, max_len, Tpad);
    MEDP_CHECK_ARG(lengths || T_uniform > 0, "feats_to_input: neither per-sample lengths nor a uniform length");
    MEDP_CHECK_ARG(aug_noise >= 0.f && aug_mask >= 0.f && aug_mask <= 1.f, "feats_to_input: bad augmentation parameters");
    MEDP_CHECK_ARG((long long)B * Tpad * (2 * V + 1) < (1ll << 31), "feats_to_input: batch too large for the 32-bit RNG index");
    FeatsParams p{ts_ptrs, ts_base, ts_stride, time_ptrs, time_base, time_stride, lengths, T_uniform, static_in, xs_ts, xs_times,
                  xs_static, B, V, Ds, max_len, Tpad, aug_noise, aug_mask, seed, stream_id, medp_rng_epoch_ptr()};
    const long long n = (long long)B * Tpad * (2 * V + 2) + (long long)B * Ds;
    const int blocks = (int)min((n + 255) / 256, (long long)4096);
    feats_to_input_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_feats_to_input");
    return 0;
}

extern "C" int medp_ssl_mask_batch(const float* xs_ts, const int* mask_t, const int* event_idx, const unsigned char* keep, float* clipped,
                                   float* y_ts, float* y_masks, float* y_events, float* y_events_mask, int B, int T, int V, void* stream) {
    MEDP_CHECK_ARG(xs_ts && mask_t && clipped && y_ts && y_masks, "ssl_mask_batch: null operand");
    MEDP_CHECK_ARG(!event_idx || (y_events && y_events_mask), "ssl_mask_batch: event targets requested without output buffers");
    MEDP_CHECK_ARG(B > 0 && T > 0 && V > 0, "ssl_mask_batch: bad shape B=%d T=%d V=%d", B, T, V);
    SslParams p{xs_ts, mask_t, event_idx, keep, clipped, y_ts, y_masks, y_events, y_events_mask, B, T, V};
    const long long n = (long long)B * T * (2 * V + 2) + (long long)B * V;
    const int blocks = (int)min((n + 255) / 256, (long long)4096);
    ssl_mask_batch_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p);
    MEDP_LAUNCH_CHECK("medp_ssl_mask_batch");
    return 0;
}
