// Row-wise argmax and temperature sampling over the vocabulary (HBM-bound passes over the
// logits).  One 1024-thread workgroup per row; 16-byte loads; (value, index) reductions keep
// the LOWEST index among equal values, like a sequential scan.
//
// Sampling uses the exponential-race form of the reference: token = argmax_i p_i / E_i with
// E_i ~ Exp(1), p = softmax(logits / T):
// score_i = (exp(z_i - max z) / sum_j exp(z_j - max z)) / max(E_i, 1e-10), z = logits / T, in fp32.
#include "common.h"

struct ValIdx {
    float v;
    int64_t i;
};
__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ValIdx block_argmax(ValIdx x, ValIdx* sm) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        ValIdx o;
        o.v = __shfl_xor(x.v, s, 64);
        o.i = __shfl_xor(x.i, s, 64);
        x = better(x, o);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = x;
    __syncthreads();
    ValIdx r = sm[0];
    for (int k = 1; k < nw; ++k) r = better(r, sm[k]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float x, float* sm) {
    x = wave_max(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = x;
    __syncthreads();
    float r = sm[0];
    for (int k = 1; k < nw; ++k) r = fmaxf(r, sm[k]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float block_sum(float x, float* sm) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = x;
    __syncthreads();
    float r = sm[0];
    for (int k = 1; k < nw; ++k) r += sm[k];
    __syncthreads();
    return r;
}

// Load 4 consecutive logits as fp32 (vocab tail handled by the caller's bounds).
template <bool F32>
__device__ __forceinline__ void load4(const void* row, int64_t i, int64_t vocab, bool vec_ok, float (&o)[4]) {
    if (vec_ok && i + 3 < vocab) {
        if constexpr (F32) {
            const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = t[e];
        } else {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>((const bf16*)row + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf2f(t[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (i + e < vocab) ? (F32 ? ((const float*)row)[i + e] : bf2f(((const bf16*)row)[i + e])) : -INFINITY;
    }
}

// Philox4x32-10 counter generator.
__device__ __forceinline__ void philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// MODE 0: argmax   MODE 1: sample with generated noise   MODE 2: sample with supplied noise
template <bool F32, int MODE>
__global__ __launch_bounds__(1024) void sample_kernel(const void* __restrict__ logits, int64_t row_stride,
                                                      const float* __restrict__ temps, const float* __restrict__ noise,
                                                      int64_t* __restrict__ out, int64_t vocab, uint64_t seed, uint64_t offset,
                                                      bool vec_ok) {
    __shared__ ValIdx sm_vi[16];
    __shared__ float sm_f[16];
    const int64_t row = blockIdx.x;
    const void* rp = F32 ? (const void*)((const float*)logits + row * row_stride) : (const void*)((const bf16*)logits + row * row_stride);
    float v[4];
    ValIdx best{-INFINITY, (int64_t)0x7fffffffffffffffLL};
    if constexpr (MODE == 0) {
        for (int64_t i = (int64_t)threadIdx.x * 4; i < vocab; i += 4096) {
            load4<F32>(rp, i, vocab, vec_ok, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < vocab) best = better(best, ValIdx{v[e], i + e});
        }
    } else {
        const float t = temps[row];
        // one pass: per-thread online (max, sum of exp), merged across the workgroup
        float mt = -INFINITY, st = 0.f;
        for (int64_t i = (int64_t)threadIdx.x * 4; i < vocab; i += 4096) {
            load4<F32>(rp, i, vocab, vec_ok, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < vocab) {
                    const float z = v[e] / t;
                    const float mn = fmaxf(mt, z);
                    st = st * expf(mt - mn) + expf(z - mn);
                    mt = mn;
                }
        }
        const float mx = block_max(mt, sm_f);
        const float denom = block_sum(mt == -INFINITY ? 0.f : st * expf(mt - mx), sm_f);
        for (int64_t i = (int64_t)threadIdx.x * 4; i < vocab; i += 4096) {
            load4<F32>(rp, i, vocab, vec_ok, v);
            float e4[4];
            if constexpr (MODE == 1) {
                uint32_t c[4] = {(uint32_t)(i >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)};
                philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = ((float)(c[e] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
                    e4[e] = -logf(u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) e4[e] = (i + e < vocab) ? noise[row * vocab + i + e] : 1.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < vocab) {
                    const float p = expf(v[e] / t - mx) / denom;
                    best = better(best, ValIdx{p / fmaxf(e4[e], 1e-10f), i + e});
                }
        }
    }
    best = block_argmax(best, sm_vi);
    if (threadIdx.x == 0) out[row] = best.i;
}

template <int MODE>
static int32_t launch_sample(const char* name, const void* logits, int32_t is_f32, int64_t row_stride, const float* temps,
                             const float* noise, int64_t* out, int64_t rows, int64_t vocab, uint64_t seed, uint64_t offset,
                             nvh_stream_t stream) {
    NVH_REQUIRE(logits && out && (MODE == 0 || temps) && (MODE != 2 || noise), NVH_EINVAL, "%s: null pointer", name);
    NVH_REQUIRE(vocab > 0 && rows >= 0, NVH_EINVAL, "%s: bad shape", name);
    NVH_REQUIRE(((uintptr_t)logits & (is_f32 ? 3 : 1)) == 0, NVH_EINVAL, "%s: logits misaligned", name);
    // rows that do not start on a 4-element boundary are read with scalar loads
    const bool vec_ok = row_stride % 4 == 0 && ((uintptr_t)logits & (is_f32 ? 15 : 7)) == 0;
    if (rows == 0) return NVH_OK;
    if (is_f32)
        sample_kernel<true, MODE><<<(unsigned)rows, 1024, 0, (hipStream_t)stream>>>(logits, row_stride, temps, noise, out, vocab, seed, offset, vec_ok);
    else
        sample_kernel<false, MODE><<<(unsigned)rows, 1024, 0, (hipStream_t)stream>>>(logits, row_stride, temps, noise, out, vocab, seed, offset, vec_ok);
    NVH_CHECK_LAUNCH(name);
    return NVH_OK;
}

extern "C" int32_t nvh_argmax(const void* logits, int32_t logits_is_f32, int64_t row_stride, int64_t* out, int64_t rows,
                              int64_t vocab, nvh_stream_t stream) {
    return launch_sample<0>("nvh_argmax", logits, logits_is_f32, row_stride, nullptr, nullptr, out, rows, vocab, 0, 0, stream);
}
extern "C" int32_t nvh_sample(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures,
                              int64_t* out, int64_t rows, int64_t vocab, uint64_t seed, uint64_t offset, nvh_stream_t stream) {
    return launch_sample<1>("nvh_sample", logits, logits_is_f32, row_stride, temperatures, nullptr, out, rows, vocab, seed, offset, stream);
}
extern "C" int32_t nvh_sample_with_noise(const void* logits, int32_t logits_is_f32, int64_t row_stride,
                                         const float* temperatures, const float* noise, int64_t* out, int64_t rows,
                                         int64_t vocab, nvh_stream_t stream) {
    return launch_sample<2>("nvh_sample_with_noise", logits, logits_is_f32, row_stride, temperatures, noise, out, rows, vocab, 0, 0, stream);
}
