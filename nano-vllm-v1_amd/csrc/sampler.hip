// Row-wise argmax and temperature sampling over the vocabulary: ONE pass over the logits, rows cut
// into slices so that ~1024 workgroups stream them; (value, index) reductions keep the LOWEST
// index among equal values, like a sequential scan.
//
// Sampling is the reference's exponential race, token = argmax_i p_i / E_i with E_i ~ Exp(1) and
// p = softmax(logits / T), evaluated in the log domain: log(p_i / E_i) = z_i - log E_i - logsumexp(z),
// z = logits / T.  The last term is a per-row constant, so score_i = z_i - log(max(E_i, 1e-10)) has the
// same argmax and needs neither the row maximum nor the softmax denominator (no extra passes, no
// underflow).  It can differ from the reference's fp32 evaluation only when two candidates are within
// rounding of each other.
#include "common.h"

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

// MODE 0: argmax   MODE 1: sample with generated noise   MODE 2: sample with supplied noise
// Grid (slices, rows): every workgroup scans one slice of one row and leaves its best candidate in
// `cand` (or in `out` directly when there is one slice); sample_final_kernel picks among slices.
template <bool F32, int MODE>
__global__ __launch_bounds__(256) void sample_slice_kernel(const void* __restrict__ logits, int64_t row_stride,
                                                           const float* __restrict__ temps, const float* __restrict__ noise,
                                                           int64_t* __restrict__ out, ValIdx* __restrict__ cand, int64_t vocab,
                                                           int64_t slice_len, uint64_t seed, uint64_t offset, bool vec_ok,
                                                           int64_t col_offset) {
    __shared__ ValIdx sm_vi[4];
    const int64_t row = blockIdx.y;
    const int64_t lo = (int64_t)blockIdx.x * slice_len;
    const int64_t hi = min(vocab, lo + slice_len);
    const void* rp = F32 ? (const void*)((const float*)logits + row * row_stride) : (const void*)((const bf16*)logits + row * row_stride);
    float t = 1.f, inv_t = 1.f;
    if constexpr (MODE != 0) {
        t = temps[row];
        inv_t = 1.0f / t;
    }
    // The lane scans increasing indices, so "strictly greater" keeps the lowest index among equals; the index stays
    // 32-bit (relative to the slice start) until the cross-lane reduction.
    float best_v = -INFINITY;
    int best_i = 0x7fffffff;
    // two independent 4-element groups per trip keep two loads in flight per lane
    for (int64_t i0 = lo + (int64_t)threadIdx.x * 4; i0 < hi; i0 += 2048) {
        float v[2][4];
        load4<F32>(rp, i0, hi, vec_ok, v[0]);
        const int64_t i1 = i0 + 1024;
        if (i1 < hi) load4<F32>(rp, i1, hi, vec_ok, v[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = u ? i1 : i0;
            if (i >= hi) continue;
            float sc[4];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[e] = v[u][e];
            } else if constexpr (MODE == 1) {
                // score = z - ln(E), E = -ln(u), u uniform in (0,1): hardware log2 (1 ulp) is ample for noise
                // (col_offset: the row is a vocabulary shard; noise and result are those of the full row's columns)
                nvh_race_scores(v[u], inv_t, i + col_offset, row, seed, offset, sc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ex = (i + e < hi) ? noise[row * vocab + i + e] : 1.f;
                    sc[e] = v[u][e] / t - logf(fmaxf(ex, 1e-10f));
                }
            }
            const int rel = (int)(i - lo);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < hi && sc[e] > best_v) {
                    best_v = sc[e];
                    best_i = rel + e;
                }
        }
    }
    // a lane whose elements are all -inf still offers its first index (rows of -inf resolve to index 0, as a scan would)
    const int64_t first = lo + (int64_t)threadIdx.x * 4;
    ValIdx best{best_v, best_i != 0x7fffffff ? lo + best_i : (first < hi ? first : (int64_t)0x7fffffffffffffffLL)};
    best = block_argmax(best, sm_vi);
    if (threadIdx.x == 0) {
        if (best.i != (int64_t)0x7fffffffffffffffLL) best.i += col_offset;
        if (out != nullptr && gridDim.x == 1) out[row] = best.i;
        else cand[row * gridDim.x + blockIdx.x] = best;
    }
}

__global__ __launch_bounds__(64) void sample_final_kernel(const ValIdx* __restrict__ cand, int64_t* __restrict__ out, int slices) {
    const int64_t row = blockIdx.x;
    ValIdx best{-INFINITY, (int64_t)0x7fffffffffffffffLL};
    for (int s = threadIdx.x; s < slices; s += 64) best = better(best, cand[row * slices + s]);
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        ValIdx o;
        o.v = __shfl_xor(best.v, sft, 64);
        o.i = __shfl_xor(best.i, sft, 64);
        best = better(best, o);
    }
    if (threadIdx.x == 0) out[row] = best.i;
}

int32_t nvh_launch_sample_final(const void* cand, int64_t* out, int64_t rows, int slices, hipStream_t st) {
    sample_final_kernel<<<(unsigned)rows, 64, 0, st>>>((const ValIdx*)cand, out, slices);
    NVH_CHECK_LAUNCH("sample_final");
    return NVH_OK;
}

static int sample_slices(int64_t rows, int64_t vocab) {
    // ~1024 workgroups in total, at least 2048 elements per slice, at most 64 slices
    int64_t s = 1024 / (rows > 0 ? rows : 1);
    const int64_t cap = vocab / 2048 > 1 ? vocab / 2048 : 1;
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t nvh_sample_workspace_bytes(int64_t rows, int64_t vocab) {
    const int s = sample_slices(rows, vocab);
    return s > 1 ? rows * s * (int64_t)sizeof(ValIdx) : 0;
}

template <int MODE>
static int32_t launch_sample(const char* name, const void* logits, int32_t is_f32, int64_t row_stride, const float* temps,
                             const float* noise, int64_t* out, int64_t rows, int64_t vocab, uint64_t seed, uint64_t offset,
                             void* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(logits && out && (MODE == 0 || temps) && (MODE != 2 || noise), NVH_EINVAL, "%s: null pointer", name);
    NVH_REQUIRE(vocab > 0 && rows >= 0 && rows <= 65535, NVH_EINVAL, "%s: bad shape", name);
    NVH_REQUIRE(((uintptr_t)logits & (is_f32 ? 3 : 1)) == 0, NVH_EINVAL, "%s: logits misaligned", name);
    if (rows == 0) return NVH_OK;
    const int slices = sample_slices(rows, vocab);
    NVH_REQUIRE(slices == 1 || (workspace && nvh_aligned16(workspace)), NVH_EINVAL, "%s: workspace of nvh_sample_workspace_bytes() needed", name);
    // rows that do not start on a 4-element boundary are read with scalar loads
    int64_t slice_len = (nvh_cdiv(vocab, slices) + 3) & ~int64_t(3);
    const bool vec_ok = row_stride % 4 == 0 && ((uintptr_t)logits & (is_f32 ? 15 : 7)) == 0;
    dim3 grid((unsigned)slices, (unsigned)rows);
    hipStream_t st = (hipStream_t)stream;
    if (is_f32)
        sample_slice_kernel<true, MODE><<<grid, 256, 0, st>>>(logits, row_stride, temps, noise, out, (ValIdx*)workspace, vocab, slice_len, seed, offset, vec_ok, 0);
    else
        sample_slice_kernel<false, MODE><<<grid, 256, 0, st>>>(logits, row_stride, temps, noise, out, (ValIdx*)workspace, vocab, slice_len, seed, offset, vec_ok, 0);
    NVH_CHECK_LAUNCH(name);
    if (slices > 1) {
        sample_final_kernel<<<(unsigned)rows, 64, 0, st>>>((const ValIdx*)workspace, out, slices);
        NVH_CHECK_LAUNCH(name);
    }
    return NVH_OK;
}

extern "C" int32_t nvh_argmax(const void* logits, int32_t logits_is_f32, int64_t row_stride, int64_t* out, int64_t rows,
                              int64_t vocab, void* workspace, nvh_stream_t stream) {
    return launch_sample<0>("nvh_argmax", logits, logits_is_f32, row_stride, nullptr, nullptr, out, rows, vocab, 0, 0, workspace, stream);
}
extern "C" int32_t nvh_sample(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures,
                              int64_t* out, int64_t rows, int64_t vocab, uint64_t seed, uint64_t offset, void* workspace,
                              nvh_stream_t stream) {
    return launch_sample<1>("nvh_sample", logits, logits_is_f32, row_stride, temperatures, nullptr, out, rows, vocab, seed, offset, workspace, stream);
}
extern "C" int32_t nvh_sample_with_noise(const void* logits, int32_t logits_is_f32, int64_t row_stride,
                                         const float* temperatures, const float* noise, int64_t* out, int64_t rows,
                                         int64_t vocab, void* workspace, nvh_stream_t stream) {
    return launch_sample<2>("nvh_sample_with_noise", logits, logits_is_f32, row_stride, temperatures, noise, out, rows, vocab, 0, 0, workspace, stream);
}

// ---- sampling over vocabulary shards (tensor parallelism) without gathering the logits: every rank picks the best
// candidate of its shard with the noise of the full row's columns, the candidates (score + index: 7 bytes) travel as bf16
// bytes through the ranks' sum - each rank fills its own segment of `enc` [world][rows][8], zeros elsewhere - and every
// rank picks the same winner.  Same result as nvh_sample over the gathered row: the candidates are compared with the same
// rule (higher score, then lower index), whose outcome does not depend on how the row is cut.
__global__ __launch_bounds__(64) void sample_shard_final_kernel(const ValIdx* __restrict__ cand, int slices, bf16* __restrict__ enc, int rank,
                                                                int world, int rows) {
    const int64_t row = blockIdx.x;
    ValIdx best{-INFINITY, (int64_t)0x7fffffffffffffffLL};
    for (int s = threadIdx.x; s < slices; s += 64) best = better(best, cand[row * slices + s]);
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        ValIdx o;
        o.v = __shfl_xor(best.v, sft, 64);
        o.i = __shfl_xor(best.i, sft, 64);
        best = better(best, o);
    }
    const uint32_t vb = __float_as_uint(best.v);
    const uint32_t ib = (uint32_t)(best.i & 0xffffff);
    for (int t = threadIdx.x; t < world * 8; t += 64) {
        const int r = t >> 3, k = t & 7;
        float byte = 0.f;
        if (r == rank) byte = (float)(k < 4 ? (vb >> (8 * k)) & 255u : k < 7 ? (ib >> (8 * (k - 4))) & 255u : 0u);
        enc[((int64_t)r * rows + row) * 8 + k] = f2bf(byte);
    }
}
__global__ __launch_bounds__(256) void sample_pick_kernel(const bf16* __restrict__ enc, int64_t* __restrict__ out, int rows, int world) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    ValIdx best{-INFINITY, (int64_t)0x7fffffffffffffffLL};
    for (int r = 0; r < world; ++r) {
        const bf16* e = enc + ((int64_t)r * rows + row) * 8;
        uint32_t vb = 0, ib = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) vb |= (uint32_t)bf2f(e[k]) << (8 * k);
#pragma unroll
        for (int k = 0; k < 3; ++k) ib |= (uint32_t)bf2f(e[4 + k]) << (8 * k);
        best = better(best, ValIdx{__uint_as_float(vb), (int64_t)ib});
    }
    out[row] = best.i;
}

extern "C" int64_t nvh_sample_shard_workspace_bytes(int64_t rows, int64_t shard_vocab) {
    return rows * sample_slices(rows, shard_vocab) * (int64_t)sizeof(ValIdx);
}

extern "C" int32_t nvh_sample_shard(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures, void* enc,
                                    int64_t rows, int64_t shard_vocab, int64_t col_offset, int32_t rank, int32_t world, uint64_t seed,
                                    uint64_t offset, void* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(logits && temperatures && enc && workspace && nvh_aligned16(workspace), NVH_EINVAL, "nvh_sample_shard: null pointer");
    NVH_REQUIRE(shard_vocab > 0 && rows >= 0 && rows <= 65535 && world >= 1 && world <= 8 && rank >= 0 && rank < world, NVH_EINVAL,
                "nvh_sample_shard: bad shape");
    NVH_REQUIRE(col_offset >= 0 && col_offset % 4 == 0 && col_offset + shard_vocab <= (1 << 24), NVH_EINVAL,
                "nvh_sample_shard: the shard must start on a multiple of 4 columns and the vocabulary fit 24 bits");
    NVH_REQUIRE(((uintptr_t)logits & (logits_is_f32 ? 3 : 1)) == 0, NVH_EINVAL, "nvh_sample_shard: logits misaligned");
    if (rows == 0) return NVH_OK;
    const int slices = sample_slices(rows, shard_vocab);
    const int64_t slice_len = (nvh_cdiv(shard_vocab, slices) + 3) & ~int64_t(3);
    const bool vec_ok = row_stride % 4 == 0 && ((uintptr_t)logits & (logits_is_f32 ? 15 : 7)) == 0;
    dim3 grid((unsigned)slices, (unsigned)rows);
    hipStream_t st = (hipStream_t)stream;
    if (logits_is_f32)
        sample_slice_kernel<true, 1><<<grid, 256, 0, st>>>(logits, row_stride, temperatures, nullptr, nullptr, (ValIdx*)workspace, shard_vocab, slice_len, seed, offset, vec_ok, col_offset);
    else
        sample_slice_kernel<false, 1><<<grid, 256, 0, st>>>(logits, row_stride, temperatures, nullptr, nullptr, (ValIdx*)workspace, shard_vocab, slice_len, seed, offset, vec_ok, col_offset);
    NVH_CHECK_LAUNCH("nvh_sample_shard");
    sample_shard_final_kernel<<<(unsigned)rows, 64, 0, st>>>((const ValIdx*)workspace, slices, (bf16*)enc, rank, world, (int)rows);
    NVH_CHECK_LAUNCH("nvh_sample_shard");
    return NVH_OK;
}

extern "C" int32_t nvh_sample_pick(const void* enc, int64_t* out, int64_t rows, int32_t world, nvh_stream_t stream) {
    NVH_REQUIRE(enc && out && rows >= 0 && rows <= 65535 && world >= 1 && world <= 8, NVH_EINVAL, "nvh_sample_pick: bad arguments");
    if (rows == 0) return NVH_OK;
    sample_pick_kernel<<<(unsigned)nvh_cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>((const bf16*)enc, out, (int)rows, world);
    NVH_CHECK_LAUNCH("nvh_sample_pick");
    return NVH_OK;
}
