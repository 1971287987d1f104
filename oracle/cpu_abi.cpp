// CPU build of include/nanovllm_hip.h's operator entry points (TEST INFRASTRUCTURE ONLY).
//
// Plain C++ restatement of the nano-vllm operator arithmetic behind the SAME C ABI the HIP library exports: same
// names, same arguments, same error codes - but every pointer is HOST memory and `stream` is ignored.  It exists so the
// oracle also exists in compiled form next to oracle/ref_ops.py (both are pinned by tests/golden, the vectors produced
// by the reference's own modules) and so the ABI's argument conventions - strides in elements, slot -1, padded rows,
// split-K slabs, paged and packed K/V - can be exercised without a GPU.  Built by oracle/build_cpu_abi.sh into
// oracle/_build/libnanovllm_cpu.so; loaded only by tests/.  The product package never loads it: its ops fail loudly when
// libnanovllm_hip.so is missing.
//
// Rounding points follow the reference sources (citations are /root/reference paths), as oracle/ref_ops.py does:
//   rms_forward        layers/layernorm.py:87-92     y = bf16(f32(x) * rsqrt(mean(x^2) + eps)) * w
//   add_rms_forward    layers/layernorm.py:134-140   variance from the unrounded fp32 sum
//   rotary embedding   layers/rotary_embedding.py:61-65, 148-153   half-split rotation in fp32, one rounding
//   SiluAndMul         layers/activation.py:36-58    silu rounds to bf16, the product rounds again
//   linear             layers/linear.py:198,407; layers/embed_head.py:204   fp32 accumulation, one rounding (+ bias in fp32)
//   embedding / rows   layers/embed_head.py:119-128, 197-202
//   store_kvcache      layers/attention.py:75-90
//   attention          layers/attention.py:222-248 (flash-attn semantics: bottom-right causal mask, kv head = q head / group,
//                      fp32 softmax); unpinned by reference outputs, like the Python restatement
//   sampler            layers/sampler.py:76-80       argmax(softmax(logits / T) / max(noise, 1e-10))
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/nanovllm_hip.h"

namespace {
thread_local char g_err[512] = "";
int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define REQUIRE(cond, code, ...) \
    do {                         \
        if (!(cond)) return fail(code, __VA_ARGS__); \
    } while (0)

typedef uint16_t bf16;
inline float bf2f(bf16 h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline bf16 f2bf(float f) {                       // round to nearest, ties to even
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16)(u >> 16);
}
inline float round_bf(float f) { return bf2f(f2bf(f)); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// rsqrt(mean(v^2) + eps) of `dim` fp32 values (sum in double: the order of an fp32 sum is not part of the contract)
inline float inv_rms(const float* v, int64_t dim, float eps) {
    double s = 0.0;
    for (int64_t i = 0; i < dim; ++i) s += (double)v[i] * (double)v[i];
    return 1.0f / std::sqrt((float)(s / (double)dim) + eps);
}

// K/V row of key j of a sequence: packed (tables null) or through the page table
struct KvView {
    const bf16 *k, *v;
    int64_t stride;              // packed: elements between tokens
    const int32_t* table;        // this sequence's page-table row or null
    int64_t hkv, first;          // packed: first token of the sequence
    const bf16* krow(int64_t j, int64_t h) const { return row(k, j, h); }
    const bf16* vrow(int64_t j, int64_t h) const { return row(v, j, h); }
    const bf16* row(const bf16* base, int64_t j, int64_t h) const {
        if (!table) return base + (first + j) * stride + h * NVH_HEAD_DIM;
        const int64_t page = table[j / NVH_KV_PAGE];
        return base + ((page * NVH_KV_PAGE + j % NVH_KV_PAGE) * hkv + h) * NVH_HEAD_DIM;
    }
};

// one query row against keys 0..visible-1: fp32 scores, softmax, P.V, one rounding (zeros when nothing is visible)
void attend_row(const bf16* q, const KvView& kv, int64_t kvh, int64_t visible, float scale, bf16* out) {
    constexpr int D = NVH_HEAD_DIM;
    if (visible <= 0) {
        for (int d = 0; d < D; ++d) out[d] = 0;
        return;
    }
    float qf[D];
    for (int d = 0; d < D; ++d) qf[d] = bf2f(q[d]);
    std::vector<float> s((size_t)visible);
    float mx = -INFINITY;
    for (int64_t j = 0; j < visible; ++j) {
        const bf16* kr = kv.krow(j, kvh);
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += qf[d] * bf2f(kr[d]);
        s[(size_t)j] = acc * scale;
        mx = std::max(mx, s[(size_t)j]);
    }
    double denom = 0.0;
    for (int64_t j = 0; j < visible; ++j) {
        s[(size_t)j] = std::exp(s[(size_t)j] - mx);
        denom += s[(size_t)j];
    }
    float o[D];
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    const float inv = (float)(1.0 / denom);
    for (int64_t j = 0; j < visible; ++j) {
        const bf16* vr = kv.vrow(j, kvh);
        const float p = s[(size_t)j] * inv;
        for (int d = 0; d < D; ++d) o[d] += p * bf2f(vr[d]);
    }
    for (int d = 0; d < D; ++d) out[d] = f2bf(o[d]);
}

float logit_at(const void* logits, int32_t is_f32, int64_t idx) {
    return is_f32 ? ((const float*)logits)[idx] : bf2f(((const bf16*)logits)[idx]);
}
}  // namespace

extern "C" {

int32_t nvh_version(void) { return 100; }
const char* nvh_last_error(void) { return g_err; }
int32_t nvh_device_check(int32_t) { return fail(NVH_ENODEV, "nvh_device_check: this is the CPU build of the header (no gfx950 device behind it)"); }

int32_t nvh_store_kvcache(const void* key, int64_t key_token_stride, const void* value, int64_t value_token_stride,
                          void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t num_tokens, int64_t row_elems,
                          nvh_stream_t) {
    REQUIRE(key && value && k_cache && v_cache && slot_mapping, NVH_EINVAL, "nvh_store_kvcache: null pointer");
    REQUIRE(num_tokens >= 0 && row_elems > 0, NVH_EINVAL, "nvh_store_kvcache: bad sizes");
    for (int64_t t = 0; t < num_tokens; ++t) {
        const int64_t slot = slot_mapping[t];
        if (slot < 0) continue;
        memcpy((bf16*)k_cache + slot * row_elems, (const bf16*)key + t * key_token_stride, (size_t)row_elems * 2);
        memcpy((bf16*)v_cache + slot * row_elems, (const bf16*)value + t * value_token_stride, (size_t)row_elems * 2);
    }
    return NVH_OK;
}

int32_t nvh_rmsnorm(const void* x, int64_t x_row_stride, const void* weight, void* y, int64_t y_row_stride, int64_t rows,
                    int64_t dim, float eps, nvh_stream_t) {
    REQUIRE(x && weight && y, NVH_EINVAL, "nvh_rmsnorm: null pointer");
    REQUIRE(rows >= 0 && dim > 0, NVH_EINVAL, "nvh_rmsnorm: bad sizes");
    std::vector<float> v((size_t)dim);
    for (int64_t r = 0; r < rows; ++r) {
        const bf16* xr = (const bf16*)x + r * x_row_stride;
        bf16* yr = (bf16*)y + r * y_row_stride;
        for (int64_t i = 0; i < dim; ++i) v[(size_t)i] = bf2f(xr[i]);
        const float inv = inv_rms(v.data(), dim, eps);
        for (int64_t i = 0; i < dim; ++i) yr[i] = f2bf(round_bf(v[(size_t)i] * inv) * bf2f(((const bf16*)weight)[i]));
    }
    return NVH_OK;
}

int32_t nvh_add_rmsnorm(const void* x, void* residual, const void* weight, void* y, int64_t rows, int64_t dim, float eps,
                        const float* partials, int64_t num_partials, nvh_stream_t) {
    REQUIRE(residual && weight && y && (x || partials), NVH_EINVAL, "nvh_add_rmsnorm: null pointer");
    REQUIRE(rows >= 0 && dim > 0 && (!partials || num_partials > 0), NVH_EINVAL, "nvh_add_rmsnorm: bad sizes");
    std::vector<float> s((size_t)dim);
    for (int64_t r = 0; r < rows; ++r) {
        bf16* res = (bf16*)residual + r * dim;
        for (int64_t i = 0; i < dim; ++i) {
            float xv;
            if (partials) {                      // bf16(sum of the slabs) stands in for x: the rounding point of the unsplit GEMM
                float acc = 0.f;
                for (int64_t p = 0; p < num_partials; ++p) acc += partials[(p * rows + r) * dim + i];
                xv = round_bf(acc);
            } else {
                xv = bf2f(((const bf16*)x)[r * dim + i]);
            }
            s[(size_t)i] = xv + bf2f(res[i]);
        }
        const float inv = inv_rms(s.data(), dim, eps);
        for (int64_t i = 0; i < dim; ++i) {
            res[i] = f2bf(s[(size_t)i]);
            ((bf16*)y)[r * dim + i] = f2bf(round_bf(s[(size_t)i] * inv) * bf2f(((const bf16*)weight)[i]));
        }
    }
    return NVH_OK;
}

static void rope_head(bf16* h, const float* cs, int64_t d) {
    const int64_t half = d / 2;
    for (int64_t i = 0; i < half; ++i) {
        const float a = bf2f(h[i]), b = bf2f(h[half + i]), c = cs[i], s = cs[half + i];
        const float lo = a * c - b * s, hi = b * c + a * s;
        h[i] = f2bf(lo);
        h[half + i] = f2bf(hi);
    }
}

int32_t nvh_rope(const int64_t* positions, void* q, int64_t q_token_stride, int64_t num_q_heads, void* k, int64_t k_token_stride,
                 int64_t num_kv_heads, const float* cos_sin, int64_t max_position, int64_t num_tokens, int64_t head_dim,
                 nvh_stream_t) {
    REQUIRE(positions && q && k && cos_sin, NVH_EINVAL, "nvh_rope: null pointer");
    REQUIRE(head_dim == NVH_HEAD_DIM, NVH_ENOSYS, "nvh_rope: head_dim %ld unsupported (128 only)", (long)head_dim);
    for (int64_t t = 0; t < num_tokens; ++t) {
        REQUIRE(positions[t] >= 0 && positions[t] < max_position, NVH_EINVAL, "nvh_rope: position %ld outside the table", (long)positions[t]);
        const float* cs = cos_sin + positions[t] * head_dim;
        for (int64_t h = 0; h < num_q_heads; ++h) rope_head((bf16*)q + t * q_token_stride + h * head_dim, cs, head_dim);
        for (int64_t h = 0; h < num_kv_heads; ++h) rope_head((bf16*)k + t * k_token_stride + h * head_dim, cs, head_dim);
    }
    return NVH_OK;
}

int32_t nvh_qknorm_rope_store(void* qkv, int64_t token_stride, int64_t num_q_heads, int64_t num_kv_heads, const void* q_norm_weight,
                              const void* k_norm_weight, float eps, const int64_t* positions, const float* cos_sin,
                              int64_t max_position, void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t num_tokens,
                              nvh_stream_t stream) {
    REQUIRE(qkv && q_norm_weight && k_norm_weight && positions && cos_sin, NVH_EINVAL, "nvh_qknorm_rope_store: null pointer");
    REQUIRE((k_cache == nullptr) == (v_cache == nullptr), NVH_EINVAL, "nvh_qknorm_rope_store: k_cache and v_cache must both be set or both null");
    constexpr int64_t D = NVH_HEAD_DIM;
    bf16* base = (bf16*)qkv;
    const int64_t qsz = num_q_heads * D, kvsz = num_kv_heads * D;
    // the three steps of models/qwen3.py:191-197 over views of the fused buffer
    int32_t rc = NVH_OK;
    for (int64_t t = 0; t < num_tokens; ++t) {
        bf16* row = base + t * token_stride;
        rc = nvh_rmsnorm(row, D, q_norm_weight, row, D, num_q_heads, D, eps, stream);
        if (rc == NVH_OK) rc = nvh_rmsnorm(row + qsz, D, k_norm_weight, row + qsz, D, num_kv_heads, D, eps, stream);
        if (rc != NVH_OK) return rc;
    }
    rc = nvh_rope(positions, base, token_stride, num_q_heads, base + qsz, token_stride, num_kv_heads, cos_sin, max_position, num_tokens,
                  D, stream);
    if (rc != NVH_OK || !k_cache || !slot_mapping) return rc;
    return nvh_store_kvcache(base + qsz, token_stride, base + qsz + kvsz, token_stride, k_cache, v_cache, slot_mapping, num_tokens, kvsz,
                             stream);
}

int32_t nvh_silu_mul(const void* x, void* y, int64_t rows, int64_t inter, nvh_stream_t) {
    REQUIRE(x && y, NVH_EINVAL, "nvh_silu_mul: null pointer");
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t i = 0; i < inter; ++i) {
            const float g = bf2f(((const bf16*)x)[r * 2 * inter + i]), u = bf2f(((const bf16*)x)[r * 2 * inter + inter + i]);
            const float silu = round_bf(g / (1.0f + std::exp(-g)));
            ((bf16*)y)[r * inter + i] = f2bf(silu * u);
        }
    return NVH_OK;
}

int32_t nvh_embedding(const int64_t* ids, const void* table, void* out, int64_t num_tokens, int64_t dim, int64_t vocab_start,
                      int64_t vocab_rows, nvh_stream_t) {
    REQUIRE(ids && table && out, NVH_EINVAL, "nvh_embedding: null pointer");
    for (int64_t t = 0; t < num_tokens; ++t) {
        const int64_t r = ids[t] - vocab_start;
        if (r >= 0 && r < vocab_rows) memcpy((bf16*)out + t * dim, (const bf16*)table + r * dim, (size_t)dim * 2);
        else memset((bf16*)out + t * dim, 0, (size_t)dim * 2);
    }
    return NVH_OK;
}

int32_t nvh_gather_last_rows(const void* hidden, const int32_t* cu_seqlens_q, const int32_t* need, void* out, int64_t num_rows,
                             int64_t dim, nvh_stream_t) {
    REQUIRE(hidden && cu_seqlens_q && out, NVH_EINVAL, "nvh_gather_last_rows: null pointer");
    for (int64_t i = 0; i < num_rows; ++i) {
        const int64_t j = need ? need[i] : i;
        memcpy((bf16*)out + i * dim, (const bf16*)hidden + ((int64_t)cu_seqlens_q[j + 1] - 1) * dim, (size_t)dim * 2);
    }
    return NVH_OK;
}

int64_t nvh_gemm_workspace_bytes(int64_t M, int64_t N, int32_t split_k) {
    return split_k > 1 ? (int64_t)split_k * M * N * (int64_t)sizeof(float) : 0;
}

int32_t nvh_gemm_bf16_nt(const void* x, int64_t x_row_stride, const void* w, const void* bias, void* y, int64_t y_row_stride, int64_t M,
                         int64_t N, int64_t K, int32_t split_k, float* workspace, nvh_stream_t) {
    REQUIRE(x && w && (y || workspace), NVH_EINVAL, "nvh_gemm_bf16_nt: null pointer");
    REQUIRE(K > 0 && K % 64 == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: K=%ld must be a multiple of 64", (long)K);
    REQUIRE(N > 0 && N % 4 == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: N=%ld must be a multiple of 4", (long)N);
    REQUIRE(split_k >= 1 && split_k <= 64, NVH_EINVAL, "nvh_gemm_bf16_nt: split_k %d out of range", split_k);
    if (split_k > 1) {
        REQUIRE(workspace, NVH_EINVAL, "nvh_gemm_bf16_nt: split_k needs a workspace");
        REQUIRE(!bias, NVH_EINVAL, "nvh_gemm_bf16_nt: bias is applied by the split-K consumer");
        REQUIRE(K / 64 >= split_k, NVH_EINVAL, "nvh_gemm_bf16_nt: split_k %d exceeds the %ld k-tiles", split_k, (long)(K / 64));
    } else {
        REQUIRE(y, NVH_EINVAL, "nvh_gemm_bf16_nt: null output");
    }
    const int64_t k_per_split = cdiv(K / 64, split_k) * 64;      // whole 64-wide k-tiles per slab, the last slab may be shorter
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        std::vector<float> wf((size_t)K);
        for (int64_t k = 0; k < K; ++k) wf[(size_t)k] = bf2f(((const bf16*)w)[n * K + k]);
        for (int64_t m = 0; m < M; ++m) {
            const bf16* xr = (const bf16*)x + m * x_row_stride;
            for (int32_t p = 0; p < split_k; ++p) {
                const int64_t k0 = p * k_per_split, k1 = std::min(K, k0 + k_per_split);
                float acc = 0.f;
                for (int64_t k = k0; k < k1; ++k) acc += bf2f(xr[k]) * wf[(size_t)k];
                if (split_k > 1) workspace[((int64_t)p * M + m) * N + n] = acc;
                else ((bf16*)y)[m * y_row_stride + n] = f2bf(bias ? acc + bf2f(((const bf16*)bias)[n]) : acc);
            }
        }
    }
    return NVH_OK;
}

// Residual add + RMSNorm + the projection that consumes it (models/qwen3.py:335-349 between two projections): the two
// operators above composed, with the new residual written to a second buffer; swiglu != 0: the gate/up epilogue of
// layers/activation.py:36-58 over rows interleaved in blocks of 16 (gate 16b.. | up 16b..).
int32_t nvh_add_rmsnorm_gemm_bf16_nt(const void* prev, const float* slabs, int64_t num_slabs, const void* residual_in, void* residual_out,
                                     const void* norm_weight, float eps, const void* w, void* y, int64_t y_row_stride, int64_t M, int64_t N,
                                     int64_t K, int32_t swiglu, nvh_stream_t) {
    REQUIRE(residual_in && residual_out && norm_weight && w && y, NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: null pointer");
    REQUIRE(residual_in != residual_out, NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: residual_out must not alias residual_in");
    REQUIRE(M >= 1 && M <= 64 && K >= 64 && K <= 1024 && K % 64 == 0 && N >= 1 && (swiglu ? N % 32 == 0 : N % 4 == 0), NVH_ENOSYS,
            "nvh_add_rmsnorm_gemm_bf16_nt: %ld x %ld x %ld unsupported", (long)M, (long)N, (long)K);
    std::memcpy(residual_out, residual_in, (size_t)(M * K) * sizeof(bf16));
    std::vector<bf16> xn((size_t)(M * K));
    // neither a previous projection nor slabs: the plain RMSNorm of the rows (the residual stream starts with them)
    int32_t rc = (!prev && !slabs) ? nvh_rmsnorm(residual_in, K, norm_weight, xn.data(), K, M, K, eps, nullptr)
                                   : nvh_add_rmsnorm(slabs ? nullptr : prev, residual_out, norm_weight, xn.data(), M, K, eps, slabs, slabs ? num_slabs : 0, nullptr);
    if (rc != NVH_OK) return rc;
    if (!swiglu) return nvh_gemm_bf16_nt(xn.data(), K, w, nullptr, y, y_row_stride, M, N, K, 1, nullptr, nullptr);
    std::vector<bf16> gu((size_t)(M * N));
    rc = nvh_gemm_bf16_nt(xn.data(), K, w, nullptr, gu.data(), N, M, N, K, 1, nullptr, nullptr);
    if (rc != NVH_OK) return rc;
    for (int64_t m = 0; m < M; ++m)
        for (int64_t b = 0; b < N / 32; ++b)
            for (int64_t j = 0; j < 16; ++j) {
                const float g = bf2f(gu[(size_t)(m * N + 32 * b + j)]), u = bf2f(gu[(size_t)(m * N + 32 * b + 16 + j)]);
                ((bf16*)y)[m * y_row_stride + 16 * b + j] = f2bf(round_bf(g / (1.0f + std::exp(-g))) * u);
            }
    return NVH_OK;
}

int32_t nvh_splitk_reduce(const float* partials, int64_t num_partials, const void* bias, void* y, int64_t y_row_stride, int64_t M,
                          int64_t N, nvh_stream_t) {
    REQUIRE(partials && y && num_partials >= 1, NVH_EINVAL, "nvh_splitk_reduce: bad arguments");
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int64_t p = 0; p < num_partials; ++p) acc += partials[(p * M + m) * N + n];
            if (bias) acc += bf2f(((const bf16*)bias)[n]);
            ((bf16*)y)[m * y_row_stride + n] = f2bf(acc);
        }
    return NVH_OK;
}

int32_t nvh_attn_varlen(const void* q, int64_t q_token_stride, const void* k, const void* v, int64_t kv_token_stride,
                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, const int32_t* block_tables, int64_t table_stride,
                        int64_t batch, int64_t max_seqlen_q, int64_t max_seqlen_k, int64_t total_q, int64_t num_q_heads,
                        int64_t num_kv_heads, int64_t head_dim, float softmax_scale, void* out, int32_t num_splits, void*,
                        nvh_stream_t) {
    REQUIRE(q && k && v && cu_seqlens_q && cu_seqlens_k && out, NVH_EINVAL, "nvh_attn_varlen: null pointer");
    REQUIRE(head_dim == NVH_HEAD_DIM, NVH_ENOSYS, "nvh_attn_varlen: head_dim %ld unsupported (128 only)", (long)head_dim);
    REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, NVH_EINVAL, "nvh_attn_varlen: Hq %% Hkv != 0");
    REQUIRE(num_splits >= 1 && num_splits <= 64, NVH_EINVAL, "nvh_attn_varlen: num_splits %d out of range", num_splits);
    (void)max_seqlen_q, (void)max_seqlen_k, (void)total_q;       // (key ranges are a launch matter: the result does not depend on them)
    const int64_t g = num_q_heads / num_kv_heads;
    for (int64_t b = 0; b < batch; ++b) {
        const int64_t q0 = cu_seqlens_q[b], lq = cu_seqlens_q[b + 1] - q0, lk = cu_seqlens_k[b + 1] - cu_seqlens_k[b];
        const KvView kv{(const bf16*)k, (const bf16*)v, kv_token_stride, block_tables ? block_tables + b * table_stride : nullptr,
                        num_kv_heads, cu_seqlens_k[b]};
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int64_t i = 0; i < lq; ++i)
            for (int64_t h = 0; h < num_q_heads; ++h)
                attend_row((const bf16*)q + (q0 + i) * q_token_stride + h * NVH_HEAD_DIM, kv, h / g, std::min(lk, i + (lk - lq) + 1),
                           softmax_scale, (bf16*)out + ((q0 + i) * num_q_heads + h) * NVH_HEAD_DIM);
    }
    return NVH_OK;
}

int64_t nvh_attn_varlen_workspace_bytes(int64_t total_q, int64_t num_q_heads, int64_t head_dim, int32_t num_splits) {
    return num_splits <= 1 ? 0 : total_q * num_q_heads * num_splits * (head_dim + 2) * (int64_t)sizeof(float);
}

int32_t nvh_attn_decode(const void* q, int64_t q_token_stride, const void* k_cache, const void* v_cache, const int32_t* context_lens,
                        const int32_t* block_tables, int64_t table_stride, int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                        int64_t head_dim, float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys, void*,
                        nvh_stream_t) {
    REQUIRE(q && k_cache && v_cache && context_lens && block_tables && out, NVH_EINVAL, "nvh_attn_decode: null pointer");
    REQUIRE(head_dim == NVH_HEAD_DIM, NVH_ENOSYS, "nvh_attn_decode: head_dim %ld unsupported (128 only)", (long)head_dim);
    REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, NVH_EINVAL, "nvh_attn_decode: Hq %% Hkv != 0");
    REQUIRE(num_splits >= 1 && num_splits <= 64 && min_range_keys % 64 == 0, NVH_EINVAL, "nvh_attn_decode: bad key ranges");
    const int64_t g = num_q_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t h = 0; h < num_q_heads; ++h) {
            const KvView kv{(const bf16*)k_cache, (const bf16*)v_cache, 0, block_tables + b * table_stride, num_kv_heads, 0};
            attend_row((const bf16*)q + b * q_token_stride + h * NVH_HEAD_DIM, kv, h / g, context_lens[b], softmax_scale,
                       (bf16*)out + (b * num_q_heads + h) * NVH_HEAD_DIM);
        }
    return NVH_OK;
}

int64_t nvh_attn_decode_workspace_bytes(int64_t batch, int64_t num_q_heads, int64_t head_dim, int32_t num_splits) {
    return num_splits <= 1 ? 0 : batch * num_q_heads * num_splits * (head_dim + 2) * (int64_t)sizeof(float);
}

int64_t nvh_sample_workspace_bytes(int64_t, int64_t) { return 0; }

int32_t nvh_argmax(const void* logits, int32_t logits_is_f32, int64_t row_stride, int64_t* out, int64_t rows, int64_t vocab, void*,
                   nvh_stream_t) {
    REQUIRE(logits && out && rows >= 0 && vocab > 0, NVH_EINVAL, "nvh_argmax: bad arguments");
    for (int64_t r = 0; r < rows; ++r) {
        int64_t best = 0;
        float bv = logit_at(logits, logits_is_f32, r * row_stride);
        for (int64_t i = 1; i < vocab; ++i) {
            const float v = logit_at(logits, logits_is_f32, r * row_stride + i);
            if (v > bv) bv = v, best = i;                       // strictly greater: the lowest index of the maximum
        }
        out[r] = best;
    }
    return NVH_OK;
}

int32_t nvh_sample_with_noise(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures, const float* noise,
                              int64_t* out, int64_t rows, int64_t vocab, void*, nvh_stream_t) {
    REQUIRE(logits && temperatures && noise && out && rows >= 0 && vocab > 0, NVH_EINVAL, "nvh_sample_with_noise: bad arguments");
    std::vector<float> z((size_t)vocab);
    for (int64_t r = 0; r < rows; ++r) {
        float mx = -INFINITY;
        for (int64_t i = 0; i < vocab; ++i) {
            z[(size_t)i] = logit_at(logits, logits_is_f32, r * row_stride + i) / temperatures[r];
            mx = std::max(mx, z[(size_t)i]);
        }
        double denom = 0.0;
        for (int64_t i = 0; i < vocab; ++i) denom += std::exp(z[(size_t)i] - mx);
        int64_t best = 0;
        float bv = -INFINITY;
        for (int64_t i = 0; i < vocab; ++i) {
            const float p = (float)(std::exp(z[(size_t)i] - mx) / denom);
            const float sc = p / std::max(noise[r * vocab + i], 1e-10f);
            if (sc > bv) bv = sc, best = i;
        }
        out[r] = best;
    }
    return NVH_OK;
}

}  // extern "C"
