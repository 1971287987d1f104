// Shared device helpers and host-side error plumbing for libnanovllm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/nanovllm_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ---- host side --------------------------------------------------------------------
void nvh_set_error(const char* fmt, ...);

#define NVH_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            nvh_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

#define NVH_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            nvh_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return NVH_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// Merge per-range attention partials ([row][split][128] fp32 + [row][split][2] (max, sum)) into bf16 rows.
int32_t nvh_launch_attn_combine(const float* ws_acc, const float* ws_ml, void* out, int64_t rows, int num_splits,
                                const int32_t* ctx_lens, int hq, int min_range, const int32_t* seq_active, hipStream_t st);

// Device address, on the current GPU, of the built-in partial-result array of nvh_attn_decode_noalloc (resolved on
// first use; nvh_device_check resolves it ahead of time for callers that will run under graph capture).
int32_t nvh_noalloc_scratch(float** out);
// Device address, on the current GPU, of the arrival counters the decode attention uses to merge key ranges in its own
// launch (resolved on first use, or by nvh_device_check before a graph capture).
int32_t nvh_decode_arrivals(unsigned** out);

// Routing and measurement switches.  Every kernel choice in this library is a pure function of the call's shapes; the
// switches below exist for A/B measurements and for tests that compare two routes.  They are read from the environment
// (variable NVH_<name>) ONCE per process, and only when NVH_TUNING=1 is set, so that a stray variable in a user's
// environment cannot change routing; afterwards only nvh_set_switch() changes them.  A switch is either unset (-1) or holds
// a non-negative value (a variable that is set to something that is not a number counts as 1).
#define NVH_SWITCHES(X)                                                                                              \
    X(DECODE_TARGET_WGS) X(DECODE_FULL_WGS) X(DECODE_MERGE_IN_LAUNCH) X(ATTN_KS) X(VARLEN_LONG_SPLITS)             \
    X(ATTN_HEAD_MAJOR) X(ATTN_RB1) X(NORM_NO_WIDE) X(GEMM_NO_WIDE_STORE) X(GEMM_NO_XCD) X(GEMM_NO_NT)              \
    X(GEMM_NO_SKINNY) X(GEMM_M32_SKINNY) X(GEMM_FORCE_TILE) X(GEMM_NO_256) X(NO_KWAVES) X(KWAVES_KMIN)             \
    X(KWAVES_NO_8) X(GEMM_S128) X(GEMM_S64) X(GEMM_NO_ROUND_CUT) X(GEMM_NO_16x64) X(GEMM_NO_KGROUPS)               \
    X(QKV_NO_8WAVE) X(SWIGLU_NO_192) X(SWIGLU_NO_96) X(NO_FUSED_QKV) X(NO_NORM_GEMM) X(NORM_GEMM_KB)               \
    X(NORM_GEMM_ROWS) X(DECODE_SLABS) X(NO_FUSED_EXCHANGE) X(NO_FUSED_DECODE)        \
    X(LM_HEAD_NO_CUT) X(NO_ROWSCALE) X(EXP_A) X(EXP_B)
enum NvhSwitch : int {
#define NVH_SW_ENUM(n) SW_##n,
    NVH_SWITCHES(NVH_SW_ENUM)
#undef NVH_SW_ENUM
    SW_COUNT
};
int64_t nvh_switch(NvhSwitch s);                                   // -1: unset
static inline bool nvh_switch_on(NvhSwitch s) { return nvh_switch(s) >= 0; }
static inline int64_t nvh_switch_or(NvhSwitch s, int64_t dflt) { const int64_t v = nvh_switch(s); return v >= 0 ? v : dflt; }

static inline bool nvh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t nvh_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device side ------------------------------------------------------------------
// Sum / max over the 16 lanes of a DPP row (lanes 16r..16r+15); every lane of the row
// receives the result.  quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror.
#define NVH_DPP(x, ctrl) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xF, 0xF, true))

__device__ __forceinline__ float row16_sum(float x) {
    x += NVH_DPP(x, 0xB1);
    x += NVH_DPP(x, 0x4E);
    x += NVH_DPP(x, 0x141);
    x += NVH_DPP(x, 0x140);
    return x;
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, NVH_DPP(x, 0xB1));
    x = fmaxf(x, NVH_DPP(x, 0x4E));
    x = fmaxf(x, NVH_DPP(x, 0x141));
    x = fmaxf(x, NVH_DPP(x, 0x140));
    return x;
}
// Full 64-lane reductions (result in every lane).
__device__ __forceinline__ float wave_sum(float x) {
    x = row16_sum(x);
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
    x = row16_max(x);
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// Per-head RMSNorm -> bf16 -> x weight -> bf16 -> half-split RoPE -> bf16 for one head of 128 elements held
// by a 16-lane DPP row: lane c owns elements [8c, 8c+8).  The rotation partner (i +- 64) lives in lane c^8
// and arrives through a row rotation by 8.  cs_row: fp32 [cos(64) | sin(64)] of the token's position.
// Used by every kernel that applies the q/k epilogue, so they all produce identical bits.
__device__ __forceinline__ bf16x8 qknorm_rope_head_cs(bf16x8 x, bf16x8 w, f32x4 c0, f32x4 c1, f32x4 s0, f32x4 s1, int c, float eps) {
    float v[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = bf2f(x[j]);
        ss += v[j] * v[j];
    }
    ss = row16_sum(ss);
    const float inv = 1.0f / sqrtf(ss / 128.0f + eps);
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float n = bf2f(f2bf(bf2f(f2bf(v[j] * inv)) * bf2f(w[j])));
        const float p = NVH_DPP(n, 0x128);                       // row_ror:8 -> lane c^8's value
        const float cs = j < 4 ? c0[j] : c1[j - 4];
        const float sn = j < 4 ? s0[j] : s1[j - 4];
        // first half: x1*cos - x2*sin (own = x1); second half: x2*cos + x1*sin (own = x2)
        const float y = (c < 8) ? n * cs - p * sn : n * cs + p * sn;
        out[j] = f2bf(y);
    }
    return out;
}
__device__ __forceinline__ bf16x8 qknorm_rope_head(bf16x8 x, bf16x8 w, const float* __restrict__ cs_row, int c, float eps) {
    // lane c's cos / sin values: columns 8 (c & 7) .. + 7 of the position's row
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs_row + 8 * (c & 7));
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs_row + 8 * (c & 7) + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs_row + 64 + 8 * (c & 7));
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(cs_row + 64 + 8 * (c & 7) + 4);
    return qknorm_rope_head_cs(x, w, c0, c1, s0, s1, c, eps);
}

// ---- sampling pieces shared by the sampler kernels and the LM-head GEMM epilogue ------------------
struct ValIdx {
    float v;
    int64_t i;
};
__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
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

// Scores of the exponential race for 4 consecutive columns col .. col+3 (col a multiple of 4) of `row`:
// z / T - ln(E), E ~ Exp(1) from Philox keyed by (seed; col / 4, row, offset).  score = z - ln(E), E = -ln(u), u uniform
// in (0,1): hardware log2 (1 ulp) is ample for noise.  23 random bits -> u = (k + 1/2) * 2^-23: every step is exact in
// fp32 and u stays inside [2^-24, 1 - 2^-24], so E is never 0; the clamp is the one the reference applies
// (layers/sampler.py:80, clamp_min_(1e-10)).
__device__ __forceinline__ void nvh_race_scores(const float (&z)[4], float inv_t, int64_t col, int64_t row, uint64_t seed, uint64_t offset,
                                                float (&sc)[4]) {
    uint32_t c[4] = {(uint32_t)(col >> 2), (uint32_t)row, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    constexpr float LN2 = 0.6931471805599453f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float un = (float)(c[e] >> 9) * 0x1p-23f + 0x1p-24f;
        const float ex = fmaxf(-LN2 * __builtin_amdgcn_logf(un), 1e-10f);       // Exp(1) sample
        sc[e] = __builtin_fmaf(z[e], inv_t, -LN2 * __builtin_amdgcn_logf(ex));
    }
}

// Pick among `slices` candidates per row (sampler.hip): out[row] = index of the best.
int32_t nvh_launch_sample_final(const void* cand, int64_t* out, int64_t rows, int slices, hipStream_t st);
