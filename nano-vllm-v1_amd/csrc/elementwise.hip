// HBM-bound row kernels: RMSNorm, add+RMSNorm, RoPE, fused QK-norm+RoPE+KV-store,
// SiLU*mul, embedding gather, LM-head row gather, KV-cache scatter store.
// Every access is a 16-byte (8 x bf16) vector per lane; reductions are wave/DPP shuffles.
// Compiled with -ffp-contract=off so fp32 products and sums round separately, matching the
// rounding points of the reference's eager arithmetic.
#include "common.h"
#include <cstdlib>

// ------------------------------------------------------------------ RMSNorm family
// LPR lanes cooperate on one row (16 for head_dim-sized rows, 64 for hidden-sized rows);
// each lane keeps up to MAXC 16-byte chunks of its row in registers.
// With PARTIALS the input is the sum of `num_partials` fp32 split-K slabs, rounded once to bf16 as the GEMM
// would have.  The slabs are always summed the same way whatever the launch shape: slab p goes to partial sum
// p mod 4 in increasing p, the total is (s0 + s1) + (s2 + s3).  WIDE (a handful of rows, many slabs): four waves
// per row, wave w gathers partial sum w, the first wave finishes the row.
template <int LPR, int MAXC, bool ADD, bool PARTIALS, int THREADS = 256, bool WIDE = false>
__global__ __launch_bounds__(THREADS) void rmsnorm_kernel(
    const bf16* __restrict__ x, int64_t x_stride, bf16* __restrict__ residual,
    const bf16* __restrict__ w, bf16* __restrict__ y, int64_t y_stride,
    int64_t rows, int dim, float eps, const float* __restrict__ partials, int num_partials) {
    static_assert(!WIDE || (PARTIALS && LPR == 64 && THREADS == 256), "WIDE: four waves on one row of slabs");
    constexpr int RPB = WIDE ? 1 : THREADS / LPR;        // rows per block
    const int lane_in_row = threadIdx.x % LPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + (WIDE ? 0 : threadIdx.x / LPR);
    const bool active = row < rows;
    const int nchunk = dim >> 3;
    const int64_t r = active ? row : rows - 1;           // inactive lanes shadow the last row

    __shared__ float wide_part[WIDE ? 3 : 1][WIDE ? MAXC : 1][WIDE ? 64 : 1][8];
    if constexpr (WIDE) {
        const int wave = threadIdx.x >> 6;
        float acc[MAXC][8];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        }
        const int64_t slab = rows * (int64_t)dim;
        for (int p = wave; p < num_partials; p += 4) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane_in_row + i * LPR;
                if (c < nchunk) {
                    const float* src = partials + p * slab + r * (int64_t)dim + c * 8;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[i][j] += a[j]; acc[i][4 + j] += b[j]; }
                }
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) wide_part[wave - 1][i][lane_in_row][j] = acc[i][j];
            }
        }
        __syncthreads();
        if (wave > 0) return;
        // first wave: (s0 + s1) + (s2 + s3), then the row as below
        float v[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane_in_row + i * LPR;
            if (c < nchunk) {
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = (acc[i][j] + wide_part[0][i][lane_in_row][j]) + (wide_part[1][i][lane_in_row][j] + wide_part[2][i][lane_in_row][j]);
                    xv[j] = bf2f(f2bf(t));
                }
                if constexpr (ADD) {
                    bf16x8 t = *reinterpret_cast<const bf16x8*>(residual + r * (int64_t)dim + c * 8);
                    bf16x8 nr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xv[j] = xv[j] + bf2f(t[j]);
                        nr[j] = f2bf(xv[j]);
                    }
                    *reinterpret_cast<bf16x8*>(residual + r * (int64_t)dim + c * 8) = nr;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[i][j] = xv[j];
                    ss += xv[j] * xv[j];
                }
            }
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane_in_row + i * LPR;
            if (c < nchunk) {
                bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c * 8);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(v[i][j] * inv)) * bf2f(wv[j]));
                *reinterpret_cast<bf16x8*>(y + r * y_stride + c * 8) = o;
            }
        }
        return;
    }

    float v[MAXC][8];
    float ss = 0.f;
    bf16x8 wv_pre[MAXC];                                  // requested ahead of the rows: no second trip to memory after the reduction
#pragma unroll
    for (int i = 0; i < MAXC; ++i) wv_pre[i] = *reinterpret_cast<const bf16x8*>(w + min(lane_in_row + i * LPR, nchunk - 1) * 8);
    // every chunk of the row (and of the residual) requested before the first one is used: loads inside the per-chunk
    // condition below would be waited for one chunk after the other (clamped index: past the row's end a lane re-reads
    // the last chunk and drops it).  1024 rows of 1024: 5.1 -> 4.8 us per launch.  (The slabs requested the same way, all
    // chunks of a group of four slabs at once: no change at 256 rows, not kept.)
    bf16x8 x_pre[PARTIALS ? 1 : MAXC], r_pre[ADD ? MAXC : 1];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int cc = min(lane_in_row + i * LPR, nchunk - 1);
        if constexpr (!PARTIALS) x_pre[i] = *reinterpret_cast<const bf16x8*>(x + r * x_stride + cc * 8);
        if constexpr (ADD) r_pre[i] = *reinterpret_cast<const bf16x8*>(residual + r * (int64_t)dim + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane_in_row + i * LPR;
        if (c < nchunk) {
            float xv[8];
            if constexpr (PARTIALS) {
                float acc[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
                }
                const float* src0 = partials + r * (int64_t)dim + c * 8;
                const int64_t slab = rows * (int64_t)dim;
                int p = 0;
                for (; p + 4 <= num_partials; p += 4) {          // 8 independent loads in flight
                    f32x4 a[4], b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab);
                        b[u] = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[u][j] += a[u][j]; acc[u][4 + j] += b[u][j]; }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (p + u < num_partials) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { acc[u][j] += a[j]; acc[u][4 + j] += b[j]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = bf2f(f2bf((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])));   // GEMM output rounds to bf16
            } else {
                const bf16x8 t = x_pre[PARTIALS ? 0 : i];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = bf2f(t[j]);
            }
            if constexpr (ADD) {
                const bf16x8 t = r_pre[ADD ? i : 0];
                bf16x8 nr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[j] = xv[j] + bf2f(t[j]);
                    nr[j] = f2bf(xv[j]);
                }
                if (active) *reinterpret_cast<bf16x8*>(residual + r * (int64_t)dim + c * 8) = nr;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = xv[j];
                ss += xv[j] * xv[j];
            }
        }
    }
    ss = (LPR == 16) ? row16_sum(ss) : wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane_in_row + i * LPR;
        if (c < nchunk && active) {
            const bf16x8 wv = wv_pre[i];
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(v[i][j] * inv)) * bf2f(wv[j]));
            *reinterpret_cast<bf16x8*>(y + r * y_stride + c * 8) = o;
        }
    }
}

// Wide rows (hidden sizes above 1024): one 256-thread workgroup per row, thread t owns the 16-byte chunks
// t, t + 256, ..  The one-wave-per-row kernel above leaves a decode batch of 8 rows of 5120 on 8 waves of the whole
// chip (17 us per call at 3 split-K slabs); here the same call is 4 waves per row with a quarter of the loads each.
// Same arithmetic per element (slab p -> partial sum p mod 4, total (s0 + s1) + (s2 + s3), one bf16 rounding);
// the sum of squares is reduced per wave, then (w0 + w1) + (w2 + w3).
template <int MAXC, bool ADD, bool PARTIALS, bool UPFRONT>
__global__ __launch_bounds__(256) void rmsnorm_wide_kernel(
    const bf16* __restrict__ x, int64_t x_stride, bf16* __restrict__ residual, const bf16* __restrict__ w,
    bf16* __restrict__ y, int64_t y_stride, int64_t rows, int dim, float eps, const float* __restrict__ partials,
    int num_partials) {
    __shared__ float wave_ss[4];
    const int64_t r = blockIdx.x;
    const int nchunk = dim >> 3;
    float v[MAXC][8];
    float ss = 0.f;
    // Everything a thread needs of its row is requested before the first use, with clamped chunk indices (past the row's
    // end a thread re-reads the last chunk and drops it): loads under the per-chunk condition below would be waited for
    // one after the other - with three slabs of 5120-wide rows a chain of about ten trips to memory.  Up to four slabs
    // (each is then one partial sum: 0 + slab, as the loop below computes it) are requested the same way with UPFRONT -
    // decode-sized batches (Qwen3-14B shape, 8 sequences: 6.71 -> 6.57 ms per step); at 1024 rows the registers that takes
    // cost more than the chain (a prefill chunk 27.15 -> 27.3 ms), so large batches and more slabs take the loop.
    bf16x8 wv_pre[MAXC], x_pre[PARTIALS ? 1 : MAXC], r_pre[ADD ? MAXC : 1];
    f32x4 s_lo[PARTIALS && UPFRONT ? MAXC : 1][4], s_hi[PARTIALS && UPFRONT ? MAXC : 1][4];
    const bool slabs_up_front = PARTIALS && UPFRONT && num_partials <= 4;           // uniform
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int cc = min((int)threadIdx.x + i * 256, nchunk - 1);
        wv_pre[i] = *reinterpret_cast<const bf16x8*>(w + cc * 8);
        if constexpr (!PARTIALS) x_pre[i] = *reinterpret_cast<const bf16x8*>(x + r * x_stride + cc * 8);
        if constexpr (ADD) r_pre[i] = *reinterpret_cast<const bf16x8*>(residual + r * (int64_t)dim + cc * 8);
        if constexpr (PARTIALS) {
            if (slabs_up_front) {
                const float* src0 = partials + r * (int64_t)dim + cc * 8;
                const int64_t slab = rows * (int64_t)dim;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* src = src0 + min(u, num_partials - 1) * slab;
                    s_lo[i][u] = *reinterpret_cast<const f32x4*>(src);
                    s_hi[i][u] = *reinterpret_cast<const f32x4*>(src + 4);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nchunk) {
            float xv[8];
            if constexpr (PARTIALS) {
                float acc[4][8];
                if (slabs_up_front) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool on = u < num_partials;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[u][j] = on ? 0.f + s_lo[i][u][j] : 0.f;
                            acc[u][4 + j] = on ? 0.f + s_hi[i][u][j] : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
                    }
                    const float* src0 = partials + r * (int64_t)dim + c * 8;
                    const int64_t slab = rows * (int64_t)dim;
                    for (int p = 0; p < num_partials; p += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (p + u < num_partials) {
                                const f32x4 a = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab);
                                const f32x4 b = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab + 4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) { acc[u][j] += a[j]; acc[u][4 + j] += b[j]; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = bf2f(f2bf((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])));
            } else {
                const bf16x8 t = x_pre[PARTIALS ? 0 : i];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = bf2f(t[j]);
            }
            if constexpr (ADD) {
                const bf16x8 t = r_pre[ADD ? i : 0];
                bf16x8 nr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[j] = xv[j] + bf2f(t[j]);
                    nr[j] = f2bf(xv[j]);
                }
                *reinterpret_cast<bf16x8*>(residual + r * (int64_t)dim + c * 8) = nr;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = xv[j];
                ss += xv[j] * xv[j];
            }
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) wave_ss[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = (wave_ss[0] + wave_ss[1]) + (wave_ss[2] + wave_ss[3]);
    const float inv = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nchunk) {
            const bf16x8 wv = wv_pre[i];
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(v[i][j] * inv)) * bf2f(wv[j]));
            *reinterpret_cast<bf16x8*>(y + r * y_stride + c * 8) = o;
        }
    }
}

template <bool ADD, bool PARTIALS>
static int32_t launch_rmsnorm(const bf16* x, int64_t xs, bf16* res, const bf16* w, bf16* y, int64_t ys,
                              int64_t rows, int64_t dim, float eps, const float* partials, int np,
                              hipStream_t st) {
    if (rows == 0) return NVH_OK;
    if (dim <= 128) {
        rmsnorm_kernel<16, 1, ADD, PARTIALS><<<nvh_cdiv(rows, 16), 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    } else if (dim <= 1024) {
        // decode-sized batches: one wave per workgroup, so a few hundred rows still reach every CU
        const bool no_wide = nvh_switch_on(SW_NORM_NO_WIDE);      // A/B switch for measurements
        if (PARTIALS && rows <= 64 && np >= 8 && !no_wide)
            rmsnorm_kernel<64, 2, ADD, PARTIALS, 256, PARTIALS><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else if (rows <= 512)               // (1024 rows: 2.907 -> 2.892 ms per prefill chunk with four rows per workgroup; 1 / 2 / 8 rows: 2.721 / 2.719 / 2.721 against 2.710 ms)
            rmsnorm_kernel<64, 2, ADD, PARTIALS, 64><<<(unsigned)rows, 64, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else
            rmsnorm_kernel<64, 2, ADD, PARTIALS><<<nvh_cdiv(rows, 4), 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    } else if (dim <= 2048) {
        if (rows <= 256) rmsnorm_wide_kernel<1, ADD, PARTIALS, true><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else rmsnorm_wide_kernel<1, ADD, PARTIALS, false><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    } else if (dim <= 4096) {
        if (rows <= 256) rmsnorm_wide_kernel<2, ADD, PARTIALS, true><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else rmsnorm_wide_kernel<2, ADD, PARTIALS, false><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    } else if (dim <= 6144) {
        if (rows <= 256) rmsnorm_wide_kernel<3, ADD, PARTIALS, true><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else rmsnorm_wide_kernel<3, ADD, PARTIALS, false><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    } else {
        if (rows <= 256) rmsnorm_wide_kernel<4, ADD, PARTIALS, true><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
        else rmsnorm_wide_kernel<4, ADD, PARTIALS, false><<<(unsigned)rows, 256, 0, st>>>(x, xs, res, w, y, ys, rows, (int)dim, eps, partials, np);
    }
    NVH_CHECK_LAUNCH("rmsnorm");
    return NVH_OK;
}

extern "C" int32_t nvh_rmsnorm(const void* x, int64_t x_row_stride, const void* weight, void* y,
                               int64_t y_row_stride, int64_t rows, int64_t dim, float eps, nvh_stream_t stream) {
    NVH_REQUIRE(x && weight && y, NVH_EINVAL, "nvh_rmsnorm: null pointer");
    NVH_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, NVH_EINVAL, "nvh_rmsnorm: dim %ld must be a multiple of 8 and <= 8192", (long)dim);
    NVH_REQUIRE(x_row_stride % 8 == 0 && y_row_stride % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(y) && nvh_aligned16(weight),
                NVH_EINVAL, "nvh_rmsnorm: rows must be 16-byte aligned");
    return launch_rmsnorm<false, false>((const bf16*)x, x_row_stride, nullptr, (const bf16*)weight, (bf16*)y, y_row_stride,
                                        rows, dim, eps, nullptr, 0, (hipStream_t)stream);
}

extern "C" int32_t nvh_add_rmsnorm(const void* x, void* residual, const void* weight, void* y, int64_t rows,
                                   int64_t dim, float eps, const float* partials, int64_t num_partials,
                                   nvh_stream_t stream) {
    NVH_REQUIRE(residual && weight && y && (x || partials), NVH_EINVAL, "nvh_add_rmsnorm: null pointer");
    NVH_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, NVH_EINVAL, "nvh_add_rmsnorm: dim %ld must be a multiple of 8 and <= 8192", (long)dim);
    NVH_REQUIRE(nvh_aligned16(residual) && nvh_aligned16(y) && nvh_aligned16(weight) && (partials || nvh_aligned16(x)) &&
                    (!partials || nvh_aligned16(partials)),
                NVH_EINVAL, "nvh_add_rmsnorm: pointers must be 16-byte aligned");
    if (partials) {
        NVH_REQUIRE(num_partials >= 1 && num_partials <= 64, NVH_EINVAL, "nvh_add_rmsnorm: num_partials %ld out of range", (long)num_partials);
        return launch_rmsnorm<true, true>(nullptr, dim, (bf16*)residual, (const bf16*)weight, (bf16*)y, dim, rows, dim, eps,
                                          partials, (int)num_partials, (hipStream_t)stream);
    }
    return launch_rmsnorm<true, false>((const bf16*)x, dim, (bf16*)residual, (const bf16*)weight, (bf16*)y, dim, rows, dim, eps,
                                       nullptr, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------ RoPE (head_dim 128)
// One 16-lane row per (token, head): lane c owns elements 4c..4c+3 of each half, so the
// rotation pair (i, i+64) sits in one lane.
__device__ __forceinline__ void rope_rotate(float (&a)[4], float (&b)[4], const f32x4& cs, const f32x4& sn) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float y1 = a[j] * cs[j] - b[j] * sn[j];
        const float y2 = b[j] * cs[j] + a[j] * sn[j];
        a[j] = y1;
        b[j] = y2;
    }
}

__global__ __launch_bounds__(256) void rope_kernel(const int64_t* __restrict__ positions, bf16* __restrict__ q,
                                                   int64_t q_stride, int hq, bf16* __restrict__ k, int64_t k_stride,
                                                   int hkv, const float* __restrict__ cos_sin, int64_t num_tokens) {
    const int c = threadIdx.x & 15;
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int heads = hq + hkv;
    if (item >= num_tokens * heads) return;
    const int64_t t = item / heads;
    const int h = (int)(item % heads);
    bf16* base = (h < hq) ? q + t * q_stride + (int64_t)h * 128 : k + t * k_stride + (int64_t)(h - hq) * 128;
    const int64_t pos = positions[t];
    const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_sin + pos * 128 + c * 4);
    const f32x4 sn = *reinterpret_cast<const f32x4*>(cos_sin + pos * 128 + 64 + c * 4);
    bf16x4 lo = *reinterpret_cast<const bf16x4*>(base + c * 4);
    bf16x4 hi = *reinterpret_cast<const bf16x4*>(base + 64 + c * 4);
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = bf2f(lo[j]); b[j] = bf2f(hi[j]); }
    rope_rotate(a, b, cs, sn);
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo[j] = f2bf(a[j]); hi[j] = f2bf(b[j]); }
    *reinterpret_cast<bf16x4*>(base + c * 4) = lo;
    *reinterpret_cast<bf16x4*>(base + 64 + c * 4) = hi;
}

extern "C" int32_t nvh_rope(const int64_t* positions, void* q, int64_t q_token_stride, int64_t num_q_heads, void* k,
                            int64_t k_token_stride, int64_t num_kv_heads, const float* cos_sin, int64_t max_position,
                            int64_t num_tokens, int64_t head_dim, nvh_stream_t stream) {
    NVH_REQUIRE(positions && q && k && cos_sin, NVH_EINVAL, "nvh_rope: null pointer");
    NVH_REQUIRE(head_dim == 128, NVH_ENOSYS, "nvh_rope: head_dim %ld unsupported (128 only)", (long)head_dim);
    NVH_REQUIRE(q_token_stride % 4 == 0 && k_token_stride % 4 == 0 && ((uintptr_t)q & 7) == 0 && ((uintptr_t)k & 7) == 0 && nvh_aligned16(cos_sin),
                NVH_EINVAL, "nvh_rope: q/k must be 8-byte aligned, table 16-byte aligned");
    (void)max_position;
    if (num_tokens == 0) return NVH_OK;
    const int64_t items = num_tokens * (num_q_heads + num_kv_heads);
    rope_kernel<<<nvh_cdiv(items, 16), 256, 0, (hipStream_t)stream>>>(positions, (bf16*)q, q_token_stride, (int)num_q_heads,
                                                                     (bf16*)k, k_token_stride, (int)num_kv_heads, cos_sin, num_tokens);
    NVH_CHECK_LAUNCH("rope");
    return NVH_OK;
}

// ------------------------------------------------------------------ fused QK-norm + RoPE + KV store
// Grid: (tokens, ceil(heads/16)); a 16-lane row per head of the fused qkv row [q heads | k heads | v heads].
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(
    bf16* __restrict__ qkv, int64_t token_stride, int hq, int hkv, const bf16* __restrict__ qw,
    const bf16* __restrict__ kw, float eps, const int64_t* __restrict__ positions,
    const float* __restrict__ cos_sin, bf16* __restrict__ k_cache, bf16* __restrict__ v_cache,
    const int32_t* __restrict__ slot_mapping) {
    const int c = threadIdx.x & 15;
    const int h = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int64_t t = blockIdx.x;
    if (h >= hq + 2 * hkv) return;
    bf16* base = qkv + t * token_stride + (int64_t)h * 128;
    const bool do_store = k_cache != nullptr && slot_mapping != nullptr;
    const int64_t slot = do_store ? (int64_t)slot_mapping[t] : -1;
    if (h >= hq + hkv) {                                 // V head: plain copy to its cache slot
        if (slot >= 0) {
            const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + c * 8);
            *reinterpret_cast<bf16x8*>(v_cache + (slot * hkv + (h - hq - hkv)) * 128 + c * 8) = vv;
        }
        return;
    }
    const bf16* w = (h < hq) ? qw : kw;
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(base + c * 8);
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c * 8);
    const bf16x8 y = qknorm_rope_head(x, wv, cos_sin + positions[t] * 128, c, eps);
    *reinterpret_cast<bf16x8*>(base + c * 8) = y;
    if (h >= hq && slot >= 0) *reinterpret_cast<bf16x8*>(k_cache + (slot * hkv + (h - hq)) * 128 + c * 8) = y;
}

extern "C" int32_t nvh_qknorm_rope_store(void* qkv, int64_t token_stride, int64_t num_q_heads, int64_t num_kv_heads,
                                         const void* q_norm_weight, const void* k_norm_weight, float eps,
                                         const int64_t* positions, const float* cos_sin, int64_t max_position,
                                         void* k_cache, void* v_cache, const int32_t* slot_mapping, int64_t num_tokens,
                                         nvh_stream_t stream) {
    NVH_REQUIRE(qkv && q_norm_weight && k_norm_weight && positions && cos_sin, NVH_EINVAL, "nvh_qknorm_rope_store: null pointer");
    NVH_REQUIRE((k_cache == nullptr) == (v_cache == nullptr), NVH_EINVAL, "nvh_qknorm_rope_store: k_cache and v_cache must both be set or both null");
    NVH_REQUIRE(token_stride % 8 == 0 && nvh_aligned16(qkv) && nvh_aligned16(cos_sin) && (!k_cache || (nvh_aligned16(k_cache) && nvh_aligned16(v_cache))),
                NVH_EINVAL, "nvh_qknorm_rope_store: 16-byte alignment required");
    (void)max_position;
    if (num_tokens == 0) return NVH_OK;
    const int heads = (int)(num_q_heads + 2 * num_kv_heads);
    dim3 grid((unsigned)num_tokens, (unsigned)nvh_cdiv(heads, 16));
    qknorm_rope_store_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((bf16*)qkv, token_stride, (int)num_q_heads, (int)num_kv_heads,
                                                                    (const bf16*)q_norm_weight, (const bf16*)k_norm_weight, eps, positions,
                                                                    cos_sin, (bf16*)k_cache, (bf16*)v_cache, slot_mapping);
    NVH_CHECK_LAUNCH("qknorm_rope_store");
    return NVH_OK;
}

// ------------------------------------------------------------------ KV-cache scatter store
__global__ __launch_bounds__(256) void store_kvcache_kernel(const bf16* __restrict__ key, int64_t ks,
                                                            const bf16* __restrict__ value, int64_t vs,
                                                            bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                            const int32_t* __restrict__ slot_mapping, int row_elems) {
    const int64_t t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;
    const int nchunk = row_elems >> 3;
    for (int i = threadIdx.x; i < 2 * nchunk; i += 256) {
        const bool is_v = i >= nchunk;
        const int c = is_v ? i - nchunk : i;
        const bf16* src = is_v ? value + t * vs : key + t * ks;
        bf16* dst = (is_v ? vc : kc) + slot * row_elems;
        *reinterpret_cast<bf16x8*>(dst + c * 8) = *reinterpret_cast<const bf16x8*>(src + c * 8);
    }
}

extern "C" int32_t nvh_store_kvcache(const void* key, int64_t key_token_stride, const void* value,
                                     int64_t value_token_stride, void* k_cache, void* v_cache,
                                     const int32_t* slot_mapping, int64_t num_tokens, int64_t row_elems,
                                     nvh_stream_t stream) {
    NVH_REQUIRE(key && value && k_cache && v_cache && slot_mapping, NVH_EINVAL, "nvh_store_kvcache: null pointer");
    NVH_REQUIRE(row_elems > 0 && row_elems % 8 == 0 && key_token_stride % 8 == 0 && value_token_stride % 8 == 0, NVH_EINVAL,
                "nvh_store_kvcache: row_elems/strides must be multiples of 8");
    NVH_REQUIRE(nvh_aligned16(key) && nvh_aligned16(value) && nvh_aligned16(k_cache) && nvh_aligned16(v_cache), NVH_EINVAL,
                "nvh_store_kvcache: 16-byte alignment required");
    if (num_tokens == 0) return NVH_OK;
    store_kvcache_kernel<<<(unsigned)num_tokens, 256, 0, (hipStream_t)stream>>>((const bf16*)key, key_token_stride, (const bf16*)value,
                                                                              value_token_stride, (bf16*)k_cache, (bf16*)v_cache,
                                                                              slot_mapping, (int)row_elems);
    NVH_CHECK_LAUNCH("store_kvcache");
    return NVH_OK;
}

// ------------------------------------------------------------------ SiLU * mul
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                       int64_t rows, int inter) {
    const int cpr = inter >> 3;                          // chunks per output row
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i % cpr);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(x + r * 2 * inter + c * 8);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(x + r * 2 * inter + inter + c * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = bf2f(g[j]);
            const float s = gf / (1.0f + expf(-gf));
            o[j] = f2bf(bf2f(f2bf(s)) * bf2f(u[j]));
        }
        *reinterpret_cast<bf16x8*>(y + r * inter + c * 8) = o;
    }
}

extern "C" int32_t nvh_silu_mul(const void* x, void* y, int64_t rows, int64_t inter, nvh_stream_t stream) {
    NVH_REQUIRE(x && y, NVH_EINVAL, "nvh_silu_mul: null pointer");
    NVH_REQUIRE(inter > 0 && inter % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(y), NVH_EINVAL, "nvh_silu_mul: inter %% 8 and 16-byte alignment required");
    if (rows == 0) return NVH_OK;
    const int64_t total = rows * (inter / 8);
    const unsigned grid = (unsigned)(nvh_cdiv(total, 256) < 4096 ? nvh_cdiv(total, 256) : 4096);
    silu_mul_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16*)x, (bf16*)y, rows, (int)inter);
    NVH_CHECK_LAUNCH("silu_mul");
    return NVH_OK;
}

// ------------------------------------------------------------------ embedding / row gathers
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ table,
                                                        bf16* __restrict__ out, int dim, int64_t vstart, int64_t vrows) {
    const int64_t t = blockIdx.x;
    const int64_t id = ids[t] - vstart;
    const bool hit = id >= 0 && id < vrows;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < (dim >> 3); c += 256) {
        bf16x8 v = zero;
        if (hit) v = *reinterpret_cast<const bf16x8*>(table + id * dim + c * 8);
        *reinterpret_cast<bf16x8*>(out + t * dim + c * 8) = v;
    }
}

extern "C" int32_t nvh_embedding(const int64_t* ids, const void* table, void* out, int64_t num_tokens, int64_t dim,
                                 int64_t vocab_start, int64_t vocab_rows, nvh_stream_t stream) {
    NVH_REQUIRE(ids && table && out, NVH_EINVAL, "nvh_embedding: null pointer");
    NVH_REQUIRE(dim > 0 && dim % 8 == 0 && nvh_aligned16(table) && nvh_aligned16(out), NVH_EINVAL, "nvh_embedding: dim %% 8 and 16-byte alignment required");
    if (num_tokens == 0) return NVH_OK;
    embedding_kernel<<<(unsigned)num_tokens, 256, 0, (hipStream_t)stream>>>(ids, (const bf16*)table, (bf16*)out, (int)dim, vocab_start, vocab_rows);
    NVH_CHECK_LAUNCH("embedding");
    return NVH_OK;
}

__global__ __launch_bounds__(256) void gather_last_rows_kernel(const bf16* __restrict__ hidden, const int32_t* __restrict__ cu_q,
                                                               const int32_t* __restrict__ need, bf16* __restrict__ out, int dim) {
    const int64_t i = blockIdx.x;
    const int j = need ? need[i] : (int)i;
    const int64_t src = (int64_t)cu_q[j + 1] - 1;
    for (int c = threadIdx.x; c < (dim >> 3); c += 256)
        *reinterpret_cast<bf16x8*>(out + i * dim + c * 8) = *reinterpret_cast<const bf16x8*>(hidden + src * dim + c * 8);
}

extern "C" int32_t nvh_gather_last_rows(const void* hidden, const int32_t* cu_seqlens_q, const int32_t* need, void* out,
                                        int64_t num_rows, int64_t dim, nvh_stream_t stream) {
    NVH_REQUIRE(hidden && cu_seqlens_q && out, NVH_EINVAL, "nvh_gather_last_rows: null pointer");
    NVH_REQUIRE(dim > 0 && dim % 8 == 0 && nvh_aligned16(hidden) && nvh_aligned16(out), NVH_EINVAL, "nvh_gather_last_rows: dim %% 8 and 16-byte alignment required");
    if (num_rows == 0) return NVH_OK;
    gather_last_rows_kernel<<<(unsigned)num_rows, 256, 0, (hipStream_t)stream>>>((const bf16*)hidden, cu_seqlens_q, need, (bf16*)out, (int)dim);
    NVH_CHECK_LAUNCH("gather_last_rows");
    return NVH_OK;
}

// input_ids[dst[i]] = sampled[src[i]]: the token ids a step takes from the previous step's sampler output while the
// host has not read them back yet.
__global__ __launch_bounds__(256) void feed_token_ids_kernel(int64_t* __restrict__ input_ids, const int32_t* __restrict__ dst,
                                                             const int64_t* __restrict__ sampled, const int32_t* __restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) input_ids[dst[i]] = sampled[src[i]];
}

extern "C" int32_t nvh_feed_token_ids(int64_t* input_ids, const int32_t* dst_index, const int64_t* sampled, const int32_t* src_index,
                                      int64_t count, nvh_stream_t stream) {
    NVH_REQUIRE(count >= 0 && count < (1 << 30), NVH_EINVAL, "nvh_feed_token_ids: count %ld out of range", (long)count);
    if (count == 0) return NVH_OK;
    NVH_REQUIRE(input_ids && dst_index && sampled && src_index, NVH_EINVAL, "nvh_feed_token_ids: null pointer");
    feed_token_ids_kernel<<<(unsigned)nvh_cdiv(count, 256), 256, 0, (hipStream_t)stream>>>(input_ids, dst_index, sampled, src_index, (int)count);
    NVH_CHECK_LAUNCH("feed_token_ids");
    return NVH_OK;
}

// Token ids as three bf16 bytes each (0..255: exact), so that "rank 0's sampled ids, zeros elsewhere" can ride the same
// sum over the ranks as the activations and every rank ends up with the ids: enc[i], enc[n + i], enc[2n + i] = byte 0, 1, 2
// of ids[i]; ids == nullptr writes zeros.  Entries n*3 .. padded are zeroed (the exchange wants a multiple of 8).
__global__ __launch_bounds__(256) void ids_to_bytes_kernel(const int64_t* __restrict__ ids, bf16* __restrict__ enc, int n, int padded) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= padded) return;
    float v = 0.f;
    if (ids != nullptr && i < 3 * n) v = (float)((ids[i % n] >> (8 * (i / n))) & 255);
    enc[i] = f2bf(v);
}
__global__ __launch_bounds__(256) void bytes_to_ids_kernel(const bf16* __restrict__ enc, int64_t* __restrict__ ids, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] = (int64_t)bf2f(enc[i]) + ((int64_t)bf2f(enc[n + i]) << 8) + ((int64_t)bf2f(enc[2 * n + i]) << 16);
}

extern "C" int32_t nvh_ids_to_bytes_bf16(const int64_t* ids, void* enc, int64_t count, int64_t padded_count, nvh_stream_t stream) {
    NVH_REQUIRE(enc && count >= 0 && padded_count >= 3 * count && padded_count < (1 << 30), NVH_EINVAL, "nvh_ids_to_bytes_bf16: bad arguments");
    if (padded_count == 0) return NVH_OK;
    ids_to_bytes_kernel<<<(unsigned)nvh_cdiv(padded_count, 256), 256, 0, (hipStream_t)stream>>>(ids, (bf16*)enc, (int)count, (int)padded_count);
    NVH_CHECK_LAUNCH("ids_to_bytes");
    return NVH_OK;
}

extern "C" int32_t nvh_bytes_bf16_to_ids(const void* enc, int64_t* ids, int64_t count, nvh_stream_t stream) {
    NVH_REQUIRE(enc && ids && count >= 0 && count < (1 << 28), NVH_EINVAL, "nvh_bytes_bf16_to_ids: bad arguments");
    if (count == 0) return NVH_OK;
    bytes_to_ids_kernel<<<(unsigned)nvh_cdiv(count, 256), 256, 0, (hipStream_t)stream>>>((const bf16*)enc, ids, (int)count);
    NVH_CHECK_LAUNCH("bytes_to_ids");
    return NVH_OK;
}
