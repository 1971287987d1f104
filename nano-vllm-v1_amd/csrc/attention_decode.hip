// Paged decode attention (one query per sequence): HBM-bound streaming of the KV pages.
//
// Grid (num_splits, Hkv, B); 256 threads.  A workgroup owns one (sequence, kv head, key
// range) and serves all G = Hq/Hkv query heads of that kv head, so every K/V byte is read
// from HBM once.  Lane layout: 16 consecutive lanes ("row") cover one token's 256-byte K
// (or V) row of this head with one 16-byte load each; a wave holds 4 rows, the workgroup 16,
// and each loop trip consumes 64 consecutive tokens (always inside one 256-token page, so the
// block-table entry is a single scalar load per trip).  Every row keeps its own online-softmax
// state (max, sum, 8 output columns per lane); rows are merged once at the end (shuffles inside
// the wave, LDS across waves).  Key ranges: the grid offers num_splits ranges per sequence, a sequence of ctx keys
// uses ranges of max(ceil(ctx / num_splits) rounded up to 64, min_range) keys, i.e. only as many as its length
// calls for (workgroups of unused ranges leave at once).  A sequence that fits ONE range is finished by its
// workgroup; longer ones leave (max, sum, unnormalised fp32 output) per range in a workspace and the merge kernel
// finishes them.  So one launch serves a batch of short sequences with a few very long ones without making the
// short ones pay for partial results, and without the long ones walking their keys in a single workgroup.
#include "common.h"
#include <stdlib.h>

#define LOG2E 1.4426950408889634f

// Keys per range of a sequence of `ctx` keys when the launch offers `num_splits` ranges (see the file header).
__device__ __forceinline__ int decode_range_keys(int ctx, int num_splits, int min_range) {
    const int even = (((ctx + num_splits - 1) / num_splits) + 63) & ~63;
    return even > min_range ? even : min_range;
}

// Explicit work list (instead of the uniform num_splits x batch grid): item i of gridDim.z is range item_range[i] of
// sequence item_seq[i] (-1: unused slot, the workgroup leaves); sequence b has seq_active[b] ranges of seq_chunk[b]
// keys.  The caller lists one item per sequence plus extra items for the ranges 1.. of its long sequences, so a
// batch of hundreds of short sequences and a few very long ones costs neither a serial walk over the long ones
// nor a grid of empty workgroups (measured: ~10 ns per empty workgroup, 32768 of them per layer for 16 uniform
// ranges at batch 256).  Partial results use max_ranges as the slab stride.
struct DecodeItems {
    const int32_t *item_seq, *item_range, *seq_active, *seq_chunk;
    int batch;
    int max_ranges;
};

// FUSED: `q` is the raw fused qkv row of the step's single new token per sequence.  The kernel applies the
// q/k epilogue itself (per-head RMSNorm + RoPE, qknorm_rope_head), uses the new token's K/V straight from
// registers for its own dot product and writes them to their cache slot (token ctx-1 of the sequence), so
// the separate norm/rope/store launch disappears from a decode step.
struct FusedQkv {
    const bf16* q_norm_w;
    const bf16* k_norm_w;
    const int64_t* positions;
    const float* cos_sin;
    float eps;
};

__device__ __forceinline__ void merge_row_wave(const float* __restrict__ ws_acc, const float* __restrict__ ws_ml, bf16* __restrict__ out,
                                               int num_splits, int64_t bh, int active, int lane);

template <int G, bool FUSED>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const bf16* __restrict__ q, int64_t q_stride, bf16* __restrict__ kc, bf16* __restrict__ vc,
    const int32_t* __restrict__ ctx_lens, const int32_t* __restrict__ tables, int64_t table_stride, int hkv,
    float scale_log2, bf16* __restrict__ out, float* __restrict__ ws_acc, float* __restrict__ ws_ml, int min_range, DecodeItems items, FusedQkv fz,
    unsigned* __restrict__ arrivals) {
    __shared__ float sm_acc[4][G][128];
    __shared__ float sm_ml[4][G][2];
    __shared__ int sm_last;

    const int kvh = blockIdx.y;
    const int hq = hkv * G;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15;
    const int rowid = w * 4 + (lane >> 4);

    int split, b, num_splits, ctx, chunk, active;
    if (items.item_seq) {                                    // work list
        b = items.item_seq[blockIdx.z];
        if (b < 0 || b >= items.batch) return;               // block-uniform: unused slot
        split = items.item_range[blockIdx.z];
        num_splits = items.max_ranges;
        ctx = ctx_lens[b];
        chunk = max(items.seq_chunk[b] & ~63, 64);           // whole 64-key trips: a malformed list may drop keys, never leave a page
        active = min(items.seq_active[b], num_splits);
        if (split < 0) return;
    } else {                                                 // uniform grid
        // (workgroup id = kvh + 8 * b reaches XCD id % 8: with 8 kv heads every XCD streams ONE head of all sequences.  The
        // transposed order - the 8 heads of a sequence side by side on one XCD, whole 2-KB token rows per XCD - measured
        // 6.49 ms against 6.29 ms per step of 256 sequences: not taken)
        split = blockIdx.x;
        b = blockIdx.z;
        num_splits = gridDim.x;
        ctx = ctx_lens[b];
        chunk = decode_range_keys(ctx, num_splits, min_range);
        active = ctx > 0 ? (ctx + chunk - 1) / chunk : 0;
    }
    if (split >= max(active, 1)) return;                     // block-uniform: this range holds no key of the sequence
    const bool final_here = active <= 1;                     // the whole sequence (or a padded row) is this workgroup's
    const int start = split * chunk;
    const int end = min(ctx, start + chunk);

    bf16x8 qv[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
        qv[g] = *reinterpret_cast<const bf16x8*>(q + (int64_t)b * q_stride + (int64_t)(kvh * G + g) * 128 + c * 8);
    bf16x8 k_new = {0, 0, 0, 0, 0, 0, 0, 0}, v_new = k_new;
    if constexpr (FUSED) {
        if (ctx > 0) {                                       // block-uniform
            const float* cs_row = fz.cos_sin + fz.positions[b] * 128;
            const bf16x8 qw = *reinterpret_cast<const bf16x8*>(fz.q_norm_w + c * 8);
            const bf16x8 kw = *reinterpret_cast<const bf16x8*>(fz.k_norm_w + c * 8);
#pragma unroll
            for (int g = 0; g < G; ++g) qv[g] = qknorm_rope_head(qv[g], qw, cs_row, c, fz.eps);
            const bf16* row = q + (int64_t)b * q_stride;
            k_new = qknorm_rope_head(*reinterpret_cast<const bf16x8*>(row + (int64_t)(hq + kvh) * 128 + c * 8), kw, cs_row, c, fz.eps);
            v_new = *reinterpret_cast<const bf16x8*>(row + (int64_t)(hq + hkv + kvh) * 128 + c * 8);
        }
    }

    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }

    const int32_t* table = tables + (int64_t)b * table_stride;
    for (int base = start; base < end; base += 64) {
        const int64_t slot0 = (int64_t)table[base >> 8] * NVH_KV_PAGE + (base & 255);
        bf16x8 kf[4], vf[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int toff = u * 16 + rowid;
            valid[u] = base + toff < end;
            const int64_t off = ((slot0 + (valid[u] ? toff : 0)) * hkv + kvh) * 128 + c * 8;
            kf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kc + off));   // streamed once
            vf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(vc + off));
            if constexpr (FUSED) {
                if (base + toff == ctx - 1) {                // the new token: not in the cache yet
                    kf[u] = k_new;
                    vf[u] = v_new;
                    *reinterpret_cast<bf16x8*>(kc + off) = k_new;
                    *reinterpret_cast<bf16x8*>(vc + off) = v_new;
                }
            }
        }
        float s[G][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x2 a = {qv[g][2 * j], qv[g][2 * j + 1]};
                    bf16x2 k2 = {kf[u][2 * j], kf[u][2 * j + 1]};
                    d = __builtin_amdgcn_fdot2_f32_bf16(a, k2, d, false);
                }
                d = row16_sum(d) * scale_log2;
                s[g][u] = valid[u] ? d : -INFINITY;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mn = fmaxf(fmaxf(fmaxf(s[g][0], s[g][1]), fmaxf(s[g][2], s[g][3])), m[g]);
            const float mu = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = __builtin_amdgcn_exp2f(m[g] - mu);
            float p[4];
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                p[u] = __builtin_amdgcn_exp2f(s[g][u] - mu);
                ps += p[u];
            }
            l[g] = l[g] * alpha + ps;
            m[g] = mn;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = acc[g][j] * alpha;
#pragma unroll
                for (int u = 0; u < 4; ++u) a += p[u] * bf2f(vf[u][j]);
                acc[g][j] = a;
            }
        }
    }

    // merge the 4 rows of the wave (lanes l, l^16, l^32 hold the same output columns)
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float mo = __shfl_xor(m[g], step, 64);
            const float lo = __shfl_xor(l[g], step, 64);
            const float mn = fmaxf(m[g], mo);
            const float mu = (mn == -INFINITY) ? 0.f : mn;
            const float a = __builtin_amdgcn_exp2f(m[g] - mu), bsc = __builtin_amdgcn_exp2f(mo - mu);
            l[g] = l[g] * a + lo * bsc;
            m[g] = mn;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float ao = __shfl_xor(acc[g][j], step, 64);
                acc[g][j] = acc[g][j] * a + ao * bsc;
            }
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sm_acc[w][g][c * 8 + j] = acc[g][j];
            if (c == 0) {
                sm_ml[w][g][0] = m[g];
                sm_ml[w][g][1] = l[g];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < G * 128; idx += 256) {
        const int g = idx >> 7, d = idx & 127;
        float M = fmaxf(fmaxf(sm_ml[0][g][0], sm_ml[1][g][0]), fmaxf(sm_ml[2][g][0], sm_ml[3][g][0]));
        const float mu = (M == -INFINITY) ? 0.f : M;
        float L = 0.f, A = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float sc = __builtin_amdgcn_exp2f(sm_ml[ww][g][0] - mu);
            L += sm_ml[ww][g][1] * sc;
            A += sm_acc[ww][g][d] * sc;
        }
        const int h = kvh * G + g;
        if (final_here) {
            out[((int64_t)b * hq + h) * 128 + d] = f2bf(L > 0.f ? A / L : 0.f);
        } else {
            // write-through (agent-scope) stores: the partial results need no release fence before the arrival count
            const int64_t slab = ((int64_t)b * hq + h) * num_splits + split;
            __hip_atomic_store(ws_acc + slab * 128 + d, A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(ws_ml + slab * 2, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ws_ml + slab * 2 + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (final_here || arrivals == nullptr) return;           // block-uniform
    // The workgroup that finishes last among the `active` ranges of this (sequence, kv head) merges them - no second
    // launch.  Every storing wave drains its stores, one lane counts the arrival (agent-scope atomic), the last arriver
    // invalidates its CU's L1 (agent-scope acquire) before any wave of it reads the other workgroups' records, and puts
    // the counter back to zero for the next launch.  Placement- and order-independent.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = arrivals + (int64_t)b * hkv + kvh;
        const unsigned before = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = before + 1u == (unsigned)active;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sm_last = last;
    }
    __syncthreads();
    if (!sm_last) return;
    for (int g = w; g < G; g += 4) merge_row_wave(ws_acc, ws_ml, out, num_splits, (int64_t)b * hq + kvh * G + g, active, lane);
}

// Merge per-range partials of one output row (sequence/token x query head).  Both kernels below compute the same
// bits: range s goes to partial accumulator s mod 4 in increasing s, the row is ((p0 + p1) + (p2 + p3)) / L with
// L summed by the same wave reduction; ranges that saw no key (sum 0) weigh 0.
//  * wave-per-row (4 rows per workgroup, no LDS): many rows (prefill with key ranges, large decode batches);
//  * workgroup-per-row (wave w owns accumulator w): few rows, where the chain of dependent loads of one wave
//    would be the whole cost.  With `ctx_lens` it only visits the ranges the decode kernel gave keys to.
// Ranges of row bh that hold partial results.  Decode rows (ctx_lens given) follow decode_range_keys; a row whose
// sequence fits one range was finished by the attention kernel itself and reports -1 (nothing to merge).
__device__ __forceinline__ int active_ranges(const int32_t* ctx_lens, int64_t bh, int hq, int num_splits, int min_range,
                                             const int32_t* seq_active) {
    if (seq_active) {
        const int a = min(seq_active[bh / hq], num_splits);
        return a > 1 ? a : -1;
    }
    if (!ctx_lens) return num_splits;
    const int ctx = ctx_lens[bh / hq];
    if (ctx <= 0) return -1;
    const int chunk = decode_range_keys(ctx, num_splits, min_range);
    const int active = (ctx + chunk - 1) / chunk;
    return active > 1 ? active : -1;
}

// One wave merges the `active` partial results of row bh (one query head of one sequence).
__device__ __forceinline__ void merge_row_wave(const float* __restrict__ ws_acc, const float* __restrict__ ws_ml, bf16* __restrict__ out,
                                               int num_splits, int64_t bh, int active, int lane) {
    const f32x2* ml = reinterpret_cast<const f32x2*>(ws_ml) + bh * num_splits;
    float m_loc = -INFINITY;
    for (int s = lane; s < active; s += 64) {
        const f32x2 v = ml[s];
        if (v[1] > 0.f) m_loc = fmaxf(m_loc, v[0]);
    }
    const float M = wave_max(m_loc);
    const float mu = (M == -INFINITY) ? 0.f : M;
    float L = 0.f;
    f32x2 A[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const f32x2* acc = reinterpret_cast<const f32x2*>(ws_acc + bh * num_splits * 128) + lane;
    for (int s0 = 0; s0 < active; s0 += 64) {
        const int s = s0 + lane;
        float wgt = 0.f, l = 0.f;
        if (s < active) {
            const f32x2 v = ml[s];
            l = v[1];
            wgt = l > 0.f ? __builtin_amdgcn_exp2f(v[0] - mu) : 0.f;
        }
        L += l * wgt;
        const int n = min(64, active - s0);
        for (int j0 = 0; j0 < n; j0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wj = __shfl(wgt, (j0 + u) & 63, 64);
                if (j0 + u < n && wj != 0.f) {                     // wave-uniform
                    const f32x2 v = acc[(int64_t)(s0 + j0 + u) * 64];
                    A[u][0] += v[0] * wj;
                    A[u][1] += v[1] * wj;
                }
            }
        }
    }
    L = wave_sum(L);
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    const float a0 = (A[0][0] + A[1][0]) + (A[2][0] + A[3][0]), a1 = (A[0][1] + A[1][1]) + (A[2][1] + A[3][1]);
    bf16x2 o = {f2bf(a0 * inv), f2bf(a1 * inv)};
    *reinterpret_cast<bf16x2*>(out + bh * 128 + lane * 2) = o;
}

__global__ __launch_bounds__(256) void attn_decode_combine_kernel(const float* __restrict__ ws_acc,
                                                                  const float* __restrict__ ws_ml,
                                                                  bf16* __restrict__ out, int num_splits, int64_t rows,
                                                                  const int32_t* __restrict__ ctx_lens, int hq, int min_range,
                                                                  const int32_t* __restrict__ seq_active) {
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= rows) return;
    const int active = active_ranges(ctx_lens, bh, hq, num_splits, min_range, seq_active);
    if (active < 0) return;
    merge_row_wave(ws_acc, ws_ml, out, num_splits, bh, active, lane);
}

// Work-list launches: one workgroup per sequence, one wave per query head (16 at a time).  Nearly every sequence of
// such a batch has a single range and nothing to merge, so its workgroup leaves at once - a quarter as many
// workgroups to dispatch as with four rows each.
__global__ __launch_bounds__(1024) void attn_decode_combine_seq_kernel(const float* __restrict__ ws_acc,
                                                                       const float* __restrict__ ws_ml,
                                                                       bf16* __restrict__ out, int num_splits, int hq,
                                                                       const int32_t* __restrict__ seq_active) {
    const int active = min(seq_active[blockIdx.x], num_splits);
    if (active <= 1) return;                                       // block-uniform
    const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    for (int h = threadIdx.x >> 6; h < hq; h += waves)
        merge_row_wave(ws_acc, ws_ml, out, num_splits, (int64_t)blockIdx.x * hq + h, active, lane);
}

__global__ __launch_bounds__(256) void attn_decode_combine_row_kernel(const float* __restrict__ ws_acc,
                                                                      const float* __restrict__ ws_ml,
                                                                      bf16* __restrict__ out, int num_splits,
                                                                      const int32_t* __restrict__ ctx_lens, int hq, int min_range,
                                                                  const int32_t* __restrict__ seq_active) {
    __shared__ f32x2 part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t bh = blockIdx.x;
    const int active = active_ranges(ctx_lens, bh, hq, num_splits, min_range, seq_active);
    if (active < 0) return;                                        // block-uniform
    const f32x2* ml = reinterpret_cast<const f32x2*>(ws_ml) + bh * num_splits;
    const f32x2* acc = reinterpret_cast<const f32x2*>(ws_acc + bh * num_splits * 128) + lane;
    float L = 0.f;
    f32x2 A = {0.f, 0.f};
    if (active <= 32) {                                            // block-uniform
        // Up to 32 ranges (what the decode attention offers a short batch): this wave's accumulator records (ranges w,
        // w + 4, ..) are requested together with the (max, sum) records instead of after the weights are known - one trip
        // to memory less.  All eight loads are unconditional (a range past the last repeats the last: a load under a
        // runtime condition would be waited for on its own); every range below `active` was written by this launch's
        // attention kernel.  Same sums in the same order as the loop below.
        f32x2 pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = acc[(int64_t)min(w + 4 * k, active - 1) * 64];
        const f32x2 got = ml[min(lane, active - 1)];
        const f32x2 rec0 = lane < active ? got : f32x2{-INFINITY, 0.f};
        const float M = wave_max(rec0[1] > 0.f ? rec0[0] : -INFINITY);
        const float mu = (M == -INFINITY) ? 0.f : M;
        const float wgt = rec0[1] > 0.f ? __builtin_amdgcn_exp2f(rec0[0] - mu) : 0.f;
        L += rec0[1] * wgt;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = w + 4 * k;
            const float wj = j < active ? __shfl(wgt, j & 63, 64) : 0.f;
            if (wj != 0.f) {                                       // wave-uniform
                A[0] += pv[k][0] * wj;
                A[1] += pv[k][1] * wj;
            }
        }
    } else {
    // every wave derives the row maximum and the weights itself (<= 256 ranges: 4 records per lane)
    f32x2 rec[4];
    float m_loc = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = i * 64 + lane;
        rec[i] = f32x2{-INFINITY, 0.f};
        if (s < active) rec[i] = ml[s];
        if (rec[i][1] > 0.f) m_loc = fmaxf(m_loc, rec[i][0]);
    }
    const float M = wave_max(m_loc);
    const float mu = (M == -INFINITY) ? 0.f : M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i * 64 >= active) break;                               // block-uniform
        const float wgt = rec[i][1] > 0.f ? __builtin_amdgcn_exp2f(rec[i][0] - mu) : 0.f;
        L += rec[i][1] * wgt;
        const int n = min(64, active - i * 64);
        for (int j0 = w; j0 < n; j0 += 16) {                       // ranges w, w+4, .. of this block of 64
            f32x2 v[4];
            float wj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
                wj[u] = j < n ? __shfl(wgt, j & 63, 64) : 0.f;
                v[u] = f32x2{0.f, 0.f};
                if (wj[u] != 0.f) v[u] = acc[(int64_t)(i * 64 + j) * 64];      // wave-uniform; loads go out together
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (wj[u] != 0.f) {
                    A[0] += v[u][0] * wj[u];
                    A[1] += v[u][1] * wj[u];
                }
            }
        }
    }
    }
    part[w][lane] = A;
    __syncthreads();
    if (w == 0) {
        L = wave_sum(L);
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        const f32x2 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
        const float a0 = (p0[0] + p1[0]) + (p2[0] + p3[0]), a1 = (p0[1] + p1[1]) + (p2[1] + p3[1]);
        bf16x2 o = {f2bf(a0 * inv), f2bf(a1 * inv)};
        *reinterpret_cast<bf16x2*>(out + bh * 128 + lane * 2) = o;
    }
}

// Arrival counters of the in-kernel merge: one word per (sequence, kv head), zero when the code object is loaded and
// put back to zero by every merge, so no launch has to clear them.  One copy per GPU; launches that overlap on a GPU
// (two streams) would share them and must not.
#define NVH_DECODE_ARRIVALS 65536
static __device__ __align__(256) unsigned decode_arrivals[NVH_DECODE_ARRIVALS];
namespace {
unsigned* g_arrivals_ptr[64] = {};
}
int32_t nvh_decode_arrivals(unsigned** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    NVH_REQUIRE(e == hipSuccess && dev >= 0 && dev < 64, NVH_ELAUNCH, "nvh_attn_decode: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    if (g_arrivals_ptr[dev] == nullptr) {
        unsigned* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(decode_arrivals));
        NVH_REQUIRE(e == hipSuccess && p != nullptr, NVH_ELAUNCH, "nvh_attn_decode: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
        g_arrivals_ptr[dev] = p;
    }
    if (out) *out = g_arrivals_ptr[dev];
    return NVH_OK;
}

int32_t nvh_launch_attn_combine(const float* ws_acc, const float* ws_ml, void* out, int64_t rows, int num_splits,
                                const int32_t* ctx_lens, int hq, int min_range, const int32_t* seq_active, hipStream_t st) {
    if (seq_active != nullptr && rows / hq > 16)
        attn_decode_combine_seq_kernel<<<(unsigned)(rows / hq), 64 * (hq < 16 ? hq : 16), 0, st>>>(ws_acc, ws_ml, (bf16*)out, num_splits, hq, seq_active);
    else if (rows <= 2048 && num_splits <= 256)
        attn_decode_combine_row_kernel<<<(unsigned)rows, 256, 0, st>>>(ws_acc, ws_ml, (bf16*)out, num_splits, ctx_lens, hq, min_range, seq_active);
    else
        attn_decode_combine_kernel<<<(unsigned)nvh_cdiv(rows, 4), 256, 0, st>>>(ws_acc, ws_ml, (bf16*)out, num_splits, rows, ctx_lens, hq, min_range,
                                                                                seq_active);
    NVH_CHECK_LAUNCH("attn_combine");
    return NVH_OK;
}

extern "C" int64_t nvh_attn_decode_workspace_bytes(int64_t batch, int64_t num_q_heads, int64_t head_dim, int32_t num_splits) {
    if (num_splits <= 1) return 0;
    return batch * num_q_heads * num_splits * (head_dim + 2) * (int64_t)sizeof(float);
}

extern "C" int32_t nvh_attn_decode_suggest_splits(int64_t batch, int64_t num_kv_heads, int64_t max_context) {
    // Aim for >= 512 workgroups (2 per CU) while keeping >= 128 keys per range at the longest context, at most 32 ranges,
    // and none once every CU has a workgroup anyway.  Measured on Qwen3-0.6B decode steps at ~1040 keys (ms per step by
    // the workgroup count aimed for, 256 / 512 / 1024 / 2048): 1 sequence 0.90 / 0.92 (64 ranges offered; 0.90 with 32)
    // / - / -, 8 sequences 1.13 / 1.15 / 1.17 / 1.31, 16 sequences 1.43 / 1.39 / 1.50 / 1.60, 32 sequences 1.70 (one
    // range, no merge launch) / 1.73 / 1.81 / 1.93: every range costs a workgroup's prologue and a record for the merge,
    // every range less a longer chain of dependent trips.  Ranges for batches that already give every CU a workgroup
    // (tools/decode_splits_ab.sh; one range / 1024 / 2048 workgroups aimed for): 32 sequences 1.67 / 1.77 / 1.89 ms, 64
    // sequences 2.39 / 2.60 / 2.67, 128 sequences 3.95 / 3.95 / 4.22.
    const int64_t target = nvh_switch_or(SW_DECODE_TARGET_WGS, 512);   // tuning switch
    const int64_t wgs = batch * num_kv_heads > 0 ? batch * num_kv_heads : 1;
    const int64_t full = nvh_switch_or(SW_DECODE_FULL_WGS, 256);        // tuning switch
    int64_t want = wgs >= full ? 1 : nvh_cdiv(target, wgs);
    const int64_t cap = max_context / 128 > 1 ? max_context / 128 : 1;
    if (want > cap) want = cap;
    if (want > 32) want = 32;
    return (int32_t)(want < 1 ? 1 : want);
}

namespace {
struct DecodeArgs {
    const void* q;
    int64_t q_token_stride;
    void *k_cache, *v_cache;
    const int32_t *context_lens, *block_tables;
    int64_t table_stride, batch, num_q_heads, num_kv_heads, head_dim;
    float softmax_scale;
    void* out;
    int32_t num_splits, min_range;
    void* workspace;
    hipStream_t st;
    DecodeItems items;              // item_seq == nullptr: uniform grid
    int64_t num_items;
};

template <bool FUSED>
int32_t launch_decode(const char* name, const DecodeArgs& a, const FusedQkv& fz) {
    NVH_REQUIRE(a.q && a.k_cache && a.v_cache && a.context_lens && a.block_tables && a.out, NVH_EINVAL, "%s: null pointer", name);
    NVH_REQUIRE(a.head_dim == 128, NVH_ENOSYS, "%s: head_dim %ld unsupported (128 only)", name, (long)a.head_dim);
    NVH_REQUIRE(a.num_kv_heads > 0 && a.num_q_heads % a.num_kv_heads == 0, NVH_EINVAL, "%s: Hq %% Hkv != 0", name);
    NVH_REQUIRE(a.q_token_stride % 8 == 0 && nvh_aligned16(a.q) && nvh_aligned16(a.k_cache) && nvh_aligned16(a.v_cache), NVH_EINVAL,
                "%s: 16-byte alignment required", name);
    NVH_REQUIRE(a.num_splits >= 1 && a.num_splits <= 256 && (a.num_splits == 1 || a.workspace), NVH_EINVAL,
                "%s: num_splits %d needs a workspace", name, a.num_splits);
    NVH_REQUIRE(a.batch <= 65535 && a.num_kv_heads <= 65535, NVH_EINVAL, "%s: batch/heads exceed grid limits", name);
    NVH_REQUIRE(a.min_range >= 0 && a.min_range % 64 == 0, NVH_EINVAL, "%s: min_range_keys %d must be a multiple of 64 (0 = 64)", name, a.min_range);
    const int min_range = a.min_range > 0 ? a.min_range : 64;
    if (a.batch == 0) return NVH_OK;
    const int g = (int)(a.num_q_heads / a.num_kv_heads);
    float* ws_acc = (float*)a.workspace;
    float* ws_ml = ws_acc ? ws_acc + a.batch * a.num_q_heads * a.num_splits * 128 : nullptr;
    dim3 grid((unsigned)a.num_splits, (unsigned)a.num_kv_heads, (unsigned)a.batch);
    if (a.items.item_seq) {
        NVH_REQUIRE(a.items.item_range && a.items.seq_active && a.items.seq_chunk && a.num_items >= a.batch && a.num_items <= 65535 &&
                        a.items.max_ranges == a.num_splits,
                    NVH_EINVAL, "%s: work list needs all four arrays, batch <= items <= 65535 and max_ranges == num_splits", name);
        grid = dim3(1, (unsigned)a.num_kv_heads, (unsigned)a.num_items);
    }
    const float sl2 = a.softmax_scale * LOG2E;
    // NVH_DECODE_MERGE_IN_LAUNCH=1: key ranges are merged by the last workgroup to arrive instead of by a second launch
    // (same bits).  Off by default: measured on Qwen3-0.6B decode steps it loses at 1 sequence (1.286 ms against 1.170 ms
    // per step: the drain, the arrival atomic, the acquire and a merge by one workgroup cost more than the 4.7 us launch
    // they replace) and gains under 1 % at 32 sequences (1.909 against 1.926 ms).
    unsigned* arrivals = nullptr;
    if (a.num_splits > 1 && a.batch * a.num_kv_heads <= NVH_DECODE_ARRIVALS && nvh_switch_on(SW_DECODE_MERGE_IN_LAUNCH)) {
        const int32_t rc = nvh_decode_arrivals(&arrivals);
        if (rc != NVH_OK) return rc;
    }
#define LAUNCH_DEC(GG)                                                                                                  \
    attn_decode_kernel<GG, FUSED><<<grid, 256, 0, a.st>>>((const bf16*)a.q, a.q_token_stride, (bf16*)a.k_cache, (bf16*)a.v_cache, \
                                                          a.context_lens, a.block_tables, a.table_stride, (int)a.num_kv_heads, \
                                                          sl2, (bf16*)a.out, ws_acc, ws_ml, min_range, a.items, fz, arrivals)
    switch (g) {
        case 1: LAUNCH_DEC(1); break;
        case 2: LAUNCH_DEC(2); break;
        case 4: LAUNCH_DEC(4); break;
        case 5: LAUNCH_DEC(5); break;
        case 8: LAUNCH_DEC(8); break;
        default:
            nvh_set_error("%s: group size Hq/Hkv = %d unsupported (1,2,4,5,8)", name, g);
            return NVH_ENOSYS;
    }
#undef LAUNCH_DEC
    NVH_CHECK_LAUNCH(name);
    if (a.num_splits > 1 && arrivals == nullptr)
        return nvh_launch_attn_combine(ws_acc, ws_ml, a.out, a.batch * a.num_q_heads, a.num_splits, a.context_lens, (int)a.num_q_heads,
                                       min_range, a.items.seq_active, a.st);
    return NVH_OK;
}
}  // namespace

extern "C" int32_t nvh_attn_decode(const void* q, int64_t q_token_stride, const void* k_cache, const void* v_cache,
                                   const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                   int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                   float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys, void* workspace,
                                   nvh_stream_t stream) {
    DecodeArgs a{q, q_token_stride, const_cast<void*>(k_cache), const_cast<void*>(v_cache), context_lens, block_tables,
                 table_stride, batch, num_q_heads, num_kv_heads, head_dim, softmax_scale, out, num_splits, min_range_keys, workspace,
                 (hipStream_t)stream, DecodeItems{}, 0};
    return launch_decode<false>("nvh_attn_decode", a, FusedQkv{});
}

// ---- nvh_attn_decode for callers that cannot allocate: the per-range partial results live in a device array of the
// code object instead of a caller workspace.  Every GPU the library runs on holds its own copy of the array (at its own
// address) until the process ends; calls that overlap on one GPU share it and race.  Nothing in it has to start at
// zero: the merge kernel reads exactly the records the attention kernel of the same call wrote.
#define NVH_NOALLOC_SCRATCH_FLOATS (2 * 1024 * 1024)          /* 8 MiB */
static __device__ __align__(256) float scratch[NVH_NOALLOC_SCRATCH_FLOATS];
namespace {
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};                       // device address of `scratch` on each GPU, resolved once
}  // namespace

// Resolves the current device's copy of the array (no allocation, no stream work).  Host code must never pass the
// array's name to a kernel: the name is a host-side stand-in, only this address is device memory.
int32_t nvh_noalloc_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    NVH_REQUIRE(e == hipSuccess && dev >= 0 && dev < kMaxDevices, NVH_ELAUNCH, "nvh_attn_decode_noalloc: hipGetDevice: %s (device %d)",
                hipGetErrorString(e), dev);
    if (g_scratch_ptr[dev] == nullptr) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        NVH_REQUIRE(e == hipSuccess && p != nullptr, NVH_ELAUNCH, "nvh_attn_decode_noalloc: hipGetSymbolAddress on device %d: %s", dev,
                    hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return NVH_OK;
}

extern "C" int64_t nvh_attn_decode_noalloc_max_bytes(void) { return (int64_t)NVH_NOALLOC_SCRATCH_FLOATS * (int64_t)sizeof(float); }

extern "C" int32_t nvh_attn_decode_noalloc(const void* q, int64_t q_token_stride, const void* k_cache, const void* v_cache,
                                           const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                           int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                           float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys,
                                           nvh_stream_t stream) {
    NVH_REQUIRE(batch >= 0 && num_q_heads > 0 && num_splits >= 1, NVH_EINVAL, "nvh_attn_decode_noalloc: batch / heads / num_splits out of range");
    const int64_t need = nvh_attn_decode_workspace_bytes(batch, num_q_heads, head_dim, num_splits);
    NVH_REQUIRE(need <= nvh_attn_decode_noalloc_max_bytes(), NVH_EINVAL,
                "nvh_attn_decode_noalloc: batch %ld x %ld heads x %d key ranges needs %ld bytes of partial results, the built-in "
                "array holds %ld; use fewer ranges or nvh_attn_decode with a workspace",
                (long)batch, (long)num_q_heads, num_splits, (long)need, (long)nvh_attn_decode_noalloc_max_bytes());
    float* ws = nullptr;
    const int32_t rc = nvh_noalloc_scratch(&ws);
    if (rc != NVH_OK) return rc;
    DecodeArgs a{q, q_token_stride, const_cast<void*>(k_cache), const_cast<void*>(v_cache), context_lens, block_tables,
                 table_stride, batch, num_q_heads, num_kv_heads, head_dim, softmax_scale, out, num_splits, min_range_keys, ws,
                 (hipStream_t)stream, DecodeItems{}, 0};
    return launch_decode<false>("nvh_attn_decode_noalloc", a, FusedQkv{});
}

extern "C" int32_t nvh_attn_decode_items(const void* q, int64_t q_token_stride, const void* k_cache, const void* v_cache,
                                         const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                         int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                         float softmax_scale, void* out, const int32_t* item_seq, const int32_t* item_range,
                                         const int32_t* seq_active, const int32_t* seq_chunk, int64_t num_items, int32_t max_ranges,
                                         void* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(item_seq, NVH_EINVAL, "nvh_attn_decode_items: null work list");
    DecodeArgs a{q, q_token_stride, const_cast<void*>(k_cache), const_cast<void*>(v_cache), context_lens, block_tables,
                 table_stride, batch, num_q_heads, num_kv_heads, head_dim, softmax_scale, out, max_ranges, 0, workspace,
                 (hipStream_t)stream, DecodeItems{item_seq, item_range, seq_active, seq_chunk, (int)batch, max_ranges}, num_items};
    return launch_decode<false>("nvh_attn_decode_items", a, FusedQkv{});
}

extern "C" int32_t nvh_attn_decode_fused_qkv(const void* qkv, int64_t token_stride, const void* q_norm_weight,
                                             const void* k_norm_weight, float eps, const int64_t* positions,
                                             const float* cos_sin, void* k_cache, void* v_cache,
                                             const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                             int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                             float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys,
                                             void* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(q_norm_weight && k_norm_weight && positions && cos_sin, NVH_EINVAL, "nvh_attn_decode_fused_qkv: null pointer");
    NVH_REQUIRE(nvh_aligned16(q_norm_weight) && nvh_aligned16(k_norm_weight) && nvh_aligned16(cos_sin), NVH_EINVAL,
                "nvh_attn_decode_fused_qkv: 16-byte alignment required");
    DecodeArgs a{qkv, token_stride, k_cache, v_cache, context_lens, block_tables, table_stride, batch, num_q_heads,
                 num_kv_heads, head_dim, softmax_scale, out, num_splits, min_range_keys, workspace, (hipStream_t)stream, DecodeItems{}, 0};
    return launch_decode<true>("nvh_attn_decode_fused_qkv", a,
                               FusedQkv{(const bf16*)q_norm_weight, (const bf16*)k_norm_weight, positions, cos_sin, eps});
}

extern "C" int32_t nvh_attn_decode_fused_qkv_items(const void* qkv, int64_t token_stride, const void* q_norm_weight,
                                                   const void* k_norm_weight, float eps, const int64_t* positions,
                                                   const float* cos_sin, void* k_cache, void* v_cache,
                                                   const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                                   int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                                   float softmax_scale, void* out, const int32_t* item_seq,
                                                   const int32_t* item_range, const int32_t* seq_active, const int32_t* seq_chunk,
                                                   int64_t num_items, int32_t max_ranges, void* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(q_norm_weight && k_norm_weight && positions && cos_sin && item_seq, NVH_EINVAL, "nvh_attn_decode_fused_qkv_items: null pointer");
    NVH_REQUIRE(nvh_aligned16(q_norm_weight) && nvh_aligned16(k_norm_weight) && nvh_aligned16(cos_sin), NVH_EINVAL,
                "nvh_attn_decode_fused_qkv_items: 16-byte alignment required");
    DecodeArgs a{qkv, token_stride, k_cache, v_cache, context_lens, block_tables, table_stride, batch, num_q_heads,
                 num_kv_heads, head_dim, softmax_scale, out, max_ranges, 0, workspace, (hipStream_t)stream,
                 DecodeItems{item_seq, item_range, seq_active, seq_chunk, (int)batch, max_ranges}, num_items};
    return launch_decode<true>("nvh_attn_decode_fused_qkv_items", a,
                               FusedQkv{(const bf16*)q_norm_weight, (const bf16*)k_norm_weight, positions, cos_sin, eps});
}
