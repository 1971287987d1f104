// y[M,N] = x[M,K] @ W[N,K]^T (+bias): bf16 in, fp32 MFMA accumulation, bf16 out (or fp32
// split-K slabs).  Both operands are K-contiguous, which is exactly the MFMA 16x16x32 bf16
// fragment shape (8 consecutive k per lane), so tiles go global -> LDS by 16-byte LDS-DMA
// (global_load_lds) with no register staging and no transposes.
//
// Workgroup = 4 waves, tile BM(x rows) x BN(W rows) x 64(k), a ring of STAGES LDS buffers and
// ONE barrier per k-tile.  The DMA of tile t+STAGES-1 is issued while tile t is multiplied and
// stays in flight across the barriers: each wave waits with a COUNTED s_waitcnt vmcnt(n) for its
// own pieces of tile t only, then a raw s_barrier makes every wave's pieces visible.  That keeps
// STAGES-1 tiles of HBM/L2 latency covered per workgroup, which is what small-M (decode) and
// short-K shapes need: they are bound by the dependent load chain, not by MFMA or bandwidth.
// W rows are the MFMA "A" operand and x rows the "B" operand, so each lane ends up with 4
// consecutive output columns (n) of one output row (m) -> 8-byte bf16 / 16-byte fp32 stores.
//
// LDS tile rows are 128 B (64 k).  16-byte chunk `c` of tile row `r` is stored at chunk
// position c ^ ((r >> 1) & 7): with that XOR every 16-lane group of a ds_read_b128 fragment
// read touches all 64 banks exactly once.  LDS-DMA writes lane-linear, so the XOR is applied
// to the per-lane SOURCE address and again on the read.
//
// The tile shape is picked per call so that the grid covers the 256 CUs (small tiles for narrow
// outputs, BM = 16/32/64 for decode batches); split-K (gridDim.z) adds workgroups for narrow-N
// projections, writing fp32 slabs that the consumer kernel sums.
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include "common.h"

#define BK 64

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate is 6 bits");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wait until at most `ahead` tiles (PW DMA instructions each) of this wave are still in flight.
template <int PW, int MAXAHEAD>
__device__ __forceinline__ void wait_tiles_in_flight(int ahead) {
    if constexpr (MAXAHEAD == 0) {
        wait_vmcnt<0>();
    } else {
        if (ahead >= MAXAHEAD) wait_vmcnt<(PW * MAXAHEAD < 64 ? PW * MAXAHEAD : 63)>();
        else wait_tiles_in_flight<PW, MAXAHEAD - 1>(ahead);
    }
}

// Epilogue of the fused QKV projection: the output row is [q heads | k heads | v heads] of 128 columns each, a
// 128-column tile is exactly one head.  Rows below raw_rows are stored as they are (their consumer applies the
// q/k epilogue itself); the others get per-head RMSNorm + RoPE on q and k heads, and k/v heads also go to their
// KV-cache slot (slot_mapping null or -1: no store).
struct QkvEpilogue {
    const bf16 *q_norm_w, *k_norm_w;
    const int64_t* positions;
    const float* cos_sin;
    bf16 *k_cache, *v_cache;
    const int32_t* slot_mapping;
    float eps;
    int hq, hkv, raw_rows;
};

// Epilogue of the LM-head projection fused with the sampler (SAMPLE): the logits never leave the workgroup.  Each lane
// holds 4 consecutive columns of a row = one group of the sampler's noise generator; it scores them exactly as
// sample_slice_kernel does (logit rounded to bf16 first), the workgroup keeps the best candidate per row of its 256
// columns and leaves it in cand[row][column tile]; nvh_launch_sample_final then picks among the column tiles.
struct SampleEpilogue {
    const float* temps;
    ValIdx* cand;
    uint64_t seed, offset;
    int64_t col_offset;       // first column of this launch's columns in the full row (shard of the vocabulary, column cut)
    int cand_stride;          // candidates per row in `cand` (all launches of a call together)
    int cand_base;            // this launch's first slot in a row's candidates
};

// RMSNorm as a row scale (QKV and SwiGLU epilogues of prefill-sized steps): the projection runs over the UN-normalised
// rows with the norm weight folded into its weight matrix (W' = W * w along K), and every accumulator of row m is
// multiplied by 1 / sqrt(sum(x_m^2) / K + eps) before the epilogue proper.  sum(x_m^2) arrives as `parts` partial sums per
// row, left by the epilogue of the projection that produced x (nvh_gemm_residual_rowsq_bf16_nt); they are added in
// index order.  sumsq == nullptr: no scaling.
struct RowScale {
    const float* sumsq;       // [M][parts]
    int parts;                // multiple of 4
    float inv_k, eps;
};

// Tile of this workgroup.  Workgroups go to the 8 XCDs round-robin by linear id.  With xcd_strip = 8 or 4 (host: gridDim.x a
// multiple of it) XCD i takes the i-th eighth of the tiles in an order that walks strips of that many column tiles row
// by row, so the workgroups resident on an XCD at one time share a few column tiles of W and a few row tiles of x in
// its L2 (Qwen3-14B shapes, 1024 rows: a prefill chunk 27.8 -> 27.15 ms; K = 1024: 2.81 -> 2.78 ms).  K-slabs (gridDim.z)
// count as further tile rows: slab z of row tile y is row y + gridDim.y * z.
__device__ __forceinline__ void xcd_tile(int xcd_strip, int& bx, int& by, int& bz) {
    bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_strip) {
        const int rows = gridDim.y * gridDim.z;
        const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total = gridDim.x * rows;
        const int xcd = id & 7, nl = xcd * (total >> 3) + min(xcd, total & 7) + (id >> 3);
        const int sh = xcd_strip == 8 ? 3 : 2, per_strip = rows << sh;
        const int strip = nl / per_strip, in = nl - strip * per_strip, row = in >> sh;
        bx = (strip << sh) + (in & (xcd_strip - 1));
        bz = row / (int)gridDim.y;
        by = row - bz * (int)gridDim.y;
    }
}

template <int BM, int BN, int WAVES_N, int WAVES_M, int STAGES, bool NT_W, bool SWIGLU, bool QKV = false, bool SAMPLE = false, bool ROWSCALE = false>
__global__ __launch_bounds__(64 * WAVES_N * WAVES_M) void gemm_bf16_nt_kernel(
    const bf16* __restrict__ x, int64_t x_stride, const bf16* __restrict__ w, const bf16* __restrict__ bias,
    bf16* __restrict__ y, int64_t y_stride, float* __restrict__ partials, int M, int N, int K, int k_per_split,
    QkvEpilogue qe, bool wide_store, SampleEpilogue se, int xcd_strip, RowScale rs) {
    constexpr int NW = WAVES_N * WAVES_M;                 // waves per workgroup (4; 2 for the skinny decode tiles)
    static_assert(NW == 8 || NW == 4 || NW == 2, "2, 4 or 8 waves");
    constexpr int WN = BN / WAVES_N, WM = BM / WAVES_M;   // wave tile
    constexpr int TN = WN / 16, TM = WM / 16;             // 16x16 MFMA tiles per wave
    static_assert(TN >= 1 && TM >= 1, "wave tile");
    constexpr int A_BYTES = BN * 128, B_BYTES = BM * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_GROUPS = BN / 8, B_GROUPS = BM / 8;   // 8-row x 128-B pieces = one DMA each
    constexpr int PIECES = A_GROUPS + B_GROUPS;
    constexpr int PW = (PIECES + NW - 1) / NW;            // DMA instructions per wave per tile (uniform)
    constexpr int AHEAD = STAGES - 1;                     // tiles kept in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    int bx, by, bz;
    xcd_tile(xcd_strip, bx, by, bz);
    const int n0 = bx * BN, m0 = by * BM;
    const int kbeg = bz * k_per_split;
    // QKV epilogue: the positions of this thread's tokens, requested before the main loop, so that the epilogue waits for
    // ONE memory latency (the cos / sin rows they select) instead of two in a row
    constexpr int QPASSES = QKV ? BM / (NW * 4) : 1;
    int64_t qpos[QPASSES];
    if constexpr (QKV) {
#pragma unroll
        for (int p = 0; p < QPASSES; ++p) qpos[p] = qe.positions[min(m0 + p * NW * 4 + (tid >> 4), M - 1)];
    }
    // ... and, in the eight-wave form (registers to spare), the cos / sin rows themselves right after the first stages' DMA
    constexpr bool CS_EARLY = QKV && NW == 8;
    f32x4 cs_pre[CS_EARLY ? QPASSES : 1][4];
    auto request_cs = [&]() {
        if constexpr (CS_EARLY) {
            if ((n0 >> 7) < qe.hq + qe.hkv) {                    // block-uniform: a q or k head
#pragma unroll
                for (int p = 0; p < QPASSES; ++p) {
                    const float* cs_row = qe.cos_sin + qpos[p] * 128 + 8 * (tid & 7);
                    cs_pre[p][0] = *reinterpret_cast<const f32x4*>(cs_row);
                    cs_pre[p][1] = *reinterpret_cast<const f32x4*>(cs_row + 4);
                    cs_pre[p][2] = *reinterpret_cast<const f32x4*>(cs_row + 64);
                    cs_pre[p][3] = *reinterpret_cast<const f32x4*>(cs_row + 68);
                }
            }
        }
    };
    const int kend = min(K, kbeg + k_per_split);
    const int ntiles = max(0, (kend - kbeg) / BK);

    // per-lane DMA source: row (lane>>3) of the 8-row piece, chunk position lane&7
    const int dma_r = lane >> 3, dma_pos = lane & 7;
    // source row pointers of this wave's PW pieces (k offset added per tile)
    const bf16* src_row[PW];
    int lds_off[PW];
    bool piece_is_w[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int p = wave + i * NW;
        if (p >= PIECES) p -= PIECES;                     // duplicate piece keeps the per-wave count uniform
        const bool is_b = p >= A_GROUPS;
        const int grp = is_b ? p - A_GROUPS : p;
        const int r = grp * 8 + dma_r;
        const int chunk = dma_pos ^ ((r >> 1) & 7);
        if (is_b) src_row[i] = x + (int64_t)min(m0 + r, M - 1) * x_stride + chunk * 8;
        else src_row[i] = w + (int64_t)min(n0 + r, N - 1) * K + chunk * 8;
        lds_off[i] = (is_b ? A_BYTES : 0) + grp * 1024;
        piece_is_w[i] = !is_b;
    }
    auto stage = [&](int t) {
        const int k0 = kbeg + t * BK;
        char* base = smem + (t % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            // weights that only this workgroup row reads (one M tile: decode) are streamed non-temporally
            if (NT_W && piece_is_w[i])
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src_row[i] + k0), (LDS_AS void*)(base + lds_off[i]), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src_row[i] + k0), (LDS_AS void*)(base + lds_off[i]), 16, 0, 0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // Row scales of this lane's TM rows (see RowScale).  Eight-wave tiles (the only ones the step executor sends here) request
    // the partial sums ahead of the main loop with loads the compiler does not track and first look at them in the
    // epilogue: up to 32 partial sums per row wait in registers, and their latency - the rows were written by the kernel
    // before this one, on other XCDs - hides behind the loop instead of delaying its start by ~3 us.  Other tiles read
    // them in the epilogue.
    constexpr bool RS = ROWSCALE && (QKV || SWIGLU);     // (instantiated for the eight-wave 128-row tiles only: launch_gemm_nt)
    constexpr bool RS_EARLY = RS && NW == 8 && TM == 2;
    // (the four lanes fq = 0..3 that share a row split its partial sums: lane fq takes groups fq and fq + 4 of four)
    f32x4 rsq[RS_EARLY ? TM * 2 : 1];
    if constexpr (RS_EARLY) {
        if (rs.sumsq != nullptr) {                           // block-uniform
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const float* ps = rs.sumsq + (int64_t)min(m0 + wm * WM + j * 16 + fr, M - 1) * rs.parts;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float* pa = ps + 4 * min(fq + 4 * u, (rs.parts >> 2) - 1);      // (groups past parts / 4 re-read the last one and are not added)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rsq[j * 2 + u]) : "v"(pa) : "memory");
                }
            }
        }
    }
    auto row_scales = [&](float (&rscale)[TM]) {
#pragma unroll
        for (int j = 0; j < TM; ++j) rscale[j] = 1.0f;
        if constexpr (RS) {
            if (rs.sumsq != nullptr) {
                if constexpr (RS_EARLY) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsq[0]), "+v"(rsq[1]), "+v"(rsq[2]), "+v"(rsq[3])::"memory");
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    float ss = 0.f;
                    if constexpr (RS_EARLY) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (fq + 4 * u < (rs.parts >> 2)) {
                                const f32x4 v = rsq[j * 2 + u];
                                ss += v[0];
                                ss += v[1];
                                ss += v[2];
                                ss += v[3];
                            }
                        ss += __shfl_xor(ss, 16, 64);
                        ss += __shfl_xor(ss, 32, 64);
                    } else {
                        const f32x4* ps = reinterpret_cast<const f32x4*>(rs.sumsq + (int64_t)min(m0 + wm * WM + j * 16 + fr, M - 1) * rs.parts);
                        for (int q = 0; q < (rs.parts >> 2); ++q) {
                            const f32x4 v = ps[q];
                            ss += v[0];
                            ss += v[1];
                            ss += v[2];
                            ss += v[3];
                        }
                    }
                    rscale[j] = 1.0f / sqrtf(ss * rs.inv_k + rs.eps);
                }
            }
        }
    };
    if constexpr (NW == 8) {
        // Eight waves, two per SIMD.  Every wave runs the same four phases per k-tile, a barrier before each:
        //   READ k0 (12 fragment reads) | MFMA k0 (32) | READ k1 | MFMA k1
        // and waves 4-7 run ONE barrier behind waves 0-3 (an extra barrier before their loop, one after the others'),
        // so on every SIMD one wave issues MFMAs while the other reads LDS.  Barrier n of the workgroup is phase
        // n mod 4 of tile n / 4 for the first half and phase (n - 1) mod 4 for the second.  Hence:
        //  * a wave waits for its DMA pieces of tile t+1 before its 4th barrier of tile t: by the time the first half
        //    passes its 1st barrier of tile t+1 (the same barrier for the second half) every piece has landed;
        //  * the DMA of tile t+1 (two LDS stages: the buffer of tile t-1) is issued after the 1st barrier of tile t;
        //    the last readers of tile t-1 are the second half's READ k1, drained (lgkmcnt) before that very barrier.
        // With more than two stages the DMA runs STAGES - 1 tiles ahead and the wait before the fourth barrier leaves
        // the tiles behind t+1 in flight (measured with 128 x 128 tiles and three stages: 13.8 us against 15.1 us for
        // 1024 x 4096 x 1024, slower on every grid that is not exactly one workgroup per CU; not dispatched).
        const bool second_half = wave >= 4;              // wave-uniform
        bf16x8 af[TN], bfr[TM];
        auto read_frags = [&](const char* a_base, int ks) {
            const char* b_base = a_base + A_BYTES;
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * WM + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * WN + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(a_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
        };
        auto mfmas = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto phase_barrier = [&]() {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
        };
        for (int t = 0; t < AHEAD && t < ntiles; ++t) stage(t);
        request_cs();                                            // younger than the stages: the counted waits below only get stricter while these are in flight
        wait_tiles_in_flight<PW, (AHEAD > 0 ? AHEAD - 1 : 0)>(min(AHEAD - 1, ntiles - 1));
        if (second_half) phase_barrier();
        for (int t = 0; t < ntiles; ++t) {
            const char* base = smem + (t % STAGES) * STAGE_BYTES;
            phase_barrier();
            if (t + AHEAD < ntiles) stage(t + AHEAD);
            read_frags(base, 0);
            phase_barrier();
            mfmas();
            phase_barrier();
            read_frags(base, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this tile's reads
            wait_tiles_in_flight<PW, (AHEAD > 0 ? AHEAD - 1 : 0)>(max(0, min(AHEAD - 1, ntiles - 2 - t)));   // own pieces of tile t+1
            __builtin_amdgcn_sched_barrier(0);
            phase_barrier();
            mfmas();
        }
        if (!second_half) phase_barrier();
    } else {
    for (int t = 0; t < AHEAD && t < ntiles; ++t) stage(t);
    for (int t = 0; t < ntiles; ++t) {
        // tiles t+1 .. min(t+AHEAD-1, ntiles-1) may stay in flight; tile t must have landed
        wait_tiles_in_flight<PW, (AHEAD > 0 ? AHEAD - 1 : 0)>(min(AHEAD - 1, ntiles - 1 - t));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + AHEAD < ntiles) stage(t + AHEAD);          // slot (t-1) % STAGES: consumed before this barrier
        const char* a_base = smem + (t % STAGES) * STAGE_BYTES;
        const char* b_base = a_base + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TN], bfr[TM];
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * WN + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(a_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * WM + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    }

    // epilogue: lane holds n = n_base + fq*4 .. +3 of row m = m_base + fr
    if constexpr (SAMPLE) {
        static_assert((NW == 8 || NW == 4) && WAVES_N == 2 && !SWIGLU && !QKV, "sampling epilogue: two column halves per workgroup");
        // the stages are free: in the eight-wave loop every read of them was waited for before its last barrier; the
        // four-wave loop ends without one
        if constexpr (NW != 8) __syncthreads();
        ValIdx (*sm_cand)[TM][16] = reinterpret_cast<ValIdx (*)[TM][16]>(smem);
        ValIdx best[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * WM + j * 16 + fr;
            const float inv_t = 1.0f / se.temps[min(m, M - 1)];
            best[j] = ValIdx{-INFINITY, (int64_t)0x7fffffffffffffffLL};
#pragma unroll
            for (int i = 0; i < TN; ++i) {                       // increasing columns: "strictly greater" keeps the lowest index
                const int n = n0 + wn * WN + i * 16 + fq * 4;
                if (n >= N) continue;
                float z[4], sc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = bf2f(f2bf(acc[i][j][e]));
                nvh_race_scores(z, inv_t, n + se.col_offset, m, se.seed, se.offset, sc);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (sc[e] > best[j].v) best[j] = ValIdx{sc[e], n + e + se.col_offset};
            }
            if (best[j].i == (int64_t)0x7fffffffffffffffLL) best[j].i = n0 + wn * WN + fq * 4 + se.col_offset;    // all -inf: offer the first column
            // the 4 lanes fq = 0..3 of a row, then the two column halves (waves wn = 0, 1) through LDS
#pragma unroll
            for (int sft = 16; sft <= 32; sft <<= 1) {
                ValIdx o;
                o.v = __shfl_xor(best[j].v, sft, 64);
                o.i = __shfl_xor(best[j].i, sft, 64);
                best[j] = better(best[j], o);
            }
        }
        if (wn == 1 && fq == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j) sm_cand[wm][j][fr] = best[j];
        }
        __syncthreads();
        if (wn == 0 && fq == 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * WM + j * 16 + fr;
                if (m < M) se.cand[(int64_t)m * se.cand_stride + se.cand_base + bx] = better(best[j], sm_cand[wm][j][fr]);
            }
        }
        return;
    }
    if constexpr (QKV) {
        // The tile (one head of BM tokens) goes through LDS as bf16 - the rounding point of the separate GEMM - and
        // is then walked like the stand-alone q/k epilogue kernel: 16 lanes per row, 8 columns per lane, the same
        // device function, so both routes give the same bits.
        static_assert(BN == 128 && (NW == 4 || NW == 8) && !SWIGLU, "one head per tile, 256 or 512 threads");
        constexpr int RPP = NW * 4;                              // rows of the tile per pass: 16 lanes per row
        constexpr int TS = 136;                                  // LDS row stride in elements (272 B: rows land in different banks)
        const int c = tid & 15, rr = tid >> 4;
        const int head = n0 >> 7;                                // block-uniform
        const bool normed = head < qe.hq + qe.hkv;
        // cos / sin rows of this thread's 8 tokens: requested before the tile goes through LDS, so their latency
        // (position, then table row) hides behind the staging
        f32x4 cs[BM / RPP][4];
        if constexpr (CS_EARLY) {
#pragma unroll
            for (int p = 0; p < BM / RPP; ++p)
#pragma unroll
                for (int u = 0; u < 4; ++u) cs[p][u] = cs_pre[p][u];
        } else if (normed) {
#pragma unroll
            for (int p = 0; p < BM / RPP; ++p) {
                const float* cs_row = qe.cos_sin + qpos[p] * 128 + 8 * (c & 7);
                cs[p][0] = *reinterpret_cast<const f32x4*>(cs_row);
                cs[p][1] = *reinterpret_cast<const f32x4*>(cs_row + 4);
                cs[p][2] = *reinterpret_cast<const f32x4*>(cs_row + 64);
                cs[p][3] = *reinterpret_cast<const f32x4*>(cs_row + 68);
            }
        }
        float rscale[TM];
        row_scales(rscale);
        __syncthreads();                                         // every wave is done with the last stage
        bf16* tile = reinterpret_cast<bf16*>(smem);              // [BM][TS]
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WM + j * 16 + fr, col = wn * WN + i * 16 + fq * 4;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][e] * rscale[j]);
                *reinterpret_cast<bf16x4*>(tile + row * TS + col) = o;
            }
        __syncthreads();
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>((head < qe.hq ? qe.q_norm_w : qe.k_norm_w) + c * 8);
        bf16* cache = head < qe.hq ? nullptr : (normed ? qe.k_cache : qe.v_cache);
        const int kvh = head - qe.hq - (normed ? 0 : qe.hkv);
        // all epilogue arithmetic first, the stores afterwards
        bf16x8 outv[BM / RPP];
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const int row = p * RPP + rr, m = min(m0 + row, M - 1);          // rows past M compute on row M-1, store nothing
            bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * TS + c * 8);
            if (m >= qe.raw_rows && normed) v = qknorm_rope_head_cs(v, wv, cs[p][0], cs[p][1], cs[p][2], cs[p][3], c, qe.eps);   // uniform over a 16-lane row
            outv[p] = v;
        }
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const int m = m0 + p * RPP + rr;
            if (m >= M) continue;
            *reinterpret_cast<bf16x8*>(y + (int64_t)m * y_stride + n0 + c * 8) = outv[p];
            if (m >= qe.raw_rows && cache != nullptr && qe.slot_mapping != nullptr) {
                const int64_t slot = qe.slot_mapping[m];
                if (slot >= 0) *reinterpret_cast<bf16x8*>(cache + (slot * qe.hkv + kvh) * 128 + c * 8) = outv[p];
            }
        }
        return;
    }
    if constexpr (SWIGLU) {
        // W rows are interleaved in blocks of 16: [gate 16b..16b+15 | up 16b..16b+15], so tile i (even) holds
        // gate and tile i+1 the matching up columns in the same lanes.  y[m][16b + ..] = silu(gate) * up with the
        // rounding points of the separate kernels: GEMM output -> bf16, silu -> bf16, product -> bf16.
        // pairs of 16-row tiles: configurations with an odd TN are never launched with SWIGLU (dispatch_gemm)
        float rscale[TM];
        row_scales(rscale);
#pragma unroll
        for (int i = 0; i + 1 < TN; i += 2) {
            const int n = n0 + wn * WN + i * 16;                 // first interleaved row of the gate block
            if (n >= N) continue;
            const int col = (n >> 5) * 16 + fq * 4;              // output column
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * WM + j * 16 + fr;
                if (m >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = bf2f(f2bf(acc[i][j][e] * rscale[j]));
                    const float uv = bf2f(f2bf(acc[i + 1][j][e] * rscale[j]));
                    const float sl = gv / (1.0f + expf(-gv));
                    o[e] = f2bf(bf2f(f2bf(sl)) * uv);
                }
                *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + col) = o;
            }
        }
        return;
    }
    if constexpr (BN == 128 && NW == 4 && BM >= 64) {
        // full 128-column tiles with a bf16 result: through LDS, so every lane stores 16 contiguous bytes and a row
        // of the tile leaves as one 256-B line (the fragment layout would store 8 B per lane, 32 B per row piece)
        if (partials == nullptr && n0 + BN <= N && wide_store) {                 // block-uniform
            constexpr int TS = 136;
            __syncthreads();                                     // every wave is done with the last stage
            bf16* tile = reinterpret_cast<bf16*>(smem);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int col = wn * WN + i * 16 + fq * 4;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias != nullptr) {
                    const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + n0 + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][e] + bv[e]);
                    *reinterpret_cast<bf16x4*>(tile + (wm * WM + j * 16 + fr) * TS + col) = o;
                }
            }
            __syncthreads();
            const int c = tid & 15, rr = tid >> 4;
#pragma unroll
            for (int p = 0; p < BM / 16; ++p) {
                const int row = p * 16 + rr, m = m0 + row;
                if (m < M) *reinterpret_cast<bf16x8*>(y + (int64_t)m * y_stride + n0 + c * 8) = *reinterpret_cast<const bf16x8*>(tile + row * TS + c * 8);
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn * WN + i * 16 + fq * 4;
        if (n >= N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
            const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * WM + j * 16 + fr;
            if (m >= M) continue;
            if (partials != nullptr) {
                *reinterpret_cast<f32x4*>(partials + ((int64_t)bz * M + m) * N + n) = acc[i][j];
            } else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][e] + bv[e]);
                *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ 64 x 64 tiles, two wave groups over alternate k-tiles
// A grid of one 64 x 64 tile per CU (the O / down projections of a prefill chunk, the QKV projection of 256 decode rows)
// leaves every SIMD with ONE wave whose LDS reads and MFMAs of a k-tile run one after the other; two such workgroups
// on a CU take 1.3-1.4 times as long for twice the work.  Here a workgroup is eight waves = two groups of four: a stage
// holds TWO k-tiles, group g multiplies tile 2t + g (32 x 32 per wave, the operand reuse of the four-wave tile), one
// barrier per pair, and the groups' sums meet in LDS at the end: group 0's + group 1's, so NOT the sequential-k sum of the
// other tiles (within the GEMM tolerance of it; no caller compares these shapes bit for bit with another route).
// Alone, on weights the Infinity Cache holds: 1024 x 1024 x 2048 / 3072 11.8 / 16.1 -> 10.7 / 14.3 us; inside a step,
// where every XCD pulls x and W through its own L2 first: a prefill chunk 2.745 -> 2.736 ms, a decode step of 256
// sequences 6.251 -> 6.240 ms.  (The same for K-slab grids - the O / down projections of 256 decode rows in four slabs:
// no change, 6.251 against 6.253 ms; not built in.)
template <int STAGES, bool RESIDUAL = false>
__global__ __launch_bounds__(512) void gemm_kgroups_bf16_nt_kernel(const bf16* __restrict__ x, int64_t x_stride,
                                                                   const bf16* __restrict__ w, const bf16* __restrict__ bias,
                                                                   bf16* __restrict__ y, int64_t y_stride, int M, int N, int K,
                                                                   int xcd_strip, bf16* __restrict__ residual, float* __restrict__ row_sumsq) {
    constexpr int BM = 64, BN = 64, NW = 8, TN = 2, TM = 2;
    constexpr int A_BYTES = BN * 128, B_BYTES = BM * 128, TILE_BYTES = A_BYTES + B_BYTES, STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int GROUPS8 = BN / 8 + BM / 8;
    constexpr int PW = 2 * GROUPS8 / NW;
    constexpr int AHEAD = STAGES - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wn = w4 & 1, wm = w4 >> 1;
    int bx, by, bz;
    xcd_tile(xcd_strip, bx, by, bz);
    const int n0 = bx * BN, m0 = by * BM;
    const int nstages = K / (2 * BK);
    const int dma_r = lane >> 3, dma_pos = lane & 7;
    const bf16* src_row[PW];
    int lds_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = wave + i * NW;
        const int half = p / GROUPS8, q = p % GROUPS8;
        const bool is_b = q >= BN / 8;
        const int g8 = is_b ? q - BN / 8 : q;
        const int r = g8 * 8 + dma_r;
        const int chunk = dma_pos ^ ((r >> 1) & 7);
        if (is_b) src_row[i] = x + (int64_t)min(m0 + r, M - 1) * x_stride + chunk * 8 + half * BK;
        else src_row[i] = w + (int64_t)min(n0 + r, N - 1) * K + chunk * 8 + half * BK;
        lds_off[i] = half * TILE_BYTES + (is_b ? A_BYTES : 0) + g8 * 1024;
    }
    auto stage = [&](int t) {
        const int k0 = t * 2 * BK;
        char* base = smem + (t % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src_row[i] + k0), (LDS_AS void*)(base + lds_off[i]), 16, 0, 0);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // residual mode: this lane's 16 residual elements, requested ahead of the main loop by loads the compiler does not track
    // (first looked at in the epilogue, behind a wait of their own)
    bf16x4 res_pre[TN][TM];
    if (RESIDUAL && grp == 0) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const bf16* rp = residual + (int64_t)(m0 + wm * 32 + j * 16 + fr) * N + n0 + wn * 32 + i * 16 + fq * 4;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(res_pre[i][j]) : "v"(rp) : "memory");
            }
    }
    for (int t = 0; t < AHEAD && t < nstages; ++t) stage(t);
    for (int t = 0; t < nstages; ++t) {
        wait_tiles_in_flight<PW, (AHEAD > 0 ? AHEAD - 1 : 0)>(min(AHEAD - 1, nstages - 1 - t));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + AHEAD < nstages) stage(t + AHEAD);
        const char* a_base = smem + (t % STAGES) * STAGE_BYTES + grp * TILE_BYTES;
        const char* b_base = a_base + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TN], bfr[TM];
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * 32 + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(a_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * 32 + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) red[(w4 * TN * TM + i * TM + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (grp == 1) return;
    if constexpr (RESIDUAL) {
        // The product is added to the residual stream in place (y is not written) with the arithmetic of the add + RMSNorm
        // launch that would follow: product rounded to bf16, sum in fp32, new residual = that sum rounded to bf16; and the
        // fp32 sums' squares are added up per row over this wave's 32 columns: row_sumsq[m][2 * column tile + wn], for the
        // consumer that applies the norm as a row scale (RowScale).  Whole tiles only (host-checked).
        float sq[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) sq[j] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(res_pre[0][0]), "+v"(res_pre[0][1]), "+v"(res_pre[1][0]), "+v"(res_pre[1][1])::"memory");
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * 32 + i * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * 32 + j * 16 + fr;
                const f32x4 other = red[(w4 * TN * TM + i * TM + j) * 64 + lane];
                bf16* rp = residual + (int64_t)m * N + n;
                const bf16x4 r4 = res_pre[i][j];
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xs = bf2f(f2bf(acc[i][j][e] + other[e])) + bf2f(r4[e]);
                    sq[j] += xs * xs;
                    o[e] = f2bf(xs);
                }
                *reinterpret_cast<bf16x4*>(rp) = o;
            }
        }
        const int parts = 2 * (int)gridDim.x;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            float t = sq[j];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (fq == 0) row_sumsq[(int64_t)(m0 + wm * 32 + j * 16 + fr) * parts + 2 * bx + wn] = t;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn * 32 + i * 16 + fq * 4;
        if (n >= N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
            const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = bf2f(b4[e]);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * 32 + j * 16 + fr;
            if (m >= M) continue;
            const f32x4 other = red[(w4 * TN * TM + i * TM + j) * 64 + lane];
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf((acc[i][j][e] + other[e]) + bv[e]);
            *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + n) = o;
        }
    }
}

// ------------------------------------------------------------------ residual add + RMSNorm in the GEMM's prologue
// Decode batches: the add + RMSNorm launch in front of a projection moves a few KB and costs a whole dependent
// launch.  Here every workgroup derives the normalised rows itself - same arithmetic, lane for lane, as
// rmsnorm_kernel<64, 2, true, PARTIALS> (slab p into partial sum p mod 4, (s0 + s1) + (s2 + s3), one bf16 rounding,
// residual add, sum of squares per lane in chunk order, wave_sum) - and keeps them in LDS as the MFMA "B" operand
// of every k-tile: [K / 64][BM rows][128 B], chunks XOR-swizzled like the ring's.  The weight ring is the generic
// kernel's (the DMA of the first tiles is issued before the prologue) and so are the MFMA order and the epilogues,
// hence the same bits as nvh_add_rmsnorm followed by nvh_gemm_bf16_nt / nvh_gemm_swiglu_bf16_nt.  Workgroup 0 writes
// the new residual to res_out, which must not alias res_in (other workgroups may still be reading it).
struct NormPrologue {
    const bf16* prev;        // bf16 output of the previous projection [M, K]; ignored when slabs != null
    const float* slabs;      // its split-K slabs [num_slabs][M][K] (fp32)
    const bf16* res_in;
    bf16* res_out;
    const bf16* norm_w;
    float eps;
    int num_slabs;
};

template <int BM, int BN, int WAVES_N, int WAVES_M, int STAGES, bool SWIGLU, int LW = 1>
__global__ __launch_bounds__(64 * (8 + LW)) void gemm_addnorm_bf16_nt_kernel(
    NormPrologue np, const bf16* __restrict__ w, bf16* __restrict__ y, int64_t y_stride, int M, int N, int K) {
    // 8 + LW waves: eight normalise rows (row r goes to wave r mod 8), the first WAVES_N x WAVES_M of them then multiply,
    // and the last LW (one; two for 64-column tiles) do nothing but request the workgroup's weight slice (every k-tile
    // has a stage of its own: K <= 64 STAGES) and wait for it tile by tile.  A wave's loads retire in order, so a wave
    // that requested both would see its rows only after the whole slice had landed.
    constexpr int NW = WAVES_N * WAVES_M;
    constexpr int PWAVES = 8;                             // waves that normalise rows; the first NW of them go on to multiply
    static_assert(NW <= PWAVES, "compute waves are prologue waves");
    constexpr int WN = BN / WAVES_N, WM = BM / WAVES_M;
    constexpr int TN = WN / 16, TM = WM / 16;
    static_assert(TN >= 1 && TM >= 1, "wave tile");
    constexpr int A_BYTES = BN * 128;
    constexpr int A_GROUPS = BN / 8;                      // 8-row x 128-B pieces = one DMA each
    constexpr int LP = A_GROUPS / LW;                     // pieces of a k-tile per loader wave
    static_assert(A_GROUPS % LW == 0 && LP * (STAGES - 1) < 64, "a loader's tiles in flight must fit the 6-bit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* panel = smem + STAGES * A_BYTES;                // [K / 64][BM][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;                       // this workgroup's rows of the batch: m0 .. m0 + rows - 1
    const int rows = min(BM, M - m0);
    const int ntiles = K / BK;                            // <= STAGES (checked on the host)

    if (wave >= PWAVES) {
        const int dma_r = lane >> 3, dma_pos = lane & 7;
        const int first = (wave - PWAVES) * LP;           // this loader's pieces: first .. first + LP - 1 of every k-tile
        const bf16* src_row[LP];
#pragma unroll
        for (int p = 0; p < LP; ++p) {
            const int r = (first + p) * 8 + dma_r;
            src_row[p] = w + (int64_t)min(n0 + r, N - 1) * K + (dma_pos ^ ((r >> 1) & 7)) * 8;
        }
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll
            for (int p = 0; p < LP; ++p)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src_row[p] + t * BK), (LDS_AS void*)(smem + t * A_BYTES + (first + p) * 1024), 16, 0, 2);
        }
        // Every k-tile is waited for at once and ONE barrier publishes the slice: by the time the rows are normalised
        // the slice has landed anyway, and a multiply loop without barriers lets the LDS reads of a k-step run ahead
        // of the MFMAs of the one before.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }

    // ---- prologue: rows wave, wave + PWAVES, .. of the batch (the waves past NW only help here)
    const int nchunk = K >> 3;
    bf16x8 wv[2];                                         // requested ahead of the rows: no second trip to memory after the reduction
#pragma unroll
    for (int i = 0; i < 2; ++i) wv[i] = *reinterpret_cast<const bf16x8*>(np.norm_w + min(lane + i * 64, nchunk - 1) * 8);
    // residual add, sum of squares, scale and weight of one row whose projection output (bf16 values) is in xv
    auto finish_row = [&](int row, float (&xv)[2][8], const bf16x8 (&rs)[2]) {
        bf16x8 nr[2];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (lane + i * 64 < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[i][j] = xv[i][j] + bf2f(rs[i][j]);
                    nr[i][j] = f2bf(xv[i][j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += xv[i][j] * xv[i][j];
            }
        }
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(ss / (float)K + np.eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(xv[i][j] * inv)) * bf2f(wv[i][j]));
                *reinterpret_cast<bf16x8*>(panel + (c >> 3) * (BM * 128) + row * 128 + (((c & 7) ^ ((row >> 1) & 7)) << 4)) = o;
                if (blockIdx.x == 0) *reinterpret_cast<bf16x8*>(np.res_out + (int64_t)(m0 + row) * K + c * 8) = nr[i];
            }
        }
    };
    if (np.slabs != nullptr) {
        for (int row = wave; row < rows; row += PWAVES) {
            float xv[2][8];
            bf16x8 rs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + i * 64;
                if (c < nchunk) {
                    rs[i] = *reinterpret_cast<const bf16x8*>(np.res_in + (int64_t)(m0 + row) * K + c * 8);
                    float acc[4][8];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
                    }
                    const float* src0 = np.slabs + (int64_t)(m0 + row) * K + c * 8;
                    const int64_t slab = (int64_t)M * K;
                    int p = 0;
                    for (; p + 4 <= np.num_slabs; p += 4) {
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
                        if (p + u < np.num_slabs) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab);
                            const f32x4 b = *reinterpret_cast<const f32x4*>(src0 + (p + u) * slab + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { acc[u][j] += a[j]; acc[u][4 + j] += b[j]; }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[i][j] = bf2f(f2bf((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])));
                }
            }
            finish_row(row, xv, rs);
        }
    } else {
        // BM rows over PWAVES = 8 waves: a wave's rows are all requested before the first is reduced
        constexpr int RPW = (BM + PWAVES - 1) / PWAVES;
        bf16x8 pv[RPW][2], rs[RPW][2];
        if (np.prev != nullptr) {                            // launch-uniform, outside the loops: no load under a per-element condition
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int row = wave + q * PWAVES;
                if (row < rows) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int c = min(lane + i * 64, nchunk - 1);
                        pv[q][i] = *reinterpret_cast<const bf16x8*>(np.prev + (int64_t)(m0 + row) * K + c * 8);
                        rs[q][i] = *reinterpret_cast<const bf16x8*>(np.res_in + (int64_t)(m0 + row) * K + c * 8);
                    }
                }
            }
        } else {                                             // no previous projection (the first layer): 0 + r = r exactly
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int row = wave + q * PWAVES;
                if (row < rows) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int c = min(lane + i * 64, nchunk - 1);
                        pv[q][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        rs[q][i] = *reinterpret_cast<const bf16x8*>(np.res_in + (int64_t)(m0 + row) * K + c * 8);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int row = wave + q * PWAVES;
            if (row < rows) {
                float xv[2][8];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[i][j] = bf2f(pv[q][i][j]);
                finish_row(row, xv, rs[q]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's rows are in the panel before it ends or meets the first barrier
    if (wave >= NW) return;                                // helper waves (a wave that has ended counts as arrived at every barrier)

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // the loader waves pass it once the whole slice has landed
    asm volatile("" ::: "memory");
#pragma unroll 4
    for (int t = 0; t < ntiles; ++t) {
        const char* a_base = smem + t * A_BYTES;
        const char* b_base = panel + t * (BM * 128);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TN], bfr[TM];
            const int kc = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * WN + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(a_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = min(wm * WM + j * 16 + fr, rows - 1);    // rows past the batch shadow its last row
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_base + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    if constexpr (SWIGLU && TN == 1) {
        // Two waves per 32-row block of the interleaved weight: wave 0 holds the 16 gate columns, wave 1 the matching up
        // columns (half the chain of dependent MFMAs per wave); the up values cross to wave 0 through LDS as bf16, the
        // rounding point of the separate kernels.  The weight slice is no longer read: its first bytes carry them.
        static_assert(WAVES_N == 2 && WAVES_M == 1 && TM == 1, "gate wave + up wave");
        bf16x4* cross = reinterpret_cast<bf16x4*>(smem);
        __builtin_amdgcn_s_barrier();                              // both waves are done with the slice
        if (wn == 1) {
            bf16x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = f2bf(acc[0][0][e]);
            cross[lane] = u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wn == 0) {
            const bf16x4 u = cross[lane];
            const int n = n0;                                     // first interleaved row of the gate block
            const int col = (n >> 5) * 16 + fq * 4;
            const int m = m0 + fr;
            if (n < N && m < M) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = bf2f(f2bf(acc[0][0][e]));
                    const float uv = bf2f(u[e]);
                    const float sl = gv / (1.0f + expf(-gv));
                    o[e] = f2bf(bf2f(f2bf(sl)) * uv);
                }
                *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + col) = o;
            }
        }
    } else if constexpr (SWIGLU) {
#pragma unroll
        for (int i = 0; i + 1 < TN; i += 2) {
            const int n = n0 + wn * WN + i * 16;
            if (n >= N) continue;
            const int col = (n >> 5) * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * WM + j * 16 + fr;
                if (m >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = bf2f(f2bf(acc[i][j][e]));
                    const float uv = bf2f(f2bf(acc[i + 1][j][e]));
                    const float sl = gv / (1.0f + expf(-gv));
                    o[e] = f2bf(bf2f(f2bf(sl)) * uv);
                }
                *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + col) = o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * WN + i * 16 + fq * 4;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * WM + j * 16 + fr;
                if (m >= M) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[i][j][e]);
                *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + n) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ narrow outputs over a long K, decode batches
// The O and down projections of a decode step: a few rows, 1024 output columns, K of 2048..3072.  Tiles of 32 columns
// give 32 workgroups, so the generic kernel cuts K into fp32 slabs (more workgroups) and leaves their sum to the
// consumer - which then cannot be a kernel that re-derives its input per workgroup.  Here a workgroup owns 16 output
// columns over the WHOLE K and its KW waves split K between them: wave w requests its own k-tiles of the 16 weight
// rows (LDS-DMA, 128-B rows, the ring's swizzle) and the matching x fragments (global -> registers) up front, waits once,
// multiplies, and the partial tiles meet in LDS: ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)), one bf16 rounding.
// No barrier until that sum: a wave reads only what it requested itself.
template <int KW, int COLS = 16>
__global__ __launch_bounds__(64 * KW) void gemm_kwaves_bf16_nt_kernel(
    const bf16* __restrict__ x, int64_t x_stride, const bf16* __restrict__ w, const bf16* __restrict__ bias,
    bf16* __restrict__ y, int64_t y_stride, int M, int N, int K) {
    static_assert(KW == 4 || KW == 8, "four or eight waves over K");
    // COLS = 8: half a MFMA tile of output columns per workgroup (the other eight operand rows repeat them and their
    // results are dropped) - twice the workgroups when 16-column ones leave most CUs without one, and one CU takes in
    // only ~35-40 GB/s from HBM
    static_assert(COLS == 16 || COLS == 8, "16 or 8 output columns");
    constexpr int TILE = COLS * 128;                      // bytes of one k-tile of the workgroup's weight rows
    constexpr int MAXT = 9;                               // k-tiles per wave (K <= 64 * KW * MAXT)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int TM = 1;                                 // one 16-row tile of the batch per workgroup (blockIdx.y)
    const int n0 = blockIdx.x * COLS, m0 = blockIdx.y * 16;
    const int ntiles = K / BK;
    const int per = (ntiles + KW - 1) / KW;               // <= MAXT (checked on the host)
    const int t0 = wave * per, t1 = min(ntiles, t0 + per);
    char* mine = smem + wave * (per * TILE);              // this wave's k-tiles: [tile][COLS rows][128 B]

    const int dma_r = lane >> 3, dma_pos = lane & 7;
    const bf16* src[COLS / 8];
#pragma unroll
    for (int p = 0; p < COLS / 8; ++p) {
        const int r = p * 8 + dma_r;
        src[p] = w + (int64_t)min(n0 + r, N - 1) * K + (dma_pos ^ ((r >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        if (t0 + i < t1) {
#pragma unroll
            for (int p = 0; p < COLS / 8; ++p)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src[p] + (t0 + i) * BK), (LDS_AS void*)(mine + i * TILE + p * 1024), 16, 0, 2);
        }
    }
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 xf[TM][MAXT][2];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const bf16* xrow = x + (int64_t)min(m0 + j * 16 + fr, M - 1) * x_stride + fq * 8;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            if (t0 + i < t1) {
                xf[j][i][0] = *reinterpret_cast<const bf16x8*>(xrow + (t0 + i) * BK);
                xf[j][i][1] = *reinterpret_cast<const bf16x8*>(xrow + (t0 + i) * BK + 32);
            }
        }
    }
    f32x4 acc[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        if (t0 + i < t1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kc = ks * 4 + fq;
                const int ar = fr & (COLS - 1);             // (COLS = 8: operand rows 8..15 repeat rows 0..7)
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(mine + i * TILE + ar * 128 + ((kc ^ ((ar >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xf[j][i][ks], acc[j], 0, 0, 0);
            }
        }
    }
    // the partial tiles meet in LDS (past the k-tiles: other waves may still be reading theirs)
    f32x4* part = reinterpret_cast<f32x4*>(smem + KW * per * TILE);          // [KW][TM][64]
#pragma unroll
    for (int j = 0; j < TM; ++j) part[(wave * TM + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (wave >= TM) return;
    const int j = wave;                                   // wave j finishes row tile j
    f32x4 s[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) s[k] = part[(k * TM + j) * 64 + lane];
    f32x4 tot;
    if constexpr (KW == 8) tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    else tot = (s[0] + s[1]) + (s[2] + s[3]);
    const int n = n0 + fq * 4, m = m0 + j * 16 + fr;
    if (n < N && m < M && fq * 4 < COLS) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(tot[e] + (bias ? bf2f(bias[n + e]) : 0.f));
        *reinterpret_cast<bf16x4*>(y + (int64_t)m * y_stride + n) = o;
    }
}

namespace {
struct GemmArgs {
    const bf16 *x, *w, *bias;
    bf16* y;
    float* partials;
    int64_t xs, ys;
    int M, N, K, split_k;
    hipStream_t st;
    bool swiglu;
    const QkvEpilogue* qkv = nullptr;   // set: the QKV epilogue (128x128 tiles only)
    const SampleEpilogue* sample = nullptr;   // set: the sampling epilogue (256x256 tiles only)
    char* plan = nullptr;               // set: launch nothing, append the name of every launch that would be made (nvh_gemm_plan)
    size_t plan_bytes = 0;
    RowScale rowscale = RowScale{nullptr, 0, 0.f, 0.f};   // QKV / SwiGLU epilogues: RMSNorm as a row scale
    bf16* residual = nullptr;           // set (with row_sumsq): add the product to this [M, N] residual in place (two-group 64 x 64 tiles only)
    float* row_sumsq = nullptr;
    mutable int32_t status = NVH_OK;    // set by a launcher that cannot serve the request (row scale on a tile without that epilogue)
};

// nvh_gemm_plan: one entry per launch, "; " between them
void plan_note(const GemmArgs& a, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void plan_note(const GemmArgs& a, const char* fmt, ...) {
    const size_t used = strlen(a.plan);
    if (used + 3 >= a.plan_bytes) return;
    size_t at = used;
    if (used) { a.plan[at++] = ';'; a.plan[at++] = ' '; a.plan[at] = 0; }
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(a.plan + at, a.plan_bytes - at, fmt, ap);
    va_end(ap);
}

template <int BM, int BN, int WAVES_N, int WAVES_M, int STAGES, bool NT_W, bool SWIGLU, bool QKV = false, bool SAMPLE = false>
void launch_gemm_nt(const GemmArgs& a) {
    constexpr int smem = STAGES * (BM + BN) * 128;
    static_assert(smem <= 160 * 1024, "LDS");
    static bool configured = false;
    auto kern = gemm_bf16_nt_kernel<BM, BN, WAVES_N, WAVES_M, STAGES, NT_W, SWIGLU, QKV, SAMPLE, false>;
    // the row-scaled epilogue (RowScale) exists for the tiles the step executor sends it to: eight waves, 128 rows
    constexpr bool HAS_RS = (QKV || SWIGLU) && !NT_W && WAVES_N * WAVES_M == 8 && BM == 128;
    if (a.rowscale.sumsq != nullptr) {
        if constexpr (HAS_RS) kern = gemm_bf16_nt_kernel<BM, BN, WAVES_N, WAVES_M, STAGES, NT_W, SWIGLU, QKV, SAMPLE, true>;
        else {
            a.status = NVH_ENOSYS;
            nvh_set_error("gemm: the row-scaled epilogue does not exist on %d x %d tiles of %d waves (M=%d N=%d K=%d)", BM, BN, WAVES_N * WAVES_M, a.M, a.N, a.K);
            return;
        }
    }
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<BM, BN, WAVES_N, WAVES_M, STAGES, NT_W, SWIGLU, QKV, SAMPLE, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if constexpr (HAS_RS)
            (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<BM, BN, WAVES_N, WAVES_M, STAGES, NT_W, SWIGLU, QKV, SAMPLE, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    if (a.plan) {
        plan_note(a, "tile %dx%d waves %dx%d stages %d%s%s%s%s%s grid %ldx%ldx%d", BM, BN, WAVES_N, WAVES_M, STAGES, NT_W ? " nt" : "", SWIGLU ? " swiglu" : "",
                  QKV ? " qkv" : "", SAMPLE ? " sample" : "", a.partials || a.split_k > 1 ? " slabs" : "", (long)nvh_cdiv(a.N, BN), (long)nvh_cdiv(a.M, BM), a.split_k);
        return;
    }
    const int ktiles = a.K / BK;
    const bool no_wide = nvh_switch_on(SW_GEMM_NO_WIDE_STORE);   // A/B switch for measurements
    // wide stores need 16-byte aligned output rows and the staging tile inside the kernel's LDS
    const bool wide = !no_wide && a.ys % 8 == 0 && nvh_aligned16(a.y) && smem >= BM * 136 * 2;
    const int k_per_split = (int)nvh_cdiv(ktiles, a.split_k) * BK;
    dim3 grid((unsigned)nvh_cdiv(a.N, BN), (unsigned)nvh_cdiv(a.M, BM), (unsigned)a.split_k);
    const bool no_xcd = nvh_switch_on(SW_GEMM_NO_XCD);            // A/B switch for measurements
    // prefill-sized products only: at 256 rows (a decode batch, 64-row tiles) the same order made a step 0.7 % slower
    const int xcd_strip = (no_xcd || grid.y * grid.z == 1 || a.M < 512) ? 0 : grid.x % 8 == 0 ? 8 : grid.x % 4 == 0 ? 4 : 0;
    kern<<<grid, 64 * WAVES_N * WAVES_M, smem, a.st>>>(a.x, a.xs, a.w, a.bias, a.y, a.ys, a.partials, a.M, a.N, a.K, k_per_split,
                                                       a.qkv ? *a.qkv : QkvEpilogue{}, wide, a.sample ? *a.sample : SampleEpilogue{}, xcd_strip, a.rowscale);
}

void launch_gemm_kgroups(const GemmArgs& a) {
    constexpr int STAGES = 3, smem = STAGES * 2 * (64 + 64) * 128;
    static bool configured = false;
    auto kern = a.residual ? gemm_kgroups_bf16_nt_kernel<STAGES, true> : gemm_kgroups_bf16_nt_kernel<STAGES, false>;
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)gemm_kgroups_bf16_nt_kernel<STAGES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)gemm_kgroups_bf16_nt_kernel<STAGES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    if (a.plan) {
        plan_note(a, "kgroups 64x64 stages %d%s grid %ldx%ld", STAGES, a.residual ? " residual" : "", (long)nvh_cdiv(a.N, 64), (long)nvh_cdiv(a.M, 64));
        return;
    }
    dim3 grid((unsigned)nvh_cdiv(a.N, 64), (unsigned)nvh_cdiv(a.M, 64), 1);
    const bool no_xcd = nvh_switch_on(SW_GEMM_NO_XCD);
    const int xcd_strip = (no_xcd || grid.y == 1 || a.M < 512) ? 0 : grid.x % 8 == 0 ? 8 : grid.x % 4 == 0 ? 4 : 0;   // as launch_gemm_nt
    kern<<<grid, 512, smem, a.st>>>(a.x, a.xs, a.w, a.bias, a.y, a.ys, a.M, a.N, a.K, xcd_strip, a.residual, a.row_sumsq);
}

template <int BM, int BN, int WAVES_N, int WAVES_M, int STAGES>
void launch_gemm(const GemmArgs& a) {
    const bool no_nt = nvh_switch_on(SW_GEMM_NO_NT);   // A/B switch for tuning
    const bool nt = a.M <= BM && !no_nt;
    if constexpr (BM == 128 && BN == 128 && WAVES_N == 2 && (WAVES_M == 2 || WAVES_M == 4)) {
        if (a.qkv) {
            if (nt) launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, true, false, true>(a);
            else launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, false, false, true>(a);
            return;
        }
    }
    if constexpr ((BM == 256 && BN == 256) || (BM == 128 && BN == 128 && WAVES_N == 2 && WAVES_M == 2 && STAGES == 2)) {
        if (a.sample) {
            launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, false, false, false, true>(a);
            return;
        }
    }
    if (a.swiglu) {
        if (nt) launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, true, true>(a);
        else launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, false, true>(a);
    } else {
        if (nt) launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, true, false>(a);
        else launch_gemm_nt<BM, BN, WAVES_N, WAVES_M, STAGES, false, false>(a);
    }
}

// Tile choice: the largest tile whose grid (x split_k) still gives every CU a workgroup.
enum Tile { T16x32, T32x32, T16x128, T32x128, T64x128, T64x64, T128x64, T128x128, T256x128, T256x256 };
Tile pick_tile(int64_t M, int64_t N, int split_k) {
    // decode batches: 128-row weight tiles leave most CUs idle unless the output is very wide; 32-row tiles in
    // two-wave workgroups (three fit a CU) keep the weight stream spread over the chip
    const bool no_skinny = nvh_switch_on(SW_GEMM_NO_SKINNY);
    const bool narrow = nvh_cdiv(N, 128) * split_k < 384 && !no_skinny;
    if (M <= 16) return narrow ? T16x32 : T16x128;
    const bool m32_skinny = nvh_switch_on(SW_GEMM_M32_SKINNY);   // tuning switch
    if (M <= 32) return (narrow && (split_k == 1 || m32_skinny)) ? T32x32 : T32x128;      // with split-K the 128-row tile measured faster
    auto wgs = [&](int bm, int bn) { return nvh_cdiv(M, bm) * nvh_cdiv(N, bn) * split_k; };
    const int force = (int)nvh_switch(SW_GEMM_FORCE_TILE);   // tuning switch
    if (force >= T16x128 && force <= T256x256) return (Tile)force;
    if (M <= 64) return nvh_cdiv(N, 128) * split_k >= 384 ? T64x128 : T64x64;   // 33..64 rows: enough workgroups first
    const bool no256 = nvh_switch_on(SW_GEMM_NO_256);
    // big products (prefill of the 14B / 32B shapes, square GEMMs from 4096 up, the LM head from 256 rows): 256 x 256
    // tiles in eight-wave workgroups once they fill the chip and few of their rows are padding (4096^3: 985 -> 1196
    // TFLOP/s; 1024 x 34816 x 5120: 705 -> 877; LM head at 256 / 512 rows: 132 -> 118 / 244 -> 215 us; 128 tiles: 983 -> 710)
    // ... also for K-slabs when tiles x slabs make ONE nearly full round (Qwen3-14B o / down at 1024 rows: 80 tiles x 3
    // slabs, 73 -> 55 us and 188 -> 157 us against 128-row tiles with the best slab count)
    // Rows: up to one row tile, at most 1/8 of it padding; beyond that the 256-row tiles never compute more than 4/3 of the
    // rows 128-row tiles would, and a 256 x 256 tile costs about three 128 x 128 ones (LM head, 151936 x 1024, at
    // 300 / 520 / 850 rows: 193 / 283 / 402 us against 209 / 338 / 484 us)
    const bool rows_ok = (M <= 256 || split_k > 1) ? nvh_cdiv(M, 256) * 256 * 8 <= M * 9 : true;   // (K-slabs: measured on full row tiles only)
    if (!no256 && wgs(256, 256) >= 240 && (split_k == 1 || wgs(256, 256) <= 256) && rows_ok) return T256x256;
    // very wide outputs with few rows: all rows in one tile, so every weight row is fetched once; three stages, one
    // workgroup per CU
    if (!no256 && M >= 256 && M <= 512 && wgs(256, 128) >= 768) return T256x128;
    if (wgs(128, 128) >= 200) return T128x128;
    if (wgs(128, 64) >= 200) return T128x64;
    return T64x64;
}
int tile_bm(Tile t) { return (t == T16x128 || t == T16x32) ? 16 : (t == T32x128 || t == T32x32) ? 32 : (t == T64x128 || t == T64x64) ? 64 : (t == T256x128 || t == T256x256) ? 256 : 128; }
int tile_bn(Tile t) { return (t == T16x32 || t == T32x32) ? 32 : (t == T64x64 || t == T128x64) ? 64 : t == T256x256 ? 256 : 128; }
}  // namespace
static int32_t dispatch_gemm(const GemmArgs& a);

extern "C" int64_t nvh_gemm_workspace_bytes(int64_t M, int64_t N, int32_t split_k) {
    return split_k > 1 ? (int64_t)split_k * M * N * (int64_t)sizeof(float) : 0;
}

namespace {
// Workgroups of a tile that fit one CU (LDS: stages x (BM + BN) x 128 B of 160 KB).
int tile_residents(Tile t) {
    switch (t) {
        case T16x32: case T32x32: return 3;
        case T16x128: case T32x128: case T64x64: case T128x128: return 2;
        default: return 1;                                     // T64x128, T128x64, T256x128, T256x256
    }
}
// Time model of one GEMM launch with `s` K-slabs, in microseconds.  These kernels are bound by the L2 -> LDS fill
// (about 50 GB/s per CU, shared by the workgroups resident on it), so a round of workgroups costs
// (k-tiles per slab + fixed part) x tile bytes x residents / fill rate, and a grid takes a whole number of rounds:
// 320 workgroups on 256 slots cost two rounds.  Slabs add their write and the consumer's read.  It ranks the
// measured split counts of the 14B projections at 256..1024 rows in the measured order.
double splitk_estimate_us(int64_t M, int64_t N, int64_t K, int s) {
    const Tile t = pick_tile(M, N, s);
    const int64_t wgs = nvh_cdiv(M, tile_bm(t)) * nvh_cdiv(N, tile_bn(t)) * s;
    int64_t res = nvh_cdiv(wgs, 256);
    if (res > tile_residents(t)) res = tile_residents(t);
    const int64_t rounds = nvh_cdiv(wgs, 256 * res);
    const double ktiles = (double)nvh_cdiv(nvh_cdiv(K, BK), s) + 6.0;
    const double tile_bytes = (double)(tile_bm(t) + tile_bn(t)) * 128.0;
    double us = (double)rounds * ktiles * tile_bytes * (double)res / 50000.0;
    if (s > 1) us += 2.0 * (double)s * (double)M * (double)N * 4.0 / 5.0e6;
    return us;
}
}  // namespace

extern "C" int32_t nvh_gemm_suggest_split_k(int64_t M, int64_t N, int64_t K) {
    // at least 4 k-tiles (256 k) per slab, at most 16 slabs
    const int64_t kmax = K / 256 > 1 ? K / 256 : 1;
    Tile t = pick_tile(M, N, 1);
    if (t == T16x32) {
        // two-wave decode tiles: three workgroups fit a CU (768 slots).  Stay inside ONE round of workgroups: a grid
        // a little over a multiple of 768 pays a whole workgroup lifetime for its last few tiles (160 tiles: 3 slabs
        // 10.5 us, 4 slabs 11.1 us, 5 slabs = 800 workgroups 13.3 us).
        const int64_t tiles = nvh_cdiv(N, 32) * nvh_cdiv(M, 16);
        if (tiles >= 512) return 1;
        int64_t s = (768 * 4 / 5) / tiles;
        if (s > kmax) s = kmax;
        if (s > 16) s = 16;
        return (int32_t)(s < 1 ? 1 : s);
    }
    // everything else: the slab count the time model likes best; a larger count has to win by 5 %
    int best = 1;
    double best_us = splitk_estimate_us(M, N, K, 1);
    for (int s = 2; s <= 8 && s <= kmax; ++s) {
        const double us = splitk_estimate_us(M, N, K, s);
        if (us < 0.95 * best_us) {
            best = s;
            best_us = us;
        }
    }
    return best;
}

extern "C" int32_t nvh_gemm_bf16_nt(const void* x, int64_t x_row_stride, const void* w, const void* bias, void* y,
                                    int64_t y_row_stride, int64_t M, int64_t N, int64_t K, int32_t split_k,
                                    float* workspace, nvh_stream_t stream) {
    NVH_REQUIRE(x && w && (y || workspace), NVH_EINVAL, "nvh_gemm_bf16_nt: null pointer");
    NVH_REQUIRE(K > 0 && K % BK == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: K=%ld must be a multiple of 64", (long)K);
    NVH_REQUIRE(N > 0 && N % 4 == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: N=%ld must be a multiple of 4", (long)N);
    NVH_REQUIRE(M >= 0 && M < (1 << 30) && N < (1 << 30), NVH_EINVAL, "nvh_gemm_bf16_nt: M/N out of range");
    NVH_REQUIRE(x_row_stride % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(w), NVH_EINVAL, "nvh_gemm_bf16_nt: x/w rows must be 16-byte aligned");
    NVH_REQUIRE(split_k >= 1 && split_k <= 64, NVH_EINVAL, "nvh_gemm_bf16_nt: split_k %d out of range", split_k);
    if (split_k > 1) {
        NVH_REQUIRE(workspace && nvh_aligned16(workspace), NVH_EINVAL, "nvh_gemm_bf16_nt: split_k needs a 16-byte aligned workspace");
        NVH_REQUIRE(!bias, NVH_EINVAL, "nvh_gemm_bf16_nt: bias is applied by the split-K consumer");
        NVH_REQUIRE(K / BK >= split_k, NVH_EINVAL, "nvh_gemm_bf16_nt: split_k %d exceeds the %ld k-tiles", split_k, (long)(K / BK));
    } else {
        NVH_REQUIRE(y && y_row_stride % 4 == 0 && ((uintptr_t)y & 7) == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: y rows must be 8-byte aligned");
        NVH_REQUIRE(!bias || ((uintptr_t)bias & 7) == 0, NVH_EINVAL, "nvh_gemm_bf16_nt: bias must be 8-byte aligned");
    }
    if (M == 0) return NVH_OK;
    GemmArgs a{(const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)y, split_k > 1 ? workspace : nullptr,
               x_row_stride, y_row_stride, (int)M, (int)N, (int)K, split_k, (hipStream_t)stream, false};
    return dispatch_gemm(a);
}

// Decode batches and an output narrow enough that 16-column workgroups (x 16-row tiles of the batch) stay within about
// a round and a half of the chip: gemm_kwaves_bf16_nt_kernel.  The O / down projections of a decode step and their
// tensor-parallel shards, which cannot hand K-slabs to a consumer (measured against the generic tiles, N = 1024 / 2048 /
// 5120 at 1..32 rows: 2.2-8.7 us against 2.8-19 us for K = 256..4096; N = 5120 at 32 rows, 640 workgroups: slower).
// Only from K = 1536: below that a gate/up projection could take it, and its sum over K would no longer be the one of the
// SwiGLU kernel (nvh_gemm_swiglu_bf16_nt promises the bits of nvh_gemm_bf16_nt + nvh_silu_mul).
// Shapes that take the two-group 64 x 64 tiles (see gemm_kgroups_bf16_nt_kernel): a plain product whose 64 x 64 grid is
// one nearly full round.
static bool takes_kgroups(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles64 = nvh_cdiv(M, 64) * nvh_cdiv(N, 64);
    return !nvh_switch_on(SW_GEMM_NO_KGROUPS) && M > 64 && K % 128 == 0 && K >= 1024 && tiles64 >= 200 && tiles64 <= 256;
}

static bool takes_kwaves(const GemmArgs& a) {
    const bool off = nvh_switch_on(SW_NO_KWAVES);                   // A/B switch for measurements
    const int kmin = (int)nvh_switch_or(SW_KWAVES_KMIN, 1536);   // tuning switch
    return !off && a.split_k == 1 && !a.swiglu && !a.qkv && !a.sample && a.M <= 64 && a.K >= kmin && a.K <= 64 * 8 * 9 &&
           nvh_cdiv(a.N, 16) * nvh_cdiv(a.M, 16) <= 384;
}
template <int KW, int COLS>
static void launch_kwaves(const GemmArgs& a) {
    constexpr int TM = 1;
    if (a.plan) {
        plan_note(a, "kwaves %d cols %d grid %ldx%ld", KW, COLS, (long)nvh_cdiv(a.N, COLS), (long)nvh_cdiv(a.M, 16));
        return;
    }
    auto kern = gemm_kwaves_bf16_nt_kernel<KW, COLS>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, KW * 9 * COLS * 128 + KW * TM * 1024);
        configured = true;
    }
    const int per = (int)nvh_cdiv(a.K / BK, KW);
    kern<<<dim3((unsigned)nvh_cdiv(a.N, COLS), (unsigned)nvh_cdiv(a.M, 16)), 64 * KW, KW * per * COLS * 128 + KW * TM * 1024, a.st>>>(a.x, a.xs, a.w, a.bias, a.y,
                                                                                                                                   a.ys, a.M, a.N, a.K);
}

static int32_t dispatch_gemm(const GemmArgs& a) {
    if (takes_kwaves(a)) {
        // eight output columns per workgroup while that still leaves CUs without one
        const bool no8 = nvh_switch_on(SW_KWAVES_NO_8);             // A/B switch for measurements
        const bool narrow = !no8 && nvh_cdiv(a.N, 8) * nvh_cdiv(a.M, 16) <= 256;
        if (a.K / BK >= 32) { if (narrow) launch_kwaves<8, 8>(a); else launch_kwaves<8, 16>(a); }
        else { if (narrow) launch_kwaves<4, 8>(a); else launch_kwaves<4, 16>(a); }
        if (!a.plan) NVH_CHECK_LAUNCH("gemm_kwaves_bf16_nt");
        return NVH_OK;
    }
    const int s128 = (int)nvh_switch_or(SW_GEMM_S128, 0);   // tuning switches
    const int s64 = (int)nvh_switch_or(SW_GEMM_S64, 0);
    const Tile tile = pick_tile(a.M, a.N, a.split_k);
    if (a.sample && a.sample->cand_base > 0) {            // the left-over columns of a cut LM head (below): 128 x 128 tiles
        launch_gemm<128, 128, 2, 2, 2>(a);
        if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt");
        return NVH_OK;
    }
    if (tile == T256x256 && a.split_k == 1 && !a.qkv) {
        // One 256 x 256 workgroup fits a CU, so the grid runs in rounds of 256 tiles and a last round with few tiles
        // costs a whole round (Qwen3-14B gate/up at 1024 rows: 544 tiles, 403 us; 512 tiles 300 us).  Columns that make
        // whole rounds go first; a remainder of at most half a round follows as a launch of its own on the tile that
        // fits IT (34 us for the 32 left-over tiles above).
        const bool no_cut = nvh_switch_on(SW_GEMM_NO_ROUND_CUT);       // A/B switch for measurements
        const int64_t tm = nvh_cdiv(a.M, 256), tn = nvh_cdiv(a.N, 256);
        int64_t g = tm, r = 256;
        while (r) { const int64_t t = g % r; g = r; r = t; }                         // gcd(tm, 256)
        const int64_t cols_per_round = 256 / g;                                      // column tiles that complete whole rounds
        const int64_t main_tn = tn / cols_per_round * cols_per_round;
        const int64_t rest = (tn - main_tn) * tm;
        if (!no_cut && main_tn > 0 && rest > 0 && rest <= 128 && !(a.sample && nvh_switch_on(SW_LM_HEAD_NO_CUT))) {
            GemmArgs head = a, tail = a;
            const int64_t n_head = main_tn * 256;
            head.N = (int)n_head;
            tail.N = a.N - (int)n_head;
            tail.w = a.w + n_head * a.K;
            if (a.bias) tail.bias = a.bias + n_head;
            SampleEpilogue se_tail;
            if (a.sample) {
                // the sampling epilogue: the left-over columns leave their candidates (one per 128 columns) in the slots
                // behind the whole rounds' (one per 256 columns); the pick runs over all of them
                se_tail = *a.sample;
                se_tail.col_offset += n_head;
                se_tail.cand_base = (int)main_tn;
                tail.sample = &se_tail;
            } else {
                tail.y = a.y + (a.swiglu ? n_head / 2 : n_head);
            }
            const int32_t rc = dispatch_gemm(head);
            return rc != NVH_OK ? rc : dispatch_gemm(tail);
        }
    }
    if (tile == T128x128 && s128 == 3) { launch_gemm<128, 128, 2, 2, 3>(a); if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt"); return NVH_OK; }
    if (tile == T128x128 && s128 == 4) { launch_gemm<128, 128, 2, 2, 4>(a); if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt"); return NVH_OK; }
    if (tile == T64x64 && s64 == 6) { launch_gemm<64, 64, 2, 2, 6>(a); if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt"); return NVH_OK; }
    if (tile == T64x64 && s64 == 8) { launch_gemm<64, 64, 2, 2, 8>(a); if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt"); return NVH_OK; }
    switch (tile) {
        case T16x32: {
            // outputs a little wider than one round of 32-column workgroups (768 fit the chip): 64-column tiles put
            // them all in flight at once instead of leaving a short second round (Qwen3-14B gate/up: 1088 vs 544)
            const bool no64 = nvh_switch_on(SW_GEMM_NO_16x64);       // A/B switch for measurements
            const int64_t t32 = nvh_cdiv(a.N, 32) * a.split_k, t64 = nvh_cdiv(a.N, 64) * a.split_k;
            const bool wide64 = !no64 && a.M <= 16 && t32 > 768 && t64 <= 768 && (a.swiglu || t32 <= 1024);   // measured: 896 tiles 57.8 -> 49.8 us, 1088 plain 66.7 -> 68.6
            if (a.swiglu) {
                if (wide64) launch_gemm<16, 64, 2, 1, 5>(a);    // each wave: one gate block and its up block
                else launch_gemm<32, 32, 1, 2, 6>(a);           // gate and up of a pair in the same wave; 6 stages = 48 KB: three workgroups per CU
            } else {
                if (wide64) launch_gemm<16, 64, 2, 1, 5>(a);
                else launch_gemm<16, 32, 2, 1, 8>(a);
            }
            break;
        }
        case T32x32:
            if (a.swiglu) launch_gemm<32, 32, 1, 2, 6>(a);
            else launch_gemm<32, 32, 2, 1, 8>(a);
            break;
        case T16x128: launch_gemm<16, 128, 4, 1, 4>(a); break;
        case T32x128: launch_gemm<32, 128, 4, 1, 4>(a); break;
        case T64x128: launch_gemm<64, 128, 2, 2, 4>(a); break;
        case T64x64: {
            if (!a.swiglu && a.qkv == nullptr && a.sample == nullptr && a.split_k == 1 && a.partials == nullptr &&
                (a.y != nullptr || a.residual != nullptr) && takes_kgroups(a.M, a.N, a.K))
                launch_gemm_kgroups(a);
            else launch_gemm<64, 64, 2, 2, 4>(a);
            break;
        }
        case T128x64: launch_gemm<128, 64, 2, 2, 4>(a); break;
        case T128x128: {
            // up to one workgroup per CU: nothing shares the CU's LDS, so a third stage (96 KB) keeps two tiles in
            // flight instead of one (1024 x 4096 x 1024: 19.1 -> 17.2 us); larger grids want two residents of 64 KB
            // a nearly full round of one workgroup per CU over a short K (the QKV projection of a prefill chunk, epilogue
            // included): eight waves (two per SIMD, one barrier apart; the QKV epilogue walks 32 rows per pass).  1024 / 896 x
            // 4096 x 1024: 16.1 / 15.1 -> 14.9 / 14.4 us, a 1024-token chunk 2.89 -> 2.81 ms; K = 2048: 25.5 -> 26.2 us (not taken)
            const int64_t tiles128 = nvh_cdiv(a.M, 128) * nvh_cdiv(a.N, 128) * a.split_k;
            if (!a.swiglu && !nvh_switch_on(SW_QKV_NO_8WAVE) && tiles128 >= 200 && tiles128 <= 256 && a.K / a.split_k <= 1536) launch_gemm<128, 128, 2, 4, 3>(a);
            else if (nvh_cdiv(a.M, 128) * nvh_cdiv(a.N, 128) * a.split_k <= 256) launch_gemm<128, 128, 2, 2, 3>(a);
            // gate/up + SwiGLU over a grid whose last round of 128 x 128 tiles is half empty (1024 x 6144: 384 tiles = 1.5
            // rounds of 256 CUs): 128 x 96 tiles make whole rounds (512).  Taken when rounds x tile area, the narrower tile
            // priced 5 % dearer per column, comes out lower: a 1024-token chunk 2.80 -> 2.77 ms.  (128 x 64 tiles, 768 = three
            // whole rounds: 2.81; a third stage on the 96-wide tile, one resident per CU: 2.95.)
            // ... and a nearly full single round of eight-wave 128 x 192 tiles (1024 x 6144: exactly 256) ahead of both, over a
            // short K like the QKV case above: a 1024-token chunk 2.789 -> 2.771 ms
            else if (a.swiglu && a.N % 192 == 0 && !nvh_switch_on(SW_SWIGLU_NO_192) && a.K <= 1536 && a.split_k == 1 &&
                     nvh_cdiv(a.M, 128) * (a.N / 192) >= 200 && nvh_cdiv(a.M, 128) * (a.N / 192) <= 256)
                launch_gemm<128, 192, 2, 4, 3>(a);
            else if (a.swiglu && a.N % 96 == 0 && !nvh_switch_on(SW_SWIGLU_NO_96) &&
                     nvh_cdiv(nvh_cdiv(a.M, 128) * (a.N / 96) * a.split_k, 256) * 96 * 105 < nvh_cdiv(tiles128, 256) * 128 * 100)
                launch_gemm<128, 96, 1, 4, 2>(a);
            else launch_gemm<128, 128, 2, 2, 2>(a);
            break;
        }
        case T256x128: launch_gemm<256, 128, 2, 2, 3>(a); break;
        case T256x256: launch_gemm<256, 256, 2, 4, 2>(a); break;
    }
    if (!a.plan) NVH_CHECK_LAUNCH("gemm_bf16_nt");
    return NVH_OK;
}

// Which kernel(s) dispatch_gemm launches for a shape, without launching: for tests that pin the tuned shapes.
extern "C" int32_t nvh_gemm_plan(int64_t M, int64_t N, int64_t K, int32_t split_k, int32_t flavor, char* out, int64_t out_bytes) {
    NVH_REQUIRE(out && out_bytes >= 16, NVH_EINVAL, "nvh_gemm_plan: no room for the answer");
    NVH_REQUIRE(M >= 1 && M < (1 << 30) && N >= 4 && N < (1 << 30) && K >= BK && K % BK == 0 && split_k >= 1 && split_k <= 64 && K / BK >= split_k &&
                    flavor >= 0 && flavor <= 3, NVH_EINVAL, "nvh_gemm_plan: shape or flavor out of range");
    out[0] = 0;
    static const QkvEpilogue qe{};
    static const SampleEpilogue se{};
    GemmArgs a{nullptr, nullptr, nullptr, (bf16*)out, nullptr, K, N, (int)M, (int)N, (int)K, split_k, nullptr, flavor == 1};
    a.partials = split_k > 1 ? (float*)out : nullptr;     // (never dereferenced: nothing is launched)
    a.plan = out;
    a.plan_bytes = (size_t)out_bytes;
    if (flavor == 2) {
        // the route nvh_gemm_qkv_rope_store takes: the epilogue inside 128 x 128 tiles of one round, else two launches
        if (!nvh_switch_on(SW_NO_FUSED_QKV) && pick_tile(M, N, 1) == T128x128 && nvh_cdiv(M, 128) * nvh_cdiv(N, 128) <= 256) a.qkv = &qe;
    }
    if (flavor == 3) {
        NVH_REQUIRE(nvh_lm_head_sample_fused(M, N, K), NVH_ENOSYS, "nvh_gemm_plan: this shape does not take the sampling epilogue");
        a.sample = &se;
        a.y = nullptr;
    }
    const int32_t rc = dispatch_gemm(a);
    if (rc == NVH_OK && flavor == 2 && !a.qkv) plan_note(a, "qknorm_rope_store");
    return rc;
}

// ------------------------------------------------------------------ split-K slab reduction
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partials, int num_partials,
                                                            const bf16* __restrict__ bias, bf16* __restrict__ y,
                                                            int64_t y_stride, int64_t M, int N) {
    const int cpr = N >> 2;
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / cpr;
        const int n = (int)(i % cpr) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < num_partials; ++p) a += *reinterpret_cast<const f32x4*>(partials + ((int64_t)p * M + m) * N + n);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(a[e] + (bias ? bf2f(bias[n + e]) : 0.f));
        *reinterpret_cast<bf16x4*>(y + m * y_stride + n) = o;
    }
}

extern "C" int32_t nvh_splitk_reduce(const float* partials, int64_t num_partials, const void* bias, void* y,
                                     int64_t y_row_stride, int64_t M, int64_t N, nvh_stream_t stream) {
    NVH_REQUIRE(partials && y, NVH_EINVAL, "nvh_splitk_reduce: null pointer");
    NVH_REQUIRE(N % 4 == 0 && y_row_stride % 4 == 0 && nvh_aligned16(partials) && ((uintptr_t)y & 7) == 0, NVH_EINVAL,
                "nvh_splitk_reduce: alignment");
    if (M == 0) return NVH_OK;
    const int64_t total = M * (N / 4);
    const unsigned grid = (unsigned)(nvh_cdiv(total, 256) < 2048 ? nvh_cdiv(total, 256) : 2048);
    splitk_reduce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(partials, (int)num_partials, (const bf16*)bias, (bf16*)y,
                                                                y_row_stride, M, (int)N);
    NVH_CHECK_LAUNCH("splitk_reduce");
    return NVH_OK;
}

// y[M, N/2] = silu(x W_gate^T) * (x W_up^T) in one kernel.  `w` holds the N = 2I rows of the merged
// gate/up weight INTERLEAVED in blocks of 16 (rows 32b..32b+15 = gate 16b.., rows 32b+16..32b+31 = up 16b..),
// see nvh_interleave_gate_up.  Same arithmetic and rounding points as nvh_gemm_bf16_nt + nvh_silu_mul.
extern "C" int32_t nvh_gemm_swiglu_bf16_nt(const void* x, int64_t x_row_stride, const void* w, void* y,
                                           int64_t y_row_stride, int64_t M, int64_t N, int64_t K, nvh_stream_t stream) {
    NVH_REQUIRE(x && w && y, NVH_EINVAL, "nvh_gemm_swiglu_bf16_nt: null pointer");
    NVH_REQUIRE(K > 0 && K % BK == 0 && N > 0 && N % 32 == 0, NVH_EINVAL, "nvh_gemm_swiglu_bf16_nt: K %% 64 and N %% 32 required (K=%ld N=%ld)", (long)K, (long)N);
    NVH_REQUIRE(M >= 0 && M < (1 << 30) && N < (1 << 30), NVH_EINVAL, "nvh_gemm_swiglu_bf16_nt: M/N out of range");
    NVH_REQUIRE(x_row_stride % 8 == 0 && y_row_stride % 4 == 0 && nvh_aligned16(x) && nvh_aligned16(w) && ((uintptr_t)y & 7) == 0,
                NVH_EINVAL, "nvh_gemm_swiglu_bf16_nt: alignment");
    if (M == 0) return NVH_OK;
    GemmArgs a{(const bf16*)x, (const bf16*)w, nullptr, (bf16*)y, nullptr, x_row_stride, y_row_stride, (int)M, (int)N, (int)K, 1,
               (hipStream_t)stream, true};
    return dispatch_gemm(a);
}

// qkv[M, (Hq + 2 Hkv) * 128] = x W^T followed by the q/k epilogue (per-head RMSNorm, RoPE) and the KV-cache store of
// rows raw_rows.. - in the GEMM's own epilogue when the shape takes 128 x 128 tiles (one head per tile), otherwise
// as nvh_gemm_bf16_nt + nvh_qknorm_rope_store.  Both routes give the same bits.
static bool qkv_epilogue_in_gemm(int64_t M, int64_t N, int64_t K) {
    return M > 0 && !nvh_switch_on(SW_NO_FUSED_QKV) && K > 0 && K % BK == 0 && M < (1 << 30) && pick_tile(M, N, 1) == T128x128 &&
           nvh_cdiv(M, 128) * nvh_cdiv(N, 128) <= 256;   // one workgroup per CU: the epilogue's registers cost no occupancy (two per CU, the Qwen3-14B shape's 448 tiles: a chunk 27.1 -> 28.9 ms)
}

static int32_t gemm_qkv_rope_store(const void* x, int64_t x_row_stride, const void* w, void* qkv, int64_t qkv_row_stride,
                                   int64_t M, int64_t K, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                   const void* q_norm_weight, const void* k_norm_weight, float eps,
                                   const int64_t* positions, const float* cos_sin, int64_t max_position, void* k_cache,
                                   void* v_cache, const int32_t* slot_mapping, int64_t raw_rows, const RowScale& rs, nvh_stream_t stream) {
    NVH_REQUIRE(x && w && qkv && q_norm_weight && k_norm_weight && positions && cos_sin, NVH_EINVAL, "nvh_gemm_qkv_rope_store: null pointer");
    NVH_REQUIRE(head_dim == 128, NVH_ENOSYS, "nvh_gemm_qkv_rope_store: head_dim %ld unsupported (128 only)", (long)head_dim);
    NVH_REQUIRE(num_q_heads > 0 && num_kv_heads > 0 && raw_rows >= 0 && raw_rows <= M, NVH_EINVAL, "nvh_gemm_qkv_rope_store: heads / raw_rows out of range");
    NVH_REQUIRE((k_cache == nullptr) == (v_cache == nullptr), NVH_EINVAL, "nvh_gemm_qkv_rope_store: k_cache and v_cache must both be set or both null");
    const int64_t N = (num_q_heads + 2 * num_kv_heads) * 128;
    NVH_REQUIRE(qkv_row_stride % 8 == 0 && nvh_aligned16(qkv) && nvh_aligned16(cos_sin) && nvh_aligned16(q_norm_weight) && nvh_aligned16(k_norm_weight) &&
                    (!k_cache || (nvh_aligned16(k_cache) && nvh_aligned16(v_cache))),
                NVH_EINVAL, "nvh_gemm_qkv_rope_store: 16-byte alignment required");
    if (qkv_epilogue_in_gemm(M, N, K) && x_row_stride % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(w)) {
        const bool store = k_cache && slot_mapping;
        QkvEpilogue qe{(const bf16*)q_norm_weight, (const bf16*)k_norm_weight, positions, cos_sin, store ? (bf16*)k_cache : nullptr,
                       store ? (bf16*)v_cache : nullptr, store ? slot_mapping : nullptr, eps, (int)num_q_heads, (int)num_kv_heads, (int)raw_rows};
        GemmArgs a{(const bf16*)x, (const bf16*)w, nullptr, (bf16*)qkv, nullptr, x_row_stride, qkv_row_stride, (int)M, (int)N, (int)K, 1,
                   (hipStream_t)stream, false, &qe};
        a.rowscale = rs;
        const int32_t rc = dispatch_gemm(a);
        return rc != NVH_OK ? rc : a.status;
    }
    NVH_REQUIRE(rs.sumsq == nullptr, NVH_ENOSYS, "nvh_gemm_qkv_rope_store_rowscaled: this shape does not take the epilogue inside the GEMM (M=%ld N=%ld K=%ld)", (long)M, (long)N, (long)K);
    const int32_t rc = nvh_gemm_bf16_nt(x, x_row_stride, w, nullptr, qkv, qkv_row_stride, M, N, K, 1, nullptr, stream);
    if (rc != NVH_OK || M == raw_rows) return rc;
    return nvh_qknorm_rope_store((bf16*)qkv + raw_rows * qkv_row_stride, qkv_row_stride, num_q_heads, num_kv_heads, q_norm_weight, k_norm_weight,
                                 eps, positions + raw_rows, cos_sin, max_position, k_cache, v_cache, slot_mapping ? slot_mapping + raw_rows : nullptr,
                                 M - raw_rows, stream);
}

extern "C" int32_t nvh_gemm_qkv_rope_store(const void* x, int64_t x_row_stride, const void* w, void* qkv, int64_t qkv_row_stride,
                                           int64_t M, int64_t K, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                           const void* q_norm_weight, const void* k_norm_weight, float eps,
                                           const int64_t* positions, const float* cos_sin, int64_t max_position, void* k_cache,
                                           void* v_cache, const int32_t* slot_mapping, int64_t raw_rows, nvh_stream_t stream) {
    return gemm_qkv_rope_store(x, x_row_stride, w, qkv, qkv_row_stride, M, K, num_q_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, eps,
                               positions, cos_sin, max_position, k_cache, v_cache, slot_mapping, raw_rows, RowScale{nullptr, 0, 0.f, 0.f}, stream);
}

// ---- RMSNorm as a row scale (prefill-sized steps): three entry points that together replace
//   add + RMSNorm launch -> projection        by        projection over the un-normalised residual, scaled per row.
// nvh_gemm_residual_rowsq_bf16_nt is the producer: residual += x W^T in place (product rounded to bf16 first, sum in fp32,
// rounded to bf16: the arithmetic of nvh_add_rmsnorm's add) and row_sumsq[m][p], p < parts, = partial sums of the squares of
// the fp32 sums of row m.  The two consumers take `w` with the norm weight folded in (w[n][k] * norm_weight[k], rounded
// to bf16) and multiply row m of the product by 1 / sqrt(sum_p row_sumsq[m][p] / K + norm_eps) before their epilogue.
// Against the separate launches the rounding points move (the normalised row is never rounded to bf16; the folded
// weight is): same tolerance against an fp32 reference, not the same bits.
extern "C" int32_t nvh_gemm_residual_rowsq_parts(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 64 || N % 128 || K % BK || M >= (1 << 30) || N >= (1 << 30)) return 0;
    return pick_tile(M, N, 1) == T64x64 && takes_kgroups(M, N, K) ? (int32_t)(N / 32) : 0;
}

extern "C" int32_t nvh_gemm_residual_rowsq_bf16_nt(const void* x, int64_t x_row_stride, const void* w, void* residual, float* row_sumsq,
                                                   int64_t M, int64_t N, int64_t K, nvh_stream_t stream) {
    NVH_REQUIRE(x && w && residual && row_sumsq, NVH_EINVAL, "nvh_gemm_residual_rowsq_bf16_nt: null pointer");
    NVH_REQUIRE(nvh_gemm_residual_rowsq_parts(M, N, K) > 0, NVH_ENOSYS,
                "nvh_gemm_residual_rowsq_bf16_nt: M=%ld N=%ld K=%ld does not take the two-group 64 x 64 tiles (nvh_gemm_residual_rowsq_parts)", (long)M, (long)N, (long)K);
    NVH_REQUIRE(x_row_stride % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(w) && nvh_aligned16(residual) && nvh_aligned16(row_sumsq), NVH_EINVAL,
                "nvh_gemm_residual_rowsq_bf16_nt: alignment");
    GemmArgs a{(const bf16*)x, (const bf16*)w, nullptr, nullptr, nullptr, x_row_stride, N, (int)M, (int)N, (int)K, 1, (hipStream_t)stream, false};
    a.residual = (bf16*)residual;
    a.row_sumsq = row_sumsq;
    return dispatch_gemm(a);
}

extern "C" int32_t nvh_gemm_qkv_epilogue_in_gemm(int64_t M, int64_t N, int64_t K) { return qkv_epilogue_in_gemm(M, N, K) ? 1 : 0; }

extern "C" int32_t nvh_gemm_qkv_rope_store_rowscaled(const void* x, int64_t x_row_stride, const void* w_folded, void* qkv, int64_t qkv_row_stride,
                                                     int64_t M, int64_t K, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                                     const void* q_norm_weight, const void* k_norm_weight, float eps,
                                                     const int64_t* positions, const float* cos_sin, int64_t max_position, void* k_cache,
                                                     void* v_cache, const int32_t* slot_mapping, int64_t raw_rows, const float* row_sumsq,
                                                     int32_t parts, float norm_eps, nvh_stream_t stream) {
    NVH_REQUIRE(row_sumsq && parts > 0 && parts % 4 == 0 && nvh_aligned16(row_sumsq), NVH_EINVAL, "nvh_gemm_qkv_rope_store_rowscaled: row sums missing or parts %% 4 != 0");
    NVH_REQUIRE(parts <= 32, NVH_ENOSYS, "nvh_gemm_qkv_rope_store_rowscaled: at most 32 partial sums per row (%d)", parts);
    return gemm_qkv_rope_store(x, x_row_stride, w_folded, qkv, qkv_row_stride, M, K, num_q_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, eps,
                               positions, cos_sin, max_position, k_cache, v_cache, slot_mapping, raw_rows,
                               RowScale{row_sumsq, parts, 1.0f / (float)K, norm_eps}, stream);
}

extern "C" int32_t nvh_gemm_swiglu_rowscaled_bf16_nt(const void* x, int64_t x_row_stride, const void* w_folded, void* y, int64_t y_row_stride,
                                                     int64_t M, int64_t N, int64_t K, const float* row_sumsq, int32_t parts, float norm_eps,
                                                     nvh_stream_t stream) {
    NVH_REQUIRE(x && w_folded && y, NVH_EINVAL, "nvh_gemm_swiglu_rowscaled_bf16_nt: null pointer");
    NVH_REQUIRE(row_sumsq && parts > 0 && parts % 4 == 0 && nvh_aligned16(row_sumsq), NVH_EINVAL, "nvh_gemm_swiglu_rowscaled_bf16_nt: row sums missing or parts %% 4 != 0");
    NVH_REQUIRE(parts <= 32, NVH_ENOSYS, "nvh_gemm_swiglu_rowscaled_bf16_nt: at most 32 partial sums per row (%d)", parts);
    NVH_REQUIRE(K > 0 && K % BK == 0 && N > 0 && N % 32 == 0, NVH_EINVAL, "nvh_gemm_swiglu_rowscaled_bf16_nt: K %% 64 and N %% 32 required (K=%ld N=%ld)", (long)K, (long)N);
    NVH_REQUIRE(M >= 0 && M < (1 << 30) && N < (1 << 30), NVH_EINVAL, "nvh_gemm_swiglu_rowscaled_bf16_nt: M/N out of range");
    NVH_REQUIRE(x_row_stride % 8 == 0 && y_row_stride % 4 == 0 && nvh_aligned16(x) && nvh_aligned16(w_folded) && ((uintptr_t)y & 7) == 0,
                NVH_EINVAL, "nvh_gemm_swiglu_rowscaled_bf16_nt: alignment");
    if (M == 0) return NVH_OK;
    GemmArgs a{(const bf16*)x, (const bf16*)w_folded, nullptr, (bf16*)y, nullptr, x_row_stride, y_row_stride, (int)M, (int)N, (int)K, 1,
               (hipStream_t)stream, true};
    a.rowscale = RowScale{row_sumsq, parts, 1.0f / (float)K, norm_eps};
    const int32_t rc = dispatch_gemm(a);
    return rc != NVH_OK ? rc : a.status;
}

// 1 when the row-scaled form of the gate/up + SwiGLU (flavor 1) or of the QKV projection with its epilogue (flavor 2) exists for
// this shape (a dry run of the dispatch: nothing is launched)
extern "C" int32_t nvh_gemm_rowscaled_supported(int64_t M, int64_t N, int64_t K, int32_t flavor) {
    if (M < 1 || M >= (1 << 30) || N < 32 || N >= (1 << 30) || K < BK || K % BK || (flavor != 1 && flavor != 2)) return 0;
    if (flavor == 2 && !qkv_epilogue_in_gemm(M, N, K)) return 0;
    if (flavor == 1 && N % 32) return 0;
    char buf[256];
    buf[0] = 0;
    static const QkvEpilogue qe{};
    static const float one = 1.0f;
    GemmArgs a{nullptr, nullptr, nullptr, (bf16*)buf, nullptr, K, N, (int)M, (int)N, (int)K, 1, nullptr, flavor == 1};
    if (flavor == 2) a.qkv = &qe;
    a.plan = buf;
    a.plan_bytes = sizeof(buf);
    a.rowscale = RowScale{&one, 4, 1.0f, 0.f};          // (never dereferenced: nothing is launched)
    return dispatch_gemm(a) == NVH_OK && a.status == NVH_OK ? 1 : 0;
}

// ---- residual add + RMSNorm + projection in one launch (decode batches)
namespace {
template <int BM, int BN, int WAVES_N, int WAVES_M, int STAGES, bool SWIGLU, int LW = 1>
void launch_gemm_addnorm(const NormPrologue& np, const bf16* w, bf16* y, int64_t ys, int M, int N, int K, hipStream_t st) {
    auto kern = gemm_addnorm_bf16_nt_kernel<BM, BN, WAVES_N, WAVES_M, STAGES, SWIGLU, LW>;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * BN * 128 + BM * 1024 * 2);
        configured = true;
    }
    const int smem = STAGES * BN * 128 + BM * K * 2;
    kern<<<dim3((unsigned)nvh_cdiv(N, BN), (unsigned)nvh_cdiv(M, BM)), 64 * (8 + LW), smem, st>>>(np, w, y, ys, M, N, K);
}
}  // namespace

extern "C" int32_t nvh_add_rmsnorm_gemm_fused(int64_t M, int64_t N, int64_t K, int64_t num_slabs, int32_t swiglu) {
    const bool off = nvh_switch_on(SW_NO_NORM_GEMM);                               // A/B switch for measurements and tests
    // KB a workgroup reads to derive the rows again (every workgroup of the grid does): beyond this the norm launch is cheaper
    const int64_t budget_kb = nvh_switch_or(SW_NORM_GEMM_KB, 128);
    // 1: take it.  2 (33..64 rows; measured 2.6-3.7 % per decode step at 40..64 sequences): take it only where every step
    // of that size does - every row count that changes route changes the order of a few fp32 sums, and steps of 33..64
    // tokens are mostly mixed ones whose neighbours take the generic route; the step executor takes it for pure decode steps.
    const int64_t max_rows = nvh_switch_or(SW_NORM_GEMM_ROWS, 64);   // tuning switch
    if (off || M < 1 || M > 64 || M > max_rows || K < BK || K > 1024 || K % BK != 0 || N < 1 || N >= (1 << 30) || num_slabs < 0 || num_slabs > 64) return 0;
    if (swiglu ? N % 32 != 0 : N % 4 != 0) return 0;
    // one workgroup per CU (its LDS holds the whole weight slice): beyond 32 rows only grids of one round
    if (M > 32 && nvh_cdiv(N, 64) * nvh_cdiv(M, 16) > 256) return 0;
    const int64_t rows = M < 16 ? M : 16;                                          // per workgroup
    const int64_t kb = rows * (num_slabs > 0 ? num_slabs * 4 * K : 2 * K) / 1024;
    if (kb > budget_kb) return 0;
    return M <= 32 ? 1 : 2;
}

extern "C" int32_t nvh_add_rmsnorm_gemm_bf16_nt(const void* prev, const float* slabs, int64_t num_slabs, const void* residual_in,
                                                void* residual_out, const void* norm_weight, float eps, const void* w, void* y,
                                                int64_t y_row_stride, int64_t M, int64_t N, int64_t K, int32_t swiglu,
                                                nvh_stream_t stream) {
    NVH_REQUIRE(residual_in && residual_out && norm_weight && w && y, NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: null pointer");
    NVH_REQUIRE(residual_in != residual_out, NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: residual_out must not alias residual_in");
    NVH_REQUIRE(nvh_aligned16(residual_in) && nvh_aligned16(residual_out) && nvh_aligned16(norm_weight) && nvh_aligned16(w) &&
                    nvh_aligned16(slabs) && nvh_aligned16(prev) && ((uintptr_t)y & 7) == 0 && y_row_stride % 4 == 0,
                NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: alignment");
    NVH_REQUIRE(!slabs || (num_slabs >= 1 && num_slabs <= 64), NVH_EINVAL, "nvh_add_rmsnorm_gemm_bf16_nt: num_slabs %ld out of range", (long)num_slabs);
    NVH_REQUIRE(M >= 1 && M <= 64 && K >= BK && K <= 1024 && K % BK == 0 && N >= 1 && N < (1 << 30) && (swiglu ? N % 32 == 0 : N % 4 == 0),
                NVH_ENOSYS, "nvh_add_rmsnorm_gemm_bf16_nt: %ld x %ld x %ld unsupported (1..64 rows, K a multiple of 64 up to 1024; ask nvh_add_rmsnorm_gemm_fused)",
                (long)M, (long)N, (long)K);
    NormPrologue np{(const bf16*)prev, slabs, (const bf16*)residual_in, (bf16*)residual_out, (const bf16*)norm_weight, eps,
                    slabs ? (int)num_slabs : 0};
    hipStream_t st = (hipStream_t)stream;
    // 16 stages = every k-tile of a 1024-wide row.  16 rows per workgroup (more rows: row tiles, so that no workgroup
    // derives more than 16 rows) while the grid stays within one workgroup per CU (its LDS holds the whole weight
    // slice): 32-column tiles while they fit one round, else 64-column ones fed by two loader waves (the gate/up
    // projection at 17..32 rows, the QKV projection at 33..64)
    const bool wide = nvh_cdiv(N, 32) * nvh_cdiv(M, 16) > 256;
    if (swiglu && wide) launch_gemm_addnorm<16, 64, 2, 1, 16, true, 2>(np, (const bf16*)w, (bf16*)y, y_row_stride, (int)M, (int)N, (int)K, st);
    else if (swiglu) launch_gemm_addnorm<16, 32, 2, 1, 16, true>(np, (const bf16*)w, (bf16*)y, y_row_stride, (int)M, (int)N, (int)K, st);
    else if (wide) launch_gemm_addnorm<16, 64, 2, 1, 16, false, 2>(np, (const bf16*)w, (bf16*)y, y_row_stride, (int)M, (int)N, (int)K, st);
    else launch_gemm_addnorm<16, 32, 2, 1, 16, false>(np, (const bf16*)w, (bf16*)y, y_row_stride, (int)M, (int)N, (int)K, st);
    NVH_CHECK_LAUNCH("gemm_addnorm_bf16_nt");
    return NVH_OK;
}

// ---- LM head + sampler in one GEMM launch (+ the pick among the column tiles)
extern "C" int32_t nvh_lm_head_sample_fused(int64_t M, int64_t N, int64_t K) {
    return M > 0 && M <= 65535 && N > 0 && N % 4 == 0 && N + 0 < (1 << 30) && K > 0 && K % BK == 0 && pick_tile(M, N, 1) == T256x256;
}
// Candidates per row: one per 256 columns of the whole rounds + one per 128 columns of a left-over launch (dispatch_gemm).
static int64_t lm_head_sample_slots(int64_t M, int64_t N) {
    const int64_t tm = nvh_cdiv(M, 256), tn = nvh_cdiv(N, 256);
    int64_t g = tm, r = 256;
    while (r) { const int64_t t = g % r; g = r; r = t; }
    const int64_t cols_per_round = 256 / g, main_tn = tn / cols_per_round * cols_per_round, rest = (tn - main_tn) * tm;
    if (nvh_switch_on(SW_GEMM_NO_ROUND_CUT) || nvh_switch_on(SW_LM_HEAD_NO_CUT) || main_tn == 0 || rest == 0 || rest > 128) return tn;
    return main_tn + nvh_cdiv(N - main_tn * 256, 128);
}
extern "C" int64_t nvh_lm_head_sample_workspace_bytes(int64_t M, int64_t N) { return M * nvh_cdiv(N, 128) * (int64_t)sizeof(ValIdx); }

extern "C" int32_t nvh_lm_head_sample(const void* x, int64_t x_row_stride, const void* w, const float* temperatures, int64_t* out,
                                      int64_t M, int64_t N, int64_t K, uint64_t seed, uint64_t offset, void* workspace,
                                      nvh_stream_t stream) {
    NVH_REQUIRE(x && w && temperatures && out && workspace && nvh_aligned16(workspace), NVH_EINVAL, "nvh_lm_head_sample: null pointer");
    NVH_REQUIRE(x_row_stride % 8 == 0 && nvh_aligned16(x) && nvh_aligned16(w), NVH_EINVAL, "nvh_lm_head_sample: x/w rows must be 16-byte aligned");
    NVH_REQUIRE(nvh_lm_head_sample_fused(M, N, K), NVH_ENOSYS,
                "nvh_lm_head_sample: %ld x %ld x %ld does not take the 256 x 256 tile (ask nvh_lm_head_sample_fused; use nvh_gemm_bf16_nt + nvh_sample)",
                (long)M, (long)N, (long)K);
    const int slots = (int)lm_head_sample_slots(M, N);
    SampleEpilogue se{temperatures, (ValIdx*)workspace, seed, offset, 0, slots, 0};
    GemmArgs a{(const bf16*)x, (const bf16*)w, nullptr, nullptr, nullptr, x_row_stride, 0, (int)M, (int)N, (int)K, 1, (hipStream_t)stream, false};
    a.sample = &se;
    const int32_t rc = dispatch_gemm(a);
    if (rc != NVH_OK) return rc;
    return nvh_launch_sample_final(workspace, out, M, slots, (hipStream_t)stream);
}

// dst rows <- src rows of a merged [gate (I rows) | up (I rows)] weight in the interleaved order above.
__global__ __launch_bounds__(256) void interleave_gate_up_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int inter, int k) {
    const int r = blockIdx.x;                       // destination row
    const int blk = r >> 5, j = r & 31;
    const int64_t srow = j < 16 ? (int64_t)blk * 16 + j : (int64_t)inter + blk * 16 + (j - 16);
    for (int c = threadIdx.x; c < (k >> 3); c += 256)
        *reinterpret_cast<bf16x8*>(dst + (int64_t)r * k + c * 8) = *reinterpret_cast<const bf16x8*>(src + srow * k + c * 8);
}

extern "C" int32_t nvh_interleave_gate_up(const void* w, void* w_interleaved, int64_t inter, int64_t K, nvh_stream_t stream) {
    NVH_REQUIRE(w && w_interleaved && w != w_interleaved, NVH_EINVAL, "nvh_interleave_gate_up: null or aliased pointers");
    NVH_REQUIRE(inter > 0 && inter % 16 == 0 && K % 8 == 0 && nvh_aligned16(w) && nvh_aligned16(w_interleaved), NVH_EINVAL,
                "nvh_interleave_gate_up: inter %% 16, K %% 8 and 16-byte alignment required");
    interleave_gate_up_kernel<<<(unsigned)(2 * inter), 256, 0, (hipStream_t)stream>>>((const bf16*)w, (bf16*)w_interleaved, (int)inter, (int)K);
    NVH_CHECK_LAUNCH("interleave_gate_up");
    return NVH_OK;
}
