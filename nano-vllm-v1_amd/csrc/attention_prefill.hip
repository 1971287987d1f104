// Var-len causal GQA attention over packed new tokens or the paged KV cache (prefill,
// chunked prefill, mixed batches): flash-style online softmax on 32x32x16 bf16 MFMA.
//
// Grid (q blocks, Hkv, B).  A workgroup owns BQ = 32*RB consecutive query rows of one
// sequence for ALL G = Hq/Hkv query heads of one kv head, so a K/V tile staged in LDS
// is shared by the whole GQA group.  One wave = 32 query rows of one query head.
//
// Per 64-key tile:
//   S^T = K . Q^T    A = K rows from LDS (ds_read_b128, XOR-swizzled rows), B = Q^T kept in
//                    registers for the whole kernel; result: key on the register index, query
//                    on the lane -> a query's row max / sum are in-lane plus one lane^32 exchange;
//   P   = exp2(S^T * scale*log2e - m), rounded to bf16: registers 8s..8s+7 of a 32x32 S^T
//         tile ARE the B fragment of k-step s of the next product (no lane movement);
//   O^T += V^T . P^T A = V^T read straight from the row-major V tile with the transposing
//                    LDS read (ds_read_b64_tr_b16), k order matched to the P fragment.
// K/V tiles go global -> LDS by 16-byte LDS-DMA (scalar base + one 32-bit lane offset per piece) into a ring of
// STAGES steps, one barrier per step; the swizzles are applied to the per-lane SOURCE address because the DMA writes
// lane-linear.  Each wave is software-pipelined over the steps: in iteration s it computes the scores of step s + 1
// and then the P.V product of step s, with the softmax of step s + 1 placed between the four MFMA groups of that
// product (MFMA and VALU work of one wave overlap; the O rescale a new maximum asks for waits until the product
// against the old maximum is complete).  So the K half of a slot is filled one iteration ahead of its V half.
// Nothing in the loop may make the compiler wait for vmcnt(0) on its own (that drains the prefetch): Q fragments
// are pinned in registers before the loop, the transposing V reads and the DMA are issued from inline asm, and the
// loop must stay free of register spills.  By counters (1024 queries x 8192 keys, 8 waves per workgroup) a wave
// spends 36 % of its cycles issuing (about 190 vector, 210 scalar, 47 LDS instructions per 32 MFMAs), 36 % in waits
// and barriers and 28 % stalled on dependencies; the matrix pipe is busy 31 % of the time.
// Key groups (KS): the waves of a workgroup can be KS copies of the (query head, row block) set; copy ks takes
// key tiles ks, ks + KS, ... of the range, every step stages KS tiles and the copies merge their (max, sum, O)
// through LDS at the end.  That puts twice the waves on a query block without partial results leaving the CU
// (a 1024-token chunk of 16 heads is only 512 row blocks for 1024 SIMDs).
// Long key ranges can also be cut into `num_splits` ranges of whole tiles (gridDim.z = B * splits):
// each range leaves an unnormalised fp32 partial + (max, sum) and the decode path's combine
// kernel merges them - this is what fills the chip when a chunk has few query blocks and many keys.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define LOG2E 1.4426950408889634f


template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
// Wait until at most `ahead` (<= MAXAHEAD) tiles of PW DMA instructions each are still in flight for this wave.
template <int PW, int MAXAHEAD>
__device__ __forceinline__ void attn_wait_tiles(int ahead) {
    if constexpr (MAXAHEAD == 0) {
        attn_wait_vmcnt<0>();
    } else {
        if (ahead >= MAXAHEAD) attn_wait_vmcnt<PW * MAXAHEAD>();
        else attn_wait_tiles<PW, MAXAHEAD - 1>(ahead);
    }
}

__device__ __forceinline__ int v_off(int row, int ch) {          // V tile: 256-B rows, chunk XOR
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ int k_off(int row, int ch) {          // K tile
    return 256 * row + 16 * (ch ^ (row & 15));
}

// Scheduling request for the region it stands in: four times {one MFMA, then N VALU instructions}.
template <int N>
__device__ __forceinline__ void interleave_mfma_valu() {
    if constexpr (N > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, N, 0);
        }
    }
}

template <int G, int RB, int KS, int STAGES, bool PAGED>
__global__ __launch_bounds__(G * RB * KS * 64, (G * RB * KS <= 4 && STAGES * KS == 2 ? 2 : 1)) void attn_varlen_kernel(
    const bf16* __restrict__ q, int64_t q_stride, const bf16* __restrict__ kp, const bf16* __restrict__ vp,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ tables, int64_t table_stride, int hkv, float scale_log2, bf16* __restrict__ out,
    int num_splits, int tiles_per_split, float* __restrict__ ws_acc, float* __restrict__ ws_ml, int head_major) {
    constexpr int NW1 = G * RB, NW = NW1 * KS, BQ = 32 * RB;     // NW1: waves of one key group
    constexpr int PIECES = 32 * KS;                              // 1-KiB DMA pieces per step: KS x (K + V tile), 4 keys each
    constexpr int PW = (PIECES + NW - 1) / NW;                   // pieces per wave (uniform; extras duplicate)
    constexpr int STAGE = 2 * 64 * 256;                          // K tile + V tile
    static_assert(STAGES == 2 || STAGES == 4, "the stage offset is folded into LDS addresses as a bit field");
    static_assert(STAGES == 2 || 16 % NW == 0, "counted DMA waits need the same number of K and of V pieces in every wave");
    static_assert(KS == 1 || KS == 2, "one or two key groups");
    static_assert(KS == 1 || 32 % NW == 0, "with key groups a wave's piece i must belong to the same tile for every wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES x KS x (K tile + V tile)

    // head_major: the kv head is the fastest grid dimension, so with 8 (4, 2) kv heads the workgroups of one head all land
    // on the same XCD (pair, four) and its L2 fetches that head's K/V once for all of its query blocks
    const int b = blockIdx.z / num_splits, split = blockIdx.z % num_splits, kvh = head_major ? blockIdx.x : blockIdx.y;
    const int qblock = head_major ? blockIdx.y : blockIdx.x, qblocks = head_major ? gridDim.y : gridDim.x;
    const int q_beg = cu_q[b], lq = cu_q[b + 1] - q_beg;
    const int k_beg = cu_k[b], lk = cu_k[b + 1] - k_beg;
    const int qb0 = (qblocks - 1 - qblock) * BQ;           // last (longest, under the causal mask) query blocks first
    if (qb0 >= lq) return;
    const int hq = hkv * G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks_id = wave / NW1, w1 = wave % NW1;               // key group, wave inside the group
    const int g = w1 / RB, rb = w1 % RB;
    const int h = kvh * G + g;
    const int ql = lane & 31, hh = lane >> 5;
    const int shift = lk - lq;                                    // causal diagonal offset
    const int qrow_raw = qb0 + rb * 32 + ql;
    const int qrow = min(qrow_raw, lq - 1);

    // key tiles of this workgroup: [kt_beg, kt_end) within the tiles its query block can see
    const int wg_last_key = min(qb0 + BQ, lq) - 1 + shift;
    const int kt_beg = split * tiles_per_split;
    const int kt_end = min(wg_last_key / 64 + 1, kt_beg + tiles_per_split);
    const int wave_first_row = min(qb0 + rb * 32, lq - 1);
    const int wave_last_key = min(qb0 + rb * 32 + 31, lq - 1) + shift;

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (kt_beg < kt_end) {
        bf16x8 qf[8];                                            // Q^T fragments, filled below
        // LDS-DMA staging: piece p < 16 -> K keys 4p..4p+3, p >= 16 -> V keys; lane -> (key = 4p + l4, chunk position).
        // The swizzled source chunk is (lane term) ^ (piece term) for both tiles, so three lane registers and a few
        // wave-uniform values describe every piece; a tile adds one wave-uniform offset.
        const int l4 = lane >> 4, pos = lane & 15;
        const int64_t row_stride = PAGED ? (int64_t)hkv * 128 : kv_stride;       // elements between consecutive keys
        const int dma_row = l4 * (int)row_stride + kvh * 128;                    // lane's key row inside a 4-key piece
        const int dma_ck = (pos ^ l4) * 8, dma_cv = (pos ^ (l4 << 2)) * 8;       // K: pos ^ (key & 15); V: pos ^ ((key&3)<<2 | (key>>2)&3)
        // page ids, fetched an iteration ahead of their use so the scalar-load latency is never exposed
        const int32_t* table_row = PAGED ? tables + (int64_t)b * table_stride : nullptr;
        const int last_page = PAGED ? max(lk - 1, 0) >> 8 : 0;
        const int last_tile = max(lk - 1, 0) >> 6;
        const int nsteps = (kt_end - kt_beg + KS - 1) / KS;
        auto page_of_step = [&](int st) -> int {
            if constexpr (PAGED) return table_row[min((kt_beg + max(st, 0) * KS) >> 2, last_page)];
            return 0;
        };
        // Step s = key tiles kt_beg + s*KS + j, j < KS (kt_beg is a multiple of KS and KS divides the 4 tiles of a page, so
        // the tiles of a step share one page-table entry).  Slot (s % 2) * KS + j holds [K tile | V tile] of that tile, but the
        // two halves are filled one iteration apart: a wave multiplies P(s) V(s) while it computes the scores of step s + 1,
        // so iteration s reads V(s) and K(s+1) and the DMA issued in it brings K(s+2) and V(s+1) into the halves that were
        // read last in iteration s - 1.  A tile past the last key of the sequence (odd tile count) re-reads the last valid
        // tile; steps past the range are skipped (every wait is vmcnt(0), so the DMA count need not be uniform).
        // One 1-KiB piece (4 keys x 256 B): wave-uniform 64-bit base in SGPRs + one 32-bit byte offset per lane, LDS
        // destination in M0 (saved and restored: the register belongs to the compiler).  Not counted by the compiler's
        // own vmcnt bookkeeping - every wait on these is explicit.
        auto dma16 = [&](const char* sbase, uint32_t lane_bytes, uint32_t lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_bytes), "s"(sbase), "s"(lds_dst) : "memory");
        };
        auto tile_base = [&](const bf16* ptr, int kt, int page) -> const char* {
            int64_t tile_off;                                    // wave-uniform element offset of key kt*64
            if constexpr (PAGED) tile_off = ((int64_t)page * NVH_KV_PAGE + (kt & 3) * 64) * row_stride;
            else tile_off = (int64_t)(k_beg + kt * 64) * row_stride;
            return reinterpret_cast<const char*>(ptr + tile_off);
        };
        // With 4, 8 or 16 waves piece i of a wave (p = i * NW + wave) belongs to the same tile and half for every wave, its
        // keys are 4 * ((i * NW) % 16 + wave) .., and (p & 3) == (wave & 3): the lane part of the address is one register
        // per half and everything else is a handful of scalar adds per piece.
        constexpr bool FAST_DMA = (NW % 4 == 0) && (16 % NW == 0);
        const uint32_t lane_k = 2u * (uint32_t)(dma_row + (dma_ck ^ (32 * (wave & 3))));
        const uint32_t lane_v = 2u * (uint32_t)(dma_row + (dma_cv ^ (8 * (wave & 3))));
        const uint32_t wave_piece = 8u * (uint32_t)row_stride * (uint32_t)wave;          // bytes from the tile's first key to this wave's keys
        const uint32_t lds_base = (uint32_t)(uintptr_t)((LDS_AS char*)smem);
        auto stage = [&](int k_step, int page_k, int v_step, int page_v) {
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                // piece p = wave + i * NW of the step: key group p >> 5, K or V half (p >> 4) & 1, keys 4 (p & 15) ..
                int p, isv, pv, jt;
                if constexpr (FAST_DMA) {                         // i alone decides the tile and the half: compile-time constants
                    jt = (i * NW) >> 5, isv = ((i * NW) >> 4) & 1, pv = ((i * NW) & 15) + wave;
                    p = i * NW + wave;
                } else {
                    p = wave + i * NW;
                    if (p >= PIECES) p -= PIECES;
                    jt = p >> 5, isv = (p >> 4) & 1, pv = p & 15;
                }
                const int st = isv ? v_step : k_step;             // wave-uniform
                if (st < 0 || st >= nsteps) continue;
                const int kt_raw = kt_beg + st * KS + jt;
                const int kt = min(kt_raw, last_tile);
                const uint32_t lds_dst = lds_base + (uint32_t)((st & (STAGES - 1)) * (KS * STAGE)) + (uint32_t)p * 1024u;
                const char* tb = tile_base(isv ? vp : kp, kt, isv ? page_v : page_k);
                if constexpr (FAST_DMA) {
                    if (kt_raw * 64 + 63 < lk) {                  // a full tile of the sequence (all but its last one)
                        dma16(tb + (wave_piece + 8u * (uint32_t)row_stride * (uint32_t)((i * NW) & 15)), isv ? lane_v : lane_k, lds_dst);
                        continue;
                    }
                }
                // last, partly filled tile (or a tile past the end, which re-reads the last one): keys past the last key
                // re-read the last valid key (they are masked anyway)
                int off = dma_row + (isv ? (dma_cv ^ (8 * (pv & 3))) : (dma_ck ^ (32 * (pv & 3))));
                const int over = max(kt * 64 + 4 * pv + l4 - (lk - 1), 0);
                off -= over * (int)row_stride;
                const bf16* src = reinterpret_cast<const bf16*>(tb) + (int64_t)(4 * pv) * row_stride + off;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(smem + (lds_dst - lds_base)), 16, 0, 0);
            }
        };

        // LDS read addresses: every fragment read is (lane term ^ wave-uniform term) + (compile-time offset).
        //   K fragment of k-step ks, key t2*32 + ql:   256*key + 16*((2ks + hh) ^ (key & 15))
        //   V^T fragment (dt, t2, sft, half):          v_off(row, ch) with the row/chunk split below
        const int gi = lane >> 4, li = lane & 15;
        const int tr_q = li >> 2, tr_p = li & 3;
        // lane parts; the (ks) / (dt, half) parts and the slot offset are wave-uniform XOR terms applied per tile
        const int kb0 = 256 * ql + ((hh ^ (ql & 15)) << 4);
        const int vb0 = 256 * (4 * (gi >> 1) + tr_q) + 16 * ((2 * (gi & 1) + (tr_p >> 1)) ^ ((tr_q << 2) | (gi >> 1))) + 8 * (tr_p & 1);
        LDS_AS const char* lds = (LDS_AS const char*)smem;

        f32x16 st[2];                                            // scores of the tile whose softmax comes next
        bf16x8 pb[2][2];                                         // P of the tile whose P.V product comes next
        constexpr float RESCALE_LOG2 = 6.0f;

        // ---- S^T = K . Q^T for the two 32-key halves of the tile in `slot`
        auto s_phase = [&](int slot) {
            const int soff = slot * STAGE;
            int kb[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kb[ks] = kb0 ^ ((32 * ks) ^ soff);
            // The fragments of step ks + 1 are requested before the MFMAs of step ks are issued (two register sets), so an
            // LDS round trip hides behind two MFMAs instead of standing between every pair; the two 32-key halves are
            // independent accumulation chains that share the Q fragment.
            // (inline asm with counted waits: the compiler drains the LDS queue - lgkmcnt(0) - at every other step)
            bf16x8 kf[2][2];
            auto request = [&](int ks, bf16x8 (&dst)[2]) {
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(kb[ks]) : "memory");
            };
            request(0, kf[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) {
                    request(ks + 1, kf[(ks + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[ks & 1][0]), "+v"(kf[ks & 1][1])::"memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[ks & 1][0]), "+v"(kf[ks & 1][1])::"memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][0], qf[ks], ks == 0 ? zero : st[0], 0, 0, 0);   // (zero: an inline operand)
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][1], qf[ks], ks == 0 ? zero : st[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- online softmax of st (query on the lane), in three slices so that they can sit between the MFMA groups of
        // the P.V product of the previous tile.  Scores stay unscaled: p = exp2(s * scale*log2e - m), one FMA per element.
        // The running max only moves when some query's max grew by more than 2^RESCALE_LOG2 (below that p <= 2^RESCALE_LOG2
        // is still exact enough in bf16 / fp32); the O rescale it asks for is applied by the caller once the P.V product
        // that is in flight against the old max has finished.
        float alpha = 1.0f, neg_m = 0.f;
        bool resc = false;
        f32x2 psum2 = {0.f, 0.f};
        auto sm_mask = [&](int kt, bool masked) {
            if (masked) {
                const int vis = qrow + shift - kt * 64 - 4 * hh;   // last visible key, relative to the lane's first
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[t2][r] = (t2 * 32 + (r & 3) + 8 * (r >> 2)) > vis ? -INFINITY : st[t2][r];
            }
        };
        auto sm_max = [&]() {
            float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mx0 = fmaxf(mx0, st[0][r]);
                mx1 = fmaxf(mx1, st[1][r]);
            }
            float mx = fmaxf(mx0, mx1);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2;
            resc = __any(mx > m_run + RESCALE_LOG2);
            alpha = 1.0f;
            if (resc) {
                const float m_new = fmaxf(m_run, mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                alpha = __builtin_amdgcn_exp2f(m_run - m_use);
                m_run = m_new;
            }
            neg_m = (m_run == -INFINITY) ? 0.f : -m_run;
            psum2 = f32x2{0.f, 0.f};
        };
        auto sm_exp = [&](int t2, bf16x8 (&pn)[2][2]) {
            const f32x2 sc2 = {scale_log2, scale_log2}, nm2 = {neg_m, neg_m};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 s2 = {st[t2][r], st[t2][r + 1]};
                const f32x2 e2 = __builtin_elementwise_fma(s2, sc2, nm2);
                const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                psum2 += p2;
                pn[t2][r >> 3][r & 7] = f2bf(p2[0]);
                pn[t2][r >> 3][(r & 7) + 1] = f2bf(p2[1]);
            }
        };
        auto sm_finish = [&]() { l_run = l_run * alpha + (psum2[0] + psum2[1]); };
        auto rescale_o = [&]() {
            if (resc) {                                          // wave-uniform, rare
                asm volatile("" : "+v"(alpha));                 // opaque: keeps the 64 multiplies inside this branch
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
        };
        // ---- one 32-column block dt of O^T += V^T . P^T over the 64 keys of the tile in `slot`.  The transposing LDS reads
        // are issued from inline asm: as compiler-visible loads they would each be preceded by a vmcnt(0) (LDS reads are
        // ordered after every LDS-DMA in flight), which drains the prefetched tiles.  lgkmcnt(0) is the only wait (LDS
        // returns in order; over-waiting is safe).
        struct VFrags { bf16x4 lo[4], hi[4]; };
        auto pv_request = [&](int slot, int dt, VFrags& f) {
            const int soff = slot * STAGE;
            const int v0 = vb0 ^ ((16 * ((4 * dt) ^ 0)) ^ soff), v1 = vb0 ^ ((16 * ((4 * dt) ^ 2)) ^ soff);
            asm volatile(
                "ds_read_b64_tr_b16 %0, %8 offset:16384\n\t"
                "ds_read_b64_tr_b16 %1, %9 offset:18432\n\t"
                "ds_read_b64_tr_b16 %2, %8 offset:20480\n\t"
                "ds_read_b64_tr_b16 %3, %9 offset:22528\n\t"
                "ds_read_b64_tr_b16 %4, %8 offset:24576\n\t"
                "ds_read_b64_tr_b16 %5, %9 offset:26624\n\t"
                "ds_read_b64_tr_b16 %6, %8 offset:28672\n\t"
                "ds_read_b64_tr_b16 %7, %9 offset:30720"
                : "=&v"(f.lo[0]), "=&v"(f.hi[0]), "=&v"(f.lo[1]), "=&v"(f.hi[1]), "=&v"(f.lo[2]), "=&v"(f.hi[2]), "=&v"(f.lo[3]), "=&v"(f.hi[3])
                : "v"(v0), "v"(v1)
                : "memory");
        };
        // OUTSTANDING = LDS reads requested after these fragments that may stay in flight (LDS returns in order)
        auto pv_mfma = [&](int dt, VFrags& f, const bf16x8 (&pc)[2][2], auto outstanding) {
            constexpr int OUTSTANDING = decltype(outstanding)::value;
            if constexpr (OUTSTANDING == 8)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3])::"memory");
            __builtin_amdgcn_sched_barrier(0);                   // no MFMA above the wait
#pragma unroll
            for (int u = 0; u < 4; ++u) {                        // u = 2*t2 + sft: keys 32*t2 + 16*sft ..
                const bf16x8 vfrag = __builtin_shufflevector(f.lo[u], f.hi[u], 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag, pc[u >> 1][u & 1], o[dt], 0, 0, 0);
            }
        };
        using Ahead = std::integral_constant<int, 8>;
        using Last = std::integral_constant<int, 0>;

        // this wave's tile of step s: its key tile, whether it exists / is visible, whether it crosses the causal diagonal
        auto tile_of = [&](int s) { return kt_beg + s * KS + ks_id; };
        auto tile_valid = [&](int s) { return s < nsteps && tile_of(s) < kt_end && tile_of(s) * 64 <= wave_last_key; };
        auto tile_masked = [&](int s) { return tile_of(s) * 64 + 63 > wave_first_row + shift; };
        auto slot_of = [&](int s) { return (s & (STAGES - 1)) * KS + ks_id; };
        // DMA ring of STAGES steps: iteration s issues K(s + STAGES) and V(s + STAGES - 1) into the halves read last in
        // iteration s - 1, and needs what iteration s + 1 - STAGES issued (K(s+1), V(s)): the issues of the STAGES - 2
        // iterations in between may stay in flight.  The count only holds while those iterations issued all their pieces;
        // near the end of the range (and for wave counts that split K and V pieces unevenly) the wait is for everything.
        auto wait_for_iteration = [&](int s) {
            if constexpr (STAGES > 2) {
                if (s + STAGES <= nsteps) {
                    attn_wait_vmcnt<PW * (STAGES - 2)>();
                    return;
                }
            }
            attn_wait_vmcnt<0>();
        };

        // ---- prologue: K(0), then what iterations 1 - STAGES .. -1 would have issued; scores and softmax of step 0 with
        // nothing to overlap
        int page_ring[STAGES + 1];                               // pages of steps s + 1 .. s + STAGES + 1 at the top of iteration s
#pragma unroll
        for (int j = 0; j <= STAGES; ++j) page_ring[j] = page_of_step(j);
        stage(0, page_ring[0], -1, 0);
        // Q^T fragments: lane (q = ql, half hh) holds Q[q][16ks + 8hh .. +7].  Loaded AFTER the first DMAs were
        // issued (the latencies overlap) and pinned by an empty asm: from here on the compiler sees registers, not a
        // pending global load, so it does not drain the DMA queue (vmcnt(0)) at their first use in every tile.
        {
            const bf16* qp = q + (int64_t)(q_beg + qrow) * q_stride + (int64_t)h * 128 + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
        }
#pragma unroll
        for (int j = 1; j < STAGES; ++j) stage(j, page_ring[j], j - 1, page_ring[j - 1]);
        if (STAGES > 2 && STAGES <= nsteps) attn_wait_vmcnt<PW * (STAGES - 1)>();      // K(0) landed, full issues behind it
        else attn_wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (page_ring[j] is now the page of step j; at the top of iteration s it is shifted to hold steps s .. s + STAGES)
        if (tile_valid(0)) {
            s_phase(slot_of(0));
            sm_mask(tile_of(0), tile_masked(0));
            sm_max();
            sm_exp(0, pb);
            sm_exp(1, pb);
            sm_finish();
            rescale_o();                                         // (O is still zero: keeps the state consistent)
        }
        // iteration s: P.V of step s from pb, scores + softmax of step s + 1 into pn, which then becomes pb
        for (int s = 0; s < nsteps; ++s) {
            // V(s) and K(s+1) have landed for this wave ...
            wait_for_iteration(s);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // ... for everyone, and the halves read in iteration s-1 are free
            asm volatile("" ::: "memory");
            stage(s + STAGES, page_ring[STAGES], s + STAGES - 1, page_ring[STAGES - 1]);
#pragma unroll
            for (int j = 0; j < STAGES; ++j) page_ring[j] = page_ring[j + 1];
            page_ring[STAGES] = page_of_step(s + STAGES + 1);
            if (!tile_valid(s)) continue;                        // (wave-uniform; the valid steps of a wave are a prefix)
            const int vslot = slot_of(s);
            if (tile_valid(s + 1)) {
                // scores of the next tile, then its softmax in the shadow of this tile's P.V MFMAs
                // (the first V fragments are requested before the score MFMAs and land behind them; every later block is
                // requested before the MFMAs of the block ahead of it)
                VFrags vf[2];
                pv_request(vslot, 0, vf[0]);
                s_phase(slot_of(s + 1));
                bf16x8 pn[2][2];
                sm_mask(tile_of(s + 1), tile_masked(s + 1));
                // each block of 4 MFMAs (one accumulator: every MFMA waits for the one before it) shares its scheduling
                // region with a slice of the softmax, and the scheduler is told to put VALU instructions between the MFMAs:
                // a wave issues in order, so a softmax slice placed after the block would only start once the last
                // MFMA of the chain has been ISSUED, i.e. after three full MFMA latencies with nothing else to do
                __builtin_amdgcn_sched_barrier(0);
                pv_request(vslot, 1, vf[1]);
                pv_mfma(0, vf[0], pb, Ahead{});
                sm_max();
                interleave_mfma_valu<4>();
                __builtin_amdgcn_sched_barrier(0);
                pv_request(vslot, 2, vf[0]);
                pv_mfma(1, vf[1], pb, Ahead{});
                sm_exp(0, pn);
                asm volatile("" : "+v"(pn[0][0]), "+v"(pn[0][1]), "+v"(psum2));     // computed HERE, not sunk below the branches
                interleave_mfma_valu<10>();
                __builtin_amdgcn_sched_barrier(0);
                pv_request(vslot, 3, vf[1]);
                pv_mfma(2, vf[0], pb, Ahead{});
                sm_exp(1, pn);
                asm volatile("" : "+v"(pn[1][0]), "+v"(pn[1][1]), "+v"(psum2));
                interleave_mfma_valu<10>();
                __builtin_amdgcn_sched_barrier(0);
                pv_mfma(3, vf[1], pb, Last{});
                __builtin_amdgcn_sched_barrier(0);
                sm_finish();
                rescale_o();                                     // after the whole product against the old max
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) pb[t2][k2] = pn[t2][k2];
            } else {
                VFrags vf[2];
                pv_request(vslot, 0, vf[0]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (dt < 3) {
                        pv_request(vslot, dt + 1, vf[(dt + 1) & 1]);
                        pv_mfma(dt, vf[dt & 1], pb, Ahead{});
                    } else {
                        pv_mfma(dt, vf[dt & 1], pb, Last{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    if constexpr (KS > 1) {
        // Merge the key groups: group 1 leaves (max, sum, O) of every lane in LDS (value-major, lane-minor: conflict
        // free), group 0 folds it into its own state.  Both states are consistent (O, sum relative to their max).
        __syncthreads();                                        // every wave is done with the staged tiles
        float* mg = reinterpret_cast<float*>(smem) + (size_t)w1 * (66 * 64) + lane;
        if (ks_id == 1) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mg[(dt * 16 + r) * 64] = o[dt][r];
            mg[64 * 64] = m_run;
            mg[65 * 64] = l_run;
        }
        __syncthreads();
        if (ks_id == 1) return;
        const float m_b = mg[64 * 64], l_b = mg[65 * 64];
        const float m_new = fmaxf(m_run, m_b);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float fa = __builtin_amdgcn_exp2f(m_run - m_use), fb = __builtin_amdgcn_exp2f(m_b - m_use);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = o[dt][r] * fa + mg[(dt * 16 + r) * 64] * fb;
        l_run = l_run * fa + l_b * fb;
        m_run = m_new;
    }

    // ---- lane (q, hh) holds O[q][32dt + 8(r>>2) + 4hh + (r&3)]
    l_run += __shfl_xor(l_run, 32, 64);
    if (qrow_raw >= lq) return;
    if (num_splits == 1) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16* op = out + ((int64_t)(q_beg + qrow_raw) * hq + h) * 128 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(o[dt][r4 * 4 + e] * inv);
                *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * r4) = v;
            }
    } else {
        // unnormalised partial + (max, sum) for the combine kernel; a range with no visible key leaves sum 0
        const int64_t slab = ((int64_t)(q_beg + qrow_raw) * hq + h) * num_splits + split;
        if (l_run > 0.f) {
            float* op = ws_acc + slab * 128 + 4 * hh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[dt][r4 * 4 + e];
                    *reinterpret_cast<f32x4*>(op + 32 * dt + 8 * r4) = v;
                }
        }
        if (hh == 0) {
            ws_ml[slab * 2] = m_run;
            ws_ml[slab * 2 + 1] = l_run;
        }
    }
}

namespace {
struct VarlenArgs {
    const bf16 *q, *k, *v;
    int64_t qs, kvs;
    const int32_t *cu_q, *cu_k, *tables;
    int64_t ts;
    int hkv;
    float sl2;
    bf16* out;
    int num_splits, tiles_per_split;
    float *ws_acc, *ws_ml;
    int head_major;
};

template <int G, int RB, int KS, int STAGES>
void launch_varlen(bool paged, dim3 grid, hipStream_t st, const VarlenArgs& a) {
    constexpr int smem = STAGES * KS * 2 * 64 * 256;
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void*)attn_varlen_kernel<G, RB, KS, STAGES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)attn_varlen_kernel<G, RB, KS, STAGES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    if (paged)
        attn_varlen_kernel<G, RB, KS, STAGES, true><<<grid, G * RB * KS * 64, smem, st>>>(a.q, a.qs, a.k, a.v, a.kvs, a.cu_q, a.cu_k, a.tables, a.ts,
                                                                                         a.hkv, a.sl2, a.out, a.num_splits, a.tiles_per_split, a.ws_acc, a.ws_ml, a.head_major);
    else
        attn_varlen_kernel<G, RB, KS, STAGES, false><<<grid, G * RB * KS * 64, smem, st>>>(a.q, a.qs, a.k, a.v, a.kvs, a.cu_q, a.cu_k, a.tables, a.ts,
                                                                                          a.hkv, a.sl2, a.out, a.num_splits, a.tiles_per_split, a.ws_acc, a.ws_ml, a.head_major);
}

// Query rows per workgroup: 4-5 waves per key group (8 for G = 8), each 32 rows of one query head.
int rows_per_block(int g) { return g == 1 ? 128 : g == 2 ? 64 : 32; }
// Key groups per workgroup (1 or 2) for a launch of `wgs` workgroups: two groups (8 waves, one workgroup per CU) while
// the grid leaves CUs without a second workgroup anyway; four-wave workgroups, two per CU, beyond that.
// Measured on 16 q / 8 kv heads (us per launch, one group vs two): Lq = Lk = 1024: 34.6 (29.9 with 4 key ranges and the
// merge launch) vs 28.5; Lq 1024 / Lk 4096 in 2 ranges: 72.1 vs 58.1; 8 x 128 tokens: 10.0 vs 11.1 (nothing to share out);
// 16 queries x 4 sequences over 4096 keys in many ranges: 26.4 vs 34.5.
int key_groups(int g, int64_t wgs, int64_t total_q, int64_t max_seqlen_k) {
    const int force = (int)nvh_switch_or(SW_ATTN_KS, 0);      // A/B switch for measurements
    if (g != 1 && g != 2 && g != 4) return 1;
    if (force == 1 || force == 2) return force;
    return (wgs <= 384 && total_q > 256 && max_seqlen_k >= 512) ? 2 : 1;
}
}  // namespace

extern "C" int32_t nvh_attn_varlen_suggest_splits(int64_t total_q, int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                                                  int64_t max_seqlen_k) {
    // Key ranges cost partial results (fp32 O per query row, head and range, written and read back), so a chunk with
    // many query rows is only cut while every range keeps 1024 keys and the grid is short of one workgroup per CU
    // (1024 queries of 16 heads over 4096 keys: 85.1 us uncut, 58.1 us in two ranges); a handful of queries over a long
    // context (the tail of a prompt) is cut down to 256 keys per range.
    if (num_kv_heads <= 0 || num_q_heads % num_kv_heads) return 1;
    const int bq = rows_per_block((int)(num_q_heads / num_kv_heads));
    const int64_t blocks = (total_q / bq + batch) * num_kv_heads;
    const int64_t tiles = nvh_cdiv(max_seqlen_k, 64);
    int64_t s, cap;
    if (total_q <= 256) {
        s = nvh_cdiv(512, blocks > 0 ? blocks : 1);
        cap = tiles / 4;
    } else {
        s = (256 + blocks / 2) / (blocks > 0 ? blocks : 1);     // nearest count to one workgroup per CU
        // from 6144 keys (where the launch walks its grid kv-head-major) four ranges: 8192 keys 85-88 us against 88-93 us in
        // two; the chunks 6-8 of an 8192-token prompt 4.02 / 4.26 / 4.57 -> 3.96 / 4.18 / 4.47 ms
        const int64_t long_s = nvh_switch_or(SW_VARLEN_LONG_SPLITS, 4);   // tuning switch
        if (max_seqlen_k >= 6144 && s < long_s) s = long_s;
        cap = tiles / 16;                                      // at least 1024 keys per range
    }
    if (s > cap) s = cap;
    if (s > 32) s = 32;
    return (int32_t)(s < 1 ? 1 : s);
}

extern "C" int64_t nvh_attn_varlen_workspace_bytes(int64_t total_q, int64_t num_q_heads, int64_t head_dim, int32_t num_splits) {
    if (num_splits <= 1) return 0;
    return total_q * num_q_heads * num_splits * (head_dim + 2) * (int64_t)sizeof(float);
}

extern "C" int32_t nvh_attn_varlen(const void* q, int64_t q_token_stride, const void* k, const void* v,
                                   int64_t kv_token_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                   const int32_t* block_tables, int64_t table_stride, int64_t batch, int64_t max_seqlen_q,
                                   int64_t max_seqlen_k, int64_t total_q, int64_t num_q_heads, int64_t num_kv_heads,
                                   int64_t head_dim, float softmax_scale, void* out, int32_t num_splits, void* workspace,
                                   nvh_stream_t stream) {
    NVH_REQUIRE(q && k && v && cu_seqlens_q && cu_seqlens_k && out, NVH_EINVAL, "nvh_attn_varlen: null pointer");
    NVH_REQUIRE(head_dim == 128, NVH_ENOSYS, "nvh_attn_varlen: head_dim %ld unsupported (128 only)", (long)head_dim);
    NVH_REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, NVH_EINVAL, "nvh_attn_varlen: Hq %% Hkv != 0");
    NVH_REQUIRE(q_token_stride % 8 == 0 && kv_token_stride % 8 == 0 && nvh_aligned16(q) && nvh_aligned16(k) && nvh_aligned16(v) && ((uintptr_t)out & 7) == 0,
                NVH_EINVAL, "nvh_attn_varlen: 16-byte alignment required");
    NVH_REQUIRE(num_splits >= 1 && num_splits <= 64 && (num_splits == 1 || (workspace && nvh_aligned16(workspace))), NVH_EINVAL,
                "nvh_attn_varlen: num_splits %d needs a 16-byte aligned workspace", num_splits);
    NVH_REQUIRE(batch * num_splits <= 65535 && num_kv_heads <= 65535, NVH_EINVAL, "nvh_attn_varlen: batch/heads exceed grid limits");
    if (batch == 0 || max_seqlen_q == 0 || total_q == 0) return NVH_OK;
    const int g = (int)(num_q_heads / num_kv_heads);
    const bool paged = block_tables != nullptr;
    hipStream_t st = (hipStream_t)stream;
    float* ws_acc = (float*)workspace;
    float* ws_ml = ws_acc ? ws_acc + total_q * num_q_heads * num_splits * 128 : nullptr;
    const int tiles = (int)nvh_cdiv(max_seqlen_k > 0 ? max_seqlen_k : 1, 64);
    const int64_t wgs = nvh_cdiv(max_seqlen_q, rows_per_block(g)) * num_kv_heads * batch * num_splits;
    const int ks = key_groups(g, wgs, total_q, max_seqlen_k);
    // whole steps of `ks` tiles per key range (a step's tiles then share one page of the block table)
    VarlenArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, q_token_stride, kv_token_stride, cu_seqlens_q, cu_seqlens_k,
                 block_tables, table_stride, (int)num_kv_heads, softmax_scale * LOG2E, (bf16*)out, num_splits,
                 (int)(nvh_cdiv(nvh_cdiv(tiles, num_splits), ks) * ks), ws_acc, ws_ml, 0};
    // Grid order (us per launch, 16 q / 8 kv heads, 1024 query rows, query-block-major vs head-major): uncut 1024 keys 22.5 vs
    // 22.1, 8192 keys 128.1 vs 127.6; 2 x 512 x 512: 15.3 vs 13.4; 8 x 128 x 128: 9.35 vs 8.97; in two key ranges 2048 keys
    // 35.3 vs 38.2, 4096 keys 52.6 vs 53.8, 8192 keys 93.3 vs 87.8 (four ranges 91.1 vs 85.0): head-major unless a cut
    // launch has fewer than 6144 keys (NVH_ATTN_HEAD_MAJOR=0 / 1 forces one order for measurements).
    const int force_hm = (int)nvh_switch(SW_ATTN_HEAD_MAJOR);
    a.head_major = force_hm >= 0 ? force_hm : (num_splits == 1 || max_seqlen_k >= 6144);
    if (nvh_cdiv(max_seqlen_q, 32) > 65535) a.head_major = 0;          // the query blocks would not fit gridDim.y
#define LAUNCH_VL(GG, RBB, KSS, STG)                                                                                        \
    do {                                                                                                                    \
        dim3 grid((unsigned)nvh_cdiv(max_seqlen_q, 32 * RBB), (unsigned)num_kv_heads, (unsigned)(batch * num_splits));      \
        if (a.head_major) grid = dim3(grid.y, grid.x, grid.z);                                                              \
        launch_varlen<GG, RBB, KSS, STG>(paged, grid, st, a);                                                               \
    } while (0)
    switch (g) {
        case 1: if (ks == 2) LAUNCH_VL(1, 4, 2, 2); else LAUNCH_VL(1, 4, 1, 2); break;
        // measured and not kept for 2 query heads per kv head: 32-row blocks (twice the K/V staging per MFMA), 128-row
        // blocks (half the workgroups; 99.9 us against 98 at 8192 keys in 4 ranges) and a ring of 4 steps instead of 2 (no
        // change: the loop is bound by instruction issue, not by the DMA latency)
        case 2: {
            // a grid that leaves half the CUs without a workgroup (one 1024-token chunk: 16 blocks x 8 kv heads): 32-row
            // blocks put a four-wave workgroup on every CU instead of an eight-wave one on every other
            const int rb1 = (int)nvh_switch_or(SW_ATTN_RB1, 128);     // A/B switch: 0 = never
            if (ks == 2 && wgs <= rb1) LAUNCH_VL(2, 1, 2, 2);
            else if (ks == 2) LAUNCH_VL(2, 2, 2, 2);
            else LAUNCH_VL(2, 2, 1, 2);
            break;
        }
        case 4: if (ks == 2) LAUNCH_VL(4, 1, 2, 2); else LAUNCH_VL(4, 1, 1, 2); break;
        case 5: LAUNCH_VL(5, 1, 1, 2); break;
        case 8: LAUNCH_VL(8, 1, 1, 2); break;
        default:
            nvh_set_error("nvh_attn_varlen: group size Hq/Hkv = %d unsupported (1,2,4,5,8)", g);
            return NVH_ENOSYS;
    }
#undef LAUNCH_VL
    NVH_CHECK_LAUNCH("attn_varlen");
    if (num_splits > 1) {
        return nvh_launch_attn_combine(ws_acc, ws_ml, out, total_q * num_q_heads, num_splits, nullptr, 1, 64, nullptr, st);   // same records as decode
    }
    return NVH_OK;
}
