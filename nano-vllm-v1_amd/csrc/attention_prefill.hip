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
// K/V tiles go global -> LDS by 16-byte LDS-DMA into a ring of LDS stages (2 in use; 4 measured no faster), one
// barrier per tile and a counted vmcnt wait: the DMA of tile t+1 runs under the MFMA work of tile t; the
// swizzles are applied to the per-lane SOURCE address because the DMA writes lane-linear.  Two workgroups fit a
// CU (<= 256 registers, 64 KB LDS), so one's softmax overlaps the other's MFMAs.  Nothing in the tile loop may
// make the compiler wait for vmcnt(0) (that drains the prefetch): Q fragments are pinned in registers before
// the loop, the transposing V reads are issued from inline asm, and the loop must stay free of register spills.
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

#ifdef NVH_ATTN_PROF
// Tuning aid (never defined in the shipped library): per-phase cycle totals of wave 0 of every workgroup.
__device__ unsigned long long g_attn_prof[8];
extern "C" int32_t nvh_debug_attn_prof(unsigned long long* out, int32_t reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_prof), sizeof(g_attn_prof)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define PROF_DECL unsigned long long prof_acc[5] = {0, 0, 0, 0, 0}
#define PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) prof_acc[slot] += (b) - (a)
#define PROF_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&g_attn_prof[i_], prof_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#define PROF_FLUSH
#endif

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

template <int G, int RB, int KS, int STAGES, bool PAGED>
__global__ __launch_bounds__(G * RB * KS * 64, (G * RB * KS <= 4 && STAGES * KS == 2 ? 2 : 1)) void attn_varlen_kernel(
    const bf16* __restrict__ q, int64_t q_stride, const bf16* __restrict__ kp, const bf16* __restrict__ vp,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ tables, int64_t table_stride, int hkv, float scale_log2, bf16* __restrict__ out,
    int num_splits, int tiles_per_split, float* __restrict__ ws_acc, float* __restrict__ ws_ml) {
    constexpr int NW1 = G * RB, NW = NW1 * KS, BQ = 32 * RB;     // NW1: waves of one key group
    constexpr int PIECES = 32 * KS;                              // 1-KiB DMA pieces per step: KS x (K + V tile), 4 keys each
    constexpr int PW = (PIECES + NW - 1) / NW;                   // pieces per wave (uniform; extras duplicate)
    constexpr int STAGE = 2 * 64 * 256;                          // K tile + V tile
    static_assert(STAGES == 2 || STAGES == 4, "the stage offset is folded into LDS addresses as a bit field");
    static_assert(KS == 1 || KS == 2, "one or two key groups");
    static_assert(KS == 1 || 32 % NW == 0, "with key groups a wave's piece i must belong to the same tile for every wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES x KS x (K tile + V tile)

    const int b = blockIdx.z / num_splits, split = blockIdx.z % num_splits, kvh = blockIdx.y;
    const int q_beg = cu_q[b], lq = cu_q[b + 1] - q_beg;
    const int k_beg = cu_k[b], lk = cu_k[b + 1] - k_beg;
    const int qb0 = (gridDim.x - 1 - blockIdx.x) * BQ;           // last (longest, under the causal mask) query blocks first
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
    PROF_DECL;

    if (kt_beg < kt_end) {
        bf16x8 qf[8];                                            // Q^T fragments, filled below
        // LDS-DMA staging: piece p < 16 -> K keys 4p..4p+3, p >= 16 -> V keys; lane -> (key = 4p + l4, chunk position).
        // The swizzled source chunk is (lane term) ^ (piece term) for both tiles, so three lane registers and a few
        // wave-uniform values describe every piece; a tile adds one wave-uniform offset.
        const int l4 = lane >> 4, pos = lane & 15;
        const int64_t row_stride = PAGED ? (int64_t)hkv * 128 : kv_stride;       // elements between consecutive keys
        const int dma_row = l4 * (int)row_stride + kvh * 128;                    // lane's key row inside a 4-key piece
        const int dma_ck = (pos ^ l4) * 8, dma_cv = (pos ^ (l4 << 2)) * 8;       // K: pos ^ (key & 15); V: pos ^ ((key&3)<<2 | (key>>2)&3)
        // page id of the next tile to stage, fetched one iteration ahead so its scalar-load latency is never exposed
        const int32_t* table_row = PAGED ? tables + (int64_t)b * table_stride : nullptr;
        const int last_page = PAGED ? max(lk - 1, 0) >> 8 : 0;
        int page_ahead = 0;
        auto fetch_page = [&](int kt) {
            if constexpr (PAGED) page_ahead = table_row[min(kt >> 2, last_page)];
        };
        // Step `st` stages key tiles kt_beg + st*KS + j, j < KS, into slots (st % STAGES) * KS + j.  kt_beg is a multiple
        // of KS and KS divides the 4 tiles of a page, so the tiles of a step share one page-table entry.  A tile past the
        // last key of the sequence (odd tile count) re-reads the last valid tile: the DMA count per step stays uniform.
        const int last_tile = max(lk - 1, 0) >> 6;
        auto stage = [&](int st) {
            const int kt0 = kt_beg + st * KS;
            LDS_AS char* base = (LDS_AS char*)smem + (st & (STAGES - 1)) * (KS * STAGE);
            int64_t page = 0;
            if constexpr (PAGED) {
                page = page_ahead;                               // fetch_page(kt0) ran an iteration ago
                fetch_page(kt0 + KS);
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                int p = wave + i * NW;
                if (p >= PIECES) p -= PIECES;
                const int kt = min(kt0 + (p >> 5), last_tile);   // (p >> 5: the same for every wave, see the static_assert)
                int64_t tile_off;                                // wave-uniform element offset of key kt*64
                if constexpr (PAGED) tile_off = (page * NVH_KV_PAGE + (kt & 3) * 64) * row_stride;
                else tile_off = (int64_t)(k_beg + kt * 64) * row_stride;
                const int isv = (p >> 4) & 1, pv = p & 15;
                const bf16* sbase = (isv ? vp : kp) + tile_off + (int64_t)(4 * pv) * row_stride;
                int off = dma_row + (isv ? (dma_cv ^ (8 * (pv & 3))) : (dma_ck ^ (32 * (pv & 3))));
                if (kt * 64 + 63 >= lk) {
                    // last, partly filled tile: keys past the end re-read the last valid key (they are masked anyway)
                    const int over = max(kt * 64 + 4 * pv + l4 - (lk - 1), 0);
                    off -= over * (int)row_stride;
                }
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sbase + off), (LDS_AS void*)(base + p * 1024), 16, 0, 0);
            }
        };

        // LDS read addresses: every fragment read is (lane term ^ wave-uniform term) + (compile-time offset).
        //   K fragment of k-step ks, key t2*32 + ql:   256*key + 16*((2ks + hh) ^ (key & 15))
        //   V^T fragment (dt, t2, sft, half):          v_off(row, ch) with the row/chunk split below
        const int gi = lane >> 4, li = lane & 15;
        const int tr_q = li >> 2, tr_p = li & 3;
        // lane parts; the (ks) / (dt, half) parts and the stage bit are wave-uniform XOR terms applied per tile
        const int kb0 = 256 * ql + ((hh ^ (ql & 15)) << 4);
        const int vb0 = 256 * (4 * (gi >> 1) + tr_q) + 16 * ((2 * (gi & 1) + (tr_p >> 1)) ^ ((tr_q << 2) | (gi >> 1))) + 8 * (tr_p & 1);
        LDS_AS const char* lds = (LDS_AS const char*)smem;

        // One 64-key tile.  MASK: the tile crosses this wave's causal diagonal.
        auto tile = [&](int kt, int slot, auto mask_tag) {
            constexpr bool MASK = decltype(mask_tag)::value;
            const int soff = slot * STAGE;
            int kb[8], vb[4][2];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kb[ks] = kb0 ^ ((32 * ks) ^ soff);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int half = 0; half < 2; ++half) vb[dt][half] = vb0 ^ ((16 * ((4 * dt) ^ (2 * half))) ^ soff);
            // ---- S^T = K . Q^T for the two 32-key halves
            f32x16 st[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                if (t2) __builtin_amdgcn_sched_barrier(0);      // 8 K fragments (32 registers) in flight, not 16
#pragma unroll
                for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks == 2 || ks == 4 || ks == 6) __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 kfrag = *reinterpret_cast<LDS_AS const bf16x8*>(lds + kb[ks] + 8192 * t2);
                    st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[ks], st[t2], 0, 0, 0);
                }
            }
            // ---- causal mask, online softmax (query on the lane).  Scores stay unscaled: the scale is
            // folded into the exponent, p = exp2(s * scale*log2e - m), one FMA per element.
            if constexpr (MASK) {
                const int vis = qrow + shift - kt * 64 - 4 * hh;   // last visible key, relative to the lane's first
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[t2][r] = (t2 * 32 + (r & 3) + 8 * (r >> 2)) > vis ? -INFINITY : st[t2][r];
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2;
            // The running max only moves when some query's max grew by more than 2^RESCALE_LOG2: below
            // that, p <= 2^RESCALE_LOG2 is still exact enough in bf16/fp32 and the O rescale is skipped.
            constexpr float RESCALE_LOG2 = 6.0f;
            float alpha = 1.0f;
            if (__any(mx > m_run + RESCALE_LOG2)) {
                const float m_new = fmaxf(m_run, mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                alpha = __builtin_amdgcn_exp2f(m_run - m_use);
                asm volatile("" : "+v"(alpha));                 // opaque: keeps the 64 multiplies inside this rare branch
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            const float neg_m = (m_run == -INFINITY) ? 0.f : -m_run;
            const f32x2 sc2 = {scale_log2, scale_log2}, nm2 = {neg_m, neg_m};
            f32x2 psum2 = {0.f, 0.f};
            bf16x8 pb[2][2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 s2 = {st[t2][r], st[t2][r + 1]};
                    const f32x2 e2 = __builtin_elementwise_fma(s2, sc2, nm2);
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                    psum2 += p2;
                    pb[t2][r >> 3][r & 7] = f2bf(p2[0]);
                    pb[t2][r >> 3][(r & 7) + 1] = f2bf(p2[1]);
                }
            l_run = l_run * alpha + (psum2[0] + psum2[1]);
            // ---- O^T += V^T . P^T.  The transposing LDS reads are issued from inline asm: as compiler-visible loads
            // they would each be preceded by a vmcnt(0) (LDS reads are ordered after every LDS-DMA in flight), which
            // drains the prefetched tiles.  One group = the 4 reads that feed two MFMAs (the SIMD's other wave covers their
            // latency).  lgkmcnt(0) is the only wait (LDS returns in order; over-waiting is safe).
            bf16x4 vl[2], vh[2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {                  // rows 32*t2 + 16*sft (+8 for the high half)
                    if (t2 == 0)
                        asm volatile(
                            "ds_read_b64_tr_b16 %0, %4 offset:16384\n\t"
                            "ds_read_b64_tr_b16 %1, %5 offset:18432\n\t"
                            "ds_read_b64_tr_b16 %2, %4 offset:20480\n\t"
                            "ds_read_b64_tr_b16 %3, %5 offset:22528\n\t"
                            "s_waitcnt lgkmcnt(0)"
                            : "=&v"(vl[0]), "=&v"(vh[0]), "=&v"(vl[1]), "=&v"(vh[1])
                            : "v"(vb[dt][0]), "v"(vb[dt][1])
                            : "memory");
                    else
                        asm volatile(
                            "ds_read_b64_tr_b16 %0, %4 offset:24576\n\t"
                            "ds_read_b64_tr_b16 %1, %5 offset:26624\n\t"
                            "ds_read_b64_tr_b16 %2, %4 offset:28672\n\t"
                            "ds_read_b64_tr_b16 %3, %5 offset:30720\n\t"
                            "s_waitcnt lgkmcnt(0)"
                            : "=&v"(vl[0]), "=&v"(vh[0]), "=&v"(vl[1]), "=&v"(vh[1])
                            : "v"(vb[dt][0]), "v"(vb[dt][1])
                            : "memory");
#pragma unroll
                    for (int sft = 0; sft < 2; ++sft) {
                        const bf16x8 vfrag = __builtin_shufflevector(vl[sft], vh[sft], 0, 1, 2, 3, 4, 5, 6, 7);
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag, pb[t2][sft], o[dt], 0, 0, 0);
                    }
                }
        };

        constexpr int AHEAD = STAGES - 1;                       // steps kept in flight
        const int nsteps = (kt_end - kt_beg + KS - 1) / KS;
        fetch_page(kt_beg);
        for (int t = 0; t < AHEAD && t < nsteps; ++t) stage(t);
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
        for (int st = 0; st < nsteps; ++st) {
            // this wave's pieces of step st have landed (later steps may still be in flight) ...
            PROF_T(p0);
            attn_wait_tiles<PW, AHEAD - 1>(min(AHEAD - 1, nsteps - 1 - st));
            asm volatile("" ::: "memory");
            PROF_T(p1);
            __builtin_amdgcn_s_barrier();                       // ... everyone's have, and step st-1 is consumed
            asm volatile("" ::: "memory");
            PROF_T(p2);
            if (st + AHEAD < nsteps) stage(st + AHEAD);         // into the slots of step st-1
            PROF_T(p3);
            PROF_ADD(0, p0, p1);
            PROF_ADD(1, p1, p2);
            PROF_ADD(2, p2, p3);
            const int kt = kt_beg + st * KS + ks_id;            // this key group's tile of the step
            if (kt < kt_end && kt * 64 <= wave_last_key) {      // else: no such tile, or entirely above this wave's diagonal
                const int slot = (st & (STAGES - 1)) * KS + ks_id;
                if (kt * 64 + 63 > wave_first_row + shift) tile(kt, slot, std::true_type{});
                else tile(kt, slot, std::false_type{});
            }
            PROF_T(p4);
            PROF_ADD(3, p3, p4);
            PROF_ADD(4, p0, p0 + 1);
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

    PROF_FLUSH;
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
                                                                                         a.hkv, a.sl2, a.out, a.num_splits, a.tiles_per_split, a.ws_acc, a.ws_ml);
    else
        attn_varlen_kernel<G, RB, KS, STAGES, false><<<grid, G * RB * KS * 64, smem, st>>>(a.q, a.qs, a.k, a.v, a.kvs, a.cu_q, a.cu_k, a.tables, a.ts,
                                                                                          a.hkv, a.sl2, a.out, a.num_splits, a.tiles_per_split, a.ws_acc, a.ws_ml);
}

// Query rows per workgroup: 4-5 waves per key group (8 for G = 8), each 32 rows of one query head.
int rows_per_block(int g) { return g == 1 ? 128 : g == 2 ? 64 : 32; }
// Key groups per workgroup (1 or 2) for a launch of `wgs` workgroups: two groups (8 waves, one workgroup per CU) while
// the grid leaves CUs without a second workgroup anyway; four-wave workgroups, two per CU, beyond that.
// Measured on 16 q / 8 kv heads (us per launch, one group vs two): Lq = Lk = 1024: 34.6 (29.9 with 4 key ranges and the
// merge launch) vs 28.5; Lq 1024 / Lk 4096 in 2 ranges: 72.1 vs 58.1; 8 x 128 tokens: 10.0 vs 11.1 (nothing to share out);
// 16 queries x 4 sequences over 4096 keys in many ranges: 26.4 vs 34.5.
int key_groups(int g, int64_t wgs, int64_t total_q, int64_t max_seqlen_k) {
    static const int force = getenv("NVH_ATTN_KS") ? atoi(getenv("NVH_ATTN_KS")) : 0;      // A/B switch for measurements
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
                 (int)(nvh_cdiv(nvh_cdiv(tiles, num_splits), ks) * ks), ws_acc, ws_ml};
#define LAUNCH_VL(GG, RBB, KSS)                                                                                             \
    do {                                                                                                                    \
        dim3 grid((unsigned)nvh_cdiv(max_seqlen_q, 32 * RBB), (unsigned)num_kv_heads, (unsigned)(batch * num_splits));      \
        launch_varlen<GG, RBB, KSS, 2>(paged, grid, st, a);                                                                 \
    } while (0)
    switch (g) {
        case 1: if (ks == 2) LAUNCH_VL(1, 4, 2); else LAUNCH_VL(1, 4, 1); break;
        case 2: if (ks == 2) LAUNCH_VL(2, 2, 2); else LAUNCH_VL(2, 2, 1); break;   // 32- and 128-row blocks measured slower
        case 4: if (ks == 2) LAUNCH_VL(4, 1, 2); else LAUNCH_VL(4, 1, 1); break;
        case 5: LAUNCH_VL(5, 1, 1); break;
        case 8: LAUNCH_VL(8, 1, 1); break;
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
