// Var-len causal GQA attention for LONG key ranges and two query heads per kv head (Qwen3-0.6B: 16 / 8): the same
// flash-style algorithm, LDS layouts and MFMA operand maps as attn_varlen_kernel (attention_prefill.hip), with another
// division of the work between waves.
//
// attn_varlen_kernel gives a wave 32 query rows of ONE head, so every 32x32x16 MFMA takes a fresh 1-KiB K or V^T fragment
// from LDS, and two such waves share a SIMD.  Over thousands of keys that loop moves data most of the time (with the
// arithmetic compiled out it still takes 70 % of its time): four SIMDs asking for a fragment per MFMA want half of what
// the LDS array can deliver before the tile staging's own writes.  Here a wave owns 32 query rows of BOTH heads of the
// pair: a K fragment feeds two score MFMAs (head 0 and head 1 against the same keys), a V^T fragment two P.V MFMAs -
// half the LDS reads per MFMA, and half the staging per MFMA because a workgroup (four waves, one per SIMD, the whole
// 512-register file each) covers 128 query rows x 2 heads per staged tile.  One wave per SIMD means nothing else hides its
// own latencies, so the loop is software-pipelined inside the wave: the softmax of tile s+1 (both heads, ~600 VALU
// instructions) is issued between the P.V MFMAs of tile s, the K and V^T fragment reads run one step ahead of the MFMAs
// that use them, the tile staging (LDS-DMA) STAGES - 1 tiles ahead.
//
// Used by nvh_attn_varlen for Hq / Hkv == 2 when the launch walks at least 2048 keys per range; results agree with
// attn_varlen_kernel within the rounding of the fp32 sums (another split of the keys between launches' ranges; the
// per-row arithmetic is the same).
#include "common.h"
#include <type_traits>
#include <utility>

#define LOG2E_PAIR 1.4426950408889634f
#ifndef PAIR_STAMPS
#define PAIR_STAMPS 0      // diagnostic build: cycle stamps around the phases of the tile loop, printed by one wave
#endif
#ifndef PAIR_ABLATE
#define PAIR_ABLATE 0      // timing-only builds compile parts of the loop out (results are then wrong)
#endif

#define PAIR_AGPR_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"
#define PAIR_A_O(g, dt) (16 * (4 * (g) + (dt)))
#define PAIR_A_Q(g, ks) (128 + 4 * (8 * (g) + (ks)))

namespace {
template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void pair_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
template <int N>
__device__ __forceinline__ void pair_interleave() {             // four times {one MFMA, then N VALU instructions}
    if constexpr (N > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, N, 0);
        }
    }
}
}  // namespace

template <int STAGES, bool PAGED>
__global__ __launch_bounds__(256, 1) void attn_pair_kernel(
    const bf16* __restrict__ q, int64_t q_stride, const bf16* __restrict__ kp, const bf16* __restrict__ vp,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ tables, int64_t table_stride, int hkv, float scale_log2, bf16* __restrict__ out,
    int num_splits, int tiles_per_split, float* __restrict__ ws_acc, float* __restrict__ ws_ml, int head_major) {
    constexpr int G = 2, NW = 4, BQ = 32 * NW;
    constexpr int PIECES = 32, PW = PIECES / NW;                 // 1-KiB DMA pieces per tile (K + V), per wave
    constexpr int STAGE = 2 * 64 * 256;                          // K tile + V tile
    static_assert(STAGES == 2 || STAGES == 4, "the stage offset is folded into LDS addresses as a bit field");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES x (K tile + V tile)

    const int b = blockIdx.z / num_splits, split = blockIdx.z % num_splits, kvh = head_major ? blockIdx.x : blockIdx.y;
    const int qblock = head_major ? blockIdx.y : blockIdx.x, qblocks = head_major ? gridDim.y : gridDim.x;
    const int q_beg = cu_q[b], lq = cu_q[b + 1] - q_beg;
    const int k_beg = cu_k[b], lk = cu_k[b + 1] - k_beg;
    const int qb0 = (qblocks - 1 - qblock) * BQ;                 // longest query blocks first
    if (qb0 >= lq) return;
    const int hq = hkv * G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row block of the workgroup
    const int ql = lane & 31, hh = lane >> 5;
    const int shift = lk - lq;                                    // causal diagonal offset
    const int qrow_raw = qb0 + wave * 32 + ql;
    const int qrow = min(qrow_raw, lq - 1);

    const int wg_last_key = min(qb0 + BQ, lq) - 1 + shift;
    const int kt_beg = split * tiles_per_split;
    const int kt_end = min(wg_last_key / 64 + 1, kt_beg + tiles_per_split);
    const int wave_first_row = min(qb0 + wave * 32, lq - 1);
    const int wave_last_key = min(qb0 + wave * 32 + 31, lq - 1) + shift;

    // The accumulator half of the register file is laid out by hand and only ever named inside asm statements:
    //   a[16 (4 g + dt) .. + 15]        O^T block dt of head g   (a0 .. a127)
    //   a[128 + 4 (8 g + ks) .. + 3]    Q fragment ks of head g  (a128 .. a191)
    // (with "a" operands the allocator keeps MFMA operands in VGPRs and copies ~500 registers per tile between the two
    // halves).  The file is compiled with VGPR spilling into AGPRs switched off, so the compiler has no use of its own for
    // these registers; the clobber list makes the kernel descriptor allocate them.
    asm volatile("" ::: PAIR_AGPR_CLOBBERS);
    static_for<128>([&](auto i_c) { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(decltype(i_c)::value)); });
    float m_run[G] = {-INFINITY, -INFINITY}, l_run[G] = {0.f, 0.f};

    if (kt_beg < kt_end) {
        const int l4 = lane >> 4, pos = lane & 15;
        const int64_t row_stride = PAGED ? (int64_t)hkv * 128 : kv_stride;
        const int dma_row = l4 * (int)row_stride + kvh * 128;
        const int dma_ck = (pos ^ l4) * 8, dma_cv = (pos ^ (l4 << 2)) * 8;
        const int32_t* table_row = PAGED ? tables + (int64_t)b * table_stride : nullptr;
        const int last_page = PAGED ? max(lk - 1, 0) >> 8 : 0;
        const int last_tile = max(lk - 1, 0) >> 6;
        const int nsteps = kt_end - kt_beg;
        auto page_of_step = [&](int st) -> int {
            if constexpr (PAGED) return table_row[min((kt_beg + max(st, 0)) >> 2, last_page)];
            return 0;
        };
        auto dma16 = [&](const char* sbase, uint32_t lane_bytes, uint32_t lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane_bytes), "s"(sbase), "s"(lds_dst) : "memory");
        };
        auto tile_base = [&](const bf16* ptr, int kt, int page) -> const char* {
            int64_t tile_off;
            if constexpr (PAGED) tile_off = ((int64_t)page * NVH_KV_PAGE + (kt & 3) * 64) * row_stride;
            else tile_off = (int64_t)(k_beg + kt * 64) * row_stride;
            return reinterpret_cast<const char*>(ptr + tile_off);
        };
        // piece i of a wave: i < 4 -> K keys 4 (4i + wave) .., i >= 4 -> V keys 4 (4 (i - 4) + wave) ..
        const uint32_t lane_k = 2u * (uint32_t)(dma_row + (dma_ck ^ (32 * (wave & 3))));
        const uint32_t lane_v = 2u * (uint32_t)(dma_row + (dma_cv ^ (8 * (wave & 3))));
        const uint32_t wave_piece = 8u * (uint32_t)row_stride * (uint32_t)wave;
        const uint32_t lds_base = (uint32_t)(uintptr_t)((LDS_AS char*)smem);
        auto stage_piece = [&](int i, int k_step, int page_k, int v_step, int page_v) {
            {
                const int isv = ((i * NW) >> 4) & 1, pv = ((i * NW) & 15) + wave, p = i * NW + wave;
                const int st = isv ? v_step : k_step;
                if (st < 0 || st >= nsteps) return;
                const int kt_raw = kt_beg + st;
                const int kt = min(kt_raw, last_tile);
                const uint32_t lds_dst = lds_base + (uint32_t)((st & (STAGES - 1)) * STAGE) + (uint32_t)p * 1024u;
                const char* tb = tile_base(isv ? vp : kp, kt, isv ? page_v : page_k);
                if (kt_raw * 64 + 63 < lk) {
                    dma16(tb + (wave_piece + 8u * (uint32_t)row_stride * (uint32_t)((i * NW) & 15)), isv ? lane_v : lane_k, lds_dst);
                    return;
                }
                int off = dma_row + (isv ? (dma_cv ^ (8 * (pv & 3))) : (dma_ck ^ (32 * (pv & 3))));
                const int over = max(kt * 64 + 4 * pv + l4 - (lk - 1), 0);
                off -= over * (int)row_stride;
                const bf16* src = reinterpret_cast<const bf16*>(tb) + (int64_t)(4 * pv) * row_stride + off;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(smem + (lds_dst - lds_base)), 16, 0, 0);
            }
        };
        auto stage = [&](int k_step, int page_k, int v_step, int page_v) {
#pragma unroll
            for (int i = 0; i < PW; ++i) stage_piece(i, k_step, page_k, v_step, page_v);
        };

        const int gi = lane >> 4, li = lane & 15;
        const int tr_q = li >> 2, tr_p = li & 3;
        const int kb0 = 256 * ql + ((hh ^ (ql & 15)) << 4);
        const int vb0 = 256 * (4 * (gi >> 1) + tr_q) + 16 * ((2 * (gi & 1) + (tr_p >> 1)) ^ ((tr_q << 2) | (gi >> 1))) + 8 * (tr_p & 1);

        f32x16 st[G][2];                                         // scores of the tile whose softmax comes next
        u32x4 pbuf[2][G][2][2];                                 // P of the tile whose P.V product comes next / of the one after (by step parity)
        constexpr float RESCALE_LOG2 = 6.0f;

        // The MFMAs are inline asm (register files chosen here, see the layout above): scores are written to VGPRs where the
        // softmax reads them, K / V^T fragments and P are VGPRs.  The compiler does not know these statements are MFMAs, so
        // the wait states are written out: two states ahead of each MFMA (an operand may have just been written by a VALU
        // instruction), and a 12-state drain before anything but an accumulating MFMA touches a result (8-pass MFMA).
#if PAIR_ABLATE & 2
#define PAIR_MFMA_ST0(d, a, g, ks) asm volatile("v_mov_b32 %0, 0" : "=&v"(d[0]) : "v"(a))
#define PAIR_MFMA_ST(d, a, g, ks) asm volatile("" : "+v"(d) : "v"(a))
#else
#define PAIR_MFMA_ST0(d, a, g, ks) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "n"(PAIR_A_Q(g, ks)), "n"(PAIR_A_Q(g, ks) + 3))
#define PAIR_MFMA_ST(d, a, g, ks) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "n"(PAIR_A_Q(g, ks)), "n"(PAIR_A_Q(g, ks) + 3))
#endif
#if PAIR_ABLATE & 4
#define PAIR_MFMA_O(g, dt, a, b) asm volatile("" ::"v"(a), "v"(b))
#else
#define PAIR_MFMA_O(g, dt, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(PAIR_A_O(g, dt)), "n"(PAIR_A_O(g, dt) + 15))
#endif
        auto drain_scores = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            (void)&st;                                           // (a name used only as an asm operand is not captured by a generic lambda)
            asm volatile("s_nop 11" : "+v"(st[g][0]), "+v"(st[g][1]));
        };
        auto drain_o = [&]() { asm volatile("s_nop 11"); };

        // ---- S^T = K . Q^T of both heads over the tile in `slot`: 16 steps of two MFMAs (the two 32-key halves): steps 0..7
        // walk the head dimension for head 0, steps 8..15 walk it again for head 1 (the K fragments are read twice: a wave that
        // ran both heads' MFMAs off one read would have both heads' softmax left over with nothing to hide it behind).  One
        // wave per SIMD hides nothing by itself: the K fragments of step k + KA are requested before the MFMAs of step k
        // (an LDS read takes ~200 cycles to come back, a step 64), and `gap(j)` is the code issued behind MFMA j (0..31)
        // while it runs: a tile-staging piece or a slice of the other head's softmax (~24 of an MFMA's 32 cycles take other issue).
        constexpr int KA = 3;
        auto qk_tile = [&](int slot, auto&& gap) {
            (void)&st;
            const int soff = slot * STAGE;
            bf16x8 kf[KA + 1][2];
            auto request = [&](int addr, bf16x8 (&dst)[2]) {
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(addr) : "memory");
            };
            static_for<KA>([&](auto k_c) {
                constexpr int k = decltype(k_c)::value;
                request(kb0 ^ ((32 * (k & 7)) ^ soff), kf[k]);
            });
            static_for<16>([&](auto k_c) {
                constexpr int k = decltype(k_c)::value, g = k >> 3, ks = k & 7, cur = k % (KA + 1);
                constexpr int ahead = (k + KA < 16) ? KA : 15 - k;        // requests still in flight behind this step's
                if constexpr (k + KA < 16 && !(PAIR_ABLATE & 16)) request(kb0 ^ ((32 * ((k + KA) & 7)) ^ soff), kf[(k + KA) % (KA + 1)]);
                asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(kf[cur][0]), "+v"(kf[cur][1]) : "n"((PAIR_ABLATE & 16) ? 0 : 2 * ahead) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == 0) PAIR_MFMA_ST0(st[g][0], kf[cur][0], g, 0);
                else PAIR_MFMA_ST(st[g][0], kf[cur][0], g, ks);
                gap(IC<2 * k>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == 0) PAIR_MFMA_ST0(st[g][1], kf[cur][1], g, 0);
                else PAIR_MFMA_ST(st[g][1], kf[cur][1], g, ks);
                gap(IC<2 * k + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // ---- online softmax of one head's scores (query on the lane).  Scores stay unscaled: p = exp2(s * scale*log2e - m).
        // The running max only moves when some query's max grew by more than 2^RESCALE_LOG2; the O rescale it asks for is
        // applied once the P.V product that is in flight against the old max has finished.
        float alpha[G] = {1.0f, 1.0f}, neg_m[G] = {0.f, 0.f}, mx_lane[G] = {0.f, 0.f};
        bool resc[G] = {false, false};
        float psum[G][2] = {{0.f, 0.f}, {0.f, 0.f}};
        auto sm_mask = [&](auto g_c, int kt, bool masked) {
            constexpr int g = decltype(g_c)::value;
            if (masked) {
                const int vis = qrow + shift - kt * 64 - 4 * hh;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[g][t2][r] = (t2 * 32 + (r & 3) + 8 * (r >> 2)) > vis ? -INFINITY : st[g][t2][r];
            }
        };
        auto sm_max_a = [&](auto g_c) {                          // the lane's 32 scores
            constexpr int g = decltype(g_c)::value;
            float mx0 = fmaxf(st[g][0][0], st[g][0][1]), mx1 = fmaxf(st[g][1][0], st[g][1][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) {                    // v_max3_f32: two values per instruction, two chains
                mx0 = __builtin_fmaxf(__builtin_fmaxf(mx0, st[g][0][r]), st[g][0][r + 1]);
                mx1 = __builtin_fmaxf(__builtin_fmaxf(mx1, st[g][1][r]), st[g][1][r + 1]);
            }
            mx_lane[g] = fmaxf(mx0, mx1);
        };
        auto sm_max_b = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            // the other 16 keys of each half sit in lane ^ 32: a half-wave exchange in the vector unit, no trip through LDS
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx_lane[g]), __builtin_bit_cast(unsigned, mx_lane[g]), false, false);
            const float mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])) * scale_log2;
            resc[g] = __any(mx > m_run[g] + RESCALE_LOG2);
            alpha[g] = 1.0f;
            if (resc[g]) {
                const float m_new = fmaxf(m_run[g], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                alpha[g] = __builtin_amdgcn_exp2f(m_run[g] - m_use);
                m_run[g] = m_new;
            }
            neg_m[g] = (m_run[g] == -INFINITY) ? 0.f : -m_run[g];
            psum[g][0] = 0.f;
            psum[g][1] = 0.f;
        };
        // score pair P of a head (registers 2 (P & 7), + 1 of key half P >> 3): exp, row sum, bf16
        auto sm_pair = [&](auto g_c, auto p_c, u32x4 (&pn)[G][2][2]) {
            constexpr int g = decltype(g_c)::value, P = decltype(p_c)::value;
            constexpr int t2 = P >> 3, r = (P & 7) * 2;
            if constexpr (PAIR_ABLATE & 1) {
                pn[g][t2][r >> 3][(r & 7) >> 1] = __builtin_bit_cast(unsigned, st[g][t2][r]);
                return;
            }
            // single-issue f32 instructions: beside MFMAs the packed forms cost more than two plain ones
            float e0 = __builtin_fmaf(st[g][t2][r], scale_log2, neg_m[g]), e1 = __builtin_fmaf(st[g][t2][r + 1], scale_log2, neg_m[g]);
            asm volatile("" : "+v"(e0), "+v"(e1));
            const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
            psum[g][0] += p0;
            asm volatile("" : "+v"(psum[g][0]));
            psum[g][1] += p1;
            asm volatile("" : "+v"(psum[g][1]));
            // (volatile: keeps the pair's instructions in the gap they are written in - its value is not used before the
            // next iteration and would otherwise sink towards that use; the nop is the wait state after a transcendental)
            unsigned pk;
            asm volatile("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(p0), "v"(p1));
            pn[g][t2][r >> 3][(r & 7) >> 1] = pk;
        };
        auto sm_finish = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            l_run[g] = l_run[g] * alpha[g] + (psum[g][0] + psum[g][1]);
        };
        auto rescale_o = [&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            if (resc[g]) {                                       // wave-uniform, rare
                drain_o();
                static_for<64>([&](auto i_c) {
                    float t;
                    (void)&alpha;
                    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0"
                                 : "=&v"(t) : "v"(alpha[g]), "n"(PAIR_A_O(g, 0) + decltype(i_c)::value));
                });
            }
        };
        struct VFrags { bf16x4 lo[4], hi[4]; };
        auto pv_request = [&](int slot, int dt, VFrags& f) {
            const int soff = slot * STAGE;
            const int v0 = vb0 ^ ((16 * ((4 * dt) ^ 0)) ^ soff), v1 = vb0 ^ ((16 * ((4 * dt) ^ 2)) ^ soff);
            if constexpr (PAIR_ABLATE & 32) {
                asm volatile("" : "=v"(f.lo[0]), "=v"(f.hi[0]), "=v"(f.lo[1]), "=v"(f.hi[1]), "=v"(f.lo[2]), "=v"(f.hi[2]), "=v"(f.lo[3]), "=v"(f.hi[3]) : "v"(v0), "v"(v1));
                return;
            }
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
        // one 32-column block DT of O^T += V^T . P^T for both heads (MFMAs 8 DT .. 8 DT + 7 of the product): every V^T
        // fragment feeds two MFMAs; gap(j) as in qk_head
        auto pv_block = [&](auto dt_c, VFrags& f, const u32x4 (&pc)[G][2][2], bool ahead, auto&& gap) {
            constexpr int dt = decltype(dt_c)::value;
            if (ahead)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3])::"memory");
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto u_c) {
                constexpr int u = decltype(u_c)::value;
                const bf16x8 vfrag = __builtin_shufflevector(f.lo[u], f.hi[u], 0, 1, 2, 3, 4, 5, 6, 7);
                PAIR_MFMA_O(0, dt, vfrag, pc[0][u >> 1][u & 1]);
                gap(IC<8 * dt + 2 * u>{});
                __builtin_amdgcn_sched_barrier(0);
                PAIR_MFMA_O(1, dt, vfrag, pc[1][u >> 1][u & 1]);
                gap(IC<8 * dt + 2 * u + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        auto tile_of = [&](int s) { return kt_beg + s; };
        auto tile_valid = [&](int s) { return s < nsteps && tile_of(s) * 64 <= wave_last_key; };
        auto tile_masked = [&](int s) { return tile_of(s) * 64 + 63 > wave_first_row + shift; };
        auto slot_of = [&](int s) { return s & (STAGES - 1); };
        auto wait_for_iteration = [&](int s) {
            if constexpr (STAGES > 2) {
                if (s + STAGES <= nsteps) {
                    pair_wait_vmcnt<PW * (STAGES - 2)>();
                    return;
                }
            }
            pair_wait_vmcnt<0>();
        };

        int page_ring[STAGES + 1];
#pragma unroll
        for (int j = 0; j <= STAGES; ++j) page_ring[j] = page_of_step(j);
        stage(0, page_ring[0], -1, 0);
        static_for<G>([&](auto g_c) {                              // Q fragments straight into their AGPRs
            constexpr int g = decltype(g_c)::value;
            const bf16* qp = q + (int64_t)(q_beg + qrow) * q_stride + (int64_t)(kvh * G + g) * 128 + 8 * hh;
            static_for<8>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                (void)&qp;
                asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(qp), "n"(PAIR_A_Q(g, ks)), "n"(PAIR_A_Q(g, ks) + 3), "n"(32 * ks) : "memory");
            });
        });
#pragma unroll
        for (int j = 1; j < STAGES; ++j) stage(j, page_ring[j], j - 1, page_ring[j - 1]);
        if (STAGES > 2 && STAGES <= nsteps) pair_wait_vmcnt<PW * (STAGES - 1)>();
        else pair_wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tile_valid(0)) {                                     // scores and softmax of tile 0, nothing to overlap
            qk_tile(slot_of(0), [](auto) {});
            static_for<G>([&](auto g_c) {
                drain_scores(g_c);
                sm_mask(g_c, tile_of(0), tile_masked(0));
                sm_max_a(g_c);
                sm_max_b(g_c);
                static_for<16>([&](auto p_c) { sm_pair(g_c, p_c, pbuf[0]); });
                sm_finish(g_c);
                rescale_o(g_c);                                  // (O is still zero: keeps the state consistent)
            });
        }
#if PAIR_STAMPS
        unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int acc_n = 0;
#define STAMP(i) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); acc_t[i] += now__ - last__; last__ = now__; }
#else
#define STAMP(i)
#endif
        // iteration s (PAR = s & 1): P.V of tile s from pbuf[PAR]; scores + softmax of tile s + 1 into pbuf[PAR ^ 1]:
        //   A  scores of head 0 (16 MFMAs)      behind them: four staging pieces
        //   B  scores of head 1 (16 MFMAs)      behind them: head 0's max and NB of its 16 exp pairs
        //   C  P.V of both heads (32 MFMAs)     behind them: the rest of head 0's pairs, head 1's softmax, four staging pieces
        constexpr int NB = 12;
        auto iteration = [&](int s, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
#if PAIR_STAMPS
            unsigned long long last__ = __builtin_amdgcn_s_memtime();
#endif
            wait_for_iteration(s);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // V(s), K(s+1) landed for everyone; the halves read in iteration s-1 are free
            asm volatile("" ::: "memory");
            const int pk = page_ring[STAGES], pvg = page_ring[STAGES - 1];
#pragma unroll
            for (int j = 0; j < STAGES; ++j) page_ring[j] = page_ring[j + 1];
            page_ring[STAGES] = page_of_step(s + STAGES + 1);
            STAMP(0)
            const bool both = tile_valid(s) && tile_valid(s + 1);
            if (!both) stage(s + STAGES, pk, s + STAGES - 1, pvg);          // (no score phase to spread the pieces over)
            if (!tile_valid(s)) return;
            const int vslot = slot_of(s), kslot = slot_of(s + 1);
            auto piece = [&](int i) { if constexpr (!(PAIR_ABLATE & 8)) stage_piece(i, s + STAGES, pk, s + STAGES - 1, pvg); };
            u32x4 (&pc)[G][2][2] = pbuf[PAR];
            u32x4 (&pn)[G][2][2] = pbuf[PAR ^ 1];
            VFrags vf[2];
            if (both) {
                const bool masked = tile_masked(s + 1);
                // ---- A, B
                qk_tile(kslot, [&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    if constexpr (j < 16) {
                        if constexpr ((j & 3) == 3) piece(j >> 2);             // pieces 0..3
                    } else if constexpr (j == 16) {
                        drain_scores(IC<0>{});                                 // (head 0's last MFMA is two back)
                        sm_mask(IC<0>{}, tile_of(s + 1), masked);
                        sm_max_a(IC<0>{});
                    } else if constexpr (j == 17) sm_max_b(IC<0>{});
                    else if constexpr (j < 18 + NB) sm_pair(IC<0>{}, IC<j - 18>{}, pn);
                    else if constexpr (j == 31) pv_request(vslot, 0, vf[0]);   // (behind the wait for the last K fragments)
                });
                drain_scores(IC<1>{});
                STAMP(2)
                // ---- C
                sm_mask(IC<1>{}, tile_of(s + 1), masked);
                auto c_gap = [&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    if constexpr (j == 0) sm_max_a(IC<1>{});
                    else if constexpr (j == 1) sm_max_b(IC<1>{});
                    else {
                        constexpr int i = j - 2;                  // 30 gaps: of every three, two take an exp pair; four of the others a staging piece
                        if constexpr (i % 3 == 2) {
                            if constexpr (i % 9 == 2) piece(4 + i / 9);
                        } else {
                            constexpr int qi = (i / 3) * 2 + (i % 3);
                            if constexpr (qi < 16 - NB) sm_pair(IC<0>{}, IC<NB + qi>{}, pn);
                            else sm_pair(IC<1>{}, IC<qi - (16 - NB)>{}, pn);
                        }
                    }
                };
                static_assert(16 - NB + 16 == 20, "the C gaps hold 20 exp pairs");
                pv_request(vslot, 1, vf[1]);
                pv_block(IC<0>{}, vf[0], pc, true, c_gap);
                STAMP(3)
                pv_request(vslot, 2, vf[0]);
                pv_block(IC<1>{}, vf[1], pc, true, c_gap);
                STAMP(4)
                pv_request(vslot, 3, vf[1]);
                pv_block(IC<2>{}, vf[0], pc, true, c_gap);
                STAMP(5)
                pv_block(IC<3>{}, vf[1], pc, false, c_gap);
                STAMP(6)
                static_for<G>([&](auto g_c) {
                    sm_finish(g_c);
                    rescale_o(g_c);                              // after the whole product against the old max
                });
                STAMP(7)
#if PAIR_STAMPS
                ++acc_n;
#endif
            } else {
                auto none = [](auto) {};
                pv_request(vslot, 0, vf[0]);
                pv_request(vslot, 1, vf[1]);
                pv_block(IC<0>{}, vf[0], pc, true, none);
                pv_request(vslot, 2, vf[0]);
                pv_block(IC<1>{}, vf[1], pc, true, none);
                pv_request(vslot, 3, vf[1]);
                pv_block(IC<2>{}, vf[0], pc, true, none);
                pv_block(IC<3>{}, vf[1], pc, false, none);
            }
        };
        for (int s = 0; s < nsteps; s += 2) {
            iteration(s, IC<0>{});
            if (s + 1 < nsteps) iteration(s + 1, IC<1>{});
        }
        drain_o();
#if PAIR_STAMPS
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0 && acc_n > 0)
            printf("pair stamps over %d tiles, cycles per tile: wait+barrier %llu | scores (+4 dma, max0, 12 pairs) %llu | pv block 0 %llu | 1 %llu | 2 %llu | 3 %llu | finish %llu\n",
                   acc_n, acc_t[0] / acc_n, acc_t[2] / acc_n, acc_t[3] / acc_n, acc_t[4] / acc_n, acc_t[5] / acc_n, acc_t[6] / acc_n, acc_t[7] / acc_n);
#endif
    }

    // ---- lane (q, hh) holds O[q][32dt + 8(r>>2) + 4hh + (r&3)] of both heads
    if (qrow_raw >= lq) return;
    f32x16 o[G][4];
    static_for<G * 4 * 16>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        float t;
        asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(t) : "n"(i));
        o[i >> 6][(i >> 4) & 3][i & 15] = t;
    });
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int h = kvh * G + g;
        const float l_all = l_run[g] + __shfl_xor(l_run[g], 32, 64);
        if (num_splits == 1) {
            const float inv = l_all > 0.f ? 1.0f / l_all : 0.f;
            bf16* op = out + ((int64_t)(q_beg + qrow_raw) * hq + h) * 128 + 4 * hh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = f2bf(o[g][dt][r4 * 4 + e] * inv);
                    *reinterpret_cast<bf16x4*>(op + 32 * dt + 8 * r4) = v;
                }
        } else {
            const int64_t slab = ((int64_t)(q_beg + qrow_raw) * hq + h) * num_splits + split;
            if (l_all > 0.f) {
                float* op = ws_acc + slab * 128 + 4 * hh;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = o[g][dt][r4 * 4 + e];
                        *reinterpret_cast<f32x4*>(op + 32 * dt + 8 * r4) = v;
                    }
            }
            if (hh == 0) {
                ws_ml[slab * 2] = m_run[g];
                ws_ml[slab * 2 + 1] = l_all;
            }
        }
    }
}

// Launch (called by nvh_attn_varlen): grid (query blocks of 128 rows, kv heads, sequences x key ranges) or head-major.
int32_t nvh_launch_attn_pair(bool paged, int stages, dim3 grid, hipStream_t st, const void* q, int64_t qs, const void* k, const void* v,
                             int64_t kvs, const int32_t* cu_q, const int32_t* cu_k, const int32_t* tables, int64_t ts, int hkv, float sl2,
                             void* out, int num_splits, int tiles_per_split, float* ws_acc, float* ws_ml, int head_major) {
#define PAIR_LAUNCH(STG, PG)                                                                                                              \
    do {                                                                                                                                   \
        static bool configured = false;                                                                                                    \
        if (!configured) {                                                                                                                 \
            (void)hipFuncSetAttribute((const void*)attn_pair_kernel<STG, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, STG * 32768);   \
            configured = true;                                                                                                             \
        }                                                                                                                                  \
        attn_pair_kernel<STG, PG><<<grid, 256, STG * 32768, st>>>((const bf16*)q, qs, (const bf16*)k, (const bf16*)v, kvs, cu_q, cu_k,      \
                                                                   tables, ts, hkv, sl2, (bf16*)out, num_splits, tiles_per_split, ws_acc,  \
                                                                   ws_ml, head_major);                                                     \
    } while (0)
    if (stages == 4) { if (paged) PAIR_LAUNCH(4, true); else PAIR_LAUNCH(4, false); }
    else { if (paged) PAIR_LAUNCH(2, true); else PAIR_LAUNCH(2, false); }
#undef PAIR_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        nvh_set_error("attn_pair: launch failed: %s", hipGetErrorString(e));
        return NVH_ELAUNCH;
    }
    return NVH_OK;
}
