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
// K/V tiles go global -> registers -> LDS; the loads for tile t+1 are issued before the
// MFMA work of tile t and written to LDS after it.
#include "common.h"

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int v_off(int row, int ch) {          // V tile: 256-B rows, chunk XOR
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ int k_off(int row, int ch) {          // K tile
    return 256 * row + 16 * (ch ^ (row & 15));
}

template <int G, int RB, bool PAGED>
__global__ __launch_bounds__(G * RB * 64) void attn_varlen_kernel(
    const bf16* __restrict__ q, int64_t q_stride, const bf16* __restrict__ kp, const bf16* __restrict__ vp,
    int64_t kv_stride, const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k,
    const int32_t* __restrict__ tables, int64_t table_stride, int hkv, float scale_log2, bf16* __restrict__ out) {
    constexpr int NW = G * RB, NT = NW * 64, BQ = 32 * RB;
    constexpr int LD = 2048 / NT + ((2048 % NT) ? 1 : 0);        // 16-B chunks per thread per tile (K+V)
    __shared__ __attribute__((aligned(16))) char sm_k[64 * 256];
    __shared__ __attribute__((aligned(16))) char sm_v[64 * 256];

    const int b = blockIdx.z, kvh = blockIdx.y;
    const int q_beg = cu_q[b], lq = cu_q[b + 1] - q_beg;
    const int k_beg = cu_k[b], lk = cu_k[b + 1] - k_beg;
    const int qb0 = blockIdx.x * BQ;
    if (qb0 >= lq) return;
    const int hq = hkv * G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / RB, rb = wave % RB;
    const int h = kvh * G + g;
    const int ql = lane & 31, hh = lane >> 5;
    const int shift = lk - lq;                                    // causal diagonal offset
    const int qrow_raw = qb0 + rb * 32 + ql;
    const int qrow = min(qrow_raw, lq - 1);

    // keys needed by this workgroup / this wave
    const int wg_last_key = min(qb0 + BQ, lq) - 1 + shift;
    const int num_kt = wg_last_key / 64 + 1;
    const int wave_first_row = min(qb0 + rb * 32, lq - 1);
    const int wave_last_key = min(qb0 + rb * 32 + 31, lq - 1) + shift;

    // Q^T fragments: lane (q = ql, half hh) holds Q[q][16ks + 8hh .. +7]
    bf16x8 qf[8];
    {
        const bf16* qp = q + (int64_t)(q_beg + qrow) * q_stride + (int64_t)h * 128 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: chunk id -> (K|V, key, 16-B chunk)
    bf16x8 stage[LD];
    auto issue_loads = [&](int kt) {
        int64_t page_base = 0;
        if constexpr (PAGED) page_base = (int64_t)tables[(int64_t)b * table_stride + ((kt * 64) >> 8)] * NVH_KV_PAGE;
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int id = i * NT + tid;
            if (id < 2048) {
                const int isv = id >> 10, key = (id >> 4) & 63, ch = id & 15;
                const int kk = min(kt * 64 + key, lk - 1);
                const bf16* base = isv ? vp : kp;
                int64_t row;
                if constexpr (PAGED) row = ((page_base + (kk & 255)) * hkv + kvh) * 128;
                else row = (int64_t)(k_beg + kk) * kv_stride + (int64_t)kvh * 128;
                stage[i] = *reinterpret_cast<const bf16x8*>(base + row + ch * 8);
            }
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int id = i * NT + tid;
            if (id < 2048) {
                const int isv = id >> 10, key = (id >> 4) & 63, ch = id & 15;
                if (isv) *reinterpret_cast<bf16x8*>(sm_v + v_off(key, ch)) = stage[i];
                else *reinterpret_cast<bf16x8*>(sm_k + k_off(key, ch)) = stage[i];
            }
        }
    };

    issue_loads(0);
    for (int kt = 0; kt < num_kt; ++kt) {
        __syncthreads();                       // previous tile fully consumed
        write_lds();
        __syncthreads();
        if (kt + 1 < num_kt) issue_loads(kt + 1);
        if (kt * 64 > wave_last_key) continue; // tile entirely above this wave's diagonal (wave-uniform)

        // ---- S^T = K . Q^T for the two 32-key halves
        f32x16 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
            const int key = t2 * 32 + ql;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(sm_k + k_off(key, 2 * ks + hh));
                st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[ks], st[t2], 0, 0, 0);
            }
        }
        // ---- scale, causal mask, online softmax (query on the lane)
        const bool need_mask = kt * 64 + 63 > wave_first_row + shift;
        const int vis = qrow + shift;          // last visible key of this lane's query
        float mx = -INFINITY;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = st[t2][r] * scale_log2;
                if (need_mask) {
                    const int key = kt * 64 + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    s = key > vis ? -INFINITY : s;
                }
                st[t2][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pb[2][2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[t2][r] - m_use);
                psum += p;
                pb[t2][r >> 3][r & 7] = f2bf(p);
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        // ---- O^T += V^T . P^T
        const int gi = lane >> 4, li = lane & 15;
        const int tr_q = li >> 2, tr_p = li & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int ch = 4 * dt + 2 * (gi & 1) + (tr_p >> 1);   // 16-B chunk of columns 32dt+16(gi&1)+4p
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int r0 = 32 * t2 + 16 * s + 4 * (gi >> 1) + tr_q;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (LDS_AS bf16x4*)(sm_v + v_off(r0, ch) + 8 * (tr_p & 1)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (LDS_AS bf16x4*)(sm_v + v_off(r0 + 8, ch) + 8 * (tr_p & 1)));
                    const bf16x8 vfrag = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag, pb[t2][s], o[dt], 0, 0, 0);
                }
        }
    }

    // ---- normalise and store: lane (q, hh) holds O[q][32dt + 8(r>>2) + 4hh + (r&3)]
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    if (qrow_raw < lq) {
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
    }
}

template <int G, int RB>
static void launch_varlen(bool paged, dim3 grid, hipStream_t st, const bf16* q, int64_t qs, const bf16* k, const bf16* v,
                          int64_t kvs, const int32_t* cu_q, const int32_t* cu_k, const int32_t* tables, int64_t ts, int hkv,
                          float sl2, bf16* out) {
    if (paged)
        attn_varlen_kernel<G, RB, true><<<grid, G * RB * 64, 0, st>>>(q, qs, k, v, kvs, cu_q, cu_k, tables, ts, hkv, sl2, out);
    else
        attn_varlen_kernel<G, RB, false><<<grid, G * RB * 64, 0, st>>>(q, qs, k, v, kvs, cu_q, cu_k, tables, ts, hkv, sl2, out);
}

extern "C" int32_t nvh_attn_varlen(const void* q, int64_t q_token_stride, const void* k, const void* v,
                                   int64_t kv_token_stride, const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                                   const int32_t* block_tables, int64_t table_stride, int64_t batch, int64_t max_seqlen_q,
                                   int64_t total_q, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                   float softmax_scale, void* out, nvh_stream_t stream) {
    NVH_REQUIRE(q && k && v && cu_seqlens_q && cu_seqlens_k && out, NVH_EINVAL, "nvh_attn_varlen: null pointer");
    NVH_REQUIRE(head_dim == 128, NVH_ENOSYS, "nvh_attn_varlen: head_dim %ld unsupported (128 only)", (long)head_dim);
    NVH_REQUIRE(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, NVH_EINVAL, "nvh_attn_varlen: Hq %% Hkv != 0");
    NVH_REQUIRE(q_token_stride % 8 == 0 && kv_token_stride % 8 == 0 && nvh_aligned16(q) && nvh_aligned16(k) && nvh_aligned16(v) && ((uintptr_t)out & 7) == 0,
                NVH_EINVAL, "nvh_attn_varlen: 16-byte alignment required");
    NVH_REQUIRE(batch <= 65535 && num_kv_heads <= 65535, NVH_EINVAL, "nvh_attn_varlen: batch/heads exceed grid limits");
    (void)total_q;
    if (batch == 0 || max_seqlen_q == 0) return NVH_OK;
    const int g = (int)(num_q_heads / num_kv_heads);
    const bool paged = block_tables != nullptr;
    const float sl2 = softmax_scale * LOG2E;
    hipStream_t st = (hipStream_t)stream;
    const bf16 *qp = (const bf16*)q, *kp = (const bf16*)k, *vp = (const bf16*)v;
    bf16* op = (bf16*)out;
#define LAUNCH_VL(GG, RBB)                                                                                          \
    do {                                                                                                            \
        dim3 grid((unsigned)nvh_cdiv(max_seqlen_q, 32 * RBB), (unsigned)num_kv_heads, (unsigned)batch);             \
        launch_varlen<GG, RBB>(paged, grid, st, qp, q_token_stride, kp, vp, kv_token_stride, cu_seqlens_q,          \
                               cu_seqlens_k, block_tables, table_stride, (int)num_kv_heads, sl2, op);               \
    } while (0)
    switch (g) {
        case 1: LAUNCH_VL(1, 4); break;
        case 2: LAUNCH_VL(2, 2); break;
        case 4: LAUNCH_VL(4, 1); break;
        case 5: LAUNCH_VL(5, 1); break;
        case 8: LAUNCH_VL(8, 1); break;
        default:
            nvh_set_error("nvh_attn_varlen: group size Hq/Hkv = %d unsupported (1,2,4,5,8)", g);
            return NVH_ENOSYS;
    }
#undef LAUNCH_VL
    NVH_CHECK_LAUNCH("attn_varlen");
    return NVH_OK;
}
