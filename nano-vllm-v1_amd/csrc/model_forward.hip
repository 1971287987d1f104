// Native step executor: enqueues every kernel of a Qwen3 forward pass from one call, so the host
// cost of a step is ~250 HIP launches instead of ~250 Python -> ctypes round trips.  It calls the
// same entry points a caller would call one by one (identical kernels, order and rounding).
// Under tensor parallelism (model->comm set) every rank runs it on its shard; the row-parallel GEMMs
// write into the exchange staging area and one all-reduce kernel per projection follows.
#include "common.h"
#include <cstdlib>

namespace {
constexpr int32_t kMaxVarlenSplits = 8;
struct Workspace {
    bf16 *residual, *residual2, *x, *y, *qkv, *attn, *gate_up, *act;
    float* slabs;
    void* decode_ws;
    void* varlen_ws;
    int64_t total;
};

inline int64_t align256(int64_t n) { return (n + 255) & ~int64_t(255); }

int32_t split_for(int64_t m, int64_t n, int64_t k) { return nvh_gemm_suggest_split_k(m, n, k); }

// The QKV projection has no consumer that sums split-K slabs, so splitting it costs a reduce launch.  Worth it only
// for a few rows over a long K with too few column tiles to occupy the chip (Qwen3-14B at decode: 224 tiles,
// 21.3 us unsplit, 15.9 us as two slabs + reduce); never for K = 1024 (the 0.6B model), where a tile lives 4 us.
int32_t split_for_qkv(int64_t m, int64_t n, int64_t k) {
    if (k < 4096 || (n + 31) / 32 > 300) return 1;
    return m <= 16 ? 2 : m <= 32 ? 4 : 1;      // 17..32 rows (128-column tiles): 24.4 us unsplit, 18.8 us as four slabs + reduce
}

Workspace carve(const nvh_qwen3_model* md, int64_t t, int64_t b, int32_t decode_splits, int32_t varlen_splits, char* base) {
    const int64_t h = md->hidden, d = md->head_dim;
    const int64_t qsz = (int64_t)md->num_q_heads * d, kvsz = (int64_t)md->num_kv_heads * d;
    Workspace w;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w.residual = (bf16*)take(t * h * 2);
    w.residual2 = (bf16*)take((t <= 64 ? t : 0) * h * 2);   // decode batches: the residual alternates between two buffers (see forward_impl)
    w.x = (bf16*)take(t * h * 2);
    w.y = (bf16*)take(t * h * 2);
    w.qkv = (bf16*)take(t * (qsz + 2 * kvsz) * 2);
    w.attn = (bf16*)take(t * qsz * 2);
    w.gate_up = (bf16*)take(t * 2 * md->inter * 2);
    w.act = (bf16*)take(t * md->inter * 2);
    const int64_t so = split_for(t, h, qsz), sd = split_for(t, h, md->inter);
    const int64_t smax = so > sd ? so : sd;
    const int64_t sq = split_for_qkv(t, qsz + 2 * kvsz, h);
    const int64_t slab_floats = (smax > 1 ? smax * t * h : 0) > (sq > 1 ? sq * t * (qsz + 2 * kvsz) : 0)
                                    ? (smax > 1 ? smax * t * h : 0) : (sq > 1 ? sq * t * (qsz + 2 * kvsz) : 0);
    w.slabs = (float*)take(slab_floats * 4);
    w.decode_ws = take(nvh_attn_decode_workspace_bytes(b, md->num_q_heads, d, decode_splits));
    w.varlen_ws = take(nvh_attn_varlen_workspace_bytes(t, md->num_q_heads, d, varlen_splits));
    w.total = off;
    return w;
}
}  // namespace

extern "C" int64_t nvh_qwen3_workspace_bytes(const nvh_qwen3_model* model, int64_t num_tokens, int64_t batch,
                                             int32_t decode_splits) {
    if (!model || num_tokens < 0) return -1;
    // sized for the widest key split the executor lets the var-len path choose
    return carve(model, num_tokens, batch, decode_splits < 1 ? 1 : decode_splits, kMaxVarlenSplits, nullptr).total;
}

// The fused decode attention of layer l over the first `rows` sequences of the step: uniform grid or work list.
static int32_t decode_attention(const nvh_qwen3_model* md, const nvh_step* st, int l, const bf16* qkv, int64_t qkv_n, void* kc, void* vc,
                                int64_t rows, float scale, bf16* attn, int32_t dsplits, void* ws, nvh_stream_t stream) {
    if (st->decode_items > 0)
        return nvh_attn_decode_fused_qkv_items(qkv, qkv_n, md->q_norm[l], md->k_norm[l], md->eps, st->positions, md->cos_sin, kc, vc,
                                               st->context_lens, st->block_tables, st->table_stride, rows, md->num_q_heads,
                                               md->num_kv_heads, md->head_dim, scale, attn, st->decode_item_seq, st->decode_item_range,
                                               st->decode_seq_active, st->decode_seq_chunk, st->decode_items, dsplits, ws, stream);
    return nvh_attn_decode_fused_qkv(qkv, qkv_n, md->q_norm[l], md->k_norm[l], md->eps, st->positions, md->cos_sin, kc, vc,
                                     st->context_lens, st->block_tables, st->table_stride, rows, md->num_q_heads, md->num_kv_heads,
                                     md->head_dim, scale, attn, dsplits, st->decode_min_range, ws, stream);
}

#define FWD(call)                 \
    do {                          \
        int32_t rc__ = (call);    \
        if (rc__ != NVH_OK) return rc__; \
    } while (0)

// ---- two steps in flight: per-layer events between consecutive steps on different streams (nvh_step.layer_events_*)
namespace {
constexpr int kMaxDevices = 16, kEventSets = 2, kMaxEventLayers = 128;
struct LayerEvents {
    hipEvent_t ev[kEventSets][kMaxEventLayers];
    int layers = 0;                      // 0: not prepared
};
LayerEvents g_layer_events[kMaxDevices];

#define HIPCHK(call, what)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            nvh_set_error("%s: %s", what, hipGetErrorString(e__));                      \
            return NVH_ELAUNCH;                                                         \
        }                                                                               \
    } while (0)
}  // namespace

extern "C" int32_t nvh_qwen3_prepare_layer_events(int32_t layers) {
    int dev = -1;
    HIPCHK(hipGetDevice(&dev), "nvh_qwen3_prepare_layer_events: hipGetDevice");
    NVH_REQUIRE(dev >= 0 && dev < kMaxDevices, NVH_EINVAL, "nvh_qwen3_prepare_layer_events: device %d out of range", dev);
    NVH_REQUIRE(layers >= 1 && layers <= kMaxEventLayers, NVH_EINVAL, "nvh_qwen3_prepare_layer_events: %d layers (1..%d)", layers, kMaxEventLayers);
    LayerEvents& L = g_layer_events[dev];
    for (int l = L.layers; l < layers; ++l) {
        for (int s = 0; s < kEventSets; ++s) HIPCHK(hipEventCreateWithFlags(&L.ev[s][l], hipEventDisableTiming), "nvh_qwen3_prepare_layer_events: event");
        L.layers = l + 1;
    }
    return L.layers;
}

// The step is a chain of parts separated by the sums over the tensor-parallel ranks:
//   part 0            embedding                                        (its sum: the residual stream)
//   part 2l+1         layer l: input norm .. O projection             (its sum: the projection output)
//   part 2l+2         layer l: post-attention norm .. down projection (its sum: the projection output)
//   part 2L+1         final norm
// `external`: the caller sums between the parts (a collective library on the same stream); no split-K slab hand-off
// then, because partial sums of different ranks meet in bf16.
static int32_t forward_impl(const nvh_qwen3_model* md, const nvh_step* st, void* hidden_out, void* workspace,
                            int64_t workspace_bytes, nvh_stream_t stream, int first_part, int last_part, bool external) {
    NVH_REQUIRE(md && st && hidden_out, NVH_EINVAL, "nvh_qwen3_forward: null pointer");
    NVH_REQUIRE(md->head_dim == 128, NVH_ENOSYS, "nvh_qwen3_forward: head_dim %d unsupported", md->head_dim);
    const int64_t t = st->num_tokens, b = st->batch;
    if (t == 0) return NVH_OK;
    const int32_t dsplits = st->decode_splits < 1 ? 1 : st->decode_splits;
    NVH_REQUIRE(workspace && nvh_aligned16(workspace), NVH_EINVAL, "nvh_qwen3_forward: workspace missing or misaligned");
    // one-token sequences at the head of a mixed step go through the decode kernel (one token = one row there)
    const int64_t prefix = st->mode == NVH_MODE_PAGED ? st->decode_prefix : 0;
    NVH_REQUIRE(prefix >= 0 && prefix <= b && prefix <= t, NVH_EINVAL, "nvh_qwen3_forward: decode_prefix %ld outside the batch", (long)prefix);
    NVH_REQUIRE(prefix == 0 || (st->context_lens && st->slot_mapping && st->cu_seqlens_q_tail && st->cu_seqlens_k_tail &&
                                md->k_cache && md->v_cache), NVH_EINVAL,
                "nvh_qwen3_forward: decode_prefix needs a KV cache, context_lens, slot_mapping and the tail offsets");
    int32_t vsplits = 1;
    if (st->mode != NVH_MODE_DECODE && b > prefix) {
        vsplits = nvh_attn_varlen_suggest_splits(t - prefix, b - prefix, md->num_q_heads, md->num_kv_heads, st->max_seqlen_k);
        if (vsplits > kMaxVarlenSplits) vsplits = kMaxVarlenSplits;
    }
    Workspace w = carve(md, t, b, dsplits, kMaxVarlenSplits, (char*)workspace);
    NVH_REQUIRE(w.total <= workspace_bytes, NVH_EINVAL, "nvh_qwen3_forward: workspace too small (%ld < %ld)",
                (long)workspace_bytes, (long)w.total);
    NVH_REQUIRE(st->mode == NVH_MODE_PACKED || st->block_tables, NVH_EINVAL, "nvh_qwen3_forward: paged modes need block_tables");
    NVH_REQUIRE(st->mode != NVH_MODE_DECODE || (st->context_lens && t == b), NVH_EINVAL, "nvh_qwen3_forward: decode mode needs context_lens and one token per sequence");
    // Two steps in flight (see nvh_step.layer_events_wait): this step's event sets on the current device
    hipEvent_t *ev_wait = nullptr, *ev_record = nullptr;
    if (st->layer_events_wait || st->layer_events_record) {
        int dev = -1;
        NVH_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices && g_layer_events[dev].layers >= md->layers, NVH_EINVAL,
                    "nvh_qwen3_forward: layer events asked for but not prepared (nvh_qwen3_prepare_layer_events)");
        NVH_REQUIRE(st->layer_events_wait >= 0 && st->layer_events_wait <= kEventSets && st->layer_events_record >= 0 &&
                        st->layer_events_record <= kEventSets && (st->layer_events_wait == 0 || st->layer_events_wait != st->layer_events_record),
                    NVH_EINVAL, "nvh_qwen3_forward: layer event sets %d / %d", st->layer_events_wait, st->layer_events_record);
        if (st->layer_events_wait) ev_wait = g_layer_events[dev].ev[st->layer_events_wait - 1];
        if (st->layer_events_record) ev_record = g_layer_events[dev].ev[st->layer_events_record - 1];
    }
    const int64_t h = md->hidden, d = md->head_dim, hq = md->num_q_heads, hkv = md->num_kv_heads;
    const int64_t qsz = hq * d, kvsz = hkv * d, qkv_n = qsz + 2 * kvsz;
    const float scale = 1.0f / sqrtf((float)d);
    // partial sums of different ranks meet in bf16: no split-K slab hand-off under tensor parallelism
    const bool sharded = md->comm || external;
    // decode batches whose add + norm happens in the next projection's prologue: no slabs either (every workgroup of
    // that projection would sum them again); the O / down projections then split K between the waves of a workgroup
    const bool no_full_k = nvh_switch_on(SW_DECODE_SLABS);                        // A/B switch for measurements and tests
    // (an answer of 2 - 33..64 rows - only for pure decode steps: every decode step of that size then takes the same route)
    const int32_t fused_limit = st->mode == NVH_MODE_DECODE ? 2 : 1;
    auto fused_ok = [&](int64_t n, int64_t slabs, int32_t swiglu) {
        const int32_t r = nvh_add_rmsnorm_gemm_fused(t, n, h, slabs, swiglu);
        return r >= 1 && r <= fused_limit;
    };
    const bool full_k = !no_full_k && t <= 64 && !md->comm && fused_ok(qkv_n, 0, 0);
    const int32_t so = sharded || full_k ? 1 : split_for(t, h, qsz), sd = sharded || full_k ? 1 : split_for(t, h, md->inter);
    auto runs = [&](int part) { return part >= first_part && part <= last_part; };
    const int32_t sq = split_for_qkv(t, qkv_n, h);

    nvh_comm* comm = md->comm;
    bf16* proj_out = w.y;               // where the row-parallel projections write their bf16 result
    if (comm) {
        NVH_REQUIRE(t * h * 2 <= nvh_comm_max_bytes(comm), NVH_EINVAL, "nvh_qwen3_forward: %ld tokens exceed the exchange staging area", (long)t);
        proj_out = (bf16*)nvh_comm_buffer(comm);   // partial sums are produced straight into the exported area
    }
    // decode-sized messages: the sum, the residual add and the norm are one kernel; larger ones keep the two-shot
    // all-reduce (less xGMI traffic) followed by the norm
    const bool no_fused = nvh_switch_on(SW_NO_FUSED_EXCHANGE);
    const bool fused_exchange = comm && !no_fused && t * h * 2 <= (int64_t)(1 << 20);
    if (runs(0)) {
        FWD(nvh_embedding(st->input_ids, md->embed, w.residual, t, h, md->vocab_start, md->vocab_rows, stream));
        if (comm) FWD(nvh_allreduce_bf16(comm, w.residual, w.residual, t * h, stream));
    }
    // what the down projection of the previous layer left behind: split-K slabs in w.slabs, or a bf16 result in w.y
    // (summed over the ranks already, unless the fused exchange takes the partial sums straight from proj_out)
    const bool pending_slabs = sd > 1;
    const int32_t pending_count = sd > 1 ? sd : 0;
    bf16* const pending_y = sd > 1 ? nullptr : w.y;
    // Decode batches without the built-in exchange: the add + norm in front of the QKV and the gate/up projection
    // happens in that projection's own prologue (one launch less each).  Such a launch reads the residual from one
    // buffer and writes it to the other, so the residual stream alternates between w.residual and w.residual2; which
    // one holds it follows from the number of fused sites passed (the same for every layer, so a step cut into
    // parts agrees with itself).
    const bool fuse_in = !comm && t <= 64 && sq == 1 && fused_ok(qkv_n, pending_count, 0);
    const bool fuse_post = !comm && t <= 64 && fused_ok(2 * md->inter, so > 1 ? so : 0, 1) && fused_ok(2 * md->inter, so > 1 ? so : 0, 0);
    auto residual_after = [&](int in_sites, int post_sites) {
        const int swaps = (fuse_in ? in_sites : 0) + (fuse_post ? post_sites : 0);
        return (swaps & 1) ? w.residual2 : w.residual;
    };
    // Prefill-sized single-rank steps: the two RMSNorms of a layer as a row scale inside the projections that consume them
    // (see nvh_gemm_residual_rowsq_bf16_nt).  The O and down projections add their product to the residual stream in place and
    // leave per-row sums of squares (in w.y: those projections write no bf16 result on this route), the QKV and gate/up
    // projections read the residual itself with the folded weights.  Layer 0's input norm stays a launch (the embedding
    // leaves no sums), the final norm becomes a plain RMSNorm of the residual.
    const int32_t parts_o = nvh_gemm_residual_rowsq_parts(t, h, qsz), parts_d = nvh_gemm_residual_rowsq_parts(t, h, md->inter);
    const bool rowscale = !nvh_switch_on(SW_NO_ROWSCALE) && md->qkv_w_scaled && md->gate_up_scaled && md->gate_up_inter && !sharded && t > 64 &&
                          so == 1 && sd == 1 && sq == 1 && d == 128 && st->mode != NVH_MODE_DECODE && t > prefix && parts_o > 0 && parts_d > 0 && parts_o <= 32 && parts_d <= 32 &&
                          nvh_gemm_rowscaled_supported(t, qkv_n, h, 2) && nvh_gemm_rowscaled_supported(t, 2 * md->inter, h, 1) && (int64_t)(parts_o > parts_d ? parts_o : parts_d) * 4 <= h * 2;
    float* const rowsq = reinterpret_cast<float*>(w.y);
    for (int l = 0; l < md->layers; ++l) {
      if (runs(2 * l + 1)) {
        // the step before this one (on another stream) is done with the attention of this layer: its keys are in the
        // cache, and it no longer reads a page this step's KV store may write
        if (ev_wait) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, ev_wait[l], 0), "nvh_qwen3_forward: layer event wait");
        // ---- input norm: layer 0 normalises the embedding; later layers add the MLP output first
        bf16* const res = residual_after(l, l);                     // input sites of layers 0..l-1, post sites of layers 0..l-1
        bool qkv_done = false;
        if (rowscale && l > 0) {
            // (the down projection of layer l - 1 has added its product to `res` and left the row sums)
        } else if (fuse_in) {
            // (layer 0: nothing to add - the embedding rows are the residual stream)
            FWD(nvh_add_rmsnorm_gemm_bf16_nt(l == 0 ? nullptr : pending_y, l > 0 && pending_slabs ? w.slabs : nullptr, l == 0 ? 0 : pending_count, res,
                                             residual_after(l + 1, l), md->input_norm[l], md->eps, md->qkv_w[l], w.qkv, qkv_n, t, qkv_n, h, 0, stream));
            qkv_done = true;
        } else if (l == 0) {
            FWD(nvh_rmsnorm(res, h, md->input_norm[l], w.x, h, t, h, md->eps, stream));
        } else if (comm && fused_exchange) {
            FWD(nvh_allreduce_add_rmsnorm_bf16(comm, proj_out, res, md->input_norm[l], w.x, t, h, md->eps, stream));
        } else {
            FWD(nvh_add_rmsnorm(pending_y, res, md->input_norm[l], w.x, t, h, md->eps,
                                pending_slabs ? w.slabs : nullptr, pending_count, stream));
        }
        // ---- attention block
        void* kc = md->k_cache ? md->k_cache[l] : nullptr;
        void* vc = md->v_cache ? md->v_cache[l] : nullptr;
        const bool store = kc && vc && st->slot_mapping;
        const bool fuse_decode = !nvh_switch_on(SW_NO_FUSED_DECODE);   // A/B switch for measurements
        const bool decode_fused = st->mode == NVH_MODE_DECODE && store && fuse_decode;
        bool qk_done = false;               // the q/k epilogue and the KV store of rows prefix.. already happened
        if (qkv_done) {
            // (projected in the launch above)
        } else if (sq > 1) {
            FWD(nvh_gemm_bf16_nt(w.x, h, md->qkv_w[l], nullptr, nullptr, qkv_n, t, qkv_n, h, sq, w.slabs, stream));
            FWD(nvh_splitk_reduce(w.slabs, sq, nullptr, w.qkv, qkv_n, t, qkv_n, stream));
        } else if (rowscale && l > 0) {
            FWD(nvh_gemm_qkv_rope_store_rowscaled(res, h, md->qkv_w_scaled[l], w.qkv, qkv_n, t, h, hq, hkv, d, md->q_norm[l], md->k_norm[l], md->eps,
                                                  st->positions, md->cos_sin, md->max_position, store ? kc : nullptr, store ? vc : nullptr,
                                                  store ? st->slot_mapping : nullptr, prefix, rowsq, parts_d, md->eps, stream));
            qk_done = true;
        } else if (!decode_fused && t > prefix && d == 128) {
            // projection + q/k norm + RoPE + KV store in one call (one launch when the shape takes 128x128 tiles)
            FWD(nvh_gemm_qkv_rope_store(w.x, h, md->qkv_w[l], w.qkv, qkv_n, t, h, hq, hkv, d, md->q_norm[l], md->k_norm[l], md->eps,
                                        st->positions, md->cos_sin, md->max_position, store ? kc : nullptr, store ? vc : nullptr,
                                        store ? st->slot_mapping : nullptr, prefix, stream));
            qk_done = true;
        } else {
            FWD(nvh_gemm_bf16_nt(w.x, h, md->qkv_w[l], nullptr, w.qkv, qkv_n, t, qkv_n, h, 1, nullptr, stream));
        }
        if (decode_fused) {
            // one launch: q/k norm + RoPE + KV store of the new token + attention
            FWD(decode_attention(md, st, l, w.qkv, qkv_n, kc, vc, b, scale, w.attn, dsplits, w.decode_ws, stream));
        } else {
            if (t > prefix && !qk_done)
                FWD(nvh_qknorm_rope_store(w.qkv + prefix * qkv_n, qkv_n, hq, hkv, md->q_norm[l], md->k_norm[l], md->eps,
                                          st->positions + prefix, md->cos_sin, md->max_position, store ? kc : nullptr,
                                          store ? vc : nullptr, store ? st->slot_mapping + prefix : nullptr, t - prefix, stream));
            if (st->mode == NVH_MODE_DECODE) {
                FWD(nvh_attn_decode(w.qkv, qkv_n, kc, vc, st->context_lens, st->block_tables, st->table_stride, b, hq, hkv,
                                    d, scale, w.attn, dsplits, st->decode_min_range, w.decode_ws, stream));
            } else if (prefix > 0) {
                // (the qk-norm / rope / store launch above covered only the tokens past the one-token sequences)
                FWD(decode_attention(md, st, l, w.qkv, qkv_n, kc, vc, prefix, scale, w.attn, dsplits, w.decode_ws, stream));
                if (b > prefix)
                    FWD(nvh_attn_varlen(w.qkv + prefix * qkv_n, qkv_n, kc, vc, kvsz, st->cu_seqlens_q_tail, st->cu_seqlens_k_tail,
                                        st->block_tables + prefix * st->table_stride, st->table_stride, b - prefix, st->max_seqlen_q,
                                        st->max_seqlen_k, t - prefix, hq, hkv, d, scale, w.attn + prefix * qsz, vsplits, w.varlen_ws, stream));
            } else if (st->mode == NVH_MODE_PAGED) {
                FWD(nvh_attn_varlen(w.qkv, qkv_n, kc, vc, kvsz, st->cu_seqlens_q, st->cu_seqlens_k, st->block_tables,
                                    st->table_stride, b, st->max_seqlen_q, st->max_seqlen_k, t, hq, hkv, d, scale, w.attn, vsplits, w.varlen_ws, stream));
            } else {
                FWD(nvh_attn_varlen(w.qkv, qkv_n, w.qkv + qsz, w.qkv + qsz + kvsz, qkv_n, st->cu_seqlens_q, st->cu_seqlens_k,
                                    nullptr, 0, b, st->max_seqlen_q, st->max_seqlen_k, t, hq, hkv, d, scale, w.attn, vsplits, w.varlen_ws, stream));
            }
        }
        if (ev_record) HIPCHK(hipEventRecord(ev_record[l], (hipStream_t)stream), "nvh_qwen3_forward: layer event record");
        // ---- O projection
        if (rowscale) FWD(nvh_gemm_residual_rowsq_bf16_nt(w.attn, qsz, md->o_w[l], residual_after(l + 1, l), rowsq, t, h, qsz, stream));
        else if (so > 1) FWD(nvh_gemm_bf16_nt(w.attn, qsz, md->o_w[l], nullptr, nullptr, h, t, h, qsz, so, w.slabs, stream));
        else FWD(nvh_gemm_bf16_nt(w.attn, qsz, md->o_w[l], nullptr, proj_out, h, t, h, qsz, 1, nullptr, stream));
      }
      if (runs(2 * l + 2)) {
        // ---- add + norm
        bf16* const res = residual_after(l + 1, l);                 // input sites of layers 0..l, post sites of layers 0..l-1
        const bool interleaved = md->gate_up_inter && md->gate_up_inter[l];
        if (rowscale) {
            FWD(nvh_gemm_swiglu_rowscaled_bf16_nt(res, h, md->gate_up_scaled[l], w.act, md->inter, t, 2 * md->inter, h, rowsq, parts_o, md->eps, stream));
            FWD(nvh_gemm_residual_rowsq_bf16_nt(w.act, md->inter, md->down_w[l], res, rowsq, t, h, md->inter, stream));
        } else {
        if (fuse_post) {
            FWD(nvh_add_rmsnorm_gemm_bf16_nt(so > 1 ? nullptr : w.y, so > 1 ? w.slabs : nullptr, so > 1 ? so : 0, res, residual_after(l + 1, l + 1),
                                             md->post_norm[l], md->eps, interleaved ? md->gate_up_inter[l] : md->gate_up_w[l],
                                             interleaved ? w.act : w.gate_up, interleaved ? md->inter : 2 * md->inter, t, 2 * md->inter, h,
                                             interleaved ? 1 : 0, stream));
            if (!interleaved) FWD(nvh_silu_mul(w.gate_up, w.act, t, md->inter, stream));
        } else if (so > 1) {
            FWD(nvh_add_rmsnorm(nullptr, res, md->post_norm[l], w.x, t, h, md->eps, w.slabs, so, stream));
        } else {
            if (comm && fused_exchange) {
                FWD(nvh_allreduce_add_rmsnorm_bf16(comm, proj_out, res, md->post_norm[l], w.x, t, h, md->eps, stream));
            } else {
                if (comm) FWD(nvh_allreduce_bf16(comm, proj_out, w.y, t * h, stream));
                FWD(nvh_add_rmsnorm(w.y, res, md->post_norm[l], w.x, t, h, md->eps, nullptr, 0, stream));
            }
        }
        // ---- MLP
        if (fuse_post) {
            // (gate/up projected in the launch above)
        } else if (md->gate_up_inter && md->gate_up_inter[l]) {
            FWD(nvh_gemm_swiglu_bf16_nt(w.x, h, md->gate_up_inter[l], w.act, md->inter, t, 2 * md->inter, h, stream));
        } else {
            FWD(nvh_gemm_bf16_nt(w.x, h, md->gate_up_w[l], nullptr, w.gate_up, 2 * md->inter, t, 2 * md->inter, h, 1, nullptr, stream));
            FWD(nvh_silu_mul(w.gate_up, w.act, t, md->inter, stream));
        }
        if (sd > 1) {
            FWD(nvh_gemm_bf16_nt(w.act, md->inter, md->down_w[l], nullptr, nullptr, h, t, h, md->inter, sd, w.slabs, stream));
        } else {
            FWD(nvh_gemm_bf16_nt(w.act, md->inter, md->down_w[l], nullptr, proj_out, h, t, h, md->inter, 1, nullptr, stream));
            if (comm && !fused_exchange) FWD(nvh_allreduce_bf16(comm, proj_out, w.y, t * h, stream));
        }
        }
      }
    }
    if (runs(2 * md->layers + 1)) {
        bf16* const res = residual_after(md->layers, md->layers);
        if (rowscale) {
            FWD(nvh_rmsnorm(res, h, md->final_norm, hidden_out, h, t, h, md->eps, stream));
        } else if (comm && fused_exchange) {
            FWD(nvh_allreduce_add_rmsnorm_bf16(comm, proj_out, res, md->final_norm, hidden_out, t, h, md->eps, stream));
        } else {
            FWD(nvh_add_rmsnorm(pending_y, res, md->final_norm, hidden_out, t, h, md->eps,
                                pending_slabs ? w.slabs : nullptr, pending_count, stream));
        }
    }
    return NVH_OK;
}

extern "C" int32_t nvh_qwen3_forward(const nvh_qwen3_model* md, const nvh_step* st, void* hidden_out, void* workspace,
                                     int64_t workspace_bytes, nvh_stream_t stream) {
    return forward_impl(md, st, hidden_out, workspace, workspace_bytes, stream, 0, md ? 2 * md->layers + 1 : 0, false);
}

extern "C" int32_t nvh_qwen3_forward_parts(const nvh_qwen3_model* md, const nvh_step* st, void* hidden_out, void* workspace,
                                           int64_t workspace_bytes, int32_t first_part, int32_t last_part, nvh_stream_t stream) {
    NVH_REQUIRE(md && !md->comm, NVH_EINVAL, "nvh_qwen3_forward_parts: for ranks WITHOUT the built-in exchange (model->comm must be null)");
    NVH_REQUIRE(first_part >= 0 && first_part <= last_part && last_part <= 2 * md->layers + 1, NVH_EINVAL,
                "nvh_qwen3_forward_parts: parts %d..%d outside 0..%d", first_part, last_part, 2 * md->layers + 1);
    return forward_impl(md, st, hidden_out, workspace, workspace_bytes, stream, first_part, last_part, true);
}

extern "C" int32_t nvh_qwen3_exchange_buffers(const nvh_qwen3_model* md, int64_t num_tokens, int64_t batch, int32_t decode_splits,
                                              void* workspace, void** embedding_sum, void** projection_sum) {
    NVH_REQUIRE(md && workspace && embedding_sum && projection_sum, NVH_EINVAL, "nvh_qwen3_exchange_buffers: null pointer");
    const Workspace w = carve(md, num_tokens, batch, decode_splits < 1 ? 1 : decode_splits, kMaxVarlenSplits, (char*)workspace);
    *embedding_sum = w.residual;
    *projection_sum = w.y;
    return NVH_OK;
}

extern "C" int32_t nvh_qwen3_logits(const nvh_qwen3_model* md, const void* hidden, const int32_t* cu_seqlens_q,
                                    const int32_t* need, int64_t num_rows, void* rows_ws, void* logits,
                                    nvh_stream_t stream) {
    NVH_REQUIRE(md && hidden && (logits || md->comm), NVH_EINVAL, "nvh_qwen3_logits: null pointer");
    if (num_rows == 0) return NVH_OK;
    const void* rows = hidden;
    if (cu_seqlens_q) {
        NVH_REQUIRE(rows_ws, NVH_EINVAL, "nvh_qwen3_logits: rows_ws missing");
        FWD(nvh_gather_last_rows(hidden, cu_seqlens_q, need, rows_ws, num_rows, md->hidden, stream));
        rows = rows_ws;
    }
    if (md->comm) {
        NVH_REQUIRE(num_rows * md->vocab_rows * 2 <= nvh_comm_max_bytes(md->comm), NVH_EINVAL,
                    "nvh_qwen3_logits: %ld rows of vocabulary shard exceed the exchange staging area", (long)num_rows);
        void* shard = nvh_comm_buffer(md->comm);
        FWD(nvh_gemm_bf16_nt(rows, md->hidden, md->lm_head, nullptr, shard, md->vocab_rows, num_rows, md->vocab_rows,
                             md->hidden, 1, nullptr, stream));
        return nvh_gather_cols_bf16(md->comm, shard, logits, num_rows, md->vocab_rows, 0, stream);
    }
    return nvh_gemm_bf16_nt(rows, md->hidden, md->lm_head, nullptr, logits, md->vocab_rows, num_rows, md->vocab_rows,
                            md->hidden, 1, nullptr, stream);
}
