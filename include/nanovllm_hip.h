/*
 * nanovllm_hip.h - C ABI of libnanovllm_hip.so: gfx950 (MI355X / CDNA4) kernels for
 * the nano-vllm operator path.
 *
 * The reference has no FFI layer; each entry point below replaces one delegation the
 * reference makes to a third-party / generated GPU kernel.  The citation on every
 * declaration is the reference call site it stands in for (paths are relative to the
 * reference repository).
 *
 * Conventions (all entry points)
 *   - return 0 on success, a negative NVH_E* code on failure; nothing throws across the
 *     ABI; nvh_last_error() returns a thread-local message for the last failure;
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr()); the
 *     library allocates nothing, never synchronises the host and launches only on the
 *     `stream` argument (a hipStream_t passed as void*), so calls can be captured into a
 *     hipGraph; scratch memory is passed in by the caller, sized by *_workspace_bytes();
 *   - activations / weights / KV cache are bf16 (uint16 bit patterns); sizes are int64
 *     element counts; strides are in ELEMENTS, given wherever the reference relies on a
 *     strided view (q/k/v are slices of the fused qkv buffer: layers/attention.py:121-123);
 *   - head_dim is 128 (every Qwen3 size); the KV page is 256 tokens (engine/sequence.py:73).
 *   - single caller thread per device; asynchronous with respect to `stream`.
 */
#ifndef NANOVLLM_HIP_H
#define NANOVLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVH_OK 0
#define NVH_EINVAL (-22)   /* bad shape / alignment / null pointer */
#define NVH_ENOSYS (-38)   /* unsupported configuration (dtype, head_dim, group size) */
#define NVH_ELAUNCH (-5)   /* hipLaunch / runtime error */
#define NVH_ENODEV (-19)   /* device is not gfx950 */

#define NVH_KV_PAGE 256
#define NVH_HEAD_DIM 128

typedef void* nvh_stream_t;

/* ---- library ---- */
int32_t nvh_version(void);
const char* nvh_last_error(void);
/* Refuses any device that is not gfx950.  Replaces the implicit CUDA device checks of
 * engine/model_runner.py:121-127. */
int32_t nvh_device_check(int32_t device_index);

/* Routing / measurement switches.  Every kernel choice is a function of the call's shapes; a switch overrides one such
 * choice for A/B measurements and for tests that compare two routes (no counterpart in the reference).  Switches are
 * read from the environment (NVH_<NAME>) once per process and only under NVH_TUNING=1; nvh_set_switch changes one at run
 * time (value < 0: unset; name with or without the NVH_ prefix; NVH_EINVAL for an unknown name).  Not synchronised with
 * launches in flight on other threads.  nvh_get_switch: the value, -1 when unset, -2 for an unknown name. */
int32_t nvh_set_switch(const char* name, int64_t value);
int64_t nvh_get_switch(const char* name);

/* ---- K1: KV-cache scatter store.  layers/attention.py:34-127 (Triton store_kvcache_kernel).
 * For token t: slot = slot_mapping[t]; slot < 0 -> skipped; otherwise cache row `slot`
 * (row_elems = Hkv*D contiguous elements) <- key[t] / value[t].  num_slots bounds-checks
 * nothing on the device; the caller guarantees slot < num_slots. */
int32_t nvh_store_kvcache(const void* key, int64_t key_token_stride,
                          const void* value, int64_t value_token_stride,
                          void* k_cache, void* v_cache, const int32_t* slot_mapping,
                          int64_t num_tokens, int64_t row_elems, nvh_stream_t stream);

/* ---- K4: RMSNorm.  layers/layernorm.py:57-93 (Inductor kernel of rms_forward).
 * y = bf16(f32(x) * rsqrt(mean(x^2) + eps)) * w, rows of `dim` contiguous elements at the
 * given row strides (x and y may alias). */
int32_t nvh_rmsnorm(const void* x, int64_t x_row_stride, const void* weight,
                    void* y, int64_t y_row_stride, int64_t rows, int64_t dim, float eps,
                    nvh_stream_t stream);

/* ---- K5: fused residual add + RMSNorm.  layers/layernorm.py:95-141 (add_rms_forward).
 * s = f32(x)+f32(residual); residual <- bf16(s) (in place); y = bf16(s*rsqrt(mean(s^2)+eps))*w.
 * When `partials` is non-null, x is ignored and bf16(sum_i partials[i]) is used in its
 * place: `num_partials` fp32 slabs of [rows, dim] written by a split-K nvh_gemm_bf16_nt. */
int32_t nvh_add_rmsnorm(const void* x, void* residual, const void* weight, void* y,
                        int64_t rows, int64_t dim, float eps,
                        const float* partials, int64_t num_partials, nvh_stream_t stream);

/* ---- K6: rotary embedding, half-split rotation in fp32.
 * layers/rotary_embedding.py:27-65,114-153 (Inductor kernel of RotaryEmbedding.forward).
 * cos_sin: fp32 [max_position, D] (cos | sin halves); positions int64 [T]; q [T,Hq,D] and
 * k [T,Hkv,D] rotated in place, token strides in elements. */
int32_t nvh_rope(const int64_t* positions, void* q, int64_t q_token_stride, int64_t num_q_heads,
                 void* k, int64_t k_token_stride, int64_t num_kv_heads,
                 const float* cos_sin, int64_t max_position, int64_t num_tokens, int64_t head_dim,
                 nvh_stream_t stream);

/* Fused per-head RMSNorm(q), RMSNorm(k) -> RoPE -> KV-cache store over the fused qkv
 * buffer [T, (Hq+2Hkv)*D]: q and k are updated in place, k and v rows are copied to their
 * cache slots (slot < 0 or null caches: no store).  Same arithmetic and rounding points as
 * nvh_rmsnorm + nvh_rope + nvh_store_kvcache in sequence, i.e. models/qwen3.py:191-197. */
int32_t nvh_qknorm_rope_store(void* qkv, int64_t token_stride, int64_t num_q_heads, int64_t num_kv_heads,
                              const void* q_norm_weight, const void* k_norm_weight, float eps,
                              const int64_t* positions, const float* cos_sin, int64_t max_position,
                              void* k_cache, void* v_cache, const int32_t* slot_mapping,
                              int64_t num_tokens, nvh_stream_t stream);

/* The fused QKV projection with its q/k epilogue and KV store in one call: Qwen3Attention.forward's
 * qkv_proj -> q_norm/k_norm -> rotary_emb (models/qwen3.py:183-195) and the store_kvcache half of
 * Attention.forward (layers/attention.py:93-127, called at :168-249).  qkv[M, (Hq+2Hkv)*128] = x[M,K] w[(Hq+2Hkv)*128, K]^T; rows
 * raw_rows.. then get nvh_qknorm_rope_store's treatment (positions, slot_mapping index rows of x; caches may be
 * null), rows below raw_rows stay the plain projection (the decode attention applies the epilogue itself).  Shapes
 * that take 128x128 GEMM tiles in a grid of at most 256 (769..1024 rows of the Qwen3-0.6B projection) do all of it
 * in the GEMM's epilogue; the others run nvh_gemm_bf16_nt and nvh_qknorm_rope_store back to back.  Bit-identical
 * either way. */
int32_t nvh_gemm_qkv_rope_store(const void* x, int64_t x_row_stride, const void* w, void* qkv, int64_t qkv_row_stride,
                                int64_t M, int64_t K, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                const void* q_norm_weight, const void* k_norm_weight, float eps,
                                const int64_t* positions, const float* cos_sin, int64_t max_position, void* k_cache,
                                void* v_cache, const int32_t* slot_mapping, int64_t raw_rows, nvh_stream_t stream);
/* RMSNorm as a row scale - no single counterpart in the reference: together these three replace, for prefill-sized steps,
 * RMSNorm.add_rms_forward (layers/layernorm.py:31-46) followed by the projection that reads its output
 * (QKVParallelLinear / MergedColumnParallelLinear, layers/linear.py).  The producer adds its product to the residual stream
 * in place (the add of add_rms_forward, same rounding points) and leaves `parts` partial sums of squares per row; the
 * consumers run over the un-normalised residual with the norm weight folded into `w_folded` (w[n][k] * norm[k], bf16)
 * and multiply row m of the product by 1 / sqrt(sum_p row_sumsq[m][p] / K + norm_eps) before their usual epilogue.
 * nvh_gemm_residual_rowsq_parts: partial sums per row the producer writes for this shape, 0 = shape not supported
 * (supported: the shapes nvh_gemm_bf16_nt runs on its two-group 64 x 64 tiles, M % 64 == 0, N % 128 == 0).
 * nvh_gemm_qkv_epilogue_in_gemm: 1 when nvh_gemm_qkv_rope_store runs its epilogue inside the GEMM for this shape (the
 * row-scaled form exists only then; otherwise it returns NVH_ENOSYS). */
int32_t nvh_gemm_residual_rowsq_parts(int64_t M, int64_t N, int64_t K);
int32_t nvh_gemm_residual_rowsq_bf16_nt(const void* x, int64_t x_row_stride, const void* w, void* residual, float* row_sumsq,
                                        int64_t M, int64_t N, int64_t K, nvh_stream_t stream);
int32_t nvh_gemm_qkv_epilogue_in_gemm(int64_t M, int64_t N, int64_t K);
/* 1 when the row-scaled gate/up + SwiGLU (flavor 1) or QKV projection (flavor 2) exists for this shape (they are built for the
 * eight-wave 128-row tiles; other shapes return NVH_ENOSYS from the calls below) */
int32_t nvh_gemm_rowscaled_supported(int64_t M, int64_t N, int64_t K, int32_t flavor);
int32_t nvh_gemm_qkv_rope_store_rowscaled(const void* x, int64_t x_row_stride, const void* w_folded, void* qkv, int64_t qkv_row_stride,
                                          int64_t M, int64_t K, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                          const void* q_norm_weight, const void* k_norm_weight, float eps,
                                          const int64_t* positions, const float* cos_sin, int64_t max_position, void* k_cache,
                                          void* v_cache, const int32_t* slot_mapping, int64_t raw_rows, const float* row_sumsq,
                                          int32_t parts, float norm_eps, nvh_stream_t stream);
int32_t nvh_gemm_swiglu_rowscaled_bf16_nt(const void* x, int64_t x_row_stride, const void* w_folded, void* y, int64_t y_row_stride,
                                          int64_t M, int64_t N, int64_t K, const float* row_sumsq, int32_t parts, float norm_eps,
                                          nvh_stream_t stream);


/* ---- K7: SiLU(x[:, :I]) * x[:, I:].  layers/activation.py:36-58.  x [rows, 2I] -> y [rows, I]. */
int32_t nvh_silu_mul(const void* x, void* y, int64_t rows, int64_t inter, nvh_stream_t stream);

/* ---- E1: embedding row gather with vocab-parallel masking.  layers/embed_head.py:95-131.
 * ids int64 [T]; table [vocab_rows, dim]; ids outside [vocab_start, vocab_start+vocab_rows)
 * produce zero rows. */
int32_t nvh_embedding(const int64_t* ids, const void* table, void* out, int64_t num_tokens,
                      int64_t dim, int64_t vocab_start, int64_t vocab_rows, nvh_stream_t stream);

/* Row selection for the LM head.  layers/embed_head.py:197-202: row i of `out` is
 * hidden[cu_seqlens_q[j+1]-1] with j = need[i] (need non-null) or j = i. */
int32_t nvh_gather_last_rows(const void* hidden, const int32_t* cu_seqlens_q, const int32_t* need,
                             void* out, int64_t num_rows, int64_t dim, nvh_stream_t stream);

/* input_ids[dst_index[i]] = sampled[src_index[i]] for i < count.  No counterpart in the reference, whose step reads
 * the sampled ids back (engine/llm_engine.py:117-126 -> scheduler.postprocess) before it builds the next step's
 * input_ids (engine/model_runner.py:438-495): an engine that schedules one step ahead launches the next step first
 * and lets it take those ids from the sampler's output on the device.  The indices must be in range (not checked
 * on the device). */
int32_t nvh_feed_token_ids(int64_t* input_ids, const int32_t* dst_index, const int64_t* sampled,
                           const int32_t* src_index, int64_t count, nvh_stream_t stream);

/* Token ids <-> three bf16 bytes each (enc[i], enc[count + i], enc[2 count + i]; exact in bf16), padded_count >= 3 count
 * entries written (ids null: all zeros).  With these, rank 0's sampled ids reach every tensor-parallel rank through the
 * SAME sum over the ranks as the activations (layers/linear.py:407-409) - the other ranks add zeros - instead of the
 * reference's host round trip (engine/model_runner.py:602-638 returns the ids to the scheduler, which sends them back
 * with the next step). */
int32_t nvh_ids_to_bytes_bf16(const int64_t* ids, void* enc, int64_t count, int64_t padded_count, nvh_stream_t stream);
int32_t nvh_bytes_bf16_to_ids(const void* enc, int64_t* ids, int64_t count, nvh_stream_t stream);

/* ---- G1-G5: y[M,N] = x[M,K] @ W[N,K]^T (+ bias[N]), fp32 accumulation, bf16 out.
 * layers/linear.py:130,198,407 and layers/embed_head.py:204 (F.linear -> cuBLAS).
 * K % 64 == 0.  split_k > 1: fp32 partial slabs [split_k, M, N] are written to `workspace`
 * instead of y (consumed by nvh_add_rmsnorm or nvh_splitk_reduce); bias must be null then. */
int32_t nvh_gemm_bf16_nt(const void* x, int64_t x_row_stride, const void* w, const void* bias,
                         void* y, int64_t y_row_stride, int64_t M, int64_t N, int64_t K,
                         int32_t split_k, float* workspace, nvh_stream_t stream);
int64_t nvh_gemm_workspace_bytes(int64_t M, int64_t N, int32_t split_k);
/* Suggested split (1 = none) for a weight-streaming GEMM of this shape on 256 CUs. */
int32_t nvh_gemm_suggest_split_k(int64_t M, int64_t N, int64_t K);
/* Which kernel(s) nvh_gemm_bf16_nt (flavor 0), nvh_gemm_swiglu_bf16_nt (1), nvh_gemm_qkv_rope_store (2; N = (Hq + 2 Hkv) *
 * 128) or nvh_lm_head_sample (3) would launch for this shape, as text in `out` ("; " between launches), without launching
 * anything.  The choice is a pure function of the shape (and of the routing switches); tests pin the tuned shapes with it. */
int32_t nvh_gemm_plan(int64_t M, int64_t N, int64_t K, int32_t split_k, int32_t flavor, char* out, int64_t out_bytes);
/* y = bf16(sum of slabs (+bias)). */
int32_t nvh_splitk_reduce(const float* partials, int64_t num_partials, const void* bias,
                          void* y, int64_t y_row_stride, int64_t M, int64_t N, nvh_stream_t stream);

/* gate_up GEMM (layers/linear.py:198 via models/qwen3.py:270) fused with the SwiGLU gate
 * (layers/activation.py:36-58): y[M, N/2] = silu(x Wg^T) * (x Wu^T), same rounding points as
 * nvh_gemm_bf16_nt followed by nvh_silu_mul.  `w_interleaved` is the merged [gate | up] weight
 * ([N = 2I, K]) re-ordered by nvh_interleave_gate_up: blocks of 16 gate rows followed by the 16 up
 * rows of the same outputs, so both land in one lane's accumulators. */
int32_t nvh_gemm_swiglu_bf16_nt(const void* x, int64_t x_row_stride, const void* w_interleaved, void* y,
                                int64_t y_row_stride, int64_t M, int64_t N, int64_t K, nvh_stream_t stream);
int32_t nvh_interleave_gate_up(const void* w, void* w_interleaved, int64_t inter, int64_t K, nvh_stream_t stream);

/* Residual add + RMSNorm + the projection that consumes it in ONE launch, for decode-sized batches: what
 * Qwen3DecoderLayer.forward does between two projections (models/qwen3.py:335-349: input_layernorm /
 * post_attention_layernorm with residual, layers/layernorm.py:95-141, then qkv_proj / gate_up_proj,
 * layers/linear.py:198).  s = f32(prev) + f32(residual_in) with prev [M, K] bf16, or bf16(sum of `num_slabs` fp32
 * split-K slabs [num_slabs, M, K]) when `slabs` is non-null, or nothing at all when both are null (the first layer's
 * input_layernorm, layers/layernorm.py:57-93: s = f32(residual_in)); residual_out <- bf16(s) (must NOT alias residual_in);
 * y[M, N] = rmsnorm(s) w[N, K]^T, or with swiglu != 0 y[M, N/2] = silu(gate) * up over a weight interleaved by
 * nvh_interleave_gate_up.  Same bits as nvh_add_rmsnorm followed by nvh_gemm_bf16_nt / nvh_gemm_swiglu_bf16_nt.
 * 1..64 rows, K a multiple of 64 up to 1024 (NVH_ENOSYS otherwise).  Every workgroup derives the rows again, so it pays only
 * while that is a few KB: nvh_add_rmsnorm_gemm_fused says whether a caller should take it - 0: no; 1: yes; 2 (33..64 rows): only
 * where every step of that size takes it (nvh_qwen3_forward does for pure decode steps), because a row count that changes route
 * changes the order of a few fp32 sums. */
int32_t nvh_add_rmsnorm_gemm_bf16_nt(const void* prev, const float* slabs, int64_t num_slabs, const void* residual_in,
                                     void* residual_out, const void* norm_weight, float eps, const void* w, void* y,
                                     int64_t y_row_stride, int64_t M, int64_t N, int64_t K, int32_t swiglu,
                                     nvh_stream_t stream);
int32_t nvh_add_rmsnorm_gemm_fused(int64_t M, int64_t N, int64_t K, int64_t num_slabs, int32_t swiglu);

/* ---- K2: paged / packed var-len causal GQA attention (prefill, chunked prefill, mixed).
 * layers/attention.py:236-248 (flash_attn_varlen_func).  q [T,Hq,D] packed by
 * cu_seqlens_q[B+1].  block_tables == NULL: k,v are the packed new tokens [Tk,Hkv,D]
 * indexed by cu_seqlens_k (token strides given); otherwise k,v are the page pools
 * [num_pages,256,Hkv,D] and block_tables is int32 [B, table_stride] (entries past
 * ceil(len/256) are never read).  Mask: query i of a sequence with Lq queries / Lk keys
 * sees keys j <= i + (Lk - Lq).  softmax in fp32, out bf16 [T,Hq,D] contiguous. */
int32_t nvh_attn_varlen(const void* q, int64_t q_token_stride,
                        const void* k, const void* v, int64_t kv_token_stride,
                        const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                        const int32_t* block_tables, int64_t table_stride,
                        int64_t batch, int64_t max_seqlen_q, int64_t max_seqlen_k, int64_t total_q,
                        int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                        float softmax_scale, void* out, int32_t num_splits, void* workspace,
                        nvh_stream_t stream);
/* num_splits > 1 cuts every sequence's keys into that many ranges of whole 64-key tiles handled by
 * separate workgroups (fills the chip when a chunk has few query blocks); needs `workspace`. */
int64_t nvh_attn_varlen_workspace_bytes(int64_t total_q, int64_t num_q_heads, int64_t head_dim, int32_t num_splits);
int32_t nvh_attn_varlen_suggest_splits(int64_t total_q, int64_t batch, int64_t num_q_heads, int64_t num_kv_heads,
                                       int64_t max_seqlen_k);

/* ---- K3: paged decode attention, one query per sequence.
 * layers/attention.py:222-234 (flash_attn_with_kvcache).  q [B,Hq,D]; context_lens int32 [B]
 * (0 = padded row -> zero output); new K/V already stored.  num_splits > 1 offers every sequence that many key
 * ranges, each handled by its own workgroup (fixed grid: safe under graph replay with changing context_lens), and
 * needs `workspace`.  A sequence of ctx keys uses ranges of max(ceil(ctx / num_splits) rounded up to 64,
 * min_range_keys) keys: min_range_keys (a multiple of 64; 0 = 64) keeps short sequences in one range, finished
 * without partial results, while a long sequence in the same batch is spread over up to num_splits workgroups. */
int32_t nvh_attn_decode(const void* q, int64_t q_token_stride,
                        const void* k_cache, const void* v_cache,
                        const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                        int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                        float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys, void* workspace,
                        nvh_stream_t stream);
/* Decode step fusion: `qkv` is the raw fused projection output [B, (Hq+2Hkv)*D] of the single new token of
 * every sequence.  One launch applies the per-head RMSNorm + RoPE to q and k (models/qwen3.py:191-195),
 * writes the new K/V rows to the cache slot of token context_lens[b]-1 (layers/attention.py:219-220) and
 * attends (:229-232).  Same bits as nvh_qknorm_rope_store followed by nvh_attn_decode; qkv is left untouched. */
int32_t nvh_attn_decode_fused_qkv(const void* qkv, int64_t token_stride, const void* q_norm_weight,
                                  const void* k_norm_weight, float eps, const int64_t* positions,
                                  const float* cos_sin, void* k_cache, void* v_cache,
                                  const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                  int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                  float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys,
                                  void* workspace, nvh_stream_t stream);
/* The same two launches over an explicit work list instead of the uniform num_splits x batch grid (no counterpart in
 * the reference, whose flash_attn_with_kvcache call splits keys internally): item i is key range item_range[i] of
 * sequence item_seq[i] (-1 = unused slot); sequence b has seq_active[b] ranges of seq_chunk[b] keys (a multiple of
 * 64), seq_active[b] <= max_ranges.  Every sequence needs an item for each of its ranges (range 0 also for padded
 * rows, which write zeros); num_items >= batch.  A sequence with one range is finished by its workgroup, the
 * others through `workspace` (nvh_attn_decode_workspace_bytes(batch, Hq, D, max_ranges)).  For batches of many
 * short sequences and a few very long ones: neither a serial walk over the long ones nor a grid of idle workgroups. */
int32_t nvh_attn_decode_items(const void* q, int64_t q_token_stride,
                              const void* k_cache, const void* v_cache,
                              const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                              int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                              float softmax_scale, void* out, const int32_t* item_seq, const int32_t* item_range,
                              const int32_t* seq_active, const int32_t* seq_chunk, int64_t num_items, int32_t max_ranges,
                              void* workspace, nvh_stream_t stream);
int32_t nvh_attn_decode_fused_qkv_items(const void* qkv, int64_t token_stride, const void* q_norm_weight,
                                        const void* k_norm_weight, float eps, const int64_t* positions,
                                        const float* cos_sin, void* k_cache, void* v_cache,
                                        const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                        int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                        float softmax_scale, void* out, const int32_t* item_seq, const int32_t* item_range,
                                        const int32_t* seq_active, const int32_t* seq_chunk, int64_t num_items,
                                        int32_t max_ranges, void* workspace, nvh_stream_t stream);
/* nvh_attn_decode for callers that cannot allocate (code under hipGraph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same kernels, same arguments and same bits as nvh_attn_decode,
 * but the per-range partial results go to a fixed device array inside the library instead of a caller workspace; the
 * call performs no allocation of any kind.  Limit: nvh_attn_decode_workspace_bytes(batch, Hq, D, num_splits) must not
 * exceed nvh_attn_decode_noalloc_max_bytes() (8 MiB; e.g. 32 sequences x 16 heads x 32 ranges), larger requests fail
 * with NVH_EINVAL.  Every GPU the library runs on holds its own copy of the array until the process ends; all calls
 * on one GPU share it, so calls that overlap there (two streams) race.  The array's device address is looked up on the
 * first call on a GPU, or by nvh_device_check() of the current device. */
int32_t nvh_attn_decode_noalloc(const void* q, int64_t q_token_stride,
                                const void* k_cache, const void* v_cache,
                                const int32_t* context_lens, const int32_t* block_tables, int64_t table_stride,
                                int64_t batch, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim,
                                float softmax_scale, void* out, int32_t num_splits, int32_t min_range_keys,
                                nvh_stream_t stream);
int64_t nvh_attn_decode_noalloc_max_bytes(void);
int64_t nvh_attn_decode_workspace_bytes(int64_t batch, int64_t num_q_heads, int64_t head_dim, int32_t num_splits);
int32_t nvh_attn_decode_suggest_splits(int64_t batch, int64_t num_kv_heads, int64_t max_context);

/* ---- K8: sampling.  layers/sampler.py:41-81.
 * nvh_sample: p = softmax(f32(logits)/T); token = argmax(p / max(E,1e-10)), E ~ Exp(1) from
 * a counter-based generator keyed by (seed, offset, row, column); evaluated as
 * argmax(logits/T - log(max(E,1e-10))), which has the same maximiser.  Rows are cut into slices
 * scanned by separate workgroups; `workspace` (nvh_sample_workspace_bytes) holds their candidates.
 * nvh_sample_with_noise: same with the noise tensor supplied (fp32 [rows, vocab]).
 * nvh_argmax: lowest index of the row maximum (the pre-sampler greedy token).
 * logits bf16 or fp32 (logits_is_f32), row stride in elements; out int64 [rows]. */
int64_t nvh_sample_workspace_bytes(int64_t rows, int64_t vocab);   /* 0: no workspace needed */
int32_t nvh_sample(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures,
                   int64_t* out, int64_t rows, int64_t vocab, uint64_t seed, uint64_t offset,
                   void* workspace, nvh_stream_t stream);
int32_t nvh_sample_with_noise(const void* logits, int32_t logits_is_f32, int64_t row_stride,
                              const float* temperatures, const float* noise, int64_t* out,
                              int64_t rows, int64_t vocab, void* workspace, nvh_stream_t stream);
int32_t nvh_argmax(const void* logits, int32_t logits_is_f32, int64_t row_stride, int64_t* out,
                   int64_t rows, int64_t vocab, void* workspace, nvh_stream_t stream);

/* Sampling over vocabulary shards without gathering the logits.  The reference gathers every rank's shard of the
 * logits on rank 0 (layers/embed_head.py:205-209) and samples there (layers/sampler.py:76-80); here every rank picks the
 * best candidate of its own shard - same noise per column of the full row (col_offset = first column of the shard, a
 * multiple of 4), same comparison rule - and leaves (score, index) as 8 bf16 bytes per row in ITS segment of
 * enc [world][rows][8], zeros in the others.  After the ranks' sum over enc (the all-reduce the activations take),
 * nvh_sample_pick gives every rank the id nvh_sample would have drawn from the gathered row. */
int64_t nvh_sample_shard_workspace_bytes(int64_t rows, int64_t shard_vocab);
int32_t nvh_sample_shard(const void* logits, int32_t logits_is_f32, int64_t row_stride, const float* temperatures, void* enc,
                         int64_t rows, int64_t shard_vocab, int64_t col_offset, int32_t rank, int32_t world,
                         uint64_t seed, uint64_t offset, void* workspace, nvh_stream_t stream);
int32_t nvh_sample_pick(const void* enc, int64_t* out, int64_t rows, int32_t world, nvh_stream_t stream);

/* LM head and sampler in one GEMM launch: ids = nvh_sample(x W^T, temperatures, seed, offset) without the logits ever
 * reaching memory (layers/embed_head.py:204 followed by layers/sampler.py:76-80 in the reference).  Each workgroup of the
 * projection scores its 256 columns with the sampler's own noise and comparison rule and leaves one candidate per row in
 * `workspace` (nvh_lm_head_sample_workspace_bytes); a second small launch picks among the column tiles.  Same ids as the
 * two separate calls.  Only for shapes that take the 256 x 256 GEMM tile: nvh_lm_head_sample_fused(M, N, K) says so (the
 * call returns NVH_ENOSYS otherwise). */
int32_t nvh_lm_head_sample_fused(int64_t M, int64_t N, int64_t K);
int64_t nvh_lm_head_sample_workspace_bytes(int64_t M, int64_t N);
int32_t nvh_lm_head_sample(const void* x, int64_t x_row_stride, const void* w, const float* temperatures, int64_t* out,
                           int64_t M, int64_t N, int64_t K, uint64_t seed, uint64_t offset, void* workspace,
                           nvh_stream_t stream);

/* ---- C1/C2: tensor-parallel exchange over xGMI peer mappings.
 * Replaces the NCCL all-reduce after every row-parallel projection (layers/linear.py:407-409) and after the
 * vocabulary-parallel embedding (layers/embed_head.py:128-129).  One rank per process and device.
 * nvh_comm_create allocates this rank's exchange region (flags + `max_bytes` of staging; the only allocation
 * in the library, made once at start-up); the ranks swap nvh_comm_get_handle() blobs of
 * nvh_comm_handle_bytes() each over any control channel and pass them, in rank order, to
 * nvh_comm_open_peers.  nvh_allreduce_bf16 is then one asynchronous, hipGraph-capturable kernel per rank:
 * `out` = sum over ranks of `in` (fp32 accumulation in rank order, so every rank gets the same bits),
 * count % 8 == 0, count*2 <= max_bytes.  `in` may be nvh_comm_buffer() itself (a producer wrote there; no
 * staging copy) or any other device pointer; `out` must lie outside the staging area and may equal `in`
 * otherwise.  All ranks must issue the same sequence of calls with the same counts.  A rank that waits more
 * than 20 s for a peer records a fault and lets its grid drain; nvh_comm_error (synchronises) reports it. */
typedef struct nvh_comm nvh_comm;
int32_t nvh_comm_handle_bytes(void);
int32_t nvh_comm_create(int32_t rank, int32_t world, int64_t max_bytes, nvh_comm** out);
int32_t nvh_comm_get_handle(nvh_comm* comm, void* handle_out);
int32_t nvh_comm_open_peers(nvh_comm* comm, const void* handles);
void* nvh_comm_buffer(nvh_comm* comm);
int64_t nvh_comm_max_bytes(nvh_comm* comm);
int32_t nvh_allreduce_bf16(nvh_comm* comm, const void* in, void* out, int64_t count, nvh_stream_t stream);
/* All-reduce fused with its consumer, the residual add + RMSNorm after a row-parallel projection
 * (layers/linear.py:407-409 followed by layers/layernorm.py:95-141): residual += sum over ranks of in[rows, dim]
 * (each rounded to bf16 as the separate calls do), y = RMSNorm(residual) * weight.  One kernel per rank, one-shot
 * exchange (meant for decode-sized messages); same bits as nvh_allreduce_bf16 + nvh_add_rmsnorm. */
int32_t nvh_allreduce_add_rmsnorm_bf16(nvh_comm* comm, const void* in, void* residual, const void* weight, void* y,
                                       int64_t rows, int64_t dim, float eps, nvh_stream_t stream);
/* C3: logits gather of ParallelLMHead.forward (layers/embed_head.py:205-209): on `root`,
 * out[rows, world*cols] = column-wise concatenation in rank order of every rank's in[rows, cols]
 * (cols % 8 == 0, rows*cols*2 <= max_bytes); `out` is ignored on the other ranks. */
int32_t nvh_gather_cols_bf16(nvh_comm* comm, const void* in, void* out, int64_t rows, int64_t cols, int32_t root,
                             nvh_stream_t stream);
int32_t nvh_comm_error(nvh_comm* comm);
int32_t nvh_comm_destroy(nvh_comm* comm);

/* ---- whole-model executor: the layer loop of models/qwen3.py:401-441 (Qwen3Model.forward) and
 * compute_logits (:509-533), issued from native code so one call enqueues every kernel of a step
 * (one rank's share of the model; with `comm` set the row-parallel outputs and the embedding are summed
 * and the logits gathered through the xGMI exchange above, otherwise the rank stands alone).
 * Same kernels, same order and same rounding points as calling the entry points above one by one:
 *   embedding -> L x { (add+)RMSNorm -> QKV GEMM -> q/k norm + RoPE + KV store -> attention ->
 *   O GEMM (split-K slabs) -> add+RMSNorm (sums slabs) -> gate/up GEMM -> SiLU*mul -> down GEMM } ->
 *   add+RMSNorm.
 * All pointer arrays have `layers` entries and live in HOST memory; they are read during the call only. */
typedef struct {
    int32_t hidden, layers, num_q_heads, num_kv_heads, head_dim, inter;
    int32_t vocab_rows, vocab_start;   /* this rank's rows of the vocabulary: [vocab_start, vocab_start + vocab_rows) */
    float eps;
    int32_t reserved2;
    int64_t max_position;
    const void* embed;                 /* [vocab_rows, hidden] */
    const void* final_norm;            /* [hidden] */
    const void* lm_head;               /* [vocab_rows, hidden] */
    const float* cos_sin;              /* fp32 [max_position, head_dim] */
    const void* const* qkv_w;          /* [(Hq+2Hkv)*D, hidden] */
    const void* const* o_w;            /* [hidden, Hq*D] */
    const void* const* q_norm;         /* [D] */
    const void* const* k_norm;         /* [D] */
    const void* const* gate_up_w;      /* [2*inter, hidden], rows [gate | up] */
    const void* const* gate_up_inter;  /* same rows in nvh_interleave_gate_up order, or NULL: unfused SiLU*mul */
    const void* const* down_w;         /* [hidden, inter] */
    const void* const* input_norm;     /* [hidden] */
    const void* const* post_norm;      /* [hidden] */
    void* const* k_cache;              /* [pages,256,Hkv,D] per layer, or NULL array entries before the pool exists */
    void* const* v_cache;
    nvh_comm* comm;                    /* NULL: single rank.  Else all sizes above are this rank's shard */
    /* Optional (NULL: off): the QKV / gate-up weights with the RMSNorm weight of their input folded in
     * (w[n][k] * norm[k] rounded to bf16; gate_up_scaled in nvh_interleave_gate_up order).  With both set, prefill-sized
     * single-rank steps whose shapes allow it apply the two RMSNorms of a layer as a row scale inside the projection that
     * consumes them (nvh_gemm_residual_rowsq_bf16_nt and the *_rowscaled entry points): two launches less per layer. */
    const void* const* qkv_w_scaled;
    const void* const* gate_up_scaled;
} nvh_qwen3_model;

#define NVH_MODE_PACKED 0   /* layers/attention.py: block_tables is None (K/V = new tokens) */
#define NVH_MODE_PAGED 1    /* var-len queries over the paged cache */
#define NVH_MODE_DECODE 2   /* one query per sequence (graph decode / eager decode) */

typedef struct {
    int64_t num_tokens, batch;
    const int64_t* input_ids;          /* [T] */
    const int64_t* positions;          /* [T] */
    const int32_t* cu_seqlens_q;       /* [B+1] (unused in NVH_MODE_DECODE) */
    const int32_t* cu_seqlens_k;       /* [B+1] (unused in NVH_MODE_DECODE) */
    const int32_t* slot_mapping;       /* [T] or NULL (no KV store) */
    const int32_t* context_lens;       /* [B] (NVH_MODE_DECODE) */
    const int32_t* block_tables;       /* [B, table_stride] or NULL */
    int64_t table_stride;
    int32_t max_seqlen_q, max_seqlen_k;
    int32_t mode, decode_splits;       /* decode_splits >= 1 */
    int32_t decode_min_range;          /* min_range_keys of the decode attention launches (0 = 64) */
    /* Mixed steps (mode PAGED) whose first `decode_prefix` sequences have exactly one new token each (the running
     * sequences of a serving step): their attention goes through the decode kernel (with `decode_splits` key
     * ranges), the remaining sequences through the var-len kernel with the two arrays below = cu_seqlens of
     * sequences decode_prefix.. rebased to start at 0, [batch - decode_prefix + 1] each.  The reference sends the
     * whole mixed batch through one var-len call (layers/attention.py:236-248); one 64-row block per one-token
     * sequence made that call half of the GPU time of a saturated serving run.  0 = no split. */
    int64_t decode_prefix;
    const int32_t *cu_seqlens_q_tail, *cu_seqlens_k_tail;
    /* decode_items > 0: the decode attention launches (mode DECODE, or the decode_prefix of a mixed step) run over
     * this work list (nvh_attn_decode_fused_qkv_items) with decode_splits as max_ranges. */
    int64_t decode_items;
    const int32_t *decode_item_seq, *decode_item_range, *decode_seq_active, *decode_seq_chunk;
    /* Two steps in flight (no counterpart in the reference, whose step is one chain of launches on one stream).  A
     * caller that launches step n+1 on ANOTHER stream before step n has finished orders the two per layer instead of as
     * a whole: step n runs with layer_events_record = s (1 or 2) and records event s[l] after the attention of layer l;
     * step n+1 runs with layer_events_wait = s and waits for s[l] before its QKV projection and KV store of layer l.
     * That is all the two steps share: by then step n's keys of that layer are in the cache (a later chunk of the same
     * prompt, or a prompt that borrows its pages, reads them) and step n no longer reads a page of that layer that step
     * n+1 may write (pages released by one step and handed to the next).  Everything else - workspace, hidden_out, the
     * step's metadata - must be the step's own.  The two steps then fill each other's launch-latency-bound time (a
     * 1024-token chunk of Qwen3-0.6B is ~200 dependent launches that cannot fill the chip one at a time).  0 = none.
     * The events live in the library (nvh_qwen3_prepare_layer_events, once per device); not for use under graph capture. */
    int32_t layer_events_wait, layer_events_record;
} nvh_step;

/* Creates the two per-layer event sets of nvh_step.layer_events_* on the CURRENT device (idempotent; the only call of
 * the executor that creates anything - make it at start-up).  Returns the number of layers covered. */
int32_t nvh_qwen3_prepare_layer_events(int32_t layers);

int64_t nvh_qwen3_workspace_bytes(const nvh_qwen3_model* model, int64_t num_tokens, int64_t batch, int32_t decode_splits);
/* hidden_out: bf16 [T, hidden] (the final-norm output). */
int32_t nvh_qwen3_forward(const nvh_qwen3_model* model, const nvh_step* step, void* hidden_out,
                          void* workspace, int64_t workspace_bytes, nvh_stream_t stream);
/* The same step for a tensor-parallel rank WITHOUT the built-in exchange (model->comm null): the sums over the ranks
 * that the reference makes with dist.all_reduce (layers/linear.py:407-409, layers/embed_head.py:128-129) are made by the
 * caller between the parts of the step, with any collective library on the same stream (RCCL through
 * torch.distributed; capturable into a hipGraph with the launches).  Parts: 0 = embedding; 2l+1 = layer l from its
 * input norm to the O projection; 2l+2 = layer l from the post-attention norm to the down projection; 2L+1 = final
 * norm.  After part 0 the caller sums *embedding_sum ([T, hidden] bf16) in place over the ranks, after every other
 * part but the last *projection_sum; nvh_qwen3_exchange_buffers names those two areas of `workspace`.  Several
 * consecutive parts may be run by one call only when no sum falls between them (first_part == last_part otherwise). */
int32_t nvh_qwen3_forward_parts(const nvh_qwen3_model* model, const nvh_step* step, void* hidden_out,
                                void* workspace, int64_t workspace_bytes, int32_t first_part, int32_t last_part,
                                nvh_stream_t stream);
int32_t nvh_qwen3_exchange_buffers(const nvh_qwen3_model* model, int64_t num_tokens, int64_t batch, int32_t decode_splits,
                                   void* workspace, void** embedding_sum, void** projection_sum);
/* logits [num_rows, vocab_rows] bf16 = lm_head(hidden[last rows]); layers/embed_head.py:197-204.
 * rows_ws: bf16 [num_rows, hidden] scratch; cu_seqlens_q NULL -> hidden rows are used as they are.
 * With `comm` set: logits is [num_rows, world*vocab_rows] on rank 0 (:205-209) and unused (may be NULL) elsewhere. */
int32_t nvh_qwen3_logits(const nvh_qwen3_model* model, const void* hidden, const int32_t* cu_seqlens_q,
                         const int32_t* need, int64_t num_rows, void* rows_ws, void* logits, nvh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NANOVLLM_HIP_H */
