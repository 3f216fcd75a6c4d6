/* libmebt_hip.so — C ABI of the MI355X-native (gfx950) MeBT transformer hot path.
 *
 * This is the drop-in boundary for the path BASELINE.json:north_star names: everything below
 * `mebt.transformer.Net2NetTransformer.forward / reconstruct_mask / shared_step / sample*` and the
 * optimiser step of the reference (Ugness/MeBT) — i.e. what the reference hands to ATen / cuBLAS /
 * NCCL through torch — as hand-written HIP kernels.  Plain pointers and sizes only: no torch types.
 * Every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library never
 * allocates device memory: its only scratch — the GEMM tuner's cache-flush buffer and the split-K
 * slabs — is part of the workspace the caller sizes with mebt_workspace_bytes), `stream` is a
 * hipStream_t passed as an opaque pointer (NULL = default stream).  Entry points launch on that
 * stream only and do not synchronise with the host, with ONE exception: in bf16 mode the first
 * launch of a GEMM signature the process has not seen times its tile candidates on the caller's
 * stream (hipEventSynchronize) before it returns.  mebt_gemm_autotune(0), MEBT_GEMM_AUTOTUNE=0, a
 * populated MEBT_GEMM_TUNE_CACHE or a table merged with mebt_gemm_tune_import (the Python host side
 * merges the shipped mebt_amd/tune/gfx950.txt) remove that exception for the signatures they cover;
 * then every entry point is capturable.
 * Re-entrancy: one thread per model handle + workspace; the tuned-configuration table shared by
 * all handles is a mutex-guarded shape -> configuration cache.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 * All functions return 0 on success, non-zero on error (MEBT_E*); mebt_last_error() holds a
 * thread-local message (bad shape / unsupported dtype / HIP error string).  The Python host side
 * (mebt_amd/_lib.py) turns a non-zero status into RuntimeError, mirroring the reference's
 * exceptions / asserts.
 */
#ifndef MEBT_HIP_H
#define MEBT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEBT_ABI_VERSION 2
#define MEBT_MAX_LAYERS 128

typedef void* mebt_stream_t;            /* hipStream_t */
typedef struct mebt_model mebt_model;    /* opaque */

enum mebt_status { MEBT_STATUS_OK = 0, MEBT_STATUS_EINVAL = 1, MEBT_STATUS_ESHAPE = 2, MEBT_STATUS_EHIP = 3,
                   MEBT_STATUS_EWORKSPACE = 4, MEBT_STATUS_EDTYPE = 5 };
enum mebt_dtype { MEBT_DTYPE_F32 = 0, MEBT_DTYPE_BF16 = 1, MEBT_DTYPE_F16 = 2 /* 3D-VQGAN operators only */ };
/* Block routing modes, reference mebt/modules/gpt.py:164-179 */
enum mebt_mode { MEBT_MODE_LATENT_ENC = 0, MEBT_MODE_LATENT_SELF = 1, MEBT_MODE_LATENT_DEC = 2, MEBT_MODE_LT2L = 3,
                 MEBT_MODE_MASKGIT = 4 /* full attention over cat[contexts, targets]: the padding mode of gpt.py:176-178,191-192,208-209 */ };

/* GPT hyper-parameters (reference mebt/modules/gpt.py:200-220, mebt/transformer.py:105-140). */
typedef struct mebt_model_desc {
    int32_t n_layer, n_head, n_embd, vocab, n_latent /* sos_emb */, block_size;
    int32_t dtype;                       /* compute precision: MEBT_DTYPE_BF16 (MFMA bf16, fp32 accumulate) or
                                            MEBT_DTYPE_F32 (exact-fp32 MFMA; the 1e-3 parity mode) */
    int32_t modes[MEBT_MAX_LAYERS];
    float label_smoothing;               /* transformer.py:97-100 */
    float embd_pdrop, resid_pdrop, attn_pdrop;   /* gpt.py:113-114,154,211 */
} mebt_model_desc;

const char* mebt_last_error(void);
int mebt_abi_version(void);

/* ---- model handle ------------------------------------------------------------------------------- */
int mebt_model_create(const mebt_model_desc* desc, mebt_model** out);
void mebt_model_destroy(mebt_model* m);
/* Flat parameter layout.  W = every nn.Linear weight (the decayed AdamW group, transformer.py:769-771),
 * per layer [query,key,value,proj,mlp.0,mlp.2] then head; P = everything else, per layer
 * [ln1.w,ln1.b,ln2.w,ln2.b,query.b,key.b,value.b,proj.b,mlp.0.b,mlp.2.b] then ln_f.w, ln_f.b,
 * mask_emb, sos_emb, pos_emb, tok_emb.  The state-dict tensors of SURVEY.md §A.2 are views into
 * these two buffers. */
int mebt_model_param_counts(const mebt_model* m, int64_t* n_w, int64_t* n_p);
/* W, P: fp32 master parameters; W_lp: bf16 mirror of W (NULL in fp32 mode); gW, gP: fp32 gradient
 * buffers of the same layouts (may be NULL for inference). */
int mebt_model_bind(mebt_model* m, float* W, void* W_lp, float* gW, float* P, float* gP);
/* Refresh the bf16 mirror from the fp32 master weights (after load_state_dict / init). */
int mebt_model_sync_lowp(mebt_model* m, mebt_stream_t stream);
int64_t mebt_workspace_bytes(const mebt_model* m, int32_t B, int32_t NC, int32_t NT, int32_t training);

/* embed + GPT.forward: replaces reference transformer.py:255-283 / :298-322 + gpt.py:234-253.
 * x_ids [B,N] i64 token grid, ci [B,NC] / ti [B,NT] i64 position sets, logits [B,NT,V] fp32 out.
 * training: bit 0 keeps the activations in `ws` for mebt_loss / mebt_backward_*, bit 1 enables
 * dropout (masks keyed by dropout_seed; ignored when all p_drop are 0), bit 2 (inference of a bf16 model only): `logits` points at
 * a bf16 [B,NT,V] buffer and the head stores its fp32 accumulators rounded to bf16 (the sampling loops' draw reads them with
 * mebt_op_sample_lp; the public logits of reconstruct_mask stay fp32). */
int mebt_forward(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t N, int32_t NC, int32_t NT,
                 const int64_t* x_ids, const int64_t* ci, const int64_t* ti, float* logits,
                 int32_t training, uint64_t dropout_seed, mebt_stream_t stream);
/* The same forward (inference, bf16 model) for the sampling loops (transformer.py:353-447, :544-663), which call it dozens of times
 * on token grids that differ in a few hundred positions: the key / value projections of the latent_enc blocks — `contexts` is
 * read-only through the network (gpt.py:187-192), so a context position's K / V row depends on its token id, its position and the
 * block's weights only — are kept in a caller-owned cache over ALL N positions, [latent_enc blocks][B][N][2 d] bf16
 * (mebt_kvcache_bytes).  Each call first recomputes the rows of the positions `dirty` [B, ND] (a subset of every sample's `ci`
 * row: embedding, LN1, projection on B * ND rows instead of B * NC) and then reads every block's keys / values at `ci`.  The caller
 * keeps the invariant that a position in `ci` is either in `dirty` or was projected by an earlier call with the token id it still
 * has; the first call of a loop passes dirty = ci.  Same logits as mebt_forward up to the bf16 rounding of a projection tiled for
 * another row count.  flags: 0, or 4 = bf16 logits. */
int64_t mebt_kvcache_bytes(const mebt_model* m, int32_t B, int32_t N);
int mebt_forward_kvcache(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t N, int32_t NC, int32_t NT,
                         const int64_t* x_ids, const int64_t* ci, const int64_t* ti, void* logits, int32_t flags,
                         void* kv_cache, const int64_t* dirty, int32_t ND, mebt_stream_t stream);
/* GPT.forward on caller-embedded inputs (reference gpt.py:234-253): sos [B,NS,d], contexts [B,NC,d],
 * targets [B,NT,d] fp32 -> logits [B,NT,V].  Inference (no activations kept); see mebt_gpt_forward_train. */
int mebt_gpt_forward(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t NC, int32_t NT, const float* sos,
                     const float* contexts, const float* targets, float* logits, mebt_stream_t stream);
/* The same in training mode: keeps the activations in `ws` for mebt_gpt_backward; dropout != 0 enables the
 * embd / attn / resid dropout sites of gpt.py:135,140,154,238-240 under `dropout_seed`. */
int mebt_gpt_forward_train(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t NC, int32_t NT, const float* sos,
                           const float* contexts, const float* targets, float* logits, int32_t dropout, uint64_t dropout_seed,
                           mebt_stream_t stream);
/* Backward of GPT.forward (the autograd backward of gpt.py:234-253) from dL/dlogits [B,NT,V] fp32 after
 * mebt_gpt_forward_train: block / ln_f / head parameter gradients into gW, gP (the embedding slices of gP are left zero) and the
 * gradients of the three embedded inputs, fp32 like the inputs (each may be NULL).  Inputs the logits do not depend on
 * get zeros. */
int mebt_gpt_backward(mebt_model* m, void* ws, const float* dlogits, float* d_sos, float* d_contexts, float* d_targets,
                      mebt_stream_t stream);
/* Masked-token loss + top-1/top-5 of the last mebt_forward: replaces F.cross_entropy(sum,
 * label_smoothing) and utils.accuracy (transformer.py:726-731, utils.py:80-94).
 * out4 (device, 4 doubles) = { CE sum, #top-1 hits, #top-5 hits, #rows }. */
int mebt_loss(mebt_model* m, void* ws, const float* logits, double* out4, mebt_stream_t stream);
/* The same, and additionally the gradient of (CE sum * loss_scale) with respect to the logits from the same pass over them (kept
 * in the workspace): a following mebt_backward_head with the same loss_scale and no upstream skips its cross-entropy backward
 * (F.cross_entropy forward + backward of transformer.py:726 in one read of the logits).  Vocabulary sizes the fused kernel does
 * not cover behave like mebt_loss. */
int mebt_loss_with_grad(mebt_model* m, void* ws, const float* logits, double* out4, float loss_scale, mebt_stream_t stream);
/* Backward of loss = CE_sum * loss_scale (* *upstream if non-NULL, a device scalar), split so the
 * caller can overlap the data-parallel all-reduce of finished gradient buckets with the rest
 * (reference: DDP reducer, train_transformer.py:39-41).  Order: head, layers hi..lo descending, embed. */
int mebt_backward_head(mebt_model* m, void* ws, const float* logits, const float* upstream, float loss_scale, mebt_stream_t stream);
/* Same, from an explicit upstream gradient dL/dlogits [B,NT,V] fp32 (caller-computed loss). */
int mebt_backward_head_dlogits(mebt_model* m, void* ws, const float* dlogits, mebt_stream_t stream);
int mebt_backward_layers(mebt_model* m, void* ws, int32_t layer_hi, int32_t layer_lo, mebt_stream_t stream);
int mebt_backward_embed(mebt_model* m, void* ws, mebt_stream_t stream);
/* Fused AdamW over both flat buffers + refresh of the bf16 mirror: replaces torch.optim.AdamW with
 * the 4 groups of transformer.py:790-797 (W decayed, P not).  step >= 1; grad_scale multiplies the
 * gradients (1/world_size when the all-reduce summed). Parameters the loss cannot reach (blocks
 * after the last latent_dec) are skipped like torch skips grad=None. */
int mebt_adamw_step(mebt_model* m, float* mW, float* vW, float* mP, float* vP, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, float grad_scale, mebt_stream_t stream);

/* The same update restricted to one gradient bucket — kind 0: head weight; 1: blocks layer_lo..layer_hi;
 * 2: the P tail (ln_f + embeddings); 3: everything; 4: everything except the blocks' Linear weights (after a
 * backward with the fused optimizer armed).  Buckets are disjoint: a caller may issue each on its
 * own stream as soon as that bucket's gradients are final (after its all-reduce), overlapping the HBM-bound
 * optimizer with the rest of backward. */
int mebt_adamw_range(mebt_model* m, float* mW, float* vW, float* mP, float* vP, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int32_t step, float grad_scale, int32_t kind, int32_t layer_hi,
                     int32_t layer_lo, mebt_stream_t stream);

/* The same update on an arbitrary slice [off, off + n) of W (which = 0) or P (which = 1) with the gradient passed
 * separately (`grad` = gradient of element `off`; bf16 when grad_bf16): what a data-parallel rank applies to its shard
 * of a gradient bucket after the reduce-scatter — optimizer state and fp32 masters are only touched by the owner
 * (reference: the DDP all-reduce + replicated optimizer of train_transformer.py:39-41). */
int mebt_adamw_slice(mebt_model* m, int32_t which, int64_t off, int64_t n, const void* grad, int32_t grad_bf16, float* mW,
                     float* vW, float* mP, float* vP, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int32_t step, float grad_scale, mebt_stream_t stream);

/* The same update with the slice's gradient given as `pieces` bf16 copies of it, piece j at grad_pieces + j * n elements: rank j's
 * contribution to this rank's shard as an all-to-all delivers them.  The kernel adds the pieces in fp32, in piece order, then
 * applies AdamW to the sum: 2 B per parameter on the wire like a bf16 reduce-scatter, but the cross-rank sum is taken in fp32 as
 * the reference's DDP all-reduce takes it (train_transformer.py:39-41) instead of in bf16 inside RCCL. */
int mebt_adamw_slice_pieces(mebt_model* m, int32_t which, int64_t off, int64_t n, const void* grad_pieces, int32_t pieces, float* mW,
                            float* vW, float* mP, float* vP, float lr, float beta1, float beta2, float eps, float weight_decay,
                            int32_t step, float grad_scale, mebt_stream_t stream);

/* Optimizer-in-backward (single-process training): armed with step >= 1, the next mebt_backward_layers applies
 * AdamW to every block's Linear weights INSIDE the weight-gradient launch (the gradient tile is consumed from
 * registers; it is not stored in gW, and W / the bf16 mirror / mW / vW are updated in place), which removes
 * 8 of the 34 bytes per parameter the separate optimizer pass moves and hides the rest behind the MFMA work.
 * The remaining parameters are then updated with mebt_adamw_range(kind = 4).  step <= 0 disarms.  Same math as
 * mebt_adamw_step (torch.optim.AdamW, transformer.py:790-797); not usable when gradients must be all-reduced
 * or inspected first. */
int mebt_model_set_fused_adamw(mebt_model* m, float* mW, float* vW, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int32_t step, float grad_scale);

/* Data-parallel wire format: gWb = bf16 buffer of n_w elements laid out like gW (NULL: off).  While bound (bf16 compute mode, no
 * gradient accumulation, fused optimizer disarmed) mebt_backward_* store the Linear weight gradients THERE, rounded once from
 * the fp32 MFMA accumulators, and leave gW untouched: the reduce-scatter of SURVEY.md §5.8 sends the buffer as is. */
int mebt_model_bind_wire_grads(mebt_model* m, void* gWb);

/* Sharded data parallelism (mebt_amd/parallel.py; the reference's counterpart is DDP's all-reduce, train_transformer.py:39-41):
 * the all-gathers that bring the other ranks' updated parameters may still be running when the next forward is enqueued.
 * `events[i]` (a hipEvent_t the caller recorded after the gather) must have completed before the forward touches the
 * parameters first read by block `layer[i]`; layer -1 = everything outside the Linear weights (embeddings, biases,
 * LayerNorms: read from the first kernel on), layer n_layer = the head weight.  One-shot: the next forward on this
 * model (training or inference) waits for them on its stream (hipStreamWaitEvent, no host synchronisation) and forgets
 * them.  The events stay owned by the caller and must outlive that forward's enqueue. */
int mebt_model_set_forward_waits(mebt_model* m, int32_t n, const int32_t* layer, void* const* events);
/* Gradient accumulation over micro-batches (reference train_transformer.py:46-49, Lightning's accumulate_grad_batches):
 * on = 1: the following mebt_backward_* calls ADD to gW / gP; on = 0 (default): they overwrite.  Not together with
 * mebt_model_set_fused_adamw. */
int mebt_model_set_grad_accumulate(mebt_model* m, int32_t on);

/* ---- operator entry points (building blocks; also what the parity tests call) ---------------------- */
/* C[M,N] = sum_k A(m,k) B(n,k) + bias, epilogue 0 none / 1 GELU (C=pre, C2=gelu) / 2 +aux residual /
 * 3 * gelu'(aux).  a_kc/b_kc: 1 = operand stored [rows][K], 0 = stored [K][rows].  Replaces nn.Linear
 * forward / dgrad / wgrad (gpt.py:126-128,140,150-155,248). */
int mebt_op_gemm(int32_t dtype, const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                 int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t ld_aux,
                 int32_t a_kc, int32_t b_kc, int32_t epilogue, int32_t c_f32, int32_t beta, int32_t split_k,
                 mebt_stream_t stream);
/* nn.LayerNorm(d), eps 1e-5 (gpt.py:147-148,216). */
int mebt_op_layernorm_fwd(int32_t dtype, const void* x, void* y, const float* gamma, const float* beta, float* mean,
                          float* rstd, int32_t rows, int32_t d, mebt_stream_t stream);
int mebt_op_layernorm_bwd(int32_t dtype, const void* x, const void* dy, const float* gamma, const float* mean,
                          const float* rstd, void* dx, float* dgamma, float* dbeta, int32_t rows, int32_t d,
                          mebt_stream_t stream);
/* softmax(q k^T / sqrt(hd)) v (gpt.py:131-137).  q [B,NQ,H,hd] row stride ldq etc.; lse [B,H,NQ]. */
int mebt_op_attention_fwd(int32_t dtype, const void* q, const void* k, const void* v, void* o, float* lse, int32_t B,
                          int32_t H, int32_t NQ, int32_t NK, int32_t HD, int32_t ldq, int32_t ldk, int32_t ldv,
                          int32_t ldo, int32_t force_generic, mebt_stream_t stream);
int mebt_op_attention_bwd(int32_t dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                          const void* d_o, void* dq, void* dk, void* dv, float* delta, int32_t B, int32_t H,
                          int32_t NQ, int32_t NK, int32_t HD, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                          int32_t force_generic, mebt_stream_t stream);
/* Embedding gather (transformer.py:255-277): sos/ctx/tgt [B,*,d] of element type `dtype`. */
int mebt_op_embed_fwd(int32_t dtype, const int64_t* x_ids, const int64_t* ci, const int64_t* ti, const float* tok_emb,
                      const float* pos_emb, const float* mask_emb, const float* sos_emb, void* sos, void* ctx, void* tgt,
                      int32_t B, int32_t N, int32_t NC, int32_t NT, int32_t NS, int32_t d, int32_t vocab,
                      int32_t block_size, mebt_stream_t stream);
/* Categorical draw of sample_from_logits / gumbel_sort (transformer.py:826-910) with the Exp(1) noise
 * as an explicit input: ids = argmax(p/noise), score = p[ids]; probs optional [rows,V]. */
int mebt_op_sample(const float* logits, const float* noise, float temperature, int32_t top_k, float top_p, int64_t* ids,
                   float* score, float* probs, int32_t rows, int32_t V, mebt_stream_t stream);
/* The same with q drawn INSIDE the kernel from a counter-based generator (element (row, e) of the step keyed by `seed`; CPU twin
 * oracle/closed_form.py:exp1_counter): only the logits move through HBM.  The explicit-noise entry above stays the bit-exact
 * interface for golden tests. */
int mebt_op_sample_seeded(const float* logits, uint64_t seed, float temperature, int32_t top_k, float top_p, int64_t* ids,
                          float* score, float* probs, int32_t rows, int32_t V, mebt_stream_t stream);
/* The weight gradients of one block in ONE launch (the autograd backward of nn.Linear weights and biases, gpt.py:126-128,140,150-155):
 * item i: dW_i[n_out_i, k_in_i] = dY_i^T X_i over tokens_i rows (bf16 row-major operands) -> gW + w_off[i] (fp32); bias[i] (or NULL)
 * += column sums of dY_i (atomic adds: zero it first).  fused != 0 applies AdamW in place of the store (optimizer-in-backward,
 * transformer.py:665-681,790-797): W, mW, vW (+ bf16 mirror Wlp, may be NULL) at the same offsets.  n <= 8. */
int mebt_op_wgrad_grouped(int32_t n, const void* const* dY, const void* const* X, const int32_t* n_out, const int32_t* k_in,
                          const int32_t* tokens, const int64_t* w_off, float* const* bias, float* W, float* gW, float* mW, float* vW,
                          void* Wlp, int32_t fused, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                          float grad_scale, mebt_stream_t stream);
/* kth[r] = the top_k-th largest value of row r of logits [rows, V] (1 <= top_k < V): the threshold of the reference's module-level
 * `top_k_logits` (transformer.py:891-895: everything below it becomes -inf, ties are kept).  ids_scratch: [rows] int64. */
int mebt_op_topk_threshold(const float* logits, int32_t top_k, float* kth, int64_t* ids_scratch, int32_t rows, int32_t V,
                           mebt_stream_t stream);
/* x[b, ti[b,j]] = ids[b,j]  (the sparse_coo/to_dense/where scatter of transformer.py:413-439). */
/* sample(debug=True) (transformer.py:395,426-436): the same draw for rows = B * NT target positions, each row's probabilities
 * written straight to row ti[b, j] of the [B, N, V] probability map (the reference materialises [B, NT, V] and scatter_()s it).
 * noise = NULL: Exp(1) drawn in the kernel from `seed`.  V = 16384, no top-p. */
int mebt_op_sample_scatter(const float* logits, const float* noise, uint64_t seed, float temperature, int32_t top_k, int64_t* ids,
                           float* score, float* probs_map, const int64_t* ti, int32_t B, int32_t N, int32_t NT, int32_t V,
                           mebt_stream_t stream);
/* The same draw on the logits as the head of an in-engine sampling loop wrote them: fp32 (logits_bf16 = 0) or bf16 (1: mebt_forward
 * with flag 4 — half the bytes of the [rows, V] tensor on both sides).  rows = B * NT; noise = NULL: Exp(1) drawn in the kernel from
 * `seed`; probs_map + ti (both or neither): the debug=True probability map as in mebt_op_sample_scatter.  V = 16384, no top-p.
 * draw: 0 = arg-max p / q with q ~ Exp(1) per element (the reference's arithmetic, transformer.py:826-841: bit-exact ids for an
 * injected `noise`); 1 (noise = NULL only) = inverse CDF from ONE uniform per row keyed by (seed, row) — a sample of the same
 * categorical distribution p (what the reference's draw is) without the per-element hash, logarithm and reciprocal: the production
 * draw of the sampling loops. */
int mebt_op_sample_lp(const void* logits, int32_t logits_bf16, const float* noise, uint64_t seed, float temperature, int32_t top_k,
                      int64_t* ids, float* score, float* probs_map, const int64_t* ti, int32_t B, int32_t N, int32_t NT, int32_t V,
                      int32_t draw, mebt_stream_t stream);
int mebt_op_scatter_ids(int64_t* x, const int64_t* ti, const int64_t* ids, int32_t B, int32_t N, int32_t NT,
                        mebt_stream_t stream);
/* MaskGen.generate_next_mask + gumbel_top_k (mask_sampler.py:178-246): order targets by
 * (score/sum)/noise^ctemp descending; first n_new go to the context (appended), the rest stay
 * targets in score order. */
int mebt_op_next_mask(const int64_t* ci, const int64_t* ti, const float* score, const float* noise, float ctemp,
                      int32_t n_new, int32_t B, int32_t NC, int32_t NT, int64_t* new_ci, int64_t* new_ti,
                      mebt_stream_t stream);
/* fp32 -> bf16 cast of a flat buffer (n multiple of 4). */
int mebt_op_cast_bf16(const float* src, void* dst, int64_t n, mebt_stream_t stream);

/* 0: never time GEMM candidates (measured heuristic or cached choices only: no host synchronisation anywhere);
 * 1: tune unseen signatures at their first launch (default; environment MEBT_GEMM_AUTOTUNE). */
void mebt_gemm_autotune(int32_t mode);
int32_t mebt_gemm_autotune_enabled(void);      /* the current mode */
/* The tuned-configuration table as text (one `n k_0 .. k_{n-1} value` entry per line + a version entry): export returns the bytes
 * needed incl. the terminating 0 and writes them when `cap` suffices; import merges a text (overwrite = 1: its entries replace the
 * ones this process holds; 2: the table is dropped first, i.e. replaced; 0: they only fill gaps) and returns the number of entries taken (0 for a text written by a build
 * with other variant codes).  A data-parallel job broadcasts rank 0's table after its first step so that every rank launches
 * identical kernels (reference train_transformer.py:39-46: DDP replicas run the same program); mebt_amd ships a default table
 * (mebt_amd/tune/gfx950.txt) merged with overwrite = 0 at load time, so a fresh process does not stall on in-situ tuning. */
int64_t mebt_gemm_tune_export(char* buf, int64_t cap);
int32_t mebt_gemm_tune_import(const char* text, int32_t overwrite);
/* diagnostics: the fastest few candidates (value, isolated microseconds) of every signature this process tuned, one line each
 * (`n k.. : v us v us ...`); tools/step_tune.py tries them inside the train step.  Same size protocol as mebt_gemm_tune_export. */
int64_t mebt_gemm_tune_alternatives(char* buf, int64_t cap);

/* ---- 3D-VQGAN first stage (reference mebt/vqgan.py:82-93,255-424, modules/codebook.py:52-62) ------------
 * Activations are channels-last [B, T, H, W, C] of `dtype` (MEBT_DTYPE_F16: MFMA fast mode, MEBT_DTYPE_F32: parity mode); the
 * video at the network boundary stays the reference's fp32 [B, C, T, H, W]. */
#define MEBT_CONV_MAX_TAPS 64
/* One (sub-lattice) convolution as an implicit GEMM.  Output voxel o = o' * os + pi for o' < (cT, cH, cW) reads, for tap j, the
 * input voxel clamp(o' * sm + tap[j]) (replicate padding, SamePadConv3d vqgan.py:374-398).  A stride-2 transposed convolution
 * (SamePadConvTranspose3d :401-424) is issued once per output parity class with the taps of that class.  w: [Cout][ntaps][Cin]
 * of `dtype`; bias fp32 [Cout] or NULL; resid (optional) is added to the result (ResBlock x + h, :370), laid out like `out`. */
typedef struct mebt_conv3d_desc {
    const void* in; void* out; const void* w; const float* bias; const void* resid;
    int32_t B, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout;
    int32_t cT, cH, cW;
    int32_t os[3], pi[3], sm[3];
    int32_t ntaps;
    int8_t tap[MEBT_CONV_MAX_TAPS][4];
    int32_t in_mode;    /* 0: channels-last `dtype`; 1: fp32 [B, C, T, H, W] */
    int32_t out_mode;   /* 0: channels-last `dtype`; 1: channels-last fp32; 2: fp32 [B, C, T, H, W] */
} mebt_conv3d_desc;
/* allow_mfma = 0 forces the direct fp32-FMA kernel (the on-GPU reference of the MFMA kernel). */
int mebt_op_conv3d(int32_t dtype, const mebt_conv3d_desc* d, int32_t allow_mfma, mebt_stream_t stream);
/* y = silu(GroupNorm_32(x)), eps 1e-6 (vqgan.py:255-258,17-18); stats: scratch [B, 32, 2] fp32. */
int mebt_op_groupnorm_silu(int32_t dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats, int32_t B,
                           int64_t vox_per_sample, int32_t C, mebt_stream_t stream);
/* ids[m] = argmin_j |z_m - e_j|^2 in the reference's evaluation order (codebook.py:54-58); z fp32 [M, d], embeddings fp32
 * [n_codes, d]; score (scratch [M, n_codes] fp32) receives z E^T from the exact-fp32 MFMA GEMM, esq scratch [n_codes]. */
int mebt_op_codebook_argmin(const float* z, const float* embeddings, float* score, float* esq, int64_t* ids, int32_t M,
                            int32_t n_codes, int32_t d, mebt_stream_t stream);
/* The same ids through a filtered search: approximate scores on the bf16 MFMA GEMM (stored as bf16), then the exact fp32 distance of
 * every code that can still be the arg-min given the rounding bound (never a reduced-precision decision).  score: scratch of at
 * least M * n_codes * 2 bytes (the exact search's fp32 matrix qualifies); lowp: scratch of (M + n_codes) * d bf16 values; esq:
 * scratch [n_codes + 1]; embedding_dim a multiple of 64. */
int mebt_op_codebook_argmin_filtered(const float* z, const float* embeddings, void* score, float* esq, void* lowp, int64_t* ids,
                                     int32_t M, int32_t n_codes, int32_t d, mebt_stream_t stream);
/* out[m, :] = embeddings[ids[m], :] (F.embedding of VQGAN.decode, vqgan.py:91), out of `dtype`. */
int mebt_op_embedding_rows(int32_t dtype, const int64_t* ids, const float* embeddings, void* out, int64_t rows, int32_t d,
                           int32_t n_codes, mebt_stream_t stream);
/* fp32 -> fp16 cast of a flat buffer (n multiple of 4). */
int mebt_op_cast_f16(const float* src, void* dst, int64_t n, mebt_stream_t stream);

/* ---- instrumentation ----------------------------------------------------------------------------- */
/* Per-kernel-family timing with HIP events on the launch stream (bench.py roofline): enable, run,
 * then read {launches, total ms, total algorithmic flops} of the GEMM family (family 0), or the
 * algorithmic operand + result bytes in place of the flops (family 1); family 2 / 3: the embedding gather forward /
 * its scatter-add backward with their algorithmic bytes. */
int mebt_profile_enable(int32_t on);
int mebt_profile_read(int32_t family, double* launches, double* total_ms, double* total_flops);
/* Every GEMM-family launch recorded while profiling was enabled, one text line each (`tag dims... ms gflop`; tag g = one product:
 * M N K a_kc b_kc epilogue c_f32; p = pair launch: M0 N0 K0 M1 N1 K1 b_kc; w = grouped weight gradients: items, output Ki-elements,
 * longest K).  Returns the bytes needed incl. the terminating 0 and writes the text when `cap` suffices (negative: HIP error). */
int64_t mebt_profile_dump(char* buf, int64_t cap);
/* Data parallel, while profiling is enabled: every wait of a forward for a deferred parameter gather
 * (mebt_model_set_forward_waits) is bracketed by an event pair.  Writes up to `cap` (first layer reading the bucket, ms the
 * compute stream stood still) pairs in launch order and returns how many were recorded (negative: HIP error). */
int32_t mebt_profile_read_waits(int32_t cap, int32_t* layer, double* ms);
/* Tests only: multiplies out[0..n) (fp32, n % 4 == 0) in place by the dropout keep-scales (0 or 1/(1-p))
 * of site `site` under `seed` — fill `out` with ones to read the mask the kernels use.  Site ids: 16*layer
 * + {0 attention probabilities, 1 proj output, 2 MLP output}; 0xFFFF0/1/2 = embedded sos/contexts/targets. */
int mebt_debug_dropout_mask(uint64_t seed, uint32_t site, float p, int64_t n, float* out, mebt_stream_t stream);
/* Diagnostics (bench.py): one wave on `stream` writes {s_memtime, s_memrealtime} at its start and again after `ref_ticks_100mhz`
 * ticks of the constant 100 MHz reference counter (<= 1 s) into out4[0..3]: (out4[2] - out4[0]) / (out4[3] - out4[1]) x 100 MHz is the
 * shader clock the GPU ran at in between, under whatever load the other streams put on it. */
int mebt_debug_clock_probe(unsigned long long* out4, uint64_t ref_ticks_100mhz, mebt_stream_t stream);
/* Benchmarking / tests only: 0 = keep every launch on the caller's stream (no side stream for gradient leaves). */
void mebt_debug_side_stream(mebt_model* m, int32_t on);
/* experiment: run that second stream's work on a caller-owned stream instead (NULL: back to the internal one) */
void mebt_debug_set_side_stream(mebt_model* m, mebt_stream_t stream);
/* Benchmarking / tests only: force the bf16 GEMM block tile (bm, bn in {128, 64}); (0, 0) restores the heuristic. */
void mebt_debug_gemm_tile(int32_t bm, int32_t bn);
/* Benchmarking / tests only: force the bf16 GEMM staging: 0 register-staged, 2 LDS-DMA 2 stages, 1 LDS-DMA
 * 3-stage ring; -1 restores the measured heuristic. */
void mebt_debug_gemm_variant(int32_t dma);
/* Benchmarking only: a caller-owned device buffer (>= 496 MiB) the operator-level mebt_op_gemm may use as tuner /
 * split-K scratch (model-level entry points use their workspace); NULL removes it. */
void mebt_debug_gemm_scratch(void* buf, int64_t bytes);
/* Diagnostics only: with a device buffer of 4 x 8 bytes per workgroup installed, wave 0 of every workgroup of the plain bf16
 * GEMM kernels stamps s_memtime at entry / first k-tile landed / main loop done / epilogue stores retired (NULL: off). */
void mebt_debug_gemm_stamps(unsigned long long* buf);
/* Diagnostics (tools/wgrad_bench.py): every grouped weight-gradient launch uses this tile (128 or 64 each way) and LDS ring depth
 * (2-4) instead of the tuned / shipped choice; tbm = 0 switches the override off. */
void mebt_debug_grouped_config(int32_t tbm, int32_t tbn, int32_t ring);
/* Benchmarking only: LDS-DMA ring depth (2 or 3) of the grouped weight-gradient GEMM. */
void mebt_debug_grouped_stages(int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* MEBT_HIP_H */
