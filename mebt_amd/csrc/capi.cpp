// Operator-level entry points of the C ABI (include/mebt_hip.h): thin argument marshalling onto
// the kernel launchers.  The model-level entry points live in engine.cpp.
#include "common.h"
#include "kernels.h"
#include <cmath>
#include "../../include/mebt_hip.h"
#include <string.h>

static hipStream_t S(mebt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static int check_dtype(int dtype) {
    if (dtype == MEBT_F32 || dtype == MEBT_BF16) return MEBT_OK;
    mebt_set_error("unsupported dtype (0 = f32, 1 = bf16)");
    return MEBT_EDTYPE;
}

extern "C" int mebt_op_gemm(int32_t dtype, const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                            int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int32_t ld_aux,
                            int32_t a_kc, int32_t b_kc, int32_t epilogue, int32_t c_f32, int32_t beta, int32_t split_k,
                            mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!A || !B || (!C && epilogue != EPI_GELU)) { mebt_set_error("gemm: null operand"); return MEBT_EINVAL; }
    if (epilogue < EPI_NONE || epilogue > EPI_GELU_BWD) { mebt_set_error("gemm: bad epilogue"); return MEBT_EINVAL; }
    if ((epilogue == EPI_RESID || epilogue == EPI_GELU_BWD) && !aux) { mebt_set_error("gemm: epilogue needs aux"); return MEBT_EINVAL; }
    if (epilogue == EPI_GELU && !C2) { mebt_set_error("gemm: GELU epilogue needs C2"); return MEBT_EINVAL; }
    if (M < 0 || N < 0 || K <= 0) { mebt_set_error("gemm: bad extents"); return MEBT_ESHAPE; }
    static bool inited = false;
    if (!inited) { if (int rc = gemm_init_attributes()) return rc; inited = true; }
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.C2 = C2; p.bias = bias; p.aux = aux; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ld_aux = ld_aux; p.a_kc = a_kc; p.b_kc = b_kc;
    p.epilogue = epilogue; p.c_f32 = c_f32; p.beta = beta; p.split_k = split_k;
    return launch_gemm(p, dtype, S(stream));
}

// The grouped weight-gradient launch of one block as an operator (tests, benchmarks): item i is dW_i[n_out_i, k_in_i] = dY_i^T X_i over
// tokens_i rows (dY_i [tokens, n_out] bf16, X_i [tokens, k_in] bf16), written at gW + w_off[i] (fp32); bias[i] (or NULL) += column sums of
// dY_i.  fused != 0: AdamW (torch semantics, step >= 1) is applied to W / mW / vW (+ the bf16 mirror Wlp) at the same offsets instead.
extern "C" int mebt_op_wgrad_grouped(int32_t n, const void* const* dY, const void* const* X, const int32_t* n_out, const int32_t* k_in,
                                     const int32_t* tokens, const int64_t* w_off, float* const* bias, float* W, float* gW, float* mW, float* vW,
                                     void* Wlp, int32_t fused, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                     float grad_scale, mebt_stream_t stream) {
    if (n < 1 || n > MEBT_MAX_GROUP || !dY || !X || !n_out || !k_in || !tokens || !w_off || !gW) { mebt_set_error("wgrad_grouped: bad arguments"); return MEBT_EINVAL; }
    if (fused && (!W || !mW || !vW || step < 1)) { mebt_set_error("wgrad_grouped: fused AdamW needs W, mW, vW and step >= 1"); return MEBT_EINVAL; }
    static bool inited = false;
    if (!inited) { if (int rc = gemm_init_attributes()) return rc; inited = true; }
    GroupedWgrad w;
    w.n = 0;
    for (int i = 0; i < n; ++i) {
        if (!dY[i] || !X[i] || n_out[i] <= 0 || k_in[i] <= 0 || tokens[i] <= 0 || (n_out[i] % 8) || (k_in[i] % 8)) { mebt_set_error("wgrad_grouped: bad item"); return MEBT_ESHAPE; }
        GroupedWgrad::Item& it = w.g[w.n++];
        it.A = dY[i]; it.B = X[i]; it.C = gW + w_off[i]; it.M = n_out[i]; it.N = k_in[i]; it.K = tokens[i];
        it.lda = n_out[i]; it.ldb = k_in[i]; it.ldc = k_in[i]; it.ntx = 0; it.bias = bias ? bias[i] : nullptr;
    }
    w.gW = gW;
    if (fused) {
        w.fused = 1; w.W = W; w.mW = mW; w.vW = vW; w.Wlp = Wlp;
        w.opt = {lr, beta1, beta2, eps, weight_decay, (float)(1.0 - pow((double)beta1, step)), (float)(1.0 - pow((double)beta2, step)), grad_scale};
    }
    return launch_wgrad_grouped(w, MEBT_BF16, S(stream));
}

extern "C" int mebt_op_layernorm_fwd(int32_t dtype, const void* x, void* y, const float* gamma, const float* beta, float* mean,
                                     float* rstd, int32_t rows, int32_t d, mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!x || !y || !gamma || !beta) { mebt_set_error("layernorm: null pointer"); return MEBT_EINVAL; }
    LnFwdParams p;
    p.x = x; p.y = y; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.rows = rows; p.d = d;
    p.seg = 0; p.seg_stride = 0; p.seg_off = 0;
    return launch_ln_fwd(p, dtype, S(stream));
}

extern "C" int mebt_op_layernorm_bwd(int32_t dtype, const void* x, const void* dy, const float* gamma, const float* mean,
                                     const float* rstd, void* dx, float* dgamma, float* dbeta, int32_t rows, int32_t d,
                                     mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!x || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) { mebt_set_error("layernorm_bwd: null pointer"); return MEBT_EINVAL; }
    LnBwdParams p;
    p.x = x; p.dy = dy; p.dy2 = nullptr; p.dx_add = nullptr; p.gamma = gamma; p.mean = mean; p.rstd = rstd; p.dx = dx; p.dx_f32 = dtype == MEBT_F32;
    p.dx_accumulate = 0; p.dgamma = dgamma; p.dbeta = dbeta; p.rows = rows; p.d = d; p.seg = 0; p.seg_stride = 0; p.seg_off = 0;
    return launch_ln_bwd(p, dtype, S(stream));
}

extern "C" int mebt_op_attention_fwd(int32_t dtype, const void* q, const void* k, const void* v, void* o, float* lse, int32_t B,
                                     int32_t H, int32_t NQ, int32_t NK, int32_t HD, int32_t ldq, int32_t ldk, int32_t ldv,
                                     int32_t ldo, int32_t force_generic, mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!q || !o || (NK > 0 && (!k || !v))) { mebt_set_error("attention: null pointer"); return MEBT_EINVAL; }
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.B = B; p.H = H; p.NQ = NQ; p.NK = NK; p.HD = HD;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    mebt_attn_force_generic(force_generic);
    const int rc = launch_attn_fwd(p, dtype, S(stream));
    mebt_attn_force_generic(0);
    return rc;
}

extern "C" int mebt_op_attention_bwd(int32_t dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                                     const void* d_o, void* dq, void* dk, void* dv, float* delta, int32_t B, int32_t H,
                                     int32_t NQ, int32_t NK, int32_t HD, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                     int32_t force_generic, mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!q || !o || !lse || !d_o || !dq || !delta || (NK > 0 && (!k || !v || !dk || !dv))) { mebt_set_error("attention_bwd: null pointer"); return MEBT_EINVAL; }
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.o = const_cast<void*>(o); p.lse = const_cast<float*>(lse); p.B = B; p.H = H; p.NQ = NQ; p.NK = NK; p.HD = HD;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.d_o = d_o; p.dq = dq; p.dk = dk; p.dv = dv; p.delta = delta; p.lddo = ldo; p.lddq = ldq; p.lddk = ldk; p.lddv = ldv;
    mebt_attn_force_generic(force_generic);
    const int rc = launch_attn_bwd(p, dtype, S(stream));
    mebt_attn_force_generic(0);
    return rc;
}

extern "C" int mebt_op_embed_fwd(int32_t dtype, const int64_t* x_ids, const int64_t* ci, const int64_t* ti, const float* tok_emb,
                                 const float* pos_emb, const float* mask_emb, const float* sos_emb, void* sos, void* ctx, void* tgt,
                                 int32_t B, int32_t N, int32_t NC, int32_t NT, int32_t NS, int32_t d, int32_t vocab,
                                 int32_t block_size, mebt_stream_t stream) {
    if (int rc = check_dtype(dtype)) return rc;
    if (!x_ids || !tok_emb || !pos_emb || !mask_emb) { mebt_set_error("embed: null pointer"); return MEBT_EINVAL; }
    if (N > block_size) { mebt_set_error("embed: N exceeds block_size"); return MEBT_ESHAPE; }
    EmbedParams p;
    p.x_ids = x_ids; p.ci = ci; p.ti = ti; p.tok_emb = tok_emb; p.pos_emb = pos_emb; p.mask_emb = mask_emb; p.sos_emb = sos_emb;
    p.sos = sos; p.ctx = ctx; p.tgt = tgt; p.B = B; p.N = N; p.NC = NC; p.NT = NT; p.NS = NS; p.d = d; p.vocab = vocab; p.block_size = block_size;
    p.drop = make_drop(0, 0, 0.f);
    return launch_embed_fwd(p, dtype, S(stream));
}

extern "C" int mebt_op_sample(const float* logits, const float* noise, float temperature, int32_t top_k, float top_p, int64_t* ids,
                              float* score, float* probs, int32_t rows, int32_t V, mebt_stream_t stream) {
    if (!logits || !noise || !ids) { mebt_set_error("sample: null pointer"); return MEBT_EINVAL; }
    SampleParams p;
    p.logits = logits; p.noise = noise; p.temperature = temperature; p.top_k = top_k; p.top_p = top_p; p.ids = ids; p.score = score;
    p.probs = probs; p.rows = rows; p.V = V;
    return launch_sample(p, S(stream));
}

// the same draw with the Exp(1) noise generated inside the kernel (counter-based, keyed by `seed`): the logits are the only
// [rows, V] tensor that moves
extern "C" int mebt_op_sample_seeded(const float* logits, uint64_t seed, float temperature, int32_t top_k, float top_p, int64_t* ids,
                                     float* score, float* probs, int32_t rows, int32_t V, mebt_stream_t stream) {
    if (!logits || !ids) { mebt_set_error("sample: null pointer"); return MEBT_EINVAL; }
    SampleParams p;
    p.logits = logits; p.noise = nullptr; p.noise_seed = seed; p.temperature = temperature; p.top_k = top_k; p.top_p = top_p; p.ids = ids;
    p.score = score; p.probs = probs; p.rows = rows; p.V = V;
    return launch_sample(p, S(stream));
}

// sample(debug=True) (transformer.py:395,426-436): the draw of rows = B * NT target positions, with each row's probabilities written
// straight to its row ti[b, j] of the [B, N, V] probability map (the reference materialises [B, NT, V] and scatter_()s it).
// noise = NULL: Exp(1) drawn in the kernel from `seed`.  V = 16384, no top-p.
extern "C" int mebt_op_sample_scatter(const float* logits, const float* noise, uint64_t seed, float temperature, int32_t top_k, int64_t* ids,
                                      float* score, float* probs_map, const int64_t* ti, int32_t B, int32_t N, int32_t NT, int32_t V,
                                      mebt_stream_t stream) {
    if (!logits || !ids || !probs_map || !ti) { mebt_set_error("sample_scatter: null pointer"); return MEBT_EINVAL; }
    if (B <= 0 || NT <= 0 || N < NT) { mebt_set_error("sample_scatter: need B > 0 and 0 < NT <= N"); return MEBT_ESHAPE; }
    SampleParams p;
    p.logits = logits; p.noise = noise; p.noise_seed = seed; p.temperature = temperature; p.top_k = top_k; p.top_p = 0.f; p.ids = ids;
    p.score = score; p.probs = probs_map; p.rows = B * NT; p.V = V; p.probs_ti = ti; p.probs_N = N; p.probs_NT = NT;
    return launch_sample(p, S(stream));
}

// The draw of the in-engine sampling loops on the logits as the head wrote them: fp32 (logits_bf16 = 0) or bf16 (1: the head's
// bf16 output of a bf16 model, mebt_forward flag 4).  rows = B * NT.  noise = NULL: Exp(1) drawn in the kernel from `seed`.
// probs_map / ti = NULL: no probabilities are written; otherwise each row's probabilities go to row ti[b, j] of the [B, N, V]
// map (sample(debug=True), transformer.py:426-436).  V = 16384, no top-p (the register kernel).
// draw: 0 = arg-max p / q, q ~ Exp(1) per element (the reference's arithmetic, transformer.py:826-841; q = `noise`, or generated per
// element from `seed`); 1 (noise = NULL only) = inverse CDF from one uniform per row: the same categorical distribution at a
// fraction of the arithmetic (the production draw of the sampling loops).
extern "C" int mebt_op_sample_lp(const void* logits, int32_t logits_bf16, const float* noise, uint64_t seed, float temperature, int32_t top_k,
                                 int64_t* ids, float* score, float* probs_map, const int64_t* ti, int32_t B, int32_t N, int32_t NT, int32_t V,
                                 int32_t draw, mebt_stream_t stream) {
    if (!logits || !ids) { mebt_set_error("sample_lp: null pointer"); return MEBT_EINVAL; }
    if ((probs_map != nullptr) != (ti != nullptr)) { mebt_set_error("sample_lp: probs_map and ti go together"); return MEBT_EINVAL; }
    if (B <= 0 || NT <= 0 || (probs_map && N < NT)) { mebt_set_error("sample_lp: need B > 0 and 0 < NT <= N"); return MEBT_ESHAPE; }
    SampleParams p;
    p.logits = reinterpret_cast<const float*>(logits); p.logits_bf16 = logits_bf16 ? 1 : 0; p.noise = noise; p.noise_seed = seed;
    p.temperature = temperature; p.top_k = top_k; p.top_p = 0.f; p.ids = ids; p.score = score; p.probs = probs_map; p.rows = B * NT; p.V = V;
    p.probs_ti = ti; p.probs_N = probs_map ? N : 0; p.probs_NT = probs_map ? NT : 0;
    if (draw != 0 && draw != 1) { mebt_set_error("sample_lp: draw must be 0 (arg-max p / q) or 1 (inverse CDF)"); return MEBT_EINVAL; }
    if (draw == 1 && noise) { mebt_set_error("sample_lp: the inverse-CDF draw takes no noise tensor"); return MEBT_EINVAL; }
    p.icdf = draw;
    return launch_sample(p, S(stream));
}

// the k-th largest logit of every row (exact: temperature 1 leaves the values untouched): the threshold of `top_k_logits`
extern "C" int mebt_op_topk_threshold(const float* logits, int32_t top_k, float* kth, int64_t* ids_scratch, int32_t rows, int32_t V, mebt_stream_t stream) {
    if (!logits || !kth || !ids_scratch) { mebt_set_error("topk_threshold: null pointer"); return MEBT_EINVAL; }
    if (top_k <= 0 || top_k >= V) { mebt_set_error("topk_threshold: k must be in [1, V)"); return MEBT_ESHAPE; }
    SampleParams p;
    p.logits = logits; p.noise = nullptr; p.noise_seed = 0; p.temperature = 1.0f; p.top_k = top_k; p.top_p = 0.f; p.ids = ids_scratch;
    p.score = nullptr; p.probs = nullptr; p.rows = rows; p.V = V; p.kth = kth;
    return launch_sample(p, S(stream));
}

extern "C" int mebt_op_scatter_ids(int64_t* x, const int64_t* ti, const int64_t* ids, int32_t B, int32_t N, int32_t NT, mebt_stream_t stream) {
    if (!x || !ti || !ids) { mebt_set_error("scatter_ids: null pointer"); return MEBT_EINVAL; }
    return launch_scatter_ids(x, ti, ids, B, N, NT, S(stream));
}

extern "C" int mebt_op_next_mask(const int64_t* ci, const int64_t* ti, const float* score, const float* noise, float ctemp,
                                 int32_t n_new, int32_t B, int32_t NC, int32_t NT, int64_t* new_ci, int64_t* new_ti, mebt_stream_t stream) {
    if (!ti || !score || !noise || !new_ci || (NT - n_new > 0 && !new_ti) || (NC > 0 && !ci)) { mebt_set_error("next_mask: null pointer"); return MEBT_EINVAL; }
    NextMaskParams p;
    p.ci = ci; p.ti = ti; p.score = score; p.noise = noise; p.ctemp = ctemp; p.n_new = n_new; p.B = B; p.NC = NC; p.NT = NT;
    p.new_ci = new_ci; p.new_ti = new_ti;
    return launch_next_mask(p, S(stream));
}

extern "C" int mebt_op_cast_bf16(const float* src, void* dst, int64_t n, mebt_stream_t stream) {
    if (!src || !dst) { mebt_set_error("cast: null pointer"); return MEBT_EINVAL; }
    return launch_cast_f32_to_bf16(src, dst, (size_t)n, S(stream));
}
