// Host-side launch interface of the gfx950 kernels (internal; the public C ABI is include/mebt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "common.h"

enum { MEBT_F32 = 0, MEBT_BF16 = 1 };
// dropout site ids (per layer: 16*layer + k)
enum { SITE_ATTN = 0, SITE_PROJ = 1, SITE_MLP = 2, SITE_EMB_SOS = 0xFFFF0, SITE_EMB_CTX = 0xFFFF1, SITE_EMB_TGT = 0xFFFF2 };
static inline DropCfg make_drop(uint64_t seed, uint32_t site, float p) {
    DropCfg d;
    d.seed = seed; d.site = site;
    d.thresh = p > 0.f ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u;
    d.inv_keep = d.thresh ? (float)(65536.0 / (65536.0 - d.thresh)) : 1.0f;
    d.small = 0;
    d.base32 = site * 0x632BE5ABu + (uint32_t)(seed >> 32);
    return d;
}
// launcher side: this site indexes `n_elems` elements
static inline void drop_mark_small(DropCfg& d, uint64_t n_elems) {
    static const bool on = [] { const char* e = getenv("MEBT_DROP_SMALL"); return !(e && e[0] == '0'); }();
    d.small = (on && n_elems < (1ull << 33)) ? 1u : 0u;
}
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_GELU_BWD = 3, EPI_ADAMW = 4 /* internal: grouped weight gradients only */ };

// Caller-owned scratch of the GEMM autotuner (cache-flush buffer) and of the split-K variant (fp32 partial slabs).
struct GemmScratch { void* flush; size_t flush_bytes; float* splitk; size_t splitk_bytes; };
#define MEBT_TUNE_FLUSH_BYTES (384ull << 20)      // > L2 + Infinity Cache: candidates are timed on cold operands
#define MEBT_TUNE_SPLITK_BYTES (112ull << 20)     // 4 slabs of the largest output split-K is offered for (256 tiles of 192 x 128)

// C[M,N] = sum_k A(m,k) B(n,k) (+bias) (+epilogue).  *_kc = 1: operand stored [rows][K];
// *_kc = 0: stored [K][rows].  ld* are element strides of the slow dimension.
struct GemmParams {
    const void* A;
    const void* B;
    void* C;
    void* C2;            // EPI_GELU: gelu(C)
    const float* bias;   // [N] fp32 or null
    const void* aux;     // EPI_RESID: residual [M,ld_aux]; EPI_GELU_BWD: pre-activation [M,ld_aux]
    int M, N, K;
    int lda, ldb, ldc, ld_aux;
    int a_kc, b_kc;
    int epilogue;
    int c_f32;           // bf16 kernel: write fp32 C (weight gradients, logits)
    int beta;            // fp32 C only: C += result
    int split_k;         // 0 = heuristic, 1 = never, >1 = forced (fp32 C, EPI_NONE only)
    long slab = 0;       // internal: split-K without atomics — grid.z slice z stores its partial product at C + z * slab
    DropCfg drop;        // EPI_RESID: C = dropout(acc + bias) + aux   (resid_pdrop, reference gpt.py:140,154)
    // EPI_ADAMW: the result is a weight gradient; it is not stored, the parameter is updated in place
    float* opt_p = nullptr; float* opt_m = nullptr; float* opt_v = nullptr; void* opt_lp = nullptr;   // [M,ldc] like C
    AdamWHyper opt = {0, 0, 0, 0, 0, 1, 1, 1};
    const GemmScratch* scratch = nullptr;   // host side only: where the tuner / split-K may put their scratch (null: heuristic, no split-K)
    // Weight prefetch: while this product runs, its workgroups touch one dword of every 128-byte line of [pf, pf + pf_bytes) —
    // the bf16 weights the NEXT product of the dependency chain will read — so that they sit in the Infinity Cache instead of
    // HBM when that launch starts (every weight is cold when its GEMM starts: 676 MB of weights cycle through 256 MB).
    const void* pf = nullptr;
    unsigned pf_bytes = 0;
    const void* pf2 = nullptr;              // a second range, 4 lines per thread: saved forward activations the NEXT kernels of the
    unsigned pf2_bytes = 0;                 // backward chain read (they were written a whole forward + half a backward ago)
    unsigned* pf_sink = nullptr;            // always null at run time: keeps the prefetch loads alive for the compiler
    int coarse_m = 0;                       // host side only: the tuner buckets M to the next power of two instead of the next multiple of 128 (a row count that
                                            // differs at almost every call: the re-projected rows of the key / value cache)
    int narrow_store = 0;                   // experiments (MEBT_EPI_NARROW=1): 8-byte instead of 16-byte bf16 stores in the plain epilogues
    float* rowsum_a = nullptr;              // internal (grouped weight gradients): [M] fp32 += sum_k A(m, k), added by the tiles of column 0
    unsigned long long* stamps = nullptr;   // diagnostics only (mebt_debug_gemm_stamps): [workgroup][4] s_memtime at entry / first tile landed /
                                            // main loop done / epilogue done, written by wave 0 of each workgroup
};
int launch_gemm(const GemmParams& p, int dtype, hipStream_t stream);
// two independent products with the same operand layouts in ONE launch (query- and key-side projections of a
// block, forward and dgrad): one kernel boundary less and a fuller grid than either product alone
int launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, int dtype, hipStream_t stream);
struct GemmPair { GemmParams p[2]; int tiles0; int ntx[2]; };     // kernel argument of the pair launch
// Up to MEBT_MAX_GROUP independent weight-gradient products dW_g = dY_g^T X_g (both operands row-contiguous, fp32
// result) in ONE launch: at batch 6 a single dW has 64-256 tiles, a layer's worth fills the chip.
#define MEBT_MAX_GROUP 12      // two blocks' worth of weight gradients (2 x (q, k|v, proj, fc1, fc2) + spare)
struct GroupedWgrad {
    int n;
    int tile_start[MEBT_MAX_GROUP + 1];
    // bias: optional [M] fp32, += sum over the K tokens of dY (the Linear layer's bias gradient, nn.Linear backward): dY is this
    // product's A operand, so the workgroups of tile column 0 add it up from the fragments they multiply anyway (one extra MFMA
    // against a ones fragment per A fragment) instead of a separate column-sum pass re-reading every dY of the block (28 MB at C2)
    struct Item { const void* A; const void* B; float* C; int M, N, K, lda, ldb, ldc, ntx; float* bias = nullptr; } g[MEBT_MAX_GROUP];
    // optimizer-in-backward: when `fused`, item i's gradient is applied to W + (C - gW) etc. instead of stored
    int fused = 0;
    float* W = nullptr; float* gW = nullptr; float* mW = nullptr; float* vW = nullptr; void* Wlp = nullptr;
    AdamWHyper opt = {0, 0, 0, 0, 0, 1, 1, 1};
    int beta = 0;                           // C += result (gradient accumulation over micro-batches); not with `fused`
    void* Cb = nullptr;                     // bf16 gradient buffer laid out like gW: item i is stored (rounded once) at Cb + (C - gW) instead of fp32 C
    const GemmScratch* scratch = nullptr;   // host side only, see GemmParams
};
int launch_wgrad_grouped(GroupedWgrad& w, int dtype, hipStream_t stream);   // fills tile_start / ntx
struct GroupedColsum {
    int n;
    int blk_start[MEBT_MAX_GROUP + 1];
    struct Item { const void* X; float* out; int M, N, ldx, gx, rpb; } g[MEBT_MAX_GROUP];
};
int launch_colsum_grouped(GroupedColsum& c, int dtype, hipStream_t stream);
int gemm_init_attributes();
extern "C" int32_t mebt_gemm_autotune_enabled(void);      // is in-situ tuning on (MEBT_GEMM_AUTOTUNE / mebt_gemm_autotune)?  (public: include/mebt_hip.h)
void mebt_gemm_force_split(int s);

// ---- embedding gather (reference transformer.py:255-277) and its scatter-add backward -----------
struct EmbedParams {
    const int64_t* x_ids;   // [B,N]
    const int64_t* ci;      // [B,NC]
    const int64_t* ti;      // [B,NT]
    const float* tok_emb;   // [V,d]
    const float* pos_emb;   // [block,d]
    const float* mask_emb;  // [d]
    const float* sos_emb;   // [NS,d]
    void* sos;              // [B,NS,d] T
    void* ctx;              // [B,NC,d] T
    void* tgt;              // [B,NT,d] T
    int B, N, NC, NT, NS, d, vocab, block_size;
    DropCfg drop;           // embd_pdrop on sos / contexts / targets (reference gpt.py:238-240); site = SITE_EMB_*
};
int launch_embed_fwd(const EmbedParams& p, int dtype, hipStream_t stream);
struct EmbedBwdParams {
    const int64_t* x_ids; const int64_t* ci; const int64_t* ti;
    const float* g_ctx;     // [B,NC,d] fp32 (accumulated over the latent_enc blocks)
    const void* g_tgt;      // [B,NT,d] T
    const void* g_sos;      // [B,NS,d] T
    float* g_tok_emb; float* g_pos_emb; float* g_mask_emb; float* g_sos_emb;   // fp32, accumulated into
    int B, N, NC, NT, NS, d;
};
int launch_embed_bwd(const EmbedBwdParams& p, int dtype, hipStream_t stream);

// ---- LayerNorm (eps 1e-5, affine) ------------------------------------------------------------------
// Output row of input row r:  (r / seg) * seg_stride + seg_off + (r % seg)   (seg = 0: identity).
// This writes LN(sos) and LN(targets) straight into the concatenated key buffer of an `lt2l` block.
struct LnFwdParams {
    const void* x; void* y; const float* gamma; const float* beta;
    float* mean; float* rstd;     // indexed by OUTPUT row; may be null (inference)
    int rows, d; int seg, seg_stride, seg_off;
};
int launch_ln_fwd(const LnFwdParams& p, int dtype, hipStream_t stream);
// several LayerNorms in ONE launch (the query and key sides of a block share LN1: gpt.py:180-181)
constexpr int MEBT_LN_MAXJ = 3;
int launch_ln_fwd_multi(const LnFwdParams* jobs, int n, int dtype, hipStream_t stream);
struct LnBwdParams {
    const void* x;                // LN input rows [rows,d] T
    const void* dy;               // grad wrt LN output, rows mapped like LnFwdParams (seg..)
    const void* dy2;              // optional second grad (contiguous [rows,d]) added to dy BEFORE the LN backward
    const void* dx_add;           // optional [rows,d] T added to the result AFTER it (residual branch)
    const float* gamma; const float* mean; const float* rstd;   // stats indexed by mapped row
    void* dx; int dx_f32; int dx_accumulate;                     // dx (+)= ...
    float* dgamma; float* dbeta;                                 // fp32, atomically accumulated
    int rows, d; int seg, seg_stride, seg_off;
    void* dx2 = nullptr;          // optional second output: dx * dropout mask `drop2` (element index row*d + e), type of dx
    DropCfg drop2 = {0, 0, 0, 1.0f};
};
// dx rows and the dgamma/dbeta column reduction of `n` jobs in ONE launch on `stream` (param_stream is ignored: kept for callers)
int launch_ln_bwd(const LnBwdParams& p, int dtype, hipStream_t stream, hipStream_t param_stream = nullptr);
struct GroupedColsum;
// `colsums` (optional): grouped column sums (bias gradients) executed by extra workgroups of the same launch
int launch_ln_bwd_multi(const LnBwdParams* jobs, int n, int dtype, hipStream_t stream, hipStream_t param_stream = nullptr, const GroupedColsum* colsums = nullptr);

// dst[i] = src[i] * keep(site, i)  (dropout mask re-applied in backward); TS/TD chosen by flags
int launch_apply_dropout(const void* src, void* dst, size_t n, int src_f32, int dst_f32, const DropCfg& d, hipStream_t stream);

// dst[map_d(r)] = src[map_s(r)] for r < rows, rows of d elements, with element conversion; map(r) = (r / seg) * stride + off
// + r % seg (seg = 0: identity).  Splits / concatenates the [contexts; targets] stream of a 'maskgit' block.
int launch_copy_rows(const void* src, void* dst, long rows, int d, int src_f32, int dst_f32, int s_seg, int s_stride, int s_off,
                     int d_seg, int d_stride, int d_off, hipStream_t stream);

// rows moved by a per-sample position list idx [B, n] (gather: dst[b * n + j] = src[b * npos + idx[b, j]]; scatter: the inverse)
int launch_cast_i64_i32(const int64_t* src, int32_t* dst, size_t n, hipStream_t stream);
int launch_index_rows(const void* src, void* dst, const int64_t* idx, int B, int n, int npos, int row_bytes, int scatter, hipStream_t stream);

// out[n] += sum_m X[m,n]   (bias gradients, mask_emb / sos_emb gradients)
int launch_colsum(const void* X, int M, int N, int ldx, float* out, int dtype, hipStream_t stream);

// ---- masked-token loss (reference transformer.py:717-732, utils.py:80-94) -------------------------
struct CeParams {
    const float* logits;      // [rows,V] fp32
    const int64_t* x_ids;     // [B,N]: target id of row (b,j) = x_ids[b, ti[b,j]]
    const int64_t* ti;        // [B,NT]
    int rows, V, B, N, NT;
    float label_smoothing;
    float* row_lse;           // [rows]
    float* row_loss;          // [rows]
    int* row_rank;            // [rows] number of classes ranked above the target
    double* out;              // [4]: loss_sum, n_top1, n_top5, rows
    // optional: also d(loss_sum * grad_scale)/dlogits [rows,V] (bf16 if dl_bf16 else fp32) from the same pass over the row — what
    // launch_ce_bwd computes (same expressions; agrees to fp32 rounding), without reading the 200 MB of logits a second time.  Only for V = 16384
    // (launch_ce_fwd returns with `dlogits` untouched otherwise: check ce_fwd_can_fuse_grad first).
    void* dlogits = nullptr;
    float grad_scale = 0.f;
    int dl_bf16 = 0;
};
int launch_ce_fwd(const CeParams& p, hipStream_t stream);
inline bool ce_fwd_can_fuse_grad(int V) { return V == 16384; }
struct CeBwdParams {
    const float* logits; const int64_t* x_ids; const int64_t* ti; const float* row_lse;
    void* dlogits;            // [rows,V] T
    const float* upstream;    // device scalar (dL/dloss) or null (=1)
    float scale;              // 1 / (B * seq_len * weight)
    float label_smoothing;
    int rows, V, B, N, NT;
};
int launch_ce_bwd(const CeBwdParams& p, int dtype, hipStream_t stream);

// ---- fused AdamW over a flat buffer (reference transformer.py:790-797) -----------------------------
struct AdamWParams {
    float* p; const float* g; float* m; float* v; void* p_bf16;   // p_bf16 may be null
    size_t n; float lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale;
    int g_bf16 = 0;    // g points at bf16 gradients (the reduce-scattered wire format of the data-parallel path)
    int g_pieces = 1;  // bf16 only: the gradient is the fp32 sum of g_pieces bf16 copies, copy j at g + j * g_stride elements (all-to-all exchange)
    size_t g_stride = 0;
};
int launch_adamw(const AdamWParams& p, hipStream_t stream);
int launch_cast_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t stream);

// ---- attention ---------------------------------------------------------------------------------------
// q [B,NQ,h,hd] (row stride ldq), k,v [B,NK,h,hd] (ldk, ldv), o [B,NQ,h,hd] (ldo); lse [B,h,NQ] fp32.
// softmax(q k^T / sqrt(hd)) v, no mask (attn_bias == 0 at every reference call site).
struct AttnParams {
    const void* q; const void* k; const void* v; void* o; float* lse;
    int B, H, NQ, NK, HD; int ldq, ldk, ldv, ldo;
    // backward
    const void* d_o; void* dq; void* dk; void* dv; float* delta; int lddo, lddq, lddk, lddv;
    DropCfg drop;        // attn_pdrop on the probabilities (reference gpt.py:135); element = ((b*H+h)*NQ+q)*NK+key
    // MFMA kernels, training with attention dropout: the forward writes the keep bits it evaluated, the two backward kernels read
    // them instead of re-hashing every element (the mask was hashed three times per step: forward, dQ, dK/dV).  Layout:
    // 16-bit fields [(b*H+h)][q][key tile of 64, padded to a multiple of 4 tiles][g = 0..3]; field bit 4*kb + r <-> key
    // 64*tile + 16*kb + 4*g + r (the 16 score elements one lane of the forward / dQ kernels holds per tile).
    // mebt_attn_dmask_bytes() bytes, or null: hash everywhere.
    uint16_t* dmask = nullptr;
    // MFMA forward only (the sampling loops' key / value cache, mebt_forward_kvcache): key / value row r of sample b is row
    // kidx[b * NK + r] of a buffer that holds `kidx_rows` rows per sample (k, v point at sample 0, position 0).  NK <= 8192.
    const int32_t* kidx = nullptr;
    int kidx_rows = 0;
};
__host__ __device__ static inline int mebt_attn_dmask_tiles(int NK) { return 4 * ((NK + 255) / 256); }
static inline size_t mebt_attn_dmask_bytes(int B, int H, int NQ, int NK) { return (size_t)B * H * NQ * mebt_attn_dmask_tiles(NK) * 8; }
int launch_attn_fwd(const AttnParams& p, int dtype, hipStream_t stream);
int launch_attn_bwd(const AttnParams& p, int dtype, hipStream_t stream);
void mebt_attn_force_generic(int on);
bool attn_fwd_can_gather(int dtype, int HD);     // the forward of this (dtype, head size) honours AttnParams::kidx

// ---- sampler (reference transformer.py:826-910, :413-439, mask_sampler.py:178-246) -----------------
struct SampleParams {
    const float* logits;   // [rows,V]
    int logits_bf16 = 0;   // `logits` points at bf16 values (the head's bf16 output of the in-engine sampling loops): register kernel only
    int icdf = 0;          // noise == null only, register kernel only: draw by inverse CDF from ONE uniform per row (seed, row) instead of
                           // arg-max p / q over per-element Exp(1) noise — the same categorical distribution (sampler.hip)
    const float* noise;    // [rows,V] Exp(1), or null: drawn in the kernel from the counter-based generator keyed by noise_seed
    uint64_t noise_seed = 0;
    float temperature; int top_k; float top_p;   // top_k <= 0 / top_p <= 0: disabled
    int64_t* ids;          // [rows]
    float* score;          // [rows] p[ids] after temperature/top-k/top-p
    float* probs;          // optional [rows,V]
    int rows, V;
    float* kth = nullptr;  // optional [rows]: the k-th largest value of logits / (temperature + 1e-8) of each row (top_k > 0), i.e. the
                           // threshold below which `top_k_logits` writes -inf (reference transformer.py:891-895)
    // optional: rows = B * probs_NT and `probs` is the [B, probs_N, V] probability map of sample(debug=True) (transformer.py:395,
    // 426-436): the probabilities of row (b, j) go to map row b * probs_N + probs_ti[b * probs_NT + j] (the reference's scatter_)
    const int64_t* probs_ti = nullptr;
    int probs_N = 0, probs_NT = 0;
};
int launch_sample(const SampleParams& p, hipStream_t stream);
int launch_scatter_ids(int64_t* x, const int64_t* ti, const int64_t* ids, int B, int N, int NT, hipStream_t stream);
struct NextMaskParams {
    const int64_t* ci; const int64_t* ti; const float* score; const float* noise;   // [B,NC],[B,NT],[B,NT],[B,NT]
    float ctemp; int n_new; int B, NC, NT;
    int64_t* new_ci;   // [B,NC+n_new]
    int64_t* new_ti;   // [B,NT-n_new]
};
int launch_next_mask(const NextMaskParams& p, hipStream_t stream);
