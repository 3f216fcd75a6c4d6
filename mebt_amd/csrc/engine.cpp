// Model handle + forward/backward orchestration behind the C ABI (include/mebt_hip.h).
// No kernels here: this file sequences the launches of gemm.hip / attention*.hip / elementwise.hip
// on the caller's stream, carving every activation out of the caller-provided workspace.
//
// Reference being replaced: embed (mebt/transformer.py:255-277), GPT.forward / Block.forward
// (mebt/modules/gpt.py:159-195,234-253), shared_step loss (transformer.py:717-732), the autograd
// backward of all of it, and AdamW (transformer.py:790-797).
#include "common.h"
#include "kernels.h"
#include "../../include/mebt_hip.h"
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <cstdlib>

namespace {

struct LayerOffsets {
    // W flat (elements)
    int64_t wq, wk, wv, wp, w1, w2;
    // P flat
    int64_t ln1w, ln1b, ln2w, ln2b, bq, bk, bv, bp, b1, b2;
};

struct Carve {   // bump allocator over the workspace; query mode when base == nullptr
    char* base;
    int64_t off;
    void* take(int64_t bytes) {
        const int64_t o = off;
        off += (bytes + 255) & ~(int64_t)255;
        return base ? base + o : nullptr;
    }
};

struct LayerAct {   // saved activations of one block
    void *qn, *kn, *q, *k, *v, *att, *x, *hn, *pre, *u, *out;
    float *mean1q, *rstd1q, *mean1k, *rstd1k, *mean2, *rstd2, *lse;
    uint16_t* dmask = nullptr;   // attention-dropout keep bits written by the MFMA forward, read by its two backward kernels (AttnParams::dmask)
    const void* q_in;   // LN1 input of the query side (previous stream value); maskgit: the contexts stream
    const void* k_in;   // LN1 input of the key side (enc: contexts, dec: sos); lt2l uses S and T; maskgit: the targets stream
    void *c_out, *t_out;   // maskgit: the block output split back into contiguous contexts / targets (gpt.py:191-192)
    int NQ, NK, ldqkv_q, ldqkv_k;
};

struct FwdCtx {
    bool valid = false;
    void* ws = nullptr;
    int B = 0, N = 0, NC = 0, NT = 0;
    const int64_t *x_ids = nullptr, *ci = nullptr, *ti = nullptr;
    int32_t* ci32 = nullptr;         // inference: int32 copy of `ci` (the gathering attention of the key / value cache)
    void *sos0 = nullptr, *ctx = nullptr, *tgt0 = nullptr;
    std::vector<LayerAct> L;
    const void *S_final = nullptr, *T_final = nullptr;
    void* hf = nullptr; float *meanf = nullptr, *rstdf = nullptr;
    float* logits_ws = nullptr;
    float *row_lse = nullptr, *row_loss = nullptr; int* row_rank = nullptr; double* loss_out = nullptr;
    bool loss_done = false;
    bool dlogits_ready = false;      // mebt_loss_with_grad has already written x.dlogits for `dlogits_scale`
    float dlogits_scale = 0.f;
    // backward scratch
    void *g_S = nullptr, *g_T = nullptr; float* g_C = nullptr;
    void* g_cat = nullptr;     // maskgit: cat[g_C, g_T], the gradient of a block output that spans both streams
    void *dlogits = nullptr, *dhf = nullptr, *datt = nullptr;
    // per-layer backward scratch, two sets (layer parity): the side stream may still read layer i's
    // operands while the main stream already produces layer i-1's
    struct Scratch { void *d4, *dh, *dx, *dqkv_q, *dqkv_k, *dqn, *dkn, *dout_m, *dx_m; } sc[2];
    float* delta = nullptr;
    bool gS_defined = false, gT_defined = false, gC_defined = false;
    int doutm_ready = -1;      // block whose dropout-masked output gradient was already written by the LN1 backward above it
    int last_bwd_lo = -1;      // lowest block the previous mebt_backward_layers call finished (doutm_ready is only valid for a contiguous descent)
    bool drop_on = false; uint64_t drop_seed = 0;
    GemmScratch tune = {nullptr, 0, nullptr, 0};     // GEMM tuner / split-K scratch, carved from the caller's workspace (bf16 mode)
};

}  // namespace

struct mebt_model {
    mebt_model_desc d;
    std::vector<LayerOffsets> lo;
    int64_t n_w = 0, n_p = 0;
    int64_t head_w = 0, lnf_w = 0, lnf_b = 0, mask_emb = 0, sos_emb = 0, pos_emb = 0, tok_emb = 0;
    float *W = nullptr, *gW = nullptr, *P = nullptr, *gP = nullptr;
    void* Wlp = nullptr;
    bool attn_bits = true;     // fixed at creation (workspace layout): MEBT_ATTN_DROP_BITS=0 -> the backward kernels re-hash the mask
    bool tune_flush = true;    // fixed at creation (the workspace layout depends on it): was in-situ GEMM tuning on?
    void* gWb = nullptr;       // mebt_model_bind_wire_grads: bf16 gradient buffer laid out like gW; when set, the Linear weight
                               // gradients are stored there (and ONLY there) straight from the MFMA accumulators

    std::vector<char> live;   // per layer: does the loss depend on this block?
    bool tok_live = false;
    bool has_maskgit = false;
    FwdCtx ctx;
    // side stream for work that nothing on the critical path waits for in backward (weight / bias / LN-affine
    // gradients).  It bought +14 % while the per-layer kernels were too small to fill 256 CUs; with the tuned,
    // paired and grouped launches of today it is neutral to slightly slower (measured), so the Python side
    // switches it off unless MEBT_SIDE_STREAM=1.  Fork/join with events only.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_e1 = nullptr, ev_layer[2] = {nullptr, nullptr}, ev_join = nullptr;
    bool use_side = true;
    hipStream_t side_own = nullptr;      // the stream this object created (destroyed with it); `side` may point at a caller's stream instead
    // optimizer-in-backward (mebt_model_set_fused_adamw): when armed, the weight gradients of the blocks are applied
    // to W (AdamW) inside the weight-gradient launch instead of being stored in gW
    bool fused_on = false;
    bool grad_acc = false;     // mebt_model_set_grad_accumulate: backward adds to gW / gP instead of overwriting them
    // Weight gradients of TWO consecutive blocks in one grouped launch (bf16, no side stream; MEBT_WGRAD_PAIR=0: off): the first block of a
    // pair parks its items here; its operands live in scratch set (i & 1) and in saved activations, which nothing touches before the
    // next block's flush point (backward_layer).  A launch of 1536 tiles of 128 x 128 fills the chip in whole rounds, 768 do not.
    GroupedWgrad pend_w;
    bool pend = false;
    float *fused_mW = nullptr, *fused_vW = nullptr;
    AdamWHyper fused_h = {0, 0, 0, 0, 0, 1, 1, 1};
    bool wire() const { return gWb && d.dtype == MEBT_BF16 && !grad_acc && !fused_on; }
    // mebt_model_set_forward_waits: one-shot (first layer that reads the parameters, event) pairs the next forward honours
    std::vector<std::pair<int, hipEvent_t>> fw_waits;
    int esz() const { return d.dtype == MEBT_BF16 ? 2 : 4; }
    // weight operand for GEMMs (bf16 mirror in bf16 mode)
    const void* Wop(int64_t off) const {
        return d.dtype == MEBT_BF16 ? (const void*)((const char*)Wlp + off * 2) : (const void*)(W + off);
    }
};

static hipStream_t S(mebt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static int fork_side(mebt_model* m, hipStream_t st);
static int join_side(mebt_model* m, hipStream_t st);

// ---------------------------------------------------------------------------------------------------
// profiling of the GEMM family with HIP events on the launch stream
// ---------------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
    hipEvent_t a, b; double flops, bytes; int kind = 0;      // kind 0: GEMM family, 2: embed forward, 3: embed backward, 4: forward wait
    int tag = 0;              // kind 0: 'g' single product, 'p' pair, 'w' grouped weight gradients
    int dims[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // g: M N K a_kc b_kc epilogue c_f32; p: M0 N0 K0 M1 N1 K1 b_kc; w: n, sum M*N, K of the longest reduction
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_ev_pool;
hipEvent_t get_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

static int gemm(const mebt_model* m, GemmParams p, hipStream_t st) {
    if (p.K <= 0) {   // empty reduction (e.g. NC = 0): the product is zero
        if (p.c_f32 && !p.beta && p.M > 0 && p.N > 0 && p.epilogue == EPI_NONE && !p.bias)
            MEBT_HIP_CHECK(hipMemset2DAsync(p.C, (size_t)p.ldc * 4, 0, (size_t)p.N * 4, p.M, st));
        else if (p.M > 0 && p.N > 0) { mebt_set_error("gemm: K = 0 with a fused epilogue is not supported"); return MEBT_ESHAPE; }
        return MEBT_OK;
    }
    p.scratch = m->ctx.tune.flush ? &m->ctx.tune : nullptr;
    ProfRec r;
    const bool prof = g_prof_on && p.M > 0 && p.N > 0;
    if (prof) {
        r.a = get_event(); r.b = get_event(); r.flops = 2.0 * p.M * p.N * p.K;
        r.tag = 'g'; r.dims[0] = p.M; r.dims[1] = p.N; r.dims[2] = p.K; r.dims[3] = p.a_kc; r.dims[4] = p.b_kc; r.dims[5] = p.epilogue; r.dims[6] = p.c_f32;
        const double esz = m->d.dtype == MEBT_BF16 ? 2.0 : 4.0, csz = (p.c_f32 || m->d.dtype == MEBT_F32) ? 4.0 : 2.0;
        r.bytes = ((double)p.M * p.K + (double)p.N * p.K) * esz + (double)p.M * p.N * (csz * (p.C ? 1 : 0) + (p.C2 ? esz : 0) + (p.aux ? esz : 0));
        (void)hipEventRecord(r.a, st);
    }
    const int rc = launch_gemm(p, m->d.dtype, st);
    if (prof) { (void)hipEventRecord(r.b, st); g_prof.push_back(r); }
    return rc;
}

static int gemm_pair(const mebt_model* m, const GemmParams& p0_in, const GemmParams& p1_in, hipStream_t st) {
    GemmParams p0 = p0_in, p1 = p1_in;
    if (m->d.dtype != MEBT_BF16 || p0.K <= 0 || p1.K <= 0 || p0.M <= 0 || p1.M <= 0) {
        if (int rc = gemm(m, p0, st)) return rc;
        return gemm(m, p1, st);
    }
    p0.scratch = p1.scratch = m->ctx.tune.flush ? &m->ctx.tune : nullptr;
    ProfRec r;
    if (g_prof_on) {
        r.a = get_event(); r.b = get_event(); r.flops = 2.0 * p0.M * p0.N * p0.K + 2.0 * p1.M * p1.N * p1.K;
        r.tag = 'p'; r.dims[0] = p0.M; r.dims[1] = p0.N; r.dims[2] = p0.K; r.dims[3] = p1.M; r.dims[4] = p1.N; r.dims[5] = p1.K; r.dims[6] = p0.b_kc;
        r.bytes = 0;
        for (const GemmParams* q : {&p0, &p1})
            r.bytes += ((double)q->M * q->K + (double)q->N * q->K) * 2.0 + (double)q->M * q->N * (2.0 * (q->C ? 1 : 0) + (q->aux ? 2.0 : 0));
        (void)hipEventRecord(r.a, st);
    }
    const int rc = launch_gemm_pair(p0, p1, m->d.dtype, st);
    if (g_prof_on) { (void)hipEventRecord(r.b, st); g_prof.push_back(r); }
    return rc;
}

extern "C" int mebt_profile_enable(int32_t on) {
    g_prof_on = on != 0;
    for (auto& r : g_prof) { g_ev_pool.push_back(r.a); g_ev_pool.push_back(r.b); }
    g_prof.clear();
    return MEBT_OK;
}
extern "C" int mebt_profile_read(int32_t family, double* launches, double* total_ms, double* total_flops) {
    double ms = 0, fl = 0, n = 0;
    const int kind = family >= 2 ? family : 0;
    for (auto& r : g_prof) {
        if (r.kind != kind) continue;
        MEBT_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0;
        MEBT_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += family >= 1 ? r.bytes : r.flops;    // family 1: algorithmic operand+result bytes instead of FLOPs
        n += 1;
    }
    *launches = n; *total_ms = ms; *total_flops = fl;
    return MEBT_OK;
}

// every GEMM-family launch recorded while profiling was on, one text line each: `tag dims... ms gflop` (tag g: M N K a_kc b_kc epilogue
// c_f32; p: M0 N0 K0 M1 N1 K1 b_kc; w: items, output Ki-elements, longest K).  Same size protocol as mebt_gemm_tune_export
// (returns the bytes needed incl. the terminating 0, writes when `cap` suffices).  tools/step_gemm_table.py
extern "C" int64_t mebt_profile_dump(char* buf, int64_t cap) {
    std::string text;
    char line[256];
    for (auto& r : g_prof) {
        if (r.kind != 0 || !r.tag) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return -1;
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
        snprintf(line, sizeof line, "%c %d %d %d %d %d %d %d %.5f %.4f\n", (char)r.tag, r.dims[0], r.dims[1], r.dims[2], r.dims[3], r.dims[4], r.dims[5],
                 r.dims[6], t, r.flops * 1e-9);
        text += line;
    }
    const int64_t need = (int64_t)text.size() + 1;
    if (buf && cap >= need) memcpy(buf, text.c_str(), (size_t)need);
    return need;
}

// the forward's waits for deferred parameter gathers recorded while profiling was on: (first layer that reads the bucket, ms the
// compute stream waited), in launch order; returns the number of records (at most `cap` are written), negative on a HIP error
extern "C" int32_t mebt_profile_read_waits(int32_t cap, int32_t* layer, double* ms) {
    int32_t n = 0;
    for (auto& r : g_prof) {
        if (r.kind != 4) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return -1;
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
        if (n < cap && layer && ms) { layer[n] = (int32_t)r.flops; ms[n] = t; }
        ++n;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------
extern "C" int mebt_abi_version(void) { return MEBT_ABI_VERSION; }

extern "C" int mebt_model_create(const mebt_model_desc* desc, mebt_model** out) {
    if (!desc || !out) { mebt_set_error("model_create: null argument"); return MEBT_EINVAL; }
    const mebt_model_desc& d = *desc;
    if (d.n_layer <= 0 || d.n_layer > MEBT_MAX_LAYERS) { mebt_set_error("model_create: n_layer out of range"); return MEBT_ESHAPE; }
    if (d.n_head <= 0 || d.n_embd % d.n_head) { mebt_set_error("model_create: n_embd must be divisible by n_head"); return MEBT_ESHAPE; }   // gpt.py:107
    if (d.n_embd % 64) { mebt_set_error("model_create: n_embd must be a multiple of 64 for the gfx950 kernels"); return MEBT_ESHAPE; }
    const int hd = d.n_embd / d.n_head;
    if (hd != 32 && hd != 64 && hd != 128) { mebt_set_error("model_create: head size must be 32, 64 or 128"); return MEBT_ESHAPE; }
    if (d.vocab % 8 || d.vocab <= 0) { mebt_set_error("model_create: vocab must be a positive multiple of 8"); return MEBT_ESHAPE; }
    if (d.n_latent < 0) { mebt_set_error("model_create: sos_emb (latent tokens) must be >= 0"); return MEBT_ESHAPE; }
    if (d.n_latent == 0)          // sos_emb = 0 (transformer.py:273-276): only meaningful when no block routes through the latents
        for (int i = 0; i < d.n_layer; ++i)
            if (d.modes[i] != MEBT_MODE_MASKGIT) { mebt_set_error("model_create: sos_emb = 0 needs every block in 'maskgit' mode (the latent routings read / write the latent tokens)"); return MEBT_ESHAPE; }
    if (d.dtype != MEBT_F32 && d.dtype != MEBT_BF16) { mebt_set_error("model_create: dtype must be f32 or bf16"); return MEBT_EDTYPE; }
    for (int i = 0; i < d.n_layer; ++i)
        if (d.modes[i] < 0 || d.modes[i] > MEBT_MODE_MASKGIT) {
            mebt_set_error("model_create: unknown block mode (latent_enc / latent_self / latent_dec / lt2l / maskgit)");
            return MEBT_EINVAL;
        }
    if (d.embd_pdrop < 0.f || d.embd_pdrop >= 1.f || d.resid_pdrop < 0.f || d.resid_pdrop >= 1.f || d.attn_pdrop < 0.f || d.attn_pdrop >= 1.f) {
        mebt_set_error("model_create: dropout probabilities must be in [0, 1)");
        return MEBT_EINVAL;
    }
    mebt_model* m = new mebt_model();
    m->tune_flush = mebt_gemm_autotune_enabled() != 0;
    { const char* e = getenv("MEBT_ATTN_DROP_BITS"); m->attn_bits = !(e && e[0] == '0'); }
    m->d = d;
    const int64_t dd = (int64_t)d.n_embd * d.n_embd, e = d.n_embd;
    int64_t w = 0, p = 0;
    m->lo.resize(d.n_layer);
    for (int i = 0; i < d.n_layer; ++i) {
        LayerOffsets& o = m->lo[i];
        o.wq = w; w += dd; o.wk = w; w += dd; o.wv = w; w += dd; o.wp = w; w += dd;
        o.w1 = w; w += 4 * dd; o.w2 = w; w += 4 * dd;
        o.ln1w = p; p += e; o.ln1b = p; p += e; o.ln2w = p; p += e; o.ln2b = p; p += e;
        o.bq = p; p += e; o.bk = p; p += e; o.bv = p; p += e; o.bp = p; p += e;
        o.b1 = p; p += 4 * e; o.b2 = p; p += e;
    }
    m->head_w = w; w += (int64_t)d.vocab * e;
    m->lnf_w = p; p += e; m->lnf_b = p; p += e;
    m->mask_emb = p; p += e;
    m->sos_emb = p; p += (int64_t)d.n_latent * e;
    m->pos_emb = p; p += (int64_t)d.block_size * e;
    m->tok_emb = p; p += (int64_t)d.vocab * e;
    m->n_w = w; m->n_p = p;
    // liveness: walk backwards from the head, which reads the targets stream only (gpt.py:247)
    m->live.assign(d.n_layer, 0);
    bool gS = false, gT = true;
    for (int i = d.n_layer - 1; i >= 0; --i) {
        switch (d.modes[i]) {
            case MEBT_MODE_LATENT_ENC: if (gS) { m->live[i] = 1; m->tok_live = true; } break;
            case MEBT_MODE_LATENT_SELF: if (gS) m->live[i] = 1; break;
            case MEBT_MODE_LT2L: if (gS) { m->live[i] = 1; gT = true; } break;
            case MEBT_MODE_LATENT_DEC: if (gT) { m->live[i] = 1; gS = true; } break;
            case MEBT_MODE_MASKGIT: m->live[i] = 1; m->tok_live = true; m->has_maskgit = true; break;   // gT is always defined here
        }
    }
    // MEBT_HOST_ONLY=1 (sanitizer / layout tests on machines without a GPU): the handle answers the host-side queries
    // (offset tables, workspace sizes, argument validation); anything that launches returns the HIP error of its first call
    static const bool host_only = [] { const char* e = getenv("MEBT_HOST_ONLY"); return e && e[0] == '1'; }();
    if (host_only) { m->use_side = false; *out = m; return MEBT_OK; }
    int rc = gemm_init_attributes();
    if (rc) { delete m; return rc; }
    {   // lowest priority: the leaves must never delay the critical path on the caller's stream
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, least) != hipSuccess) { m->side = nullptr; m->use_side = false; (void)hipGetLastError(); }
    }
    m->side_own = m->side;
    if (m->side) {
        hipEvent_t* evs[] = {&m->ev_fork, &m->ev_e1, &m->ev_layer[0], &m->ev_layer[1], &m->ev_join};
        for (hipEvent_t* e : evs) MEBT_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    *out = m;
    return MEBT_OK;
}

extern "C" void mebt_model_destroy(mebt_model* m) {
    if (!m) return;
    if (m->side) {
        (void)hipStreamSynchronize(m->side);
        hipEvent_t evs[] = {m->ev_fork, m->ev_e1, m->ev_layer[0], m->ev_layer[1], m->ev_join};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        if (m->side_own) (void)hipStreamDestroy(m->side_own);
    }
    delete m;
}
extern "C" void mebt_debug_side_stream(mebt_model* m, int32_t on) { if (m) m->use_side = on != 0 && m->side != nullptr; }
// use the caller's stream (e.g. one it probed to run beside its compute stream: streams that share a hardware queue do not overlap)
extern "C" void mebt_debug_set_side_stream(mebt_model* m, mebt_stream_t s) {
    if (!m || !m->side_own) return;
    (void)hipStreamSynchronize(m->side);
    m->side = s ? reinterpret_cast<hipStream_t>(s) : m->side_own;
}

extern "C" int mebt_model_param_counts(const mebt_model* m, int64_t* n_w, int64_t* n_p) {
    if (!m) { mebt_set_error("null model"); return MEBT_EINVAL; }
    *n_w = m->n_w; *n_p = m->n_p;
    return MEBT_OK;
}

extern "C" int mebt_model_bind(mebt_model* m, float* W, void* W_lp, float* gW, float* P, float* gP) {
    if (!m || !W || !P) { mebt_set_error("model_bind: W and P are required"); return MEBT_EINVAL; }
    if (m->d.dtype == MEBT_BF16 && !W_lp) { mebt_set_error("model_bind: bf16 mode needs the bf16 weight mirror"); return MEBT_EINVAL; }
    if (m->W != W || m->Wlp != W_lp || m->P != P) m->ctx.valid = false;   // attaching gradient buffers keeps a live forward
    m->W = W; m->Wlp = W_lp; m->gW = gW; m->P = P; m->gP = gP;
    return MEBT_OK;
}

extern "C" int mebt_model_sync_lowp(mebt_model* m, mebt_stream_t stream) {
    if (!m || !m->W) { mebt_set_error("model_sync_lowp: model not bound"); return MEBT_EINVAL; }
    if (m->d.dtype != MEBT_BF16) return MEBT_OK;
    return launch_cast_f32_to_bf16(m->W, m->Wlp, (size_t)m->n_w, S(stream));
}

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
static void mode_shape(const mebt_model* m, int mode, int NC, int NT, int& NQ, int& NK) {
    const int NS = m->d.n_latent;
    switch (mode) {
        case MEBT_MODE_LATENT_ENC: NQ = NS; NK = NC; break;
        case MEBT_MODE_LATENT_SELF: NQ = NS; NK = NS; break;
        case MEBT_MODE_LATENT_DEC: NQ = NT; NK = NS; break;
        case MEBT_MODE_MASKGIT: NQ = NC + NT; NK = NC + NT; break;
        default: NQ = NS; NK = NS + NT; break;
    }
}

// Lays the workspace out; with c.base == nullptr only measures.  In inference (training == 0)
// all per-layer buffers alias one layer-sized region and the stream outputs ping-pong.
static void carve(const mebt_model* m, Carve& c, FwdCtx& x, int B, int NC, int NT, int training) {
    const int64_t d = m->d.n_embd, e = m->esz(), NS = m->d.n_latent, H = m->d.n_head, V = m->d.vocab;
    if (m->d.dtype == MEBT_BF16) {     // the library allocates nothing: the tuner's flush buffer and the split-K slabs come from here
        // the 384 MB cache-flush buffer only while in-situ tuning is on (inference-only models and every extra model of a
        // process paid for it); with tuning off a launch never times candidates.  16 bytes keep `flush` non-null = "scratch present".
        const int64_t fb = m->tune_flush ? (int64_t)MEBT_TUNE_FLUSH_BYTES : 16;
        x.tune.flush = c.take(fb); x.tune.flush_bytes = (size_t)fb;
        x.tune.splitk = (float*)c.take((int64_t)MEBT_TUNE_SPLITK_BYTES); x.tune.splitk_bytes = MEBT_TUNE_SPLITK_BYTES;
    } else {
        x.tune = {nullptr, 0, nullptr, 0};
    }
    x.sos0 = c.take(B * NS * d * e);
    x.ctx = c.take((int64_t)B * NC * d * e);
    x.tgt0 = c.take((int64_t)B * NT * d * e);
    x.ci32 = training ? nullptr : (int32_t*)c.take((int64_t)B * NC * 4);
    x.L.assign(m->d.n_layer, LayerAct());
    const int64_t layer_base = c.off;
    int64_t layer_max = c.off;
    void* pingS[2] = {nullptr, nullptr};
    void* pingT[2] = {nullptr, nullptr};
    if (!training) {
        pingS[0] = c.take(B * NS * d * e); pingS[1] = c.take(B * NS * d * e);
        pingT[0] = c.take((int64_t)B * NT * d * e); pingT[1] = c.take((int64_t)B * NT * d * e);
    }
    for (int i = 0; i < m->d.n_layer; ++i)      // a maskgit block rewrites both streams: its split outputs outlive the per-layer region
        if (m->d.modes[i] == MEBT_MODE_MASKGIT) {
            x.L[i].c_out = c.take((int64_t)B * NC * d * e);
            x.L[i].t_out = c.take((int64_t)B * NT * d * e);
        }
    const int64_t after_ping = c.off;
    int nS = 0, nT = 0;
    for (int i = 0; i < m->d.n_layer; ++i) {
        LayerAct& a = x.L[i];
        const int mode = m->d.modes[i];
        mode_shape(m, mode, NC, NT, a.NQ, a.NK);
        const int64_t Mq = (int64_t)B * a.NQ, Mk = (int64_t)B * a.NK;
        if (!training) c.off = after_ping;
        a.qn = c.take(Mq * d * e);
        a.mean1q = (float*)c.take(Mq * 4); a.rstd1q = (float*)c.take(Mq * 4);
        if (mode == MEBT_MODE_LATENT_SELF || mode == MEBT_MODE_MASKGIT) {
            a.kn = a.qn; a.mean1k = a.mean1q; a.rstd1k = a.rstd1q;
            a.q = c.take(Mq * 3 * d * e);
            a.k = (char*)a.q + d * e; a.v = (char*)a.q + 2 * d * e;
            a.ldqkv_q = 3 * d; a.ldqkv_k = 3 * d;
        } else {
            a.kn = c.take(Mk * d * e);
            a.mean1k = (float*)c.take(Mk * 4); a.rstd1k = (float*)c.take(Mk * 4);
            a.q = c.take(Mq * d * e);
            a.k = c.take(Mk * 2 * d * e); a.v = (char*)a.k + d * e;
            a.ldqkv_q = d; a.ldqkv_k = 2 * d;
        }
        a.att = c.take(Mq * d * e);
        a.lse = (float*)c.take((int64_t)B * H * a.NQ * 4);
        a.dmask = (training && m->d.attn_pdrop > 0.f && m->d.dtype == MEBT_BF16 && d / H == 64 && m->attn_bits)
                      ? (uint16_t*)c.take((int64_t)mebt_attn_dmask_bytes(B, (int)H, a.NQ, a.NK)) : nullptr;
        a.x = c.take(Mq * d * e);
        a.mean2 = (float*)c.take(Mq * 4); a.rstd2 = (float*)c.take(Mq * 4);
        a.hn = c.take(Mq * d * e);
        a.pre = training ? c.take(Mq * 4 * d * e) : nullptr;
        a.u = c.take(Mq * 4 * d * e);
        if (training || mode == MEBT_MODE_MASKGIT) a.out = c.take(Mq * d * e);
        else a.out = (mode == MEBT_MODE_LATENT_DEC) ? pingT[(nT++) & 1] : pingS[(nS++) & 1];
        if (c.off > layer_max) layer_max = c.off;
    }
    (void)layer_base;
    c.off = layer_max;
    const int64_t R = (int64_t)B * NT;
    x.hf = c.take(R * d * e);
    x.meanf = (float*)c.take(R * 4); x.rstdf = (float*)c.take(R * 4);
    x.logits_ws = nullptr;
    if (training) {
        x.row_lse = (float*)c.take(R * 4); x.row_loss = (float*)c.take(R * 4); x.row_rank = (int*)c.take(R * 4);
        x.loss_out = (double*)c.take(32);
        int64_t Mmax = (int64_t)B * (NS + NT);
        if ((int64_t)B * NC > Mmax) Mmax = (int64_t)B * NC;
        if (m->has_maskgit && (int64_t)B * (NC + NT) > Mmax) Mmax = (int64_t)B * (NC + NT);
        x.g_cat = m->has_maskgit ? c.take((int64_t)B * (NC + NT) * d * e) : nullptr;
        x.g_S = c.take(B * NS * d * e);
        x.g_T = c.take(R * d * e);
        x.g_C = (float*)c.take((int64_t)B * NC * d * 4);
        x.dlogits = c.take(R * V * e);
        x.dhf = c.take(R * d * e);
        x.datt = c.take(Mmax * d * e);
        for (int k = 0; k < 2; ++k) {
            FwdCtx::Scratch& s = x.sc[k];
            s.d4 = c.take(Mmax * 4 * d * e);
            s.dh = c.take(Mmax * d * e);
            s.dx = c.take(Mmax * d * e);
            s.dqkv_q = c.take(Mmax * 3 * d * e);
            s.dqkv_k = c.take(Mmax * 2 * d * e);
            s.dqn = c.take(Mmax * d * e);
            s.dkn = c.take(Mmax * d * e);
            s.dout_m = c.take(Mmax * d * e);
            s.dx_m = m->d.resid_pdrop > 0.f ? c.take(Mmax * d * e) : nullptr;
        }
        x.delta = (float*)c.take((int64_t)B * H * (m->has_maskgit && NC > NS ? NC + NT : NS + NT) * 4);
    }
}

extern "C" int64_t mebt_workspace_bytes(const mebt_model* m, int32_t B, int32_t NC, int32_t NT, int32_t training) {
    if (!m || B < 0 || NC < 0 || NT < 0) return -1;
    // measuring pass over a never-dereferenced, non-null base: sub-buffer pointers are derived by arithmetic on what take()
    // returns, and arithmetic on a null pointer is undefined behaviour (found by the UBSan host build, `make asan`)
    Carve c{reinterpret_cast<char*>((uintptr_t)1 << 30), 0};
    FwdCtx x;
    carve(m, c, x, B, NC, NT, training);
    return c.off + 256;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
static GemmParams gp(const void* A, const void* Bm, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kc, int b_kc) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = Bm; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.a_kc = a_kc; p.b_kc = b_kc;
    p.split_k = 1;
    return p;
}

static int ln_fwd(const mebt_model* m, const void* x, void* y, int64_t gw, int64_t gb, float* mean, float* rstd, int rows,
                  int seg, int seg_stride, int seg_off, hipStream_t st) {
    LnFwdParams p;
    p.x = x; p.y = y; p.gamma = m->P + gw; p.beta = m->P + gb; p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.d = m->d.n_embd; p.seg = seg; p.seg_stride = seg_stride; p.seg_off = seg_off;
    return launch_ln_fwd(p, m->d.dtype, st);
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// Weight prefetch along the dependency chain (GemmParams::pf): product g asks its workgroups to pull the bf16 weights of
// product g + 1 into the Infinity Cache.
static int pf_level() {      // MEBT_GEMM_PREFETCH: 0 off; 1 (default) the next product's weights; 2 also the saved activations backward reads
                             // next — measured SLOWER (11.27 vs 11.10 ms per step, level 0: 11.36; gpurun_out/r2m): kept as an experiment
    static const int lv = [] { const char* e = getenv("MEBT_GEMM_PREFETCH"); return e ? atoi(e) : 1; }();
    return lv;
}
static void set_pf(const mebt_model* m, GemmParams& p, int64_t w_off, int64_t n_elems) {
    if (pf_level() < 1 || m->d.dtype != MEBT_BF16 || w_off < 0 || n_elems <= 0) return;
    p.pf = m->Wop(w_off);
    p.pf_bytes = (unsigned)(n_elems * 2 > 0x7FFFFFFF ? 0x7FFFFFFF : n_elems * 2);
}
// second range: saved forward activations [lo, hi) of a block (contiguous in the training workspace, see carve())
static void set_pf2(const mebt_model* m, GemmParams& p, const void* lo, const void* hi) {
    if (pf_level() < 2 || m->d.dtype != MEBT_BF16 || !lo || !hi || hi <= lo) return;
    const size_t n = (const char*)hi - (const char*)lo;
    p.pf2 = lo;
    p.pf2_bytes = (unsigned)(n > 0x7FFFFFFFull ? 0x7FFFFFFFull : n);
}
// the two halves of a block's saved activations in the order backward consumes them
static void pf2_tail(const mebt_model* m, GemmParams& p, const LayerAct& a) { set_pf2(m, p, a.x, (const char*)a.out); }        // x, LN2 stats, hn, pre, u
static void pf2_head(const mebt_model* m, GemmParams& p, const LayerAct& a) { set_pf2(m, p, a.qn, (const char*)a.x); }          // qn, kn, q, k, v, att, lse

// Parameter dependencies of the next forward (sharded data parallelism: the all-gather of a bucket's updated weights may
// still be in flight when the forward starts; it only has to be complete when the first block of that bucket runs).
static int fw_wait(const mebt_model* m, int layer, hipStream_t st) {
    for (const auto& w : m->fw_waits)
        if (w.first == layer) {
            // profiling: an event pair around the wait = how long the compute stream really stood still for this bucket's
            // parameters (0 when the all-gather had finished before the forward got here)
            ProfRec r;
            if (g_prof_on) { r.a = get_event(); r.b = get_event(); r.flops = layer; r.bytes = 0; r.kind = 4; (void)hipEventRecord(r.a, st); }
            MEBT_HIP_CHECK(hipStreamWaitEvent(st, w.second, 0));
            if (g_prof_on) { (void)hipEventRecord(r.b, st); g_prof.push_back(r); }
        }
    return MEBT_OK;
}
static bool fw_pending(const mebt_model* m, int layer) {
    for (const auto& w : m->fw_waits)
        if (w.first == layer) return true;
    return false;
}
extern "C" int mebt_model_set_forward_waits(mebt_model* m, int32_t n, const int32_t* layer, void* const* events) {
    if (!m || n < 0 || (n > 0 && (!layer || !events))) { mebt_set_error("set_forward_waits: bad arguments"); return MEBT_EINVAL; }
    m->fw_waits.clear();
    for (int i = 0; i < n; ++i) {
        if (layer[i] < -1 || layer[i] > m->d.n_layer || !events[i]) { mebt_set_error("set_forward_waits: layer outside [-1, n_layer] or null event"); m->fw_waits.clear(); return MEBT_EINVAL; }
        m->fw_waits.emplace_back(layer[i], reinterpret_cast<hipEvent_t>(events[i]));
    }
    return MEBT_OK;
}

// Cache of the latent_enc blocks' key / value projections over ALL positions of the token grid (mebt_forward_kvcache): `contexts` is
// read-only through the network (gpt.py:187-192), so the K / V row of a context position depends on its token id, its position and
// the block's weights only — the sampling loops change a few hundred positions per forward and re-project thousands.
struct KvCacheArgs { void* cache; const int64_t* dirty; int ND; };

static int forward_impl(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t N, int32_t NC, int32_t NT,
                        const int64_t* x_ids, const int64_t* ci, const int64_t* ti, const float* const* embedded,
                        float* logits, int32_t training, uint64_t dropout_seed, mebt_stream_t stream, const KvCacheArgs* kv = nullptr) {
    if (!m || !m->W) { mebt_set_error("forward: model not bound"); return MEBT_EINVAL; }
    if (B <= 0 || N <= 0 || NC < 0 || NT <= 0) { mebt_set_error("forward: need B > 0, N > 0, NC >= 0, NT > 0"); return MEBT_ESHAPE; }
    if (N > m->d.block_size) { mebt_set_error("forward: sequence longer than block_size (pos_emb rows)"); return MEBT_ESHAPE; }
    if (!ws || !logits) { mebt_set_error("forward: null pointer"); return MEBT_EINVAL; }
    if (!embedded && (!x_ids || !ti || (NC > 0 && !ci))) { mebt_set_error("forward: null pointer"); return MEBT_EINVAL; }
    const bool drop_on = (training & 2) != 0 && (m->d.embd_pdrop > 0.f || m->d.resid_pdrop > 0.f || m->d.attn_pdrop > 0.f);
    const bool logits_lp = (training & 4) != 0;
    if (logits_lp && ((training & 1) || m->d.dtype != MEBT_BF16)) { mebt_set_error("forward: bf16 logits (flag 4) are for inference of a bf16 model"); return MEBT_EINVAL; }
    training = training & 1;
    const bool kvc = kv && kv->cache;
    if (kvc && (training || embedded || m->d.dtype != MEBT_BF16 || kv->ND < 0 || kv->ND > NC || (kv->ND > 0 && !kv->dirty))) {
        mebt_set_error("forward_kvcache: inference of a bf16 model on token ids, with 0 <= ND <= NC");
        return MEBT_EINVAL;
    }
    const int ND = kvc ? kv->ND : 0;
    FwdCtx& x = m->ctx;
    x.valid = false;
    Carve c{(char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), 0};
    carve(m, c, x, B, NC, NT, training);
    if (c.off + 256 > ws_bytes) { mebt_set_error("forward: workspace too small (see mebt_workspace_bytes)"); return MEBT_EWORKSPACE; }
    hipStream_t st = S(stream);
    const int d = m->d.n_embd, NS = m->d.n_latent, H = m->d.n_head, V = m->d.vocab, dt = m->d.dtype;
    RC(fw_wait(m, -1, st));                                // the non-Linear parameters (embeddings, every bias / LayerNorm)
    x.ws = ws; x.B = B; x.N = N; x.NC = NC; x.NT = NT; x.x_ids = x_ids; x.ci = ci; x.ti = ti;
    x.drop_on = drop_on; x.drop_seed = dropout_seed;
    const float p_emb = drop_on ? m->d.embd_pdrop : 0.f, p_res = drop_on ? m->d.resid_pdrop : 0.f, p_att = drop_on ? m->d.attn_pdrop : 0.f;

    if (embedded) {   // GPT.forward boundary (gpt.py:234): the caller hands over fp32 embeddings
        void* dst[3] = {x.sos0, x.ctx, x.tgt0};
        const size_t cnt[3] = {(size_t)B * NS * d, (size_t)B * NC * d, (size_t)B * NT * d};
        for (int k = 0; k < 3; ++k) {
            if (!cnt[k]) continue;
            if (!embedded[k]) { mebt_set_error("forward: null embedded input"); return MEBT_EINVAL; }
            if (dt == MEBT_BF16) RC(launch_cast_f32_to_bf16(embedded[k], dst[k], cnt[k], st));
            else MEBT_HIP_CHECK(hipMemcpyAsync(dst[k], embedded[k], cnt[k] * 4, hipMemcpyDeviceToDevice, st));
            static const uint32_t site[3] = {SITE_EMB_SOS, SITE_EMB_CTX, SITE_EMB_TGT};     // self.drop on the three inputs (gpt.py:238-240)
            if (p_emb > 0.f) RC(launch_apply_dropout(dst[k], dst[k], cnt[k], dt == MEBT_F32, dt == MEBT_F32, make_drop(dropout_seed, site[k], p_emb), st));
        }
        x.x_ids = nullptr; x.ci = nullptr; x.ti = nullptr;
    } else {
        EmbedParams ep;
        ep.x_ids = x_ids; ep.ci = kvc ? kv->dirty : ci; ep.ti = ti;       // cached: only the re-projected positions are embedded as contexts
        ep.tok_emb = m->P + m->tok_emb; ep.pos_emb = m->P + m->pos_emb; ep.mask_emb = m->P + m->mask_emb; ep.sos_emb = m->P + m->sos_emb;
        ep.sos = x.sos0; ep.ctx = x.ctx; ep.tgt = x.tgt0;
        ep.B = B; ep.N = N; ep.NC = kvc ? ND : NC; ep.NT = NT; ep.NS = NS; ep.d = d; ep.vocab = V; ep.block_size = m->d.block_size;
        ep.drop = make_drop(dropout_seed, 0, p_emb);       // gpt.py:238-240
        ProfRec r;
        if (g_prof_on) {     // algorithmic bytes: rows read once (fp32 tables), rows written once, the index vectors
            r.a = get_event(); r.b = get_event(); r.kind = 2; r.flops = 0;
            r.bytes = (double)B * (((double)2 * NC + NT + NS) * d * 4 + ((double)NC + NT + NS) * d * m->esz() + 8.0 * (NC + NT) + 8.0 * NC);
            (void)hipEventRecord(r.a, st);
        }
        RC(launch_embed_fwd(ep, dt, st));
        if (g_prof_on) { (void)hipEventRecord(r.b, st); g_prof.push_back(r); }
    }

    const void* Sv = x.sos0;
    const void* Tv = x.tgt0;
    const void* Cv = x.ctx;          // read-only unless the model has 'maskgit' blocks (gpt.py:191-192)
    if (kvc && m->has_maskgit) { mebt_set_error("forward_kvcache: 'maskgit' blocks rewrite the contexts"); return MEBT_EINVAL; }
    int enc_seen = 0;
    // keys / values of the cached blocks: gathered by the MFMA attention kernel through an int32 copy of `ci` (head size 64, at most
    // 8192 keys: the list lives in LDS), otherwise copied into the block's contiguous buffer
    const bool gather_on = [] { const char* e = getenv("MEBT_KV_GATHER"); return !(e && e[0] == '0'); }();      // read per forward (tests flip it)
    const bool kv_gather = kvc && gather_on && NC > 0 && NC <= 8192 && attn_fwd_can_gather(dt, d / H) && x.ci32 != nullptr;
    if (kv_gather) RC(launch_cast_i64_i32(ci, x.ci32, (size_t)B * NC, st));
    for (int i = 0; i < m->d.n_layer; ++i) {
        LayerAct& a = x.L[i];
        const LayerOffsets& o = m->lo[i];
        const int mode = m->d.modes[i];
        const bool kv_layer = kvc && mode == MEBT_MODE_LATENT_ENC;
        const int Mq = B * a.NQ, Mk = kv_layer ? B * ND : B * a.NK;       // cached: only the dirty positions are normalised and projected
        RC(fw_wait(m, i, st));                             // a bucket of Linear weights that starts at this block
        a.q_in = (mode == MEBT_MODE_LATENT_DEC) ? Tv : (mode == MEBT_MODE_MASKGIT) ? Cv : Sv;
        // LN1 on query and key with the SAME parameters (gpt.py:180-181), then the projections
        // (gpt.py:126-128); the three [d,d] weights are adjacent in W so QKV / KV fuse.  The key side
        // is independent of the query side: both LayerNorms are one launch, both projections are one launch.
        {
            LnFwdParams lj[MEBT_LN_MAXJ];
            int nj = 0;
            auto job = [&](const void* xin, void* y, float* mean, float* rstd, int rows, int seg, int seg_stride, int seg_off) {
                LnFwdParams& p = lj[nj++];
                p.x = xin; p.y = y; p.gamma = m->P + o.ln1w; p.beta = m->P + o.ln1b; p.mean = mean; p.rstd = rstd;
                p.rows = rows; p.d = d; p.seg = seg; p.seg_stride = seg_stride; p.seg_off = seg_off;
            };
            if (mode == MEBT_MODE_MASKGIT) {       // query = key = LN1(cat[contexts, targets]) (gpt.py:176-181)
                a.k_in = Tv;
                job(Cv, a.qn, a.mean1q, a.rstd1q, B * NC, NC, NC + NT, 0);
                job(Tv, a.qn, a.mean1q, a.rstd1q, B * NT, NT, NC + NT, NC);
            } else {
                job(a.q_in, a.qn, a.mean1q, a.rstd1q, Mq, 0, 0, 0);
            }
            if (mode == MEBT_MODE_LATENT_ENC) {
                a.k_in = Cv;
                job(Cv, a.kn, a.mean1k, a.rstd1k, Mk, 0, 0, 0);
            } else if (mode == MEBT_MODE_LATENT_DEC) {
                a.k_in = Sv;
                job(Sv, a.kn, a.mean1k, a.rstd1k, Mk, 0, 0, 0);
            } else if (mode == MEBT_MODE_LT2L) {   // key = LN1(cat[sos, targets]) (gpt.py:175,181)
                a.k_in = Tv;
                job(Sv, a.kn, a.mean1k, a.rstd1k, B * NS, NS, NS + NT, 0);
                job(Tv, a.kn, a.mean1k, a.rstd1k, B * NT, NT, NS + NT, NS);
            } else if (mode == MEBT_MODE_LATENT_SELF) {
                a.k_in = nullptr;
            }
            RC(launch_ln_fwd_multi(lj, nj, dt, st));
        }
        if (mode == MEBT_MODE_LATENT_SELF || mode == MEBT_MODE_MASKGIT) {
            GemmParams p = gp(a.qn, m->Wop(o.wq), a.q, Mq, 3 * d, d, d, d, 3 * d, 1, 1);
            p.bias = m->P + o.bq;
            set_pf(m, p, o.wp, (int64_t)d * d);
            RC(gemm(m, p, st));
        } else {
            GemmParams pk = gp(a.kn, m->Wop(o.wk), a.k, Mk, 2 * d, d, d, d, 2 * d, 1, 1);
            pk.bias = m->P + o.bk;
            pk.coarse_m = kv_layer ? 1 : 0;        // B * ND changes at every step of a sampling loop: few tuner signatures, not one per 128 rows
            set_pf(m, pk, o.wp, (int64_t)d * d);
            GemmParams p = gp(a.qn, m->Wop(o.wq), a.q, Mq, d, d, d, d, d, 1, 1);
            p.bias = m->P + o.bq;
            if (Mk <= 0) set_pf(m, p, o.wp, (int64_t)d * d);
            RC(gemm_pair(m, pk, p, st));         // key/value and query projections in one launch
            if (kv_layer) {     // dirty rows -> their positions of this block's cache; the block's keys / values = the cache rows at `ci`
                char* cache = (char*)kv->cache + (size_t)enc_seen * B * N * (2 * d) * 2;
                RC(launch_index_rows(a.k, cache, kv->dirty, B, ND, N, 2 * d * 2, 1, st));
                if (!kv_gather) RC(launch_index_rows(cache, a.k, ci, B, NC, N, 2 * d * 2, 0, st));   // no gathering attention for this shape: contiguous copy
            }
        }
        if (mode == MEBT_MODE_LATENT_ENC) ++enc_seen;
        // softmax(q k^T / sqrt(hd)) v  (gpt.py:131-137)
        AttnParams ap;
        memset(&ap, 0, sizeof(ap));
        ap.q = a.q; ap.k = a.k; ap.v = a.v; ap.o = a.att; ap.lse = a.lse;
        ap.B = B; ap.H = H; ap.NQ = a.NQ; ap.NK = a.NK; ap.HD = d / H;
        ap.ldq = a.ldqkv_q; ap.ldk = a.ldqkv_k; ap.ldv = a.ldqkv_k; ap.ldo = d;
        if (kv_layer && kv_gather) {       // the attention kernel reads the cache rows at `ci` itself (index list in LDS): no copy
            char* cache = (char*)kv->cache + (size_t)(enc_seen - 1) * B * N * (2 * d) * 2;
            ap.k = cache; ap.v = cache + (size_t)d * 2; ap.kidx = x.ci32; ap.kidx_rows = N;
        }
        ap.drop = make_drop(dropout_seed, 16 * i + SITE_ATTN, p_att);   // gpt.py:135
        ap.dmask = (p_att > 0.f && training) ? a.dmask : nullptr;
        RC(launch_attn_fwd(ap, dt, st));
        // x = LN1(query) + proj(att)   — residual on the NORMALISED query (gpt.py:180,184)
        {
            GemmParams p = gp(a.att, m->Wop(o.wp), a.x, Mq, d, d, d, d, d, 1, 1);
            p.bias = m->P + o.bp; p.epilogue = EPI_RESID; p.aux = a.qn; p.ld_aux = d;
            p.drop = make_drop(dropout_seed, 16 * i + SITE_PROJ, p_res);    // gpt.py:140
            set_pf(m, p, o.w1, (int64_t)4 * d * d);
            RC(gemm(m, p, st));
        }
        // x = x + mlp(LN2(x))  (gpt.py:185, 150-155)
        RC(ln_fwd(m, a.x, a.hn, o.ln2w, o.ln2b, a.mean2, a.rstd2, Mq, 0, 0, 0, st));
        {
            GemmParams p = gp(a.hn, m->Wop(o.w1), a.pre, Mq, 4 * d, d, d, d, 4 * d, 1, 1);
            p.bias = m->P + o.b1; p.epilogue = EPI_GELU; p.C2 = a.u;
            set_pf(m, p, o.w2, (int64_t)4 * d * d);
            RC(gemm(m, p, st));
        }
        {
            GemmParams p = gp(a.u, m->Wop(o.w2), a.out, Mq, d, 4 * d, 4 * d, 4 * d, d, 1, 1);
            p.bias = m->P + o.b2; p.epilogue = EPI_RESID; p.aux = a.x; p.ld_aux = d;
            p.drop = make_drop(dropout_seed, 16 * i + SITE_MLP, p_res);     // gpt.py:154
            if (fw_pending(m, i + 1)) {}                   // those weights may still be arriving: no touch before the wait
            else if (i + 1 < m->d.n_layer) set_pf(m, p, m->lo[i + 1].wq, (int64_t)3 * d * d);
            else set_pf(m, p, m->head_w, (int64_t)V * d);
            RC(gemm(m, p, st));
        }
        if (mode == MEBT_MODE_LATENT_DEC) Tv = a.out;                    // gpt.py:187-192
        else if (mode == MEBT_MODE_MASKGIT) {                            // contexts, targets = x[:, :NC], x[:, NC:]
            const int f32 = dt == MEBT_F32;
            RC(launch_copy_rows(a.out, a.c_out, (long)B * NC, d, f32, f32, NC, NC + NT, 0, 0, 0, 0, st));
            RC(launch_copy_rows(a.out, a.t_out, (long)B * NT, d, f32, f32, NT, NC + NT, NC, 0, 0, 0, st));
            Cv = a.c_out; Tv = a.t_out;
        } else Sv = a.out;
    }
    x.S_final = Sv; x.T_final = Tv;
    // logits = head(ln_f(targets))  (gpt.py:247-248; head has no bias)
    RC(ln_fwd(m, Tv, x.hf, m->lnf_w, m->lnf_b, x.meanf, x.rstdf, B * NT, 0, 0, 0, st));
    RC(fw_wait(m, m->d.n_layer, st));                      // the head weight
    {
        GemmParams p = gp(x.hf, m->Wop(m->head_w), logits, B * NT, V, d, d, d, V, 1, 1);
        p.c_f32 = logits_lp ? 0 : 1;
        RC(gemm(m, p, st));
    }
    m->fw_waits.clear();                                   // one-shot
    x.valid = training != 0;
    x.loss_done = false;
    x.dlogits_ready = false;
    return MEBT_OK;
}

extern "C" int mebt_forward(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t N, int32_t NC, int32_t NT,
                            const int64_t* x_ids, const int64_t* ci, const int64_t* ti, float* logits,
                            int32_t training, uint64_t dropout_seed, mebt_stream_t stream) {
    return forward_impl(m, ws, ws_bytes, B, N, NC, NT, x_ids, ci, ti, nullptr, logits, training, dropout_seed, stream);
}

// bytes of the key / value cache of mebt_forward_kvcache for B samples of N positions: [latent_enc blocks][B][N][2 d] bf16
extern "C" int64_t mebt_kvcache_bytes(const mebt_model* m, int32_t B, int32_t N) {
    if (!m || B <= 0 || N <= 0) return -1;
    int n_enc = 0;
    for (int i = 0; i < m->d.n_layer; ++i) n_enc += m->d.modes[i] == MEBT_MODE_LATENT_ENC;
    return (int64_t)n_enc * B * N * 2 * m->d.n_embd * 2;
}
// mebt_forward (inference, bf16 model) with the latent_enc blocks' keys / values taken from `kv_cache` at the positions `ci`.
// Before they are read, the rows of the positions dirty [B, ND] (a subset of every sample's `ci` row) are recomputed from x_ids —
// embedding, LN1 and key / value projection of each latent_enc block, on B * ND rows instead of B * NC — and stored.  The caller
// keeps the invariant that every position in `ci` either is in `dirty` or was projected by an earlier call with the token id it
// still has (the first call of a sampling loop passes dirty = ci).  flags: 4 = bf16 logits (see mebt_forward).
extern "C" int mebt_forward_kvcache(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t N, int32_t NC, int32_t NT,
                                    const int64_t* x_ids, const int64_t* ci, const int64_t* ti, void* logits, int32_t flags,
                                    void* kv_cache, const int64_t* dirty, int32_t ND, mebt_stream_t stream) {
    if (!kv_cache) { mebt_set_error("forward_kvcache: null cache"); return MEBT_EINVAL; }
    if (flags & ~4) { mebt_set_error("forward_kvcache: inference only (flags: 0 or 4)"); return MEBT_EINVAL; }
    const KvCacheArgs kv{kv_cache, dirty, ND};
    return forward_impl(m, ws, ws_bytes, B, N, NC, NT, x_ids, ci, ti, nullptr, reinterpret_cast<float*>(logits), flags, 0, stream, &kv);
}

extern "C" int mebt_gpt_forward(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t NC, int32_t NT,
                                const float* sos, const float* contexts, const float* targets, float* logits, mebt_stream_t stream) {
    const float* e[3] = {sos, contexts, targets};
    return forward_impl(m, ws, ws_bytes, B, 1, NC, NT, nullptr, nullptr, nullptr, e, logits, 0, 0, stream);
}

// GPT.forward in training mode on caller-embedded inputs: keeps the activations for mebt_gpt_backward
extern "C" int mebt_gpt_forward_train(mebt_model* m, void* ws, int64_t ws_bytes, int32_t B, int32_t NC, int32_t NT, const float* sos,
                                      const float* contexts, const float* targets, float* logits, int32_t dropout, uint64_t dropout_seed,
                                      mebt_stream_t stream) {
    const float* e[3] = {sos, contexts, targets};
    return forward_impl(m, ws, ws_bytes, B, 1, NC, NT, nullptr, nullptr, nullptr, e, logits, 1 | (dropout ? 2 : 0), dropout_seed, stream);
}

static int loss_impl(mebt_model* m, void* ws, const float* logits, double* out4, bool with_grad, float loss_scale, mebt_stream_t stream) {
    if (!m || !m->ctx.valid || m->ctx.ws != ws) { mebt_set_error("loss: no training-mode forward on this workspace"); return MEBT_EINVAL; }
    FwdCtx& x = m->ctx;
    if (!x.x_ids) { mebt_set_error("loss: the last forward ran on caller-embedded inputs (no token ids to score against)"); return MEBT_EINVAL; }
    if (!out4) out4 = x.loss_out;
    x.loss_done = true;
    CeParams p;
    p.logits = logits; p.x_ids = x.x_ids; p.ti = x.ti; p.rows = x.B * x.NT; p.V = m->d.vocab; p.B = x.B; p.N = x.N; p.NT = x.NT;
    p.label_smoothing = m->d.label_smoothing; p.row_lse = x.row_lse; p.row_loss = x.row_loss; p.row_rank = x.row_rank; p.out = out4;
    x.dlogits_ready = false;
    if (with_grad && x.dlogits && ce_fwd_can_fuse_grad(p.V)) {     // other vocabulary sizes: statistics only, mebt_backward_head runs the CE backward
        p.dlogits = x.dlogits; p.grad_scale = loss_scale; p.dl_bf16 = m->d.dtype == MEBT_BF16;
        x.dlogits_ready = true; x.dlogits_scale = loss_scale;
    }
    return launch_ce_fwd(p, S(stream));
}
extern "C" int mebt_loss(mebt_model* m, void* ws, const float* logits, double* out4, mebt_stream_t stream) {
    return loss_impl(m, ws, logits, out4, false, 0.f, stream);
}
extern "C" int mebt_loss_with_grad(mebt_model* m, void* ws, const float* logits, double* out4, float loss_scale, mebt_stream_t stream) {
    return loss_impl(m, ws, logits, out4, true, loss_scale, stream);
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
static int ln_bwd(const mebt_model* m, const void* x, const void* dy, const void* dy2, int64_t gw, int64_t gb, const float* mean,
                  const float* rstd, void* dx, int dx_f32, int acc, int rows, int seg, int seg_stride, int seg_off, hipStream_t st,
                  const void* dx_add = nullptr) {
    LnBwdParams p;
    p.x = x; p.dy = dy; p.dy2 = dy2; p.dx_add = dx_add; p.gamma = m->P + gw; p.mean = mean; p.rstd = rstd; p.dx = dx; p.dx_f32 = dx_f32;
    p.dx_accumulate = acc; p.dgamma = m->gP + gw; p.dbeta = m->gP + gb; p.rows = rows; p.d = m->d.n_embd;
    p.seg = seg; p.seg_stride = seg_stride; p.seg_off = seg_off;
    return launch_ln_bwd(p, m->d.dtype, st);
}

// dW[n_out,k_in] = dY^T X (reduction over tokens), into the fp32 gradient buffer
static int wgrad(const mebt_model* m, const void* dY, int ld_dy, const void* X, int ld_x, int64_t w_off, int n_out, int k_in, int tokens, hipStream_t st) {
    GemmParams p = gp(dY, X, m->gW + w_off, n_out, k_in, tokens, ld_dy, ld_x, k_in, 0, 0);
    p.c_f32 = 1; p.split_k = 0; p.beta = m->grad_acc ? 1 : 0;
    if (m->wire()) { p.C = (char*)m->gWb + (size_t)w_off * 2; p.c_f32 = 0; p.split_k = 1; }     // bf16 wire format, no fp32 copy
    return gemm(m, p, st);
}
// dX[tokens,k_in] = dY W  (+aux)
static int dgrad(const mebt_model* m, const void* dY, int ld_dy, int64_t w_off, void* dX, int tokens, int n_out, int k_in, int epilogue, const void* aux, int ld_aux, hipStream_t st,
                 int64_t pf_off = -1, int64_t pf_elems = 0, const LayerAct* pf2_layer = nullptr, int pf2_part = 0) {
    GemmParams p = gp(dY, m->Wop(w_off), dX, tokens, k_in, n_out, ld_dy, k_in, k_in, 1, 0);
    p.epilogue = epilogue; p.aux = aux; p.ld_aux = ld_aux;
    set_pf(m, p, pf_off, pf_elems);
    if (pf2_layer) { if (pf2_part) pf2_tail(m, p, *pf2_layer); else pf2_head(m, p, *pf2_layer); }
    return gemm(m, p, st);
}

// shared tail of the head backward: dlogits (already in x.dlogits) -> dW_head, d ln_f, g_T
static int head_backward_common(mebt_model* m, hipStream_t st) {
    FwdCtx& x = m->ctx;
    const int d = m->d.n_embd, V = m->d.vocab, R = x.B * x.NT;
    hipStream_t sd = st;
    if (m->use_side) {
        RC(fork_side(m, st));
        sd = m->side;
        // a fresh backward: mark both scratch sets free
        MEBT_HIP_CHECK(hipEventRecord(m->ev_layer[0], sd));
        MEBT_HIP_CHECK(hipEventRecord(m->ev_layer[1], sd));
    }
    RC(wgrad(m, x.dlogits, V, x.hf, d, m->head_w, V, d, R, sd));
    RC(dgrad(m, x.dlogits, V, m->head_w, x.dhf, R, V, d, EPI_NONE, nullptr, 0, st, m->lo[m->d.n_layer - 1].w2, (int64_t)4 * d * d,
             &x.L[m->d.n_layer - 1], 1));
    RC(ln_bwd(m, x.T_final, x.dhf, nullptr, m->lnf_w, m->lnf_b, x.meanf, x.rstdf, x.g_T, 0, 0, R, 0, 0, 0, st));
    x.gT_defined = true; x.gS_defined = false; x.gC_defined = false; x.doutm_ready = -1; x.last_bwd_lo = m->d.n_layer;
    return join_side(m, st);
}

static int backward_prologue(mebt_model* m, void* ws, hipStream_t st) {
    if (!m || !m->ctx.valid || m->ctx.ws != ws) { mebt_set_error("backward: no training-mode forward on this workspace"); return MEBT_EINVAL; }
    if (!m->gW || !m->gP) { mebt_set_error("backward: gradient buffers not bound"); return MEBT_EINVAL; }
    const int d = m->d.n_embd;
    if (m->grad_acc) {           // a further micro-batch: everything below adds to what the buffers hold
        if (m->fused_on) { mebt_set_error("backward: gradient accumulation and the optimizer-in-backward exclude each other"); return MEBT_EINVAL; }
        return MEBT_OK;
    }
    // P-side gradients are accumulated with atomics (LN affine, biases, embeddings): zero them first
    MEBT_HIP_CHECK(hipMemsetAsync(m->gP, 0, (size_t)m->n_p * 4, st));
    for (int i = 0; i < m->d.n_layer; ++i)
        if (!m->live[i]) {
            MEBT_HIP_CHECK(hipMemsetAsync(m->gW + m->lo[i].wq, 0, (size_t)12 * d * d * 4, st));
            if (m->wire()) MEBT_HIP_CHECK(hipMemsetAsync((char*)m->gWb + (size_t)m->lo[i].wq * 2, 0, (size_t)12 * d * d * 2, st));
        }
    return MEBT_OK;
}

extern "C" int mebt_backward_head(mebt_model* m, void* ws, const float* logits, const float* upstream, float loss_scale, mebt_stream_t stream) {
    hipStream_t st = S(stream);
    RC(backward_prologue(m, ws, st));
    FwdCtx& x = m->ctx;
    const int V = m->d.vocab, dt = m->d.dtype, R = x.B * x.NT;
    if (x.dlogits_ready && !upstream && x.dlogits_scale == loss_scale) {     // mebt_loss_with_grad already wrote the same dlogits
        x.dlogits_ready = false;
        return head_backward_common(m, st);
    }
    if (!x.loss_done) RC(mebt_loss(m, ws, logits, nullptr, stream));   // the CE backward needs the per-row logsumexp
    CeBwdParams cp;
    cp.logits = logits; cp.x_ids = x.x_ids; cp.ti = x.ti; cp.row_lse = x.row_lse; cp.dlogits = x.dlogits; cp.upstream = upstream;
    cp.scale = loss_scale; cp.label_smoothing = m->d.label_smoothing; cp.rows = R; cp.V = V; cp.B = x.B; cp.N = x.N; cp.NT = x.NT;
    RC(launch_ce_bwd(cp, dt, st));
    return head_backward_common(m, st);
}

// Backward from an arbitrary upstream gradient on the logits (a caller that computes its own loss
// on the returned logits, as the reference's shared_step does with F.cross_entropy).
extern "C" int mebt_backward_head_dlogits(mebt_model* m, void* ws, const float* dlogits, mebt_stream_t stream) {
    hipStream_t st = S(stream);
    RC(backward_prologue(m, ws, st));
    FwdCtx& x = m->ctx;
    if (!dlogits) { mebt_set_error("backward: null dlogits"); return MEBT_EINVAL; }
    x.dlogits_ready = false;       // x.dlogits is overwritten below: a later mebt_backward_head must not take it for the cross-entropy gradient
    const size_t n = (size_t)x.B * x.NT * m->d.vocab;
    if (m->d.dtype == MEBT_BF16) RC(launch_cast_f32_to_bf16(dlogits, x.dlogits, n, st));
    else MEBT_HIP_CHECK(hipMemcpyAsync(x.dlogits, dlogits, n * 4, hipMemcpyDeviceToDevice, st));
    return head_backward_common(m, st);
}

// fork: the side stream continues after everything enqueued on `st` so far
static int fork_side(mebt_model* m, hipStream_t st) {
    MEBT_HIP_CHECK(hipEventRecord(m->ev_fork, st));
    MEBT_HIP_CHECK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    return MEBT_OK;
}
// join: `st` continues after everything enqueued on the side stream so far
static int join_side(mebt_model* m, hipStream_t st) {
    if (!m->use_side) return MEBT_OK;
    MEBT_HIP_CHECK(hipEventRecord(m->ev_join, m->side));
    MEBT_HIP_CHECK(hipStreamWaitEvent(st, m->ev_join, 0));
    return MEBT_OK;
}

// Gradient leaves of one block (weight / bias gradients): collected while the critical path
// (dgrad chain, attention backward, LN dx) is enqueued on the main stream, then issued as ONE grouped
// wgrad GEMM + ONE grouped column-sum on the side stream, where they overlap the next block's chain.
struct Leaves {
    GroupedWgrad w;
    GroupedColsum c;
    Leaves() { w.n = 0; c.n = 0; }
    // `bias`: the layer's bias gradient [n_out] (+= column sums of dY); bf16 mode adds it up inside the grouped launch, the
    // fp32 parity mode keeps the separate column-sum pass
    void wgrad(const void* dY, int ld_dy, const void* X, int ld_x, float* C, int n_out, int k_in, int tokens, float* bias = nullptr, bool in_gemm = false) {
        GroupedWgrad::Item& it = w.g[w.n++];
        it.A = dY; it.B = X; it.C = C; it.M = n_out; it.N = k_in; it.K = tokens; it.lda = ld_dy; it.ldb = ld_x; it.ldc = k_in; it.ntx = 0;
        it.bias = in_gemm ? bias : nullptr;
        if (bias && !in_gemm) colsum(dY, tokens, n_out, ld_dy, bias);
    }
    void colsum(const void* X, int M, int N, int ldx, float* out) {
        GroupedColsum::Item& it = c.g[c.n++];
        it.X = X; it.out = out; it.M = M; it.N = N; it.ldx = ldx; it.gx = 0; it.rpb = 0;
    }
};

// AdamW (armed fused hyper-parameters) on the weight whose gradient starts at `g` (a contiguous [n_out,k_in] block of gW)
static int fused_adamw_slice(mebt_model* m, const float* g, int64_t n, hipStream_t st) {
    const int64_t off = g - m->gW;
    AdamWParams a;
    a.p = m->W + off; a.g = g; a.m = m->fused_mW + off; a.v = m->fused_vW + off; a.n = (size_t)n;
    a.p_bf16 = m->Wlp ? (void*)((char*)m->Wlp + off * 2) : nullptr;
    a.lr = m->fused_h.lr; a.beta1 = m->fused_h.beta1; a.beta2 = m->fused_h.beta2; a.eps = m->fused_h.eps;
    a.weight_decay = m->fused_h.weight_decay; a.bc1 = m->fused_h.bc1; a.bc2 = m->fused_h.bc2; a.grad_scale = m->fused_h.grad_scale;
    return launch_adamw(a, st);
}

// weight gradients of a block (one grouped launch); its bias column sums ride along with the block's LN1-backward
// launch (launch_ln_bwd_multi(..., &lv.c)) unless `with_colsum`
static int flush_leaves(mebt_model* m, Leaves& lv, hipStream_t sd, bool with_colsum = false) {
    const int dt = m->d.dtype;
    if (with_colsum) RC(launch_colsum_grouped(lv.c, dt, sd));
    for (int i = 0; i < lv.w.n; ++i) {      // empty reductions (NC = 0): the gradient is zero
        const GroupedWgrad::Item& it = lv.w.g[i];
        if (it.K <= 0 && it.M > 0 && it.N > 0 && !m->grad_acc) {
            MEBT_HIP_CHECK(hipMemset2DAsync(it.C, (size_t)it.ldc * 4, 0, (size_t)it.N * 4, it.M, sd));
            if (m->wire()) MEBT_HIP_CHECK(hipMemset2DAsync((char*)m->gWb + (size_t)(it.C - m->gW) * 2, (size_t)it.ldc * 2, 0, (size_t)it.N * 2, it.M, sd));
        }
    }
    if (dt == MEBT_BF16) {
        ProfRec r;
        const bool prof = g_prof_on;
        if (prof) {
            r.a = get_event(); r.b = get_event(); r.flops = 0; r.bytes = 0;
            r.tag = 'w'; r.dims[0] = lv.w.n;
            for (int i = 0; i < lv.w.n; ++i) {
                const GroupedWgrad::Item& it = lv.w.g[i];
                if (it.K <= 0) continue;
                r.dims[1] += it.M * it.N / 1024; if (it.K > r.dims[2]) r.dims[2] = it.K;
                r.flops += 2.0 * it.M * it.N * it.K;
                // fp32 gradient store, or (optimizer-in-backward) read p,m,v + write p,m,v and the bf16 mirror
                r.bytes += ((double)it.M * it.K + (double)it.N * it.K) * 2.0 + (double)it.M * it.N * (m->fused_on ? 26.0 : 4.0);
            }
            (void)hipEventRecord(r.a, sd);
        }
        lv.w.scratch = m->ctx.tune.flush ? &m->ctx.tune : nullptr;
        lv.w.beta = m->grad_acc ? 1 : 0;
        lv.w.gW = m->gW;
        lv.w.Cb = m->wire() ? m->gWb : nullptr;
        if (m->fused_on) {
            lv.w.fused = 1; lv.w.W = m->W; lv.w.gW = m->gW; lv.w.mW = m->fused_mW; lv.w.vW = m->fused_vW; lv.w.Wlp = m->Wlp;
            lv.w.opt = m->fused_h;
            for (int i = 0; i < lv.w.n; ++i) {      // an empty reduction still decays / advances the moments: its gradient is the zero just written
                const GroupedWgrad::Item& it = lv.w.g[i];
                if (it.K <= 0 && it.M > 0 && it.N > 0) RC(fused_adamw_slice(m, it.C, (int64_t)it.M * it.N, sd));
            }
        }
        const int rc = launch_wgrad_grouped(lv.w, dt, sd);
        if (prof) { (void)hipEventRecord(r.b, sd); g_prof.push_back(r); }
        return rc;
    }
    for (int i = 0; i < lv.w.n; ++i) {
        const GroupedWgrad::Item& it = lv.w.g[i];
        if (it.K > 0) {
            GemmParams p = gp(it.A, it.B, it.C, it.M, it.N, it.K, it.lda, it.ldb, it.ldc, 0, 0);
            p.c_f32 = 1; p.beta = m->grad_acc ? 1 : 0;
            RC(gemm(m, p, sd));
        }
        // fp32 parity mode has no fused epilogue: same semantics with the streaming kernel on this weight
        if (m->fused_on && it.M > 0 && it.N > 0) RC(fused_adamw_slice(m, it.C, (int64_t)it.M * it.N, sd));
    }
    return MEBT_OK;
}

// The flush point of a block's weight gradients.  bf16 engine, leaves on the main stream (MEBT_WGRAD_PAIR=0 switches it off): every
// other block parks its items and the next block launches both blocks' products together (see mebt_model::pend_w) - round 6, one
// MI355X, three alternating runs: 10.43 -> 10.07 ms per Sky-16f step, GEMM family 8.25 -> 7.86 ms (profiles/r06_wgrad_pair_ab.txt).
static int flush_or_park(mebt_model* m, Leaves& lv, hipStream_t sd, bool side) {
    static const bool pair_on = [] { const char* e = getenv("MEBT_WGRAD_PAIR"); return !(e && e[0] == '0'); }();
    if (!pair_on || side || m->d.dtype != MEBT_BF16 || lv.w.n > MEBT_MAX_GROUP / 2) return flush_leaves(m, lv, sd);
    if (!m->pend) {
        m->pend_w = lv.w;
        m->pend = true;
        return MEBT_OK;
    }
    for (int k = 0; k < m->pend_w.n && lv.w.n < MEBT_MAX_GROUP; ++k) lv.w.g[lv.w.n++] = m->pend_w.g[k];
    m->pend = false;
    return flush_leaves(m, lv, sd);
}
static int flush_parked(mebt_model* m, hipStream_t sd) {
    if (!m->pend) return MEBT_OK;
    Leaves lv;
    lv.w = m->pend_w;
    m->pend = false;
    return flush_leaves(m, lv, sd);
}

static int backward_layer(mebt_model* m, int i, hipStream_t st) {
    FwdCtx& x = m->ctx;
    LayerAct& a = x.L[i];
    const LayerOffsets& o = m->lo[i];
    const int mode = m->d.modes[i], d = m->d.n_embd, dt = m->d.dtype, B = x.B, NS = m->d.n_latent, NT = x.NT, H = m->d.n_head;
    const int Mq = B * a.NQ, Mk = B * a.NK;
    const bool isdec = mode == MEBT_MODE_LATENT_DEC, ismg = mode == MEBT_MODE_MASKGIT;
    if (ismg ? !(x.gT_defined || x.gC_defined) : isdec ? !x.gT_defined : !x.gS_defined) return MEBT_OK;   // the loss does not depend on this block
    const bool side = m->use_side;
    hipStream_t sd = side ? m->side : st;          // leaves (dW, db, dLN-affine) go here
    FwdCtx::Scratch& sc = x.sc[i & 1];
    if (side) MEBT_HIP_CHECK(hipStreamWaitEvent(st, m->ev_layer[i & 1], 0));   // side readers of this scratch set (layer i+2) are done
    const void* dout = isdec ? x.g_T : x.g_S;
    const int f32 = dt == MEBT_F32;
    const size_t esz = m->esz();
    if (ismg) {      // the block output spans both streams: dout = cat[g_C (fp32 accumulator), g_T]
        const int NC = x.NC;
        if (x.gC_defined) RC(launch_copy_rows(x.g_C, x.g_cat, (long)B * NC, d, 1, f32, 0, 0, 0, NC, NC + NT, 0, st));
        else if (NC > 0) MEBT_HIP_CHECK(hipMemset2DAsync(x.g_cat, (size_t)(NC + NT) * d * esz, 0, (size_t)NC * d * esz, B, st));
        if (x.gT_defined) RC(launch_copy_rows(x.g_T, x.g_cat, (long)B * NT, d, f32, f32, 0, 0, 0, NT, NC + NT, NC, st));
        else MEBT_HIP_CHECK(hipMemset2DAsync((char*)x.g_cat + (size_t)NC * d * esz, (size_t)(NC + NT) * d * esz, 0, (size_t)NT * d * esz, B, st));
        dout = x.g_cat;
    }
    const float p_res = x.drop_on ? m->d.resid_pdrop : 0.f, p_att = x.drop_on ? m->d.attn_pdrop : 0.f;
    Leaves lv;
    // out = x + dropout(u W2^T + b2): the branch gradient is dout * mask (the mask is recomputed, never
    // stored).  It is materialised in scratch either way: the leaves read it after this block's LN1
    // backward has overwritten the stream gradient.
    if (x.doutm_ready == i) x.doutm_ready = -1;     // written by the LN1 backward of the block above (see the end of this function)
    else if (p_res > 0.f) RC(launch_apply_dropout(dout, sc.dout_m, (size_t)Mq * d, f32, f32, make_drop(x.drop_seed, 16 * i + SITE_MLP, p_res), st));
    else MEBT_HIP_CHECK(hipMemcpyAsync(sc.dout_m, dout, (size_t)Mq * d * esz, hipMemcpyDeviceToDevice, st));
    const void* dmlp = sc.dout_m;
    static const bool bias_in_wgrad = [] { const char* e = getenv("MEBT_BIAS_IN_WGRAD"); return !(e && e[0] == '0'); }();
    const bool bg = dt == MEBT_BF16 && bias_in_wgrad;           // bias gradients inside the grouped weight-gradient launch
    const int64_t dd = (int64_t)d * d;
    lv.wgrad(dmlp, d, a.u, 4 * d, m->gW + o.w2, d, 4 * d, Mq, m->gP + o.b2, bg);
    RC(dgrad(m, dmlp, d, o.w2, sc.d4, Mq, d, 4 * d, EPI_GELU_BWD, a.pre, 4 * d, st, o.w1, 4 * dd));   // d(pre) = (dmlp W2) * gelu'(pre)
    lv.wgrad(sc.d4, 4 * d, a.hn, d, m->gW + o.w1, 4 * d, d, Mq, m->gP + o.b1, bg);
    RC(dgrad(m, sc.d4, 4 * d, o.w1, sc.dh, Mq, 4 * d, d, EPI_NONE, nullptr, 0, st, o.wp, dd, &a, 0));
    // dx = dout + LN2'(dh); the same kernel reduces dgamma/dbeta and writes the dropout-masked copy the
    // projection branch reads (x = qn + dropout(att Wp^T + bp))
    const void* dproj = sc.dx;
    {
        LnBwdParams p;
        p.x = a.x; p.dy = sc.dh; p.dy2 = nullptr; p.dx_add = dout; p.gamma = m->P + o.ln2w; p.mean = a.mean2; p.rstd = a.rstd2;
        p.dx = sc.dx; p.dx_f32 = f32; p.dx_accumulate = 0; p.dgamma = m->gP + o.ln2w; p.dbeta = m->gP + o.ln2b;
        p.rows = Mq; p.d = d; p.seg = 0; p.seg_stride = 0; p.seg_off = 0;
        if (p_res > 0.f) {
            p.dx2 = sc.dx_m; p.drop2 = make_drop(x.drop_seed, 16 * i + SITE_PROJ, p_res);
            dproj = sc.dx_m;
        }
        if (side) RC(fork_side(m, st));
        RC(launch_ln_bwd(p, dt, st, sd));
    }
    lv.wgrad(dproj, d, a.att, d, m->gW + o.wp, d, d, Mq, m->gP + o.bp, bg);
    RC(dgrad(m, dproj, d, o.wp, x.datt, Mq, d, d, EPI_NONE, nullptr, 0, st, o.wq, 3 * dd));
    // attention backward
    AttnParams ap;
    memset(&ap, 0, sizeof(ap));
    ap.q = a.q; ap.k = a.k; ap.v = a.v; ap.o = a.att; ap.lse = a.lse; ap.B = B; ap.H = H; ap.NQ = a.NQ; ap.NK = a.NK; ap.HD = d / H;
    ap.ldq = a.ldqkv_q; ap.ldk = a.ldqkv_k; ap.ldv = a.ldqkv_k; ap.ldo = d;
    ap.d_o = x.datt; ap.lddo = d; ap.delta = x.delta;
    ap.drop = make_drop(x.drop_seed, 16 * i + SITE_ATTN, p_att);
    ap.dmask = p_att > 0.f ? a.dmask : nullptr;
    if (mode == MEBT_MODE_LATENT_SELF || ismg) {
        ap.dq = sc.dqkv_q; ap.dk = (char*)sc.dqkv_q + (size_t)d * esz; ap.dv = (char*)sc.dqkv_q + (size_t)2 * d * esz;
        ap.lddq = ap.lddk = ap.lddv = 3 * d;
    } else {
        ap.dq = sc.dqkv_q; ap.lddq = d;
        ap.dk = sc.dqkv_k; ap.dv = (char*)sc.dqkv_k + (size_t)d * esz; ap.lddk = ap.lddv = 2 * d;
    }
    RC(launch_attn_bwd(ap, dt, st));
    // LN1 backward of the query and key sides: one launch (they share LN1's dgamma/dbeta)
    LnBwdParams lj[MEBT_LN_MAXJ];
    int nj = 0;
    auto ln1 = [&](const void* xin, const void* dy, const void* dy2, const float* mean, const float* rstd, void* dxp, int dx_f32, int acc,
                   int rows, int seg, int seg_stride, int seg_off) {
        LnBwdParams& p = lj[nj++];
        p.x = xin; p.dy = dy; p.dy2 = dy2; p.dx_add = nullptr; p.gamma = m->P + o.ln1w; p.mean = mean; p.rstd = rstd;
        p.dx = dxp; p.dx_f32 = dx_f32 || f32; p.dx_accumulate = acc; p.dgamma = m->gP + o.ln1w; p.dbeta = m->gP + o.ln1b;
        p.rows = rows; p.d = d; p.seg = seg; p.seg_stride = seg_stride; p.seg_off = seg_off;
    };
    if (mode == MEBT_MODE_LATENT_SELF || ismg) {
        lv.wgrad(sc.dqkv_q, 3 * d, a.qn, d, m->gW + o.wq, 3 * d, d, Mq, m->gP + o.bq, bg);
        RC(dgrad(m, sc.dqkv_q, 3 * d, o.wq, sc.dqn, Mq, 3 * d, d, EPI_RESID, sc.dx, d, st, i > 0 ? m->lo[i - 1].w2 : -1, 4 * dd,
                 i > 0 ? &x.L[i - 1] : nullptr, 1));   // + dx (residual on qn)
        if (side) RC(fork_side(m, st));
        RC(flush_or_park(m, lv, sd, side));
        if (ismg) {      // LN1 rows [0,NC) of each sample came from the contexts stream, the rest from the targets stream
            const int NC = x.NC;
            ln1(a.q_in, sc.dqn, nullptr, a.mean1q, a.rstd1q, x.g_C, 1, 0, B * NC, NC, NC + NT, 0);
            ln1(a.k_in, sc.dqn, nullptr, a.mean1q, a.rstd1q, x.g_T, 0, 0, B * NT, NT, NC + NT, NC);
            x.gC_defined = NC > 0; x.gT_defined = true;
        } else {
            ln1(a.q_in, sc.dqn, nullptr, a.mean1q, a.rstd1q, x.g_S, 0, 0, Mq, 0, 0, 0);
        }
    } else {
        lv.wgrad(sc.dqkv_q, d, a.qn, d, m->gW + o.wq, d, d, Mq, m->gP + o.bq, bg);
        lv.wgrad(sc.dqkv_k, 2 * d, a.kn, d, m->gW + o.wk, 2 * d, d, Mk, m->gP + o.bk, bg);
        {
            GemmParams pq = gp(sc.dqkv_q, m->Wop(o.wq), sc.dqn, Mq, d, d, d, d, d, 1, 0);
            pq.epilogue = EPI_RESID; pq.aux = sc.dx; pq.ld_aux = d;
            GemmParams pk = gp(sc.dqkv_k, m->Wop(o.wk), sc.dkn, Mk, d, 2 * d, 2 * d, d, d, 1, 0);
            if (i > 0) {
                GemmParams& lead = Mk > 0 ? pk : pq;
                set_pf(m, lead, m->lo[i - 1].w2, 4 * dd);
                pf2_tail(m, lead, x.L[i - 1]);
            }
            if (Mk > 0) RC(gemm_pair(m, pk, pq, st)); else RC(gemm(m, pq, st));
        }
        if (side) RC(fork_side(m, st));
        RC(flush_or_park(m, lv, sd, side));
        if (mode == MEBT_MODE_LATENT_ENC) {
            ln1(a.q_in, sc.dqn, nullptr, a.mean1q, a.rstd1q, x.g_S, 0, 0, Mq, 0, 0, 0);
            if (Mk > 0) {   // contexts feed every latent_enc block: accumulate in fp32
                ln1(a.k_in, sc.dkn, nullptr, a.mean1k, a.rstd1k, x.g_C, 1, x.gC_defined ? 1 : 0, Mk, 0, 0, 0);
                x.gC_defined = true;
            }
        } else if (mode == MEBT_MODE_LATENT_DEC) {
            ln1(a.q_in, sc.dqn, nullptr, a.mean1q, a.rstd1q, x.g_T, 0, 0, Mq, 0, 0, 0);
            ln1(a.k_in, sc.dkn, nullptr, a.mean1k, a.rstd1k, x.g_S, 0, x.gS_defined ? 1 : 0, Mk, 0, 0, 0);
            x.gS_defined = true;
        } else {   // lt2l: key rows [0,NS) come from the same LN as the query
            ln1(a.q_in, sc.dkn, sc.dqn, a.mean1k, a.rstd1k, x.g_S, 0, 0, B * NS, NS, NS + NT, 0);
            ln1(a.k_in, sc.dkn, nullptr, a.mean1k, a.rstd1k, x.g_T, 0, x.gT_defined ? 1 : 0, B * NT, NT, NS + NT, NS);
            x.gT_defined = true;
        }
    }
    // The job that finalises the stream gradient the block below consumes also writes that block's dropout-masked copy
    // (its MLP branch gradient, dout * mask): one elementwise launch less per block.  Not with the side stream: the
    // leaves of block i+1 may still be reading the other scratch set's dout_m.
    static const int fuse_doutm = [] { const char* e = getenv("MEBT_FUSE_DOUTM"); return e ? atoi(e) : 1; }();
    if (i > 0 && !side && p_res > 0.f && fuse_doutm && m->d.modes[i - 1] != MEBT_MODE_MASKGIT) {
        const bool below_dec = m->d.modes[i - 1] == MEBT_MODE_LATENT_DEC;
        const void* want = below_dec ? x.g_T : x.g_S;
        int k = -1;
        for (int j = 0; j < nj; ++j) if (lj[j].dx == want) k = j;       // the last job writing it
        if (k >= 0 && !lj[k].dx_f32) {
            lj[k].dx2 = x.sc[(i - 1) & 1].dout_m;
            lj[k].drop2 = make_drop(x.drop_seed, 16 * (i - 1) + SITE_MLP, p_res);
            x.doutm_ready = i - 1;
        }
    }
    RC(launch_ln_bwd_multi(lj, nj, dt, st, sd, &lv.c));
    if (side) MEBT_HIP_CHECK(hipEventRecord(m->ev_layer[i & 1], sd));   // this scratch set is free once the side stream gets here
    return MEBT_OK;
}

extern "C" int mebt_backward_layers(mebt_model* m, void* ws, int32_t layer_hi, int32_t layer_lo, mebt_stream_t stream) {
    if (!m || !m->ctx.valid || m->ctx.ws != ws) { mebt_set_error("backward: no training-mode forward on this workspace"); return MEBT_EINVAL; }
    if (layer_hi >= m->d.n_layer || layer_lo < 0 || layer_lo > layer_hi) { mebt_set_error("backward_layers: bad layer range"); return MEBT_EINVAL; }
    if (m->ctx.last_bwd_lo != layer_hi + 1) m->ctx.doutm_ready = -1;     // not the block right below the previous call's range
    m->pend = false;
    for (int i = layer_hi; i >= layer_lo; --i) RC(backward_layer(m, i, S(stream)));
    RC(flush_parked(m, S(stream)));            // an odd block count, or the last block of a gradient bucket
    m->ctx.last_bwd_lo = layer_lo;
    return join_side(m, S(stream));          // the caller may all-reduce these gradients next
}

// embd dropout (gpt.py:238-240) on the stream gradients that reached the network inputs
static int input_grad_dropout(mebt_model* m, hipStream_t st);

// Backward of GPT.forward (gpt.py:234-253) from dL/dlogits: parameter gradients of the blocks / ln_f / head into gW, gP and
// the gradients with respect to the three embedded inputs (fp32, like the inputs; NULL = not wanted).
extern "C" int mebt_gpt_backward(mebt_model* m, void* ws, const float* dlogits, float* d_sos, float* d_contexts, float* d_targets,
                                 mebt_stream_t stream) {
    RC(mebt_backward_head_dlogits(m, ws, dlogits, stream));
    RC(mebt_backward_layers(m, ws, m->d.n_layer - 1, 0, stream));
    FwdCtx& x = m->ctx;
    hipStream_t st = S(stream);
    RC(input_grad_dropout(m, st));
    const int d = m->d.n_embd, f32 = m->d.dtype == MEBT_F32;
    struct { float* dst; const void* src; bool defined; long rows; int src_f32; } o[3] = {
        {d_sos, x.g_S, x.gS_defined, (long)x.B * m->d.n_latent, f32}, {d_contexts, x.g_C, x.gC_defined, (long)x.B * x.NC, 1},
        {d_targets, x.g_T, x.gT_defined, (long)x.B * x.NT, f32}};
    for (auto& t : o) {
        if (!t.dst || t.rows <= 0) continue;
        if (t.defined) RC(launch_copy_rows(t.src, t.dst, t.rows, d, t.src_f32, 1, 0, 0, 0, 0, 0, 0, st));
        else MEBT_HIP_CHECK(hipMemsetAsync(t.dst, 0, (size_t)t.rows * d * 4, st));      // the logits do not depend on this input
    }
    return MEBT_OK;
}

static int input_grad_dropout(mebt_model* m, hipStream_t st) {
    FwdCtx& x = m->ctx;
    if (x.drop_on && m->d.embd_pdrop > 0.f) {
        const int f32 = m->d.dtype == MEBT_F32;
        const size_t dd = m->d.n_embd;
        const float pe = m->d.embd_pdrop;
        if (x.gS_defined) RC(launch_apply_dropout(x.g_S, x.g_S, (size_t)x.B * m->d.n_latent * dd, f32, f32, make_drop(x.drop_seed, SITE_EMB_SOS, pe), st));
        if (x.gT_defined) RC(launch_apply_dropout(x.g_T, x.g_T, (size_t)x.B * x.NT * dd, f32, f32, make_drop(x.drop_seed, SITE_EMB_TGT, pe), st));
        if (x.gC_defined) RC(launch_apply_dropout(x.g_C, x.g_C, (size_t)x.B * x.NC * dd, 1, 1, make_drop(x.drop_seed, SITE_EMB_CTX, pe), st));
    }
    return MEBT_OK;
}

extern "C" int mebt_backward_embed(mebt_model* m, void* ws, mebt_stream_t stream) {
    if (!m || !m->ctx.valid || m->ctx.ws != ws) { mebt_set_error("backward: no training-mode forward on this workspace"); return MEBT_EINVAL; }
    FwdCtx& x = m->ctx;
    if (!x.x_ids) { mebt_set_error("backward_embed: the last forward ran on caller-embedded inputs: use mebt_gpt_backward"); return MEBT_EINVAL; }
    RC(input_grad_dropout(m, S(stream)));
    EmbedBwdParams p;
    p.x_ids = x.x_ids; p.ci = x.ci; p.ti = x.ti;
    p.g_ctx = x.g_C; p.g_tgt = x.g_T; p.g_sos = x.g_S;
    p.g_tok_emb = m->gP + m->tok_emb; p.g_pos_emb = m->gP + m->pos_emb; p.g_mask_emb = m->gP + m->mask_emb; p.g_sos_emb = m->gP + m->sos_emb;
    p.B = x.B; p.N = x.N; p.NC = x.gC_defined ? x.NC : 0; p.NT = x.gT_defined ? x.NT : 0; p.NS = x.gS_defined ? m->d.n_latent : 0; p.d = m->d.n_embd;
    if (p.NC != x.NC && x.NC > 0) {   // contexts unused by any live block: only the target rows scatter; keep index strides right
        mebt_set_error("backward_embed: contexts without a live latent_enc block are not supported");
        return MEBT_EINVAL;
    }
    ProfRec r;
    if (g_prof_on) {         // stream gradients read once; fp32 atomic adds: 2 table rows per context row, 1 per target row; mask / sos column sums
        r.a = get_event(); r.b = get_event(); r.kind = 3; r.flops = 0;
        const double dd = p.d;
        r.bytes = (double)p.B * ((double)p.NC * dd * 4 + ((double)p.NT + p.NS) * dd * m->esz() + ((double)2 * p.NC + p.NT) * dd * 4);
        (void)hipEventRecord(r.a, S(stream));
    }
    const int rc = launch_embed_bwd(p, m->d.dtype, S(stream));
    if (g_prof_on) { (void)hipEventRecord(r.b, S(stream)); g_prof.push_back(r); }
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// optimiser
// ---------------------------------------------------------------------------------------------------
// AdamW over one gradient bucket: kind 0 = head weight, 1 = blocks layer_lo..layer_hi (their W and P
// slices), 2 = the P tail (ln_f, mask/sos/pos/tok embeddings), 3 = everything, 4 = everything EXCEPT the
// blocks' Linear weights (what is left after a backward with the fused optimizer armed).  Buckets are
// disjoint, so the caller may run each one on its own stream as soon as that bucket's gradients are final.
extern "C" int mebt_adamw_range(mebt_model* m, float* mW, float* vW, float* mP, float* vP, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int32_t step, float grad_scale, int32_t kind, int32_t layer_hi,
                                int32_t layer_lo, mebt_stream_t stream) {
    if (!m || !m->W || !m->gW || !m->gP) { mebt_set_error("adamw: model / gradients not bound"); return MEBT_EINVAL; }
    if (step < 1) { mebt_set_error("adamw: step must be >= 1"); return MEBT_EINVAL; }
    if (kind == 1 && (layer_hi >= m->d.n_layer || layer_lo < 0 || layer_lo > layer_hi)) { mebt_set_error("adamw: bad layer range"); return MEBT_EINVAL; }
    hipStream_t st = S(stream);
    AdamWParams a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    a.bc1 = (float)(1.0 - pow((double)beta1, step)); a.bc2 = (float)(1.0 - pow((double)beta2, step));
    const int d = m->d.n_embd;
    auto run = [&](float* p, const float* g, float* mm, float* vv, void* lp, int64_t off, int64_t n, float wd) -> int {
        a.p = p + off; a.g = g + off; a.m = mm + off; a.v = vv + off; a.n = (size_t)n; a.weight_decay = wd;
        a.p_bf16 = lp ? (void*)((char*)lp + off * 2) : nullptr;
        return launch_adamw(a, st);
    };
    if (kind == 1 || kind == 3 || kind == 4) {   // contiguous runs of live layers (torch skips parameters whose grad is None)
        const int lo = kind != 1 ? 0 : layer_lo, hi = kind != 1 ? m->d.n_layer - 1 : layer_hi;
        int i = lo;
        while (i <= hi) {
            if (!m->live[i]) { ++i; continue; }
            int j = i;
            while (j + 1 <= hi && m->live[j + 1]) ++j;
            if (kind != 4) RC(run(m->W, m->gW, mW, vW, m->Wlp, m->lo[i].wq, (int64_t)(j - i + 1) * 12 * d * d, weight_decay));
            RC(run(m->P, m->gP, mP, vP, nullptr, m->lo[i].ln1w, (int64_t)(j - i + 1) * 13 * d, 0.f));
            i = j + 1;
        }
    }
    if (kind == 0 || kind == 3 || kind == 4) RC(run(m->W, m->gW, mW, vW, m->Wlp, m->head_w, (int64_t)m->d.vocab * d, weight_decay));
    if (kind == 2 || kind == 3 || kind == 4) {   // ln_f, mask_emb, sos_emb, pos_emb (always reached) and tok_emb (only through a live latent_enc)
        const int64_t tail_n = (m->tok_live ? m->n_p : m->tok_emb) - m->lnf_w;
        RC(run(m->P, m->gP, mP, vP, nullptr, m->lnf_w, tail_n, 0.f));
    }
    return MEBT_OK;
}

// AdamW on an arbitrary slice [off, off + n) of the flat W (which = 0: decayed, bf16 mirror refreshed) or P (which = 1)
// buffer, with the gradient given separately: `grad` points at the gradient of element `off` (fp32, or bf16 when
// grad_bf16).  This is the update a data-parallel rank applies to ITS shard of a gradient bucket after the
// reduce-scatter (ZeRO-1 style: m, v and the fp32 master are only ever touched on the owning rank).  Elements of blocks
// the loss cannot reach are skipped, exactly like mebt_adamw_range.
static int adamw_slice_impl(mebt_model* m, int32_t which, int64_t off, int64_t n, const void* grad, int32_t grad_bf16, int32_t pieces, float* mW,
                            float* vW, float* mP, float* vP, float lr, float beta1, float beta2, float eps, float weight_decay,
                            int32_t step, float grad_scale, mebt_stream_t stream) {
    if (!m || !m->W || !grad) { mebt_set_error("adamw_slice: model not bound / null gradient"); return MEBT_EINVAL; }
    if (pieces < 1 || (pieces > 1 && !grad_bf16)) { mebt_set_error("adamw_slice: pieces >= 1, several pieces only of bf16 gradients"); return MEBT_EINVAL; }
    if (step < 1) { mebt_set_error("adamw: step must be >= 1"); return MEBT_EINVAL; }
    const int64_t total = which == 0 ? m->n_w : m->n_p;
    if (off < 0 || n < 0 || off + n > total || (off % 4) || (n % 4)) { mebt_set_error("adamw_slice: bad range (multiples of 4 inside the buffer)"); return MEBT_EINVAL; }
    hipStream_t st = S(stream);
    AdamWParams a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale; a.g_bf16 = grad_bf16 ? 1 : 0;
    a.g_pieces = pieces; a.g_stride = (size_t)n;      // piece j of the gradient starts n elements behind piece j - 1
    a.bc1 = (float)(1.0 - pow((double)beta1, step)); a.bc2 = (float)(1.0 - pow((double)beta2, step));
    a.weight_decay = which == 0 ? weight_decay : 0.f;
    const int d = m->d.n_embd;
    const int64_t gsz = grad_bf16 ? 2 : 4;
    auto run = [&](int64_t lo, int64_t hi) -> int {          // intersection of [lo, hi) (a live range) with the slice
        const int64_t a0 = lo > off ? lo : off, a1 = hi < off + n ? hi : off + n;
        if (a1 <= a0) return MEBT_OK;
        float* pb = which == 0 ? m->W : m->P;
        a.p = pb + a0; a.m = (which == 0 ? mW : mP) + a0; a.v = (which == 0 ? vW : vP) + a0; a.n = (size_t)(a1 - a0);
        a.g = reinterpret_cast<const float*>(reinterpret_cast<const char*>(grad) + (a0 - off) * gsz);
        a.p_bf16 = (which == 0 && m->Wlp) ? (void*)((char*)m->Wlp + a0 * 2) : nullptr;
        return launch_adamw(a, st);
    };
    for (int i = 0; i < m->d.n_layer; ++i) {
        if (!m->live[i]) continue;
        if (which == 0) RC(run(m->lo[i].wq, m->lo[i].wq + (int64_t)12 * d * d));
        else RC(run(m->lo[i].ln1w, m->lo[i].ln1w + (int64_t)13 * d));
    }
    if (which == 0) RC(run(m->head_w, m->n_w));
    else RC(run(m->lnf_w, m->tok_live ? m->n_p : m->tok_emb));
    return MEBT_OK;
}
extern "C" int mebt_adamw_slice(mebt_model* m, int32_t which, int64_t off, int64_t n, const void* grad, int32_t grad_bf16, float* mW,
                                float* vW, float* mP, float* vP, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int32_t step, float grad_scale, mebt_stream_t stream) {
    return adamw_slice_impl(m, which, off, n, grad, grad_bf16, 1, mW, vW, mP, vP, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}
// The gradient of the slice arrives as `pieces` bf16 copies of it (piece j = rank j's contribution to this rank's shard, received by
// an all-to-all; piece j at grad + j * n elements): the kernel adds them up in fp32, in piece order, and applies AdamW to the sum —
// the reference's DDP sums fp32 gradients (train_transformer.py:39-41); a bf16 reduce-scatter lets RCCL add in bf16 (one rounding
// per ring hop), this keeps 2 B per parameter on the wire and the sum in fp32 (one rounding per RANK, where each rank rounded its
// own fp32 accumulators once).
extern "C" int mebt_adamw_slice_pieces(mebt_model* m, int32_t which, int64_t off, int64_t n, const void* grad_pieces, int32_t pieces, float* mW,
                                       float* vW, float* mP, float* vP, float lr, float beta1, float beta2, float eps, float weight_decay,
                                       int32_t step, float grad_scale, mebt_stream_t stream) {
    return adamw_slice_impl(m, which, off, n, grad_pieces, 1, pieces, mW, vW, mP, vP, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

// bf16 wire-format gradients for the data-parallel path: with a buffer bound (bf16 compute mode, no accumulation, no fused
// optimizer) the weight-gradient launches round their fp32 accumulators once and store bf16 there instead of fp32 in gW —
// the reduce-scatter reads it as is (no fp32 store, no cast pass: 6 bytes per parameter less HBM traffic per step).
extern "C" int mebt_model_bind_wire_grads(mebt_model* m, void* gWb) {
    if (!m) { mebt_set_error("bind_wire_grads: null model"); return MEBT_EINVAL; }
    m->gWb = gWb;
    return MEBT_OK;
}

// Gradient accumulation over micro-batches (reference train_transformer.py:46-49, Lightning accumulate_grad_batches):
// on = 1 makes the following backward ADD its gradients to gW / gP (fp32 C += in the weight-gradient epilogues, no
// zero-fill of the atomically accumulated P side); on = 0 (default) overwrites.
extern "C" int mebt_model_set_grad_accumulate(mebt_model* m, int32_t on) {
    if (!m) { mebt_set_error("set_grad_accumulate: null model"); return MEBT_EINVAL; }
    m->grad_acc = on != 0;
    return MEBT_OK;
}

// Arm (step >= 1) or disarm (step <= 0) the optimizer-in-backward for the blocks' Linear weights.
extern "C" int mebt_model_set_fused_adamw(mebt_model* m, float* mW, float* vW, float lr, float beta1, float beta2, float eps,
                                          float weight_decay, int32_t step, float grad_scale) {
    if (!m) { mebt_set_error("set_fused_adamw: null model"); return MEBT_EINVAL; }
    if (step <= 0) { m->fused_on = false; return MEBT_OK; }
    if (!mW || !vW || !m->W || !m->gW) { mebt_set_error("set_fused_adamw: optimizer state / model not bound"); return MEBT_EINVAL; }
    m->fused_on = true; m->fused_mW = mW; m->fused_vW = vW;
    m->fused_h = {lr, beta1, beta2, eps, weight_decay, (float)(1.0 - pow((double)beta1, step)), (float)(1.0 - pow((double)beta2, step)), grad_scale};
    return MEBT_OK;
}

extern "C" int mebt_adamw_step(mebt_model* m, float* mW, float* vW, float* mP, float* vP, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int32_t step, float grad_scale, mebt_stream_t stream) {
    return mebt_adamw_range(m, mW, vW, mP, vP, lr, beta1, beta2, eps, weight_decay, step, grad_scale, 3, 0, 0, stream);
}

// Test hook: the keep-scale (0 or 1/(1-p)) of elements 0..n-1 of a dropout site, as the kernels compute it.
extern "C" int mebt_debug_dropout_mask(uint64_t seed, uint32_t site, float p, int64_t n, float* out, mebt_stream_t stream) {
    if (!out || n < 0 || (n % 4)) { mebt_set_error("dropout_mask: bad arguments (n must be a multiple of 4)"); return MEBT_EINVAL; }
    return launch_apply_dropout(out, out, (size_t)n, 1, 1, make_drop(seed, site, p), S(stream));
}
