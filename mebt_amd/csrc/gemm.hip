// Host side of the bf16 / f32 GEMM family: validation, in-situ autotuning, split-K scratch, and dispatch to the four
// per-layout translation units (gemm_kk/kr/rr/rk.hip, which hold the bf16 kernel instantiations; the device code and
// the design notes are in gemm_kernels.h).  The fp32 parity kernel and the split-K reduce kernel are instantiated here.
#include "gemm_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <cstring>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

void mebt_gemm_cfg_kk(const GemmParams&, int, int, int, int, hipStream_t);
void mebt_gemm_cfg_kr(const GemmParams&, int, int, int, int, hipStream_t);
void mebt_gemm_cfg_rr(const GemmParams&, int, int, int, int, hipStream_t);
void mebt_gemm_cfg_rk(const GemmParams&, int, int, int, int, hipStream_t);
void mebt_gemm_ks2_kk(const GemmParams&, int, int, int, hipStream_t);
void mebt_gemm_ks2_kr(const GemmParams&, int, int, int, hipStream_t);
void mebt_gemm_ks2_rr(const GemmParams&, int, int, int, hipStream_t);
void mebt_gemm_ks2_rk(const GemmParams&, int, int, int, hipStream_t);
void mebt_gemm_w8_kk(const GemmParams&, int, hipStream_t);
void mebt_gemm_w8_kr(const GemmParams&, int, hipStream_t);
void mebt_gemm_w8_rr(const GemmParams&, int, hipStream_t);
void mebt_gemm_w8_rk(const GemmParams&, int, hipStream_t);
void mebt_gemm_pp_kk(const GemmParams&, hipStream_t);
void mebt_gemm_pp_kr(const GemmParams&, hipStream_t);
void mebt_gemm_pp_rr(const GemmParams&, hipStream_t);
void mebt_gemm_pp_rk(const GemmParams&, hipStream_t);
int mebt_gemm_attrs_kk(); int mebt_gemm_attrs_kr(); int mebt_gemm_attrs_rr(); int mebt_gemm_attrs_rk();
void mebt_gemm_pair_kk(GemmPair&, int, int, int, hipStream_t);
void mebt_gemm_pair_kr(GemmPair&, int, int, int, hipStream_t);
void mebt_gemm_grouped(GroupedWgrad&, int, int, int, hipStream_t);


static void launch_bf16_ks2(const GemmParams& p, int tbm, int tbn, int ring, hipStream_t stream) {
    if (p.a_kc && p.b_kc) mebt_gemm_ks2_kk(p, tbm, tbn, ring, stream);
    else if (p.a_kc && !p.b_kc) mebt_gemm_ks2_kr(p, tbm, tbn, ring, stream);
    else if (!p.a_kc && !p.b_kc) mebt_gemm_ks2_rr(p, tbm, tbn, ring, stream);
    else mebt_gemm_ks2_rk(p, tbm, tbn, ring, stream);
}
static void launch_pair_config(GemmPair& g, int tbm, int tbn, int staging, hipStream_t stream) {
    if (g.p[0].b_kc) mebt_gemm_pair_kk(g, tbm, tbn, staging, stream);
    else mebt_gemm_pair_kr(g, tbm, tbn, staging, stream);
}
static void launch_grouped_config(GroupedWgrad& c, int tbm, int tbn, int stages, hipStream_t stream) { mebt_gemm_grouped(c, tbm, tbn, stages, stream); }


// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
static int g_gemm_force_split = 0;
static int g_gemm_force_tile = 0;     // (BM << 12) | BN, tests / tools only
static int g_grouped_stages = 2;
extern "C" void mebt_debug_grouped_stages(int n) { g_grouped_stages = n; }
static int g_grouped_force = 0;       // (tbm << 20) | (tbn << 8) | ring: tools/wgrad_bench.py times tile shapes the table does not hold
extern "C" void mebt_debug_grouped_config(int32_t tbm, int32_t tbn, int32_t ring) { g_grouped_force = (tbm && tbn) ? ((tbm << 20) | (tbn << 8) | ring) : 0; }
static int g_gemm_dma = -1;           // -1 autotune / heuristic; forced (tests, tools): 0 register-staged, 2..5 LDS-DMA ring depth, 16+r two pipelines, 32+r / 64+r split-K 2 / 4
static int g_gemm_nostore = 0;        // experiments only (variant >= 100): skip the C store of plain epilogues
extern "C" void mebt_debug_gemm_variant(int dma) { g_gemm_nostore = dma >= 100; g_gemm_dma = dma >= 100 ? (dma == 199 ? -1 : dma - 100) : dma; }
void mebt_gemm_force_split(int s) { g_gemm_force_split = s; }
static unsigned long long* g_gemm_stamps = nullptr;
extern "C" void mebt_debug_gemm_stamps(unsigned long long* buf) { g_gemm_stamps = buf; }
extern "C" void mebt_debug_gemm_tile(int bm, int bn) { g_gemm_force_tile = (bm && bn) ? ((bm << 12) | bn) : 0; }

// one bf16 launch with an explicit (block tile, staging) choice; staging 0 = register-staged 2 stages,
// 2..5 = LDS-DMA ring with that many stages (clamped to what the tile's LDS footprint admits)
// scratch for split-K partials (library-owned, grown on demand; launches that use it are ordered on ONE stream: the
// autotuner only offers split configurations for bf16-output products, i.e. the forward / dgrad chain)
// Scratch of the tuner and of the split-K variant.  Caller-owned (the engine carves it out of the workspace PyTorch
// allocated, `GemmParams::scratch`); operator-level callers without one get the heuristic configuration, or the
// process-wide buffer a benchmark tool installed through mebt_debug_gemm_scratch().
static GemmScratch g_default_scratch = {nullptr, 0, nullptr, 0};
static const GemmScratch* scratch_of(const GemmScratch* s) { return (s && s->flush) ? s : (g_default_scratch.flush ? &g_default_scratch : nullptr); }
// Tuning needs the flush buffer at its real size: a model created while tuning was off carved 16 bytes (engine.cpp carve()), and
// candidates timed behind a 16-byte "flush" are timed warm — the regime that measured worse picks.  Such a model
// keeps the heuristic / cached configurations and says so once.
static const GemmScratch* tune_scratch_of(const GemmScratch* s) {
    const GemmScratch* sc = scratch_of(s);
    if (sc && sc->flush_bytes < (64u << 20)) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "[mebt gemm autotune] tuning was switched on after this model's workspace was laid out without the cache-flush buffer: "
                            "its launches keep the heuristic / cached configurations (create the model with tuning on to tune in situ)\n");
        }
        return nullptr;
    }
    return sc;
}
extern "C" void mebt_debug_gemm_scratch(void* buf, int64_t bytes) {
    if (!buf || bytes < (int64_t)(MEBT_TUNE_FLUSH_BYTES + MEBT_TUNE_SPLITK_BYTES)) { g_default_scratch = {nullptr, 0, nullptr, 0}; return; }
    g_default_scratch = {buf, MEBT_TUNE_FLUSH_BYTES, (float*)((char*)buf + MEBT_TUNE_FLUSH_BYTES), (size_t)bytes - MEBT_TUNE_FLUSH_BYTES};
}
static void launch_bf16_config(const GemmParams& p, int tbm, int tbn, int staging, int split, hipStream_t stream) {
    if (tbm == 256 && tbn == 256 && staging == 9 && p.a_kc && p.K % BK == 0) {   // two staggered wave groups (gemm_bf16_pp_kernel)
        if (p.b_kc) mebt_gemm_pp_kk(p, stream); else mebt_gemm_pp_kr(p, stream);
        return;
    }
    if (tbm == 256 && tbn == 256) {   // the 8-wave tile: whole reduction in the workgroup, ring 2
        if (p.a_kc && p.b_kc) mebt_gemm_w8_kk(p, 2, stream);
        else if (p.a_kc && !p.b_kc) mebt_gemm_w8_kr(p, 2, stream);
        else if (!p.a_kc && !p.b_kc) mebt_gemm_w8_rr(p, 2, stream);
        else mebt_gemm_w8_rk(p, 2, stream);
        return;
    }
    if (staging >= 16) {          // two pipelines: whole reduction in the workgroup, an even number of k-tiles
        if (split == 1 && p.K % (2 * BK) == 0 && ks2_lds(tbm, tbn, staging - 16)) { launch_bf16_ks2(p, tbm, tbn, staging - 16, stream); return; }
        staging -= 16;
    }
    if (p.a_kc && p.b_kc) mebt_gemm_cfg_kk(p, tbm, tbn, staging, split, stream);
    else if (p.a_kc && !p.b_kc) mebt_gemm_cfg_kr(p, tbm, tbn, staging, split, stream);
    else if (!p.a_kc && !p.b_kc) mebt_gemm_cfg_rr(p, tbm, tbn, staging, split, stream);
    else mebt_gemm_cfg_rk(p, tbm, tbn, staging, split, stream);
}
static bool splitk_fits(const GemmParams& p, int S) {
    const GemmScratch* sc = scratch_of(p.scratch);
    return sc && sc->splitk && (size_t)S * p.M * p.N * 4 <= sc->splitk_bytes;
}
static int launch_bf16_splitk(const GemmParams& p, int tbm, int tbn, int ring, int S, hipStream_t stream) {
    const GemmScratch* sc = scratch_of(p.scratch);
    if (!splitk_fits(p, S)) { mebt_set_error("gemm: split-K scratch missing or too small"); return MEBT_EWORKSPACE; }
    float* g_sk_buf = sc->splitk;
    GemmParams q = p;
    q.C = g_sk_buf; q.C2 = nullptr; q.c_f32 = 1; q.ldc = p.N; q.epilogue = EPI_NONE; q.bias = nullptr; q.aux = nullptr; q.beta = 0;
    q.drop.thresh = 0; q.slab = (long)p.M * p.N;
    launch_bf16_config(q, tbm, tbn, ring, S, stream);
    const long total = (long)p.M * (p.N / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (S == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3(blocks), dim3(256), 0, stream, p, g_sk_buf, q.slab);
    else hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, stream, p, g_sk_buf, q.slab);
    return MEBT_OK;
}

// ------------------------------------------------------------------------------------------------
// In-situ autotuning.  A train step has ~25 distinct GEMM signatures and for each one the best block
// tile is the one whose tile count lands on a whole number of workgroup slots of the 256 CUs
// (profiles/r01_gemm_variants_cold.txt: up to 1.35x between neighbouring tiles, no closed-form rule
// survives all shapes).  The first launch of a signature times every (tile, ring depth) candidate on the
// caller's stream with the caller's operands and keeps the fastest.  The plain candidates accumulate every
// output element in the same k order (identical bits); the two-pipeline and split-K candidates add two or four
// partial sums at the end, i.e. differ in the last fp32 bits before the bf16 rounding of the output — inside
// the bf16 tolerance of the parity tests, and the fp32 parity mode is never tuned.  Only idempotent
// launches are tuned (no beta accumulation, no split-K atomics).  MEBT_GEMM_AUTOTUNE=0 disables it
// (heuristic below), MEBT_GEMM_TUNE_LOG=1 prints the choices.
// ------------------------------------------------------------------------------------------------
// A signature = {kind | flags, then (M, N, K) of every product of the launch} with the real dimensions (no hashing:
// two different shape sets never share an entry).  The token-count dimensions (M of forward / dgrad, K of the weight
// gradients) are bucketed to multiples of 128: in real training t ~ U(0,1) makes NC / NT differ at almost every step,
// and an exact key would tune ~20 new signatures per step forever; bucketed, a config has 48 buckets per token count
// (weight dimensions are multiples of 128 and unaffected).  The table is process-wide (a pure shape -> configuration
// cache) and guarded by a mutex; tuning itself runs under the same lock with the CALLER's scratch and events of its own.
typedef std::vector<int> TuneKey;
static int tune_bucket(int v) { return v <= 128 ? v : ((v + 127) / 128) * 128; }
static int tune_bucket_m(const GemmParams& p) {          // GemmParams::coarse_m: next power of two above 128 rows
    if (!p.coarse_m || p.M <= 128) return tune_bucket(p.M);
    int b = 128;
    while (b < p.M) b <<= 1;
    return b;
}
static std::map<TuneKey, int> g_tuned;      // -> (tbm << 20) | (tbn << 8) | staging
// the fastest few candidates of every signature tuned in this process, with their isolated (cold) times: what tools/step_tune.py
// tries one by one IN the train step (a candidate that is second in isolation can be first between its real neighbours)
static std::map<TuneKey, std::vector<std::pair<int, float>>> g_alts;
static void alts_keep(const TuneKey& key, std::vector<std::pair<int, float>> v) {
    std::sort(v.begin(), v.end(), [](const std::pair<int, float>& a, const std::pair<int, float>& b) { return a.second < b.second; });
    if (v.size() > 4) v.resize(4);
    g_alts[key] = v;
}
static std::mutex g_tune_mutex;
static int g_autotune = -1, g_tune_log = 0;
static const char* g_tune_cache = nullptr;                       // MEBT_GEMM_TUNE_CACHE: text file of tuned choices

// environment: MEBT_GEMM_AUTOTUNE=0 disables tuning, MEBT_GEMM_TUNE_LOG=1 prints choices, MEBT_GEMM_TUNE_CACHE=<file>
// loads earlier choices at start-up and appends new ones (a later process then launches no tuning candidates)
// Text form of the table (MEBT_GEMM_TUNE_CACHE files, mebt_gemm_tune_export / _import, the shipped mebt_amd/tune/*.txt): one entry
// per line, `n k_0 ... k_{n-1} value`.  The entry with the key {-1} carries MEBT_TUNE_VERSION: a text written by a build whose
// variant codes meant something else is ignored as a whole (an unversioned text is taken as version 1).
#define MEBT_TUNE_VERSION 4
static int tune_parse(const char* text, bool overwrite, bool replace = false) {
    std::vector<std::pair<TuneKey, int>> ent;
    int version = 1;
    const char* c = text;
    auto next_int = [&](int& v) -> bool {
        char* e = nullptr;
        const long x = strtol(c, &e, 10);
        if (e == c) return false;
        c = e; v = (int)x;
        return true;
    };
    int n;
    while (next_int(n) && n > 0 && n < 64) {
        TuneKey k(n);
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) ok = next_int(k[i]);
        int v;
        if (!ok || !next_int(v)) break;
        if (n == 1 && k[0] == -1) version = v;
        else ent.emplace_back(k, v);
    }
    if (version != MEBT_TUNE_VERSION) return 0;
    if (replace) g_tuned.clear();
    int taken = 0;
    for (auto& e : ent) {
        if (!overwrite && g_tuned.count(e.first)) continue;
        g_tuned[e.first] = e.second;
        ++taken;
    }
    return taken;
}
// version line of a table text (1 for an unversioned text), without taking any entry
static int tune_text_version(const char* text) {
    int version = 1;
    const char* c = text;
    while (*c) {
        char* e = nullptr;
        const long n = strtol(c, &e, 10);
        if (e == c || n <= 0 || n >= 64) break;
        c = e;
        long first = 0, v = 0;
        bool ok = true;
        for (long i = 0; i < n && ok; ++i) { const long x = strtol(c, &e, 10); ok = e != c; c = e; if (i == 0) first = x; }
        if (!ok) break;
        v = strtol(c, &e, 10);
        if (e == c) break;
        c = e;
        if (n == 1 && first == -1) version = (int)v;
    }
    return version;
}
static void tune_init() {
    if (g_autotune >= 0) return;
    const char* e = getenv("MEBT_GEMM_AUTOTUNE");
    g_autotune = (e && e[0] == '0') ? 0 : 1;
    const char* l = getenv("MEBT_GEMM_TUNE_LOG");
    g_tune_log = l ? atoi(l) : 0;                     // 1: decisions, 2: every candidate
    g_tune_cache = getenv("MEBT_GEMM_TUNE_CACHE");
    if (g_tune_cache && g_tune_cache[0]) {
        // Data-parallel ranks inherit the same variable and run this concurrently: the read / version check / restart of the file is
        // done under an exclusive flock on a sibling lock file, a file of another version is MOVED to <cache>.v<its version> (never
        // truncated: it may be a file the user pointed at by mistake) and the fresh file appears by rename.
        const std::string lock_path = std::string(g_tune_cache) + ".lock";
        const int lock_fd = open(lock_path.c_str(), O_CREAT | O_RDWR, 0644);
        if (lock_fd >= 0) (void)flock(lock_fd, LOCK_EX);
        bool fresh = true;
        if (FILE* f = fopen(g_tune_cache, "r")) {
            std::string text;
            char buf[4096];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
            fclose(f);
            fresh = text.empty();
            const int ver = fresh ? MEBT_TUNE_VERSION : tune_text_version(text.c_str());
            if (!fresh && ver != MEBT_TUNE_VERSION) {
                // a cache written by a build whose variant codes meant something else (or an unversioned round-3 file): nothing of it
                // is usable, and appending to it would only grow a file every later process ignores again — start a new one
                const std::string keep = std::string(g_tune_cache) + ".v" + std::to_string(ver);
                const bool moved = rename(g_tune_cache, keep.c_str()) == 0;
                fprintf(stderr, "[mebt gemm autotune] %s holds a table of version %d: %s, new table for version %d\n", g_tune_cache, ver,
                        moved ? ("kept as " + keep).c_str() : "could not be moved aside (left in place, not used)", MEBT_TUNE_VERSION);
                if (!moved) g_tune_cache = nullptr;       // never truncate a file that is not ours
                fresh = moved;
            } else {
                tune_parse(text.c_str(), true);
            }
        }
        if (fresh && g_tune_cache) {
            const std::string tmp = std::string(g_tune_cache) + ".tmp" + std::to_string((long)getpid());
            if (FILE* f = fopen(tmp.c_str(), "w")) {
                fprintf(f, "1 -1 %d\n", MEBT_TUNE_VERSION);
                fclose(f);
                if (rename(tmp.c_str(), g_tune_cache) != 0) (void)remove(tmp.c_str());
            }
        }
        if (lock_fd >= 0) { (void)flock(lock_fd, LOCK_UN); close(lock_fd); }
    } else {
        g_tune_cache = nullptr;
    }
}
static void tune_remember(const TuneKey& k, int v) {
    if (!g_tune_cache) return;
    if (FILE* f = fopen(g_tune_cache, "a")) {
        fprintf(f, "%d", (int)k.size());
        for (int x : k) fprintf(f, " %d", x);
        fprintf(f, " %d\n", v);
        fclose(f);
    }
}
// The table as text (see tune_parse): returns the bytes needed incl. the terminating 0; writes them when `cap` suffices.  A
// data-parallel job broadcasts rank 0's table after the first step so that every rank launches the same kernels (VERDICT r03:
// eight ranks tuning on their own can end with eight tables and a permanent straggler).
extern "C" int64_t mebt_gemm_tune_export(char* buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    tune_init();
    std::string text = "1 -1 " + std::to_string(MEBT_TUNE_VERSION) + "\n";
    for (auto& e : g_tuned) {
        text += std::to_string(e.first.size());
        for (int x : e.first) { text += ' '; text += std::to_string(x); }
        text += ' '; text += std::to_string(e.second); text += '\n';
    }
    const int64_t need = (int64_t)text.size() + 1;
    if (buf && cap >= need) memcpy(buf, text.c_str(), (size_t)need);
    return need;
}
// The runner-up candidates of the signatures THIS process tuned: one line per signature, `n k_0 .. k_{n-1} : v us v us ...` (fastest
// first, at most four; v as in the table).  Diagnostics for tools/step_tune.py; same size protocol as mebt_gemm_tune_export.
extern "C" int64_t mebt_gemm_tune_alternatives(char* buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    std::string text;
    char tmp[64];
    for (auto& e : g_alts) {
        text += std::to_string(e.first.size());
        for (int x : e.first) { text += ' '; text += std::to_string(x); }
        text += " :";
        for (auto& a : e.second) { snprintf(tmp, sizeof tmp, " %d %.2f", a.first, a.second * 1e3f); text += tmp; }
        text += '\n';
    }
    const int64_t need = (int64_t)text.size() + 1;
    if (buf && cap >= need) memcpy(buf, text.c_str(), (size_t)need);
    return need;
}
// Merge a text table: overwrite = 1 replaces entries this process already holds, 2 drops the whole table first (a data-parallel
// rank adopting rank 0's table: identical tables afterwards), 0 keeps them (a shipped default table never overrides what was
// tuned here or loaded from MEBT_GEMM_TUNE_CACHE).  Returns the number of entries taken
// (0 for a text of another MEBT_TUNE_VERSION), negative on a null argument.
extern "C" int32_t mebt_gemm_tune_import(const char* text, int32_t overwrite) {
    if (!text) return -1;
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    tune_init();
    return tune_parse(text, overwrite != 0, overwrite == 2);
}
// 0: never tune (heuristic / cached choices only; every entry point is then free of host synchronisation and
// capture-safe), 1: tune unseen signatures at their first launch.  Default: MEBT_GEMM_AUTOTUNE (1).
extern "C" int32_t mebt_gemm_autotune_enabled(void) { std::lock_guard<std::mutex> lk(g_tune_mutex); tune_init(); return g_autotune; }
extern "C" void mebt_gemm_autotune(int32_t mode) { std::lock_guard<std::mutex> lk(g_tune_mutex); tune_init(); g_autotune = mode ? 1 : 0; }

static void heuristic_config(const GemmParams& p, int& tbm, int& tbn, int& staging) {
    // cold-operand measurements (profiles/r01_gemm_variants_cold.txt), all three layouts alike:
    // many tiles -> 128x128 with 2 stages (2 workgroups/CU); a chip's worth or less -> deeper rings
    const long n128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (n128 >= 384) { tbm = 128; tbn = 128; staging = 2; }
    else if (n128 > 256) { tbm = 128; tbn = 64; staging = 2; }
    else if (n128 >= 160) { tbm = 128; tbn = 128; staging = 3; }
    else if (n128 > 64) { tbm = 64; tbn = 128; staging = p.K >= 2048 ? 4 : 3; }
    else { tbm = 64; tbn = 64; staging = 4; }
}

// Candidates are timed with COLD caches: in the train step the weights (676 MB in bf16) never survive in the
// 256 MB Infinity Cache from one use to the next, and a warm-cache timing favours shallow rings.  Before every
// timed launch a 384 MB scratch buffer is overwritten (L2 and Infinity Cache hold nothing of the operands).
struct TuneRun {        // one tuning session: the caller's flush buffer + two events of its own
    const GemmScratch* sc = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int begin(const GemmScratch* s) {
        sc = s;
        MEBT_HIP_CHECK(hipEventCreate(&e0));
        MEBT_HIP_CHECK(hipEventCreate(&e1));
        return MEBT_OK;
    }
    ~TuneRun() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
template <typename F>
static int time_cold(F&& launch, hipStream_t stream, const TuneRun& tr, float& best_ms) {
    float tot = 0.f;
    hipEvent_t e0 = tr.e0, e1 = tr.e1;
    launch();                                                                    // code object, TLBs
    static const int warm = [] { const char* e = getenv("MEBT_GEMM_TUNE_WARM"); return e ? atoi(e) : 0; }();   // experiment: time candidates on warm caches
    // How cold?  In the step a product's weights were prefetched into the Infinity Cache by its predecessor and its activations
    // were just written: L2-cold, Infinity-Cache-warm.  A 384 MB flush also empties the 256 MB Infinity Cache and ranks the
    // candidates for a situation that never occurs: fresh-tuned steps were 0.2-0.3 ms slower than with 96-256 MB flushes
    // (three boxes, profiles/r03_tuner_flush_size_ab.txt).  128 MB evicts the L2s (32 MB) several times over and leaves the
    // just-touched operands in the Infinity Cache.  MEBT_GEMM_TUNE_FLUSH_MB overrides.
    static const size_t flush_cap = [] { const char* e = getenv("MEBT_GEMM_TUNE_FLUSH_MB"); return (size_t)(e ? atol(e) : 128) << 20; }();
    for (int r = 0; r < 2; ++r) {
        if (!warm) MEBT_HIP_CHECK(hipMemsetAsync(tr.sc->flush, r, tr.sc->flush_bytes < flush_cap ? tr.sc->flush_bytes : flush_cap, stream));
        MEBT_HIP_CHECK(hipEventRecord(e0, stream));
        launch();
        MEBT_HIP_CHECK(hipEventRecord(e1, stream));
        MEBT_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        MEBT_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
    }
    best_ms = tot / 2;
    return MEBT_OK;
}

static int autotune_config(const GemmParams& p, hipStream_t stream, int& tbm, int& tbn, int& staging, std::vector<std::pair<int, float>>* alts = nullptr) {
    TuneRun tr;
    if (int rc = tr.begin(scratch_of(p.scratch))) return rc;
    static const int tiles[7][2] = {{192, 128}, {128, 128}, {96, 128}, {128, 64}, {64, 128}, {96, 64}, {64, 64}};
    const long out = (long)p.M * p.N;
    float best = 1e30f;
    struct Cand { float ms; int bm, bn, staging; };
    std::vector<Cand> cands;
    for (int t = 0; t < 7; ++t) {
        const int bm = tiles[t][0], bn = tiles[t][1];
        const long nt = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
        const bool few = nt < 96 && (long)bm * bn > 64 * 64 && out > 64 * 64;    // unsplit, this tile would leave most of the chip idle
        for (int st = 2; st <= 4 && !few; ++st) {
            if (st * (bm + bn) * BK * 2 > 128 * 1024) continue;
            float ms = 0.f;
            if (int rc = time_cold([&] { launch_bf16_config(p, bm, bn, st, 1, stream); }, stream, tr, ms)) return rc;
            if (g_tune_log >= 2) fprintf(stderr, "    cand %dx%d ring %d: %.1f us\n", bm, bn, st, ms * 1e3f);
            cands.push_back({ms, bm, bn, st});
            if (ms < best) { best = ms; tbm = bm; tbn = bn; staging = st; }
        }
        for (int st = 2; st <= 4 && !few; ++st) {                                 // software-pipelined main loop (staging 8 + ring depth)
            if (st * (bm + bn) * BK * 2 > 160 * 1024) continue;
            float ms = 0.f;
            if (int rc = time_cold([&] { launch_bf16_config(p, bm, bn, 8 + st, 1, stream); }, stream, tr, ms)) return rc;
            if (g_tune_log >= 2) fprintf(stderr, "    cand %dx%d ring %d pipelined: %.1f us\n", bm, bn, st, ms * 1e3f);
            cands.push_back({ms, bm, bn, 8 + st});
            if (ms < best) { best = ms; tbm = bm; tbn = bn; staging = 8 + st; }
        }
        if (!p.c_f32 && p.C && nt <= 256 && (long)bm * bn >= 128 * 128)          // split-K into fp32 slabs + reduce/epilogue kernel (staging 32 * log2(S) + ring)
            for (int S = 2; S <= 4; S *= 2) {
                if (p.K % (S * BK) || p.K / S < 8 * BK || !splitk_fits(p, S)) continue;
                for (int st = 2; st <= 3; ++st) {
                    if (st * (bm + bn) * BK * 2 > 128 * 1024) continue;
                    float ms = 0.f;
                    int rc2 = MEBT_OK;
                    if (int rc = time_cold([&] { rc2 |= launch_bf16_splitk(p, bm, bn, st, S, stream); }, stream, tr, ms)) return rc;
                    if (rc2) return rc2;
                    if (g_tune_log >= 2) fprintf(stderr, "    cand %dx%d ring %d split-K %d: %.1f us\n", bm, bn, st, S, ms * 1e3f);
                    cands.push_back({ms, bm, bn, (S == 2 ? 32 : 64) + st});
                    if (ms < best) { best = ms; tbm = bm; tbn = bn; staging = (S == 2 ? 32 : 64) + st; }
                }
            }
        if (p.K % (2 * BK) == 0 && p.K >= 8 * BK && nt <= 640 && !few)         // two pipelines per workgroup (staging 16 + ring depth)
            for (int st = 2; st <= 3; ++st) {
                if (!ks2_lds(bm, bn, st)) continue;
                float ms = 0.f;
                if (int rc = time_cold([&] { launch_bf16_config(p, bm, bn, 16 + st, 1, stream); }, stream, tr, ms)) return rc;
                if (g_tune_log >= 2) fprintf(stderr, "    cand %dx%d ring %d x2 pipelines: %.1f us\n", bm, bn, st, ms * 1e3f);
                cands.push_back({ms, bm, bn, 16 + st});
                if (ms < best) { best = ms; tbm = bm; tbn = bn; staging = 16 + st; }
            }
    }
    if ((long)((p.M + 255) / 256) * ((p.N + 255) / 256) >= 96 && (p.a_kc ? p.K % BK == 0 : true)) {   // 8-wave 256 x 256 tile
        float ms = 0.f;
        if (int rc = time_cold([&] { launch_bf16_config(p, 256, 256, 2, 1, stream); }, stream, tr, ms)) return rc;
        if (g_tune_log >= 2) fprintf(stderr, "    cand 256x256 (8 waves): %.1f us\n", ms * 1e3f);
        cands.push_back({ms, 256, 256, 2});
        if (ms < best) { best = ms; tbm = 256; tbn = 256; staging = 2; }
        if (p.a_kc && p.K % BK == 0) {                     // the same tile as two staggered wave groups (staging code 9)
            if (int rc = time_cold([&] { launch_bf16_config(p, 256, 256, 9, 1, stream); }, stream, tr, ms)) return rc;
            if (g_tune_log >= 2) fprintf(stderr, "    cand 256x256 (2 x 4 waves, staggered): %.1f us\n", ms * 1e3f);
            cands.push_back({ms, 256, 256, 9});
            if (ms < best) { best = ms; tbm = 256; tbn = 256; staging = 9; }
        }
    }
    // Second round.  The sweep's minimum over ~40 two-sample means is biased towards a lucky sample (run-to-run the choice moved
    // between neighbours and the step time with it, +-0.1 ms at config 2): the four fastest are timed again, three more cold pairs
    // each, and the best mean of all eight samples wins.
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.ms < b.ms; });
    const int finalists = (int)std::min<size_t>(4, cands.size());
    if (finalists > 1) {
        best = 1e30f;
        for (int c = 0; c < finalists; ++c) {
            Cand& k = cands[c];
            float sum = k.ms;
            for (int r = 0; r < 3; ++r) {
                float ms = 0.f;
                int rc2 = MEBT_OK;
                if (int rc = time_cold([&] {
                        if (k.staging >= 32 && k.bm != 256) rc2 |= launch_bf16_splitk(p, k.bm, k.bn, k.staging & 15, k.staging >= 64 ? 4 : 2, stream);
                        else launch_bf16_config(p, k.bm, k.bn, k.staging, 1, stream);
                    }, stream, tr, ms)) return rc;
                if (rc2) return rc2;
                sum += ms;
            }
            k.ms = sum / 4;
            if (g_tune_log >= 2) fprintf(stderr, "    finalist %dx%d code %d: %.1f us (mean of 8)\n", k.bm, k.bn, k.staging, k.ms * 1e3f);
            if (k.ms < best) { best = k.ms; tbm = k.bm; tbn = k.bn; staging = k.staging; }
        }
    }
    if (alts)
        for (int c = 0; c < finalists; ++c) alts->emplace_back((cands[c].bm << 20) | (cands[c].bn << 8) | cands[c].staging, cands[c].ms);
    if (g_tune_log)
        fprintf(stderr, "[mebt gemm autotune] M=%d N=%d K=%d a_kc=%d b_kc=%d epi=%d c_f32=%d -> %dx%d ring %d%s (%.1f us cold)\n", p.M, p.N, p.K,
                p.a_kc, p.b_kc, p.epilogue, p.c_f32, tbm, tbn, (tbm == 256 && staging == 9) ? 2 : (staging >= 8 && staging < 16) ? staging - 8 : (staging & 15),
                (tbm == 256 && staging == 9) ? " two staggered groups, persistent" : staging >= 64 ? " split-K 4" : staging >= 32 ? " split-K 2" : staging >= 16 ? " x2 pipelines" : staging >= 8 ? " pipelined" : "", best * 1e3f);
    return MEBT_OK;
}

int launch_gemm(const GemmParams& p_in, int dtype, hipStream_t stream) {
    GemmParams p = p_in;
    static const int narrow = [] { const char* e = getenv("MEBT_EPI_NARROW"); return (e && e[0] == '1') ? 1 : 0; }();
    p.narrow_store = narrow;
    if (p.M <= 0 || p.N <= 0) return MEBT_OK;
    drop_mark_small(p.drop, (uint64_t)p.M * p.ldc);        // the residual epilogue's dropout indexes row * ldc + column
    if (p.N % 4 != 0) { mebt_set_error("gemm: N must be a multiple of 4"); return MEBT_ESHAPE; }
    const int esz = dtype == MEBT_BF16 ? 2 : 4;
    const int kq = dtype == MEBT_BF16 ? BK : FBK;
    if ((p.a_kc || p.b_kc) && (p.K % kq) != 0) { mebt_set_error("gemm: K of a k-contiguous operand must be a multiple of the k-tile"); return MEBT_ESHAPE; }
    if (!p.a_kc && (size_t)p.K * p.lda * esz >= 0x7FFFFFFFull) { mebt_set_error("gemm: RC operand A exceeds 2 GiB"); return MEBT_ESHAPE; }
    if (!p.b_kc && (size_t)p.K * p.ldb * esz >= 0x7FFFFFFFull) { mebt_set_error("gemm: RC operand B exceeds 2 GiB"); return MEBT_ESHAPE; }
    if ((!p.a_kc && (p.M % 8)) || (!p.b_kc && (p.N % 8))) { mebt_set_error("gemm: row extent of an RC operand must be a multiple of 8"); return MEBT_ESHAPE; }
    if (dtype == MEBT_F32) p.c_f32 = 1;
    if (g_gemm_nostore && p.epilogue == EPI_NONE) p.C = nullptr;
    p.stamps = g_gemm_stamps;
    if (p.K <= 0) { mebt_set_error("gemm: K must be positive (empty reductions are handled by the caller)"); return MEBT_ESHAPE; }
    int split = 1;
    const bool can_split = p.c_f32 && p.epilogue == EPI_NONE && p.split_k > 1;   // atomics split-K only on request
    if (can_split) {
        const int nkt = (p.K + kq - 1) / kq;
        split = max(1, min(p.split_k, nkt));
    }
    if (g_gemm_force_split > 0 && p.c_f32 && p.epilogue == EPI_NONE) split = g_gemm_force_split;
    if (split > 1 && !p.beta) {
        // split-K accumulates with fp32 atomics into a zeroed C
        MEBT_HIP_CHECK(hipMemset2DAsync(p.C, (size_t)p.ldc * 4, 0, (size_t)p.N * 4, p.M, stream));
    }
    if (dtype == MEBT_BF16) {
        int tbm = 128, tbn = 128, staging = 2;
        const bool forced = g_gemm_force_tile || g_gemm_dma >= 0;
        // only launches that can be repeated are timed: no accumulation into C, no output aliasing the aux operand (an in-place
        // residual).  C = null (the inference form of the GELU product: only gelu(C) is stored) is repeatable — comparing it with a
        // null aux made every inference fc1 skip the tuner and run the heuristic tile (round 5: 427 -> 270 us at 32 768 rows)
        const bool idempotent = !p.beta && split == 1 && (p.C == nullptr || p.C != p.aux);
        bool have = false;
        if (!forced && split == 1 && (long)p.M * p.N >= 128 * 128) {
            std::lock_guard<std::mutex> lk(g_tune_mutex);
            tune_init();
            const TuneKey key{p.a_kc | (p.b_kc << 1) | (p.epilogue << 2) | (p.c_f32 << 5) | ((p.bias != nullptr) << 6) | ((p.drop.thresh != 0) << 7) |
                                  ((p.C == nullptr) << 8),        // bit 8: no primary output (inference GELU product: half the store bytes)
                              tune_bucket_m(p), p.N, tune_bucket(p.K)};
            auto it = g_tuned.find(key);
            if (it == g_tuned.end() && g_autotune && idempotent && tune_scratch_of(p.scratch)) {
                heuristic_config(p, tbm, tbn, staging);
                std::vector<std::pair<int, float>> alts;
                if (int rc = autotune_config(p, stream, tbm, tbn, staging, &alts)) return rc;
                alts_keep(key, alts);
                it = g_tuned.emplace(key, (tbm << 20) | (tbn << 8) | staging).first;
                tune_remember(key, it->second);
            }
            if (it != g_tuned.end()) {
                tbm = it->second >> 20; tbn = (it->second >> 8) & 0xFFF; staging = it->second & 255;
                have = true;
                if ((staging >= 16 && staging < 32 && p.K % (2 * BK)) || (tbm == 256 && p.a_kc && p.K % BK)) have = false;   // a bucket neighbour's variant that this K cannot run
            }
        }
        if (!have) {
            heuristic_config(p, tbm, tbn, staging);
            if (g_gemm_force_tile) { tbm = g_gemm_force_tile >> 12; tbn = g_gemm_force_tile & 0xFFF; }
            if (g_gemm_dma >= 0) staging = g_gemm_dma == 1 ? 3 : g_gemm_dma;       // forced: 0 reg, 2..5 LDS-DMA stages (1 = 3)
        }
        if (staging >= 32) {
            const int S = staging >= 64 ? 4 : 2;
            if (!p.c_f32 && p.C && split == 1 && p.K % (S * BK) == 0 && !p.beta && splitk_fits(p, S)) { if (int rc = launch_bf16_splitk(p, tbm, tbn, staging & 15, S, stream)) return rc; }
            else launch_bf16_config(p, tbm, tbn, staging & 15, split, stream);
        } else {
            launch_bf16_config(p, tbm, tbn, staging, split, stream);
        }
    } else if (dtype == MEBT_F32) {
        // the largest tile whose grid still fills the chip (256 CUs): 128 x 128, then the longer side halved, then 64 x 64
        // (MEBT_F32_TILE=128 keeps 128 x 128 everywhere: A/B of round 6)
        static const int f32_tile = [] { const char* e = getenv("MEBT_F32_TILE"); return e ? atoi(e) : 0; }();
        int tbm = 128, tbn = 128;
        auto wgs = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * split; };
        if (f32_tile != 128 && wgs(128, 128) < 256) {
            if (p.M >= p.N) { tbm = 64; tbn = 128; } else { tbm = 128; tbn = 64; }
            if (wgs(tbm, tbn) < 256) { tbm = 64; tbn = 64; }
        }
        dim3 grid((p.N + tbn - 1) / tbn, (p.M + tbm - 1) / tbm, split);
#define LAUNCH_F32_T(AK, BKC)                                                                                                 \
        do {                                                                                                                      \
            if (tbm == 128 && tbn == 128) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 128, 128>), grid, dim3(256), 0, stream, p);   \
            else if (tbm == 128) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 128, 64>), grid, dim3(256), 0, stream, p);          \
            else if (tbn == 128) hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 64, 128>), grid, dim3(256), 0, stream, p);          \
            else hipLaunchKernelGGL((gemm_f32_kernel<AK, BKC, 64, 64>), grid, dim3(256), 0, stream, p);                          \
        } while (0)
#define LAUNCH_F32(AK, BKC) LAUNCH_F32_T(AK, BKC)
        if (p.a_kc && p.b_kc) LAUNCH_F32(true, true);
        else if (p.a_kc && !p.b_kc) LAUNCH_F32(true, false);
        else if (!p.a_kc && !p.b_kc) LAUNCH_F32(false, false);
        else LAUNCH_F32(false, true);
#undef LAUNCH_F32_T
#undef LAUNCH_F32
    } else {
        mebt_set_error("gemm: unsupported dtype");
        return MEBT_EDTYPE;
    }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, int dtype, hipStream_t stream) {
    const bool ok = dtype == MEBT_BF16 && p0.a_kc && p1.a_kc && p0.b_kc == p1.b_kc && p0.M > 0 && p0.N > 0 && p1.M > 0 && p1.N > 0 &&
                    p0.K > 0 && p1.K > 0 && !p0.beta && !p1.beta && !p0.c_f32 && !p1.c_f32 && !g_gemm_force_tile && g_gemm_dma < 0 &&
                    p0.K % BK == 0 && p1.K % BK == 0 && p0.N % 8 == 0 && p1.N % 8 == 0 && !g_gemm_nostore;
    if (!ok) {                       // anything unusual: two ordinary launches (with their own validation)
        if (int rc = launch_gemm(p0, dtype, stream)) return rc;
        return launch_gemm(p1, dtype, stream);
    }
    GemmPair g;
    g.p[0] = p0; g.p[1] = p1;
    int tbm = 96, tbn = 128, staging = 3;
    int choice = -1;
    {
        std::unique_lock<std::mutex> lk(g_tune_mutex);
        tune_init();
        const TuneKey key{0x20000000 | p0.b_kc | (p0.epilogue << 2) | (p1.epilogue << 5) | ((p0.drop.thresh != 0) << 8),
                          tune_bucket_m(p0), p0.N, tune_bucket(p0.K), tune_bucket_m(p1), p1.N, tune_bucket(p1.K)};
        auto it = g_tuned.find(key);
        if (it == g_tuned.end() && g_autotune && tune_scratch_of(p0.scratch)) {
            TuneRun tr;
            if (int rc = tr.begin(scratch_of(p0.scratch))) return rc;
            lk.unlock();            // the separate-launch baseline below goes through launch_gemm, which takes the lock itself
            static const int tiles[7][2] = {{192, 128}, {128, 128}, {96, 128}, {128, 64}, {64, 128}, {96, 64}, {64, 64}};
            float best = 1e30f;
            std::vector<std::pair<int, float>> alts;
            for (int t = 0; t < 7; ++t)
                for (int st = 2; st <= 4; ++st) {
                    if (st * (tiles[t][0] + tiles[t][1]) * BK * 2 > 128 * 1024) continue;
                    float ms = 0.f;
                    if (int rc = time_cold([&] { launch_pair_config(g, tiles[t][0], tiles[t][1], st, stream); }, stream, tr, ms)) return rc;
                    alts.emplace_back((tiles[t][0] << 20) | (tiles[t][1] << 8) | st, ms);
                    if (ms < best) { best = ms; tbm = tiles[t][0]; tbn = tiles[t][1]; staging = st; }
                }
            // ... against the two products launched one after the other with their own tuned configurations
            float sep = 0.f;
            if (int rc = time_cold([&] { launch_gemm(p0, MEBT_BF16, stream); launch_gemm(p1, MEBT_BF16, stream); }, stream, tr, sep)) return rc;
            if (int rc = time_cold([&] { launch_gemm(p0, MEBT_BF16, stream); launch_gemm(p1, MEBT_BF16, stream); }, stream, tr, sep)) return rc;
            if (g_tune_log)
                fprintf(stderr, "[mebt gemm autotune] pair %dx%dx%d + %dx%dx%d b_kc=%d -> %dx%d ring %d (%.1f us cold; separate launches %.1f us)\n",
                        p0.M, p0.N, p0.K, p1.M, p1.N, p1.K, p0.b_kc, tbm, tbn, staging, best * 1e3f, sep * 1e3f);
            lk.lock();
            alts.emplace_back(0, sep);                 // value 0 = the two products launched separately
            alts_keep(key, alts);
            it = g_tuned.emplace(key, sep <= best ? 0 : ((tbm << 20) | (tbn << 8) | staging)).first;
            tune_remember(key, it->second);
        }
        if (it != g_tuned.end()) choice = it->second;
    }
    if (choice == 0) {           // the pair did not win on this shape
        if (int rc = launch_gemm(p0, dtype, stream)) return rc;
        return launch_gemm(p1, dtype, stream);
    }
    if (choice > 0) { tbm = choice >> 20; tbn = (choice >> 8) & 0xFFF; staging = choice & 255; }
    launch_pair_config(g, tbm, tbn, staging, stream);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_wgrad_grouped(GroupedWgrad& w, int dtype, hipStream_t stream) {
    if (dtype != MEBT_BF16) { mebt_set_error("grouped wgrad: bf16 only"); return MEBT_EDTYPE; }
    int n = 0;
    GroupedWgrad c;
    for (int i = 0; i < w.n; ++i) {          // drop empty products (their C was zero-filled by the caller)
        if (w.g[i].M <= 0 || w.g[i].N <= 0 || w.g[i].K <= 0) continue;
        c.g[n++] = w.g[i];
    }
    c.n = n;
    if (!n) return MEBT_OK;
    c.beta = w.beta; c.scratch = w.scratch; c.Cb = w.Cb; c.gW = w.gW;
    c.fused = w.fused; c.W = w.W; c.gW = w.gW; c.mW = w.mW; c.vW = w.vW; c.Wlp = w.Wlp; c.opt = w.opt;
    for (int i = 1; i < n; ++i)              // insertion sort, K descending: the canonical order of the tuner key (the launchers order the items themselves)
        for (int j = i; j > 0 && c.g[j].K > c.g[j - 1].K; --j) { const GroupedWgrad::Item t = c.g[j]; c.g[j] = c.g[j - 1]; c.g[j - 1] = t; }
    int tbm = 128, tbn = 128, stages = g_grouped_stages == 3 ? 3 : 2;
    {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        tune_init();
        TuneKey key{0x40000000 | n | (c.fused ? 0x100 : 0) | (c.beta ? 0x200 : 0) | (c.Cb ? 0x400 : 0)};
        for (int i = 0; i < n; ++i) { key.push_back(c.g[i].M); key.push_back(c.g[i].N); key.push_back(tune_bucket(c.g[i].K)); }
        auto it = g_tuned.find(key);
        if (it == g_tuned.end() && g_autotune && !c.beta && tune_scratch_of(w.scratch)) {
            TuneRun tr;
            if (int rc = tr.begin(scratch_of(w.scratch))) return rc;
            static const int tiles[5][2] = {{256, 128}, {128, 128}, {128, 64}, {64, 128}, {64, 64}};
            float best = 1e30f;
            // candidates are timed in the mode that will run; a fused launch is not idempotent, so its candidates
            // run with a zero learning rate, zero decay and beta1 = beta2 = 1 (p, m, v are rewritten unchanged)
            GroupedWgrad tc = c;
            for (int i = 0; i < n; ++i) tc.g[i].bias = nullptr;      // the bias row sums are atomic adds: not in the repeated candidate runs
            if (tc.fused) { tc.opt.lr = 0.f; tc.opt.weight_decay = 0.f; tc.opt.beta1 = 1.f; tc.opt.beta2 = 1.f; }
            std::vector<std::pair<int, float>> alts;
            for (int t = 0; t < 5; ++t)
                for (int st = 2; st <= 4; ++st) {
                    if (tiles[t][0] == 256 && st > 3) continue;            // 8-wave 256 x 128: ring 2 or 3 (4 x 48 KiB does not fit the 160 KiB LDS)
                    float ms = 0.f;
                    if (int rc = time_cold([&] { launch_grouped_config(tc, tiles[t][0], tiles[t][1], st, stream); }, stream, tr, ms)) return rc;
                    alts.emplace_back((tiles[t][0] << 20) | (tiles[t][1] << 8) | st, ms);
                    if (g_tune_log >= 2) fprintf(stderr, "    cand grouped %dx%d ring %d%s%s: %.1f us\n", tiles[t][0], tiles[t][1], st, tc.fused ? " +adamw" : "", tc.Cb ? " bf16-out" : "", ms * 1e3f);
                    if (ms < best) { best = ms; tbm = tiles[t][0]; tbn = tiles[t][1]; stages = st; }
                }
            if (g_tune_log) {
                fprintf(stderr, "[mebt gemm autotune] grouped wgrad");
                for (int i = 0; i < n; ++i) fprintf(stderr, " %dx%dx%d", c.g[i].M, c.g[i].N, c.g[i].K);
                fprintf(stderr, " -> %dx%d ring %d (%.1f us cold)\n", tbm, tbn, stages, best * 1e3f);
            }
            alts_keep(key, alts);
            it = g_tuned.emplace(key, (tbm << 20) | (tbn << 8) | stages).first;
            tune_remember(key, it->second);
        }
        if (it != g_tuned.end()) { tbm = it->second >> 20; tbn = (it->second >> 8) & 0xFFF; stages = it->second & 255; }
    }
    {   // experiments (tools/env_ab_cached.sh): MEBT_GROUPED_FORCE=256x128x2 = that tile / ring for every grouped launch
        static const int env_force = [] {
            const char* e = getenv("MEBT_GROUPED_FORCE");
            int a = 0, b = 0, r = 0;
            return (e && sscanf(e, "%dx%dx%d", &a, &b, &r) == 3 && a && b) ? ((a << 20) | (b << 8) | r) : 0;
        }();
        if (env_force && !g_grouped_force) { tbm = env_force >> 20; tbn = (env_force >> 8) & 0xFFF; stages = env_force & 255; }
    }
    if (g_grouped_force) { tbm = g_grouped_force >> 20; tbn = (g_grouped_force >> 8) & 0xFFF; stages = g_grouped_force & 255; }
    launch_grouped_config(c, tbm, tbn, stages, stream);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int gemm_init_attributes() {
    if (int rc = mebt_gemm_attrs_kk()) return rc;
    if (int rc = mebt_gemm_attrs_kr()) return rc;
    if (int rc = mebt_gemm_attrs_rr()) return rc;
    return mebt_gemm_attrs_rk();
}
