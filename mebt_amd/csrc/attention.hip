// Attention kernels, generic path: softmax(q k^T / sqrt(hd)) v with fp32 arithmetic for any
// element type T (fp32 parity mode; bf16 with head sizes the MFMA kernel does not cover).
// Replaces reference mebt/modules/gpt.py:131-137 and its autograd backward.  Flash-style: the
// [NQ,NK] score matrix is never written to HBM; K/V tiles are staged in LDS and read as
// wave-wide broadcasts, one query row (forward, dQ) or one key row (dK/dV) per lane.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT = 32;   // keys (or queries) per LDS tile

template <typename T, int HD>
__device__ __forceinline__ void load_row(const T* src, float (&dst)[HD], float scale = 1.f) {
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
        const f32x4 v = load4<T>(src + e);
        dst[e] = v[0] * scale; dst[e + 1] = v[1] * scale; dst[e + 2] = v[2] * scale; dst[e + 3] = v[3] * scale;
    }
}
template <typename T, int HD>
__device__ __forceinline__ void store_row(T* dst, const float (&src)[HD], float scale = 1.f) {
#pragma unroll
    for (int e = 0; e < HD; e += 4) store4<T>(dst + e, f32x4{src[e] * scale, src[e + 1] * scale, src[e + 2] * scale, src[e + 3] * scale});
}

// cooperative copy of `n` rows x HD of a strided [.., ld] tensor into an fp32 LDS tile [KT][HD]
template <typename T, int HD>
__device__ __forceinline__ void stage_tile(float (*dst)[HD], const T* src, int ld, int n, int tid, int nthreads) {
    constexpr int CPR = HD / 4;
    for (int c = tid; c < KT * CPR; c += nthreads) {
        const int r = c / CPR, e = (c % CPR) * 4;
        f32x4 v = {0, 0, 0, 0};
        if (r < n) v = load4<T>(src + (size_t)r * ld + e);
        *reinterpret_cast<f32x4*>(&dst[r][e]) = v;
    }
}

// NS waves per workgroup share the 64 rows a workgroup owns (one row per lane in EVERY wave) and split the streamed side: wave w
// takes the tiles w, w + NS, ... through a staging area of its own, and the NS partial results are merged through LDS at the end
// (round 6: with one wave per workgroup the fp32 engine's attention ran on a third of the SIMDs, one wave each - 256 x 96 / 64 = 384
// waves for 1024 SIMDs - and was 60 % of the exact-fp32 train step).  Every wave runs the same number of iterations (empty tiles at
// the end) so that the workgroup barriers match.
template <typename T, int HD, int NS>
__global__ __launch_bounds__(64 * NS) void attn_fwd_generic(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float (*sK)[HD] = reinterpret_cast<float (*)[HD]>(smem_raw) + (size_t)wave * 2 * KT;
    float (*sV)[HD] = sK + KT;
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + lane;
    const bool valid = qi < p.NQ;
    const float scale = rsqrtf((float)HD);
    float q[HD], o[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) { q[e] = 0.f; o[e] = 0.f; }
    if (valid) load_row<T, HD>(reinterpret_cast<const T*>(p.q) + ((size_t)b * p.NQ + qi) * p.ldq + h * HD, q);
    float m = -INFINITY, l = 0.f;
    const T* kb = reinterpret_cast<const T*>(p.k) + (size_t)b * p.NK * p.ldk + h * HD;
    const T* vb = reinterpret_cast<const T*>(p.v) + (size_t)b * p.NK * p.ldv + h * HD;
    const int ntile = (p.NK + KT - 1) / KT, iters = (ntile + NS - 1) / NS;
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it * NS + wave) * KT;
        const int n = max(0, min(KT, p.NK - k0));
        __syncthreads();
        if (n > 0) {
            stage_tile<T, HD>(sK, kb + (size_t)k0 * p.ldk, p.ldk, n, lane, 64);
            stage_tile<T, HD>(sV, vb + (size_t)k0 * p.ldv, p.ldv, n, lane, 64);
        }
        __syncthreads();
        if (n <= 0) continue;
        // groups of four keys: their scores, one rescale of (l, o) to the new running maximum, then the four P V updates.  (The whole
        // 32-key tile at once - 32 score registers beside q and o, both loops unrolled - spilled 6.8 KB per lane to scratch and made the
        // forward 8 x slower than the backward kernels: 2.07 ms per launch in the exact-fp32 Sky-16f step, round 6.)
        const uint64_t dbase = (((uint64_t)b * p.H + h) * p.NQ + qi) * p.NK + k0;
        for (int j0 = 0; j0 < n; j0 += 4) {
            float s4[4];
            float tmax = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) a += q[e] * sK[j0 + jj][e];        // rows n .. KT-1 of the tile are zero-filled
                s4[jj] = j0 + jj < n ? a * scale : -INFINITY;
                tmax = fmaxf(tmax, s4[jj]);
            }
            const float mn = fmaxf(m, tmax);                                     // finite: key j0 exists
            const float alpha = __expf(m - mn);                                  // m = -inf (first group): 0
            l *= alpha;
#pragma unroll
            for (int e = 0; e < HD; ++e) o[e] *= alpha;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float pj = __expf(s4[jj] - mn);   // exp(-inf) = 0 for padded keys
                l += pj;
                if (p.drop.thresh) pj *= drop_keep(p.drop, dbase + j0 + jj);   // attn_drop acts on the normalised probabilities (gpt.py:135)
#pragma unroll
                for (int e = 0; e < HD; ++e) o[e] += pj * sV[j0 + jj][e];
            }
            m = mn;
        }
    }
    if (NS > 1) {        // (m, l, o) of waves 1 .. NS-1 -> wave 0, through the staging area
        constexpr int RL = HD + 2;
        float* red = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
        if (wave > 0) {
            float* r = red + ((size_t)(wave - 1) * 64 + lane) * RL;
            r[0] = m; r[1] = l;
#pragma unroll
            for (int e = 0; e < HD; ++e) r[2 + e] = o[e];
        }
        __syncthreads();
        if (wave > 0) return;
        for (int w = 1; w < NS; ++w) {
            const float* r = red + ((size_t)(w - 1) * 64 + lane) * RL;
            const float mw = r[0], lw = r[1];
            if (!(mw > -INFINITY)) continue;             // that wave saw no key
            const float mn = fmaxf(m, mw);
            const float a = __expf(m - mn), bw = __expf(mw - mn);     // m = -inf: a = 0
            l = l * a + lw * bw;
#pragma unroll
            for (int e = 0; e < HD; ++e) o[e] = o[e] * a + r[2 + e] * bw;
            m = mn;
        }
    }
    if (valid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;    // NK == 0: softmax over an empty axis -> output 0
        store_row<T, HD>(reinterpret_cast<T*>(p.o) + ((size_t)b * p.NQ + qi) * p.ldo + h * HD, o, inv);
        if (p.lse) p.lse[((size_t)b * p.H + h) * p.NQ + qi] = l > 0.f ? m + logf(l) : -INFINITY;
    }
}

// dQ (one query row per lane) + delta = rowsum(dO * O) written for the dK/dV kernel
template <typename T, int HD, int NS>
__global__ __launch_bounds__(64 * NS) void attn_bwd_dq_generic(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float (*sK)[HD] = reinterpret_cast<float (*)[HD]>(smem_raw) + (size_t)wave * 2 * KT;
    float (*sV)[HD] = sK + KT;
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + lane;
    const bool valid = qi < p.NQ;
    const float scale = rsqrtf((float)HD);
    float q[HD], go[HD], dq[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) { q[e] = 0.f; go[e] = 0.f; dq[e] = 0.f; }
    float lse = 0.f, delta = 0.f;
    if (valid) {
        load_row<T, HD>(reinterpret_cast<const T*>(p.q) + ((size_t)b * p.NQ + qi) * p.ldq + h * HD, q);
        load_row<T, HD>(reinterpret_cast<const T*>(p.d_o) + ((size_t)b * p.NQ + qi) * p.lddo + h * HD, go);
        float oo[HD];
        load_row<T, HD>(reinterpret_cast<const T*>(p.o) + ((size_t)b * p.NQ + qi) * p.ldo + h * HD, oo);
#pragma unroll
        for (int e = 0; e < HD; ++e) delta += go[e] * oo[e];
        lse = p.lse[((size_t)b * p.H + h) * p.NQ + qi];
        if (wave == 0) p.delta[((size_t)b * p.H + h) * p.NQ + qi] = delta;
    }
    const T* kb = reinterpret_cast<const T*>(p.k) + (size_t)b * p.NK * p.ldk + h * HD;
    const T* vb = reinterpret_cast<const T*>(p.v) + (size_t)b * p.NK * p.ldv + h * HD;
    const int ntile = (p.NK + KT - 1) / KT, iters = (ntile + NS - 1) / NS;
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it * NS + wave) * KT;
        const int n = max(0, min(KT, p.NK - k0));
        __syncthreads();
        if (n > 0) {
            stage_tile<T, HD>(sK, kb + (size_t)k0 * p.ldk, p.ldk, n, lane, 64);
            stage_tile<T, HD>(sV, vb + (size_t)k0 * p.ldv, p.ldv, n, lane, 64);
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) { s += q[e] * sK[j][e]; dp += go[e] * sV[j][e]; }
            const float pj = __expf(s * scale - lse);
            if (p.drop.thresh) dp *= drop_keep(p.drop, (((uint64_t)b * p.H + h) * p.NQ + qi) * p.NK + k0 + j);
            const float ds = pj * (dp - delta) * scale;
#pragma unroll
            for (int e = 0; e < HD; ++e) dq[e] += ds * sK[j][e];
        }
    }
    if (NS > 1) {        // partial dQ of waves 1 .. NS-1 -> wave 0
        float* red = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
        if (wave > 0) {
            float* r = red + ((size_t)(wave - 1) * 64 + lane) * (HD + 1);
#pragma unroll
            for (int e = 0; e < HD; ++e) r[e] = dq[e];
        }
        __syncthreads();
        if (wave > 0) return;
        for (int w = 1; w < NS; ++w) {
            const float* r = red + ((size_t)(w - 1) * 64 + lane) * (HD + 1);
#pragma unroll
            for (int e = 0; e < HD; ++e) dq[e] += r[e];
        }
    }
    if (valid) store_row<T, HD>(reinterpret_cast<T*>(p.dq) + ((size_t)b * p.NQ + qi) * p.lddq + h * HD, dq);
}

// dK, dV (one key row per lane); queries / dO / lse / delta staged per tile
template <typename T, int HD, int NS>
__global__ __launch_bounds__(64 * NS) void attn_bwd_dkv_generic(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float (*sQ)[HD] = reinterpret_cast<float (*)[HD]>(smem_raw) + (size_t)wave * 2 * KT;
    float (*sG)[HD] = sQ + KT;
    float* sL = reinterpret_cast<float*>(smem_raw) + (size_t)NS * 2 * KT * HD + wave * 2 * KT;      // behind every wave's tiles
    float* sD = sL + KT;
    const int b = blockIdx.z, h = blockIdx.y, ki = blockIdx.x * 64 + lane;
    const bool valid = ki < p.NK;
    const float scale = rsqrtf((float)HD);
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) { k[e] = 0.f; v[e] = 0.f; dk[e] = 0.f; dv[e] = 0.f; }
    if (valid) {
        load_row<T, HD>(reinterpret_cast<const T*>(p.k) + ((size_t)b * p.NK + ki) * p.ldk + h * HD, k);
        load_row<T, HD>(reinterpret_cast<const T*>(p.v) + ((size_t)b * p.NK + ki) * p.ldv + h * HD, v);
    }
    const T* qb = reinterpret_cast<const T*>(p.q) + (size_t)b * p.NQ * p.ldq + h * HD;
    const T* gb = reinterpret_cast<const T*>(p.d_o) + (size_t)b * p.NQ * p.lddo + h * HD;
    const float* lb = p.lse + ((size_t)b * p.H + h) * p.NQ;
    const float* db = p.delta + ((size_t)b * p.H + h) * p.NQ;
    const int ntile = (p.NQ + KT - 1) / KT, iters = (ntile + NS - 1) / NS;
    for (int it = 0; it < iters; ++it) {
        const int q0 = (it * NS + wave) * KT;
        const int n = max(0, min(KT, p.NQ - q0));
        __syncthreads();
        if (n > 0) {
            stage_tile<T, HD>(sQ, qb + (size_t)q0 * p.ldq, p.ldq, n, lane, 64);
            stage_tile<T, HD>(sG, gb + (size_t)q0 * p.lddo, p.lddo, n, lane, 64);
            if (lane < KT) {
                sL[lane] = lane < n ? lb[q0 + lane] : 0.f;
                sD[lane] = lane < n ? db[q0 + lane] : 0.f;
            }
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) { s += sQ[i][e] * k[e]; dp += sG[i][e] * v[e]; }
            const float pi = __expf(s * scale - sL[i]);
            const float keep = p.drop.thresh ? drop_keep(p.drop, (((uint64_t)b * p.H + h) * p.NQ + q0 + i) * p.NK + ki) : 1.0f;
            const float ds = pi * (dp * keep - sD[i]) * scale;
            const float pd = pi * keep;
#pragma unroll
            for (int e = 0; e < HD; ++e) { dv[e] += pd * sG[i][e]; dk[e] += ds * sQ[i][e]; }
        }
    }
    if (NS > 1) {        // partial dK, then dV, of waves 1 .. NS-1 -> wave 0 (two rounds through the staging area)
        float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            if (wave > 0) {
                float* r = red + ((size_t)(wave - 1) * 64 + lane) * (HD + 1);
#pragma unroll
                for (int e = 0; e < HD; ++e) r[e] = which ? dv[e] : dk[e];
            }
            __syncthreads();
            if (wave == 0)
                for (int w = 1; w < NS; ++w) {
                    const float* r = red + ((size_t)(w - 1) * 64 + lane) * (HD + 1);
#pragma unroll
                    for (int e = 0; e < HD; ++e) { if (which) dv[e] += r[e]; else dk[e] += r[e]; }
                }
        }
        if (wave > 0) return;
    }
    if (valid) {
        store_row<T, HD>(reinterpret_cast<T*>(p.dk) + ((size_t)b * p.NK + ki) * p.lddk + h * HD, dk);
        store_row<T, HD>(reinterpret_cast<T*>(p.dv) + ((size_t)b * p.NK + ki) * p.lddv + h * HD, dv);
    }
}

// waves per workgroup: four where the streamed side has at least four tiles and the staging areas fit 64 KiB (head size <= 64)
template <int HD> constexpr int split_of() { return HD <= 64 ? 4 : 1; }
template <int HD, int NS> constexpr int attn_lds_bytes() { return NS * 2 * KT * HD * 4 + NS * 2 * KT * 4; }

// more than 64 KiB of dynamic LDS needs the attribute (once per instantiation)
template <typename K>
static bool allow_lds(K kernel, int bytes) {
    return bytes <= 64 * 1024 || hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

template <typename T, int HD>
int run_fwd(const AttnParams& p, hipStream_t stream) {
    const dim3 grid((p.NQ + 63) / 64, p.H, p.B);
    constexpr int NS = split_of<HD>();
    static const bool ok_lds = allow_lds(&attn_fwd_generic<T, HD, NS>, attn_lds_bytes<HD, NS>());
    if (!ok_lds) { mebt_set_error("attention: cannot reserve the LDS of the split generic kernel"); return MEBT_EHIP; }
    static const bool one = [] { const char* e = getenv("MEBT_ATTN_GENERIC_SPLIT"); return e && e[0] == '1'; }();     // A/B: 1 = one wave per workgroup
    if (NS > 1 && !one && p.NK >= NS * KT) hipLaunchKernelGGL((attn_fwd_generic<T, HD, NS>), grid, dim3(64 * NS), (attn_lds_bytes<HD, NS>()), stream, p);
    else hipLaunchKernelGGL((attn_fwd_generic<T, HD, 1>), grid, dim3(64), (attn_lds_bytes<HD, 1>()), stream, p);
    return MEBT_OK;
}
template <typename T, int HD>
int run_bwd(const AttnParams& p, hipStream_t stream) {
    constexpr int NS = split_of<HD>();
    static const bool one = [] { const char* e = getenv("MEBT_ATTN_GENERIC_SPLIT"); return e && e[0] == '1'; }();
    static const bool ok_lds = allow_lds(&attn_bwd_dq_generic<T, HD, NS>, attn_lds_bytes<HD, NS>()) && allow_lds(&attn_bwd_dkv_generic<T, HD, NS>, attn_lds_bytes<HD, NS>());
    if (!ok_lds) { mebt_set_error("attention: cannot reserve the LDS of the split generic kernels"); return MEBT_EHIP; }
    const dim3 gq((p.NQ + 63) / 64, p.H, p.B);
    if (NS > 1 && !one && p.NK >= NS * KT) hipLaunchKernelGGL((attn_bwd_dq_generic<T, HD, NS>), gq, dim3(64 * NS), (attn_lds_bytes<HD, NS>()), stream, p);
    else hipLaunchKernelGGL((attn_bwd_dq_generic<T, HD, 1>), gq, dim3(64), (attn_lds_bytes<HD, 1>()), stream, p);
    if (p.NK > 0) {
        const dim3 gk((p.NK + 63) / 64, p.H, p.B);
        if (NS > 1 && !one && p.NQ >= NS * KT) hipLaunchKernelGGL((attn_bwd_dkv_generic<T, HD, NS>), gk, dim3(64 * NS), (attn_lds_bytes<HD, NS>()), stream, p);
        else hipLaunchKernelGGL((attn_bwd_dkv_generic<T, HD, 1>), gk, dim3(64), (attn_lds_bytes<HD, 1>()), stream, p);
    }
    return MEBT_OK;
}

}  // namespace

#ifdef MEBT_HAVE_ATTN_MFMA
int launch_attn_fwd_mfma(const AttnParams& p, hipStream_t stream);   // attention_mfma.hip
int launch_attn_bwd_mfma(const AttnParams& p, hipStream_t stream);
#else
static int launch_attn_fwd_mfma(const AttnParams&, hipStream_t) { return -1; }
static int launch_attn_bwd_mfma(const AttnParams&, hipStream_t) { return -1; }
#endif
#ifdef MEBT_HAVE_ATTN_MFMA
static int g_force_generic = 0;
#else
static int g_force_generic = 1;
#endif
#ifdef MEBT_HAVE_ATTN_MFMA
void mebt_attn_force_generic(int on) { g_force_generic = on; }
#else
void mebt_attn_force_generic(int) {}
#endif

#define DISPATCH_HD(FN, T)                                             \
    switch (p.HD) {                                                    \
        case 32: FN<T, 32>(p, stream); break;                          \
        case 64: FN<T, 64>(p, stream); break;                          \
        case 128: FN<T, 128>(p, stream); break;                        \
        default: mebt_set_error("attention: head size must be 32, 64 or 128"); return MEBT_ESHAPE; \
    }

// does the forward of this (dtype, head size) run on the MFMA kernels, i.e. can it gather its keys / values (AttnParams::kidx)?
bool attn_fwd_can_gather(int dtype, int HD) { return dtype == MEBT_BF16 && HD == 64 && !g_force_generic; }

int launch_attn_fwd(const AttnParams& p, int dtype, hipStream_t stream) {
    if (p.B <= 0 || p.NQ <= 0) return MEBT_OK;
    if (dtype == MEBT_BF16 && p.HD == 64 && !g_force_generic && p.NK > 0) return launch_attn_fwd_mfma(p, stream);
    if (p.kidx && p.NK > 0) { mebt_set_error("attention: gathered keys / values (kidx) need the MFMA forward (bf16, head size 64)"); return MEBT_ESHAPE; }
    if (dtype == MEBT_BF16) { DISPATCH_HD(run_fwd, bf16_t) } else { DISPATCH_HD(run_fwd, float) }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_attn_bwd(const AttnParams& p, int dtype, hipStream_t stream) {
    if (p.B <= 0 || p.NQ <= 0) return MEBT_OK;
    if (dtype == MEBT_BF16 && p.HD == 64 && !g_force_generic && p.NK > 0) return launch_attn_bwd_mfma(p, stream);
    if (dtype == MEBT_BF16) { DISPATCH_HD(run_bwd, bf16_t) } else { DISPATCH_HD(run_bwd, float) }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}
