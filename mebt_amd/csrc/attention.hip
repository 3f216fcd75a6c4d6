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

template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_fwd_generic(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float sK[KT][HD];
    __shared__ __attribute__((aligned(16))) float sV[KT][HD];
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + threadIdx.x;
    const bool valid = qi < p.NQ;
    const float scale = rsqrtf((float)HD);
    float q[HD], o[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) { q[e] = 0.f; o[e] = 0.f; }
    if (valid) load_row<T, HD>(reinterpret_cast<const T*>(p.q) + ((size_t)b * p.NQ + qi) * p.ldq + h * HD, q);
    float m = -INFINITY, l = 0.f;
    const T* kb = reinterpret_cast<const T*>(p.k) + (size_t)b * p.NK * p.ldk + h * HD;
    const T* vb = reinterpret_cast<const T*>(p.v) + (size_t)b * p.NK * p.ldv + h * HD;
    for (int k0 = 0; k0 < p.NK; k0 += KT) {
        const int n = min(KT, p.NK - k0);
        __syncthreads();
        stage_tile<T, HD>(sK, kb + (size_t)k0 * p.ldk, p.ldk, n, threadIdx.x, 64);
        stage_tile<T, HD>(sV, vb + (size_t)k0 * p.ldv, p.ldv, n, threadIdx.x, 64);
        __syncthreads();
        float s[KT];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) a += q[e] * sK[j][e];
            s[j] = j < n ? a * scale : -INFINITY;
            tmax = fmaxf(tmax, s[j]);
        }
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < HD; ++e) o[e] *= alpha;
        const uint64_t dbase = (((uint64_t)b * p.H + h) * p.NQ + qi) * p.NK + k0;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            float pj = __expf(s[j] - mn);   // exp(-inf) = 0 for padded keys
            l += pj;
            if (p.drop.thresh) pj *= drop_keep(p.drop, dbase + j);   // attn_drop acts on the normalised probabilities (gpt.py:135)
#pragma unroll
            for (int e = 0; e < HD; ++e) o[e] += pj * sV[j][e];
        }
        m = mn;
    }
    if (valid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;    // NK == 0: softmax over an empty axis -> output 0
        store_row<T, HD>(reinterpret_cast<T*>(p.o) + ((size_t)b * p.NQ + qi) * p.ldo + h * HD, o, inv);
        if (p.lse) p.lse[((size_t)b * p.H + h) * p.NQ + qi] = l > 0.f ? m + logf(l) : -INFINITY;
    }
}

// dQ (one query row per lane) + delta = rowsum(dO * O) written for the dK/dV kernel
template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_bwd_dq_generic(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float sK[KT][HD];
    __shared__ __attribute__((aligned(16))) float sV[KT][HD];
    const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x * 64 + threadIdx.x;
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
        p.delta[((size_t)b * p.H + h) * p.NQ + qi] = delta;
    }
    const T* kb = reinterpret_cast<const T*>(p.k) + (size_t)b * p.NK * p.ldk + h * HD;
    const T* vb = reinterpret_cast<const T*>(p.v) + (size_t)b * p.NK * p.ldv + h * HD;
    for (int k0 = 0; k0 < p.NK; k0 += KT) {
        const int n = min(KT, p.NK - k0);
        __syncthreads();
        stage_tile<T, HD>(sK, kb + (size_t)k0 * p.ldk, p.ldk, n, threadIdx.x, 64);
        stage_tile<T, HD>(sV, vb + (size_t)k0 * p.ldv, p.ldv, n, threadIdx.x, 64);
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
    if (valid) store_row<T, HD>(reinterpret_cast<T*>(p.dq) + ((size_t)b * p.NQ + qi) * p.lddq + h * HD, dq);
}

// dK, dV (one key row per lane); queries / dO / lse / delta staged per tile
template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_bwd_dkv_generic(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float sQ[KT][HD];
    __shared__ __attribute__((aligned(16))) float sG[KT][HD];
    __shared__ float sL[KT], sD[KT];
    const int b = blockIdx.z, h = blockIdx.y, ki = blockIdx.x * 64 + threadIdx.x;
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
    for (int q0 = 0; q0 < p.NQ; q0 += KT) {
        const int n = min(KT, p.NQ - q0);
        __syncthreads();
        stage_tile<T, HD>(sQ, qb + (size_t)q0 * p.ldq, p.ldq, n, threadIdx.x, 64);
        stage_tile<T, HD>(sG, gb + (size_t)q0 * p.lddo, p.lddo, n, threadIdx.x, 64);
        if (threadIdx.x < KT) {
            sL[threadIdx.x] = threadIdx.x < n ? lb[q0 + threadIdx.x] : 0.f;
            sD[threadIdx.x] = threadIdx.x < n ? db[q0 + threadIdx.x] : 0.f;
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
    if (valid) {
        store_row<T, HD>(reinterpret_cast<T*>(p.dk) + ((size_t)b * p.NK + ki) * p.lddk + h * HD, dk);
        store_row<T, HD>(reinterpret_cast<T*>(p.dv) + ((size_t)b * p.NK + ki) * p.lddv + h * HD, dv);
    }
}

template <typename T, int HD>
int run_fwd(const AttnParams& p, hipStream_t stream) {
    const dim3 grid((p.NQ + 63) / 64, p.H, p.B);
    hipLaunchKernelGGL((attn_fwd_generic<T, HD>), grid, dim3(64), 0, stream, p);
    return MEBT_OK;
}
template <typename T, int HD>
int run_bwd(const AttnParams& p, hipStream_t stream) {
    const dim3 gq((p.NQ + 63) / 64, p.H, p.B);
    hipLaunchKernelGGL((attn_bwd_dq_generic<T, HD>), gq, dim3(64), 0, stream, p);
    if (p.NK > 0) {
        const dim3 gk((p.NK + 63) / 64, p.H, p.B);
        hipLaunchKernelGGL((attn_bwd_dkv_generic<T, HD>), gk, dim3(64), 0, stream, p);
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
