// Sampler kernels of the MeBT inference loops (reference mebt/transformer.py:826-910, :413-439 and
// mebt/mask_sampler.py:178-246).  The reference draws each token with a FULL descending sort over
// the 16384-entry vocabulary and keeps column 0 (transformer.py:839,877); that is arg-max of
// p_norm/q, q ~ Exp(1), so one pass over the row in LDS suffices.  All noise is an explicit input
// (the caller owns the RNG), which makes sampled ids bit-comparable with the oracle.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SV_MAX = 16384;

__device__ __forceinline__ float blk_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float blk_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// Counter-based Exp(1) draw for element `idx` of the step keyed by `seed`: 24 uniform bits of a 3-round multiply / xor-shift
// hash -> u in (0, 1) -> -log(u).  Generated where it is consumed: the [rows, V] noise tensor of the explicit-noise entry
// (written by a torch RNG kernel, then re-read) doubled the sampler's HBM bytes.  CPU twin: oracle/closed_form.py:exp1_counter.
__device__ __forceinline__ float exp1_counter(uint64_t seed, uint64_t idx) {
    uint32_t h = pair_hash(seed, 0x5A3B1Eu, idx);
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return -logf(u);
}
__device__ __forceinline__ uint32_t fkey(float f) {   // monotone float -> uint map
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one workgroup (256 threads) per row; the row lives in LDS
template <bool TOP_P>
__global__ __launch_bounds__(256) void sample_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sv = reinterpret_cast<float*>(smem);                              // [V] values / probabilities
    unsigned short* si = reinterpret_cast<unsigned short*>(smem + (size_t)SV_MAX * 4);   // TOP_P: ids sorted by probability
    __shared__ float sh[4];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel_prefix, sel_k;
    __shared__ int sJ;
    __shared__ double chunk_sum[256];
    const int tid = threadIdx.x, V = p.V, row = blockIdx.x;
    const float* lg = p.logits + (size_t)row * V;
    const float inv_t = 1.0f / (p.temperature + 1e-8f);
    for (int e = tid; e < V; e += 256) {
        float v = lg[e] / (p.temperature + 1e-8f);      // transformer.py:860 (division, not multiply by reciprocal)
        (void)inv_t;
        if (v != v) v = -INFINITY;                       // :866-868
        sv[e] = v;
    }
    __syncthreads();
    if (p.top_k > 0 && p.top_k < V) {                    // :863-864, :891-895 — keep everything >= k-th largest
        if (tid == 0) { sel_prefix = 0; sel_k = (unsigned)p.top_k; }
        unsigned mask = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned pre = sel_prefix;
            for (int e = tid; e < V; e += 256) {
                const uint32_t k = fkey(sv[e]);
                if ((k & mask) == pre) atomicAdd(&hist[(k >> shift) & 255], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned need = sel_k, acc = 0;
                int b = 255;
                for (; b > 0; --b) { if (acc + hist[b] >= need) break; acc += hist[b]; }
                sel_prefix = pre | ((unsigned)b << shift);
                sel_k = need - acc;
            }
            mask |= 255u << shift;
            __syncthreads();
        }
        const uint32_t kth = sel_prefix;
        if (p.kth && tid == 0) p.kth[row] = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7FFFFFFFu) : ~kth);    // fkey^-1
        for (int e = tid; e < V; e += 256) if (fkey(sv[e]) < kth) sv[e] = -INFINITY;
        __syncthreads();
    }
    // softmax (:871)
    float mx = -INFINITY;
    for (int e = tid; e < V; e += 256) mx = fmaxf(mx, sv[e]);
    mx = blk_max(mx, sh);
    float se = 0.f;
    for (int e = tid; e < V; e += 256) { const float x = expf(sv[e] - mx); sv[e] = x; se += x; }
    se = blk_sum(se, sh);
    for (int e = tid; e < V; e += 256) sv[e] = sv[e] / se;
    __syncthreads();
    if (TOP_P) {                                         // :873-874, :898-910
        // bitonic sort (descending) of the ids by (probability, then lower id first).  Only entries with a non-zero
        // probability take part: after top-k filtering that is k (+ ties) of the 16384, and the zeros neither move the
        // cumulative sum nor need zeroing — same kept set as sorting the whole row (transformer.py:898-910).
        __shared__ unsigned int nz_count;
        if (tid == 0) nz_count = 0;
        __syncthreads();
        for (int e = tid; e < V; e += 256)
            if (sv[e] > 0.f) si[atomicAdd(&nz_count, 1u)] = (unsigned short)e;
        __syncthreads();
        const int nnz = (int)nz_count;
        int n = 1;
        while (n < nnz) n <<= 1;
        auto pr = [&](int i) -> float { const unsigned id = si[i]; return id < (unsigned)V ? sv[id] : -1.0f; };
        for (int e = nnz + tid; e < n; e += 256) si[e] = (unsigned short)0xFFFF;
        __syncthreads();
        for (int k = 2; k <= n; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const bool desc = (i & k) == 0;
                        const float a = pr(i), b = pr(l);
                        const bool before = (a > b) || (a == b && si[i] < si[l]);   // ties: lower id first
                        if (desc ? !before : before) { const unsigned short t = si[i]; si[i] = si[l]; si[l] = t; }
                    }
                }
                __syncthreads();
            }
        // J = first j > 0 whose exclusive prefix sum of the sorted probabilities reaches top_p (everything from J on is
        // dropped, :904-908).  Block scan: thread t owns the contiguous chunk [t c, (t+1) c) of the sorted order, chunk totals
        // are prefixed in LDS, then every thread walks its chunk.  Accumulated in double and compared as float, like
        // torch.cumsum on the CPU (acc_type<float> = double, result rounded to float); a one-thread loop over up to 16384
        // dependent LDS reads took ~1 ms per row.
        {
            const int c = (nnz + 255) / 256;
            const int j0 = min(tid * c, nnz), j1 = min(j0 + c, nnz);
            double t = 0.0;
            for (int j = j0; j < j1; ++j) t += (double)sv[si[j]];
            chunk_sum[tid] = t;
            if (tid == 0) sJ = nnz;
            __syncthreads();
            double acc = 0.0;
            for (int k = 0; k < tid; ++k) acc += chunk_sum[k];            // exclusive prefix of the chunk totals (256 LDS reads, broadcast)
            int J = nnz;
            for (int j = j0; j < j1; ++j) {
                if (j > 0 && (float)acc >= p.top_p) { J = j; break; }
                acc += (double)sv[si[j]];
            }
            if (J < nnz) atomicMin(&sJ, J);
        }
        __syncthreads();
        for (int j = sJ + tid; j < nnz; j += 256) sv[si[j]] = 0.f;
        __syncthreads();
        float s = 0.f;
        for (int e = tid; e < V; e += 256) s += sv[e];
        s = blk_sum(s, sh);
        for (int e = tid; e < V; e += 256) sv[e] = sv[e] / s;
        __syncthreads();
    }
    if (p.probs) for (int e = tid; e < V; e += 256) p.probs[(size_t)row * V + e] = sv[e];
    // gumbel_sort (:834-841): arg-max of (p / sum p) / q, zero-probability entries forced to 0
    float tot = 0.f;
    for (int e = tid; e < V; e += 256) tot += sv[e];
    tot = blk_sum(tot, sh);
    const float* nz = p.noise ? p.noise + (size_t)row * V : nullptr;
    float best = -1.f;
    int bi = V;
    for (int e = tid; e < V; e += 256) {
        const float pe = sv[e];
        const float q = nz ? nz[e] : exp1_counter(p.noise_seed, (uint64_t)row * V + e);
        const float key = pe > 0.f ? (pe / tot) / q : 0.f;
        if (key > best || (key == best && e < bi)) { best = key; bi = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __shared__ float wb[4];
    __shared__ int wi[4];
    if ((tid & 63) == 0) { wb[tid >> 6] = best; wi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (wb[w] > best || (wb[w] == best && wi[w] < bi)) { best = wb[w]; bi = wi[w]; }
        p.ids[row] = bi;
        if (p.score) p.score[row] = sv[bi];
    }
}

__global__ void scatter_ids_kernel(int64_t* x, const int64_t* ti, const int64_t* ids, int B, int N, int NT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * NT) return;
    const int b = i / NT;
    const int64_t pos = ti[i];
    if (pos >= 0 && pos < N) x[(size_t)b * N + pos] = ids[i];
}

// one workgroup per batch row: bitonic sort (descending) of (score/sum)/noise^ctemp over NT <= 8192
__global__ __launch_bounds__(256) void next_mask_kernel(const NextMaskParams p, int n_pow2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* key = reinterpret_cast<float*>(smem);
    int* idx = reinterpret_cast<int*>(smem + (size_t)n_pow2 * 4);
    __shared__ float sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, NT = p.NT;
    const float* sc = p.score + (size_t)b * NT;
    const float* nz = p.noise + (size_t)b * NT;
    float s = 0.f;
    for (int j = tid; j < NT; j += 256) s += sc[j];
    s = blk_sum(s, sh);
    for (int j = tid; j < n_pow2; j += 256) {
        float k = -INFINITY;
        if (j < NT) {
            const float q = p.ctemp == 0.f ? 1.0f : powf(nz[j], p.ctemp);   // mask_sampler.py:183
            k = (sc[j] / s) / q;                                            // :180,:183
        }
        key[j] = k; idx[j] = j < NT ? j : 0x7FFFFFFF;
    }
    __syncthreads();
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const float a = key[i], c = key[l];
                    const bool before = (a > c) || (a == c && idx[i] < idx[l]);
                    if (desc ? !before : before) {
                        key[i] = c; key[l] = a;
                        const int t = idx[i]; idx[i] = idx[l]; idx[l] = t;
                    }
                }
            }
            __syncthreads();
        }
    const int64_t* ci = p.ci + (size_t)b * p.NC;
    const int64_t* ti = p.ti + (size_t)b * NT;
    int64_t* nc = p.new_ci + (size_t)b * (p.NC + p.n_new);
    int64_t* nt = p.new_ti + (size_t)b * (NT - p.n_new);
    for (int j = tid; j < p.NC; j += 256) nc[j] = ci[j];                        // :228
    for (int j = tid; j < p.n_new; j += 256) nc[p.NC + j] = ti[idx[j]];         // :232-233
    for (int j = p.n_new + tid; j < NT; j += 256) nt[j - p.n_new] = ti[idx[j]]; // :231,:234
}

}  // namespace

int launch_sample(const SampleParams& p, hipStream_t stream) {
    if (p.rows <= 0) return MEBT_OK;
    if (p.V > SV_MAX || p.V <= 0) { mebt_set_error("sample: vocabulary must be in [1, 16384]"); return MEBT_ESHAPE; }
    const bool tp = p.top_p > 0.f;
    const size_t lds = tp ? (size_t)SV_MAX * 6 : (size_t)SV_MAX * 4;
    if (tp) {
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sample_kernel<true>, dim3(p.rows), dim3(256), lds, stream, p);
    } else {
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sample_kernel<false>, dim3(p.rows), dim3(256), lds, stream, p);
    }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_scatter_ids(int64_t* x, const int64_t* ti, const int64_t* ids, int B, int N, int NT, hipStream_t stream) {
    if (B * NT <= 0) return MEBT_OK;
    hipLaunchKernelGGL(scatter_ids_kernel, dim3((B * NT + 255) / 256), dim3(256), 0, stream, x, ti, ids, B, N, NT);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_next_mask(const NextMaskParams& p, hipStream_t stream) {
    if (p.B <= 0) return MEBT_OK;
    if (p.n_new < 0 || p.n_new > p.NT) { mebt_set_error("next_mask: n_new out of range"); return MEBT_ESHAPE; }
    if (p.NT > 16384) { mebt_set_error("next_mask: more than 16384 targets"); return MEBT_ESHAPE; }
    int n = 1;
    while (n < p.NT) n <<= 1;
    const size_t lds = (size_t)n * 8;
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&next_mask_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(next_mask_kernel, dim3(p.B), dim3(256), lds, stream, p, n);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}
