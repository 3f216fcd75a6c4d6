// Sampler kernels of the MeBT inference loops (reference mebt/transformer.py:826-910, :413-439 and
// mebt/mask_sampler.py:178-246).  The reference draws each token with a FULL descending sort over
// the 16384-entry vocabulary and keeps column 0 (transformer.py:839,877); that is arg-max of
// p_norm/q, q ~ Exp(1), so one pass over the row in LDS suffices.  All noise is an explicit input
// (the caller owns the RNG), which makes sampled ids bit-comparable with the oracle.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

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
// The same draw with the logarithm on v_log_f32: u = (n + 0.5) / 2^24; for u within 2^-6 of 1 (where log2's absolute error would be a
// large relative one) the series of -log(1 - d), d = 1 - u exact in fp32; elsewhere -log2(u) * ln 2 (relative error < 5e-6).  In-kernel
// noise has no reference to be bit-equal to; its CPU twin (float64 log of the same u) agrees to that accuracy, inside the 1e-4 the
// twin test allows at near-ties.
__device__ __forceinline__ float exp1_fast(uint64_t seed, uint64_t idx) {
    uint32_t h = pair_hash(seed, 0x5A3B1Eu, idx);
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    const uint32_t n = h >> 8;
    const float u = ((float)n + 0.5f) * (1.0f / 16777216.0f);
    const float d = ((float)(16777215u - n) + 0.5f) * (1.0f / 16777216.0f);      // 1 - u, exact
    const float series = d * (1.0f + d * (0.5f + d * (0.33333334f + d * 0.25f)));
    const float lg = -__builtin_amdgcn_logf(u) * 0.6931471805599453f;
    return d < 0.015625f ? series : lg;
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
    int n_nan = 0;
    for (int e = tid; e < V; e += 256) {
        float v = lg[e] / (p.temperature + 1e-8f);      // transformer.py:860 (division, not multiply by reciprocal)
        (void)inv_t;
        if (v != v) { v = -INFINITY; ++n_nan; }          // :866-868
        sv[e] = v;
    }
    // torch.topk ranks NaN above every number and the reference filters before it replaces NaNs (see sample_fast_kernel)
    int top_k = p.top_k;
    if (__syncthreads_or(n_nan) && top_k > 0 && top_k < V) top_k -= (int)blk_sum((float)n_nan, sh);
    __syncthreads();
    if (top_k > 0 && top_k < V) {                        // :863-864, :891-895 — keep everything >= k-th largest
        if (tid == 0) { sel_prefix = 0; sel_k = (unsigned)top_k; }
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

// ------------------------------------------------------------------------------------------------
// The same draw for V = 16384 without top-p (every shipped script), the row in REGISTERS: 512 threads, thread t owns elements
// t + 512 j (j = 0 .. 31), one pass over the logits instead of six to ten passes over an LDS copy:
//  * top-k threshold (the k-th largest value, ties kept, transformer.py:891-895) without a radix select over 16384 entries (its four
//    LDS-histogram passes serialised on a handful of bins — logits share their exponent bits — and were most of the 1.9 ms this
//    kernel averaged on [32768, 16384] inputs): the k-th largest of the 512 per-thread maxima is a lower bound L of the
//    threshold (at least k elements are >= L); the elements >= L are collected in LDS (a few dozen on ordinary rows) and the
//    exact k-th largest is picked among them by rank counting.  More than 1024 candidates (degenerate rows: many equal values)
//    or k > 512: the block falls back to the radix select, reading registers.
//  * exp / division / noise only where the filtered value is not -inf (wave-uniform skips: with top_k = 32 that is 32 of 16384).
//  * the probability map of debug=True is written straight to its rows of the [B, N, V] map (16-byte stores through an LDS transpose).
// ------------------------------------------------------------------------------------------------
constexpr int FAST_T = 512;                 // threads per row
constexpr int FAST_E = SV_MAX / FAST_T;     // elements per thread: 32
constexpr int FAST_CAP = 1024;              // candidate list
__device__ __forceinline__ float blk_sum8(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]));
}
__device__ __forceinline__ float blk_max8(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])), fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7])));
}

// LT: float, or bf16_t for the logits the head wrote in bf16 for the sampling loops of a bf16 model (half the bytes on both sides of
// the [rows, V] tensor; the values are the bf16 roundings of the same fp32 accumulators, converted back exactly on load)
template <typename LT>
__global__ __launch_bounds__(FAST_T) void sample_fast_kernel(const SampleParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB only when probabilities are written
    __shared__ float sh[8];
    __shared__ float lmax[8];
    __shared__ float cand[FAST_CAP];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel_prefix, sel_k, ncand;
    __shared__ float thr;
    constexpr int V = SV_MAX;
    const int tid = threadIdx.x, row = blockIdx.x;
    const LT* lg = reinterpret_cast<const LT*>(p.logits) + (size_t)row * V;
    const float tdiv = p.temperature + 1e-8f;
    // element ownership: thread t holds x[8 g + k] = element 8 t + k + 4096 g (g = 0 .. 3, k = 0 .. 7): four runs of eight consecutive
    // elements, read with 16-byte loads (one per run for bf16 logits, two for fp32) — 32 four-byte (two-byte) loads per thread before.
    // Within a thread the element index grows with the register index (the tie rules below rely on it).
    float x[FAST_E];
#pragma unroll
    for (int g8 = 0; g8 < FAST_E / 8; ++g8) {
        const LT* src = lg + 8 * tid + 4096 * g8;
        if constexpr (sizeof(LT) == 2) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[8 * g8 + k] = (float)v[k];
        } else {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[8 * g8 + k] = lo[k]; x[8 * g8 + 4 + k] = hi[k]; }
        }
    }
    auto elem_of = [&](int j) { return 8 * tid + (j & 7) + 4096 * (j >> 3); };
    if (tdiv != 1.0f) {                                   // x / 1.0f == x: the division is skipped only where it is the identity
#pragma unroll
        for (int j = 0; j < FAST_E; ++j) x[j] = x[j] / tdiv;      // transformer.py:860
    }
    float lm = -INFINITY;
    int n_nan = 0;
#pragma unroll
    for (int j = 0; j < FAST_E; ++j) {
        if (x[j] != x[j]) { x[j] = -INFINITY; ++n_nan; }  // :866-868 (applied AFTER the top-k filter in the reference: see top_k below)
        lm = fmaxf(lm, x[j]);
    }
    // torch.topk ranks NaN above every number (:891-895 runs before the NaN -> -inf replacement :866-868): a row with n NaNs keeps
    // everything >= its (k - n)-th largest NUMBER, and nothing is filtered when n >= k (the k-th value is NaN, `out < NaN` is false)
    int top_k = p.top_k;
    if (top_k > 0 && top_k < V && __syncthreads_or(n_nan)) top_k -= (int)blk_sum8((float)n_nan, sh);
    if (top_k > 0 && top_k < V) {                         // :863-864, :891-895 — keep everything >= the k-th largest
        bool radix = top_k > FAST_T;
        if (!radix) {
            // lower bound L of the threshold: every wave takes the m-th largest of its 64 per-thread maxima, m = ceil(k / 8) (m rounds
            // of wave maximum + retire one lane that holds it), L = the smallest of the eight: at least 8 m >= k elements are >= L
            const int mrounds = (top_k + 7) >> 3;
            float v = lm, wm = lm;
            for (int r = 0; r < mrounds; ++r) {
                wm = wave_max(v);
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(v == wm);
                if ((tid & 63) == (int)__builtin_ctzll(hit)) v = -INFINITY;       // one holder leaves (equal maxima leave one per round)
            }
            if ((tid & 63) == 0) lmax[tid >> 6] = wm;
            if (tid == 0) ncand = 0;
            __syncthreads();
            if (tid == 0) thr = fminf(fminf(fminf(lmax[0], lmax[1]), fminf(lmax[2], lmax[3])), fminf(fminf(lmax[4], lmax[5]), fminf(lmax[6], lmax[7])));
            __syncthreads();
            const float L = thr;
#pragma unroll
            for (int j = 0; j < FAST_E; ++j)
                if (x[j] >= L) {
                    const unsigned pos = atomicAdd(&ncand, 1u);
                    if (pos < (unsigned)FAST_CAP) cand[pos] = x[j];
                }
            __syncthreads();
            const int n = (int)ncand;
            radix = n > FAST_CAP;                         // block-uniform
            if (!radix) {
                for (int i = tid; i < n; i += FAST_T) {   // the k-th largest of the candidates: #greater < k <= #greater-or-equal
                    const float c = cand[i];
                    int gt = 0, ge = 0;
                    for (int s = 0; s < n; ++s) { const float o = cand[s]; gt += o > c ? 1 : 0; ge += o >= c ? 1 : 0; }
                    if (gt < top_k && top_k <= ge) thr = c;       // every thread that qualifies writes the same value
                }
                __syncthreads();
            }
        }
        if (radix) {                                      // 4 x 8-bit radix select over the monotone keys, from registers
            if (tid == 0) { sel_prefix = 0; sel_k = (unsigned)top_k; }
            unsigned mask = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const unsigned pre = sel_prefix;
#pragma unroll
                for (int j = 0; j < FAST_E; ++j) {
                    const uint32_t k = fkey(x[j]);
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
            if (tid == 0) { const uint32_t kth = sel_prefix; thr = __uint_as_float((kth & 0x80000000u) ? (kth & 0x7FFFFFFFu) : ~kth); }
            __syncthreads();
        }
        const float T = thr;
        if (p.kth && tid == 0) p.kth[row] = T;
        const uint32_t kT = fkey(T);
#pragma unroll
        for (int j = 0; j < FAST_E; ++j) if (fkey(x[j]) < kT) x[j] = -INFINITY;     // the comparison of sample_kernel (orders -0 below +0)
    }
    // softmax (:871): max, exp, sum; p = e * (1 / sum) (one rounding away from e / sum: inside every tolerance the probabilities are
    // compared with, and the arg-max below does not depend on the common factor)
    const float mx = blk_max8(lm, sh);                     // the filter never removes the row maximum
    float se = 0.f;
    if (p.icdf && !p.noise) {
        // production draw: exp(x - mx) as one fma + v_exp_f32 (relative error ~1e-6 at |x - mx| <= 20: the probabilities are this
        // path's own, nothing is compared bit for bit); the injected-noise path below keeps expf, whose values the parity tests pin
        const float mx2 = mx * 1.4426950408889634f;
#pragma unroll
        for (int j = 0; j < FAST_E; ++j) {
            const float e = __builtin_amdgcn_exp2f(fmaf(x[j], 1.4426950408889634f, -mx2));      // exp2(-inf) = 0
            x[j] = e;
            se += e;
        }
    } else {
#pragma unroll
        for (int j = 0; j < FAST_E; ++j) {
            float e = 0.f;
            if (__builtin_amdgcn_ballot_w64(x[j] > -INFINITY)) e = expf(x[j] - mx);    // expf(-inf - mx) = 0: skipped per wave where nothing survives
            x[j] = e;
            se += e;
        }
    }
    se = blk_sum8(se, sh);
    const float inv_se = 1.0f / se;
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < FAST_E; ++j) {
        x[j] *= inv_se;
        tot += x[j];
    }
    if (p.probs) {                                        // rows of 64 KiB: transpose through LDS, 16-byte stores
        float* sv = reinterpret_cast<float*>(smem);
        size_t drow = (size_t)row;
        bool in_map = true;
        if (p.probs_ti) {                                 // precondition: 0 <= ti < N (as torch's scatter_ checks); a row outside the map is
            const int b = row / p.probs_NT;              // skipped (block-uniform), never written out of bounds (scatter_ids_kernel does the same)
            const int64_t t = p.probs_ti[row];
            in_map = t >= 0 && t < (int64_t)p.probs_N;
            drow = (size_t)b * p.probs_N + (size_t)(in_map ? t : 0);
        }
        __syncthreads();
#pragma unroll
        for (int g8 = 0; g8 < FAST_E / 8; ++g8) {           // the thread's runs of eight go to LDS as two 16-byte stores each
            f32x4* d8 = reinterpret_cast<f32x4*>(sv + 8 * tid + 4096 * g8);
            d8[0] = f32x4{x[8 * g8], x[8 * g8 + 1], x[8 * g8 + 2], x[8 * g8 + 3]};
            d8[1] = f32x4{x[8 * g8 + 4], x[8 * g8 + 5], x[8 * g8 + 6], x[8 * g8 + 7]};
        }
        __syncthreads();
        f32x4* dst = reinterpret_cast<f32x4*>(p.probs + drow * V);
        if (in_map)
#pragma unroll
            for (int i = 0; i < SV_MAX / 4 / FAST_T; ++i) dst[tid + FAST_T * i] = *reinterpret_cast<const f32x4*>(sv + 4 * (tid + FAST_T * i));
    }
    const float loc = tot;                                // this thread's share of the row's probability mass
    tot = blk_sum8(tot, sh);
    if (p.icdf && !p.noise) {
        // Production draw (no injected noise).  The reference draws arg-max p / q with q ~ Exp(1) per element (:826-841), i.e. ONE
        // sample of the categorical distribution p; per element that costs a hash, a logarithm, a reciprocal and a compare — the
        // kernel was bound by exactly that arithmetic (983 us per [32768, 16384] draw against 350 us for reading the logits).  The
        // same distribution from one uniform per row: the first element, in the order thread 0's 32 elements (8 t + k + 4096 g, g major),
        // thread 1's, ..., whose running sum of p reaches u * sum(p).  Zero-probability (filtered) elements are never chosen; if rounding leaves the
        // target above the last running sum, the last element with p > 0 is taken.
        __shared__ float wtot[8];
        __shared__ int owner_lo, owner_last;
        uint32_t h = pair_hash(p.noise_seed, 0x5A3B1Fu, (uint64_t)row);
        h *= 0xC2B2AE35u;
        h ^= h >> 16;
        const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float target = u * tot;
        const int lane = tid & 63, wv = tid >> 6;
        float incl = loc;                                  // inclusive scan of the thread sums inside the wave (Hillis-Steele)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wtot[wv] = incl;
        if (tid == 0) { owner_lo = FAST_T; owner_last = -1; }
        __syncthreads();
        float base = incl - loc;
        for (int w = 0; w < wv; ++w) base += wtot[w];
        const bool has = loc > 0.f;
        if (has && base + loc >= target) atomicMin(&owner_lo, tid);
        if (has) atomicMax(&owner_last, tid);
        __syncthreads();
        const int owner = owner_lo < FAST_T ? owner_lo : owner_last;     // owner_last >= 0: the row maximum has p > 0
        if (tid == owner) {
            float acc = base;
            int pick = -1;
#pragma unroll
            for (int j = 0; j < FAST_E; ++j)
                if (x[j] > 0.f && (pick < 0 || acc < target)) { acc += x[j]; pick = j; }
            // `pick` = the first positive element at which the running sum reached the target, else the last positive one
            float pv = 0.f;
#pragma unroll
            for (int j = 0; j < FAST_E; ++j) if (j == pick) pv = x[j];
            p.ids[row] = elem_of(pick);
            if (p.score) p.score[row] = pv;
        }
        return;
    }
    // gumbel_sort (:834-841): arg-max of (p / sum p) / q, zero-probability entries forced to 0.  First sweep: a = p * rcp(q) (the common
    // factor 1 / sum p dropped, v_rcp_f32 instead of two IEEE divisions) keeping each thread's best entry with its p and q; the row
    // maximum A of a; then the exact key (p / tot) / q of every thread's best within 4e-6 of A decides (lowest index at equal keys) —
    // the reference's choice unless TWO of one thread's 32 entries lie within 4e-6 of the row maximum (such a pair is a tie by the
    // parity tests' own criterion, 5e-4).
    const float* nz = p.noise ? p.noise + (size_t)row * V : nullptr;
    float a1 = -1.f, p1 = 0.f, q1 = 1.f;
    int e1 = V;
#pragma unroll
    for (int j = 0; j < FAST_E; ++j) {
        const int e = elem_of(j);
        const float pe = x[j];
        float a = 0.f, q = 1.f;
        if (__builtin_amdgcn_ballot_w64(pe > 0.f)) {
            if (pe > 0.f) {
                q = nz ? nz[e] : exp1_fast(p.noise_seed, (uint64_t)row * V + e);
                a = pe * __builtin_amdgcn_rcpf(q);
            }
        }
        if (a > a1) { a1 = a; e1 = e; p1 = pe; q1 = q; }  // strict: the lower index stays at equal keys.  (As four selects instead of a
                                                          // branch hipcc overlaps all 32 hash / log chains: 148 VGPRs, one row per CU instead of two.)
    }
    const float A = blk_max8(a1, sh);
    float best = -1.f, bp = 0.f;
    int bi = V;
    if (a1 >= A * (1.0f - 4e-6f)) { best = p1 > 0.f ? (p1 / tot) / q1 : 0.f; bi = e1; bp = p1; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), op = __shfl_xor(bp, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; bp = op; }
    }
    __shared__ float wb[8], wp[8];
    __shared__ int wi[8];
    if ((tid & 63) == 0) { wb[tid >> 6] = best; wi[tid >> 6] = bi; wp[tid >> 6] = bp; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 8; ++w) if (wb[w] > best || (wb[w] == best && wi[w] < bi)) { best = wb[w]; bi = wi[w]; bp = wp[w]; }
        p.ids[row] = bi;
        if (p.score) p.score[row] = bp;
    }
}

__global__ void scatter_ids_kernel(int64_t* x, const int64_t* ti, const int64_t* ids, int B, int N, int NT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * NT) return;
    const int b = i / NT;
    const int64_t pos = ti[i];
    if (pos >= 0 && pos < N) x[(size_t)b * N + pos] = ids[i];
}

// one workgroup (1024 threads) per batch row: bitonic sort (descending) of (score/sum)/noise^ctemp over NT <= 16384 in LDS.
// Every thread works on a compare-exchange PAIR per iteration (pair q of stride j: i = insert a 0 bit at position log2 j, l = i | j),
// 16 waves hide the LDS latency of each other: 450 us -> tens of us per call at NT = 8192 (the 256-thread loop over elements,
// half of them idle, was latency-bound on its dependent LDS round trips; profiles/r04_c4_kernel_stats.csv).
constexpr int NM_THREADS = 1024;
__global__ __launch_bounds__(NM_THREADS) void next_mask_kernel(const NextMaskParams p, int n_pow2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* key = reinterpret_cast<float*>(smem);
    int* idx = reinterpret_cast<int*>(smem + (size_t)n_pow2 * 4);
    __shared__ float sh[NM_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, NT = p.NT;
    const float* sc = p.score + (size_t)b * NT;
    const float* nz = p.noise + (size_t)b * NT;
    // the row sum in the order of the 256-thread kernel this replaces (thread t of 256 adds elements t, t + 256, ...; wave sums;
    // the four wave totals in order): the keys — and with them the order at near-ties — stay bit-identical
    float s = 0.f;
    if (tid < 256)
        for (int j = tid; j < NT; j += 256) s += sc[j];
    s = wave_sum(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    s = sh[0] + sh[1] + sh[2] + sh[3];
    for (int j = tid; j < n_pow2; j += NM_THREADS) {
        float k = -INFINITY;
        if (j < NT) {
            const float q = p.ctemp == 0.f ? 1.0f : powf(nz[j], p.ctemp);   // mask_sampler.py:183
            k = (sc[j] / s) / q;                                            // :180,:183
        }
        key[j] = k; idx[j] = j < NT ? j : 0x7FFFFFFF;
    }
    __syncthreads();
    const int half = n_pow2 >> 1;
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < half; q += NM_THREADS) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), l = i | j;
                const bool desc = (i & k) == 0;
                const float a = key[i], c = key[l];
                const int ia = idx[i], ic = idx[l];
                const bool before = (a > c) || (a == c && ia < ic);
                if (desc ? !before : before) { key[i] = c; key[l] = a; idx[i] = ic; idx[l] = ia; }
            }
            __syncthreads();
        }
    const int64_t* ci = p.ci + (size_t)b * p.NC;
    const int64_t* ti = p.ti + (size_t)b * NT;
    int64_t* nc = p.new_ci + (size_t)b * (p.NC + p.n_new);
    int64_t* nt = p.new_ti + (size_t)b * (NT - p.n_new);
    for (int j = tid; j < p.NC; j += NM_THREADS) nc[j] = ci[j];                        // :228
    for (int j = tid; j < p.n_new; j += NM_THREADS) nc[p.NC + j] = ti[idx[j]];         // :232-233
    for (int j = p.n_new + tid; j < NT; j += NM_THREADS) nt[j - p.n_new] = ti[idx[j]]; // :231,:234
}

}  // namespace

int launch_sample(const SampleParams& p, hipStream_t stream) {
    if (p.rows <= 0) return MEBT_OK;
    if (p.V > SV_MAX || p.V <= 0) { mebt_set_error("sample: vocabulary must be in [1, 16384]"); return MEBT_ESHAPE; }
    const bool tp = p.top_p > 0.f;
    static const bool fast_on = [] { const char* e = getenv("MEBT_SAMPLE_FAST"); return !(e && e[0] == '0'); }();
    if (p.probs_ti && (tp || p.V != SV_MAX || !fast_on)) { mebt_set_error("sample: the scattered probability map needs V = 16384 without top-p"); return MEBT_ESHAPE; }
    if (p.icdf && (p.noise || tp || p.V != SV_MAX || !fast_on)) { mebt_set_error("sample: the inverse-CDF draw needs in-kernel noise, V = 16384 and no top-p (the register kernel)"); return MEBT_ESHAPE; }
    if (p.logits_bf16 && (tp || p.V != SV_MAX || !fast_on)) { mebt_set_error("sample: bf16 logits need V = 16384 without top-p (the register kernel)"); return MEBT_ESHAPE; }
    if (!tp && p.V == SV_MAX && fast_on) {
        const size_t fl = p.probs ? (size_t)SV_MAX * 4 : 0;
        if (p.logits_bf16) {
            if (fl) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_fast_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl));
            hipLaunchKernelGGL(sample_fast_kernel<bf16_t>, dim3(p.rows), dim3(FAST_T), fl, stream, p);
        } else {
            if (fl) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_fast_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl));
            hipLaunchKernelGGL(sample_fast_kernel<float>, dim3(p.rows), dim3(FAST_T), fl, stream, p);
        }
        MEBT_HIP_CHECK(hipGetLastError());
        return MEBT_OK;
    }
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
    hipLaunchKernelGGL(next_mask_kernel, dim3(p.B), dim3(NM_THREADS), lds, stream, p, n);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}
