// HBM-bound kernels of the MeBT path: embedding gather / scatter-add, LayerNorm fwd/bwd, column
// sums, masked-token cross-entropy (+top-1/top-5), fused AdamW.  One wave64 per row for the
// row-wise kernels, 16-byte accesses per lane, wave-shuffle reductions (no LDS round trip for the
// per-row statistics).
#include "common.h"
#include <cstdlib>
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// embedding gather:  contexts = tok_emb[x[ci]] + pos_emb[ci];  targets = mask_emb + pos_emb[ti];
// sos = sos_emb broadcast.   (reference mebt/transformer.py:255-277 / :298-317; the reference
// materialises a [B,block,d] copy of pos_emb and two int64 [B,*,d] address tensors, we move only
// the algorithmic bytes.)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbedParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long per_b = (long)p.NS + p.NC + p.NT;
    if (row >= per_b * p.B) return;
    const int b = (int)(row / per_b);
    const int r = (int)(row % per_b);
    const float* src0;
    const float* src1 = nullptr;
    T* dst;
    DropCfg dc = p.drop;
    uint64_t ebase;
    if (r < p.NS) {
        src0 = p.sos_emb + (size_t)r * p.d;
        dst = reinterpret_cast<T*>(p.sos) + ((size_t)b * p.NS + r) * p.d;
        dc.site = SITE_EMB_SOS; ebase = ((uint64_t)b * p.NS + r) * p.d;
    } else if (r < p.NS + p.NC) {
        const int i = r - p.NS;
        long pos = p.ci[(size_t)b * p.NC + i];
        pos = pos < 0 ? 0 : (pos >= p.N ? p.N - 1 : pos);
        long tok = p.x_ids[(size_t)b * p.N + pos];
        tok = tok < 0 ? 0 : (tok >= p.vocab ? p.vocab - 1 : tok);
        src0 = p.tok_emb + (size_t)tok * p.d;
        src1 = p.pos_emb + (size_t)pos * p.d;
        dst = reinterpret_cast<T*>(p.ctx) + ((size_t)b * p.NC + i) * p.d;
        dc.site = SITE_EMB_CTX; ebase = ((uint64_t)b * p.NC + i) * p.d;
    } else {
        const int j = r - p.NS - p.NC;
        long pos = p.ti[(size_t)b * p.NT + j];
        pos = pos < 0 ? 0 : (pos >= p.N ? p.N - 1 : pos);
        src0 = p.mask_emb;
        src1 = p.pos_emb + (size_t)pos * p.d;
        dst = reinterpret_cast<T*>(p.tgt) + ((size_t)b * p.NT + j) * p.d;
        dc.site = SITE_EMB_TGT; ebase = ((uint64_t)b * p.NT + j) * p.d;
    }
    for (int e = lane * 4; e < p.d; e += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(src0 + e);
        if (src1) v += *reinterpret_cast<const f32x4*>(src1 + e);
        if (dc.thresh) {
            v *= drop_keep4(dc, ebase + e);
        }
        store4<T>(dst + e, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const EmbedBwdParams p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long per_b = (long)p.NC + p.NT;
    if (row >= per_b * p.B) return;
    const int b = (int)(row / per_b);
    const int r = (int)(row % per_b);
    // float atomics run at full rate only when a wave-instruction covers contiguous bytes: lane l adds
    // element l + 64k (256 contiguous bytes per instruction), not 4 consecutive elements per lane
    if (r < p.NC) {
        const long pos = p.ci[(size_t)b * p.NC + r];
        const long tok = p.x_ids[(size_t)b * p.N + pos];
        const float* g = p.g_ctx + ((size_t)b * p.NC + r) * p.d;
        float* gt = p.g_tok_emb + (size_t)tok * p.d;
        float* gp = p.g_pos_emb + (size_t)pos * p.d;
        for (int e = lane; e < p.d; e += 64) {
            const float v = g[e];
            atomicAdd(gt + e, v);
            atomicAdd(gp + e, v);
        }
    } else {
        const int j0 = r - p.NC;
        const long pos = p.ti[(size_t)b * p.NT + j0];
        const T* g = reinterpret_cast<const T*>(p.g_tgt) + ((size_t)b * p.NT + j0) * p.d;
        float* gp = p.g_pos_emb + (size_t)pos * p.d;
        for (int e = lane; e < p.d; e += 64) atomicAdd(gp + e, to_f32<T>(g[e]));
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXC = 8;   // d <= 2048

__device__ __forceinline__ long map_row(long r, int seg, int seg_stride, int seg_off) {
    return seg > 0 ? (r / seg) * (long)seg_stride + seg_off + (r % seg) : r;
}

struct LnFwdMulti { LnFwdParams j[MEBT_LN_MAXJ]; int blk_start[MEBT_LN_MAXJ + 1]; int n; };
struct LnBwdMulti { LnBwdParams j[MEBT_LN_MAXJ]; int blk_start[MEBT_LN_MAXJ + 1]; int rpw[MEBT_LN_MAXJ]; int n; };

// one wave per row, 4 rows per workgroup; the workgroups of all jobs are enumerated by blockIdx.x
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdMulti mj) {
    int job = 0;
#pragma unroll
    for (int i = 1; i < MEBT_LN_MAXJ; ++i)
        if (i < mj.n && (int)blockIdx.x >= mj.blk_start[i]) job = i;
    const LnFwdParams& p = mj.j[job];
    const int lane = threadIdx.x & 63;
    const long row = (long)(blockIdx.x - mj.blk_start[job]) * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const T* x = reinterpret_cast<const T*>(p.x) + (size_t)row * p.d;
    f32x4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int e = lane * 4 + 256 * c;
        if (e < p.d) {
            v[c] = load4<T>(x + e);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(s) / p.d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int e = lane * 4 + 256 * c;
        if (e < p.d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = v[c][j] - mean; q += t * t; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / p.d + 1e-5f);
    const long orow = map_row(row, p.seg, p.seg_stride, p.seg_off);
    T* y = reinterpret_cast<T*>(p.y) + (size_t)orow * p.d;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int e = lane * 4 + 256 * c;
        if (e < p.d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + e);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.beta + e);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            store4<T>(y + e, o);
        }
    }
    if (p.mean && lane == 0) { p.mean[orow] = mean; p.rstd[orow] = rstd; }
}

// LayerNorm backward, two kernels (each takes several jobs per launch):
//  (1) dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) [+ dx_add] (+ optional dropout'd copy dx2)
//      — one wave per row, 4 rows per workgroup;
//  (2) dgamma += sum_rows dy*xhat, dbeta += sum_rows dy — column reduction (64 columns x a chunk of
//      rows per workgroup, LDS reduce, ONE atomic per column per workgroup: float atomics collapse
//      when hundreds of workgroups meet on the same 2*d addresses — a fused variant with one adder per
//      16 rows ran 4x slower than both kernels together).
template <typename T>
__device__ __forceinline__ void ln_bwd_dx_body(const LnBwdMulti& mj, int bid) {
    int job = 0;
#pragma unroll
    for (int i = 1; i < MEBT_LN_MAXJ; ++i)
        if (i < mj.n && bid >= mj.blk_start[i]) job = i;
    const LnBwdParams& p = mj.j[job];
    const int lane = threadIdx.x & 63;
    const int d = p.d;
    const long row = (long)(bid - mj.blk_start[job]) * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const long mrow = map_row(row, p.seg, p.seg_stride, p.seg_off);
    const T* x = reinterpret_cast<const T*>(p.x) + (size_t)row * d;
    const T* dy = reinterpret_cast<const T*>(p.dy) + (size_t)mrow * d;
    const T* dy2 = p.dy2 ? reinterpret_cast<const T*>(p.dy2) + (size_t)row * d : nullptr;
    const float mean = p.mean[mrow], rstd = p.rstd[mrow];
    f32x4 xh[LN_MAXC], gy[LN_MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int e = lane * 4 + 256 * c;
        if (e < d) {
            const f32x4 xv = load4<T>(x + e);
            f32x4 dv = load4<T>(dy + e);
            if (dy2) dv += load4<T>(dy2 + e);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[c][j] = (xv[j] - mean) * rstd;
                gy[c][j] = dv[j] * g[j];
                s1 += gy[c][j];
                s2 += gy[c][j] * xh[c][j];
            }
        }
    }
    const float m1 = wave_sum(s1) / d, m2 = wave_sum(s2) / d;
    const bool f32out = sizeof(T) == 4 || p.dx_f32;           // job-uniform: fp32 gradient accumulators (contexts) or T
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int e = lane * 4 + 256 * c;
        if (e < d) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (gy[c][j] - m1 - xh[c][j] * m2);
            if (p.dx_add) o += load4<T>(reinterpret_cast<const T*>(p.dx_add) + (size_t)row * d + e);
            const size_t oi = (size_t)row * d + e;
            if (f32out) {
                float* dx = reinterpret_cast<float*>(p.dx) + oi;
                if (p.dx_accumulate) o += load4<float>(dx);
                store4<float>(dx, o);
            } else {
                T* dx = reinterpret_cast<T*>(p.dx) + oi;
                if (p.dx_accumulate) o += load4<T>(dx);
                store4<T>(dx, o);
            }
            if (p.dx2) {
                f32x4 m = o;                        // mask the value as stored (rounded to the type of dx)
                if (!f32out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = (float)(T)o[j];
                }
                m *= drop_keep4(p.drop2, (uint64_t)oi);
                if (f32out) store4<float>(reinterpret_cast<float*>(p.dx2) + oi, m);
                else store4<T>(reinterpret_cast<T*>(p.dx2) + oi, m);
            }
        }
    }
}

// ---- fast paths: d == NCH * 64 * V with V = 16 bytes of T per lane -------------------------------------
// The generic kernels above guard every 256-element chunk with `e < d` (runtime d): hipcc then branches
// around each load and waits for it (s_waitcnt vmcnt(0) per chunk), i.e. the row is fetched as a chain of
// dependent round trips (1.5 TB/s measured).  With the chunk count a template parameter every load of
// the row is issued before the first use.
template <typename T> struct LaneVec;
template <> struct LaneVec<float> { static constexpr int V = 4; };
template <> struct LaneVec<bf16_t> { static constexpr int V = 8; };
template <typename T, int V> __device__ __forceinline__ void loadv(const T* p, float (&o)[V]);
template <> __device__ __forceinline__ void loadv<float, 4>(const float* p, float (&o)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j];
}
template <> __device__ __forceinline__ void loadv<bf16_t, 8>(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
}
template <typename T, int V> __device__ __forceinline__ void storev(T* p, const float (&o)[V]);
template <> __device__ __forceinline__ void storev<float, 4>(float* p, const float (&o)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
}
template <> __device__ __forceinline__ void storev<bf16_t, 8>(bf16_t* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)o[j];
    *reinterpret_cast<bf16x8*>(p) = v;
}
template <int V> __device__ __forceinline__ void loadf(const float* p, float (&o)[V]) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * q + j] = v[j];
    }
}
template <int V> __device__ __forceinline__ void storef(float* p, const float (&o)[V]) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_fast_kernel(const LnFwdMulti mj) {
    constexpr int V = LaneVec<T>::V, D = NCH * 64 * V;
    int job = 0;
#pragma unroll
    for (int i = 1; i < MEBT_LN_MAXJ; ++i)
        if (i < mj.n && (int)blockIdx.x >= mj.blk_start[i]) job = i;
    const LnFwdParams& p = mj.j[job];
    const int lane = threadIdx.x & 63;
    const long row = (long)(blockIdx.x - mj.blk_start[job]) * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const T* x = reinterpret_cast<const T*>(p.x) + (size_t)row * D + lane * V;
    float v[NCH][V], g[NCH][V], b[NCH][V];
#pragma unroll
    for (int c = 0; c < NCH; ++c) loadv<T, V>(x + c * 64 * V, v[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) { loadf<V>(p.gamma + c * 64 * V + lane * V, g[c]); loadf<V>(p.beta + c * 64 * V + lane * V, b[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) s += v[c][j];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) { const float t = v[c][j] - mean; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
    const long orow = map_row(row, p.seg, p.seg_stride, p.seg_off);
    T* y = reinterpret_cast<T*>(p.y) + (size_t)orow * D + lane * V;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = (v[c][j] - mean) * rstd * g[c][j] + b[c][j];
        storev<T, V>(y + c * 64 * V, o);
    }
    if (p.mean && lane == 0) { p.mean[orow] = mean; p.rstd[orow] = rstd; }
}

template <typename T, int NCH>
__device__ __forceinline__ void ln_bwd_dx_fast_body(const LnBwdMulti& mj, int bid) {
    constexpr int V = LaneVec<T>::V, D = NCH * 64 * V;
    int job = 0;
#pragma unroll
    for (int i = 1; i < MEBT_LN_MAXJ; ++i)
        if (i < mj.n && bid >= mj.blk_start[i]) job = i;
    const LnBwdParams& p = mj.j[job];
    const int lane = threadIdx.x & 63;
    const long row = (long)(bid - mj.blk_start[job]) * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const long mrow = map_row(row, p.seg, p.seg_stride, p.seg_off);
    const size_t ro = (size_t)row * D + lane * V, mo = (size_t)mrow * D + lane * V;
    const bool f32out = sizeof(T) == 4 || p.dx_f32;
    // optional operands are loaded unconditionally (from the x row, scaled by 0, when absent): a branch
    // around a load makes hipcc drain the load queue at the join
    const T* dy2p = p.dy2 ? reinterpret_cast<const T*>(p.dy2) + ro : reinterpret_cast<const T*>(p.x) + ro;
    const T* addp = p.dx_add ? reinterpret_cast<const T*>(p.dx_add) + ro : reinterpret_cast<const T*>(p.x) + ro;
    const float k2 = p.dy2 ? 1.f : 0.f, ka = p.dx_add ? 1.f : 0.f;
    float xv[NCH][V], dv[NCH][V], g[NCH][V], add[NCH][V], d2[NCH][V];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        loadv<T, V>(reinterpret_cast<const T*>(p.x) + ro + c * 64 * V, xv[c]);
        loadv<T, V>(reinterpret_cast<const T*>(p.dy) + mo + c * 64 * V, dv[c]);
        loadv<T, V>(dy2p + c * 64 * V, d2[c]);
        loadv<T, V>(addp + c * 64 * V, add[c]);
        loadf<V>(p.gamma + c * 64 * V + lane * V, g[c]);
    }
    const float mean = p.mean[mrow], rstd = p.rstd[mrow];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) { dv[c][j] += k2 * d2[c][j]; add[c][j] *= ka; }
    if (p.dx_accumulate) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float t[V];
            if (f32out) loadf<V>(reinterpret_cast<const float*>(p.dx) + ro + c * 64 * V, t);
            else loadv<T, V>(reinterpret_cast<const T*>(p.dx) + ro + c * 64 * V, t);
#pragma unroll
            for (int j = 0; j < V; ++j) add[c][j] += t[j];
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) {
            xv[c][j] = (xv[c][j] - mean) * rstd;        // xhat
            dv[c][j] *= g[c][j];                        // dy * gamma
            s1 += dv[c][j];
            s2 += dv[c][j] * xv[c][j];
        }
    const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = rstd * (dv[c][j] - m1 - xv[c][j] * m2) + add[c][j];
        const size_t oi = ro + c * 64 * V;
        if (f32out) storef<V>(reinterpret_cast<float*>(p.dx) + oi, o);
        else storev<T, V>(reinterpret_cast<T*>(p.dx) + oi, o);
        if (p.dx2) {
            float m[V];
#pragma unroll
            for (int q4 = 0; q4 < V / 4; ++q4) {
                const f32x4 k4 = drop_keep4(p.drop2, (uint64_t)oi + 4 * q4);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[4 * q4 + j] = (f32out ? o[4 * q4 + j] : (float)(T)o[4 * q4 + j]) * k4[j];
            }
            if (f32out) storef<V>(reinterpret_cast<float*>(p.dx2) + oi, m);
            else storev<T, V>(reinterpret_cast<T*>(p.dx2) + oi, m);
        }
    }
}

// blocks of job j: gx = ceil(d/64) column groups x ceil(rows/rpw) row chunks (rpw = rows per workgroup here)
template <typename T>
__device__ __forceinline__ void ln_bwd_param_body(const LnBwdMulti& mj, int bid) {
    __shared__ float red[2][16][64 + 4];
    int job = 0;
#pragma unroll
    for (int i = 1; i < MEBT_LN_MAXJ; ++i)
        if (i < mj.n && bid >= mj.blk_start[i]) job = i;
    const LnBwdParams& p = mj.j[job];
    const int rows_per_block = mj.rpw[job];
    const int gx = (p.d + 63) / 64;
    const int lb = bid - mj.blk_start[job];
    const int bx = lb % gx, by = lb / gx;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = bx * 64 + cg * 4;
    const long m0 = (long)by * rows_per_block;
    const long m1 = m0 + rows_per_block < p.rows ? m0 + rows_per_block : p.rows;
    f32x4 ag = {0, 0, 0, 0}, ab = {0, 0, 0, 0};
    if (n < p.d)
#pragma unroll 4
        for (long m = m0 + rl; m < m1; m += 16) {
            const long mrow = map_row(m, p.seg, p.seg_stride, p.seg_off);
            const f32x4 xv = load4<T>(reinterpret_cast<const T*>(p.x) + (size_t)m * p.d + n);
            f32x4 dv = load4<T>(reinterpret_cast<const T*>(p.dy) + (size_t)mrow * p.d + n);
            if (p.dy2) dv += load4<T>(reinterpret_cast<const T*>(p.dy2) + (size_t)m * p.d + n);
            const float mean = p.mean[mrow], rstd = p.rstd[mrow];
#pragma unroll
            for (int j = 0; j < 4; ++j) { ag[j] += dv[j] * ((xv[j] - mean) * rstd); ab[j] += dv[j]; }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cg * 4 + j] = ag[j]; red[1][rl][cg * 4 + j] = ab[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[which][r][col];
        const int c = bx * 64 + col;
        if (c < p.d) atomicAdd((which ? p.dbeta : p.dgamma) + c, t);
    }
}

// ONE launch for both parts: workgroups [0, dx_blocks) compute dx rows, the rest the dgamma/dbeta column chunks.
// They read the same x / dy at the same time (the second reader hits L2) and a kernel boundary disappears.
// NCH = 0 selects the generic-d row kernel.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdMulti mj, const LnBwdMulti mp, int dx_blocks) {
    if ((int)blockIdx.x < dx_blocks) {
        if constexpr (NCH > 0) ln_bwd_dx_fast_body<T, NCH>(mj, blockIdx.x);
        else ln_bwd_dx_body<T>(mj, blockIdx.x);
    } else {
        ln_bwd_param_body<T>(mp, blockIdx.x - dx_blocks);
    }
}

// ------------------------------------------------------------------------------------------------
// column sums: out[n] += sum_m X[m,n]
// ------------------------------------------------------------------------------------------------
// Workgroup = 64 columns x `rows_per_block` rows: thread (cg = tid&15, rl = tid>>4) sums rows
// rl, rl+16, ... of its 4 columns (16 threads x 8/16 B = one 128/256-B segment per row), the 16
// row-lanes are reduced through LDS and ONE atomic per column leaves the workgroup, so at most
// M/rows_per_block adders meet on an address (float atomics collapse under contention).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, int M, int N, int ldx, float* out, int rows_per_block) {
    __shared__ float red[16][64 + 4];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + cg * 4;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    f32x4 acc = {0, 0, 0, 0};
    if (n < N)
#pragma unroll 8
        for (int m = m0 + rl; m < m1; m += 16) acc += load4<T>(X + (size_t)m * ldx + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cg * 4 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        const int col = blockIdx.x * 64 + threadIdx.x;
        if (col < N) atomicAdd(out + col, t);
    }
}

template <typename T>
__device__ __forceinline__ void colsum_grouped_body(const GroupedColsum& c, int bid) {
    __shared__ float red[16][64 + 4];
    int g = 0;
#pragma unroll
    for (int i = 1; i < MEBT_MAX_GROUP; ++i)
        if (i < c.n && bid >= c.blk_start[i]) g = i;
    const GroupedColsum::Item& it = c.g[g];
    const int b = bid - c.blk_start[g];
    const int bx = b % it.gx, by = b / it.gx;
    const T* X = reinterpret_cast<const T*>(it.X);
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = bx * 64 + cg * 4;
    const int m0 = by * it.rpb, m1 = min(it.M, m0 + it.rpb);
    f32x4 acc = {0, 0, 0, 0};
    if (n < it.N)
#pragma unroll 8
        for (int m = m0 + rl; m < m1; m += 16) acc += load4<T>(X + (size_t)m * it.ldx + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cg * 4 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        const int col = bx * 64 + threadIdx.x;
        if (col < it.N) atomicAdd(it.out + col, t);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const GroupedColsum c) { colsum_grouped_body<T>(c, blockIdx.x); }

// LayerNorm backward + the bias column sums of the same block in ONE launch: workgroups [0, dx_blocks) dx rows,
// [dx_blocks, cs_start) dgamma/dbeta column chunks, [cs_start, ...) grouped column sums (ln_bwd_kernel above is the
// two-part form; the body functions are shared)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_colsum_kernel(const LnBwdMulti mj, const LnBwdMulti mp, const GroupedColsum cs, int dx_blocks, int cs_start) {
    if ((int)blockIdx.x < dx_blocks) {
        if constexpr (NCH > 0) ln_bwd_dx_fast_body<T, NCH>(mj, blockIdx.x);
        else ln_bwd_dx_body<T>(mj, blockIdx.x);
    } else if ((int)blockIdx.x < cs_start) {
        ln_bwd_param_body<T>(mp, blockIdx.x - dx_blocks);
    } else {
        colsum_grouped_body<T>(cs, blockIdx.x - cs_start);
    }
}

// ------------------------------------------------------------------------------------------------
// cross-entropy over V with label smoothing + rank of the target (top-1 / top-5 accuracy)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const CeParams p) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const int b = row / p.NT, j = row % p.NT;
    const long pos = p.ti[(size_t)b * p.NT + j];
    const long tgt = p.x_ids[(size_t)b * p.N + pos];
    const float* l = p.logits + (size_t)row * p.V;
    const float lt = l[tgt];
    float mx = -INFINITY, sum = 0.f;
    int rank = 0;
    for (int e = threadIdx.x * 4; e < p.V; e += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(l + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mx = fmaxf(mx, v[q]);
            sum += v[q];
            rank += (v[q] > lt) || (v[q] == lt && (e + q) < tgt);
        }
    }
    mx = block_max(mx, sh);
    float se = 0.f;
    for (int e = threadIdx.x * 4; e < p.V; e += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(l + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) se += __expf(v[q] - mx);
    }
    se = block_sum(se, sh);
    sum = block_sum(sum, sh);
    const float rk = block_sum((float)rank, sh);
    if (threadIdx.x == 0) {
        const float lse = mx + logf(se);
        const float eps = p.label_smoothing;
        p.row_lse[row] = lse;
        p.row_loss[row] = (1.f - eps) * (lse - lt) + eps * (lse - sum / p.V);
        p.row_rank[row] = (int)(rk + 0.5f);
    }
}

// The same loss statistics AND the gradient of (loss_sum * scale) with respect to the logits from ONE pass over the row: the
// 16384 logits of a row stay in registers (16 float4 per thread), so the row is read from HBM once instead of three times
// (ce_fwd twice — the second time from L2 —, ce_bwd once).  Same per-thread summation order and the same expressions as
// ce_fwd_kernel / ce_bwd_kernel: statistics and gradient agree with the two-kernel path to fp32 rounding.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ce_fused_kernel(const CeParams p) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const int b = row / p.NT, j = row % p.NT;
    const long pos = p.ti[(size_t)b * p.NT + j];
    const long tgt = p.x_ids[(size_t)b * p.N + pos];
    const float* l = p.logits + (size_t)row * p.V;
    f32x4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = *reinterpret_cast<const f32x4*>(l + threadIdx.x * 4 + 1024 * c);
    const float lt = l[tgt];
    float mx = -INFINITY, sum = 0.f;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e = threadIdx.x * 4 + 1024 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mx = fmaxf(mx, v[c][q]);
            sum += v[c][q];
            rank += (v[c][q] > lt) || (v[c][q] == lt && (e + q) < tgt);
        }
    }
    mx = block_max(mx, sh);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) se += __expf(v[c][q] - mx);
    se = block_sum(se, sh);
    sum = block_sum(sum, sh);
    const float rk = block_sum((float)rank, sh);
    const float lse = mx + logf(se);
    const float eps = p.label_smoothing;
    if (threadIdx.x == 0) {
        p.row_lse[row] = lse;
        p.row_loss[row] = (1.f - eps) * (lse - lt) + eps * (lse - sum / p.V);
        p.row_rank[row] = (int)(rk + 0.5f);
    }
    T* d = reinterpret_cast<T*>(p.dlogits) + (size_t)row * p.V;
    const float sc = p.grad_scale, u = eps / p.V;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e = threadIdx.x * 4 + 1024 * c;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (__expf(v[c][q] - lse) - ((e + q) == tgt ? 1.f - eps : 0.f) - u) * sc;
        store4<T>(d + e, o);
    }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss, const int* row_rank, int rows, double* out) {
    __shared__ double sd[3][256];
    double s = 0, t1 = 0, t5 = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        s += row_loss[r];
        t1 += row_rank[r] < 1;
        t5 += row_rank[r] < 5;
    }
    sd[0][threadIdx.x] = s; sd[1][threadIdx.x] = t1; sd[2][threadIdx.x] = t5;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sd[0][threadIdx.x] += sd[0][threadIdx.x + o];
            sd[1][threadIdx.x] += sd[1][threadIdx.x + o];
            sd[2][threadIdx.x] += sd[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sd[0][0]; out[1] = sd[1][0]; out[2] = sd[2][0]; out[3] = rows; }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const CeBwdParams p) {
    const int row = blockIdx.x;
    const int b = row / p.NT, j = row % p.NT;
    const long pos = p.ti[(size_t)b * p.NT + j];
    const long tgt = p.x_ids[(size_t)b * p.N + pos];
    const float* l = p.logits + (size_t)row * p.V;
    T* d = reinterpret_cast<T*>(p.dlogits) + (size_t)row * p.V;
    const float lse = p.row_lse[row];
    const float sc = p.scale * (p.upstream ? *p.upstream : 1.0f);
    const float eps = p.label_smoothing, u = eps / p.V;
    for (int e = threadIdx.x * 4; e < p.V; e += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(l + e);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (__expf(v[q] - lse) - ((e + q) == tgt ? 1.f - eps : 0.f) - u) * sc;
        store4<T>(d + e, o);
    }
}

// ------------------------------------------------------------------------------------------------
// AdamW (decoupled weight decay, torch.optim.AdamW semantics) + bf16 mirror of the updated weights
// ------------------------------------------------------------------------------------------------
template <typename TG>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWParams a) {
    // (nontemporal loads/stores and grids of 2048..16384 workgroups were tried: all within noise, 4.5-5.4 TB/s
    //  depending on the box, of the ~6.3 TB/s a float4 copy reaches)
    const AdamWHyper h = {a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.bc1, a.bc2, a.grad_scale};
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < a.n; i += (size_t)gridDim.x * 1024) {
        f32x4 p = *reinterpret_cast<const f32x4*>(a.p + i);
        f32x4 g = load4<TG>(reinterpret_cast<const TG*>(a.g) + i);
        if constexpr (sizeof(TG) == 2)       // gradient exchanged by all-to-all: the ranks' bf16 contributions are added here, in fp32, in rank order
            for (int j = 1; j < a.g_pieces; ++j) g += load4<TG>(reinterpret_cast<const TG*>(a.g) + (size_t)j * a.g_stride + i);
        f32x4 m = *reinterpret_cast<const f32x4*>(a.m + i);
        f32x4 v = *reinterpret_cast<const f32x4*>(a.v + i);
        adamw_update4(p, g, m, v, h);
        *reinterpret_cast<f32x4*>(a.p + i) = p;
        *reinterpret_cast<f32x4*>(a.m + i) = m;
        *reinterpret_cast<f32x4*>(a.v + i) = v;
        if (a.p_bf16) store4<bf16_t>(reinterpret_cast<bf16_t*>(a.p_bf16) + i, p);
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void apply_dropout_kernel(const TS* src, TD* dst, size_t n, DropCfg d) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        f32x4 v = load4<TS>(src + i);
        v *= drop_keep4(d, i);
        store4<TD>(dst + i, v);
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_rows_kernel(const TS* src, TD* dst, long rows, int d, int s_seg, int s_stride, int s_off,
                                                        int d_seg, int d_stride, int d_off) {
    const int d4 = d / 4;
    const long total = rows * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / d4;
        const int e = (int)(i % d4) * 4;
        store4<TD>(dst + (size_t)map_row(r, d_seg, d_stride, d_off) * d + e, load4<TS>(src + (size_t)map_row(r, s_seg, s_stride, s_off) * d + e));
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, bf16_t* dst, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024)
        store4<bf16_t>(dst + i, *reinterpret_cast<const f32x4*>(src + i));
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define CHECK_LAUNCH() MEBT_HIP_CHECK(hipGetLastError())

int launch_embed_fwd(const EmbedParams& p, int dtype, hipStream_t stream) {
    if (p.d % 4) { mebt_set_error("embed: d must be a multiple of 4"); return MEBT_ESHAPE; }
    const long rows = (long)p.B * (p.NS + p.NC + p.NT);
    if (rows == 0) return MEBT_OK;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == MEBT_BF16) hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, stream, p);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_embed_bwd(const EmbedBwdParams& p, int dtype, hipStream_t stream) {
    const long rows = (long)p.B * (p.NC + p.NT);
    if (rows > 0) {
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (dtype == MEBT_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, stream, p);
        CHECK_LAUNCH();
    }
    int rc = launch_colsum(p.g_tgt, p.B * p.NT, p.d, p.d, p.g_mask_emb, dtype, stream);   // mask_emb: sum over all target rows
    if (rc) return rc;
    if (p.NS > 0) rc = launch_colsum(p.g_sos, p.B, p.NS * p.d, p.NS * p.d, p.g_sos_emb, dtype, stream);   // sos_emb: sum over batch
    return rc;
}

int launch_ln_fwd_multi(const LnFwdParams* jobs, int n, int dtype, hipStream_t stream) {
    LnFwdMulti mj;
    int k = 0, blocks = 0;
    for (int i = 0; i < n; ++i) {
        const LnFwdParams& p = jobs[i];
        if (p.rows <= 0) continue;
        if (p.d % 4 || p.d > 256 * LN_MAXC) { mebt_set_error("layernorm: d must be a multiple of 4 and <= 2048"); return MEBT_ESHAPE; }
        if (k == MEBT_LN_MAXJ) { mebt_set_error("layernorm: too many jobs in one launch"); return MEBT_EINVAL; }
        mj.j[k] = p;
        mj.blk_start[k] = blocks;
        blocks += (p.rows + 3) / 4;
        ++k;
    }
    if (!k) return MEBT_OK;
    mj.n = k;
    for (int i = k; i <= MEBT_LN_MAXJ; ++i) mj.blk_start[i] = blocks;
    bool same_d = true;
    for (int i = 1; i < k; ++i) same_d = same_d && mj.j[i].d == mj.j[0].d;
    const int d0 = mj.j[0].d;
    if (dtype == MEBT_BF16) {
        if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_fwd_fast_kernel<bf16_t, 2>), dim3(blocks), dim3(256), 0, stream, mj);
        else if (same_d && d0 == 512) hipLaunchKernelGGL((ln_fwd_fast_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, stream, mj);
        else if (same_d && d0 == 2048) hipLaunchKernelGGL((ln_fwd_fast_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 0, stream, mj);
        else hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, mj);
    } else {
        if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_fwd_fast_kernel<float, 4>), dim3(blocks), dim3(256), 0, stream, mj);
        else if (same_d && d0 == 256) hipLaunchKernelGGL((ln_fwd_fast_kernel<float, 1>), dim3(blocks), dim3(256), 0, stream, mj);
        else hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, mj);
    }
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_ln_fwd(const LnFwdParams& p, int dtype, hipStream_t stream) { return launch_ln_fwd_multi(&p, 1, dtype, stream); }

static int prepare_colsum(const GroupedColsum& c, GroupedColsum& k, int& blocks_out);
int launch_ln_bwd_multi(const LnBwdParams* jobs, int n, int dtype, hipStream_t stream, hipStream_t param_stream, const GroupedColsum* colsums) {
    LnBwdMulti mj, mp;
    int k = 0, blocks = 0, pblocks = 0;
    for (int i = 0; i < n; ++i) {
        const LnBwdParams& p = jobs[i];
        if (p.rows <= 0) continue;
        if (p.d % 4 || p.d > 256 * LN_MAXC) { mebt_set_error("layernorm: d must be a multiple of 4 and <= 2048"); return MEBT_ESHAPE; }
        if (k == MEBT_LN_MAXJ) { mebt_set_error("layernorm: too many jobs in one launch"); return MEBT_EINVAL; }
        mj.j[k] = p; mp.j[k] = p;
        mj.rpw[k] = 1;
        mj.blk_start[k] = blocks;
        blocks += (p.rows + 3) / 4;
        const int gx = (p.d + 63) / 64;
        int rpb = 256;
        while ((p.rows + rpb - 1) / rpb > 32) rpb *= 2;                                   // <= 32 adders per address
        while (rpb > 64 && (long)gx * ((p.rows + rpb - 1) / rpb) < 256) rpb /= 2;         // but fill the chip
        mp.rpw[k] = rpb;
        mp.blk_start[k] = pblocks;
        pblocks += gx * ((p.rows + rpb - 1) / rpb);
        ++k;
    }
    if (!k) return MEBT_OK;
    mj.n = mp.n = k;
    for (int i = k; i <= MEBT_LN_MAXJ; ++i) { mj.blk_start[i] = blocks; mp.blk_start[i] = pblocks; }
    (void)param_stream;                      // all parts are one launch on `stream` now
    bool same_d = true;
    for (int i = 1; i < k; ++i) same_d = same_d && mj.j[i].d == mj.j[0].d;
    const int d0 = mj.j[0].d;
    GroupedColsum cs;
    int cblocks = 0;
    if (colsums) { if (int rc = prepare_colsum(*colsums, cs, cblocks)) return rc; }
    if (cblocks) {
        const dim3 grid(blocks + pblocks + cblocks);
        const int cs_start = blocks + pblocks;
        if (dtype == MEBT_BF16) {
            if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_bwd_colsum_kernel<bf16_t, 2>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
            else if (same_d && d0 == 512) hipLaunchKernelGGL((ln_bwd_colsum_kernel<bf16_t, 1>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
            else if (same_d && d0 == 2048) hipLaunchKernelGGL((ln_bwd_colsum_kernel<bf16_t, 4>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
            else hipLaunchKernelGGL((ln_bwd_colsum_kernel<bf16_t, 0>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
        } else {
            if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_bwd_colsum_kernel<float, 4>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
            else if (same_d && d0 == 256) hipLaunchKernelGGL((ln_bwd_colsum_kernel<float, 1>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
            else hipLaunchKernelGGL((ln_bwd_colsum_kernel<float, 0>), grid, dim3(256), 0, stream, mj, mp, cs, blocks, cs_start);
        }
        CHECK_LAUNCH();
        return MEBT_OK;
    }
    const dim3 grid(blocks + pblocks);
    if (dtype == MEBT_BF16) {
        if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 2>), grid, dim3(256), 0, stream, mj, mp, blocks);
        else if (same_d && d0 == 512) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 1>), grid, dim3(256), 0, stream, mj, mp, blocks);
        else if (same_d && d0 == 2048) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, stream, mj, mp, blocks);
        else hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 0>), grid, dim3(256), 0, stream, mj, mp, blocks);
    } else {
        if (same_d && d0 == 1024) hipLaunchKernelGGL((ln_bwd_kernel<float, 4>), grid, dim3(256), 0, stream, mj, mp, blocks);
        else if (same_d && d0 == 256) hipLaunchKernelGGL((ln_bwd_kernel<float, 1>), grid, dim3(256), 0, stream, mj, mp, blocks);
        else hipLaunchKernelGGL((ln_bwd_kernel<float, 0>), grid, dim3(256), 0, stream, mj, mp, blocks);
    }
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_ln_bwd(const LnBwdParams& p, int dtype, hipStream_t stream, hipStream_t param_stream) { return launch_ln_bwd_multi(&p, 1, dtype, stream, param_stream, nullptr); }

int launch_colsum(const void* X, int M, int N, int ldx, float* out, int dtype, hipStream_t stream) {
    if (M <= 0 || N <= 0) return MEBT_OK;
    if (N % 4) { mebt_set_error("colsum: N must be a multiple of 4"); return MEBT_ESHAPE; }
    const int gx = (N + 63) / 64;
    int rpb = 256;
    while ((M + rpb - 1) / rpb > 32) rpb *= 2;          // <= 32 adders per address
    while (rpb > 64 && (long)gx * ((M + rpb - 1) / rpb) < 256) rpb /= 2;   // but fill the chip
    const dim3 grid(gx, (M + rpb - 1) / rpb);
    if (dtype == MEBT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(X), M, N, ldx, out, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, reinterpret_cast<const float*>(X), M, N, ldx, out, rpb);
    CHECK_LAUNCH();
    return MEBT_OK;
}

static int prepare_colsum(const GroupedColsum& c, GroupedColsum& k, int& blocks_out) {
    int n = 0, blocks = 0;
    for (int i = 0; i < c.n; ++i) {
        if (c.g[i].M <= 0 || c.g[i].N <= 0) continue;
        if (c.g[i].N % 4) { mebt_set_error("colsum: N must be a multiple of 4"); return MEBT_ESHAPE; }
        k.g[n] = c.g[i];
        k.g[n].gx = (c.g[i].N + 63) / 64;
        int rpb = 256;
        while ((c.g[i].M + rpb - 1) / rpb > 32) rpb *= 2;
        while (rpb > 64 && (long)k.g[n].gx * ((c.g[i].M + rpb - 1) / rpb) < 128) rpb /= 2;
        k.g[n].rpb = rpb;
        k.blk_start[n] = blocks;
        blocks += k.g[n].gx * ((c.g[i].M + rpb - 1) / rpb);
        ++n;
    }
    k.n = n;
    for (int i = n; i <= MEBT_MAX_GROUP; ++i) k.blk_start[i] = blocks;
    blocks_out = blocks;
    return MEBT_OK;
}

int launch_colsum_grouped(GroupedColsum& c, int dtype, hipStream_t stream) {
    GroupedColsum k;
    int blocks = 0;
    if (int rc = prepare_colsum(c, k, blocks)) return rc;
    if (!blocks) return MEBT_OK;
    if (dtype == MEBT_BF16) hipLaunchKernelGGL(colsum_grouped_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, k);
    else hipLaunchKernelGGL(colsum_grouped_kernel<float>, dim3(blocks), dim3(256), 0, stream, k);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_ce_fwd(const CeParams& p, hipStream_t stream) {
    if (p.V % 4) { mebt_set_error("cross-entropy: V must be a multiple of 4"); return MEBT_ESHAPE; }
    if (p.rows > 0) {
        if (p.dlogits && ce_fwd_can_fuse_grad(p.V)) {
            if (p.dl_bf16) hipLaunchKernelGGL((ce_fused_kernel<bf16_t, 16>), dim3(p.rows), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((ce_fused_kernel<float, 16>), dim3(p.rows), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL(ce_fwd_kernel, dim3(p.rows), dim3(256), 0, stream, p);
        }
        CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, stream, p.row_loss, p.row_rank, p.rows, p.out);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_ce_bwd(const CeBwdParams& p, int dtype, hipStream_t stream) {
    if (p.rows <= 0) return MEBT_OK;
    if (dtype == MEBT_BF16) hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(p.rows), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(p.rows), dim3(256), 0, stream, p);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_adamw(const AdamWParams& p, hipStream_t stream) {
    if (p.n == 0) return MEBT_OK;
    if (p.n % 4) { mebt_set_error("adamw: flat buffer length must be a multiple of 4"); return MEBT_ESHAPE; }
    const size_t blocks = (p.n / 4 + 255) / 256;
    if (p.g_bf16) hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(adamw_kernel<float>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, p);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_apply_dropout(const void* src, void* dst, size_t n, int src_f32, int dst_f32, const DropCfg& d, hipStream_t stream) {
    if (n == 0) return MEBT_OK;
    if (n % 4) { mebt_set_error("dropout: length must be a multiple of 4"); return MEBT_ESHAPE; }
    const size_t blocks = (n / 4 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096));
    if (src_f32 && dst_f32) hipLaunchKernelGGL((apply_dropout_kernel<float, float>), grid, dim3(256), 0, stream, (const float*)src, (float*)dst, n, d);
    else if (!src_f32 && !dst_f32) hipLaunchKernelGGL((apply_dropout_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n, d);
    else { mebt_set_error("dropout: mixed element types are not supported"); return MEBT_EDTYPE; }
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_copy_rows(const void* src, void* dst, long rows, int d, int src_f32, int dst_f32, int s_seg, int s_stride, int s_off,
                     int d_seg, int d_stride, int d_off, hipStream_t stream) {
    if (rows <= 0 || d <= 0) return MEBT_OK;
    if (d % 4) { mebt_set_error("copy_rows: d must be a multiple of 4"); return MEBT_ESHAPE; }
    const long blocks = (rows * (d / 4) + 255) / 256;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
#define COPY_ROWS(TS, TD) hipLaunchKernelGGL((copy_rows_kernel<TS, TD>), grid, dim3(256), 0, stream, (const TS*)src, (TD*)dst, rows, d, s_seg, s_stride, s_off, d_seg, d_stride, d_off)
    if (src_f32 && dst_f32) COPY_ROWS(float, float);
    else if (src_f32) COPY_ROWS(float, bf16_t);
    else if (dst_f32) COPY_ROWS(bf16_t, float);
    else COPY_ROWS(bf16_t, bf16_t);
#undef COPY_ROWS
    CHECK_LAUNCH();
    return MEBT_OK;
}

// Rows of `row_bytes` bytes (a multiple of 16) moved by a per-sample position list idx [B, n]: gather dst[b * n + j] = src[b * npos +
// idx[b, j]], or scatter dst[b * npos + idx[b, j]] = src[b * n + j] (duplicate-free lists, as every index set of the mask sampler
// is).  One wave per row, 16 bytes per lane and pass.  Positions outside [0, npos) are skipped.
namespace {
__global__ __launch_bounds__(256) void index_rows_kernel(const char* src, char* dst, const int64_t* idx, long rows, int n, int npos, int row_bytes, int scatter) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long b = r / n;
    const long pos = idx[r];
    if (pos < 0 || pos >= npos) return;
    const char* s = src + (size_t)(scatter ? r : b * npos + pos) * row_bytes;
    char* d = dst + (size_t)(scatter ? b * npos + pos : r) * row_bytes;
    for (int o = lane * 16; o < row_bytes; o += 1024) *reinterpret_cast<u32x4*>(d + o) = *reinterpret_cast<const u32x4*>(s + o);
}
}  // namespace
namespace {
__global__ __launch_bounds__(256) void cast_i64_i32_kernel(const int64_t* src, int32_t* dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (int32_t)src[i];
}
}  // namespace
int launch_cast_i64_i32(const int64_t* src, int32_t* dst, size_t n, hipStream_t stream) {
    if (!n) return MEBT_OK;
    hipLaunchKernelGGL(cast_i64_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
    CHECK_LAUNCH();
    return MEBT_OK;
}
int launch_index_rows(const void* src, void* dst, const int64_t* idx, int B, int n, int npos, int row_bytes, int scatter, hipStream_t stream) {
    const long rows = (long)B * n;
    if (rows <= 0) return MEBT_OK;
    if (row_bytes % 16) { mebt_set_error("index_rows: rows must be a multiple of 16 bytes"); return MEBT_ESHAPE; }
    hipLaunchKernelGGL(index_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const char*)src, (char*)dst, idx, rows, n, npos, row_bytes, scatter);
    CHECK_LAUNCH();
    return MEBT_OK;
}

int launch_cast_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t stream) {
    if (n == 0) return MEBT_OK;
    if (n % 4) { mebt_set_error("cast: length must be a multiple of 4"); return MEBT_ESHAPE; }
    const size_t blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, src, reinterpret_cast<bf16_t*>(dst), n);
    CHECK_LAUNCH();
    return MEBT_OK;
}

// Diagnostics (bench.py): ONE wave records the shader clock counter (s_memtime) and the constant 100 MHz reference (s_memrealtime),
// sleeps until `ref_ticks` reference ticks have passed and records both again: (d memtime / d realtime) x 100 MHz is the shader clock
// the chip really ran at in between - while whatever else is on the GPU runs (the sysfs sclk reading of these boxes does not move).
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ref_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < ref_ticks) __builtin_amdgcn_s_sleep(127);
    out[0] = t0; out[1] = r0; out[2] = __builtin_amdgcn_s_memtime(); out[3] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace
extern "C" int mebt_debug_clock_probe(unsigned long long* out4, uint64_t ref_ticks_100mhz, void* stream) {
    if (!out4 || ref_ticks_100mhz > 100000000ull) { mebt_set_error("clock_probe: null output or more than one second"); return MEBT_EINVAL; }
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out4, (unsigned long long)ref_ticks_100mhz);
    CHECK_LAUNCH();
    return MEBT_OK;
}
