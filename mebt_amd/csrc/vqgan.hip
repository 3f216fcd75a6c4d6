// 3D-VQGAN first stage on gfx950 (BASELINE.json configs[4]): the inference path of reference
// mebt/vqgan.py:82-93 (VQGAN.encode / decode) — SamePadConv3d / SamePadConvTranspose3d (:374-424), GroupNorm(32, eps 1e-6) +
// SiLU (:255-258, :17-18), ResBlock (:338-370), the nearest-codebook-entry search (modules/codebook.py:52-58) and the
// embedding lookup of decode (:91).
//
// Layout: activations are channels-last [B, T, H, W, C] (fp16 in the fast mode, fp32 in the parity mode); the network
// boundary (video in, video out) is the reference's fp32 [B, C, T, H, W], read / written directly by the first / last
// convolution.  A convolution is an implicit GEMM: M = output voxels, N = Cout, K = taps x Cin, weights pre-arranged as
// [Cout][tap][Cin] so that a K-slice of one tap is contiguous for both operands (an input row of Cin channels, a weight
// row).  Replicate padding = clamping the gathered input coordinate.  A transposed convolution with stride 2 is run as
// 2^k ordinary "sub-lattice" convolutions: the outputs of one parity class o = 2 o' + pi use a fixed subset of the taps
// (k == (pi + 1) mod 2), so each class is an implicit GEMM with 1/8 (or 1/4) of the taps and no zero-stuffing.
//
// Kernels: conv3d_mfma_f16_dma (v_mfma_f32_16x16x32_f16, LDS-DMA ring with per-lane gather addresses; Cin, Cout % 64 == 0),
// conv3d_mfma_f16 (the first version: 128 x 64 x 32 tiles, register-staged double buffer; Cin % 32 == 0,
// Cout % 64 == 0), conv3d_direct (any shape / dtype, fp32 FMA: the parity mode, the 3-channel first / last layers and odd
// channel counts), group-norm statistics + normalise/SiLU, codebook distance arg-min on an exact-fp32 MFMA score matrix,
// embedding rows.
#include "common.h"
#include "kernels.h"
#include "../../include/mebt_hip.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<f16_t>(const f16_t* p) {
    const f16x4 v = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, f32x4 v) {
    const f16x4 o = {(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
    *reinterpret_cast<f16x4*>(p) = o;
}

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// class-local voxel index -> (b, t', h', w')
struct Vox { int b, t, h, w; };
__device__ __forceinline__ Vox decode_vox(const mebt_conv3d_desc& p, long m) {
    Vox v;
    v.w = (int)(m % p.cW); m /= p.cW;
    v.h = (int)(m % p.cH); m /= p.cH;
    v.t = (int)(m % p.cT); v.b = (int)(m / p.cT);
    return v;
}
// channels-last element offset of the input voxel tap `j` of output voxel v reads (replicate padding = clamp)
__device__ __forceinline__ size_t in_voxel(const mebt_conv3d_desc& p, const Vox& v, int j) {
    const int ti = clampi(v.t * p.sm[0] + p.tap[j][0], p.Ti - 1);
    const int hi = clampi(v.h * p.sm[1] + p.tap[j][1], p.Hi - 1);
    const int wi = clampi(v.w * p.sm[2] + p.tap[j][2], p.Wi - 1);
    return (((size_t)v.b * p.Ti + ti) * p.Hi + hi) * p.Wi + wi;
}
__device__ __forceinline__ size_t out_voxel(const mebt_conv3d_desc& p, const Vox& v) {
    const int t = v.t * p.os[0] + p.pi[0], h = v.h * p.os[1] + p.pi[1], w = v.w * p.os[2] + p.pi[2];
    return (((size_t)v.b * p.To + t) * p.Ho + h) * p.Wo + w;
}

// ------------------------------------------------------------------------------------------------
// direct convolution: one thread per (output voxel, output channel), fp32 FMA chain over taps x Cin
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(const mebt_conv3d_desc p) {
    const long total = (long)p.B * p.cT * p.cH * p.cW * p.Cout;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % p.Cout);
    const Vox v = decode_vox(p, idx / p.Cout);
    float acc = p.bias ? p.bias[co] : 0.f;
    const T* wrow = reinterpret_cast<const T*>(p.w) + (size_t)co * p.ntaps * p.Cin;
    for (int j = 0; j < p.ntaps; ++j) {
        if (p.in_mode == 1) {            // fp32 [B, C, T, H, W] (the video)
            const int ti = clampi(v.t * p.sm[0] + p.tap[j][0], p.Ti - 1);
            const int hi = clampi(v.h * p.sm[1] + p.tap[j][1], p.Hi - 1);
            const int wi = clampi(v.w * p.sm[2] + p.tap[j][2], p.Wi - 1);
            const size_t plane = (size_t)p.Ti * p.Hi * p.Wi;
            const float* x = reinterpret_cast<const float*>(p.in) + (size_t)v.b * p.Cin * plane + ((size_t)ti * p.Hi + hi) * p.Wi + wi;
            for (int ci = 0; ci < p.Cin; ++ci) acc += x[ci * plane] * (float)wrow[j * p.Cin + ci];
        } else {
            const T* x = reinterpret_cast<const T*>(p.in) + in_voxel(p, v, j) * p.Cin;
            const T* wj = wrow + (size_t)j * p.Cin;
            if ((p.Cin & 3) == 0) {
                for (int ci = 0; ci < p.Cin; ci += 4) {
                    const f32x4 a = ld4<T>(x + ci), b = ld4<T>(wj + ci);
                    acc += a[0] * b[0]; acc += a[1] * b[1]; acc += a[2] * b[2]; acc += a[3] * b[3];
                }
            } else {
                for (int ci = 0; ci < p.Cin; ++ci) acc += (float)x[ci] * (float)wj[ci];
            }
        }
    }
    const size_t ov = out_voxel(p, v);
    if (p.resid) acc += (float)reinterpret_cast<const T*>(p.resid)[ov * p.Cout + co];
    if (p.out_mode == 0) reinterpret_cast<T*>(p.out)[ov * p.Cout + co] = (T)acc;
    else if (p.out_mode == 1) reinterpret_cast<float*>(p.out)[ov * p.Cout + co] = acc;
    else {                                  // fp32 [B, C, T, H, W]
        const size_t plane = (size_t)p.To * p.Ho * p.Wo;
        const size_t bv = ov / plane, sp = ov % plane;
        reinterpret_cast<float*>(p.out)[(bv * p.Cout + co) * plane + sp] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// thin convolutions: the 3-channel first / last layers (Cin * Cout small, output voxels many).  One thread per output voxel
// computes ALL output channels (accumulators in registers); the whole weight set sits in LDS as fp32 [tap][ci][co] and is
// read as broadcasts (every lane the same address), so an input value is loaded once for all its output channels.  The
// per-(voxel, channel) direct kernel spent 2.9 ms on each of these layers (33 M threads x 81 dependent scalar loads, or
// 3 M threads x 1728 MACs each); this one is bound by the input gathers.
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void conv3d_voxel_kernel(const mebt_conv3d_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem_v[];
    float* wl = reinterpret_cast<float*>(smem_v);                  // [ntaps][Cin][MAXC]
    const int nw = p.ntaps * p.Cin;
    for (int i = threadIdx.x; i < nw * MAXC; i += 256) {
        const int co = i % MAXC, tc = i / MAXC;                    // tc = tap * Cin + ci ; source layout [Cout][tap][Cin]
        wl[i] = co < p.Cout ? (float)reinterpret_cast<const T*>(p.w)[(size_t)co * nw + tc] : 0.f;
    }
    __syncthreads();
    const long total = (long)p.B * p.cT * p.cH * p.cW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const Vox v = decode_vox(p, idx);
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
    const size_t plane = (size_t)p.Ti * p.Hi * p.Wi;
    for (int j = 0; j < p.ntaps; ++j) {
        const int ti = clampi(v.t * p.sm[0] + p.tap[j][0], p.Ti - 1);
        const int hi = clampi(v.h * p.sm[1] + p.tap[j][1], p.Hi - 1);
        const int wi = clampi(v.w * p.sm[2] + p.tap[j][2], p.Wi - 1);
        const size_t sp = ((size_t)ti * p.Hi + hi) * p.Wi + wi;
        const float* wt = wl + (size_t)j * p.Cin * MAXC;
        if (p.in_mode == 1) {                                   // fp32 [B, C, T, H, W]
            const float* x = reinterpret_cast<const float*>(p.in) + (size_t)v.b * p.Cin * plane + sp;
            for (int ci = 0; ci < p.Cin; ++ci) {
                const float xv = x[ci * plane];
#pragma unroll
                for (int c = 0; c < MAXC; c += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + ci * MAXC + c);
                    acc[c] += xv * w4[0]; acc[c + 1] += xv * w4[1]; acc[c + 2] += xv * w4[2]; acc[c + 3] += xv * w4[3];
                }
            }
        } else {
            const T* x = reinterpret_cast<const T*>(p.in) + ((size_t)v.b * plane + sp) * p.Cin;
            for (int ci = 0; ci < p.Cin; ci += 4) {             // Cin % 4 == 0 on this path (checked by the launcher)
                const f32x4 x4 = ld4<T>(x + ci);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < MAXC; c += 4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + (ci + q) * MAXC + c);
                        acc[c] += x4[q] * w4[0]; acc[c + 1] += x4[q] * w4[1]; acc[c + 2] += x4[q] * w4[2]; acc[c + 3] += x4[q] * w4[3];
                    }
            }
        }
    }
    const size_t ov = out_voxel(p, v);
    const size_t oplane = (size_t)p.To * p.Ho * p.Wo;
    for (int c = 0; c < p.Cout; ++c) {
        float a = acc[c];
        if (p.resid) a += (float)reinterpret_cast<const T*>(p.resid)[ov * p.Cout + c];
        if (p.out_mode == 0) reinterpret_cast<T*>(p.out)[ov * p.Cout + c] = (T)a;
        else if (p.out_mode == 1) reinterpret_cast<float*>(p.out)[ov * p.Cout + c] = a;
        else reinterpret_cast<float*>(p.out)[((ov / oplane) * p.Cout + c) * oplane + ov % oplane] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// The same thin layers on the matrix cores (fp16 mode).  conv3d_voxel_kernel ran them on the vector ALUs: 1.0 ms for the 3 -> 32
// first layer and 1.9 ms for the 64 -> 3 last layer at batch 16 (12 % of encode + decode; profiles/r04_vqgan16.txt).
// Both use v_mfma_f32_16x16x32_f16 with the operands swapped (D^T = W A^T: a lane owns 4 consecutive output channels of one voxel).
//
// conv3d_first_mfma_kernel (fp32 [B, C, T, H, W] video in, CIN = 3): K = taps x CIN (81) padded to a multiple of 32.  There is no
// contiguous k-run to copy — k = 3 tap + channel — so every lane gathers the 8 values of its A fragment (voxel lane & 15, k group
// lane >> 4) from the video with clamped coordinates (replicate padding) and rounds them to fp16; the weight fragments
// (Cout / 16 x K / 32, at most 8) stay in registers for the whole kernel.  A wave walks 16-voxel row blocks with a grid stride.
// ------------------------------------------------------------------------------------------------
template <int CIN, int NKS, int NCF>
__global__ __launch_bounds__(256) void conv3d_first_mfma_kernel(const mebt_conv3d_desc p) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int K = p.ntaps * CIN;
    const f16_t* w = reinterpret_cast<const f16_t*>(p.w);
    f16x8 wf[NKS][NCF];
    uint32_t tk[NKS][8];           // this lane's k -> (dt + 8, dh + 8, dw + 8, channel | valid << 7), voxel-independent
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * ks + 8 * g + j;
            const int tap = k / CIN, ci = k - tap * CIN;
            tk[ks][j] = k < K ? ((uint32_t)(p.tap[tap][0] + 8) | ((uint32_t)(p.tap[tap][1] + 8) << 8) | ((uint32_t)(p.tap[tap][2] + 8) << 16) | ((uint32_t)(ci | 128) << 24)) : 0u;
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) wf[ks][cf][j] = k < K ? w[(size_t)(16 * cf + r) * K + k] : (f16_t)0.f;
        }
    }
    const long Mcls = (long)p.B * p.cT * p.cH * p.cW;
    const long nblk = (Mcls + 15) / 16;
    const uint32_t plane = (uint32_t)(p.Ti * p.Hi * p.Wi);       // the launcher vouches for < 2^31 input elements: 32-bit index arithmetic
    const float* x = reinterpret_cast<const float*>(p.in);
    for (long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += (long)gridDim.x * 4) {
        long m = blk * 16 + r;
        const bool live = m < Mcls;
        if (!live) m = Mcls - 1;
        const Vox v = decode_vox(p, m);
        const float* xb = x + (uint32_t)v.b * CIN * plane;
        const int t0 = v.t * p.sm[0] - 8, h0 = v.h * p.sm[1] - 8, w0 = v.w * p.sm[2] - 8;
        f32x4 acc[NCF];
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) acc[cf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            f16x8 af;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t c = tk[ks][j];
                const int ti = clampi(t0 + (int)(c & 255u), p.Ti - 1);
                const int hi = clampi(h0 + (int)((c >> 8) & 255u), p.Hi - 1);
                const int wi = clampi(w0 + (int)((c >> 16) & 255u), p.Wi - 1);
                const float val = xb[((c >> 24) & 127u) * plane + (uint32_t)((ti * p.Hi + hi) * p.Wi + wi)];  // k >= K: element (0, clamped origin), times a zero weight
                af[j] = (f16_t)val;
            }
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][cf], af, acc[cf], 0, 0, 0);
        }
        if (!live) continue;
        const size_t ov = out_voxel(p, v);
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) {
            const int n = 16 * cf + 4 * g;
            f32x4 o = acc[cf];
            if (p.bias) o += *reinterpret_cast<const f32x4*>(p.bias + n);
            st4<f16_t>(reinterpret_cast<f16_t*>(p.out) + ov * p.Cout + n, o);
        }
    }
}

// conv3d_last_mfma_kernel (channels-last fp16 in, Cin % 32 == 0, Cout <= 16 padded to one 16-column fragment; any output mode):
// K = taps x Cin (1728) in 32-channel steps of one tap, so an A fragment is ONE 16-byte load per lane (voxel lane & 15, channels
// c0 + 8 (lane >> 4) .. + 7 of the tap's input voxel).  The weights sit in LDS already arranged as fragments ([k-step][lane] x 16 B:
// conflict-free, one ds_read_b128 per k-step); a wave multiplies FOUR 16-voxel row blocks per k-step against each fragment.
template <int KC2>
__global__ __launch_bounds__(256) void conv3d_last_mfma_kernel(const mebt_conv3d_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem_l[];
    f16x8* wl = reinterpret_cast<f16x8*>(smem_l);            // [nks][64]
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int kc = p.Cin / 32, nks = p.ntaps * kc, K = p.ntaps * p.Cin;
    const f16_t* w = reinterpret_cast<const f16_t*>(p.w);
    for (int i = threadIdx.x; i < nks * 64; i += 256) {
        const int ks = i >> 6, l = i & 63, n = l & 15, k0 = 32 * ks + 8 * (l >> 4);
        f16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < p.Cout) f = *reinterpret_cast<const f16x8*>(w + (size_t)n * K + k0);
        wl[i] = f;
    }
    __syncthreads();
    constexpr int RF = 4;
    const long Mcls = (long)p.B * p.cT * p.cH * p.cW;
    const long nblk = (Mcls + 16 * RF - 1) / (16 * RF);
    const f16_t* in = reinterpret_cast<const f16_t*>(p.in);
    for (long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += (long)gridDim.x * 4) {
        Vox v[RF];
        bool live[RF];
#pragma unroll
        for (int i = 0; i < RF; ++i) {
            long m = (blk * RF + i) * 16 + r;
            live[i] = m < Mcls;
            if (!live[i]) m = Mcls - 1;
            v[i] = decode_vox(p, m);
        }
        f32x4 acc[RF];
#pragma unroll
        for (int i = 0; i < RF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the loads of tap j + 1 are issued before the MFMAs of tap j (two register sets): one exposed L1 / L2 round trip per
        // 64 voxels instead of one per k-step.  KC2 = Cin / 32 k-steps per tap (2 for the 64-channel last layer).
        f16x8 af[2][KC2][RF];
        auto load_tap = [&](int j, int buf) {
#pragma unroll
            for (int i = 0; i < RF; ++i) {
                const int ti = clampi(v[i].t * p.sm[0] + p.tap[j][0], p.Ti - 1);
                const int hi = clampi(v[i].h * p.sm[1] + p.tap[j][1], p.Hi - 1);
                const int wi = clampi(v[i].w * p.sm[2] + p.tap[j][2], p.Wi - 1);
                const uint32_t vox = (uint32_t)(((v[i].b * p.Ti + ti) * p.Hi + hi) * p.Wi + wi);      // < 2^31 input elements (launcher)
                const f16_t* src = in + vox * (uint32_t)p.Cin + 8 * g;
#pragma unroll
                for (int c = 0; c < KC2; ++c)
                    if (c < kc) af[buf][c][i] = *reinterpret_cast<const f16x8*>(src + 32 * c);
            }
        };
        load_tap(0, 0);
        for (int j = 0; j < p.ntaps; j += 2) {
            if (j + 1 < p.ntaps) load_tap(j + 1, 1);
#pragma unroll
            for (int c = 0; c < KC2; ++c)
                if (c < kc) {
                    const f16x8 wf = wl[(j * kc + c) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < RF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[0][c][i], acc[i], 0, 0, 0);
                }
            if (j + 1 >= p.ntaps) break;
            if (j + 2 < p.ntaps) load_tap(j + 2, 0);
#pragma unroll
            for (int c = 0; c < KC2; ++c)
                if (c < kc) {
                    const f16x8 wf = wl[((j + 1) * kc + c) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < RF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[1][c][i], acc[i], 0, 0, 0);
                }
        }
        const size_t oplane = (size_t)p.To * p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < RF; ++i) {
            if (!live[i]) continue;
            const size_t ov = out_voxel(p, v[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * g + q;
                if (c >= p.Cout) continue;
                float a = acc[i][q] + (p.bias ? p.bias[c] : 0.f);
                if (p.resid) a += (float)reinterpret_cast<const f16_t*>(p.resid)[ov * p.Cout + c];
                if (p.out_mode == 0) reinterpret_cast<f16_t*>(p.out)[ov * p.Cout + c] = (f16_t)a;
                else if (p.out_mode == 1) reinterpret_cast<float*>(p.out)[ov * p.Cout + c] = a;
                else reinterpret_cast<float*>(p.out)[((ov / oplane) * p.Cout + c) * oplane + ov % oplane] = a;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA implicit GEMM, fp16: C[m, n] = sum_{tap, ci} in[vox(m, tap), ci] * w[n][tap][ci]
// ------------------------------------------------------------------------------------------------
constexpr int CBM = 128, CBN = 64, CBK = 32, CLD = 40;      // LDS rows of 32 halfs padded to 40 (80 B): conflict-free b128 reads

__global__ __launch_bounds__(256) void conv3d_mfma_f16_kernel(const mebt_conv3d_desc p) {
    __shared__ __attribute__((aligned(16))) f16_t sA[2][CBM * CLD];
    __shared__ __attribute__((aligned(16))) f16_t sB[2][CBN * CLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long Mcls = (long)p.B * p.cT * p.cH * p.cW;
    const long m0 = (long)blockIdx.x * CBM;
    const int n0 = blockIdx.y * CBN;
    const f16_t* in = reinterpret_cast<const f16_t*>(p.in);
    const f16_t* w = reinterpret_cast<const f16_t*>(p.w);
    // loader roles: A chunks id = tid + 256 i (i < 2): row id >> 2, 16-byte chunk id & 3; B chunk: row tid >> 2, chunk tid & 3
    Vox va[2];
    int arow[2], ach[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i;
        arow[i] = id >> 2; ach[i] = id & 3;
        long m = m0 + arow[i];
        if (m >= Mcls) m = Mcls - 1;
        va[i] = decode_vox(p, m);
    }
    const int brow = tid >> 2, bch = tid & 3;
    const f16_t* wrow = w + (size_t)(n0 + brow) * p.ntaps * p.Cin + bch * 8;
    const int kc = p.Cin / CBK;                     // channel chunks per tap
    const int nk = p.ntaps * kc;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[2], rb;
    size_t abase[2] = {0, 0};
    auto gload = [&](int kk) {
        const int j = kk / kc, c0 = (kk % kc) * CBK;
        if (kk % kc == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) abase[i] = in_voxel(p, va[i], j) * p.Cin;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const u32x4*>(in + abase[i] + c0 + ach[i] * 8);
        rb = *reinterpret_cast<const u32x4*>(wrow + (size_t)j * p.Cin + c0);
    };
    auto lstore = [&](int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(&sA[s][arow[i] * CLD + ach[i] * 8]) = ra[i];
        *reinterpret_cast<u32x4*>(&sB[s][brow * CLD + bch * 8]) = rb;
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fc = (lane >> 4) * 8;
    for (int kk = 0; kk < nk; ++kk) {
        const int s = kk & 1;
        if (kk + 1 < nk) gload(kk + 1);
        f16x8 af[2], bf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(&sA[s][(32 * wave + 16 * i + fr) * CLD + fc]);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(&sB[s][(16 * j + fr) * CLD + fc]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (kk + 1 < nk) lstore(s ^ 1);
        __syncthreads();
    }
    // D^T = B A^T: lane holds row (lane & 15) of row block i, columns 16 j + 4 (lane >> 4) .. + 3
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = m0 + 32 * wave + 16 * i + fr;
        if (m >= Mcls) continue;
        const size_t ov = out_voxel(p, decode_vox(p, m));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 16 * j + 4 * (lane >> 4);
            f32x4 v = acc[i][j];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.resid) v += ld4<f16_t>(reinterpret_cast<const f16_t*>(p.resid) + ov * p.Cout + n);
            if (p.out_mode == 1) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ov * p.Cout + n) = v;
            else st4<f16_t>(reinterpret_cast<f16_t*>(p.out) + ov * p.Cout + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same implicit GEMM on the Linear layers' machinery (gemm_kernels.h): k-steps of 64 channels of one tap, both operands
// copied global -> LDS by LDS-DMA into a ring of NST stages (counted vmcnt waits, one raw s_barrier per k-step), KC images
// [row][64 k] with the XOR swizzle chunk ^ (row >> 1 & 7) read with ds_read_b128, TBM x TBN tiles (128 x 128 for the 128- and
// 256-channel layers, 256 x 64 for the 64-channel ones), 4 waves of 64 rows each (32 MFMAs per k-step against 8 in the
// register-staged kernel above), epilogue staged through LDS so that every lane stores 8 consecutive channels (16 bytes).
// An A row of a k-step is the 128 contiguous bytes in[vox(m, tap)][c0 .. c0 + 63]: each lane of a DMA piece supplies its own
// source address (row base of ITS row, recomputed when the tap changes — replicate padding is the clamp inside in_voxel), so
// the gather costs no register pass.  Cin % 64 == 0, Cout % 64 == 0.
// ------------------------------------------------------------------------------------------------
template <int TBM, int TBN, int NST>
__global__ __launch_bounds__(256) void conv3d_mfma_f16_dma_kernel(const mebt_conv3d_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KB = 64;
    constexpr int STAGE = (TBM + TBN) * KB * 2;
    constexpr int PA = TBM / 32, PB = TBN / 32;           // DMA pieces (8 rows x 128 B) per wave and stage
    constexpr int LPT = PA + PB;
    constexpr int WM = TBM / 64, WN = 4 / WM;             // waves over rows x columns: a wave always owns 64 rows
    constexpr int TM = 4, TN = TBN / (16 * WN);           // 16 x 16 fragments per wave
    constexpr int AHEAD = NST - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const long Mcls = (long)p.B * p.cT * p.cH * p.cW;
    const long m0 = (long)blockIdx.x * TBM;
    const int n0 = blockIdx.y * TBN;
    const int kc = p.Cin / KB, nk = p.ntaps * kc;
    const size_t in_bytes = (size_t)p.B * p.Ti * p.Hi * p.Wi * p.Cin * 2, w_bytes = (size_t)p.Cout * p.ntaps * p.Cin * 2;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.in, in_bytes), rB = make_rsrc(p.w, w_bytes);

    // this lane's rows: piece q = wave + 4 i covers tile rows 8 q .. 8 q + 7, lane -> row 8 q + (lane >> 3), 16-byte slot lane & 7
    Vox va[PA];
    uint32_t aoff[PA], boff[PB];
    unsigned alds[PA], blds[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = 8 * (wave + 4 * i) + (lane >> 3);
        long m = m0 + row;
        if (m >= Mcls) m = Mcls - 1;
        va[i] = decode_vox(p, m);
        aoff[i] = 0;
        alds[i] = (wave + 4 * i) * 1024;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = 8 * (wave + 4 * i) + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        boff[i] = (uint32_t)(n0 + row) * p.ntaps * p.Cin * 2 + ch * 16;
        blds[i] = TBM * KB * 2 + (wave + 4 * i) * 1024;
    }
    int cur_tap = -1;
    auto issue = [&](int kk) {
        char* st = smem + (kk % NST) * STAGE;
        const int j = kk / kc, c0 = (kk - j * kc) * KB;
        if (j != cur_tap) {
            cur_tap = j;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int row = 8 * (wave + 4 * i) + (lane >> 3);
                const int ch = (lane & 7) ^ ((row >> 1) & 7);
                aoff[i] = (uint32_t)(in_voxel(p, va[i], j) * p.Cin * 2) + ch * 16;
            }
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) dma16(rA, (unsigned)(size_t)(lds_char_ptr)(st + alds[i]), aoff[i] + c0 * 2);
        const uint32_t bk = ((uint32_t)j * p.Cin + c0) * 2;
#pragma unroll
        for (int i = 0; i < PB; ++i) dma16(rB, (unsigned)(size_t)(lds_char_ptr)(st + blds[i]), boff[i] + bk);
    };
    auto frag = [&](const char* tile, int blk16, int ks) -> f16x8 {
        const int row = blk16 * 16 + (lane & 15);
        const int c = 4 * ks + (lane >> 4);
        return *reinterpret_cast<const f16x8*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < nk) issue(a);
    for (int t = 0; t < nk; ++t) {
        const int younger = min(AHEAD - 1, nk - 1 - t);
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        if (t + AHEAD < nk) issue(t + AHEAD);
        const char* sA = smem + (t % NST) * STAGE;
        const char* sB = sA + TBM * KB * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = frag(sA, wm * TM + i, ks);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = frag(sB, wn * TN + j, ks);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: a 16-row fp32 slab per wave through LDS, then 8 consecutive channels per lane
    constexpr int COLS = TN * 16, LD = COLS + 4, LPR = COLS / 8, RPI = 64 / LPR, NPASS = 16 / RPI;
    __syncthreads();
    float* st = reinterpret_cast<float*>(smem) + wave * (16 * LD);
    const int r0 = lane / LPR, c8 = (lane % LPR) * 8;
    const int n = n0 + wn * COLS + c8;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) { b_lo = *reinterpret_cast<const f32x4*>(p.bias + n); b_hi = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(st + (lane & 15) * LD + j * 16 + 4 * (lane >> 4)) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int row = r * RPI + r0;
            f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * LD + c8);
            f32x4 hi = *reinterpret_cast<const f32x4*>(st + row * LD + c8 + 4);
            const long m = m0 + wm * 64 + i * 16 + row;
            if (m >= Mcls) continue;
            const size_t oi = out_voxel(p, decode_vox(p, m)) * p.Cout + n;
            lo += b_lo; hi += b_hi;
            if (p.resid) {
                const f16x8 rv = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16_t*>(p.resid) + oi);
#pragma unroll
                for (int q = 0; q < 4; ++q) { lo[q] += (float)rv[q]; hi[q] += (float)rv[4 + q]; }
            }
            if (p.out_mode == 1) {
                float* o = reinterpret_cast<float*>(p.out) + oi;
                *reinterpret_cast<f32x4*>(o) = lo;
                *reinterpret_cast<f32x4*>(o + 4) = hi;
            } else {
                const f16x8 o = {(f16_t)lo[0], (f16_t)lo[1], (f16_t)lo[2], (f16_t)lo[3], (f16_t)hi[0], (f16_t)hi[1], (f16_t)hi[2], (f16_t)hi[3]};
                *reinterpret_cast<f16x8*>(reinterpret_cast<f16_t*>(p.out) + oi) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slab is rewritten by the next row block
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (32 groups, eps 1e-6) + SiLU, channels-last
// ------------------------------------------------------------------------------------------------
// pass 1: stats[b][g] = {sum, sum of squares}.  A workgroup takes a slab of voxels of one sample: thread = 4 consecutive
// channels of every (256 / (C/4))-th row, per-channel partial sums in registers -> LDS -> one atomic pair per group.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* x, float* stats, long vox_per_sample, int C, int rows_per_block) {
    __shared__ float red[2][256 * 4];
    const int b = blockIdx.y;
    const int lanes_per_row = C / 4, rows_par = 256 / lanes_per_row;
    const int cl = threadIdx.x % lanes_per_row, rl = threadIdx.x / lanes_per_row;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < vox_per_sample ? r0 + rows_per_block : vox_per_sample;
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    if (rl < rows_par)
        for (long r = r0 + rl; r < r1; r += rows_par) {
            const f32x4 v = ld4<T>(x + ((size_t)b * vox_per_sample + r) * C + cl * 4);
            s += v; q += v * v;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][threadIdx.x * 4 + j] = s[j]; red[1][threadIdx.x * 4 + j] = q[j]; }
    __syncthreads();
    // thread c < C: total of channel c over the row lanes, then per group (C/32 consecutive channels)
    const int cpg = C / 32;
    if ((int)threadIdx.x < 32) {
        const int g = threadIdx.x;
        float gs = 0.f, gq = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const int lane_c = c / 4, j = c % 4;
            for (int r = 0; r < rows_par; ++r) {
                gs += red[0][(r * lanes_per_row + lane_c) * 4 + j];
                gq += red[1][(r * lanes_per_row + lane_c) * 4 + j];
            }
        }
        atomicAdd(stats + ((size_t)b * 32 + g) * 2, gs);
        atomicAdd(stats + ((size_t)b * 32 + g) * 2 + 1, gq);
    }
}
// pass 2: y = silu((x - mean_g) * rstd_g * gamma_c + beta_c)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_silu_kernel(const T* x, T* y, const float* stats, const float* gamma, const float* beta,
                                                            long vox_per_sample, int C, long total4) {
    const int cpg = C / 32;
    const float inv_n = 1.0f / ((float)vox_per_sample * cpg);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int c = (int)(e % C);
        const long b = e / ((long)vox_per_sample * C);
        const f32x4 v = ld4<T>(x + e);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c), b4 = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (c + j) / cpg;
            const float sum = stats[((size_t)b * 32 + g) * 2], sq = stats[((size_t)b * 32 + g) * 2 + 1];
            const float mean = sum * inv_n;
            const float var = fmaxf(sq * inv_n - mean * mean, 0.f);
            const float h = (v[j] - mean) * rsqrtf(var + 1e-6f) * g4[j] + b4[j];
            o[j] = h / (1.0f + __expf(-h));
        }
        st4<T>(y + e, o);
    }
}

// ------------------------------------------------------------------------------------------------
// codebook: ids[m] = argmin_j (|z_m|^2 - 2 z_m . e_j) + |e_j|^2, evaluated in the reference's order (codebook.py:54-58)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* e, float* out, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) { const float v = e[(size_t)row * d + k]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}
__global__ __launch_bounds__(256) void codebook_argmin_kernel(const float* score /*[M, n_codes] = z E^T*/, const float* z, const float* esq,
                                                              int64_t* ids, int n_codes, int d) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float szz;
    const int m = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        float s = 0.f;
        for (int k = lane; k < d; k += 64) { const float v = z[(size_t)m * d + k]; s += v * v; }
        s = wave_sum(s);
        if (lane == 0) szz = s;
    }
    __syncthreads();
    const float zz = szz;
    float best = INFINITY;
    int bi = 0x7FFFFFFF;
    for (int j = threadIdx.x; j < n_codes; j += 256) {
        const float dist = (zz - 2.0f * score[(size_t)m * n_codes + j]) + esq[j];
        if (dist < best) { best = dist; bi = j; }          // ascending j per thread: the first minimum is kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sv[wv] = best; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (sv[k] < best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        ids[m] = bi;
    }
}

// ------------------------------------------------------------------------------------------------
// Filtered search (large M x n_codes): the [M, n_codes] score matrix costs 137 GFLOP at config 5 / batch 16 — 1.16 ms on the exact
// fp32 MFMA GEMM (77 % of its peak), a third of the whole encode.  The bf16 GEMM family computes APPROXIMATE scores 5x faster; this
// kernel turns them into the same ids the exact search gives:
//   d~_j = (|z|^2 - 2 s~_j) + |e_j|^2,  |d~_j - d_j| <= mu := (2^-7 + 2^-8) |z| max_j |e_j| (1 + 2^-6)   (bf16 round-to-nearest: |x - x~| <= 2^-9 |x|
//   per vector, Cauchy-Schwarz on both factors, doubled by the -2: 2^-7; the score itself is stored as bf16: another 2^-9, doubled:
//   2^-8; the fp32 accumulation of the MFMA is inside the 2^-6 slack),
// so the true arg-min is among C = { j : d~_j <= min d~ + 2 mu }.  For every j in C the distance is re-evaluated in exact fp32 (a
// sequential FMA chain over k, the order of the exact GEMM) and the first minimum in index order wins, as in codebook_argmin_kernel.
// |C| is a handful on real codebooks; more than 1024 candidates (degenerate codebook) -> every code is re-evaluated.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vec_max_sqrt_kernel(const float* v, float* out, int n) {      // out[0] = sqrt(max v)
    __shared__ float sh[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, v[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrtf(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])));
}
constexpr int CB_CAP = 1024;
// scores arrive as bf16 (half the bytes of the [M, n_codes] matrix in both directions: the search is bound by that matrix, 1 GB in
// fp32 at config 5 / batch 16); rounding the fp32 accumulator to bf16 adds <= 2^-9 |s| <= 2^-9 |z| max|e| (1 + 2^-6) per score, i.e.
// 2^-8 |z| max|e| (1 + 2^-6) to d~, on top of mu above.  FAST (n_codes = 16384): the 64 approximate distances of a thread stay in
// registers between the minimum pass and the candidate pass — one pass over the row, 16-byte loads.
template <bool FAST>
__global__ __launch_bounds__(256) void codebook_filter_kernel(const bf16_t* score /*[M, n_codes] ~ z E^T (bf16 operands, bf16 result)*/, const float* z,
                                                              const float* e, const float* esq, const float* emax, int64_t* ids, int n_codes, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    float* zl = reinterpret_cast<float*>(smem_c);                  // [d]
    int* cand = reinterpret_cast<int*>(zl + d);                    // [CB_CAP]
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float szz, smin;
    __shared__ unsigned int ncand;
    const int m = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < d; k += 256) zl[k] = z[(size_t)m * d + k];
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    if (wv == 0) {           // |z|^2 in the summation order of codebook_argmin_kernel
        float s = 0.f;
        for (int k = lane; k < d; k += 64) { const float v = zl[k]; s += v * v; }
        s = wave_sum(s);
        if (lane == 0) szz = s;
    }
    __syncthreads();
    const float zz = szz;
    const bf16_t* srow = score + (size_t)m * n_codes;
    constexpr int NV = FAST ? 8 : 1;                                // 16-byte chunks per thread (FAST: 8 x 8 x 256 = 16384 codes)
    float dv[NV][8];
    float best = INFINITY;
    if (FAST) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int j0 = (c * 256 + threadIdx.x) * 8;
            const bf16x8 sc = *reinterpret_cast<const bf16x8*>(srow + j0);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(esq + j0), q1 = *reinterpret_cast<const f32x4*>(esq + j0 + 4);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                dv[c][k] = (zz - 2.0f * (float)sc[k]) + (k < 4 ? q0[k] : q1[k - 4]);
                best = fminf(best, dv[c][k]);
            }
        }
    } else {
        for (int j = threadIdx.x; j < n_codes; j += 256) best = fminf(best, (zz - 2.0f * (float)srow[j]) + esq[j]);
    }
    best = -wave_max(-best);
    if (lane == 0) sv[wv] = best;
    __syncthreads();
    if (threadIdx.x == 0) smin = fminf(fminf(sv[0], sv[1]), fminf(sv[2], sv[3]));
    __syncthreads();
    const float mu = (0.0078125f + 0.00390625f) * 1.015625f * sqrtf(zz) * emax[0];
    const float thr = smin + 2.0f * mu + 1e-30f;
    if (FAST) {
#pragma unroll
        for (int c = 0; c < NV; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (dv[c][k] <= thr) {
                    const unsigned pos = atomicAdd(&ncand, 1u);
                    if (pos < (unsigned)CB_CAP) cand[pos] = (c * 256 + threadIdx.x) * 8 + k;
                }
    } else {
        for (int j = threadIdx.x; j < n_codes; j += 256)
            if ((zz - 2.0f * (float)srow[j]) + esq[j] <= thr) {
                const unsigned pos = atomicAdd(&ncand, 1u);
                if (pos < (unsigned)CB_CAP) cand[pos] = j;
            }
    }
    __syncthreads();
    const bool all = ncand > (unsigned)CB_CAP;               // block-uniform
    const int n = all ? n_codes : (int)ncand;
    float bd = INFINITY;
    int bi = 0x7FFFFFFF;
    for (int c = threadIdx.x; c < n; c += 256) {
        const int j = all ? c : cand[c];
        const float* ej = e + (size_t)j * d;
        float s = 0.f;
        for (int k = 0; k < d; ++k) s = fmaf(zl[k], ej[k], s);          // exact fp32, k ascending (the exact GEMM's chain)
        const float dist = (zz - 2.0f * s) + esq[j];
        if (dist < bd || (dist == bd && j < bi)) { bd = dist; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(bd, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < bd || (ob == bd && oi < bi)) { bd = ob; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wv] = bd; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (sv[k] < bd || (sv[k] == bd && si[k] < bi)) { bd = sv[k]; bi = si[k]; }
        ids[m] = bi;
    }
}

// decode: out[m, :] = E[ids[m], :]  (F.embedding, vqgan.py:91), channels-last
template <typename T>
__global__ __launch_bounds__(256) void embedding_rows_kernel(const int64_t* ids, const float* e, T* out, long rows, int d, int n_codes) {
    const int d4 = d / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * d4; i += (long)gridDim.x * 256) {
        const long r = i / d4;
        const int c = (int)(i % d4) * 4;
        long id = ids[r];
        id = id < 0 ? 0 : (id >= n_codes ? n_codes - 1 : id);
        st4<T>(out + (size_t)r * d + c, *reinterpret_cast<const f32x4*>(e + (size_t)id * d + c));
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* src, TD* dst, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) st4<TD>(dst + i, ld4<TS>(src + i));
}

}  // namespace

static hipStream_t S(mebt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static bool is_f16(int dtype) { return dtype == MEBT_DTYPE_F16; }

extern "C" int mebt_op_conv3d(int32_t dtype, const mebt_conv3d_desc* d, int32_t allow_mfma, mebt_stream_t stream) {
    if (!d || !d->in || !d->out || !d->w) { mebt_set_error("conv3d: null pointer"); return MEBT_EINVAL; }
    if (dtype != MEBT_DTYPE_F32 && dtype != MEBT_DTYPE_F16) { mebt_set_error("conv3d: dtype must be f32 or f16"); return MEBT_EDTYPE; }
    const mebt_conv3d_desc& p = *d;
    if (p.ntaps < 1 || p.ntaps > MEBT_CONV_MAX_TAPS || p.Cin < 1 || p.Cout < 1) { mebt_set_error("conv3d: bad tap / channel count"); return MEBT_ESHAPE; }
    const long mcls = (long)p.B * p.cT * p.cH * p.cW;
    if (mcls <= 0) return MEBT_OK;
    const bool mfma = allow_mfma && is_f16(dtype) && p.in_mode == 0 && p.out_mode != 2 && p.Cin % CBK == 0 && p.Cout % CBN == 0;
    // thin layers (video boundary): Cout <= 4 with any Cin % 4 == 0 / NCDHW input, or NCDHW input with Cout <= 32
    const bool chan_ok = p.in_mode == 1 || p.Cin % 4 == 0;
    const int maxc = p.Cout <= 4 ? 4 : (p.Cout <= 32 && p.in_mode == 1 ? 32 : 0);
    const size_t vlds = (size_t)p.ntaps * p.Cin * maxc * 4;
    static const int conv_dma = [] { const char* e = getenv("MEBT_CONV_DMA"); return (e && e[0] == '0') ? 0 : 1; }();   // 0: the first (register-staged) kernel
    const bool small_in = (size_t)p.B * p.Ti * p.Hi * p.Wi * p.Cin < (1ull << 31);      // the thin MFMA kernels index the input with 32 bits
    static const int thin_mfma = [] { const char* e = getenv("MEBT_CONV_THIN_MFMA"); return (e && e[0] == '0') ? 0 : 1; }();   // 0: the 3-channel boundary layers on the vector ALUs (conv3d_voxel_kernel)
    if (mfma && conv_dma && p.Cin % 64 == 0 && p.Cout % 64 == 0 && (size_t)p.B * p.Ti * p.Hi * p.Wi * p.Cin * 2 < (1ull << 32)) {
        // tile: 128 x 128 (two workgroups per CU at ring depth 2) where Cout allows, else 256 x 64; MEBT_CONV_TILE=bm,bn,ring (experiments)
        static int force[3] = {0, 0, 0};
        static const int attrs = [] {
#define CONV_ATTR(BM_, BN_, ST_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_mfma_f16_dma_kernel<BM_, BN_, ST_>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_ * (BM_ + BN_) * 128)
            CONV_ATTR(128, 128, 2); CONV_ATTR(128, 128, 3); CONV_ATTR(128, 64, 2); CONV_ATTR(128, 64, 3); CONV_ATTR(256, 64, 2); CONV_ATTR(256, 64, 3);
            CONV_ATTR(256, 128, 2);
#undef CONV_ATTR
            if (const char* e = getenv("MEBT_CONV_TILE")) sscanf(e, "%d,%d,%d", &force[0], &force[1], &force[2]);
            return 0;
        }();
        (void)attrs;
        int bm = p.Cout % 128 == 0 ? 128 : 256, bn = p.Cout % 128 == 0 ? 128 : 64, ring = 2;
        if (force[0] && p.Cout % force[1] == 0) { bm = force[0]; bn = force[1]; ring = force[2]; }
        const dim3 grid((unsigned)((mcls + bm - 1) / bm), p.Cout / bn);
#define CONV_GO(BM_, BN_, ST_) hipLaunchKernelGGL((conv3d_mfma_f16_dma_kernel<BM_, BN_, ST_>), grid, dim3(256), ST_ * (BM_ + BN_) * 128, S(stream), p)
        if (bm == 128 && bn == 128 && ring == 2) CONV_GO(128, 128, 2);
        else if (bm == 128 && bn == 128) CONV_GO(128, 128, 3);
        else if (bm == 128 && bn == 64 && ring == 2) CONV_GO(128, 64, 2);
        else if (bm == 128 && bn == 64) CONV_GO(128, 64, 3);
        else if (bm == 256 && bn == 64 && ring == 2) CONV_GO(256, 64, 2);
        else if (bm == 256 && bn == 64) CONV_GO(256, 64, 3);
        else CONV_GO(256, 128, 2);
#undef CONV_GO
    } else if (mfma) {
        const dim3 grid((unsigned)((mcls + CBM - 1) / CBM), p.Cout / CBN);
        hipLaunchKernelGGL(conv3d_mfma_f16_kernel, grid, dim3(256), 0, S(stream), p);
    } else if (allow_mfma && thin_mfma && small_in && is_f16(dtype) && p.in_mode == 1 && p.Cin == 3 && p.out_mode == 0 && !p.resid && p.Cout == 32 && p.ntaps * 3 > 64 &&
               p.ntaps * 3 <= 96) {          // the TATS first layer: 3 -> 32 channels, 27 taps (K = 81 in three k-steps, two column fragments)
        const long nblk = (mcls + 15) / 16;
        const unsigned wgs = (unsigned)((nblk + 3) / 4 < 2048 ? (nblk + 3) / 4 : 2048);
        hipLaunchKernelGGL((conv3d_first_mfma_kernel<3, 3, 2>), dim3(wgs), dim3(256), 0, S(stream), p);
    } else if (allow_mfma && thin_mfma && small_in && is_f16(dtype) && p.in_mode == 0 && (p.Cin == 32 || p.Cin == 64 || p.Cin == 128) && p.Cout <= 16 && (size_t)p.ntaps * (p.Cin / 32) * 1024 <= 96 * 1024) {
        const size_t lds = (size_t)p.ntaps * (p.Cin / 32) * 1024;
        static bool attr = false;
        if (!attr) {
            MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_last_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_last_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_last_mfma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr = true;
        }
        const long nblk = (mcls + 63) / 64;
        const unsigned wgs = (unsigned)((nblk + 3) / 4 < 1024 ? (nblk + 3) / 4 : 1024);
        if (p.Cin == 32) hipLaunchKernelGGL(conv3d_last_mfma_kernel<1>, dim3(wgs), dim3(256), lds, S(stream), p);
        else if (p.Cin == 64) hipLaunchKernelGGL(conv3d_last_mfma_kernel<2>, dim3(wgs), dim3(256), lds, S(stream), p);
        else hipLaunchKernelGGL(conv3d_last_mfma_kernel<4>, dim3(wgs), dim3(256), lds, S(stream), p);
    } else if (allow_mfma && maxc && chan_ok && vlds <= 64 * 1024) {
        const dim3 grid((unsigned)((mcls + 255) / 256));
#define VOX(T_, C_) hipLaunchKernelGGL((conv3d_voxel_kernel<T_, C_>), grid, dim3(256), vlds, S(stream), p)
        if (is_f16(dtype)) { if (maxc == 4) VOX(f16_t, 4); else VOX(f16_t, 32); }
        else { if (maxc == 4) VOX(float, 4); else VOX(float, 32); }
#undef VOX
    } else {
        const long total = mcls * p.Cout;
        const dim3 grid((unsigned)((total + 255) / 256));
        if (is_f16(dtype)) hipLaunchKernelGGL(conv3d_direct_kernel<f16_t>, grid, dim3(256), 0, S(stream), p);
        else hipLaunchKernelGGL(conv3d_direct_kernel<float>, grid, dim3(256), 0, S(stream), p);
    }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

extern "C" int mebt_op_groupnorm_silu(int32_t dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                      int32_t B, int64_t vox_per_sample, int32_t C, mebt_stream_t stream) {
    if (!x || !y || !gamma || !beta || !stats) { mebt_set_error("groupnorm: null pointer"); return MEBT_EINVAL; }
    if (C % 32 || C > 1024 || (C / 4) > 256) { mebt_set_error("groupnorm: channels must be a multiple of 32 (32 groups) and <= 1024"); return MEBT_ESHAPE; }
    if (dtype != MEBT_DTYPE_F32 && dtype != MEBT_DTYPE_F16) { mebt_set_error("groupnorm: dtype must be f32 or f16"); return MEBT_EDTYPE; }
    if (B <= 0 || vox_per_sample <= 0) return MEBT_OK;
    MEBT_HIP_CHECK(hipMemsetAsync(stats, 0, (size_t)B * 32 * 2 * 4, S(stream)));
    int rpb = 64;
    while ((vox_per_sample + rpb - 1) / rpb > 512) rpb *= 2;               // <= 512 adders per statistic
    const dim3 grid((unsigned)((vox_per_sample + rpb - 1) / rpb), B);
    const long total4 = (long)B * vox_per_sample * C / 4;
    const unsigned ablocks = (unsigned)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    if (is_f16(dtype)) {
        hipLaunchKernelGGL(gn_stats_kernel<f16_t>, grid, dim3(256), 0, S(stream), (const f16_t*)x, stats, (long)vox_per_sample, C, rpb);
        hipLaunchKernelGGL(gn_apply_silu_kernel<f16_t>, dim3(ablocks), dim3(256), 0, S(stream), (const f16_t*)x, (f16_t*)y, stats, gamma, beta, (long)vox_per_sample, C, total4);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, S(stream), (const float*)x, stats, (long)vox_per_sample, C, rpb);
        hipLaunchKernelGGL(gn_apply_silu_kernel<float>, dim3(ablocks), dim3(256), 0, S(stream), (const float*)x, (float*)y, stats, gamma, beta, (long)vox_per_sample, C, total4);
    }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

// ids[m] = nearest codebook entry of z[m, :] (fp32 [M, d]).  `score` is caller-provided scratch [M, n_codes] fp32 for
// z E^T (exact-fp32 MFMA: the search is never run at reduced precision), `esq` scratch [n_codes].
extern "C" int mebt_op_codebook_argmin(const float* z, const float* embeddings, float* score, float* esq, int64_t* ids, int32_t M,
                                       int32_t n_codes, int32_t d, mebt_stream_t stream) {
    if (!z || !embeddings || !score || !esq || !ids) { mebt_set_error("codebook: null pointer"); return MEBT_EINVAL; }
    if (M <= 0) return MEBT_OK;
    if (d % 16 || n_codes % 4) { mebt_set_error("codebook: embedding_dim must be a multiple of 16 and n_codes of 4"); return MEBT_ESHAPE; }
    if (int rc = gemm_init_attributes()) return rc;
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = z; g.B = embeddings; g.C = score; g.M = M; g.N = n_codes; g.K = d; g.lda = d; g.ldb = d; g.ldc = n_codes;
    g.a_kc = 1; g.b_kc = 1; g.c_f32 = 1; g.split_k = 1;
    if (int rc = launch_gemm(g, MEBT_F32, S(stream))) return rc;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n_codes + 3) / 4), dim3(256), 0, S(stream), embeddings, esq, n_codes, d);
    hipLaunchKernelGGL(codebook_argmin_kernel, dim3(M), dim3(256), 0, S(stream), score, z, esq, ids, n_codes, d);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

// The same ids through the filtered search (see codebook_filter_kernel): approximate scores on the bf16 MFMA GEMM, exact fp32
// re-evaluation of the candidates that can be the arg-min.  lowp: scratch of (M + n_codes) * d bf16 values (the rounded copies of
// z and of the embeddings; the library allocates nothing); esq: scratch [n_codes + 1].  d a multiple of 64.
extern "C" int mebt_op_codebook_argmin_filtered(const float* z, const float* embeddings, void* score, float* esq, void* lowp, int64_t* ids,
                                                int32_t M, int32_t n_codes, int32_t d, mebt_stream_t stream) {
    if (!z || !embeddings || !score || !esq || !lowp || !ids) { mebt_set_error("codebook: null pointer"); return MEBT_EINVAL; }
    if (M <= 0) return MEBT_OK;
    if (d % 64 || n_codes % 8 || d > 4096) { mebt_set_error("codebook (filtered): embedding_dim must be a multiple of 64 (<= 4096) and n_codes of 8"); return MEBT_ESHAPE; }
    if (int rc = gemm_init_attributes()) return rc;
    char* zb = reinterpret_cast<char*>(lowp);
    char* eb = zb + (size_t)M * d * 2;
    if (int rc = launch_cast_f32_to_bf16(z, zb, (size_t)M * d, S(stream))) return rc;
    if (int rc = launch_cast_f32_to_bf16(embeddings, eb, (size_t)n_codes * d, S(stream))) return rc;
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = zb; g.B = eb; g.C = score; g.M = M; g.N = n_codes; g.K = d; g.lda = d; g.ldb = d; g.ldc = n_codes;
    g.a_kc = 1; g.b_kc = 1; g.c_f32 = 0; g.split_k = 1;        // bf16 scores: the first half of the caller's scratch
    if (int rc = launch_gemm(g, MEBT_BF16, S(stream))) return rc;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n_codes + 3) / 4), dim3(256), 0, S(stream), embeddings, esq, n_codes, d);
    hipLaunchKernelGGL(vec_max_sqrt_kernel, dim3(1), dim3(256), 0, S(stream), esq, esq + n_codes, n_codes);
    const bf16_t* sb = reinterpret_cast<const bf16_t*>(score);
    if (n_codes == 16384) hipLaunchKernelGGL(codebook_filter_kernel<true>, dim3(M), dim3(256), (size_t)d * 4 + CB_CAP * 4, S(stream), sb, z, embeddings, esq, esq + n_codes, ids, n_codes, d);
    else hipLaunchKernelGGL(codebook_filter_kernel<false>, dim3(M), dim3(256), (size_t)d * 4 + CB_CAP * 4, S(stream), sb, z, embeddings, esq, esq + n_codes, ids, n_codes, d);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

extern "C" int mebt_op_embedding_rows(int32_t dtype, const int64_t* ids, const float* embeddings, void* out, int64_t rows, int32_t d,
                                      int32_t n_codes, mebt_stream_t stream) {
    if (!ids || !embeddings || !out) { mebt_set_error("embedding_rows: null pointer"); return MEBT_EINVAL; }
    if (d % 4) { mebt_set_error("embedding_rows: d must be a multiple of 4"); return MEBT_ESHAPE; }
    if (rows <= 0) return MEBT_OK;
    const long n4 = rows * (d / 4);
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (is_f16(dtype)) hipLaunchKernelGGL(embedding_rows_kernel<f16_t>, dim3(blocks), dim3(256), 0, S(stream), ids, embeddings, (f16_t*)out, (long)rows, d, n_codes);
    else if (dtype == MEBT_DTYPE_F32) hipLaunchKernelGGL(embedding_rows_kernel<float>, dim3(blocks), dim3(256), 0, S(stream), ids, embeddings, (float*)out, (long)rows, d, n_codes);
    else { mebt_set_error("embedding_rows: dtype must be f32 or f16"); return MEBT_EDTYPE; }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

// fp32 -> fp16 of a flat buffer (weights of the fast mode), n % 4 == 0
extern "C" int mebt_op_cast_f16(const float* src, void* dst, int64_t n, mebt_stream_t stream) {
    if (!src || !dst || n < 0 || (n % 4)) { mebt_set_error("cast_f16: bad arguments"); return MEBT_EINVAL; }
    if (!n) return MEBT_OK;
    const size_t blocks = ((size_t)n / 4 + 255) / 256;
    hipLaunchKernelGGL((cast_kernel<float, f16_t>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, S(stream), src, (f16_t*)dst, (size_t)n);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}
