// Shared device helpers for the MeBT gfx950 kernels (CDNA4, wave64).  HIP only — no CUDA paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define MEBT_WAVE 64

// status codes returned through the C ABI (include/mebt_hip.h)
enum { MEBT_OK = 0, MEBT_EINVAL = 1, MEBT_ESHAPE = 2, MEBT_EHIP = 3, MEBT_EWORKSPACE = 4, MEBT_EDTYPE = 5 };

#define MEBT_HIP_CHECK(expr)                                   \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) { mebt_set_hip_error(_e, #expr); (void)hipGetLastError(); return MEBT_EHIP; } \
    } while (0)
void mebt_set_hip_error(hipError_t e, const char* what);
void mebt_set_error(const char* msg);

// ---------------------------------------------------------------------------------------------
// buffer resources: out-of-range loads return 0 and out-of-range stores are dropped by hardware,
// which is how every ragged tile edge (NC, NT are arbitrary) is handled without branches.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    const uint32_t n = bytes > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
#define MEBT_OOB 0x7FFFFFFF   // voffset guaranteed >= num_records

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// ---------------------------------------------------------------------------------------------
// direct global -> LDS copies (LDS-DMA)
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
// One LDS-DMA instruction as inline asm: hipcc models the builtin form as a store to LDS and then drains
// the whole DMA queue (s_waitcnt vmcnt(0)) in front of every ds_read_b64_tr_b16, which defeats the ring
// for the transposed-operand layouts.  In asm the compiler sees no memory operation; completion is
// tracked by the counted waits of the kernel.  M0 (the LDS base of the DMA) is saved and restored inside
// the statement because the compiler owns it.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(rsrc)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(lds_char_ptr)(char*)p; }

// ---------------------------------------------------------------------------------------------
// element conversion
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements <-> 4 floats (16-B or 8-B vector access)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions (butterfly over all 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-erf GELU (nn.GELU default, reference gpt.py:152) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Cheap erf for the bf16 epilogues: Abramowitz-Stegun 7.1.27, |error| <= 5e-4 (bf16 resolves 4e-3
// relative), 4 FMAs + one reciprocal, no exponential.  The fp32 parity kernels keep erff.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    float d = 1.0f + ax * (0.278393f + ax * (0.230389f + ax * (0.000972f + ax * 0.078108f)));
    d *= d; d *= d;
    return copysignf(1.0f - __frcp_rn(d), x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);   // exp(-x^2/2)
}

// ---------------------------------------------------------------------------------------------
// AdamW (decoupled weight decay, torch.optim.AdamW semantics): one definition for the streaming kernel and
// for the weight-gradient epilogue that applies the update in place (optimizer-in-backward)
// ---------------------------------------------------------------------------------------------
struct AdamWHyper { float lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale; };
__device__ __forceinline__ void adamw_update4(f32x4& p, const f32x4& g, f32x4& m, f32x4& v, const AdamWHyper& a) {
    const float step_size = a.lr / a.bc1;
    const float inv_sqrt_bc2 = rsqrtf(a.bc2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float gj = g[j] * a.grad_scale;
        p[j] *= 1.0f - a.lr * a.weight_decay;
        m[j] = a.beta1 * m[j] + (1.0f - a.beta1) * gj;
        v[j] = a.beta2 * v[j] + (1.0f - a.beta2) * gj * gj;
        const float denom = sqrtf(v[j]) * inv_sqrt_bc2 + a.eps;
        p[j] -= step_size * (m[j] / denom);
    }
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG for dropout (Philox-like mixing of a 64-bit counter; one 32-bit draw per call).
// The mask of element `idx` at site `site` of step `seed` is recomputed in backward, never stored.
// ---------------------------------------------------------------------------------------------
// One 32-bit mix (two multiplies, the costly part: v_mul_lo_u32 is quarter rate) yields two 16-bit draws:
// elements idx and idx^1 share a hash.  `pair_hash` takes the pair index (idx >> 1).
__device__ __forceinline__ uint32_t pair_hash(uint64_t seed, uint32_t site, uint64_t pair) {
    uint32_t h = ((uint32_t)pair ^ (uint32_t)seed) + ((uint32_t)(pair >> 32) + site) * 0x632BE5ABu + (uint32_t)(seed >> 32);
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}
__device__ __forceinline__ uint32_t mix_hash(uint64_t seed, uint32_t site, uint64_t idx) {
    const uint32_t h = pair_hash(seed, site, idx >> 1);
    return (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
}
// Dropout descriptor: element `idx` of site `site` is kept iff its 16-bit draw >= thresh
// (thresh = round(p * 65536)); kept values are scaled by inv_keep = 65536 / (65536 - thresh).
// thresh == 0 disables the site.
struct DropCfg {
    uint64_t seed;
    uint32_t site;
    uint32_t thresh;
    float inv_keep;
    uint32_t small;     // set by the launcher when the site has fewer than 2^33 elements: every pair index fits 32 bits
    uint32_t base32;    // site * 0x632BE5AB + (seed >> 32): the wave-uniform part of pair_hash when the pair's high word is 0
};
// pair_hash(d.seed, d.site, pair) with one multiply less when the launcher vouches for pair < 2^32 (identical value:
// with a zero high word the first product is the uniform site * 0x632BE5AB, folded into base32 on the host)
__device__ __forceinline__ uint32_t pair_hash_cfg(const DropCfg& d, uint64_t pair) {
    if (!d.small) return pair_hash(d.seed, d.site, pair);
    uint32_t h = ((uint32_t)pair ^ (uint32_t)d.seed) + d.base32;
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}
__device__ __forceinline__ float drop_keep(const DropCfg& d, uint64_t idx) {
    const uint32_t h = pair_hash_cfg(d, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xFFFFu)) >= d.thresh ? d.inv_keep : 0.0f;
}
// keep factors of 4 consecutive elements idx0 .. idx0+3: two hashes when idx0 is even (the usual case:
// vectors start at multiples of 4), identical values to drop_keep element by element
__device__ __forceinline__ f32x4 drop_keep4(const DropCfg& d, uint64_t idx0) {
    f32x4 k;
    if ((idx0 & 1) == 0) {
        const uint32_t h0 = pair_hash_cfg(d, idx0 >> 1), h1 = pair_hash_cfg(d, (idx0 >> 1) + 1);
        k[0] = (h0 & 0xFFFFu) >= d.thresh ? d.inv_keep : 0.0f;
        k[1] = (h0 >> 16) >= d.thresh ? d.inv_keep : 0.0f;
        k[2] = (h1 & 0xFFFFu) >= d.thresh ? d.inv_keep : 0.0f;
        k[3] = (h1 >> 16) >= d.thresh ? d.inv_keep : 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = drop_keep(d, idx0 + j);
    }
    return k;
}
