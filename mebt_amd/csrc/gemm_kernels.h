// MFMA GEMM kernels for gfx950 (device code + per-layout launch helpers; included by gemm.hip and the four
// per-layout translation units gemm_kk/kr/rr/rk.hip so that the instantiations compile in parallel):  C[M,N] = sum_k A(m,k) * B(n,k)  (+ fused epilogue)
//
// Replaces the implicit ATen dispatches behind nn.Linear forward/backward on the MeBT hot path
// (reference mebt/modules/gpt.py:126-128,140,150-155,248 and their autograd backward).
//
// Operand layouts.  Each operand is either
//   KC  "k-contiguous":   stored [rows][K]  (K fastest)  -> LDS image [row][64 k], ds_read_b128
//   RC  "row-contiguous": stored [K][rows]  (row fastest)-> LDS image [k][128 rows],
//                                                           ds_read_b64_tr_b16 (hardware transpose)
// so the three products of a Linear layer need no transposed copies in HBM:
//   forward  Y  = X  W^T : A = X  (KC)  B = W  (KC)
//   dgrad    dX = dY W   : A = dY (KC)  B = W  (RC, stored [n_out][k_in] = [K][rows])
//   wgrad    dW = dY^T X : A = dY (RC)  B = X  (RC)      (reduction over tokens)
//
// bf16 kernels: block tiles 192x128 ... 64x64 x 64 (k), 4 waves (2x2), v_mfma_f32_16x16x32_bf16 with the
// operands swapped (D^T = B A^T) so that each lane owns 4 consecutive output columns (vector epilogue
// loads/stores).  Global->LDS staging is LDS-DMA into a ring of 2-4 k-tiles with counted waits and a raw
// s_barrier (gemm_tile_dma); a register-staged two-stage variant (gemm_tile_regstaged) is kept as the
// reference implementation of the same LDS images.  Variants on top: two pipelines per workgroup on
// alternate k-tiles, two independent products per launch (pair), the grouped weight gradients of a block
// (optionally with AdamW in the epilogue), split-K into fp32 slabs + a reduce/epilogue kernel.  The host
// picks tile / ring depth / variant per GEMM signature by timing them in situ (autotune_config).
// Ragged edges are handled by buffer-resource bounds (OOB loads return 0).
//
// f32 kernel (parity mode): same tiling on v_mfma_f32_32x32x2_f32, which is an exact fp32 FMA
// chain (no TF32-like truncation exists on gfx950).
#pragma once
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// ------------------------------------------------------------------------------------------------
// epilogue shared by both kernels.  v = 4 consecutive output columns n..n+3 of row m.
// ------------------------------------------------------------------------------------------------
template <typename TA /*activation dtype of aux / C2*/>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, f32x4 v, bool add_bias, bool atomic) {
    if (add_bias && p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v += b;
    }
    const size_t ci = (size_t)m * p.ldc + n;
    if (atomic) {
        float* c = reinterpret_cast<float*>(p.C) + ci;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(c + j, v[j]);
        return;
    }
    constexpr bool fast = sizeof(TA) == 2;       // bf16 kernel: cheap erf (1.5e-7 abs error)
    if (p.epilogue == EPI_GELU) {
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = fast ? gelu_fast(v[j]) : gelu_f(v[j]);
        store4<TA>(reinterpret_cast<TA*>(p.C2) + ci, g);
    } else if (p.epilogue == EPI_RESID) {
        if (p.drop.thresh) {
            const uint64_t e0 = (uint64_t)m * p.N + n;
            v *= drop_keep4(p.drop, e0);
        }
        v += load4<TA>(reinterpret_cast<const TA*>(p.aux) + (size_t)m * p.ld_aux + n);
    } else if (p.epilogue == EPI_GELU_BWD) {
        const f32x4 x = load4<TA>(reinterpret_cast<const TA*>(p.aux) + (size_t)m * p.ld_aux + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= fast ? gelu_grad_fast(x[j]) : gelu_grad_f(x[j]);
    }
    if (!p.C) return;                   // EPI_GELU in inference: only gelu(C) is needed
    if (p.c_f32) {
        float* c = reinterpret_cast<float*>(p.C) + ci;
        if (p.beta) v += *reinterpret_cast<const f32x4*>(c);
        *reinterpret_cast<f32x4*>(c) = v;
    } else {
        store4<TA>(reinterpret_cast<TA*>(p.C) + ci, v);
    }
}

// Epilogue through LDS.  The MFMA accumulators hold, per lane, 4 consecutive columns of ONE row of each
// 16x16 fragment, so a direct store instruction touches 16 rows x 32 bytes — a quarter of every 128-byte
// line for the C store and for the aux / residual loads.  After the k-loop the LDS ring is idle: each
// wave parks its (TM*16) x (TN*16) fp32 tile there (padded rows: conflict-free ds_write_b128), then walks
// it row-contiguously, 4 columns per lane, so every global access of the epilogue covers whole rows of the
// wave's tile (TN*32 bytes of bf16 per row).
//
// Every operand the epilogue reads (bias: one f32x4 per lane, the same for all rows; aux: one 8-byte
// vector per row pass) is fetched BEFORE the accumulators are staged, all loads back to back at clamped
// (always valid) addresses: a load under a per-row bounds/epilogue branch makes hipcc wait for it inside
// the branch, which turned the 16 row passes of a wave into 16-32 dependent L2/HBM round trips.
// EPI_ADAMW: the tile is a weight gradient; apply AdamW to the parameter tile in place (p, m, v fp32 + bf16 mirror)
// instead of storing it.  Straight-line per 16-row slab, software-pipelined over the slabs: the 12 loads of slab
// i + 1 are in flight while slab i is staged through LDS, updated and stored (a tile's epilogue is one exposed HBM
// round trip instead of one per slab).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_adamw(const GemmParams& p, const f32x4 (&acc)[TM][TN], char* smem, int wave, int lane,
                                               int mbase, int nbase) {
    constexpr int COLS = TN * 16, LD = COLS + 4;
    float* st = reinterpret_cast<float*>(smem) + wave * (16 * LD);
    constexpr int LPR = COLS / 4, RPI = 64 / LPR, NPASS = 16 / RPI;
    const int r0 = lane / LPR, c4 = (lane % LPR) * 4;
    const int n = nbase + c4;
    const bool n_ok = n < p.N;
    const int nc = n_ok ? n : 0;
    f32x4 pp[2][NPASS], mm[2][NPASS], vv[2][NPASS];
    auto prefetch = [&](int i, int buf) {
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int m = min(mbase + i * 16 + r * RPI + r0, p.M - 1);
            const size_t ci = (size_t)m * p.ldc + nc;
            pp[buf][r] = *reinterpret_cast<const f32x4*>(p.opt_p + ci);
            mm[buf][r] = *reinterpret_cast<const f32x4*>(p.opt_m + ci);
            vv[buf][r] = *reinterpret_cast<const f32x4*>(p.opt_v + ci);
        }
    };
    prefetch(0, 0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) prefetch(i + 1, (i + 1) & 1);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(st + (lane & 15) * LD + j * 16 + 4 * (lane >> 4)) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int row = r * RPI + r0;
            const f32x4 g = *reinterpret_cast<const f32x4*>(st + row * LD + c4);
            adamw_update4(pp[i & 1][r], g, mm[i & 1][r], vv[i & 1][r], p.opt);
            const int m = mbase + i * 16 + row;
            if (m < p.M && n_ok) {
                const size_t ci = (size_t)m * p.ldc + n;
                *reinterpret_cast<f32x4*>(p.opt_p + ci) = pp[i & 1][r];
                *reinterpret_cast<f32x4*>(p.opt_m + ci) = mm[i & 1][r];
                *reinterpret_cast<f32x4*>(p.opt_v + ci) = vv[i & 1][r];
                if (p.opt_lp) store4<bf16_t>(reinterpret_cast<bf16_t*>(p.opt_lp) + ci, pp[i & 1][r]);
            }
        }
    }
}

// Stores of the plain epilogues.  Every lane owns EIGHT consecutive columns of a row, so that a bf16 output leaves as one 16-byte
// store per lane: the store tail of a tile is bound by store INSTRUCTIONS, not bytes (8-byte stores: ≈7 B/clk/CU — 128 KiB of a
// 256 x 256 bf16 tile took ≈9 us, as long as its whole K = 1024 main loop; profiles/r03_pp_bench_components.txt), and 16-byte stores
// halve the count.  `wide` = the row pitch and the column keep 16-byte alignment; otherwise (N or ldc not a multiple of 8) the two
// 4-column halves go out separately, each only if its columns exist.
template <typename T>
__device__ __forceinline__ void store8(T* c, const f32x4& lo, const f32x4& hi, bool ok_hi, bool wide) {
    if constexpr (sizeof(T) == 2) {
        if (wide && ok_hi) {
            const bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
            *reinterpret_cast<bf16x8*>(c) = o;
            return;
        }
    }
    store4<T>(c, lo);
    if (ok_hi) store4<T>(c + 4, hi);
}

// SWZ: the slab rows are 64 floats without padding, the 16-byte chunk index XORed with the row (exactly 4 KiB per wave: the
// persistent 256 x 256 kernel has 32 KiB of LDS left beside its ring); `pre_bias`: bias values the caller loaded earlier.
template <int TM, int TN, bool SYNC = true, bool SWZ = false>
__device__ __forceinline__ void epilogue_via_lds(const GemmParams& p, const f32x4 (&acc)[TM][TN], char* smem, int wave, int lane,
                                                 int mbase, int nbase, bool add_bias, bool atomic, const f32x4* pre_bias = nullptr) {
    typedef bf16_t TA;
    if (!SWZ && p.epilogue == EPI_ADAMW) { epilogue_adamw<TM, TN>(p, acc, smem, wave, lane, mbase, nbase); return; }
    constexpr int COLS = TN * 16, LD = SWZ ? COLS : COLS + 4;          // fp32 elements per staged row
    static_assert(!SWZ || COLS == 64, "the swizzled slab is 16 chunks wide");
    float* st = reinterpret_cast<float*>(smem) + wave * (16 * LD);   // one 16-row slab per wave (<= 4.3 KiB)
    constexpr int LPR = COLS / 8;                         // lanes per row (8 columns each)
    constexpr int RPI = 64 / LPR;                         // rows per pass
    constexpr int NPASS = 16 / RPI;
    static_assert(NPASS >= 1, "wave tiles are at least 32 columns wide");
    const int r0 = lane / LPR, c8 = (lane % LPR) * 8;
    const int n = nbase + c8;
    const bool n_ok = n < p.N, hi_ok = n + 4 < p.N;
    const bool wide = !p.narrow_store && (p.ldc & 7) == 0 && (p.ld_aux & 7) == 0 && (((size_t)p.C | (size_t)p.C2 | (size_t)p.aux) & 15) == 0;
    const int nc = n_ok ? n : 0, nh = hi_ok ? n + 4 : nc;
    const bool need_aux = p.epilogue == EPI_RESID || p.epilogue == EPI_GELU_BWD;
    f32x4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f};
    if (pre_bias) { bias_lo = pre_bias[0]; bias_hi = pre_bias[1]; }
    else if (add_bias && p.bias) {
        bias_lo = *reinterpret_cast<const f32x4*>(p.bias + nc);
        bias_hi = *reinterpret_cast<const f32x4*>(p.bias + nh);
    }
    bf16x4 aux_lo[TM][NPASS], aux_hi[TM][NPASS];
    if (need_aux) {
        const TA* aux = reinterpret_cast<const TA*>(p.aux);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                const int m = min(mbase + i * 16 + r * RPI + r0, p.M - 1);
                if (wide && hi_ok) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(aux + (size_t)m * p.ld_aux + nc);
                    aux_lo[i][r] = bf16x4{a[0], a[1], a[2], a[3]};
                    aux_hi[i][r] = bf16x4{a[4], a[5], a[6], a[7]};
                } else {
                    aux_lo[i][r] = *reinterpret_cast<const bf16x4*>(aux + (size_t)m * p.ld_aux + nc);
                    aux_hi[i][r] = *reinterpret_cast<const bf16x4*>(aux + (size_t)m * p.ld_aux + nh);
                }
            }
    }
    if (SYNC) __syncthreads();                            // every wave is done reading the last k-tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(st + (lane & 15) * LD + (SWZ ? (((j * 4 + (lane >> 4)) ^ (lane & 15)) << 2) : j * 16 + 4 * (lane >> 4))) = acc[i][j];
        // the same wave reads back what it wrote (LDS operations of a wave execute in order): only its LDS
        // queue has to drain, no workgroup barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int row = r * RPI + r0;
            f32x4 lo = *reinterpret_cast<const f32x4*>(st + row * LD + (SWZ ? ((((c8 >> 2)) ^ (row & 15)) << 2) : c8));
            f32x4 hi = *reinterpret_cast<const f32x4*>(st + row * LD + (SWZ ? ((((c8 >> 2) + 1) ^ (row & 15)) << 2) : c8 + 4));
            const int m = mbase + i * 16 + row;
            if (!(m < p.M && n_ok)) continue;
            lo += bias_lo;
            hi += bias_hi;
            const size_t ci = (size_t)m * p.ldc + n;
            if (atomic) {
                float* c = reinterpret_cast<float*>(p.C) + ci;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(c + j, lo[j]);
                if (hi_ok)
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(c + 4 + j, hi[j]);
                continue;
            }
            if (p.epilogue == EPI_GELU) {
                f32x4 glo, ghi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { glo[j] = gelu_fast(lo[j]); ghi[j] = gelu_fast(hi[j]); }
                store8<TA>(reinterpret_cast<TA*>(p.C2) + ci, glo, ghi, hi_ok, wide);
            } else if (p.epilogue == EPI_RESID) {
                if (p.drop.thresh) {
                    const uint64_t e0 = (uint64_t)m * p.N + n;
                    lo *= drop_keep4(p.drop, e0);
                    hi *= drop_keep4(p.drop, e0 + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] += (float)aux_lo[i][r][j]; hi[j] += (float)aux_hi[i][r][j]; }
            } else if (p.epilogue == EPI_GELU_BWD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] *= gelu_grad_fast((float)aux_lo[i][r][j]); hi[j] *= gelu_grad_fast((float)aux_hi[i][r][j]); }
            }
            if (!p.C) continue;                 // EPI_GELU in inference: only gelu(C) is needed
            if (p.c_f32) {
                float* c = reinterpret_cast<float*>(p.C) + ci;
                if (p.beta) {
                    lo += *reinterpret_cast<const f32x4*>(c);
                    if (hi_ok) hi += *reinterpret_cast<const f32x4*>(c + 4);
                }
                store8<float>(c, lo, hi, hi_ok, wide);
            } else {
                store8<TA>(reinterpret_cast<TA*>(p.C) + ci, lo, hi, hi_ok, wide);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 kernel, templated on the block tile (TBM x TBN in {128,64}) so that small outputs still fill
// 256 CUs: 4 waves as 2x2, each wave (TBM/2) x (TBN/2) = (TBM/32) x (TBN/32) MFMA tiles of 16x16.
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;

// swizzle of the 16-byte chunk index of an RC image row (k-row `krow`): conflict-free transposed reads
template <int ROWS> __device__ __forceinline__ int rc_swizzle(int krow);
template <> __device__ __forceinline__ int rc_swizzle<128>(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }          // 256-B rows
template <> __device__ __forceinline__ int rc_swizzle<64>(int krow) { return (((krow >> 1) & 1) << 1) | (((krow >> 3) & 1) << 2); }   // 128-B rows
// 192-B rows: k-rows q = 0..3 of a 16-lane group already land 64 B apart; rows r and r+8 (the two groups of a
// 32-lane half) coincide mod 256 B -> move the second group to the neighbouring 32-B pair
template <> __device__ __forceinline__ int rc_swizzle<96>(int krow) { return ((krow >> 3) & 1) << 1; }
// 384-B rows: rows q and q+2 coincide mod 256 B, and so do r and r+8
// 512-B rows: every k-row starts on the same bank -> the 8 k-rows of a 32-lane half (q = 0..3 of two groups) get 8 different 32-B pairs
template <> __device__ __forceinline__ int rc_swizzle<256>(int krow) { return ((krow & 3) << 1) | (((krow >> 3) & 1) << 3); }
template <> __device__ __forceinline__ int rc_swizzle<192>(int krow) { return (((krow >> 1) & 1) << 1) | (((krow >> 3) & 1) << 2); }

// XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (consecutive linear ids on
// consecutive XCDs) and each XCD has its own 4 MB L2, so the tiles an XCD works on at the same time should
// share operand rows: XCD x gets a COMPACT xr x xc sub-grid of the tile grid (xr * xc = 8), the shape chosen
// to minimise the bytes all XCDs fetch together, xc * |A| + xr * |B| with |A| ~ M and |B| ~ N (both x K).
// Measured with rocprofv3 FETCH_SIZE before this mapping: 3-6x the algorithmic operand bytes (the grouped
// weight-gradient launch streamed 357 MB for 62 MB of operands at 4.3 TB/s — bandwidth-bound on re-reads).
// `t` = linear workgroup index within the product; any t with equal t % 8 share an XCD.
__device__ __forceinline__ void xcd_tile(int t, int ntx, int nty, int M, int N, int& tr, int& tc) {
    const int T = ntx * nty;
    if (T & 7) {                              // no whole sub-grids: contiguous runs along N (bijective for any T)
        const int q = T >> 3, r = T & 7, xcd = t & 7, idx = t >> 3;
        const int b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tr = b / ntx; tc = b % ntx;
        return;
    }
    int xr = 0, xc = 0;
    long best = 0x7FFFFFFFFFFFFFFFl;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = 8 >> s, c = 1 << s;     // (8,1) (4,2) (2,4) (1,8)
        if (nty % r || ntx % c) continue;
        const long cost = (long)c * M + (long)r * N;
        if (cost < best) { best = cost; xr = r; xc = c; }
    }
    if (!xr) {                                // T % 8 == 0 but neither dimension splits (e.g. 3 x 8 ... handled above); fall back
        tr = t / ntx; tc = t % ntx;
        return;
    }
    const int xcd = t & 7, idx = t >> 3;
    const int lr = nty / xr, lc = ntx / xc;
    tr = (xcd / xc) * lr + idx / lc;
    tc = (xcd % xc) * lc + idx % lc;
}

// Weight prefetch for the next product of the chain (GemmParams::pf): thread t of workgroup w touches line (w * T + t) and
// line (W * T + w * T + t) — one dword per 128-byte line, 2 VGPRs, issued before the workgroup's own loads so that the
// counted waits of the main loop (which only assume "older operations finish first") are unaffected.  The values are only
// consumed behind `pf_sink` (null at run time).
struct PrefetchRegs { uint32_t v[6]; };
__device__ __forceinline__ PrefetchRegs prefetch_next(const GemmParams& p, int wg, int nwg, int nthreads) {
    PrefetchRegs r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.v[i] = 0;
    if (p.pf) {
        const size_t nlines = p.pf_bytes >> 7;
        const char* base = reinterpret_cast<const char*>(p.pf);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t l = (size_t)i * nwg * nthreads + (size_t)wg * nthreads + threadIdx.x;
            if (l < nlines) r.v[i] = *reinterpret_cast<const uint32_t*>(base + (l << 7));
        }
    }
    if (p.pf2) {
        const size_t nlines = p.pf2_bytes >> 7;
        const char* base = reinterpret_cast<const char*>(p.pf2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t l = (size_t)i * nwg * nthreads + (size_t)wg * nthreads + threadIdx.x;
            if (l < nlines) r.v[2 + i] = *reinterpret_cast<const uint32_t*>(base + (l << 7));
        }
    }
    return r;
}
__device__ __forceinline__ void prefetch_sink(const GemmParams& p, const PrefetchRegs& r) {
    if (p.pf_sink) *p.pf_sink = r.v[0] | r.v[1] | r.v[2] | r.v[3] | r.v[4] | r.v[5];
}

template <bool KC, int ROWS>
struct TileLoader {
    // per-thread 16-byte chunks of a [ROWS x 64] (KC) or [64 x ROWS] (RC) bf16 tile
    static constexpr int NCH = ROWS / 32;
    uint32_t goff[NCH];
    uint32_t loff[NCH];
    bool ok[NCH];
    uint32_t step;   // byte advance per k-tile
    __amdgpu_buffer_rsrc_t rsrc;

    __device__ __forceinline__ void init(const void* base, int rows, int K, int ld, int r0, int tid) {
        const bf16_t* b = reinterpret_cast<const bf16_t*>(base);
        if (KC) {
            rsrc = make_rsrc(b + (size_t)r0 * ld, (size_t)(rows - r0) * ld * 2);
            step = BK * 2;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i, row = id >> 3, c = id & 7;
                goff[i] = ((uint32_t)row * ld + 8 * c) * 2;
                loff[i] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
                ok[i] = true;
            }
        } else {
            constexpr int CPR = ROWS / 8;          // chunks per k-row
            rsrc = make_rsrc(b, (size_t)K * ld * 2);
            step = (uint32_t)BK * ld * 2;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i, krow = id / CPR, c = id % CPR;
                const int col = r0 + 8 * c;
                ok[i] = col < rows;
                goff[i] = ((uint32_t)krow * ld + col) * 2;
                loff[i] = krow * (ROWS * 2) + ((c ^ rc_swizzle<ROWS>(krow)) << 4);
            }
        }
    }
    __device__ __forceinline__ void load(u32x4 (&r)[NCH], int kt) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) r[i] = buf_load16(rsrc, ok[i] ? goff[i] + (uint32_t)kt * step : (uint32_t)MEBT_OOB);
    }
    __device__ __forceinline__ void store(char* lds, const u32x4 (&r)[NCH]) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(lds + loff[i]) = r[i];
    }
};

// fragment of 16 rows x 32 k for v_mfma_f32_16x16x32_bf16: lane l holds row (l&15), k = 8*(l>>4)+j
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int blk16 /*16-row block in tile*/, int ks, int lane) {
    if (KC) {
        const int row = blk16 * 16 + (lane & 15);
        const int c = 4 * ks + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int ch = 2 * blk16 + (pp >> 1);
        const int k0 = 32 * ks + 8 * g + q, k1 = k0 + 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + k0 * (ROWS * 2) + ((ch ^ rc_swizzle<ROWS>(k0)) << 4) + 8 * (pp & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + k1 * (ROWS * 2) + ((ch ^ rc_swizzle<ROWS>(k1)) << 4) + 8 * (pp & 1)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// one output tile (m0, n0), k-tiles [kt0, kt1): register-staged 2-stage main loop + epilogue
template <bool A_KC, bool B_KC, int TBM, int TBN>
__device__ __forceinline__ void gemm_tile_regstaged(const GemmParams& p, int m0, int n0, int kt0, int kt1, char* smem, bool atomic, bool add_bias) {
    constexpr int STAGE = (TBM + TBN) * BK * 2;
    constexpr int TM = TBM / 32, TN = TBN / 32;         // MFMA tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    TileLoader<A_KC, TBM> la;
    TileLoader<B_KC, TBN> lb;
    la.init(p.A, p.M, p.K, p.lda, m0, tid);
    lb.init(p.B, p.N, p.K, p.ldb, n0, tid);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[TBM / 32], rb[TBN / 32];
    la.load(ra, kt0);
    lb.load(rb, kt0);
    la.store(smem, ra);
    lb.store(smem + TBM * BK * 2, rb);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int s = (kt - kt0) & 1;
        const char* sA = smem + s * STAGE;
        const char* sB = sA + TBM * BK * 2;
        const bool more = kt + 1 < kt1;
        if (more) {
            la.load(ra, kt + 1);
            lb.load(rb, kt + 1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<A_KC, TBM>(sA, wm * TM + i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<B_KC, TBN>(sB, wn * TN + j, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* dA = smem + (s ^ 1) * STAGE;
            la.store(dA, ra);
            lb.store(dA + TBM * BK * 2, rb);
        }
        __syncthreads();
    }
    epilogue_via_lds<TM, TN>(p, acc, smem, wave, lane, m0 + wm * (TBM / 2), n0 + wn * (TBN / 2), add_bias, atomic);
}

template <bool A_KC, bool B_KC, int TBM, int TBN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = gridDim.x, nty = gridDim.y;
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, p.M, p.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    const int nkt = (p.K + BK - 1) / BK;
    const int per = (nkt + gridDim.z - 1) / gridDim.z;
    const int kt0 = blockIdx.z * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
    gemm_tile_regstaged<A_KC, B_KC, TBM, TBN>(p, m0, n0, kt0, kt1, smem, gridDim.z > 1, blockIdx.z == 0);
}

// ------------------------------------------------------------------------------------------------
// bf16 kernel, direct-to-LDS variant: the same LDS images and fragment reads, but the tiles are
// written by `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write) into a 3-stage ring, so
// two k-tiles are in flight behind the one being multiplied.  An LDS-DMA wave-instruction writes 1 KiB
// at a wave-uniform base + lane*16, so the XOR swizzle moves to the per-lane SOURCE address.  Waits
// are counted (`s_waitcnt vmcnt(N)`, never 0 inside the loop) and the barrier is the raw s_barrier
// (a __syncthreads() would drain the DMA queue).
// ------------------------------------------------------------------------------------------------
template <bool KC, int ROWS, int NW = 4>
struct DmaLoader {
    static constexpr int NP = ROWS / (8 * NW);   // 1-KiB pieces per wave per tile (ROWS/8 pieces over NW waves)
    uint32_t goff[NP];
    bool ok[NP];
    uint32_t step;
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ void init(const void* base, int rows, int K, int ld, int r0, int wave, int lane) {
        const bf16_t* b = reinterpret_cast<const bf16_t*>(base);
        if (KC) {                             // piece = 8 rows x 128 B
            rsrc = make_rsrc(b + (size_t)r0 * ld, (size_t)(rows - r0) * ld * 2);
            step = BK * 2;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int row = 8 * (NW * i + wave) + (lane >> 3), slot = lane & 7;
                const int c = slot ^ ((row >> 1) & 7);
                goff[i] = ((uint32_t)row * ld + 8 * c) * 2;
                ok[i] = true;
            }
        } else {
            constexpr int CPR = ROWS / 8;     // 16-byte slots per k-row; a 1-KiB piece = 64 consecutive slots of the image
            rsrc = make_rsrc(b, (size_t)K * ld * 2);
            step = (uint32_t)BK * ld * 2;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int id = 64 * (NW * i + wave) + lane;
                const int krow = id / CPR, slot = id % CPR;
                const int c = slot ^ rc_swizzle<ROWS>(krow);
                const int col = r0 + 8 * c;
                ok[i] = col < rows;
                goff[i] = ((uint32_t)krow * ld + col) * 2;
            }
        }
    }
    // tile kt -> LDS image at `tile` (wave-uniform pointer)
    __device__ __forceinline__ void issue(char* tile, int kt, int wave) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const uint32_t off = ok[i] ? goff[i] + (uint32_t)kt * step : (uint32_t)MEBT_OOB;
            dma16(rsrc, (unsigned)(size_t)(lds_char_ptr)(tile + (NW * i + wave) * 1024), off);
        }
    }
};

// one output tile (m0, n0), k-tiles [kt0, kt1): LDS-DMA ring main loop + epilogue
// KS = 2: an 8-wave workgroup runs TWO of these 4-wave pipelines on the same output tile, one on the even and one on
// the odd k-tiles (each with its own LDS ring), and adds the two accumulator sets through LDS before the epilogue.
// For outputs of a chip's worth of tiles or less and a deep reduction (the MLP down-projection and its dgrad:
// 256 tiles x 64 k-tiles) a single 4-wave workgroup per CU is bound by the latency of one k-step (barrier, LDS
// read, 12-24 dependent MFMAs); the second pipeline fills exactly those bubbles.
// NW = 8 (KS = 1 only): ONE pipeline of eight waves, 4 (M) x 2 (N), on a tile twice as tall (256 x 128: the wave tile stays 64 x 64) -
// per k-tile 48 KiB for 4.2 MFLOP, a quarter less L2 -> LDS traffic per output than 128 x 128 (grouped weight gradients of a block pair:
// 768 tiles = three whole rounds of one workgroup per CU).
template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE, int KS = 1, bool ROWSUM = false, int NW = 4>
__device__ __forceinline__ void gemm_tile_dma(const GemmParams& p, int m0, int n0, int kt0, int kt1, char* smem, bool atomic, bool add_bias) {
    static_assert(NW == 4 || (NW == 8 && KS == 1), "eight waves run as one pipeline");
    constexpr int STAGE = (TBM + TBN) * BK * 2;
    constexpr int TM = TBM / (8 * NW), TN = TBN / 32;  // waves (NW / 2) x 2
    constexpr int LPT = (TBM + TBN) / (8 * NW);        // DMA instructions per wave per tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KS == 2 ? wave8 >> 2 : 0;          // pipeline of this wave
    const int wave = NW == 8 ? wave8 : (wave8 & 3);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (kt1 - kt0) / KS;                    // k-tiles per pipeline (the launcher guarantees divisibility)
    char* ring = smem + grp * (NSTAGE * STAGE);
    const int ktb = kt0 + grp;                          // pipeline tile t is k-tile ktb + KS * t

    unsigned long long* stamp = nullptr;                // diagnostics: wave 0, lane 0 of the workgroup
    if (p.stamps && tid == 0) {
        stamp = p.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4;
        stamp[0] = __builtin_amdgcn_s_memtime();
    }
    DmaLoader<A_KC, TBM, NW> la;
    DmaLoader<B_KC, TBN, NW> lb;
    la.init(p.A, p.M, p.K, p.lda, m0, wave, lane);
    lb.init(p.B, p.N, p.K, p.ldb, n0, wave, lane);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ROWSUM: row sums of A over k (bias gradients) ride on the fragments of tile column 0: rs[i] = ones . af[i]^T per k-step
    const bool do_rs = ROWSUM && p.rowsum_a != nullptr && n0 == 0 && wn == 0;
    f32x4 rs[ROWSUM ? TM : 1];
    bf16x8 ones;
    if (ROWSUM) {
#pragma unroll
        for (int i = 0; i < (ROWSUM ? TM : 1); ++i) rs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
    }
    constexpr int AHEAD = NSTAGE - 1;                   // tiles in flight behind the one being multiplied
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < nk) {
            la.issue(ring + a * STAGE, ktb + KS * a, wave);
            lb.issue(ring + a * STAGE + TBM * BK * 2, ktb + KS * a, wave);
        }
    int st = 0;                                          // ring slot of tile t
    for (int t = 0; t < nk; ++t) {
        // tile t must have landed; the (up to AHEAD-1) younger tiles may stay in flight.  vmcnt takes an
        // immediate, so the tail (fewer younger tiles than AHEAD-1) selects the count by a uniform switch.
        const int younger = min(AHEAD - 1, nk - 1 - t);
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        if (stamp && t == 0) stamp[1] = __builtin_amdgcn_s_memtime();
        if (t + AHEAD < nk) {                            // slot of tile t-1: every wave is past its reads
            int s2 = st + AHEAD; if (s2 >= NSTAGE) s2 -= NSTAGE;
            la.issue(ring + s2 * STAGE, ktb + KS * (t + AHEAD), wave);
            lb.issue(ring + s2 * STAGE + TBM * BK * 2, ktb + KS * (t + AHEAD), wave);
        }
        const char* sA = ring + st * STAGE;
        const char* sB = sA + TBM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<A_KC, TBM>(sA, wm * TM + i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<B_KC, TBN>(sB, wn * TN + j, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            if (ROWSUM && do_rs) {
#pragma unroll
                for (int i = 0; i < TM; ++i) rs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], rs[i], 0, 0, 0);
            }
        }
        if (++st == NSTAGE) st = 0;
    }
    if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();
    if (ROWSUM && do_rs && (lane >> 4) == 0) {          // every column of rs[i] holds the sum of row (lane & 15) of fragment i
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + (lane & 15);
            if (m < p.M) atomicAdd(p.rowsum_a + m, rs[i][0]);
        }
    }
    if (KS == 2) {
        // odd pipeline -> LDS -> even pipeline.  The exchange area starts above the epilogue's staging slabs
        // (4 waves x 16 rows x <= 132 floats = 33 KiB) and fits the two rings (checked by the launcher).
        constexpr int RED_OFF = 34 * 1024;
        f32x4* red = reinterpret_cast<f32x4*>(smem + RED_OFF) + (size_t)(wave * TM * TN) * 64 + lane;
        __syncthreads();                                 // both pipelines are done with the rings
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) red[(i * TN + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += red[(i * TN + j) * 64];
        epilogue_via_lds<TM, TN, false>(p, acc, smem, wave, lane, m0 + wm * (TM * 16), n0 + wn * (TBN / 2), add_bias, atomic);
    } else {
        epilogue_via_lds<TM, TN>(p, acc, smem, wave, lane, m0 + wm * (TM * 16), n0 + wn * (TBN / 2), add_bias, atomic);
    }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's own stores have left
        stamp[3] = __builtin_amdgcn_s_memtime();
    }
}

// Software-pipelined variant of gemm_tile_dma (same ring, same images, same counted waits): the fragment reads of one
// 32-deep k-step are issued BEFORE the MFMAs of the previous one, across the k-tile barrier as well, in two register sets:
//     read F1(t) | MFMA F0(t) | lgkmcnt(0), vmcnt, barrier | read F0(t+1), DMA tile t+NSTAGE -> slot of tile t | MFMA F1(t)
// In gemm_tile_dma every k-step starts with an exposed LDS round trip (the compiler waits for the reads right in front of
// the MFMAs that consume them) and every k-tile with the DMA issue; tools/fill_bench.hip (the same loops without reads and
// MFMAs) showed those loops running at 0.5-0.6 of the rate their own staging sustains (fc1 at 1536 rows: 11 us of fill,
// 23 us with the multiplications), i.e. the launches were bound by the serial chain inside a k-step, not by the fill rate.
// The slot of tile t is free for the next DMA at the barrier of tile t + 1 because every wave drains its LDS queue
// (lgkmcnt(0): the F1(t) reads were issued a whole MFMA group earlier) before it arrives there.
template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE>
__device__ __forceinline__ void gemm_tile_pipe(const GemmParams& p, int m0, int n0, int kt0, int kt1, char* smem, bool atomic, bool add_bias) {
    constexpr int STAGE = (TBM + TBN) * BK * 2;
    constexpr int TM = TBM / 32, TN = TBN / 32;
    constexpr int LPT = (TBM + TBN) / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = kt1 - kt0;
    DmaLoader<A_KC, TBM> la;
    DmaLoader<B_KC, TBN> lb;
    la.init(p.A, p.M, p.K, p.lda, m0, wave, lane);
    lb.init(p.B, p.N, p.K, p.ldb, n0, wave, lane);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = NSTAGE - 1;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < nk) {
            la.issue(smem + a * STAGE, kt0 + a, wave);
            lb.issue(smem + a * STAGE + TBM * BK * 2, kt0 + a, wave);
        }
    auto wait_tile = [&](int t) {       // tile t has landed; the (up to AHEAD - 1) younger ones may stay in flight
        const int younger = min(AHEAD - 1, nk - 1 - t);
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
    };
    bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
    wait_tile(0);
    __builtin_amdgcn_s_barrier();
    if (AHEAD < nk) {
        la.issue(smem + AHEAD * STAGE, kt0 + AHEAD, wave);
        lb.issue(smem + AHEAD * STAGE + TBM * BK * 2, kt0 + AHEAD, wave);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) a0[i] = read_frag<A_KC, TBM>(smem, wm * TM + i, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) b0[j] = read_frag<B_KC, TBN>(smem + TBM * BK * 2, wn * TN + j, 0, lane);
    int st = 0;
    for (int t = 0; t < nk; ++t) {
        const char* sA = smem + st * STAGE;
        const char* sB = sA + TBM * BK * 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = read_frag<A_KC, TBM>(sA, wm * TM + i, 1, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = read_frag<B_KC, TBN>(sB, wn * TN + j, 1, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
        int sn = st + 1; if (sn == NSTAGE) sn = 0;
        if (t + 1 < nk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave is done with slot `st` (F1 is in registers)
            wait_tile(t + 1);
            __builtin_amdgcn_s_barrier();
            const char* nA = smem + sn * STAGE;
            const char* nB = nA + TBM * BK * 2;
#pragma unroll
            for (int i = 0; i < TM; ++i) a0[i] = read_frag<A_KC, TBM>(nA, wm * TM + i, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) b0[j] = read_frag<B_KC, TBN>(nB, wn * TN + j, 0, lane);
            if (t + 1 + AHEAD < nk) {                                // into the slot of tile t: every wave is past its reads
                la.issue(smem + st * STAGE, kt0 + t + 1 + AHEAD, wave);
                lb.issue(smem + st * STAGE + TBM * BK * 2, kt0 + t + 1 + AHEAD, wave);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        st = sn;
    }
    epilogue_via_lds<TM, TN>(p, acc, smem, wave, lane, m0 + wm * (TBM / 2), n0 + wn * (TBN / 2), add_bias, atomic);
}

template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE>
__global__ __launch_bounds__(256) void gemm_bf16_pipe_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = gridDim.x, nty = gridDim.y;
    const PrefetchRegs pfr = prefetch_next(p, blockIdx.y * ntx + blockIdx.x, ntx * nty, 256);
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, p.M, p.N, tr, tc);
    gemm_tile_pipe<A_KC, B_KC, TBM, TBN, NSTAGE>(p, tr * TBM, tc * TBN, 0, (p.K + BK - 1) / BK, smem, false, true);
    prefetch_sink(p, pfr);
}

// 8-wave workgroup on ONE 256 x 256 tile: waves 2 (M) x 4 (N), wave tile 128 x 64.  Per k-tile it moves 64 KiB for
// 8.4 MFLOP (7.8 B per KFLOP, 40 % less than 192 x 128), which lifts the fill-rate cap of the large products
// (outputs of >= 12 M elements: the MLP up-projection at 3072 rows, the vocabulary head).
template <bool A_KC, bool B_KC, int NSTAGE>
__global__ __launch_bounds__(512) void gemm_bf16_w8_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TBM = 256, TBN = 256, NW = 8;
    constexpr int STAGE = (TBM + TBN) * BK * 2;
    constexpr int TM = 8, TN = 4;                       // 16x16 MFMA tiles per wave
    constexpr int LPT = (TBM + TBN) / (8 * NW);         // DMA instructions per wave per tile
    const int ntx = gridDim.x, nty = gridDim.y;
    const PrefetchRegs pfr = prefetch_next(p, blockIdx.y * ntx + blockIdx.x, ntx * nty, 512);
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, p.M, p.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = (p.K + BK - 1) / BK;

    DmaLoader<A_KC, TBM, NW> la;
    DmaLoader<B_KC, TBN, NW> lb;
    la.init(p.A, p.M, p.K, p.lda, m0, wave, lane);
    lb.init(p.B, p.N, p.K, p.ldb, n0, wave, lane);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = NSTAGE - 1;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < nk) {
            la.issue(smem + a * STAGE, a, wave);
            lb.issue(smem + a * STAGE + TBM * BK * 2, a, wave);
        }
    int st = 0;
    for (int t = 0; t < nk; ++t) {
        const int younger = min(AHEAD - 1, nk - 1 - t);
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");       // NSTAGE <= 3: at most one younger tile
        __builtin_amdgcn_s_barrier();
        if (t + AHEAD < nk) {
            int s2 = st + AHEAD; if (s2 >= NSTAGE) s2 -= NSTAGE;
            la.issue(smem + s2 * STAGE, t + AHEAD, wave);
            lb.issue(smem + s2 * STAGE + TBM * BK * 2, t + AHEAD, wave);
        }
        const char* sA = smem + st * STAGE;
        const char* sB = sA + TBM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<A_KC, TBM>(sA, wm * TM + i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = read_frag<B_KC, TBN>(sB, wn * TN + j, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (++st == NSTAGE) st = 0;
    }
    epilogue_via_lds<TM, TN>(p, acc, smem, wave, lane, m0 + wm * (TBM / 2), n0 + wn * (TBN / 4), true, false);
    prefetch_sink(p, pfr);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves in two staggered groups ("ping-pong"): while the four waves of one group (one per SIMD) issue their
// 16 MFMAs of a phase, the four waves of the other group read the fragments of their next phase from LDS and issue the LDS-DMA
// of a later k-tile; at the next s_barrier the roles swap.  Every SIMD's matrix pipe then has a wave ready in every slot — the
// structure the one-barrier-per-k-tile loops above lack when only one workgroup fits a CU (their waves read, wait and multiply
// in lock-step: 38 % MFMA issue rate at 192 x 256).  Geometry: waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4
// fragments, processed as four quadrant phases of 64 x 32 x 64 (16 MFMAs); k-tile 64; LDS = 2 k-tile buffers x {B half 0, B half 1,
// A half 0, A half 1} x 16 KiB = 128 KiB.  Group g = wave >> 2 owns A half g; the waves of both groups read both B halves.
//
// Slots (one s_barrier each).  Slot s = 8 t + 2 ph + g is the LOAD part of phase ph of k-tile t for group g and the MFMA part of
// the other group's previous phase.  Reads of k-tile t: B in slots 8t .. 8t+3 (phases 0, 1; B of quadrant column 0 is kept in
// registers for phase 3), A half 0 in 8t and 8t+4, A half 1 in 8t+1 and 8t+5.  Its buffer is refilled with k-tile t + 2 as early as
// the reads allow, a whole 16-KiB half-tile per slot issued by the loading group (4 pieces per wave): B half 0 in slot 8t+4 (group
// 0, phase 2), B half 1 in 8t+5 (group 1, phase 2), A half 0 in 8t+6 (group 0, phase 3), A half 1 in 8t+7 (group 1, phase 3) —
// a full k-tile (64 KiB) is in flight for eight slots, the depth the fill needs (issued -> landed is ~0.8 us under load; spreading
// the same pieces evenly over slots 8t+5 .. 8t+12 left the last ones three slots and the loop waited for them: 1.6 instead of 1.0 us
// per k-tile, profiles/r03_pp_bench_components.txt).  The B reads of phase 1 are drained before that slot's barrier, so every
// overwrite is issued at least one barrier after the last read of its target has completed.  Each group waits once per k-tile, at
// the end of its phase-3 LOAD part: vmcnt(8) = everything but the two half-tile shares of k-tile t + 2 it has just issued.
// ------------------------------------------------------------------------------------------------
// Persistent over the tile list (one workgroup per CU, tiles w, w + G, ...): after the main loop of tile i the prologue DMA of
// tile i + 1 is issued, then the epilogue of tile i runs out of 8 x 4 KiB swizzled slabs beside the ring (128 + 32 = 160 KiB of
// LDS), and the wait in front of the next main loop is vmcnt(16 | 32): everything but this wave's own store instructions of the
// epilogue — the 128 KiB store burst of a tile drains behind the next tile's MFMAs instead of in front of them (config 4's
// 31 744 x 2048 x 1024: 147 -> 126-136 us, the 31 744 x 16 384 head 1230 -> 1040 us = 1.02 PFLOP/s).  The counted wait is only
// taken where the instruction count after the DMA is known (full tile, plain epilogue, no accumulation); otherwise vmcnt(0).
// DBG (tools/pp_bench.hip only): bit 0 = no refill DMA in the loop, bit 1 = no MFMAs, bit 2 = no fragment reads
template <bool B_KC, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF = 128 * BK * 2;                       // one half-tile image: 16 KiB
    constexpr int TILE = 4 * HALF;                           // [B0 | B1 | A0 | A1]
    char* slabs = smem + 2 * TILE;                           // 8 x 4 KiB: the epilogue's swizzled staging slabs, outside the ring
    const int ntx = (p.N + 255) / 256, nty = (p.M + 255) / 256, ntiles = ntx * nty;
    const int G = gridDim.x;                                 // persistent: workgroup w takes tiles w, w + G, w + 2 G, ...
    const PrefetchRegs pfr = prefetch_next(p, blockIdx.x, G, 512);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;                // group = A half; wq = the wave's 64-column slice of the tile
    const int nk = p.K / BK;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B);
    const uint32_t a_step = BK * 2, b_step = B_KC ? BK * 2 : (uint32_t)BK * p.ldb * 2;

    // ---- DMA addressing of one output tile.  Piece i (0 .. 15) of a half-tile = 1 KiB: KC: rows 8 i .. 8 i + 7 x 128 B;
    // RC: k-rows 4 i .. 4 i + 3 x 256 B.  The per-lane offsets of the pieces this wave issues are kept in registers.
    int m0 = 0, n0 = 0;
    __amdgpu_buffer_rsrc_t rA, rB;
    uint32_t offB[4], offA[4];                               // main-loop refill: pieces 4 wq .. 4 wq + 3 of B half `grp` / A half `grp`
    auto kc_off = [&](int piece, int ld) -> uint32_t {       // byte offset of this lane's 16 B of piece `piece` inside a KC half-tile at k-tile 0
        const int row = 8 * piece + (lane >> 3), slot = lane & 7;
        return ((uint32_t)row * ld + 8 * (slot ^ ((row >> 1) & 7))) * 2;
    };
    auto rc_off = [&](int piece, int half) -> uint32_t {     // B stored [K][N]: image [64 k][128 columns]
        const int id = 64 * piece + lane, krow = id >> 4, slot = id & 15;
        const int col = n0 + 128 * half + 8 * (slot ^ rc_swizzle<128>(krow));
        return col < p.N ? ((uint32_t)krow * p.ldb + col) * 2 : (uint32_t)MEBT_OOB;
    };
    auto piece_off = [&](int h, int piece) -> uint32_t {     // half-tile h: 0, 1 = B halves; 2, 3 = A halves
        if (h >= 2) return kc_off(piece, p.lda) + (uint32_t)(h - 2) * 128 * p.lda * 2;
        if (B_KC) return kc_off(piece, p.ldb) + (uint32_t)h * 128 * p.ldb * 2;
        return rc_off(piece, h);
    };
    auto issue_piece = [&](char* buf, int h, int piece, uint32_t off0, int kt) {
        const unsigned dst = (unsigned)(size_t)(lds_char_ptr)(buf + h * HALF + piece * 1024);
        dma16(h >= 2 ? rA : rB, dst, off0 == (uint32_t)MEBT_OOB ? off0 : off0 + (uint32_t)kt * (h >= 2 ? a_step : b_step));
    };
    // make `tile` the current one and issue its prologue: k-tiles 0 and 1 completely, two pieces of every half-tile per wave
    auto begin_tile = [&](int tile) {
        int tr, tc;
        xcd_tile(tile, ntx, nty, p.M, p.N, tr, tc);
        m0 = tr * 256; n0 = tc * 256;
        rA = make_rsrc(Ab + (size_t)m0 * p.lda, (size_t)(p.M - m0) * p.lda * 2);
        rB = B_KC ? make_rsrc(Bb + (size_t)n0 * p.ldb, (size_t)(p.N - n0) * p.ldb * 2) : make_rsrc(Bb, (size_t)p.K * p.ldb * 2);
        for (int t0 = 0; t0 < 2 && t0 < nk && !(DBG & 16); ++t0)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) issue_piece(smem + t0 * TILE, h, 2 * wave + j, piece_off(h, 2 * wave + j), t0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { offB[j] = piece_off(grp, 4 * wq + j); offA[j] = piece_off(2 + grp, 4 * wq + j); }
    };

    const int bhalf = wq >> 1, bblk = (wq & 1) * 4;          // the wave's B fragments: half-tile wq >> 1, 16-row blocks bblk .. bblk + 3
    int tile = blockIdx.x;
    if (tile < ntiles) begin_tile(tile);
    int pending = 0;                                         // store instructions of the previous tile's epilogue that may still be in flight (0, 16 or 32)
    while (tile < ntiles) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the prologue of this tile has landed; the previous tile's stores (younger than it) need not have
        if (pending == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (pending == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();          // the stagger: group 1 runs one slot behind group 0
        const int cur_m0 = m0, cur_n0 = n0;

        bf16x8 af[2][4] = {}, bq0[2][2] = {}, bq1[2][2] = {};
        for (int t = 0; t < nk; ++t) {
            char* buf = smem + (t & 1) * TILE;
            const char* sB = buf + bhalf * HALF;
            const char* sA = buf + (2 + grp) * HALF;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                // ------------------------------------------------ LOAD part of phase ph (slot s = 8 t + 2 ph + grp)
                if (ph == 0 && !(DBG & 4)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) bq0[ks][jj] = read_frag<B_KC, 128>(sB, bblk + jj, ks, lane);
                }
                if (ph == 1 && !(DBG & 4)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) bq1[ks][jj] = read_frag<B_KC, 128>(sB, bblk + 2 + jj, ks, lane);
                }
                if ((ph == 0 || ph == 2) && !(DBG & 4)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) af[ks][ii] = read_frag<true, 128>(sA, (ph >> 1) * 4 + ii, ks, lane);
                }
                if (!(DBG & 1) && ph >= 2 && t + 2 < nk) {   // refill: a whole half-tile of k-tile t + 2 per LOAD part, 4 pieces per wave
#pragma unroll
                    for (int j = 0; j < 4; ++j) issue_piece(buf, ph == 2 ? grp : 2 + grp, 4 * wq + j, ph == 2 ? offB[j] : offA[j], t + 2);
                }
                if (ph == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the last reads of this k-tile's B are done before the slot the other group may overwrite it in
                if (ph == 3) {                               // every piece of k-tile t + 1 this wave issued has landed; still in flight: its 8 pieces of t + 2
                    if (t + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // ------------------------------------------------ MFMA part of phase ph: quadrant (ph >> 1, (ph == 1 || ph == 2))
                __builtin_amdgcn_s_setprio(1);
                if (!(DBG & 2))
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int i = (ph >> 1) * 4 + ii;
                            if (ph == 1 || ph == 2) acc[i][2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq1[ks][jj], af[ks][ii], acc[i][2 + jj], 0, 0, 0);
                            else acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq0[ks][jj], af[ks][ii], acc[i][jj], 0, 0, 0);
                        }
                __builtin_amdgcn_s_setprio(0);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();          // group 0 waits for group 1's last slot: every wave is past its reads of the ring
        // the bias is fetched (and waited for) BEFORE the next tile's prologue is issued: a compiler-inserted wait on a load
        // younger than those DMA pieces would also wait for them
        const int nb = cur_n0 + wq * 64 + (lane % 8) * 8;
        f32x4 bias2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (p.bias) {
            bias2[0] = *reinterpret_cast<const f32x4*>(p.bias + (nb < p.N ? nb : 0));
            bias2[1] = *reinterpret_cast<const f32x4*>(p.bias + (nb + 4 < p.N ? nb + 4 : 0));
            asm volatile("" ::"v"(bias2[0]), "v"(bias2[1]) : "memory");
        }
        const int next = tile + G;
        // the store burst of this tile drains behind the next tile's main loop when nothing but plain stores follows the
        // prologue DMA in this wave's memory queue (full tile, plain epilogue: exactly 16 (bf16) or 32 (fp32) store instructions per wave)
        const bool counted = next < ntiles && cur_m0 + 256 <= p.M && cur_n0 + 256 <= p.N && p.epilogue == EPI_NONE && !p.beta && p.C != nullptr;
        if (next < ntiles) begin_tile(next);
        if (!(DBG & 8)) epilogue_via_lds<8, 4, false, true>(p, acc, slabs, wave, lane, cur_m0 + grp * 128, cur_n0 + wq * 64, true, false, bias2);
        else if (acc[0][0][0] == 123.f) *reinterpret_cast<float*>(p.C) = acc[7][3][1];
        pending = (counted && !(DBG & 8)) ? (p.c_f32 ? 32 : 16) : 0;       // fp32 output: two 16-byte stores per lane and pass
        tile = next;
    }
    prefetch_sink(p, pfr);
}

// two pipelines per workgroup (KS = 2): whole reduction in one workgroup, K a multiple of 128
template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE>
__global__ __launch_bounds__(512) void gemm_bf16_dma_ks2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = gridDim.x, nty = gridDim.y;
    const PrefetchRegs pfr = prefetch_next(p, blockIdx.y * ntx + blockIdx.x, ntx * nty, 512);
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, p.M, p.N, tr, tc);
    gemm_tile_dma<A_KC, B_KC, TBM, TBN, NSTAGE, 2>(p, tr * TBM, tc * TBN, 0, p.K / BK, smem, false, true);
    prefetch_sink(p, pfr);
}

template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE>
__global__ __launch_bounds__(256) void gemm_bf16_dma_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = gridDim.x, nty = gridDim.y;
    const PrefetchRegs pfr = prefetch_next(p, (blockIdx.z * nty + blockIdx.y) * ntx + blockIdx.x, ntx * nty * gridDim.z, 256);
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, p.M, p.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    const int nkt = (p.K + BK - 1) / BK;
    const int per = (nkt + gridDim.z - 1) / gridDim.z;
    const int kt0 = blockIdx.z * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) { prefetch_sink(p, pfr); return; }
    if (p.slab) {                 // partial product of this k-range into its own fp32 slab (reduced by splitk_reduce_kernel)
        GemmParams q = p;
        q.C = reinterpret_cast<float*>(p.C) + (size_t)blockIdx.z * p.slab;
        gemm_tile_dma<A_KC, B_KC, TBM, TBN, NSTAGE>(q, m0, n0, kt0, kt1, smem, false, false);
        prefetch_sink(p, pfr);
        return;
    }
    gemm_tile_dma<A_KC, B_KC, TBM, TBN, NSTAGE>(p, m0, n0, kt0, kt1, smem, gridDim.z > 1, blockIdx.z == 0);
    prefetch_sink(p, pfr);
}

// Split-K for small outputs with a deep reduction: S workgroups per LARGE tile each reduce K/S and store fp32
// partials (slab mode above); this kernel adds the S slabs and applies the product's real epilogue.  Why: the
// global->LDS fill rate caps a tile shape at ~13 TB/s / (1/bm + 1/bn) B per flop; a 1536 x 1024 output only
// fills the chip with 96 x 64 tiles (~500 TFLOP/s cap), split 4 ways it can use 192 x 128 tiles (~1000).
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const float* part, long slab) {
    const long n4 = p.N / 4;
    const long total = (long)p.M * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        f32x4 v[S];
#pragma unroll
        for (int s = 0; s < S; ++s) v[s] = *reinterpret_cast<const f32x4*>(part + (size_t)s * slab + (size_t)m * p.N + n);
#pragma unroll
        for (int s = 1; s < S; ++s) v[0] += v[s];
        epilogue_store<bf16_t>(p, m, n, v[0], true, false);
    }
}

template <bool A_KC, bool B_KC, int TBM, int TBN, int NSTAGE>
__global__ __launch_bounds__(256) void gemm_pair_kernel(const GemmPair g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PrefetchRegs pfr = prefetch_next(g.p[0], blockIdx.x, gridDim.x, 256);       // the pair's prefetch request rides on product 0
    const int which = (int)blockIdx.x >= g.tiles0 ? 1 : 0;
    const GemmParams& p = g.p[which];
    const int t = blockIdx.x - (which ? g.tiles0 : 0);
    int tr, tc;
    xcd_tile(t, g.ntx[which], (p.M + TBM - 1) / TBM, p.M, p.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    gemm_tile_dma<A_KC, B_KC, TBM, TBN, NSTAGE>(p, m0, n0, 0, (p.K + BK - 1) / BK, smem, false, true);
    prefetch_sink(g.p[0], pfr);
}

// grouped weight gradients: blockIdx.x enumerates the tiles of all groups (RC x RC, fp32 result)
template <int TBM, int TBN, int NSTAGE, int NW = 4>
__global__ __launch_bounds__(NW * 64) void wgrad_grouped_kernel(const GroupedWgrad w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x;
    int g = 0;
#pragma unroll
    for (int i = 1; i < MEBT_MAX_GROUP; ++i)
        if (i < w.n && tile >= w.tile_start[i]) g = i;
    const GroupedWgrad::Item& it = w.g[g];
    const int t = tile - w.tile_start[g];
    GemmParams p;
    p.A = it.A; p.B = it.B; p.C = it.C; p.C2 = nullptr; p.bias = nullptr; p.aux = nullptr;
    p.M = it.M; p.N = it.N; p.K = it.K; p.lda = it.lda; p.ldb = it.ldb; p.ldc = it.ldc; p.ld_aux = 0;
    p.a_kc = 0; p.b_kc = 0; p.epilogue = EPI_NONE; p.c_f32 = 1; p.beta = w.beta; p.split_k = 1; p.drop.thresh = 0;
    p.rowsum_a = it.bias;
    if (w.Cb) { p.C = reinterpret_cast<bf16_t*>(w.Cb) + (it.C - w.gW); p.c_f32 = 0; }
    if (w.fused) {
        const ptrdiff_t off = it.C - w.gW;
        p.epilogue = EPI_ADAMW; p.opt = w.opt;
        p.opt_p = w.W + off; p.opt_m = w.mW + off; p.opt_v = w.vW + off;
        p.opt_lp = w.Wlp ? (void*)(reinterpret_cast<bf16_t*>(w.Wlp) + off) : nullptr;
    }
    int tr, tc;
    xcd_tile(t, it.ntx, (it.M + TBM - 1) / TBM, it.M, it.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    gemm_tile_dma<false, false, TBM, TBN, NSTAGE, 1, true, NW>(p, m0, n0, 0, (it.K + BK - 1) / BK, smem, false, false);
}

// ------------------------------------------------------------------------------------------------
// f32 kernel (parity mode).  LDS image is always [k][row] (row fastest); KC operands are
// transposed by the staging write, RC operands are copied.  BK = 16.
// ------------------------------------------------------------------------------------------------
constexpr int FBK = 16, FLD = 132;   // padded row length (floats)

// ROWS = 128 or 64 tile rows: ROWS * 4 float4 chunks per k-tile, ROWS / 64 per thread
template <bool KC, int ROWS>
struct TileLoaderF32 {
    static constexpr int NCH = ROWS / 64;
    uint32_t goff[NCH];
    bool ok[NCH];
    int l0[NCH], l1[NCH];
    uint32_t step;
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ void init(const void* base, int rows, int K, int ld, int r0, int tid) {
        const float* b = reinterpret_cast<const float*>(base);
        if (KC) {   // tile [ROWS rows][16 k]
            rsrc = make_rsrc(b + (size_t)r0 * ld, (size_t)(rows - r0) * ld * 4);
            step = FBK * 4;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i, row = id >> 2, c = id & 3;
                goff[i] = ((uint32_t)row * ld + 4 * c) * 4;
                ok[i] = true;
                l0[i] = 4 * c;   // k
                l1[i] = row;
            }
        } else {    // tile [16 k][ROWS rows]
            constexpr int CPR = ROWS / 4;
            rsrc = make_rsrc(b, (size_t)K * ld * 4);
            step = (uint32_t)FBK * ld * 4;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = tid + 256 * i, krow = id / CPR, c = id % CPR;
                const int col = r0 + 4 * c;
                ok[i] = col < rows;
                goff[i] = ((uint32_t)krow * ld + col) * 4;
                l0[i] = krow;
                l1[i] = 4 * c;
            }
        }
    }
    __device__ __forceinline__ void load(f32x4 (&r)[NCH], int kt) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            r[i] = __builtin_bit_cast(f32x4, buf_load16(rsrc, ok[i] ? goff[i] + (uint32_t)kt * step : (uint32_t)MEBT_OOB));
    }
    __device__ __forceinline__ void store(float* lds, const f32x4 (&r)[NCH]) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (KC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[(l0[i] + j) * FLD + l1[i]] = r[i][j];
            } else {
                *reinterpret_cast<f32x4*>(lds + l0[i] * FLD + l1[i]) = r[i];
            }
        }
    }
};

// TBM, TBN in {128, 64}: 4 waves as 2 x 2, wave tile (TBM / 2) x (TBN / 2) = 1 or 2 MFMA tiles of 32 x 32 each way.  The small tiles exist
// for outputs that leave CUs idle at 128 x 128 (a 1536 x 1024 product is 96 tiles for 256 CUs; a 1024 x 1024 weight gradient 64): round 6,
// the exact-fp32 engine is the reference's own arithmetic and its step was 14 x the bf16 step's time.
template <bool A_KC, bool B_KC, int TBM = 128, int TBN = 128>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float sA[FBK * FLD];
    __shared__ __attribute__((aligned(16))) float sB[FBK * FLD];
    constexpr int MI = TBM / 64, NJ = TBN / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TBM, n0 = blockIdx.x * TBN;
    const int nkt = (p.K + FBK - 1) / FBK;
    const int per = (nkt + gridDim.z - 1) / gridDim.z;
    const int kt0 = blockIdx.z * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;

    TileLoaderF32<A_KC, TBM> la;
    TileLoaderF32<B_KC, TBN> lb;
    la.init(p.A, p.M, p.K, p.lda, m0, tid);
    lb.init(p.B, p.N, p.K, p.ldb, n0, tid);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[TBM / 64], rb[TBN / 64];
    la.load(ra, kt0);
    lb.load(rb, kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        la.store(sA, ra);
        lb.store(sB, rb);
        __syncthreads();
        if (kt + 1 < kt1) {
            la.load(ra, kt + 1);
            lb.load(rb, kt + 1);
        }
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2) {
            const int k = kk + (lane >> 5);
            float a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = sA[k * FLD + wm * (TBM / 2) + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = sB[k * FLD + wn * (TBN / 2) + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D'[n_local][m_local]: lane owns m_local = lane&31; register r -> n_local = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool atomic = gridDim.z > 1;
    const bool add_bias = blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (TBM / 2) + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * (TBN / 2) + j * 32 + 8 * g + 4 * (lane >> 5);
                if (m < p.M && n < p.N) {
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    epilogue_store<float>(p, m, n, v, add_bias, atomic);
                }
            }
    }
}

}  // namespace

// LDS bytes of a two-pipeline (KS = 2) configuration; 0 = not available for this tile / ring depth
static inline int ks2_lds(int tbm, int tbn, int ring) {
    const bool tile_ok = (tbm == 96 && tbn == 64) || (tbm == 64 && tbn == 64) || (tbm == 96 && tbn == 128) || (tbm == 64 && tbn == 128) || (tbm == 128 && tbn == 64);
    if (!tile_ok || ring < 2 || ring > 3) return 0;
    const int bytes = 2 * ring * (tbm + tbn) * BK * 2;
    const int need = 34 * 1024 + 4 * (tbm / 32) * (tbn / 32) * 1024;
    return (bytes <= 160 * 1024 && bytes >= need) ? bytes : 0;
}
template <bool AK, bool BKC>
static void layout_launch_ks2(const GemmParams& p, int tbm, int tbn, int ring, hipStream_t stream) {
    const dim3 grid((p.N + tbn - 1) / tbn, (p.M + tbm - 1) / tbm, 1);
    const int lds = ks2_lds(tbm, tbn, ring);
#define KS2_T(TM_, TN_)                                                                                                        \
    do {                                                                                                                               \
        if (ring == 3) hipLaunchKernelGGL((gemm_bf16_dma_ks2_kernel<AK, BKC, TM_, TN_, 3>), grid, dim3(512), lds, stream, p);            \
        else hipLaunchKernelGGL((gemm_bf16_dma_ks2_kernel<AK, BKC, TM_, TN_, 2>), grid, dim3(512), lds, stream, p);                      \
    } while (0)
#define KS2_L()                                                      \
    do {                                                            \
        if (tbm == 96 && tbn == 64) KS2_T(96, 64);         \
        else if (tbm == 64 && tbn == 64) KS2_T(64, 64);    \
        else if (tbm == 96 && tbn == 128) KS2_T(96, 128);  \
        else if (tbm == 64 && tbn == 128) KS2_T(64, 128);  \
        else KS2_T(128, 64);                               \
    } while (0)
    KS2_L();
#undef KS2_L
#undef KS2_T
}

// staging 8 + ring (10 .. 12): the software-pipelined main loop (gemm_tile_pipe), whole reduction per workgroup
template <bool AK, bool BKC>
static bool layout_launch_pipe(const GemmParams& p, int tbm, int tbn, int ring, hipStream_t stream) {
    const dim3 grid((p.N + tbn - 1) / tbn, (p.M + tbm - 1) / tbm, 1);
#define PIPE_T(TM_, TN_)                                                                                                                      \
    do {                                                                                                                                      \
        if (ring * (TM_ + TN_) * BK * 2 > 160 * 1024) return false;                                                                           \
        if (ring >= 4) hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 160 * 1024 ? 4 : 2)>), grid, dim3(256), 4 * (TM_ + TN_) * BK * 2, stream, p); \
        else if (ring == 3) hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, 3>), grid, dim3(256), 3 * (TM_ + TN_) * BK * 2, stream, p); \
        else hipLaunchKernelGGL((gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, 2>), grid, dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, p);            \
        return true;                                                                                                                          \
    } while (0)
    if (tbm == 128 && tbn == 128) PIPE_T(128, 128);
    else if (tbm == 192 && tbn == 128) PIPE_T(192, 128);
    else if (tbm == 96 && tbn == 128) PIPE_T(96, 128);
    else if (tbm == 96 && tbn == 64) PIPE_T(96, 64);
    else if (tbm == 128 && tbn == 64) PIPE_T(128, 64);
    else if (tbm == 64 && tbn == 128) PIPE_T(64, 128);
    else if (tbm == 64 && tbn == 64) PIPE_T(64, 64);
#undef PIPE_T
    return false;
}

template <bool AK, bool BKC>
static void layout_launch_cfg(const GemmParams& p, int tbm, int tbn, int staging, int split, hipStream_t stream) {
    if (staging >= 10 && staging <= 12 && split == 1 && !p.slab) {
        if (layout_launch_pipe<AK, BKC>(p, tbm, tbn, staging - 8, stream)) return;
        staging -= 8;
    }
    const dim3 grid((p.N + tbn - 1) / tbn, (p.M + tbm - 1) / tbm, split);
#define LAUNCH_T(TM_, TN_)                                                                                           \
        do {                                                                                                         \
            if (staging == 5 && 5 * (TM_ + TN_) * BK * 2 <= 160 * 1024) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, (5 * (TM_ + TN_) * BK * 2 <= 160 * 1024 ? 5 : 2)>), grid, dim3(256), 5 * (TM_ + TN_) * BK * 2, stream, p); \
            else if (staging >= 4) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 4>), grid, dim3(256), 4 * (TM_ + TN_) * BK * 2, stream, p); \
            else if (staging == 3) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 3>), grid, dim3(256), 3 * (TM_ + TN_) * BK * 2, stream, p); \
            else if (staging == 2) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 2>), grid, dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, p); \
            else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKC, TM_, TN_>), grid, dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, p);          \
        } while (0)
        /* tiles that exist as LDS-DMA kernels only (2..4 stages) */
#define LAUNCH_D(TM_, TN_)                                                                                           \
        do {                                                                                                         \
            if (staging >= 4 && 4 * (TM_ + TN_) * BK * 2 <= 128 * 1024) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024 ? 4 : 2)>), grid, dim3(256), 4 * (TM_ + TN_) * BK * 2, stream, p); \
            else if (staging >= 3) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 3>), grid, dim3(256), 3 * (TM_ + TN_) * BK * 2, stream, p); \
            else hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 2>), grid, dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, p); \
        } while (0)
#define LAUNCH_BF16()                                                  \
        do {                                                           \
            if (tbm == 128 && tbn == 128) LAUNCH_T(128, 128); \
            else if (tbm == 192 && tbn == 128) LAUNCH_D(192, 128); \
            else if (tbm == 96 && tbn == 128) LAUNCH_D(96, 128); \
            else if (tbm == 96 && tbn == 64) LAUNCH_D(96, 64); \
            else if (tbm == 128 && tbn == 64) LAUNCH_T(128, 64); \
            else if (tbm == 64 && tbn == 128) LAUNCH_T(64, 128); \
            else LAUNCH_T(64, 64);                            \
        } while (0)
    LAUNCH_BF16();
#undef LAUNCH_BF16
#undef LAUNCH_D
#undef LAUNCH_T
}

template <bool AK, bool BKC>
static void layout_launch_pair(GemmPair& g, int tbm, int tbn, int staging, hipStream_t stream) {
    int tiles[2];
    for (int i = 0; i < 2; ++i) {
        g.ntx[i] = (g.p[i].N + tbn - 1) / tbn;
        tiles[i] = ((g.p[i].M + tbm - 1) / tbm) * g.ntx[i];
    }
    g.tiles0 = tiles[0];
    const dim3 grid(tiles[0] + tiles[1]);
#define PAIR_T(TM_, TN_)                                                                                                                                  \
    do {                                                                                                                                                  \
        if (staging >= 4 && 4 * (TM_ + TN_) * BK * 2 <= 128 * 1024) hipLaunchKernelGGL((gemm_pair_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024 ? 4 : 2)>), grid, dim3(256), 4 * (TM_ + TN_) * BK * 2, stream, g); \
        else if (staging >= 3) hipLaunchKernelGGL((gemm_pair_kernel<AK, BKC, TM_, TN_, 3>), grid, dim3(256), 3 * (TM_ + TN_) * BK * 2, stream, g);          \
        else hipLaunchKernelGGL((gemm_pair_kernel<AK, BKC, TM_, TN_, 2>), grid, dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, g);                           \
    } while (0)
#define PAIR_L()                                                         \
    do {                                                                 \
        if (tbm == 128 && tbn == 128) PAIR_T(128, 128);         \
        else if (tbm == 192 && tbn == 128) PAIR_T(192, 128);    \
        else if (tbm == 96 && tbn == 128) PAIR_T(96, 128);      \
        else if (tbm == 96 && tbn == 64) PAIR_T(96, 64);        \
        else if (tbm == 128 && tbn == 64) PAIR_T(128, 64);      \
        else if (tbm == 64 && tbn == 128) PAIR_T(64, 128);      \
        else PAIR_T(64, 64);                                    \
    } while (0)
    PAIR_L();
#undef PAIR_L
#undef PAIR_T
}

// grouped weight gradients of one block.  Items are ordered by reduction length, longest first (the key
// projection reduces over twice as many tokens as the rest: started last, its tiles were the tail of the
// launch), and the block tile / ring depth are autotuned per group signature like the single GEMMs.
static void layout_launch_grouped(GroupedWgrad& c, int tbm, int tbn, int stages, hipStream_t stream) {
    for (int i = 1; i < c.n; ++i)              // insertion sort, K descending (longest reduction first: its tiles are not the tail)
        for (int j = i; j > 0 && c.g[j].K > c.g[j - 1].K; --j) { const GroupedWgrad::Item t = c.g[j]; c.g[j] = c.g[j - 1]; c.g[j - 1] = t; }
    int tiles = 0;
    for (int i = 0; i < c.n; ++i) {
        c.g[i].ntx = (c.g[i].N + tbn - 1) / tbn;
        c.tile_start[i] = tiles;
        tiles += ((c.g[i].M + tbm - 1) / tbm) * c.g[i].ntx;
    }
    for (int i = c.n; i <= MEBT_MAX_GROUP; ++i) c.tile_start[i] = tiles;
#define LAUNCH_G(TM_, TN_)                                                                                                                   \
    do {                                                                                                                                    \
        if (stages >= 4) hipLaunchKernelGGL((wgrad_grouped_kernel<TM_, TN_, 4>), dim3(tiles), dim3(256), 4 * (TM_ + TN_) * BK * 2, stream, c);      \
        else if (stages == 3) hipLaunchKernelGGL((wgrad_grouped_kernel<TM_, TN_, 3>), dim3(tiles), dim3(256), 3 * (TM_ + TN_) * BK * 2, stream, c); \
        else hipLaunchKernelGGL((wgrad_grouped_kernel<TM_, TN_, 2>), dim3(tiles), dim3(256), 2 * (TM_ + TN_) * BK * 2, stream, c);                 \
    } while (0)
    if (tbm == 256 && tbn == 128) {          // eight waves, one workgroup per CU (ring 2: 96 KiB, ring 3: 144 KiB)
        if (stages >= 3) hipLaunchKernelGGL((wgrad_grouped_kernel<256, 128, 3, 8>), dim3(tiles), dim3(512), 3 * 384 * BK * 2, stream, c);
        else hipLaunchKernelGGL((wgrad_grouped_kernel<256, 128, 2, 8>), dim3(tiles), dim3(512), 2 * 384 * BK * 2, stream, c);
    }
    else if (tbm == 128 && tbn == 128) LAUNCH_G(128, 128);
    else if (tbm == 128 && tbn == 64) LAUNCH_G(128, 64);
    else if (tbm == 64 && tbn == 128) LAUNCH_G(64, 128);
    else LAUNCH_G(64, 64);
#undef LAUNCH_G
}

template <bool AK, bool BKC>
static void layout_launch_w8(const GemmParams& p, int ring, hipStream_t stream) {
    const dim3 grid((p.N + 255) / 256, (p.M + 255) / 256, 1);
    if (ring >= 3) return;                          // 3 x 64 KiB does not fit the 160 KiB LDS: ring 2 only
    hipLaunchKernelGGL((gemm_bf16_w8_kernel<AK, BKC, 2>), grid, dim3(512), 2 * 512 * BK * 2, stream, p);
}

// the staggered two-group 256 x 256 kernel: A KC only, K a multiple of 64
template <bool AK, bool BKC>
static void layout_launch_pp(const GemmParams& p, hipStream_t stream) {
    const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256);
    const dim3 grid(tiles < 256 ? tiles : 256, 1, 1);          // persistent over the tile list: one workgroup per CU
    if (AK) hipLaunchKernelGGL((gemm_bf16_pp_kernel<BKC>), grid, dim3(512), 8 * 128 * BK * 2 + 8 * 4096, stream, p);
}

// dynamic-LDS attributes of every instantiation of one operand layout (called once per process)
template <bool AK, bool BKC>
static int layout_set_attrs() {
#define SET_T(TM_, TN_)                                                                                                  \
    do {                                                                                                                         \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<AK, BKC, TM_, TN_>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TM_ + TN_) * BK * 2)); \
        if (5 * (TM_ + TN_) * BK * 2 <= 160 * 1024) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, (5 * (TM_ + TN_) * BK * 2 <= 160 * 1024 ? 5 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (TM_ + TN_) * BK * 2)); \
    } while (0)
#define SET_D(TM_, TN_)                                                                                                  \
    do {                                                                                                                         \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TM_ + TN_) * BK * 2)); \
        if (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024 ? 4 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TM_ + TN_) * BK * 2)); \
    } while (0)
#define SET_K(TM_, TN_)                                                                                                                           \
    do {                                                                                                                                                  \
        if (ks2_lds(TM_, TN_, 2)) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_ks2_kernel<AK, BKC, TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ks2_lds(TM_, TN_, 2))); \
        if (ks2_lds(TM_, TN_, 3)) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_ks2_kernel<AK, BKC, TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ks2_lds(TM_, TN_, 3))); \
    } while (0)
    SET_T(128, 128); SET_T(128, 64); SET_T(64, 128); SET_T(64, 64); SET_D(192, 128); SET_D(96, 128); SET_D(96, 64);
#define SET_PIPE(TM_, TN_)                                                                                                       \
    do {                                                                                                                         \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TM_ + TN_) * BK * 2)); \
        if (4 * (TM_ + TN_) * BK * 2 <= 160 * 1024) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pipe_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 160 * 1024 ? 4 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TM_ + TN_) * BK * 2)); \
    } while (0)
    SET_PIPE(128, 128); SET_PIPE(192, 128); SET_PIPE(96, 128); SET_PIPE(96, 64); SET_PIPE(128, 64); SET_PIPE(64, 128); SET_PIPE(64, 64);
#undef SET_PIPE
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w8_kernel<AK, BKC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * BK * 2));
    if (AK) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<BKC>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * BK * 2 + 8 * 4096));
    SET_K(96, 64); SET_K(64, 64); SET_K(96, 128); SET_K(64, 128); SET_K(128, 64);
#undef SET_K
#undef SET_D
#undef SET_T
    return MEBT_OK;
}
template <bool AK, bool BKC>
static int layout_set_pair_attrs() {
#define SET_P(TM_, TN_)                                                                                                                           \
    do {                                                                                                                                                  \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pair_kernel<AK, BKC, TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pair_kernel<AK, BKC, TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TM_ + TN_) * BK * 2)); \
        if (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024) MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pair_kernel<AK, BKC, TM_, TN_, (4 * (TM_ + TN_) * BK * 2 <= 128 * 1024 ? 4 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TM_ + TN_) * BK * 2)); \
    } while (0)
    SET_P(192, 128); SET_P(128, 128); SET_P(96, 128); SET_P(128, 64); SET_P(64, 128); SET_P(96, 64); SET_P(64, 64);
#undef SET_P
    return MEBT_OK;
}
static int layout_set_grouped_attrs() {
#define SET_G(TM_, TN_)                                                                                                                                        \
    do {                                                                                                                                                  \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<TM_, TN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<TM_, TN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TM_ + TN_) * BK * 2)); \
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<TM_, TN_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TM_ + TN_) * BK * 2)); \
    } while (0)
    SET_G(128, 128); SET_G(128, 64); SET_G(64, 128); SET_G(64, 64);
#undef SET_G
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<256, 128, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 384 * BK * 2));
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<256, 128, 3, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 384 * BK * 2));
    return MEBT_OK;
}
