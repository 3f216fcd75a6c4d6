// MFMA flash attention for gfx950, bf16, head size 64: forward, dQ and dK/dV kernels.
// Replaces reference mebt/modules/gpt.py:131-137 (q@k^T * 1/sqrt(hd) -> softmax -> @v) and its
// autograd backward for the four routings (NQ x NK) = (NS,NC), (NS,NS), (NT,NS), (NS,NS+NT).
//
// Structure (all three kernels): a workgroup = 8 waves, each wave owns 16 rows (queries in
// forward/dQ, keys in dK/dV) of one (batch, head); the other side is streamed through LDS in CHUNKS
// of 256 rows (four 64-row tile images per operand), copied global->LDS by LDS-DMA into a two-deep
// ring: for the shapes of this model (<= 768 streamed rows) every load of the kernel is in flight
// within the first microsecond and the 64-row tiles of a chunk are multiplied back to back with no
// barrier between them (the previous register-staged 64-row pipeline paid one exposed HBM round
// trip and one barrier per tile: 20-29 us per launch for ~3 us of math).  Two barriers per chunk.
// The score tile is computed TRANSPOSED with respect
// to the owned rows so that (i) the softmax statistics of a row live in one lane (+2 shuffles
// across the 4 lane groups) and (ii) the score accumulators are directly the B operand of the
// next product (v_mfma_f32_16x16x32_bf16: lane = column, registers = 4 rows), with no LDS round
// trip.  The operand that must be read column-wise (V in forward, K in dQ, Q/dO in dK/dV) comes
// from the same row-major LDS image through ds_read_b64_tr_b16.
//
// LDS tile image: [64 rows][64 bf16] = 128-byte rows, 16-byte chunk index XOR-swizzled with
// ((row >> 1) & 3) << 1, which is conflict-free for both the ds_read_b128 row reads and the
// transposed reads used here.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TILE = 64;                 // rows per streamed tile
constexpr int TILE_BYTES = TILE * 64 * 2;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) << 1; }

// row-wise fragment: 16 rows (blk16) x 32 k (ks): lane l holds row l&15, k = 32ks + 8(l>>4) + j
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int blk16, int ks, int lane) {
    const int row = blk16 * 16 + (lane & 15);
    const int c = 4 * ks + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ swz(row)) << 4));
}
// column-wise fragment (transposed read): the operand's "row" is tile column 16*cb + (l&15); its
// k index 8g + j maps to tile row  32*kk + 16*(j>>2) + 4g + (j&3)  — the order in which a pair of
// 16-row score accumulators presents its rows (see pack_acc).
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int cb, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3;
    const int ch = 2 * cb + (pp >> 1);
    const int r0 = 32 * kk + 4 * g + q4, r1 = r0 + 16;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + r0 * 128 + ((ch ^ swz(r0)) << 4) + 8 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + r1 * 128 + ((ch ^ swz(r1)) << 4) + 8 * (pp & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// The fragment byte offsets depend only on the lane: computed once per kernel, the loop adds the stage base.
struct FragOffsets {
    int rows[4][2];        // frag_rows(blk16, ks)
    int cols[4][2][2];     // frag_cols(cb, kk): the two transposed reads
    __device__ __forceinline__ void init(int lane) {
        const int g = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = b * 16 + i, c = 4 * ks + g;
                rows[b][ks] = row * 128 + ((c ^ swz(row)) << 4);
            }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = 2 * cb + (pp >> 1);
                const int r0 = 32 * kk + 4 * g + q4, r1 = r0 + 16;
                cols[cb][kk][0] = r0 * 128 + ((ch ^ swz(r0)) << 4) + 8 * (pp & 1);
                cols[cb][kk][1] = r1 * 128 + ((ch ^ swz(r1)) << 4) + 8 * (pp & 1);
            }
    }
    __device__ __forceinline__ bf16x8 row_frag(const char* tile, int b, int ks) const {
        return *reinterpret_cast<const bf16x8*>(tile + rows[b][ks]);
    }
    __device__ __forceinline__ bf16x8 col_frag(const char* tile, int cb, int kk) const {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + cols[cb][kk][0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + cols[cb][kk][1]));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
};
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; exp2(-inf) = 0

// two 16-row accumulators (rows 4g+r of blocks 2kk and 2kk+1) -> B operand of the next product
__device__ __forceinline__ bf16x8 pack_acc(const f32x4& a, const f32x4& b) {
    bf16x8 v = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return v;
}
// fragment straight from global memory: lane l holds row (l&15) of a 16-row block, k = 32ks + 8(l>>4) + j
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* base, int row, int rows, int ld, int ks, int lane) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row >= rows) return z;
    return *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + 32 * ks + 8 * (lane >> 4));
}
// Across the 4 lane groups (same l & 15) without the LDS crossbar: v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- /
// 32-lane rows between two registers in the vector ALU.  With both operands = v: after the 16-lane swap one register holds rows
// {0, 0, 2, 2} and the other {1, 1, 3, 3}; after the 32-lane swap {lo, lo} and {hi, hi} — one add / max of the pair is the xor-16 /
// xor-32 butterfly step.  __shfl_xor compiled to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): four exposed LDS round trips per 64-key tile
// in the forward's softmax (profiles/r04_attention_forward.txt).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float group_sum(float v) {   // across the 4 lane groups (same l&15)
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float group_max(float v) {
    u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// chunked LDS-DMA staging shared by the three kernels
// ------------------------------------------------------------------------------------------------
constexpr int WAVES = 8;                         // 16 owned rows per wave -> 128 rows per workgroup
constexpr int BLOCK_ROWS = WAVES * 16;
constexpr int CHUNK_TILES = 4;                   // 64-row tiles per chunk
constexpr int CHUNK = CHUNK_TILES * TILE;        // 256 streamed rows per chunk
constexpr int CHUNK_BYTES = CHUNK_TILES * TILE_BYTES;          // one operand: 32 KiB
constexpr int STAT_BYTES = CHUNK * 4;                          // one fp32 vector of a chunk (dK/dV kernel)
constexpr int MASK_BYTES = 4096;                               // dropout keep bits of a chunk (backward kernels): 4 DMA pieces
constexpr int MASK_OFF = 2 * CHUNK_BYTES + 2 * STAT_BYTES;
constexpr int STAGE_BYTES = MASK_OFF + MASK_BYTES;             // two operands (+ lse, delta) + keep bits
constexpr int ATTN_LDS = 2 * STAGE_BYTES;                      // 140 KiB
constexpr int DMA_PER_WAVE = 2 * (CHUNK_BYTES / 1024) / WAVES; // 8 LDS-DMA instructions per wave per chunk

// One operand of a chunk = 32 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+8, w+16, w+24.
// A piece is written lane-linearly (lane l -> bytes 16 l), so the XOR swizzle of the tile image is applied
// to the per-lane SOURCE address.  Rows beyond `rows` are zero-filled by the buffer bounds.
// NW waves copy a chunk of 32 NW rows (NW / 2 tiles): four pieces per wave and operand whatever NW is.
template <int NW = WAVES>
struct ChunkDma {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t goff[4];
    uint32_t step;
    __device__ __forceinline__ void init(const bf16_t* base, int rows, int ld, int wave, int lane) {
        rsrc = make_rsrc(base, rows > 0 ? ((size_t)(rows - 1) * ld + 64) * 2 : 0);
        step = (uint32_t)(32 * NW) * ld * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (wave + NW * i) + (lane >> 3), slot = lane & 7;
            goff[i] = ((uint32_t)row * ld + 8 * (slot ^ swz(row))) * 2;
        }
    }
    __device__ __forceinline__ void issue(char* dst, int chunk, int wave) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rsrc, lds_addr_of(dst + (wave + NW * i) * 1024), goff[i] + (uint32_t)chunk * step);
    }
    // the same chunk with its rows taken from positions of a larger buffer: row r of the chunk image <- buffer row sidx[first + r]
    // (an index list staged in LDS; rows at or beyond `nrows` are zero-filled).  The swizzle belongs to the image row, the address to
    // the gathered row.  `rsrc` must span the whole buffer of this sample.
    __device__ __forceinline__ void issue_gather(char* dst, int chunk, int wave, int lane, const int* sidx, int nrows, int ld) const {
        int src[4];                 // the four index reads first: one LDS round trip in front of the four DMAs, not one in front of each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = chunk * (32 * NW) + 8 * (wave + NW * i) + (lane >> 3);
            src[i] = r < nrows ? sidx[r] : -1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (wave + NW * i) + (lane >> 3), slot = lane & 7;
            const uint32_t off = src[i] >= 0 ? ((uint32_t)src[i] * ld + 8 * (slot ^ swz(row))) * 2 : (uint32_t)MEBT_OOB;
            dma16(rsrc, lds_addr_of(dst + (wave + NW * i) * 1024), off);
        }
    }
};
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// NW = 8: 128 query rows per workgroup, chunks of 256 keys; NW = 4: 64 query rows, chunks of 128 keys (32 KiB per ring stage).
// The 4-wave form exists for the routings whose 128-row grid does not fill the chip (NQ = 256 latents: 2 x heads x batch
// workgroups — 128 at config 4, 192 at config 2): the kernel is bound by vector issue (softmax), two waves of one workgroup share
// a SIMD, so the same waves spread over twice the CUs run up to twice as fast; NST = ring depth (stages of K + V).
// SPLIT = 2 (long key sets on a grid that still leaves SIMD slots idle: config 4's 7936 keys for 256 latent queries at batch 4): TWO
// groups of NW waves work on the SAME 16 NW query rows, group s on the key chunks s, s + 2, s + 4, ... with a ring of its own, and the
// groups' (m, l, O) are merged through LDS at the end — twice the waves per query row without a second kernel or partial results
// in HBM.  The groups run in lock step (the workgroup barrier is shared): a group without a chunk in an iteration only keeps the
// barrier count.
template <int NW, int NST, int SPLIT = 1>
__global__ __launch_bounds__(NW * SPLIT * 64) void attn_fwd_mfma(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr int CT = NW / 2, CROWS = CT * TILE, CBYTES = CT * TILE_BYTES, SBYTES = 2 * CBYTES;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = SPLIT == 1 ? 0 : wave_all / NW, wave = SPLIT == 1 ? wave_all : wave_all % NW;
    char* smem = smem_all + half * (NST * SBYTES);
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * (NW * 16) + wave * 16 + (lane & 15);
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * p.NQ * p.ldq + h * 64;
    const bool gather = p.kidx != nullptr;          // keys / values = rows kidx[b, :] of a cache of kidx_rows positions per sample
    const int krows = gather ? p.kidx_rows : p.NK;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * krows * p.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * krows * p.ldv + h * 64;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = frag_global(Q, q, p.NQ, p.ldq, ks, lane);
    ChunkDma<NW> lk, lv;
    lk.init(K, krows, p.ldk, wave, lane);
    lv.init(V, krows, p.ldv, wave, lane);
    // gathered keys: this sample's index list goes to LDS behind the rings first (plain loads, older than every DMA of the kernel
    // and waited for here: the counted waits of the loop see DMAs only)
    int* sidx = reinterpret_cast<int*>(smem_all + SPLIT * NST * SBYTES);
    if (gather) {
        const int32_t* gi = p.kidx + (size_t)b * p.NK;
        for (int i = tid; i < p.NK; i += NW * SPLIT * 64) sidx[i] = gi[i];
        __syncthreads();
    }
    const int nchunks_all = (p.NK + CROWS - 1) / CROWS;
    const int nchunks = SPLIT == 1 ? nchunks_all : (nchunks_all - half + SPLIT - 1) / SPLIT;      // this group's chunks: half, half + SPLIT, ...
    const int niter = (nchunks_all + SPLIT - 1) / SPLIT;                                            // barrier rounds of the workgroup
#pragma unroll
    for (int a = 0; a < NST; ++a)
        if (a < nchunks) {
            if (gather) {
                lk.issue_gather(smem + a * SBYTES, a * SPLIT + half, wave, lane, sidx, p.NK, p.ldk);
                lv.issue_gather(smem + a * SBYTES + CBYTES, a * SPLIT + half, wave, lane, sidx, p.NK, p.ldv);
            } else {
                lk.issue(smem + a * SBYTES, a * SPLIT + half, wave);
                lv.issue(smem + a * SBYTES + CBYTES, a * SPLIT + half, wave);
            }
        }
    const float c = 0.125f * LOG2E;     // 1/sqrt(64) folded with log2(e)
    const int mtiles = mebt_attn_dmask_tiles(p.NK);
    FragOffsets fo;
    fo.init(lane);

    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0, 0, 0, 0};
    float m = -INFINITY, l = 0.f;

    int st = 0;
    for (int it = 0; it < niter; ++it) {
        const int ci = it;                       // index among this group's chunks
        const int ch = it * SPLIT + half;        // chunk of the key set
        char* stage = smem + st * SBYTES;
        // this wave's pieces of chunk ch have landed; the (up to NST - 1) younger chunks may stay in flight
        const int younger = min(NST - 1, nchunks - 1 - ci);
        if (younger <= 0) wait_vm<0>();
        else if (younger == 1) wait_vm<DMA_PER_WAVE>();
        else if (younger == 2) wait_vm<2 * DMA_PER_WAVE>();
        else wait_vm<3 * DMA_PER_WAVE>();
        __builtin_amdgcn_s_barrier();                                          // ... and everybody else's
        const int nt = ci < nchunks ? min(CT, (p.NK - ch * CROWS + TILE - 1) / TILE) : 0;
        // TWO 64-key tiles per pass: their 16 score MFMAs are independent, and the softmax pays ONE maximum / sum exchange across
        // the lane groups, one rescale decision and one dependent chain per 128 keys instead of per 64 (the per-tile chain — reads,
        // 8 MFMAs, max, exchange, 16 exp, exchange, reads, 8 MFMAs — ran strictly in sequence: ~2000 cycles per tile and wave for
        // ~700 of work).  A missing second tile (odd count) is a tile of masked keys: the DMA zero-fills rows beyond NK.
        for (int t = 0; t < nt; t += 2) {
            const char* sK = stage + t * TILE_BYTES;
            const char* sV = stage + CBYTES + t * TILE_BYTES;
            // S^T[key][q] for the 128 keys of the two tiles
            f32x4 s[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    s[u][kb] = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) s[u][kb] = MFMA(fo.row_frag(sK + u * TILE_BYTES, kb, ks), qf[ks], s[u][kb]);
                }
            const int k0 = ch * CROWS + t * TILE;
            float tmax = -INFINITY;
            if (k0 + 2 * TILE > p.NK) {     // ragged end only: keys beyond NK are masked out
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (k0 + 64 * u + 16 * kb + 4 * g + r >= p.NK) s[u][kb][r] = -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[u][kb][r]);
            tmax = group_max(tmax) * c;                       // c > 0: max commutes with the scale
            const float mn = fmaxf(m, tmax);
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[u][kb][r] = fast_exp2(fmaf(s[u][kb][r], c, -mn)); ps += s[u][kb][r]; }
            ps = group_sum(ps);
            // the running maximum moves in the first tiles and then rarely: when no lane of the wave saw it move, alpha = exp2(0) = 1
            // exactly and the rescale of l and of the 16 output accumulators is skipped (same values, 17 vector instructions less)
            if (__builtin_amdgcn_ballot_w64(mn != m)) {
                const float alpha = fast_exp2(m - mn);
                l *= alpha;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= alpha;
            }
            l += ps;
            m = mn;
            if (p.drop.thresh) {   // attn_drop on the probabilities (gpt.py:135); the row sum above stays undropped
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (t + u >= nt) break;
                    const uint64_t dbase = (((uint64_t)b * p.H + h) * p.NQ + q) * p.NK + k0 + 64 * u;
                    uint32_t bits = 0;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const f32x4 keep = drop_keep4(p.drop, dbase + 16 * kb + 4 * g);
                        s[u][kb] *= keep;
#pragma unroll
                        for (int r = 0; r < 4; ++r) bits |= (keep[r] != 0.f ? 1u : 0u) << (4 * kb + r);
                    }
                    if (p.dmask && q < p.NQ)      // the keep bits of this lane's 16 elements: read back by the backward kernels
                        p.dmask[((((size_t)b * p.H + h) * p.NQ + q) * mtiles + ((k0 >> 6) + u)) * 4 + g] = (uint16_t)bits;
                }
            }
            // O^T[e][q] += V^T[e][key] P^T[key][q]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 pf = pack_acc(s[u][2 * kk], s[u][2 * kk + 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = MFMA(fo.col_frag(sV + u * TILE_BYTES, e, kk), pf, o[e]);
                }
        }
        if (SPLIT == 1 ? ci + NST < nchunks : it + NST < niter) {      // SPLIT: both groups keep the same barrier count
            __builtin_amdgcn_s_barrier();                     // every wave is done reading this stage
            if (ci + NST < nchunks) {
                if (gather) {
                    lk.issue_gather(stage, (ci + NST) * SPLIT + half, wave, lane, sidx, p.NK, p.ldk);
                    lv.issue_gather(stage + CBYTES, (ci + NST) * SPLIT + half, wave, lane, sidx, p.NK, p.ldv);
                } else {
                    lk.issue(stage, (ci + NST) * SPLIT + half, wave);
                    lv.issue(stage + CBYTES, (ci + NST) * SPLIT + half, wave);
                }
            }
        }
        if (++st == NST) st = 0;
    }
    if (SPLIT > 1) {                             // merge the groups: group 1 parks (m, l, O) in LDS, group 0 combines
        __syncthreads();                         // both rings are idle
        float* mg = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * 18;
        if (half == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) mg[4 * e + r] = o[e][r];
            mg[16] = m; mg[17] = l;
        }
        __syncthreads();
        if (half == 1) return;
        const float m1 = mg[16], l1 = mg[17];
        const float mn = fmaxf(m, m1);
        const float a0 = m == mn ? 1.f : fast_exp2(m - mn), a1 = m1 == mn ? 1.f : fast_exp2(m1 - mn);    // -inf == -inf cannot occur: NK > 0
        l = l * a0 + l1 * a1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[e][r] = o[e][r] * a0 + mg[4 * e + r] * a1;
        m = mn;
    }
    if (q < p.NQ) {
        const float inv = 1.0f / l;
        bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + ((size_t)b * p.NQ + q) * p.ldo + h * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) store4<bf16_t>(O + 16 * e + 4 * g, o[e] * inv);
        if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.NQ + q] = m * 0.6931471805599453f + logf(l);
    }
}

// ------------------------------------------------------------------------------------------------
// forward, long key sets on a grid of at most one workgroup per CU: two groups of four waves in ANTI-PHASE
// ------------------------------------------------------------------------------------------------
// Same decomposition as attn_fwd_mfma<4, 2, 2> (64 query rows per workgroup, group s = waves 4s .. 4s+3 takes the 128-key chunks
// s, s + 2, ... through a ring of its own, (m, l, O) merged through LDS at the end), different schedule.  There the two waves that
// share a SIMD (wave w of each group) ran the same part of a pass at the same time — both wait for their score MFMAs, both run the
// softmax's vector instructions, both wait for the P V MFMAs: ~6000 cycles per 128 keys for ~1700 of work per wave, with keys and
// values resident (tools/attn_layout_experiment.py).  Here a pass is cut into a MATRIX phase (O += V(i-1) P(i-1), then S(i) = K(i) Q
// and the row maxima of S(i): 32 MFMAs) and a VECTOR phase (softmax of S(i) to the packed P(i), rescale of O; the DMA of K(i+2) and
// V(i+1) is issued at its start), one workgroup barrier between phases, and group 1 runs one phase behind group 0: a SIMD's matrix
// pipe and vector ALU are each busy for one of its two waves in every phase.  P(i) is applied one pass late, so K(i) and V(i-1) are
// what a matrix phase reads; a stage's K half is refilled with K(i+2) and the other stage's V half with V(i+1) right after it.
__global__ __launch_bounds__(512) void attn_fwd_pp(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    constexpr int NW = 4, CROWS = 2 * TILE, CBYTES = 2 * TILE_BYTES, SBYTES = 2 * CBYTES;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave_all >> 2, wave = wave_all & 3;
    char* smem = smem_all + half * (2 * SBYTES);
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * (NW * 16) + wave * 16 + (lane & 15);
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * p.NQ * p.ldq + h * 64;
    const bool gather = p.kidx != nullptr;
    const int krows = gather ? p.kidx_rows : p.NK;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * krows * p.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * krows * p.ldv + h * 64;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = frag_global(Q, q, p.NQ, p.ldq, ks, lane);
    ChunkDma<NW> lk, lv;
    lk.init(K, krows, p.ldk, wave, lane);
    lv.init(V, krows, p.ldv, wave, lane);
    int* sidx = reinterpret_cast<int*>(smem_all + 4 * SBYTES);
    if (gather) {
        const int32_t* gi = p.kidx + (size_t)b * p.NK;
        for (int i = tid; i < p.NK; i += 512) sidx[i] = gi[i];
        __syncthreads();
    }
    const int nchunks_all = (p.NK + CROWS - 1) / CROWS;
    const int n = (nchunks_all - half + 1) / 2;                     // this group's chunks: half, half + 2, ...
    const int n0 = (nchunks_all + 1) / 2, n1 = nchunks_all / 2;
    const int nphase = max(2 * n0 + 1, 2 * n1 + 2);                // group 0: phases 0 .. 2 n0, group 1: 1 .. 2 n1 + 1
    auto issue_k = [&](int j) {
        if (gather) lk.issue_gather(smem + (j & 1) * SBYTES, 2 * j + half, wave, lane, sidx, p.NK, p.ldk);
        else lk.issue(smem + (j & 1) * SBYTES, 2 * j + half, wave);
    };
    auto issue_v = [&](int j) {
        if (gather) lv.issue_gather(smem + (j & 1) * SBYTES + CBYTES, 2 * j + half, wave, lane, sidx, p.NK, p.ldv);
        else lv.issue(smem + (j & 1) * SBYTES + CBYTES, 2 * j + half, wave);
    };
    if (0 < n) { issue_k(0); issue_v(0); }
    if (1 < n) issue_k(1);
    if (n <= 0) wait_vm<0>();                   // K(0) has landed (this wave's pieces); V(0) and K(1) may stay in flight
    else if (1 < n) wait_vm<8>();
    else wait_vm<4>();
    const float c = 0.125f * LOG2E;
    const int mtiles = mebt_attn_dmask_tiles(p.NK);
    FragOffsets fo;
    fo.init(lane);
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0, 0, 0, 0};
    float m = -INFINITY, l = 0.f, tmax = -INFINITY;
    f32x4 s[2][4];
    bf16x8 pf[2][2];

    for (int ph = 0; ph < nphase; ++ph) {
        __builtin_amdgcn_s_barrier();
        const int r = ph - half;
        if (r < 0 || r > 2 * n) continue;
        const int i = r >> 1;
        if (!(r & 1)) {
            // ---- matrix phase: O += V(i-1) P(i-1);  S(i) = K(i) Q and its row maxima.  All sixteen K fragments are requested first
            // (64 registers) and the V fragments of a 32-key step while the previous step's MFMAs run: the LDS round trips overlap the
            // matrix pipe instead of alternating with it (four reads, wait, four MFMAs was what the compiler made of the plain loop nest)
            bf16x8 kf[2][4][2];
            if (i < n) {
                const char* sK = smem + (i & 1) * SBYTES;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) kf[u][kb][ks] = fo.row_frag(sK + u * TILE_BYTES, kb, ks);
            }
            if (i > 0) {
                const char* sV = smem + ((i - 1) & 1) * SBYTES + CBYTES;
                bf16x8 vf[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[0][e] = fo.col_frag(sV, e, 0);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int u = st >> 1, kk = st & 1;
                    if (st + 1 < 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vf[(st + 1) & 1][e] = fo.col_frag(sV + ((st + 1) >> 1) * TILE_BYTES, e, (st + 1) & 1);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = MFMA(vf[st & 1][e], pf[u][kk], o[e]);
                }
            }
            if (i < n) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) s[u][kb] = MFMA(kf[u][kb][0], qf[0], (f32x4{0, 0, 0, 0}));
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) s[u][kb] = MFMA(kf[u][kb][1], qf[1], s[u][kb]);
                const int k0 = (2 * i + half) * CROWS;
                if (k0 + CROWS > p.NK) {        // ragged end only: keys beyond NK are masked out
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr)
                                if (k0 + 64 * u + 16 * kb + 4 * g + rr >= p.NK) s[u][kb][rr] = -INFINITY;
                }
                float t = -INFINITY;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) t = fmaxf(t, s[u][kb][rr]);
                tmax = group_max(t) * c;
            }
        } else {
            // ---- vector phase: refill the halves the matrix phase has just released, softmax of S(i)
            const bool more_v = i + 1 < n, more_k = i + 2 < n;
            if (more_v) issue_v(i + 1);
            if (more_k) issue_k(i + 2);
            const int k0 = (2 * i + half) * CROWS;
            const float mn = fmaxf(m, tmax);
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) { s[u][kb][rr] = fast_exp2(fmaf(s[u][kb][rr], c, -mn)); ps += s[u][kb][rr]; }
            ps = group_sum(ps);
            if (__builtin_amdgcn_ballot_w64(mn != m)) {
                const float alpha = fast_exp2(m - mn);
                l *= alpha;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= alpha;
            }
            l += ps;
            m = mn;
            if (p.drop.thresh) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (k0 + 64 * u >= p.NK) break;
                    const uint64_t dbase = (((uint64_t)b * p.H + h) * p.NQ + q) * p.NK + k0 + 64 * u;
                    uint32_t bits = 0;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const f32x4 keep = drop_keep4(p.drop, dbase + 16 * kb + 4 * g);
                        s[u][kb] *= keep;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) bits |= (keep[rr] != 0.f ? 1u : 0u) << (4 * kb + rr);
                    }
                    if (p.dmask && q < p.NQ)
                        p.dmask[((((size_t)b * p.H + h) * p.NQ + q) * mtiles + ((k0 >> 6) + u)) * 4 + g] = (uint16_t)bits;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) pf[u][kk] = pack_acc(s[u][2 * kk], s[u][2 * kk + 1]);
            // what the next matrix phase reads — K(i+1), V(i) — is older than this phase's refills
            if (more_v && more_k) wait_vm<8>();
            else if (more_v || more_k) wait_vm<4>();
            else wait_vm<0>();
        }
    }
    // merge the groups: group 1 parks (m, l, O) in LDS, group 0 combines
    __syncthreads();
    float* mg = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * 18;
    if (half == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) mg[4 * e + rr] = o[e][rr];
        mg[16] = m; mg[17] = l;
    }
    __syncthreads();
    if (half == 1) return;
    {
        const float m1 = mg[16], l1 = mg[17];
        const float mn = fmaxf(m, m1);
        const float a0 = m == mn ? 1.f : fast_exp2(m - mn), a1 = m1 == mn ? 1.f : fast_exp2(m1 - mn);
        l = l * a0 + l1 * a1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) o[e][rr] = o[e][rr] * a0 + mg[4 * e + rr] * a1;
        m = mn;
    }
    if (q < p.NQ) {
        const float inv = 1.0f / l;
        bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + ((size_t)b * p.NQ + q) * p.ldo + h * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) store4<bf16_t>(O + 16 * e + 4 * g, o[e] * inv);
        if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.NQ + q] = m * 0.6931471805599453f + logf(l);
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ (+ delta = rowsum(dO * O)).  One wave = 16 query rows; K/V chunks streamed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dq_mfma(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * BLOCK_ROWS + wave * 16 + (lane & 15);
    const bool qv = q < p.NQ;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * p.NQ * p.ldq + h * 64;
    const bf16_t* G = reinterpret_cast<const bf16_t*>(p.d_o) + (size_t)b * p.NQ * p.lddo + h * 64;
    const bf16_t* Oo = reinterpret_cast<const bf16_t*>(p.o) + (size_t)b * p.NQ * p.ldo + h * 64;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * p.NK * p.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * p.NK * p.ldv + h * 64;
    ChunkDma<> lk, lv;
    lk.init(K, p.NK, p.ldk, wave, lane);
    lv.init(V, p.NK, p.ldv, wave, lane);
    const int nchunks = (p.NK + CHUNK - 1) / CHUNK;
    // the wave's own operands first (oldest in the vmcnt queue), then the chunk copies; nothing else touches
    // memory until the epilogue, so "at most DMA_PER_WAVE outstanding" means "chunk ch has landed"
    bf16x8 qf[2], gf[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = frag_global(Q, q, p.NQ, p.ldq, ks, lane);
        gf[ks] = frag_global(G, q, p.NQ, p.lddo, ks, lane);
        of[ks] = frag_global(Oo, q, p.NQ, p.ldo, ks, lane);
    }
    const size_t sidx = ((size_t)b * p.H + h) * p.NQ + q;
    const float lse_q = qv ? p.lse[sidx] : 0.f;
    // keep bits of (this workgroup's 128 queries) x (the chunk's 4 key tiles): 32 B per query = 4 KiB per chunk = four 1-KiB DMA
    // pieces, issued by waves 0-3 with the chunk (always: a null / zero-sized mask reads as zeros without traffic, and the
    // counted waits stay the same); piece `wave`, lane l -> query 32 wave + l / 2, 16-byte half l & 1
    const bool use_bits = p.drop.thresh && p.dmask != nullptr;
    const int mtiles = mebt_attn_dmask_tiles(p.NK);
    const size_t mrow0 = ((size_t)b * p.H + h) * p.NQ;
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(p.dmask ? p.dmask + mrow0 * mtiles * 4 : (const uint16_t*)p.q, use_bits ? (size_t)p.NQ * mtiles * 8 : 0);
    const uint32_t moff = (uint32_t)((blockIdx.x * BLOCK_ROWS + 32 * (wave & 3) + (lane >> 1)) * mtiles * 8 + (lane & 1) * 16);
    auto issue_mask = [&](char* stage, int chunk) {
        if (wave < 4) dma16(rmask, lds_addr_of(stage + MASK_OFF + wave * 1024), moff + (uint32_t)chunk * 32);
    };
    lk.issue(smem, 0, wave);
    lv.issue(smem + CHUNK_BYTES, 0, wave);
    issue_mask(smem, 0);
    if (nchunks > 1) {
        lk.issue(smem + STAGE_BYTES, 1, wave);
        lv.issue(smem + STAGE_BYTES + CHUNK_BYTES, 1, wave);
        issue_mask(smem + STAGE_BYTES, 1);
    }
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)gf[ks][j] * (float)of[ks][j];
    delta = group_sum(delta);
    const float lse2 = lse_q * LOG2E;
    const float c = 0.125f * LOG2E;
    FragOffsets fo;
    fo.init(lane);

    f32x4 dq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dq[e] = f32x4{0, 0, 0, 0};
    const uint64_t drow = (((uint64_t)b * p.H + h) * p.NQ + q) * p.NK;     // dropout index of (q, key 0)
    const int mlds = (wave * 16 + (lane & 15)) * 32 + g * 2;                // this lane's field of tile 0 in the staged mask
    for (int ch = 0; ch < nchunks; ++ch) {
        char* stage = smem + (ch & 1) * STAGE_BYTES;
        if (ch + 1 < nchunks) { if (wave < 4) wait_vm<DMA_PER_WAVE + 1>(); else wait_vm<DMA_PER_WAVE>(); } else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const int nt = min(CHUNK_TILES, (p.NK - ch * CHUNK + TILE - 1) / TILE);
        // TWO key tiles per pass, unconditionally (a chunk always holds four tile images; rows beyond NK are zero-filled and their
        // probabilities forced to 0): 16 independent score / dP chains and 32 exponentials in flight instead of a strictly
        // sequential reads -> MFMA -> exp -> pack -> MFMA chain per tile (as in the forward)
        for (int t = 0; t < nt; t += 2) {
            f32x4 ds[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* sK = stage + (t + u) * TILE_BYTES;
                const char* sV = stage + CHUNK_BYTES + (t + u) * TILE_BYTES;
                const int k0 = ch * CHUNK + (t + u) * TILE;
                const uint32_t bits = use_bits ? *reinterpret_cast<const uint16_t*>(stage + MASK_OFF + mlds + (t + u) * 8) : 0u;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sc = MFMA(fo.row_frag(sK, kb, ks), qf[ks], sc);     // S^T[key][q]
                        dp = MFMA(fo.row_frag(sV, kb, ks), gf[ks], dp);    // dP^T[key][q] = V dO^T
                    }
                    f32x4 keep = {1.f, 1.f, 1.f, 1.f};
                    if (use_bits) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) keep[r] = ((bits >> (4 * kb + r)) & 1u) ? p.drop.inv_keep : 0.f;
                    } else if (p.drop.thresh) {
                        keep = drop_keep4(p.drop, drow + (uint64_t)(k0 + 16 * kb + 4 * g));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + 16 * kb + 4 * g + r;
                        const float pr = key < p.NK ? fast_exp2(fmaf(sc[r], c, -lse2)) : 0.f;
                        ds[u][kb][r] = pr * (dp[r] * keep[r] - delta) * 0.125f;
                    }
                }
            }
            // dQ^T[e][q] += K^T[e][key] dS^T[key][q]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 df = pack_acc(ds[u][2 * kk], ds[u][2 * kk + 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[e] = MFMA(fo.col_frag(stage + (t + u) * TILE_BYTES, e, kk), df, dq[e]);
                }
        }
        if (ch + 2 < nchunks) {
            __builtin_amdgcn_s_barrier();
            lk.issue(stage, ch + 2, wave);
            lv.issue(stage + CHUNK_BYTES, ch + 2, wave);
            issue_mask(stage, ch + 2);
        }
    }
    if (qv) {
        bf16_t* D = reinterpret_cast<bf16_t*>(p.dq) + ((size_t)b * p.NQ + q) * p.lddq + h * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) store4<bf16_t>(D + 16 * e + 4 * g, dq[e]);
        if (g == 0) p.delta[sidx] = delta;
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV.  One wave = 16 key rows; Q / dO chunks (+ lse, delta) streamed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dkv_mfma(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int key = blockIdx.x * BLOCK_ROWS + wave * 16 + (lane & 15);
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * p.NQ * p.ldq + h * 64;
    const bf16_t* G = reinterpret_cast<const bf16_t*>(p.d_o) + (size_t)b * p.NQ * p.lddo + h * 64;
    const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * p.NK * p.ldk + h * 64;
    const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * p.NK * p.ldv + h * 64;
    const float* L = p.lse + ((size_t)b * p.H + h) * p.NQ;
    const float* Dl = p.delta + ((size_t)b * p.H + h) * p.NQ;
    ChunkDma<> lq, lg;
    lq.init(Q, p.NQ, p.ldq, wave, lane);
    lg.init(G, p.NQ, p.lddo, wave, lane);
    // lse / delta of a chunk: 256 floats = one 1-KiB piece each, copied by waves 0 and 1 (zero beyond NQ)
    const __amdgpu_buffer_rsrc_t rstat = make_rsrc(wave == 0 ? L : Dl, (size_t)p.NQ * 4);
    const int nchunks = (p.NQ + CHUNK - 1) / CHUNK;
    // keep bits of (the chunk's 256 queries) x (this workgroup's 2 key tiles): 16 B per query = 4 KiB per chunk = four 1-KiB DMA
    // pieces issued by waves 0-3 (always, see the dQ kernel); piece `wave`, lane l -> query 64 wave + l of the chunk
    const bool use_bits = p.drop.thresh && p.dmask != nullptr;
    const int mtiles = mebt_attn_dmask_tiles(p.NK);
    const size_t mrow0 = ((size_t)b * p.H + h) * p.NQ;
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(p.dmask ? p.dmask + mrow0 * mtiles * 4 : (const uint16_t*)p.q, use_bits ? (size_t)p.NQ * mtiles * 8 : 0);
    const uint32_t moff = (uint32_t)((64 * (wave & 3) + lane) * mtiles * 8 + blockIdx.x * 16);
    auto issue = [&](char* stage, int chunk) {
        lq.issue(stage, chunk, wave);
        lg.issue(stage + CHUNK_BYTES, chunk, wave);
        if (wave < 2) dma16(rstat, lds_addr_of(stage + 2 * CHUNK_BYTES + wave * STAT_BYTES), (uint32_t)(chunk * CHUNK * 4 + lane * 16));
        if (wave < 4) dma16(rmask, lds_addr_of(stage + MASK_OFF + wave * 1024), moff + (uint32_t)chunk * CHUNK * (uint32_t)mtiles * 8);
    };
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = frag_global(K, key, p.NK, p.ldk, ks, lane);
        vf[ks] = frag_global(V, key, p.NK, p.ldv, ks, lane);
    }
    issue(smem, 0);
    if (nchunks > 1) issue(smem + STAGE_BYTES, 1);
    const float c = 0.125f * LOG2E;
    FragOffsets fo;
    fo.init(lane);

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { dk[e] = f32x4{0, 0, 0, 0}; dv[e] = f32x4{0, 0, 0, 0}; }
    const uint64_t dcol = ((uint64_t)b * p.H + h) * p.NQ * p.NK + key;      // dropout index of (query 0, key)
    for (int ch = 0; ch < nchunks; ++ch) {
        char* stage = smem + (ch & 1) * STAGE_BYTES;
        if (ch + 1 < nchunks) {
            if (wave < 2) wait_vm<DMA_PER_WAVE + 2>(); else if (wave < 4) wait_vm<DMA_PER_WAVE + 1>(); else wait_vm<DMA_PER_WAVE>();
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        const float* cl = reinterpret_cast<const float*>(stage + 2 * CHUNK_BYTES);
        const float* cd = cl + CHUNK;
        // this lane's key is bit 4 (wave & 3) + (lane & 3) of field [q][key tile wave >> 2][(lane >> 2) & 3]
        const char* cm = stage + MASK_OFF + (wave >> 2) * 8 + ((lane >> 2) & 3) * 2;
        const int mbit = 4 * (wave & 3) + (lane & 3);
        const int nt = min(CHUNK_TILES, (p.NQ - ch * CHUNK + TILE - 1) / TILE);
        // two query tiles per pass, unconditionally (see the dQ kernel; padded query rows are zero rows of Q and dO: they add nothing)
        for (int t = 0; t < nt; t += 2) {
            f32x4 pr[2][4], ds[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* sQ = stage + (t + u) * TILE_BYTES;
                const char* sG = stage + CHUNK_BYTES + (t + u) * TILE_BYTES;
                const int q0 = ch * CHUNK + (t + u) * TILE;
#pragma unroll
                for (int qb = 0; qb < 4; ++qb) {
                    f32x4 sc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        sc = MFMA(fo.row_frag(sQ, qb, ks), kf[ks], sc);     // S[q][key]
                        dp = MFMA(fo.row_frag(sG, qb, ks), vf[ks], dp);    // dP[q][key] = dO V^T
                    }
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(cl + (t + u) * TILE + 16 * qb + 4 * g);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(cd + (t + u) * TILE + 16 * qb + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = q0 + 16 * qb + 4 * g + r;
                        // padded query rows (qi >= NQ): Q and dO rows are zero-filled, so whatever p is they add nothing
                        const float pv = fast_exp2(fmaf(sc[r], c, -l4[r] * LOG2E));
                        float keep = 1.0f;
                        if (use_bits) keep = ((*reinterpret_cast<const uint16_t*>(cm + ((t + u) * TILE + 16 * qb + 4 * g + r) * 16) >> mbit) & 1u) ? p.drop.inv_keep : 0.f;
                        else if (p.drop.thresh) keep = drop_keep(p.drop, dcol + (uint64_t)(uint32_t)qi * (uint32_t)p.NK);
                        ds[u][qb][r] = pv * (dp[r] * keep - d4[r]) * 0.125f;
                        pr[u][qb][r] = pv * keep;
                    }
                }
            }
            // dV^T[e][key] += dO^T[e][q] P[q][key];   dK^T[e][key] += Q^T[e][q] dS[q][key]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 pf = pack_acc(pr[u][2 * kk], pr[u][2 * kk + 1]);
                    const bf16x8 df = pack_acc(ds[u][2 * kk], ds[u][2 * kk + 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dv[e] = MFMA(fo.col_frag(stage + CHUNK_BYTES + (t + u) * TILE_BYTES, e, kk), pf, dv[e]);
                        dk[e] = MFMA(fo.col_frag(stage + (t + u) * TILE_BYTES, e, kk), df, dk[e]);
                    }
                }
        }
        if (ch + 2 < nchunks) {
            __builtin_amdgcn_s_barrier();
            issue(stage, ch + 2);
        }
    }
    if (key < p.NK) {
        bf16_t* DK = reinterpret_cast<bf16_t*>(p.dk) + ((size_t)b * p.NK + key) * p.lddk + h * 64;
        bf16_t* DV = reinterpret_cast<bf16_t*>(p.dv) + ((size_t)b * p.NK + key) * p.lddv + h * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            store4<bf16_t>(DK + 16 * e + 4 * g, dk[e]);
            store4<bf16_t>(DV + 16 * e + 4 * g, dv[e]);
        }
    }
}

}  // namespace

static int check_layout(const AttnParams& p) {
    if (p.HD != 64) { mebt_set_error("mfma attention: head size must be 64"); return MEBT_ESHAPE; }
    if ((p.ldq | p.ldk | p.ldv | p.ldo) % 8) { mebt_set_error("mfma attention: row strides must be multiples of 8 elements"); return MEBT_ESHAPE; }
    static bool inited = false;
    if (!inited) {
        // the forward kernels may carry a key index list behind their rings (AttnParams::kidx): the whole LDS is admissible
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma<4, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS));
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS));
        inited = true;
    }
    return MEBT_OK;
}
// one stage is enough when the streamed side fits one chunk: two workgroups per CU
static int lds_bytes(int streamed_rows) { return streamed_rows <= CHUNK ? STAGE_BYTES : ATTN_LDS; }

int launch_attn_fwd_mfma(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    drop_mark_small(p.drop, (uint64_t)p.B * p.H * p.NQ * p.NK);
    if (int rc = check_layout(p)) return rc;
    // 128-row workgroups (8 waves) unless their grid leaves CUs idle; then 64-row workgroups (4 waves, twice the workgroups).
    // MEBT_ATTN_FWD_WAVES = 4 | 8 forces one form (A/B runs).
    static const int force = [] { const char* e = getenv("MEBT_ATTN_FWD_WAVES"); return e ? atoi(e) : 0; }();
    // measured (tools/attn_bench.py, profiles/r04_attention_forward.txt): with 192 workgroups of 128 rows (config 2, batch 6) the 8-wave
    // form wins on every routing (10.7 vs 15.0 us at 512 keys); with 128 (batch 4) the 4-wave form does: 93 vs 112 us at 7936 keys
    const long grid8 = (long)((p.NQ + 127) / 128) * p.H * p.B;
    const bool four = force ? force == 4 : grid8 <= 128;
    const int xl = p.kidx ? ((p.NK * 4 + 15) & ~15) : 0;       // the key index list of a sample, staged behind the rings
    if (xl && xl + 4 * CHUNK_BYTES > 160 * 1024) { mebt_set_error("mfma attention: a gathered key set holds at most 8192 keys"); return MEBT_ESHAPE; }
    if (four) {
        const dim3 grid((p.NQ + 63) / 64, p.H, p.B);
        // short key sets: two stages of 128 keys (64 KiB: two workgroups per CU); long ones: four stages, one workgroup per CU
        static const int split_on = [] { const char* e = getenv("MEBT_ATTN_FWD_SPLIT"); return e ? atoi(e) : 1; }();
        if (p.NK <= 1024) hipLaunchKernelGGL((attn_fwd_mfma<4, 2>), grid, dim3(256), 2 * CHUNK_BYTES + xl, stream, p);
        else if (split_on && (long)grid.x * p.H * p.B <= 256) {    // one workgroup per CU at most: a second group of waves per query block
            if (split_on == 2) hipLaunchKernelGGL((attn_fwd_mfma<4, 2, 2>), grid, dim3(512), 4 * CHUNK_BYTES + xl, stream, p);   // both groups in phase (A/B)
            else hipLaunchKernelGGL(attn_fwd_pp, grid, dim3(512), 4 * CHUNK_BYTES + xl, stream, p);
        }
        else hipLaunchKernelGGL((attn_fwd_mfma<4, 4>), grid, dim3(256), 4 * CHUNK_BYTES + xl, stream, p);
    } else {
        const dim3 grid((p.NQ + BLOCK_ROWS - 1) / BLOCK_ROWS, p.H, p.B);
        if (p.NK <= CHUNK) hipLaunchKernelGGL((attn_fwd_mfma<8, 1>), grid, dim3(WAVES * 64), 2 * CHUNK_BYTES + xl, stream, p);
        else hipLaunchKernelGGL((attn_fwd_mfma<8, 2>), grid, dim3(WAVES * 64), 2 * 2 * CHUNK_BYTES + xl, stream, p);
    }
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}

int launch_attn_bwd_mfma(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    drop_mark_small(p.drop, (uint64_t)p.B * p.H * p.NQ * p.NK);
    if (int rc = check_layout(p)) return rc;
    if ((p.lddo | p.lddq | p.lddk | p.lddv) % 8) { mebt_set_error("mfma attention: row strides must be multiples of 8 elements"); return MEBT_ESHAPE; }
    const dim3 gq((p.NQ + BLOCK_ROWS - 1) / BLOCK_ROWS, p.H, p.B);
    hipLaunchKernelGGL(attn_bwd_dq_mfma, gq, dim3(WAVES * 64), lds_bytes(p.NK), stream, p);
    const dim3 gk((p.NK + BLOCK_ROWS - 1) / BLOCK_ROWS, p.H, p.B);
    hipLaunchKernelGGL(attn_bwd_dkv_mfma, gk, dim3(WAVES * 64), lds_bytes(p.NQ), stream, p);
    MEBT_HIP_CHECK(hipGetLastError());
    return MEBT_OK;
}
