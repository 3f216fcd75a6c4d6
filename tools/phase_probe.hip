// Diagnostic (not product): WHICH resource do the two halves of the fused weight-gradient launch share?
// The launch is (a) k-loops filling LDS from L2 / Infinity Cache (1-2 GB per launch through the CUs' vector memory path at
// 25-33 B/clk/CU) and (b) the AdamW stream (327 MB of HBM traffic).  Three earlier builds put (a) and (b) side by side on every CU
// (split roles, a second stream, early loads: profiles/r03_wgrad_adamw_overlap_experiments.txt, r05_early_epilogue_loads_ab.txt)
// and the launch took the SUM each time.  This probe runs the two access patterns of the real kernel - LDS-DMA k-tile fills of an
// L2-resident operand and a read-modify-write pass over HBM-cold p / m / v (+ bf16 mirror) - as PHASES of 768 persistent
// workgroups (3 per CU), and only changes WHO is in which phase at the same time:
//   mode 0  fill only                    mode 1  stream only
//   mode 2  every workgroup F S F S      (all CUs in the same phase: today's launch)
//   mode 3  mixed inside every CU        (slot parity on the CU: F S F S beside S F S F - what the split-role builds did)
//   mode 4  CU parity                    (all three workgroups of a CU in the same phase, neighbouring CUs in opposite phases)
//   mode 5  XCD parity                   (whole XCDs in opposite phases)
//   mode 6  SE parity
// If 3 ~ 2 (sum) and 4 or 5 ~ max(0, 1), the shared resource is per CU (the in-order vector memory return path: an L2 hit
// queued behind an HBM miss takes the miss's latency) resp. per XCD, and the launch can overlap its halves by de-phasing CUs.
// Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mebt_amd/csrc -I include tools/phase_probe.hip -o tools/bin/phase_probe && tools/bin/phase_probe
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

void mebt_set_hip_error(hipError_t, const char*) {}
void mebt_set_error(const char*) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

constexpr int TILE_ELEMS = 128 * 64;                 // parameters per tile (one workgroup's AdamW epilogue)
constexpr int KT_BYTES = (128 + 64) * 64 * 2;        // one k-tile of a 128 x 64 product: 24 KiB
constexpr int RING = 2;

struct Args {
    const char* fill;            // [8 XCDs][region] L2-resident operand bytes
    unsigned region;             // bytes per XCD region
    float* p; float* m; float* v; unsigned short* lp;   // [rounds * grid * TILE_ELEMS]
    int rounds, ktiles, mode;
    unsigned* cu_slots;          // [8 * 8 * 2 * 16] zeroed: slot counter per CU
    unsigned long long* stamps;  // [grid][2 * rounds * 2]
    unsigned* census;            // [grid]: (xcc << 16) | hw id bits
    unsigned* sink;
    unsigned long long* clocks;  // [4]: s_memtime (shader clock) and s_memrealtime (100 MHz) at entry and exit of workgroup 0
};

__device__ __forceinline__ void fill_phase(const Args& a, char* smem, int xcc, int wg_in_xcd, int round, int wave, int lane) {
    // the k-loop's memory side: NP = 6 LDS-DMA pieces of 1 KiB per wave per k-tile into a ring of 2, counted waits, raw barrier
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.fill + (size_t)xcc * a.region, a.region);
    constexpr int NP = KT_BYTES / 1024 / 4;
    const unsigned span = a.region - KT_BYTES;
    unsigned base = ((unsigned)(wg_in_xcd * 7 + round * 3) * (unsigned)KT_BYTES * 5u) % span;
    base &= ~1023u;
    auto issue = [&](int t, int slot) {
        const unsigned off = (base + (unsigned)t * KT_BYTES) % span;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            dma16(rsrc, lds_addr_of(smem + slot * KT_BYTES + (4 * i + wave) * 1024), (off & ~1023u) + (4 * i + wave) * 1024 + lane * 16);
    };
    issue(0, 0);
    unsigned acc = 0;
    for (int t = 0; t < a.ktiles; ++t) {
        if (t + 1 < a.ktiles) { issue(t + 1, (t + 1) & 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<const unsigned*>(smem + (t & 1) * KT_BYTES + threadIdx.x * 4);
        __builtin_amdgcn_s_barrier();
    }
    if (a.sink && acc == 0x12345678u) *a.sink = acc;
}

__device__ __forceinline__ void stream_phase(const Args& a, int round) {
    // the AdamW epilogue's memory side: 12 B read + 14 B written per parameter, 16-byte accesses, whole rows of 256 B per 16 lanes
    const size_t t0 = ((size_t)round * gridDim.x + blockIdx.x) * TILE_ELEMS;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        f32x4 p[4], m[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = t0 + ((size_t)(c * 4 + i) * 256 + threadIdx.x) * 4;
            p[i] = *reinterpret_cast<const f32x4*>(a.p + e);
            m[i] = *reinterpret_cast<const f32x4*>(a.m + e);
            v[i] = *reinterpret_cast<const f32x4*>(a.v + e);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t e = t0 + ((size_t)(c * 4 + i) * 256 + threadIdx.x) * 4;
            m[i] = m[i] * 0.9f + p[i] * 0.1f;
            v[i] = v[i] * 0.95f + p[i] * p[i] * 0.05f;
            p[i] = p[i] - m[i] * 1e-9f;
            *reinterpret_cast<f32x4*>(a.p + e) = p[i];
            *reinterpret_cast<f32x4*>(a.m + e) = m[i];
            *reinterpret_cast<f32x4*>(a.v + e) = v[i];
            typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
            const u16x4 o = {(unsigned short)(__builtin_bit_cast(unsigned, p[i][0]) >> 16), (unsigned short)(__builtin_bit_cast(unsigned, p[i][1]) >> 16),
                             (unsigned short)(__builtin_bit_cast(unsigned, p[i][2]) >> 16), (unsigned short)(__builtin_bit_cast(unsigned, p[i][3]) >> 16)};
            *reinterpret_cast<u16x4*>(a.lp + e) = o;
        }
    }
}

__global__ __launch_bounds__(256) void probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_slot;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc &= 7;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (threadIdx.x == 0) {
        s_slot = atomicAdd(a.cu_slots + (((xcc * 8 + se) * 2 + sh) * 16 + cu), 1u);
        a.census[blockIdx.x] = (xcc << 16) | (hw & 0xFF00);
    }
    __syncthreads();
    const unsigned slot = s_slot;
    int grp = 0;
    if (a.mode == 3) grp = slot & 1;
    else if (a.mode == 4) grp = (cu ^ se) & 1;
    else if (a.mode == 5) grp = xcc & 1;
    else if (a.mode == 6) grp = se & 1;
    unsigned long long* st = a.stamps + (size_t)blockIdx.x * (4 * a.rounds);
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.clocks[0] = __builtin_amdgcn_s_memtime(); a.clocks[1] = __builtin_amdgcn_s_memrealtime(); }
    for (int r = 0; r < a.rounds; ++r) {
        const bool do_fill = a.mode != 1, do_stream = a.mode != 0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1, t2;
        if (grp == 0) {
            if (do_fill) fill_phase(a, smem, xcc, blockIdx.x >> 3, r, wave, lane);
            t1 = __builtin_amdgcn_s_memtime();
            if (do_stream) stream_phase(a, r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) { st[4 * r] = t0; st[4 * r + 1] = t1; st[4 * r + 2] = t1; st[4 * r + 3] = t2; }
        } else {
            if (do_stream) stream_phase(a, r);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            if (do_fill) fill_phase(a, smem, xcc, blockIdx.x >> 3, r, wave, lane);
            t2 = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) { st[4 * r] = t1; st[4 * r + 1] = t2; st[4 * r + 2] = t0; st[4 * r + 3] = t1; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.clocks[2] = __builtin_amdgcn_s_memtime(); a.clocks[3] = __builtin_amdgcn_s_memrealtime(); }
}

}  // namespace

int main(int argc, char** argv) {
    const int grid = 768, rounds = argc > 1 ? atoi(argv[1]) : 2, ktiles = argc > 2 ? atoi(argv[2]) : 48, reps = 8, pool = 4;
    const unsigned region = 3u << 20;              // 3 MiB per XCD: resident in its 4 MiB L2 after the first pass
    char* fill;
    CK(hipMalloc(&fill, (size_t)8 * region));
    CK(hipMemset(fill, 1, (size_t)8 * region));
    const size_t per = (size_t)rounds * grid * TILE_ELEMS;
    float *p, *m, *v; unsigned short* lp;
    CK(hipMalloc(&p, per * 4 * pool)); CK(hipMalloc(&m, per * 4 * pool)); CK(hipMalloc(&v, per * 4 * pool)); CK(hipMalloc(&lp, per * 2 * pool));
    CK(hipMemset(p, 0, per * 4 * pool)); CK(hipMemset(m, 0, per * 4 * pool)); CK(hipMemset(v, 0, per * 4 * pool));
    unsigned* slots; unsigned long long* stamps; unsigned* census; unsigned long long* clocks;
    CK(hipMalloc(&clocks, 4 * 8));
    CK(hipMalloc(&slots, 8 * 8 * 2 * 16 * 4));
    CK(hipMalloc(&stamps, (size_t)grid * 4 * rounds * 8));
    CK(hipMalloc(&census, grid * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, RING * KT_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %d workgroups x %d rounds; fill = %d k-tiles of %d KiB per round (%.2f GB per launch through LDS-DMA, L2-resident), stream = %.0f MB of p/m/v/mirror traffic per launch (HBM-cold, %d sets rotated)\n",
           grid, rounds, ktiles, KT_BYTES / 1024, (double)grid * rounds * ktiles * KT_BYTES / 1e9, (double)per * 26 / 1e6, pool);
    const char* names[7] = {"fill only", "stream only", "all in phase (F S F S)", "mixed inside each CU (slot parity)", "CU parity", "XCD parity", "SE parity"};
    double base[2] = {0, 0};
    for (int mode = 0; mode < 7; ++mode) {
        float best = 1e30f, sum = 0;
        std::vector<unsigned long long> hs((size_t)grid * 4 * rounds);
        for (int it = 0; it < reps + 2; ++it) {
            Args a;
            a.fill = fill; a.region = region;
            const size_t o = (size_t)(it % pool) * per;
            a.p = p + o; a.m = m + o; a.v = v + o; a.lp = lp + o;
            a.rounds = rounds; a.ktiles = ktiles; a.mode = mode; a.cu_slots = slots; a.stamps = stamps; a.census = census; a.sink = nullptr; a.clocks = clocks;
            CK(hipMemsetAsync(slots, 0, 8 * 8 * 2 * 16 * 4, 0));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), RING * KT_BYTES, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { best = std::min(best, ms); sum += ms; }
        }
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned> hc(grid), hslots(8 * 8 * 2 * 16);
        CK(hipMemcpy(hc.data(), census, grid * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hslots.data(), slots, hslots.size() * 4, hipMemcpyDeviceToHost));
        int cus = 0, full = 0;
        for (unsigned s : hslots) { cus += s > 0; full += s == 3; }
        double f = 0, s = 0;
        for (int w = 0; w < grid; ++w)
            for (int r = 0; r < rounds; ++r) {
                f += (double)(hs[(size_t)w * 4 * rounds + 4 * r + 1] - hs[(size_t)w * 4 * rounds + 4 * r]);
                s += (double)(hs[(size_t)w * 4 * rounds + 4 * r + 3] - hs[(size_t)w * 4 * rounds + 4 * r + 2]);
            }
        unsigned long long hk[4];
        CK(hipMemcpy(hk, clocks, sizeof hk, hipMemcpyDeviceToHost));
        const double ghz = (double)(hk[2] - hk[0]) / (double)(hk[3] - hk[1]) * 0.1;     // shader-clock ticks per 100 MHz reference tick
        f /= (double)grid * rounds * ghz * 1e3; s /= (double)grid * rounds * ghz * 1e3;   // s_memtime ticks -> us at the measured clock
        if (mode < 2) base[mode] = sum / reps;
        printf("mode %d  %-36s  %7.1f us avg  %7.1f us best   per workgroup and round: fill phase %6.1f us, stream phase %6.1f us   shader clock during the last launch %.2f GHz   (%d CUs used, %d with 3 workgroups)\n",
               mode, names[mode], sum / reps * 1e3, best * 1e3, f, s, ghz, cus, full);
    }
    printf("# sum of the halves %.1f us, max %.1f us\n", (base[0] + base[1]) * 1e3, std::max(base[0], base[1]) * 1e3);
    return 0;
}
