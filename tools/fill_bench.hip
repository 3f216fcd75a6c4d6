// Microbenchmark (diagnostics, not product): how fast can the CUs pull GEMM operand k-tiles from L2 / Infinity Cache into LDS?
// The main loops of gemm_kernels.h WITHOUT fragment reads and MFMAs: same tile -> workgroup mapping, same LDS-DMA / register
// staging instructions, same counted waits and barriers.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mebt_amd/csrc -I include tools/fill_bench.hip -o /tmp/fill_bench && /tmp/fill_bench
#include "gemm_kernels.h"
#include <cstdio>
#include <vector>

void mebt_set_hip_error(hipError_t, const char*) {}
void mebt_set_error(const char*) {}

namespace {

// MODE 0: LDS-DMA ring (NSTAGE deep, AHEAD = NSTAGE - 1 tiles in flight), one s_barrier per k-tile when BARRIER
// MODE 1: register staging, one tile ahead in VGPRs, two LDS stages (gemm_tile_regstaged's loop)
// MODE 2: plain global loads into registers only, `NSTAGE` tiles in flight, nothing written to LDS (ceiling of the load path)
// WORK (MODE 0, 4 waves as 2 x 2): bit 0 = the fragment reads of the real main loop, bit 1 = its MFMAs (on whatever the registers hold
// when the reads are off): which of the two is it that slows the fill down?
template <int MODE, int NW, int TBM, int TBN, int NSTAGE, bool BARRIER, int WORK = 0>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const bf16_t* A, const bf16_t* B, int M, int N, int K, unsigned* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (TBM + TBN) * BK * 2;
    const int ntx = gridDim.x, nty = gridDim.y;
    int tr, tc;
    xcd_tile(blockIdx.y * ntx + blockIdx.x, ntx, nty, M, N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K / BK;
    unsigned acc = 0;
    if constexpr (MODE == 0) {
        constexpr int LPT = (TBM + TBN) / (8 * NW);
        constexpr int AHEAD = NSTAGE - 1;
        DmaLoader<true, TBM, NW> la;
        DmaLoader<true, TBN, NW> lb;
        la.init(A, M, K, K, m0, wave, lane);
        lb.init(B, N, K, K, n0, wave, lane);
        constexpr int TM = TBM / 32, TN = TBN / 32;
        const int wm = (wave & 3) >> 1, wn = wave & 1;
        f32x4 cc[TM][TN];
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { af[i] = bf16x8{}; 
#pragma unroll
            for (int j = 0; j < TN; ++j) cc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = bf16x8{};
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int a = 0; a < AHEAD; ++a)
                if (a < nk) { la.issue(smem + a * STAGE, a, wave); lb.issue(smem + a * STAGE + TBM * BK * 2, a, wave); }
            int st = 0;
            for (int t = 0; t < nk; ++t) {
                const int younger = min(AHEAD - 1, nk - 1 - t);
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
                else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT) : "memory");
                if (BARRIER) __builtin_amdgcn_s_barrier();
                if (t + AHEAD < nk) {
                    int s2 = st + AHEAD; if (s2 >= NSTAGE) s2 -= NSTAGE;
                    la.issue(smem + s2 * STAGE, t + AHEAD, wave);
                    lb.issue(smem + s2 * STAGE + TBM * BK * 2, t + AHEAD, wave);
                }
                if constexpr (WORK == 0) {
                    acc += *reinterpret_cast<const unsigned*>(smem + st * STAGE + tid * 4);      // one dword per thread: the tile is "used"
                } else {
                    const char* sA = smem + st * STAGE;
                    const char* sB = sA + TBM * BK * 2;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if constexpr (WORK & 1) {
#pragma unroll
                            for (int i = 0; i < TM; ++i) af[i] = read_frag<true, TBM>(sA, wm * TM + i, ks, lane);
#pragma unroll
                            for (int j = 0; j < TN; ++j) bf[j] = read_frag<true, TBN>(sB, wn * TN + j, ks, lane);
                        }
                        if constexpr (WORK & 2) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], cc[i][j], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[j]));
                        }
                    }
                }
                if (++st == NSTAGE) st = 0;
            }
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
        if constexpr ((WORK & 2) != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc += __builtin_bit_cast(unsigned, cc[i][j][0] + cc[i][j][3]);
        }
    } else if constexpr (MODE == 1) {
        static_assert(NW == 4, "register staging: 256 threads");
        TileLoader<true, TBM> la;
        TileLoader<true, TBN> lb;
        la.init(A, M, K, K, m0, tid);
        lb.init(B, N, K, K, n0, tid);
        u32x4 ra[TBM / 32], rb[TBN / 32];
        for (int r = 0; r < reps; ++r) {
            la.load(ra, 0); lb.load(rb, 0);
            la.store(smem, ra); lb.store(smem + TBM * BK * 2, rb);
            __syncthreads();
            for (int t = 0; t < nk; ++t) {
                const int s = t & 1;
                const bool more = t + 1 < nk;
                if (more) { la.load(ra, t + 1); lb.load(rb, t + 1); }
                acc += *reinterpret_cast<const unsigned*>(smem + s * STAGE + tid * 4);
                if (more) { char* d = smem + (s ^ 1) * STAGE; la.store(d, ra); lb.store(d + TBM * BK * 2, rb); }
                __syncthreads();
            }
        }
    } else {
        static_assert(NW == 4, "256 threads");
        TileLoader<true, TBM> la;
        TileLoader<true, TBN> lb;
        la.init(A, M, K, K, m0, tid);
        lb.init(B, N, K, K, n0, tid);
        u32x4 ra[NSTAGE][TBM / 32], rb[NSTAGE][TBN / 32];
        for (int r = 0; r < reps; ++r) {
            for (int t = 0; t < nk; t += NSTAGE) {
#pragma unroll
                for (int s = 0; s < NSTAGE; ++s) { la.load(ra[s], min(t + s, nk - 1)); lb.load(rb[s], min(t + s, nk - 1)); }
#pragma unroll
                for (int s = 0; s < NSTAGE; ++s) {
#pragma unroll
                    for (int i = 0; i < TBM / 32; ++i) acc += ra[s][i][0] ^ ra[s][i][3];
#pragma unroll
                    for (int i = 0; i < TBN / 32; ++i) acc += rb[s][i][1] ^ rb[s][i][2];
                }
            }
        }
    }
    if (sink && acc == 0x12345678u) *sink = acc;
}

template <int MODE, int NW, int TBM, int TBN, int NSTAGE, bool BARRIER, int WORK = 0>
void run(const char* name, const bf16_t* A, const bf16_t* B, int M, int N, int K, int lds_extra = 0) {
    const int lds = (MODE == 2 ? 0 : (MODE == 1 ? 2 : NSTAGE) * (TBM + TBN) * BK * 2) + lds_extra;
    auto k = fill_kernel<MODE, NW, TBM, TBN, NSTAGE, BARRIER, WORK>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const dim3 grid(N / TBN, M / TBM);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 1;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, 0, A, B, M, N, K, (unsigned*)nullptr, reps);
    float best = 1e30f, tot = 0.f;
    const int iters = 20;
    for (int i = 0; i < iters; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, 0, A, B, M, N, K, (unsigned*)nullptr, reps);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    const hipError_t err = hipGetLastError();
    const double bytes = (double)grid.x * grid.y * (TBM + TBN) * K * 2.0 * reps;
    const int tiles = grid.x * grid.y;
    printf("%-44s %4d tiles lds %6d  best %7.1f us  avg %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU (256)  flops-equivalent %6.0f TF/s %s\n", name, tiles, lds,
           best * 1e3, tot / iters * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e9 / 256.0,
           2.0 * M * N * K / (best * 1e-3) / 1e12, err == hipSuccess ? "" : hipGetErrorString(err));
}

}  // namespace

int main() {
    const int MAXE = 4608 * 4096;
    bf16_t *A, *B;
    (void)hipMalloc(&A, (size_t)MAXE * 2 * 2);
    (void)hipMalloc(&B, (size_t)16384 * 4096 * 2);
    (void)hipMemset(A, 1, (size_t)MAXE * 2 * 2);
    (void)hipMemset(B, 2, (size_t)16384 * 4096 * 2);
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{3072, 4096, 1024, "fc1 M3072"}, {1536, 4096, 1024, "fc1 M1536"}, {1536, 1024, 4096, "fc2 M1536"}, {3072, 1024, 4096, "fc2 M3072"},
                            {1536, 1024, 1024, "proj M1536"}};
    for (const Shape& s : shapes) {
        printf("== %s: M %d N %d K %d (operands once: %.1f MB)\n", s.what, s.M, s.N, s.K, (s.M + s.N) * (double)s.K * 2 / 1e6);
        const int M = s.M, N = s.N, K = s.K;
        run<0, 4, 96, 128, 2, true, 0>("96x128 ring2: fill only", A, B, M, N, K);
        run<0, 4, 96, 128, 2, true, 1>("96x128 ring2: fill + fragment reads", A, B, M, N, K);
        run<0, 4, 96, 128, 2, true, 2>("96x128 ring2: fill + MFMAs (no reads)", A, B, M, N, K);
        run<0, 4, 96, 128, 2, true, 3>("96x128 ring2: fill + reads + MFMAs", A, B, M, N, K);
        run<0, 4, 96, 64, 3, true, 0>("96x64 ring3: fill only", A, B, M, N, K);
        run<0, 4, 96, 64, 3, true, 1>("96x64 ring3: fill + fragment reads", A, B, M, N, K);
        run<0, 4, 96, 64, 3, true, 2>("96x64 ring3: fill + MFMAs (no reads)", A, B, M, N, K);
        run<0, 4, 96, 64, 3, true, 3>("96x64 ring3: fill + reads + MFMAs", A, B, M, N, K);
        run<0, 4, 128, 128, 2, true, 0>("128x128 ring2: fill only", A, B, M, N, K);
        run<0, 4, 128, 128, 2, true, 1>("128x128 ring2: fill + fragment reads", A, B, M, N, K);
        run<0, 4, 128, 128, 2, true, 2>("128x128 ring2: fill + MFMAs (no reads)", A, B, M, N, K);
        run<0, 4, 128, 128, 2, true, 3>("128x128 ring2: fill + reads + MFMAs", A, B, M, N, K);
        run<0, 4, 192, 128, 3, true, 0>("192x128 ring3: fill only", A, B, M, N, K);
        run<0, 4, 192, 128, 3, true, 1>("192x128 ring3: fill + fragment reads", A, B, M, N, K);
        run<0, 4, 192, 128, 3, true, 2>("192x128 ring3: fill + MFMAs (no reads)", A, B, M, N, K);
        run<0, 4, 192, 128, 3, true, 3>("192x128 ring3: fill + reads + MFMAs", A, B, M, N, K);
    }
    return 0;
}
