// Microbenchmark (diagnostics, not product): the staggered two-group 256 x 256 kernel (gemm_bf16_pp_kernel) with parts of its
// loop removed — which of refill DMA / fragment reads / MFMAs sets the slot time?  Constant operands, warm caches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mebt_amd/csrc -I include tools/pp_bench.hip -o tools/bin/pp_bench && tools/bin/pp_bench
#include "gemm_kernels.h"
#include <cstdio>

void mebt_set_hip_error(hipError_t, const char*) {}
void mebt_set_error(const char*) {}

namespace {
template <int DBG>
void run(const char* name, const GemmParams& p) {
    auto k = gemm_bf16_pp_kernel<true, DBG>;
    const int lds = 8 * 128 * BK * 2 + 8 * 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int all_tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256);
    const dim3 grid(all_tiles < 256 ? all_tiles : 256);      // persistent over the tile list
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, p);
    float best = 1e30f, tot = 0.f;
    const int iters = 20;
    for (int i = 0; i < iters; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, p);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; tot += ms;
    }
    const hipError_t err = hipGetLastError();
    const int tiles = all_tiles, rounds = (tiles + 255) / 256, nk = p.K / BK;
    printf("  %-52s best %7.1f us  avg %7.1f us  %6.0f TF/s-equivalent   %6.0f ns per k-tile per round %s\n", name, best * 1e3, tot / iters * 1e3,
           2.0 * p.M * p.N * p.K / (best * 1e-3) / 1e12, best * 1e6 / (rounds * nk), err == hipSuccess ? "" : hipGetErrorString(err));
}
}  // namespace

int main() {
    bf16_t *A, *B, *C;
    (void)hipMalloc(&A, (size_t)32768 * 4096 * 2);
    (void)hipMalloc(&B, (size_t)16384 * 4096 * 2);
    (void)hipMalloc(&C, (size_t)32768 * 16384 * 2);
    (void)hipMemset(A, 0, (size_t)32768 * 4096 * 2);
    (void)hipMemset(B, 0, (size_t)16384 * 4096 * 2);
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{4096, 4096, 1024, "256 tiles (one round), K 1024"}, {4096, 4096, 4096, "256 tiles, K 4096"}, {3072, 4096, 1024, "fc1 M3072 (192 tiles)"},
                            {32768, 2048, 1024, "c4 kv-like (1024 tiles)"}};
    for (const Shape& s : shapes) {
        GemmParams p{};
        p.A = A; p.B = B; p.C = C; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.K; p.ldc = s.N; p.a_kc = 1; p.b_kc = 1;
        printf("== %s: M %d N %d K %d\n", s.what, s.M, s.N, s.K);
        run<0>("full", p);
        run<1>("no refill DMA (reads + MFMAs)", p);
        run<2>("no MFMAs (refill + reads)", p);
        run<4>("no fragment reads (refill + MFMAs)", p);
        run<3>("fragment reads only", p);
        run<5>("MFMAs only", p);
        run<6>("refill only", p);
        run<7>("barriers only", p);
        run<7 + 8>("barriers only, no epilogue", p);
        run<7 + 16>("barriers only, no prologue loads", p);
        run<7 + 24>("barriers only, neither", p);
        run<8>("full loop, no epilogue", p);
        run<5 + 8>("MFMAs only, no epilogue", p);
        run<6 + 8>("refill only, no epilogue", p);
        run<1 + 8>("reads + MFMAs, no epilogue", p);
        run<4 + 8>("refill + MFMAs, no epilogue", p);
        run<2 + 8>("refill + reads, no epilogue", p);
    }
    return 0;
}
