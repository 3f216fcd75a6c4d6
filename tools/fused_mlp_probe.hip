// Probe (diagnostics, not product): does a tile-level hand-off INSIDE one launch beat a kernel boundary on this chip?
// The MLP forward of one Sky-16f block at batch 6 — u = gelu(h W1^T + b1) [1536 x 4096], out = x + u W2^T + b2 [1536 x 1024] — as
//   (a) two launches (the step's own kernels: gemm_bf16_dma<96,128,2>, gemm_bf16_dma_ks2<96,64,3>), and
//   (b) ONE launch of 256 workgroups whose row blocks are owned by an XCD: XCD x computes the fc1 tiles of row blocks x and x + 8
//       (64 tiles on its 32 CUs), counts them per row block, and each of its workgroups then takes one fc2 tile of those row blocks
//       as soon as that block's 32 fc1 tiles are counted.  Producer and consumer share the XCD's L2, so the hand-off is plain
//       stores + s_waitcnt vmcnt(0) + one atomic per tile, and plain loads (the consumer CU never read those lines before).
// Weights rotate through LAYERS sets (16 MB each) so that they come from HBM / Infinity Cache as in the step.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mebt_amd/csrc -I include tools/fused_mlp_probe.hip -o tools/bin/fused_mlp_probe
#include "gemm_kernels.h"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
#include <functional>

void mebt_set_hip_error(hipError_t, const char*) {}
void mebt_set_error(const char*) {}

namespace {
constexpr int RB = 96;           // rows per row block
template <int TBN1, int NST1, int KS1, int TBN2, int NST2, int KS2>
__global__ __launch_bounds__(512) void mlp_fused_kernel(const GemmParams p1, const GemmParams p2, int* counters, int target, int phases) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;          // workgroup i runs on XCD i % 8
    const int nslot = gridDim.x >> 3;
    const int nrb = p1.M / RB, rb_per_xcd = nrb / 8;
    const int ncol1 = p1.N / TBN1, ncol2 = p2.N / TBN2;
    for (int t = slot; (phases & 1) && t < rb_per_xcd * ncol1; t += nslot) {
        const int rb = xcd + 8 * (t / ncol1), c = t % ncol1;
        gemm_tile_dma<true, true, RB, TBN1, NST1, KS1>(p1, rb * RB, c * TBN1, 0, p1.K / BK, smem, false, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores have left for L2
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(counters + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = slot; (phases & 2) && t < rb_per_xcd * ncol2; t += nslot) {
        const int rb = xcd + 8 * (t / ncol2), c = t % ncol2;
        if (threadIdx.x == 0 && phases == 3) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz; a probe must not hang the box: give up after 20 ms
            while (__hip_atomic_load(counters + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target * ncol1) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { __hip_atomic_fetch_add(counters + 63, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        gemm_tile_dma<true, true, RB, TBN2, NST2, KS2>(p2, rb * RB, c * TBN2, 0, p2.K / BK, smem, false, true);
        __syncthreads();
    }
}

float time_us(int iters, const std::function<void(int)>& fn) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) fn(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) fn(i);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}
}  // namespace

int main() {
    const int M = 1536, D = 1024, H = 4096, LAYERS = 18;
    bf16_t *h, *x, *pre, *u, *out_a, *out_b, *W1, *W2;
    float *b1, *b2;
    int* counters;
    (void)hipMalloc(&h, (size_t)M * D * 2); (void)hipMalloc(&x, (size_t)M * D * 2);
    (void)hipMalloc(&pre, (size_t)M * H * 2); (void)hipMalloc(&u, (size_t)M * H * 2);
    (void)hipMalloc(&out_a, (size_t)M * D * 2); (void)hipMalloc(&out_b, (size_t)M * D * 2);
    (void)hipMalloc(&W1, (size_t)LAYERS * H * D * 2); (void)hipMalloc(&W2, (size_t)LAYERS * D * H * 2);
    (void)hipMalloc(&b1, H * 4); (void)hipMalloc(&b2, D * 4);
    (void)hipMalloc(&counters, 64 * 4);
    (void)hipMemset(counters, 0, 64 * 4);
    {   // deterministic small values
        std::vector<uint16_t> v((size_t)LAYERS * H * D);
        auto fill = [&](void* dst, size_t n, float scale, unsigned seed) {
            unsigned s = seed;
            for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const float f = (((s >> 9) & 0xFFFF) / 65536.f - 0.5f) * scale; unsigned bits; std::memcpy(&bits, &f, 4); v[i] = (uint16_t)(bits >> 16); }
            (void)hipMemcpy(dst, v.data(), n * 2, hipMemcpyHostToDevice);
        };
        fill(h, (size_t)M * D, 2.f, 1); fill(x, (size_t)M * D, 2.f, 2);
        fill(W1, (size_t)LAYERS * H * D, 0.06f, 3); fill(W2, (size_t)LAYERS * D * H, 0.03f, 4);
        (void)hipMemset(b1, 0, H * 4); (void)hipMemset(b2, 0, D * 4);
    }
    auto params = [&](int layer, bf16_t* out, GemmParams& p1, GemmParams& p2) {
        p1 = GemmParams{}; p2 = GemmParams{};
        p1.A = h; p1.B = W1 + (size_t)layer * H * D; p1.C = pre; p1.C2 = u; p1.bias = b1; p1.M = M; p1.N = H; p1.K = D;
        p1.lda = D; p1.ldb = D; p1.ldc = H; p1.a_kc = 1; p1.b_kc = 1; p1.epilogue = EPI_GELU; p1.split_k = 1;
        p2.A = u; p2.B = W2 + (size_t)layer * D * H; p2.C = out; p2.bias = b2; p2.aux = x; p2.ld_aux = D; p2.M = M; p2.N = D; p2.K = H;
        p2.lda = H; p2.ldb = H; p2.ldc = D; p2.a_kc = 1; p2.b_kc = 1; p2.epilogue = EPI_RESID; p2.split_k = 1;
    };
    // (a) the step's two launches
    auto k1 = gemm_bf16_dma_kernel<true, true, 96, 128, 2>;
    auto k2 = gemm_bf16_dma_ks2_kernel<true, true, 96, 64, 3>;
    const int lds1 = 2 * (96 + 128) * BK * 2, lds2 = 2 * 3 * (96 + 64) * BK * 2;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    auto separate = [&](int i) {
        GemmParams p1, p2;
        params(i % LAYERS, out_a, p1, p2);
        hipLaunchKernelGGL(k1, dim3(H / 128, M / 96), dim3(256), lds1, 0, p1);
        hipLaunchKernelGGL(k2, dim3(D / 64, M / 96), dim3(512), lds2, 0, p2);
    };
    // the same two products as two launches of the 512-thread forms the fused kernel is built from (like for like)
    auto k1b = gemm_bf16_dma_ks2_kernel<true, true, 96, 128, 2>;
    const int lds1b = 2 * 2 * (96 + 128) * BK * 2;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1b), hipFuncAttributeMaxDynamicSharedMemorySize, lds1b);
    auto separate_ks2 = [&](int i) {
        GemmParams p1, p2;
        params(i % LAYERS, out_a, p1, p2);
        hipLaunchKernelGGL(k1b, dim3(H / 128, M / 96), dim3(512), lds1b, 0, p1);
        hipLaunchKernelGGL(k2, dim3(D / 64, M / 96), dim3(512), lds2, 0, p2);
    };
    // (b) one launch
    auto kf = mlp_fused_kernel<128, 2, 2, 64, 3, 2>;
    const int ldsf = lds1b > lds2 ? lds1b : lds2;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, ldsf);
    int epoch = 0;
    auto fused = [&](int i) {
        GemmParams p1, p2;
        params(i % LAYERS, out_b, p1, p2);
        ++epoch;
        hipLaunchKernelGGL(kf, dim3(256), dim3(512), ldsf, 0, p1, p2, counters, epoch, 3);
    };
    // the same row-block-per-XCD tile assignment as TWO launches (no hand-off): what the ownership itself costs
    auto owned_two = [&](int i) {
        GemmParams p1, p2;
        params(i % LAYERS, out_b, p1, p2);
        hipLaunchKernelGGL(kf, dim3(256), dim3(512), ldsf, 0, p1, p2, counters, 0, 1);
        hipLaunchKernelGGL(kf, dim3(256), dim3(512), ldsf, 0, p1, p2, counters, 0, 2);
    };
    // correctness: layer 5 both ways (against the 8-wave forms: the same arithmetic, so bit for bit)
    separate_ks2(5); (void)hipDeviceSynchronize();
    fused(5); (void)hipDeviceSynchronize();
    const hipError_t err = hipGetLastError();
    std::vector<uint16_t> a((size_t)M * D), b((size_t)M * D);
    (void)hipMemcpy(a.data(), out_a, a.size() * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), out_b, b.size() * 2, hipMemcpyDeviceToHost);
    size_t diff = 0; double amax = 0;
    for (size_t i = 0; i < a.size(); ++i) { diff += a[i] != b[i]; unsigned bits = (unsigned)a[i] << 16; float f; std::memcpy(&f, &bits, 4); amax = std::fmax(amax, std::fabs(f)); }
    int gave_up = 0; (void)hipMemcpy(&gave_up, counters + 63, 4, hipMemcpyDeviceToHost);
    printf("ONE launch vs two launches of the same tile code: %zu of %zu elements differ (max |out| %.3f); waits given up: %d %s\n", diff, a.size(), amax, gave_up, err == hipSuccess ? "" : hipGetErrorString(err));
    for (int round = 0; round < 3; ++round) {
        const float ta = time_us(90, separate), tb = time_us(90, separate_ks2), tc = time_us(90, fused), td = time_us(90, owned_two);
        printf("round %d: two launches (the step's kernels) %6.1f us   two launches (8-wave forms) %6.1f us   ONE launch, same-XCD hand-off %6.1f us   its tile assignment as two launches %6.1f us\n", round, ta, tb, tc, td);
    }
    return 0;
}
