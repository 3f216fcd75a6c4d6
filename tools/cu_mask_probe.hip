// Diagnostic (not product): which CUs does a stream created with hipExtStreamCreateWithCUMask really run on?
// A census kernel records (XCC id, SE id, CU id) of every workgroup; the host prints, per mask, how many distinct CUs of each XCD
// were used and whether two complementary masks are disjoint.  The engine's spatial split of the backward (MEBT_CU_SPLIT, engine.cpp)
// relies on the bit order found here: bit b of the mask <-> XCD b % 8, CU slot b / 8 of that XCD.
// Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe && tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void census(uint32_t* out, int spin) {
    // keep the workgroup resident for a while so that the grid spreads over every CU the queue may use
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}

static int run(const char* name, hipStream_t st, uint32_t* d, std::set<uint32_t>& cus) {
    const int G = 4096;
    std::vector<uint32_t> h(2 * G);
    hipLaunchKernelGGL(census, dim3(G), dim3(256), 0, st, d, 20000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    int per_xcc[8] = {0};
    std::set<uint32_t> seen[8];
    for (int i = 0; i < G; ++i) {
        const uint32_t xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
        // HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (xcc < 8) seen[xcc].insert(id);
        cus.insert(id);
    }
    int total = 0;
    printf("%-28s CUs per XCD:", name);
    for (int x = 0; x < 8; ++x) { per_xcc[x] = (int)seen[x].size(); total += per_xcc[x]; printf(" %2d", per_xcc[x]); }
    printf("   total %d\n", total);
    return 0;
}

int main() {
    uint32_t* d;
    CK(hipMalloc(&d, 2 * 4096 * 4));
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    std::set<uint32_t> all;
    if (run("unmasked", plain, d, all)) return 1;
    for (int n : {8, 16, 20, 24}) {
        // assumed order: bit b <-> XCD b % 8, slot b / 8: the first 8 n bits = n CUs of every XCD
        uint32_t lo[8] = {0}, hi[8] = {0};
        for (int b = 0; b < 256; ++b) (b < 8 * n ? lo : hi)[b >> 5] |= 1u << (b & 31);
        hipStream_t a, b2;
        CK(hipExtStreamCreateWithCUMask(&a, 8, lo));
        CK(hipExtStreamCreateWithCUMask(&b2, 8, hi));
        char nm[64];
        std::set<uint32_t> sa, sb;
        snprintf(nm, sizeof nm, "bits [0, %d)", 8 * n);
        if (run(nm, a, d, sa)) return 1;
        snprintf(nm, sizeof nm, "bits [%d, 256)", 8 * n);
        if (run(nm, b2, d, sb)) return 1;
        int common = 0;
        for (uint32_t c : sa) common += sb.count(c);
        printf("    overlap of the two: %d CUs\n", common);
        CK(hipStreamDestroy(a));
        CK(hipStreamDestroy(b2));
    }
    {   // whole XCDs: bits with b % 8 < k
        for (int k : {4, 6}) {
            uint32_t lo[8] = {0};
            for (int b = 0; b < 256; ++b) if (b % 8 < k) lo[b >> 5] |= 1u << (b & 31);
            hipStream_t a;
            CK(hipExtStreamCreateWithCUMask(&a, 8, lo));
            char nm[64];
            std::set<uint32_t> sa;
            snprintf(nm, sizeof nm, "bits b %% 8 < %d", k);
            if (run(nm, a, d, sa)) return 1;
            CK(hipStreamDestroy(a));
        }
    }
    return 0;
}
