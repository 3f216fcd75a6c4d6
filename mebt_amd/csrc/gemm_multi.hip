// Mixed-layout multi-product launch of the bf16 GEMM family (kernels.h: GemmMulti): one grid enumerates the tiles of up to six
// independent products; a workgroup picks its product, then runs that product's layout of the LDS-DMA tile loop (gemm_kernels.h)
// with the product's own epilogue.  Used by the backward of a block to launch each dgrad product (critical path) together with
// the weight-gradient product that reads the same dY (long K, on nobody's dependency chain; AdamW in its epilogue).
// Reference being replaced: the autograd backward of nn.Linear (mebt/modules/gpt.py:126-128,140,150-155) + AdamW (transformer.py:790-797).
#include "gemm_kernels.h"

namespace {

template <int TBM, int TBN, int NSTAGE>
__global__ __launch_bounds__(256) void gemm_multi_kernel(const GemmMulti g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PrefetchRegs pfr = prefetch_next(g.p[0], blockIdx.x, gridDim.x, 256);
    int i = 0;
#pragma unroll
    for (int k = 1; k < MEBT_MAX_MULTI; ++k)
        if (k < g.n && (int)blockIdx.x >= g.tile_start[k]) i = k;
    const GemmParams& p = g.p[i];
    const int t = blockIdx.x - g.tile_start[i];
    int tr, tc;
    xcd_tile(t, g.ntx[i], (p.M + TBM - 1) / TBM, p.M, p.N, tr, tc);
    const int m0 = tr * TBM, n0 = tc * TBN, nkt = (p.K + BK - 1) / BK;
    if (p.a_kc && p.b_kc) gemm_tile_dma<true, true, TBM, TBN, NSTAGE>(p, m0, n0, 0, nkt, smem, false, true);
    else if (p.a_kc) gemm_tile_dma<true, false, TBM, TBN, NSTAGE>(p, m0, n0, 0, nkt, smem, false, true);
    else gemm_tile_dma<false, false, TBM, TBN, NSTAGE, 1, true>(p, m0, n0, 0, nkt, smem, false, true);
    prefetch_sink(g.p[0], pfr);
}

template <int TBM, int TBN>
void launch_t(const GemmMulti& g, int tiles, int ring, hipStream_t stream) {
    if (ring >= 4 && 4 * (TBM + TBN) * BK * 2 <= 128 * 1024)
        hipLaunchKernelGGL((gemm_multi_kernel<TBM, TBN, (4 * (TBM + TBN) * BK * 2 <= 128 * 1024 ? 4 : 2)>), dim3(tiles), dim3(256), 4 * (TBM + TBN) * BK * 2, stream, g);
    else if (ring >= 3) hipLaunchKernelGGL((gemm_multi_kernel<TBM, TBN, 3>), dim3(tiles), dim3(256), 3 * (TBM + TBN) * BK * 2, stream, g);
    else hipLaunchKernelGGL((gemm_multi_kernel<TBM, TBN, 2>), dim3(tiles), dim3(256), 2 * (TBM + TBN) * BK * 2, stream, g);
}
template <int TBM, int TBN>
int attrs_t() {
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_kernel<TBM, TBN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TBM + TBN) * BK * 2));
    MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_kernel<TBM, TBN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (TBM + TBN) * BK * 2));
    if (4 * (TBM + TBN) * BK * 2 <= 128 * 1024)
        MEBT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_kernel<TBM, TBN, (4 * (TBM + TBN) * BK * 2 <= 128 * 1024 ? 4 : 2)>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (TBM + TBN) * BK * 2));
    return MEBT_OK;
}

}  // namespace

// fills tile_start / ntx for the tile shape and launches; the tile shapes the RC x RC loop exists for: 128 / 96 / 64 rows x 128 / 64 columns
void mebt_gemm_multi_cfg(GemmMulti& g, int tbm, int tbn, int ring, hipStream_t stream) {
    int tiles = 0;
    for (int i = 0; i < g.n; ++i) {
        g.ntx[i] = (g.p[i].N + tbn - 1) / tbn;
        g.tile_start[i] = tiles;
        tiles += ((g.p[i].M + tbm - 1) / tbm) * g.ntx[i];
    }
    for (int i = g.n; i <= MEBT_MAX_MULTI; ++i) g.tile_start[i] = tiles;
    if (tbm == 128 && tbn == 128) launch_t<128, 128>(g, tiles, ring, stream);
    else if (tbm == 128 && tbn == 64) launch_t<128, 64>(g, tiles, ring, stream);
    else if (tbm == 96 && tbn == 128) launch_t<96, 128>(g, tiles, ring, stream);
    else if (tbm == 96 && tbn == 64) launch_t<96, 64>(g, tiles, ring, stream);
    else if (tbm == 64 && tbn == 128) launch_t<64, 128>(g, tiles, ring, stream);
    else launch_t<64, 64>(g, tiles, ring, stream);
}
int mebt_gemm_multi_attrs() {
    if (int rc = attrs_t<128, 128>()) return rc;
    if (int rc = attrs_t<128, 64>()) return rc;
    if (int rc = attrs_t<96, 128>()) return rc;
    if (int rc = attrs_t<96, 64>()) return rc;
    if (int rc = attrs_t<64, 128>()) return rc;
    return attrs_t<64, 64>();
}
