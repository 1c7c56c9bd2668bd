// Column-mapped reduction skeleton shared by the BatchNorm / depthwise / stem kernels.
#pragma once
#include "cdrl_kernels.h"

namespace cdrl {

// ------------------------------------------------------------------------------------------
// column-mapped reduction skeleton
// ------------------------------------------------------------------------------------------
template <int NQ, class F>
__global__ void __launch_bounds__(256) colreduce_kernel(F f, int Mg, int C, int rb, double* __restrict__ part) {
    extern __shared__ double sm[];   // [CY][NQ][CX]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y;
    const int nb = gridDim.x;
    const int r0 = blockIdx.x * rb;
    const int r1 = min(r0 + rb, Mg);
    for (int c0 = 0; c0 < C; c0 += CX) {
        const int c = c0 + tx;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        if (c < C) {
            for (int r = r0 + ty; r < r1; r += CY) f(g, (int64_t)g * Mg + r, c, acc);
        }
        if (CY > 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) sm[(ty * NQ + q) * CX + tx] = acc[q];
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    double s = acc[q];
                    for (int y = 1; y < CY; ++y) s += sm[(y * NQ + q) * CX + tx];
                    acc[q] = s;
                }
            }
            __syncthreads();
        }
        if (ty == 0 && c < C) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) part[(((int64_t)g * nb + blockIdx.x) * NQ + q) * C + c] = acc[q];
        }
    }
}

template <int NQ, class F>
static int launch_colreduce(F f, int G, int Mg, int C, double* part, hipStream_t st, int max_blocks = NB_STATS) {
    ColGeom g = col_geom(Mg, C, max_blocks);
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    size_t sm = (size_t)g.cy * NQ * g.cx * sizeof(double);
    hipLaunchKernelGGL((colreduce_kernel<NQ, F>), grid, block, sm, st, f, Mg, C, g.rb, part);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
