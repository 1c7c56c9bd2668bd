// Which SIMD does wave w of a 512-thread workgroup land on (gfx950)?  HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8],
// sh [12], se [15:13] (gfx9 layout).  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_wave_simd.hip -o scratch/ubench_wave_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__global__ void __launch_bounds__(512, 1) probe(uint32_t* out) {
    extern __shared__ char lds[];
    const uint32_t id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    lds[threadIdx.x] = 0;
}

int main() {
    uint32_t* d;
    const int nb = 512;
    CK(hipMalloc(&d, nb * 8 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 90 * 1024, 0, d);
    CK(hipDeviceSynchronize());
    uint32_t h[nb * 8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    int hist[8][4] = {};
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; ++b) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 8; ++w) printf(" [w%d simd %u wave_id %u cu %u se %u]", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15, (h[b * 8 + w] >> 13) & 7);
        printf("\n");
    }
    return 0;
}
