// Are 4-byte loads at 2-byte-aligned addresses (bf16 rows of 58 / 116 channels start at 116 / 232-byte strides, channel offsets
// of 29 elements) correct and fast on gfx950?  buffer_load_dword / global_load_dword at byte offset 4*i + 2.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_unaligned.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0: aligned dword, 1: misaligned dword (buffer), 2: two ushort loads, 3: misaligned dword (global pointer)
__global__ void k(const unsigned short* __restrict__ src, unsigned* __restrict__ dst, int n, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, n * 2, 0x00020000);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned e = ((unsigned)tid * 2u + (unsigned)it * 1024u * 1024u) % (unsigned)(n - 4);        // even element index
        const unsigned off = (e + (MODE == 0 ? 0u : 1u)) * 2u;
        if (MODE == 0 || MODE == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
        else if (MODE == 2) acc += (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0) | ((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off + 2, 0, 0) << 16);
        else acc += *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(src) + off);
    }
    dst[tid] = acc;
}

int main() {
    const int n = 64 << 20;
    unsigned short* src; unsigned* dst;
    CK(hipMalloc(&src, (size_t)n * 2)); CK(hipMalloc(&dst, (size_t)(1 << 22) * 4));
    std::vector<unsigned short> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 2654435761u >> 13);
    CK(hipMemcpy(src, h.data(), (size_t)n * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 8192, iters = 16;
    std::vector<unsigned> out[4];
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, dst, n, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, dst, n, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, dst, n, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, src, dst, n, iters);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("mode %d: %.1f us (%.0f GB/s of 4-byte requests)\n", mode, ms * 1e3, (double)blocks * 256 * iters * 4 / ms / 1e6);
        }
        out[mode].resize(blocks * 256);
        CK(hipMemcpy(out[mode].data(), dst, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost));
    }
    size_t bad12 = 0, bad13 = 0;
    for (size_t i = 0; i < out[1].size(); ++i) { bad12 += out[1][i] != out[2][i]; bad13 += out[1][i] != out[3][i]; }
    printf("misaligned dword (buffer) vs two ushorts: %zu differ; vs misaligned global dword: %zu differ\n", bad12, bad13);
    return 0;
}
