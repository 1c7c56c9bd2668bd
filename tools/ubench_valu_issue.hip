// VALU issue rate of ONE wave against several waves per SIMD (gfx950): 8 independent chains of v_fma_f32 / v_pk_fma_f32 / v_cvt_pk_bf16_f32
// per wave, one workgroup per CU with 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.  Reports clocks per instruction and SIMD
// (at the 2.4 GHz peak clock) -- if one wave cannot issue a VALU instruction every 4 clocks, kernels at 1-2 waves per SIMD are issue-bound.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu_issue.hip -o scratch/ubench_valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NT>
__global__ void __launch_bounds__(NT) probe(float* out, int iters) {
    float x[8];
    f32x2 p[8];
    for (int j = 0; j < 8; ++j) {
        x[j] = 1.0f + j + threadIdx.x;
        p[j] = f32x2{1.0f + j, 2.0f + threadIdx.x};
    }
    const float m = 1.0001f, c = 0.5f;
    const f32x2 pm = f32x2{1.0001f, 0.9999f}, pc = f32x2{0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
                if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[j]) : "v"(m));
                if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
            }
    }
    float s = 0.0f;
    for (int j = 0; j < 8; ++j) s += x[j] + p[j][0] + p[j][1];
    out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int KIND, int NT>
static void run(float* d, int iters) {
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_perm_b32"};
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    hipLaunchKernelGGL((probe<KIND, NT>), dim3(256), dim3(NT), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(t0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<KIND, NT>), dim3(256), dim3(NT), 0, 0, d, iters);
    CK(hipEventRecord(t1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, t0, t1));
    const double us = ms * 1e3 / 5, instr_per_wave = (double)iters * 32, waves_per_simd = NT / 256.0;
    printf("%-18s %d wave(s) per SIMD: %8.1f us, %5.2f clocks per instruction of a wave, %5.2f clocks per instruction of the SIMD\n", names[KIND],
           NT / 256, us, us * 2400.0 / instr_per_wave, us * 2400.0 / (instr_per_wave * waves_per_simd));
}

int main() {
    float* d;
    CK(hipMalloc(&d, 256 * 1024 * 4));
    const int iters = 4096;
    run<0, 256>(d, iters); run<0, 512>(d, iters); run<0, 1024>(d, iters);
    run<1, 256>(d, iters); run<1, 512>(d, iters); run<1, 1024>(d, iters);
    run<2, 256>(d, iters); run<2, 512>(d, iters); run<2, 1024>(d, iters);
    run<3, 256>(d, iters); run<3, 512>(d, iters); run<3, 1024>(d, iters);
    return 0;
}
