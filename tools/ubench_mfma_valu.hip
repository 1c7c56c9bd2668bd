// Do a dependent MFMA chain of one wave and the VALU work of ANOTHER wave on the same SIMD overlap (gfx950)?  One 512-thread workgroup
// per CU: waves w and w + 4 share a SIMD (tools/ubench_wave_simd.hip).  Waves 0-3 run v_mfma_f32_32x32x16_bf16 on NACC independent
// accumulators (NACC = 1: every MFMA depends on the previous one), waves 4-7 run a dependent v_pk_fma_f32 chain.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma_valu.hip -o scratch/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND of the VALU chain: 0 four v_pk_fma_f32 chains, 1 eight v_fma_f32 chains (the same number of FMAs), 2 eight v_pk_fma_f32 chains (twice),
// 3 eight v_cvt_pk_bf16_f32 + widen chains (the three-way split's instructions)
template <int NACC, int KIND>
__global__ void __launch_bounds__(512, 1) probe(float* out, int mfma_iters, int valu_iters, int who) {
    const int hwave = threadIdx.x >> 6;
    const int role = __builtin_amdgcn_readfirstlane(hwave >> 2);
    if (role == 0) {
        if (!(who & 1)) return;
        f32x16 acc[NACC];
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (__bf16)(float)(threadIdx.x & 3);
            b[i] = (__bf16)1.0f;
        }
        for (int i = 0; i < mfma_iters; i += NACC) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0.0f;
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(who & 2)) return;
        f32x2 v[8];
        for (int j = 0; j < 8; ++j) v[j] = f32x2{1.0f + j + threadIdx.x, 2.0f + j};
        const f32x2 m = f32x2{1.0001f, 0.9999f}, c = f32x2{0.5f, 0.25f};
        for (int i = 0; i < valu_iters; ++i) {
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
            } else if (KIND == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j][0], y = v[j][1];
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m[0]), "v"(c[0]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(m[1]), "v"(c[1]));
                    v[j][0] = x;
                    v[j][1] = y;
                }
            } else if (KIND == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], m, c);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[j], bf16x2));
                    v[j] = f32x2{__uint_as_float(w << 16) + 1.0f, __uint_as_float(w & 0xffff0000u)};
                }
            }
        }
        float s = 0.0f;
        for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int NACC, int KIND>
static void run(float* d, int mi, int vi) {
    const char* names[4] = {"", "MFMA waves only", "VALU waves only", "both"};
    for (int who = 1; who <= 3; ++who) {
        hipEvent_t t0, t1;
        CK(hipEventCreate(&t0));
        CK(hipEventCreate(&t1));
        hipLaunchKernelGGL((probe<NACC, KIND>), dim3(256), dim3(512), 0, 0, d, mi, vi, who);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, 0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<NACC, KIND>), dim3(256), dim3(512), 0, 0, d, mi, vi, who);
        CK(hipEventRecord(t1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, t0, t1));
        printf("accumulators %d, VALU kind %d, %-16s %8.1f us per launch\n", NACC, KIND, names[who], ms * 1e3 / 5);
    }
}

int main(int argc, char** argv) {
    float* d;
    CK(hipMalloc(&d, 256 * 512 * 4));
    const int mi = 16384;                                   // MFMAs per wave
    const int vi = argc > 1 ? atoi(argv[1]) : 16384;        // 4 packed FMAs per iteration
    printf("%d MFMAs per wave (waves 0-3), %d x 4 v_pk_fma_f32 per wave (waves 4-7)\n", mi, vi);
    run<1, 0>(d, mi, vi);
    run<4, 0>(d, mi, vi);
    run<4, 1>(d, mi, vi);
    run<4, 2>(d, mi, vi);
    run<4, 3>(d, mi, vi);
    return 0;
}
