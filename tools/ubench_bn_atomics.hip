// Stand-alone microbenchmark (gfx950): how should a train-mode BatchNorm's statistics travel from the kernel that produces the
// BN input to the kernel that consumes the normalised values?
//   A  producer -> per-workgroup partial rows (double) -> bn_finalize-like kernel (few workgroups) -> consumer reads scale / shift
//   B  producer -> order-independent FIXED-POINT int64 atomics (2 limbs per sum: value = hi * 2^-20 + lo * 2^-70, no return value)
//      -> consumer derives scale / shift from the four integers of its channel in its prologue (no kernel in between)
// Both are bit-wise deterministic (A: fixed order; B: integer addition commutes).  Shapes: one stage-1 ShuffleNet unit of the
// benchmark (G = 4 time slices x Mg = 12288 rows x C = 116 channels), nb workgroups per group.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench_bn_atomics.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int C = 116, G = 4, CX = 29, CY = 8;     // 4 channels per thread, 232 threads

__device__ __forceinline__ void fx_add(long long* slot, double v) {
    const long long hi = __double2ll_rn(v * 1048576.0);
    const double r = v - (double)hi * (1.0 / 1048576.0);
    const long long lo = __double2ll_rn(r * 1180591620717411303424.0);
    atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)hi);
    atomicAdd(reinterpret_cast<unsigned long long*>(slot) + 1, (unsigned long long)lo);
}
__device__ __forceinline__ double fx_get(const long long* slot) {
    return (double)slot[0] * (1.0 / 1048576.0) + (double)slot[1] * (1.0 / 1180591620717411303424.0);
}

// MODE 2: the same fixed-point sums, but WORKGROUP-scope atomics (performed in the issuing XCD's L2, no trip to the memory side) into
// one replica per XCD (8 replicas, indexed by HW_REG_XCC_ID); the consumer adds the 8 replicas.  A replica is only ever touched
// by workgroups of one XCD, so no cross-L2 coherence is needed inside the kernel; the kernel boundary publishes it.
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7;
}
__device__ __forceinline__ void fx_add_wg(long long* slot, double v) {
    const long long hi = __double2ll_rn(v * 1048576.0);
    const double r = v - (double)hi * (1.0 / 1048576.0);
    const long long lo = __double2ll_rn(r * 1180591620717411303424.0);
    __hip_atomic_fetch_add(slot, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(slot + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// producer: column (sum, sum^2) of its rows; MODE 0: partial row, MODE 1: atomics
template <int MODE>
__global__ void __launch_bounds__(256) producer(const float* __restrict__ y, int Mg, int rb, double* __restrict__ part, long long* __restrict__ acc) {
    __shared__ double sm[CY][4][CX];
    const int tx = threadIdx.x, ty = threadIdx.y, g = blockIdx.y, nb = gridDim.x;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    double a[2][4] = {};
    for (int r = r0 + ty; r < r1; r += CY) {
        const float4 v = *reinterpret_cast<const float4*>(y + ((int64_t)g * Mg + r) * C + tx * 4);
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[0][i] += f[i]; a[1][i] += (double)f[i] * f[i]; }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sm[ty][i][tx] = a[q][i];
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { double s = a[q][i]; for (int yy = 1; yy < CY; ++yy) s += sm[yy][i][tx]; a[q][i] = s; }
        }
        __syncthreads();
    }
    if (ty == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tx * 4 + i;
                if (MODE == 0) part[(((int64_t)g * nb + blockIdx.x) * 2 + q) * C + c] = a[q][i];
                else if (MODE == 1) fx_add(acc + (((int64_t)g * C + c) * 2 + q) * 2, a[q][i]);
                else fx_add_wg(acc + ((((int64_t)xcc_id() * G + g) * C + c) * 2 + q) * 2, a[q][i]);
            }
    }
}

__global__ void __launch_bounds__(512) finalize(const double* __restrict__ part, int nb, int Mg, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ stats) {
    __shared__ double sm[2][64][8];
    const int tx = threadIdx.x, ty = threadIdx.y;       // (8, 64)
    const int c = blockIdx.x * 8 + tx;
    for (int g = 0; g < G; ++g) {
        double s = 0, q = 0;
        if (c < C) for (int b = ty; b < nb; b += 64) { s += part[(((int64_t)g * nb + b) * 2 + 0) * C + c]; q += part[(((int64_t)g * nb + b) * 2 + 1) * C + c]; }
        sm[0][ty][tx] = s; sm[1][ty][tx] = q;
        __syncthreads();
        if (ty == 0 && c < C) {
            s = q = 0;
            for (int yy = 0; yy < 64; ++yy) { s += sm[0][yy][tx]; q += sm[1][yy][tx]; }
            const double mean = s / Mg; double var = q / Mg - mean * mean; if (var < 0) var = 0;
            const float inv = (float)(1.0 / sqrt(var + 1e-3));
            stats[(0 * G + g) * C + c] = gamma[c] * inv;
            stats[(1 * G + g) * C + c] = beta[c] - (float)mean * gamma[c] * inv;
        }
        __syncthreads();
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) consumer(const float* __restrict__ y, float* __restrict__ out, int Mg, int rb, const float* __restrict__ stats,
                                                const long long* __restrict__ acc, const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int tx = threadIdx.x, ty = threadIdx.y, g = blockIdx.y;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tx * 4 + i;
        if (MODE == 0) { sc[i] = stats[(0 * G + g) * C + c]; sh[i] = stats[(1 * G + g) * C + c]; }
        else {
            long long t[4] = {0, 0, 0, 0};
            const int nrep = MODE == 2 ? 8 : 1;
            for (int x = 0; x < nrep; ++x) {
                const long long* p = acc + (((int64_t)x * G + g) * C + c) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] += p[k];
            }
            const double s = fx_get(t), q = fx_get(t + 2);
            const double mean = s / Mg; double var = q / Mg - mean * mean; if (var < 0) var = 0;
            const float inv = (float)(1.0 / sqrt(var + 1e-3));
            sc[i] = gamma[c] * inv; sh[i] = beta[c] - (float)mean * gamma[c] * inv;
        }
    }
    for (int r = r0 + ty; r < r1; r += CY) {
        const int64_t o = ((int64_t)g * Mg + r) * C + tx * 4;
        const float4 v = *reinterpret_cast<const float4*>(y + o);
        *reinterpret_cast<float4*>(out + o) = make_float4(fmaf(sc[0], v.x, sh[0]), fmaf(sc[1], v.y, sh[1]), fmaf(sc[2], v.z, sh[2]), fmaf(sc[3], v.w, sh[3]));
    }
}

int main() {
    const int Mg = 12288;
    const size_t n = (size_t)G * Mg * C;
    float *y, *out, *gamma, *beta, *stats; double* part; long long* acc;
    CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
    CK(hipMalloc(&stats, 2 * G * C * 4)); CK(hipMalloc(&part, (size_t)G * 2048 * 2 * C * 8)); CK(hipMalloc(&acc, (size_t)8 * G * C * 4 * 8));
    std::vector<float> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 1.f;
    CK(hipMemcpy(y, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> one(C, 1.f), zero(C, 0.f);
    CK(hipMemcpy(gamma, one.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, zero.data(), C * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nb : {32, 64, 128, 256, 512}) {
        const int rb = (Mg + nb - 1) / nb;
        dim3 grid(nb, G), block(CX, CY), cgrid(512, G);
        const int crb = (Mg + 511) / 512;
        float ms[4];
        for (int mode = 0; mode < 4; ++mode) {
            const int iters = 300;
            for (int it = -20; it < iters; ++it) {
                if (it == 0) CK(hipEventRecord(e0, st));
                if (mode == 0) {
                    hipLaunchKernelGGL(producer<0>, grid, block, 0, st, y, Mg, rb, part, acc);
                    hipLaunchKernelGGL(finalize, dim3((C + 7) / 8), dim3(8, 64), 0, st, part, nb, Mg, gamma, beta, stats);
                    hipLaunchKernelGGL(consumer<0>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
                } else if (mode == 1) {
                    CK(hipMemsetAsync(acc, 0, (size_t)G * C * 4 * 8, st));       // (the engine zeroes ALL accumulators of a pass with one memset)
                    hipLaunchKernelGGL(producer<1>, grid, block, 0, st, y, Mg, rb, part, acc);
                    hipLaunchKernelGGL(consumer<1>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
                } else if (mode == 2) {
                    hipLaunchKernelGGL(producer<1>, grid, block, 0, st, y, Mg, rb, part, acc);      // no memset (values grow; timing only)
                    hipLaunchKernelGGL(consumer<1>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
                } else {
                    hipLaunchKernelGGL(producer<2>, grid, block, 0, st, y, Mg, rb, part, acc);      // per-XCD replicas, L2 atomics
                    hipLaunchKernelGGL(consumer<2>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
                }
            }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
            ms[mode] = ms[mode] / iters * 1e3f;
        }
        // correctness: B == A
        CK(hipMemsetAsync(acc, 0, (size_t)8 * G * C * 4 * 8, st));
        hipLaunchKernelGGL(producer<0>, grid, block, 0, st, y, Mg, rb, part, acc);
        hipLaunchKernelGGL(finalize, dim3((C + 7) / 8), dim3(8, 64), 0, st, part, nb, Mg, gamma, beta, stats);
        hipLaunchKernelGGL(consumer<0>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
        std::vector<float> oa(n), ob(n);
        CK(hipStreamSynchronize(st)); CK(hipMemcpy(oa.data(), out, n * 4, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(producer<1>, grid, block, 0, st, y, Mg, rb, part, acc);
        hipLaunchKernelGGL(consumer<1>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
        CK(hipStreamSynchronize(st)); CK(hipMemcpy(ob.data(), out, n * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < n; ++i) diff += oa[i] != ob[i];
        CK(hipMemsetAsync(acc, 0, (size_t)8 * G * C * 4 * 8, st));
        hipLaunchKernelGGL(producer<2>, grid, block, 0, st, y, Mg, rb, part, acc);
        hipLaunchKernelGGL(consumer<2>, cgrid, block, 0, st, y, out, Mg, crb, stats, acc, gamma, beta);
        CK(hipStreamSynchronize(st)); CK(hipMemcpy(ob.data(), out, n * 4, hipMemcpyDeviceToHost));
        size_t diff2 = 0; for (size_t i = 0; i < n; ++i) diff2 += oa[i] != ob[i];
        printf("nb=%4d workgroups/group: A partials+finalize %.2f us | B device-scope atomics+memset %.2f us | B atomics only %.2f us | C per-XCD L2 atomics %.2f us | "
               "elements differing B vs A: %zu, C vs A: %zu of %zu\n", nb, ms[0], ms[1], ms[2], ms[3], diff, diff2, n);
    }
    return 0;
}
