// GRU gate kernels (gfx950).  Keras GRU v2 cell, reset_after=True, gate order z, r, h
// (reference core/networks.py:47-50):
//   z = sig(xz + hz); r = sig(xr + hr); hh = tanh(xh + r * hh_p); h' = z*h + (1-z)*hh
// The projections xp = x K + b0 and hp = h R + b1 are MFMA GEMMs (gemm.hip); these kernels are
// the fused elementwise gate math and its BPTT counterpart.
#include "cdrl_kernels.h"

namespace cdrl {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void gru_gates_fwd_kernel(const float* __restrict__ xp, const float* __restrict__ hp,
                                     const float* __restrict__ hprev, float* __restrict__ z, float* __restrict__ r,
                                     float* __restrict__ hh, float* __restrict__ hnew, int B, int u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * u) return;
    const int b = i / u, j = i % u;
    const float* x = xp + (int64_t)b * 3 * u;
    const float* h = hp + (int64_t)b * 3 * u;
    const float zz = sigm(x[j] + h[j]);
    const float rr = sigm(x[u + j] + h[u + j]);
    const float cand = tanhf(x[2 * u + j] + rr * h[2 * u + j]);
    z[i] = zz;
    r[i] = rr;
    hh[i] = cand;
    hnew[i] = zz * hprev[i] + (1.0f - zz) * cand;
}

int gru_gates_fwd(const float* xp, const float* hp, const float* hprev, float* z, float* r, float* hh, float* hnew,
                  int B, int u, hipStream_t st) {
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(cdiv(B * u, 256)), dim3(256), 0, st, xp, hp, hprev, z, r, hh, hnew, B, u);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void gru_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z,
                                     const float* __restrict__ r, const float* __restrict__ hh,
                                     const float* __restrict__ hp, const float* __restrict__ hprev,
                                     float* __restrict__ dxp, float* __restrict__ dhp, float* __restrict__ dhprev, int B,
                                     int u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * u) return;
    const int b = i / u, j = i % u;
    const float g = dh[i], zz = z[i], rr = r[i], cand = hh[i];
    const float hhp = hp[(int64_t)b * 3 * u + 2 * u + j];
    const float dcand = g * (1.0f - zz);
    const float dz = g * (hprev[i] - cand);
    const float dpre_h = dcand * (1.0f - cand * cand);
    const float dr = dpre_h * hhp;
    const float dpre_z = dz * zz * (1.0f - zz);
    const float dpre_r = dr * rr * (1.0f - rr);
    float* dx = dxp + (int64_t)b * 3 * u;
    float* dhh = dhp + (int64_t)b * 3 * u;
    dx[j] = dpre_z;
    dx[u + j] = dpre_r;
    dx[2 * u + j] = dpre_h;
    dhh[j] = dpre_z;
    dhh[u + j] = dpre_r;
    dhh[2 * u + j] = dpre_h * rr;
    dhprev[i] = g * zz;
}

int gru_gates_bwd(const float* dh, const float* z, const float* r, const float* hh, const float* hp,
                  const float* hprev, float* dxp, float* dhp, float* dhprev, int B, int u, hipStream_t st) {
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(cdiv(B * u, 256)), dim3(256), 0, st, dh, z, r, hh, hp, hprev, dxp, dhp,
                       dhprev, B, u);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
