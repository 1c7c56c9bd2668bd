// GRU time step as ONE kernel per direction (gfx950, v_mfma_f32_16x16x4_f32).
//
// Keras GRU v2 cell, reset_after=True, gate order z, r, h (reference core/networks.py:47-50):
//   hp = h R + b1;  z = sig(xz + hz); r = sig(xr + hr); hh = tanh(xh + r * hp_h); h' = z*h + (1-z)*hh
// with xp = x K + b0 precomputed for all T steps by one batched GEMM (gemm.hip).
//
// The recurrence couples only the columns of one batch row, so a workgroup owns a 16-row x 16-unit tile of the step:
//   forward : the recurrent product for its three gate column tiles (z, r, h) from an LDS-staged copy of h[rows, :]
//             (K = u split over the four waves, fixed-order fold through LDS), then the gate math and all saved tensors;
//   backward: the gate derivatives of its 16 rows for ALL 3u pre-activations into LDS (that is the A operand of
//             dh_prev = dhp R^T; the column-tile-0 workgroups also write dxp / dhp for the weight gradients), the product
//             against R^T (K = 3u split over the waves), plus the direct term dh * z.
// Before: recurrent GEMM + split-K reduce + gate kernel = 3 launches per step and direction on the critical stream.
#include "cdrl_kernels.h"

namespace cdrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// Tile = 16 batch rows x 16 hidden units per workgroup (v_mfma_f32_16x16x4_f32; B = u = 256 -> 256 workgroups, one per CU:
// the step is a latency chain, so the tile is sized for parallelism, not for operand reuse -- R stays L2-resident).
constexpr int GT = 16;

struct GruFwdArgs {
    const float* xp;      // [B][3u]
    const float* hprev;   // [B][u]
    const float* R;       // [u][3u]
    const float* b1;      // [3u]
    float *z, *r, *hh, *hp, *hnew;
    View out;             // optional second destination of hnew (the concat view after the last step)
    int B, u;
};

__global__ void __launch_bounds__(256) gru_step_fwd_kernel(GruFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int u = a.u, U3 = 3 * u, LDA = u + 4;      // +4: rows stay 16-byte aligned, fragment reads spread over the banks
    float* As = smem;                               // [16][u + 4]
    float* red = smem + GT * LDA;                   // [4 waves][3 gates][16][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * GT, c0 = blockIdx.y * GT;
    {   // rows r0 .. r0+15 of h are one contiguous block of 16*u floats
        const int u4 = u >> 2;
        for (int i = tid; i < GT * u4; i += 256) {
            const int r = i / u4, k4 = i - r * u4;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (r0 + r < a.B) v = *reinterpret_cast<const float4*>(a.hprev + (int64_t)(r0 + r) * u + 4 * k4);
            *reinterpret_cast<float4*>(As + r * LDA + 4 * k4) = v;
        }
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    const int KW = u >> 2, k0 = wave * KW;          // this wave's slice of K
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float* Rb = a.R + c0 + li;
#pragma unroll 8
    for (int s = 0; s < KW / 4; ++s) {
        const int k = k0 + 4 * s + lk;
        const float av = As[li * LDA + k];
        const float* rk = Rb + (int64_t)k * U3;
        const float b0 = rk[0], b1 = rk[u], b2 = rk[2 * u];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc[2], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * 3 + g) * GT + lk * 4 + i) * 17 + li] = acc[g][i];
    __syncthreads();
    {   // 256 threads = the 16 x 16 tile: fixed-order fold of the four K slices, then the gate math
        const int r = tid >> 4, c = tid & 15;
        const int row = r0 + r, j = c0 + c;
        if (row < a.B) {
            float hpv[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float s = red[((0 * 3 + g) * GT + r) * 17 + c];
                s += red[((1 * 3 + g) * GT + r) * 17 + c];
                s += red[((2 * 3 + g) * GT + r) * 17 + c];
                s += red[((3 * 3 + g) * GT + r) * 17 + c];
                hpv[g] = s + a.b1[g * u + j];
            }
            const float* x = a.xp + (int64_t)row * U3;
            const float zz = sigm(x[j] + hpv[0]);
            const float rr = sigm(x[u + j] + hpv[1]);
            const float cand = tanhf(x[2 * u + j] + rr * hpv[2]);
            const float hn = zz * As[r * LDA + j] + (1.0f - zz) * cand;
            const int64_t o = (int64_t)row * u + j;
            a.z[o] = zz;
            a.r[o] = rr;
            a.hh[o] = cand;
            float* hpo = a.hp + (int64_t)row * U3;
            hpo[j] = hpv[0];
            hpo[u + j] = hpv[1];
            hpo[2 * u + j] = hpv[2];
            a.hnew[o] = hn;
            if (a.out.p) a.out.p[(int64_t)row * a.out.ld + a.out.coff + j] = hn;
        }
    }
}

bool gru_step_supported(int u) { return u >= 16 && u % 16 == 0 && (size_t)(GT * (3 * u + 4) + 4 * GT * 17) * 4 <= 160 * 1024; }

int gru_step_fwd(const float* xp, const float* hprev, const float* R, const float* b1, float* z, float* r, float* hh, float* hp,
                 float* hnew, View out, int B, int u, hipStream_t st) {
    if (!gru_step_supported(u)) {
        set_error("gru_step_fwd: units %d unsupported (multiple of 16)", u);
        return -1;
    }
    GruFwdArgs a{xp, hprev, R, b1, z, r, hh, hp, hnew, out, B, u};
    const size_t lds = (size_t)(GT * (u + 4) + 4 * 3 * GT * 17) * sizeof(float);
    static LdsAttrOnce attr;
    if (attr.need()) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(cdiv(B, GT), u / GT), dim3(256), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

struct GruBwdArgs {
    View dh;              // gradient w.r.t. h_t: element (row, j) at dh.p[row * dh.ld + dh.coff + j]
    const float *z, *r, *hh, *hp, *hprev;
    const float* RT;      // [3u][u] = R^T
    float *dxp, *dhp;     // [B][3u]
    float* dhprev;        // [B][u] (may be null: first time step)
    int B, u;
};

__global__ void __launch_bounds__(256) gru_step_bwd_kernel(GruBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int u = a.u, U3 = 3 * u, LDD = U3 + 4;
    float* Ds = smem;                               // [16][3u + 4]: dhp of the workgroup's rows
    float* red = smem + GT * LDD;                   // [4][16][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * GT, c0 = blockIdx.y * GT;
    const bool writer = blockIdx.y == 0;
    {
        int r = tid / u, j = tid - r * u;           // 256 consecutive elements per pass; (r, j) advanced without divisions
        const int dr = 256 / u, dj = 256 - dr * u;
        for (int i = tid; i < GT * u; i += 256) {
            const int row = r0 + r;
            float pz = 0.0f, pr = 0.0f, ph = 0.0f, rr = 0.0f;
            if (row < a.B) {
                const int64_t o = (int64_t)row * u + j;
                const float g = a.dh.p[(int64_t)row * a.dh.ld + a.dh.coff + j];
                const float zz = a.z[o], cand = a.hh[o];
                rr = a.r[o];
                const float hhp = a.hp[(int64_t)row * U3 + 2 * u + j];
                const float dcand = g * (1.0f - zz);
                const float dz = g * (a.hprev[o] - cand);
                ph = dcand * (1.0f - cand * cand);
                const float drr = ph * hhp;
                pz = dz * zz * (1.0f - zz);
                pr = drr * rr * (1.0f - rr);
                if (writer) {
                    float* dx = a.dxp + (int64_t)row * U3;
                    float* dhh = a.dhp + (int64_t)row * U3;
                    dx[j] = pz;
                    dx[u + j] = pr;
                    dx[2 * u + j] = ph;
                    dhh[j] = pz;
                    dhh[u + j] = pr;
                    dhh[2 * u + j] = ph * rr;
                }
            }
            Ds[r * LDD + j] = pz;
            Ds[r * LDD + u + j] = pr;
            Ds[r * LDD + 2 * u + j] = ph * rr;
            r += dr;
            j += dj;
            if (j >= u) {
                j -= u;
                ++r;
            }
        }
    }
    if (!a.dhprev) return;
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    const int KW = U3 >> 2, k0 = wave * KW;
    f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};      // two chains: 40-cycle dependent latency
    const float* Tb = a.RT + c0 + li;
#pragma unroll 4
    for (int s = 0; s < KW / 8; ++s) {
        const int k = k0 + 8 * s + lk;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[li * LDD + k], Tb[(int64_t)k * u], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[li * LDD + k + 4], Tb[(int64_t)(k + 4) * u], acc[1], 0, 0, 0);
    }
    if (KW % 8) {                                   // KW = 3u/4 is a multiple of 4; odd multiples leave one step
        const int k = k0 + KW - 4 + lk;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[li * LDD + k], Tb[(int64_t)k * u], acc[0], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * GT + lk * 4 + i) * 17 + li] = acc[0][i] + acc[1][i];
    __syncthreads();
    {
        const int r = tid >> 4, c = tid & 15;
        const int row = r0 + r, j = c0 + c;
        if (row < a.B) {
            float s = red[(0 * GT + r) * 17 + c];
            s += red[(1 * GT + r) * 17 + c];
            s += red[(2 * GT + r) * 17 + c];
            s += red[(3 * GT + r) * 17 + c];
            const int64_t o = (int64_t)row * u + j;
            a.dhprev[o] = a.dh.p[(int64_t)row * a.dh.ld + a.dh.coff + j] * a.z[o] + s;
        }
    }
}

int gru_step_bwd(View dh, const float* z, const float* r, const float* hh, const float* hp, const float* hprev, const float* RT,
                 float* dxp, float* dhp, float* dhprev, int B, int u, hipStream_t st) {
    if (!gru_step_supported(u)) {
        set_error("gru_step_bwd: units %d unsupported (multiple of 16)", u);
        return -1;
    }
    GruBwdArgs a{dh, z, r, hh, hp, hprev, RT, dxp, dhp, dhprev, B, u};
    const size_t lds = (size_t)(GT * (3 * u + 4) + 4 * GT * 17) * sizeof(float);
    static LdsAttrOnce attr;
    if (attr.need()) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        attr.mark();
    }
    // first time step: no dh_prev consumer -> only the gate derivatives (one column of workgroups)
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(cdiv(B, GT), dhprev ? u / GT : 1), dim3(256), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
