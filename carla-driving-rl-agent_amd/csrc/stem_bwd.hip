// One-pass backward of the stem block  conv3x3/s2 -> BatchNorm(train)+ReLU6 -> max-pool  from the POOLED gradient
// (reference core/architectures.py:159-161), gfx950.
//
// The stem conv is the first layer: no input gradient is needed, only dW, db, dgamma, dbeta.  With
//     dy = k1 * (dz - k2 - xhat * k3),   k2 = mean(dz), k3 = mean(dz * xhat)   (per time slice g and channel c)
// the filter gradient is LINEAR in three sums that do not depend on k2 / k3:
//     dW[:, c] = sum_g k1[g,c] * ( P^T dz  -  k2[g,c] * P^T 1  -  k3[g,c] * P^T xhat )[:, c]
// (P = im2col patch matrix of the observations).  So ONE pass over the data produces, per time slice,
//     P^T dz, P^T xhat, P^T 1, sum dz, sum xhat, sum dz*xhat
// and a tiny finalize kernel derives k2, k3, dgamma, dbeta, dW and db -- the two-phase BatchNorm backward (sums,
// finalize, apply) and its 255 MB intermediates vanish, the pool gather is evaluated once.
//
// The three products are one MFMA TN product  P_ext^T D_ext  over 128-row slices staged in LDS: P_ext = 27 patch taps + a
// ones column (32 rows x 28), D_ext = [dz (C) | xhat (C) | dz*xhat (C) | 1] packed into 3C+1 <= 96 columns = three 32x32
// accumulators per wave.  The D tile is filled 4 channels at a time (one pool gather + one float4 load of y per item).
//   column c        : dz       -> rows 0..26: P^T dz,   row 27 (ones tap): sum dz
//   column C + c    : xhat     -> rows 0..26: P^T xhat, row 27: sum xhat
//   column 2C + c   : dz*xhat  -> row 27: sum dz*xhat
//   column 3C       : 1        -> rows 0..26: P^T 1, row 27: pixel count
// (A register-fragment variant without LDS -- every lane gathering its own element -- was measured at 1157 us: it cannot
// vectorise the gather over channels; this LDS form runs the same work in ~1/3 of that.)
// Partials: float [block][group][3][28][32], reduced in fixed order (double accumulation) -> deterministic.
#include "colreduce.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SBD_TILE (3 * 28 * 32)
#define SBD_LDD 97

__global__ void __launch_bounds__(256) stem_bwd_onepass_kernel(const float* __restrict__ x, PoolSrc ps, const float* __restrict__ y,
                                                               const float* __restrict__ stats, float* __restrict__ part, int B,
                                                               int T, int H, int W, int Ho, int Wo, int Cout, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Pt = smem;                                   // [4][32][29]
    float* Dt = smem + 4 * 32 * 29;                     // [4][32][SBD_LDD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lcol = lane & 31, lk = lane >> 5;
    const int g = blockIdx.y;
    const int Mg = B * Ho * Wo;
    const int r0 = blockIdx.x * rows_per, r1 = min(r0 + rows_per, Mg);
    float* P = Pt + wave * 32 * 29;
    float* D = Dt + wave * 32 * SBD_LDD;
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = 0.0f;
    for (int i = lane; i < 32 * 29; i += 64) P[i] = 0.0f;               // zero the padding columns once
    for (int i = lane; i < 32 * SBD_LDD; i += 64) D[i] = 0.0f;
    const int C4 = Cout >> 2, GC = T * Cout;
    for (int base = r0; base < r1; base += 128) {
        const int wrow0 = base + wave * 32;
        __syncthreads();
        {   // patch tile: lane l owns row (l & 31) and the 14 columns [14*(l>>5), +14): one row decode per lane per slice
            const int r = lane & 31, half = lane >> 5;
            const int row = wrow0 + r;
            const bool ok = row < r1;
            const float* xp = x;
            if (ok) {
                const int ox = row % Wo;
                const int t2 = row / Wo;
                const int oy = t2 % Ho;
                const int b = t2 / Ho;
                xp = x + ((((int64_t)b * T + g) * H + 2 * oy) * W + 2 * ox) * 3;
            }
#pragma unroll
            for (int jj = 0; jj < 14; ++jj) {
                const int64_t o0 = (int64_t)(jj / 9) * W * 3 + (jj % 9);
                const int64_t o1 = (int64_t)((14 + jj) / 9) * W * 3 + ((14 + jj) % 9);
                float val = 0.0f;
                if (ok) val = (half && jj == 13) ? 1.0f : xp[half ? o1 : o0];
                P[r * 29 + half * 14 + jj] = val;
            }
        }
        for (int idx = lane; idx < 32 * C4; idx += 64) {
            const int r = idx / C4, c0 = (idx - r * C4) * 4;
            const int row = wrow0 + r;
            float dz[4] = {0.0f, 0.0f, 0.0f, 0.0f}, xh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (row < r1) {
                const int64_t grow = (int64_t)g * Mg + row;
                const VecF<4> d = pool_gather<4>(ps, grow, c0, Cout);
                const VecF<4> v = vload<4>(y + grow * Cout + c0);
                const float* sp = stats + g * Cout + c0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = fmaf(sp[2 * GC + i], v.v[i], sp[3 * GC + i]);
                    dz[i] = relu6_open(z) ? d.v[i] : 0.0f;
                    xh[i] = (v.v[i] - sp[i]) * sp[GC + i];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                D[r * SBD_LDD + c0 + i] = dz[i];
                D[r * SBD_LDD + Cout + c0 + i] = xh[i];
                D[r * SBD_LDD + 2 * Cout + c0 + i] = dz[i] * xh[i];
            }
        }
        if (lane < 32) D[lane * SBD_LDD + 3 * Cout] = (wrow0 + lane) < r1 ? 1.0f : 0.0f;
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 32; mm += 2) {
            const float a = P[(mm + lk) * 29 + (lcol < 28 ? lcol : 28)];
            const float* dr = &D[(mm + lk) * SBD_LDD + lcol];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dr[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dr[32], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dr[64], acc2, 0, 0, 0);
        }
    }
    // fold the 4 waves (fixed order) and write one partial per workgroup: [b][g][3][28][32]
    float* out = part + ((int64_t)blockIdx.x * gridDim.y + g) * SBD_TILE;
    float* red = smem;                                  // [4][32][33] (the tiles are dead)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const f32x16& acc = t == 0 ? acc0 : (t == 1 ? acc1 : acc2);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 33 + lcol] = acc[r];
        __syncthreads();
        for (int idx = tid; idx < 28 * 32; idx += 256) {
            const int k = idx >> 5, n = idx & 31;
            out[t * 28 * 32 + idx] = (red[(0 * 32 + k) * 33 + n] + red[(1 * 32 + k) * 33 + n]) +
                                     (red[(2 * 32 + k) * 33 + n] + red[(3 * 32 + k) * 33 + n]);
        }
    }
}

// red: [G][3][28][32] (sums over the workgroups of each time slice).  One workgroup.
__global__ void __launch_bounds__(1024) stem_bwd_finalize_kernel(const float* __restrict__ red, const float* __restrict__ stats, int G,
                                                               int Mg, int Cout, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ coef,
                                                               float* __restrict__ dw, float* __restrict__ db) {
    const int GC = G * Cout;
    const double n = (double)Mg;
    for (int idx = threadIdx.x; idx < 28 * Cout; idx += blockDim.x) {
        const int k = idx / Cout, c = idx - k * Cout;
        double acc = 0.0, dg = 0.0, dbt = 0.0;
        for (int g = 0; g < G; ++g) {
            const float* R = red + (int64_t)g * SBD_TILE;
            // packed columns: q -> tile q / 32, column q % 32;  dz: c, xhat: Cout + c, dz*xhat: 2*Cout + c, ones: 3*Cout
            auto at = [&](int row, int q) { return (double)R[(q >> 5) * 896 + row * 32 + (q & 31)]; };
            const double sdz = at(27, c), sxh = at(27, Cout + c), sdx = at(27, 2 * Cout + c);
            const double k1 = stats[2 * GC + g * Cout + c], k2 = sdz / n, k3 = sdx / n;
            if (k < 27) {
                const double a1 = at(k, c), a3 = at(k, Cout + c), a2 = at(k, 3 * Cout);
                acc += k1 * (a1 - k2 * a2 - k3 * a3);
            } else {
                acc += k1 * (sdz - k2 * n - k3 * sxh);           // bias gradient (analytically 0)
                dg += sdx;
                dbt += sdz;
                coef[0 * GC + g * Cout + c] = (float)k1;
                coef[1 * GC + g * Cout + c] = (float)k2;
                coef[2 * GC + g * Cout + c] = (float)k3;
            }
        }
        if (k < 27) {
            dw[k * Cout + c] = (float)acc;
        } else {
            db[c] = (float)acc;
            dgamma[c] = (float)dg;
            dbeta[c] = (float)dbt;
        }
    }
}

bool stem_bwd_direct_supported(int Cout) { return Cout >= 4 && (Cout % 4) == 0 && 3 * Cout + 1 <= 96; }

static int sbd_nb(int Mg) {
    int nb = cdiv(Mg, 512);               // >= 512 pixel rows per workgroup
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return nb;
}

int64_t stem_bwd_direct_ws_floats(int B, int T, int H, int W) {
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    return (int64_t)(sbd_nb(B * Ho * Wo) + 1) * T * SBD_TILE;
}

int stem_bwd_direct(const float* x, const PoolSrc& ps, const float* y, const float* stats, float* dgamma, float* dbeta, float* coef,
                    float* dw, float* db, int B, int T, int H, int W, int Cout, float* ws, hipStream_t st) {
    if (!stem_bwd_direct_supported(Cout)) {
        set_error("stem_bwd_direct: Cout=%d not supported (multiple of 4, <= 28)", Cout);
        return -1;
    }
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const int Mg = B * Ho * Wo;
    const int nb = sbd_nb(Mg);
    const int rows_per = cdiv(cdiv(Mg, nb), 128) * 128;
    const int nbu = cdiv(Mg, rows_per);
    float* part = ws;
    float* red = ws + (int64_t)nbu * T * SBD_TILE;
    const size_t lds = (size_t)(4 * 32 * 29 + 4 * 32 * SBD_LDD) * sizeof(float);
    hipLaunchKernelGGL(stem_bwd_onepass_kernel, dim3(nbu, T), dim3(256), lds, st, x, ps, y, stats, part, B, T, H, W, Ho, Wo, Cout, rows_per);
    CDRL_LAUNCH_CHECK();
    const int64_t n = (int64_t)T * SBD_TILE;
    CDRL_TRY(reduce_partials_f32(part, nbu, n, n, red, 0, st));
    hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, red, stats, T, Mg, Cout, dgamma, dbeta, coef, dw, db);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
