// BatchNorm family + small elementwise kernels (gfx950).
//
// Reference semantics: Keras BatchNormalization applied **per time slice** with shared
// gamma/beta (reference core/architectures.py:44-57).  Rows of one time
// slice are contiguous (frame order f = t*B + b), so a statistics group is a row range.
//
// All kernels here are HBM-bound column-mapped streams: blockDim = (CX channel lanes, CY row
// lanes); consecutive lanes touch consecutive channels of one NHWC pixel (coalesced), and the
// per-channel reductions finish with an LDS tree over the CY row lanes.  Partial sums are kept
// in double so that E[x^2]-E[x]^2 and the (sum dz, sum dz*xhat) pair are exact to fp32 output
// precision and bit-wise deterministic (fixed reduction order, no atomics).
#include "colreduce.h"

namespace cdrl {

#define BN_EPS 1e-3f

template <int VEC, class T = float>
struct StatsF {
    View y;
    bool al;
    __device__ void operator()(int, int64_t row, int c0, double (*acc)[VEC]) const {
        const VecF<VEC> v = vload_view<VEC, T>(y, row, c0, 0, al);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const double d = (double)v.v[i];
            acc[0][i] += d;
            acc[1][i] += d * d;
        }
    }
};

int colstats(View y, int G, int Mg, int C, double* part, hipStream_t st, int at) {
    const bool al = view_aligned(y, vcol_geom(Mg, C).vec);
    if (at) return launch_vcolreduce_t<2, StatsF, bf16_t>(G, Mg, C, part, st, NB_STATS, y, al);
    return launch_vcolreduce_t<2, StatsF, float>(G, Mg, C, part, st, NB_STATS, y, al);
}

template <int VEC>
struct SumF {
    View x;
    bool al;
    __device__ void operator()(int, int64_t row, int c0, double (*acc)[VEC]) const {
        const VecF<VEC> v = vload_view<VEC>(x, row, c0, 0, al);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[0][i] += (double)v.v[i];
    }
};

int colsum(View x, int rows, int C, double* part, hipStream_t st) {
    const bool al = view_aligned(x, vcol_geom(rows, C).vec);
    return launch_vcolreduce<1, SumF>(1, rows, C, part, st, NB_STATS, x, al);
}

// ------------------------------------------------------------------------------------------
// forward finalize: stats[0]=mean, [1]=invstd, [2]=scale, [3]=shift, each [G][C]
// ------------------------------------------------------------------------------------------
// blockDim = (8 channel lanes, 128 partial lanes) (measured: 16x64 -> 21.84, 8x128 -> 21.78, 4x256 -> 21.99 ms/update-step).  The 64 lanes are split over the G groups (time slices) and, inside
// a group, over the nb per-block partials; every thread issues its loads in batches of FIN_U independent requests
// (a serial chain of nb dependent L2 round trips was 10 us of a 13 us kernel) and the lanes are combined through LDS
// in a fixed order -> deterministic.
#define FIN_CX 8
#define FIN_PY_MAX 128
#define FIN_U 8

// (sum_b p0[b*step], sum_b p1[b*step]) over b = first, first+stride, ... < count; 2*FIN_U loads in flight
__device__ __forceinline__ void strided_sum2(const double* __restrict__ p0, const double* __restrict__ p1, int64_t step,
                                             int first, int stride, int count, double& s0, double& s1) {
    s0 = 0.0;
    s1 = 0.0;
    for (int b0 = first; b0 < count; b0 += stride * FIN_U) {
        double v0[FIN_U], v1[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int b = b0 + u * stride;
            v0[u] = b < count ? p0[(int64_t)b * step] : 0.0;
            v1[u] = b < count ? p1[(int64_t)b * step] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            s0 += v0[u];
            s1 += v1[u];
        }
    }
}

// partial lanes of the finalize blocks (x 8 channel lanes = threads per block).  A 1024-thread block needs 16 free wave
// slots on one CU before it starts; in the backward pass the side stream keeps the CUs busy and the finalize kernels of the
// critical stream then queue behind it (outliers of 74 us for a 5.6 us kernel).  CDRL_FIN_PY = 128 | 64 | 32.
static int fin_py() {
    static const int v = 64;
    return v >= 128 ? 128 : (v >= 64 ? 64 : 32);
}

__device__ __forceinline__ int fin_group_slots(int G) { return G <= 1 ? 1 : (G <= 2 ? 2 : (G <= 4 ? 4 : 8)); }

// sums the [G][nb][2][C] partials of channel c for the groups g0 .. g0+Gp-1 into red[group slot][2][FIN_CX]
template <int FIN_PY>
__device__ __forceinline__ void fin_reduce(const double* __restrict__ part, int nb, int G, int C, int c, bool ok, int g0, int Gp,
                                           double (*sm)[FIN_PY][FIN_CX], double (*red)[2][FIN_CX]) {
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int L = FIN_PY / Gp;
    const int gi = g0 + ty / L, bl = ty % L;
    double s = 0.0, q = 0.0;
    if (ok && gi < G) {
        const double* base = part + ((int64_t)gi * nb * 2) * C + c;
        strided_sum2(base, base + C, (int64_t)2 * C, bl, L, nb, s, q);
    }
    sm[0][ty][tx] = s;
    sm[1][ty][tx] = q;
    __syncthreads();
    if (ok && ty < 2 * Gp) {
        const int gg = ty >> 1, qq = ty & 1;
        double a = 0.0;
        for (int y = 0; y < L; ++y) a += sm[qq][gg * L + y][tx];
        red[gg][qq][tx] = a;
    }
    __syncthreads();
}

template <int FIN_PY>
__global__ void __launch_bounds__(FIN_CX * FIN_PY) bn_finalize_kernel(const double* __restrict__ part, int nb, int G, int Mg, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ mov_mean, float* __restrict__ mov_var,
                                                          int bessel, int training, float* __restrict__ stats) {
    __shared__ double sm[2][FIN_PY][FIN_CX];
    __shared__ double red[8][2][FIN_CX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * FIN_CX + tx;
    const bool ok = c < C;
    const int GC = G * C;
    if (!training) {
        if (ok && ty == 0) {
            const float gm = gamma[c], bt = beta[c];
            const float mean = mov_mean[c];
            const float invstd = (float)(1.0 / sqrt((double)mov_var[c] + (double)BN_EPS));
            for (int g = 0; g < G; ++g) {
                stats[0 * GC + g * C + c] = mean;
                stats[1 * GC + g * C + c] = invstd;
                stats[2 * GC + g * C + c] = gm * invstd;
                stats[3 * GC + g * C + c] = bt - mean * gm * invstd;
            }
        }
        return;
    }
    float mm = 0.0f, mv = 0.0f, gm = 0.0f, bt = 0.0f;
    if (ok && ty == 0) {
        mm = mov_mean[c];
        mv = mov_var[c];
        gm = gamma[c];
        bt = beta[c];
    }
    const double n = (double)Mg;
    const float corr = (bessel && Mg > 1) ? (float)(n / (n - 1.0)) : 1.0f;
    const int Gp = fin_group_slots(G);
    for (int g0 = 0; g0 < G; g0 += Gp) {
        fin_reduce<FIN_PY>(part, nb, G, C, c, ok, g0, Gp, sm, red);
        if (ok && ty == 0) {
            const int ng = min(Gp, G - g0);
            for (int gg = 0; gg < ng; ++gg) {
                const int g = g0 + gg;
                const double mean = red[gg][0][tx] / n;
                double var = red[gg][1][tx] / n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float meanf = (float)mean, varf = (float)var;
                const float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
                stats[0 * GC + g * C + c] = meanf;
                stats[1 * GC + g * C + c] = invstd;
                stats[2 * GC + g * C + c] = gm * invstd;
                stats[3 * GC + g * C + c] = bt - meanf * gm * invstd;
                // Keras: moving -= (moving - value) * (1 - momentum), once per time slice (t ascending)
                mm = mm - (mm - meanf) * 0.01f;
                mv = mv - (mv - varf * corr) * 0.01f;
            }
        }
        __syncthreads();
    }
    if (ok && ty == 0) {
        mov_mean[c] = mm;
        mov_var[c] = mv;
    }
}

int bn_finalize(const double* part, int nb, int G, int Mg, int C, const float* gamma, const float* beta,
                float* mov_mean, float* mov_var, int bessel, int training, float* stats, hipStream_t st) {
    // block = (8 channel lanes, fin_py() partial lanes)
    const int py = fin_py();
    if (py == 128) hipLaunchKernelGGL(bn_finalize_kernel<128>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 128), 0, st, part, nb, G, Mg, C, gamma, beta, mov_mean, mov_var, bessel, training, stats);
    else if (py == 64) hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 64), 0, st, part, nb, G, Mg, C, gamma, beta, mov_mean, mov_var, bessel, training, stats);
    else hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 32), 0, st, part, nb, G, Mg, C, gamma, beta, mov_mean, mov_var, bessel, training, stats);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// Inference-mode statistics blocks of MANY BatchNorm layers in one launch (rollout inference, reference core/networks.py:181-193:
// every layer normalises with its moving statistics): the per-layer inference branch of bn_finalize above was a third of the
// launches of a predict() call, each doing a few hundred flops.  Same arithmetic as that branch.
__global__ void __launch_bounds__(64) bn_inference_stats_many_kernel(const BnInfEntry* __restrict__ tab) {
    const BnInfEntry e = tab[blockIdx.y];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= e.C) return;
    const int GC = e.G * e.C;
    const float gm = e.gamma[c], bt = e.beta[c];
    const float mean = e.mov_mean[c];
    const float invstd = (float)(1.0 / sqrt((double)e.mov_var[c] + (double)BN_EPS));
    for (int g = 0; g < e.G; ++g) {
        e.stats[0 * GC + g * e.C + c] = mean;
        e.stats[1 * GC + g * e.C + c] = invstd;
        e.stats[2 * GC + g * e.C + c] = gm * invstd;
        e.stats[3 * GC + g * e.C + c] = bt - mean * gm * invstd;
    }
}

int bn_inference_stats_many(const BnInfEntry* tab_dev, int n, int max_c, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(bn_inference_stats_many_kernel, dim3(cdiv(max_c, 64), n), dim3(64), 0, st, tab_dev);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Single-group BatchNorm over a few hundred rows (the control branches' and the trunk's dense BNs: B x 320..512) as ONE
// kernel per direction: statistics + finalize + apply (forward), sums + coefficients + apply (backward).  As three
// launches each they were 15-19 us of pure dispatch latency per BatchNorm on the stretch between the trunk's forward and
// backward.  block = (16 channel lanes, 16 row lanes); same arithmetic as bn_finalize / bn_bwd_finalize / the apply functors.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_small_fwd_kernel(View x, int M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ mov_mean,
                                                           float* __restrict__ mov_var, float* __restrict__ stats, View out) {
    __shared__ double sm[2][16][16];
    __shared__ float cf[2][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < C;
    const float* xp = x.p + x.coff + (ok ? c : 0);
    double s = 0.0, q = 0.0;
    for (int r0 = rl; r0 < M; r0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)min(r0 + 16 * u, M - 1) * x.ld];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 16 * u < M) {
                s += (double)v[u];
                q += (double)v[u] * (double)v[u];
            }
    }
    sm[0][rl][cl] = s;
    sm[1][rl][cl] = q;
    __syncthreads();
    if (rl == 0 && ok) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int y = 0; y < 16; ++y) {
            a += sm[0][y][cl];
            b += sm[1][y][cl];
        }
        const double n = (double)M;
        const double mean = a / n;
        double var = b / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean, varf = (float)var;
        const float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
        const float gm = gamma[c], bt = beta[c];
        const float sc = gm * invstd, sh = bt - meanf * gm * invstd;
        stats[0 * C + c] = meanf;
        stats[1 * C + c] = invstd;
        stats[2 * C + c] = sc;
        stats[3 * C + c] = sh;
        // Keras: moving -= (moving - value) * (1 - momentum); rank-2 inputs: no Bessel correction
        mov_mean[c] = mov_mean[c] - (mov_mean[c] - meanf) * 0.01f;
        mov_var[c] = mov_var[c] - (mov_var[c] - varf) * 0.01f;
        cf[0][cl] = sc;
        cf[1][cl] = sh;
    }
    __syncthreads();
    if (!ok) return;
    const float sc = cf[0][cl], sh = cf[1][cl];
    float* op = out.p + out.coff + c;
    for (int r0 = rl; r0 < M; r0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)min(r0 + 16 * u, M - 1) * x.ld];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 16 * u < M) op[(int64_t)(r0 + 16 * u) * out.ld] = fmaf(sc, v[u], sh);
    }
}

int bn_small_fwd(View x, int M, int C, const float* gamma, const float* beta, float* mov_mean, float* mov_var, float* stats,
                 View out, hipStream_t st) {
    hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, x, M, C, gamma, beta, mov_mean, mov_var, stats, out);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// backward: dz = dout (no activation), xhat from the raw input; dx = k1 * (dz - k2 - xhat * k3)
__global__ void __launch_bounds__(256) bn_small_bwd_kernel(View dout, View x, int M, int C, const float* __restrict__ stats,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef, float* __restrict__ dx) {
    __shared__ double sm[2][16][16];
    __shared__ float cf[3][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < C;
    const float* xp = x.p + x.coff + (ok ? c : 0);
    const float* dp = dout.p + dout.coff + (ok ? c : 0);
    const float mean = ok ? stats[0 * C + c] : 0.0f, inv = ok ? stats[1 * C + c] : 0.0f;
    double s = 0.0, q = 0.0;
    for (int r0 = rl; r0 < M; r0 += 16 * 8) {
        float v[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = min(r0 + 16 * u, M - 1);
            v[u] = xp[r * x.ld];
            d[u] = dp[r * dout.ld];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 16 * u < M) {
                const float xh = (v[u] - mean) * inv;
                s += (double)d[u];
                q += (double)d[u] * (double)xh;
            }
    }
    sm[0][rl][cl] = s;
    sm[1][rl][cl] = q;
    __syncthreads();
    if (rl == 0 && ok) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int y = 0; y < 16; ++y) {
            a += sm[0][y][cl];
            b += sm[1][y][cl];
        }
        const double n = (double)M;
        const float k1 = stats[2 * C + c], k2 = (float)(a / n), k3 = (float)(b / n);
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
        coef[0 * C + c] = k1;
        coef[1 * C + c] = k2;
        coef[2 * C + c] = k3;
        cf[0][cl] = k1;
        cf[1][cl] = k2;
        cf[2][cl] = k3;
    }
    __syncthreads();
    if (!ok) return;
    const float k1 = cf[0][cl], k2 = cf[1][cl], k3 = cf[2][cl];
    for (int r0 = rl; r0 < M; r0 += 16 * 8) {
        float v[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = min(r0 + 16 * u, M - 1);
            v[u] = xp[r * x.ld];
            d[u] = dp[r * dout.ld];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 16 * u < M) {
                const float xh = (v[u] - mean) * inv;
                dx[(int64_t)(r0 + 16 * u) * C + c] = k1 * (d[u] - k2 - xh * k3);
            }
    }
}

int bn_small_bwd(View dout, View x, int M, int C, const float* stats, float* dgamma, float* dbeta, float* coef, float* dx,
                 hipStream_t st) {
    hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, dout, x, M, C, stats, dgamma, dbeta, coef, dx);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// forward apply
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float z, int act) {
    if (act == ACT_RELU6) return fminf(fmaxf(z, 0.0f), 6.0f);
    return z;
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_apply_kernel(View y, int Mg, int C, int rb, int nloop, int GC,
                                                       const float* __restrict__ stats, int act, View dst,
                                                       int shuffle_ctot, bool al_in, bool al_out, View psrc, View pdst,
                                                       bool al_ps) {
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y;
    const int r0 = blockIdx.x * rb;
    const int r1 = min(r0 + rb, Mg);
    for (int l = 0; l < nloop; ++l) {
        const int c0 = (l * CX + tx) * VEC;
        if (c0 >= C) continue;
        VecF<VEC> sc, sh;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sc.v[i] = 1.0f;
            sh.v[i] = 0.0f;
        }
        if (stats) {
            sc = vload<VEC>(stats + 2 * GC + g * C + c0);
            sh = vload<VEC>(stats + 3 * GC + g * C + c0);
        }
        for (int r = r0 + ty; r < r1; r += CY) {
            const int64_t row = (int64_t)g * Mg + r;
            VecF<VEC> v = vload_view<VEC, T>(y, row, c0, 0, al_in);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v.v[i] = apply_act(fmaf(sc.v[i], v.v[i], sh.v[i]), act);
            vstore_view<VEC, T>(dst, row, c0, shuffle_ctot, al_out, v);
            if (psrc.p) {       // the unit's identity half goes through the same concat + shuffle store (same C channels)
                const VecF<VEC> pv = vload_view<VEC, T>(psrc, row, c0, 0, al_ps);
                vstore_view<VEC, T>(pdst, row, c0, shuffle_ctot, false, pv);
            }
        }
    }
}

// two adjacent floats at 4-byte alignment (one 8-byte access: global memory takes it unaligned)
struct __attribute__((packed, aligned(4))) F2U {
    float x, y;
};

// Fast path of bn_apply for the unit output (concat + channel shuffle as a destination permutation, optionally with the
// identity half): RU rows of loads are issued before the first store -- gfx9 counts loads and stores in one in-order
// counter, so "load, store, load, use" (the generic loop) waits for a store round trip per row.
template <int VEC, bool PASS, class T>
__global__ void __launch_bounds__(256) bn_apply_shuf_kernel(View y, int Mg, int C, int rb, int GC, const float* __restrict__ stats,
                                                            int act, View dst, int ctot, View psrc, View pdst) {
    const T* yp = vptr<T>(y);
    const T* psp = vptr<T>(psrc);
    T* dstp = vptr<T>(dst);
    T* pdp = vptr<T>(pdst);
    // (even ctot / 2 and even channel offsets: checked by the launcher through view alignment; 2-byte-aligned 4-byte stores are fine)
    constexpr bool PAIR = sizeof(T) == 2 && VEC == 4;
    constexpr int RU = 4;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CY = blockDim.y;
    const int g = blockIdx.y;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    const int c0 = tx * VEC;
    if (c0 >= C) return;
    const VecF<VEC> sc = vload<VEC>(stats + 2 * GC + g * C + c0), sh = vload<VEC>(stats + 3 * GC + g * C + c0);
    int dcol[VEC], pcol[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        dcol[i] = shuffle_dst(dst.coff + c0 + i, ctot);
        pcol[i] = PASS ? shuffle_dst(pdst.coff + c0 + i, ctot) : 0;
    }
    const int64_t gbase = (int64_t)g * Mg;
    for (int rr = r0 + ty; rr < r1; rr += CY * RU) {
        VecF<VEC> v[RU], pv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t row = gbase + min(rr + u * CY, r1 - 1);
            v[u] = vload<VEC>(yp + row * y.ld + y.coff + c0);
            if (PASS) pv[u] = vload<VEC>(psp + row * psrc.ld + psrc.coff + c0);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (rr + u * CY >= r1) break;
            const int64_t row = gbase + rr + u * CY;
            T* dr = dstp + row * dst.ld;
            if (PAIR) {     // bf16, 4 channels: the de-interleave sends (0, 2) and (1, 3) to two pairs of ADJACENT columns -> 4-byte stores
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = apply_act(fmaf(sc.v[i % VEC], v[u].v[i % VEC], sh.v[i % VEC]), act);
                *reinterpret_cast<uint32_t*>(dr + dcol[0]) = bf_pack(o[0], o[2]);
                *reinterpret_cast<uint32_t*>(dr + dcol[1 % VEC]) = bf_pack(o[1], o[3]);
                if (PASS) {
                    T* pr = pdp + row * pdst.ld;
                    *reinterpret_cast<uint32_t*>(pr + pcol[0]) = bf_pack(pv[u].v[0], pv[u].v[2 % VEC]);
                    *reinterpret_cast<uint32_t*>(pr + pcol[1 % VEC]) = bf_pack(pv[u].v[1 % VEC], pv[u].v[3 % VEC]);
                }
                continue;
            }
            if (sizeof(T) == 4 && VEC == 4) {      // float32, 4 channels: the same two pairs as 8-byte stores at 4-byte alignment
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = apply_act(fmaf(sc.v[i % VEC], v[u].v[i % VEC], sh.v[i % VEC]), act);
                *reinterpret_cast<F2U*>(dr + dcol[0]) = F2U{o[0], o[2]};
                *reinterpret_cast<F2U*>(dr + dcol[1 % VEC]) = F2U{o[1], o[3]};
                if (PASS) {
                    T* pr = pdp + row * pdst.ld;
                    *reinterpret_cast<F2U*>(pr + pcol[0]) = F2U{pv[u].v[0], pv[u].v[2 % VEC]};
                    *reinterpret_cast<F2U*>(pr + pcol[1 % VEC]) = F2U{pv[u].v[1 % VEC], pv[u].v[3 % VEC]};
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) stf(dr + dcol[i], apply_act(fmaf(sc.v[i], v[u].v[i], sh.v[i]), act));
            if (PASS) {
                T* pr = pdp + row * pdst.ld;
#pragma unroll
                for (int i = 0; i < VEC; ++i) stf(pr + pcol[i], pv[u].v[i]);
            }
        }
    }
}

template <class T>
static int bn_apply_t(View y, int G, int Mg, int C, const float* stats, int act, View dst, int shuffle_ctot,
                      hipStream_t st, const View* pass_src, const View* pass_dst) {
    {
        static const bool fast = true;
        const VColGeom g = vcol_geom(Mg, C, 2048);
        View ps{nullptr, 0, 0}, pd{nullptr, 0, 0};
        if (pass_src && pass_dst) {
            ps = *pass_src;
            pd = *pass_dst;
        }
        if (fast && stats && shuffle_ctot && g.nloop == 1 && g.vec >= 2 && view_aligned(y, g.vec) && (!ps.p || view_aligned(ps, g.vec))) {
            dim3 grid(g.nb, G), block(g.cx, g.cy);
            if (g.vec == 4) {
                if (ps.p) hipLaunchKernelGGL((bn_apply_shuf_kernel<4, true, T>), grid, block, 0, st, y, Mg, C, g.rb, G * C, stats, act, dst, shuffle_ctot, ps, pd);
                else hipLaunchKernelGGL((bn_apply_shuf_kernel<4, false, T>), grid, block, 0, st, y, Mg, C, g.rb, G * C, stats, act, dst, shuffle_ctot, ps, pd);
            } else {
                if (ps.p) hipLaunchKernelGGL((bn_apply_shuf_kernel<2, true, T>), grid, block, 0, st, y, Mg, C, g.rb, G * C, stats, act, dst, shuffle_ctot, ps, pd);
                else hipLaunchKernelGGL((bn_apply_shuf_kernel<2, false, T>), grid, block, 0, st, y, Mg, C, g.rb, G * C, stats, act, dst, shuffle_ctot, ps, pd);
            }
            CDRL_LAUNCH_CHECK();
            return 0;
        }
    }
    VColGeom g = vcol_geom(Mg, C, 2048);
    const bool ai = view_aligned(y, g.vec), ao = view_aligned(dst, g.vec);
    View ps{nullptr, 0, 0}, pd{nullptr, 0, 0};
    if (pass_src && pass_dst) {
        ps = *pass_src;
        pd = *pass_dst;
    }
    const bool aps = ps.p && view_aligned(ps, g.vec);
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    if (g.vec == 4)
        hipLaunchKernelGGL((bn_apply_kernel<4, T>), grid, block, 0, st, y, Mg, C, g.rb, g.nloop, G * C, stats, act, dst, shuffle_ctot, ai, ao, ps, pd, aps);
    else if (g.vec == 2)
        hipLaunchKernelGGL((bn_apply_kernel<2, T>), grid, block, 0, st, y, Mg, C, g.rb, g.nloop, G * C, stats, act, dst, shuffle_ctot, ai, ao, ps, pd, aps);
    else
        hipLaunchKernelGGL((bn_apply_kernel<1, T>), grid, block, 0, st, y, Mg, C, g.rb, g.nloop, G * C, stats, act, dst, shuffle_ctot, ai, ao, ps, pd, aps);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int bn_apply(View y, int G, int Mg, int C, const float* stats, int act, View dst, int shuffle_ctot,
             hipStream_t st, const View* pass_src, const View* pass_dst, int at) {
    if (at) return bn_apply_t<bf16_t>(y, G, Mg, C, stats, act, dst, shuffle_ctot, st, pass_src, pass_dst);
    return bn_apply_t<float>(y, G, Mg, C, stats, act, dst, shuffle_ctot, st, pass_src, pass_dst);
}

// BatchNorm apply + activation + global average pool in one pass over the raw conv output (the head of the tower: the 12288 x 768
// activated tensor is neither written nor read; same operation order as bn_apply followed by gap_fwd_kernel, so the same bits)
template <class T>
__global__ void bn_act_gap_fwd_kernel(const T* __restrict__ y, const float* __restrict__ stats, float* __restrict__ out, int N,
                                      int P, int C, int GC, int frames_per_group, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * C) return;
    const int c = (int)(i % C);
    const int64_t n = i / C;
    const int g = (int)(n / frames_per_group);
    const float sc = stats[2 * GC + g * C + c], sh = stats[3 * GC + g * C + c];
    float s = 0.0f;
    for (int p = 0; p < P; ++p) s += apply_act(fmaf(sc, ldf(y + (n * P + p) * C + c), sh), act);
    out[i] = s / (float)P;
}

int bn_act_gap_fwd(const float* y, const float* stats, float* out, int G, int frames_per_group, int P, int C, int act, hipStream_t st,
                   int at) {
    const int N = G * frames_per_group;
    if (at)
        hipLaunchKernelGGL(bn_act_gap_fwd_kernel<bf16_t>, dim3((unsigned)cdiv64((int64_t)N * C, 256)), dim3(256), 0, st,
                           reinterpret_cast<const bf16_t*>(y), stats, out, N, P, C, G * C, frames_per_group, act);
    else
        hipLaunchKernelGGL(bn_act_gap_fwd_kernel<float>, dim3((unsigned)cdiv64((int64_t)N * C, 256)), dim3(256), 0, st, y, stats, out, N, P, C,
                           G * C, frames_per_group, act);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <int VEC, class T = float>
struct BnBwdReduceF {
    View da;
    int shuffle_ctot;
    View y;
    const float* stats;
    int GC, C, act;
    bool al_da, al_y;
    bool use_pool;
    PoolSrc pool;
    View pgsrc, pgdst;      // optional: gradient of the unit's identity half, gathered through the same shuffle map
    bool al_pg;
    int bcast;              // > 0: da has one row per `bcast` rows of y and is divided by it (gradient of a global average pool)
    __device__ void operator()(int g, int64_t row, int c0, double (*acc)[VEC]) const {
        VecF<VEC> d;
        if (bcast > 0) {        // the pooled gradient (one row per frame) is a float32 tensor in either storage mode
            d = vload_view<VEC, float>(da, row / bcast, c0, 0, al_da);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d.v[i] = d.v[i] / (float)bcast;
        } else {
            d = use_pool ? pool_gather<VEC, T>(pool, row, c0, C) : vload_view<VEC, T>(da, row, c0, shuffle_ctot, al_da);
        }
        const VecF<VEC> v = vload_view<VEC, T>(y, row, c0, 0, al_y);
        const VecF<VEC> mean = vload<VEC>(stats + 0 * GC + g * C + c0), invstd = vload<VEC>(stats + 1 * GC + g * C + c0);
        if (pgsrc.p) {
            const VecF<VEC> pv = vload_view<VEC, T>(pgsrc, row, c0, shuffle_ctot, false);
            vstore_view<VEC, T>(pgdst, row, c0, 0, al_pg, pv);
        }
        if (act == ACT_RELU6) {
            const VecF<VEC> sc = vload<VEC>(stats + 2 * GC + g * C + c0), sh = vload<VEC>(stats + 3 * GC + g * C + c0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = fmaf(sc.v[i], v.v[i], sh.v[i]);
                if (!relu6_open(z)) d.v[i] = 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xh = (v.v[i] - mean.v[i]) * invstd.v[i];
            acc[0][i] += (double)d.v[i];
            acc[1][i] += (double)d.v[i] * (double)xh;
        }
    }
};

// Fast path of bn_bwd_reduce for the unit's last BatchNorm (dz gathered through the channel-shuffle map, ReLU6 mask,
// optionally the identity half's gradient gathered + stored in the same pass): same partial layout as the generic
// skeleton, but the per-channel columns / coefficients are computed once per thread and RU rows are loaded before any of
// them is consumed (the generic functor re-derives everything per row and, with its pass-through store between the loads,
// runs one row's round trip at a time; at 6-21 rows per thread the kernel was latency, not bandwidth).
template <int VEC, bool PASS, class T>
__global__ void __launch_bounds__(256) bn_bwd_reduce_shuf_kernel(View da, int ctot, const T* __restrict__ y,
                                                                 const float* __restrict__ stats, int GC, int C, int Mg, int rb,
                                                                 View pgs, View pgd, double* __restrict__ part) {
    extern __shared__ double sm[];   // [CY][VEC][CX]
    constexpr int RU = 4;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y, nb = gridDim.x;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    const int c0 = tx * VEC;
    const bool on = c0 < C;
    double acc[2][VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[0][i] = acc[1][i] = 0.0;
    if (on) {
        int dcol[VEC], pcol[VEC];
        float mean[VEC], inv[VEC], sc[VEC], sh[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            dcol[i] = shuffle_dst(da.coff + c0 + i, ctot);
            pcol[i] = PASS ? shuffle_dst(pgs.coff + c0 + i, ctot) : 0;
            mean[i] = stats[0 * GC + g * C + c0 + i];
            inv[i] = stats[1 * GC + g * C + c0 + i];
            sc[i] = stats[2 * GC + g * C + c0 + i];
            sh[i] = stats[3 * GC + g * C + c0 + i];
        }
        const int64_t gbase = (int64_t)g * Mg;
        for (int rr = r0 + ty; rr < r1; rr += CY * RU) {
            float dz[RU][VEC], pv[RU][VEC];
            VecF<VEC> yv[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t row = gbase + min(rr + u * CY, r1 - 1);          // clamped: unconditional loads
                const T* dr = vptr<T>(da) + row * da.ld;
                constexpr bool PAIR = sizeof(T) == 2 && VEC == 4;       // (see bn_apply_shuf_kernel: columns (0, 2) and (1, 3) are adjacent)
                // float32: the same pairs as 8-byte loads at 4-byte alignment (columns coff/2 + 2 tx: a wave's 4-byte loads covered
                // every second dword of its segment, twice)
                constexpr bool PAIRF = sizeof(T) == 4 && VEC == 4;
                if (PAIR) {
                    const uint32_t w0 = *reinterpret_cast<const uint32_t*>(dr + dcol[0]), w1 = *reinterpret_cast<const uint32_t*>(dr + dcol[1 % VEC]);
                    dz[u][0] = bf_lo(w0);
                    dz[u][2 % VEC] = bf_hi(w0);
                    dz[u][1 % VEC] = bf_lo(w1);
                    dz[u][3 % VEC] = bf_hi(w1);
                } else if (PAIRF) {
                    const F2U w0 = *reinterpret_cast<const F2U*>(dr + dcol[0]), w1 = *reinterpret_cast<const F2U*>(dr + dcol[1 % VEC]);
                    dz[u][0] = w0.x;
                    dz[u][2 % VEC] = w0.y;
                    dz[u][1 % VEC] = w1.x;
                    dz[u][3 % VEC] = w1.y;
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) dz[u][i] = ldf(dr + dcol[i]);
                }
                yv[u] = vload<VEC>(y + row * C + c0);
                if (PASS) {
                    const T* pr = vptr<T>(pgs) + row * pgs.ld;
                    if (PAIR) {
                        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(pr + pcol[0]), w1 = *reinterpret_cast<const uint32_t*>(pr + pcol[1 % VEC]);
                        pv[u][0] = bf_lo(w0);
                        pv[u][2 % VEC] = bf_hi(w0);
                        pv[u][1 % VEC] = bf_lo(w1);
                        pv[u][3 % VEC] = bf_hi(w1);
                    } else if (PAIRF) {
                        const F2U w0 = *reinterpret_cast<const F2U*>(pr + pcol[0]), w1 = *reinterpret_cast<const F2U*>(pr + pcol[1 % VEC]);
                        pv[u][0] = w0.x;
                        pv[u][2 % VEC] = w0.y;
                        pv[u][1 % VEC] = w1.x;
                        pv[u][3 % VEC] = w1.y;
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) pv[u][i] = ldf(pr + pcol[i]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (rr + u * CY >= r1) break;
                const int64_t row = gbase + rr + u * CY;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float v = yv[u].v[i];
                    const float z = fmaf(sc[i], v, sh[i]);
                    const float d = relu6_open(z) ? dz[u][i] : 0.0f;
                    const float xh = (v - mean[i]) * inv[i];
                    acc[0][i] += (double)d;
                    acc[1][i] += (double)d * (double)xh;
                }
                if (PASS) {
                    VecF<VEC> o;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o.v[i] = pv[u][i];
                    vstore<VEC>(vptr<T>(pgd) + row * pgd.ld + pgd.coff + c0, o);
                }
            }
        }
    }
    // block reduction over the row lanes, one quantity at a time (as vcolreduce_kernel)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (CY > 1) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) sm[(ty * VEC + i) * CX + tx] = acc[q][i];
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    double s = acc[q][i];
                    for (int yy = 1; yy < CY; ++yy) s += sm[(yy * VEC + i) * CX + tx];
                    acc[q][i] = s;
                }
            }
            __syncthreads();
        }
    }
    if (ty == 0 && on) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < VEC; ++i) part[(((int64_t)g * nb + blockIdx.x) * 2 + q) * C + c0 + i] = acc[q][i];
    }
}

template <int VEC, class T>
static void launch_bbr_shuf(const VColGeom& g, int G, hipStream_t st, View da, int ctot, const float* y, const float* stats, int C,
                            int Mg, View pgs, View pgd, double* part) {
    dim3 grid(g.nb, G), block(g.cx, g.cy);
    size_t sm = (size_t)g.cy * VEC * g.cx * sizeof(double);
    if (sm < 16) sm = 16;
    const T* yt = reinterpret_cast<const T*>(y);
    if (pgs.p)
        hipLaunchKernelGGL((bn_bwd_reduce_shuf_kernel<VEC, true, T>), grid, block, sm, st, da, ctot, yt, stats, G * C, C, Mg, g.rb, pgs, pgd, part);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_shuf_kernel<VEC, false, T>), grid, block, sm, st, da, ctot, yt, stats, G * C, C, Mg, g.rb, pgs, pgd, part);
}

int bn_bwd_reduce(View da, int shuffle_ctot, View y, int G, int Mg, int C, const float* stats, int act,
                  double* part, hipStream_t st, const PoolSrc* pool, const View* pass_gsrc, const View* pass_gdst, int bcast_rows, int at) {
    {
        static const bool fast = true;
        const VColGeom g = vcol_geom(Mg, C, NB_STATS);
        View pgs{nullptr, 0, 0}, pgd{nullptr, 0, 0};
        if (pass_gsrc && pass_gdst) {
            pgs = *pass_gsrc;
            pgd = *pass_gdst;
        }
        const bool ydense = y.ld == C && y.coff == 0 && view_aligned(y, g.vec);
        const bool pok = !pgs.p || (pgd.p && view_aligned(pgd, g.vec));
        if (fast && !pool && !bcast_rows && shuffle_ctot && act == ACT_RELU6 && g.nloop == 1 && ydense && pok && g.vec >= 2) {
            if (at) {
                if (g.vec == 4) launch_bbr_shuf<4, bf16_t>(g, G, st, da, shuffle_ctot, y.p, stats, C, Mg, pgs, pgd, part);
                else launch_bbr_shuf<2, bf16_t>(g, G, st, da, shuffle_ctot, y.p, stats, C, Mg, pgs, pgd, part);
            } else {
                if (g.vec == 4) launch_bbr_shuf<4, float>(g, G, st, da, shuffle_ctot, y.p, stats, C, Mg, pgs, pgd, part);
                else launch_bbr_shuf<2, float>(g, G, st, da, shuffle_ctot, y.p, stats, C, Mg, pgs, pgd, part);
            }
            CDRL_LAUNCH_CHECK();
            return 0;
        }
    }
    const int vec = vcol_geom(Mg, C).vec;
    PoolSrc ps{};
    if (pool) ps = *pool;
    View pgs{nullptr, 0, 0}, pgd{nullptr, 0, 0};
    if (pass_gsrc && pass_gdst) {
        pgs = *pass_gsrc;
        pgd = *pass_gdst;
    }
    if (at)
        return launch_vcolreduce_t<2, BnBwdReduceF, bf16_t>(G, Mg, C, part, st, NB_STATS, da, shuffle_ctot, y, stats, G * C, C, act,
                                                            pool ? false : view_aligned(da, vec), view_aligned(y, vec), pool != nullptr, ps,
                                                            pgs, pgd, pgd.p && view_aligned(pgd, vec), bcast_rows);
    return launch_vcolreduce_t<2, BnBwdReduceF, float>(G, Mg, C, part, st, NB_STATS, da, shuffle_ctot, y, stats, G * C, C, act,
                                                       pool ? false : view_aligned(da, vec), view_aligned(y, vec), pool != nullptr, ps,
                                                       pgs, pgd, pgd.p && view_aligned(pgd, vec), bcast_rows);
}

// BN-backward sums of a BatchNorm+ReLU6 whose output feeds a 3x3/s2 max-pool, in SCATTER form: iterate over the POOLED
// gradient (4x fewer elements than the pre-pool tensor) -- every pooled element contributes dp to exactly one pre-pool
// location (its saved argmax), so  sum_a dz[a] = sum_o dp[o]*mask(a(o))  and  sum_a dz[a]*xhat[a] = sum_o dp[o]*mask*xhat(y[a(o)]).
// Traffic: dp + argmax + one gathered y per pooled element, instead of y + up to 4 window probes per pre-pool element
// (the gather form took 506 us on the 255 MB stem tensor for ~70 us of HBM time).
template <int VEC, class T = float>
struct PoolBnReduceF {
    PoolSrc ps;
    const float* y;         // pre-pool BN input [N][H][W][C] (T elements)
    const float* stats;
    int GC, C;
    __device__ void operator()(int g, int64_t row, int c0, double (*acc)[VEC]) const {
        const int ox = (int)(row % ps.Wo);
        const int64_t q = row / ps.Wo;
        const int oy = (int)(q % ps.Ho);
        const int64_t n = q / ps.Ho;
        const VecF<VEC> d = vload<VEC>(reinterpret_cast<const T*>(ps.dp) + row * C + c0);
        const VecF<VEC> mean = vload<VEC>(stats + 0 * GC + g * C + c0), invstd = vload<VEC>(stats + 1 * GC + g * C + c0);
        const VecF<VEC> sc = vload<VEC>(stats + 2 * GC + g * C + c0), sh = vload<VEC>(stats + 3 * GC + g * C + c0);
        uint32_t am = 0;
        if (VEC == 4) am = *reinterpret_cast<const uint32_t*>(ps.argmax + row * C + c0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int k = (VEC == 4 ? (int)((am >> (8 * i)) & 0xffu) : (int)ps.argmax[row * C + c0 + i]) & 0x7f;      // (bit 7: ReLU6 flag of maxpool_bn_fwd)
            const int ky = k / 3, kx = k - 3 * ky;
            const int iy = 2 * oy - ps.pt + ky, ix = 2 * ox - ps.pl + kx;
            const float v = ldf(reinterpret_cast<const T*>(y) + ((n * ps.H + iy) * ps.W + ix) * C + c0 + i);
            const float z = fmaf(sc.v[i], v, sh.v[i]);
            if (relu6_open(z)) {
                const float xh = (v - mean.v[i]) * invstd.v[i];
                acc[0][i] += (double)d.v[i];
                acc[1][i] += (double)d.v[i] * (double)xh;
            }
        }
    }
};

// Fast path of pool_bn_bwd_reduce (4 channels per thread): the scatter-form sums need two DEPENDENT round trips per pooled
// element (argmax, then the gathered pre-pool value); the generic row loop paid them row by row (16 rows per thread).  Here
// RU rows go through the two phases together: all pooled-gradient / argmax loads, then all 4*RU gathers, then the sums.
// POOLED (ps.pa given): the gathered pre-pool value is only needed for (a) the ReLU6 mask and (b) xhat -- and both
// follow from the pooled activated output a = relu6(scale y + shift) the forward stored: 0 < a < 6 is the same decision as
// 0 < scale y + shift < 6 (a IS that value, clamped), and where it holds y = (a - shift) / scale.  One dense 16-byte load per 4
// channels instead of 4 scattered 4-byte gathers out of a tensor 4x the size (the kernel sits in the exposed tail of every pass:
// 150 -> 60 us at B = 256).  Channels whose |scale| is below 0.05 (gamma ~ 0: the division would amplify the rounding of a) take
// the gather path, thread by thread.
template <class T, bool POOLED>
__global__ void __launch_bounds__(256) pool_bn_bwd_reduce_v4_kernel(PoolSrc ps, const T* __restrict__ y,
                                                                    const float* __restrict__ stats, int GC, int C, int Mg,
                                                                    int rb, double* __restrict__ part) {
    extern __shared__ double sm[];   // [CY][4][CX]
    constexpr int RU = 4, VEC = 4;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y, nb = gridDim.x;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    const int c0 = tx * VEC;
    const bool on = c0 < C;
    double acc[2][VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[0][i] = acc[1][i] = 0.0;
    if (on) {
        const VecF<VEC> mean = vload<VEC>(stats + 0 * GC + g * C + c0), invstd = vload<VEC>(stats + 1 * GC + g * C + c0);
        const VecF<VEC> sc = vload<VEC>(stats + 2 * GC + g * C + c0), sh = vload<VEC>(stats + 3 * GC + g * C + c0);
        const int64_t gbase = (int64_t)g * Mg;
        bool pooled = POOLED;
        if (POOLED) {
            // v = (a - shift) / scale recovers the pre-BatchNorm value from the pooled activation with an absolute error of
            // eps (|a| + |shift|) / |scale|, i.e. eps (|a| + |shift|) / |gamma| in xhat: the shortcut is taken only where that stays
            // near 3e-6 (|shift| <= 8, |gamma| >= 0.25, |scale| >= 0.05); other channels use the gather form below
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                pooled = pooled && fabsf(sc.v[i]) >= 0.05f && fabsf(sh.v[i]) <= 8.0f && fabsf(sc.v[i]) >= 0.25f * fabsf(invstd.v[i]);
        }
        if (POOLED && pooled) {
            float isc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) isc[i] = 1.0f / sc.v[i];
            for (int rr = r0 + ty; rr < r1; rr += CY * RU) {
                VecF<VEC> d[RU], a[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int64_t row = gbase + min(rr + u * CY, r1 - 1);          // clamped: unconditional loads
                    d[u] = vload<VEC>(reinterpret_cast<const T*>(ps.dp) + row * C + c0);
                    a[u] = vload<VEC>(reinterpret_cast<const T*>(ps.pa) + row * C + c0);
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    if (rr + u * CY >= r1) break;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        if (relu6_open(a[u].v[i])) {
                            const float v = (a[u].v[i] - sh.v[i]) * isc[i];
                            const float xh = (v - mean.v[i]) * invstd.v[i];
                            acc[0][i] += (double)d[u].v[i];
                            acc[1][i] += (double)d[u].v[i] * (double)xh;
                        } else if (sizeof(T) == 2 && a[u].v[i] == 6.0f) {
                            // bf16 storage: an activation in (6 - half a bf16 ulp, 6) was STORED as 6.0, but its ReLU6 is open and the
                            // apply side (stem_bwd_filter_fused) takes the mask from the raw conv output -- decide from that here too
                            // (rare: one gather for the element)
                            const int64_t row = gbase + rr + u * CY;
                            const int k = (int)ps.argmax[row * C + c0 + i] & 0x7f;
                            const int ox = (int)(row % ps.Wo);
                            const int64_t q = row / ps.Wo;
                            const int oy = (int)(q % ps.Ho);
                            const int64_t n = q / ps.Ho;
                            const int ky = k / 3, kx = k - 3 * ky;
                            const float yv = ldf(y + ((n * ps.H + (2 * oy - ps.pt + ky)) * ps.W + (2 * ox - ps.pl + kx)) * C + c0 + i);
                            if (relu6_open(fmaf(sc.v[i], yv, sh.v[i]))) {
                                const float xh = (yv - mean.v[i]) * invstd.v[i];
                                acc[0][i] += (double)d[u].v[i];
                                acc[1][i] += (double)d[u].v[i] * (double)xh;
                            }
                        }
                    }
                }
            }
        } else
        for (int rr = r0 + ty; rr < r1; rr += CY * RU) {
            VecF<VEC> d[RU];
            uint32_t am[RU];
            int64_t ybase[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t row = gbase + min(rr + u * CY, r1 - 1);          // clamped: unconditional loads
                d[u] = vload<VEC>(reinterpret_cast<const T*>(ps.dp) + row * C + c0);
                am[u] = *reinterpret_cast<const uint32_t*>(ps.argmax + row * C + c0);
                const int ox = (int)(row % ps.Wo);
                const int64_t q = row / ps.Wo;
                const int oy = (int)(q % ps.Ho);
                const int64_t n = q / ps.Ho;
                // element offset of the window's top-left input pixel (may lie in the padding: only used with the tap offset)
                ybase[u] = ((n * ps.H + (2 * oy - ps.pt)) * ps.W + (2 * ox - ps.pl)) * C + c0;
            }
            float v[RU][VEC];
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int k = (int)((am[u] >> (8 * i)) & 0x7fu);
                    const int ky = k / 3, kx = k - 3 * ky;
                    v[u][i] = ldf(y + ybase[u] + ((int64_t)ky * ps.W + kx) * C + i);
                }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (rr + u * CY >= r1) break;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float z = fmaf(sc.v[i], v[u][i], sh.v[i]);
                    if (relu6_open(z)) {
                        const float xh = (v[u][i] - mean.v[i]) * invstd.v[i];
                        acc[0][i] += (double)d[u].v[i];
                        acc[1][i] += (double)d[u].v[i] * (double)xh;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (CY > 1) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) sm[(ty * VEC + i) * CX + tx] = acc[q][i];
            __syncthreads();
            if (ty == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    double s = acc[q][i];
                    for (int yy = 1; yy < CY; ++yy) s += sm[(yy * VEC + i) * CX + tx];
                    acc[q][i] = s;
                }
            }
            __syncthreads();
        }
    }
    if (ty == 0 && on) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < VEC; ++i) part[(((int64_t)g * nb + blockIdx.x) * 2 + q) * C + c0 + i] = acc[q][i];
    }
}

int pool_bn_bwd_reduce(const PoolSrc& ps, const float* y, int G, int frames_per_group, int C, const float* stats, double* part,
                       hipStream_t st, int at) {
    {
        static const bool fast = true;
        const int Mg = frames_per_group * ps.Ho * ps.Wo;
        const VColGeom g = vcol_geom(Mg, C, NB_STATS);
        if (fast && g.vec == 4 && g.nloop == 1) {
            dim3 grid(g.nb, G), block(g.cx, g.cy);
            const size_t smb = (size_t)g.cy * 4 * g.cx * sizeof(double);
            static const bool pooled_env = true;
            if (at && ps.pa && pooled_env) hipLaunchKernelGGL((pool_bn_bwd_reduce_v4_kernel<bf16_t, true>), grid, block, smb, st, ps, reinterpret_cast<const bf16_t*>(y), stats, G * C, C, Mg, g.rb, part);
            else if (at) hipLaunchKernelGGL((pool_bn_bwd_reduce_v4_kernel<bf16_t, false>), grid, block, smb, st, ps, reinterpret_cast<const bf16_t*>(y), stats, G * C, C, Mg, g.rb, part);
            else if (ps.pa && pooled_env) hipLaunchKernelGGL((pool_bn_bwd_reduce_v4_kernel<float, true>), grid, block, smb, st, ps, y, stats, G * C, C, Mg, g.rb, part);
            else hipLaunchKernelGGL((pool_bn_bwd_reduce_v4_kernel<float, false>), grid, block, smb, st, ps, y, stats, G * C, C, Mg, g.rb, part);
            CDRL_LAUNCH_CHECK();
            return 0;
        }
    }
    if (at) return launch_vcolreduce_t<2, PoolBnReduceF, bf16_t>(G, frames_per_group * ps.Ho * ps.Wo, C, part, st, NB_STATS, ps, y, stats, G * C, C);
    return launch_vcolreduce_t<2, PoolBnReduceF, float>(G, frames_per_group * ps.Ho * ps.Wo, C, part, st, NB_STATS, ps, y, stats, G * C, C);
}

template <int FIN_PY>
__global__ void __launch_bounds__(FIN_CX * FIN_PY) bn_bwd_finalize_kernel(const double* __restrict__ part, int nb, int G, int Mg,
                                                              int C, const float* __restrict__ stats,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    __shared__ double sm[2][FIN_PY][FIN_CX];
    __shared__ double red[8][2][FIN_CX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c = blockIdx.x * FIN_CX + tx;
    const bool ok = c < C;
    const int GC = G * C;
    double dg = 0.0, db = 0.0;
    const double n = (double)Mg;
    const int Gp = fin_group_slots(G);
    for (int g0 = 0; g0 < G; g0 += Gp) {
        // k1 = gamma * invstd of this round's groups: issued before the reduction so that the loads overlap it (they were a
        // serial chain of dependent global loads in the single-thread tail: 11 us vs 6 us for the forward finalize)
        float k1v[8];
        if (ok && ty == 0) {
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) k1v[gg] = (gg < Gp && g0 + gg < G) ? stats[2 * GC + (g0 + gg) * C + c] : 0.0f;
        }
        fin_reduce<FIN_PY>(part, nb, G, C, c, ok, g0, Gp, sm, red);
        if (ok && ty == 0) {
            const int ng = min(Gp, G - g0);
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
                if (gg >= ng) break;
                const int g = g0 + gg;
                const double s = red[gg][0][tx], q = red[gg][1][tx];
                db += s;
                dg += q;
                coef[0 * GC + g * C + c] = k1v[gg];                     // k1 = gamma * invstd
                coef[1 * GC + g * C + c] = (float)(s / n);              // k2 = mean(dz)
                coef[2 * GC + g * C + c] = (float)(q / n);              // k3 = mean(dz * xhat)
            }
        }
        __syncthreads();
    }
    if (ok && ty == 0) {
        dgamma[c] = (float)dg;      // shared gamma/beta: summed over the T applications (Appendix E)
        dbeta[c] = (float)db;
    }
}

int bn_bwd_finalize(const double* part, int nb, int G, int Mg, int C, const float* stats, float* dgamma,
                    float* dbeta, float* coef, hipStream_t st) {
    const int py = fin_py();
    if (py == 128) hipLaunchKernelGGL(bn_bwd_finalize_kernel<128>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 128), 0, st, part, nb, G, Mg, C, stats, dgamma, dbeta, coef);
    else if (py == 64) hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 64), 0, st, part, nb, G, Mg, C, stats, dgamma, dbeta, coef);
    else hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, dim3(cdiv(C, FIN_CX)), dim3(FIN_CX, 32), 0, st, part, nb, G, Mg, C, stats, dgamma, dbeta, coef);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int VEC, class T = float>
struct BnBwdApplyF {
    View da;
    int shuffle_ctot;
    View y;
    const float* stats;
    const float* coef;
    int GC, C, act;
    float* dy;
    bool al_da, al_y;
    bool use_pool;
    PoolSrc pool;
    int bcast;              // as in BnBwdReduceF
    __device__ void operator()(int g, int64_t row, int c0, double (*acc)[VEC]) const {
        VecF<VEC> d;
        if (bcast > 0) {        // float32 pooled gradient (see BnBwdReduceF)
            d = vload_view<VEC, float>(da, row / bcast, c0, 0, al_da);
#pragma unroll
            for (int i = 0; i < VEC; ++i) d.v[i] = d.v[i] / (float)bcast;
        } else {
            d = use_pool ? pool_gather<VEC, T>(pool, row, c0, C) : vload_view<VEC, T>(da, row, c0, shuffle_ctot, al_da);
        }
        const VecF<VEC> v = vload_view<VEC, T>(y, row, c0, 0, al_y);
        const VecF<VEC> mean = vload<VEC>(stats + 0 * GC + g * C + c0), invstd = vload<VEC>(stats + 1 * GC + g * C + c0);
        if (act == ACT_RELU6) {
            const VecF<VEC> sc = vload<VEC>(stats + 2 * GC + g * C + c0), sh = vload<VEC>(stats + 3 * GC + g * C + c0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = fmaf(sc.v[i], v.v[i], sh.v[i]);
                if (!relu6_open(z)) d.v[i] = 0.0f;
            }
        }
        const VecF<VEC> k1 = vload<VEC>(coef + 0 * GC + g * C + c0), k2 = vload<VEC>(coef + 1 * GC + g * C + c0),
                        k3 = vload<VEC>(coef + 2 * GC + g * C + c0);
        VecF<VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float xh = (v.v[i] - mean.v[i]) * invstd.v[i];
            o.v[i] = k1.v[i] * (d.v[i] - k2.v[i] - xh * k3.v[i]);
            acc[0][i] += (double)o.v[i];
        }
        vstore<VEC>(reinterpret_cast<T*>(dy) + row * C + c0, o);
    }
};

// Fast path of bn_bwd_apply (no pool source): RU rows of loads ahead of the stores, columns / coefficients once per thread;
// same partial layout ([G][nb][C] column sums of dy) as the generic skeleton.
template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_bwd_apply_fast_kernel(View da, int ctot, const T* __restrict__ y,
                                                                const float* __restrict__ stats, const float* __restrict__ coef,
                                                                int GC, int C, int Mg, int rb, int act, T* __restrict__ dy,
                                                                double* __restrict__ part, int bcast) {
    extern __shared__ double sm[];   // [CY][VEC][CX]
    constexpr int RU = 4;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int g = blockIdx.y, nb = gridDim.x;
    const int r0 = blockIdx.x * rb, r1 = min(r0 + rb, Mg);
    const int c0 = tx * VEC;
    const bool on = c0 < C;
    double acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0;
    if (on) {
        int dcol[VEC];
        float mean[VEC], inv[VEC], sc[VEC], sh[VEC], k1[VEC], k2[VEC], k3[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int cc = da.coff + c0 + i;
            dcol[i] = ctot ? shuffle_dst(cc, ctot) : cc;
            const int o = g * C + c0 + i;
            mean[i] = stats[0 * GC + o];
            inv[i] = stats[1 * GC + o];
            sc[i] = stats[2 * GC + o];
            sh[i] = stats[3 * GC + o];
            k1[i] = coef[0 * GC + o];
            k2[i] = coef[1 * GC + o];
            k3[i] = coef[2 * GC + o];
        }
        const bool relu = act == ACT_RELU6;
        const int64_t gbase = (int64_t)g * Mg;
        for (int rr = r0 + ty; rr < r1; rr += CY * RU) {
            float dz[RU][VEC];
            VecF<VEC> yv[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t row = gbase + min(rr + u * CY, r1 - 1);
                if (bcast > 0) {        // float32 pooled gradient
                    const float* dr = da.p + (row / bcast) * da.ld;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) dz[u][i] = dr[dcol[i]];
                } else {
                    const T* dr = vptr<T>(da) + row * da.ld;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) dz[u][i] = ldf(dr + dcol[i]);
                }
                yv[u] = vload<VEC>(y + row * C + c0);
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (rr + u * CY >= r1) break;
                const int64_t row = gbase + rr + u * CY;
                VecF<VEC> o;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float v = yv[u].v[i];
                    float d = dz[u][i];
                    if (bcast > 0) d = d / (float)bcast;        // gradient of the global average pool over `bcast` rows
                    if (relu) {
                        const float z = fmaf(sc[i], v, sh[i]);
                        d = relu6_open(z) ? d : 0.0f;
                    }
                    const float xh = (v - mean[i]) * inv[i];
                    o.v[i] = k1[i] * (d - k2[i] - xh * k3[i]);
                    acc[i] += (double)o.v[i];
                }
                vstore<VEC>(dy + row * C + c0, o);
            }
        }
    }
    if (CY > 1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) sm[(ty * VEC + i) * CX + tx] = acc[i];
        __syncthreads();
        if (ty == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                double s = acc[i];
                for (int yy = 1; yy < CY; ++yy) s += sm[(yy * VEC + i) * CX + tx];
                acc[i] = s;
            }
        }
    }
    if (ty == 0 && on) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[((int64_t)g * nb + blockIdx.x) * C + c0 + i] = acc[i];
    }
}

int bn_bwd_apply(View da, int shuffle_ctot, View y, int G, int Mg, int C, const float* stats, const float* coef,
                 int act, float* dy, double* part2, hipStream_t st, const PoolSrc* pool, int bcast_rows, int at) {
    {
        static const bool fast = true;
        const VColGeom g = vcol_geom(Mg, C, NB_STATS);
        const bool ydense = y.ld == C && y.coff == 0 && view_aligned(y, g.vec);
        if (fast && !pool && g.nloop == 1 && g.vec >= 2 && ydense && (reinterpret_cast<uintptr_t>(dy) % (4 * g.vec)) == 0) {
            dim3 grid(g.nb, G), block(g.cx, g.cy);
            const size_t sm = (size_t)g.cy * g.vec * g.cx * sizeof(double);
            const bf16_t* yb = reinterpret_cast<const bf16_t*>(y.p);
            bf16_t* dyb = reinterpret_cast<bf16_t*>(dy);
            if (at && g.vec == 4)
                hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<4, bf16_t>), grid, block, sm, st, da, shuffle_ctot, yb, stats, coef, G * C, C, Mg, g.rb, act, dyb, part2, bcast_rows);
            else if (at)
                hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<2, bf16_t>), grid, block, sm, st, da, shuffle_ctot, yb, stats, coef, G * C, C, Mg, g.rb, act, dyb, part2, bcast_rows);
            else if (g.vec == 4)
                hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<4, float>), grid, block, sm, st, da, shuffle_ctot, y.p, stats, coef, G * C, C, Mg, g.rb, act, dy, part2, bcast_rows);
            else
                hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<2, float>), grid, block, sm, st, da, shuffle_ctot, y.p, stats, coef, G * C, C, Mg, g.rb, act, dy, part2, bcast_rows);
            CDRL_LAUNCH_CHECK();
            return 0;
        }
    }
    const int vec = vcol_geom(Mg, C).vec;
    PoolSrc ps{};
    if (pool) ps = *pool;
    if ((reinterpret_cast<uintptr_t>(dy) % (4 * vec)) != 0) {
        set_error("bn_bwd_apply: dy must be %d-byte aligned", 4 * vec);
        return -1;
    }
    if (at)
        return launch_vcolreduce_t<1, BnBwdApplyF, bf16_t>(G, Mg, C, part2, st, NB_STATS, da, shuffle_ctot, y, stats, coef, G * C, C, act,
                                                           dy, pool ? false : view_aligned(da, vec), view_aligned(y, vec), pool != nullptr, ps, bcast_rows);
    return launch_vcolreduce_t<1, BnBwdApplyF, float>(G, Mg, C, part2, st, NB_STATS, da, shuffle_ctot, y, stats, coef, G * C, C, act,
                                                      dy, pool ? false : view_aligned(da, vec), view_aligned(y, vec), pool != nullptr, ps, bcast_rows);
}

// Block = (CX outputs, 1024/CX partial lanes).  CX = 16 gives 128-byte row segments; CX = 4 is used when there are few
// outputs (n/16 workgroups would leave most of the 256 CUs idle while each workgroup walks hundreds of KB alone: the
// 1392-output / 2048-partial reductions of the first unit took ~300 us that way).
template <int CX>
__global__ void __launch_bounds__(256) reduce_partials_kernel(const double* __restrict__ part, int nparts, int n,
                                                             int64_t stride, float* __restrict__ out, int accumulate,
                                                             int n1, float* __restrict__ out2) {
    // columns [0, n1) go to out, [n1, n) to out2 (two parameter tensors reduced by one launch, e.g. filter + bias)
    // 256-thread blocks: these reductions run on the side stream next to the main stream's kernels, and a 1024-thread
    // block needs 16 free wave slots on ONE CU before it can start (measured 21 us per launch in the step, 5 us alone).
    constexpr int PY = 256 / CX;
    __shared__ double sm[PY][CX];
    const int tx = threadIdx.x % CX, ty = threadIdx.x / CX;
    const int i = blockIdx.x * CX + tx;
    double s = 0.0;
    if (i < n) {
        const double* src = part + i;
        for (int p0 = ty; p0 < nparts; p0 += PY * FIN_U) {      // FIN_U independent loads in flight (clamped addresses)
            double v[FIN_U];
#pragma unroll
            for (int u = 0; u < FIN_U; ++u) {
                const int p = p0 + u * PY;
                v[u] = src[(int64_t)min(p, nparts - 1) * stride];
                if (p >= nparts) v[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < FIN_U; ++u) s += v[u];
        }
    }
    sm[ty][tx] = s;
    __syncthreads();
    // fold the PY lane sums in a fixed order
    if (ty == 0 && i < n) {
        s = 0.0;
#pragma unroll 8
        for (int y = 0; y < PY; ++y) s += sm[y][tx];
        float* o = i < n1 ? &out[i] : &out2[i - n1];
        *o = accumulate ? *o + (float)s : (float)s;
    }
}

int reduce_partials2(const double* part, int nparts, int n1, int n2, int64_t stride, float* out1, float* out2, int accumulate,
                     hipStream_t st) {
    const int n = n1 + n2;
    if (cdiv(n, 16) >= 128 || nparts <= 64)
        hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3(cdiv(n, 16)), dim3(256), 0, st, part, nparts, n, stride, out1, accumulate, n1, out2);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<4>, dim3(cdiv(n, 4)), dim3(256), 0, st, part, nparts, n, stride, out1, accumulate, n1, out2);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int reduce_partials(const double* part, int nparts, int n, int64_t stride, float* out, int accumulate,
                    hipStream_t st) {
    return reduce_partials2(part, nparts, n, 0, stride, out, nullptr, accumulate, st);
}

__global__ void __launch_bounds__(256) gather_view_kernel(View src, int shuffle_ctot, int rows, int C, int rb,
                                                          View dst, int accumulate) {
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int r0 = blockIdx.x * rb;
    const int r1 = min(r0 + rb, rows);
    for (int c = tx; c < C; c += CX) {
        int sc = src.coff + c;
        if (shuffle_ctot) sc = shuffle_dst(sc, shuffle_ctot);
        for (int r = r0 + ty; r < r1; r += CY) {
            const float v = src.p[(int64_t)r * src.ld + sc];
            float* d = &dst.p[(int64_t)r * dst.ld + dst.coff + c];
            *d = accumulate ? *d + v : v;
        }
    }
}

int gather_view(View src, int shuffle_ctot, int rows, int C, View dst, int accumulate, hipStream_t st) {
    ColGeom g = col_geom(rows, C, 4096);
    hipLaunchKernelGGL(gather_view_kernel, dim3(g.nb), dim3(g.cx, g.cy), 0, st, src, shuffle_ctot, rows, C, g.rb, dst,
                       accumulate);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// dense activations (relu6 for feature nets, swish6 for the control branches)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void act_fwd_kernel(const float* __restrict__ z, float* __restrict__ a, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = z[i];
        float o = x;
        if (act == ACT_RELU6) o = fminf(fmaxf(x, 0.0f), 6.0f);
        else if (act == ACT_SWISH6) o = fminf(x * sigmoidf_(x), 6.0f);      // rl/utils.py:420-421
        else if (act == ACT_TANH) o = tanhf(x);
        else if (act == ACT_SIGMOID) o = sigmoidf_(x);
        a[i] = o;
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ da, float* __restrict__ dz,
                               int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = z[i];
        float d = da[i];
        if (act == ACT_RELU6) {
            if (!relu6_open(x)) d = 0.0f;
        } else if (act == ACT_SWISH6) {
            const float s = sigmoidf_(x);
            d = (x * s < 6.0f) ? d * (s * (1.0f + x * (1.0f - s))) : 0.0f;
        } else if (act == ACT_TANH) {
            const float t = tanhf(x);
            d *= (1.0f - t * t);
        } else if (act == ACT_SIGMOID) {
            const float s = sigmoidf_(x);
            d *= s * (1.0f - s);
        }
        dz[i] = d;
    }
}

static inline int flat_grid(int64_t n) {
    int64_t b = cdiv64(n, 256);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

int act_fwd(const float* z, float* a, int64_t n, int act, hipStream_t st) {
    hipLaunchKernelGGL(act_fwd_kernel, dim3(flat_grid(n)), dim3(256), 0, st, z, a, n, act);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int act_bwd(const float* z, const float* da, float* dz, int64_t n, int act, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(flat_grid(n)), dim3(256), 0, st, z, da, dz, n, act);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

int fill(float* p, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(flat_grid(n)), dim3(256), 0, st, p, n, v);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// (B,T,D) -> rows t*B+b
__global__ void permute_bt_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int D,
                                  int inverse) {
    const int64_t n = (int64_t)B * T * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int64_t bt = i / D;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int64_t j = ((int64_t)t * B + b) * D + d;
        if (inverse) dst[i] = src[j];
        else dst[j] = src[i];
    }
}

int permute_bt(const float* src, float* dst, int B, int T, int D, hipStream_t st) {
    hipLaunchKernelGGL(permute_bt_kernel, dim3(flat_grid((int64_t)B * T * D)), dim3(256), 0, st, src, dst, B, T, D, 0);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int permute_tb_bwd(const float* src, float* dst, int B, int T, int D, hipStream_t st) {
    hipLaunchKernelGGL(permute_bt_kernel, dim3(flat_grid((int64_t)B * T * D)), dim3(256), 0, st, src, dst, B, T, D, 1);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// minibatch row gather (K16): dst[i, :] = src[idx[i], :]; one workgroup column per row, 16-byte
// vector copies when the row length allows (the 4x90x120x3 observation rows are 518 KB each).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t row_elems, int vec4) {
    const int64_t row = blockIdx.y;
    const int64_t s = (int64_t)idx[row] * row_elems, d = row * row_elems;
    if (vec4) {
        const float4* sp = reinterpret_cast<const float4*>(src + s);
        float4* dp = reinterpret_cast<float4*>(dst + d);
        const int64_t n4 = row_elems >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
            dp[i] = sp[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_elems; i += (int64_t)gridDim.x * blockDim.x)
            dst[d + i] = src[s + i];
    }
}

int gather_rows(const float* src, const int* idx, float* dst, int nrows, int64_t row_elems, hipStream_t st) {
    if (nrows <= 0 || row_elems <= 0) return 0;
    const int vec4 = (row_elems % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    int64_t bx = cdiv64(vec4 ? row_elems / 4 : row_elems, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)bx, nrows), dim3(256), 0, st, src, idx, dst, row_elems, vec4);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
