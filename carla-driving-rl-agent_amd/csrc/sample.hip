// On-device Beta sampling with pathwise (implicit-reparameterisation) derivatives, gfx950.
//
// The reference's PolicyNetwork.call re-samples an action from the NEW Beta(alpha, beta) inside the
// loss and lets the gradient flow through the sample (reference core/networks.py:96-110,
// 133-137).  TFP draws Beta(a, b) as g1 / (g1 + g2) with g1 ~ Gamma(a), g2 ~ Gamma(b) and differentiates
// each Gamma sample implicitly (Figurnov et al. 2018):  dg/da = -(dP/da)(a, g) / p(g; a)  where P is the
// regularised lower incomplete gamma function and p the Gamma density.  This kernel does the same:
//   * Philox-4x32-10 counter RNG (seed, offset, element index) -> reproducible, stateless;
//   * Marsaglia-Tsang rejection sampler (alpha, beta >= 1.01 on this path, so a >= 1 always);
//   * dP/da from the term-by-term derivative of the series
//         P(a, x) = e^{-x} sum_n x^{a+n} / Gamma(a+n+1)
//     => dP/da    = e^{-x} sum_n x^{a+n} / Gamma(a+n+1) * (ln x - psi(a+n+1)),
//     evaluated in double (terms by recurrence, psi by recurrence);
//   * u = g1/(g1+g2), du/dalpha = dg1/da * g2/(g1+g2)^2, du/dbeta = -dg2/db * g1/(g1+g2)^2.
// B*A is a few hundred elements: one thread per element, latency-bound.
#include "cdrl_kernels.h"
#include "philox.h"

namespace cdrl {

__device__ double digamma_s(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x +
           f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0)))));
}

// Marsaglia-Tsang, a >= 1
__device__ double gamma_sample(double a, Philox& rng) {
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 64; ++it) {
        const double x = rng.normal();
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = rng.uniform();
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
    return d;       // (probability ~1e-60) fall back to the mode-ish value
}

// dg/da at fixed Gamma(a) quantile: -(dP/da)(a, g) / p(g; a)
__device__ double gamma_grad(double a, double g) {
    const double lx = log(g);
    // term_n = g^{a+n} e^{-g} / Gamma(a+n+1), n = 0..; ratio term_{n+1}/term_n = g / (a+n+1)
    double term = exp(a * lx - g - lgamma(a + 1.0));
    double psi = digamma_s(a + 1.0);
    double dP = 0.0;
    for (int n = 0; n < 2000; ++n) {
        dP += term * (lx - psi);
        const double an1 = a + n + 1.0;
        term *= g / an1;
        psi += 1.0 / an1;
        if (an1 > g && term * (fabs(lx - psi) + 1.0) < 1e-17 * (fabs(dP) + 1e-300)) break;
    }
    const double pdf = exp((a - 1.0) * lx - g - lgamma(a));
    return -dP / pdf;
}

// element i = (row, col): alpha[row*ld + col], beta[row*ld + col]; outputs are dense [rows][A]
__global__ void beta_sample_kernel(const float* __restrict__ alpha, const float* __restrict__ beta, int n, int A, int ld,
                                   uint64_t seed, uint64_t offset, float* __restrict__ u, float* __restrict__ du_da,
                                   float* __restrict__ du_db, float* __restrict__ logp, double* __restrict__ gammas) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = i / A, col = i - row * A;
    const double a = (double)alpha[(int64_t)row * ld + col], b = (double)beta[(int64_t)row * ld + col];
    Philox rng(seed, offset, (uint64_t)i);
    const double g1 = gamma_sample(a, rng), g2 = gamma_sample(b, rng);
    const double s = g1 + g2;
    if (u) u[i] = (float)(g1 / s);
    if (gammas) {       // test hook: the two Gamma draws behind u
        gammas[2 * (int64_t)i] = g1;
        gammas[2 * (int64_t)i + 1] = g2;
    }
    if (du_da) du_da[i] = (float)(gamma_grad(a, g1) * g2 / (s * s));
    if (du_db) du_db[i] = (float)(-gamma_grad(b, g2) * g1 / (s * s));
    if (logp) {     // log-density of the CLIPPED sample, as PolicyNetwork.call evaluates it (core/networks.py:100-103,139-144)
        const float eps = 1.1920929e-07f;
        const double x = (double)fminf(fmaxf((float)(g1 / s), eps), 1.0f - eps);
        logp[i] = (float)((a - 1.0) * log(x) + (b - 1.0) * log1p(-x) - (lgamma(a) + lgamma(b) - lgamma(a + b)));
    }
}

int beta_sample(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset, float* u,
                float* du_da, float* du_db, hipStream_t st, float* logp, double* gammas) {
    const int n = rows * A;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(beta_sample_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, alpha, beta, n, A, ld, seed, offset, u, du_da,
                       du_db, logp, gammas);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// test hook: the first `nblocks` raw 128-bit blocks of the streams (seed, offset, idx0 + i), i < n -> out[i][nblocks][4]
__global__ void philox_words_kernel(uint64_t seed, uint64_t offset, uint64_t idx0, int n, int nblocks, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Philox rng(seed, offset, idx0 + (uint64_t)i);
    for (int b = 0; b < nblocks; ++b) {
        rng.refill();
        for (int w = 0; w < 4; ++w) out[((int64_t)i * nblocks + b) * 4 + w] = rng.out[w];
    }
}

int philox_words(uint64_t seed, uint64_t offset, uint64_t idx0, int n, int nblocks, uint32_t* out, hipStream_t st) {
    // stream contract (philox.h): element indices below 2^48 (the top 16 bits of the index word count the element's blocks) and at
    // most 2^16 blocks per element -- beyond either an element would silently alias another one's words
    if (n < 0 || idx0 >= PHILOX_MAX_ELEMENTS || (uint64_t)n > PHILOX_MAX_ELEMENTS - idx0 || nblocks < 0 || nblocks > PHILOX_MAX_BLOCKS) {
        set_error("philox_words: element range [%llu, +%d) or block count %d outside the stream contract (2^48 elements, 2^16 blocks)",
                  (unsigned long long)idx0, n, nblocks);
        return -1;
    }
    if (n <= 0 || nblocks <= 0) return 0;
    hipLaunchKernelGGL(philox_words_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, seed, offset, idx0, n, nblocks, out);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// test hook: implicit gamma derivative for given (a, g)
__global__ void gamma_grad_kernel(const double* __restrict__ a, const double* __restrict__ g, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gamma_grad(a[i], g[i]);
}

int gamma_implicit_grad(const double* a, const double* g, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(gamma_grad_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, a, g, n, out);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
