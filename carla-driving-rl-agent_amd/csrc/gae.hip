// Returns + GAE(lambda) + sp-norm for one env shard (gfx950).
//
// Reference: PPOMemory.compute_returns / compute_advantages (rl/agents/ppo.py:699-727),
// utils.gae / discount_cumsum / decompose_number / tf_sp_norm (rl/utils.py:57-84,140-151,
// 344-349).  discount_cumsum is scipy.signal.lfilter([1],[1,-d]) on the reversed sequence,
// which promotes float32 input to float64: y[n] = x[n] + d*y[n+1] with one rounding per
// multiply and per add (no FMA) -- reproduced here with __dmul_rn/__dadd_rn so the scan is
// bit-exact against scipy.  Element-wise float32 steps use the _rn intrinsics as well so that
// hipcc cannot contract them into FMAs (TF evaluates them as separate float32 ops).
// One workgroup per shard: the recurrences are latency-bound chains (staged through LDS, the two of them on two waves), everything
// else is a wavefront-parallel map / reduction.
#include "cdrl_kernels.h"

namespace cdrl {

// y[j] = x[j] + d * y[j + 1] over a[0 .. n) in place, from the top down, `acc` = y[n]; eight independent LDS reads are issued ahead of
// each group of eight dependent (multiply, add) steps -- left to the compiler, every step waited for its own read.
__device__ __forceinline__ double gae_scan(double* a, int n, double d, double acc) {
    int j = n - 1;
    for (; j >= 7; j -= 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = a[j - u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __dadd_rn(x[u], __dmul_rn(d, acc));
            x[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[j - u] = x[u];
    }
    for (; j >= 0; --j) {
        acc = __dadd_rn(a[j], __dmul_rn(d, acc));
        a[j] = acc;
    }
    return acc;
}

__global__ void __launch_bounds__(256) gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values_be,
                                                  int N, double gamma, double lambda, float scale,
                                                  float* __restrict__ returns, float* __restrict__ returns_be,
                                                  float* __restrict__ adv_raw, float* __restrict__ adv,
                                                  double* __restrict__ scratch) {
    __shared__ float smax[256], smin[256];
    const int tid = threadIdx.x;
    double* delta = scratch;             // [N]
    double* ret = scratch + (N + 1);     // [N+1]
    const float g32 = (float)gamma;
    // values = base * 10^exp (float32), deltas (float32, three separately rounded ops)
    for (int i = tid; i < N; i += 256) {
        const float v0 = __fmul_rn(values_be[2 * i], (float)pow(10.0, (double)values_be[2 * i + 1]));
        const float v1 = __fmul_rn(values_be[2 * i + 2], (float)pow(10.0, (double)values_be[2 * i + 3]));
        const float d = __fsub_rn(__fadd_rn(rewards[i], __fmul_rn(g32, v1)), v0);
        delta[i] = (double)d;
    }
    __syncthreads();
    // The two recurrences (advantages over N deltas, returns over N + 1 rewards) are sequential by definition -- the result must be
    // bit-identical to scipy's lfilter -- but nothing forces them to walk global memory: read straight from the scratch arrays,
    // every step was a dependent L2 round trip (171 ns per time step: 11.2 ms for a 65536-step buffer).  Chunks of GAE_CH steps are
    // staged in LDS by the whole workgroup (coalesced), lane 0 of wave 0 scans the deltas while lane 0 of wave 1 scans the rewards,
    // and the chunk goes back coalesced: the chains run at LDS latency, side by side.
    {
        constexpr int GAE_CH = 2048;
        __shared__ double sd[GAE_CH], sr[GAE_CH];
        const double dl = __dmul_rn(gamma, lambda);
        double accd = 0.0, accr = 0.0;              // carries (live in tid 0 / tid 64 only)
        for (int hi = N + 1; hi > 0; hi -= GAE_CH) {      // chunk = indices [lo, hi) of the (N + 1)-long reward sequence
            const int lo = hi > GAE_CH ? hi - GAE_CH : 0;
            for (int j = tid; j < hi - lo; j += 256) {
                sr[j] = (double)rewards[lo + j];
                sd[j] = (lo + j < N) ? delta[lo + j] : 0.0;
            }
            __syncthreads();
            if (tid == 0 && lambda != 0.0) accd = gae_scan(sd, (hi <= N ? hi : N) - lo, dl, accd);     // utils.gae: lambda == 0 -> advantages = deltas
            if (tid == 64) accr = gae_scan(sr, hi - lo, gamma, accr);
            __syncthreads();
            for (int j = tid; j < hi - lo; j += 256) {
                ret[lo + j] = sr[j];
                if (lo + j < N) delta[lo + j] = sd[j];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    float mx = -INFINITY, mn = INFINITY;
    for (int i = tid; i < N; i += 256) {
        const float a = (float)delta[i];
        adv_raw[i] = a;
        mx = fmaxf(mx, a);
        mn = fminf(mn, a);
        // decompose_number: while |x| > 1: x /= 10 (float32 division)
        float x = (float)ret[i];
        returns[i] = x;
        int e = 0;
        while (fabsf(x) > 1.0f) {
            x = __fdiv_rn(x, 10.0f);
            ++e;
        }
        returns_be[2 * i] = x;
        returns_be[2 * i + 1] = (float)e;
    }
    smax[tid] = mx;
    smin[tid] = mn;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) {
            smax[tid] = fmaxf(smax[tid], smax[tid + k]);
            smin[tid] = fminf(smin[tid], smin[tid + k]);
        }
        __syncthreads();
    }
    // tf_sp_norm: positives / (max + eps) + negatives / -(min - eps), then * scale
    const float pden = __fadd_rn(smax[0], 1e-3f);
    const float nden = -__fsub_rn(smin[0], 1e-3f);
    for (int i = tid; i < N; i += 256) {
        const float a = adv_raw[i];
        const float pos = a > 0.0f ? a : __fmul_rn(a, 0.0f);
        const float neg = a < 0.0f ? a : __fmul_rn(a, 0.0f);
        adv[i] = __fmul_rn(__fadd_rn(__fdiv_rn(pos, pden), __fdiv_rn(neg, nden)), scale);
    }
}

int gae_returns(const float* rewards, const float* values_be, int N, double gamma, double lambda, float scale,
                float* returns, float* returns_be, float* adv_raw, float* adv, double* scratch, hipStream_t st) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(gae_kernel, dim3(1), dim3(256), 0, st, rewards, values_be, N, gamma, lambda, scale, returns,
                       returns_be, adv_raw, adv, scratch);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
