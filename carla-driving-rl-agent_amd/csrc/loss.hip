// Fused loss kernels (forward + analytic backward w.r.t. the linear head outputs), gfx950.
//
//  policy_loss : CARLAgent.policy_objective (reference core/carla_agent.py:394-428) on top of
//                PolicyNetwork.call (core/networks.py:96-110): Beta(alpha,beta) log-prob of the
//                clipped sample, entropy, mean-over-actions ratio, spinning-up style clipping,
//                0.5*MSE aux speed / similarity heads.  The Beta sample u and (optionally) its
//                pathwise Jacobians are inputs (F8 / Appendix C-1).
//  value_loss  : CARLAgent.value_objective (core/carla_agent.py:469-486).
// One workgroup, wavefront/LDS reductions in double: B is a few hundred rows, the kernels are
// latency-bound, so everything transcendental (lgamma / digamma / trigamma) runs in double.
#include "cdrl_kernels.h"

namespace cdrl {

__device__ double digamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0)))));
    return r + log(x) - 0.5 / x + t;
}

__device__ double trigamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r += 1.0 / (x * x);
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = 1.0 / x + 0.5 * f +
                     (1.0 / x) * f * (1.0 / 6.0 + f * (-1.0 / 30.0 + f * (1.0 / 42.0 + f * (-1.0 / 30.0 + f * (5.0 / 66.0)))));
    return r + t;
}

__device__ __forceinline__ double softplus_d(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

template <int NQ>
__device__ void block_reduce(double* acc, double* sm) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) sm[q * blockDim.x + tid] = acc[q];
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) sm[q * blockDim.x + tid] += sm[q * blockDim.x + tid + s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = sm[q * blockDim.x];
    __syncthreads();
}

#define F32_EPS 1.1920929e-07f

__global__ void __launch_bounds__(256) policy_loss_kernel(PolicyLossArgs p) {
    __shared__ double sm[6 * 256];
    const DevHP* hp = reinterpret_cast<const DevHP*>(p.hp);
    const double clip = (double)hp->clip_ratio, cent = (double)hp->entropy_coef;
    const int B = p.B, A = p.A, L = 2 * A + 2;
    const double invB = 1.0 / (double)B, invA = 1.0 / (double)A;
    const double gs = (double)p.inv_world;
    double acc[6] = {0, 0, 0, 0, 0, 0};   // policy term, entropy, speed sq, sim sq, ratio, logp
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* lin = p.lin + (int64_t)b * L;
        float* dlin = p.dlin + (int64_t)b * L;
        double ea[8], dlp_da[8], dlp_db[8], dH_da[8], dH_db[8], sg_a[8], sg_b[8];
        double ratio = 0.0;
        for (int a = 0; a < A; ++a) {
            const double la = (double)lin[a], lb = (double)lin[A + a];
            const double al = softplus_d(la) + 1.01, be = softplus_d(lb) + 1.01;
            const float uf = p.u[(int64_t)b * A + a];
            const float xf = fminf(fmaxf(uf, F32_EPS), 1.0f - F32_EPS);      // _clip_actions
            const bool inside = (uf >= F32_EPS) && (uf <= 1.0f - F32_EPS);
            const double x = (double)xf;
            const double lx = log(x), l1x = log1p(-x);
            const double lnB = lgamma(al) + lgamma(be) - lgamma(al + be);
            const double pa = digamma_d(al), pb = digamma_d(be), pab = digamma_d(al + be);
            const double logp = (al - 1.0) * lx + (be - 1.0) * l1x - lnB;
            const double ent = lnB - (al - 1.0) * pa - (be - 1.0) * pb + (al + be - 2.0) * pab;
            const double e = exp(logp - (double)p.old_logp[(int64_t)b * A + a]);
            ea[a] = e;
            ratio += e;
            double dlx = inside ? ((al - 1.0) / x - (be - 1.0) / (1.0 - x)) : 0.0;
            const double ja = p.du_da ? (double)p.du_da[(int64_t)b * A + a] : 0.0;
            const double jb = p.du_db ? (double)p.du_db[(int64_t)b * A + a] : 0.0;
            dlp_da[a] = lx - pa + pab + dlx * ja;
            dlp_db[a] = l1x - pb + pab + dlx * jb;
            const double tab = trigamma_d(al + be);
            dH_da[a] = -(al - 1.0) * trigamma_d(al) + (al + be - 2.0) * tab;
            dH_db[a] = -(be - 1.0) * trigamma_d(be) + (al + be - 2.0) * tab;
            sg_a[a] = sigmoid_d(la);
            sg_b[a] = sigmoid_d(lb);
            acc[1] += ent;
            acc[5] += logp;
            if (p.aux) {
                float* ax = p.aux + (int64_t)b * 4 * A;
                ax[a] = (float)al;
                ax[A + a] = (float)be;
                ax[2 * A + a] = (float)logp;
                ax[3 * A + a] = (float)ent;
            }
        }
        ratio *= invA;
        const double adv = (double)p.adv[b];
        const double minadv = adv > 0.0 ? (1.0 + clip) * adv : (1.0 - clip) * adv;
        const double s = ratio * adv;
        const bool pass = s <= minadv;                 // tf.minimum routes the gradient to x where x <= y
        acc[0] += pass ? s : minadv;
        acc[4] += ratio;
        for (int a = 0; a < A; ++a) {
            const double dL_dlogp = pass ? (-invB * adv * invA * ea[a]) : 0.0;
            const double dL_dH = -cent * invB * invA;
            dlin[a] = (float)(gs * (dL_dlogp * dlp_da[a] + dL_dH * dH_da[a]) * sg_a[a]);
            dlin[A + a] = (float)(gs * (dL_dlogp * dlp_db[a] + dL_dH * dH_db[a]) * sg_b[a]);
        }
        const double sim = tanh((double)lin[2 * A]);
        const double sgs = sigmoid_d((double)lin[2 * A + 1]);
        const double spd = 2.0 * sgs;
        const double dsim = sim - (double)p.similarity[b], dspd = spd - (double)p.speed[b];
        acc[3] += dsim * dsim;
        acc[2] += dspd * dspd;
        dlin[2 * A] = (float)(gs * invB * dsim * (1.0 - sim * sim));
        dlin[2 * A + 1] = (float)(gs * invB * dspd * 2.0 * sgs * (1.0 - sgs));
    }
    block_reduce<6>(acc, sm);
    if (threadIdx.x == 0) {
        const double policy_loss = -acc[0] * invB;
        const double entropy = acc[1] * invB * invA;
        const double speed_loss = 0.5 * acc[2] * invB, sim_loss = 0.5 * acc[3] * invB;
        p.metrics[0] = (float)(policy_loss - cent * entropy + speed_loss + sim_loss);
        p.metrics[1] = (float)policy_loss;
        p.metrics[2] = (float)entropy;
        p.metrics[3] = (float)speed_loss;
        p.metrics[4] = (float)sim_loss;
        p.metrics[5] = (float)(acc[4] * invB);
        p.metrics[6] = (float)(acc[5] * invB * invA);
    }
}

int policy_loss(const PolicyLossArgs& a, hipStream_t st) {
    if (a.A > 8) {
        set_error("policy_loss: num_actions %d > 8 unsupported", a.A);
        return -1;
    }
    hipLaunchKernelGGL(policy_loss_kernel, dim3(1), dim3(256), 0, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(256) value_loss_kernel(ValueLossArgs p) {
    __shared__ double sm[4 * 256];
    const int B = p.B;
    const double invB = 1.0 / (double)B, es = (double)p.exp_scale, gs = (double)p.inv_world;
    double acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* lin = p.lin + (int64_t)b * 4;
        float* dlin = p.dlin + (int64_t)b * 4;
        const double base = tanh((double)lin[0]);
        const double se = sigmoid_d((double)lin[1]), ex = es * se;
        const double ss = sigmoid_d((double)lin[2]), spd = 2.0 * ss;
        const double sim = tanh((double)lin[3]);
        const double d0 = base - (double)p.returns[2 * b], d1 = ex - (double)p.returns[2 * b + 1];
        const double d2 = spd - (double)p.speed[b], d3 = sim - (double)p.similarity[b];
        acc[0] += d0 * d0;
        acc[1] += d1 * d1;
        acc[2] += d2 * d2;
        acc[3] += d3 * d3;
        // total = 0.25 * (0.25*mean d0^2 + mean d1^2 / es^2 + mean d2^2 + mean d3^2)
        dlin[0] = (float)(gs * 0.25 * 0.25 * 2.0 * d0 * invB * (1.0 - base * base));
        dlin[1] = (float)(gs * 0.25 * 2.0 * d1 * invB / (es * es) * es * se * (1.0 - se));
        dlin[2] = (float)(gs * 0.25 * 2.0 * d2 * invB * 2.0 * ss * (1.0 - ss));
        dlin[3] = (float)(gs * 0.25 * 2.0 * d3 * invB * (1.0 - sim * sim));
        if (p.values) {
            p.values[2 * b] = (float)base;
            p.values[2 * b + 1] = (float)ex;
        }
    }
    block_reduce<4>(acc, sm);
    if (threadIdx.x == 0) {
        const double value_loss = 0.25 * acc[0] * invB + acc[1] * invB / (es * es);
        const double speed_loss = acc[2] * invB, sim_loss = acc[3] * invB;
        p.metrics[0] = (float)(0.25 * (value_loss + speed_loss + sim_loss));
        p.metrics[1] = (float)value_loss;
        p.metrics[2] = (float)speed_loss;
        p.metrics[3] = (float)sim_loss;
    }
}

int value_loss(const ValueLossArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(256), 0, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// inference-side head activations (reference core/networks.py:96-110, 267-275)
__global__ void policy_dist_kernel(const float* __restrict__ lin, float* __restrict__ out, int B, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * A) return;
    const int b = i / A, a = i % A, L = 2 * A + 2;
    const double al = softplus_d((double)lin[(int64_t)b * L + a]) + 1.01;
    const double be = softplus_d((double)lin[(int64_t)b * L + A + a]) + 1.01;
    float* o = out + (int64_t)b * 4 * A;
    o[a] = (float)al;
    o[A + a] = (float)be;
    o[2 * A + a] = (float)(al / (al + be));
    o[3 * A + a] = (float)sqrt(al * be / ((al + be) * (al + be) * (al + be + 1.0)));
}

int policy_dist(const float* lin, float* out, int B, int A, hipStream_t st) {
    hipLaunchKernelGGL(policy_dist_kernel, dim3(cdiv(B * A, 256)), dim3(256), 0, st, lin, out, B, A);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void value_act_kernel(const float* __restrict__ lin, float* __restrict__ out, int B, float exp_scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[4 * b + 0] = (float)tanh((double)lin[4 * b + 0]);
    out[4 * b + 1] = (float)((double)exp_scale * sigmoid_d((double)lin[4 * b + 1]));
    out[4 * b + 2] = (float)(2.0 * sigmoid_d((double)lin[4 * b + 2]));
    out[4 * b + 3] = (float)tanh((double)lin[4 * b + 3]);
}

int value_act(const float* lin, float* out, int B, float exp_scale, hipStream_t st) {
    hipLaunchKernelGGL(value_act_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, lin, out, B, exp_scale);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
