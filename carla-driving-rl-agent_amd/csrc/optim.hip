// Per-tensor clip-by-norm + Keras Adam over flat parameter arenas (gfx950).
//
// Reference: utils.clip_gradients -> tf.clip_by_norm per tensor (rl/utils.py:120-121), applied to
// the policy / value heads only (F9); Keras Adam (beta1 .9, beta2 .999, eps 1e-7) in the fused
// `ResourceApplyAdam` form:
//     alpha = lr * sqrt(1 - b2^t) / (1 - b1^t);  m += (g - m)(1 - b1);  v += (g*g - v)(1 - b2)
//     theta -= m * alpha / (sqrt(v) + eps)
// Hyper-parameters and the Adam step counters live in a device block (DevHP) so that a captured
// hipGraph of the whole update step can be replayed while learning rates change between steps.
// One workgroup handles one 1024-element chunk of one tensor: HBM-bound streaming, no atomics,
// norms reduced in two deterministic stages.
#include "cdrl_kernels.h"

namespace cdrl {

#define CHUNK 1024

__global__ void __launch_bounds__(256) sqnorm_chunk_kernel(const float* __restrict__ g,
                                                           const TensorSeg* __restrict__ segs,
                                                           const int* __restrict__ chunk_tensor,
                                                           const int64_t* __restrict__ chunk_off,
                                                           double* __restrict__ chunk_part, DevHP* hp, int tick) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    // Adam step counters advanced here instead of by launches of their own: the trunk's (its update ran in FRONT of this kernel) and the
    // head's, whose update runs BEHIND it and is told so (clip_adam's `ticked`)
    // (bit mask: 1 policy, 2 value, 4 trunk)
    if (tick > 0 && c == 0 && threadIdx.x == 0) {
        if (tick & 1) hp->t_policy += 1;
        if (tick & 2) hp->t_value += 1;
        if (tick & 4) hp->t_dynamics += 1;
    }
    const TensorSeg s = segs[chunk_tensor[c]];
    const int64_t beg = chunk_off[c];
    int64_t end = beg + CHUNK;
    if (end > s.off + s.n) end = s.off + s.n;
    double acc = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        const double v = (double)g[i];
        acc += v * v;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_part[c] = sm[0];
}

// one wave per tensor: lane l sums the chunks l, l + 64, ... in order, then a fixed shuffle tree (one THREAD per tensor walked up to
// 768 chunk partials one dependent load at a time: 17 us)
__global__ void __launch_bounds__(256) sqnorm_final_kernel(const TensorSeg* __restrict__ segs, int ntensors,
                                                           const double* __restrict__ chunk_part, float* __restrict__ sqnorms) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntensors) return;
    const TensorSeg s = segs[t];
    double acc = 0.0;
    for (int c = lane; c < s.nchunks; c += 64) acc += chunk_part[s.first_chunk + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) sqnorms[t] = (float)acc;
}

int tensor_sqnorms(const float* g, const TensorSeg* segs_dev, int ntensors, const int* chunk_tensor_dev,
                   const int64_t* chunk_off_dev, int nchunks, double* chunk_part, float* sqnorms, hipStream_t st, DevHP* tick_hp,
                   int tick_mask, bool fold_final) {
    hipLaunchKernelGGL(sqnorm_chunk_kernel, dim3(nchunks), dim3(256), 0, st, g, segs_dev, chunk_tensor_dev, chunk_off_dev,
                       chunk_part, tick_hp, tick_hp ? tick_mask : 0);
    CDRL_LAUNCH_CHECK();
    if (fold_final) return 0;           // the consumer (clip_adam with chunk_part) folds the chunk partials of its tensor itself
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(cdiv(ntensors, 4)), dim3(256), 0, st, segs_dev, ntensors, chunk_part,
                       sqnorms);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const int* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off,
                                                        const TensorSeg* __restrict__ segs,
                                                        const float* __restrict__ sqnorms, const DevHP* __restrict__ hp,
                                                        int which, const double* __restrict__ chunk_part, int ticked) {
    int64_t beg, end;
    float cn = 0.0f, denom = 1.0f;
    const float clip_norm = which == 0 ? hp->clip_norm_policy : (which == 1 ? hp->clip_norm_value : 0.0f);
    if (chunk_tensor) {
        const int c = blockIdx.x;
        const int t = chunk_tensor[c];
        const TensorSeg s = segs[t];
        beg = chunk_off[c];
        end = beg + CHUNK;
        if (end > s.off + s.n) end = s.off + s.n;
        if ((sqnorms || chunk_part) && clip_norm > 0.0f) {
            float l2;
            if (chunk_part) {
                // the tensor's squared norm from its chunk partials, exactly as sqnorm_final_kernel folds them (lane l: chunks l, l + 64,
                // ... in order, fixed shuffle tree) -- every wave of every block of the tensor computes the same bits
                const int lane = threadIdx.x & 63;
                double acc = 0.0;
                for (int cc = lane; cc < s.nchunks; cc += 64) acc += chunk_part[s.first_chunk + cc];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
                l2 = (float)__shfl(acc, 0, 64);
            } else {
                l2 = sqnorms[t];
            }
            const float norm = l2 > 0.0f ? sqrtf(l2) : l2;
            cn = clip_norm;
            denom = fmaxf(norm, clip_norm);
        }
    } else {
        beg = (int64_t)blockIdx.x * CHUNK;
        end = beg + CHUNK;
        if (end > n) end = n;
    }
    const float lr = which == 0 ? hp->lr_policy : (which == 1 ? hp->lr_value : hp->lr_dynamics);
    const int t1 = (which == 0 ? hp->t_policy : (which == 1 ? hp->t_value : hp->t_dynamics)) + (ticked ? 0 : 1);
    const float b1 = hp->beta1, b2 = hp->beta2, eps = hp->eps;
    const float alpha = lr * sqrtf(1.0f - powf(b2, (float)t1)) / (1.0f - powf(b1, (float)t1));
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        float gi = g[i];
        if (cn > 0.0f) gi = (gi * cn) / denom;
        float mi = m[i], vi = v[i];
        mi += (gi - mi) * (1.0f - b1);
        vi += (gi * gi - vi) * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        p[i] -= (mi * alpha) / (sqrtf(vi) + eps);
    }
}

int clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const int* chunk_tensor_dev,
              const int64_t* chunk_off_dev, int nchunks, const TensorSeg* segs_dev, const float* sqnorms, DevHP* hp,
              int which, hipStream_t st, const double* chunk_part, int ticked) {
    const int grid = chunk_tensor_dev ? nchunks : (int)cdiv64(n, CHUNK);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, chunk_tensor_dev, chunk_off_dev,
                       segs_dev, sqnorms, hp, which, chunk_part, ticked);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// two device-to-device copies as ONE launch of the library's own (old_policy <- policy: trainable and state slices): hipMemcpyAsync ran
// two blit kernels of the runtime with their own fences in the middle of the apply
__global__ void __launch_bounds__(256) copy_two_kernel(float* __restrict__ d0, const float* __restrict__ s0, int64_t n0,
                                                       float* __restrict__ d1, const float* __restrict__ s1, int64_t n1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n0) d0[i] = s0[i];
    else if (i < n0 + n1) d1[i - n0] = s1[i - n0];
}

int copy_two(float* d0, const float* s0, int64_t n0, float* d1, const float* s1, int64_t n1, hipStream_t st) {
    if (n0 + n1 <= 0) return 0;
    hipLaunchKernelGGL(copy_two_kernel, dim3((unsigned)cdiv64(n0 + n1, 256)), dim3(256), 0, st, d0, s0, n0, d1, s1, n1);
    CDRL_LAUNCH_CHECK();
    return 0;
}

__global__ void adam_tick_kernel(DevHP* hp, int which) {
    if (which == 0) hp->t_policy += 1;
    else if (which == 1) hp->t_value += 1;
    else hp->t_dynamics += 1;
}

int adam_tick(DevHP* hp, int which, hipStream_t st) {
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, hp, which);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
