// Rollout-time image augmentation on the device (reference core/carla_agent.py:545-577: color jitter -> random-kernel
// "blur" -> salt & pepper -> gaussian noise -> per-image min-max normalisation -> cutout -> coarse dropout; ops from
// rl/augmentations/augmentations.py and rl/augmentations/simclr.py:49-63).
//
// The host draws the plan (which ops fire and their scalar parameters: tf_chance / tf.image.random_* in the reference);
// the per-pixel random fields (salt & pepper masks, gaussian noise, dropout grid) come from Philox streams keyed by
// (seed, offset*8 + stream id, element index) so that a plan + seed determines the output exactly (oracle/augment.py
// reproduces the streams bit for bit).  The stack is tiny (T x 90 x 120 x 3 floats): kernels are simple grid-stride
// loops, reductions use one workgroup per image; nothing here is on the learner's timed path.
#include "cdrl_kernels.h"
#include "philox.h"

namespace cdrl {

enum { AUG_SP_SELECT = 1, AUG_SP_NOISE = 2, AUG_GN_SELECT = 3, AUG_GN_NOISE = 4, AUG_DROPOUT = 5 };

__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
    v = fmaxf(fmaxf(r, g), b);
    const float mn = fminf(fminf(r, g), b);
    const float d = v - mn;
    s = v > 0.0f ? d / v : 0.0f;
    if (d > 0.0f) {
        float hh;
        if (v == r) hh = (g - b) / d;
        else if (v == g) hh = 2.0f + (b - r) / d;
        else hh = 4.0f + (r - g) / d;
        hh /= 6.0f;
        h = hh - floorf(hh);
    } else {
        h = 0.0f;
    }
}

__device__ __forceinline__ float hsv_f(float n, float h6, float s, float v) {
    const float k = fmodf(n + h6, 6.0f);
    return v - v * s * fmaxf(0.0f, fminf(fminf(k, 4.0f - k), 1.0f));
}

__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float h6 = h * 6.0f;
    r = hsv_f(5.0f, h6, s, v);
    g = hsv_f(3.0f, h6, s, v);
    b = hsv_f(1.0f, h6, s, v);
}

// per-(image, channel) mean over H*W: one workgroup per image
__global__ void __launch_bounds__(1024) aug_channel_mean_kernel(const float* __restrict__ x, int P, float add, float* __restrict__ mean) {
    __shared__ double sm[3][1024];
    const float* xp = x + (int64_t)blockIdx.x * P * 3;
    double s[3] = {0.0, 0.0, 0.0};
    for (int p = threadIdx.x; p < P; p += blockDim.x)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += (double)(xp[p * 3 + c] + add);
#pragma unroll
    for (int c = 0; c < 3; ++c) sm[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int c = 0; c < 3; ++c) sm[c][threadIdx.x] += sm[c][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) mean[blockIdx.x * 3 + threadIdx.x] = (float)(sm[threadIdx.x][0] / (double)P);
}

__global__ void aug_jitter_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t npix, int P,
                                  const float* __restrict__ mean, float brightness, float contrast, float saturation, float hue) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / P);
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float m = mean[t * 3 + k];
            c[k] = (x[i * 3 + k] + brightness - m) * contrast + m;
        }
        float h, s, v;
        rgb_to_hsv(c[0], c[1], c[2], h, s, v);
        s = fminf(fmaxf(s * saturation, 0.0f), 1.0f);
        hsv_to_rgb(h, s, v, c[0], c[1], c[2]);
        rgb_to_hsv(c[0], c[1], c[2], h, s, v);
        h = h + hue;
        h = h - floorf(h);
        hsv_to_rgb(h, s, v, c[0], c[1], c[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) y[i * 3 + k] = fminf(fmaxf(c[k], 0.0f), 1.0f);
    }
}

struct BlurK {
    float w[75];
};

__global__ void aug_blur_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int H, int W, int k, BlurK bk) {
    const int64_t n = (int64_t)T * H * W * 3;
    const int r = k / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        int64_t p = i / 3;
        const int xx = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H);
        const int t = (int)(p / H);
        float acc = 0.0f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = yy + ky - r;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = xx + kx - r;
                if (ix < 0 || ix >= W) continue;
                acc = fmaf(x[(((int64_t)t * H + iy) * W + ix) * 3 + c], bk.w[(ky * k + kx) * 3 + c], acc);
            }
        }
        y[i] = acc;
    }
}

__global__ void aug_noise_kernel(float* __restrict__ x, int64_t npix, int salt_pepper, float sp_p, float sp_prob, int gauss, float gn_amount,
                                 float gn_std, uint64_t seed, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        float c[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
        if (salt_pepper) {
            Philox a(seed, offset * 8 + AUG_SP_SELECT, (uint64_t)i), b(seed, offset * 8 + AUG_SP_NOISE, (uint64_t)i);
            const float sel = a.uniform() < (double)sp_p ? 1.0f : 0.0f;
            const float nz = b.uniform() < (double)sp_prob ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = c[k] * (1.0f - sel) + nz * sel;
        }
        if (gauss) {
            Philox a(seed, offset * 8 + AUG_GN_SELECT, (uint64_t)i);
            const float sel = a.uniform() < (double)gn_amount ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Philox g(seed, offset * 8 + AUG_GN_NOISE, (uint64_t)(i * 3 + k));
                const float nz = (float)(g.normal() * (double)gn_std);
                c[k] += fminf(fmaxf(sel * nz, 0.0f), 1.0f);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) x[i * 3 + k] = c[k];
    }
}

// per-image (min, max - min): one workgroup per image
__global__ void __launch_bounds__(1024) aug_minmax_kernel(const float* __restrict__ x, int n_per_image, float* __restrict__ mm) {
    __shared__ float smin[1024], smax[1024];
    const float* xp = x + (int64_t)blockIdx.x * n_per_image;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n_per_image; i += blockDim.x) {
        lo = fminf(lo, xp[i]);
        hi = fmaxf(hi, xp[i]);
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + st]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mm[blockIdx.x * 2] = smin[0];
        mm[blockIdx.x * 2 + 1] = smax[0] - smin[0];
    }
}

__device__ __forceinline__ int nearest_src(int dst, int out, int in) {
    int s = (int)floorf(((float)dst + 0.5f) * (float)in / (float)out);
    return s < in - 1 ? s : in - 1;
}

__global__ void aug_final_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int H, int W, int normalize,
                                 const float* __restrict__ mm, float eps, int cutout_size, int cutout_cell, int dropout_size,
                                 float dropout_keep, uint64_t seed, uint64_t offset) {
    const int64_t npix = (int64_t)T * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i;
        const int xx = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H);
        const int t = (int)(p / H);
        float mask = 1.0f;
        if (cutout_size > 0) {
            const int cy = nearest_src(yy, H, cutout_size), cx = nearest_src(xx, W, cutout_size);
            if (cy * cutout_size + cx == cutout_cell) mask = 0.0f;
        }
        if (dropout_size > 0) {
            const int cy = nearest_src(yy, H, dropout_size), cx = nearest_src(xx, W, dropout_size);
            Philox d(seed, offset * 8 + AUG_DROPOUT, (uint64_t)(cy * dropout_size + cx));
            if (!(d.uniform() < (double)dropout_keep)) mask = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = x[i * 3 + k];
            if (normalize) v = (v - mm[t * 2]) / (mm[t * 2 + 1] + eps);
            y[i * 3 + k] = v * mask;
        }
    }
}

int augment_images(const float* in, float* out, int T, int H, int W, const AugPlan& p, float* workspace, hipStream_t st) {
    if (T <= 0 || H <= 0 || W <= 0) {
        set_error("augment_images: bad shape %dx%dx%d", T, H, W);
        return -1;
    }
    if (p.blur_size != 0 && p.blur_size != 3 && p.blur_size != 5) {
        set_error("augment_images: blur_size must be 0, 3 or 5");
        return -1;
    }
    const int P = H * W;
    const int64_t npix = (int64_t)T * P, n = npix * 3;
    float* bufA = workspace;                // [n]
    float* bufB = workspace + n;            // [n]
    float* small = workspace + 2 * n;       // means [T][3] | minmax [T][2]
    const int grid = (int)((npix + 255) / 256 < 1024 ? (npix + 255) / 256 : 1024);
    const float* cur = in;
    if (p.jitter) {
        hipLaunchKernelGGL(aug_channel_mean_kernel, dim3(T), dim3(1024), 0, st, cur, P, p.brightness, small);
        hipLaunchKernelGGL(aug_jitter_kernel, dim3(grid), dim3(256), 0, st, cur, bufA, npix, P, small, p.brightness, p.contrast,
                           p.saturation, p.hue);
        cur = bufA;
    }
    if (p.blur_size) {
        BlurK bk;
        for (int i = 0; i < 75; ++i) bk.w[i] = p.blur_kernel[i];
        float* dst = cur == bufA ? bufB : bufA;
        hipLaunchKernelGGL(aug_blur_kernel, dim3(grid * 3 < 1024 ? grid * 3 : 1024), dim3(256), 0, st, cur, dst, T, H, W, p.blur_size, bk);
        cur = dst;
    }
    if (p.salt_pepper || p.gauss_noise) {
        float* buf = const_cast<float*>(cur);
        if (cur == in) {        // never write into the caller's input
            CDRL_HIP(hipMemcpyAsync(bufA, in, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            buf = bufA;
        }
        hipLaunchKernelGGL(aug_noise_kernel, dim3(grid), dim3(256), 0, st, buf, npix, p.salt_pepper, p.sp_amount / 10.0f, p.sp_prob,
                           p.gauss_noise, p.gn_amount, p.gn_std, p.seed, p.offset);
        cur = buf;
    }
    if (p.normalize) hipLaunchKernelGGL(aug_minmax_kernel, dim3(T), dim3(1024), 0, st, cur, P * 3, small + 3 * T);
    hipLaunchKernelGGL(aug_final_kernel, dim3(grid), dim3(256), 0, st, cur, out, T, H, W, p.normalize, small + 3 * T, 1.1920929e-07f,
                       p.cutout_size, p.cutout_cell, p.dropout_size, 1.0f - p.dropout_amount, p.seed, p.offset);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
