// The linear output heads of a control branch (reference core/networks.py:128-137 policy alpha / beta / similarity /
// speed, :267-275 value base / exp / speed / similarity) as ONE launch per direction.
//
// Each head is Dense(320 -> 1..3, linear) on the same (B, 320) activation.  As four separate GEMMs they were four
// two-workgroup launches of 15 us each in the forward and four backward-data GEMMs plus 16 side-stream launches (column
// sums, reductions, filter-gradient GEMM + reduction per head) in the backward -- all on the latency-bound stretch of the
// step between the trunk's forward and backward.  The outputs are 256 x <= 8 numbers: plain dot products, accumulated in
// double in a fixed order (deterministic), no MFMA.
#include "cdrl_kernels.h"

namespace cdrl {

// flat view of the output columns: column c of lin belongs to head h(c), weight column n(c)
struct HeadCols {
    int L;
    const float* w[HEADS_MAX_OUT];     // &W_h[0][n]
    int ws[HEADS_MAX_OUT];             // row stride of W_h (= n_h)
    int lc[HEADS_MAX_OUT];             // column in lin
    float bias[HEADS_MAX_OUT];         // (forward only; filled on the device)
};

__device__ __forceinline__ HeadCols head_cols(const HeadSet& hs) {
    HeadCols hc;
    int c = 0;
#pragma unroll
    for (int h = 0; h < HEADS_MAX; ++h) {
        if (h >= hs.nheads) break;
        for (int n = 0; n < hs.n[h]; ++n, ++c) {
            hc.w[c] = hs.w[h] + n;
            hc.ws[c] = hs.n[h];
            hc.lc[c] = hs.off[h] + n;
            hc.bias[c] = hs.b[h] ? hs.b[h][n] : 0.0f;
        }
    }
    hc.L = c;
    for (; c < HEADS_MAX_OUT; ++c) {       // unused columns alias column 0 (loaded, never stored)
        hc.w[c] = hs.w[0];
        hc.ws[c] = hs.n[0];
        hc.lc[c] = hs.off[0];
        hc.bias[c] = 0.0f;
    }
    return hc;
}

// forward: lin[b][off_h + n] = bias_h[n] + sum_k a[b][k] * W_h[k][n];  one wave per row, lane-strided k, butterfly sum
__global__ void __launch_bounds__(256) heads_fwd_kernel(const float* __restrict__ a, int lda, HeadSet hs, float* __restrict__ lin,
                                                        int ldl, int B, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const HeadCols hc = head_cols(hs);
    double acc[HEADS_MAX_OUT];
#pragma unroll
    for (int c = 0; c < HEADS_MAX_OUT; ++c) acc[c] = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float x = a[(int64_t)b * lda + k];
        float wv[HEADS_MAX_OUT];
#pragma unroll
        for (int c = 0; c < HEADS_MAX_OUT; ++c) wv[c] = hc.w[c][(int64_t)k * hc.ws[c]];
#pragma unroll
        for (int c = 0; c < HEADS_MAX_OUT; ++c) acc[c] += (double)x * (double)wv[c];
    }
#pragma unroll
    for (int c = 0; c < HEADS_MAX_OUT; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);      // every lane ends with the same fixed-order sum
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < HEADS_MAX_OUT; ++c)
            if (c < hc.L) lin[(int64_t)b * ldl + hc.lc[c]] = (float)acc[c] + hc.bias[c];
    }
}

// backward, one launch: blocks [0, nbd) write da[b][k] = sum_c dlin[b][c] * W[k][c] (overwrite);
//                       blocks [nbd, ..) write dW_h[k][n] = sum_b a[b][k] * dlin[b][off_h + n] and db_h[n] = sum_b dlin[b][off_h + n]
__global__ void __launch_bounds__(256) heads_bwd_kernel(const float* __restrict__ a, int lda, HeadSet hs,
                                                        const float* __restrict__ dlin, int ldl, float* __restrict__ da, int ldda,
                                                        int B, int K, int nbd) {
    const int tid = threadIdx.x;
    const HeadCols hc = head_cols(hs);
    if ((int)blockIdx.x < nbd) {
        const int64_t i = (int64_t)blockIdx.x * 256 + tid;
        if (i >= (int64_t)B * K) return;
        const int b = (int)(i / K), k = (int)(i - (int64_t)b * K);
        float dv[HEADS_MAX_OUT], wv[HEADS_MAX_OUT];
#pragma unroll
        for (int c = 0; c < HEADS_MAX_OUT; ++c) {
            dv[c] = dlin[(int64_t)b * ldl + hc.lc[c]];
            wv[c] = hc.w[c][(int64_t)k * hc.ws[c]];
        }
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < HEADS_MAX_OUT; ++c)
            if (c < hc.L) s += (double)dv[c] * (double)wv[c];
        da[(int64_t)b * ldda + k] = (float)s;
        return;
    }
    // weight / bias gradients: block = 16 k lanes (k == K: the bias row) x 16 row lanes; a row lane walks rows rl, rl+16, ...
    // with 4 rows in flight, the 16 lane sums are folded through LDS in a fixed order (a single thread per k walking all
    // rows was one 256-step chain of dependent L2 round trips: 125 us)
    __shared__ double red[16][16][HEADS_MAX_OUT];
    const int kl = tid & 15, rl = tid >> 4;
    const int k = ((int)blockIdx.x - nbd) * 16 + kl;
    double acc[HEADS_MAX_OUT];
#pragma unroll
    for (int c = 0; c < HEADS_MAX_OUT; ++c) acc[c] = 0.0;
    if (k <= K) {
        for (int b0 = rl; b0 < B; b0 += 16 * 4) {
            float x[4], d[4][HEADS_MAX_OUT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = min(b0 + 16 * u, B - 1);          // clamped: unconditional loads
                x[u] = (k < K) ? a[(int64_t)bb * lda + k] : 1.0f;
#pragma unroll
                for (int c = 0; c < HEADS_MAX_OUT; ++c) d[u][c] = dlin[(int64_t)bb * ldl + hc.lc[c]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b0 + 16 * u >= B) break;
#pragma unroll
                for (int c = 0; c < HEADS_MAX_OUT; ++c) acc[c] += (double)x[u] * (double)d[u][c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < HEADS_MAX_OUT; ++c) red[rl][kl][c] = acc[c];
    __syncthreads();
    if (rl == 0 && k <= K) {
        int c = 0;
        for (int h = 0; h < hs.nheads; ++h)
            for (int n = 0; n < hs.n[h]; ++n, ++c) {
                double s = 0.0;
#pragma unroll
                for (int y = 0; y < 16; ++y) s += red[y][kl][c];
                if (k < K) hs.gw[h][(int64_t)k * hs.n[h] + n] = (float)s;
                else hs.gb[h][n] = (float)s;
            }
    }
}

static int heads_check(const HeadSet& hs) {
    int tot = 0;
    if (hs.nheads < 1 || hs.nheads > HEADS_MAX) {
        set_error("heads: %d heads not supported", hs.nheads);
        return -1;
    }
    for (int h = 0; h < hs.nheads; ++h) tot += hs.n[h];
    if (tot > HEADS_MAX_OUT) {
        set_error("heads: %d outputs > %d", tot, HEADS_MAX_OUT);
        return -1;
    }
    return 0;
}

int heads_fwd(const float* a, int lda, const HeadSet& hs, float* lin, int ldl, int B, int K, hipStream_t st) {
    CDRL_TRY(heads_check(hs));
    hipLaunchKernelGGL(heads_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, a, lda, hs, lin, ldl, B, K);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int heads_bwd(const float* a, int lda, const HeadSet& hs, const float* dlin, int ldl, float* da, int ldda, int B, int K,
              hipStream_t st) {
    CDRL_TRY(heads_check(hs));
    const int nbd = da ? (int)cdiv64((int64_t)B * K, 256) : 0;
    const int nbw = hs.gw[0] ? cdiv(K + 1, 16) : 0;
    if (nbd + nbw == 0) return 0;
    hipLaunchKernelGGL(heads_bwd_kernel, dim3(nbd + nbw), dim3(256), 0, st, a, lda, hs, dlin, ldl, da, ldda, B, K, nbd);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
