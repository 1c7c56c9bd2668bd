// Bodies of the per-pass weight-packing kernels (pw_pack_many / pw_x3_pack_many / gemm_x3_pack_many / transpose_many) as device functions,
// so that ONE launch can run all four at the start of a training pass (pack_all, gemm_pw.hip): they were four dependent 4-13 us
// launches of the critical stream in front of every stem conv.  `bx` / `nbx` stand for blockIdx.x / gridDim.x of the original kernels.
#pragma once
#include "cdrl_kernels.h"

namespace cdrl {

typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pk_split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)x;
    const float r1 = x - (float)h1;             // exact
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);              // exact difference, final rounding below 2^-24 |x|
}

// B(k, n) = w[k * sbk + n * sbn] -> fragment order [ct = n / 32][lk = k & 1][n & 31][s = k >> 1] (zero padded to KSM, 32)
__device__ __forceinline__ void pw_pack_body(const PwPack& d, int bx, int nbx) {
    const int total = d.ntiles * 2 * 32 * d.ksm;
    if (d.bf16) {
        // bf16 fragments of the BF variant: [ct][s = k / 16][lk][lrow][8], element e <-> k = 16 s + 8 lk + e (round-to-nearest-even)
        __bf16* wb = reinterpret_cast<__bf16*>(d.wp);
        const int ks = d.ksm / 8;
        for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
            const int e = i % 8, lrow = (i / 8) % 32, lk = (i / 256) % 2, s = (i / 512) % ks, ct = i / (512 * ks);
            const int k = 16 * s + 8 * lk + e, n = ct * 32 + lrow;
            wb[i] = (__bf16)((k < d.K && n < d.N) ? d.w[(int64_t)k * d.sbk + (int64_t)n * d.sbn] : 0.0f);
        }
        return;
    }
    for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
        const int s = i % d.ksm, lrow = (i / d.ksm) % 32, lk = (i / (d.ksm * 32)) % 2, ct = i / (d.ksm * 64);
        const int k = 2 * s + lk, n = ct * 32 + lrow;
        d.wp[i] = (k < d.K && n < d.N) ? d.w[(int64_t)k * d.sbk + (int64_t)n * d.sbn] : 0.0f;
    }
}

// B(k, n) -> three bf16 planes of MFMA B fragments, column blocks of 128: [block][3][KP/16][2][128][8]
__device__ __forceinline__ void pw_x3_pack_body(const PwX3Pack& d, int bx, int nbx) {
    const int ks = d.kp / 16, total = ks * 2 * 128, nblk = (d.N + 127) / 128;
    for (int ii = bx * 256 + threadIdx.x; ii < total * nblk; ii += nbx * 256) {
        const int blk = ii / total, i = ii % total;
        const int nl = i % 128, n = blk * 128 + nl, lk = (i / 128) % 2, s = i / 256;
        pk_bf16x8 v[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * s + 8 * lk + e;
            const float x = (k < d.K && n < d.N) ? d.w[(int64_t)k * d.sbk + (int64_t)n * d.sbn] : 0.0f;
            __bf16 h1, h2, h3;
            pk_split3(x, h1, h2, h3);
            v[0][e] = h1;
            v[1][e] = h2;
            v[2][e] = h3;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<pk_bf16x8*>(d.wp + (((int64_t)blk * 3 + p) * total + i) * 8) = v[p];
    }
}

// B(k, n) -> [3][KS][2][NP][8] bf16
__device__ __forceinline__ void gemm_x3_pack_body(const GemmX3Pack& d, int bx, int nbx) {
    const int64_t plane = (int64_t)d.KS * 2 * d.NP * 8;
    const int total = d.KS * 2 * d.NP;
    for (int i = bx * 256 + threadIdx.x; i < total; i += nbx * 256) {
        const int n = i % d.NP, lk = (i / d.NP) % 2, ks = i / (2 * d.NP);
        pk_bf16x8 v[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * lk + e;
            const float x = (k < d.K && n < d.N) ? d.w[(int64_t)k * d.sbk + (int64_t)n * d.sbn] : 0.0f;
            __bf16 h1, h2, h3;
            pk_split3(x, h1, h2, h3);
            v[0][e] = h1;
            v[1][e] = h2;
            v[2][e] = h3;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<pk_bf16x8*>(d.wp + p * plane + (int64_t)i * 8) = v[p];
    }
}

// one 32 x 32 tile per block through LDS (t: a [32][33] float tile of the calling kernel); blocks beyond the matrix's tiles do nothing
__device__ __forceinline__ void transpose_body(const PwTranspose& d, int bx, float (*t)[33]) {
    const int tk = (d.cin + 31) / 32, tn = (d.cout + 31) / 32;
    if (bx >= tk * tn) return;
    const int k0 = (bx / tn) * 32, n0 = (bx % tn) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        t[r][tx] = (k0 + r < d.cin && n0 + tx < d.cout) ? d.w[(int64_t)(k0 + r) * d.cout + n0 + tx] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < d.cout && k0 + tx < d.cin) d.wt[(int64_t)(n0 + r) * d.cin + k0 + tx] = t[tx][r];
}

}  // namespace cdrl
