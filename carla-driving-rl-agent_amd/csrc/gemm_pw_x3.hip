// Float32 pointwise (1x1) convolution on the bf16 matrix pipe by exact operand splitting (gfx950, v_mfma_f32_32x32x16_bf16).
//
// Reference: core/architectures.py:130,140 (Conv2D(k=1)).  The float32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16
// rate, and the persistent float32 kernel of gemm_pw.hip is measurably matrix-bound (MfmaUtil 0.36 at K = N = 116,
// profiles/r02_pmc_mfma.json).  A float32 number is the exact sum of three bf16 numbers (3 x 8 significand bits):
//     a = a1 + a2 + a3,   b = b1 + b2 + b3
// and  a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a b|):  six bf16 products accumulated in float32 carry
// the product to float32 accuracy (the dropped terms are below half an ulp of the product).  Six K=16 steps of 32 cycles replace
// eight K=2 steps of 64 cycles: 2.7x less matrix-pipe time for the same float32 inputs, outputs and HBM traffic.
// The result is NOT bit-identical to a k-ordered fmaf chain (it is at least as accurate: every product is exact to 2^-24 and the
// accumulation is float32 either way); parity is asserted against the float64 reference at 1e-5 (tests/test_gpu_ops.py).
// Same skeleton as gemm_pw.hip: persistent workgroups over the row tiles of one BatchNorm group, W (three bf16 planes, packed
// once per weight version) in registers, A split on its way into LDS, BN-apply prologue, statistics epilogue.
#include <stdlib.h>

#include "cdrl_kernels.h"
#include "pack_bodies.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct PwX3Args {
    View A;
    const float* pro_stats;     // [4][G][K] or null
    const __bf16* Wp;           // [3][KP/16][2][128][8]
    const float* bias;
    View C;
    double* part;               // [G][nbpg][2][N] or null
    int N, K, G, Mg, nbpg;
};

__device__ __forceinline__ void split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)x;
    const float r1 = x - (float)h1;             // exact
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);              // exact difference, final rounding below 2^-24 |x|
}

template <int KP, int NT, bool PRO, bool EPI>
__global__ void __launch_bounds__(256, 2) pw_x3_kernel(PwX3Args a) {
    constexpr int WC = NT, WR = 4 / WC, BM = 32 * WR, KS = KP / 16;
    constexpr int LDA = KP + 8;
    constexpr int CPR = KP / 4;                      // 16-byte chunks (4 floats) per row
    constexpr int NCH = BM * CPR / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDA];
    __shared__ float pc[2][KP];
    __shared__ double red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int lrow = lane & 31, lk = lane >> 5;
    const int g = blockIdx.x / a.nbpg, b = blockIdx.x % a.nbpg;
    const int K = a.K, N = a.N;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int tiles_g = (a.Mg + BM - 1) / BM;
    const int t0 = (int)((int64_t)b * tiles_g / a.nbpg), t1 = (int)((int64_t)(b + 1) * tiles_g / a.nbpg);
    const int n = wc * 32 + lrow;
    bf16x8 breg[3][KS];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            breg[p][s] = *reinterpret_cast<const bf16x8*>(a.Wp + (((int64_t)(p * KS + s) * 2 + lk) * 128 + n) * 8);
    const float bv = (a.bias && n < N) ? a.bias[n] : 0.0f;
    if (PRO) {
        const int GK = a.G * K;
        for (int k = tid; k < KP; k += 256) {
            pc[0][k] = k < K ? a.pro_stats[2 * GK + g * K + k] : 0.0f;
            pc[1][k] = k < K ? a.pro_stats[3 * GK + g * K + k] : 0.0f;
        }
        __syncthreads();
    }
    double s1 = 0.0, s2 = 0.0;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const uint32_t OOR = 0x80000000u;
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.A.p, 0, (int)(Mtot * a.A.ld * 4), 0x00020000);
    u32x4_t ra[NCH];
    auto load_tile = [&](int t) {
        const int64_t m0 = mbeg + (int64_t)t * BM;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
            const bool ok = (m0 + r < mend) && (k4 < K);
            const uint32_t off = (uint32_t)(((m0 + r) * a.A.ld + a.A.coff + k4) * 4);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : OOR, 0, 0);
        }
    };
    auto store_tile = [&]() {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        // pairs: the arithmetic compiles to v_pk_{fma,add}_f32, the three conversions to one v_cvt_pk_bf16_f32 each, and the
        // conversion results ARE the packed LDS words (no per-element repacking): 19.9 -> see DESIGN.md for the isolated timing
        auto widen = [](uint32_t w) -> f32x2 { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
            uint32_t hw[3][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 x = f32x2{__uint_as_float(ra[i][2 * h]), __uint_as_float(ra[i][2 * h + 1])};
                // columns beyond K inside the last 4-wide chunk (K = 58: 58, 59) are whatever the neighbouring channels of a wider
                // tensor hold: a non-finite value there would survive the zero weights (0 * NaN); zeroed here
                if (k4 + 2 * h >= K) x[0] = 0.0f;
                if (k4 + 2 * h + 1 >= K) x[1] = 0.0f;
                if (PRO) {
                    // (columns beyond K carry scale 0 / shift 0 in pc)
                    const f32x2 sc = f32x2{pc[0][k4 + 2 * h], pc[0][k4 + 2 * h + 1]}, sh = f32x2{pc[1][k4 + 2 * h], pc[1][k4 + 2 * h + 1]};
                    x = __builtin_elementwise_fma(sc, x, sh);
                }
                hw[0][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
                const f32x2 r1 = x - widen(hw[0][h]);             // exact
                hw[1][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
                hw[2][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1 - widen(hw[1][h]), bf16x2));
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2_t*>(&As[p][r * LDA + k4]) = u32x2_t{hw[p][0], hw[p][1]};
        }
    };
    auto compute_tile = [&](int t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const int ao = (wr * 32 + lrow) * LDA + 8 * lk;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[0][ao + 16 * s]);
            const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[1][ao + 16 * s]);
            const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&As[2][ao + 16 * s]);
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, breg[0][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[2][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[1][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[0][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[1][s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[0][s], acc, 0, 0, 0);
        }
        if (n >= N) return;
        const int64_t m0 = mbeg + (int64_t)t * BM + wr * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < mend) {
                const float v = acc[r] + bv;
                if (EPI) {
                    s1 += (double)v;
                    s2 += (double)v * (double)v;
                }
                a.C.p[m * a.C.ld + a.C.coff + n] = v;
            }
        }
    };
    if (t0 < t1) load_tile(t0);
    for (int t = t0; t < t1; ++t) {
        store_tile();
        __syncthreads();
        if (t + 1 < t1) load_tile(t + 1);
        compute_tile(t);
        __syncthreads();
    }
    if (EPI && a.part) {
        const double f1 = s1 + __shfl_down(s1, 32), f2 = s2 + __shfl_down(s2, 32);
        if (lk == 0) {
            red[0][wave][lrow] = f1;
            red[1][wave][lrow] = f2;
        }
        __syncthreads();
        if (wr == 0 && lk == 0 && n < N) {
            double u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int w = 0; w < WR; ++w) {
                u1 += red[0][wc * WR + w][lrow];
                u2 += red[1][wc * WR + w][lrow];
            }
            double* p = a.part + ((int64_t)g * a.nbpg + b) * 2 * N;
            p[n] = u1;
            p[N + n] = u2;
        }
    }
}


// K, N up to 256 (the 232-channel convs of stage 2; M = 12288 rows at B = 256): the product is tiny and the persistent form -- one
// workgroup per CU that walks three tiles with all of W in its registers (gemm_pw.hip <128, 4>: 116 float32-MFMA steps of 64 cycles per
// tile and column tile, one wave per SIMD, 26-36 us per launch for 23 MB) -- is a serial chain of load / MFMA / store phases.  Here a
// workgroup takes ONE 32-row tile and one block of 128 columns: W^(column block) as three bf16 planes in registers (16 K steps x 3 planes
// x 4 VGPRs = 192), the A tile split into three planes in LDS (50 KB), 96 bf16 MFMAs of 32 cycles per wave, two workgroups per CU so
// that one loads / stores while the other multiplies.  Grid = (groups x tiles, column blocks); statistics: one partial row per tile.
template <bool PRO, bool EPI>
__global__ void __launch_bounds__(256, 2) pw_x3_wide_kernel(PwX3Args a) {
    constexpr int KP = 256, BM = 32, KS = KP / 16, LDA = KP + 8, CPR = KP / 4, NCH = BM * CPR / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDA];
    __shared__ float pc[2][KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int g = blockIdx.x / a.nbpg, t = blockIdx.x % a.nbpg;        // nbpg = tiles per group
    const int K = a.K, N = a.N;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int64_t m0 = mbeg + (int64_t)t * BM;
    const int nl = wave * 32 + lrow, n = blockIdx.y * 128 + nl;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t OOR = 0x80000000u;
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.A.p, 0, (int)(Mtot * a.A.ld * 4), 0x00020000);
    // the A tile first (HBM / Infinity Cache), the weight fragments (L2) behind it: both in flight before anything is consumed
    u32x4_t ra[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
        const bool ok = (m0 + r < mend) && (k4 < K);
        const uint32_t off = (uint32_t)(((m0 + r) * a.A.ld + a.A.coff + k4) * 4);
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? off : OOR, 0, 0);
    }
    if (PRO) {
        const int GK = a.G * K;
        for (int k = tid; k < KP; k += 256) {
            pc[0][k] = k < K ? a.pro_stats[2 * GK + g * K + k] : 0.0f;
            pc[1][k] = k < K ? a.pro_stats[3 * GK + g * K + k] : 0.0f;
        }
    }
    // A workgroup uses every weight fragment ONCE (one tile), so holding all 3 x 16 of them (192 VGPRs) buys nothing but the early issue
    // of their loads -- and spilled next to the prologue.  A ring of PF K-steps (3 planes each) is requested ahead instead.
    const __bf16* wp = a.Wp + (int64_t)blockIdx.y * 3 * KS * 2 * 128 * 8 + ((int64_t)lk * 128 + nl) * 8;
    constexpr int PF = 6;
    bf16x8 ring[PF][3];
    auto wload = [&](int s, bf16x8* dst) {
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(p * KS + s) * 2 * 128 * 8);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) wload(s, ring[s]);
    const float bv = (a.bias && n < N) ? a.bias[n] : 0.0f;
    if (PRO) __syncthreads();
    {
        auto widen = [](uint32_t w) -> f32x2 { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
            uint32_t hw[3][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 x = f32x2{__uint_as_float(ra[i][2 * h]), __uint_as_float(ra[i][2 * h + 1])};
                if (k4 + 2 * h >= K) x[0] = 0.0f;          // (columns beyond K inside the last chunk: see pw_x3_kernel)
                if (k4 + 2 * h + 1 >= K) x[1] = 0.0f;
                if (PRO) {
                    const f32x2 sc = f32x2{pc[0][k4 + 2 * h], pc[0][k4 + 2 * h + 1]}, sh = f32x2{pc[1][k4 + 2 * h], pc[1][k4 + 2 * h + 1]};
                    x = __builtin_elementwise_fma(sc, x, sh);
                }
                hw[0][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
                const f32x2 r1 = x - widen(hw[0][h]);
                hw[1][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
                hw[2][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1 - widen(hw[1][h]), bf16x2));
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2_t*>(&As[p][r * LDA + k4]) = u32x2_t{hw[p][0], hw[p][1]};
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int ao = lrow * LDA + 8 * lk;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[0][ao + 16 * s]);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[1][ao + 16 * s]);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&As[2][ao + 16 * s]);
        const bf16x8 b1 = ring[s % PF][0], b2 = ring[s % PF][1], b3 = ring[s % PF][2];
        if (s + PF < KS) wload(s + PF, ring[s % PF]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);       // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < mend) {
            const float v = acc[r] + bv;
            if (EPI) {
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
            a.C.p[m * a.C.ld + a.C.coff + n] = v;
        }
    }
    if (EPI && a.part) {
        const double f1 = s1 + __shfl_down(s1, 32), f2 = s2 + __shfl_down(s2, 32);
        if (lk == 0) {
            double* p = a.part + ((int64_t)g * a.nbpg + t) * 2 * N;
            p[n] = f1;
            p[N + n] = f2;
        }
    }
}


// Backward-data of the same convs, one tile per workgroup (round 6): C[M, N] (+)= dy[M, K] W^T with the BatchNorm backward of the BatchNorm
// behind the conv applied on load -- dy = k1 (mask dz - k2 - xhat(y) k3), dz optionally gathered through the channel shuffle (destination
// columns c, c + 2 and c + 1, c + 3 are adjacent pairs of the source), column sums of dy (the conv's bias gradient) as one partial row per
// tile -- and optionally the backward sums (sum c, sum c xhat(ey)) of the BatchNorm the OUTPUT feeds (EPI) or accumulation onto the old
// output (ACC).  Same contract as pw_nn_kernel<128, 4, PRO_BNBWD, EPI, 0, ACC> (gemm_pw.hip), which ran these shapes as three serial tiles
// per CU on the float32 matrix pipe (32-53 us per launch at M = 12288).  The weight fragments are requested behind the prologue (its raw
// tiles and the 192 fragment VGPRs do not fit together at two workgroups per CU); the co-resident workgroup covers that latency.
struct PwX3BwdArgs {
    View A;                     // dz
    int a_shuffle, a_act;
    const float* a_y;           // raw input of the BatchNorm behind the conv [M][K] dense
    const float* pro_stats;     // [4][G][K]
    const float* pro_coef;      // [3][G][K]
    double* part2;              // [G][nbpg][K] or null
    const __bf16* Wp;           // pw_x3 packing of B(k, n), column blocks of 128
    View C;
    const float* ey;            // EPI: [M][N] dense
    const float* epi_stats;     // EPI: [4][G][N]
    double* part;               // EPI: [G][nbpg][2][N]
    int N, K, G, Mg, nbpg;
};

template <bool SHUF, bool EPI, bool ACC>
__global__ void __launch_bounds__(256, 2) pw_x3_wide_bwd_kernel(PwX3BwdArgs a) {
    constexpr int KP = 256, BM = 32, KS = KP / 16, LDA = KP + 8, CPR = KP / 4, NCH = BM * CPR / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDA];
    __shared__ float qc[7][KP];         // mean, invstd, scale, shift, k1, k2, k3 of the dy columns (0 beyond K)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int g = blockIdx.x / a.nbpg, t = blockIdx.x % a.nbpg;
    const int K = a.K, N = a.N;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int64_t m0 = mbeg + (int64_t)t * BM;
    const int nl = wave * 32 + lrow, n = blockIdx.y * 128 + nl;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const uint32_t OOR = 0x80000000u;
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.A.p, 0, (int)(Mtot * a.A.ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a_y), 0, (int)(Mtot * K * 4), 0x00020000);
    // a thread's chunks: rows r0 + 4 i, ALWAYS the same four columns k4 .. k4 + 3 (256 % CPR == 0)
    const int k4 = 4 * (tid % CPR), r0 = tid / CPR;
    const bool kon = k4 < K;                    // (K % 4 == 0: a chunk is valid or not as a whole)
    uint32_t voD0, voD1;                        // byte offsets inside a row: SHUF -> sources of columns (0, 2) and (1, 3); dense -> one 16-byte chunk
    if (SHUF) {
        voD0 = kon ? (uint32_t)shuffle_dst(a.A.coff + k4, a.a_shuffle) * 4u : OOR;
        voD1 = kon ? (uint32_t)shuffle_dst(a.A.coff + k4 + 1, a.a_shuffle) * 4u : OOR;
    } else {
        voD0 = kon ? (uint32_t)(a.A.coff + k4) * 4u : OOR;
        voD1 = OOR;
    }
    const uint32_t voY = kon ? (uint32_t)k4 * 4u : OOR;
    const uint32_t rowA = (uint32_t)a.A.ld * 4u, rowY = (uint32_t)K * 4u;
    u32x4_t rz[NCH], ry[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int64_t m = m0 + r0 + 4 * i;
        const uint32_t msk = m < mend ? 0u : OOR;
        const uint32_t mu = (uint32_t)m;
        if (SHUF) {
            const u32x2_t p0 = __builtin_amdgcn_raw_buffer_load_b64(rsA, (voD0 + mu * rowA) | msk | (voD0 & OOR), 0, 0);
            const u32x2_t p1 = __builtin_amdgcn_raw_buffer_load_b64(rsA, (voD1 + mu * rowA) | msk | (voD1 & OOR), 0, 0);
            rz[i] = u32x4_t{p0[0], p1[0], p0[1], p1[1]};
        } else {
            rz[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (voD0 + mu * rowA) | msk | (voD0 & OOR), 0, 0);
        }
        ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (voY + mu * rowY) | msk | (voY & OOR), 0, 0);
    }
    {
        const int GK = a.G * K;
        for (int i = tid; i < 7 * KP; i += 256) {
            const int q = i / KP, k = i % KP;
            qc[q][k] = k < K ? (q < 4 ? a.pro_stats[q * GK + g * K + k] : a.pro_coef[(q - 4) * GK + g * K + k]) : 0.0f;
        }
    }
    __syncthreads();
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    {
        auto widen = [](uint32_t w) -> f32x2 { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
        const bool relu6 = a.a_act == ACT_RELU6;
        f32x2 cmean[2], cinv[2], csc[2], csh[2], ck1[2], ck2[2], ck3[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cmean[h] = f32x2{qc[0][k4 + 2 * h], qc[0][k4 + 2 * h + 1]};
            cinv[h] = f32x2{qc[1][k4 + 2 * h], qc[1][k4 + 2 * h + 1]};
            csc[h] = f32x2{qc[2][k4 + 2 * h], qc[2][k4 + 2 * h + 1]};
            csh[h] = f32x2{qc[3][k4 + 2 * h], qc[3][k4 + 2 * h + 1]};
            ck1[h] = f32x2{qc[4][k4 + 2 * h], qc[4][k4 + 2 * h + 1]};
            ck2[h] = f32x2{qc[5][k4 + 2 * h], qc[5][k4 + 2 * h + 1]};
            ck3[h] = f32x2{qc[6][k4 + 2 * h], qc[6][k4 + 2 * h + 1]};
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = r0 + 4 * i;
            const bool rok = m0 + r < mend;
            uint32_t hw[3][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 d = f32x2{__uint_as_float(rz[i][2 * h]), __uint_as_float(rz[i][2 * h + 1])};
                const f32x2 yv = f32x2{__uint_as_float(ry[i][2 * h]), __uint_as_float(ry[i][2 * h + 1])};
                if (relu6) {
                    const f32x2 z = __builtin_elementwise_fma(csc[h], yv, csh[h]);      // = fmaf(scale, y, shift) of the forward
                    if (!relu6_open(z[0])) d[0] = 0.0f;
                    if (!relu6_open(z[1])) d[1] = 0.0f;
                }
                const f32x2 xh = (yv - cmean[h]) * cinv[h];
                f32x2 v = ck1[h] * (d - ck2[h] - xh * ck3[h]);          // columns beyond K: every coefficient 0 -> 0
                if (!rok) v = f32x2{0.0f, 0.0f};
                cs[2 * h] += (double)v[0];
                cs[2 * h + 1] += (double)v[1];
                hw[0][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                const f32x2 r1 = v - widen(hw[0][h]);
                hw[1][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
                hw[2][h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1 - widen(hw[1][h]), bf16x2));
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2_t*>(&As[p][r * LDA + k4]) = u32x2_t{hw[p][0], hw[p][1]};
        }
    }
    // weight fragments + the epilogue's operands: requested now, consumed behind the barrier
    const __bf16* wp = a.Wp + (int64_t)blockIdx.y * 3 * KS * 2 * 128 * 8 + ((int64_t)lk * 128 + nl) * 8;
    constexpr int PF = 6;                       // ring of K-steps requested ahead (see pw_x3_wide_kernel)
    bf16x8 ring[PF][3];
    auto wload = [&](int s, bf16x8* dst) {
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(p * KS + s) * 2 * 128 * 8);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) wload(s, ring[s]);
    const bool non = n < N;
    float ext[(EPI || ACC) ? 16 : 1];           // EPI: raw input of the BatchNorm the output feeds; ACC: old output values
    if (EPI || ACC) {
        const __amdgpu_buffer_rsrc_t rsE = EPI ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ey), 0, (int)(Mtot * N * 4), 0x00020000)
                                               : __builtin_amdgcn_make_buffer_rsrc(a.C.p, 0, (int)(Mtot * a.C.ld * 4), 0x00020000);
        const uint32_t rowE = EPI ? (uint32_t)N * 4u : (uint32_t)a.C.ld * 4u;
        const uint32_t voE = non ? (uint32_t)((EPI ? 0 : a.C.coff) + n) * 4u : OOR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            ext[(EPI || ACC) ? r : 0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsE, (voE + (uint32_t)m * rowE) | (m < mend ? 0u : OOR) | (voE & OOR), 0, 0));
        }
    }
    float emean = 0.0f, einv = 0.0f;
    if (EPI && non) {
        emean = a.epi_stats[0 * a.G * N + g * N + n];
        einv = a.epi_stats[1 * a.G * N + g * N + n];
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int ao = lrow * LDA + 8 * lk;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[0][ao + 16 * s]);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[1][ao + 16 * s]);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&As[2][ao + 16 * s]);
        const bf16x8 b1 = ring[s % PF][0], b2 = ring[s % PF][1], b3 = ring[s % PF][2];
        if (s + PF < KS) wload(s + PF, ring[s % PF]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);       // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    }
    double s1 = 0.0, s2 = 0.0;
    if (non) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < mend) {
                float v = acc[r];
                if (EPI) {
                    const float xh = (ext[EPI ? r : 0] - emean) * einv;
                    s1 += (double)v;
                    s2 += (double)v * (double)xh;
                }
                if (ACC) v += ext[ACC ? r : 0];
                a.C.p[m * a.C.ld + a.C.coff + n] = v;
            }
        }
    }
    if (EPI && a.part) {
        const double f1 = s1 + __shfl_down(s1, 32), f2 = s2 + __shfl_down(s2, 32);
        if (lk == 0 && non) {
            double* p = a.part + ((int64_t)g * a.nbpg + t) * 2 * N;
            p[n] = f1;
            p[N + n] = f2;
        }
    }
    if (a.part2 && blockIdx.y == 0) {
        // column sums of dy: the four row groups of a column chunk folded in fixed order (the LDS planes are dead: barrier first)
        __syncthreads();
        double* red = reinterpret_cast<double*>(&As[0][0]);     // [4][KP]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[r0 * KP + k4 + e] = cs[e];
        __syncthreads();
        for (int k = tid; k < K; k += 256) a.part2[((int64_t)g * a.nbpg + t) * K + k] = (red[k] + red[KP + k]) + (red[2 * KP + k] + red[3 * KP + k]);
    }
}

static inline int x3_kp(int K) { return K <= 32 ? 32 : (K <= 64 ? 64 : (K <= 128 ? 128 : 256)); }
static inline bool x3_wide(int N, int K) { return K > 128 || N > 128; }
static inline int x3_nt(int N) { return N <= 32 ? 1 : (N <= 64 ? 2 : 4); }

// B(k, n) = w[k * sbk + n * sbn] -> three bf16 planes of MFMA B fragments [3][KP/16][2][128][8]
__global__ void pw_x3_pack_many_kernel(const PwX3Pack* __restrict__ tab) {
    const PwX3Pack d = tab[blockIdx.y];
    pw_x3_pack_body(d, blockIdx.x, gridDim.x);
}

int64_t pw_x3_packed_bytes(int K) { return (int64_t)3 * (x3_kp(K) / 16) * 2 * 128 * 8 * 2; }      // one column block (N <= 128)
int pw_x3_ksteps(int K) { return x3_kp(K) / 16; }       // K = 16 steps per plane of a pack (K <= 128: 2 | 4 | 8)
// (the wide form always runs with KP = 256: its kernel is instantiated for that padding only)
int64_t pw_x3_packed_bytes_n(int K, int N) { return x3_wide(N, K) ? pw_x3_packed_bytes(256) * ((N + 127) / 128) : pw_x3_packed_bytes(K); }

PwX3Pack pw_x3_pack_entry(const float* w, void* wp, int K, int N, int sbk, int sbn) {
    PwX3Pack e;
    e.w = w;
    e.wp = reinterpret_cast<__bf16*>(wp);
    e.K = K;
    e.N = N;
    e.sbk = sbk;
    e.sbn = sbn;
    e.kp = x3_wide(N, K) ? 256 : x3_kp(K);
    return e;
}

int pw_x3_pack_many(const PwX3Pack* tab_dev, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pw_x3_pack_many_kernel, dim3(8, n), dim3(256), 0, st, tab_dev);
    CDRL_LAUNCH_CHECK();
    return 0;
}

bool pw_x3_supported(View A, int N, int K) {
    // K <= 128 (the persistent kernel): its A chunks are 16-byte BUFFER loads, which need dword alignment only, and columns beyond K
    // inside the last chunk are zeroed -- so the 58-channel rows of stage 0 (232-byte pitch, channel offset 58) qualify (round 6;
    // CDRL_PW_X3_UNALIGNED=0 -> 16-byte aligned rows only, as before: those convs then stay on the float32 matrix pipe)
    static const bool unal = !(cdrl_getenv("CDRL_PW_X3_UNALIGNED") && atoi(cdrl_getenv("CDRL_PW_X3_UNALIGNED")) == 0);
    if (unal && K >= 4 && K <= 128 && N >= 1 && N <= 128 && K % 2 == 0 && A.ld % 2 == 0 && A.coff % 2 == 0 &&
        (reinterpret_cast<uintptr_t>(A.p) & 15) == 0)
        return true;
    return K >= 4 && K <= 256 && N >= 1 && N <= 256 && K % 4 == 0 && A.ld % 4 == 0 && A.coff % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(A.p) & 15) == 0;
}

static int x3_occ() {
    static const int v = 2;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
}

int pw_x3_partial_rows(int G, int Mg, int N, int K) {
    if (x3_wide(N, K)) return cdiv(Mg, 32);         // one partial row per 32-row tile
    const int bm = 32 * (4 / x3_nt(N)), tiles = cdiv(Mg, bm);
    int nb = 256 * x3_occ() / G;
    if (nb < 1) nb = 1;
    return nb > tiles ? tiles : nb;
}

template <int KP, int NT>
static void pw_x3_launch(const PwX3Args& a, bool pro, bool epi, hipStream_t st) {
    const dim3 grid(a.G * a.nbpg), block(256);
    if (pro && epi) hipLaunchKernelGGL((pw_x3_kernel<KP, NT, true, true>), grid, block, 0, st, a);
    else if (pro) hipLaunchKernelGGL((pw_x3_kernel<KP, NT, true, false>), grid, block, 0, st, a);
    else if (epi) hipLaunchKernelGGL((pw_x3_kernel<KP, NT, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_x3_kernel<KP, NT, false, false>), grid, block, 0, st, a);
}

int pw_x3(View A, const float* pro_stats, const void* Wp, const float* bias, View C, int G, int Mg, int N, int K, double* part,
          hipStream_t st, int nbpg) {
    if (!pw_x3_supported(A, N, K) || !Wp) {
        set_error("pw_x3: unsupported shape / alignment K=%d N=%d ld=%d coff=%d", K, N, A.ld, A.coff);
        return -1;
    }
    if ((int64_t)G * Mg * A.ld * 4 >= (int64_t)1 << 31) {
        set_error("pw_x3: operand of 2 GB or more");
        return -1;
    }
    PwX3Args a{A, pro_stats, reinterpret_cast<const __bf16*>(Wp), bias, C, part, N, K, G, Mg, nbpg > 0 ? nbpg : pw_x3_partial_rows(G, Mg, N, K)};
    const int kp = x3_kp(K), nt = x3_nt(N);
    const bool pro = pro_stats != nullptr, epi = part != nullptr;
    if (x3_wide(N, K)) {
        if (a.nbpg != cdiv(Mg, 32)) {
            set_error("pw_x3: the wide form writes one partial row per 32-row tile (%d), not %d", cdiv(Mg, 32), a.nbpg);
            return -1;
        }
        const dim3 grid(G * a.nbpg, cdiv(N, 128)), block(256);
        if (pro && epi) hipLaunchKernelGGL((pw_x3_wide_kernel<true, true>), grid, block, 0, st, a);
        else if (pro) hipLaunchKernelGGL((pw_x3_wide_kernel<true, false>), grid, block, 0, st, a);
        else if (epi) hipLaunchKernelGGL((pw_x3_wide_kernel<false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_x3_wide_kernel<false, false>), grid, block, 0, st, a);
        CDRL_LAUNCH_CHECK();
        return 0;
    }
#define CDRL_X3(KPV, NTV) pw_x3_launch<KPV, NTV>(a, pro, epi, st)
    if (kp == 32) { if (nt == 1) CDRL_X3(32, 1); else if (nt == 2) CDRL_X3(32, 2); else CDRL_X3(32, 4); }
    else if (kp == 64) { if (nt == 1) CDRL_X3(64, 1); else if (nt == 2) CDRL_X3(64, 2); else CDRL_X3(64, 4); }
    else { if (nt == 1) CDRL_X3(128, 1); else if (nt == 2) CDRL_X3(128, 2); else CDRL_X3(128, 4); }
#undef CDRL_X3
    CDRL_LAUNCH_CHECK();
    return 0;
}


bool pw_x3_wide_bwd_supported(View dz, View C, int N, int K, int shuffle_ctot) {
    // N = conv input channels (columns of the output), K = conv output channels (columns of dz / y)
    if (!(K > 128 || N > 128) || K > 256 || N > 256 || (K & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(dz.p) & 15) || (dz.ld & 1)) return false;
    if (shuffle_ctot) return (dz.coff & 1) == 0 && ((shuffle_ctot >> 1) & 1) == 0;
    return (dz.ld & 3) == 0 && (dz.coff & 3) == 0;
}

int pw_x3_wide_bwd_rows(int Mg) { return cdiv(Mg, 32); }

int pw_x3_wide_bwd(View dz, const PwBnBwd& bb, const void* Wp, View C, int accumulate, int G, int Mg, int N, int K, const float* ey,
                   const float* epi_stats, double* part, hipStream_t st) {
    if (!pw_x3_wide_bwd_supported(dz, C, N, K, bb.shuffle_ctot) || !Wp || !bb.y || !bb.stats || !bb.coef) {
        set_error("pw_x3_wide_bwd: unsupported shape / alignment K=%d N=%d ld=%d coff=%d", K, N, dz.ld, dz.coff);
        return -1;
    }
    if ((ey != nullptr) && accumulate) {
        set_error("pw_x3_wide_bwd: the BatchNorm-sum epilogue and accumulation are not instantiated together");
        return -1;
    }
    const int64_t Mtot = (int64_t)G * Mg;
    if (Mtot * dz.ld * 4 >= (int64_t)1 << 31 || Mtot * K * 4 >= (int64_t)1 << 31 || Mtot * C.ld * 4 >= (int64_t)1 << 31) {
        set_error("pw_x3_wide_bwd: operand of 2 GB or more");
        return -1;
    }
    PwX3BwdArgs a;
    a.A = dz;
    a.a_shuffle = bb.shuffle_ctot;
    a.a_act = bb.act;
    a.a_y = bb.y;
    a.pro_stats = bb.stats;
    a.pro_coef = bb.coef;
    a.part2 = bb.part2;
    a.Wp = reinterpret_cast<const __bf16*>(Wp);
    a.C = C;
    a.ey = ey;
    a.epi_stats = epi_stats;
    a.part = part;
    a.N = N;
    a.K = K;
    a.G = G;
    a.Mg = Mg;
    a.nbpg = cdiv(Mg, 32);
    const dim3 grid(G * a.nbpg, cdiv(N, 128)), block(256);
    const bool shuf = bb.shuffle_ctot != 0, epi = ey != nullptr, acc = accumulate != 0;
#define CDRL_X3B(SH)                                                                                          \
    do {                                                                                                      \
        if (epi) hipLaunchKernelGGL((pw_x3_wide_bwd_kernel<SH, true, false>), grid, block, 0, st, a);         \
        else if (acc) hipLaunchKernelGGL((pw_x3_wide_bwd_kernel<SH, false, true>), grid, block, 0, st, a);    \
        else hipLaunchKernelGGL((pw_x3_wide_bwd_kernel<SH, false, false>), grid, block, 0, st, a);            \
    } while (0)
    if (shuf) CDRL_X3B(true);
    else CDRL_X3B(false);
#undef CDRL_X3B
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
