// General float32 GEMM  C[M,N] (+)= A[M,K] B[K,N] + bias  on the bf16 matrix pipe by exact three-way operand splitting
// (gfx950, v_mfma_f32_32x32x16_bf16) -- the 464 -> 768 head convolution of the tower (reference core/architectures.py:170) and
// its backward-data product, where K and N exceed what gemm_pw_x3.hip keeps in registers.
//
// The float32-MFMA form of these products is the most matrix-bound kernel of the step (MfmaUtil 0.31-0.47,
// profiles/r02_pmc_mfma.json: 8.7 GFLOP = 56 us at the float32 MFMA peak for 60 MB of HBM traffic).  As in gemm_pw_x3.hip:
// a = a1 + a2 + a3, b = b1 + b2 + b3 exactly in bf16, six products per K = 16 step carry a b to 2^-24.
//   * B (the weights) is split and packed ONCE per pass into MFMA fragment order  Bp[plane][k step][k half][n][8]  (gemm_x3_pack);
//     a wave reads its fragments straight from that array (16-byte loads, L2 resident), no LDS for B;
//   * A is loaded as float32 (16-byte lanes), split on its way into LDS (three bf16 planes, 128 rows x 32 k per stage), double
//     buffered: the loads of stage i+1 are issued before the MFMAs of stage i;
//   * 128 x 128 output tile per workgroup, 2 x 2 waves of 64 x 64 (four 32x32 accumulators each).
#include "colreduce.h"
#include "pack_bodies.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct GemmX3Args {
    View A;
    const __bf16* Bp;           // [3][KS][2][NP][8], KS = ceil(K/16), NP = N padded to 128
    const float* bias;          // [N] or null
    View C;
    int accumulate;
    int M, N, K, KS, NP;
};

__device__ __forceinline__ void gx3_split(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// NS = 3: exact three-way split (float32-accurate product).  NS = 1: the bf16-operand compute mode of configuration 3 -- both
// operands rounded once to bf16 (plane 0 of the split = round-to-nearest-even), one MFMA per K = 16 step.
// BH (with NS = 1 only): bf16 ACTIVATION STORAGE -- A and C are bf16 in HBM: the A chunks (4 elements, 8 bytes) go to LDS as they
// are, C is rounded on store (and read as bf16 when accumulating).
template <int NS, bool BH = false>
__global__ void __launch_bounds__(256, 2) gemm_x3_kernel(GemmX3Args a) {
    static_assert(!BH || NS == 1, "bf16 storage implies bf16 operands");
    typedef typename std::conditional<BH, bf16_t, float>::type T;
    constexpr uint32_t ESZ = BH ? 2u : 4u;
    constexpr int BM = 128, BK = 32, LDA = BK + 8;          // bf16 elements per LDS row (+16 bytes)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][BM * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int lrow = lane & 31, lk = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * 128;
    const int K = a.K, N = a.N;
    const int nstage = (K + BK - 1) / BK;
    // A stage loads: 128 rows x 8 chunks of 4 floats = 1024 chunks, 4 per thread
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const uint32_t OOR = 0x80000000u;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.A.p, 0, (int)((int64_t)a.M * a.A.ld * ESZ), 0x00020000);
    const int cr = tid >> 3, ck = 4 * (tid & 7);            // chunk row (0..31, +32 i), chunk k
    auto load_stage = [&](int s, u32x4_t (&ra)[4]) {
        const int k = s * BK + ck;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + cr + 32 * i;
            const bool ok = m < a.M && k < K;
            const uint32_t off = ok ? (uint32_t)((m * a.A.ld + a.A.coff + k) * ESZ) : OOR;
            if (BH) {
                const u32x2_t h = __builtin_amdgcn_raw_buffer_load_b64(rsA, off, 0, 0);
                ra[i][0] = h[0];
                ra[i][1] = h[1];
            } else {
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
            }
        }
    };
    auto store_stage = [&](int buf, const u32x4_t (&ra)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BH) {       // already bf16: 4 elements = 2 dwords
                u32x2_t h2;
                h2[0] = ra[i][0];
                h2[1] = ra[i][1];
                *reinterpret_cast<u32x2_t*>(&As[buf][0][(cr + 32 * i) * LDA + ck]) = h2;
                continue;
            }
            bf16x4 h[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h1, h2, h3;
                gx3_split(__uint_as_float(ra[i][e]), h1, h2, h3);
                h[0][e] = h1;
                h[1][e] = h2;
                h[2][e] = h3;
            }
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<bf16x4*>(&As[buf][p][(cr + 32 * i) * LDA + ck]) = h[p];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // B fragments of one K = 16 step for this wave's two column tiles: [plane][column tile]
    const int64_t plane = (int64_t)a.KS * 2 * a.NP * 8;
    auto load_b = [&](int ks, bf16x8 (&bf)[NS][2]) {
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wc * 64 + j * 32 + lrow;
                bf[p][j] = *reinterpret_cast<const bf16x8*>(a.Bp + p * plane + (((int64_t)ks * 2 + lk) * a.NP + n) * 8);
            }
    };
    auto mma = [&](int buf, int kk, const bf16x8 (&bf)[NS][2]) {
        bf16x8 af[NS][2];
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[p][i] = *reinterpret_cast<const bf16x8*>(&As[buf][p][(wr * 64 + i * 32 + lrow) * LDA + 16 * kk + 8 * lk]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];           // smallest terms first
                if constexpr (NS == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };
    // B fragments (L2) are fetched one K = 16 step ahead; A stages (HBM, ~2 us under load) TWO stages ahead in two register
    // sets -- one stage is only ~1500 matrix-pipe cycles and left every stage waiting on its loads (91 us for 24 us of MFMA)
    bf16x8 b0[NS][2], b1[NS][2];
    u32x4_t raA[4], raB[4];
    load_stage(0, raA);
    load_b(0, b0);
    if (nstage > 1) load_stage(1, raB);
    store_stage(0, raA);
    if (nstage > 2) load_stage(2, raA);
    __syncthreads();
    auto iter = [&](int s, u32x4_t (&ruse)[4]) {
        // on entry: stage s is in LDS buffer s & 1, stage s + 1 sits in `ruse`, stage s + 2 is in flight in the other set
        const int buf = s & 1, ks = 2 * s;
        if (ks + 1 < a.KS) load_b(ks + 1, b1);
        mma(buf, 0, b0);
        if (ks + 1 < a.KS) {
            if (ks + 2 < a.KS) load_b(ks + 2, b0);
            mma(buf, 1, b1);
        }
        if (s + 1 < nstage) store_stage(buf ^ 1, ruse);   // the other buffer: its last readers passed the previous barrier
        if (s + 3 < nstage) load_stage(s + 3, ruse);
        __syncthreads();
    };
    for (int s = 0; s < nstage; s += 2) {
        iter(s, raB);
        if (s + 1 < nstage) iter(s + 1, raA);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + lrow;
        if (n >= N) continue;
        const float bv = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < a.M) {
                    T* c = vptr<T>(a.C) + m * a.C.ld + a.C.coff + n;
                    float v = acc[i][j][r] + bv;
                    if (a.accumulate) v += ldf(c);
                    stf(c, v);
                }
            }
    }
}

// B(k, n) = w[k * sbk + n * sbn] -> [3][KS][2][NP][8] bf16
__global__ void gemm_x3_pack_kernel(const GemmX3Pack* __restrict__ tab) {
    const GemmX3Pack d = tab[blockIdx.y];
    gemm_x3_pack_body(d, blockIdx.x, gridDim.x);
}

int64_t gemm_x3_packed_bytes(int N, int K) { return (int64_t)3 * cdiv(K, 16) * 2 * (cdiv(N, 128) * 128) * 8 * 2; }

GemmX3Pack gemm_x3_pack_entry(const float* w, void* wp, int K, int N, int sbk, int sbn) {
    GemmX3Pack e;
    e.w = w;
    e.wp = reinterpret_cast<__bf16*>(wp);
    e.K = K;
    e.N = N;
    e.sbk = sbk;
    e.sbn = sbn;
    e.KS = cdiv(K, 16);
    e.NP = cdiv(N, 128) * 128;
    return e;
}

int gemm_x3_pack_many(const GemmX3Pack* tab_dev, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gemm_x3_pack_kernel, dim3(32, n), dim3(256), 0, st, tab_dev);
    CDRL_LAUNCH_CHECK();
    return 0;
}

bool gemm_x3_supported(View A, int K) {
    return K >= 4 && K % 4 == 0 && A.ld % 4 == 0 && A.coff % 4 == 0 && (reinterpret_cast<uintptr_t>(A.p) & 15) == 0;
}

int gemm_x3(View A, const void* Bp, const float* bias, View C, int M, int N, int K, int accumulate, hipStream_t st, bool bf16_operands,
            int at) {
    if (M <= 0 || N <= 0) return 0;
    if (!gemm_x3_supported(A, K) || !Bp) {
        set_error("gemm_x3: unsupported alignment K=%d ld=%d coff=%d", K, A.ld, A.coff);
        return -1;
    }
    if ((int64_t)M * A.ld * 4 >= (int64_t)1 << 31) {
        set_error("gemm_x3: operand of 2 GB or more");
        return -1;
    }
    GemmX3Args a{A, reinterpret_cast<const __bf16*>(Bp), bias, C, accumulate, M, N, K, cdiv(K, 16), cdiv(N, 128) * 128};
    if (at && !bf16_operands) {
        set_error("gemm_x3: bf16 activation storage needs the bf16-operand variant");
        return -1;
    }
    if (at) hipLaunchKernelGGL((gemm_x3_kernel<1, true>), dim3(cdiv(M, 128), cdiv(N, 128)), dim3(256), 0, st, a);
    else if (bf16_operands) hipLaunchKernelGGL(gemm_x3_kernel<1>, dim3(cdiv(M, 128), cdiv(N, 128)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_x3_kernel<3>, dim3(cdiv(M, 128), cdiv(N, 128)), dim3(256), 0, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
