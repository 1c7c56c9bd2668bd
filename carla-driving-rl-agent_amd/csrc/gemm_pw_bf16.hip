// Pointwise (1x1) convolution with bf16 activations and bf16 MFMA (gfx950: v_mfma_f32_32x32x16_bf16), float32 accumulate --
// the first kernel of BASELINE.json's configuration 3 ("same network bf16 with MFMA 1x1-conv/FC GEMMs, batch 1024").
//
// Reference: core/architectures.py:130,140 (Conv2D(k=1) -> BatchNormalization per time slice).  Storage contract of the bf16
// path: activations (A, C) are bf16 in HBM; weights, bias and every BatchNorm quantity stay float32 ("master") -- the weight
// fragments are rounded to bf16 once per workgroup on their way into registers; products accumulate in float32; the
// statistics of the following BatchNorm are taken from the ROUNDED outputs (what the consumers will read) in double.
// Same skeleton as the float32 kernel (gemm_pw.hip): persistent workgroups walking the row tiles of ONE BatchNorm group, W in
// registers for the whole kernel, A staged through LDS, optional BN-apply of the previous layer on load (PRO) and statistics
// partials of the next BatchNorm in the epilogue (EPI).  At K = N = 116 the float32 kernel spends 58 MFMA steps of 64 cycles
// per 32x32 tile (36 % MfmaUtil measured, profiles/r02_pmc_mfma.json); here it is 8 steps of 32 cycles.
#include <stdlib.h>

#include "cdrl_kernels.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct PwBf16Args {
    const __bf16* A;            // [G*Mg][lda] (+ a_coff)
    int lda, a_coff;
    const float* pro_stats;     // [4][G][K] or null: a = scale * a + shift on load
    const float* W;             // [K][N] float32 master weights (used when Wp == null)
    const __bf16* Wp;           // packed bf16 fragments [KP/16][2][128][8] (pw_bf16_pack) or null
    const float* bias;          // [N] or null
    __bf16* C;                  // [G*Mg][ldc] (+ c_coff)
    int ldc, c_coff;
    double* part;               // [G][nbpg][2][N] or null
    int N, K, G, Mg, nbpg;
};

// KP = K padded to a multiple of 16 (32, 64, 128); NT = 32-column tiles (1, 2, 4): one tile per wave, WR = 4/NT wave rows
template <int KP, int NT, bool PRO, bool EPI>
__global__ void __launch_bounds__(256) pw_bf16_kernel(PwBf16Args a) {
    constexpr int WC = NT, WR = 4 / WC, BM = 32 * WR, KS = KP / 16;
    constexpr int LDA = KP + 8;                      // bf16 elements; +16 bytes: ds_read_b128 of 16 rows hits 16 distinct slots
    constexpr int CPR = KP / 4;                      // 8-byte chunks (4 bf16) per row
    constexpr int NCH = BM * CPR / 256;              // chunks per thread per tile
    __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDA];
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

    // ---- W fragments -> registers: B[k = 16 s + 8 lk + e][n], rounded to bf16 once
    const int n = wc * 32 + lrow;
    bf16x8 breg[KS];
    if (a.Wp) {                 // pre-packed in fragment order: one 16-byte load per k step instead of eight strided float loads
#pragma unroll
        for (int s = 0; s < KS; ++s) breg[s] = *reinterpret_cast<const bf16x8*>(a.Wp + ((int64_t)(s * 2 + lk) * 128 + n) * 8);
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s + 8 * lk + e;
                breg[s][e] = (__bf16)((k < K && n < N) ? a.W[(int64_t)k * N + n] : 0.0f);
            }
    }
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

    // tile loads: chunk c = tid + 256 i -> row c / CPR, 4 bf16 at column 4 (c % CPR); 8-byte buffer loads, masked lanes read 0
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t OOR = 0x80000000u;
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.A), 0, (int)(Mtot * a.lda * 2), 0x00020000);
    u32x2_t ra[NCH];
    auto load_tile = [&](int t) {
        const int64_t m0 = mbeg + (int64_t)t * BM;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
            const bool ok = (m0 + r < mend) && (k4 < K);
            const uint32_t off = (uint32_t)(((m0 + r) * a.lda + a.a_coff + k4) * 2);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b64(rsA, ok ? off : OOR, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, r = c / CPR, k4 = 4 * (c % CPR);
            u32x2_t v = ra[i];
            if (PRO) {
                bf16x4 x = __builtin_bit_cast(bf16x4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (__bf16)fmaf(pc[0][k4 + e], (float)x[e], pc[1][k4 + e]);
                if (k4 >= K) x = bf16x4{(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};   // padded columns stay 0 (shift != 0)
                v = __builtin_bit_cast(u32x2_t, x);
            }
            *reinterpret_cast<u32x2_t*>(&As[r * LDA + k4]) = v;
        }
    };
    auto compute_tile = [&](int t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const __bf16* arow = &As[(wr * 32 + lrow) * LDA + 8 * lk];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, breg[s], acc, 0, 0, 0);
        }
        if (n >= N) return;
        const int64_t m0 = mbeg + (int64_t)t * BM + wr * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < mend) {
                const __bf16 o = (__bf16)(acc[r] + bv);
                if (EPI) {
                    const double v = (double)(float)o;
                    s1 += v;
                    s2 += v * v;
                }
                a.C[m * a.ldc + a.c_coff + n] = o;
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
        // lanes lk = 0, 1 hold different rows of the same column; wave rows wr likewise: fold in a fixed order
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

// W [K][N] float32 -> bf16 MFMA B fragments [KP/16][2][128][8]: element e of lane (lk, n) at k step s is W[16 s + 8 lk + e][n]
// (zero beyond K, N).  Run once per weight version (like the W^T copies of the float32 path), 8 KB .. 32 KB per conv.
__global__ void pw_bf16_pack_kernel(const float* __restrict__ W, int K, int N, int KP, __bf16* __restrict__ Wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;         // (s, lk, n)
    const int total = (KP / 16) * 2 * 128;
    if (i >= total) return;
    const int n = i % 128, lk = (i / 128) % 2, s = i / 256;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * lk + e;
        v[e] = (__bf16)((k < K && n < N) ? W[(int64_t)k * N + n] : 0.0f);
    }
    *reinterpret_cast<bf16x8*>(Wp + (int64_t)i * 8) = v;
}

static int pw_bf16_nbpg(int G, int Mg, int BM, int occ) {
    const int tiles = cdiv(Mg, BM);
    int nb = 256 * occ / G;
    if (nb < 1) nb = 1;
    if (nb > tiles) nb = tiles;
    return nb;
}

static inline int pw_bf16_kp(int K) { return K <= 32 ? 32 : (K <= 64 ? 64 : 128); }
static inline int pw_bf16_nt(int N) { return N <= 32 ? 1 : (N <= 64 ? 2 : 4); }

bool pw_bf16_supported(int lda, int a_coff, int N, int K) { return K >= 4 && K <= 128 && N >= 1 && N <= 128 && lda % 4 == 0 && a_coff % 4 == 0 && K % 4 == 0; }

static int pw_bf16_occ() {
    static const int v = 2;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
}

int pw_bf16_partial_rows(int G, int Mg, int N, int K) { return pw_bf16_nbpg(G, Mg, 32 * (4 / pw_bf16_nt(N)), pw_bf16_occ()); }

int64_t pw_bf16_packed_elems(int K) { return (int64_t)(pw_bf16_kp(K) / 16) * 2 * 128 * 8; }

int pw_bf16_pack(const float* W, int K, int N, void* Wp, hipStream_t st) {
    if (K < 1 || K > 128 || N < 1 || N > 128) {
        set_error("pw_bf16_pack: K, N <= 128");
        return -1;
    }
    const int kp = pw_bf16_kp(K), total = (kp / 16) * 2 * 128;
    hipLaunchKernelGGL(pw_bf16_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, W, K, N, kp, reinterpret_cast<__bf16*>(Wp));
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int KP, int NT>
static void pw_bf16_launch(const PwBf16Args& a, bool pro, bool epi, hipStream_t st) {
    const dim3 grid(a.G * a.nbpg), block(256);
    if (pro && epi) hipLaunchKernelGGL((pw_bf16_kernel<KP, NT, true, true>), grid, block, 0, st, a);
    else if (pro) hipLaunchKernelGGL((pw_bf16_kernel<KP, NT, true, false>), grid, block, 0, st, a);
    else if (epi) hipLaunchKernelGGL((pw_bf16_kernel<KP, NT, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_bf16_kernel<KP, NT, false, false>), grid, block, 0, st, a);
}

int pw_bf16(const void* A, int lda, int a_coff, const float* pro_stats, const float* W, const float* bias, void* C, int ldc,
            int c_coff, int G, int Mg, int N, int K, double* part, hipStream_t st, const void* Wp) {
    if (!pw_bf16_supported(lda, a_coff, N, K)) {
        set_error("pw_bf16: unsupported shape K=%d N=%d lda=%d coff=%d (K, N <= 128; K, lda, coff multiples of 4)", K, N, lda, a_coff);
        return -1;
    }
    if ((int64_t)G * Mg * lda * 2 >= (int64_t)1 << 31) {
        set_error("pw_bf16: operand larger than 2 GB");
        return -1;
    }
    PwBf16Args a;
    a.A = reinterpret_cast<const __bf16*>(A);
    a.lda = lda;
    a.a_coff = a_coff;
    a.pro_stats = pro_stats;
    a.W = W;
    a.Wp = reinterpret_cast<const __bf16*>(Wp);
    a.bias = bias;
    a.C = reinterpret_cast<__bf16*>(C);
    a.ldc = ldc;
    a.c_coff = c_coff;
    a.part = part;
    a.N = N;
    a.K = K;
    a.G = G;
    a.Mg = Mg;
    a.nbpg = pw_bf16_partial_rows(G, Mg, N, K);
    const int kp = pw_bf16_kp(K), nt = pw_bf16_nt(N);
    const bool pro = pro_stats != nullptr, epi = part != nullptr;
#define CDRL_PWB(KPV, NTV) pw_bf16_launch<KPV, NTV>(a, pro, epi, st)
    if (kp == 32) { if (nt == 1) CDRL_PWB(32, 1); else if (nt == 2) CDRL_PWB(32, 2); else CDRL_PWB(32, 4); }
    else if (kp == 64) { if (nt == 1) CDRL_PWB(64, 1); else if (nt == 2) CDRL_PWB(64, 2); else CDRL_PWB(64, 4); }
    else { if (nt == 1) CDRL_PWB(128, 1); else if (nt == 2) CDRL_PWB(128, 2); else CDRL_PWB(128, 4); }
#undef CDRL_PWB
    CDRL_LAUNCH_CHECK();
    return 0;
}

// float32 <-> bf16 (round to nearest even), 4 elements per thread
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4*>(y + i) = o;
    } else {
        for (int64_t j = i; j < n; ++j) y[j] = (__bf16)x[j];
    }
}

__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + i);
        *reinterpret_cast<float4*>(y + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
        for (int64_t j = i; j < n; ++j) y[j] = (float)x[j];
    }
}

int f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, x, reinterpret_cast<__bf16*>(y), n);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(x), y, n);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
