// Pointwise (1x1) convolution with 128 < K, N <= 256 -- the 232-channel convs of stage 2 (reference core/architectures.py:130,140,
// channel counts of ShuffleNet-V2 x1: 464 / 2) -- on the bf16 matrix pipe by exact three-way operand splitting (gfx950,
// v_mfma_f32_32x32x16_bf16), with the neighbouring BatchNorm work folded in like gemm_pw.hip.
//
// Why a second form: at K = 232 the register-resident-W design of gemm_pw.hip needs 128 VGPRs of float32 W fragments per wave, runs
// one workgroup of four waves per CU, covers N in two column blocks and spends 116 v_mfma_f32_32x32x2_f32 steps (64 cycles each) per
// 32-row tile: 26-42 us per launch for 23 MB of traffic at the benchmark shape (M = 12288), matrix-pipe- and latency-bound.  Here
//   * a workgroup of EIGHT waves owns a 64-row panel of A for ALL output columns: wave w computes columns 32 w .. 32 w + 31 for both
//     32-row tiles of the panel;
//   * the panel is loaded once (16-byte lanes), transformed by the prologue, split a = a1 + a2 + a3 exactly in bf16 and kept in
//     LDS whole (3 planes x 64 rows x K) -- the K loop has NO barrier;
//   * W is split and packed once per pass in MFMA fragment order (gemm_x3_pack, shared with gemm_x3.hip); a wave streams ITS column
//     tile's fragments from L2 four K = 16 steps ahead, no two waves of a workgroup read the same fragment;
//   * six MFMAs per (row tile, K step) carry the product to 2^-24 (smallest terms first), 12 MFMAs per 3 fragment loads;
//   * the next panel's rows are requested before the K loop of the current one.
// Prologues / epilogues (same contracts as gemm_pw.hip): PRO 1 BatchNorm-apply on load; EPI 1 statistics partials of the following
// BatchNorm, one row per workgroup, double, combined in fixed order by bn_finalize (bit-wise reproducible).
#include "colreduce.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pww_u32x4 __attribute__((ext_vector_type(4)));

struct PwwArgs {
    View A;
    const float* pro_stats;     // PRO 1: [4][G][K]
    const __bf16* Bp;           // [3][KS][2][NP][8] (gemm_x3_pack), NP = N padded to 128
    const float* bias;
    View C;
    double* part;               // EPI 1: [G][nbpg][2][N]
    int N, K, KS, NP, G, Mg, nbpg;
    int diag;       // CDRL_DIAG_PWW bits (timing experiments, wrong results): 1 no W fragment loads in the K loop, 2 no MFMA, 4 no epilogue, 8 no panel split / LDS writes
};

constexpr int PWW_NL = 8;      // 16-byte panel loads per thread

__device__ __forceinline__ void pww_split(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// KS = K steps of 16 (compile time: the LDS layout, the chunk -> (row, k) map and the plane offsets are constants; with run-time values
// the ~40 loop-invariant addresses were kept in registers and spilled)
// WV = waves per workgroup: 8 -> 64-row panels, wave w owns column tile w for both row tiles (one workgroup per CU: 95 KB of LDS);
//                           4 -> 32-row panels, wave w owns column tiles 2 w, 2 w + 1 (two workgroups per CU, whose phases -- panel
//                                load, split, K loop, epilogue -- overlap each other; twice the W fragment traffic per row)
template <int PRO, int EPI, int KS, int WV>
__global__ void __launch_bounds__(64 * WV, 2) pww_kernel(PwwArgs a) {
    constexpr int RT = WV / 4, CT = 8 / WV, PWW_BM = 32 * RT, NT_ = 64 * WV;
    constexpr int PWW_PF = WV == 8 ? 4 : 3;        // fragment prefetch depth (K steps)
    extern __shared__ __attribute__((aligned(16))) __bf16 As[];      // [3][64][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int g = blockIdx.x / a.nbpg, b = blockIdx.x % a.nbpg;
    const int K = a.K, N = a.N;
    constexpr int KP = KS * 16, LDA = KP + 8, C4 = KP / 4;          // row of 4-float chunks, zero-filled beyond K
    constexpr int plane_lds = PWW_BM * LDA;
    const int tiles_g = (a.Mg + PWW_BM - 1) / PWW_BM;
    const int t0 = (int)((int64_t)b * tiles_g / a.nbpg), t1 = (int)((int64_t)(b + 1) * tiles_g / a.nbpg);
    const int ncol = wave * 32 * CT + lrow;        // + 32 j
    bool col_ok[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) col_ok[j] = ncol + 32 * j < N;
    const int GK = a.G * K;
    const uint32_t OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.A.p, 0, (int)((int64_t)a.G * a.Mg * a.A.ld * 4), 0x00020000);
    const int64_t plane = (int64_t)KS * 2 * a.NP * 8;
    static_assert(PWW_BM * C4 <= NT_ * PWW_NL, "panel load plan");
    float* cf = reinterpret_cast<float*>(As + 3 * plane_lds);      // PRO 1: [2][KP] scale | shift of this group

    const __bf16* bcol = a.Bp + ((int64_t)lk * a.NP + ncol) * 8;       // + p * plane + ks * 2 * NP * 8
    auto load_b = [&](int ks, bf16x8 (&dst)[CT][3]) {
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[j][p] = *reinterpret_cast<const bf16x8*>(bcol + p * plane + ((int64_t)ks * 2 * a.NP + 32 * j) * 8);
    };
    // panel rows: chunk idx = tid + 512 i -> (row, 4-float chunk); every thread keeps the same chunk column set for all panels
    auto load_panel = [&](int t, pww_u32x4 (&ra)[PWW_NL]) {
        const int valid = min(PWW_BM, a.Mg - t * PWW_BM);
        const int64_t m0 = (int64_t)g * a.Mg + (int64_t)t * PWW_BM;
#pragma unroll
        for (int i = 0; i < PWW_NL; ++i) {
            const int idx = tid + NT_ * i;
            const int row = idx / C4, k = 4 * (idx - row * C4);
            const bool ok = row < valid && k < K;
            const uint32_t off = ok ? (uint32_t)(((m0 + row) * a.A.ld + a.A.coff + k) * 4) : OOR;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        }
    };
    auto store_panel = [&](const pww_u32x4 (&ra)[PWW_NL]) {
#pragma unroll
        for (int i = 0; i < PWW_NL; ++i) {
            const int idx = tid + NT_ * i;
            const int row = idx / C4, k = 4 * (idx - row * C4);
            if (row >= PWW_BM) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(ra[i][e]);
            if (PRO == 1 && k < K) {
                // (from LDS: read from memory the 16 coefficient registers per chunk are loop invariants that the compiler keeps live
                //  across all panels -- 64 VGPRs, spilled)
                const float4 sc = *reinterpret_cast<const float4*>(&cf[k]);
                const float4 sh = *reinterpret_cast<const float4*>(&cf[KP + k]);
                v[0] = fmaf(sc.x, v[0], sh.x);
                v[1] = fmaf(sc.y, v[1], sh.y);
                v[2] = fmaf(sc.z, v[2], sh.z);
                v[3] = fmaf(sc.w, v[3], sh.w);
                // (rows beyond the group's end hold zeros from the out-of-range loads and would become `shift`: their outputs are
                //  never stored and never enter the statistics, so they may hold anything)
            }
            bf16x4 h[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h1, h2, h3;
                pww_split(v[e], h1, h2, h3);
                h[0][e] = h1;
                h[1][e] = h2;
                h[2][e] = h3;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4*>(&As[p * plane_lds + row * LDA + k]) = h[p];
            __builtin_amdgcn_sched_barrier(0);      // chunk by chunk: interleaving the eight chunks' temporaries spilled 20-60 VGPRs
        }
    };
    if (PRO == 1) {
        for (int k = tid; k < K; k += NT_) {
            cf[k] = a.pro_stats[2 * GK + g * K + k];
            cf[KP + k] = a.pro_stats[3 * GK + g * K + k];
        }
    }
    float bv[CT];
    double s1[CT], s2[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        bv[j] = (a.bias && col_ok[j]) ? a.bias[ncol + 32 * j] : 0.0f;
        s1[j] = s2[j] = 0.0;
    }
    pww_u32x4 ra[PWW_NL];
    if (t0 < t1) load_panel(t0, ra);
    for (int t = t0; t < t1; ++t) {
        const int valid = min(PWW_BM, a.Mg - t * PWW_BM);
        const int64_t m0 = (int64_t)g * a.Mg + (int64_t)t * PWW_BM;
        bf16x8 bq[PWW_PF][CT][3];
#pragma unroll
        for (int u = 0; u < PWW_PF; ++u)
            if (u < KS) load_b(u, bq[u]);
        __syncthreads();                // the previous panel's fragment reads are done
        if (!(a.diag & 8)) store_panel(ra);
        if (t + 1 < t1) load_panel(t + 1, ra);
        __syncthreads();
        f32x16 acc[RT][CT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll 1
        for (int ks0 = 0; ks0 < KS; ks0 += PWW_PF) {
#pragma unroll
            for (int u = 0; u < PWW_PF; ++u) {
                const int ks = ks0 + u;
                if (ks >= KS) break;
                bf16x8 bf[CT][3];
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[j][p] = bq[u][j][p];
                if (ks + PWW_PF < KS && !(a.diag & 1)) load_b(ks + PWW_PF, bq[u]);
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    if (a.diag & 2) break;
                    bf16x8 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        af[p] = *reinterpret_cast<const bf16x8*>(&As[p * plane_lds + (i * 32 + lrow) * LDA + 16 * ks + 8 * lk]);
#pragma unroll
                    for (int j = 0; j < CT; ++j) {
                        f32x16 c = acc[i][j];          // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
            }
        }
        if (!(a.diag & 4)) {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (!col_ok[j]) continue;
                float* cp = a.C.p + a.C.coff + ncol + 32 * j;
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (row < valid) {
                            const float v = acc[i][j][r] + bv[j];
                            cp[(m0 + row) * a.C.ld] = v;
                            if (EPI == 1) {
                                s1[j] += (double)v;
                                s2[j] += (double)v * (double)v;
                            }
                        }
                    }
            }
        }
    }
    if (EPI == 1) {
        // the two lane halves hold different rows of the same column: lane < 32 adds its partner's sums and writes the row
        double* pr = a.part + ((int64_t)g * a.nbpg + b) * 2 * N;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const double t1_ = s1[j] + __shfl_xor(s1[j], 32), t2_ = s2[j] + __shfl_xor(s2[j], 32);
            if (lk == 0 && col_ok[j]) {
                pr[ncol + 32 * j] = t1_;
                pr[N + ncol + 32 * j] = t2_;
            }
        }
    }
}

bool pw_wide_supported(View A, View C, int N, int K) {
    (void)C;
    // K steps of 16: 15 (K = 228 .. 240, the 232-channel convs) and 16 (K = 244 .. 256) are instantiated
    return N > 128 && N <= 256 && K > 224 && K <= 256 && K % 4 == 0 && A.ld % 4 == 0 && A.coff % 4 == 0 && (reinterpret_cast<uintptr_t>(A.p) & 15) == 0;
}

// waves per workgroup: 8 (64-row panels, one workgroup per CU) | 4 (32-row panels, two per CU); CDRL_PWW_WAVES
static int pww_waves() {
    static const int wv = cdrl_getenv("CDRL_PWW_WAVES") ? atoi(cdrl_getenv("CDRL_PWW_WAVES")) : 4;
    return wv == 8 ? 8 : 4;
}

// workgroups per BatchNorm group: one panel each up to the number resident at once (256 or 512 on the chip), several panels each beyond
int pw_wide_nbpg(int G, int Mg, int N, int K) {
    (void)N;
    (void)K;
    const int wv = pww_waves();
    const int tiles_g = cdiv(Mg, 8 * wv);
    int target = (wv == 8 ? 256 : 512) / (G > 0 ? G : 1);
    if (target < 1) target = 1;
    return tiles_g < target ? tiles_g : target;
}

template <int PRO, int EPI, int KS, int WV>
static int launch_pww_ks(const PwwArgs& a, hipStream_t st) {
    const size_t lds = (size_t)3 * (8 * WV) * (KS * 16 + 8) * sizeof(__bf16) + (PRO == 1 ? (size_t)2 * KS * 16 * sizeof(float) : 0);
    auto kern = pww_kernel<PRO, EPI, KS, WV>;
    static size_t allowed = 64 * 1024;
    if (lds > allowed) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    hipLaunchKernelGGL(kern, dim3(a.G * a.nbpg), dim3(64 * WV), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int PRO, int EPI>
static int launch_pww(const PwwArgs& a, hipStream_t st) {
    if (pww_waves() == 8) return a.KS == 15 ? launch_pww_ks<PRO, EPI, 15, 8>(a, st) : launch_pww_ks<PRO, EPI, 16, 8>(a, st);
    return a.KS == 15 ? launch_pww_ks<PRO, EPI, 15, 4>(a, st) : launch_pww_ks<PRO, EPI, 16, 4>(a, st);
}

// C = (PRO ? scale * A + shift : A) W + bias;  part (or null): statistics partials, pw_wide_nbpg rows per group.
// Wp: gemm_x3_pack of B(k, n) = W (K x N).
int pw_wide(View A, const float* pro_stats, const void* Wp, const float* bias, View C, int G, int Mg, int N, int K, double* part,
            hipStream_t st) {
    if (G <= 0 || Mg <= 0) return 0;
    if (!pw_wide_supported(A, C, N, K) || !Wp) {
        set_error("pw_wide: unsupported shape / alignment N=%d K=%d ld=%d coff=%d", N, K, A.ld, A.coff);
        return -1;
    }
    if ((int64_t)G * Mg * A.ld * 4 >= (int64_t)1 << 31) {
        set_error("pw_wide: operand of 2 GB or more");
        return -1;
    }
    static const int diag = cdrl_getenv("CDRL_DIAG_PWW") ? atoi(cdrl_getenv("CDRL_DIAG_PWW")) : 0;
    PwwArgs a{A, pro_stats, reinterpret_cast<const __bf16*>(Wp), bias, C, part, N, K, cdiv(K, 16), cdiv(N, 128) * 128, G, Mg, pw_wide_nbpg(G, Mg, N, K), diag};
    if (pro_stats) return part ? launch_pww<1, 1>(a, st) : launch_pww<1, 0>(a, st);
    return part ? launch_pww<0, 1>(a, st) : launch_pww<0, 0>(a, st);
}

}  // namespace cdrl
