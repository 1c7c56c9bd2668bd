// fp32 MFMA GEMMs for gfx950 (CDNA4): v_mfma_f32_32x32x2_f32, 64-lane wavefronts.
//
// Used for every GEMM-shaped op of the learner (K4 pointwise convs, K8 dense layers, K10 GRU
// projections).  The f32-input MFMA is an exact k-ordered fmaf chain, so the
// results carry plain fp32 rounding (needed for the 1e-4 parity bar) at the matrix-core rate.
//
//   gemm_nn : C[M,N] (+)= A[M,K] * B(k,n) + bias     (forward, backward-data via strided B)
//   gemm_tn : W[K,N]  = sum_m A[m,K]^T D[m,N]        (backward-filter: gemm_tn_direct.hip; split over M, two-stage
//                                                      deterministic reduction, no atomics)
//
// Tiling: 256-thread workgroups = 4 wavefronts.  gemm_nn: 128 x (32*NT) output tile, every
// wave owns 32 rows x NT 32x32 accumulators; A/B staged through LDS in BK=32 slices (A tile
// padded to 33 floats per row -> conflict-free ds_read_b32 for the MFMA A fragment, B fragment
// reads are lane-consecutive).  Awkward channel counts (24/58/92/116/232/464) are zero-padded in
// LDS only; HBM tensors stay dense NHWC.
#include <stdlib.h>

#include "cdrl_kernels.h"
#include "pack_bodies.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32

// Software-pipelined K loop: the global loads of slice k+1 are issued into registers before the
// MFMAs of slice k run out of the other LDS buffer (one barrier per slice).  BT selects the lane
// order of the B staging so that the global reads stay coalesced for both W (sbn == 1, forward)
// and W^T (sbk == 1, backward-data); the B tile is padded to BN+1 floats per row so that both
// store orders are bank-conflict free.
// WR = wave rows: 4 -> 128-row tile, every wave owns 32 rows x all NT column tiles;
//                 2 -> 64-row tile, 2x2 wave grid, every wave owns 32 rows x NT/2 column tiles (more,
//                      smaller workgroups: balances the late, small-M layers over 256 CUs).
template <int NT, bool BT, bool VEC, int WR>
__global__ void __launch_bounds__(256, 2) gemm_nn_kernel(View A, const float* __restrict__ Bp, int sbk, int sbn,
                                                         const float* __restrict__ bias, View C, int M, int N, int Kfull,
                                                         int accumulate, int kchunk, int64_t cz_stride) {
    // split-K: blockIdx.z owns the K range [kb, kb + K) and writes its own dense partial slab (C.p + z * cz_stride)
    const int kb = blockIdx.z * kchunk;
    const int K = min(kchunk, Kfull - kb);
    A.coff += kb;
    Bp += (int64_t)kb * sbk;
    C.p += (int64_t)blockIdx.z * cz_stride;
    constexpr int BM = 32 * WR;
    constexpr int WC = 4 / WR;                   // wave columns
    constexpr int NTW = NT / WC;                 // column tiles per wave
    constexpr int BN = 32 * NT;
    constexpr int NA = (BM * BK) / 256;          // A elements per thread per slice
    constexpr int NB = (BK * BN) / 256;          // 4*NT B elements per thread per slice
    __shared__ float As[2][BM][BK + 1];
    __shared__ float Bs[2][BK][BN + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int lrow = lane & 31, lk = lane >> 5;
    float ra[NA], rb[NB];

    auto load_slice = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 4, kk = (idx & 15) * 2;
                const int64_t m = m0 + r;
                float2 v = make_float2(0.0f, 0.0f);
                if (m < M && (k0 + kk) < K) v = *reinterpret_cast<const float2*>(&A.p[m * A.ld + A.coff + k0 + kk]);
                ra[2 * i] = v.x;
                ra[2 * i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 5, kk = idx & 31;
                const int64_t m = m0 + r;
                ra[i] = (m < M && (k0 + kk) < K) ? A.p[m * A.ld + A.coff + k0 + kk] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            const int kk = BT ? (idx & (BK - 1)) : (idx / BN);
            const int nn = BT ? (idx / BK) : (idx % BN);
            rb[i] = ((k0 + kk) < K && (n0 + nn) < N) ? Bp[(int64_t)(k0 + kk) * sbk + (int64_t)(n0 + nn) * sbn] : 0.0f;
        }
    };
    auto store_slice = [&](int buf) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 4, kk = (idx & 15) * 2;
                As[buf][r][kk] = ra[2 * i];
                As[buf][r][kk + 1] = ra[2 * i + 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + 256 * i;
                As[buf][idx >> 5][idx & 31] = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i;
            const int kk = BT ? (idx & (BK - 1)) : (idx / BN);
            const int nn = BT ? (idx / BK) : (idx % BN);
            Bs[buf][kk][nn] = rb[i];
        }
    };

    load_slice(0);
    store_slice(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = (k0 + BK) < K;
        if (more) load_slice(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[buf][wr * 32 + lrow][kk + lk];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const float b = Bs[buf][kk + lk][(wc + j * WC) * 32 + lrow];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        if (more) store_slice(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = n0 + (wc + j * WC) * 32 + lrow;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < M) {
                float* c = &C.p[m * C.ld + C.coff + n];
                float v = acc[j][r] + bv;
                if (accumulate) v += *c;
                *c = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent "skinny" variant for K <= 128, N <= 128 (all stage-1/2 pointwise convs = most of the
// GEMM time): the whole weight panel W[K x BN] is loaded into LDS ONCE per workgroup, which then
// walks over 64-row tiles of A with a continuous register-prefetch pipeline across tile borders.
// Per slice a thread issues 4 float2 loads instead of 4 + 16 (A + B) -> far fewer VMEM/LDS
// instructions per MFMA, no per-tile prologue bubble, and the K tail (116 = 3*32 + 20) runs only
// the k-steps that exist.
// ------------------------------------------------------------------------------------------
template <int NT, bool BT, bool VEC>
__global__ void __launch_bounds__(256, 2) gemm_nn_persist_kernel(View A, const float* __restrict__ Bp, int sbk, int sbn,
                                                                 const float* __restrict__ bias, View C, int M, int N,
                                                                 int K, int accumulate, int ntiles) {
    constexpr int BMp = 64;
    constexpr int NTW = NT / 2;
    constexpr int BN = 32 * NT;
    constexpr int NA = (BMp * BK) / 256;      // 8
    extern __shared__ float smem[];
    const int Kp = (K + 1) & ~1;
    float* Ws = smem;                          // [Kp][BN + 1]
    float* As = smem + (size_t)Kp * (BN + 1);  // [2][64][33]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int lrow = lane & 31, lk = lane >> 5;
    for (int idx = tid; idx < Kp * BN; idx += 256) {
        const int kk = BT ? (idx % Kp) : (idx / BN);
        const int nn = BT ? (idx / Kp) : (idx % BN);
        Ws[kk * (BN + 1) + nn] = (kk < K && nn < N) ? Bp[(int64_t)kk * sbk + (int64_t)nn * sbn] : 0.0f;
    }
    const int nchunk = (K + BK - 1) / BK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * nchunk;
    float ra[NA];

    auto load_slice = [&](int step) {
        const int tile = blockIdx.x + (step / nchunk) * gridDim.x;
        const int k0 = (step % nchunk) * BK;
        const int64_t m0 = (int64_t)tile * BMp;
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 4, kk = (idx & 15) * 2;
                const int64_t m = m0 + r;
                float2 v = make_float2(0.0f, 0.0f);
                if (m < M && (k0 + kk) < K) v = *reinterpret_cast<const float2*>(&A.p[m * A.ld + A.coff + k0 + kk]);
                ra[2 * i] = v.x;
                ra[2 * i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 5, kk = idx & 31;
                const int64_t m = m0 + r;
                ra[i] = (m < M && (k0 + kk) < K) ? A.p[m * A.ld + A.coff + k0 + kk] : 0.0f;
            }
        }
    };
    auto store_slice = [&](int buf) {
        float* Ab = As + buf * (BMp * (BK + 1));
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 4, kk = (idx & 15) * 2;
                Ab[r * (BK + 1) + kk] = ra[2 * i];
                Ab[r * (BK + 1) + kk + 1] = ra[2 * i + 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int idx = tid + 256 * i;
                Ab[(idx >> 5) * (BK + 1) + (idx & 31)] = ra[i];
            }
        }
    };

    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    if (nsteps > 0) {
        load_slice(0);
        store_slice(0);
    }
    __syncthreads();
    int buf = 0;
    for (int step = 0; step < nsteps; ++step) {
        const bool more = (step + 1) < nsteps;
        if (more) load_slice(step + 1);
        const int chunk = step % nchunk;
        const int k0 = chunk * BK;
        const int kmax = min(BK, Kp - k0);
        const float* Ab = As + buf * (BMp * (BK + 1)) + (wr * 32 + lrow) * (BK + 1) + lk;
        const float* Wb = Ws + (size_t)(k0 + lk) * (BN + 1) + wc * 32 + lrow;
#pragma unroll 4
        for (int kk = 0; kk < kmax; kk += 2) {
            const float a = Ab[kk];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const float b = Wb[kk * (BN + 1) + j * 64];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        if (chunk == nchunk - 1) {      // tile finished: epilogue, then reset the accumulators
            const int tile = blockIdx.x + (step / nchunk) * gridDim.x;
            const int64_t m0 = (int64_t)tile * BMp;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = (wc + j * 2) * 32 + lrow;
                if (n < N) {
                    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m < M) {
                            float* c = &C.p[m * C.ld + C.coff + n];
                            float v = acc[j][r] + bv;
                            if (accumulate) v += *c;
                            *c = v;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            }
        }
        if (more) store_slice(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

template <int NT>
static int launch_nn_persist(bool bt, bool vec, hipStream_t st, View A, const float* Bp, int sbk, int sbn, const float* bias,
                             View C, int M, int N, int K, int acc) {
    const int Kp = (K + 1) & ~1;
    const size_t lds = ((size_t)Kp * (32 * NT + 1) + 2 * 64 * (BK + 1)) * sizeof(float);
    const int ntiles = cdiv(M, 64);
    int grid = 512;                 // 2 resident workgroups per CU
    if (grid > ntiles) grid = ntiles;
#define CDRL_PERSIST(BTv, VECv)                                                                                         \
    do {                                                                                                                \
        static LdsAttrOnce attr_done;                                                                                   \
        if (attr_done.need()) {                                                                                               \
            CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nn_persist_kernel<NT, BTv, VECv>),           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                       \
            attr_done.mark();                                                                                             \
        }                                                                                                               \
        hipLaunchKernelGGL((gemm_nn_persist_kernel<NT, BTv, VECv>), dim3(grid), dim3(256), lds, st, A, Bp, sbk, sbn, bias, \
                           C, M, N, K, acc, ntiles);                                                                    \
    } while (0)
    if (bt) {
        if (vec) CDRL_PERSIST(true, true);
        else CDRL_PERSIST(true, false);
    } else {
        if (vec) CDRL_PERSIST(false, true);
        else CDRL_PERSIST(false, false);
    }
#undef CDRL_PERSIST
    return 0;
}

// ------------------------------------------------------------------------------------------
// Row-stacked variant for the wide late layers (stage-3 units K = N = 232, head conv 464 -> 768): a (32*RT) x 128
// tile, wave w owns column tile w and RT row tiles, so one B fragment feeds RT MFMAs (LDS reads per MFMA: (RT+1)/RT
// instead of 3/2) and M = 12288 splits into exactly 256 workgroups at RT = 3 (the 64 x 128 tiling gave 384 = 1.5
// rounds over 256 CUs: 33 us for 8.4 us of MFMA time).  A and B slices are fetched with 16-byte loads (the generic
// kernel issues 16 scalar B loads per thread and slice), the K tail (232 = 7*32 + 8) runs only the k-steps that exist.
// Requires K % 4 == 0, 16-byte aligned A rows and weight rows (N % 4 == 0 forward, K % 4 == 0 transposed).
// ------------------------------------------------------------------------------------------
template <int RT, bool BT>
__global__ void __launch_bounds__(256, 2) gemm_nn_rt_kernel(View A, const float* __restrict__ Bp, int sbk, int sbn,
                                                            const float* __restrict__ bias, View C, int M, int N, int K,
                                                            int accumulate) {
    constexpr int BM = 32 * RT, BN = 128;
    constexpr int NA4 = (BM * BK / 4) / 256;     // float4 loads of A per thread and slice (RT)
    constexpr int NB4 = (BK * BN / 4) / 256;     // 4
    __shared__ float As[2][BM][BK + 1];
    __shared__ float Bs[2][BK][BN + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    // Three-stage pipeline: slice s is multiplied out of LDS while slice s+1 waits in one register set and the global
    // loads of slice s+2 are issued into the other.  With the loads issued only one slice ahead (1.3 us of MFMA work) every
    // slice waited ~1 us for HBM at one workgroup per CU: 29 us for the 12288 x 232 x 232 GEMM whose MFMA time is 9 us.
    float4 ra[2][NA4], rb[2][NB4];
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    auto load_slice = [&](int k0, float4* qa, float4* qb) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, kk = (idx & 7) * 4;
            const int64_t m = m0 + r;
            qa[i] = (m < M && (k0 + kk) < K) ? *reinterpret_cast<const float4*>(&A.p[m * A.ld + A.coff + k0 + kk]) : z4;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int idx = tid + 256 * i;
            if (BT) {       // W^T: contiguous along k
                const int kk = (idx & 7) * 4, nn = idx >> 3;
                qb[i] = ((k0 + kk) < K && (n0 + nn) < N)
                            ? *reinterpret_cast<const float4*>(&Bp[(int64_t)(k0 + kk) * sbk + (int64_t)(n0 + nn) * sbn]) : z4;
            } else {        // W: contiguous along n
                const int kk = idx >> 5, nn = (idx & 31) * 4;
                qb[i] = ((k0 + kk) < K && (n0 + nn) < N)
                            ? *reinterpret_cast<const float4*>(&Bp[(int64_t)(k0 + kk) * sbk + (int64_t)(n0 + nn) * sbn]) : z4;
            }
        }
    };
    auto store_slice = [&](int buf, const float4* qa, const float4* qb) {
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, kk = (idx & 7) * 4;
            As[buf][r][kk] = qa[i].x;
            As[buf][r][kk + 1] = qa[i].y;
            As[buf][r][kk + 2] = qa[i].z;
            As[buf][r][kk + 3] = qa[i].w;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int idx = tid + 256 * i;
            if (BT) {
                const int kk = (idx & 7) * 4, nn = idx >> 3;
                Bs[buf][kk][nn] = qb[i].x;
                Bs[buf][kk + 1][nn] = qb[i].y;
                Bs[buf][kk + 2][nn] = qb[i].z;
                Bs[buf][kk + 3][nn] = qb[i].w;
            } else {
                const int kk = idx >> 5, nn = (idx & 31) * 4;
                Bs[buf][kk][nn] = qb[i].x;
                Bs[buf][kk][nn + 1] = qb[i].y;
                Bs[buf][kk][nn + 2] = qb[i].z;
                Bs[buf][kk][nn + 3] = qb[i].w;
            }
        }
    };
    auto mma_slice = [&](int buf, int k0) {
        const int kmax = min(BK, K - k0);        // K % 4 == 0 -> even
        if (kmax == BK) {
#pragma unroll
            for (int h = 0; h < BK; h += 16) {
                float bf[8], af[8][RT];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    bf[q] = Bs[buf][h + 2 * q + lk][wave * 32 + lrow];
#pragma unroll
                    for (int i = 0; i < RT; ++i) af[q][i] = As[buf][i * 32 + lrow][h + 2 * q + lk];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][i], bf[q], acc[i], 0, 0, 0);
            }
        } else {
            for (int kk = 0; kk < kmax; kk += 2) {
                const float b = Bs[buf][kk + lk][wave * 32 + lrow];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const float a = As[buf][i * 32 + lrow][kk + lk];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                }
            }
        }
    };

    load_slice(0, ra[0], rb[0]);
    if (BK < K) load_slice(BK, ra[1], rb[1]);
    store_slice(0, ra[0], rb[0]);
    __syncthreads();
    // invariant at the top of step s (k0 = s*BK): LDS[s & 1] holds slice s, register set (s+1) & 1 holds slice s+1
    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
        // even step: LDS 0, next slice in set 1, fetch slice s+2 into set 0
        if (k0 + 2 * BK < K) load_slice(k0 + 2 * BK, ra[0], rb[0]);
        mma_slice(0, k0);
        if (k0 + BK < K) store_slice(1, ra[1], rb[1]);
        __syncthreads();
        if (k0 + BK >= K) break;
        // odd step: LDS 1, next slice in set 0, fetch slice s+2 into set 1
        if (k0 + 3 * BK < K) load_slice(k0 + 3 * BK, ra[1], rb[1]);
        mma_slice(1, k0 + BK);
        if (k0 + 2 * BK < K) store_slice(0, ra[0], rb[0]);
        __syncthreads();
    }
    const int n = n0 + wave * 32 + lrow;
    if (n >= N) return;
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < M) {
                float* c = &C.p[m * C.ld + C.coff + n];
                float v = acc[i][r] + bv;
                if (accumulate) v += *c;
                *c = v;
            }
        }
    }
}

template <int RT>
static void launch_nn_rt(bool bt, hipStream_t st, View A, const float* Bp, int sbk, int sbn, const float* bias, View C, int M,
                         int N, int K, int acc) {
    dim3 grid(cdiv(M, 32 * RT), cdiv(N, 128));
    if (bt) hipLaunchKernelGGL((gemm_nn_rt_kernel<RT, true>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc);
    else hipLaunchKernelGGL((gemm_nn_rt_kernel<RT, false>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc);
}

template <int NT, int WR>
static void launch_nn(bool bt, bool vec, dim3 grid, hipStream_t st, View A, const float* Bp, int sbk, int sbn,
                      const float* bias, View C, int M, int N, int K, int acc, int kchunk = 1 << 30, int64_t czs = 0) {
    if (bt) {
        if (vec) hipLaunchKernelGGL((gemm_nn_kernel<NT, true, true, WR>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc, kchunk, czs);
        else hipLaunchKernelGGL((gemm_nn_kernel<NT, true, false, WR>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc, kchunk, czs);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_nn_kernel<NT, false, true, WR>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc, kchunk, czs);
        else hipLaunchKernelGGL((gemm_nn_kernel<NT, false, false, WR>), grid, dim3(256), 0, st, A, Bp, sbk, sbn, bias, C, M, N, K, acc, kchunk, czs);
    }
}

static int g_nn_bm64_threshold = -1;

// C(view) (+)= bias + sum_z part[z]  (fixed order)
// act_out (dense [M][N], optional): the layer's activation of the reduced value as well -- same expressions as act_fwd_kernel (bn.hip),
// which was a launch of its own behind every split-K Dense layer
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, int M, int N, const float* __restrict__ bias,
                                     View C, int accumulate, float* __restrict__ act_out, int act) {
    const int64_t n = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N;
        const int c = (int)(i - m * N);
        float s = bias ? bias[c] : 0.0f;
        for (int z = 0; z < nsplit; ++z) s += part[(int64_t)z * n + i];
        float* o = &C.p[m * C.ld + C.coff + c];
        const float x = accumulate ? *o + s : s;
        *o = x;
        if (act_out) {
            float a = x;
            if (act == ACT_RELU6) a = fminf(fmaxf(x, 0.0f), 6.0f);
            else if (act == ACT_SWISH6) a = fminf(x * (1.0f / (1.0f + expf(-x))), 6.0f);
            else if (act == ACT_TANH) a = tanhf(x);
            else if (act == ACT_SIGMOID) a = 1.0f / (1.0f + expf(-x));
            act_out[i] = a;
        }
    }
}

int64_t gemm_nn_splitk_elems(int M, int N, int K) {
    if (M > 2048 || K < 256) return 0;
    return (int64_t)8 * M * N;
}

int gemm_nn(View A, const float* Bp, int sbk, int sbn, const float* bias, View C, int M, int N, int K,
            int accumulate, hipStream_t st, float* splitk_ws, float* act_out, int act, bool* act_done) {
    if (act_done) *act_done = false;
    if (M <= 0 || N <= 0) return 0;
    if (splitk_ws && gemm_nn_splitk_elems(M, N, K) > 0) {
        // few output tiles, long reduction (GRU projections: M = B or T*B, K = 256 / 768): split K over blockIdx.z so that
        // the chip is filled, partial slabs + fixed-order reduction (the 4x2-workgroup launch took 60 us for 0.1 GFLOP)
        int nt = cdiv(N, 32);
        if (nt > 4) nt = 4;
        const int ncb = cdiv(N, 32 * nt);
        nt = cdiv(cdiv(N, ncb), 32);
        if (nt % 2 == 0) {
            const int gy = cdiv(N, 32 * nt), gx = cdiv(M, 64);
            int nsplit = cdiv(512, gx * gy);
            if (nsplit > 8) nsplit = 8;
            int kchunk = cdiv(cdiv(K, nsplit), 32) * 32;
            nsplit = cdiv(K, kchunk);
            if (nsplit > 1) {
                const bool bt = sbn != 1;
                const bool vec = (K % 2 == 0) && (A.ld % 2 == 0) && (A.coff % 2 == 0) && ((reinterpret_cast<uintptr_t>(A.p) & 7) == 0);
                dim3 grid(gx, gy, nsplit);
                View P = make_view(splitk_ws, N);
                if (nt == 2) launch_nn<2, 2>(bt, vec, grid, st, A, Bp, sbk, sbn, nullptr, P, M, N, K, 0, kchunk, (int64_t)M * N);
                else launch_nn<4, 2>(bt, vec, grid, st, A, Bp, sbk, sbn, nullptr, P, M, N, K, 0, kchunk, (int64_t)M * N);
                CDRL_LAUNCH_CHECK();
                const int64_t n = (int64_t)M * N;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st,
                                   splitk_ws, nsplit, M, N, bias, C, accumulate, act_done ? act_out : nullptr, act);
                CDRL_LAUNCH_CHECK();
                if (act_done && act_out) *act_done = true;
                return 0;
            }
        }
    }
    if (g_nn_bm64_threshold < 0) {
        g_nn_bm64_threshold = 1 << 30;   // use 64-row tiles below this many 128-row blocks; measured: 64-row tiles win at every learner shape
    }
    int nt = cdiv(N, 32);
    if (nt > 4) nt = 4;
    // balance column blocks: e.g. N=232 -> 2 blocks of 4 tiles; N=92 -> 1 block of 3 tiles
    const int ncb = cdiv(N, 32 * nt);
    nt = cdiv(cdiv(N, ncb), 32);
    const int gy = cdiv(N, 32 * nt);
    const bool bt = sbn != 1;
    const bool vec = (K % 2 == 0) && (A.ld % 2 == 0) && (A.coff % 2 == 0) && ((reinterpret_cast<uintptr_t>(A.p) & 7) == 0);
    {   // row-stacked tiles for the wide late layers
        static const int rt = 3;
        static const int rt_minn = 129;
        const bool a16 = (K % 4 == 0) && (A.ld % 4 == 0) && (A.coff % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.p) & 15) == 0);
        const bool b16 = ((reinterpret_cast<uintptr_t>(Bp) & 15) == 0) &&
                         (bt ? (sbk == 1 && sbn % 4 == 0) : (sbn == 1 && sbk % 4 == 0 && N % 4 == 0));
        if (rt >= 2 && rt <= 4 && N >= rt_minn && K >= 64 && M >= 2048 && a16 && b16) {
            if (rt == 2) launch_nn_rt<2>(bt, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate);
            else if (rt == 3) launch_nn_rt<3>(bt, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate);
            else launch_nn_rt<4>(bt, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate);
            CDRL_LAUNCH_CHECK();
            return 0;
        }
    }
    static int persist = -1;
    if (persist < 0) {
        // measured at B=256 (768 tiles of 64 rows over 512 resident workgroups): 40.8 us vs 26.7 us for the
        // non-persistent 64-row kernel at M=49152, K=N=116 -- too few tiles per workgroup to amortise the
        // weight panel, and 1.5 tiles/workgroup is badly balanced.  Off by default; worth re-testing at B>=1024.
        persist = 0;
    }
    if (persist && gy == 1 && K <= 128 && (nt == 2 || nt == 4) && M >= 4096) {
        if (nt == 2) CDRL_TRY(launch_nn_persist<2>(bt, vec, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate));
        else CDRL_TRY(launch_nn_persist<4>(bt, vec, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate));
        CDRL_LAUNCH_CHECK();
        return 0;
    }
    const bool small = (nt % 2 == 0) && (cdiv(M, 128) * gy < g_nn_bm64_threshold);
    if (small) {
        dim3 grid(cdiv(M, 64), gy);
        if (nt == 2) launch_nn<2, 2>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate);
        else launch_nn<4, 2>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate);
    } else {
        dim3 grid(cdiv(M, 128), gy);
        switch (nt) {
            case 1: launch_nn<1, 4>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate); break;
            case 2: launch_nn<2, 4>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate); break;
            case 3: launch_nn<3, 4>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate); break;
            default: launch_nn<4, 4>(bt, vec, grid, st, A, Bp, sbk, sbn, bias, C, M, N, K, accumulate); break;
        }
    }
    CDRL_LAUNCH_CHECK();
    return 0;
}

// (the filter-gradient GEMM lives in gemm_tn_direct.hip; its split-M partials are reduced here)
// (CX output lanes) x (1024/CX split lanes) per block: the split partials are summed in parallel and combined in a
// fixed order (deterministic).  CX = 4 when there are few outputs (see reduce_partials_kernel in bn.hip).
template <int CX>
__global__ void __launch_bounds__(1024) tn_reduce_kernel(const float* __restrict__ part, int nsplit, int64_t n,
                                                         int64_t stride, float* __restrict__ out, int accumulate) {
    constexpr int PY = 1024 / CX;
    __shared__ double sm[PY][CX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t i = (int64_t)blockIdx.x * CX + tx;
    double s = 0.0;
    if (i < n) {
        for (int p0 = ty; p0 < nsplit; p0 += PY * 8) {       // 8 independent loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + u * PY;
                v[u] = p < nsplit ? part[(int64_t)p * stride + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (PY > 64) {
        if (ty < 64) {
            double a = 0.0;
            for (int y = ty; y < PY; y += 64) a += sm[y][tx];
            sm[ty][tx] = a;
        }
        __syncthreads();
    }
    if (i < n && ty == 0) {
        s = 0.0;
#pragma unroll
        for (int y = 0; y < 64; ++y) s += sm[y][tx];
        out[i] = accumulate ? out[i] + (float)s : (float)s;
    }
}

// float4 variant: TX lanes x 4 consecutive outputs (16-byte loads, 256-byte row segments at TX = 16) x 256/TX split
// lanes, 8 independent 16-byte loads in flight per thread.  The CX = 16 scalar kernel above read 64-byte segments with
// 1024-thread blocks: 31 us for the 384 x 116 x 116 partials of a stage-1 filter gradient (0.66 TB/s).
template <int TX>
__global__ void __launch_bounds__(256) tn_reduce_v4_kernel(const float* __restrict__ part, int nsplit, int64_t n,
                                                           int64_t stride, float* __restrict__ out, int accumulate) {
    constexpr int TY = 256 / TX;
    __shared__ double sm[TY][TX][4];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int64_t i = ((int64_t)blockIdx.x * TX + tx) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        const float* src = part + i;
        for (int p0 = ty; p0 < nsplit; p0 += TY * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // clamped address + select AFTER the load: a conditional load becomes a branch with its own s_waitcnt
                // (the 8 loads were issued one at a time)
                const int p = p0 + u * TY;
                v[u] = *reinterpret_cast<const float4*>(src + (int64_t)min(p, nsplit - 1) * stride);
                if (p >= nsplit) v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += (double)v[u].x;
                s1 += (double)v[u].y;
                s2 += (double)v[u].z;
                s3 += (double)v[u].w;
            }
        }
    }
    sm[ty][tx][0] = s0;
    sm[ty][tx][1] = s1;
    sm[ty][tx][2] = s2;
    sm[ty][tx][3] = s3;
    __syncthreads();
    // fixed-order fold over the TY split lanes: thread (c, tx) of the first 4*TX threads owns output i + c
    if (threadIdx.x < 4 * TX) {
        const int c = threadIdx.x % 4, x = threadIdx.x / 4;
        const int64_t o = ((int64_t)blockIdx.x * TX + x) * 4 + c;
        if (o < n) {
            double s = 0.0;
#pragma unroll 8
            for (int y = 0; y < TY; ++y) s += sm[y][x][c];
            out[o] = accumulate ? out[o] + (float)s : (float)s;
        }
    }
}

// few partials (batch-sized GEMMs: GRU / dense weight gradients with 1-8 row splits): one thread per 4 outputs walks
// the partials in order.  The lane-parallel kernels above spend 1024 threads per 16 outputs on 4 partials: 97 us for
// the 768 x 768 GRU kernel gradient.
__global__ void __launch_bounds__(256) tn_reduce_few_kernel(const float* __restrict__ part, int nsplit, int64_t n,
                                                            int64_t stride, float* __restrict__ out, int accumulate) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int p0 = 0; p0 < nsplit; p0 += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            v[u] = *reinterpret_cast<const float4*>(part + (int64_t)min(p0 + u, nsplit - 1) * stride + i);
            if (p0 + u >= nsplit) v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0 += (double)v[u].x;
            s1 += (double)v[u].y;
            s2 += (double)v[u].z;
            s3 += (double)v[u].w;
        }
    }
    float4 o = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
    if (accumulate) {
        o.x += out[i];
        o.y += out[i + 1];
        o.z += out[i + 2];
        o.w += out[i + 3];
    }
    out[i] = o.x;
    out[i + 1] = o.y;
    out[i + 2] = o.z;
    out[i + 3] = o.w;
}

int reduce_partials_f32(const float* part, int nparts, int64_t n, int64_t stride, float* out, int accumulate,
                        hipStream_t st) {
    static const bool v4 = true;
    if (v4 && n % 4 == 0 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0 && nparts < 16 && n >= 4096) {
        hipLaunchKernelGGL(tn_reduce_few_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, part, nparts, n, stride, out, accumulate);
        CDRL_LAUNCH_CHECK();
        return 0;
    }
    if (v4 && n % 4 == 0 && stride % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0 && nparts >= 16) {
        if (cdiv64(n, 64) >= 160)
            hipLaunchKernelGGL(tn_reduce_v4_kernel<16>, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, st, part, nparts, n, stride, out, accumulate);
        else if (cdiv64(n, 16) >= 128 || nparts < 128)
            hipLaunchKernelGGL(tn_reduce_v4_kernel<4>, dim3((unsigned)cdiv64(n, 16)), dim3(256), 0, st, part, nparts, n, stride, out, accumulate);
        else
            hipLaunchKernelGGL(tn_reduce_v4_kernel<1>, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, st, part, nparts, n, stride, out, accumulate);
        CDRL_LAUNCH_CHECK();
        return 0;
    }
    if (cdiv64(n, 16) >= 128 || nparts <= 64)
        hipLaunchKernelGGL(tn_reduce_kernel<16>, dim3((unsigned)cdiv64(n, 16)), dim3(16, 64), 0, st, part, nparts, n, stride, out, accumulate);
    else
        hipLaunchKernelGGL(tn_reduce_kernel<4>, dim3((unsigned)cdiv64(n, 4)), dim3(4, 256), 0, st, part, nparts, n, stride, out, accumulate);
    CDRL_LAUNCH_CHECK();
    return 0;
}

// one 32 x 32 tile per block through LDS; grid = (tiles of the largest matrix, matrices)
__global__ void __launch_bounds__(256) transpose_many_kernel(const PwTranspose* __restrict__ tab) {
    __shared__ float t[32][33];
    const PwTranspose d = tab[blockIdx.y];
    transpose_body(d, blockIdx.x, t);
}

int transpose_many(const PwTranspose* tab_dev, int n, int tiles, hipStream_t st) {
    if (n <= 0 || tiles <= 0) return 0;
    hipLaunchKernelGGL(transpose_many_kernel, dim3(tiles, n), dim3(256), 0, st, tab_dev);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
