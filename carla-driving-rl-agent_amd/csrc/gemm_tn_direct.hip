// Filter-gradient GEMM  W[K,N] = sum_m A[m,K]^T D[m,N]  with MFMA operands loaded STRAIGHT from global memory
// (gfx950, v_mfma_f32_32x32x2_f32).
//
// For the transposed-A product the MFMA operand layout IS the row-major memory layout: the "A" fragment of one
// 32x32x2 step wants lane (k = lane%32, h = lane/32) to hold A[m0+h][k0+k], the "B" fragment wants D[m0+h][n0+n] --
// consecutive lanes read consecutive floats of one row (128-byte coalesced segments), two rows per instruction.  So
// there is no LDS staging, no barrier and no bank conflict: every wave is an independent stream of
//     (1 + NJ) coalesced loads  ->  NJ MFMAs           per pair of rows,
// with U row pairs of loads in flight per lane.  The LDS-tiled gemm_tn it replaces was a chain of load -> LDS ->
// barrier -> MFMA steps (~8 us per 32 rows for 0.4 us of MFMA work) on < 1 workgroup per CU: 78-100 us for the
// 49152 x 116 x 116 filter gradients that cost ~10 us of HBM time.
//
// Work split: grid = (row splits, 128-blocks of K, 128-blocks of N).  Inside a workgroup the 4 waves are spread over
// the k tiles first, then over the n tiles, and whatever factor is left splits the workgroup's rows (those waves write
// separate partial slots).  Partials are float [slot][K][N], summed in fixed order by tn_reduce_kernel (deterministic).
// Optional operand prologues, same formulas as gemm_pw.hip:
//   A <- scale[g][k]*A + shift[g][k]                    (BatchNorm apply of the layer that produced A's raw values)
//   D <- k1*(dz - k2 - xhat*k3), dz = D (shuffle-gathered, ReLU6-masked), xhat from the BN's raw input y
// (both need the row splits aligned to the G BatchNorm groups, which the plan guarantees).
// BF variants (bf16-operand compute mode, configuration 3): the contraction runs over the ROWS, so one v_mfma_f32_32x32x16_bf16
// step consumes 16 rows: lane (column = lane%32, h = lane/32) holds rows m0 + 8h .. m0 + 8h + 7 of its column -- the same
// 2-rows-per-load-instruction access pattern with the second half-wave 8 rows (instead of 1 row) further down, the same
// operand prologues in float32, then one round-to-nearest-even pack of 8 values per operand and 1/16 of the matrix-pipe time.
#include <stdlib.h>

#include <algorithm>

#include "colreduce.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct TnDirectArgs {
    View A, D;
    float* part;
    int M, N, K, G, Mg;
    int rows_per, nspg;         // rows per workgroup, workgroups per group
    int WK, NSPL, NJW, RS;      // wave mapping: waves along k, n-splits, n tiles per wave, row splits
    int TR;                     // transposed wave mapping (tn_direct_tr_kernel)
    int RS2;                    // 2: a second set of 4 waves takes the other half of every wave's rows; the two halves
                                // are added through LDS before the partial is written (half the split-M partials)
    const float* a_stats;       // [4][G][K] or null
    TnBnBwd db;                 // y == null: no D prologue
};


// MODE 0: float32; 1: BF (bf16 MFMA operands, float32 tensors); 2: BF + bf16 activation storage (A, D and the BN input y of the
// D prologue are bf16 in HBM: 2-byte buffer loads per lane, widened; prologues and partials unchanged)
template <int MODE>
__device__ __forceinline__ float tnd_ld(const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t so) {
    if (MODE == 2) return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, 0) << 16);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
}

template <int NJW, bool APRO, bool DPRO, int U, int MODE>
__global__ void __launch_bounds__(512) tn_direct_kernel(TnDirectArgs a) {
    constexpr bool BF = MODE >= 1;
    constexpr uint32_t ESZ = MODE == 2 ? 2u : 4u;
    static_assert(!BF || U == 8, "BF: one batch = one K = 16 MFMA step (8 rows per half-wave)");
    constexpr int LHR = BF ? 8 : 1;             // rows between the two half-waves of a load
    extern __shared__ float tnd_red[];          // RS2 == 2: [4 waves][NJW][16][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8;
    const int l32 = lane & 31, lh = lane >> 5;
    const int K = a.K, N = a.N;
    const int k0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
    const int KT = min(4, (K - k0 + 31) / 32), NT = min(4, (N - n0 + 31) / 32);
    const int wk = wave % a.WK, rest = wave / a.WK;
    const int nsp = rest % a.NSPL, rs = rest / a.NSPL;
    const int ki = wk, nj0 = nsp * NJW;
    const int njn = min(NJW, NT - nj0);
    // idle waves: a 4th wave for 3 k tiles, fewer n tiles than splits, or a leftover wave when the waves that remain
    // after the k tiles do not divide evenly into n splits x row splits (e.g. 4 waves over 3 n tiles: RS = 1)
    const bool active = !(ki >= KT || njn <= 0 || rs >= a.RS);
    const int grp = blockIdx.x / a.nspg;
    const int64_t gend = (int64_t)(grp + 1) * a.Mg;
    int64_t mbeg = (int64_t)grp * a.Mg + (int64_t)(blockIdx.x % a.nspg) * a.rows_per;
    int64_t mend = mbeg + a.rows_per;
    if (mend > gend) mend = gend;
    {   // this wave's share of the workgroup's rows (even number of rows per share)
        const int64_t len = mend - mbeg;
        const int TS = a.RS * a.RS2;
        const int64_t per = ((len + TS - 1) / TS + 1) / 2 * 2;
        mbeg += (rs * a.RS2 + half) * per;
        if (mbeg + per < mend) mend = mbeg + per;
        if (!active) mend = mbeg;
    }
    const int k = k0 + ki * 32 + l32;
    const bool kon = k < K;
    float asc = 1.0f, ash = 0.0f;
    if (APRO && kon) {
        asc = a.a_stats[2 * a.G * K + grp * K + k];
        ash = a.a_stats[3 * a.G * K + grp * K + k];
    }
    int ncol[NJW], dcol[NJW];
    bool non[NJW];
    float qm[NJW], qi[NJW], qsc[NJW], qsh[NJW], qk1[NJW], qk2[NJW], qk3[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        const int n = n0 + (nj0 + j) * 32 + l32;
        ncol[j] = n;
        non[j] = j < njn && n < N;
        dcol[j] = a.D.coff + n;
        if (DPRO) {
            if (a.db.shuffle_ctot) dcol[j] = shuffle_dst(dcol[j], a.db.shuffle_ctot);
            const int GN = a.G * N, o = grp * N + n;
            qm[j] = non[j] ? a.db.stats[0 * GN + o] : 0.0f;
            qi[j] = non[j] ? a.db.stats[1 * GN + o] : 0.0f;
            qsc[j] = non[j] ? a.db.stats[2 * GN + o] : 0.0f;
            qsh[j] = non[j] ? a.db.stats[3 * GN + o] : 0.0f;
            qk1[j] = non[j] ? a.db.coef[0 * GN + o] : 0.0f;
            qk2[j] = non[j] ? a.db.coef[1 * GN + o] : 0.0f;
            qk3[j] = non[j] ? a.db.coef[2 * GN + o] : 0.0f;
        }
    }
    f32x16 acc[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // one batch of U row pairs in flight per wave.  The launches of the learner put only 1.5-2 waves on a SIMD (the
    // row splits are kept coarse so that the partial buffer stays small), so a wave alternates between "wait for the
    // batch" (~2 us) and U*NJW MFMAs: the batch must be deep -- U = 4 ran the 49152 x 116 x 116 gradient in 37 us
    // (16 exposed latencies per wave) for 10 us of MFMA time.
    // Loads go through buffer descriptors: the row offset of a batch is wave-uniform (SGPR soffset), the lane's column
    // offset is a fixed 32-bit voffset, masked lanes point out of range (the bounds check returns 0) -- no 64-bit address
    // arithmetic and no per-load select in the steady state, which is what lets U = 8 / 16 fit in registers.
    const uint32_t OOR = 0x80000000u;           // host checks that every tensor is < 2 GB
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A.p), 0, (int)((int64_t)a.M * a.A.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D.p), 0, (int)((int64_t)a.M * a.D.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(DPRO ? a.db.y : a.D.p), 0, (int)((int64_t)a.M * (DPRO ? N : a.D.ld) * ESZ), 0x00020000);
    const uint32_t voA = kon ? (uint32_t)(LHR * lh * a.A.ld + a.A.coff + k) * ESZ : OOR;
    uint32_t voD[NJW], voY[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        voD[j] = non[j] ? (uint32_t)(LHR * lh * a.D.ld + dcol[j]) * ESZ : OOR;
        voY[j] = non[j] ? (uint32_t)(LHR * lh * N + ncol[j]) * ESZ : OOR;
    }
    const uint32_t sA = (uint32_t)a.A.ld * ESZ, sD = (uint32_t)a.D.ld * ESZ, sY = (uint32_t)N * ESZ;      // row strides in bytes
    float av0[U], dv0[U][NJW], yv0[DPRO ? U : 1][DPRO ? NJW : 1];
    // row range of this wave as wave-uniform 32-bit scalars (the loop counter and the row offsets live in SGPRs)
    const int mb = __builtin_amdgcn_readfirstlane((int)mbeg), me = __builtin_amdgcn_readfirstlane((int)mend);
    // TAIL = false: all 2*U rows of the batch exist, no masks at all; TAIL = true: rows past the end of the share are
    // masked per lane through the voffset (OR with the out-of-range bit: branch-free)
    auto load_batch = [&](int m0, bool tail, float* av, float (*dv)[NJW], float (*yv)[DPRO ? NJW : 1]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = (uint32_t)(BF ? m0 + u : m0 + 2 * u);
            uint32_t msk = 0u;
            if (tail) msk = (BF ? m0 + u + 8 * lh : m0 + 2 * u + lh) < me ? 0u : OOR;
            av[u] = tnd_ld<MODE>(rA, voA | msk, r * sA);
#pragma unroll
            for (int j = 0; j < NJW; ++j) {
                dv[u][j] = tnd_ld<MODE>(rD, voD[j] | msk, r * sD);
                if (DPRO) yv[u][j] = tnd_ld<MODE>(rY, voY[j] | msk, r * sY);
            }
        }
        // keep the whole batch of loads ahead of the MFMAs (left alone, the scheduler interleaves load / s_waitcnt 0 /
        // MFMA one by one through two registers: every MFMA then pays a full memory latency)
        __builtin_amdgcn_sched_barrier(0);
    };
    // Branch-free transforms: masked lanes loaded zeros and have zero coefficients (q* = 0 when the column does not exist),
    // so only the rows past the end of the share need a select, and only in the tail batch.  With the transforms inside
    // `if (row and column exist)` every batch started with s_waitcnt vmcnt(0): the loads of the NEXT batch, already in
    // flight, were waited for as well.
    auto mma_batch = [&](int m0, bool tail, const float* av, const float (*dv)[NJW], const float (*yv)[DPRO ? NJW : 1]) {
        bf16x8 xb, db[BF ? NJW : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float keep = (!tail || (BF ? m0 + u + 8 * lh : m0 + 2 * u + lh) < me) ? 1.0f : 0.0f;
            float x = av[u];
            if (APRO) x = fmaf(asc, x, ash);          // rows past the end: x = ash, but their d is 0
            if (BF) xb[u & 7] = (__bf16)x;
#pragma unroll
            for (int j = 0; j < NJW; ++j) {
                float d = dv[u][j];
                if (DPRO) {
                    const float y = yv[u][j];
                    if (a.db.act == ACT_RELU6) {
                        const float z = fmaf(qsc[j], y, qsh[j]);
                        d = relu6_open(z) ? d : 0.0f;
                    }
                    const float xh = (y - qm[j]) * qi[j];
                    d = keep * (qk1[j] * (d - qk2[j] - xh * qk3[j]));
                }
                if (BF) db[BF ? j : 0][u & 7] = (__bf16)d;
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, d, acc[j], 0, 0, 0);
            }
        }
        if (BF) {
#pragma unroll
            for (int j = 0; j < NJW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, db[BF ? j : 0], acc[j], 0, 0, 0);
        }
    };
    // steady state: two register sets, the loads of batch i+1 are in flight while the MFMAs of batch i run
    float av1[U], dv1[U][NJW], yv1[DPRO ? U : 1][DPRO ? NJW : 1];
    int m0 = mb;
    bool has = m0 + 2 * U <= me;
    if (has) load_batch(m0, false, av0, dv0, yv0);
    while (has) {
        const int m1 = m0 + 2 * U;
        const bool has1 = m1 + 2 * U <= me;
        if (has1) load_batch(m1, false, av1, dv1, yv1);
        mma_batch(m0, false, av0, dv0, yv0);
        m0 = m1;
        if (!has1) break;
        const int m2 = m1 + 2 * U;
        has = m2 + 2 * U <= me;
        if (has) load_batch(m2, false, av0, dv0, yv0);
        mma_batch(m1, false, av1, dv1, yv1);
        m0 = m2;
    }
    if (m0 < me) {
        load_batch(m0, true, av0, dv0, yv0);
        mma_batch(m0, true, av0, dv0, yv0);
    }
    if (a.RS2 == 2) {       // fold the second half's accumulators into the first half's (fixed order)
        float* red = tnd_red + (size_t)wave * NJW * 16 * 64 + lane;
        if (half == 1 && active) {
#pragma unroll
            for (int j = 0; j < NJW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64] = acc[j][r];
        }
        __syncthreads();
        if (half == 0 && active) {
#pragma unroll
            for (int j = 0; j < NJW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64];
        }
    }
    if (half != 0 || !active) return;
    // C/D layout: column (n) = lane&31, row (k) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* out = a.part + ((int64_t)blockIdx.x * a.RS + rs) * K * N;
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        if (!non[j]) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + ki * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (kk < K) out[(int64_t)kk * N + ncol[j]] = acc[j][r];
        }
    }
}

// Transposed wave mapping for the launches with a D prologue: a wave owns ONE n tile and up to NJW k tiles, so the
// BatchNorm-backward transform of a D element (2 loads + ~10 VALU) is computed once per workgroup instead of once per
// k-tile wave (4x), and a row pair costs NJW + 2 loads instead of 2*NJW + 1.  The cheap side (A, at most one fma) is the
// one that is re-read by the 4 waves.  Same partial layout and reduction as tn_direct_kernel.
template <int NJW, bool APRO, bool DPRO, int U, int MODE>
__global__ void __launch_bounds__(512) tn_direct_tr_kernel(TnDirectArgs a) {
    constexpr bool BF = MODE >= 1;
    constexpr uint32_t ESZ = MODE == 2 ? 2u : 4u;
    static_assert(!BF || U == 8, "BF: one batch = one K = 16 MFMA step (8 rows per half-wave)");
    constexpr int LHR = BF ? 8 : 1;
    extern __shared__ float tnd_red[];          // RS2 == 2: [4 waves][NJW][16][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8;
    const int l32 = lane & 31, lh = lane >> 5;
    const int K = a.K, N = a.N;
    const int k0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
    const int KT = min(4, (K - k0 + 31) / 32), NT = min(4, (N - n0 + 31) / 32);
    const int ni = wave % a.WK, rest = wave / a.WK;           // WK = waves along n here
    const int ksp = rest % a.NSPL, rs = rest / a.NSPL;        // NSPL = k splits
    const int kj0 = ksp * NJW;
    const int kjn = min(NJW, KT - kj0);
    const bool active = !(ni >= NT || kjn <= 0 || rs >= a.RS);
    const int grp = blockIdx.x / a.nspg;
    const int64_t gend = (int64_t)(grp + 1) * a.Mg;
    int64_t mbeg = (int64_t)grp * a.Mg + (int64_t)(blockIdx.x % a.nspg) * a.rows_per;
    int64_t mend = mbeg + a.rows_per;
    if (mend > gend) mend = gend;
    {
        const int64_t len = mend - mbeg;
        const int TS = a.RS * a.RS2;
        const int64_t per = ((len + TS - 1) / TS + 1) / 2 * 2;
        mbeg += (rs * a.RS2 + half) * per;
        if (mbeg + per < mend) mend = mbeg + per;
        if (!active) mend = mbeg;
    }
    const int n = n0 + ni * 32 + l32;
    const bool non = active && n < N;
    int dcol = a.D.coff + n;
    float qm = 0.0f, qi = 0.0f, qsc = 0.0f, qsh = 0.0f, qk1 = 0.0f, qk2 = 0.0f, qk3 = 0.0f;
    if (DPRO) {
        if (a.db.shuffle_ctot) dcol = shuffle_dst(dcol, a.db.shuffle_ctot);
        if (non) {
            const int GN = a.G * N, o = grp * N + n;
            qm = a.db.stats[0 * GN + o];
            qi = a.db.stats[1 * GN + o];
            qsc = a.db.stats[2 * GN + o];
            qsh = a.db.stats[3 * GN + o];
            qk1 = a.db.coef[0 * GN + o];
            qk2 = a.db.coef[1 * GN + o];
            qk3 = a.db.coef[2 * GN + o];
        }
    }
    int kcol[NJW];
    bool kon[NJW];
    float asc[NJW], ash[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        const int k = k0 + (kj0 + j) * 32 + l32;
        kcol[j] = k;
        kon[j] = j < kjn && k < K;
        asc[j] = 1.0f;
        ash[j] = 0.0f;
        if (APRO && kon[j]) {
            asc[j] = a.a_stats[2 * a.G * K + grp * K + k];
            ash[j] = a.a_stats[3 * a.G * K + grp * K + k];
        }
    }
    f32x16 acc[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    const uint32_t OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A.p), 0, (int)((int64_t)a.M * a.A.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D.p), 0, (int)((int64_t)a.M * a.D.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(DPRO ? a.db.y : a.D.p), 0, (int)((int64_t)a.M * (DPRO ? N : a.D.ld) * ESZ), 0x00020000);
    const uint32_t voD = non ? (uint32_t)(LHR * lh * a.D.ld + dcol) * ESZ : OOR;
    const uint32_t voY = non ? (uint32_t)(LHR * lh * N + n) * ESZ : OOR;
    uint32_t voA[NJW];
#pragma unroll
    for (int j = 0; j < NJW; ++j) voA[j] = kon[j] ? (uint32_t)(LHR * lh * a.A.ld + a.A.coff + kcol[j]) * ESZ : OOR;
    const uint32_t sA = (uint32_t)a.A.ld * ESZ, sD = (uint32_t)a.D.ld * ESZ, sY = (uint32_t)N * ESZ;
    const int mb = __builtin_amdgcn_readfirstlane((int)mbeg), me = __builtin_amdgcn_readfirstlane((int)mend);
    float av0[U][NJW], dv0[U], yv0[U], av1[U][NJW], dv1[U], yv1[U];
    auto load_batch = [&](int m0, bool tail, float (*av)[NJW], float* dv, float* yv) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = (uint32_t)(BF ? m0 + u : m0 + 2 * u);
            uint32_t msk = 0u;
            if (tail) msk = (BF ? m0 + u + 8 * lh : m0 + 2 * u + lh) < me ? 0u : OOR;
            dv[u] = tnd_ld<MODE>(rD, voD | msk, r * sD);
            if (DPRO) yv[u] = tnd_ld<MODE>(rY, voY | msk, r * sY);
#pragma unroll
            for (int j = 0; j < NJW; ++j) av[u][j] = tnd_ld<MODE>(rA, voA[j] | msk, r * sA);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_batch = [&](int m0, bool tail, const float (*av)[NJW], const float* dv, const float* yv) {
        bf16x8 db, xb[BF ? NJW : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float keep = (!tail || (BF ? m0 + u + 8 * lh : m0 + 2 * u + lh) < me) ? 1.0f : 0.0f;      // branch-free (see tn_direct_kernel)
            float d = dv[u];
            if (DPRO) {
                const float y = yv[u];
                if (a.db.act == ACT_RELU6) {
                    const float z = fmaf(qsc, y, qsh);
                    d = relu6_open(z) ? d : 0.0f;
                }
                const float xh = (y - qm) * qi;
                d = keep * (qk1 * (d - qk2 - xh * qk3));
            }
            if (BF) db[u & 7] = (__bf16)d;
#pragma unroll
            for (int j = 0; j < NJW; ++j) {
                float x = av[u][j];
                if (APRO) x = fmaf(asc[j], x, ash[j]);
                if (BF) xb[BF ? j : 0][u & 7] = (__bf16)x;
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, d, acc[j], 0, 0, 0);
            }
        }
        if (BF) {
#pragma unroll
            for (int j = 0; j < NJW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[BF ? j : 0], db, acc[j], 0, 0, 0);
        }
    };
    int m0 = mb;
    bool has = m0 + 2 * U <= me;
    if (has) load_batch(m0, false, av0, dv0, yv0);
    while (has) {
        const int m1 = m0 + 2 * U;
        const bool has1 = m1 + 2 * U <= me;
        if (has1) load_batch(m1, false, av1, dv1, yv1);
        mma_batch(m0, false, av0, dv0, yv0);
        m0 = m1;
        if (!has1) break;
        const int m2 = m1 + 2 * U;
        has = m2 + 2 * U <= me;
        if (has) load_batch(m2, false, av0, dv0, yv0);
        mma_batch(m1, false, av1, dv1, yv1);
        m0 = m2;
    }
    if (m0 < me) {
        load_batch(m0, true, av0, dv0, yv0);
        mma_batch(m0, true, av0, dv0, yv0);
    }
    if (a.RS2 == 2) {
        float* red = tnd_red + (size_t)wave * NJW * 16 * 64 + lane;
        if (half == 1 && active) {
#pragma unroll
            for (int j = 0; j < NJW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64] = acc[j][r];
        }
        __syncthreads();
        if (half == 0 && active) {
#pragma unroll
            for (int j = 0; j < NJW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64];
        }
    }
    if (half != 0 || !non) return;
    // C/D layout: column (n) = lane&31, row (k) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* out = a.part + ((int64_t)blockIdx.x * a.RS + rs) * K * N;
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
        if (j >= kjn) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + (kj0 + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (kk < K) out[(int64_t)kk * N + n] = acc[j][r];
        }
    }
}

struct TndPlan {
    int gy, gz, nspg, nsplit, rows_per;
    int WK, NSPL, NJW, RS, RS2;
    int TR;                     // transposed wave mapping (waves along n, NJW k tiles per wave)
};

static TndPlan tnd_plan(int M, int N, int K, int G, bool dpro = false) {
    TndPlan p;
    p.gy = cdiv(K, 128);
    p.gz = cdiv(N, 128);
    int KT = K >= 128 ? 4 : cdiv(K, 32), NT = N >= 128 ? 4 : cdiv(N, 32);   // tiles of the (first) 128-block
    static const bool tr_on = true;
    p.TR = (dpro && tr_on) ? 1 : 0;
    if (p.TR) {                                  // same mapping with the roles of k and n exchanged
        const int t = KT;
        KT = NT;
        NT = t;
    }
    p.WK = KT >= 3 ? 4 : KT;
    const int left = 4 / p.WK;                  // 1, 2 or 4 waves left for n tiles / row splits
    p.NSPL = left < NT ? left : NT;
    p.NJW = cdiv(NT, p.NSPL);
    if (p.NJW == 3) p.NJW = 4;                   // instantiated widths: 1, 2, 4
    p.RS = left / p.NSPL;
    const int Mg = M / G;
    // enough workgroups to fill the chip a few times over, but >= 128 rows each so that the partial buffer stays small
    static const int tn_target = 1024;     // 2048 doubles the split-M partial traffic (8.8 GB/update-step) for no measurable gain
    int target = tn_target / (p.gy * p.gz * G * p.RS);
    if (target < 1) target = 1;
    static const int tn_minrows = 128;
    // in-workgroup row halves (8 waves): the same rows per wave, twice the rows per partial
    static const int tn_rs2 = 2;
    p.RS2 = (tn_rs2 == 2 && Mg >= 2 * tn_minrows) ? 2 : 1;
    int ns = Mg / (tn_minrows * p.RS2);
    if (ns > target) ns = target;
    if (ns < 1) ns = 1;
    p.rows_per = cdiv(cdiv(Mg, ns), 2) * 2;
    p.nspg = cdiv(Mg, p.rows_per);
    p.nsplit = G * p.nspg;
    return p;
}

int64_t gemm_tn_part_elems(int M, int N, int K, int G) {
    const TndPlan p = tnd_plan(M, N, K, G, false), q = tnd_plan(M, N, K, G, true);
    const int64_t a = (int64_t)p.nsplit * p.RS * K * N, b = (int64_t)q.nsplit * q.RS * K * N;
    const int64_t c = gemm_tn_lds_part_elems(M, N, K, G);          // the LDS-staged form of the bf16 modes (gemm_tn_lds.hip)
    return std::max(std::max(a, b), c);
}

bool gemm_tn_dpro_supported(int) { return true; }

template <int NJW, int U, int BF>
static void launch_tnd_u(bool apro, bool dpro, dim3 grid, hipStream_t st, const TnDirectArgs& a) {
    const dim3 blk(256 * a.RS2);
    const size_t lds = a.RS2 == 2 ? (size_t)4 * NJW * 16 * 64 * sizeof(float) : 0;     // <= 64 KB
    if (a.TR) {         // (planned only with a D prologue)
        if (apro) hipLaunchKernelGGL((tn_direct_tr_kernel<NJW, true, true, U, BF>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((tn_direct_tr_kernel<NJW, false, true, U, BF>), grid, blk, lds, st, a);
        return;
    }
    if (apro && dpro) hipLaunchKernelGGL((tn_direct_kernel<NJW, true, true, U, BF>), grid, blk, lds, st, a);
    else if (apro) hipLaunchKernelGGL((tn_direct_kernel<NJW, true, false, U, BF>), grid, blk, lds, st, a);
    else if (dpro) hipLaunchKernelGGL((tn_direct_kernel<NJW, false, true, U, BF>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((tn_direct_kernel<NJW, false, false, U, BF>), grid, blk, lds, st, a);
}

template <int NJW>
static void launch_tnd(bool apro, bool dpro, dim3 grid, hipStream_t st, const TnDirectArgs& a, bool bf, int at) {
    if (at) {           // bf16 operands AND bf16 activation storage
        launch_tnd_u<NJW, 8, 2>(apro, dpro, grid, st, a);
        return;
    }
    if (bf) {           // bf16 operands: a batch is one 16-row MFMA step
        launch_tnd_u<NJW, 8, 1>(apro, dpro, grid, st, a);
        return;
    }
    static const int u = 4;
    static const int ud = 8;      // D prologue (transposed mapping: 200 VGPRs at U = 8; 19.6 -> 19.2 ms/update-step over U = 4)
    const int uu = dpro ? ud : u;
    if (uu >= 16 && !dpro) launch_tnd_u<NJW, 16, 0>(apro, dpro, grid, st, a);
    else if (uu >= 8) launch_tnd_u<NJW, 8, 0>(apro, dpro, grid, st, a);
    else launch_tnd_u<NJW, 4, 0>(apro, dpro, grid, st, a);
}

int gemm_tn(View A, View D, float* Cout, int M, int N, int K, float* part, int accumulate, hipStream_t st, int G,
            const float* pro_stats, const TnBnBwd* dpro, bool bf16_operands, int at) {
    if (G < 1 || M % G != 0) {
        set_error("gemm_tn: M=%d is not a multiple of G=%d", M, G);
        return -1;
    }
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (at && !bf16_operands) {
        set_error("gemm_tn: bf16 activation storage needs the bf16-operand variant");
        return -1;
    }
    if ((int64_t)M * A.ld * 4 >= (1ll << 31) || (int64_t)M * D.ld * 4 >= (1ll << 31) || (int64_t)M * N * 4 >= (1ll << 31)) {
        set_error("gemm_tn: operands of 2 GB or more are not supported (M=%d)", M);
        return -1;
    }
    static const bool trace = false;      // shapes of the filter-gradient GEMMs
    if (trace) fprintf(stderr, "gemm_tn M=%d K=%d N=%d G=%d apro=%d dpro=%d shuffle=%d bf16=%d at=%d\n", M, K, N, G, pro_stats != nullptr, dpro != nullptr, dpro ? dpro->shuffle_ctot : 0, (int)bf16_operands, at);
    static const bool diag_skip = cdrl_getenv("CDRL_DIAG_SKIP_TN") && atoi(cdrl_getenv("CDRL_DIAG_SKIP_TN")) == 1;   // timing diagnostics only
    if (diag_skip) return 0;
    // bf16 modes: operands staged once per workgroup through LDS (gemm_tn_lds.hip) -- the direct form is a stream of 2-byte loads
    // there; CDRL_TN_LDS=0 keeps the direct form, CDRL_TN_LDS=1 restricts the LDS form to one column block (96 <= K, N <= 128).
    // Shapes: its 128 x 128 column block with one k tile per wave only pays for wide products -- measured isolated at B = 1024:
    // K = N = 116: 47.6 vs 64.2 us, 232: 40.9 vs 58.3 us, but K = N = 58: 82.9 vs 68.2 us and 24 x 58: 264 vs 126 us.
    // (A NARROW geometry of the LDS kernel for the K, N <= 64 convs of stage 0 -- one 64-column block, 64-row chunks, the <= 4 tile
    //  products spread over the waves -- was built and measured in round 3: isolated 52 vs 68 us at 58 x 58, 150 vs 126 us at
    //  24 x 58 (B = 1024), and NO change of the update-step (32.76 vs 32.80 ms): next to the main stream these kernels take 3-4x
    //  their isolated time either way.  Not kept.)
    // (Until the ReLU6 masks became single compares -- relu6_open(), cdrl_common.h -- the shapes with several column blocks were kept
    //  off this path: next to them the fused backward-data GEMM on the main stream lost its run-to-run reproducibility.  The cause
    //  was in that kernel's mask code, not here: DESIGN.md "What round 3 found", tools/det_co.py.)
    static const int lds_mode = 2;
    const bool lds_shape = K >= 96 && N >= 96 && (lds_mode == 2 || (K <= 128 && N <= 128));
    if (bf16_operands && lds_mode != 0 && lds_shape && gemm_tn_lds_supported(A, D, N, K, dpro))
        return gemm_tn_lds(A, D, Cout, M, N, K, part, accumulate, st, G, pro_stats, dpro, at);
    // float32 tensors through the same staging: three bf16 planes per operand + six plane products on the bf16 pipe (exact split, float32-
    // accurate).  Round 3 measured it at K = N = 116, M = 196608 (B = 1024), isolated: direct 87.6 us, float32-MFMA LDS form 104.1 us,
    // this form 73.7 us; K = N = 232, M = 49152: 81.4 / 96.5 / 70.2 us -- and NO gain in the update-step (15.78 direct | 16.58 | 15.83 ms):
    // then the stage-0 / stage-1 filter gradients ran next to it on the side stream.  Since the fused conv backward took those over, what is
    // left here are the 232-channel convs of stage 2, the head conv and the shortcut convs, and the faster kernel shortens the contention
    // with the critical stream: round 5, same box, 14.27 (direct) | 14.31 (float32-MFMA LDS form) | 14.01 ms (this form).  Default since
    // round 5; CDRL_TN_LDS_F32=0 -> direct form, 1 -> float32 LDS columns + v_mfma_f32_32x32x2_f32.
    static const int lds_f32 = cdrl_getenv("CDRL_TN_LDS_F32") ? atoi(cdrl_getenv("CDRL_TN_LDS_F32")) : 2;
    if (!bf16_operands && lds_f32 && K >= 96 && N >= 96 && gemm_tn_lds_supported(A, D, N, K, dpro))
        return gemm_tn_lds(A, D, Cout, M, N, K, part, accumulate, st, G, pro_stats, dpro, 0, lds_f32 == 2 ? 2 : 1);
    const TndPlan p = tnd_plan(M, N, K, G, dpro != nullptr);
    TnDirectArgs a;
    a.A = A;
    a.D = D;
    a.part = part;
    a.M = M;
    a.N = N;
    a.K = K;
    a.G = G;
    a.Mg = M / G;
    a.rows_per = p.rows_per;
    a.nspg = p.nspg;
    a.WK = p.WK;
    a.NSPL = p.NSPL;
    a.NJW = p.NJW;
    a.RS = p.RS;
    a.RS2 = p.RS2;
    a.TR = p.TR;
    a.a_stats = pro_stats;
    a.db = TnBnBwd{};
    if (dpro) a.db = *dpro;
    dim3 grid(p.nsplit, p.gy, p.gz);
    switch (p.NJW) {
        case 1: launch_tnd<1>(pro_stats != nullptr, dpro != nullptr, grid, st, a, bf16_operands, at); break;
        case 2: launch_tnd<2>(pro_stats != nullptr, dpro != nullptr, grid, st, a, bf16_operands, at); break;
        default: launch_tnd<4>(pro_stats != nullptr, dpro != nullptr, grid, st, a, bf16_operands, at); break;
    }
    CDRL_LAUNCH_CHECK();
    const int64_t n = (int64_t)K * N;
    return reduce_partials_f32(part, p.nsplit * p.RS, n, n, Cout, accumulate, st);
}

}  // namespace cdrl
