// Filter-gradient GEMM  W[K,N] = sum_m A[m,K]^T D[m,N]  on the bf16 matrix pipe with the operands staged ONCE per workgroup through
// LDS (gfx950, v_mfma_f32_32x32x16_bf16) -- the form used by the bf16 modes of configuration 3 (bf16 MFMA operands; float32 or
// bf16 activation tensors).
//
// Why not gemm_tn_direct.hip here: that kernel feeds the MFMA straight from global memory -- lane = column, one element per lane
// and row -- which is the right shape for float32 (4-byte lanes, 128-byte segments) but is a stream of 2-byte loads once the
// tensors are bf16, and every wave that needs an operand column loads (and, with a BatchNorm-backward prologue, transforms) it
// again: 48 vector-memory instructions per wave and 16 rows, 8 waves per workgroup.  Measured at B = 1024 (M = 196608,
// K = N = 116): 285 us with float32 tensors, 390 us with bf16 tensors, for ~10 us of HBM time.
//
// Here a workgroup (4 waves) walks its rows in chunks of 32:
//   1. every thread loads a 4-row x 4-column micro-tile of A, of D (through the channel-shuffle gather when D is the gradient of
//      a unit output: the even / odd destination columns are two CONTIGUOUS source ranges) and, with the BatchNorm-backward
//      prologue, of the BN's raw input y -- 8- / 16-byte loads, 4 per tensor, all issued before the matrix work of the
//      previous chunk;
//   2. applies the prologues in registers (the thread's 4 columns are fixed: coefficients live in registers), rounds to bf16 and
//      writes the micro-tile TRANSPOSED into LDS ([column][row], odd dword stride): 2 x 4-byte writes per column;
//   3. wave w multiplies k tile w against all (<= 4) n tiles: per 16-row step 4 + 4 * NT ds_read_b32 and NT MFMAs.
// ~35x fewer vector-memory instructions per row than the direct form, every element loaded and transformed once per workgroup.
// Partials are float [slot][K][N], summed in fixed order by reduce_partials_f32 (deterministic), same as gemm_tn_direct.hip.
#include <stdlib.h>

#include "colreduce.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct TnLdsArgs {
    View A, D;
    float* part;
    int M, N, K, G, Mg;
    int rows_per, nspg;
    const float* a_stats;       // [4][G][K] or null
    TnBnBwd db;                 // y == null: no D prologue
};

constexpr int TNL_R = 32;               // rows per chunk

// MODE 2: activation tensors (A, D, y) are bf16 in HBM, bf16 MFMA operands; MODE 1: float32 tensors, operands rounded to bf16 after
// the prologues; MODE 0: float32 tensors AND float32 operands (v_mfma_f32_32x32x2_f32: the exact k-ordered fmaf chain of the
// float32 engine) -- same staging, the LDS columns hold floats.  MODE 3: float32 tensors, float32-ACCURATE product on the bf16 pipe:
// after the prologues every operand value is split into three bf16 planes (x = x1 + x2 + x3 to 24 mantissa bits, as gemm_x3.hip /
// gemm_pw_x3.hip do for the forward GEMMs) and the product is the six plane products x1 d1 + x1 d2 + x2 d1 + x1 d3 + x3 d1 + x2 d2:
// 6 x 32 matrix-pipe cycles per 16 rows and tile pair instead of 8 x 64 for v_mfma_f32_32x32x2_f32.  Three planes per operand do
// not leave room for a second LDS buffer at two workgroups per CU: one buffer, two barriers per chunk.
template <int MODE, bool APRO, bool DPRO>
__global__ void __launch_bounds__(256, 2) tn_lds_kernel(TnLdsArgs a) {
    constexpr bool BH = MODE == 2, F32 = MODE == 0, X3 = MODE == 3;
    constexpr int NBUF = X3 ? 1 : 2, PL = X3 ? 3 : 1;         // LDS buffers, bf16 planes per operand
    // dwords per LDS column (odd: conflict-free fragment reads, 2-way on the writes): row pairs (bf16) or rows (float32)
    constexpr int TNL_CS = F32 ? TNL_R + 1 : TNL_R / 2 + 1;
    __shared__ uint32_t AsT[NBUF * PL][128 * TNL_CS];
    __shared__ uint32_t DsT[NBUF * PL][128 * TNL_CS];
    constexpr uint32_t ESZ = BH ? 2u : 4u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, lh = lane >> 5;
    const int K = a.K, N = a.N;
    const int k0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
    const int KT = min(4, (K - k0 + 31) / 32), NT = min(4, (N - n0 + 31) / 32);
    const int grp = blockIdx.x / a.nspg;
    const int64_t gend = (int64_t)(grp + 1) * a.Mg;
    const int64_t mbeg64 = (int64_t)grp * a.Mg + (int64_t)(blockIdx.x % a.nspg) * a.rows_per;
    const int mb = (int)mbeg64;
    const int me = (int)min(mbeg64 + a.rows_per, gend);
    // micro-tile of this thread: rows 4 rg .. 4 rg + 3 of the chunk, columns 4 cq .. 4 cq + 3 of the 128-column block
    const int rg = tid >> 5, cq = tid & 31;
    const int ka = k0 + 4 * cq, nd = n0 + 4 * cq;
    // per-column coefficients (0 for columns that do not exist)
    float asc[4], ash[4], qm[4], qi[4], qsc[4], qsh[4], qk1[4], qk2[4], qk3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asc[j] = 1.0f;
        ash[j] = 0.0f;
        if (APRO) {
            const bool on = ka + j < K;
            asc[j] = on ? a.a_stats[2 * a.G * K + grp * K + ka + j] : 0.0f;
            ash[j] = on ? a.a_stats[3 * a.G * K + grp * K + ka + j] : 0.0f;
        }
        qm[j] = qi[j] = qsc[j] = qsh[j] = qk1[j] = qk2[j] = qk3[j] = 0.0f;
        if (DPRO && nd + j < N) {
            const int GN = a.G * N, o = grp * N + nd + j;
            qm[j] = a.db.stats[0 * GN + o];
            qi[j] = a.db.stats[1 * GN + o];
            qsc[j] = a.db.stats[2 * GN + o];
            qsh[j] = a.db.stats[3 * GN + o];
            qk1[j] = a.db.coef[0 * GN + o];
            qk2[j] = a.db.coef[1 * GN + o];
            qk3[j] = a.db.coef[2 * GN + o];
        }
    }
    const bool relu = DPRO && a.db.act == ACT_RELU6;
    // ---- global loads through buffer descriptors (masked lanes point out of range and read 0)
    const uint32_t OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A.p), 0, (int)((int64_t)a.M * a.A.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.D.p), 0, (int)((int64_t)a.M * a.D.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(DPRO ? a.db.y : a.D.p), 0, (int)((int64_t)a.M * (DPRO ? N : a.D.ld) * ESZ), 0x00020000);
    // column offsets (elements) of the thread's 4 columns; pairs (j, j + 1) resp. (j, j + 2) are adjacent in memory
    const bool a4 = (ka + 3 < K);                        // all four A columns exist (K is even: otherwise the first two or none)
    const bool a2 = (ka + 1 < K);
    const bool d4 = (nd + 3 < N), d2 = (nd + 1 < N);
    const int shuf = DPRO ? a.db.shuffle_ctot : 0;
    // D columns: plain view -> D.coff + nd + j; through the shuffle map -> even destination columns (j = 0, 2) and odd ones
    // (j = 1, 3) are each two ADJACENT source columns
    uint32_t voA = a2 ? (uint32_t)(a.A.coff + ka) * ESZ : OOR;
    uint32_t voD0, voD1;        // shuffle: source of (j = 0, 2) and of (j = 1, 3); plain: columns (0, 1) and (2, 3)
    // half micro-tile at the end of a block whose width is 2 mod 4 (58 channels): the element behind the thread's last column may
    // lie outside the tensor (last row), and a 4-byte access that straddles the end of the buffer is dropped as a whole -- such a
    // thread reads the pair (s - 1, s) instead of (s, s + 1) and keeps the upper half
    const bool dhalf = shuf && d2 && !d4;
    if (shuf) {
        voD0 = d2 ? (uint32_t)(shuffle_dst(a.D.coff + nd, shuf) - (dhalf ? 1 : 0)) * ESZ : OOR;
        voD1 = d2 ? (uint32_t)(shuffle_dst(a.D.coff + nd + 1, shuf) - (dhalf ? 1 : 0)) * ESZ : OOR;
    } else {
        voD0 = d2 ? (uint32_t)(a.D.coff + nd) * ESZ : OOR;
        voD1 = d4 ? (uint32_t)(a.D.coff + nd + 2) * ESZ : OOR;
    }
    const uint32_t voY = d2 ? (uint32_t)nd * ESZ : OOR;
    const uint32_t sA = (uint32_t)a.A.ld * ESZ, sD = (uint32_t)a.D.ld * ESZ, sY = (uint32_t)N * ESZ;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    // two adjacent elements at a byte offset, as floats.  The row offset goes into the PER-LANE offset: a wave covers two row
    // groups, and a non-uniform scalar offset would be executed as a waterfall loop (one pass per distinct value)
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t ro, float& x0, float& x1) {
        const uint32_t off = (vo & 0x80000000u) ? 0x80000000u : vo + ro;
        if (BH) {
            const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
            x0 = bf_lo(w);
            x1 = bf_hi(w);
        } else {
            const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
            x0 = __uint_as_float(w[0]);
            x1 = __uint_as_float(w[1]);
        }
    };
    struct Regs {
        float a[4][4], d[4][4], y[DPRO ? 4 : 1][DPRO ? 4 : 1];      // [row][column]
    };
    auto load_chunk = [&](int m0, Regs& r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 4 * rg + i;
            const uint32_t msk = m < me ? 0u : OOR;
            const uint32_t row = (uint32_t)m;
            ld2(rA, voA | msk, row * sA, r.a[i][0], r.a[i][1]);
            ld2(rA, (a4 ? voA + 2 * ESZ : OOR) | msk, row * sA, r.a[i][2], r.a[i][3]);
            if (shuf) {
                ld2(rD, voD0 | msk, row * sD, r.d[i][0], r.d[i][2]);
                ld2(rD, voD1 | msk, row * sD, r.d[i][1], r.d[i][3]);
                if (dhalf) {
                    r.d[i][0] = r.d[i][2];
                    r.d[i][1] = r.d[i][3];
                    r.d[i][2] = r.d[i][3] = 0.0f;
                }
            } else {
                ld2(rD, voD0 | msk, row * sD, r.d[i][0], r.d[i][1]);
                ld2(rD, voD1 | msk, row * sD, r.d[i][2], r.d[i][3]);
            }
            if (DPRO) {
                ld2(rY, voY | msk, row * sY, r.y[DPRO ? i : 0][0], r.y[DPRO ? i : 0][DPRO ? 1 : 0]);
                ld2(rY, (d4 ? voY + 2 * ESZ : OOR) | msk, row * sY, r.y[DPRO ? i : 0][DPRO ? 2 : 0], r.y[DPRO ? i : 0][DPRO ? 3 : 0]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // prologues + bf16 rounding + transposed LDS write: column c of the block at dwords [c * TNL_CS, ...), rows packed in pairs
    auto store_chunk = [&](int m0, int buf, const Regs& r) {
        float keep[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) keep[i] = (m0 + 4 * rg + i) < me ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[4], d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] = r.a[i][j];
                if (APRO) x[i] = fmaf(asc[j], x[i], ash[j]);       // (rows past the end: x = shift, but their d is 0)
                d[i] = r.d[i][j];
                if (DPRO) {
                    const float y = r.y[DPRO ? i : 0][DPRO ? j : 0];
                    if (relu) {
                        const float z = fmaf(qsc[j], y, qsh[j]);
                        d[i] = relu6_open(z) ? d[i] : 0.0f;
                    }
                    const float xh = (y - qm[j]) * qi[j];
                    d[i] = keep[i] * (qk1[j] * (d[i] - qk2[j] - xh * qk3[j]));
                }
            }
            const int c = 4 * cq + j;
            if (F32) {
                uint32_t* pa = &AsT[buf][c * TNL_CS + 4 * rg];
                uint32_t* pd = &DsT[buf][c * TNL_CS + 4 * rg];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pa[i] = __float_as_uint(x[i]);
                    pd[i] = __float_as_uint(d[i]);
                }
            } else if (X3) {
                // three bf16 planes: h1 = bf16(v), h2 = bf16(v - h1), h3 = bf16(v - h1 - h2) (the subtractions are exact)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    float hx[4], hd[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hx[i] = (float)(bf16_t)x[i];
                        hd[i] = (float)(bf16_t)d[i];
                        x[i] -= hx[i];
                        d[i] -= hd[i];
                    }
                    uint32_t* pa = &AsT[pl][c * TNL_CS + 2 * rg];
                    pa[0] = bf_pack(hx[0], hx[1]);
                    pa[1] = bf_pack(hx[2], hx[3]);
                    uint32_t* pd = &DsT[pl][c * TNL_CS + 2 * rg];
                    pd[0] = bf_pack(hd[0], hd[1]);
                    pd[1] = bf_pack(hd[2], hd[3]);
                }
            } else {
                uint32_t* pa = &AsT[buf][c * TNL_CS + 2 * rg];
                pa[0] = bf_pack(x[0], x[1]);
                pa[1] = bf_pack(x[2], x[3]);
                uint32_t* pd = &DsT[buf][c * TNL_CS + 2 * rg];
                pd[0] = bf_pack(d[0], d[1]);
                pd[1] = bf_pack(d[2], d[3]);
            }
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
    // fragment of a 16-row step: lane (column l32 of the tile, half lh) holds rows 16 s + 8 lh .. + 7 of its column
    auto frag = [&](const uint32_t* base, int col, int s) -> bf16x8 {
        const uint32_t* p = base + col * TNL_CS + 8 * s + 4 * lh;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t w;
        w[0] = p[0];
        w[1] = p[1];
        w[2] = p[2];
        w[3] = p[3];
        return __builtin_bit_cast(bf16x8, w);
    };
    auto mma_chunk = [&](int buf) {
        if (wave >= KT) return;
        if (F32) {      // K = 2 steps: lane (column l32, half lh) supplies row 2 s + lh of its column
            const uint32_t* pa = &AsT[buf][(wave * 32 + l32) * TNL_CS + lh];
#pragma unroll
            for (int s = 0; s < TNL_R / 2; ++s) {
                const float av = __uint_as_float(pa[2 * s]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < NT) {
                        const float dv = __uint_as_float(DsT[buf][(j * 32 + l32) * TNL_CS + 2 * s + lh]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dv, acc[j], 0, 0, 0);
                    }
                }
            }
            return;
        }
        if (X3) {
#pragma unroll
            for (int s = 0; s < TNL_R / 16; ++s) {
                const bf16x8 a1 = frag(AsT[0], wave * 32 + l32, s), a2 = frag(AsT[X3 ? 1 : 0], wave * 32 + l32, s),
                             a3 = frag(AsT[X3 ? 2 : 0], wave * 32 + l32, s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < NT) {
                        const bf16x8 d1 = frag(DsT[0], j * 32 + l32, s), d2 = frag(DsT[X3 ? 1 : 0], j * 32 + l32, s),
                                     d3 = frag(DsT[X3 ? 2 : 0], j * 32 + l32, s);
                        // smallest terms first
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, d2, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, d1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d3, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, d1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d2, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d1, acc[j], 0, 0, 0);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < TNL_R / 16; ++s) {
            const bf16x8 af = frag(AsT[buf], wave * 32 + l32, s);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < NT) {
                    const bf16x8 df = frag(DsT[buf], j * 32 + l32, s);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, df, acc[j], 0, 0, 0);
                }
            }
        }
    };
    Regs r0;
    int c = 0;
    if (mb < me) load_chunk(mb, r0);
    for (int m0 = mb; m0 < me; m0 += TNL_R, ++c) {
        store_chunk(m0, X3 ? 0 : (c & 1), r0);
        __syncthreads();
        if (m0 + TNL_R < me) load_chunk(m0 + TNL_R, r0);
        mma_chunk(X3 ? 0 : (c & 1));
        if (X3) __syncthreads();           // one buffer: every wave is done with the fragments before the next chunk is stored
    }
    if (wave >= KT) return;
    // C/D layout: column (n) = lane & 31, row (k) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = a.part + (int64_t)blockIdx.x * K * N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 32 + l32;
        if (j >= NT || n >= N) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int kk = k0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (kk < K) out[(int64_t)kk * N + n] = acc[j][q];
        }
    }
}

struct TnlPlan {
    int gy, gz, nspg, nsplit, rows_per;
};

static TnlPlan tnl_plan(int M, int N, int K, int G) {
    TnlPlan p;
    p.gy = cdiv(K, 128);
    p.gz = cdiv(N, 128);
    const int Mg = M / G;
    // ~1.5 workgroups per CU over all column blocks, at least 128 rows each (the partial buffer stays small), whole chunks
    // (768 | 512 | 384 | 256 workgroups: 33.3 | 33.1 | 32.8 | 33.0 ms / update-step at bf16-storage B = 1024: fewer partial slots to write
    //  and reduce; CDRL_TNL_WGS)
    static const int wgs = 384;
    int target = wgs / (p.gy * p.gz * G);
    if (target < 1) target = 1;
    int ns = Mg / 128;
    if (ns > target) ns = target;
    if (ns < 1) ns = 1;
    p.rows_per = cdiv(cdiv(Mg, ns), TNL_R) * TNL_R;
    p.nspg = cdiv(Mg, p.rows_per);
    p.nsplit = G * p.nspg;
    return p;
}

int64_t gemm_tn_lds_part_elems(int M, int N, int K, int G) { return (int64_t)tnl_plan(M, N, K, G).nsplit * K * N; }

bool gemm_tn_lds_supported(View A, View D, int N, int K, const TnBnBwd* dpro) {
    // adjacent-column pairs: even leading dimensions / offsets / widths (every tower tensor); the shuffle gather needs the two
    // halves of the concat to be even as well
    if ((A.ld & 1) || (A.coff & 1) || (D.ld & 1) || (D.coff & 1) || (K & 1) || (N & 1)) return false;
    if (dpro && dpro->shuffle_ctot && ((dpro->shuffle_ctot >> 1) & 1)) return false;
    return true;
}

int gemm_tn_lds(View A, View D, float* Cout, int M, int N, int K, float* part, int accumulate, hipStream_t st, int G, const float* pro_stats,
                const TnBnBwd* dpro, int at, int f32_mode) {
    if (G < 1 || M % G != 0) {
        set_error("gemm_tn_lds: M=%d is not a multiple of G=%d", M, G);
        return -1;
    }
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (!gemm_tn_lds_supported(A, D, N, K, dpro)) {
        set_error("gemm_tn_lds: odd leading dimension / offset / width (K=%d N=%d)", K, N);
        return -1;
    }
    const int64_t esz = at ? 2 : 4;
    if ((int64_t)M * A.ld * esz >= (1ll << 31) || (int64_t)M * D.ld * esz >= (1ll << 31) || (int64_t)M * N * esz >= (1ll << 31)) {
        set_error("gemm_tn_lds: operands of 2 GB or more are not supported (M=%d)", M);
        return -1;
    }
    const TnlPlan p = tnl_plan(M, N, K, G);
    TnLdsArgs a;
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
    a.a_stats = pro_stats;
    a.db = TnBnBwd{};
    if (dpro) a.db = *dpro;
    const dim3 grid(p.nsplit, p.gy, p.gz), blk(256);
    const bool ap = pro_stats != nullptr, dp = dpro != nullptr;
#define CDRL_TNL(BHV)                                                                               \
    do {                                                                                            \
        if (ap && dp) hipLaunchKernelGGL((tn_lds_kernel<BHV, true, true>), grid, blk, 0, st, a);    \
        else if (ap) hipLaunchKernelGGL((tn_lds_kernel<BHV, true, false>), grid, blk, 0, st, a);    \
        else if (dp) hipLaunchKernelGGL((tn_lds_kernel<BHV, false, true>), grid, blk, 0, st, a);    \
        else hipLaunchKernelGGL((tn_lds_kernel<BHV, false, false>), grid, blk, 0, st, a);           \
    } while (0)
    if (at) CDRL_TNL(2);
    else if (f32_mode == 2) CDRL_TNL(3);
    else if (f32_mode == 1) CDRL_TNL(0);
    else CDRL_TNL(1);
#undef CDRL_TNL
    CDRL_LAUNCH_CHECK();
    const int64_t n = (int64_t)K * N;
    return reduce_partials_f32(part, p.nsplit, n, n, Cout, accumulate, st);
}

}  // namespace cdrl
