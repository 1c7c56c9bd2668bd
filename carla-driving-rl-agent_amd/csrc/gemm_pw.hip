// Pointwise (1x1) convolution of the ShuffleNet unit as a *persistent skinny GEMM* with the neighbouring
// BatchNorm work folded in (gfx950, v_mfma_f32_32x32x2_f32).
//
// Reference: core/architectures.py:130,140 (Conv2D(k=1) -> BatchNormalization per time slice).  At the tower's
// channel counts (K, N <= 128) the GEMM C[M,N] = A[M,K] W[K,N] is HBM-bound (arithmetic intensity ~15-30 flop/B):
// what matters is that A is read once, C is written once, enough loads are in flight, and nothing else touches
// the tensors.  So, unlike the generic tiled gemm_nn:
//   * W lives in REGISTERS as MFMA B fragments for the whole kernel (K/2 steps x column tiles <= 192 VGPRs), loaded
//     once per workgroup; LDS only stages the A tile (conflict-free ds_read_b32 of the A fragment, stride 2*KSM+2);
//   * a workgroup walks over a contiguous range of row tiles inside ONE BatchNorm group (time slice), prefetching
//     the next A tile into registers while the MFMAs of the current tile run;
//   * prologue  PRO_BNAPPLY: a = scale[g][k] * a + shift[g][k] on the way into LDS (BN apply of the previous
//                            layer: its normalised output is never written to HBM);
//               PRO_BNBWD:   a = k1 * (dz - k2 - xhat * k3) with dz gathered (optionally through the channel-shuffle
//                            map, optionally ReLU6-masked) from the gradient of the BN OUTPUT and xhat from the BN's
//                            raw input: the BatchNorm backward "apply" happens on load, dy is never written to
//                            HBM; the column sums of a (bias gradient of the conv in front of that BN) come out
//                            as per-workgroup partials;
//   * epilogue  EPI_STATS:   per-workgroup (sum c, sum c^2) per output channel -> statistics partials of the
//                            following BatchNorm (no separate pass over C);
//               EPI_BNRED:   (sum c, sum c * xhat) with xhat from the raw input `ey` of the BatchNorm that C is the
//                            output-gradient of (backward-data GEMM feeding a BN backward: no separate reduce pass).
// Partials are double, one row per workgroup, combined in fixed order by bn_finalize / bn_bwd_finalize.
#include "colreduce.h"
#include "pack_bodies.h"
#include <algorithm>

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct PwArgs {
    View A;                     // PRO_BNBWD: gradient w.r.t. the BN output (the dz source)
    int a_shuffle;              // PRO_BNBWD: channel-shuffle gather on the columns of A (0 = none)
    int a_act;                  // PRO_BNBWD: ReLU6 mask from the BN output
    const float* a_y;           // PRO_BNBWD: raw BN input [M][K] dense
    const float* pro_coef;      // PRO_BNBWD: [3][G][K] (k1, k2, k3)
    double* part2;              // PRO_BNBWD: [G][nbpg][K] column sums of the transformed A (or null)
    const float* pro_stats;     // [4][G][K]
    const float* W;
    int sbk, sbn;
    const float* Wp;            // optional: W in fragment order [ceil(N/32)][2][32][KSM] (pw_pack_many): 16-byte loads
                                // BF variant: bf16 fragments [ceil(N/32)][KSM/8][2][32][8] (pw_pack_many with PwPack::bf16)
    const float* bias;
    View C;
    int accumulate;
    const float* ey;            // EPI_BNRED: [M][N] dense
    const float* epi_stats;     // EPI_BNRED: [4][G][N]
    double* part;               // [G][nbpg][2][N]
    int N, K, G, Mg, nbpg, tpb;
    int bf;                     // host dispatch only: Wp holds bf16 fragments -> BF variant
    int at;                     // host dispatch only: 1 = the activation tensors (A, a_y, C, ey) are bf16 in HBM (needs bf)
};

// register budget: W fragments (NTW*KSM) + accumulators + one prefetched A tile; the K > 64 and 3-column-tile variants
// need more than 256 VGPRs -> one workgroup per CU (AGPRs used as well), the others run two per CU
constexpr int pw_occ(int ksm, int nt) { return ((ksm == 32 && nt == 3) || (ksm == 64 && nt <= 2) || ksm == 128) ? 1 : 2; }
// wave grid: one 32-column tile per wave whenever the 4 waves can be spread over the column tiles (NT = 1, 2, 4):
// the W fragments then cost only KSM VGPRs per wave and 2+ workgroups fit a CU even at K = 128
constexpr int pw_wc(int nt) { return nt == 3 ? 1 : nt; }

// value of the adjacent lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float lane_swap1(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}

// bf16 storage, epilogue access pattern: a lane owns ONE column of the 32x32 MFMA output tile (16 rows in 16 registers), which as
// 2-byte elements means 16 sub-dword accesses per lane and tile.  Lanes (2i, 2i+1) own adjacent columns: the even lane takes the
// EVEN registers (rows) of both columns, the odd lane the odd ones -- 8 four-byte accesses per lane instead of 16 two-byte ones.
//   load:  lane reads the pair (col even, col odd) of ITS rows, hands the half that belongs to the partner over by DPP
//   store: lane receives the partner's value of ITS rows by DPP and writes the pair
// (N is even and the pair's first column is even, so a pair is valid or invalid as a whole; 4-byte loads at 2-byte-aligned
//  addresses are fine on gfx950: tools/ubench_unaligned.hip)
template <class F>
__device__ __forceinline__ void pair_load16(const __amdgpu_buffer_rsrc_t& rs, bool odd, F off_of /* (r) -> byte offset of (row(r), even col) or OOR */,
                                            float* out /*[16]*/) {
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = __builtin_amdgcn_raw_buffer_load_b32(rs, off_of(2 * q + (odd ? 1 : 0)), 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float lo = bf_lo(w[q]), hi = bf_hi(w[q]);
        const float recv = lane_swap1(odd ? lo : hi);
        out[2 * q] = odd ? recv : lo;
        out[2 * q + 1] = odd ? hi : recv;
    }
}

// BF (bf16-operand compute mode, configuration 3): same skeleton, same float32 tensors in HBM, float32 accumulation, statistics
// and epilogues, but both MFMA operands are rounded to bf16 (round-to-nearest-even) -- A on its way into LDS (after the
// prologue), W when it is packed -- and the product runs as v_mfma_f32_32x32x16_bf16: 1/8 of the float32 matrix-pipe time and
// half the W registers / LDS tile.
// ACC (instantiated for EPI == 0 only): C += product with the old values of the tile prefetched before the MFMA chain.
// MODE 0: float32; 1: BF (bf16 MFMA operands, float32 tensors); 2: BF + bf16 ACTIVATION STORAGE -- A / a_y / ey are read and C is
// written as bf16 (2- and 4-byte buffer loads, round-to-nearest-even stores), the EPI_STATS sums are those of the rounded outputs;
// prologues, accumulation, partials and coefficients as in mode 1.
template <int KSM, int NT, int PRO, int EPI, int MODE, bool ACC>
__global__ void __launch_bounds__(256, (pw_occ(KSM, NT))) pw_nn_kernel(PwArgs a) {
    constexpr bool BF = MODE >= 1, BH = MODE == 2;
    typedef typename std::conditional<BH, bf16_t, float>::type T;
    constexpr uint32_t ESZ = BH ? 2u : 4u;
    constexpr int WC = pw_wc(NT);                   // wave columns
    constexpr int WR = 4 / WC;                      // wave rows
    constexpr int NTW = NT / WC;                    // column tiles per wave
    constexpr int BM = 32 * WR;
    constexpr int LDA = 2 * KSM + 2;
    constexpr int KS16 = KSM / 8;                   // BF: K = 16 steps
    constexpr int LDB = 2 * KSM + 8;                // BF: bf16 elements per LDS row (16-byte fragment reads, conflict-free)
    constexpr int NA2 = BM * KSM / 256;             // float2 loads per thread per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [BM][LDA]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wc = wave / WR;
    const int lrow = lane & 31, lk = lane >> 5;
    const int g = blockIdx.x / a.nbpg, b = blockIdx.x % a.nbpg;
    const int nb0 = blockIdx.y * 128;               // column block (N > 128: stage-2 convs, 232 = 128 + 104)
    const int K = a.K, N = a.N, K2 = K >> 1;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int tiles_g = (a.Mg + BM - 1) / BM;
    // even split of the group's tiles over its nbpg workgroups (grid = a whole number of resident waves of blocks)
    const int t0 = (int)((int64_t)b * tiles_g / a.nbpg), t1 = (int)((int64_t)(b + 1) * tiles_g / a.nbpg);

    // ---- W fragments -> registers (once)
    float breg[BF ? 1 : NTW][BF ? 1 : KSM];
    bf16x8 bregb[BF ? NTW : 1][BF ? KS16 : 1];
    if (BF) {
        const __bf16* wb = reinterpret_cast<const __bf16*>(a.Wp);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int ct = (nb0 >> 5) + wc + j * WC;
#pragma unroll
            for (int s = 0; s < KS16; ++s)
                bregb[j][s] = *reinterpret_cast<const bf16x8*>(wb + ((((int64_t)ct * KS16 + s) * 2 + lk) * 32 + lrow) * 8);
        }
    } else if (a.Wp) {
        // pre-packed once per weight version: the lane's KSM fragment values are contiguous -> KSM/4 16-byte loads instead of
        // KSM dependent 4-byte loads (the strided form was ~9 us of every launch: isolated 1x1 conv 24.8 -> see DESIGN.md)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const float* wp = a.Wp + ((int64_t)(((nb0 >> 5) + wc + j * WC) * 2 + lk) * 32 + lrow) * KSM;
#pragma unroll
            for (int s = 0; s < KSM; s += 4) {
                const float4 v = *reinterpret_cast<const float4*>(wp + s);
                breg[j][s] = v.x;
                breg[j][s + 1] = v.y;
                breg[j][s + 2] = v.z;
                breg[j][s + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = nb0 + (wc + j * WC) * 32 + lrow;
#pragma unroll
            for (int s = 0; s < KSM; ++s) {
                const int k = 2 * s + lk;
                breg[j][s] = (k < K && n < N) ? a.W[(int64_t)k * a.sbk + (int64_t)n * a.sbn] : 0.0f;
            }
        }
    }
    float bv[NTW], emean[NTW], einv[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = nb0 + (wc + j * WC) * 32 + lrow;
        bv[j] = (a.bias && n < N) ? a.bias[n] : 0.0f;
        emean[j] = einv[j] = 0.0f;
        if (EPI == 2 && n < N) {
            emean[j] = a.epi_stats[0 * a.G * N + g * N + n];
            einv[j] = a.epi_stats[1 * a.G * N + g * N + n];
        }
    }
    // prologue coefficients: a thread always handles the same two k columns (256 % KSM == 0)
    const int kk_t = tid % KSM;                     // float2 column of this thread
    const bool kon = kk_t < K2;
    float psc0 = 1.0f, psc1 = 1.0f, psh0 = 0.0f, psh1 = 0.0f;
    // PRO_BNBWD: the 7 per-column coefficients stay in LDS (registers are the scarce resource of this kernel)
    float* qc = smem + BM * LDA;                    // [7][2*KSM]: mean, invstd, scale, shift, k1, k2, k3
    int dcol0 = 0, dcol1 = 0;
    // ... except in the epilogue-free variant, which has the VGPRs to spare: 14 LDS reads less per element pair
    constexpr bool QREG = PRO == 2 && EPI == 0 && (NT == 2 || NT == 4);
    float qr[QREG ? 7 : 1][2];
    if (PRO == 1 && kon) {
        const int k = 2 * kk_t, GK = a.G * K, o = g * K + k;
        psc0 = a.pro_stats[2 * GK + o];
        psc1 = a.pro_stats[2 * GK + o + 1];
        psh0 = a.pro_stats[3 * GK + o];
        psh1 = a.pro_stats[3 * GK + o + 1];
    }
    if (PRO == 2) {
        const int GK = a.G * K;
        for (int i = tid; i < 2 * KSM; i += 256) {
            const bool okk = i < K;
#pragma unroll
            for (int q = 0; q < 4; ++q) qc[q * 2 * KSM + i] = okk ? a.pro_stats[q * GK + g * K + i] : 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) qc[(4 + q) * 2 * KSM + i] = okk ? a.pro_coef[q * GK + g * K + i] : 0.0f;
        }
        if (QREG && kon) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                qr[q][0] = a.pro_stats[q * GK + g * K + 2 * kk_t];
                qr[q][1] = a.pro_stats[q * GK + g * K + 2 * kk_t + 1];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                qr[4 + q][0] = a.pro_coef[q * GK + g * K + 2 * kk_t];
                qr[4 + q][1] = a.pro_coef[q * GK + g * K + 2 * kk_t + 1];
            }
        }
        if (kon) {
            dcol0 = a.A.coff + 2 * kk_t;
            dcol1 = dcol0 + 1;
            if (a.a_shuffle) {
                dcol0 = shuffle_dst(dcol0, a.a_shuffle);
                dcol1 = shuffle_dst(dcol1, a.a_shuffle);
            }
        }
        __syncthreads();
    }
    double cs0 = 0.0, cs1 = 0.0;                    // PRO_BNBWD: column sums of the transformed A
    double s1[NTW], s2[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) s1[j] = s2[j] = 0.0;

    // the next A tile is prefetched into registers while the MFMAs of the current one run (a second register set,
    // two tiles ahead, was measured slower: it costs a wave of occupancy, 25.6 -> 29.4 us at K = N = 58)
    const bool dz_vec = PRO == 2 && !a.a_shuffle && (a.A.ld % 2 == 0) && (a.A.coff % 2 == 0) &&
                        ((reinterpret_cast<uintptr_t>(a.A.p) & 7) == 0);
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    float2 ra0[NA2], ry0[PRO == 2 ? NA2 : 1];
    // Tile loads through buffer descriptors: a thread's rows of a tile are r = tid / KSM + (256 / KSM) * i, so the row
    // offset of load i is wave-uniform (SGPR soffset) and the thread's byte offset inside it is a constant voffset --
    // no 64-bit address arithmetic and no address registers (the <64, 4, 2, 2> variant spilled 41 VGPRs with flat loads).
    // Masked lanes (k beyond K, rows beyond the group in the last tile) point out of range and read 0.
    constexpr int RSTEP = 256 / KSM;                 // rows between consecutive loads of a thread
    const uint32_t OOR = 0x80000000u;               // host checks that the operands are < 2 GB
    const int r_t = tid / KSM;
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A.p), 0, (int)(Mtot * a.A.ld * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(PRO == 2 ? a.a_y : a.A.p), 0, (int)(Mtot * (PRO == 2 ? K : a.A.ld) * ESZ), 0x00020000);
    // bf16 storage: descriptors for the epilogue's raw-BN-input tile (EPI_BNRED) and the old output tile (ACC)
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BH && EPI == 2) ? a.ey : a.A.p), 0,
                                                                         (int)(Mtot * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.C.p, 0, (int)(Mtot * a.C.ld * 2), 0x00020000);
    uint32_t voA0 = OOR, voA1 = OOR, voY = OOR;
    if (kon) {
        if (PRO == 2) {
            voA0 = (uint32_t)(r_t * a.A.ld + dcol0) * ESZ;
            voA1 = (uint32_t)(r_t * a.A.ld + dcol1) * ESZ;
            voY = (uint32_t)(r_t * K + 2 * kk_t) * ESZ;
        } else {
            voA0 = (uint32_t)(r_t * a.A.ld + a.A.coff + 2 * kk_t) * ESZ;
        }
    }
    const uint32_t rowA = (uint32_t)a.A.ld * ESZ, rowY = (uint32_t)K * ESZ;
    // Tile loads return RAW words: with bf16 storage the widening (shift / mask) happens in store_tile, not here -- a conversion
    // right behind its load is a use of the loaded register, i.e. an s_waitcnt in the middle of the prefetch (the first bf16-storage
    // version converted on load and waited for every load of the next tile before the MFMA chain of the current one).
    //   float32 tensors: ld2 = the two floats, ld1 = the float.
    //   bf16 storage:    ld2 = ONE dword (two bf16) in .x, ld1 = the zero-extended 16 bits (as float bit patterns)
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t so) -> float2 {
        if (BH) return make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0)), 0.0f);
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
        return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
    };
    auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t so) -> float {
        if (BH) return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, 0));
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
    };
    // decode of a raw tile word pair (no-ops for float32 tensors)
    auto dec2 = [&](float2 r) -> float2 {          // from ld2
        if (!BH) return r;
        const uint32_t w = __float_as_uint(r.x);
        return make_float2(bf_lo(w), bf_hi(w));
    };
    auto dec11 = [&](float2 r) -> float2 {         // from two ld1
        if (!BH) return r;
        return make_float2(__uint_as_float(__float_as_uint(r.x) << 16), __uint_as_float(__float_as_uint(r.y) << 16));
    };
    auto load_tile = [&](int t, float2* ra, float2* ry) {
        const int64_t m0 = mbeg + (int64_t)t * BM;
        const uint32_t mu = (uint32_t)m0;
        const bool full = m0 + BM <= mend;
        // (the dz_vec test is hoisted out of the load loop: inside it every load was its own basic block)
        if (PRO == 2 && !dz_vec) {
#pragma unroll
            for (int i = 0; i < NA2; ++i) {
                const uint32_t r = mu + (uint32_t)(i * RSTEP);
                uint32_t msk = 0u;
                if (!full) msk = (m0 + i * RSTEP + r_t) < mend ? 0u : OOR;      // last tile of the group only
                ra[i].x = ld1(rsA, voA0 | msk, r * rowA);
                ra[i].y = ld1(rsA, voA1 | msk, r * rowA);
                ry[PRO == 2 ? i : 0] = ld2(rsY, voY | msk, r * rowY);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA2; ++i) {
                const uint32_t r = mu + (uint32_t)(i * RSTEP);
                uint32_t msk = 0u;
                if (!full) msk = (m0 + i * RSTEP + r_t) < mend ? 0u : OOR;
                ra[i] = ld2(rsA, voA0 | msk, r * rowA);       // dense gradient / plain operand: one 8-byte (bf16: 4-byte) load
                if (PRO == 2) ry[PRO == 2 ? i : 0] = ld2(rsY, voY | msk, r * rowY);
            }
        }
    };
    auto store_tile = [&](int t, const float2* ra, const float2* ry) {
        const int64_t m0 = mbeg + (int64_t)t * BM;
#pragma unroll
        for (int i = 0; i < NA2; ++i) {
            const int r = (tid + 256 * i) / KSM;
            float2 v = (PRO == 2 && !dz_vec) ? dec11(ra[i]) : dec2(ra[i]);
            if (PRO == 1) {
                // unconditional: columns beyond K carry scale 1 / shift 0 (and loaded 0), rows beyond the group end only feed
                // output rows that are neither stored nor counted -- and a compound lane condition in front of a select is the
                // mask pattern of "What round 3 found" 
                v.x = fmaf(psc0, v.x, psh0);
                v.y = fmaf(psc1, v.y, psh1);
            } else if (PRO == 2) {
                if (m0 + r < mend && kon) {
                    const float2 yv = dec2(ry[i]);
                    if (QREG) {
                        if (a.a_act == ACT_RELU6) {
                            const float z0 = fmaf(qr[2][0], yv.x, qr[3][0]), z1 = fmaf(qr[2][1], yv.y, qr[3][1]);
                            if (!relu6_open(z0)) v.x = 0.0f;
                            if (!relu6_open(z1)) v.y = 0.0f;
                        }
                        const float xh0 = (yv.x - qr[0][0]) * qr[1][0], xh1 = (yv.y - qr[0][1]) * qr[1][1];
                        v.x = qr[4][0] * (v.x - qr[5][0] - xh0 * qr[6][0]);
                        v.y = qr[4][1] * (v.y - qr[5][1] - xh1 * qr[6][1]);
                        cs0 += (double)v.x;
                        cs1 += (double)v.y;
                        if (BF) {
                            bf16x2 h;
                            h[0] = (__bf16)v.x;
                            h[1] = (__bf16)v.y;
                            *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(As) + r * LDB + 2 * kk_t) = h;
                        } else {
                            *reinterpret_cast<float2*>(&As[r * LDA + 2 * kk_t]) = v;
                        }
                        continue;
                    }
                    const float* q0 = qc + 2 * kk_t;
                    if (a.a_act == ACT_RELU6) {
                        const float z0 = fmaf(q0[2 * 2 * KSM], yv.x, q0[3 * 2 * KSM]);
                        const float z1 = fmaf(q0[2 * 2 * KSM + 1], yv.y, q0[3 * 2 * KSM + 1]);
                        if (!relu6_open(z0)) v.x = 0.0f;
                        if (!relu6_open(z1)) v.y = 0.0f;
                    }
                    const float xh0 = (yv.x - q0[0]) * q0[2 * KSM], xh1 = (yv.y - q0[1]) * q0[2 * KSM + 1];
                    v.x = q0[4 * 2 * KSM] * (v.x - q0[5 * 2 * KSM] - xh0 * q0[6 * 2 * KSM]);
                    v.y = q0[4 * 2 * KSM + 1] * (v.y - q0[5 * 2 * KSM + 1] - xh1 * q0[6 * 2 * KSM + 1]);
                    cs0 += (double)v.x;
                    cs1 += (double)v.y;
                }
            }
            if (BF) {
                bf16x2 h;
                h[0] = (__bf16)v.x;
                h[1] = (__bf16)v.y;
                *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(As) + r * LDB + 2 * kk_t) = h;
            } else {
                *reinterpret_cast<float2*>(&As[r * LDA + 2 * kk_t]) = v;
            }
        }
    };
    auto compute_tile = [&](int t) {
        f32x16 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        // EPI_BNRED: the raw BN input of this output tile is fetched BEFORE the MFMA chain (its latency hides behind the
        // matrix work instead of sitting in the epilogue: 52 -> 22 us gap to the epilogue-free kernel otherwise)
        float eyv[EPI == 2 ? NTW : 1][EPI == 2 ? 16 : 1];
        if (EPI == 2) {
            const int64_t me0 = mbeg + (int64_t)t * BM + wr * 32;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = nb0 + (wc + j * WC) * 32 + lrow;
                if constexpr (BH && EPI == 2) {         // paired 4-byte buffer loads, unconditional (out-of-range offsets read 0)
                    pair_load16(rsE, (lane & 1) != 0, [&](int r) -> uint32_t {
                        const int64_t m = me0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        return (n < N && m < mend) ? (uint32_t)((m * N + (n & ~1)) * 2) : OOR;
                    }, &eyv[EPI == 2 ? j : 0][0]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = me0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        eyv[j][r] = (n < N && m < mend) ? a.ey[m * N + n] : 0.0f;
                    }
                }
            }
        }
        // accumulate (the stride-2 units' first conv adds its input gradient to the shortcut branch's): the old values of the
        // output tile are fetched BEFORE the MFMA chain as well -- read in the epilogue they were 16 dependent scalar round trips
        // per tile with nothing to hide behind: 124 vs 56 us for the 24-channel backward-data conv of the first unit (isolated)
        constexpr bool ACC_PF = ACC && NTW == 1;      // (three column tiles per wave: 48 more VGPRs would spill -> plain read-modify-write)
        float cold[ACC_PF ? NTW : 1][ACC_PF ? 16 : 1];
        if (ACC_PF) {
            const int64_t mc0 = mbeg + (int64_t)t * BM + wr * 32;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = nb0 + (wc + j * WC) * 32 + lrow;
                if constexpr (BH && ACC_PF) {
                    pair_load16(rsC, (lane & 1) != 0, [&](int r) -> uint32_t {
                        const int64_t m = mc0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        return (n < N && m < mend) ? (uint32_t)((m * a.C.ld + a.C.coff + (n & ~1)) * 2) : OOR;
                    }, &cold[ACC_PF ? j : 0][0]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = mc0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        cold[ACC_PF ? j : 0][ACC_PF ? r : 0] = (n < N && m < mend) ? a.C.p[m * a.C.ld + a.C.coff + n] : 0.0f;
                    }
                }
            }
        }
        if (BF) {
            const __bf16* arow = reinterpret_cast<const __bf16*>(As) + (wr * 32 + lrow) * LDB + 8 * lk;
#pragma unroll
            for (int s = 0; s < KS16; ++s) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bregb[j][s], acc[j], 0, 0, 0);
            }
        } else {
            const float* arow = &As[(wr * 32 + lrow) * LDA + lk];
#pragma unroll
            for (int s = 0; s < KSM; ++s) {
                // keep the scheduler from hoisting all KSM fragment reads above the MFMA chain (64 extra live VGPRs)
                if (KSM > 16 && (s % 16) == 0) __builtin_amdgcn_sched_barrier(0);
                const float av = arow[2 * s];
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[j][s], acc[j], 0, 0, 0);
            }
        }
        // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int64_t m0 = mbeg + (int64_t)t * BM + wr * 32;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = nb0 + (wc + j * WC) * 32 + lrow;
            if (n >= N) continue;
            if constexpr (BH) {
                if (ACC_PF || !a.accumulate) {          // paired 4-byte stores (see pair_load16)
                    const bool odd = (lane & 1) != 0;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        v[r] = acc[j][r] + bv[j];
                        if (EPI == 1) v[r] = (float)(bf16_t)v[r];       // statistics of the stored (rounded) values
                        if (m < mend) {
                            if (EPI == 1) {
                                s1[j] += (double)v[r];
                                s2[j] += (double)v[r] * (double)v[r];
                            } else if (EPI == 2) {
                                const float xh = (eyv[EPI == 2 ? j : 0][EPI == 2 ? r : 0] - emean[j]) * einv[j];
                                s1[j] += (double)v[r];
                                s2[j] += (double)v[r] * (double)xh;
                            }
                        }
                        if (ACC_PF) v[r] += cold[ACC_PF ? j : 0][ACC_PF ? r : 0];
                    }
                    bf16_t* cp = vptr<bf16_t>(a.C) + a.C.coff + (n & ~1);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float recv = lane_swap1(odd ? v[2 * q] : v[2 * q + 1]);
                        const float lo = odd ? recv : v[2 * q], hi = odd ? v[2 * q + 1] : recv;
                        const int r = 2 * q + (odd ? 1 : 0);
                        const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m < mend) *reinterpret_cast<uint32_t*>(cp + m * a.C.ld) = bf_pack(lo, hi);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < mend) {
                    T* c = vptr<T>(a.C) + m * a.C.ld + a.C.coff + n;
                    float v = acc[j][r] + bv[j];
                    if (EPI == 1) {
                        if (BH) v = (float)(bf16_t)v;       // statistics of the stored (rounded) values
                        s1[j] += (double)v;
                        s2[j] += (double)v * (double)v;
                    } else if (EPI == 2) {
                        const float xh = (eyv[j][r] - emean[j]) * einv[j];
                        s1[j] += (double)v;
                        s2[j] += (double)v * (double)xh;
                    }
                    if (ACC_PF) v += cold[ACC_PF ? j : 0][ACC_PF ? r : 0];
                    else if (a.accumulate) v += ldf(c);
                    stf(c, v);
                }
            }
        }
    };

    if (t0 < t1) load_tile(t0, ra0, ry0);
    for (int t = t0; t < t1; ++t) {
        store_tile(t, ra0, ry0);
        __syncthreads();
        if (t + 1 < t1) load_tile(t + 1, ra0, ry0);
        compute_tile(t);
        __syncthreads();             // all fragment reads of As done before the next store_tile
    }
    if (PRO == 2 && a.part2 && blockIdx.y == 0) {
        // column sums of the transformed A: threads tid = kk + KSM*j share a column pair
        double* red2 = reinterpret_cast<double*>(smem);     // [256/KSM][2*KSM]
        red2[(tid / KSM) * 2 * KSM + 2 * kk_t] = cs0;
        red2[(tid / KSM) * 2 * KSM + 2 * kk_t + 1] = cs1;
        __syncthreads();
        for (int k = tid; k < K; k += 256) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 256 / KSM; ++j) s += red2[j * 2 * KSM + k];
            a.part2[((int64_t)g * a.nbpg + b) * K + k] = s;
        }
        __syncthreads();
    }
    if (EPI != 0) {
        double* red = reinterpret_cast<double*>(smem);      // [WR][2][32*NT]; As is dead
        constexpr int NP = 32 * NT;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (lk == 0) {
                const int nl = (wc + j * WC) * 32 + lrow;
                red[(wr * 2 + 0) * NP + nl] = s1[j];
                red[(wr * 2 + 1) * NP + nl] = s2[j];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * NP; i += 256) {
            const int q = i / NP, nl = i % NP;
            if (nb0 + nl < N) {
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < WR; ++w) s += red[(w * 2 + q) * NP + nl];
                a.part[(((int64_t)g * a.nbpg + b) * 2 + q) * N + nb0 + nl] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int pw_ksm(int K) { return K <= 32 ? 16 : (K <= 64 ? 32 : (K <= 128 ? 64 : 128)); }

// B(k, n) = w[k * sbk + n * sbn] -> fragment order [ct = n / 32][lk = k & 1][n & 31][s = k >> 1] (zero padded to KSM, 32):
// what lane (lrow = n & 31, lk) of the wave that owns column tile ct keeps in registers.  One launch for all convs of a pass.
__global__ void __launch_bounds__(256) pw_pack_many_kernel(const PwPack* __restrict__ tab) {
    const PwPack d = tab[blockIdx.y];
    pw_pack_body(d, blockIdx.x, gridDim.x);
}

// All packing work of a training pass as ONE launch: blockIdx.y walks the split-precision GEMM packs, the float32-MFMA packs, the
// split-precision conv packs and the W^T transposes of the backward (pack_bodies.h).
__global__ void __launch_bounds__(256) pack_all_kernel(const GemmX3Pack* __restrict__ tg, int ng, const PwPack* __restrict__ tp, int np,
                                                       const PwX3Pack* __restrict__ t3, int n3, const PwTranspose* __restrict__ tt, int nt) {
    __shared__ float tile[32][33];
    int y = blockIdx.y;
    if (y < ng) {
        const GemmX3Pack d = tg[y];
        gemm_x3_pack_body(d, blockIdx.x, gridDim.x);
        return;
    }
    y -= ng;
    if (y < np) {
        const PwPack d = tp[y];
        pw_pack_body(d, blockIdx.x, gridDim.x);
        return;
    }
    y -= np;
    if (y < n3) {
        const PwX3Pack d = t3[y];
        pw_x3_pack_body(d, blockIdx.x, gridDim.x);
        return;
    }
    y -= n3;
    if (y < nt) {
        const PwTranspose d = tt[y];
        transpose_body(d, blockIdx.x, tile);
    }
}

int pack_all(const GemmX3Pack* tg, int ng, const PwPack* tp, int np, const PwX3Pack* t3, int n3, const PwTranspose* tt, int nt,
             int transpose_tiles, hipStream_t st) {
    const int rows = ng + np + n3 + nt;
    if (rows <= 0) return 0;
    const int gx = std::max(32, nt > 0 ? transpose_tiles : 0);
    hipLaunchKernelGGL(pack_all_kernel, dim3(gx, rows), dim3(256), 0, st, tg, ng, tp, np, t3, n3, tt, nt);
    CDRL_LAUNCH_CHECK();
    return 0;
}

int64_t pw_packed_elems(int N, int K) { return (int64_t)cdiv(N, 32) * 2 * 32 * pw_ksm(K); }

PwPack pw_pack_entry(const float* w, float* wp, int K, int N, int sbk, int sbn, bool bf16) {
    PwPack e;
    e.bf16 = bf16 ? 1 : 0;
    e.w = w;
    e.wp = wp;
    e.K = K;
    e.N = N;
    e.sbk = sbk;
    e.sbn = sbn;
    e.ksm = pw_ksm(K);
    e.ntiles = cdiv(N, 32);
    return e;
}

int pw_pack_many(const PwPack* tab_dev, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pw_pack_many_kernel, dim3(16, n), dim3(256), 0, st, tab_dev);
    CDRL_LAUNCH_CHECK();
    return 0;
}

bool pw_nn_supported(View A, int N, int K) {
    if (K > 256 || N > 256 || (K & 1) || N < 1) return false;
    const int nt = N > 128 ? 4 : cdiv(N, 32), ksm = pw_ksm(K);
    if (nt == 3 && ksm > 32) return false;
    if (ksm == 128 && nt != 4) return false;        // K > 128 is instantiated for 4 column tiles per block only
    return (A.ld % 2 == 0) && (A.coff % 2 == 0) && ((reinterpret_cast<uintptr_t>(A.p) & 7) == 0);
}

PwPlan pw_nn_plan(int G, int Mg, int N, int K) {
    PwPlan p;
    const int nt = N <= 128 ? cdiv(N, 32) : 4, ksm = pw_ksm(K);     // N > 128: column blocks of 128 (grid.y)
    p.bm = 32 * (4 / pw_wc(nt));
    const int tiles_g = cdiv(Mg, p.bm);
    // grid = exactly the number of workgroups that are resident at once (256 CUs x occupancy of the variant), so every
    // CU gets the same share; the tiles of a group are split evenly over its workgroups
    const int occ = pw_occ(ksm, nt) == 1 ? 1 : ((ksm <= 32 && nt != 1) ? 3 : 2);
    int target = 256 * occ / (G * cdiv(N, 128));
    if (target < 1) target = 1;
    p.nbpg = tiles_g < target ? tiles_g : target;
    p.tpb = cdiv(tiles_g, p.nbpg);
    return p;
}

template <int KSM, int NT, int PRO, int EPI, int MODE, bool ACC>
static int launch_pw_bf(const PwArgs& a, hipStream_t st) {
    constexpr int WR = 4 / pw_wc(NT);
    constexpr int BM = 32 * WR;
    size_t lds = (size_t)(BM * (2 * KSM + 2) + (PRO == 2 ? 14 * KSM : 0)) * sizeof(float);
    const size_t red = (size_t)WR * 2 * 32 * NT * sizeof(double);
    if (lds < red) lds = red;
    if (lds < (size_t)512 * sizeof(double)) lds = (size_t)512 * sizeof(double);      // PRO_BNBWD column-sum scratch
    auto kern = pw_nn_kernel<KSM, NT, PRO, EPI, MODE, ACC>;
    static size_t allowed = 64 * 1024;          // per instantiation: one runtime call per kernel and size, not one per launch
    if (lds > allowed) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    hipLaunchKernelGGL(kern, dim3(a.G * a.nbpg, cdiv(a.N, 128)), dim3(256), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int KSM, int NT, int PRO, int EPI>
static int launch_pw(const PwArgs& a, hipStream_t st) {
    if constexpr (EPI == 0) {
        if (a.accumulate) {
            if (a.at) return launch_pw_bf<KSM, NT, PRO, EPI, 2, true>(a, st);
            return a.bf ? launch_pw_bf<KSM, NT, PRO, EPI, 1, true>(a, st) : launch_pw_bf<KSM, NT, PRO, EPI, 0, true>(a, st);
        }
    }
    if (a.at) return launch_pw_bf<KSM, NT, PRO, EPI, 2, false>(a, st);
    return a.bf ? launch_pw_bf<KSM, NT, PRO, EPI, 1, false>(a, st) : launch_pw_bf<KSM, NT, PRO, EPI, 0, false>(a, st);
}

template <int KSM, int NT>
static int launch_pw_pe(int pro, int epi, const PwArgs& a, hipStream_t st) {
    if (pro == 0 && epi == 0) return launch_pw<KSM, NT, 0, 0>(a, st);
    if (pro == 0 && epi == 1) return launch_pw<KSM, NT, 0, 1>(a, st);
    if (pro == 0 && epi == 2) return launch_pw<KSM, NT, 0, 2>(a, st);
    if (pro == 1 && epi == 0) return launch_pw<KSM, NT, 1, 0>(a, st);
    if (pro == 1 && epi == 1) return launch_pw<KSM, NT, 1, 1>(a, st);
    if (pro == 2 && epi == 0) return launch_pw<KSM, NT, 2, 0>(a, st);
    if (pro == 2 && epi == 2) return launch_pw<KSM, NT, 2, 2>(a, st);
    set_error("pw_nn: unsupported prologue/epilogue combination %d/%d", pro, epi);
    return -1;
}

template <int KSM>
static int launch_pw_nt(int nt, int pro, int epi, const PwArgs& a, hipStream_t st) {
    switch (nt) {
        case 1: return launch_pw_pe<KSM, 1>(pro, epi, a, st);
        case 2: return launch_pw_pe<KSM, 2>(pro, epi, a, st);
        case 3:
            if constexpr (KSM <= 32) return launch_pw_pe<KSM, 3>(pro, epi, a, st);
            break;
        case 4: return launch_pw_pe<KSM, 4>(pro, epi, a, st);
    }
    set_error("pw_nn: unsupported tile shape");
    return -1;
}

int pw_nn(View A, const float* pro_stats, const float* W, int sbk, int sbn, const float* bias, View C, int accumulate, int G,
          int Mg, int N, int K, int epilogue, const float* ey, const float* epi_stats, double* part, hipStream_t st,
          const PwBnBwd* bb, const float* Wp, bool wp_bf16, int at) {
    if (!pw_nn_supported(bb ? make_view(const_cast<float*>(bb->y), K) : A, N, K)) {
        set_error("pw_nn: shape K=%d N=%d / alignment not supported", K, N);
        return -1;
    }
    if (epilogue != 0 && !part) {
        set_error("pw_nn: epilogue needs a partial buffer");
        return -1;
    }
    if ((int64_t)G * Mg * A.ld * 4 >= (1ll << 31) || (int64_t)G * Mg * K * 4 >= (1ll << 31)) {     // 32-bit buffer offsets
        set_error("pw_nn: operands of 2 GB or more are not supported (rows=%lld)", (long long)G * Mg);
        return -1;
    }
    const PwPlan p = pw_nn_plan(G, Mg, N, K);
    PwArgs a;
    a.A = A;
    a.a_shuffle = bb ? bb->shuffle_ctot : 0;
    a.a_act = bb ? bb->act : 0;
    a.a_y = bb ? bb->y : nullptr;
    a.pro_coef = bb ? bb->coef : nullptr;
    a.part2 = bb ? bb->part2 : nullptr;
    if (bb) pro_stats = bb->stats;
    a.pro_stats = pro_stats;
    a.W = W;
    a.sbk = sbk;
    a.sbn = sbn;
    a.Wp = Wp;
    a.bf = (wp_bf16 && Wp) ? 1 : 0;
    a.at = at ? 1 : 0;
    if (a.at && !a.bf) {
        set_error("pw_nn: bf16 activation storage needs the bf16-operand variant (packed bf16 weights)");
        return -1;
    }
    a.bias = bias;
    a.C = C;
    a.accumulate = accumulate;
    a.ey = ey;
    a.epi_stats = epi_stats;
    a.part = part;
    a.N = N;
    a.K = K;
    a.G = G;
    a.Mg = Mg;
    a.nbpg = p.nbpg;
    a.tpb = p.tpb;
    const int nt = N > 128 ? 4 : cdiv(N, 32), pro = bb ? 2 : (pro_stats ? 1 : 0);
    switch (pw_ksm(K)) {
        case 16: return launch_pw_nt<16>(nt, pro, epilogue, a, st);
        case 32: return launch_pw_nt<32>(nt, pro, epilogue, a, st);
        case 64: return launch_pw_nt<64>(nt, pro, epilogue, a, st);
        default: return launch_pw_pe<128, 4>(pro, epilogue, a, st);
    }
}

}  // namespace cdrl
