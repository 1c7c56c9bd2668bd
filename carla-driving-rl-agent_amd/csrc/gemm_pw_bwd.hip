// Backward of a pointwise (1x1) convolution as ONE pass over its operands (gfx950, v_mfma_f32_32x32x16_bf16 on exact three-way
// bf16 splits of float32 operands, see gemm_pw_x3.hip).
//
// Reference: the gradient of Conv2D(k=1) -> BatchNormalization (core/architectures.py:130-141) inside tape.gradient
// (core/carla_agent.py:364-365).  For y = a W + b followed by a train-mode BatchNorm whose output gradient is dz:
//     dy = k1 (mask dz - k2 - xhat(y) k3)          BatchNorm-backward apply (never written to HBM)
//     da = dy W^T                                  backward-data   -> HBM
//     dW = a^T dy,  db = column sums of dy         filter / bias gradient
// Rounds 1-3 ran backward-data (gemm_pw.hip, critical stream) and the filter gradient (gemm_tn_direct.hip, side stream) as two
// kernels that each load (dz, y) and apply the prologue; the side-stream kernel cost the update-step 1.6 ms of contention plus two
// event records per unit.  Here a workgroup stages the tile ONCE: 512 threads = 8 waves, the first four load (dz, y), build dy and
// multiply it with W^T (fragments in registers), the other four load the conv input `a` and accumulate a^T dy over all the tiles
// of the workgroup in registers; both products read the same LDS planes.
// When `a` is itself a BatchNorm output that is applied on load (unit conv 2: a = gamma xhat(y2) + beta, no activation), the
// second product is taken against xhat:  Q = xhat^T dy  per time slice, and everything downstream follows from Q in the reduce
// kernel, with no pass over da:
//     dW[k, n]            = gamma[k] sum_g Q_g[k, n] + beta[k] db[n]
//     sum_r da[r, k]      = sum_n W[k, n] db_g[n]                       (BatchNorm-backward sums of the BN that produced a:
//     sum_r da[r, k] xhat = sum_n W[k, n] Q_g[k, n]                      the EPI_BNRED epilogue of gemm_pw.hip, for free)
// Partials: one [KP][NP] float tile + [NP] doubles per workgroup, combined in fixed order by pwb_reduce_kernel (bit-wise
// reproducible, no atomics).
#include <stdlib.h>

#include "colreduce.h"

namespace cdrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));     // arithmetic on pairs compiles to v_pk_{add,mul,fma}_f32

struct PwbArgs {
    View dz;                // gradient w.r.t. the BatchNorm output (columns optionally gathered through the shuffle map)
    int dz_shuffle;         // ctot of the channel shuffle, 0 = none
    int act;                // ACT_RELU6: mask from the BatchNorm output
    const float* y;         // raw BatchNorm input = conv output [M][N] dense
    const float* stats;     // [4][G][N]
    const float* coef;      // [3][G][N]
    View a;                 // conv input [M][K]
    const float* a_stats;   // [4][G][K] (ANORM) or null
    const __bf16* Wp;       // W^T as three planes of MFMA B fragments: pw_x3 packing of B(k = n_out, n = k_in)
    int wp_ks;              // K = 16 steps per plane of that pack (pw_x3_ksteps(n_out): 2 for n_out <= 32 -- fewer than NP / 16; the rest are zero)
    View da;
    float* qpart;           // [G][nbpg][KP][NP]
    double* dbpart;         // [G][nbpg][NP]
    double* spart;          // bf16 storage + ANORM: [G][nbpg][2][KP] (sum da, sum da * xhat) taken from the accumulators in double
    int N, K, G, Mg, nbpg;
    // float32 form: the BatchNorm-backward FINALIZE of the BatchNorm behind the conv done here instead of by bn_bwd_finalize in front of
    // this launch -- fin_part = that BatchNorm's [G][fin_nb][2][N] sums (sum dz, sum dz xhat); every workgroup folds the rows of its group
    // into k2 = mean(dz), k3 = mean(dz xhat) (k1 = gamma invstd is the statistics block's scale row); the first workgroup of a group
    // leaves the group totals in fin_tot [G][2][N] for the reduce kernel (dgamma / dbeta)
    const double* fin_part;
    double* fin_tot;
    int fin_nb;
    int at;                 // host dispatch: 1 = bf16 activation storage (pwb16_kernel)
    int dbg;                // timing diagnostics (CDRL_DIAG=1 CDRL_DIAG_PWB=bits, wrong results): 1 no MFMA, 2 no LDS writes, 4 no stores, 8 no loads, 16 no partial-tile stores, 32 no W^T fragment loads
};

__device__ __forceinline__ void pwb_split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// the same split on a pair, results as packed dwords {lo = first element, hi = second}: three packed conversions, two packed
// subtractions, widening by shift / mask
__device__ __forceinline__ void pwb_split3x2(f32x2 x, uint32_t& h1, uint32_t& h2, uint32_t& h3) {
    auto widen = [](uint32_t w) -> f32x2 { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
    h1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    const f32x2 r1 = x - widen(h1);
    h2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    h3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1 - widen(h2), bf16x2));
}
// {lo16(a), lo16(b)} and {hi16(a), hi16(b)} in one v_perm_b32 each: two rows of one column from two rows of a column pair
__device__ __forceinline__ uint32_t pwb_lolo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pwb_hihi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// KP: padded input channels (k_in), NP: padded output channels (n_out); (128, 128) -> 32-row tiles, (64, 64) -> 64-row tiles.
// The two halves of the workgroup run DIFFERENT loops (scalar branch on the wave index: registers of one role are not live in
// the other) with the same barrier sequence.
template <int KP, int NP, bool SHUF, bool ANORM, bool ACC>
__global__ void __launch_bounds__(512, 1) pwb_kernel(PwbArgs a) {
    static_assert((KP == NP && (KP == 64 || KP == 128)) || (KP == 64 && NP == 128), "instantiated paddings: 64/64, 128/128, 64 -> 128");
    constexpr int BM = (KP == 128 || NP == 128) ? 32 : 64;
    constexpr int NRG = BM / 4;                     // row groups of 4
    constexpr int LDR = NP + 8;                     // row-major planes: bf16 per row (16-byte fragment reads, conflict-free)
    constexpr int LDT = BM + 8;                     // transposed planes: bf16 per column ((BM + 8) / 8 odd: conflict-free b128 reads)
    constexpr int TPD = NP * LDT, TPA = KP * LDT;   // elements of one transposed plane of dy / of a
    constexpr int KS_DA = NP / 16, KS_Q = BM / 16;
    constexpr int DA_WC = KP / 32, DA_WR = BM / 32; // backward-data tiles: DA_WC x DA_WR waves are busy (2 of 4 in the 64 -> 128 form)
    constexpr int Q_KT = KP / 32, Q_NT = NP / 32;
    constexpr int NTW = Q_KT * Q_NT / 4;            // Q tiles per wave (4 | 1 | 2): wave w owns k tile w % Q_KT, column tiles from (w / Q_KT) * NTW
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Rm = reinterpret_cast<__bf16*>(smem_raw);       // [3][BM][LDR]   dy, row-major
    __bf16* Dt = Rm + 3 * BM * LDR;                         // [3][NP][LDT]   dy, transposed
    __bf16* At = Dt + 3 * TPD;                              // [3][KP][LDT]   a (or xhat(a)), transposed
    float* cf = reinterpret_cast<float*>(At + 3 * TPA);     // [7][NP] mean, invstd, scale, shift, k1, k2, k3 of the dy columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8);     // 0: waves 0-3 (dy + backward-data), 1: waves 4-7 (a + filter product)
    const int t2 = tid & 255;
    const int rg = t2 % NRG, cg = t2 / NRG;         // micro-tile: rows 4 rg .. 4 rg + 3, columns 4 cg .. 4 cg + 3
    const int c0 = 4 * cg;
    const int g = blockIdx.x / a.nbpg, b = blockIdx.x % a.nbpg;
    const int K = a.K, N = a.N;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int tiles_g = (a.Mg + BM - 1) / BM;
    const int t0 = (int)((int64_t)b * tiles_g / a.nbpg), t1 = (int)((int64_t)(b + 1) * tiles_g / a.nbpg);
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const uint32_t OOR = 0x80000000u;

    if (a.fin_part) {
        // BatchNorm-backward finalize of this group's columns, by all eight waves before they part: wave w takes columns 16 w .. 16 w + 15,
        // lane bits 4-5 a quarter of the partial rows each (16 loads in flight), folded by two shuffles in a fixed order
        const int c = wave * 16 + (lane & 15), sl = lane >> 4;
        const int per = (a.fin_nb + 3) >> 2, b0 = sl * per, b1 = min(a.fin_nb, b0 + per);
        double s = 0.0, q = 0.0;
        if (c < N) {
            const double* ps = a.fin_part + ((int64_t)g * a.fin_nb * 2) * N + c;
            int bb = b0;
            for (; bb + 8 <= b1; bb += 8) {
                double u[8], v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    u[i] = ps[(int64_t)(bb + i) * 2 * N];
                    v[i] = ps[(int64_t)(bb + i) * 2 * N + N];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s += u[i];
                    q += v[i];
                }
            }
            for (; bb < b1; ++bb) {
                s += ps[(int64_t)bb * 2 * N];
                q += ps[(int64_t)bb * 2 * N + N];
            }
        }
        s += __shfl_xor(s, 16);
        q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (sl == 0 && c < NP) {
            const double cnt = (double)a.Mg;
            cf[5 * NP + c] = c < N ? (float)(s / cnt) : 0.0f;
            cf[6 * NP + c] = c < N ? (float)(q / cnt) : 0.0f;
            if (b == 0 && c < N) {
                a.fin_tot[((int64_t)g * 2 + 0) * N + c] = s;
                a.fin_tot[((int64_t)g * 2 + 1) * N + c] = q;
            }
        }
    }
    if (role == 0) {
        // =========================================================== dy = BatchNorm-backward(dz, y); da = dy W^T; db partials
        const int dwr = wave % DA_WR, dwc = wave / DA_WR;
        bf16x8 breg[3][KS_DA];                      // W^T fragments, once
        {
            const int n = dwc * 32 + lrow;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int s = 0; s < KS_DA; ++s) {
                    // (a pack of n_out <= 32 channels holds fewer K steps per plane than the padding NP / 16: round 6, the 24-channel
                    //  shortcut conv -- reading past it multiplied garbage, NaN included, into the input gradient)
                    bf16x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
                    breg[p][s] = z;
                    if (!(a.dbg & 32) && s < a.wp_ks) breg[p][s] = *reinterpret_cast<const bf16x8*>(a.Wp + (((int64_t)(p * a.wp_ks + s) * 2 + lk) * 128 + n) * 8);
                }
        }
        // the seven per-column coefficients live in LDS (28 registers otherwise: this role also holds the W^T fragments);
        // padded columns carry 0 everywhere, so their dy is 0
        const int GN = a.G * N;
        for (int i = tid; i < (a.fin_part ? 5 : 7) * NP; i += 256) {
            const int q = i / NP, c = i % NP;
            // (fin_part: k1 = the scale row of the statistics block, k2 / k3 were written above)
            cf[i] = c < N ? (q < 4 ? a.stats[q * GN + g * N + c] : (a.fin_part ? a.stats[2 * GN + g * N + c] : a.coef[(q - 4) * GN + g * N + c])) : 0.0f;
        }
        uint32_t vo[4], voy[2];                     // byte offsets of (row 4 rg, column) inside a tile, or OOR
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            const int dc = SHUF ? shuffle_dst(a.dz.coff + c, a.dz_shuffle) : a.dz.coff + c;
            vo[e] = c < N ? (uint32_t)((4 * rg) * a.dz.ld + dc) * 4u : OOR;
        }
        voy[0] = c0 < N ? (uint32_t)((4 * rg) * N + c0) * 4u : OOR;
        voy[1] = c0 + 2 < N ? (uint32_t)((4 * rg) * N + c0 + 2) * 4u : OOR;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(a.dz.p, 0, (int)(Mtot * a.dz.ld * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y), 0, (int)(Mtot * N * 4), 0x00020000);
        const uint32_t rowD = (uint32_t)a.dz.ld * 4u, rowY = (uint32_t)N * 4u;
        const bool relu6 = a.act == ACT_RELU6;
        f32x2 rz[4][2], ry[4][2];                   // raw tile registers: dz, y of the micro-tile as column pairs
        // Row j of tile t -> rz[j], ry[j].  The rows of the NEXT tile are requested from inside store_tile, each as soon as the
        // prologue has consumed the registers of the current one: a load then has the LDS writes, both barriers and the whole MFMA
        // phase to arrive (issued after the first barrier instead, ~10 of 40 us were exposed load latency).
        auto load_row = [&](int t, int j) {
            if (a.dbg & 8) return;
            const int64_t m0 = mbeg + (int64_t)t * BM;
            const uint32_t mu = (uint32_t)m0;
            // rows of this micro-tile beyond the group (ragged last tile): read 0, zeroed again after the prologue
            const int left = (int)(mend - (m0 + 4 * rg));
            {
                const uint32_t msk = j < left ? 0u : OOR;
                if (SHUF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rz[j][e >> 1][e & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsD, (vo[e] + (uint32_t)j * rowD) | msk, mu * rowD, 0));
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsD, (vo[2 * h] + (uint32_t)j * rowD) | msk, mu * rowD, 0);
                        rz[j][h] = __builtin_bit_cast(f32x2, v);
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsY, (voy[h] + (uint32_t)j * rowY) | msk, mu * rowY, 0);
                    ry[j][h] = __builtin_bit_cast(f32x2, v);
                }
            }
        };
        double cs[4] = {0.0, 0.0, 0.0, 0.0};        // column sums of dy (bias gradient)
        auto store_tile = [&](int t) {
            const bool more = t + 1 < t1;
            const int left = (int)(mend - (mbeg + (int64_t)t * BM + 4 * rg));
            // coefficients of this thread's four columns as two pairs
            f32x2 cmean[2], cinv[2], csc[2], csh[2], ck1[2], ck2[2], ck3[2];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(&cf[0 * NP + c0]), q1 = *reinterpret_cast<const float4*>(&cf[1 * NP + c0]);
                const float4 q2 = *reinterpret_cast<const float4*>(&cf[2 * NP + c0]), q3 = *reinterpret_cast<const float4*>(&cf[3 * NP + c0]);
                const float4 q4 = *reinterpret_cast<const float4*>(&cf[4 * NP + c0]), q5 = *reinterpret_cast<const float4*>(&cf[5 * NP + c0]);
                const float4 q6 = *reinterpret_cast<const float4*>(&cf[6 * NP + c0]);
                cmean[0] = f32x2{q0.x, q0.y}; cmean[1] = f32x2{q0.z, q0.w};
                cinv[0] = f32x2{q1.x, q1.y}; cinv[1] = f32x2{q1.z, q1.w};
                csc[0] = f32x2{q2.x, q2.y}; csc[1] = f32x2{q2.z, q2.w};
                csh[0] = f32x2{q3.x, q3.y}; csh[1] = f32x2{q3.z, q3.w};
                ck1[0] = f32x2{q4.x, q4.y}; ck1[1] = f32x2{q4.z, q4.w};
                ck2[0] = f32x2{q5.x, q5.y}; ck2[1] = f32x2{q5.z, q5.w};
                ck3[0] = f32x2{q6.x, q6.y}; ck3[1] = f32x2{q6.z, q6.w};
            }
            uint32_t hw[3][4][2];                   // [plane][row j][column pair h]: packed bf16 pairs of the micro-tile
            f32x2 ts[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};      // column sums of this tile's four rows (float), folded into cs below
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rok = j < left;
                f32x2 vrow[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 d = rz[j][h];
                    const f32x2 yv = ry[j][h];
                    if (relu6) {
                        const f32x2 z = __builtin_elementwise_fma(csc[h], yv, csh[h]);     // = fmaf(scale, y, shift) of the forward, per element
                        if (!relu6_open(z[0])) d[0] = 0.0f;
                        if (!relu6_open(z[1])) d[1] = 0.0f;
                    }
                    const f32x2 xh = (yv - cmean[h]) * cinv[h];
                    f32x2 v = ck1[h] * (d - ck2[h] - xh * ck3[h]);      // padded columns: every coefficient 0 -> 0
                    if (!rok) v = f32x2{0.0f, 0.0f};
                    vrow[h] = v;
                }
                if (more) load_row(t + 1, j);       // rz[j] / ry[j] are free
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ts[h] += vrow[h];
                    pwb_split3x2(vrow[h], hw[0][j][h], hw[1][j][h], hw[2][j][h]);
                }
                if (!(a.dbg & 2)) {                 // row-major planes: the conversion results as they are
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<u32x2*>(&Rm[p * BM * LDR + (4 * rg + j) * LDR + c0]) = u32x2{hw[p][j][0], hw[p][j][1]};
                }
            }
            cs[0] += (double)ts[0][0];
            cs[1] += (double)ts[0][1];
            cs[2] += (double)ts[1][0];
            cs[3] += (double)ts[1][1];
            if (a.dbg & 2) {
                if (hw[0][0][0] == 123u && hw[2][3][1] == 321u) Rm[0] = (__bf16)1.0f;      // keep the math alive
                return;
            }
            // transposed planes: column c0 + e, rows 4 rg .. 4 rg + 3 = one v_perm_b32 per row pair
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    *reinterpret_cast<u32x2*>(&Dt[p * TPD + (c0 + 2 * h) * LDT + 4 * rg]) =
                        u32x2{pwb_lolo(hw[p][0][h], hw[p][1][h]), pwb_lolo(hw[p][2][h], hw[p][3][h])};
                    *reinterpret_cast<u32x2*>(&Dt[p * TPD + (c0 + 2 * h + 1) * LDT + 4 * rg]) =
                        u32x2{pwb_hihi(hw[p][0][h], hw[p][1][h]), pwb_hihi(hw[p][2][h], hw[p][3][h])};
                }
        };
        // output tile through a buffer descriptor: the row offset of register r is wave-uniform (SGPR soffset), the lane's part
        // (its 4 lk rows + its column) one constant voffset -- no 64-bit address per register (the flat form kept 16 of them live
        // across the tile loop and spilled); lanes beyond K and rows beyond the group point out of range: store dropped, load 0
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.da.p, 0, (int)(Mtot * a.da.ld * 4), 0x00020000);
        const uint32_t rowC = (uint32_t)a.da.ld * 4u;
        const int ncol = dwc * 32 + lrow;
        const uint32_t voC = ncol < K ? (uint32_t)((4 * lk) * a.da.ld + a.da.coff + ncol) * 4u : OOR;
        auto compute_tile = [&](int t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int64_t m0 = mbeg + (int64_t)t * BM + dwr * 32;
            const uint32_t mu = (uint32_t)m0;
            const int left = (int)(mend - m0) - 4 * lk;         // rows (r & 3) + 8 (r >> 2) of this lane below `left` are inside the group
            float cold[ACC ? 16 : 1];
            if (ACC) {          // old values of the output tile, fetched before the MFMA chain (gemm_pw.hip ACC_PF)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    cold[ACC ? r : 0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, voC | (rr < left ? 0u : OOR), (mu + (uint32_t)rr) * rowC, 0));
                }
            }
            const int ao = (dwr * 32 + lrow) * LDR + 8 * lk;
            if (!(a.dbg & 1))
#pragma unroll
            for (int s = 0; s < KS_DA; ++s) {
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&Rm[0 * BM * LDR + ao + 16 * s]);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&Rm[1 * BM * LDR + ao + 16 * s]);
                const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&Rm[2 * BM * LDR + ao + 16 * s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, breg[0][s], acc, 0, 0, 0);       // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[2][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[1][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, breg[0][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[1][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, breg[0][s], acc, 0, 0, 0);
            }
            if (!(a.dbg & 4))
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                const float v = ACC ? acc[r] + cold[ACC ? r : 0] : acc[r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, voC | (rr < left ? 0u : OOR), (mu + (uint32_t)rr) * rowC, 0);
            }
        };
        if (t0 < t1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(t0, j);
        }
        __syncthreads();            // coefficient table
        for (int t = t0; t < t1; ++t) {
            store_tile(t);          // (requests tile t + 1)
            __syncthreads();
            if (wave < DA_WC * DA_WR) compute_tile(t);
            __syncthreads();
        }
        double* red = reinterpret_cast<double*>(smem_raw);      // [NRG][NP]; the planes are dead
#pragma unroll
        for (int e = 0; e < 4; ++e) red[rg * NP + c0 + e] = cs[e];
        __syncthreads();
        for (int c = tid; c < NP; c += 256) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NRG; ++r) s += red[r * NP + c];
            a.dbpart[((int64_t)g * a.nbpg + b) * NP + c] = s;
        }
    } else {
        // =========================================================== a (or xhat(a)) -> LDS; Q += a^T dy over all tiles
        const int qw = wave & 3;
        const int qkt = qw % Q_KT, qnt0 = (qw / Q_KT) * NTW;
        f32x2 cmean[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}}, cinv[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
        if (ANORM) {
            const int GK = a.G * K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                const bool on = c < K;
                cmean[e >> 1][e & 1] = on ? a.a_stats[0 * GK + g * K + (on ? c : 0)] : 0.0f;
                cinv[e >> 1][e & 1] = on ? a.a_stats[1 * GK + g * K + (on ? c : 0)] : 0.0f;
            }
        }
        const bool aon = c0 < KP;                   // (64 -> 128: half of this role's threads have no column group of a)
        uint32_t vo[2];
        vo[0] = c0 < K ? (uint32_t)((4 * rg) * a.a.ld + a.a.coff + c0) * 4u : OOR;
        vo[1] = c0 + 2 < K ? (uint32_t)((4 * rg) * a.a.ld + a.a.coff + c0 + 2) * 4u : OOR;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.a.p, 0, (int)(Mtot * a.a.ld * 4), 0x00020000);
        const uint32_t rowA = (uint32_t)a.a.ld * 4u;
        f32x2 rz[4][2];
        auto load_row = [&](int t, int j) {
            if (a.dbg & 8) return;
            const int64_t m0 = mbeg + (int64_t)t * BM;
            const uint32_t mu = (uint32_t)m0;
            const int left = (int)(mend - (m0 + 4 * rg));
            {
                const uint32_t msk = j < left ? 0u : OOR;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsA, (vo[h] + (uint32_t)j * rowA) | msk, mu * rowA, 0);
                    rz[j][h] = __builtin_bit_cast(f32x2, v);
                }
            }
        };
        auto store_tile = [&](int t) {
            const bool more = t + 1 < t1;
            const int left = (int)(mend - (mbeg + (int64_t)t * BM + 4 * rg));
            uint32_t hw[3][4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rok = j < left;
                f32x2 vrow[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 v = rz[j][h];
                    if (ANORM) {
                        v = (v - cmean[h]) * cinv[h];       // padded columns: loaded 0, mean 0, invstd 0 -> 0
                        if (!rok) v = f32x2{0.0f, 0.0f};
                    }
                    vrow[h] = v;
                }
                if (more) load_row(t + 1, j);
#pragma unroll
                for (int h = 0; h < 2; ++h) pwb_split3x2(vrow[h], hw[0][j][h], hw[1][j][h], hw[2][j][h]);
            }
            if (a.dbg & 2) {
                if (hw[0][0][0] == 123u && hw[2][3][1] == 321u) At[0] = (__bf16)1.0f;
                return;
            }
            if (aon) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        *reinterpret_cast<u32x2*>(&At[p * TPA + (c0 + 2 * h) * LDT + 4 * rg]) =
                            u32x2{pwb_lolo(hw[p][0][h], hw[p][1][h]), pwb_lolo(hw[p][2][h], hw[p][3][h])};
                        *reinterpret_cast<u32x2*>(&At[p * TPA + (c0 + 2 * h + 1) * LDT + 4 * rg]) =
                            u32x2{pwb_hihi(hw[p][0][h], hw[p][1][h]), pwb_hihi(hw[p][2][h], hw[p][3][h])};
                    }
            }
        };
        f32x16 qacc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) qacc[j][r] = 0.0f;
        auto compute_tile = [&]() {
            const int ao = (qkt * 32 + lrow) * LDT + 8 * lk;
            if (!(a.dbg & 1))
#pragma unroll
            for (int s = 0; s < KS_Q; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&At[0 * TPA + ao + 16 * s]);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&At[1 * TPA + ao + 16 * s]);
                const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&At[2 * TPA + ao + 16 * s]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int bo = ((qnt0 + j) * 32 + lrow) * LDT + 8 * lk + 16 * s;
                    const bf16x8 d1 = *reinterpret_cast<const bf16x8*>(&Dt[0 * TPD + bo]);
                    const bf16x8 d2 = *reinterpret_cast<const bf16x8*>(&Dt[1 * TPD + bo]);
                    const bf16x8 d3 = *reinterpret_cast<const bf16x8*>(&Dt[2 * TPD + bo]);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, d1, qacc[j], 0, 0, 0);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d3, qacc[j], 0, 0, 0);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, d2, qacc[j], 0, 0, 0);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, d1, qacc[j], 0, 0, 0);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d2, qacc[j], 0, 0, 0);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d1, qacc[j], 0, 0, 0);
                }
            }
        };
        if (t0 < t1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(t0, j);
        }
        __syncthreads();            // (the other role's coefficient table)
        for (int t = t0; t < t1; ++t) {
            store_tile(t);
            __syncthreads();
            compute_tile();
            __syncthreads();
        }
        float* qp = a.qpart + ((int64_t)g * a.nbpg + b) * KP * NP;
        if (!(a.dbg & 16))
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = (qnt0 + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = qkt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                qp[k * NP + n] = qacc[j][r];
            }
        }
        __syncthreads();            // pairs with the barrier in front of the other role's column-sum fold
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// bf16 ACTIVATION STORAGE (configuration 3, Config::compute == 2): the same skeleton on bf16 tensors.  dz, y, a are read and da is
// written as bf16 (paired 4-byte accesses: a thread's four columns are two bf16 pairs; 4-byte accesses at 2-byte alignment are
// fine on gfx950, 8-byte ones that straddle the end of a buffer are dropped as a whole, hence pairs), the BatchNorm-backward
// prologue runs in float32, both MFMA operands are ONE bf16 plane (dy and xhat / a rounded to nearest even, W^T = plane 0 of the
// packed fragments = bf16(W)), accumulation / partials / the reduce kernel in float32 / double as above.  A third of the LDS and a
// sixth of the matrix work of the float32 form: two workgroups per CU.
template <int KP, int NP, bool SHUF, bool ANORM, bool ACC>
__global__ void __launch_bounds__(512, 4) pwb16_kernel(PwbArgs a) {
    static_assert(KP == NP && (KP == 64 || KP == 128), "instantiated for square padded shapes");
    constexpr int BM = KP == 128 ? 32 : 64;
    constexpr int NRG = BM / 4;
    constexpr int LDR = NP + 8;
    constexpr int LDT = BM + 8;
    constexpr int TP = KP * LDT;
    constexpr int KS_DA = NP / 16, KS_Q = BM / 16;
    constexpr int DA_WC = KP / 32, DA_WR = 4 / DA_WC;
    constexpr int Q_KT = KP / 32, Q_NT = NP / 32;
    constexpr int NTW = Q_NT / (4 / Q_KT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* Rm = reinterpret_cast<__bf16*>(smem_raw);       // [BM][LDR]   dy, row-major
    __bf16* Dt = Rm + BM * LDR;                             // [NP][LDT]   dy, transposed
    __bf16* At = Dt + TP;                                   // [KP][LDT]   a (or xhat(a)), transposed
    float* cf = reinterpret_cast<float*>(At + TP);          // [7][NP]
    __bf16* Wl = reinterpret_cast<__bf16*>(cf + 7 * NP);    // [NP/16][2][128][8] W^T fragments (plane 0): LDS, not registers -- at two
                                                            // workgroups per CU a wave has 128 VGPRs, and 32 of them went to these
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int t2 = tid & 255;
    const int rg = t2 % NRG, cg = t2 / NRG;
    const int c0 = 4 * cg;
    const int g = blockIdx.x / a.nbpg, b = blockIdx.x % a.nbpg;
    const int K = a.K, N = a.N;
    const int64_t mbeg = (int64_t)g * a.Mg, mend = mbeg + a.Mg;
    const int tiles_g = (a.Mg + BM - 1) / BM;
    const int t0 = (int)((int64_t)b * tiles_g / a.nbpg), t1 = (int)((int64_t)(b + 1) * tiles_g / a.nbpg);
    const int64_t Mtot = (int64_t)a.G * a.Mg;
    const uint32_t OOR = 0x80000000u;
    auto widen = [](uint32_t w) -> f32x2 { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
    auto pack = [](f32x2 v) -> uint32_t { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); };

    if (role == 0) {
        const int dwr = wave % DA_WR, dwc = wave / DA_WR;
        for (int i = tid; i < KS_DA * 2 * 128; i += 256) {     // W^T fragments (plane 0 of the packed operand) -> LDS, once
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
            if (i < a.wp_ks * 2 * 128) z = *reinterpret_cast<const bf16x8*>(a.Wp + (int64_t)i * 8);      // (steps beyond the pack: zero)
            *reinterpret_cast<bf16x8*>(&Wl[i * 8]) = z;
        }
        const int GN = a.G * N;
        for (int i = tid; i < 7 * NP; i += 256) {
            const int q = i / NP, c = i % NP;
            cf[i] = c < N ? (q < 4 ? a.stats[q * GN + g * N + c] : a.coef[(q - 4) * GN + g * N + c]) : 0.0f;
        }
        // byte offsets of the two bf16 pairs of (row 4 rg) inside a tile.  Dense: pairs (c0, c0 + 1), (c0 + 2, c0 + 3).  Through the
        // shuffle gather destination columns c and c + 2 are ADJACENT in the source: pairs (c0, c0 + 2), (c0 + 1, c0 + 3).
        uint32_t vo[2], voy[2];
        bool cok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cok[e] = c0 + e < N;
        // (a pair whose second half would lie beyond the END OF THE ROW -- N = 58: destination column 57 is the row's last source
        //  element -- is fetched one element lower and taken from the high half: a 4-byte access that straddles the end of the buffer
        //  in the tensor's last row is dropped as a whole)
        bool hi_half[2] = {false, false};
        if (SHUF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int dc = shuffle_dst(a.dz.coff + c0 + h, a.dz_shuffle);
                if (dc + 1 >= a.dz.ld) {
                    dc -= 1;
                    hi_half[h] = true;
                }
                vo[h] = cok[h] ? (uint32_t)((4 * rg) * a.dz.ld + dc) * 2u : OOR;
            }
        } else {
            vo[0] = cok[0] ? (uint32_t)((4 * rg) * a.dz.ld + a.dz.coff + c0) * 2u : OOR;
            vo[1] = cok[2] ? (uint32_t)((4 * rg) * a.dz.ld + a.dz.coff + c0 + 2) * 2u : OOR;
        }
        voy[0] = cok[0] ? (uint32_t)((4 * rg) * N + c0) * 2u : OOR;
        voy[1] = cok[2] ? (uint32_t)((4 * rg) * N + c0 + 2) * 2u : OOR;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(a.dz.p, 0, (int)(Mtot * a.dz.ld * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y), 0, (int)(Mtot * N * 2), 0x00020000);
        const uint32_t rowD = (uint32_t)a.dz.ld * 2u, rowY = (uint32_t)N * 2u;
        const bool relu6 = a.act == ACT_RELU6;
        uint32_t rz[4][2], ry[4][2];                // raw words (widened where they are used, not where they are loaded)
        auto load_row = [&](int t, int j) {
            const int64_t m0 = mbeg + (int64_t)t * BM;
            const uint32_t mu = (uint32_t)m0;
            const int left = (int)(mend - (m0 + 4 * rg));
            const uint32_t msk = j < left ? 0u : OOR;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rz[j][h] = __builtin_amdgcn_raw_buffer_load_b32(rsD, (vo[h] + (uint32_t)j * rowD) | msk, mu * rowD, 0);
                ry[j][h] = __builtin_amdgcn_raw_buffer_load_b32(rsY, (voy[h] + (uint32_t)j * rowY) | msk, mu * rowY, 0);
            }
        };
        double cs[4] = {0.0, 0.0, 0.0, 0.0};
        auto store_tile = [&](int t) {
            const bool more = t + 1 < t1;
            const int left = (int)(mend - (mbeg + (int64_t)t * BM + 4 * rg));
            f32x2 cmean[2], cinv[2], csc[2], csh[2], ck1[2], ck2[2], ck3[2];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(&cf[0 * NP + c0]), q1 = *reinterpret_cast<const float4*>(&cf[1 * NP + c0]);
                const float4 q2 = *reinterpret_cast<const float4*>(&cf[2 * NP + c0]), q3 = *reinterpret_cast<const float4*>(&cf[3 * NP + c0]);
                const float4 q4 = *reinterpret_cast<const float4*>(&cf[4 * NP + c0]), q5 = *reinterpret_cast<const float4*>(&cf[5 * NP + c0]);
                const float4 q6 = *reinterpret_cast<const float4*>(&cf[6 * NP + c0]);
                cmean[0] = f32x2{q0.x, q0.y}; cmean[1] = f32x2{q0.z, q0.w};
                cinv[0] = f32x2{q1.x, q1.y}; cinv[1] = f32x2{q1.z, q1.w};
                csc[0] = f32x2{q2.x, q2.y}; csc[1] = f32x2{q2.z, q2.w};
                csh[0] = f32x2{q3.x, q3.y}; csh[1] = f32x2{q3.z, q3.w};
                ck1[0] = f32x2{q4.x, q4.y}; ck1[1] = f32x2{q4.z, q4.w};
                ck2[0] = f32x2{q5.x, q5.y}; ck2[1] = f32x2{q5.z, q5.w};
                ck3[0] = f32x2{q6.x, q6.y}; ck3[1] = f32x2{q6.z, q6.w};
            }
            uint32_t hw[4][2];                      // [row j][column pair h]
            f32x2 ts[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rok = j < left;
                f32x2 dz2[2];
                if (SHUF) {     // loaded (c0, c0 + 2), (c0 + 1, c0 + 3) -> pairs (c0, c0 + 1), (c0 + 2, c0 + 3)
                    const f32x2 pa = widen(hi_half[0] ? rz[j][0] >> 16 : rz[j][0]), pb = widen(hi_half[1] ? rz[j][1] >> 16 : rz[j][1]);
                    dz2[0] = f32x2{pa[0], pb[0]};
                    dz2[1] = f32x2{pa[1], pb[1]};
                    if (!cok[2]) dz2[1][0] = 0.0f;      // (channel counts are even: c0 + 2 and c0 + 3 are valid together; the source
                    if (!cok[3]) dz2[1][1] = 0.0f;      //  words' second halves belong to other channels there)
                } else {
                    dz2[0] = widen(rz[j][0]);
                    dz2[1] = widen(rz[j][1]);
                }
                f32x2 vrow[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 d = dz2[h];
                    const f32x2 yv = widen(ry[j][h]);
                    if (relu6) {
                        const f32x2 z = __builtin_elementwise_fma(csc[h], yv, csh[h]);
                        if (!relu6_open(z[0])) d[0] = 0.0f;
                        if (!relu6_open(z[1])) d[1] = 0.0f;
                    }
                    const f32x2 xh = (yv - cmean[h]) * cinv[h];
                    f32x2 v = ck1[h] * (d - ck2[h] - xh * ck3[h]);
                    if (!rok) v = f32x2{0.0f, 0.0f};
                    vrow[h] = v;
                }
                if (more) load_row(t + 1, j);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ts[h] += vrow[h];
                    hw[j][h] = pack(vrow[h]);
                }
                *reinterpret_cast<u32x2*>(&Rm[(4 * rg + j) * LDR + c0]) = u32x2{hw[j][0], hw[j][1]};
            }
            cs[0] += (double)ts[0][0];
            cs[1] += (double)ts[0][1];
            cs[2] += (double)ts[1][0];
            cs[3] += (double)ts[1][1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<u32x2*>(&Dt[(c0 + 2 * h) * LDT + 4 * rg]) = u32x2{pwb_lolo(hw[0][h], hw[1][h]), pwb_lolo(hw[2][h], hw[3][h])};
                *reinterpret_cast<u32x2*>(&Dt[(c0 + 2 * h + 1) * LDT + 4 * rg]) = u32x2{pwb_hihi(hw[0][h], hw[1][h]), pwb_hihi(hw[2][h], hw[3][h])};
            }
        };
        // output tile as bf16: a lane owns ONE column (16 rows in 16 registers); lanes (2i, 2i + 1) own adjacent columns -- the even
        // lane takes the even registers of both columns, the odd lane the odd ones, the partner's half arrives by DPP: 8 four-byte
        // stores per lane instead of 16 two-byte ones (gemm_pw.hip pair_load16).  Row offsets of register pair q are wave-uniform.
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.da.p, 0, (int)(Mtot * a.da.ld * 2), 0x00020000);
        const uint32_t rowC = (uint32_t)a.da.ld * 2u;
        const int ncol = dwc * 32 + lrow;
        const bool odd = (lane & 1) != 0;
        const uint32_t voC = (ncol & ~1) < K ? (uint32_t)((4 * lk + (odd ? 1 : 0)) * a.da.ld + a.da.coff + (ncol & ~1)) * 2u : OOR;
        auto swap1 = [](float x) -> float { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)); };
        double sda = 0.0, sdx = 0.0;                // ANORM: this lane's column, its 16 rows of every tile
        auto compute_tile = [&](int t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int64_t m0 = mbeg + (int64_t)t * BM + dwr * 32;
            const uint32_t mu = (uint32_t)m0;
            const int left = (int)(mend - m0) - 4 * lk - (odd ? 1 : 0);
            uint32_t cold[ACC ? 8 : 1];
            if (ACC) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int rr = ((2 * q) & 3) + 8 * ((2 * q) >> 2);
                    cold[ACC ? q : 0] = __builtin_amdgcn_raw_buffer_load_b32(rsC, voC | (rr < left ? 0u : OOR), (mu + (uint32_t)rr) * rowC, 0);
                }
            }
            const int ao = (dwr * 32 + lrow) * LDR + 8 * lk;
#pragma unroll
            for (int s = 0; s < KS_DA; ++s) {
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&Rm[ao + 16 * s]);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Wl[((s * 2 + lk) * 128 + ncol) * 8]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            }
            if (ANORM) {
                // BatchNorm-backward sums of the BatchNorm in front of the conv, from the float32 accumulators and the xhat operand
                // plane, in double: deriving them from the float32 filter-product partials (as the float32 kernel does) makes the
                // coefficients order-dependent in their last bits, and behind bf16 storage a last-bit change of a coefficient flips
                // roundings downstream (test_config3_full_size_properties: permutation invariance)
                // (every term in double: float32 partial sums over the 16 rows of a tile already break the invariance, measured)
                const int xo = ncol * LDT + dwr * 32 + 4 * lk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2 xw = *reinterpret_cast<const u32x2*>(&At[xo + 8 * i]);
                    const f32x2 x01 = widen(xw[0]), x23 = widen(xw[1]);
                    sda += ((double)acc[4 * i] + (double)acc[4 * i + 1]) + ((double)acc[4 * i + 2] + (double)acc[4 * i + 3]);
                    sdx += ((double)acc[4 * i] * (double)x01[0] + (double)acc[4 * i + 1] * (double)x01[1]) +
                           ((double)acc[4 * i + 2] * (double)x23[0] + (double)acc[4 * i + 3] * (double)x23[1]);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // even lane: rows of registers 2q (its own column -> lo, partner's -> hi); odd lane: registers 2q + 1
                const float recv = swap1(odd ? acc[2 * q] : acc[2 * q + 1]);
                float lo = odd ? recv : acc[2 * q], hi = odd ? acc[2 * q + 1] : recv;
                if (ACC) {
                    lo += bf_lo(cold[ACC ? q : 0]);
                    hi += bf_hi(cold[ACC ? q : 0]);
                }
                const int rr = ((2 * q) & 3) + 8 * ((2 * q) >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(bf_pack(lo, hi), rsC, voC | (rr < left ? 0u : OOR), (mu + (uint32_t)rr) * rowC, 0);
            }
        };
        if (t0 < t1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(t0, j);
        }
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            store_tile(t);
            __syncthreads();
            compute_tile(t);
            __syncthreads();
        }
        double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
        for (int e = 0; e < 4; ++e) red[rg * NP + c0 + e] = cs[e];
        __syncthreads();
        for (int c = tid; c < NP; c += 256) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NRG; ++r) s += red[r * NP + c];
            a.dbpart[((int64_t)g * a.nbpg + b) * NP + c] = s;
        }
        if (ANORM && a.spart) {
            __syncthreads();                        // (role 1 passes the matching barriers below)
            double* r2 = reinterpret_cast<double*>(smem_raw);       // [DA_WR][2][2][KP]: wave row, lk half, quantity, column
            r2[((dwr * 2 + lk) * 2 + 0) * KP + ncol] = sda;
            r2[((dwr * 2 + lk) * 2 + 1) * KP + ncol] = sdx;
            __syncthreads();
            for (int i = tid; i < 2 * KP; i += 256) {
                const int q = i / KP, c = i % KP;
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < DA_WR * 2; ++w) s += r2[(w * 2 + q) * KP + c];
                a.spart[(((int64_t)g * a.nbpg + b) * 2 + q) * KP + c] = s;
            }
        }
    } else {
        const int qw = wave & 3;
        const int qkt = qw % Q_KT, qnt0 = (qw / Q_KT) * NTW;
        f32x2 cmean[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}}, cinv[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
        if (ANORM) {
            const int GK = a.G * K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                const bool on = c < K;
                cmean[e >> 1][e & 1] = on ? a.a_stats[0 * GK + g * K + (on ? c : 0)] : 0.0f;
                cinv[e >> 1][e & 1] = on ? a.a_stats[1 * GK + g * K + (on ? c : 0)] : 0.0f;
            }
        }
        uint32_t vo[2];
        vo[0] = c0 < K ? (uint32_t)((4 * rg) * a.a.ld + a.a.coff + c0) * 2u : OOR;
        vo[1] = c0 + 2 < K ? (uint32_t)((4 * rg) * a.a.ld + a.a.coff + c0 + 2) * 2u : OOR;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(a.a.p, 0, (int)(Mtot * a.a.ld * 2), 0x00020000);
        const uint32_t rowA = (uint32_t)a.a.ld * 2u;
        uint32_t rz[4][2];
        auto load_row = [&](int t, int j) {
            const int64_t m0 = mbeg + (int64_t)t * BM;
            const uint32_t mu = (uint32_t)m0;
            const int left = (int)(mend - (m0 + 4 * rg));
            const uint32_t msk = j < left ? 0u : OOR;
#pragma unroll
            for (int h = 0; h < 2; ++h) rz[j][h] = __builtin_amdgcn_raw_buffer_load_b32(rsA, (vo[h] + (uint32_t)j * rowA) | msk, mu * rowA, 0);
        };
        auto store_tile = [&](int t) {
            const bool more = t + 1 < t1;
            const int left = (int)(mend - (mbeg + (int64_t)t * BM + 4 * rg));
            uint32_t hw[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rok = j < left;
                if (ANORM) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 v = (widen(rz[j][h]) - cmean[h]) * cinv[h];       // padded columns: loaded 0, mean 0, invstd 0 -> 0
                        if (!rok) v = f32x2{0.0f, 0.0f};
                        hw[j][h] = pack(v);
                    }
                } else {
                    hw[j][0] = rz[j][0];            // plain input: the stored bf16 values ARE the operand
                    hw[j][1] = rz[j][1];
                }
                if (more) load_row(t + 1, j);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<u32x2*>(&At[(c0 + 2 * h) * LDT + 4 * rg]) = u32x2{pwb_lolo(hw[0][h], hw[1][h]), pwb_lolo(hw[2][h], hw[3][h])};
                *reinterpret_cast<u32x2*>(&At[(c0 + 2 * h + 1) * LDT + 4 * rg]) = u32x2{pwb_hihi(hw[0][h], hw[1][h]), pwb_hihi(hw[2][h], hw[3][h])};
            }
        };
        f32x16 qacc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) qacc[j][r] = 0.0f;
        auto compute_tile = [&]() {
            const int ao = (qkt * 32 + lrow) * LDT + 8 * lk;
#pragma unroll
            for (int s = 0; s < KS_Q; ++s) {
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&At[ao + 16 * s]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const bf16x8 d1 = *reinterpret_cast<const bf16x8*>(&Dt[((qnt0 + j) * 32 + lrow) * LDT + 8 * lk + 16 * s]);
                    qacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d1, qacc[j], 0, 0, 0);
                }
            }
        };
        if (t0 < t1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(t0, j);
        }
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            store_tile(t);
            __syncthreads();
            compute_tile();
            __syncthreads();
        }
        float* qp = a.qpart + ((int64_t)g * a.nbpg + b) * KP * NP;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = (qnt0 + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = qkt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                qp[k * NP + n] = qacc[j][r];
            }
        }
        __syncthreads();
        if (ANORM && a.spart) {
            __syncthreads();
            __syncthreads();
        }
    }
}

// One workgroup per input channel k (row of dW), one thread per output channel n.
struct PwbReduceArgs {
    const float* qpart;     // [G][nbpg][KP][NP]
    const double* dbpart;   // [G][nbpg][NP]
    const double* spart;    // [G][nbpg][2][KP] or null: BatchNorm-backward sums taken directly by the main kernel (bf16 storage)
    const float* W;         // conv weights [K][N]
    const float* a_stats;   // [4][G][K] or null
    const float* a_gamma;   // ANORM: gamma / beta of the BatchNorm that produced a
    const float* a_beta;
    float* dW;              // [K][N]
    float* db;              // [N]
    float* a_dgamma;        // ANORM outputs: dgamma / dbeta [K], backward coefficients [3][G][K]
    float* a_dbeta;
    float* a_coef;
    const double* fin_tot;  // [G][2][N] (or null): group totals of the BatchNorm BEHIND the conv -> its dgamma / dbeta
    float* o_dgamma;
    float* o_dbeta;
    int N, K, KP, NP, G, Mg, nbpg;
    int wbf;                // bf16 storage: the backward-data product used bf16(W); the derived BatchNorm sums use the same values
};

// 1024 threads = 128 output channels n x 8 slots; a slot sums one (group, slice of the partial rows) pair with the loads of 8
// partials in flight (the first version walked all partials in one dependent loop per thread: 125 us; 4 slices x all groups: 10-13 us),
// the slices are folded through LDS in fixed order.
constexpr int PWB_RS = 8;
__global__ void __launch_bounds__(128 * PWB_RS) pwb_reduce_kernel(PwbReduceArgs a) {
    __shared__ double sq[PWB_RS][128];        // [slice * G + group][n]: Q partial sums (G * nsl <= 8 pairs)
    __shared__ double sd[PWB_RS][128];        // db partial sums
    __shared__ double red[2][8][2];           // [s1 | s2][group][wave]
    const int k = blockIdx.x, n = threadIdx.x & 127, sl = threadIdx.x >> 7;
    const int N = a.N, K = a.K, G = a.G;
    const bool on = n < N;
    const int nsl = PWB_RS / G > 0 ? PWB_RS / G : 1;          // slices per group (G = 4: 2)
    const int per = (a.nbpg + nsl - 1) / nsl;
    for (int pair = sl; pair < G * nsl; pair += PWB_RS) {
        const int g = pair % G, s = pair / G;
        const int b0 = s * per, b1 = min(a.nbpg, b0 + per);
        double q = 0.0, d = 0.0;
        if (on) {
            const float* pq = a.qpart + ((int64_t)g * a.nbpg * a.KP + k) * a.NP + n;
            const double* pd = a.dbpart + (int64_t)g * a.nbpg * a.NP + n;
            int b = b0;
            // the column sums of dy are needed for every k only when the BatchNorm sums of the conv input are derived here (ANORM); the
            // plain form writes db from k == 0 alone -- 32 of its 64 loads per thread were for values nobody reads
            const bool need_d = a.a_stats != nullptr || k == 0;
            for (; b + 32 <= b1; b += 32) {         // a slot's 32 partials in ONE round of loads (two rounds of 16 before round 5)
                float v[32];
                double u[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    v[i] = pq[(int64_t)(b + i) * a.KP * a.NP];
                    u[i] = need_d ? pd[(int64_t)(b + i) * a.NP] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    q += (double)v[i];
                    d += u[i];
                }
            }
            for (; b + 16 <= b1; b += 16) {
                float v[16];
                double u[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    v[i] = pq[(int64_t)(b + i) * a.KP * a.NP];
                    u[i] = pd[(int64_t)(b + i) * a.NP];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    q += (double)v[i];
                    d += u[i];
                }
            }
            for (; b + 8 <= b1; b += 8) {
                float v[8];
                double u[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = pq[(int64_t)(b + i) * a.KP * a.NP];
                    u[i] = pd[(int64_t)(b + i) * a.NP];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    q += (double)v[i];
                    d += u[i];
                }
            }
            for (; b < b1; ++b) {
                q += (double)pq[(int64_t)b * a.KP * a.NP];
                d += pd[(int64_t)b * a.NP];
            }
        }
        sq[pair][n] = q;
        sd[pair][n] = d;
    }
    __syncthreads();
    const bool lead = sl == 0;                  // slice 0 (threads 0 .. 127 = waves 0, 1) folds and writes
    float w = (on && lead) ? a.W[(int64_t)k * N + n] : 0.0f;
    if (a.wbf) w = (float)(__bf16)w;
    double qtot = 0.0, dbtot = 0.0;
    double s1[8], s2[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        s1[g] = s2[g] = 0.0;
        if (g < G && lead) {
            double q = sq[g][n], d = sd[g][n];
            for (int s = 1; s < nsl; ++s) {
                q += sq[s * G + g][n];
                d += sd[s * G + g][n];
            }
            qtot += q;
            dbtot += d;
            s1[g] = (double)w * d;
            s2[g] = (double)w * q;
        }
    }
    if (on && lead) {
        if (a.a_stats) a.dW[(int64_t)k * N + n] = (float)((double)a.a_gamma[k] * qtot + (double)a.a_beta[k] * dbtot);
        else a.dW[(int64_t)k * N + n] = (float)qtot;
        if (k == 0) a.db[n] = (float)dbtot;
        if (k == 0 && a.fin_tot) {      // shared gamma / beta: summed over the T applications, t ascending (as bn_bwd_finalize)
            double dg = 0.0, dbt = 0.0;
            for (int g = 0; g < G; ++g) {
                dbt += a.fin_tot[((int64_t)g * 2 + 0) * N + n];
                dg += a.fin_tot[((int64_t)g * 2 + 1) * N + n];
            }
            a.o_dgamma[n] = (float)dg;
            a.o_dbeta[n] = (float)dbt;
        }
    }
    if (!a.a_stats) return;
    // BatchNorm-backward sums of the BatchNorm that produced a: fixed-order reduction over n
    const int lane = n & 63, wave = n >> 6;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g >= G) break;
        double u1 = s1[g], u2 = s2[g];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            u1 += __shfl_down(u1, o);
            u2 += __shfl_down(u2, o);
        }
        if (lane == 0 && lead) {
            red[0][g][wave] = u1;
            red[1][g][wave] = u2;
        }
    }
    __syncthreads();
    if (a.spart && (int)(threadIdx.x >> 6) < 2 * G) {   // the directly accumulated sums replace the derived ones: one wave per (group, sum)
        // pair, lane l takes the workgroups l, l + 64, ... and the lanes fold in a fixed tree (one thread walking all 128 partials was a
        // chain of dependent loads: 53 us per launch at B = 1024 against 14 us for the rest of the kernel)
        const int pr = threadIdx.x >> 6, g = pr >> 1, q = pr & 1, l = threadIdx.x & 63;
        double u = 0.0;
        for (int b = l; b < a.nbpg; b += 64) u += a.spart[(((int64_t)g * a.nbpg + b) * 2 + q) * a.KP + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) u += __shfl_down(u, o);
        if (l == 0) {
            red[q][g][0] = u;
            red[q][g][1] = 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int GK = G * K;
        const double cnt = (double)a.Mg;
        double dg = 0.0, dbt = 0.0;
        for (int g = 0; g < G; ++g) {
            const double u1 = red[0][g][0] + red[0][g][1], u2 = red[1][g][0] + red[1][g][1];
            dbt += u1;
            dg += u2;
            a.a_coef[0 * GK + g * K + k] = a.a_stats[2 * GK + g * K + k];       // k1 = gamma * invstd
            a.a_coef[1 * GK + g * K + k] = (float)(u1 / cnt);                   // k2 = mean(dz)
            a.a_coef[2 * GK + g * K + k] = (float)(u2 / cnt);                   // k3 = mean(dz * xhat)
        }
        a.a_dgamma[k] = (float)dg;
        a.a_dbeta[k] = (float)dbt;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int pwb_pad(int c) { return c <= 64 ? 64 : 128; }

bool pw_bwd_fused_supported(View dz, View a, View da, int N, int K, int at) {
    if (N > 128 || K > 128 || N < 8 || K < 8 || (N & 1) || (K & 1)) return false;
    if (at && pwb_pad(N) != pwb_pad(K)) return false;          // bf16-storage form: equal paddings only
    if (pwb_pad(N) != pwb_pad(K) && !(pwb_pad(K) == 64 && pwb_pad(N) == 128)) return false;      // 64/64, 128/128, 64 -> 128
    auto ok = [](View v) { return (v.ld % 2 == 0) && (v.coff % 2 == 0) && ((reinterpret_cast<uintptr_t>(v.p) & 7) == 0); };
    return ok(a) && ok(da) && (dz.ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(dz.p) & 7) == 0);
}

int pw_bwd_fused_nbpg(int G, int Mg, int N, int K, int at) {
    const int bm = (pwb_pad(K) == 128 || pwb_pad(N) == 128) ? 32 : 64;
    const int tiles = cdiv(Mg, bm);
    // resident workgroups: one per CU (float32: 87 KB of LDS); bf16 storage (32 KB) fits two, taken while a workgroup still gets >= 6 tiles
    int nb = 256 / G;
    if (at && tiles >= 6 * (512 / G)) nb = 512 / G;
    if (nb < 1) nb = 1;
    return nb > tiles ? tiles : nb;
}

int64_t pw_bwd_fused_qpart_elems(int G, int Mg, int N, int K, int at) {
    return (int64_t)G * pw_bwd_fused_nbpg(G, Mg, N, K, at) * pwb_pad(K) * pwb_pad(N);
}

// (bf16 storage: [column sums of dy | (sum da, sum da xhat)] -- three rows of NP == KP doubles per workgroup)
int64_t pw_bwd_fused_dbpart_elems(int G, int Mg, int N, int K, int at) {
    return (int64_t)G * pw_bwd_fused_nbpg(G, Mg, N, K, at) * pwb_pad(N) * (at ? 3 : 1);
}

template <int KP, int NP, bool SHUF, bool ANORM, bool ACC>
static int pwb_launch(const PwbArgs& a, hipStream_t st) {
    constexpr int BM = (KP == 128 || NP == 128) ? 32 : 64;
    constexpr size_t lds = (size_t)(3 * BM * (NP + 8) + 3 * (NP + KP) * (BM + 8)) * 2 + (size_t)7 * NP * sizeof(float);
    auto kern = pwb_kernel<KP, NP, SHUF, ANORM, ACC>;
    static LdsAttrOnce attr;
    if (attr.need()) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    hipLaunchKernelGGL(kern, dim3(a.G * a.nbpg), dim3(512), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int P, bool SHUF, bool ANORM, bool ACC>
static int pwb16_launch(const PwbArgs& a, hipStream_t st) {
    constexpr int BM = P == 128 ? 32 : 64;
    constexpr size_t lds = (size_t)(BM * (P + 8) + 2 * P * (BM + 8)) * 2 + (size_t)7 * P * sizeof(float) + (size_t)(P / 16) * 2 * 128 * 8 * 2;
    auto kern = pwb16_kernel<P, P, SHUF, ANORM, ACC>;
    static LdsAttrOnce attr;
    if (attr.need() && lds >= 64 * 1024) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.mark();
    }
    hipLaunchKernelGGL(kern, dim3(a.G * a.nbpg), dim3(512), lds, st, a);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int P, bool SHUF>
static int pwb16_launch2(const PwbArgs& a, bool anorm, bool acc, hipStream_t st) {
    if (anorm) return acc ? pwb16_launch<P, SHUF, true, true>(a, st) : pwb16_launch<P, SHUF, true, false>(a, st);
    return acc ? pwb16_launch<P, SHUF, false, true>(a, st) : pwb16_launch<P, SHUF, false, false>(a, st);
}

template <int KP, int NP, bool SHUF>
static int pwb_launch2(const PwbArgs& a, bool anorm, bool acc, hipStream_t st) {
    if (anorm) return acc ? pwb_launch<KP, NP, SHUF, true, true>(a, st) : pwb_launch<KP, NP, SHUF, true, false>(a, st);
    return acc ? pwb_launch<KP, NP, SHUF, false, true>(a, st) : pwb_launch<KP, NP, SHUF, false, false>(a, st);
}

int pw_bwd_fused(const PwBwdFused& f, hipStream_t st) {
    if (!pw_bwd_fused_supported(f.dz, f.a, f.da, f.N, f.K, f.at) || !f.Wp) {
        set_error("pw_bwd_fused: unsupported shape / alignment N=%d K=%d", f.N, f.K);
        return -1;
    }
    const int64_t Mtot = (int64_t)f.G * f.Mg;
    if (f.at && ((f.dz.coff | f.dz.ld | f.a.coff | f.a.ld | f.da.coff | f.da.ld) & 1)) {
        set_error("pw_bwd_fused: bf16 storage needs even leading dimensions / channel offsets");
        return -1;
    }
    if (Mtot * f.dz.ld * 4 >= ((int64_t)1 << 31) || Mtot * f.a.ld * 4 >= ((int64_t)1 << 31) || Mtot * f.N * 4 >= ((int64_t)1 << 31) ||
        Mtot * f.da.ld * 4 >= ((int64_t)1 << 31)) {
        set_error("pw_bwd_fused: operand of 2 GB or more");
        return -1;
    }
    if (f.G > 8) {
        set_error("pw_bwd_fused: more than 8 groups");
        return -1;
    }
    PwbArgs a;
    a.dz = f.dz;
    a.dz_shuffle = f.dz_shuffle;
    a.act = f.act;
    a.y = f.y;
    a.stats = f.stats;
    a.coef = f.coef;
    a.a = f.a;
    a.a_stats = f.a_stats;
    a.Wp = reinterpret_cast<const __bf16*>(f.Wp);
    a.wp_ks = pw_x3_ksteps(f.N);
    a.da = f.da;
    a.qpart = f.qpart;
    a.dbpart = f.dbpart;
    a.N = f.N;
    a.K = f.K;
    a.G = f.G;
    a.Mg = f.Mg;
    a.nbpg = pw_bwd_fused_nbpg(f.G, f.Mg, f.N, f.K, f.at);
    a.at = f.at;
    a.fin_part = f.at ? nullptr : f.fin_part;
    a.fin_tot = f.fin_tot;
    a.fin_nb = f.fin_nb;
    if (f.fin_part && (f.at || !f.fin_tot || f.fin_nb <= 0)) {
        set_error("pw_bwd_fused: finalize-on-load needs the float32 form, fin_tot and fin_nb");
        return -1;
    }
    a.spart = (f.at && f.a_stats) ? f.dbpart + (int64_t)f.G * a.nbpg * pwb_pad(f.N) : nullptr;
    static const int dbg = cdrl_getenv("CDRL_DIAG_PWB") ? atoi(cdrl_getenv("CDRL_DIAG_PWB")) : 0;
    a.dbg = dbg;
    const bool anorm = f.a_stats != nullptr, acc = f.accumulate != 0, shuf = f.dz_shuffle != 0;
    if (f.at && pwb_pad(f.K) != pwb_pad(f.N)) {
        set_error("pw_bwd_fused: the bf16-storage form is instantiated for equal paddings only (N=%d K=%d)", f.N, f.K);
        return -1;
    }
    if (f.at) {
        if (pwb_pad(f.K) == 128) return shuf ? pwb16_launch2<128, true>(a, anorm, acc, st) : pwb16_launch2<128, false>(a, anorm, acc, st);
        return shuf ? pwb16_launch2<64, true>(a, anorm, acc, st) : pwb16_launch2<64, false>(a, anorm, acc, st);
    }
    if (pwb_pad(f.K) == 128) return shuf ? pwb_launch2<128, 128, true>(a, anorm, acc, st) : pwb_launch2<128, 128, false>(a, anorm, acc, st);
    if (pwb_pad(f.N) == 128) return shuf ? pwb_launch2<64, 128, true>(a, anorm, acc, st) : pwb_launch2<64, 128, false>(a, anorm, acc, st);
    return shuf ? pwb_launch2<64, 64, true>(a, anorm, acc, st) : pwb_launch2<64, 64, false>(a, anorm, acc, st);
}

int pw_bwd_fused_reduce(const PwBwdFused& f, hipStream_t st) {
    PwbReduceArgs r;
    r.qpart = f.qpart;
    r.dbpart = f.dbpart;
    r.W = f.W;
    r.a_stats = f.a_stats;
    r.a_gamma = f.a_gamma;
    r.a_beta = f.a_beta;
    r.dW = f.dW;
    r.db = f.db;
    r.a_dgamma = f.a_dgamma;
    r.a_dbeta = f.a_dbeta;
    r.a_coef = f.a_coef;
    r.N = f.N;
    r.K = f.K;
    r.KP = pwb_pad(f.K);
    r.NP = pwb_pad(f.N);
    r.G = f.G;
    r.Mg = f.Mg;
    r.nbpg = pw_bwd_fused_nbpg(f.G, f.Mg, f.N, f.K, f.at);
    r.wbf = f.at;
    r.fin_tot = (f.fin_part && !f.at) ? f.fin_tot : nullptr;
    r.o_dgamma = f.o_dgamma;
    r.o_dbeta = f.o_dbeta;
    if (r.fin_tot && (!f.o_dgamma || !f.o_dbeta)) {
        set_error("pw_bwd_fused_reduce: finalize-on-load needs the dgamma / dbeta outputs of the BatchNorm behind the conv");
        return -1;
    }
    r.spart = (f.at && f.a_stats) ? f.dbpart + (int64_t)f.G * r.nbpg * pwb_pad(f.N) : nullptr;
    if (f.a_stats && (!f.a_gamma || !f.a_beta || !f.a_dgamma || !f.a_dbeta || !f.a_coef)) {
        set_error("pw_bwd_fused_reduce: normalised input needs gamma / beta and the dgamma / dbeta / coef outputs");
        return -1;
    }
    hipLaunchKernelGGL(pwb_reduce_kernel, dim3(f.K), dim3(128 * PWB_RS), 0, st, r);
    CDRL_LAUNCH_CHECK();
    return 0;
}

}  // namespace cdrl
