// Fused depthwise-3x3 block of the ShuffleNet unit (gfx950): one workgroup owns whole frames.
//
// Reference: core/architectures.py:120-145 -- the unit's  pw -> BN+ReLU6 -> dw3x3 -> BN -> pw  chain.
// A depthwise conv is local to one frame and one channel, and at the tower's resolutions a frame
// (22x30 ... 3x4 pixels x <= 232 channels) fits in LDS.  So instead of streaming every tensor through
// HBM once per elementary op, a workgroup stages the frame(s) in LDS once and does everything that
// is local to the frame on it:
//
//   forward   tile <- relu6(scale1 * y1 + shift1)      (BN1 apply: a1 is never written to HBM)
//             y2   <- dw3x3(tile) + bias               (9 LDS reads per output)
//             (sum y2, sum y2^2) per channel           -> BN2 statistics partials (no second pass over y2)
//
//   backward  D    <- k1 * (da2 - k2 - xhat2 * k3)     (BN2 backward apply: dy2 is never written to HBM)
//             A    <- y1 (raw; relu6(scale1 * y1 + shift1) is re-applied on each LDS read, nothing is stored)
//             dW   += A (x) D per tap, db += D         -> filter / bias gradient partials
//             dz1  <- mask1 * dw3x3^T(D)               (ReLU6 mask of BN1's output)
//             (sum dz1, sum dz1 * xhat1)               -> BN1 backward partials
//
// HBM traffic per element: forward 1 read + 1 write (was 3 reads + 2 writes over apply / dw / stats),
// backward 3 reads + 1 write (was 9 reads + 2 writes over bn-apply /
// dw-data / dw-filter / bn-reduce).  Launches per unit: forward 3 -> 1, backward 4 -> 1.
//
// Grid = (groups * frame-blocks, channel chunks); block = (channel lanes, pixel lanes); channel chunks
// only when a full-width frame does not fit the LDS budget (the 22x30 stride-2 inputs).  Partial sums
// are double, written per block, combined in fixed order by the finalize kernels (deterministic).
// The `pre` prologue is optional (shortcut branch: the depthwise reads an activation tensor directly).
#include <stdlib.h>

#include <algorithm>

#include "colreduce.h"

namespace cdrl {

// 4 workgroups per CU (160 KB LDS): frames wider than ~60 channels at 6x8 pixels run as two channel chunks (adjacent on one XCD, see
// the block map in the kernels).  Re-measured in round 3 (ms / update-step at 38 | 52 | 76 KB): float32 B = 256 15.91 | 15.94 | 16.07,
// float32 B = 1024 49.0 | - | 49.7, bf16 storage B = 1024 41.1 | 41.7 | 43.6 -- the phases of these kernels are separated by
// barriers, and two resident workgroups (8 waves per CU) do not cover them.
static constexpr size_t DWF_LDS_BUDGET = 38 * 1024;
#define DWF_T_FWD 512                                    // threads per workgroup, forward / backward
#define DWF_T_BWD 512
#define DWF_T_FWD_DEFAULT 256                            // (tunable: CDRL_DWF_TF / CDRL_DWF_TB, <= the maxima above; 256 vs 512 forward: -0.1 ms/update-step at v39)
#define DWF_T_BWD_DEFAULT 256
#define DWF_UF 8                                         // forward: loads in flight per thread (one tensor)
#define DWF_U 4                                          // global loads in flight per thread in the tile loads

static DwsGeom dws_geom(int B, int G, int fpb, int H, int W, int C);
static DwsGeom dws2_geom(int B, int G, int fpb, int H, int W, int C);

DwfGeom dwf_geom(int B, int G, int H, int W, int C, int stride) {
    DwfGeom g;
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    g.vec = (C % 4 == 0) ? 4 : ((C % 2 == 0) ? 2 : 1);
    // tiles are zero-padded by one pixel per side
    const size_t a_px = (size_t)(H + 2) * (W + 2);
    const size_t d_px = (size_t)(Ho + 2) * (Wo + 2);
    const size_t per_c = (a_px + d_px) * sizeof(float);
    // (a separate budget for the stride-2 blocks, CDRL_DWF_LDS_KB_S2: 38 | 52 | 76 | 110 KB -> 15.76 | 15.74 | 15.82 | 16.07 ms / update-step
    //  at float32 B = 256 and 40.6 | - | 41.9 ms at bf16-storage B = 1024: their backward kernel alone is faster with the larger tile
    //  (105 vs 119 us), the step is not)
    static const size_t lds_budget_s1 = DWF_LDS_BUDGET;
    static const size_t lds_budget_s2 = lds_budget_s1;
    size_t lds_budget = stride == 2 ? lds_budget_s2 : lds_budget_s1;
    // wide frames (three-camera 90x360, 135x180): at 38 KB their channel chunks drop below ~24 channels -- a dozen channel lanes
    // per workgroup and 5-6 chunks per frame (40.9 vs 38.3 ms / update-step at 90x360 with 76 KB) -- so the budget grows in steps
    // until a chunk holds at least CDRL_DWF_MINCHUNK channels (or the whole frame)
    // (22: the 11x15x58 units of the 90x120 configuration sit exactly there and were measured best at 38 KB; the stride-2 blocks of that
    //  configuration, 8 channels per chunk, likewise: see above)
    static const int min_chunk = 22;
    static const int min_chunk_s2 = 8;   // (22x30x24 at 38 KB: 8)
    for (const size_t kb : {52, 76}) {
        const int mc = (int)(lds_budget / per_c) / g.vec * g.vec;
        if (mc >= C || mc >= (stride == 2 ? min_chunk_s2 : min_chunk)) break;
        if (lds_budget < kb * 1024) lds_budget = kb * 1024;
    }
    int maxc = (int)(lds_budget / per_c) / g.vec * g.vec;
    if (maxc < g.vec) maxc = g.vec;
    if (maxc > 256) maxc = 256;
    const int lanes = C / g.vec;
    if (maxc >= C) {
        g.nch = 1;
        g.cchunk = C;
    } else {
        int nch = cdiv(C, maxc);
        g.cchunk = cdiv(lanes, nch) * g.vec;
        g.nch = cdiv(C, g.cchunk);
    }
    // wide workgroups: the tile loads are a latency chain of (pixels / cy) rounds that every thread sits through before
    // the first barrier, so the pixel lanes are made as many as the frame has output pixels (<= 1024 threads forward)
    const int Po_ = Ho * Wo;
    static const int t_fwd = DWF_T_FWD_DEFAULT;
    static const int t_bwd = DWF_T_BWD_DEFAULT;
    g.cx = g.cchunk / g.vec;
    g.cy = t_fwd / g.cx;
    if (g.cy > Po_) g.cy = Po_;
    if (g.cy < 1) g.cy = 1;
    // small frames: several frames per workgroup (fewer partials, less per-block overhead), keeping >= 512 blocks
    int fpb = 1;
    const int work = Ho * Wo * g.cx;
    while (fpb < 8 && work * fpb * 2 <= 4096 && B % (fpb * 2) == 0 && (int64_t)G * (B / (fpb * 2)) * g.nch >= 512) fpb *= 2;
    // large batches (configuration 3): once every CU has two rounds of 4 workgroups anyway, more frames per workgroup only shrink
    // the partial rows (10 + 2 doubles per channel and workgroup: 38 MB per launch at B = 1024 with one frame each) and the
    // finalize kernels that read them
    static const int min_blocks = 2048;
    while (fpb < 8 && B % (fpb * 2) == 0 && (int64_t)G * (B / (fpb * 2)) * g.nch >= min_blocks) fpb *= 2;
    g.fpb = fpb;
    g.nb = B / fpb;
    // backward: 12 double accumulators per channel lane -> at 4 channels per thread the kernel needs > 256 VGPRs (one
    // workgroup per CU); 2 channels per thread keep it at ~150 (3 waves / SIMD)
    g.vec_bwd = g.vec > 2 ? 2 : g.vec;
    g.cx_bwd = g.cchunk / g.vec_bwd;
    g.cy_bwd = t_bwd / g.cx_bwd;
    if (g.cy_bwd > Po_) g.cy_bwd = Po_;
    if (g.cy_bwd < 1) g.cy_bwd = 1;
    const size_t red_f = (size_t)2 * g.cy * g.vec * g.cx * sizeof(double);
    const size_t red_b = (size_t)10 * g.cy_bwd * g.vec_bwd * g.cx_bwd * sizeof(double);
    g.lds_fwd = a_px * g.cchunk * sizeof(float);
    g.lds_bwd = (a_px + d_px) * g.cchunk * sizeof(float);
    if (g.lds_fwd < red_f) g.lds_fwd = red_f;
    if (g.lds_bwd < red_b) g.lds_bwd = red_b;
    // Backward in strip form (whole pixel rows per workgroup where LDS allows): its own frames-per-workgroup count -- the largest
    // power of two (<= 8) that still leaves CDRL_DWS_WGS workgroups.  Measured on the seven shapes of the 90x120 tower at B = 256
    // (isolated, cold; 128 | 256 | 512 | 1024 workgroups): ONE workgroup of 6-10 waves per CU is the optimum -- 11x15x58 - | 38.5 | 43.3 |
    // 52.1 us, 6x8x116 35.7 | 26.7 | 31.0 | 39.6, stride 2: 22x30x58 141 | 83.6 | 86.4 | 93.8, 11x15x116 79.2 | 48.9 | 52.8 | 61.1 -- except
    // for the 3x4 frames (18.0 | 16.2 at 256 | 512): frames of <= 16 pixels take two per CU
    g.fpb_bwd = g.fpb;
    g.nb_bwd = g.nb;
    g.strip.ok = false;
    static const bool strips = !(cdrl_getenv("CDRL_DWS") && atoi(cdrl_getenv("CDRL_DWS")) == 0);
    static const int want_env = 0;
    const int want_wgs = want_env ? want_env : (H * W <= 16 ? 512 : 256);
    if (strips && (stride == 1 || stride == 2) && (int64_t)G * B * H * W * C * 8 < (int64_t)1 << 31) {
        // (the channel-chunk count of the plan does not depend on fpb: plan with 1 first)
        DwsGeom d = stride == 1 ? dws_geom(B, G, 1, H, W, C) : dws2_geom(B, G, 1, H, W, C);
        if (d.ok) {
            int fb = 1;
            while (fb < 8 && B % (fb * 2) == 0 && (int64_t)G * (B / (fb * 2)) * d.nch >= want_wgs) fb *= 2;
            g.strip = stride == 1 ? dws_geom(B, G, fb, H, W, C) : dws2_geom(B, G, fb, H, W, C);
            g.fpb_bwd = fb;
            g.nb_bwd = B / fb;
        }
    }
    return g;
}

int64_t dwf_stats_part_elems(int B, int G, int H, int W, int C, int stride) {
    const DwfGeom g = dwf_geom(B, G, H, W, C, stride);
    return (int64_t)G * std::max(g.nb, g.nb_bwd) * 2 * C;
}
int64_t dwf_filter_part_elems(int B, int G, int H, int W, int C, int stride) {
    const DwfGeom g = dwf_geom(B, G, H, W, C, stride);
    return (int64_t)G * std::max(g.nb, g.nb_bwd) * 10 * C;
}

// reduce one double per (channel lane, vec) over the pixel lanes through LDS; result valid on ty == 0
template <int VEC>
__device__ __forceinline__ void block_colsum(double* sm, double (&a)[VEC], int tx, int ty, int CX, int CY) {
    if (CY == 1) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) sm[(ty * VEC + i) * CX + tx] = a[i];
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            double s = a[i];
            for (int y = 1; y < CY; ++y) s += sm[(y * VEC + i) * CX + tx];
            a[i] = s;
        }
    }
}

// All NQ per-thread quantities go to LDS at once ([q][ty][VEC][CX] doubles) and every (q, i, tx) column is summed
// over the pixel lanes by a different thread: one barrier pair for the whole tail instead of one per quantity
// (12 quantities x (2 barriers + a serial loop on ty == 0) was several microseconds per workgroup).
template <int NQ, int VEC>
__device__ __forceinline__ void block_colsum_all(double* sm, double (&a)[NQ][VEC], int tx, int ty, int CX, int CY, bool on,
                                                 double* out /* &part[row][0][c0] */, int qstride) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < VEC; ++i) sm[((q * CY + ty) * VEC + i) * CX + tx] = a[q][i];
    __syncthreads();
    const int ncol = NQ * VEC;           // columns per channel lane
    for (int j = ty; j < ncol; j += CY) {
        const int q = j / VEC, i = j - q * VEC;
        double s = 0.0;
        for (int y = 0; y < CY; ++y) s += sm[((q * CY + y) * VEC + i) * CX + tx];
        if (on) out[(int64_t)q * qstride + i] = s;
    }
}

// zero the one-pixel border of a padded [Hp][Wp][cchunk] LDS tile (the interior is overwritten by every frame)
template <int VEC>
__device__ __forceinline__ void zero_border(float* t, int Hp, int Wp, int cchunk, int tx, int ty, int CY) {
    VecF<VEC> z;
#pragma unroll
    for (int i = 0; i < VEC; ++i) z.v[i] = 0.0f;
    if (tx * VEC >= cchunk) return;
    const int nb = 2 * Wp + 2 * (Hp - 2);
    for (int j = ty; j < nb; j += CY) {
        int y, x;
        if (j < Wp) {
            y = 0;
            x = j;
        } else if (j < 2 * Wp) {
            y = Hp - 1;
            x = j - Wp;
        } else {
            const int k = j - 2 * Wp;
            y = 1 + (k >> 1);
            x = (k & 1) ? Wp - 1 : 0;
        }
        vstore<VEC>(&t[(y * Wp + x) * cchunk + tx * VEC], z);
    }
}

template <int S, int VEC, bool PRE, class T>
__global__ void __launch_bounds__(DWF_T_FWD) dwf_fwd_kernel(const T* __restrict__ x, const float* __restrict__ pre_stats,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      T* __restrict__ y, double* __restrict__ part, int Bf, int H, int W,
                                                      int Ho, int Wo, int C, int GC, int pt, int pl, int fpb, int nb,
                                                      int cchunk, int nfb) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [H*W][cc]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    // XCD-aware block -> (frame block, channel chunk) map: workgroups id, id + 8, id + 16, ... run on the same XCD (same L2)
    // back to back, so the channel chunks of ONE frame block are placed there: the chunks read interleaved 64-byte pieces of
    // the same pixel rows (row stride = C floats), and with the chunk index as the slow grid dimension every cache line of
    // the frame was fetched from HBM once per chunk (measured 3.4x the algorithmic read bytes on the 22x30 stride-2 block)
    const int nch_ = (C + cchunk - 1) / cchunk;
    const int q_ = blockIdx.x >> 3;
    const int fb_ = (q_ / nch_) * 8 + (blockIdx.x & 7);
    if (fb_ >= nfb) return;
    const int g = fb_ / nb, b = fb_ % nb;
    const int cbase = (q_ % nch_) * cchunk;
    const int cc = min(cchunk, C - cbase);
    const bool on = tx * VEC < cc;
    const int c = cbase + tx * VEC;
    VecF<VEC> wk[9], bv, sc, sh;
    if (on) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = vload<VEC>(w + k * C + c);
        bv = vload<VEC>(bias + c);
        if (PRE) {
            sc = vload<VEC>(pre_stats + 2 * GC + g * C + c);
            sh = vload<VEC>(pre_stats + 3 * GC + g * C + c);
        }
    }
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.0;
    const int P = H * W, Po = Ho * Wo;
    // The tile is zero-padded by one pixel on every side ([H+2][W+2][cc]): the 9 taps of every output pixel are then
    // plain LDS reads at constant offsets from one base address -- no per-tap bounds tests, no integer divisions
    // (the kernel was VALU-bound on exactly that index arithmetic: a wave64 instruction costs 4 issue cycles).
    const int Wp = W + 2, Hp = H + 2;
    zero_border<VEC>(tile, Hp, Wp, cchunk, tx, ty, CY);              // the border stays zero for every frame
    const float invW = 1.0f / (float)W, invWo = 1.0f / (float)Wo;
    for (int f = 0; f < fpb; ++f) {
        const int64_t n = (int64_t)g * Bf + (int64_t)b * fpb + f;
        __syncthreads();
        if (on) {
            const T* xp = x + n * P * C + c;
            for (int p0 = ty; p0 < P; p0 += CY * DWF_UF) {       // DWF_UF independent loads in flight per thread
                VecF<VEC> v[DWF_UF];
#pragma unroll
                for (int u = 0; u < DWF_UF; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) v[u] = vload_raw<VEC>(xp + (int64_t)p * C);
                }
#pragma unroll
                for (int u = 0; u < DWF_UF; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) {
                        vdecode<VEC>(v[u], xp);
                        if (PRE) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) v[u].v[i] = fminf(fmaxf(fmaf(sc.v[i], v[u].v[i], sh.v[i]), 0.0f), 6.0f);
                        }
                        const int iy = (int)(((float)p + 0.5f) * invW), ix = p - iy * W;
                        vstore<VEC>(&tile[((iy + 1) * Wp + ix + 1) * cchunk + tx * VEC], v[u]);
                    }
                }
            }
        }
        __syncthreads();
        if (on) {
            T* yp = y + n * Po * C + c;
            for (int p = ty; p < Po; p += CY) {
                const int oy = (int)(((float)p + 0.5f) * invWo), ox = p - oy * Wo;
                const int o0 = ((oy * S + 1 - pt) * Wp + ox * S + 1 - pl) * cchunk + tx * VEC;    // (index, not pointer:
                VecF<VEC> acc = bv;                                                               //  keeps ds_read)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const VecF<VEC> a = vload<VEC>(&tile[o0 + (ky * Wp + kx) * cchunk]);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(a.v[i], wk[ky * 3 + kx].v[i], acc.v[i]);
                    }
                vstore<VEC>(yp + (int64_t)p * C, acc);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    // (bf16 storage: the statistics are those of the values the consumers will read, i.e. the rounded ones)
                    const double d = sizeof(T) == 2 ? (double)(float)(bf16_t)acc.v[i] : (double)acc.v[i];
                    s1[i] += d;
                    s2[i] += d * d;
                }
            }
        }
    }
    __syncthreads();
    double* sm = reinterpret_cast<double*>(tile);
    double sq[2][VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sq[0][i] = s1[i];
        sq[1][i] = s2[i];
    }
    block_colsum_all<2, VEC>(sm, sq, tx, ty, CX, CY, on, part + ((int64_t)g * nb + b) * 2 * C + c, C);
}

template <int S, int VEC, bool PRE, class T>
__global__ void __launch_bounds__(DWF_T_BWD) dwf_bwd_kernel(const T* __restrict__ x, const float* __restrict__ pre_stats,
                                                      const T* __restrict__ dout, const T* __restrict__ y2,
                                                      const float* __restrict__ post_stats,
                                                      const float* __restrict__ post_coef, const float* __restrict__ w,
                                                      View dx, double* __restrict__ part_bn, double* __restrict__ part_w,
                                                      int Bf, int H, int W, int Ho, int Wo, int C, int GC, int pt, int pl,
                                                      int fpb, int nb, int cchunk, bool dx_al, int nfb, bool reload_y1) {
    extern __shared__ __attribute__((aligned(16))) float tile[];     // A [H*W][cc] | D [Ho*Wo][cc]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    // XCD-aware block -> (frame block, channel chunk) map: workgroups id, id + 8, id + 16, ... run on the same XCD (same L2)
    // back to back, so the channel chunks of ONE frame block are placed there: the chunks read interleaved 64-byte pieces of
    // the same pixel rows (row stride = C floats), and with the chunk index as the slow grid dimension every cache line of
    // the frame was fetched from HBM once per chunk (measured 3.4x the algorithmic read bytes on the 22x30 stride-2 block)
    const int nch_ = (C + cchunk - 1) / cchunk;
    const int q_ = blockIdx.x >> 3;
    const int fb_ = (q_ / nch_) * 8 + (blockIdx.x & 7);
    if (fb_ >= nfb) return;
    const int g = fb_ / nb, b = fb_ % nb;
    const int cbase = (q_ % nch_) * cchunk;
    const int cc = min(cchunk, C - cbase);
    const bool on = tx * VEC < cc;
    const int c = cbase + tx * VEC;
    const int P = H * W, Po = Ho * Wo;
    // zero-padded tiles (see the forward kernel): A = [H+2][W+2][cc], D = [Ho+2][Wo+2][cc]: neither the filter-gradient
    // windows nor the transposed conv need bounds tests
    const int Wp = W + 2, Hp = H + 2;
    const int Wdp = Wo + 2, Hdp = Ho + 2, dpad = 1;
    const int dbase = Hp * Wp * cchunk;          // tile D starts here (indices into `tile`, not pointers: keeps ds_* ops)
    zero_border<VEC>(tile, Hp, Wp, cchunk, tx, ty, CY);
    zero_border<VEC>(tile + dbase, Hdp, Wdp, cchunk, tx, ty, CY);
    const float invW = 1.0f / (float)W, invWo = 1.0f / (float)Wo;
    VecF<VEC> wk[9], sc, sh, mean1, inv1, mean2, inv2, k1, k2, k3;
    if (on) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = vload<VEC>(w + k * C + c);
        if (PRE) {
            mean1 = vload<VEC>(pre_stats + 0 * GC + g * C + c);
            inv1 = vload<VEC>(pre_stats + 1 * GC + g * C + c);
            sc = vload<VEC>(pre_stats + 2 * GC + g * C + c);
            sh = vload<VEC>(pre_stats + 3 * GC + g * C + c);
        }
        mean2 = vload<VEC>(post_stats + 0 * GC + g * C + c);
        inv2 = vload<VEC>(post_stats + 1 * GC + g * C + c);
        k1 = vload<VEC>(post_coef + 0 * GC + g * C + c);
        k2 = vload<VEC>(post_coef + 1 * GC + g * C + c);
        k3 = vload<VEC>(post_coef + 2 * GC + g * C + c);
    }
    // xhat1 = (v - bt1) * rg1 with v = the activated value of tile A (bt1 = beta, rg1 = 1 / gamma) or, for a thread holding a channel
    // with (6 + |beta|) / |gamma| > 170, v = y1 re-read from memory (bt1 = mean, rg1 = invstd); see the input-gradient phase
    VecF<VEC> bt1, rg1;
    bool slow1 = reload_y1;
    if (PRE && on) {
        // error of (a - beta) / gamma: ~ 2^-24 (|a| + |beta|) / |gamma| with a in (0, 6); held to ~1e-5 of xhat's unit scale
#pragma unroll
        for (int i = 0; i < VEC; ++i) slow1 |= !((6.0f + fabsf(fmaf(mean1.v[i], sc.v[i], sh.v[i]))) * fabsf(inv1.v[i]) <= 170.0f * fabsf(sc.v[i]));
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            bt1.v[i] = slow1 ? mean1.v[i] : fmaf(mean1.v[i], sc.v[i], sh.v[i]);
            rg1.v[i] = slow1 ? inv1.v[i] : inv1.v[i] / sc.v[i];
        }
    }
    // filter / bias gradient accumulators: float over the (<= 8) frames of this workgroup (a few hundred fmaf per lane),
    // double from the block reduction on -- 40 fewer VGPRs than double accumulators, one more wave per SIMD
    float gf[10][VEC];
    double gb1[VEC], gb2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        gb1[i] = gb2[i] = 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) gf[k][i] = 0.0f;
    }
    for (int f = 0; f < fpb; ++f) {
        const int64_t n = (int64_t)g * Bf + (int64_t)b * fpb + f;
        __syncthreads();
        const T* xp = x + n * P * C + c;
        if (on) {
            // one merged load loop for the three tensors (x, dout, y2): 3*DWF_U independent loads in flight per thread.
            // Tile A holds the ACTIVATED input relu6(scale1*y1+shift1) (x itself without a pre-BN); tile D the
            // BatchNorm-backward-applied gradient of the depthwise output.
            const T* dp = dout + n * Po * C + c;
            const T* yp = y2 + n * Po * C + c;
            for (int p0 = ty; p0 < P; p0 += CY * DWF_U) {
                VecF<VEC> xa[DWF_U], d[DWF_U], v[DWF_U];
#pragma unroll
                for (int u = 0; u < DWF_U; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) xa[u] = vload_raw<VEC>(xp + (int64_t)p * C);
                    if (p < Po) {
                        d[u] = vload_raw<VEC>(dp + (int64_t)p * C);
                        v[u] = vload_raw<VEC>(yp + (int64_t)p * C);
                    }
                }
#pragma unroll
                for (int u = 0; u < DWF_U; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) {
                        vdecode<VEC>(xa[u], xp);
                        if (PRE) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) xa[u].v[i] = fminf(fmaxf(fmaf(sc.v[i], xa[u].v[i], sh.v[i]), 0.0f), 6.0f);
                        }
                        const int iy = (int)(((float)p + 0.5f) * invW), ix = p - iy * W;
                        vstore<VEC>(&tile[((iy + 1) * Wp + ix + 1) * cchunk + tx * VEC], xa[u]);
                    }
                    if (p < Po) {
                        vdecode<VEC>(d[u], xp);
                        vdecode<VEC>(v[u], xp);
                        VecF<VEC> o;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const float xh = (v[u].v[i] - mean2.v[i]) * inv2.v[i];
                            o.v[i] = k1.v[i] * (d[u].v[i] - k2.v[i] - xh * k3.v[i]);
                        }
                        const int oy = (int)(((float)p + 0.5f) * invWo), ox = p - oy * Wo;
                        vstore<VEC>(&tile[dbase + ((oy + dpad) * Wdp + ox + dpad) * cchunk + tx * VEC], o);
                    }
                }
            }
        }
        __syncthreads();
        if (on) {
            // filter / bias gradient: every output pixel contributes D * A(window)
            for (int p = ty; p < Po; p += CY) {
                const int oy = (int)(((float)p + 0.5f) * invWo), ox = p - oy * Wo;
                const VecF<VEC> d = vload<VEC>(&tile[dbase + ((oy + dpad) * Wdp + ox + dpad) * cchunk + tx * VEC]);
                const int o0 = ((oy * S + 1 - pt) * Wp + ox * S + 1 - pl) * cchunk + tx * VEC;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const VecF<VEC> av = vload<VEC>(&tile[o0 + (ky * Wp + kx) * cchunk]);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gf[ky * 3 + kx][i] = fmaf(av.v[i], d.v[i], gf[ky * 3 + kx][i]);
                    }
#pragma unroll
                for (int i = 0; i < VEC; ++i) gf[9][i] += d.v[i];
            }
            // gradient w.r.t. the depthwise input (transposed conv), masked by ReLU6 of the pre BN, in batches of DXU pixels
            // (stores of a batch issued together).
            constexpr int DXU = (S == 2 && PRE) ? 4 : 1;      // (stride 1: deeper batches cost a wave of occupancy, measured slower)
            auto taps = [&](int p, VecF<VEC>& av) {
                const int iy = (int)(((float)p + 0.5f) * invW), ix = p - iy * W;
                VecF<VEC> acc;
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc.v[i] = 0.0f;
                if (S == 1) {
                    // D is zero-padded: da[iy][ix] = sum_k D[iy + pt - ky][ix + pl - kx] * w[k], no bounds tests
                    const int o0 = dbase + ((iy + pt + 1) * Wdp + ix + pl + 1) * cchunk + tx * VEC;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const VecF<VEC> d = vload<VEC>(&tile[o0 - (ky * Wdp + kx) * cchunk]);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(d.v[i], wk[ky * 3 + kx].v[i], acc.v[i]);
                        }
                } else {
                    // stride 2: which taps reach an input pixel depends only on the parity of (iy+pt, ix+pl) -- 1, 2 or 4
                    // of the 9; D is zero-padded, so the out-of-range neighbours read zeros (no bounds tests)
                    const int ny = iy + pt, nx = ix + pl;
                    const int o00 = dbase + (((ny >> 1) + 1) * Wdp + (nx >> 1) + 1) * cchunk + tx * VEC;   // D[oy0][ox0]
                    const int up = Wdp * cchunk, lf = cchunk;                                            // oy0-1 / ox0-1
                    auto tap = [&](int o, int k) {
                        const VecF<VEC> d = vload<VEC>(&tile[o]);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) acc.v[i] = fmaf(d.v[i], wk[k].v[i], acc.v[i]);
                    };
                    if (ny & 1) {
                        if (nx & 1) {
                            tap(o00, 4);
                        } else {
                            tap(o00, 3);
                            tap(o00 - lf, 5);
                        }
                    } else {
                        if (nx & 1) {
                            tap(o00, 1);
                            tap(o00 - up, 7);
                        } else {
                            tap(o00, 0);
                            tap(o00 - lf, 2);
                            tap(o00 - up, 6);
                            tap(o00 - up - lf, 8);
                        }
                    }
                }
                if (PRE) {      // ReLU6 mask of BN1's output (the activated value is in tile A)
                    av = vload<VEC>(&tile[((iy + 1) * Wp + ix + 1) * cchunk + tx * VEC]);
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        if (!relu6_open(av.v[i])) acc.v[i] = 0.0f;
                }
                return acc;
            };
            // xhat1 for BN1's backward sums comes from the ACTIVATED value in tile A: where the ReLU6 mask is open a = scale y1 + shift,
            // so xhat1 = (a - beta) / gamma (beta = shift + mean scale, 1 / gamma = invstd / scale); where it is closed the gradient is
            // zero and xhat1 is not needed.  No global re-read of y1 in this phase (it was an L2 round trip per pixel batch in front of
            // the stores, both on gfx9's single in-order memory counter).  Channels with (6 + |beta|) / |gamma| > 170 (the division would
            // amplify the rounding of a beyond 1e-5) re-read y1, thread by thread.
            for (int p0 = ty; p0 < P; p0 += CY * DXU) {
                VecF<VEC> acc[DXU], av[DXU];
#pragma unroll
                for (int u = 0; u < DXU; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) acc[u] = taps(p, av[u]);
                }
                if (PRE) {
#pragma unroll
                    for (int u = 0; u < DXU; ++u) {
                        const int p = p0 + u * CY;
                        if (p >= P) continue;
                        if (slow1) {
                            av[u] = vload_raw<VEC>(xp + (int64_t)p * C);
                            vdecode<VEC>(av[u], xp);
                        }
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const float xh = (av[u].v[i] - bt1.v[i]) * rg1.v[i];
                            gb1[i] += (double)acc[u].v[i];
                            gb2[i] += (double)acc[u].v[i] * (double)xh;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < DXU; ++u) {
                    const int p = p0 + u * CY;
                    if (p < P) vstore_view<VEC, T>(dx, n * P + p, c, 0, dx_al, acc[u]);
                }
            }
        }
    }
    __syncthreads();
    double* sm = reinterpret_cast<double*>(tile);
    double gw[10][VEC];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int i = 0; i < VEC; ++i) gw[k][i] = (double)gf[k][i];
    block_colsum_all<10, VEC>(sm, gw, tx, ty, CX, CY, on, part_w + ((int64_t)g * nb + b) * 10 * C + c, C);
    if (PRE) {
        double gq[2][VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            gq[0][i] = gb1[i];
            gq[1][i] = gb2[i];
        }
        block_colsum_all<2, VEC>(sm, gq, tx, ty, CX, CY, on, part_bn + ((int64_t)g * nb + b) * 2 * C + c, C);
    }
}

// (one runtime call per kernel and size, not one per launch.  The kernel is a NON-TYPE template parameter: every
//  dwf_fwd_kernel<S, VEC, PRE> has the same function-pointer TYPE, so a type-keyed static would be shared by all of them and the
//  second instantiation that needs > 64 KB would never get its attribute set)
template <auto Kern>
static int allow_lds(size_t bytes) {
    static size_t allowed[64];          // per device (zero-initialised: 64 KB are always allowed)
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& a = allowed[dev & 63];
    if (a < 64 * 1024) a = 64 * 1024;
    if (bytes > a) {
        CDRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        a = bytes;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Stride-1 backward in STRIP form (round 5).  SQ counters of the pixel-mapped kernel above (profiles/r05_pmc_sq.json): 3266 VALU and
// 435 LDS instructions per wave for 12 channel-pixels per thread and frame (~68 VALU per channel-pixel: three p -> (y, x) conversions,
// 20 LDS addresses and 2 x 4 double-precision operations per pixel around 38 useful FMAs), VALU busy 0.34, waves parked 52 % of
// their cycles at 1.6 waves per SIMD -- an instruction- and latency-chain per workgroup, not a bandwidth limit (2.5 TB/s at 6x8,
// 1.1 TB/s at 3x4 pixels).  Here a thread owns a channel PAIR and a row strip of SW pixels:
//   * both products of the backward come from the SAME 3x3 window of D around the strip's own pixels --
//       da[i]   = sum_k D[i + 1 - k] w[k]            (transposed conv)
//       dW[k]  += a[i] D[i + 1 - k]                  (filter gradient in scatter form: own input pixel x neighbouring D)
//     so only D goes to LDS (half the tile, half the LDS stores); the activated input strip stays in registers across the barrier;
//   * the window slides along the strip in registers: 3 x (SW + 2) LDS reads per SW pixels (3.75 per pixel at SW = 8; 20 before);
//   * strip coordinates are computed once per strip, every LDS / global offset inside a strip is a compile-time constant;
//   * BN1's backward sums are accumulated in float32 along a strip (<= 8 terms, fixed order) and in double across strips;
//   * F frames share one tile batch when the frames are small (one barrier pair per F frames), all 3 x SW global loads of a strip
//     are issued before the first one is consumed (unconditional, clamped addresses).
// Partial-row layout, frames per workgroup and the block -> (frame block, channel chunk) map are those of dwf_bwd_kernel.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

template <int NF, int ND>
__device__ __forceinline__ void block_colsum_mixed(float* smf, float (&a)[NF][2], double (&d)[ND][2], int tx, int ty, int CX, int CY, bool on,
                                                   double* outf, double* outd, int qstride) {
    // [NF][CY][2][CX] floats, then [ND][CY][2][CX] doubles: every (quantity, channel) column is summed over the strip lanes by ONE
    // thread in lane order (deterministic); float32 accumulators are widened when they are read (exact)
    double* smd = reinterpret_cast<double*>(smf + (size_t)NF * CY * 2 * CX);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NF; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) smf[((q * CY + ty) * 2 + i) * CX + tx] = a[q][i];
#pragma unroll
    for (int q = 0; q < ND; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) smd[((q * CY + ty) * 2 + i) * CX + tx] = d[q][i];
    __syncthreads();
    for (int j = ty; j < (NF + ND) * 2; j += CY) {
        const int q = j >> 1, i = j & 1;
        double s = 0.0;
        if (q < NF) {
            for (int y = 0; y < CY; ++y) s += (double)smf[((q * CY + y) * 2 + i) * CX + tx];
            if (on) outf[(int64_t)q * qstride + i] = s;
        } else {
            for (int y = 0; y < CY; ++y) s += smd[(((q - NF) * CY + y) * 2 + i) * CX + tx];
            if (on && outd) outd[(int64_t)(q - NF) * qstride + i] = s;
        }
    }
}

#ifndef DWS_LB
#define DWS_LB 640
#endif
template <int SW, int R, bool PRE, class T>
__global__ void __launch_bounds__(DWS_LB) dws_bwd_kernel(const T* __restrict__ x, const float* __restrict__ pre_stats, const T* __restrict__ dout,
                                                      const T* __restrict__ y2, const float* __restrict__ post_stats,
                                                      const float* __restrict__ post_coef, const float* __restrict__ w, View dx,
                                                      double* __restrict__ part_bn, double* __restrict__ part_w, int Bf, int H, int W, int C,
                                                      int GC, int fpb, int nb, int cchunk, bool dx_al, int nfb, bool reload_y1, int F, int S,
                                                      int tile_floats) {
    // D: [F][H + 2][S * SW + 2][cchunk], zero border and right padding | coefficient table [16][cchunk]
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int nch_ = (C + cchunk - 1) / cchunk;
    const int q_ = blockIdx.x >> 3;
    const int fb_ = (q_ / nch_) * 8 + (blockIdx.x & 7);
    if (fb_ >= nfb) return;
    const int g = fb_ / nb, b = fb_ % nb;
    const int cbase = (q_ % nch_) * cchunk;
    const int cc = min(cchunk, C - cbase);
    const bool on = tx * 2 < cc;
    const int c = cbase + (on ? tx * 2 : 0);          // (idle lanes load channel pair 0 of the chunk: unconditional loads)
    const int P = H * W, Wp = S * SW + 2, Hp = H + 2;
    const int NS1 = H * S, NSB = F * NS1;
    for (int i = (ty * CX + tx) * 2; i < tile_floats; i += CX * CY * 2) *reinterpret_cast<float2*>(&tile[i]) = make_float2(0.0f, 0.0f);
    // Per-channel constants live in LDS, not in registers: the load phase needs 7 pairs (BN1 scale / shift, BN2 mean / invstd / k1..k3),
    // the window phase the 9 filter taps -- read where they are used, they are not live across the other phase (~30 VGPRs)
    const int ctab = tile_floats + tx * 2;             // entry e of this lane's channel pair: tile[ctab + e * cchunk]
    VecF<2> bt1, rg1;
    bool slow1 = PRE && reload_y1;
    {
        VecF<2> mean1, inv1, sc, sh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            mean1.v[i] = sh.v[i] = 0.0f;        // (no pre-BN: the identity, never used)
            inv1.v[i] = sc.v[i] = 1.0f;
        }
        if (PRE) {
            mean1 = vload<2>(pre_stats + 0 * GC + g * C + c);
            inv1 = vload<2>(pre_stats + 1 * GC + g * C + c);
            sc = vload<2>(pre_stats + 2 * GC + g * C + c);
            sh = vload<2>(pre_stats + 3 * GC + g * C + c);
        }
        if (ty == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) vstore<2>(&tile[ctab + k * cchunk], vload<2>(w + k * C + c));
            vstore<2>(&tile[ctab + 9 * cchunk], sc);
            vstore<2>(&tile[ctab + 10 * cchunk], sh);
            vstore<2>(&tile[ctab + 11 * cchunk], vload<2>(post_stats + 0 * GC + g * C + c));
            vstore<2>(&tile[ctab + 12 * cchunk], vload<2>(post_stats + 1 * GC + g * C + c));
            vstore<2>(&tile[ctab + 13 * cchunk], vload<2>(post_coef + 0 * GC + g * C + c));
            vstore<2>(&tile[ctab + 14 * cchunk], vload<2>(post_coef + 1 * GC + g * C + c));
            vstore<2>(&tile[ctab + 15 * cchunk], vload<2>(post_coef + 2 * GC + g * C + c));
        }
        // xhat1 = (a - beta) / gamma from the activated value (see dwf_bwd_kernel); channels beyond the amplification bound re-read y1
#pragma unroll
        for (int i = 0; i < 2; ++i) slow1 |= !((6.0f + fabsf(fmaf(mean1.v[i], sc.v[i], sh.v[i]))) * fabsf(inv1.v[i]) <= 170.0f * fabsf(sc.v[i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bt1.v[i] = slow1 ? mean1.v[i] : fmaf(mean1.v[i], sc.v[i], sh.v[i]);
            rg1.v[i] = slow1 ? inv1.v[i] : inv1.v[i] / sc.v[i];
        }
    }
    float gf[10][2];
    double gb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gb[0][i] = gb[1][i] = 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) gf[k][i] = 0.0f;
    }
    // Buffer descriptors: a strip's pixels sit at voffset (one VGPR per tensor) + j * C * sizeof(T) (a scalar / immediate), so the 3 x SW
    // loads and SW stores of a strip carry no 64-bit address registers (48 + 16 VGPRs in the flat form); an offset beyond the tensor
    // (idle lanes: OOR) reads 0 and drops stores, the pixels of a partial strip that lie beyond its row are loaded and ignored
    constexpr uint32_t OOR = 0x80000000u;
    constexpr int ESZ = (int)sizeof(T);
    const int tot_bytes = (int)((int64_t)nfb * fpb * P * C * ESZ);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, tot_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dout), 0, tot_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(y2), 0, tot_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(vptr<T>(dx), 0, (int)((int64_t)nfb * fpb * P * dx.ld * ESZ), 0x00020000);
    auto ldp = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t so) -> VecF<2> {
        VecF<2> r;
        if (ESZ == 2) {
            r.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
            r.v[1] = 0.0f;
        } else {
            const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
            r.v[0] = __uint_as_float(t[0]);
            r.v[1] = __uint_as_float(t[1]);
        }
        return r;
    };
    const float invNS1 = 1.0f / (float)NS1, invS = 1.0f / (float)S;
    for (int f0 = 0; f0 < fpb; f0 += F) {
        VecF<2> A[R][SW];
        int pix0[R], lbase[R], nx[R];          // first pixel of the strip (global pixel index), its LDS offset, pixels in range
        __syncthreads();                          // the previous batch's windows have been read (first pass: table + zeros written)
        {
            const VecF<2> sc = vload<2>(&tile[ctab + 9 * cchunk]), sh = vload<2>(&tile[ctab + 10 * cchunk]);
            const VecF<2> mean2 = vload<2>(&tile[ctab + 11 * cchunk]), inv2 = vload<2>(&tile[ctab + 12 * cchunk]);
            const VecF<2> k1 = vload<2>(&tile[ctab + 13 * cchunk]), k2 = vload<2>(&tile[ctab + 14 * cchunk]), k3 = vload<2>(&tile[ctab + 15 * cchunk]);
#pragma unroll
            for (int rd = 0; rd < R; ++rd) {
                const int s = ty + rd * CY;
                const bool valid = on && s < NSB;
                const int sv = s < NSB ? s : 0;
                const int fl = (int)(((float)sv + 0.5f) * invNS1);
                const int rem = sv - fl * NS1;
                const int r = (int)(((float)rem + 0.5f) * invS);
                const int x0 = (rem - r * S) * SW;
                const int64_t n = (int64_t)g * Bf + (int64_t)b * fpb + f0 + fl;
                nx[rd] = valid ? min(SW, W - x0) : 0;
                pix0[rd] = (int)(n * P) + r * W + x0;            // (byte offsets fit 31 bits: checked by the launcher)
                lbase[rd] = ((fl * Hp + r + 1) * Wp + x0 + 1) * cchunk + tx * 2;
                const uint32_t vo = valid ? (uint32_t)(pix0[rd] * C + c) * ESZ : OOR;
                VecF<2> xa[SW], d[SW], v[SW];
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    xa[j] = ldp(rsX, vo, (uint32_t)(j * C * ESZ));
                    d[j] = ldp(rsD, vo, (uint32_t)(j * C * ESZ));
                    v[j] = ldp(rsY, vo, (uint32_t)(j * C * ESZ));
                }
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    vdecode<2>(xa[j], x);
                    vdecode<2>(d[j], x);
                    vdecode<2>(v[j], x);
                    VecF<2> o;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float a = PRE ? fminf(fmaxf(fmaf(sc.v[i], xa[j].v[i], sh.v[i]), 0.0f), 6.0f) : xa[j].v[i];
                        A[rd][j].v[i] = j < nx[rd] ? a : 0.0f;
                        const float xh = (v[j].v[i] - mean2.v[i]) * inv2.v[i];
                        o.v[i] = k1.v[i] * (d[j].v[i] - k2.v[i] - xh * k3.v[i]);
                    }
                    if (j < nx[rd]) vstore<2>(&tile[lbase[rd] + j * cchunk], o);
                }
            }
        }
        __syncthreads();
        {
            VecF<2> wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = vload<2>(&tile[ctab + k * cchunk]);
#pragma unroll
            for (int rd = 0; rd < R; ++rd) {
                if (nx[rd] == 0) continue;
                VecF<2> da[SW];
#pragma unroll
                for (int j = 0; j < SW; ++j) da[j].v[0] = da[j].v[1] = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // row r + 1 - ky of D, columns x0 - 1 .. x0 + SW (padded coordinates: + 1 each); one row in registers at a time
                    const int rb = lbase[rd] + ((1 - ky) * Wp - 1) * cchunk;
                    VecF<2> dr[SW + 2];
#pragma unroll
                    for (int m = 0; m < SW + 2; ++m) dr[m] = vload<2>(&tile[rb + m * cchunk]);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < SW; ++j)
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                da[j].v[i] = fmaf(dr[j + 2 - kx].v[i], wk[ky * 3 + kx].v[i], da[j].v[i]);
                                gf[ky * 3 + kx][i] = fmaf(A[rd][j].v[i], dr[j + 2 - kx].v[i], gf[ky * 3 + kx][i]);
                            }
                    if (ky == 1) {
#pragma unroll
                        for (int j = 0; j < SW; ++j)
#pragma unroll
                            for (int i = 0; i < 2; ++i) gf[9][i] += dr[j + 1].v[i];       // (columns beyond W hold zeros)
                    }
                    // One row of D in registers at a time (20 VGPRs, not 60): the accumulators are pinned here, so this row's FMAs are
                    // complete before the next row's LDS reads are issued (pure arithmetic is otherwise sunk below all 30 reads)
#pragma unroll
                    for (int j = 0; j < SW; ++j) asm volatile("" : "+v"(da[j].v[0]), "+v"(da[j].v[1]));
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) asm volatile("" : "+v"(gf[ky * 3 + kx][0]), "+v"(gf[ky * 3 + kx][1]));
                    asm volatile("" ::: "memory");
                }
                if (PRE) {
                VecF<2> av[SW];
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    av[j] = A[rd][j];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (!relu6_open(av[j].v[i])) da[j].v[i] = 0.0f;
                }
                if (slow1) {        // (rare: a channel whose |beta| / |gamma| would amplify the rounding of a beyond 1e-5 -- xhat1 from y1 itself)
                    const uint32_t vo = (uint32_t)(pix0[rd] * C + c) * ESZ;
#pragma unroll
                    for (int j = 0; j < SW; ++j) av[j] = ldp(rsX, vo, (uint32_t)(j * C * ESZ));
#pragma unroll
                    for (int j = 0; j < SW; ++j) vdecode<2>(av[j], x);
                }
                float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < SW; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float xh = (av[j].v[i] - bt1.v[i]) * rg1.v[i];
                        s1[i] += da[j].v[i];              // (masked / out-of-row pixels carry da = 0)
                        s2[i] = fmaf(da[j].v[i], xh, s2[i]);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    gb[0][i] += (double)s1[i];
                    gb[1][i] += (double)s2[i];
                }
                }
                if (dx_al) {
                    const uint32_t vo = (uint32_t)(pix0[rd] * dx.ld + dx.coff + c) * ESZ;
#pragma unroll
                    for (int j = 0; j < SW; ++j) {
                        const uint32_t voj = j < nx[rd] ? vo : OOR;
                        if (ESZ == 2) __builtin_amdgcn_raw_buffer_store_b32(bf_pack(da[j].v[0], da[j].v[1]), rsO, voj, (uint32_t)(j * dx.ld * ESZ), 0);
                        else {
                            u32x2_t t;
                            t[0] = __float_as_uint(da[j].v[0]);
                            t[1] = __float_as_uint(da[j].v[1]);
                            __builtin_amdgcn_raw_buffer_store_b64(t, rsO, voj, (uint32_t)(j * dx.ld * ESZ), 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < SW; ++j)
                        if (j < nx[rd]) vstore_view<2, T>(dx, (int64_t)pix0[rd] + j, c, 0, false, da[j]);
                }
            }
        }
    }
    block_colsum_mixed<10, 2>(tile, gf, gb, tx, ty, CX, CY, on, part_w + ((int64_t)g * nb + b) * 10 * C + c,
                              PRE ? part_bn + ((int64_t)g * nb + b) * 2 * C + c : nullptr, C);
}

// Stride-2 backward in strip form.  An input pixel (iy, ix) meets the taps k with ky = (iy + pt) mod 2 (+ 2), kx = (ix + pl) mod 2 (+ 2):
// 1, 2 or 4 of the 9, at D[(iy + pt - ky) / 2][(ix + pl - kx) / 2].  A thread owns a channel pair and walks input-row strips of 8
// pixels: the column parities inside a strip are compile-time constants (PL = pl), the row parity is a per-strip select between filter
// rows (no divergent code), and a strip reads 2 rows x 5 columns of D from LDS for its 8 pixels.  The D tile (BatchNorm-backward applied
// on load, bias gradient summed by the producing thread) is built first from the output-sized tensors; the 4x larger input is streamed
// through registers only: load strip -> [BN1 apply + ReLU6] -> da, dW += a (x) D -> mask -> BN1 sums -> store.
template <int PL, bool PRE, class T>
__global__ void __launch_bounds__(512) dws2_bwd_kernel(const T* __restrict__ x, const float* __restrict__ pre_stats, const T* __restrict__ dout,
                                                       const T* __restrict__ y2, const float* __restrict__ post_stats,
                                                       const float* __restrict__ post_coef, const float* __restrict__ w, View dx,
                                                       double* __restrict__ part_bn, double* __restrict__ part_w, int Bf, int H, int W, int Ho,
                                                       int Wo, int C, int GC, int pt, int fpb, int nb, int cchunk, bool dx_al, int nfb,
                                                       bool reload_y1, int F, int S, int Wdp, int tile_floats) {
    constexpr int SW = 8, ND = SW / 2 + 1 + PL;     // D columns a strip touches per row
    extern __shared__ __attribute__((aligned(16))) float tile[];     // D: [F][Ho + 2][Wdp][cchunk] (zero frame) | coefficient table [16][cchunk]
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int CX = blockDim.x, CY = blockDim.y;
    const int nch_ = (C + cchunk - 1) / cchunk;
    const int q_ = blockIdx.x >> 3;
    const int fb_ = (q_ / nch_) * 8 + (blockIdx.x & 7);
    if (fb_ >= nfb) return;
    const int g = fb_ / nb, b = fb_ % nb;
    const int cbase = (q_ % nch_) * cchunk;
    const int cc = min(cchunk, C - cbase);
    const bool on = tx * 2 < cc;
    const int c = cbase + (on ? tx * 2 : 0);
    const int P = H * W, Po = Ho * Wo, Hdp = Ho + 2;
    for (int i = (ty * CX + tx) * 2; i < tile_floats; i += CX * CY * 2) *reinterpret_cast<float2*>(&tile[i]) = make_float2(0.0f, 0.0f);
    const int ctab = tile_floats + tx * 2;
    VecF<2> bt1, rg1, sc, sh;
    bool slow1 = reload_y1;
    if (PRE) {
        const VecF<2> mean1 = vload<2>(pre_stats + 0 * GC + g * C + c), inv1 = vload<2>(pre_stats + 1 * GC + g * C + c);
        sc = vload<2>(pre_stats + 2 * GC + g * C + c);
        sh = vload<2>(pre_stats + 3 * GC + g * C + c);
#pragma unroll
        for (int i = 0; i < 2; ++i) slow1 |= !((6.0f + fabsf(fmaf(mean1.v[i], sc.v[i], sh.v[i]))) * fabsf(inv1.v[i]) <= 170.0f * fabsf(sc.v[i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bt1.v[i] = slow1 ? mean1.v[i] : fmaf(mean1.v[i], sc.v[i], sh.v[i]);
            rg1.v[i] = slow1 ? inv1.v[i] : inv1.v[i] / sc.v[i];
        }
    }
    if (ty == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) vstore<2>(&tile[ctab + k * cchunk], vload<2>(w + k * C + c));
        vstore<2>(&tile[ctab + 11 * cchunk], vload<2>(post_stats + 0 * GC + g * C + c));
        vstore<2>(&tile[ctab + 12 * cchunk], vload<2>(post_stats + 1 * GC + g * C + c));
        vstore<2>(&tile[ctab + 13 * cchunk], vload<2>(post_coef + 0 * GC + g * C + c));
        vstore<2>(&tile[ctab + 14 * cchunk], vload<2>(post_coef + 1 * GC + g * C + c));
        vstore<2>(&tile[ctab + 15 * cchunk], vload<2>(post_coef + 2 * GC + g * C + c));
    }
    float gf[10][2];
    double gb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gb[0][i] = gb[1][i] = 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) gf[k][i] = 0.0f;
    }
    constexpr uint32_t OOR = 0x80000000u;
    constexpr int ESZ = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)((int64_t)nfb * fpb * P * C * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dout), 0, (int)((int64_t)nfb * fpb * Po * C * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(y2), 0, (int)((int64_t)nfb * fpb * Po * C * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(vptr<T>(dx), 0, (int)((int64_t)nfb * fpb * P * dx.ld * ESZ), 0x00020000);
    auto ldp = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t vo, uint32_t so) -> VecF<2> {
        VecF<2> r;
        if (ESZ == 2) {
            r.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
            r.v[1] = 0.0f;
        } else {
            const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
            r.v[0] = __uint_as_float(t[0]);
            r.v[1] = __uint_as_float(t[1]);
        }
        return r;
    };
    const int So = (Wo + 3) >> 2;                       // output strips of 4 pixels (tile construction)
    const int NO1 = Ho * So, NS1 = H * S;
    const float invNO1 = 1.0f / (float)NO1, invSo = 1.0f / (float)So, invNS1 = 1.0f / (float)NS1, invS = 1.0f / (float)S;
    for (int f0 = 0; f0 < fpb; f0 += F) {
        __syncthreads();                                 // the previous batch's windows have been read (first pass: table + zeros written)
        {
            const VecF<2> mean2 = vload<2>(&tile[ctab + 11 * cchunk]), inv2 = vload<2>(&tile[ctab + 12 * cchunk]);
            const VecF<2> k1 = vload<2>(&tile[ctab + 13 * cchunk]), k2 = vload<2>(&tile[ctab + 14 * cchunk]), k3 = vload<2>(&tile[ctab + 15 * cchunk]);
            for (int s = ty; s < F * NO1; s += CY) {
                const int fl = (int)(((float)s + 0.5f) * invNO1);
                const int rem = s - fl * NO1;
                const int r = (int)(((float)rem + 0.5f) * invSo);
                const int x0 = (rem - r * So) * 4;
                const int64_t n = (int64_t)g * Bf + (int64_t)b * fpb + f0 + fl;
                const int nxo = min(4, Wo - x0);
                const uint32_t vo = on ? (uint32_t)(((int)(n * Po) + r * Wo + x0) * C + c) * ESZ : OOR;
                VecF<2> d[4], v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j] = ldp(rsD, vo, (uint32_t)(j * C * ESZ));
                    v[j] = ldp(rsY, vo, (uint32_t)(j * C * ESZ));
                }
                const int lb = ((fl * Hdp + r + 1) * Wdp + x0 + 1) * cchunk + tx * 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vdecode<2>(d[j], x);
                    vdecode<2>(v[j], x);
                    VecF<2> o;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float xh = (v[j].v[i] - mean2.v[i]) * inv2.v[i];
                        o.v[i] = k1.v[i] * (d[j].v[i] - k2.v[i] - xh * k3.v[i]);
                    }
                    if (on && j < nxo) {
                        vstore<2>(&tile[lb + j * cchunk], o);
                        gf[9][0] += o.v[0];
                        gf[9][1] += o.v[1];
                    }
                }
            }
        }
        __syncthreads();
        {
            // strips of the input, streamed through registers; the NEXT strip's 8 loads are in flight while this one is computed
            const int NSB = F * NS1;
            auto coords = [&](int s, int& fl, int& r, int& x0, int& pix0) {
                fl = (int)(((float)s + 0.5f) * invNS1);
                const int rem = s - fl * NS1;
                r = (int)(((float)rem + 0.5f) * invS);
                x0 = (rem - r * S) * SW;
                const int64_t n = (int64_t)g * Bf + (int64_t)b * fpb + f0 + fl;
                pix0 = (int)(n * P) + r * W + x0;
            };
            VecF<2> nxt[SW];
            {
                int fl, r, x0, pix0;
                coords(min(ty, NSB - 1), fl, r, x0, pix0);
                const uint32_t vo = (on && ty < NSB) ? (uint32_t)(pix0 * C + c) * ESZ : OOR;
#pragma unroll
                for (int j = 0; j < SW; ++j) nxt[j] = ldp(rsX, vo, (uint32_t)(j * C * ESZ));
            }
            for (int s = ty; s < NSB; s += CY) {
                int fl, r, x0, pix0;
                coords(s, fl, r, x0, pix0);
                const int nx = on ? min(SW, W - x0) : 0;
                const uint32_t vo = on ? (uint32_t)(pix0 * C + c) * ESZ : OOR;
                VecF<2> A[SW], av[SW];
#pragma unroll
                for (int j = 0; j < SW; ++j) av[j] = nxt[j];
                {
                    int fl2, r2, x02, pix02;
                    const int s2 = s + CY;
                    coords(min(s2, NSB - 1), fl2, r2, x02, pix02);
                    const uint32_t vo2 = (on && s2 < NSB) ? (uint32_t)(pix02 * C + c) * ESZ : OOR;
#pragma unroll
                    for (int j = 0; j < SW; ++j) nxt[j] = ldp(rsX, vo2, (uint32_t)(j * C * ESZ));
                }
                const int ny = r + pt;
                const bool odd = ny & 1;
                const int oyA = ny >> 1;                                 // D row of tap row kyA = odd ? 1 : 0; row B = oyA - 1 (tap row 2, even only)
                // (filter rows from the LDS table, selected by address: row kyA = odd ? 1 : 0 against D row A, row 2 -- even strips only -- against row B)
                VecF<2> wa[3], wb[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    wa[kx] = vload<2>(&tile[ctab + ((odd ? 3 : 0) + kx) * cchunk]);
                    wb[kx] = vload<2>(&tile[ctab + (6 + kx) * cchunk]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) wb[kx].v[i] = odd ? 0.0f : wb[kx].v[i];
                }
                // D columns (x0 + PL) / 2 - 1 + m, m = 0 .. ND - 1 (padded coordinates: + 1)
                const int lb = ((fl * Hdp + oyA + 1) * Wdp + ((x0 + PL) >> 1)) * cchunk + tx * 2;
                VecF<2> da_[2][ND];
#pragma unroll
                for (int m = 0; m < ND; ++m) {
                    da_[0][m] = vload<2>(&tile[lb + m * cchunk]);
                    da_[1][m] = vload<2>(&tile[lb + (m - Wdp) * cchunk]);
                }
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    vdecode<2>(av[j], x);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float a = PRE ? fminf(fmaxf(fmaf(sc.v[i], av[j].v[i], sh.v[i]), 0.0f), 6.0f) : av[j].v[i];
                        A[j].v[i] = j < nx ? a : 0.0f;
                    }
                }
                VecF<2> da[SW];
                float ca[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, cb[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    // nx_j = x0 + j + PL; column index into the 5-wide window: (j + PL') with PL' = (x0 + PL) & 1 = PL (x0 is even)
                    const int e = j + PL;                  // nx - (first window column's nx) ... the window starts at ox = (x0 + PL) / 2 - 1
                    da[j].v[0] = da[j].v[1] = 0.0f;
                    if ((e & 1) == 0) {                    // even column parity: taps kx = 0 (ox = e / 2) and kx = 2 (ox = e / 2 - 1)
                        const int m0 = (e >> 1) + 1, m2 = (e >> 1);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            da[j].v[i] = fmaf(da_[0][m0].v[i], wa[0].v[i], da[j].v[i]);
                            da[j].v[i] = fmaf(da_[0][m2].v[i], wa[2].v[i], da[j].v[i]);
                            da[j].v[i] = fmaf(da_[1][m0].v[i], wb[0].v[i], da[j].v[i]);
                            da[j].v[i] = fmaf(da_[1][m2].v[i], wb[2].v[i], da[j].v[i]);
                            ca[0][i] = fmaf(A[j].v[i], da_[0][m0].v[i], ca[0][i]);
                            ca[2][i] = fmaf(A[j].v[i], da_[0][m2].v[i], ca[2][i]);
                            cb[0][i] = fmaf(A[j].v[i], da_[1][m0].v[i], cb[0][i]);
                            cb[2][i] = fmaf(A[j].v[i], da_[1][m2].v[i], cb[2][i]);
                        }
                    } else {                               // odd: tap kx = 1 at ox = (e - 1) / 2
                        const int m1 = ((e - 1) >> 1) + 1;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            da[j].v[i] = fmaf(da_[0][m1].v[i], wa[1].v[i], da[j].v[i]);
                            da[j].v[i] = fmaf(da_[1][m1].v[i], wb[1].v[i], da[j].v[i]);
                            ca[1][i] = fmaf(A[j].v[i], da_[0][m1].v[i], ca[1][i]);
                            cb[1][i] = fmaf(A[j].v[i], da_[1][m1].v[i], cb[1][i]);
                        }
                    }
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        gf[kx][i] += odd ? 0.0f : ca[kx][i];
                        gf[3 + kx][i] += odd ? ca[kx][i] : 0.0f;
                        gf[6 + kx][i] += odd ? 0.0f : cb[kx][i];
                    }
                if (PRE) {
#pragma unroll
                    for (int j = 0; j < SW; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (!relu6_open(A[j].v[i])) da[j].v[i] = 0.0f;
                    if (slow1) {
#pragma unroll
                        for (int j = 0; j < SW; ++j) A[j] = ldp(rsX, vo, (uint32_t)(j * C * ESZ));
#pragma unroll
                        for (int j = 0; j < SW; ++j) vdecode<2>(A[j], x);
                    }
                    float s1[2] = {0.0f, 0.0f}, s2[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < SW; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const float xh = (A[j].v[i] - bt1.v[i]) * rg1.v[i];
                            const float dv = j < nx ? da[j].v[i] : 0.0f;
                            s1[i] += dv;
                            s2[i] = fmaf(dv, xh, s2[i]);
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        gb[0][i] += (double)s1[i];
                        gb[1][i] += (double)s2[i];
                    }
                }
                if (dx_al) {
                    const uint32_t vd = (uint32_t)(pix0 * dx.ld + dx.coff + c) * ESZ;
#pragma unroll
                    for (int j = 0; j < SW; ++j) {
                        const uint32_t voj = j < nx ? vd : OOR;
                        if (ESZ == 2) __builtin_amdgcn_raw_buffer_store_b32(bf_pack(da[j].v[0], da[j].v[1]), rsO, voj, (uint32_t)(j * dx.ld * ESZ), 0);
                        else {
                            u32x2_t t;
                            t[0] = __float_as_uint(da[j].v[0]);
                            t[1] = __float_as_uint(da[j].v[1]);
                            __builtin_amdgcn_raw_buffer_store_b64(t, rsO, voj, (uint32_t)(j * dx.ld * ESZ), 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < SW; ++j)
                        if (j < nx) vstore_view<2, T>(dx, (int64_t)pix0 + j, c, 0, false, da[j]);
                }
            }
        }
    }
    block_colsum_mixed<10, 2>(tile, gf, gb, tx, ty, CX, CY, on, part_w + ((int64_t)g * nb + b) * 10 * C + c,
                              PRE ? part_bn + ((int64_t)g * nb + b) * 2 * C + c : nullptr, C);
}

// strip geometry of the stride-1 backward; ok == false -> the pixel-mapped kernel runs
static DwsGeom dws_geom(int B, int G, int fpb, int H, int W, int C) {
    DwsGeom d;
    d.ok = false;
    if (C % 2 != 0) return d;
    static const int lds_kb = 56;
    static const int max_thr = 384;
    const size_t budget = (size_t)lds_kb * 1024;
    // strips of 8 pixels when every thread then has ONE strip per tile batch (the activated strips live in registers across the
    // barrier: 16 VGPRs per strip of 8), strips of 4 with up to 3 per thread otherwise
    struct Try {
        int sw, thr, rmax;
    };
    // strips of 8 pixels with ONE strip per thread and tile batch (the activated strip lives in registers across the barrier: 16 VGPRs),
    // in workgroups of up to 384 threads, then up to 640 (11x15 frames: 22 strips x 29 channel pairs); strips of 4 with up to 3 per thread
    // (whole pixel rows -- one channel chunk -- first: 232-byte accesses instead of 120-byte ones)
    for (int nch = 1; nch <= 16; ++nch) {
        const int cchunk = cdiv(C / 2, nch) * 2;
        if (cchunk < 8 && nch > 1) break;
        for (const Try t : {Try{8, max_thr, 1}, Try{8, 640, 1}, Try{4, max_thr, 3}}) {
            const int sw = t.sw;
            if (sw == 8 && W <= 4) continue;
            const int S = cdiv(W, sw);
            const int Wp = S * sw + 2, Hp = H + 2, NS1 = H * S;
            const int cx = cchunk / 2;
            const int maxcy = t.thr / cx;
            if (maxcy < 1) continue;
            const size_t lds1 = (size_t)Hp * Wp * cchunk * sizeof(float);
            if (lds1 > budget) continue;
            const int R = cdiv(NS1, maxcy);
            if (R > t.rmax) continue;
            int F = 1;
            if (R == 1)
                while (F * 2 <= fpb && fpb % (F * 2) == 0 && (size_t)(F * 2) * lds1 <= budget && F * 2 * NS1 <= maxcy) F *= 2;
            d.ok = true;
            d.sw = sw;
            d.S = S;
            d.R = R;
            d.F = F;
            d.cchunk = cchunk;
            d.nch = cdiv(C, cchunk);
            d.cx = cx;
            d.cy = cdiv(F * NS1, R);
            d.tile_floats = F * Hp * Wp * cchunk;
            const size_t red = (size_t)d.cx * d.cy * (10 * 2 * sizeof(float) + 2 * 2 * sizeof(double));
            d.lds = std::max((size_t)(d.tile_floats + 16 * cchunk) * sizeof(float), red);
            return d;
        }
    }
    return d;
}

static DwsGeom dws2_geom(int B, int G, int fpb, int H, int W, int C) {
    DwsGeom d;
    d.ok = false;
    if (C % 2 != 0) return d;
    static const int lds_kb = 60;
    static const int max_thr = 384;
    const int Ho = same_out(H, 2), Wo = same_out(W, 2);
    const int S = cdiv(W, 8), NS1 = H * S;
    const int Wdp = std::max(4 * S + 2, Wo + 2), Hdp = Ho + 2;
    for (int nch = 1; nch <= 16; ++nch) {
        const int cchunk = cdiv(C / 2, nch) * 2;
        if (cchunk < 8 && nch > 1) break;
        const int cx = cchunk / 2;
        const int maxcy = max_thr / cx;
        if (maxcy < 1) continue;
        const size_t lds1 = (size_t)Hdp * Wdp * cchunk * sizeof(float);
        if (lds1 > (size_t)lds_kb * 1024) continue;
        int F = 1;
        while (F * 2 <= fpb && fpb % (F * 2) == 0 && (size_t)(F * 2) * lds1 <= (size_t)lds_kb * 1024 && F * NS1 < maxcy) F *= 2;
        d.ok = true;
        d.sw = 8;
        d.S = S;
        d.R = 1;
        d.wdp = Wdp;
        d.F = F;
        d.cchunk = cchunk;
        d.nch = cdiv(C, cchunk);
        d.cx = cx;
        d.cy = std::min(maxcy, F * NS1);
        d.tile_floats = F * Hdp * Wdp * cchunk;
        const size_t red = (size_t)d.cx * d.cy * (10 * 2 * sizeof(float) + 2 * 2 * sizeof(double));
        d.lds = std::max((size_t)(d.tile_floats + 16 * cchunk) * sizeof(float), red);
        return d;
    }
    return d;
}

template <int PL, bool PRE, class T>
static int launch_dws2_bwd(const DwfGeom& g, const DwsGeom& d, hipStream_t st, const float* x, const float* pre_stats, const float* dout,
                           const float* y2, const float* post_stats, const float* post_coef, const float* w, View dx, double* part_bn,
                           double* part_w, int G, int B, int H, int W, int C) {
    static const bool reload_y1 = cdrl_getenv("CDRL_DWF_XHAT_RELOAD") && atoi(cdrl_getenv("CDRL_DWF_XHAT_RELOAD")) == 1;
    CDRL_TRY((allow_lds<dws2_bwd_kernel<PL, PRE, T>>(d.lds)));
    hipLaunchKernelGGL((dws2_bwd_kernel<PL, PRE, T>), dim3(cdiv(G * g.nb_bwd, 8) * 8 * d.nch), dim3(d.cx, d.cy), d.lds, st,
                       reinterpret_cast<const T*>(x), pre_stats, reinterpret_cast<const T*>(dout), reinterpret_cast<const T*>(y2), post_stats,
                       post_coef, w, dx, part_bn, part_w, B, H, W, same_out(H, 2), same_out(W, 2), C, G * C, same_pad_before(H, 2), g.fpb_bwd,
                       g.nb_bwd, d.cchunk, view_aligned(dx, 2), G * g.nb_bwd, reload_y1, d.F, d.S, d.wdp, d.tile_floats);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int SW, int R, bool PRE, class T>
static int launch_dws_bwd(const DwfGeom& g, const DwsGeom& d, hipStream_t st, const float* x, const float* pre_stats, const float* dout,
                          const float* y2, const float* post_stats, const float* post_coef, const float* w, View dx, double* part_bn,
                          double* part_w, int G, int B, int H, int W, int C) {
    static const bool reload_y1 = cdrl_getenv("CDRL_DWF_XHAT_RELOAD") && atoi(cdrl_getenv("CDRL_DWF_XHAT_RELOAD")) == 1;
    CDRL_TRY((allow_lds<dws_bwd_kernel<SW, R, PRE, T>>(d.lds)));
    hipLaunchKernelGGL((dws_bwd_kernel<SW, R, PRE, T>), dim3(cdiv(G * g.nb_bwd, 8) * 8 * d.nch), dim3(d.cx, d.cy), d.lds, st,
                       reinterpret_cast<const T*>(x), pre_stats, reinterpret_cast<const T*>(dout), reinterpret_cast<const T*>(y2), post_stats,
                       post_coef, w, dx, part_bn, part_w, B, H, W, C, G * C, g.fpb_bwd, g.nb_bwd, d.cchunk, view_aligned(dx, 2), G * g.nb_bwd, reload_y1,
                       d.F, d.S, d.tile_floats);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int S, int VEC, bool PRE, class T>
static int launch_dwf_fwd(const DwfGeom& g, hipStream_t st, const float* x, const float* pre_stats, const float* w,
                          const float* bias, float* y, double* part, int G, int B, int H, int W, int C) {
    const int Ho = same_out(H, S), Wo = same_out(W, S);
    CDRL_TRY((allow_lds<dwf_fwd_kernel<S, VEC, PRE, T>>(g.lds_fwd)));
    hipLaunchKernelGGL((dwf_fwd_kernel<S, VEC, PRE, T>), dim3(cdiv(G * g.nb, 8) * 8 * g.nch), dim3(g.cx, g.cy), g.lds_fwd, st,
                       reinterpret_cast<const T*>(x), pre_stats, w, bias, reinterpret_cast<T*>(y), part, B, H, W, Ho, Wo, C, G * C,
                       same_pad_before(H, S), same_pad_before(W, S), g.fpb, g.nb, g.cchunk, G * g.nb);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int S, int VEC>
static int launch_dwf_fwd_pre(const DwfGeom& g, hipStream_t st, const float* x, const float* pre_stats, const float* w,
                              const float* bias, float* y, double* part, int G, int B, int H, int W, int C, int at) {
    if (at) {
        if (pre_stats) return launch_dwf_fwd<S, VEC, true, bf16_t>(g, st, x, pre_stats, w, bias, y, part, G, B, H, W, C);
        return launch_dwf_fwd<S, VEC, false, bf16_t>(g, st, x, pre_stats, w, bias, y, part, G, B, H, W, C);
    }
    if (pre_stats) return launch_dwf_fwd<S, VEC, true, float>(g, st, x, pre_stats, w, bias, y, part, G, B, H, W, C);
    return launch_dwf_fwd<S, VEC, false, float>(g, st, x, pre_stats, w, bias, y, part, G, B, H, W, C);
}

int dwf_fwd(const float* x, const float* pre_stats, const float* w, const float* bias, float* y, double* part, int G,
            int B, int H, int W, int C, int stride, hipStream_t st, int at) {
    if (stride != 1 && stride != 2) {
        set_error("dwf_fwd: stride must be 1 or 2");
        return -1;
    }
    const DwfGeom g = dwf_geom(B, G, H, W, C, stride);
    if (g.lds_fwd > 150 * 1024) {
        set_error("dwf_fwd: frame %dx%d does not fit LDS even at %d channels", H, W, g.cchunk);
        return -1;
    }
#define CDRL_DWF_FWD(S, V) return launch_dwf_fwd_pre<S, V>(g, st, x, pre_stats, w, bias, y, part, G, B, H, W, C, at)
    if (stride == 1) {
        if (g.vec == 4) CDRL_DWF_FWD(1, 4);
        if (g.vec == 2) CDRL_DWF_FWD(1, 2);
        CDRL_DWF_FWD(1, 1);
    }
    if (g.vec == 4) CDRL_DWF_FWD(2, 4);
    if (g.vec == 2) CDRL_DWF_FWD(2, 2);
    CDRL_DWF_FWD(2, 1);
#undef CDRL_DWF_FWD
}

template <int S, int VEC, bool PRE, class T>
static int launch_dwf_bwd(const DwfGeom& g, hipStream_t st, const float* x, const float* pre_stats, const float* dout,
                          const float* y2, const float* post_stats, const float* post_coef, const float* w, View dx,
                          double* part_bn, double* part_w, int G, int B, int H, int W, int C) {
    const int Ho = same_out(H, S), Wo = same_out(W, S);
    // CDRL_DWF_XHAT_RELOAD=1: xhat1 from a re-read of y1 for every channel (the form of rounds 1-3) instead of the activated tile
    static const bool reload_y1 = cdrl_getenv("CDRL_DWF_XHAT_RELOAD") && atoi(cdrl_getenv("CDRL_DWF_XHAT_RELOAD")) == 1;
    CDRL_TRY((allow_lds<dwf_bwd_kernel<S, VEC, PRE, T>>(g.lds_bwd)));
    hipLaunchKernelGGL((dwf_bwd_kernel<S, VEC, PRE, T>), dim3(cdiv(G * g.nb, 8) * 8 * g.nch), dim3(g.cx_bwd, g.cy_bwd), g.lds_bwd, st,
                       reinterpret_cast<const T*>(x), pre_stats, reinterpret_cast<const T*>(dout), reinterpret_cast<const T*>(y2),
                       post_stats, post_coef, w, dx, part_bn, part_w, B, H, W, Ho, Wo, C, G * C, same_pad_before(H, S),
                       same_pad_before(W, S), g.fpb, g.nb, g.cchunk, view_aligned(dx, g.vec_bwd), G * g.nb, reload_y1);
    CDRL_LAUNCH_CHECK();
    return 0;
}

template <int S, int VEC>
static int launch_dwf_bwd_pre(const DwfGeom& g, hipStream_t st, const float* x, const float* pre_stats, const float* dout,
                              const float* y2, const float* post_stats, const float* post_coef, const float* w, View dx,
                              double* part_bn, double* part_w, int G, int B, int H, int W, int C, int at) {
    if (at) {
        if (pre_stats)
            return launch_dwf_bwd<S, VEC, true, bf16_t>(g, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C);
        return launch_dwf_bwd<S, VEC, false, bf16_t>(g, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C);
    }
    if (pre_stats)
        return launch_dwf_bwd<S, VEC, true, float>(g, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C);
    return launch_dwf_bwd<S, VEC, false, float>(g, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C);
}

int dwf_bwd(const float* x, const float* pre_stats, const float* dout, const float* y2, const float* post_stats,
            const float* post_coef, const float* w, View dx, double* part_bn, double* part_w, int G, int B, int H, int W,
            int C, int stride, hipStream_t st, int at) {
    if (stride != 1 && stride != 2) {
        set_error("dwf_bwd: stride must be 1 or 2");
        return -1;
    }
    if (pre_stats && !part_bn) {
        set_error("dwf_bwd: part_bn is required with a pre-BN prologue");
        return -1;
    }
    const DwfGeom g = dwf_geom(B, G, H, W, C, stride);
    // strip form (planned by dwf_geom: the partial-row count nb_bwd the caller finalizes with belongs to it)
    if (g.strip.ok) {
        const DwsGeom& d = g.strip;
        if ((int64_t)G * B * H * W * std::max(C, dx.ld) * 4 >= (int64_t)1 << 31) {
            set_error("dwf_bwd: gradient view too large for the strip kernels' 32-bit buffer offsets");
            return -1;
        }
        if (stride == 1) {
#define CDRL_DWS(SWV, RV)                                                                                                                                  \
    do {                                                                                                                                                   \
        if (pre_stats)                                                                                                                                     \
            return at ? launch_dws_bwd<SWV, RV, true, bf16_t>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C) \
                      : launch_dws_bwd<SWV, RV, true, float>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C); \
        return at ? launch_dws_bwd<SWV, RV, false, bf16_t>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C)   \
                  : launch_dws_bwd<SWV, RV, false, float>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C);  \
    } while (0)
            if (d.sw == 4) {
                if (d.R == 1) CDRL_DWS(4, 1);
                if (d.R == 2) CDRL_DWS(4, 2);
                CDRL_DWS(4, 3);
            }
            CDRL_DWS(8, 1);
#undef CDRL_DWS
        }
        const int pl = same_pad_before(W, 2);
#define CDRL_DWS2(PLV, PREV)                                                                                                                             \
    return at ? launch_dws2_bwd<PLV, PREV, bf16_t>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C) \
              : launch_dws2_bwd<PLV, PREV, float>(g, d, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C)
        if (pre_stats) {
            if (pl) CDRL_DWS2(1, true);
            CDRL_DWS2(0, true);
        }
        if (pl) CDRL_DWS2(1, false);
        CDRL_DWS2(0, false);
#undef CDRL_DWS2
    }
    if (g.lds_bwd > 150 * 1024) {
        set_error("dwf_bwd: frame %dx%d does not fit LDS even at %d channels", H, W, g.cchunk);
        return -1;
    }
#define CDRL_DWF_BWD(S, V) \
    return launch_dwf_bwd_pre<S, V>(g, st, x, pre_stats, dout, y2, post_stats, post_coef, w, dx, part_bn, part_w, G, B, H, W, C, at)
    if (stride == 1) {
        if (g.vec_bwd == 2) CDRL_DWF_BWD(1, 2);
        CDRL_DWF_BWD(1, 1);
    }
    if (g.vec_bwd == 2) CDRL_DWF_BWD(2, 2);
    CDRL_DWF_BWD(2, 1);
#undef CDRL_DWF_BWD
}

}  // namespace cdrl
