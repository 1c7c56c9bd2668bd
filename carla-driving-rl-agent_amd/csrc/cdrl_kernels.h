// Internal C++ launch API of libcdrl_hip.so.  Every function enqueues work on `st`, never
// allocates, never synchronises, and returns 0 on success (<0: see cdrl::last_error()).
#pragma once
#include "cdrl_common.h"

namespace cdrl {

// ---------------------------------------------------------------- BatchNorm family (bn.hip)
// Per-group column statistics of y (rows grouped contiguously: group g = rows [g*Mg,(g+1)*Mg)).
// part layout: [G][nb][2][C] (sum, sum of squares); nb from col_geom(Mg, C).
// at (here and below): element type of the ACTIVATION tensors the pointers / views refer to -- 0: float32, 1: bf16 (storage only:
// everything computed, reduced or kept as statistics stays float32 / double; configuration 3)
int colstats(View y, int G, int Mg, int C, double* part, hipStream_t st, int at = 0);
// Turns partials (training) or moving statistics (inference) into per-(group,channel) mean /
// invstd / scale / shift and applies the T sequential EMA updates.
int bn_finalize(const double* part, int nb, int G, int Mg, int C, const float* gamma, const float* beta,
                float* mov_mean, float* mov_var, int bessel, int training, float* stats /*[4][G][C]*/,
                hipStream_t st);
// One launch for the inference-mode statistics blocks of many BatchNorm layers (the training == 0 branch of bn_finalize, batched)
struct BnInfEntry {
    const float *gamma, *beta, *mov_mean, *mov_var;
    float* stats;           // [4][G][C]
    int G, C;
};
int bn_inference_stats_many(const BnInfEntry* tab_dev, int n, int max_c, hipStream_t st);
// dst = act(scale*y + shift); optional de-interleave shuffle on the destination channel index.
// stats == nullptr -> plain copy.
// pass_src / pass_dst: optional second tensor with the same C channels copied through the same shuffle store (the
// identity half of a ShuffleNet unit: concat + shuffle of both halves in one launch)
int bn_apply(View y, int G, int Mg, int C, const float* stats, int act, View dst, int shuffle_ctot,
             hipStream_t st, const View* pass_src = nullptr, const View* pass_dst = nullptr, int at = 0);
// single-group BatchNorm over few rows, one launch per direction (G = 1, no activation, no shuffle; training mode)
int bn_small_fwd(View x, int M, int C, const float* gamma, const float* beta, float* mov_mean, float* mov_var, float* stats,
                 View out, hipStream_t st);
int bn_small_bwd(View dout, View x, int M, int C, const float* stats, float* dgamma, float* dbeta, float* coef, float* dx,
                 hipStream_t st);
// Gradient source "through a 3x3/s2 SAME max-pool": d(a)[n,iy,ix,c] = sum of dp over the windows whose
// saved argmax points at (iy,ix).  Lets the stem BatchNorm backward read the pooled gradient directly
// (the 255 MB pre-pool gradient tensor is never written or re-read).
struct PoolSrc {
    const uint8_t* argmax;   // [N][Ho][Wo][C]
    const float* dp;         // [N][Ho][Wo][C]
    int H, W, Ho, Wo, pt, pl;
    const float* pa = nullptr;   // optional: the pooled ACTIVATED output [N][Ho][Wo][C] of the forward; lets
                                 // pool_bn_bwd_reduce skip the gather of the pre-pool values (see there)
};
// Backward: reduce (sum dz, sum dz*xhat) -> part [G][nb][2][C]
int bn_bwd_reduce(View da, int shuffle_ctot, View y, int G, int Mg, int C, const float* stats, int act,
                  double* part, hipStream_t st, const PoolSrc* pool = nullptr, const View* pass_gsrc = nullptr,
                  const View* pass_gdst = nullptr, int bcast_rows = 0, int at = 0);
// bcast_rows > 0 (here and in bn_bwd_apply): `da` holds ONE row per bcast_rows rows of y and is divided by bcast_rows on load --
// the gradient of a global average pool over bcast_rows pixels, never materialised (head of the tower)     // pass_*: gradient of the identity half gathered in the same pass
// Same sums for a BN+ReLU6 that feeds a 3x3/s2 max-pool, in scatter form over the POOLED gradient (ps.dp, ps.argmax);
// y: the BN's raw input [G*frames_per_group][ps.H][ps.W][C]; part [G][nb][2][C], nb = vcol_geom(frames*Ho*Wo, C).nb
int pool_bn_bwd_reduce(const PoolSrc& ps, const float* y, int G, int frames_per_group, int C, const float* stats, double* part,
                       hipStream_t st, int at = 0);
// dgamma/dbeta (+= over groups; `accumulate` keeps previous content) and coefficients coef[3][G][C]
int bn_bwd_finalize(const double* part, int nb, int G, int Mg, int C, const float* stats, float* dgamma,
                    float* dbeta, float* coef, hipStream_t st);
// dy = k1*(dz - k2 - xhat*k3) (dense [G*Mg][C]); also column sums of dy -> part2 [G][nb][C]
int bn_bwd_apply(View da, int shuffle_ctot, View y, int G, int Mg, int C, const float* stats,
                 const float* coef, int act, float* dy, double* part2, hipStream_t st, const PoolSrc* pool = nullptr,
                 int bcast_rows = 0, int at = 0);
// out[i] (+)= sum_p part[p*stride + i], i < n
int reduce_partials(const double* part, int nparts, int n, int64_t stride, float* out, int accumulate,
                    hipStream_t st);
// two outputs from one partial row: columns [0, n1) -> out1, [n1, n1+n2) -> out2 (filter + bias in one launch)
int reduce_partials2(const double* part, int nparts, int n1, int n2, int64_t stride, float* out1, float* out2, int accumulate,
                     hipStream_t st);
// column sums of a view: part [nb][C] with nb = col_geom(rows, C).nb
int colsum(View x, int rows, int C, double* part, hipStream_t st);
// dst(view) = src(view) gathered through the shuffle map on the *source* (backward of shuffle)
int gather_view(View src, int shuffle_ctot, int rows, int C, View dst, int accumulate, hipStream_t st);
// elementwise activation on dense [n] arrays
int act_fwd(const float* z, float* a, int64_t n, int act, hipStream_t st);
int act_bwd(const float* z, const float* da, float* dz, int64_t n, int act, hipStream_t st);
int fill(float* p, int64_t n, float v, hipStream_t st);
// (B,T,D) -> (T*B, D)
int permute_bt(const float* src, float* dst, int B, int T, int D, hipStream_t st);
int permute_tb_bwd(const float* src, float* dst, int B, int T, int D, hipStream_t st);
// dst[i,:] = src[idx[i],:]
int gather_rows(const float* src, const int* idx, float* dst, int nrows, int64_t row_elems, hipStream_t st);

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[M,N] (view) (+)= A[M,K] (view) * B(k,n) + bias[n];  B(k,n) = Bp[k*sbk + n*sbn].
// fp32 MFMA (v_mfma_f32_32x32x2_f32): results are bit-identical to a k-ordered fmaf chain.
// splitk_ws (>= gemm_nn_splitk_elems floats, or null): enables split-K for small-M / long-K products
int64_t gemm_nn_splitk_elems(int M, int N, int K);
int gemm_nn(View A, const float* Bp, int sbk, int sbn, const float* bias, View C, int M, int N, int K,
            int accumulate, hipStream_t st, float* splitk_ws = nullptr,
            float* act_out = nullptr, int act = 0, bool* act_done = nullptr);
// act_out / act / act_done: when the product goes through split-K, its reduce also writes act(C) to the dense [M][N] act_out and
// sets *act_done (the caller runs act_fwd otherwise)
// Cout[K,N] = sum_m A[m,K]^T * D[m,N]  (split over M; partial buffer `part` >= gemm_tn_part_elems)
// G > 1: rows are G equal BatchNorm groups; pro_stats ([4][G][K]) != null applies A <- scale[g][k]*A + shift[g][k] on load.
// dpro != null: D[m,n] <- k1*(dz - k2 - xhat*k3) on load (BatchNorm-backward apply; D views the gradient w.r.t. the BN
// output, optionally through the channel-shuffle gather / ReLU6 mask); needs gemm_tn_dpro_supported(N) and G groups.
struct TnBnBwd {
    const float* y;         // raw BN input [M][N] dense
    const float* stats;     // [4][G][N]
    const float* coef;      // [3][G][N]
    int shuffle_ctot;
    int act;
};
bool gemm_tn_dpro_supported(int N);
int64_t gemm_tn_part_elems(int M, int N, int K, int G = 1);
// bf16_operands: both operands rounded to bf16 AFTER their prologues, v_mfma_f32_32x32x16_bf16 over 16-row steps (configuration 3)
int gemm_tn(View A, View D, float* Cout, int M, int N, int K, float* part, int accumulate, hipStream_t st, int G = 1,
            const float* pro_stats = nullptr, const TnBnBwd* dpro = nullptr, bool bf16_operands = false, int at = 0);

// LDS-staged form for the bf16 modes (gemm_tn_lds.hip): same contract as gemm_tn with bf16_operands = true; at: 0 float32 / 1 bf16
// tensors.  gemm_tn dispatches to it (CDRL_TN_LDS=0 keeps the direct form).
bool gemm_tn_lds_supported(View A, View D, int N, int K, const TnBnBwd* dpro);
int64_t gemm_tn_lds_part_elems(int M, int N, int K, int G = 1);
// f32_mode (with at = 0): 1 = float32 operands on v_mfma_f32_32x32x2_f32 (the float32 engine's arithmetic); 2 = float32-accurate
// product from three bf16 planes per operand on v_mfma_f32_32x32x16_bf16 (six plane products)
int gemm_tn_lds(View A, View D, float* Cout, int M, int N, int K, float* part, int accumulate, hipStream_t st, int G,
                const float* pro_stats, const TnBnBwd* dpro, int at, int f32_mode = 0);

// ---------------------------------------------------------------- fused pointwise conv (gemm_pw.hip)
// Persistent skinny GEMM for K, N <= 128: C[m,n] (+)= sum_k pro(A[m,k]) W(k,n) + bias[n] over G groups of Mg rows.
//   pro_stats != null : A <- scale[g][k]*A + shift[g][k]  (BatchNorm apply of the previous layer)
//   epilogue 1        : part[g][b][2][N] = (sum c, sum c^2)            -> bn_finalize(part, nb = pw_nn_plan().nbpg)
//   epilogue 2        : part[g][b][2][N] = (sum c, sum c*xhat(ey))     -> bn_bwd_finalize(part, nb = ...nbpg)
struct PwPlan {
    int bm, tpb, nbpg;     // rows per tile, tiles per workgroup, workgroups (= partial rows) per group
};
// BatchNorm-backward prologue: A[m,k] = k1*(dz - k2 - xhat*k3), dz = (A view, gathered through the shuffle map when
// shuffle_ctot != 0) masked by ReLU6 of the BN output when act == ACT_RELU6, xhat from the BN's raw input y [M][K].
struct PwBnBwd {
    const float* y;
    const float* stats;     // [4][G][K]
    const float* coef;      // [3][G][K]
    int shuffle_ctot;
    int act;
    double* part2;          // optional [G][nbpg][K]: column sums of the transformed A (bias gradient partials)
};
bool pw_nn_supported(View A, int N, int K);
PwPlan pw_nn_plan(int G, int Mg, int N, int K);
// Wp (optional): B pre-packed in MFMA fragment order by pw_pack_many (pw_packed_elems(N, K) floats)
// wp_bf16: Wp holds bf16 fragments (PwPack::bf16) -> bf16-operand variant (both MFMA operands rounded to bf16, float32
// tensors / accumulation / statistics: the compute mode of configuration 3)
int pw_nn(View A, const float* pro_stats, const float* W, int sbk, int sbn, const float* bias, View C, int accumulate, int G,
          int Mg, int N, int K, int epilogue, const float* ey, const float* epi_stats, double* part, hipStream_t st,
          const PwBnBwd* bnbwd = nullptr, const float* Wp = nullptr, bool wp_bf16 = false, int at = 0);
struct PwPack {             // one operand to pack: B(k, n) = w[k * sbk + n * sbn], K x N
    const float* w;
    float* wp;
    int K, N, sbk, sbn, ksm, ntiles;
    int bf16;               // 1: bf16 fragments of the bf16-operand variant (same element count, half the bytes)
};
int64_t pw_packed_elems(int N, int K);
PwPack pw_pack_entry(const float* w, float* wp, int K, int N, int sbk, int sbn, bool bf16 = false);
int pw_pack_many(const PwPack* tab_dev, int n, hipStream_t st);
// out[i] (+)= sum_p part[p*stride + i] for float partials (double accumulation, fixed order)
int reduce_partials_f32(const float* part, int nparts, int64_t n, int64_t stride, float* out, int accumulate,
                        hipStream_t st);

// ---------------------------------------------------------------- convolutions (conv.hip)
// x: (B,T,H,W,3) reference layout -> y: [(t*B+b)][Ho][Wo][Cout], 3x3 stride 2 valid.
int stem_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int H, int W, int Cout,
             hipStream_t st);
// stem conv + (sum, sum^2) partials of its output per time slice: part [T][nb][2][Cout], nb = stem_fwd_stats_nb()
bool stem_fwd_stats_supported(int Cout);
int stem_fwd_stats_nb(int B, int T, int H, int W, int Cout);
int stem_fwd_stats(const float* x, const float* w, const float* bias, float* y, double* part, int B, int T, int H, int W, int Cout,
                   hipStream_t st, int at = 0);
int64_t stem_bwd_part_elems(int B, int T, int H, int W, int Cout);
// stem filter gradient with the stem BatchNorm's backward apply + max-pool gather fused into the operand load (dy is
// never materialised); y: raw stem conv output, stats/coef: the stem BN's [4|3][T][Cout] blocks
bool stem_bwd_fused_supported(int Cout);
int stem_bwd_filter_fused(const float* x, const PoolSrc& ps, const float* y, const float* stats, const float* coef, float* dw,
                          float* db, int B, int T, int H, int W, int Cout, double* part, hipStream_t st, int at = 0);
int stem_bwd_filter(const float* x, const float* dy, float* dw, float* db, int B, int T, int H, int W, int Cout,
                    double* part, hipStream_t st);
// depthwise 3x3, TF 'SAME' (asymmetric) padding, stride 1|2.  N = frames.
int dw_fwd(View a, const float* w, const float* bias, float* y, int N, int H, int W, int C, int stride,
           hipStream_t st);
int dw_bwd_data(const float* dy, const float* w, View da, int N, int H, int W, int C, int stride, int accumulate,
                hipStream_t st);
// dw_bwd_data fused with the BatchNorm-backward reduction of the layer that produced the depthwise input
// (a = act(bn(y_pre))): writes da AND the per-group partial sums (sum dz, sum dz*xhat) of that BN.
int dw_bwd_data_bnreduce(const float* dy, const float* w, View da, int N, int H, int W, int C, int stride, int G,
                         const float* y_pre, const float* stats_pre, int act_pre, double* part, hipStream_t st);
int64_t dw_bwd_part_elems(int N, int H, int W, int C, int stride);
int dw_bwd_filter(View a, const float* dy, float* dw, float* db, int N, int H, int W, int C, int stride,
                  double* part, hipStream_t st);
// ---- fused depthwise block (dwfused.hip): whole frames staged in LDS.  G groups x B frames per group.
// strip form of the backward (round 5): thread = channel pair x row strip of `sw` pixels; ok == false -> the pixel-mapped kernel runs
struct DwsGeom {
    bool ok;
    int sw, S, R, F, cchunk, nch, cx, cy, wdp, tile_floats;    // S strips per row, R strips per thread and tile batch (stride 1), F frames
    size_t lds;                                                // per tile batch, wdp padded tile width (stride 2)
};
struct DwfGeom {
    int vec, nch, cchunk, cx, cy, fpb, nb;     // nb: partial blocks per group (B / frames-per-block) of the FORWARD kernel
    int vec_bwd, cx_bwd, cy_bwd;               // backward lane shape (fewer channels per thread: register budget)
    size_t lds_fwd, lds_bwd;
    int fpb_bwd, nb_bwd;                       // frames per workgroup / partial rows per group of the BACKWARD kernel
    DwsGeom strip;
};
DwfGeom dwf_geom(int B, int G, int H, int W, int C, int stride);
int64_t dwf_stats_part_elems(int B, int G, int H, int W, int C, int stride);      // doubles: [G][nb][2][C]
int64_t dwf_filter_part_elems(int B, int G, int H, int W, int C, int stride);     // doubles: [G*nb][10][C]
// y = dw3x3(pre_stats ? relu6(scale*x+shift) : x) + bias and the per-block (sum y, sum y^2) partials of the
// BatchNorm that follows (layout of bn_finalize with nb = dwf_geom().nb).
int dwf_fwd(const float* x, const float* pre_stats, const float* w, const float* bias, float* y, double* part, int G,
            int B, int H, int W, int C, int stride, hipStream_t st, int at = 0);
// dout: gradient w.r.t. the output of the BatchNorm (no activation) that follows the depthwise conv; y2: the raw
// depthwise output; post_stats / post_coef: that BN's statistics and backward coefficients (k1,k2,k3).
// Writes the filter / bias partials (part_w: reduce with reduce_partials over G*nb parts, row stride 10*C) and
//   pre_stats != null: dx = ReLU6-masked gradient at the pre-BN's output + its (sum dz, sum dz*xhat) partials
//   pre_stats == null: dx = gradient w.r.t. x.
int dwf_bwd(const float* x, const float* pre_stats, const float* dout, const float* y2, const float* post_stats,
            const float* post_coef, const float* w, View dx, double* part_bn, double* part_w, int G, int B, int H, int W,
            int C, int stride, hipStream_t st, int at = 0);
int maxpool_fwd(const float* a, float* p, uint8_t* argmax, int N, int H, int W, int C, hipStream_t st);
int maxpool_bwd(const uint8_t* argmax, const float* dp, float* da, int N, int H, int W, int C, hipStream_t st);
// fused BN-apply + ReLU6 + max-pool on the raw conv output y (frames of group g = [g*frames_per_group, ...))
int maxpool_bn_fwd(const float* y, const float* stats, int G, int frames_per_group, float* p, uint8_t* argmax, int N,
                   int H, int W, int C, hipStream_t st, int at = 0);
PoolSrc make_pool_src(const uint8_t* argmax, const float* dp, int H, int W);
// mean over the P pixels of each frame: a [N][P][C] -> out [N][C]
int gap_fwd(const float* a, float* out, int N, int P, int C, hipStream_t st);
// out[n][c] = mean_p act(scale[g][c] * y[n*P + p][c] + shift[g][c]): BatchNorm apply + activation + global average pool fused
int bn_act_gap_fwd(const float* y, const float* stats, float* out, int G, int frames_per_group, int P, int C, int act, hipStream_t st,
                   int at = 0);
int gap_bwd(const float* dout, float* da, int N, int P, int C, hipStream_t st);

// batched transposes W[cin][cout] -> WT[cout][cin] of many small matrices in one launch (gemm.hip)
struct PwTranspose {
    const float* w;
    float* wt;
    int cin, cout;
};
// one launch: grid = (tiles, n) with tiles >= max over the entries of ceil(cin/32) * ceil(cout/32)
int transpose_many(const PwTranspose* tab_dev, int n, int tiles, hipStream_t st);

// ---------------------------------------------------------------- float32 pointwise conv on the bf16 matrix pipe (gemm_pw_x3.hip)
// exact 3-way bf16 split of both operands, six bf16 MFMAs per K = 16 step, float32 in / out; W packed by pw_x3_pack_many
struct PwX3Pack {
    const float* w;
    __bf16* wp;
    int K, N, sbk, sbn, kp;
};
bool pw_x3_supported(View A, int N, int K);
int pw_x3_partial_rows(int G, int Mg, int N, int K);
int64_t pw_x3_packed_bytes(int K);                 // N <= 128 (one column block)
int pw_x3_ksteps(int K);                           // K = 16 steps per plane of that pack
int64_t pw_x3_packed_bytes_n(int K, int N);        // any N <= 256: column blocks of 128
PwX3Pack pw_x3_pack_entry(const float* w, void* wp, int K, int N, int sbk, int sbn);
int pw_x3_pack_many(const PwX3Pack* tab_dev, int n, hipStream_t st);
// nbpg > 0: workgroups (= partial rows) per group chosen by the caller (the engine keeps pw_nn_plan's count)
// backward-data of the wide convs (128 < K or N <= 256) with the BatchNorm-backward prologue, one 32-row tile per workgroup: C[M][N] (+)=
// dy W^T, Wp = pw_x3_pack_entry(W, wp, K, N, 1, K) (B(k = conv output channel, n = conv input channel)); one part2 / part row per tile
bool pw_x3_wide_bwd_supported(View dz, View C, int N, int K, int shuffle_ctot);
int pw_x3_wide_bwd_rows(int Mg);
int pw_x3_wide_bwd(View dz, const PwBnBwd& bb, const void* Wp, View C, int accumulate, int G, int Mg, int N, int K, const float* ey,
                   const float* epi_stats, double* part, hipStream_t st);
int pw_x3(View A, const float* pro_stats, const void* Wp, const float* bias, View C, int G, int Mg, int N, int K, double* part,
          hipStream_t st, int nbpg = 0);


// ---------------------------------------------------------------- fused backward of a pointwise conv (gemm_pw_bwd.hip)
// ONE pass over (dz, y, a): BatchNorm-backward apply on load, da = dy W^T, Q = a^T dy (or xhat(a)^T dy), db partials; then
// pw_bwd_fused_reduce: dW, db and -- when a is a BatchNorm output applied on load (a_stats) -- that BatchNorm's backward sums
// (dgamma, dbeta, coefficients) from Q, with no pass over da.  Float32 tensors, float32-accurate (three-way bf16 operand split).
struct PwBwdFused {
    View dz;                // gradient w.r.t. the output of the BatchNorm behind the conv
    int dz_shuffle = 0;     // channel-shuffle gather on the dz columns (ctot), 0 = none
    int act = 0;            // ACT_RELU6: mask from that BatchNorm's output
    const float* y = nullptr;       // raw conv output [G*Mg][N] dense
    const float* stats = nullptr;   // [4][G][N] of that BatchNorm
    const float* coef = nullptr;    // [3][G][N] backward coefficients (bn_bwd_finalize)
    View a;                 // conv input [G*Mg][K]
    const float* a_stats = nullptr; // [4][G][K]: a = BatchNorm(a_raw) applied on load (no activation); null: plain input
    const float* a_gamma = nullptr; // with a_stats: parameters / gradient outputs of that BatchNorm
    const float* a_beta = nullptr;
    float* a_dgamma = nullptr;
    float* a_dbeta = nullptr;
    float* a_coef = nullptr;        // [3][G][K]
    const float* W = nullptr;       // conv weights [K][N]
    const void* Wp = nullptr;       // pw_x3 packing of B(k = n_out, n = k_in) = W[n * N + k]  (pw_x3_pack_entry(W, wp, N, K, 1, N))
    View da;                // gradient w.r.t. the conv input (the raw a for a_stats: gradient w.r.t. the BatchNorm OUTPUT)
    int accumulate = 0;
    float* dW = nullptr;    // [K][N]
    float* db = nullptr;    // [N]
    float* qpart = nullptr; // pw_bwd_fused_qpart_elems floats
    double* dbpart = nullptr;       // pw_bwd_fused_dbpart_elems doubles
    int N = 0, K = 0, G = 1, Mg = 0;
    int at = 0;             // 1: dz, y, a, da are bf16 in HBM (bf16 activation storage; one bf16 plane per MFMA operand)
    // float32 form, optional: the finalize of the BatchNorm behind the conv on load (no bn_bwd_finalize launch in front, `coef` unused):
    // fin_part = its [G][fin_nb][2][N] backward sums, fin_tot = [G][2][N] doubles of scratch, o_dgamma / o_dbeta [N] written by the reduce
    const double* fin_part = nullptr;
    double* fin_tot = nullptr;
    int fin_nb = 0;
    float* o_dgamma = nullptr;
    float* o_dbeta = nullptr;
};
bool pw_bwd_fused_supported(View dz, View a, View da, int N, int K, int at = 0);
int pw_bwd_fused_nbpg(int G, int Mg, int N, int K, int at = 0);
int64_t pw_bwd_fused_qpart_elems(int G, int Mg, int N, int K, int at = 0);
int64_t pw_bwd_fused_dbpart_elems(int G, int Mg, int N, int K, int at = 0);
int pw_bwd_fused(const PwBwdFused& f, hipStream_t st);
int pw_bwd_fused_reduce(const PwBwdFused& f, hipStream_t st);

// general form (gemm_x3.hip): any K (multiple of 4), any N; B packed once per pass by gemm_x3_pack_many into [3][K/16][2][Npad][8]
struct GemmX3Pack {
    const float* w;
    __bf16* wp;
    int K, N, sbk, sbn, KS, NP;
};
bool gemm_x3_supported(View A, int K);
int64_t gemm_x3_packed_bytes(int N, int K);
GemmX3Pack gemm_x3_pack_entry(const float* w, void* wp, int K, int N, int sbk, int sbn);
int gemm_x3_pack_many(const GemmX3Pack* tab_dev, int n, hipStream_t st);
// every pack table of a training pass + the W^T transposes of its backward in ONE launch (gemm_pw.hip, pack_bodies.h)
int pack_all(const GemmX3Pack* tg, int ng, const PwPack* tp, int np, const PwX3Pack* t3, int n3, const PwTranspose* tt, int nt,
             int transpose_tiles, hipStream_t st);
// bf16_operands: one product per step on the first plane only (operands rounded to bf16: configuration 3's compute mode)
int gemm_x3(View A, const void* Bp, const float* bias, View C, int M, int N, int K, int accumulate, hipStream_t st,
            bool bf16_operands = false, int at = 0);

// ---------------------------------------------------------------- bf16 pointwise conv (gemm_pw_bf16.hip)
// bf16 activations (A, C), float32 master weights / bias / BatchNorm blocks, bf16 MFMA with float32 accumulate; optional
// BN-apply prologue (pro_stats [4][G][K]) and statistics epilogue (part [G][nbpg][2][N], nbpg = pw_bf16_partial_rows)
bool pw_bf16_supported(int lda, int a_coff, int N, int K);
int pw_bf16_partial_rows(int G, int Mg, int N, int K);
// Wp (optional): the weights pre-packed as bf16 MFMA fragments by pw_bf16_pack (pw_bf16_packed_elems(K) bf16 elements)
int pw_bf16(const void* A, int lda, int a_coff, const float* pro_stats, const float* W, const float* bias, void* C, int ldc,
            int c_coff, int G, int Mg, int N, int K, double* part, hipStream_t st, const void* Wp = nullptr);
int64_t pw_bf16_packed_elems(int K);
int pw_bf16_pack(const float* W, int K, int N, void* Wp, hipStream_t st);
int f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t st);
int bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t st);

// ---------------------------------------------------------------- linear output heads (heads.hip)
#define HEADS_MAX 4
#define HEADS_MAX_OUT 8
struct HeadSet {
    int nheads;
    int n[HEADS_MAX], off[HEADS_MAX];      // outputs of head h and its first column in lin
    const float* w[HEADS_MAX];             // [K][n]
    const float* b[HEADS_MAX];             // [n]
    float* gw[HEADS_MAX];                  // gradients (null for inference-only heads)
    float* gb[HEADS_MAX];
};
// lin[b][off_h + n] = b_h[n] + sum_k a[b][k] W_h[k][n]   (a: [B][lda], lin: [B][ldl])
int heads_fwd(const float* a, int lda, const HeadSet& hs, float* lin, int ldl, int B, int K, hipStream_t st);
// da[b][k] = sum_c dlin[b][c] W[k][c] (overwrite; da may be null), dW_h = a^T dlin_h, db_h = column sums of dlin_h
int heads_bwd(const float* a, int lda, const HeadSet& hs, const float* dlin, int ldl, float* da, int ldda, int B, int K,
              hipStream_t st);

// ---------------------------------------------------------------- GRU (rnn.hip)
// gates of step t: xp,hp [B][3u]; hprev [B][u]; saves z,r,hh; writes hnew
// one fused kernel per GRU time step and direction (rnn.hip): recurrent product + gate math / gate derivatives + product
// against R^T; `out` (optional) is a second destination of h_new, `dh` a strided view, dhprev == null for the first step
bool gru_step_supported(int u);
int gru_step_fwd(const float* xp, const float* hprev, const float* R, const float* b1, float* z, float* r, float* hh, float* hp,
                 float* hnew, View out, int B, int u, hipStream_t st);
int gru_step_bwd(View dh, const float* z, const float* r, const float* hh, const float* hp, const float* hprev, const float* RT,
                 float* dxp, float* dhp, float* dhprev, int B, int u, hipStream_t st);

// ---------------------------------------------------------------- losses (loss.hip)
struct PolicyLossArgs {
    const float* lin;        // [B][2A+2] linear head outputs: alpha_pre(A) beta_pre(A) similarity_pre speed_pre
    const float* adv;        // [B]
    const float* old_logp;   // [B][A]
    const float* speed;      // [B]
    const float* similarity; // [B]
    const float* u;          // [B][A] Beta samples (or stored actions)
    const float* du_da;      // [B][A] or nullptr
    const float* du_db;      // [B][A] or nullptr
    const float* hp;         // device hyper-parameter block (DevHP)
    float* dlin;             // [B][2A+2]
    float* metrics;          // [16]
    float* aux;              // optional [B][4A]: alpha, beta, log_prob, entropy
    int B, A;
    float inv_world;         // gradient scale for data-parallel averaging (1/world_size)
};
int policy_loss(const PolicyLossArgs& a, hipStream_t st);
struct ValueLossArgs {
    const float* lin;        // [B][4]: base_pre, exp_pre, speed_pre, similarity_pre
    const float* returns;    // [B][2]
    const float* speed;
    const float* similarity;
    float* dlin;             // [B][4]
    float* metrics;          // [16]
    float* values;           // optional [B][2]
    int B;
    float exp_scale;
    float inv_world;
};
int value_loss(const ValueLossArgs& a, hipStream_t st);
// inference heads: alpha,beta,(mean,std) and value(base,exp) from linear outputs
int policy_dist(const float* lin, float* out /*[B][4A]: alpha,beta,mean,std*/, int B, int A, hipStream_t st);
int value_act(const float* lin, float* out /*[B][4] base,exp,speed,sim*/, int B, float exp_scale, hipStream_t st);

// ---------------------------------------------------------------- Beta sampling (sample.hip)
// u ~ Beta(alpha, beta) (as a Gamma ratio) with pathwise derivatives du/dalpha, du/dbeta (implicit
// reparameterisation of the two Gamma samples).  Element (row, col): alpha[row*ld + col].
int beta_sample(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset, float* u,
                float* du_da, float* du_db, hipStream_t st, float* logp = nullptr, double* gammas = nullptr);
int gamma_implicit_grad(const double* a, const double* g, int n, double* out, hipStream_t st);
int philox_words(uint64_t seed, uint64_t offset, uint64_t idx0, int n, int nblocks, uint32_t* out, hipStream_t st);

// ---------------------------------------------------------------- rollout-time augmentation (augment.hip)
struct AugPlan {           // same layout as cdrl_aug_plan (include/cdrl.h)
    int jitter;
    float brightness, contrast, saturation, hue;
    int blur_size;
    float blur_kernel[75];
    int salt_pepper;
    float sp_amount, sp_prob;
    int gauss_noise;
    float gn_amount, gn_std;
    int normalize;
    int cutout_size, cutout_cell;
    int dropout_size;
    float dropout_amount;
    uint64_t seed, offset;
};
// in/out: [T][H][W][3] floats; workspace: 2*T*H*W*3 + 5*T floats
int augment_images(const float* in, float* out, int T, int H, int W, const AugPlan& plan, float* workspace, hipStream_t st);

// ---------------------------------------------------------------- optimiser (optim.hip)
// Device hyper-parameter block, refreshed by the host before each step (graph-replay safe).
struct DevHP {
    float lr_policy, lr_value, lr_dynamics;
    float clip_ratio, entropy_coef;
    float clip_norm_policy, clip_norm_value;   // <= 0 : no clipping
    float beta1, beta2, eps;
    int t_policy, t_value, t_dynamics;         // Adam step counters (incremented on device)
    int pad[3];
};
struct TensorSeg {      // one parameter tensor inside a flat arena
    int64_t off;
    int64_t n;
    int first_chunk;    // index of its first 1024-element chunk in the chunk table
    int nchunks;
};
// sq-norm per tensor -> norms[ntensors]; deterministic two-stage reduction
int tensor_sqnorms(const float* g, const TensorSeg* segs_dev, int ntensors, const int* chunk_tensor_dev,
                   const int64_t* chunk_off_dev, int nchunks, double* chunk_part, float* sqnorms, hipStream_t st,
                   DevHP* tick_hp = nullptr, int tick_mask = 0, bool fold_final = false);
// tick_hp / tick_mask (1 policy, 2 value, 4 trunk): the chunk kernel also advances those Adam step counters; fold_final: no second
// stage -- the consumer folds the chunk partials (clip_adam with chunk_part)
int clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const int* chunk_tensor_dev /*or null*/,
              const int64_t* chunk_off_dev, int nchunks, const TensorSeg* segs_dev, const float* sqnorms /*or null*/,
              DevHP* hp, int which, hipStream_t st, const double* chunk_part = nullptr, int ticked = 0);
// chunk_part: per-tensor norms folded here from tensor_sqnorms' chunk partials; ticked: the step counter was advanced already
int adam_tick(DevHP* hp, int which, hipStream_t st);
int copy_two(float* d0, const float* s0, int64_t n0, float* d1, const float* s1, int64_t n1, hipStream_t st);

// ---------------------------------------------------------------- GAE (gae.hip)
// rewards [N+1] (bootstrap appended), values_be [N+1][2] -> returns_be [N][2], returns [N], adv_raw [N], adv [N]
int gae_returns(const float* rewards, const float* values_be, int N, double gamma, double lambda, float scale,
                float* returns, float* returns_be, float* adv_raw, float* adv, double* scratch, hipStream_t st);

}  // namespace cdrl
