/* libcdrl_hip.so -- C ABI of the MI355X-native PPO learner hot path.
 *
 * Drop-in boundary for the learner path of Luca96/carla-driving-rl-agent.  The reference is pure
 * Python/TensorFlow and has no FFI of its own; each entry point below names the
 * reference interface (file:line under the reference repo) whose work it replaces.  Python
 * (ctypes) is the only caller today: carla-driving-rl-agent_amd/_lib.py, see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; cdrl_last_error() gives the message;
 *   - all pointers are DEVICE pointers to float32 unless stated; `stream` is a hipStream_t;
 *   - nothing allocates device memory: the caller passes arenas + a workspace sized by
 *     cdrl_learner_workspace_bytes(); entry points only enqueue work (no host sync);
 *   - observation tensors use the reference layout (B, T, H, W, 3) / (B, T, D), NHWC dense.
 */
#ifndef CDRL_H
#define CDRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDRL_VERSION 1

typedef struct cdrl_learner cdrl_learner;

/* Network / batch geometry.  Defaults = CARLAgent.DEFAULT_* (core/carla_agent.py:61-68) on the
 * real CARLAEnv spaces (core/carla_env.py:18-24,64). */
typedef struct cdrl_config {
    int32_t B, T, H, W;                    /* minibatch, time_horizon, image height / width     */
    int32_t road, vehicle, navigation, A;  /* feature-vector sizes, num_actions                 */
    int32_t stem;                          /* 24                                                */
    int32_t stage_c[3];                    /* 116, 232, 464 (g = 1.0)                           */
    int32_t stage_n[3];                    /* 4, 8, 4                                           */
    int32_t last;                          /* 768                                               */
    int32_t feat, rnn_image, rnn_small, dyn, head; /* 16, 256, 32, 512, 320                     */
    float exp_scale;                       /* 6.0 (core/networks.py:169)                        */
    int32_t compute;                       /* CDRL_COMPUTE_*: arithmetic of the tower's 1x1 convolutions (default float32) */
} cdrl_config;

/* CDRL_COMPUTE_BF16_OPERANDS (BASELINE.json configs[2]): the 1x1 convolutions of the image tower (core/architectures.py:130,140,
 * 170) multiply bf16-rounded operands on v_mfma_f32_32x32x16_bf16 -- forward, backward-data and filter gradient; float32
 * accumulation, float32 tensors in HBM, float32 BatchNorm statistics, bias gradients, optimizer and master weights.
 * CDRL_COMPUTE_BF16_STORAGE (configs[2] in full): the same arithmetic AND bf16 activation storage -- every activation and
 * activation-gradient tensor of the image tower (raw conv outputs, unit outputs, their gradients, the scratch gradients) lives
 * in HBM as bf16 (round-to-nearest-even on store); BatchNorm statistics are those of the stored values; LDS tiles, accumulators,
 * statistics / partial sums (double), coefficients, weights, gradients of weights and everything behind the global average
 * pool stay float32.  Half the activation traffic of the float32 path. */
enum { CDRL_COMPUTE_F32 = 0, CDRL_COMPUTE_BF16_OPERANDS = 1, CDRL_COMPUTE_BF16_STORAGE = 2 };

enum { CDRL_TRUNK = 0, CDRL_POLICY = 1, CDRL_VALUE = 2, CDRL_OLD_POLICY = 3 };

typedef struct cdrl_param_info {
    char name[64];
    int32_t shape[4];
    int32_t ndim;
    int32_t trainable;
    int64_t numel;
    int64_t offset;      /* element offset inside the model's trainable / state region */
} cdrl_param_info;

/* DynamicParameter values of the step (rl/parameters/parameters.py; rl/agents/ppo.py:42,55,61,
 * 101-106; core/carla_agent.py:120-124) + Keras Adam constants. */
typedef struct cdrl_hparams {
    float policy_lr, value_lr, dynamics_lr;
    float clip_ratio, entropy_coef;
    float clip_norm_policy, clip_norm_value;   /* <= 0 disables tf.clip_by_norm */
    float beta1, beta2, eps;
} cdrl_hparams;

/* One policy minibatch = what CARLAgent.policy_batch_tensors yields (core/carla_agent.py:323-331)
 * plus the Beta samples of PolicyNetwork.call (core/networks.py:96-110). */
typedef struct cdrl_policy_batch {
    const float *image, *road, *vehicle, *navigation;
    const float *advantages;      /* (B)    */
    const float *old_log_prob;    /* (B, A) */
    const float *speed;           /* (B)  info_buffer['speed'] / 100 */
    const float *similarity;      /* (B)    */
    const float *u;               /* (B, A) Beta sample (or stored action) */
    const float *du_dalpha;       /* (B, A) pathwise Jacobian or NULL */
    const float *du_dbeta;        /* (B, A) or NULL */
} cdrl_policy_batch;

/* One value minibatch = CARLAgent.value_batch_tensors (core/carla_agent.py:333-349). */
typedef struct cdrl_value_batch {
    const float *image, *road, *vehicle, *navigation;
    const float *returns;         /* (B, 2) (base, exponent) */
    const float *speed, *similarity;
} cdrl_value_batch;

const char* cdrl_last_error(void);
int cdrl_version(void);
/* Environment switches (no reference counterpart: the reference has no native code).  The library reads its CDRL_* tuning
 * switches from the environment; CDRL_DIAG_* switches skip work or synchronisation (WRONG RESULTS, timing diagnostics only) and
 * are honoured only together with the master switch CDRL_DIAG=1.
 * cdrl_env_overrides: writes "NAME=VALUE NAME=VALUE ..." of every CDRL_* variable of the process environment into buf (truncated
 * to cap bytes, NUL-terminated) and returns their count.  cdrl_diag_active: number of wrong-result switches in effect (0 unless
 * CDRL_DIAG=1): benchmarks must refuse to report when it is non-zero. */
int cdrl_env_overrides(char* buf, int cap);
int cdrl_diag_active(void);
/* CRC-32C (Castagnoli) of `n` host bytes continued from `crc` (0 to start): block and tensor checksums of the
 * TensorFlow checkpoint-V2 files CARLANetwork.save_weights writes (reference core/networks.py:297-300). */
uint32_t cdrl_crc32c(uint32_t crc, const void* data, size_t n);

/* ---- learner object ------------------------------------------------------------------------
 * replaces CARLANetwork.__init__ / dynamics_model / value_network / PolicyNetwork construction
 * (core/networks.py:150-176,223-253) for one minibatch size. */
int cdrl_learner_create(const cdrl_config* cfg, cdrl_learner** out);
void cdrl_learner_destroy(cdrl_learner* l);
void cdrl_config_default(cdrl_config* cfg);

/* variable inventory: Model.trainable_variables / get_weights ordering (core/networks.py:281-285) */
int cdrl_learner_param_count(const cdrl_learner* l, int model);
int cdrl_learner_param_info(const cdrl_learner* l, int model, int index, cdrl_param_info* out);
int64_t cdrl_learner_region_offset(const cdrl_learner* l, int model, int trainable);
int64_t cdrl_learner_region_elems(const cdrl_learner* l, int model, int trainable);
int64_t cdrl_learner_params_total(const cdrl_learner* l);
int64_t cdrl_learner_grads_total(const cdrl_learner* l);
size_t cdrl_learner_workspace_bytes(const cdrl_learner* l);

/* params: [policy_tr | trunk_tr | value_tr | policy_state | trunk_state | value_state | old_policy]
 * grads / adam_m / adam_v: [policy_tr | trunk_tr | value_tr] (one contiguous all-reduce buffer) */
int cdrl_learner_bind(cdrl_learner* l, float* params, float* grads, float* adam_m, float* adam_v, void* workspace,
                      size_t workspace_bytes);
int cdrl_learner_set_hparams(cdrl_learner* l, const cdrl_hparams* hp, void* stream);
/* Makes `l` read its hyper-parameters and Adam step counters from `owner`'s device block (both bound, sharing the same
 * parameter / Adam arenas): a second learner built for the ragged LAST minibatch of an update (the reference's tf.data
 * pipeline keeps it unless drop_remainder is set, rl/utils.py:365-393) then advances the same optimizer. */
int cdrl_learner_share_hparams(cdrl_learner* l, const cdrl_learner* owner);
/* Data-parallel overlap: `stream` (caller-owned, or NULL to switch off) is made to wait inside every
 * following *_forward_backward call for the point of the backward pass at which the gradients of the head and of the trunk
 * tail (every trunk tensor behind the image tower) are final.  A collective enqueued on that stream after the call has
 * returned runs concurrently with the tower's backward; the tower's gradients are final when the call's own stream is. */
int cdrl_learner_set_comm_stream(cdrl_learner* l, void* stream);
/* Element offset, inside the trunk's trainable region, of the first TAIL tensor: gradients [tail_offset, trunk size) plus the
 * head's are final when the communication stream is released, gradients [0, tail_offset) (the image tower) only when the
 * pass's own stream is.  Equals the trunk size when the stream is never released mid-pass (hipGraph replay, CDRL_GRAPH=1). */
int64_t cdrl_learner_tail_offset(const cdrl_learner* l);
int cdrl_learner_reset_optimizer_steps(cdrl_learner* l, void* stream);

/* CARLAgent.get_policy_gradients (core/carla_agent.py:351-373): train-mode trunk forward,
 * policy_objective (:394-428), gradients w.r.t. policy and trunk variables.  grad_scale = 1 /
 * world_size for data-parallel averaging. */
int cdrl_learner_policy_forward_backward(cdrl_learner* l, const cdrl_policy_batch* b, float grad_scale, void* stream);
/* The same step split at the Beta sample: PolicyNetwork.call re-samples an action
 * from the NEW Beta(alpha, beta) (core/networks.py:96-110).  policy_forward runs the train-mode
 * forward and leaves (alpha, beta, mean, std) in CDRL_BUF_AUX_P; the caller draws u and its
 * pathwise Jacobians; policy_backward evaluates the loss on them and back-propagates. */
int cdrl_learner_policy_forward(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                                const float* navigation, void* stream);
int cdrl_learner_policy_backward(cdrl_learner* l, const cdrl_policy_batch* b, float grad_scale, void* stream);
/* Fully on-device form of the re-sampling step: forward, u ~ Beta(alpha, beta) of the new policy drawn
 * by cdrl_beta_sample with pathwise Jacobians (what TFP's reparameterised Beta provides to
 * PolicyNetwork.call, core/networks.py:96-110,136-137), backward.  b->u / du_* are ignored.
 * (seed, offset) select the Philox stream; the drawn sample is left in CDRL_BUF_SAMPLE. */
int cdrl_learner_policy_forward_backward_resample(cdrl_learner* l, const cdrl_policy_batch* b, uint64_t seed,
                                                 uint64_t offset, float grad_scale, void* stream);
/* Brackets a SEQUENCE of learner calls enqueued from one stream with nothing of the caller's own between them -- the four calls of
 * one minibatch of PPOAgent.update on one GPU (rl/agents/ppo.py:190-226: policy gradients, apply, value gradients, apply).  Each
 * learner call orders the engine's streams behind `stream` on entry and `stream` behind them on return; inside a sequence that
 * happens once, in begin and in end.  Between the two, work the caller enqueues on `stream` is NOT ordered against the learner calls
 * (collectives between a pass and its apply: do not bracket them).  Calls on other streams are unaffected.  No nesting. */
int cdrl_learner_sequence_begin(cdrl_learner* l, void* stream);
int cdrl_learner_sequence_end(cdrl_learner* l, void* stream);
/* CARLAgent.apply_policy_gradients (core/carla_agent.py:375-388) + PPOAgent.apply_policy_gradients
 * (rl/agents/ppo.py:238-252): trunk Adam, per-tensor clip, old_policy <- policy, policy Adam. */
int cdrl_learner_policy_apply(cdrl_learner* l, void* stream);
/* CARLAgent.get_value_gradients / apply_value_gradients (core/carla_agent.py:430-463;
 * rl/agents/ppo.py:264-275). */
int cdrl_learner_value_forward_backward(cdrl_learner* l, const cdrl_value_batch* b, float grad_scale, void* stream);
int cdrl_learner_value_apply(cdrl_learner* l, void* stream);
/* CARLANetwork.update_old_policy (core/networks.py:281-285). */
int cdrl_learner_update_old_policy(cdrl_learner* l, void* stream);
/* CARLANetwork.predict deterministic part (core/networks.py:181-193): inference-mode trunk,
 * old_policy -> dist (B,4A: alpha, beta, mean, std), value heads -> (B,4: base, exp, speed, sim). */
int cdrl_learner_predict(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                         const float* navigation, float* dist_out, float* value_out, float* dynamics_out, void* stream);
/* CARLANetwork.dynamics_predict_train (core/networks.py:210-212). */
int cdrl_learner_trunk_forward_train(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                                     const float* navigation, void* stream);

enum {
    CDRL_BUF_DYNAMICS = 0,   /* (B, dyn)  trunk output                       */
    CDRL_BUF_IMG_FEAT = 1,   /* (T*B, last) tower output, frame = t*B + b    */
    CDRL_BUF_METRICS_P = 2,  /* 16 floats: total, policy, entropy, speed, sim, ratio, log_prob */
    CDRL_BUF_METRICS_V = 3,  /* 16 floats: total, value, speed, sim           */
    CDRL_BUF_AUX_P = 4,      /* (B, 4A) alpha, beta, log_prob, entropy        */
    CDRL_BUF_AUX_V = 5,      /* (B, 2) value (base, exp)                      */
    CDRL_BUF_LIN_P = 6,      /* (B, 2A+2) linear head outputs                 */
    CDRL_BUF_LIN_V = 7,      /* (B, 4)                                        */
    CDRL_BUF_SAMPLE = 8      /* (B, A) Beta sample drawn by ..._resample       */
};
int cdrl_learner_get_buffer(const cdrl_learner* l, int which, float** ptr, int64_t* elems);
/* Named internal tensors of a bound learner, for parity tests: "<bn>.x" (raw BatchNorm input, dense NHWC rows with frames
 * f = t*B + b), "<bn>.stats" ([4][T][C]: mean, invstd, scale, shift of the last training forward), "<dense>.z" (dense
 * pre-activation), "img.stem.pool.argmax" (one byte ky*3+kx per pooled element).  The discrete decisions of the last forward
 * (ReLU6 region = region of fmaf(scale, x, shift); max-pool argmax) are reconstructed from them, so that the float64 oracle
 * can be evaluated on the SAME decisions (reference core/architectures.py:47,161 are the sites). */
int cdrl_learner_named_buffer(const cdrl_learner* l, const char* name, void** ptr, int64_t* bytes);
/* Out-of-bounds canaries of the workspace (sanitizer row; there is no GPU AddressSanitizer on this pool and the
 * kernels address the workspace through raw buffer descriptors).  A learner created with CDRL_GUARD=1 in the environment plans its
 * workspace with a 64 KB band behind EVERY tensor (cdrl_learner_workspace_bytes grows accordingly) and cdrl_learner_bind fills the
 * bands with a pattern; this call checks them on `stream` (synchronises): *bad_bands = number of bands that no longer hold the pattern,
 * *first_bad_offset = byte offset (inside the workspace) of the first one, -1 if none.  Without CDRL_GUARD=1 it fails with -1. */
int cdrl_learner_check_guards(cdrl_learner* l, void* stream, int64_t* bad_bands, int64_t* first_bad_offset);

/* ---- rollout-buffer post-processing ---------------------------------------------------------
 * PPOMemory.compute_returns + compute_advantages (rl/agents/ppo.py:699-727), utils.gae /
 * discount_cumsum / decompose_number / tf_sp_norm (rl/utils.py:57-84,140-151,344-349).
 * rewards (N+1) and values_be (N+1, 2) already hold the bootstrap entry of end_trajectory
 * (rl/agents/ppo.py:692-697).  scratch: >= 2*(N+1)+2 doubles. */
int cdrl_gae_returns(const float* rewards, const float* values_be, int N, double gamma, double lambda, float scale,
                     float* returns, float* returns_be, float* adv_raw, float* adv, double* scratch, void* stream);

/* tfp.distributions.Beta(alpha, beta).sample() with reparameterisation gradients (core/networks.py:
 * 136-137): u = g1/(g1+g2), g ~ Gamma via Marsaglia-Tsang on a Philox-4x32-10 stream, du/dalpha and
 * du/dbeta by implicit differentiation of the Gamma CDF.  Element (row, col) reads alpha[row*ld + col]. */
int cdrl_beta_sample(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                     float* u, float* du_dalpha, float* du_dbeta, void* stream);
/* Rollout form (CARLANetwork.predict, core/networks.py:181-193 -> PolicyNetwork.call :96-103): the sample and the
 * log-density of the sample clipped to [eps, 1 - eps], no Jacobians. */
int cdrl_beta_sample_logp(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                          float* u, float* log_prob, void* stream);
/* bf16 ACTIVATION STORAGE at op level (configuration 3): the op-level entry points that have a bf16-storage form -- cdrl_bn_train_fwd /
 * _bwd (y, out, dout, dy), cdrl_dwconv_bn_fwd / _bwd (x, y, dout, dx), cdrl_pwconv_fused_packed and cdrl_pwconv_bn_bwd(_packed) with
 * packed_bf16 = 1 (a, c, epi_y; dout, y, x, dx), cdrl_pwconv_bwd_fused (+ _workspace), cdrl_gemm_tn (A, D), cdrl_gemm_x3 (A, C),
 * cdrl_maxpool_bn_fwd (y, p), cdrl_stem_fwd_stats (y), cdrl_stem_block_bwd(_pooled) (y, dp) -- take the element type of those ACTIVATION
 * tensors as an explicit argument `int act_type` in front of the stream: 0 float32, 1 bf16 (same pointer spelling, element strides and
 * offsets; round-to-nearest-even on store).  Statistics, coefficient blocks, partial sums, weights and weight gradients stay float32 /
 * double.  The library keeps no mode of its own (rounds 3-5 had a thread-local switch, cdrl_set_op_activation_type); cdrl_learner_* is
 * not concerned (Config::compute selects its storage). */
int cdrl_gamma_implicit_grad(const double* a, const double* g, int n, double* out, void* stream);
/* Test hook: the two Gamma draws (g1 ~ Gamma(alpha), g2 ~ Gamma(beta)) behind the sample u = g1 / (g1 + g2) of the same
 * (seed, offset) stream -> gammas[rows * A][2] doubles.  Lets a test check du/dalpha, du/dbeta sample by sample. */
int cdrl_beta_sample_gammas(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                            double* gammas, void* stream);
/* Test hook: the first `nblocks` raw Philox-4x32-10 blocks of the streams (seed, offset, idx0 + i), i < n -> out[n][nblocks][4].
 * (seed, offset) selects a stream: streams of different offsets are DISJOINT for any number of blocks per element (the block
 * counter lives in the top 16 bits of the element-index word, not in the offset). */
int cdrl_philox_words(uint64_t seed, uint64_t offset, uint64_t idx0, int n, int nblocks, uint32_t* out, void* stream);

/* Float32 1x1 convolution on the bf16 matrix pipe (exact three-way bf16 split of both operands, six bf16 MFMAs per K = 16
 * step; float32 in / out, float32-accurate, not bit-identical to an fmaf chain).  W_packed: cdrl_pwconv_x3_packed_bytes(K)
 * bytes written by cdrl_pwconv_x3_pack from B(k, n) = W[k * sbk + n * sbn].  Same prologue / epilogue contract as
 * cdrl_pwconv_fused (pro_stats: BN-apply on load; part: statistics partials, cdrl_pwconv_x3_partial_rows rows per group).
 * K, N <= 128: K, lda, a_coff even (the A chunks are 16-byte buffer loads at dword alignment; columns beyond K are zeroed: the
 * 58-channel convs of stage 0 qualify since round 6); 128 < K or N <= 256: K, lda, a_coff multiples of 4.  K or N above 128 (the 232-channel convs of stage 2, core/architectures.py:130,140 at
 * num_channels 464): W_packed holds one block per 128 output columns -- cdrl_pwconv_x3_packed_bytes_n(K, N) bytes -- and the kernel
 * takes one 32-row tile per workgroup (one statistics row per tile). */
int64_t cdrl_pwconv_x3_packed_bytes(int K);
int64_t cdrl_pwconv_x3_packed_bytes_n(int K, int N);
int cdrl_pwconv_x3_partial_rows(int G, int Mg, int N, int K);
int cdrl_pwconv_x3_pack(const float* W, int K, int N, int sbk, int sbn, void* packed, void* stream);
int cdrl_pwconv_x3(const float* A, int lda, int a_coff, const float* pro_stats, const void* W_packed, const float* bias, float* C,
                   int ldc, int c_coff, int G, int Mg, int N, int K, double* part, void* stream);
/* Backward-data of those wide convs (128 < Cin or Cout <= 256, Cout % 4 == 0) with the BatchNorm backward of the BatchNorm behind the conv
 * applied on load (core/architectures.py:130-141 under tape.gradient): da[G*Mg][ldda] (+ da_coff; += when accumulate) = dy W^T with
 * dy = k1 (mask dz - k2 - xhat(y) k3); dz [G*Mg][ld_dz] (+ dz_coff, gathered through the channel shuffle of dz_shuffle channels when
 * non-zero, ReLU6-masked from the BatchNorm output when act), y [G*Mg][Cout] raw conv output, stats [4][G][Cout] / coef [3][G][Cout].
 * W_packed: cdrl_pwconv_x3_pack(W, Cout, Cin, 1, Cout, ...) with cdrl_pwconv_x3_packed_bytes_n(Cout, Cin) bytes.  part2 (or NULL):
 * [G][rows][Cout] column sums of dy (the conv's bias gradient partials), rows = cdrl_pwconv_x3_wide_bwd_rows(Mg) (one per 32-row tile).
 * ey / epi_stats / part (all or none; not with accumulate): (sum da, sum da xhat(ey)) of the BatchNorm whose raw input is ey [G*Mg][Cin],
 * part [G][rows][2][Cin].  float32 tensors only. */
int cdrl_pwconv_x3_wide_bwd_rows(int Mg);
int cdrl_pwconv_x3_wide_bwd(const float* dz, int ld_dz, int dz_coff, int dz_shuffle, int act, const float* y, const float* stats,
                            const float* coef, const void* W_packed, float* da, int ldda, int da_coff, int accumulate, int G, int Mg,
                            int Cin, int Cout, double* part2, const float* ey, const float* epi_stats, double* part, void* stream);


/* Backward of the unit's 1x1 convolution + BatchNorm (core/architectures.py:130-141 under tape.gradient, core/carla_agent.py:
 * 364-365) as ONE pass over its operands: dy = BatchNorm-backward(dz, y) on load, da = dy W^T, dW = a^T dy, db = column sums of
 * dy.  dz [G*Mg][ld_dz] (+ dz_coff, gathered through the channel shuffle of dz_shuffle channels when non-zero, ReLU6-masked from
 * the BatchNorm output when act == 1), y [G*Mg][N] raw conv output, stats [4][G][N] / coef [3][G][N] of that BatchNorm.
 * a [G*Mg][lda] (+ a_coff) conv input; a_stats ([4][G][K], or NULL): a is the raw input of a BatchNorm (no activation) applied on
 * load, and that BatchNorm's dgamma / dbeta [K] and backward coefficients a_coef [3][G][K] are produced as well (from the filter
 * product, no pass over da).  W [K][N]; W_packed: cdrl_pwconv_x3_pack(W, N, K, 1, N, ...) (the transposed operand).
 * da [G*Mg][ldda] (+ da_coff; += when accumulate).  Workspaces: qpart cdrl_pwconv_bwd_fused_workspace(G, Mg, N, K, 0) floats,
 * dbpart (..., 1) doubles.  K, N <= 128 and padded alike (both <= 64 or both > 64), even; float32-accurate.
 * act_type = 1: dz, y, a, da are bf16 (bf16 activation storage: one bf16 plane per MFMA operand, i.e. dy,
 * xhat / a and W rounded to nearest even; leading dimensions / offsets even); workspaces sized under the same setting. */
int64_t cdrl_pwconv_bwd_fused_workspace(int G, int Mg, int N, int K, int which, int act_type);
int cdrl_pwconv_bwd_fused(const float* dz, int ld_dz, int dz_coff, int dz_shuffle, int act, const float* y, const float* stats,
                          const float* coef, const float* a, int lda, int a_coff, const float* a_stats, const float* a_gamma,
                          const float* a_beta, float* a_dgamma, float* a_dbeta, float* a_coef, const float* W, const void* W_packed,
                          float* da, int ldda, int da_coff, int accumulate, float* dW, float* db, float* qpart, double* dbpart, int G,
                          int Mg, int N, int K, int act_type, void* stream);

/* General float32 GEMM C (+)= A B + bias on the bf16 matrix pipe (three-way operand split; the 464 -> 768 head conv of
 * core/architectures.py:170 and its backward-data product).  B_packed: cdrl_gemm_x3_packed_bytes(N, K) bytes written by
 * cdrl_gemm_x3_pack from B(k, n) = B[k * sbk + n * sbn].  K, lda, a_coff multiples of 4, A 16-byte aligned. */
int64_t cdrl_gemm_x3_packed_bytes(int N, int K);
int cdrl_gemm_x3_pack(const float* B, int K, int N, int sbk, int sbn, void* packed, void* stream);
int cdrl_gemm_x3(const float* A, int lda, int a_coff, const void* B_packed, const float* bias, float* C, int ldc, int c_coff, int M,
                 int N, int K, int accumulate, int act_type, void* stream);

/* bf16 path (BASELINE.json configuration 3), first kernel: the unit's 1x1 convolution (core/architectures.py:130,140) with
 * bf16 activations in HBM, float32 master weights / bias, v_mfma_f32_32x32x16_bf16 with float32 accumulate.  A [G*Mg][lda]
 * bf16 (+ a_coff), C [G*Mg][ldc] bf16; pro_stats ([4][G][K] float32 or NULL): BatchNorm-apply of the previous layer on load;
 * part ([G][rows][2][N] double or NULL, rows = cdrl_pwconv_bf16_partial_rows): (sum, sum of squares) of the ROUNDED outputs
 * per channel for the following BatchNorm.  K, N <= 128; K, lda, a_coff multiples of 4.
 * W_packed (optional): the weights as bf16 MFMA fragments, written once per weight version by cdrl_pwconv_bf16_pack
 * (cdrl_pwconv_bf16_packed_elems(K) bf16 elements); with it a workgroup's weight prologue is 8 x 16-byte loads per lane. */
int cdrl_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int cdrl_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);
int cdrl_pwconv_bf16_partial_rows(int G, int Mg, int N, int K);
int64_t cdrl_pwconv_bf16_packed_elems(int K);
int cdrl_pwconv_bf16_pack(const float* W, int K, int N, void* packed, void* stream);
int cdrl_pwconv_bf16(const void* A, int lda, int a_coff, const float* pro_stats, const float* W, const void* W_packed,
                     const float* bias, void* C, int ldc, int c_coff, int G, int Mg, int N, int K, double* part, void* stream);

/* One time step of the Keras GRU v2 cell (reset_after=True, gates z, r, h; reference core/networks.py:47-50 ->
 * keras.layers.GRU(unroll=True)) as ONE kernel per direction.  xp = x K + b0 of the step [B][3u], hprev [B][u], R [u][3u],
 * b1 [3u]; saved for the backward: z, r, hh [B][u] and hp = hprev R + b1 [B][3u].  Backward: dh [B][ld_dh] gradient w.r.t.
 * the step's output, RT = R^T [3u][u]; outputs dxp (gradient w.r.t. xp), dhp (gradient w.r.t. hp), dhprev (may be null). */
int cdrl_gru_step_fwd(const float* xp, const float* hprev, const float* R, const float* b1, float* z, float* r, float* hh,
                      float* hp, float* hnew, int B, int u, void* stream);
int cdrl_gru_step_bwd(const float* dh, int ld_dh, const float* z, const float* r, const float* hh, const float* hp,
                      const float* hprev, const float* RT, float* dxp, float* dhp, float* dhprev, int B, int u, void* stream);

/* Minibatch assembly: utils.data_to_batches' tf.data gather of the shuffled rollout rows
 * (rl/utils.py:365-393); dst[i, :] = src[idx[i], :], idx = int32 device array of row numbers. */
int cdrl_gather_rows(const float* src, const int32_t* idx, float* dst, int nrows, int64_t row_elems, void* stream);

/* ---- op-level entry points (Keras layer call -> one launch; used by the parity tests) -------- */
/* Conv2D(k=1) / Dense forward: C[M,N] (+)= A[M,K] B[K,N] + bias (core/architectures.py:130,134,140,170) */
int cdrl_gemm_nn(const float* A, int lda, int a_coff, const float* B, int sbk, int sbn, const float* bias, float* C,
                 int ldc, int c_coff, int M, int N, int K, int accumulate, void* stream);
int64_t cdrl_gemm_tn_workspace_elems(int M, int N, int K);
int cdrl_gemm_tn(const float* A, int lda, int a_coff, const float* D, int ldd, int d_coff, float* out, int M, int N,
                 int K, float* workspace, int accumulate, int act_type, void* stream);
/* Conv2D(24, 3, strides=2) stem (core/architectures.py:159) */
int cdrl_stem_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int H, int W, int Cout,
                  void* stream);
/* The same conv with the statistics of the BatchNorm behind it in the epilogue (what the engine runs in training; conv.hip
 * stem_fwd_band_kernel: the image band staged in LDS, or the window form with CDRL_STEM_FWD_BAND=0): y bit-identical to cdrl_stem_fwd,
 * part [T][rows][2][Cout] doubles = per-workgroup (sum y, sum y^2) per time slice, rows = cdrl_stem_fwd_stats_rows(), the layout
 * cdrl_bn_train_fwd's finalize consumes.  Cout % 4 == 0, <= 64; y 16-byte aligned.  Replaces Conv2D(stem) + the moments of its
 * BatchNormalization (core/architectures.py:159-160). */
int cdrl_stem_fwd_stats_rows(int B, int T, int H, int W, int Cout);
int cdrl_stem_fwd_stats(const float* x, const float* w, const float* bias, float* y, double* part, int B, int T, int H, int W, int Cout,
                        int act_type, void* stream);
int64_t cdrl_stem_bwd_workspace_doubles(int B, int T, int H, int W, int Cout);
int cdrl_stem_bwd_filter(const float* x, const float* dy, float* dw, float* db, int B, int T, int H, int W, int Cout,
                         double* workspace, void* stream);
/* DepthwiseConv2D(3, strides, 'same') (core/architectures.py:132,138) */
int cdrl_dwconv_fwd(const float* a, const float* w, const float* bias, float* y, int N, int H, int W, int C, int stride,
                    void* stream);
int cdrl_dwconv_bwd_data(const float* dy, const float* w, float* da, int N, int H, int W, int C, int stride, void* stream);
int64_t cdrl_dwconv_bwd_workspace_doubles(int N, int H, int W, int C, int stride);
int cdrl_dwconv_bwd_filter(const float* a, const float* dy, float* dw, float* db, int N, int H, int W, int C, int stride,
                           double* workspace, void* stream);
/* Pointwise Conv2D(k=1) of the ShuffleNet unit (core/architectures.py:130,140) as a persistent skinny GEMM for
 * K, N <= 128 with the neighbouring per-time-slice BatchNormalization folded in.  Rows: G groups of Mg rows.
 *   pro_stats != NULL (4*G*K block of the previous BN): a <- scale[g][k]*a + shift[g][k] on load;
 *   epilogue 1: part[g][b][2][N] = (sum c, sum c^2) of the output -> statistics of the following BN;
 *   epilogue 2: part[g][b][2][N] = (sum c, sum c*xhat) with xhat = (epi_y - mean)*invstd from epi_stats (4*G*N):
 *               BN-backward sums when c is the gradient w.r.t. that BN's output (backward-data GEMM, B = W^T via
 *               sbk/sbn as in cdrl_gemm_nn).
 * b < cdrl_pwconv_fused_partial_rows(G, Mg, N, K); returns -1 (cdrl_last_error) for unsupported shapes/alignment. */
int cdrl_pwconv_fused_partial_rows(int G, int Mg, int N, int K);
int cdrl_pwconv_fused(const float* a, int lda, int a_coff, const float* pro_stats, const float* w, int sbk, int sbn,
                      const float* bias, float* c, int ldc, int c_coff, int accumulate, int G, int Mg, int N, int K,
                      int epilogue, const float* epi_y, const float* epi_stats, double* part, void* stream);
/* Same operator with the weights pre-packed in MFMA fragment order (cdrl_pwconv_pack: cdrl_pwconv_pack_elems(N, K) floats of
 * space, written once per weight version from B(k, n) = W[k * sbk + n * sbn]).  packed_bf16 = 0: float32 fragments, results
 * bit-identical to cdrl_pwconv_fused.  packed_bf16 = 1 (cdrl_pwconv_pack(..., bf16 = 1)): the bf16-OPERAND compute mode of
 * configuration 3 (BASELINE.json configs[2]) -- both MFMA operands are rounded to bf16 (round-to-nearest-even: A after the
 * prologue on its way into LDS, W at pack time), the product runs as v_mfma_f32_32x32x16_bf16 with float32 accumulation;
 * tensors in HBM, bias, statistics and epilogues stay float32.  Reference layer: core/architectures.py:130,140. */
int64_t cdrl_pwconv_pack_elems(int N, int K);
int cdrl_pwconv_pack(const float* W, int K, int N, int sbk, int sbn, float* packed, int bf16, void* stream);
int cdrl_pwconv_fused_packed(const float* a, int lda, int a_coff, const float* pro_stats, const float* w, int sbk, int sbn,
                             const float* bias, float* c, int ldc, int c_coff, int accumulate, int G, int Mg, int N, int K,
                             int epilogue, const float* epi_y, const float* epi_stats, double* part, const float* w_packed,
                             int packed_bf16, int act_type, void* stream);
/* Backward of [Conv2D(k=1) -> BatchNormalization(training, per time slice) (+ReLU6) (+channel_shuffle on the store)]
 * (core/architectures.py:130-131,140-145) without materialising the gradient w.r.t. the conv output: the BN-backward
 * "apply" runs as the operand prologue of the two GEMMs.  dout: gradient w.r.t. the BN output (view ld/coff, read through
 * the shuffle map when shuffle_ctot != 0); y: raw conv output [G*Mg][N]; stats: that BN's 4*G*N block; x: conv input
 * view [G*Mg][K]; x_pro_stats != NULL: the conv input is BN-apply(x) with those statistics (4*G*K) as in
 * cdrl_pwconv_fused.  Outputs: dgamma, dbeta, coef (3*G*N scratch), dx (view, accumulate flag), dw (K x N), db (N).
 * workspace: cdrl_pwconv_bn_bwd_workspace_bytes().  Shapes as cdrl_pwconv_fused (K, N <= 128, even). */
int64_t cdrl_pwconv_bn_bwd_workspace_bytes(int G, int Mg, int N, int K);
int cdrl_pwconv_bn_bwd(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, int relu6, const float* y,
                       const float* stats, const float* x, int x_ld, int x_coff, const float* x_pro_stats, const float* w,
                       int G, int Mg, int N, int K, float* dgamma, float* dbeta, float* coef, float* dx, int dx_ld,
                       int dx_coff, int accumulate, float* dw, float* db, void* workspace, int act_type, void* stream);
/* ... with the backward-data operand W^T pre-packed: cdrl_pwconv_pack(W, N, K, 1, N, wt_packed, bf16) packs
 * B(k = n_out, n = k_in) = W[k_in * N + n_out]; packed_bf16 = 1 runs the backward-data GEMM AND the filter-gradient GEMM in the
 * bf16-operand mode (dz rounded to bf16 after the BatchNorm-backward prologue, x after its BatchNorm-apply prologue); db,
 * dgamma, dbeta are float32 reductions. */
int cdrl_pwconv_bn_bwd_packed(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, int relu6, const float* y,
                              const float* stats, const float* x, int x_ld, int x_coff, const float* x_pro_stats, const float* w,
                              int G, int Mg, int N, int K, float* dgamma, float* dbeta, float* coef, float* dx, int dx_ld,
                              int dx_coff, int accumulate, float* dw, float* db, void* workspace, const float* wt_packed,
                              int packed_bf16, int act_type, void* stream);
/* Fused depthwise block of the ShuffleNet unit: [BatchNormalization + ReLU6 of the previous 1x1 conv, applied on
 * load] -> DepthwiseConv2D(3, stride, 'same') -> statistics of the BatchNormalization that follows
 * (core/architectures.py:130-139; per-time-slice BN :44-57).  Whole frames are staged in LDS; the normalised
 * depthwise input is never written to memory.  x: [G*B frames][H][W][C] raw output of the previous conv
 * (pre_stats = that BN's 4*G*C statistics block from cdrl_bn_train_fwd / this function) or, with pre_stats = NULL,
 * an activation tensor used as is.  Outputs: y (raw depthwise output), post_stats (4*G*C) of the following BN
 * (moving statistics updated as in cdrl_bn_train_fwd). */
int64_t cdrl_dwconv_bn_workspace_doubles(int G, int B, int H, int W, int C, int stride);
int cdrl_dwconv_bn_fwd(const float* x, const float* pre_stats, const float* w, const float* bias, float* y, int G, int B,
                       int H, int W, int C, int stride, const float* gamma, const float* beta, float* moving_mean,
                       float* moving_var, int bessel, float* post_stats, double* workspace, int act_type, void* stream);
/* Backward of the same block.  dout: gradient w.r.t. the OUTPUT of the following BatchNormalization (no activation);
 * y: raw depthwise output.  Produces dw (3,3,C,1), db, the following BN's dgamma/dbeta (+ coef_post, 3*G*C scratch)
 * and dx = gradient w.r.t. x; with pre_stats != NULL the previous BN(+ReLU6) is back-propagated too
 * (dgamma_pre, dbeta_pre, coef_pre 3*G*C scratch) so that dx is the gradient w.r.t. the raw conv output x. */
int cdrl_dwconv_bn_bwd(const float* x, const float* pre_stats, const float* dout, const float* y, const float* post_stats,
                       const float* w, int G, int B, int H, int W, int C, int stride, float* dx, float* dw, float* db,
                       float* dgamma_post, float* dbeta_post, float* coef_post, float* dgamma_pre, float* dbeta_pre,
                       float* coef_pre, double* workspace, int act_type, void* stream);
/* MaxPooling2D(3, 2, 'same') (core/architectures.py:161) */
int cdrl_maxpool_fwd(const float* a, float* p, uint8_t* argmax, int N, int H, int W, int C, void* stream);
int cdrl_maxpool_bwd(const uint8_t* argmax, const float* dp, float* da, int N, int H, int W, int C, void* stream);
/* BatchNormalization(training=True) per time slice + optional ReLU6 + optional channel_shuffle on
 * the store (core/architectures.py:44-57,109-118).  stats: 4*G*C floats, workspace: G*256*2*C doubles. */
int cdrl_bn_train_fwd(const float* y, int G, int Mg, int C, const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, int bessel, int relu6, float* out, int out_ld, int out_coff, int shuffle_ctot,
                      float* stats, double* workspace, int act_type, void* stream);
/* Single-group BatchNorm over a few hundred rows (the dense BatchNorms of the trunk tail and of control_branch,
 * core/networks.py:59-66, :53-54) as ONE launch per direction: statistics + moving-average update + apply (forward),
 * sums + coefficients + apply (backward).  y / out / dout / dx dense [M][C]; stats 4*C, coef 3*C floats. */
int cdrl_bn_small_fwd(const float* y, int M, int C, const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, float* stats, float* out, void* stream);
int cdrl_bn_small_bwd(const float* dout, const float* y, int M, int C, const float* stats, float* dgamma, float* dbeta,
                      float* coef, float* dx, void* stream);
/* The linear output heads of a control branch (core/networks.py:128-137, :267-275: up to 4 Dense(K -> n_h) layers on the
 * same [B][K] activation, 8 outputs in total) as one launch per direction.  w[h]: [K][n[h]], b[h]: [n[h]];
 * lin / dlin: [B][sum n]; backward: da [B][K] (overwritten), dw[h], db[h]. */
int cdrl_linear_heads_fwd(const float* a, int nheads, const int* n, const float* const* w, const float* const* b, float* lin,
                          int B, int K, void* stream);
int cdrl_linear_heads_bwd(const float* a, int nheads, const int* n, const float* const* w, const float* dlin, float* da,
                          float* const* dw, float* const* db, int B, int K, void* stream);
int cdrl_bn_train_bwd(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, const float* y, int G, int Mg,
                      int C, const float* stats, int relu6, float* dgamma, float* dbeta, float* dy, float* coef,
                      double* workspace, int act_type, void* stream);
/* Fused stem block (core/architectures.py:160-161): BatchNorm-apply + ReLU6 + MaxPooling2D(3,2,'same') on the
 * raw conv output (stats from cdrl_bn_train_fwd), and the BatchNorm backward that gathers its incoming
 * gradient from the pooled gradient `dp` through the saved argmax.  argmax codes: ky * 3 + kx of the winning window position in bits 0-3;
 * bit 7 is set where the winning activation is clamped (ReLU6 closed: no gradient flows); decoders mask it away. */
int cdrl_maxpool_bn_fwd(const float* y, const float* stats, int G, int frames_per_group, float* p, uint8_t* argmax,
                        int N, int H, int W, int C, int act_type, void* stream);
int cdrl_bn_train_bwd_pooled(const uint8_t* argmax, const float* dp, int H, int W, const float* y, int G, int Mg, int C,
                             const float* stats, float* dgamma, float* dbeta, float* dy, float* coef, double* workspace,
                             void* stream);
/* Backward of the whole stem block Conv2D(3x3,s2,valid) -> BatchNormalization+ReLU6 -> MaxPooling2D(3,2,'same')
 * (core/architectures.py:159-161) from the POOLED gradient dp, without materialising any pre-pool gradient: the BN sums
 * are taken in scatter form over dp (one gathered y per pooled element), the BN-backward apply and the pool gather run
 * inside the operand load of the filter-gradient GEMM.  x: observations (B,T,H,W,3); y: raw conv output
 * [(t*B+b)][Ho][Wo][Cout]; stats: the BN's 4*T*Cout block; argmax/dp: [T*B][Hp][Wp][Cout].  Outputs dgamma, dbeta, coef
 * (3*T*Cout scratch), dw (3,3,3,Cout), db.  workspace: cdrl_stem_block_bwd_workspace_doubles().  Cout % 4 == 0, <= 32. */
int64_t cdrl_stem_block_bwd_workspace_doubles(int B, int T, int H, int W, int Cout);
int cdrl_stem_block_bwd(const float* x, const float* y, const float* stats, const uint8_t* argmax, const float* dp, int B, int T,
                        int H, int W, int Cout, float* dgamma, float* dbeta, float* coef, float* dw, float* db,
                        double* workspace, int act_type, void* stream);
/* Same, with the pooled ACTIVATED output `pooled` of cdrl_maxpool_bn_fwd (null = the form above; element type as y / dp): the
 * BatchNorm sums then take the ReLU6 decision and xhat from it instead of gathering the pre-pool value through the argmax (a dense
 * read of a tensor a quarter the size; what the engine does).  Channels with |gamma * invstd| < 0.05 keep the gather.  With bf16
 * storage xhat comes from the ROUNDED pooled value: bf16-level agreement with the gather form, not bit-wise. */
int cdrl_stem_block_bwd_pooled(const float* x, const float* y, const float* stats, const uint8_t* argmax, const float* dp,
                               const float* pooled, int B, int T, int H, int W, int Cout, float* dgamma, float* dbeta, float* coef,
                               float* dw, float* db, double* workspace, int act_type, void* stream);

/* Rollout-time image augmentation of one observation stack (CARLAgent.augment, core/carla_agent.py:545-577; ops of
 * rl/augmentations/augmentations.py and simclr.color_jitter): color jitter (brightness -> contrast -> saturation -> hue ->
 * clip) -> random-kernel blur -> salt & pepper -> gaussian noise -> per-image min-max normalisation -> cutout -> coarse
 * dropout.  The plan says which ops fire and with which scalars (the host draws it, as tf_chance / tf.image.random_* do in
 * the reference); per-pixel random fields are Philox-4x32-10 streams of (seed, offset).  in/out: [T][H][W][3] floats on
 * the device (may not alias); workspace: cdrl_augment_workspace_floats(T, H, W) floats. */
typedef struct {
    int jitter;                 /* 1: color jitter with the four scalars below */
    float brightness, contrast, saturation, hue;
    int blur_size;              /* 0 (off), 3 or 5 */
    float blur_kernel[75];      /* [k][k][3], used [0, 3*k*k) */
    int salt_pepper;            /* 1: tf_salt_and_pepper_batch(amount, prob) */
    float sp_amount, sp_prob;
    int gauss_noise;            /* 1: tf_gaussian_noise_batch(amount, std) */
    float gn_amount, gn_std;
    int normalize;              /* 1: tf_normalize_batch */
    int cutout_size, cutout_cell;   /* size 0 = off; the zeroed cell (row-major index in the size x size grid) */
    int dropout_size;           /* 0 = off; size x size Bernoulli(1 - amount) grid, nearest-neighbour upsampled */
    float dropout_amount;
    uint64_t seed, offset;
} cdrl_aug_plan;
int64_t cdrl_augment_workspace_floats(int T, int H, int W);
int cdrl_augment_images(const float* in, float* out, int T, int H, int W, const cdrl_aug_plan* plan, float* workspace,
                        void* stream);
/* CARLAgent.policy_objective / value_objective on linear head outputs (core/carla_agent.py:394-428,
 * 469-486); writes d(loss)/d(lin) and 16 metric floats. */
int cdrl_beta_ppo_loss(const float* lin, const float* adv, const float* old_logp, const float* speed,
                       const float* similarity, const float* u, const float* du_da, const float* du_db, float clip_ratio,
                       float entropy_coef, int B, int A, float grad_scale, float* dlin, float* metrics, float* aux,
                       float* hp_scratch16, void* stream);
int cdrl_value_loss(const float* lin, const float* returns, const float* speed, const float* similarity, int B,
                    float exp_scale, float grad_scale, float* dlin, float* metrics, float* values, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDRL_H */
