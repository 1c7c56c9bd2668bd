// Learner engine: owns the layer graph of CARLANetwork (trunk + policy / old-policy / value
// heads), the flat parameter-arena layout and the workspace plan for one batch size.
#pragma once
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "cdrl_kernels.h"

namespace cdrl {

struct Config {
    int B = 1, T = 4, H = 90, W = 120;
    int road = 9, vehicle = 4, navigation = 5, A = 2;
    int stem = 24;
    int stage_c[3] = {116, 232, 464};
    int stage_n[3] = {4, 8, 4};
    int last = 768;
    int feat = 16;
    int rnn_image = 256, rnn_small = 32;
    int dyn = 512;
    int head = 320;
    float exp_scale = 6.0f;
    int compute = 0;        // 0: float32 products; 1: bf16 MFMA operands in the tower's 1x1 convolutions (configuration 3);
                            // 2: 1 + bf16 ACTIVATION STORAGE: every activation / activation-gradient tensor of the image tower
                            //    is bf16 in HBM (statistics, partials, coefficients, weights, accumulators stay float32 / double)
};

enum Model : int { M_TRUNK = 0, M_POLICY = 1, M_VALUE = 2, M_OLD_POLICY = 3 };

struct ParamInfo {
    std::string name;
    int shape[4] = {1, 1, 1, 1};
    int ndim = 1;
    int64_t numel = 0;
    int trainable = 1;
    int model = 0;
    int64_t off = 0;      // element offset inside the model's trainable / state region
};

struct Buffers {
    float* params = nullptr;     // [policy_tr | trunk_tr | value_tr | policy_st | trunk_st | value_st | old_tr | old_st]
    float* grads = nullptr;      // [policy_tr | trunk_tr | value_tr]
    float* adam_m = nullptr;
    float* adam_v = nullptr;
    void* workspace = nullptr;
    size_t workspace_bytes = 0;
};

struct PolicyBatch {
    const float *image, *road, *vehicle, *navigation;       // (B,T,...) reference layout
    const float *adv, *old_logp, *speed, *similarity, *u, *du_da, *du_db;
};
struct ValueBatch {
    const float *image, *road, *vehicle, *navigation;
    const float *returns, *speed, *similarity;
};

struct Op {
    std::function<int(hipStream_t, int)> fwd;
    std::function<int(hipStream_t)> bwd;
};

// Identity half of a ShuffleNet unit carried by the unit's last BatchNorm op: forward = copy through the concat + shuffle
// store of the BN-apply kernel, backward = gather of its gradient inside the BN-backward reduction (no separate launches)
struct Passthrough {
    View fsrc{nullptr, 0, 0}, fdst{nullptr, 0, 0};     // forward: X[:, :C] -> out (shuffled, channel offset 0)
    View gsrc{nullptr, 0, 0}, gdst{nullptr, 0, 0};     // backward: out.g (shuffled) -> X.g[:, :C]
    // global average pool fused behind the BatchNorm (head of the tower): the forward writes mean_p act(BN(x)) to gap_out
    // [frames][C] instead of the activated tensor, the backward reads the pooled gradient gap_dout [frames][C] (broadcast / P)
    float* gap_out = nullptr;
    const float* gap_dout = nullptr;
    int gap_rows = 0;                                   // P pixels per frame
};

// BatchNorm work folded into a pointwise conv (gemm_pw.hip); all optional
struct PwFuse {
    bool fwd_pw = false;                 // forward through the persistent skinny GEMM
    const float* pro_stats = nullptr;    // input = BN-apply(in) with these statistics (in = raw previous conv output)
    bool epi_stats = false;              // forward epilogue: statistics partials of the following BN -> scr_main_.part
    bool bwd_pw = false;                 // backward-data through the persistent skinny GEMM
    const float* bwd_ey = nullptr;       // backward epilogue: BN-backward sums of the BN whose raw input is bwd_ey
    const float* bwd_epi_stats = nullptr;
    // BatchNorm-backward apply of the BN that follows this conv, done as the operand prologue of the backward-data and
    // filter-gradient GEMMs (the gradient w.r.t. the conv output is never materialised)
    bool bb = false;
    const float* bb_stats = nullptr;     // that BN's statistics / backward coefficients (filled by its own backward op)
    const float* bb_coef = nullptr;
    View bb_dz{nullptr, 0, 0};           // gradient w.r.t. the BN output; p == nullptr: the current scratch slot (dense)
    int bb_shuffle = 0, bb_act = 0;
    bool bb_claim_slot = false;          // this op claims the rotating scratch slot (nobody upstream did)
    // fused backward (gemm_pw_bwd.hip): when the conv input is a BatchNorm output applied on load (pro_stats; bwd_ey = its raw
    // input), that BatchNorm's parameters / gradient outputs / coefficient block -- its backward sums come out of the conv's
    // reduce kernel, and its own backward op does nothing (`a_bn_done`)
    bool a_bn = false;                   // the five pointers below are set (they are null in the dry build either way)
    const float* a_gamma = nullptr;
    const float* a_beta = nullptr;
    float* a_dgamma = nullptr;
    float* a_dbeta = nullptr;
    float* a_coef = nullptr;
    std::shared_ptr<bool> a_bn_done;
    // finalize of the BatchNorm BEHIND the conv on load (gemm_pw_bwd.hip, float32): its backward sums sit in *bb_fin_part
    // (bb_fin_nb rows per group); dgamma / dbeta come out of the conv's reduce kernel; `bb_fin_done` tells the BatchNorm's backward op
    bool bb_fin = false;
    double** bb_fin_part = nullptr;      // &scratch.part
    int bb_fin_nb = 0;
    float* bb_dgamma = nullptr;
    float* bb_dbeta = nullptr;
    std::shared_ptr<bool> bb_fin_done;
};

class Learner {
public:
    const std::string& build_error() const { return build_err_; }
    explicit Learner(const Config& cfg);
    ~Learner();

    const Config& config() const { return cfg_; }
    const std::vector<ParamInfo>& params(int model) const { return infos_[model]; }
    int64_t trainable_elems(int model) const { return tr_size_[model]; }
    int64_t state_elems(int model) const { return st_size_[model]; }
    // element offsets of the regions inside Buffers::params / grads
    int64_t tr_offset(int model) const;
    int64_t st_offset(int model) const;
    int64_t params_total() const;
    int64_t grads_total() const { return tr_size_[0] + tr_size_[1] + tr_size_[2]; }
    size_t workspace_bytes() const { return ws_bytes_; }

    int bind(const Buffers& b);
    DevHP* host_hp() { return &hp_host_; }
    int upload_hp(hipStream_t st);          // host hp block (lr, clip...) -> device, keeps device counters
    int reset_counters(hipStream_t st);

    int policy_forward_backward(const PolicyBatch& b, float inv_world, hipStream_t st);
    // split form for the re-sampling loss (F8): forward -> (host samples u from alpha, beta) -> backward
    int policy_forward(const float* image, const float* road, const float* vehicle, const float* navigation,
                       hipStream_t st);
    int policy_backward(const PolicyBatch& b, float inv_world, hipStream_t st);
    // forward -> on-device Beta re-sampling (u, du/dalpha, du/dbeta) -> backward; batch.u / du_* are ignored
    int policy_forward_backward_resample(const PolicyBatch& b, uint64_t seed, uint64_t offset, float inv_world,
                                         hipStream_t st);
    float* sample_buffer() const { return sample_u_; }
    int policy_apply(hipStream_t st);
    int sequence_begin(hipStream_t caller);
    int sequence_end(hipStream_t caller);
    int value_forward_backward(const ValueBatch& b, float inv_world, hipStream_t st);
    int value_apply(hipStream_t st);
    int update_old_policy(hipStream_t st);
    // inference: trunk (moving stats) + old_policy + value heads
    int predict(const float* image, const float* road, const float* vehicle, const float* navigation,
                float* dist_out /*[B][4A]*/, float* value_out /*[B][4]*/, float* dyn_out /*[B][dyn] or null*/,
                hipStream_t st);
    // training-mode forward only (parity tests): returns pointers inside the workspace
    int trunk_forward_train(const float* image, const float* road, const float* vehicle, const float* navigation,
                            hipStream_t st);
    float* dyn_out() const { return dyn_.p; }
    float* img_feat() const { return feat_.p; }
    float* metrics_policy() const { return metrics_p_; }
    float* metrics_value() const { return metrics_v_; }
    float* policy_aux() const { return aux_p_; }
    float* value_aux() const { return aux_v_; }
    float* policy_lin() const { return lin_p_.p; }
    float* value_lin() const { return lin_v_.p; }
    DevHP* dev_hp() const { return hp_dev_; }
    // Use another (bound) learner's device hyper-parameter block -- learning rates, clip, AND the Adam step counters -- so
    // that engines built for different minibatch sizes over the same parameter arenas behave as one optimizer.
    void share_hp(const Learner& owner) { hp_dev_ = owner.hp_dev_; }
    // Data-parallel overlap: `s` (a caller-owned stream, or null) is made to wait, in the middle of every backward pass, for
    // the point where the gradients of the heads and of the trunk tail (GRUs, feature nets, concat BN + Dense) are final --
    // a collective enqueued on `s` after the pass has been enqueued then runs UNDER the tower's backward.
    void set_comm_stream(hipStream_t s) { comm_ = s; }
    // First element (inside the trunk's trainable region) of the TAIL tensors: everything registered behind the image tower,
    // i.e. exactly the gradients that are final at the point the communication stream is released.  Fixed by the op list.
    int64_t tail_offset() const { return tail_off_; }
    bool graphs_enabled() const { return graphs_enabled_; }
    // Named internal tensors (parity tests: the raw BatchNorm inputs, statistics blocks, max-pool argmax codes and dense
    // pre-activations from which the discrete ReLU6 / max-pool decisions of the last forward are reconstructed)
    // CDRL_GUARD=1: 64 KB canary bands behind every workspace tensor (filled at bind); counts the bands that lost their pattern
    int check_guards(hipStream_t st, int64_t* bad, int64_t* first_off);
    bool named_buffer(const std::string& name, void** p, int64_t* bytes) const {
        auto it = named_.find(name);
        if (it == named_.end()) return false;
        *p = it->second.first;
        *bytes = it->second.second;
        return true;
    }

private:
    struct Tens {
        float* p = nullptr;
        float* g = nullptr;
        int rows = 0, C = 0;
        View v(int coff = 0) const { return make_view(p, C, coff); }
        View gv(int coff = 0) const { return make_view(g, C, coff); }
    };
    struct PRef {
        float* p = nullptr;
        float* g = nullptr;
    };
    struct BnRec {
        int G, Mg, C, nb;
        float* stats = nullptr;            // [4][G][C] of this BatchNorm
        float* coef = nullptr;             // [3][G][C] backward coefficients
        float* y = nullptr;                // its (dense) input
        int act = 0;
        // set by the op that produces this BN's incoming gradient when it also accumulates the BN-backward
        // sums (sum dz, sum dz*xhat) in its own pass: the BN backward then skips its reduce kernel
        std::shared_ptr<bool> reduce_fused;
        // the fused backward of the conv in front (float32) folds this BatchNorm's backward sums itself (finalize on load): the
        // BatchNorm backward then skips bn_bwd_finalize.  What that kernel needs: the scratch block the sums are left in, their row
        // count, the gradient slots of gamma / beta.
        std::shared_ptr<bool> fin_by_consumer;
        double** part_ptr = nullptr;        // &scratch.part of the stream the BatchNorm runs on (filled at the end of the build)
        float* dgamma = nullptr;
        float* dbeta = nullptr;
    };

    // --- building
    void build(bool dry);
    float* alloc(size_t n);
    double* alloc_d(size_t n);
    Tens tens(int rows, int C, bool grad = true);
    // tower tensor in the activation type of the build (float32, or bf16 with Config::compute == 2); same Tens / View spelling,
    // p and g then point to bf16 elements (ld and channel offsets count elements)
    Tens tens_a(int rows, int C, bool grad = true);
    int at_ = 0;            // activation type of the tower: 0 float32, 1 bf16 (compute == 2)
    size_t esz() const { return at_ ? 2 : 4; }
    PRef param(int model, const std::string& name, std::initializer_list<int> shape, bool trainable);
    void build_trunk(std::vector<Op>& ops);
    void build_head(std::vector<Op>& ops, int model, const std::string& prefix, Tens& lin, int nheads,
                    const int* head_dims, const char* const* head_names);
    // dx == nullptr: tower mode, the gradient w.r.t. the BN input goes to the current scratch slot
    // stats_nb > 0: the statistics partials were already written by the producing op (that many rows per group)
    BnRec add_bn(std::vector<Op>& ops, int model, const std::string& prefix, View x, int G, int Mg, int C, bool bessel,
                 int act, View out, int out_shuffle, View dout, int dout_shuffle, float* dx, int stats_nb = 0,
                 bool defer_apply = false, Passthrough pass = Passthrough());
    void add_pw(std::vector<Op>& ops, const std::string& prefix, View in, int rows, int Cin, int Cout, float* y,
                View din, int din_acc, BnRec bn_after, PwFuse fuse = PwFuse());
    void add_dw(std::vector<Op>& ops, const std::string& prefix, View in, int N, int H, int W, int C, int stride,
                float* y, View din, int din_acc, const BnRec* pre_bn = nullptr);
    // Fused depthwise block (dwfused.hip): [BN `bn_pre` (+ReLU6) of the raw 1x1-conv output x, or none] -> dw3x3 ->
    // BN `bn_post` (no activation) -> out.  Emits three ops (pre-BN, depthwise, post-BN); the normalised depthwise
    // input and the post-BN input gradient never touch HBM.  din: gradient target when there is no pre-BN.
    // pre_stats_nb > 0: the pre-BN statistics partials were already written (by the producing GEMM's epilogue) with
    // that many partial rows per group; post_apply = false: the post-BN output is not materialised (its consumer
    // applies it on load); post_bwd_nb > 0: the post-BN backward sums were written by the GEMM that produced dout.
    // Returns the post-BN statistics block.
    float* add_dw_block(std::vector<Op>& ops, const std::string& unit, const char* bn_pre, const char* dw, const char* bn_post,
                        float* x, int H, int W, int C, int stride, float* y2, View out, View dout, View din,
                        int pre_stats_nb = 0, bool post_apply = true, int post_bwd_nb = 0, float* stats1_ext = nullptr,
                        float* coef1_ext = nullptr, bool pre_defer_apply = false, float** coef2_out = nullptr,
                        std::shared_ptr<bool> post_bwd_done = nullptr, std::shared_ptr<bool> pre_fin_done = nullptr);
    bool fused_dw_ = true, fused_pw_ = true, fused_pw_wide_ = false;
    int fused_bb_ = 1;
    bool fused_bwd_ = true;             // backward-data + filter gradient of the unit convs as one kernel (gemm_pw_bwd.hip)
    int pw_fwd_nbpg(int G, int Mg, int N, int K) const;    // statistics partial rows per group written by a unit conv's forward
    bool pw_bwd_x3_wide(int G, int Mg, int N, int K) const;               // backward-data on pw_x3_wide_bwd_kernel (N = conv inputs, K = conv outputs)
    int pw_bwd_nbpg(int G, int Mg, int N, int K) const;    // partial rows per group of a unit conv's backward-data (bias sums, BatchNorm sums)
    bool pw_fwd_x3_wide(int G, int Mg, int N, int K) const;               // that forward runs on pw_x3_wide_kernel (float32 engine, 128 < K or N <= 256)
    void add_dense(std::vector<Op>& ops, int model, const std::string& prefix, View in, int M, int K, int N, int act,
                   View out, View dout, View din, int din_acc, bool need_din, const char* bias_init);
    void add_gru(std::vector<Op>& ops, const std::string& name, Tens& x, int In, int u, View out, View dout,
                 bool need_dx);
    void note_scratch(size_t part_d, size_t part2_d, size_t dy_f, size_t tn_f, size_t fpart_d = 0);

    // Every public step runs on the engine's own stream `main_` (bridged to the caller's stream with
    // events) and, when `graphable`, is captured once into a hipGraph (main + side stream, ~1000
    // kernel nodes per pass) and replayed afterwards: the update-step is launch-bound on the host
    // otherwise (measured: 22 ms of CPU enqueue time per 30 ms update-step).  The cache key is the
    // step kind + every pointer / scalar argument baked into the captured kernel arguments.
    int launch(hipStream_t caller, std::vector<uint64_t> key, bool graphable, const std::function<int(hipStream_t)>& body);
    void drop_graphs();
    hipStream_t main_ = nullptr;
    hipEvent_t ev_in_ = nullptr, ev_out_ = nullptr;
    bool graphs_enabled_ = true;
    std::map<std::vector<uint64_t>, hipGraphExec_t> graphs_;
    int policy_backward_impl(const PolicyBatch& b, float inv_world, hipStream_t st);
    int policy_forward_impl(const float* image, const float* road, const float* vehicle, const float* navigation,
                            hipStream_t st);
    int value_forward_backward_impl(const ValueBatch& b, float inv_world, hipStream_t st);
    int update_old_policy_impl(hipStream_t st);
    int policy_apply_impl(hipStream_t st);
    int value_apply_impl(hipStream_t st);
    int predict_impl(const float* image, const float* road, const float* vehicle, const float* navigation, float* dist_out,
                     float* value_out, float* dyn_out, hipStream_t st);
    int run_fwd(std::vector<Op>& ops, hipStream_t st, int training);
    int run_bwd(std::vector<Op>& ops, hipStream_t st);
    int set_inputs(const float* image, const float* road, const float* vehicle, const float* navigation);

    Config cfg_;
    std::vector<ParamInfo> infos_[3];
    std::unordered_map<std::string, int> index_[3];
    int64_t tr_size_[3] = {0, 0, 0}, st_size_[3] = {0, 0, 0};
    bool table_frozen_ = false;

    Buffers buf_;
    bool dry_ = true;
    char* ws_base_ = nullptr;
    size_t ws_off_ = 0, ws_bytes_ = 0;
    static constexpr size_t GUARD_BYTES = 65536, GUARD_TABLE_MAX = 131072;
    bool guard_ = false;
    std::vector<int64_t> guard_off_;        // byte offsets of the bands (real build)
    int64_t* guard_tab_ = nullptr;          // device copy + [GUARD_TABLE_MAX]: bad count, [+1]: first bad band
    void add_guard();
    // scratch maxima (from the dry build) and pointers
    size_t max_part_ = 0, max_part2_ = 0, max_dy_ = 0, max_tn_ = 0, max_fpart_ = 0;
    // reduction scratch of the ops on the main stream / of the auxiliary ops (feature nets + small GRUs),
    // which run concurrently on the side stream and therefore need their own
    struct Scratch {
        double* part = nullptr;
        double* part2 = nullptr;
        float* tn = nullptr;
    };
    Scratch scr_main_, scr_aux_, scr_sc_;
    hipEvent_t ev_sc_fork_[3] = {}, ev_sc_done_[3] = {};   // shortcut branch of the stride-2 units on the side stream (forward)
    Scratch* build_scr_ = &scr_main_;
    std::vector<Op> aux_ops_;
    hipEvent_t ev_aux_fork_ = nullptr, ev_aux_done_ = nullptr;
    void add_aux_fork(std::vector<Op>& ops);
    void add_aux_join(std::vector<Op>& ops);
    // Backward-pass side stream: the filter / bias gradients of the tower (gemm_tn, depthwise and stem
    // filter reductions, db) are off the critical path dy -> bwd-data -> next layer, so they run on a
    // second HIP stream and overlap the latency-bound main chain.  NSLOT rotating scratch sets
    // (dy, db partials, split-M partials, filter partials) + events make the hand-off race-free.
    static constexpr int NSLOT = 8;
    float* dys_[NSLOT] = {};
    double* part2s_[NSLOT] = {};
    float* tns_[NSLOT] = {};
    double* fparts_[NSLOT] = {};
    // fused conv backward: per-workgroup filter-product tiles / column sums.  Their own small ring (the tiles are large: 16-21 MB per
    // conv at B = 256), guarded by events recorded behind the side-stream reduce that reads them
    static constexpr int NQ = 3;
    float* qparts_[NQ] = {};
    double* dbparts_[NQ] = {};
    double* fintots_[NQ] = {};           // [8][2][128] group totals of a finalize-on-load BatchNorm (gemm_pw_bwd.hip)
    bool fin_on_load_ = true;            // the fused conv backward finalizes the BatchNorm behind it on load (CDRL_FIN_ON_LOAD=0: stand-alone launches)
    hipEvent_t ev_q_[NQ] = {};
    bool q_used_[NQ] = {};
    int qi_ = 0;
    int next_q(hipStream_t st);          // main: claim the next buffer pair (waits for the side job that last read it)
    size_t max_qpart_ = 0, max_dbpart_ = 0;
    int flush_side(hipStream_t st);      // enqueue the deferred side jobs now (one event record on `st`)
    hipStream_t side_ = nullptr;
    hipStream_t aux_ = nullptr;          // feature nets + small GRUs (forward and backward)
    bool aux_pending_ = false;
    // Optional second host thread that ENQUEUES the ~340 launches per update-step of the small-modality nets (opt-in,
    // CDRL_AUX_THREAD=1): the single-threaded enqueue of an update-step costs 12.3 ms of host time (8 us per launch incl.
    // the event traffic), which bounds the step for small images.
    struct AuxWorker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> task;
        bool has_task = false, busy = false, stop = false;
        int rc = 0, device = 0;
        std::string err;
        explicit AuxWorker(int dev);
        ~AuxWorker();
        void submit(std::function<int()> fn);
        int wait();             // blocks until the submitted task has been enqueued; returns its status
        void loop();
    };
    std::unique_ptr<AuxWorker> aux_worker_;
    bool aux_inflight_ = false;
    int aux_wait();
    hipEvent_t ev_in_sys_ = nullptr;                    // hand-over from the caller's stream with the fence (data-parallel use)
    hipEvent_t ev_out_sys_ = nullptr;                   // hand-back to the caller's stream WITH the system-scope fence (data-parallel use)
    bool dp_hint_ = false;                              // a pass ran with a gradient scale below 1 (world size > 1)
    bool packs_have_wt_ = false;                        // the last forward's pack launch also wrote the W^T copies of the backward
    bool seq_open_ = false;                             // between sequence_begin and sequence_end on seq_caller_
    hipStream_t seq_caller_ = nullptr;
    TailEvents tail_;                                   // stop events of the critical stream's kernels (cdrl_common.h)
    std::map<std::vector<uint64_t>, std::vector<uint8_t>> tail_need_;      // per body (launch key): which launches a fork follows
    int mark_stream(hipStream_t st, hipEvent_t fallback, hipEvent_t* ev);
    hipEvent_t ev_main_[NSLOT] = {};
    hipEvent_t ev_side_[NSLOT] = {};
    hipEvent_t ev_join_ = nullptr;
    bool side_enabled_ = true;
    bool slot_used_[NSLOT] = {};
    // numbered records of the side stream and what the critical stream has waited for (engine.hip: wait_side_record)
    struct SideRec { hipEvent_t ev = nullptr; uint64_t seq = 0; };
    static constexpr int SIDE_HIST = 32;
    SideRec side_hist_[SIDE_HIST];
    uint64_t side_seq_ = 0, main_waited_ = 0;
    uint64_t slot_seq_[NSLOT] = {}, q_seq_[NQ] = {};
    int side_lag_ = 4;
    uint64_t note_side_record(hipEvent_t ev);
    int wait_side_record(hipStream_t st, uint64_t need, hipEvent_t need_ev);
    int slot_ = 0;
    int next_slot(hipStream_t st);                       // main: claim a scratch slot (waits for its last side job)
    hipStream_t fork_side(hipStream_t st);               // main -> side dependency for the current slot
    int done_side(hipStream_t side);                     // side job of the current slot finished
    // Side jobs that hang off the main stream but are not urgent (partial reductions of bias / depthwise-filter gradients)
    // are queued and enqueued by the NEXT fork_side (or join_side): one event record on the critical stream serves
    // several side jobs (every record is a barrier packet between two dependent main-stream kernels).
    struct Deferred {
        int slot;
        std::function<int(hipStream_t)> fn;
    };
    std::vector<Deferred> deferred_;
    int deferred_rc_ = 0;
    int defer_side(hipStream_t st, std::function<int(hipStream_t)> fn);
    void flush_deferred();
    int join_side(hipStream_t st);

    std::vector<Op> trunk_ops_, policy_ops_, value_ops_, old_policy_ops_;
    // live input pointers (read by the first ops through these slots)
    const float* in_image_ = nullptr;
    const float* in_road_ = nullptr;
    const float* in_vehicle_ = nullptr;
    const float* in_navigation_ = nullptr;

    Tens dyn_, feat_, lin_p_, lin_v_, lin_old_;
    float *metrics_p_ = nullptr, *metrics_v_ = nullptr, *aux_p_ = nullptr, *aux_v_ = nullptr;
    float *sample_u_ = nullptr, *sample_da_ = nullptr, *sample_db_ = nullptr;
    DevHP hp_host_;
    DevHP* hp_dev_ = nullptr;
    DevHP* hp_stage_ = nullptr;     // pinned host staging
    // optimiser tables (device, inside workspace)
    struct SegTable {
        TensorSeg* segs = nullptr;
        int* chunk_tensor = nullptr;
        int64_t* chunk_off = nullptr;
        int ntensors = 0, nchunks = 0;
        double* chunk_part = nullptr;
        float* sqnorms = nullptr;
        std::vector<TensorSeg> h_segs;
        std::vector<int> h_chunk_tensor;
        std::vector<int64_t> h_chunk_off;
    } seg_[3];
    void build_seg_tables();
    int upload_seg_tables();
    // transposed copies of the pointwise-conv weights for the backward-data GEMMs: with W^T in memory the persistent GEMM
    // loads its weight fragments coalesced (the strided fragment load of W cost ~7 us of TA time per CU and launch);
    // refreshed by one batched transpose launch at the start of every trunk backward
    std::vector<PwTranspose> h_pwt_;
    std::map<std::string, float*> pwt_by_name_;
    PwTranspose* d_pwt_ = nullptr;
    int pwt_tiles_ = 0;
    // pointwise-conv weights in MFMA fragment order (forward operand W, backward-data operand W^T), re-packed by ONE launch
    // at the start of every trunk forward: the per-workgroup weight prologue of the persistent GEMM becomes KSM/4 16-byte loads
    std::string build_err_;                 // first configuration error met while the op lists were built (reported by create)
    void build_fail(const char* fmt, ...);
    // inference-mode BatchNorm statistics of the trunk (tower + trunk tail, main and aux streams): one batched launch at the
    // start of an inference forward instead of one per layer (bn_inference_stats_many)
    std::vector<BnInfEntry> h_bninf_;
    BnInfEntry* d_bninf_ = nullptr;
    int bninf_max_c_ = 0;
    void note_bn_inference(const float* gamma, const float* beta, const float* mm, const float* mv, float* stats, int G, int C);
    std::vector<PwPack> h_pack_;
    PwPack* d_pack_ = nullptr;
    float* pw_packed(const float* w, int K, int N, int sbk, int sbn, bool bf16 = false);
    std::vector<PwX3Pack> h_pack3_;         // three-plane bf16 fragments of the convs that run on the bf16 matrix pipe (gemm_pw_x3.hip)
    PwX3Pack* d_pack3_ = nullptr;
    const void* pw_x3_packed(const float* w, int K, int N, int sbk, int sbn);
    std::vector<GemmX3Pack> h_gpack_;       // general split-precision GEMM operands (head conv, wide shortcut convs: gemm_x3.hip)
    GemmX3Pack* d_gpack_ = nullptr;
    const void* gemm_x3_packed(const float* w, int K, int N, int sbk, int sbn);
    int run_trunk_fwd(hipStream_t st, int training);
    hipStream_t comm_ = nullptr;
    int64_t tail_off_ = 0;
    hipEvent_t ev_tail_main_ = nullptr, ev_tail_side_ = nullptr;
    std::vector<std::pair<void*, size_t>> zero_once_;    // workspace regions that must read as zero and are never written
    float* pw_transposed(const std::string& name, const float* w, int cin, int cout);
    bool tables_uploaded_ = false;
    std::map<std::string, std::pair<void*, int64_t>> named_;
    void note_named(const std::string& name, const void* p, size_t bytes) {
        if (!dry_) named_[name] = std::make_pair(const_cast<void*>(p), (int64_t)bytes);
    }
};

}  // namespace cdrl
