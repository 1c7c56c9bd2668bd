// Learner engine implementation (host side, gfx950 kernels from the sibling .hip files).
//
// Mirrors the structure built by the reference's Keras graph builders:
//   trunk  = dynamics_layers            (reference core/networks.py:37-56)
//            shufflenet_v2 over T slices (core/architectures.py:30-173)
//            feature_net x3             (core/architectures.py:9-27)
//   heads  = control_branch + policy / value heads (core/networks.py:59-66,115-137,255-275)
//   steps  = get_*_gradients / apply_*_gradients   (core/carla_agent.py:351-388,430-463;
//                                                    rl/agents/ppo.py:238-275)
// Frames are ordered f = t*B + b, so each per-time-slice BatchNorm group is a contiguous row
// range (F6); split / concat / channel_shuffle never materialise on their own: they are views
// (ld, channel offset) plus a destination-index permutation in the BN-apply store (F7).
#include <cstdarg>
#include <cstdio>

#include "engine.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace cdrl {

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int same_out_h(int n, int s) { return (n + s - 1) / s; }

Learner::Learner(const Config& cfg) : cfg_(cfg) {
    memset(&hp_host_, 0, sizeof(hp_host_));
    hp_host_.lr_policy = 3e-4f;
    hp_host_.lr_value = 3e-4f;
    hp_host_.lr_dynamics = 3e-4f;
    hp_host_.clip_ratio = 0.2f;
    hp_host_.entropy_coef = 1.0f;
    hp_host_.clip_norm_policy = 1.0f;
    hp_host_.clip_norm_value = 1.0f;
    hp_host_.beta1 = 0.9f;
    hp_host_.beta2 = 0.999f;
    hp_host_.eps = 1e-7f;
    at_ = cfg_.compute == 2 ? 1 : 0;
    guard_ = cdrl_getenv("CDRL_GUARD") && atoi(cdrl_getenv("CDRL_GUARD")) == 1;
    build(true);
    table_frozen_ = true;
    build_seg_tables();
}

Learner::~Learner() {
    aux_worker_.reset();
    if (hp_stage_) (void)hipHostFree(hp_stage_);
    for (int i = 0; i < NSLOT; ++i) {
        if (ev_main_[i]) (void)hipEventDestroy(ev_main_[i]);
        if (ev_side_[i]) (void)hipEventDestroy(ev_side_[i]);
    }
    drop_graphs();
    if (ev_in_) (void)hipEventDestroy(ev_in_);
    if (ev_out_) (void)hipEventDestroy(ev_out_);
    if (ev_out_sys_) (void)hipEventDestroy(ev_out_sys_);
    if (ev_in_sys_) (void)hipEventDestroy(ev_in_sys_);
    if (main_) (void)hipStreamDestroy(main_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    for (int i = 0; i < NQ; ++i)
        if (ev_q_[i]) (void)hipEventDestroy(ev_q_[i]);
    for (int i = 0; i < 32; ++i)
        if (tail_.ring[i]) (void)hipEventDestroy(tail_.ring[i]);
    if (ev_tail_main_) (void)hipEventDestroy(ev_tail_main_);
    if (ev_tail_side_) (void)hipEventDestroy(ev_tail_side_);
    for (int i = 0; i < 3; ++i) {
        if (ev_sc_fork_[i]) (void)hipEventDestroy(ev_sc_fork_[i]);
        if (ev_sc_done_[i]) (void)hipEventDestroy(ev_sc_done_[i]);
    }
    if (ev_aux_fork_) (void)hipEventDestroy(ev_aux_fork_);
    if (ev_aux_done_) (void)hipEventDestroy(ev_aux_done_);
    if (side_) (void)hipStreamDestroy(side_);
    if (aux_) (void)hipStreamDestroy(aux_);
}

Learner::AuxWorker::AuxWorker(int dev) : device(dev) { th = std::thread([this] { loop(); }); }

Learner::AuxWorker::~AuxWorker() {
    {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
}

void Learner::AuxWorker::loop() {
    (void)hipSetDevice(device);
    for (;;) {
        std::function<int()> fn;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [this] { return has_task || stop; });
            if (stop) return;
            fn = std::move(task);
            has_task = false;
        }
        const int r = fn();
        {
            std::lock_guard<std::mutex> lk(m);
            rc = r;
            err = r != 0 ? last_error() : "";
            busy = false;
        }
        cv.notify_all();
    }
}

void Learner::AuxWorker::submit(std::function<int()> fn) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return !busy; });
    task = std::move(fn);
    has_task = true;
    busy = true;
    lk.unlock();
    cv.notify_all();
}

int Learner::AuxWorker::wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return !busy; });
    if (rc != 0) set_error("%s", err.c_str());
    return rc;
}

int Learner::aux_wait() {
    if (!aux_inflight_) return 0;
    aux_inflight_ = false;
    return aux_worker_->wait();
}

void Learner::drop_graphs() {
    for (auto& kv : graphs_) (void)hipGraphExecDestroy(kv.second);
    graphs_.clear();
}

int Learner::launch(hipStream_t caller, std::vector<uint64_t> key, bool graphable,
                    const std::function<int(hipStream_t)>& body) {
    if (!main_) {
        set_error("learner not bound");
        return -1;
    }
    // inside a sequence (sequence_begin .. sequence_end on this stream) the hand-overs between the caller's stream and the engine's happen
    // once, around the whole sequence
    const bool in_seq = seq_open_ && caller == seq_caller_;
    if (!in_seq) {
        hipEvent_t ei = (comm_ || dp_hint_) ? ev_in_sys_ : ev_in_;      // (data-parallel use: fenced both ways, see below)
        CDRL_HIP(hipEventRecord(ei, caller));
        CDRL_HIP(hipStreamWaitEvent(main_, ei, 0));
    }
    int rc = 0;
    if (!graphs_enabled_ || !graphable) {
        // eager launches: kernels on the critical stream carry stop events (TailEvents, cdrl_common.h) while the body runs on this thread
        if (tail_.n > 0) {
            if (tail_need_.size() > 64) tail_need_.clear();
            std::vector<uint8_t>& need = tail_need_[key];
            if (need.empty()) need.assign(4096, 0);
            tail_.need = need.data();
            tail_.need_cap = (int)need.size();
            tail_.idx = 0;
            tail_.last = nullptr;
            tl_tail = &tail_;
        }
        rc = body(main_);
        tl_tail = nullptr;
    } else {
        auto it = graphs_.find(key);
        if (it == graphs_.end()) {
            if (graphs_.size() >= 32) drop_graphs();
            hipGraph_t graph = nullptr;
            CDRL_HIP(hipStreamBeginCapture(main_, hipStreamCaptureModeRelaxed));
            rc = body(main_);
            hipError_t ce = hipStreamEndCapture(main_, &graph);
            if (rc != 0 || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                if (rc == 0) {
                    set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce));
                    rc = -2;
                }
                return rc;
            }
            hipGraphExec_t exec = nullptr;
            hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) {
                set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ie));
                return -2;
            }
            it = graphs_.emplace(std::move(key), exec).first;
        }
        CDRL_HIP(hipGraphLaunch(it->second, main_));
    }
    if (rc != 0) return rc;
    if (in_seq) return 0;
    // data-parallel use (a communication stream is set, or a pass ran with a gradient scale below 1): what the caller enqueued before /
    // enqueues next may be a collective whose peers write / read these buffers -- both hand-overs keep the system-scope fence there
    // (unmeasurable on one GPU: insurance)
    hipEvent_t eo = (comm_ || dp_hint_) ? ev_out_sys_ : ev_out_;
    CDRL_HIP(hipEventRecord(eo, main_));
    CDRL_HIP(hipStreamWaitEvent(caller, eo, 0));
    return 0;
}

// A sequence of learner calls on ONE caller stream with nothing of the caller's own in between (an update-step on one GPU: policy pass,
// apply, value pass, apply): every call used to hand the work from the caller's stream to the engine's and back -- two markers and two
// barrier packets across two queues per call, ~10 us of signal latency each way with nothing to order.  Between begin and end the calls
// on that stream skip the hand-overs; begin orders the engine behind the caller's stream, end the caller's stream behind the engine.
int Learner::sequence_begin(hipStream_t caller) {
    if (!main_) {
        set_error("learner not bound");
        return -1;
    }
    if (seq_open_) {
        set_error("sequence_begin: a sequence is already open");
        return -1;
    }
    hipEvent_t ei = (comm_ || dp_hint_) ? ev_in_sys_ : ev_in_;
    CDRL_HIP(hipEventRecord(ei, caller));
    CDRL_HIP(hipStreamWaitEvent(main_, ei, 0));
    seq_open_ = true;
    seq_caller_ = caller;
    return 0;
}

int Learner::sequence_end(hipStream_t caller) {
    if (!seq_open_ || caller != seq_caller_) {
        set_error("sequence_end: no sequence open on this stream");
        return -1;
    }
    seq_open_ = false;
    hipEvent_t eo = (comm_ || dp_hint_) ? ev_out_sys_ : ev_out_;
    CDRL_HIP(hipEventRecord(eo, main_));
    CDRL_HIP(hipStreamWaitEvent(caller, eo, 0));
    return 0;
}

static const int g_diag_skip_fin = cdrl_getenv("CDRL_DIAG_SKIP_FIN") ? atoi(cdrl_getenv("CDRL_DIAG_SKIP_FIN")) : 0;   // timing diagnostics only (stale statistics)
static const int g_diag_noev = cdrl_getenv("CDRL_DIAG_NOEV") ? atoi(cdrl_getenv("CDRL_DIAG_NOEV")) : 0;   // timing diagnostics only (racy)

// Records on the side stream are numbered (the stream is in order: whoever has waited for record s has waited for every record
// before it), and the critical stream remembers the newest one it has waited for.  A scratch slot / partial-tile buffer is free once
// the record of its last user is covered -- so instead of one hipStreamWaitEvent (a barrier packet of its own on the critical stream,
// 0.23 ms per update-step for ~100 of them by the no-waits diagnostic) per claim, the critical stream waits for a record `side_lag_`
// behind the newest one whenever the record it needs is not covered yet, and skips the claims that wait covers (round 6).
uint64_t Learner::note_side_record(hipEvent_t ev) {
    ++side_seq_;
    side_hist_[side_seq_ % SIDE_HIST] = SideRec{ev, side_seq_};
    return side_seq_;
}

int Learner::wait_side_record(hipStream_t st, uint64_t need, hipEvent_t need_ev) {
    if (st != main_ || side_lag_ < 0) {              // another stream than the critical one (or the scheme off): the claim's own event
        CDRL_HIP(hipStreamWaitEvent(st, need_ev, 0));
        return 0;
    }
    if (need <= main_waited_) return 0;
    uint64_t target = need;
    if (side_seq_ > (uint64_t)side_lag_ && side_seq_ - (uint64_t)side_lag_ > target) target = side_seq_ - (uint64_t)side_lag_;
    hipEvent_t ev = need_ev;
    uint64_t covered = need;
    if (side_seq_ - target < SIDE_HIST && side_hist_[target % SIDE_HIST].seq == target) {
        ev = side_hist_[target % SIDE_HIST].ev;
        covered = target;
    }
    for (int i = 0; i < SIDE_HIST; ++i)             // the event may have been recorded again since: waiting for it covers that record too
        if (side_hist_[i].ev == ev && side_hist_[i].seq > covered) covered = side_hist_[i].seq;
    CDRL_HIP(hipStreamWaitEvent(st, ev, 0));
    main_waited_ = covered;
    return 0;
}

int Learner::next_slot(hipStream_t st) {
    slot_ = (slot_ + 1) % NSLOT;
    if (g_diag_noev & 1) return 0;
    if (side_enabled_ && slot_used_[slot_]) CDRL_TRY(wait_side_record(st, slot_seq_[slot_], ev_side_[slot_]));
    return 0;
}

hipStream_t Learner::fork_side(hipStream_t st) {
    if (!side_enabled_) return st;
    if (g_diag_noev & 2) return side_;
    hipEvent_t ev = nullptr;
    if (mark_stream(st, ev_main_[slot_], &ev) != 0) return st;
    if (hipStreamWaitEvent(side_, ev, 0) != hipSuccess) return st;
    flush_deferred();
    return side_;
}

// An event that covers everything `st` holds so far: the stop event of its newest kernel when the tail events are on and nothing
// else went in behind that kernel (nothing is enqueued on `st`), else `fallback` recorded on `st`.
int Learner::mark_stream(hipStream_t st, hipEvent_t fallback, hipEvent_t* ev) {
    TailEvents* t = tl_tail;
    if (t && st == t->stream && t->last) {
        *ev = t->last;
        return 0;
    }
    if (t && st == t->stream && t->idx > 0 && t->idx <= t->need_cap) t->need[t->idx - 1] = 1;     // next run: a stop event on that launch
    CDRL_HIP(hipEventRecord(fallback, st));
    *ev = fallback;
    return 0;
}

void Learner::flush_deferred() {
    for (Deferred& d : deferred_) {
        const int rc = d.fn(side_);
        if (rc != 0 && deferred_rc_ == 0) deferred_rc_ = rc;
        if (d.slot != slot_) {      // (a job of the current slot is covered by the done_side() that follows)
            if (hipEventRecord(ev_side_[d.slot], side_) != hipSuccess && deferred_rc_ == 0) deferred_rc_ = -3;
            slot_seq_[d.slot] = note_side_record(ev_side_[d.slot]);
            slot_used_[d.slot] = true;
        }
    }
    deferred_.clear();
}

int Learner::defer_side(hipStream_t st, std::function<int(hipStream_t)> fn) {
    static const bool on = true;
    if (!side_enabled_ || !on || (g_diag_noev & 2)) {
        hipStream_t side = fork_side(st);
        CDRL_TRY(fn(side));
        return done_side(side);
    }
    deferred_.push_back(Deferred{slot_, std::move(fn)});
    return 0;
}

int Learner::next_q(hipStream_t st) {
    qi_ = (qi_ + 1) % NQ;
    if (side_enabled_ && q_used_[qi_]) CDRL_TRY(wait_side_record(st, q_seq_[qi_], ev_q_[qi_]));
    return 0;
}

int Learner::flush_side(hipStream_t st) {
    if (!side_enabled_ || deferred_.empty()) return 0;
    hipStream_t side = fork_side(st);       // flushes the queue behind an event recorded on `st`
    return done_side(side);
}

int Learner::done_side(hipStream_t side) {
    if (!side_enabled_ || side != side_) return 0;
    if (deferred_rc_ != 0) {
        const int rc = deferred_rc_;
        deferred_rc_ = 0;
        return rc;
    }
    if (g_diag_noev & 4) return 0;
    CDRL_HIP(hipEventRecord(ev_side_[slot_], side_));
    slot_seq_[slot_] = note_side_record(ev_side_[slot_]);
    slot_used_[slot_] = true;
    return 0;
}

int Learner::join_side(hipStream_t st) {
    if (!side_enabled_) return 0;
    if (!deferred_.empty()) {
        hipStream_t side = fork_side(st);       // flushes the queue
        CDRL_TRY(done_side(side));
    }
    CDRL_HIP(hipEventRecord(ev_join_, side_));
    CDRL_HIP(hipStreamWaitEvent(st, ev_join_, 0));
    const uint64_t js = note_side_record(ev_join_);
    if (st == main_) main_waited_ = js;
    if (aux_pending_) {
        CDRL_TRY(aux_wait());
        CDRL_HIP(hipStreamWaitEvent(st, ev_aux_done_, 0));
        aux_pending_ = false;
    }
    for (int i = 0; i < NSLOT; ++i) slot_used_[i] = false;
    for (int i = 0; i < NQ; ++i) q_used_[i] = false;
    return 0;
}

int64_t Learner::tr_offset(int model) const {
    switch (model) {
        case M_POLICY: return 0;
        case M_TRUNK: return tr_size_[M_POLICY];
        case M_VALUE: return tr_size_[M_POLICY] + tr_size_[M_TRUNK];
        default: return grads_total() + st_size_[0] + st_size_[1] + st_size_[2];      // old policy trainable
    }
}

int64_t Learner::st_offset(int model) const {
    const int64_t base = grads_total();
    switch (model) {
        case M_POLICY: return base;
        case M_TRUNK: return base + st_size_[M_POLICY];
        case M_VALUE: return base + st_size_[M_POLICY] + st_size_[M_TRUNK];
        default: return tr_offset(M_OLD_POLICY) + tr_size_[M_POLICY];                  // old policy state
    }
}

int64_t Learner::params_total() const { return st_offset(M_OLD_POLICY) + st_size_[M_POLICY]; }

// ------------------------------------------------------------------------------------------
// allocation helpers
// ------------------------------------------------------------------------------------------
// CDRL_GUARD=1: a canary band behind every bump allocation (the only out-of-bounds detector this stack has for kernels that address
// the workspace through raw buffer descriptors: an overrun into the neighbour tensor is otherwise seen only if a parity test reads it)
void Learner::add_guard() {
    if (!guard_) return;
    if (!dry_) guard_off_.push_back((int64_t)ws_off_);
    ws_off_ += GUARD_BYTES;
}

float* Learner::alloc(size_t n) {
    const size_t bytes = align_up(n * sizeof(float), 256);
    float* p = dry_ ? nullptr : reinterpret_cast<float*>(ws_base_ + ws_off_);
    ws_off_ += bytes;
    add_guard();
    return p;
}

double* Learner::alloc_d(size_t n) {
    const size_t bytes = align_up(n * sizeof(double), 256);
    double* p = dry_ ? nullptr : reinterpret_cast<double*>(ws_base_ + ws_off_);
    ws_off_ += bytes;
    add_guard();
    return p;
}

static constexpr uint32_t GUARD_PATTERN = 0xA5C3961Eu;

__global__ void __launch_bounds__(256) guard_fill_kernel(char* base, const int64_t* __restrict__ offs, int words) {
    uint32_t* p = reinterpret_cast<uint32_t*>(base + offs[blockIdx.x]);
    for (int i = threadIdx.x; i < words; i += 256) p[i] = GUARD_PATTERN ^ (uint32_t)i;
}

__global__ void __launch_bounds__(256) guard_check_kernel(const char* base, const int64_t* __restrict__ offs, int words, int64_t* out) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + offs[blockIdx.x]);
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int b = 0;
    for (int i = threadIdx.x; i < words; i += 256) b |= p[i] != (GUARD_PATTERN ^ (uint32_t)i);
    if (b) bad = 1;          // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0 && bad) {
        atomicAdd(reinterpret_cast<unsigned long long*>(out), 1ull);
        atomicMin(reinterpret_cast<long long*>(out + 1), (long long)blockIdx.x);
    }
}

int Learner::check_guards(hipStream_t st, int64_t* bad, int64_t* first_off) {
    if (!guard_ || !guard_tab_) {
        set_error("check_guards: the learner was not created with CDRL_GUARD=1 (or is not bound)");
        return -1;
    }
    const int n = (int)guard_off_.size();
    int64_t init[2] = {0, (int64_t)1 << 40};
    CDRL_HIP(hipMemcpyAsync(guard_tab_ + GUARD_TABLE_MAX, init, sizeof(init), hipMemcpyHostToDevice, st));
    tail_invalidate(st);
    hipLaunchKernelGGL(guard_check_kernel, dim3(n), dim3(256), 0, st, ws_base_, guard_tab_, (int)(GUARD_BYTES / 4), guard_tab_ + GUARD_TABLE_MAX);
    CDRL_LAUNCH_CHECK();
    int64_t out[2];
    CDRL_HIP(hipMemcpyAsync(out, guard_tab_ + GUARD_TABLE_MAX, sizeof(out), hipMemcpyDeviceToHost, st));
    tail_invalidate(st);
    CDRL_HIP(hipStreamSynchronize(st));
    if (bad) *bad = out[0];
    if (first_off) *first_off = out[0] ? guard_off_[(size_t)out[1]] : -1;
    return 0;
}

Learner::Tens Learner::tens(int rows, int C, bool grad) {
    Tens t;
    t.rows = rows;
    t.C = C;
    t.p = alloc((size_t)rows * C);
    if (grad) t.g = alloc((size_t)rows * C);
    return t;
}

Learner::Tens Learner::tens_a(int rows, int C, bool grad) {
    Tens t;
    t.rows = rows;
    t.C = C;
    const size_t n = ((size_t)rows * C * esz() + 3) / 4;        // float-sized slots
    t.p = alloc(n);
    if (grad) t.g = alloc(n);
    return t;
}

Learner::PRef Learner::param(int model, const std::string& name, std::initializer_list<int> shape, bool trainable) {
    const int tm = model == M_OLD_POLICY ? (int)M_POLICY : model;
    int idx;
    auto it = index_[tm].find(name);
    if (it == index_[tm].end()) {
        if (table_frozen_ || model == M_OLD_POLICY) {
            set_error("unknown parameter %s", name.c_str());
            return PRef();
        }
        ParamInfo pi;
        pi.name = name;
        pi.ndim = (int)shape.size();
        int64_t n = 1;
        int k = 0;
        for (int d : shape) {
            pi.shape[k++] = d;
            n *= d;
        }
        pi.numel = n;
        pi.trainable = trainable ? 1 : 0;
        pi.model = tm;
        int64_t& sz = trainable ? tr_size_[tm] : st_size_[tm];
        pi.off = sz;
        sz += (n + 3) / 4 * 4;           // 16-byte aligned tensor starts
        idx = (int)infos_[tm].size();
        infos_[tm].push_back(pi);
        index_[tm][name] = idx;
    } else {
        idx = it->second;
    }
    PRef r;
    if (dry_) return r;
    const ParamInfo& pi = infos_[tm][idx];
    if (pi.trainable) {
        r.p = buf_.params + tr_offset(model) + pi.off;
        if (model != M_OLD_POLICY) r.g = buf_.grads + tr_offset(model) + pi.off;
    } else {
        r.p = buf_.params + st_offset(model) + pi.off;
    }
    return r;
}

float* Learner::pw_transposed(const std::string& name, const float* w, int cin, int cout) {
    auto it = pwt_by_name_.find(name);
    if (it != pwt_by_name_.end()) return it->second;
    float* wt = alloc((size_t)cin * cout);
    pwt_by_name_[name] = wt;
    h_pwt_.push_back(PwTranspose{w, wt, cin, cout});
    pwt_tiles_ = std::max(pwt_tiles_, ((cin + 31) / 32) * ((cout + 31) / 32));
    return wt;
}

void Learner::build_fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (build_err_.empty()) build_err_ = buf;
}

float* Learner::pw_packed(const float* w, int K, int N, int sbk, int sbn, bool bf16) {
    float* wp = alloc((size_t)pw_packed_elems(N, K));
    h_pack_.push_back(pw_pack_entry(w, wp, K, N, sbk, sbn, bf16));
    return wp;
}

const void* Learner::pw_x3_packed(const float* w, int K, int N, int sbk, int sbn) {
    void* wp = alloc((size_t)pw_x3_packed_bytes_n(K, N) / sizeof(float));
    h_pack3_.push_back(pw_x3_pack_entry(w, wp, K, N, sbk, sbn));
    return wp;
}

int Learner::pw_fwd_nbpg(int G, int Mg, int N, int K) const {
    // float32 engine, K or N above 128 (stage 2): the forward runs on the one-tile-per-workgroup split-precision kernel
    // (gemm_pw_x3.hip pw_x3_wide_kernel), which writes one statistics row per 32-row tile
    if (pw_fwd_x3_wide(G, Mg, N, K)) return pw_x3_partial_rows(G, Mg, N, K);
    return pw_nn_plan(G, Mg, N, K).nbpg;
}

// backward-data of the same convs (N = conv input channels, K = conv output channels): pw_x3_wide_bwd_kernel, one part2 / part row per tile
// Small products only (M = 12288 rows at B = 256: the 3x4-pixel maps of stage 2): every workgroup streams its block of W^T once, so at
// M = 49152 (the stride-2 unit's first conv, on the 6x8 maps) the fragment traffic (3072 x 196 KB) outweighs what the persistent kernel
// loses to its serial tiles -- measured in the step: 140 us against 100 us there, 36 against 53 us (BatchNorm-sum epilogue) at M = 12288.
bool Learner::pw_bwd_x3_wide(int G, int Mg, int N, int K) const {
    static const bool on = !(cdrl_getenv("CDRL_PW_X3_WIDE_BWD") && atoi(cdrl_getenv("CDRL_PW_X3_WIDE_BWD")) == 0);
    return on && pw_fwd_x3_wide(G, Mg, N, K) && (int64_t)G * Mg <= 16384;
}

int Learner::pw_bwd_nbpg(int G, int Mg, int N, int K) const {
    if (pw_bwd_x3_wide(G, Mg, N, K)) return pw_x3_wide_bwd_rows(Mg);
    return pw_nn_plan(G, Mg, N, K).nbpg;
}

bool Learner::pw_fwd_x3_wide(int G, int Mg, int N, int K) const {
    static const bool x3_env = !(cdrl_getenv("CDRL_PW_X3") && atoi(cdrl_getenv("CDRL_PW_X3")) == 0);
    static const bool wide_env = !(cdrl_getenv("CDRL_PW_X3_WIDE") && atoi(cdrl_getenv("CDRL_PW_X3_WIDE")) == 0);
    // (measured at M = 12288 and 49152 rows: 16-20 against 26-36 us, ~60 against 80 us; beyond that every 32-row tile would still stream
    //  its own copy of W -- 196 KB per tile and column block -- and the persistent kernel keeps the shape)
    return cfg_.compute == 0 && x3_env && wide_env && (K > 128 || N > 128) && K <= 256 && N <= 256 && K % 4 == 0 && (int64_t)G * Mg <= 49152;
}

const void* Learner::gemm_x3_packed(const float* w, int K, int N, int sbk, int sbn) {
    void* wp = alloc((size_t)gemm_x3_packed_bytes(N, K) / sizeof(float));
    h_gpack_.push_back(gemm_x3_pack_entry(w, wp, K, N, sbk, sbn));
    return wp;
}

void Learner::note_bn_inference(const float* gamma, const float* beta, const float* mm, const float* mv, float* stats, int G, int C) {
    h_bninf_.push_back(BnInfEntry{gamma, beta, mm, mv, stats, G, C});
    if (C > bninf_max_c_) bninf_max_c_ = C;
}

int Learner::run_trunk_fwd(hipStream_t st, int training) {
    if (!training) CDRL_TRY(bn_inference_stats_many(d_bninf_, (int)h_bninf_.size(), bninf_max_c_, st));
    // this pass's weights in fragment order (fwd + bwd-data).  On the side stream beside the stem block, joined behind it: measured twice,
    // 14.29 vs 14.30 ms with round 5's events and 12.46 vs 12.41 ms with round 6's -- three 10-us launches beside the bandwidth-bound stem
    // conv plus a join packet cost what they save; they stay in front of the stem.
    // ... and in ONE launch, with the W^T transposes of the backward (training passes) riding along: four dependent 4-13 us launches per
    // pass became one (pack_all; the weights do not change between here and the backward)
    packs_have_wt_ = training != 0;
    CDRL_TRY(pack_all(d_gpack_, (int)h_gpack_.size(), d_pack_, (int)h_pack_.size(), d_pack3_, (int)h_pack3_.size(), d_pwt_,
                      training ? (int)h_pwt_.size() : 0, pwt_tiles_, st));
    return run_fwd(trunk_ops_, st, training);
}

void Learner::note_scratch(size_t part_d, size_t part2_d, size_t dy_f, size_t tn_f, size_t fpart_d) {
    if (part_d > max_part_) max_part_ = part_d;
    if (part2_d > max_part2_) max_part2_ = part2_d;
    if (dy_f > max_dy_) max_dy_ = dy_f;
    if (tn_f > max_tn_) max_tn_ = tn_f;
    if (fpart_d > max_fpart_) max_fpart_ = fpart_d;
}

// ------------------------------------------------------------------------------------------
// op builders
// ------------------------------------------------------------------------------------------
Learner::BnRec Learner::add_bn(std::vector<Op>& ops, int model, const std::string& prefix, View x, int G, int Mg, int C,
                               bool bessel, int act, View out, int out_shuffle, View dout, int dout_shuffle, float* dx,
                               int stats_nb, bool defer_apply, Passthrough pass) {
    PRef gamma = param(model, prefix + ".gamma", {C}, true);
    PRef beta = param(model, prefix + ".beta", {C}, true);
    PRef mm = param(model, prefix + ".moving_mean", {C}, false);
    PRef mv = param(model, prefix + ".moving_var", {C}, false);
    float* stats = alloc((size_t)4 * G * C);
    float* coef = alloc((size_t)3 * G * C);
    note_named(prefix + ".stats", stats, (size_t)4 * G * C * sizeof(float));
    if (x.ld == C && x.coff == 0) note_named(prefix + ".x", x.p, (size_t)G * Mg * C * (model == M_TRUNK && prefix.compare(0, 4, "img.") == 0 ? esz() : 4));
    const int nb = vcol_geom(Mg, C).nb;
    note_scratch((size_t)G * std::max(nb, stats_nb) * 2 * C, (size_t)G * nb * C, 0, 0);
    BnRec rec;
    rec.G = G;
    rec.Mg = Mg;
    rec.C = C;
    rec.nb = nb;
    rec.stats = stats;
    rec.coef = coef;
    rec.y = (x.ld == C && x.coff == 0) ? x.p : nullptr;
    rec.act = act;
    rec.reduce_fused = std::make_shared<bool>(false);
    rec.fin_by_consumer = std::make_shared<bool>(false);
    rec.part_ptr = &build_scr_->part;
    rec.dgamma = gamma.g;
    rec.dbeta = beta.g;
    std::shared_ptr<bool> fused = rec.reduce_fused, finc = rec.fin_by_consumer;
    const int bes = bessel ? 1 : 0;
    Scratch* sc = build_scr_;
    // trunk layers: the inference-mode statistics block comes from the batched launch at the start of the forward
    const bool inf_batched = model == M_TRUNK;
    if (inf_batched) note_bn_inference(gamma.p, beta.p, mm.p, mv.p, stats, G, C);
    // single-group BatchNorm over a few hundred rows (dense BNs of the trunk tail and the control branches): one launch per
    // direction instead of three
    static const bool small_env = true;
    const bool small = small_env && G == 1 && Mg <= 2048 && !bessel && act == ACT_NONE && !out_shuffle && !dout_shuffle && dx &&
                       !stats_nb && !defer_apply && !pass.fsrc.p && !pass.gsrc.p && !pass.gap_out;
    const bool gap = pass.gap_out != nullptr;
    const int at = model == M_TRUNK && prefix.compare(0, 4, "img.") == 0 ? at_ : 0;     // tower tensors only
    std::shared_ptr<int> diag_calls3 = std::make_shared<int>(0);
    if (at && small) build_fail("%s: single-launch BatchNorm has no bf16-storage form", prefix.c_str());
    if (gap && (pass.gap_rows <= 0 || Mg % pass.gap_rows != 0 || x.ld != C || x.coff != 0 || out_shuffle || dout_shuffle || pass.fsrc.p ||
                pass.gsrc.p))
        build_fail("%s: fused global average pool needs a dense input and whole frames per group", prefix.c_str());
    Op op;
    if (small) {
        op.fwd = [=](hipStream_t st, int training) -> int {
            if (training) return bn_small_fwd(x, Mg, C, gamma.p, beta.p, mm.p, mv.p, stats, out, st);
            if (!inf_batched) CDRL_TRY(bn_finalize(sc->part, nb, G, Mg, C, gamma.p, beta.p, mm.p, mv.p, bes, training, stats, st));
            return bn_apply(x, G, Mg, C, stats, act, out, out_shuffle, st);
        };
        op.bwd = [=](hipStream_t st) -> int { return bn_small_bwd(dout, x, Mg, C, stats, gamma.g, beta.g, coef, dx, st); };
        ops.push_back(op);
        return rec;
    }
    op.fwd = [=](hipStream_t st, int training) -> int {
        if (training && !stats_nb) CDRL_TRY(colstats(x, G, Mg, C, sc->part, st, at));
        if ((training || !inf_batched) && !((g_diag_skip_fin & 4) && stats_nb && ++*diag_calls3 > 3))
            CDRL_TRY(bn_finalize(sc->part, stats_nb ? stats_nb : nb, G, Mg, C, gamma.p, beta.p, mm.p, mv.p, bes, training, stats, st));
        if (gap) return bn_act_gap_fwd(x.p, stats, pass.gap_out, G, Mg / pass.gap_rows, pass.gap_rows, C, act, st, at);
        if (pass.fsrc.p) return bn_apply(x, G, Mg, C, stats, act, out, out_shuffle, st, &pass.fsrc, &pass.fdst, at);
        return bn_apply(x, G, Mg, C, stats, act, out, out_shuffle, st, nullptr, nullptr, at);
    };
    op.bwd = [=](hipStream_t st) -> int {
        // fused global average pool: the gradient source is the pooled gradient, one row per gap_rows rows of x
        const View dsrc = gap ? make_view(const_cast<float*>(pass.gap_dout), C) : dout;
        const int bc = gap ? pass.gap_rows : 0;
        if (pass.gsrc.p) CDRL_TRY(bn_bwd_reduce(dout, dout_shuffle, x, G, Mg, C, stats, act, sc->part, st, nullptr, &pass.gsrc, &pass.gdst, 0, at));
        else if (!*fused) CDRL_TRY(bn_bwd_reduce(dsrc, dout_shuffle, x, G, Mg, C, stats, act, sc->part, st, nullptr, nullptr, nullptr, bc, at));
        if (*finc) return 0;            // sums folded, applied and turned into dgamma / dbeta by the fused conv backward that follows
        CDRL_TRY(bn_bwd_finalize(sc->part, nb, G, Mg, C, stats, gamma.g, beta.g, coef, st));
        if (defer_apply) return 0;      // applied by the consumer GEMMs on load (PwFuse::bb)
        if (dx) return bn_bwd_apply(dsrc, dout_shuffle, x, G, Mg, C, stats, coef, act, dx, sc->part2, st, nullptr, bc, at);
        CDRL_TRY(next_slot(st));       // tower: dy + db partials go to a rotating scratch slot
        return bn_bwd_apply(dsrc, dout_shuffle, x, G, Mg, C, stats, coef, act, dys_[slot_], part2s_[slot_], st, nullptr, bc, at);
    };
    ops.push_back(op);
    return rec;
}

void Learner::add_pw(std::vector<Op>& ops, const std::string& prefix, View in, int rows, int Cin, int Cout, float* y,
                     View din, int din_acc, BnRec bn_after, PwFuse fuse) {
    PRef w = param(M_TRUNK, prefix + ".w", {1, 1, Cin, Cout}, true);
    PRef b = param(M_TRUNK, prefix + ".b", {Cout}, true);
    const int G = cfg_.T, Mg = rows / G;
    Scratch* sc = build_scr_;           // statistics / BatchNorm-sum partials: the scratch of the stream this op's forward runs on (the shortcut
                                        // branch of a stride-2 unit runs beside the main branch and has its own)
    static const bool pack_env = true;
    static const bool wt_env = true;
    const float* wt = (wt_env && !pack_env && fuse.bwd_pw) ? pw_transposed(prefix, w.p, Cin, Cout) : nullptr;
    const float* wb = wt ? wt : w.p;                    // backward-data operand B(k = cout, n = cin)
    const int wb_sk = wt ? Cin : 1, wb_sn = wt ? 1 : Cout;
    // compute mode 1 (configuration 3): every 1x1 convolution of the tower multiplies bf16-rounded operands -- forward,
    // backward-data and filter gradient: the fused kernels in their BF variant, the plain wide ones through gemm_x3's
    // single-plane form, the filter gradients through tn_direct's BF variant
    const bool bfc = cfg_.compute >= 1;
    const int at = at_;
    // forward on the bf16 matrix pipe (exact three-way operand split, gemm_pw_x3.hip) where the shape allows it
    static const bool x3_env = !(cdrl_getenv("CDRL_PW_X3") && atoi(cdrl_getenv("CDRL_PW_X3")) == 0);
    const bool x3_shape = (Cin <= 128 && Cout <= 128) || pw_fwd_x3_wide(G, Mg, Cout, Cin);
    const void* w3f = (!bfc && x3_env && fuse.fwd_pw && x3_shape && pw_x3_supported(in, Cout, Cin)) ? pw_x3_packed(w.p, Cin, Cout, Cout, 1) : nullptr;
    if (fuse.fwd_pw && pw_fwd_x3_wide(G, Mg, Cout, Cin) && !w3f && !dry_)
        build_fail("%s: the wide split-precision forward needs 16-byte aligned input rows (ld %d, offset %d)", prefix.c_str(), in.ld, in.coff);
    const int nb_fwd = pw_fwd_nbpg(G, Mg, Cout, Cin);
    // plain (unfused) wide convs -- the 464 -> 768 head conv, the 232-wide shortcut conv -- on the bf16 matrix pipe too (gemm_x3.hip)
    const bool wide = Cin >= 128 || Cout > 128;
    const bool g3 = bfc || (x3_env && wide);
    const bool use_g3f = g3 && !fuse.fwd_pw && gemm_x3_supported(in, Cin);
    const bool use_g3b = g3 && !fuse.bwd_pw && !fuse.bb && Cout % 4 == 0;
    const void* g3f = use_g3f ? gemm_x3_packed(w.p, Cin, Cout, Cout, 1) : nullptr;
    const void* g3b = use_g3b ? gemm_x3_packed(w.p, Cout, Cin, 1, Cout) : nullptr;
    if (bfc && ((!fuse.fwd_pw && !use_g3f) || (!fuse.bwd_pw && !fuse.bb && !use_g3b) || (fuse.bb && !fuse.bwd_pw)))
        build_fail("bf16-operand mode: 1x1 convolution %s (%d -> %d, fwd %d bwd %d bb %d; input ld %d coff %d) has no bf16 kernel",
                   prefix.c_str(), Cin, Cout, (int)fuse.fwd_pw, (int)fuse.bwd_pw, (int)fuse.bb, in.ld, in.coff);
    if (bfc && !pack_env) build_fail("bf16-operand mode needs the packed-weight path (CDRL_PW_PACK=0 is set)");
    const float* wpf = (pack_env && fuse.fwd_pw && !w3f) ? pw_packed(w.p, Cin, Cout, Cout, 1, bfc) : nullptr;      // forward: B(k = cin, n = cout)
    const float* wpb = (pack_env && fuse.bwd_pw) ? pw_packed(w.p, Cout, Cin, 1, Cout, bfc) : nullptr;      // backward-data: W^T
    const int tn_groups = (fuse.pro_stats || fuse.bb) ? G : 1;
    note_scratch(0, 0, (size_t)rows * Cout, (size_t)gemm_tn_part_elems(rows, Cout, Cin, tn_groups));
    if (fuse.epi_stats) note_scratch((size_t)G * nb_fwd * 2 * Cout, 0, 0, 0);
    if (fuse.bwd_ey) note_scratch((size_t)G * pw_bwd_nbpg(G, Mg, Cin, Cout) * 2 * Cin, 0, 0, 0);
    Op op;
    op.fwd = [=](hipStream_t st, int) -> int {
        if (w3f)
            return pw_x3(in, fuse.pro_stats, w3f, b.p, make_view(y, Cout), G, Mg, Cout, Cin, fuse.epi_stats ? sc->part : nullptr, st,
                         nb_fwd);
        if (fuse.fwd_pw)
            return pw_nn(in, fuse.pro_stats, w.p, Cout, 1, b.p, make_view(y, Cout), 0, G, Mg, Cout, Cin, fuse.epi_stats ? 1 : 0,
                         nullptr, nullptr, sc->part, st, nullptr, wpf, bfc, at);
        if (g3f) return gemm_x3(in, g3f, b.p, make_view(y, Cout), rows, Cout, Cin, 0, st, bfc, at);
        return gemm_nn(in, w.p, Cout, 1, b.p, make_view(y, Cout), rows, Cout, Cin, 0, st);
    };
    const int nbp_bwd = fuse.bb ? pw_bwd_nbpg(G, Mg, Cin, Cout) : 0;
    if (fuse.bb) note_scratch(0, (size_t)G * nbp_bwd * Cout, 0, 0);
    // One kernel for backward-data + filter gradient + bias gradient (+ the backward sums of the BatchNorm in front of the conv):
    // float32 engine and bf16 activation storage (not the operand-only mode), both channel counts padded alike (gemm_pw_bwd.hip)
    const View dz_probe = fuse.bb_dz.p ? fuse.bb_dz : make_view(reinterpret_cast<float*>(uintptr_t(16)), Cout);
    const bool anorm = fuse.pro_stats != nullptr;
    // 24 input channels -- the first unit, the last conv of the backward -- pad to 64: 158 us against 104 us for the backward-data kernel of
    // the two-kernel form, which is why rounds 4 kept it there; but that form's filter gradient (163 us on the side stream) then bounds the
    // tail of the pass and slows the BatchNorm reduction beside it (117 us instead of 29): fused 14.00 vs 14.06 ms per update-step, and one
    // pass over (dz, y) = 0.3 GB per pass less.  CDRL_FBWD_MIN_CIN=32 -> the two-kernel form for that conv.
    static const int fbwd_min_cin = 24;
    const bool fbwd = fused_bwd_ && fuse.bb && fuse.bwd_pw && (!bfc || at) && G <= 8 && Cin >= fbwd_min_cin && pw_bwd_fused_supported(dz_probe, in, din, Cout, Cin, at) &&
                      (!anorm || (fuse.bwd_ey == in.p && fuse.bwd_epi_stats == fuse.pro_stats && fuse.a_bn && in.ld == Cin && in.coff == 0)) &&
                      (anorm || !fuse.bwd_ey);
    // 232-channel convs (stage 2), float32: backward-data with the BatchNorm-backward prologue on the one-tile-per-workgroup
    // split-precision kernel (round 6; the filter gradient stays on the side stream)
    const bool wbw = !bfc && !at && fuse.bb && fuse.bwd_pw && !fbwd && pw_bwd_x3_wide(G, Mg, Cin, Cout);
    if (wbw && !pw_x3_wide_bwd_supported(dz_probe, din, Cin, Cout, fuse.bb_shuffle))
        build_fail("%s: the wide split-precision backward needs even / 16-byte aligned gradient rows (ld %d, offset %d, shuffle %d)", prefix.c_str(),
                   dz_probe.ld, dz_probe.coff, fuse.bb_shuffle);
    const void* wpx = (fbwd || wbw) ? pw_x3_packed(w.p, Cout, Cin, 1, Cout) : nullptr;      // W^T planes: B(k = cout, n = cin)
    if (fbwd) {
        max_qpart_ = std::max(max_qpart_, (size_t)pw_bwd_fused_qpart_elems(G, Mg, Cout, Cin, at));
        max_dbpart_ = std::max(max_dbpart_, (size_t)pw_bwd_fused_dbpart_elems(G, Mg, Cout, Cin, at));
        if (fuse.a_bn_done) *fuse.a_bn_done = true;
    }
    // float32: the finalize of the BatchNorm behind the conv inside the fused kernel (no bn_bwd_finalize launch in front of it)
    const bool fin = fbwd && !at && fin_on_load_ && fuse.bb_fin;
    if (fin && fuse.bb_fin_done) *fuse.bb_fin_done = true;
    if (fin && !dry_ && (!fuse.bb_fin_part || fuse.bb_fin_nb <= 0)) build_fail("%s: finalize on load without the BatchNorm's scratch block", prefix.c_str());
    op.bwd = [=](hipStream_t st) -> int {
        if (fbwd) {
            if (fuse.bb_claim_slot) CDRL_TRY(next_slot(st));
            PwBwdFused f;
            f.dz = fuse.bb_dz.p ? fuse.bb_dz : make_view(dys_[slot_], Cout);
            f.dz_shuffle = fuse.bb_shuffle;
            f.act = fuse.bb_act;
            f.y = y;
            f.stats = fuse.bb_stats;
            f.coef = fuse.bb_coef;
            f.a = in;
            f.a_stats = anorm ? fuse.pro_stats : nullptr;
            f.a_gamma = fuse.a_gamma;
            f.a_beta = fuse.a_beta;
            f.a_dgamma = fuse.a_dgamma;
            f.a_dbeta = fuse.a_dbeta;
            f.a_coef = fuse.a_coef;
            f.W = w.p;
            f.Wp = wpx;
            f.da = din;
            f.accumulate = din_acc;
            f.dW = w.g;
            f.db = b.g;
            CDRL_TRY(next_q(st));
            const int qi = qi_;
            f.qpart = qparts_[qi];
            f.dbpart = dbparts_[qi];
            if (fin) {
                f.fin_part = *fuse.bb_fin_part;
                f.fin_nb = fuse.bb_fin_nb;
                f.fin_tot = fintots_[qi];
                f.o_dgamma = fuse.bb_dgamma;
                f.o_dbeta = fuse.bb_dbeta;
            }
            f.G = G;
            f.Mg = Mg;
            f.N = Cout;
            f.K = Cin;
            f.at = at;
            CDRL_TRY(pw_bwd_fused(f, st));
            // the reduce also finalizes the BatchNorm in front of the conv (its coefficients are the next kernel's input): critical
            // stream; without one it only produces weight gradients -> side stream, flushed once per unit
            if (anorm) return pw_bwd_fused_reduce(f, st);
            CDRL_TRY(defer_side(st, [=](hipStream_t sd) -> int {
                CDRL_TRY(pw_bwd_fused_reduce(f, sd));
                if (side_enabled_ && sd == side_) {     // the buffer pair is free again once this reduce has run
                    CDRL_HIP(hipEventRecord(ev_q_[qi], sd));
                    q_seq_[qi] = note_side_record(ev_q_[qi]);
                    q_used_[qi] = true;
                }
                return 0;
            }));
            return flush_side(st);
        }
        if (fuse.bb) {
            // BN-backward apply fused into the operand loads: dz (+ raw y, statistics, coefficients) instead of dy
            if (fuse.bb_claim_slot) CDRL_TRY(next_slot(st));
            const View dz = fuse.bb_dz.p ? fuse.bb_dz : make_view(dys_[slot_], Cout);
            hipStream_t side = fork_side(st);
            TnBnBwd tb{y, fuse.bb_stats, fuse.bb_coef, fuse.bb_shuffle, fuse.bb_act};
            CDRL_TRY(gemm_tn(in, dz, w.g, rows, Cout, Cin, tns_[slot_], 0, side, G, fuse.pro_stats, &tb, bfc, at));
            CDRL_TRY(done_side(side));
            PwBnBwd pb{y, fuse.bb_stats, fuse.bb_coef, fuse.bb_shuffle, fuse.bb_act, part2s_[slot_]};
            if (wbw)
                CDRL_TRY(pw_x3_wide_bwd(dz, pb, wpx, din, din_acc, G, Mg, Cin, Cout, fuse.bwd_ey, fuse.bwd_epi_stats,
                                        fuse.bwd_ey ? sc->part : nullptr, st));
            else
                CDRL_TRY(pw_nn(dz, nullptr, wb, wb_sk, wb_sn, nullptr, din, din_acc, G, Mg, Cin, Cout, fuse.bwd_ey ? 2 : 0, fuse.bwd_ey,
                               fuse.bwd_epi_stats, sc->part, st, &pb, wpb, bfc, at));
            // bias gradient = column sums of the (virtual) dy, reduced from the GEMM's partials: rides on the next fork
            double* p2 = part2s_[slot_];
            return defer_side(st, [=](hipStream_t sd) -> int { return reduce_partials(p2, G * nbp_bwd, Cout, Cout, b.g, 0, sd); });
        }
        float* dy = dys_[slot_];
        // side stream: bias gradient (column sums of dy, reduced per block by bn_bwd_apply) + filter gradient
        hipStream_t side = fork_side(st);
        CDRL_TRY(reduce_partials(part2s_[slot_], bn_after.G * bn_after.nb, Cout, Cout, b.g, 0, side));
        CDRL_TRY(gemm_tn(in, make_view(dy, Cout), w.g, rows, Cout, Cin, tns_[slot_], 0, side, tn_groups, fuse.pro_stats, nullptr, bfc, at));
        CDRL_TRY(done_side(side));
        // main stream: the critical path to the previous layer
        if (din.p) {
            if (fuse.bwd_pw)
                return pw_nn(make_view(dy, Cout), nullptr, wb, wb_sk, wb_sn, nullptr, din, din_acc, G, Mg, Cin, Cout,
                             fuse.bwd_ey ? 2 : 0, fuse.bwd_ey, fuse.bwd_epi_stats, sc->part, st, nullptr, wpb, bfc, at);
            if (g3b) return gemm_x3(make_view(dy, Cout), g3b, nullptr, din, rows, Cin, Cout, din_acc, st, bfc, at);
            CDRL_TRY(gemm_nn(make_view(dy, Cout), w.p, 1, Cout, nullptr, din, rows, Cin, Cout, din_acc, st));
        }
        return 0;
    };
    ops.push_back(op);
}

void Learner::add_dw(std::vector<Op>& ops, const std::string& prefix, View in, int N, int H, int W, int C, int stride,
                     float* y, View din, int din_acc, const BnRec* pre_bn) {
    PRef w = param(M_TRUNK, prefix + ".w", {3, 3, C, 1}, true);
    PRef b = param(M_TRUNK, prefix + ".b", {C}, true);
    const int Ho = same_out_h(H, stride), Wo = same_out_h(W, stride);
    note_scratch(0, 0, (size_t)N * Ho * Wo * C, 0, (size_t)dw_bwd_part_elems(N, H, W, C, stride));
    static const bool fuse_env = true;
    const bool fuse = fuse_env && pre_bn && pre_bn->y && din_acc == 0 && pre_bn->C == C && pre_bn->Mg * pre_bn->G == N * H * W;
    BnRec pre;
    if (fuse) {
        pre = *pre_bn;
        *pre.reduce_fused = true;
    }
    if (at_) build_fail("%s: the unfused depthwise path has no bf16-storage form (CDRL_FUSED_DW=0 is set)", prefix.c_str());
    Op op;
    op.fwd = [=](hipStream_t st, int) -> int { return dw_fwd(in, w.p, b.p, y, N, H, W, C, stride, st); };
    op.bwd = [=](hipStream_t st) -> int {
        float* dy = dys_[slot_];
        hipStream_t side = fork_side(st);
        CDRL_TRY(dw_bwd_filter(in, dy, w.g, b.g, N, H, W, C, stride, fparts_[slot_], side));
        CDRL_TRY(done_side(side));
        if (fuse)      // bwd-data + BN-backward sums of the layer that produced `in` in one pass over da
            return dw_bwd_data_bnreduce(dy, w.p, din, N, H, W, C, stride, pre.G, pre.y, pre.stats, pre.act, scr_main_.part, st);
        return dw_bwd_data(dy, w.p, din, N, H, W, C, stride, din_acc, st);
    };
    ops.push_back(op);
}

float* Learner::add_dw_block(std::vector<Op>& ops, const std::string& unit, const char* bn_pre, const char* dw,
                             const char* bn_post, float* x, int H, int W, int C, int stride, float* y2, View out, View dout,
                             View din, int pre_stats_nb, bool post_apply, int post_bwd_nb, float* stats1_ext, float* coef1_ext,
                             bool pre_defer_apply, float** coef2_out, std::shared_ptr<bool> post_bwd_done,
                             std::shared_ptr<bool> pre_fin_done) {
    const int B = cfg_.B, G = cfg_.T, N = B * G;
    const int Ho = same_out_h(H, stride), Wo = same_out_h(W, stride);
    const int Mi = B * H * W, Mo = B * Ho * Wo;
    const bool pre = bn_pre != nullptr;
    Scratch* sc = build_scr_;           // the shortcut branch of a stride-2 unit has its own partial buffers (it runs on the side stream)
    PRef g1, b1, mm1, mv1;
    float *stats1 = nullptr, *coef1 = nullptr;
    if (pre) {
        const std::string n1 = unit + "." + bn_pre;
        g1 = param(M_TRUNK, n1 + ".gamma", {C}, true);
        b1 = param(M_TRUNK, n1 + ".beta", {C}, true);
        mm1 = param(M_TRUNK, n1 + ".moving_mean", {C}, false);
        mv1 = param(M_TRUNK, n1 + ".moving_var", {C}, false);
        stats1 = stats1_ext ? stats1_ext : alloc((size_t)4 * G * C);
        note_bn_inference(g1.p, b1.p, mm1.p, mv1.p, stats1, G, C);
        coef1 = coef1_ext ? coef1_ext : alloc((size_t)3 * G * C);
        note_named(n1 + ".stats", stats1, (size_t)4 * G * C * sizeof(float));
        note_named(n1 + ".x", x, (size_t)N * H * W * C * esz());
    }
    PRef w = param(M_TRUNK, unit + "." + dw + ".w", {3, 3, C, 1}, true);
    PRef b = param(M_TRUNK, unit + "." + dw + ".b", {C}, true);
    const std::string n2 = unit + "." + bn_post;
    PRef g2 = param(M_TRUNK, n2 + ".gamma", {C}, true);
    PRef b2 = param(M_TRUNK, n2 + ".beta", {C}, true);
    PRef mm2 = param(M_TRUNK, n2 + ".moving_mean", {C}, false);
    PRef mv2 = param(M_TRUNK, n2 + ".moving_var", {C}, false);
    float* stats2 = alloc((size_t)4 * G * C);
    note_bn_inference(g2.p, b2.p, mm2.p, mv2.p, stats2, G, C);
    float* coef2 = alloc((size_t)3 * G * C);
    if (coef2_out) *coef2_out = coef2;
    const int nb_in = vcol_geom(Mi, C).nb, nb_out = vcol_geom(Mo, C).nb;
    const int nbf = dwf_geom(B, G, H, W, C, stride).nb;              // partial rows of the forward kernel (BN2 statistics) ...
    const int nbb = dwf_geom(B, G, H, W, C, stride).nb_bwd;          // ... and of the backward (BN1 sums, filter partials)
    const size_t nbmax = (size_t)std::max(std::max(std::max(nb_in, nb_out), std::max(nbf, nbb)), std::max(pre_stats_nb, post_bwd_nb));
    note_scratch((size_t)G * nbmax * 2 * C, (size_t)G * nbmax * C, (size_t)N * H * W * C, 0,
                 (size_t)dwf_filter_part_elems(B, G, H, W, C, stride));
    const View xv = make_view(x, C), y2v = make_view(y2, C);
    const int at = at_;
    std::shared_ptr<int> diag_calls = std::make_shared<int>(0);

    if (pre) {      // BN1: statistics only in the forward; backward = finalize of the sums the depthwise op produced
        Op op;
        op.fwd = [=](hipStream_t st, int training) -> int {
            if (!training) return 0;                            // inference: statistics block from the batched launch
            if (!pre_stats_nb) CDRL_TRY(colstats(xv, G, Mi, C, sc->part, st, at));
            if ((g_diag_skip_fin & 1) && ++*diag_calls > 6) return 0;      // timing diagnostic: stale statistics of an earlier step
            return bn_finalize(sc->part, pre_stats_nb ? pre_stats_nb : nb_in, G, Mi, C, g1.p, b1.p, mm1.p, mv1.p, 1, training,
                               stats1, st);
        };
        op.bwd = [=](hipStream_t st) -> int {
            if (pre_fin_done && *pre_fin_done) return 0;   // folded by the fused backward of the 1x1 conv in front (finalize on load)
            CDRL_TRY(bn_bwd_finalize(sc->part, nbb, G, Mi, C, stats1, g1.g, b1.g, coef1, st));
            if (pre_defer_apply) return 0;                 // the 1x1 conv in front applies it on load (PwFuse::bb)
            const float* dz = dys_[slot_];                 // masked gradient left there by the depthwise op
            CDRL_TRY(next_slot(st));
            return bn_bwd_apply(make_view(const_cast<float*>(dz), C), 0, xv, G, Mi, C, stats1, coef1, ACT_NONE, dys_[slot_],
                                part2s_[slot_], st, nullptr, 0, at);
        };
        ops.push_back(op);
    }
    {
        Op op;
        op.fwd = [=](hipStream_t st, int) -> int {
            return dwf_fwd(x, stats1, w.p, b.p, y2, sc->part, G, B, H, W, C, stride, st, at);
        };
        op.bwd = [=](hipStream_t st) -> int {
            CDRL_TRY(next_slot(st));
            double* pw = fparts_[slot_];
            const View dx = pre ? make_view(dys_[slot_], C) : din;
            CDRL_TRY(dwf_bwd(x, stats1, dout.p, y2, stats2, coef2, w.p, dx, sc->part, pw, G, B, H, W, C, stride, st, at));
            return defer_side(st, [=](hipStream_t sd) -> int {
                return reduce_partials2(pw, G * nbb, 9 * C, C, (int64_t)10 * C, w.g, b.g, 0, sd);
            });
        };
        ops.push_back(op);
    }
    {               // BN2: finalize + apply in the forward; backward = sums + coefficients only (applied by the dw op)
        Op op;
        op.fwd = [=](hipStream_t st, int training) -> int {
            if (training && !((g_diag_skip_fin & 2) && ++*diag_calls > 6))
                CDRL_TRY(bn_finalize(sc->part, nbf, G, Mo, C, g2.p, b2.p, mm2.p, mv2.p, 1, training, stats2, st));
            if (!post_apply) return 0;
            return bn_apply(y2v, G, Mo, C, stats2, ACT_NONE, out, 0, st, nullptr, nullptr, at);
        };
        op.bwd = [=](hipStream_t st) -> int {
            if (post_bwd_done && *post_bwd_done) return 0;      // dgamma, dbeta, coefficients came out of the consumer conv's reduce kernel
            if (!post_bwd_nb) CDRL_TRY(bn_bwd_reduce(dout, 0, y2v, G, Mo, C, stats2, ACT_NONE, sc->part, st, nullptr, nullptr, nullptr, 0, at));
            return bn_bwd_finalize(sc->part, post_bwd_nb ? post_bwd_nb : nb_out, G, Mo, C, stats2, g2.g, b2.g, coef2, st);
        };
        ops.push_back(op);
    }
    return stats2;
}

void Learner::add_dense(std::vector<Op>& ops, int model, const std::string& prefix, View in, int M, int K, int N,
                        int act, View out, View dout, View din, int din_acc, bool need_din, const char*) {
    PRef w = param(model, prefix + ".w", {K, N}, true);
    PRef b = param(model, prefix + ".b", {N}, true);
    float *z = nullptr, *dz = nullptr;
    if (act != ACT_NONE) {
        z = alloc((size_t)M * N);
        dz = alloc((size_t)M * N);
        note_named(prefix + ".z", z, (size_t)M * N * sizeof(float));
    }
    note_scratch((size_t)vcol_geom(M, N).nb * N, (size_t)vcol_geom(M, N).nb * N, 0, (size_t)gemm_tn_part_elems(M, N, K));
    const int nb = vcol_geom(M, N).nb;
    Scratch* sc = build_scr_;
    const size_t dskn = (size_t)std::max(gemm_nn_splitk_elems(M, N, K), gemm_nn_splitk_elems(M, K, N));
    float* dsk = dskn ? alloc(dskn) : nullptr;          // split-K scratch (batch-sized M, K >= 256)
    Op op;
    op.fwd = [=](hipStream_t st, int) -> int {
        if (act == ACT_NONE) return gemm_nn(in, w.p, N, 1, b.p, out, M, N, K, 0, st, dsk);
        bool act_done = false;      // (split-K layers: the activation rides in the slab reduce)
        const bool dense_out = out.ld == N && out.coff == 0;
        CDRL_TRY(gemm_nn(in, w.p, N, 1, b.p, make_view(z, N), M, N, K, 0, st, dsk, dense_out ? out.p : nullptr, act, &act_done));
        if (act_done) return 0;
        return act_fwd(z, out.p, (int64_t)M * N, act, st);
    };
    op.bwd = [=](hipStream_t st) -> int {
        View dzv = dout;
        if (act != ACT_NONE) {
            CDRL_TRY(act_bwd(z, dout.p, dz, (int64_t)M * N, act, st));
            dzv = make_view(dz, N);
        }
        // critical path first, then the weight / bias gradients on the side stream (main-stream layers only)
        if (need_din) CDRL_TRY(gemm_nn(dzv, w.p, 1, N, nullptr, din, M, K, N, din_acc, st, dsk));
        if (sc == &scr_main_) {
            CDRL_TRY(next_slot(st));
            hipStream_t side = fork_side(st);
            CDRL_TRY(colsum(dzv, M, N, part2s_[slot_], side));
            CDRL_TRY(reduce_partials(part2s_[slot_], nb, N, N, b.g, 0, side));
            CDRL_TRY(gemm_tn(in, dzv, w.g, M, N, K, tns_[slot_], 0, side));
            return done_side(side);
        }
        CDRL_TRY(colsum(dzv, M, N, sc->part, st));
        CDRL_TRY(reduce_partials(sc->part, nb, N, N, b.g, 0, st));
        return gemm_tn(in, dzv, w.g, M, N, K, sc->tn, 0, st);
    };
    ops.push_back(op);
}

void Learner::add_gru(std::vector<Op>& ops, const std::string& name, Tens& x, int In, int u, View out, View dout,
                      bool need_dx) {
    const int B = cfg_.B, T = cfg_.T, U3 = 3 * u;
    PRef Kp = param(M_TRUNK, name + ".kernel", {In, U3}, true);
    PRef Rp = param(M_TRUNK, name + ".recurrent", {u, U3}, true);
    PRef bp = param(M_TRUNK, name + ".bias", {2, U3}, true);
    float* XP = alloc((size_t)T * B * U3);
    float* HP = alloc((size_t)T * B * U3);
    float* Z = alloc((size_t)T * B * u);
    float* R = alloc((size_t)T * B * u);
    float* HH = alloc((size_t)T * B * u);
    float* Hs = alloc((size_t)(T + 1) * B * u);         // Hs[0] = h_{-1} = 0: zeroed once at bind, never written
    float* dXP = alloc((size_t)T * B * U3);
    float* dHP = alloc((size_t)T * B * U3);
    float* dHa = alloc((size_t)B * u);
    float* dHb = alloc((size_t)B * u);
    if (!dry_) zero_once_.push_back(std::make_pair(Hs, (size_t)B * u * sizeof(float)));
    if (!gru_step_supported(u)) build_fail("GRU units %d unsupported by the fused step kernels", u);
    const float* RT = pw_transposed(name + ".recurrent", Rp.p, u, U3);        // [3u][u], refreshed with the conv W^T copies
    note_scratch((size_t)vcol_geom(T * B, U3).nb * U3, (size_t)vcol_geom(T * B, U3).nb * U3, 0,
                 (size_t)std::max(gemm_tn_part_elems(T * B, U3, In), gemm_tn_part_elems(T * B, U3, u)));
    const int nbc = vcol_geom(T * B, U3).nb;
    View xv = x.v();
    View xg = x.gv();
    Scratch* sc = build_scr_;
    // split-K scratch for the input projections (M = T*B, K = In / 3u): own buffer per GRU, the GRUs of different
    // modalities run on different streams
    size_t skn = 0;
    for (int64_t e : {gemm_nn_splitk_elems(T * B, U3, In), gemm_nn_splitk_elems(T * B, In, U3)}) skn = std::max(skn, (size_t)e);
    float* sk = skn ? alloc(skn) : nullptr;
    Op op;
    op.fwd = [=](hipStream_t st, int) -> int {
        CDRL_TRY(gemm_nn(xv, Kp.p, U3, 1, bp.p, make_view(XP, U3), T * B, U3, In, 0, st, sk));
        for (int t = 0; t < T; ++t)         // one fused kernel per step: h R + b1, gates, saved tensors, (last step) the concat slot
            CDRL_TRY(gru_step_fwd(XP + (size_t)t * B * U3, Hs + (size_t)t * B * u, Rp.p, bp.p + U3, Z + (size_t)t * B * u,
                                  R + (size_t)t * B * u, HH + (size_t)t * B * u, HP + (size_t)t * B * U3,
                                  Hs + (size_t)(t + 1) * B * u, t == T - 1 ? out : View{nullptr, 0, 0}, B, u, st));
        return 0;
    };
    op.bwd = [=](hipStream_t st) -> int {
        View cur = dout;                    // the gradient enters through the last hidden state only
        float* nxt = dHa;
        for (int t = T - 1; t >= 0; --t) {
            CDRL_TRY(gru_step_bwd(cur, Z + (size_t)t * B * u, R + (size_t)t * B * u, HH + (size_t)t * B * u,
                                  HP + (size_t)t * B * U3, Hs + (size_t)t * B * u, RT, dXP + (size_t)t * B * U3,
                                  dHP + (size_t)t * B * U3, t > 0 ? nxt : nullptr, B, u, st));
            cur = make_view(nxt, u);
            nxt = nxt == dHa ? dHb : dHa;
        }
        // critical path first: the gradient w.r.t. the GRU input
        if (need_dx) CDRL_TRY(gemm_nn(make_view(dXP, U3), Kp.p, 1, U3, nullptr, xg, T * B, In, U3, 0, st, sk));
        // weight / bias gradients: off the critical path -> side stream with a rotating scratch slot (main-stream GRU only;
        // the small-modality GRUs already run on the aux stream with their own scratch)
        hipStream_t ws = st;
        float* tn = sc->tn;
        double* part = sc->part;
        const bool on_side = sc == &scr_main_;
        if (on_side) {
            CDRL_TRY(next_slot(st));
            ws = fork_side(st);
            tn = tns_[slot_];
            part = part2s_[slot_];
        }
        CDRL_TRY(gemm_tn(xv, make_view(dXP, U3), Kp.g, T * B, U3, In, tn, 0, ws));
        CDRL_TRY(colsum(make_view(dXP, U3), T * B, U3, part, ws));
        CDRL_TRY(reduce_partials(part, nbc, U3, U3, bp.g, 0, ws));
        CDRL_TRY(gemm_tn(make_view(Hs, u), make_view(dHP, U3), Rp.g, T * B, U3, u, tn, 0, ws));
        CDRL_TRY(colsum(make_view(dHP, U3), T * B, U3, part, ws));
        CDRL_TRY(reduce_partials(part, nbc, U3, U3, bp.g + U3, 0, ws));
        if (on_side) CDRL_TRY(done_side(ws));
        return 0;
    };
    ops.push_back(op);
}

// The road / vehicle / navigation feature nets and their GRUs are ~150 latency-bound launches that
// do not depend on the image tower: they are enqueued on the side stream at the start of the forward
// (fork) and joined right before the concat BatchNorm; in the backward they are forked as soon as the
// gradient of the concat exists and run under the tower's backward.
void Learner::add_aux_fork(std::vector<Op>& ops) {
    Op op;
    static const int diag_skip_aux = cdrl_getenv("CDRL_DIAG_SKIP_AUX") ? atoi(cdrl_getenv("CDRL_DIAG_SKIP_AUX")) : 0;     // timing diagnostics only (wrong results)
    op.fwd = [=](hipStream_t st, int training) -> int {
        if (diag_skip_aux & 1) return 0;
        if (!side_enabled_) return run_fwd(aux_ops_, st, training);
        hipEvent_t evf = nullptr;
        CDRL_TRY(mark_stream(st, ev_aux_fork_, &evf));        // parameters / inputs produced on the main stream
        if (aux_worker_ && !graphs_enabled_) {
            CDRL_TRY(aux_wait());
            aux_worker_->submit([this, training, evf]() -> int {
                CDRL_HIP(hipStreamWaitEvent(aux_, evf, 0));
                CDRL_TRY(run_fwd(aux_ops_, aux_, training));
                CDRL_HIP(hipEventRecord(ev_aux_done_, aux_));
                return 0;
            });
            aux_inflight_ = true;
            return 0;
        }
        CDRL_HIP(hipStreamWaitEvent(aux_, evf, 0));
        CDRL_TRY(run_fwd(aux_ops_, aux_, training));
        CDRL_HIP(hipEventRecord(ev_aux_done_, aux_));
        return 0;
    };
    op.bwd = [](hipStream_t) -> int { return 0; };
    ops.push_back(op);
}

void Learner::add_aux_join(std::vector<Op>& ops) {
    Op op;
    op.fwd = [=](hipStream_t st, int) -> int {
        if (side_enabled_) {
            CDRL_TRY(aux_wait());               // the worker has recorded ev_aux_done_
            CDRL_HIP(hipStreamWaitEvent(st, ev_aux_done_, 0));
        }
        return 0;
    };
    static const int diag_skip_aux = cdrl_getenv("CDRL_DIAG_SKIP_AUX") ? atoi(cdrl_getenv("CDRL_DIAG_SKIP_AUX")) : 0;     // timing diagnostics only (wrong results)
    op.bwd = [=](hipStream_t st) -> int {
        if (diag_skip_aux & 2) return 0;
        if (!side_enabled_) return run_bwd(aux_ops_, st);
        // Own stream: ~90 tiny dependent kernels (0.8 ms).  On the filter-gradient side stream they blocked, in stream
        // order, the slot events the main stream waits on (measured: a 0.84 ms hole in the critical stream per pass).
        hipEvent_t evf = nullptr;
        CDRL_TRY(mark_stream(st, ev_aux_fork_, &evf));        // gradient of the concat is ready
        aux_pending_ = true;                                  // joined by join_side() at the end of the backward
        if (aux_worker_ && !graphs_enabled_) {
            CDRL_TRY(aux_wait());
            aux_worker_->submit([this, evf]() -> int {
                CDRL_HIP(hipStreamWaitEvent(aux_, evf, 0));
                CDRL_TRY(run_bwd(aux_ops_, aux_));
                CDRL_HIP(hipEventRecord(ev_aux_done_, aux_));
                return 0;
            });
            aux_inflight_ = true;
            return 0;
        }
        CDRL_HIP(hipStreamWaitEvent(aux_, evf, 0));
        CDRL_TRY(run_bwd(aux_ops_, aux_));
        CDRL_HIP(hipEventRecord(ev_aux_done_, aux_));
        return 0;
    };
    ops.push_back(op);
}

// ------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------
void Learner::build_trunk(std::vector<Op>& ops) {
    const Config& c = cfg_;
    const int B = c.B, T = c.T, N = B * T;
    const int Hs = (c.H - 3) / 2 + 1, Ws = (c.W - 3) / 2 + 1;
    auto bnrec = [](int G, int Mg, int C) {
        BnRec r;
        r.G = G;
        r.Mg = Mg;
        r.C = C;
        r.nb = vcol_geom(Mg, C).nb;
        return r;
    };
    aux_ops_.clear();
    add_aux_fork(ops);
    {
        const char* e = cdrl_getenv("CDRL_FUSED_DW");        // 0 -> unfused bn-apply / depthwise / stats kernels
        fused_dw_ = !(e && atoi(e) == 0);
        const char* e2 = cdrl_getenv("CDRL_FUSED_PW");       // 0 -> generic tiled GEMM + separate BN passes around the 1x1 convs
        fused_pw_ = !(e2 && atoi(e2) == 0);
        fused_pw_wide_ = !(e2 && atoi(e2) == 1);        // K, N = 232 (stage-2 units) on the fused path too; 1 -> narrow layers only
        // BN-backward apply as GEMM operand prologue: bit 0 -> for the unit's first 1x1 conv (bn1), bit 1 -> for the second
        // (bn3, gathered through the shuffle map); 0 -> separate apply passes with a materialised dy
        const char* e3 = cdrl_getenv("CDRL_FUSED_BB");
        // measured at v19: 1 -> 25.5, 0 -> 25.8, 3 -> 26.0, 2 -> 26.2 ms/update-step; re-measured at v29 (buffer-load filter
        // gradient: the shuffle gather costs nothing there any more): 3 -> 20.66, 1 -> 20.80, 0 -> 21.0, 2 -> 21.3, and with
        // the wide fused pointwise path 3 -> 20.21.  Bit 2: also for the first unit's conv with 24 input channels -- slower before
        // the accumulate variant prefetched its old output tile (248 us against 165 us for apply + plain GEMM), now 15.91 vs 15.96
        // ms/update-step and one 164 MB tensor less
        fused_bb_ = e3 ? atoi(e3) : 7;
        // (round 4's LDS-resident 64-row panel form of the 232-channel forward convs, gemm_pw_wide.hip, is gone: 18 vs 26 us per launch
        //  isolated but 14.43 vs 14.48 ms per update-step, and -- like any change of a float32 forward -- it re-drew the ReLU6 decisions
        //  and moved smoke()'s worst tensor from 6.9e-5 to 9.5e-5 of the 1e-4 gate, both times it was measured: DESIGN.md section 3)
        // The fused conv backward folds the backward sums of the BatchNorm behind it in its own prologue (no bn_bwd_finalize launch in
        // front: 46 fewer critical-stream launches per update-step, 614 -> 568; losses bit-identical over 155 update-steps).  Round 4
        // measured it neutral (-0.07 ms) with 128-256 partial rows per time slice -- every one of the 64 workgroups of a slice reads all
        // of them from L2 -- and kept it opt-in; with 64 rows from the strip-form depthwise backward and 128 from the BatchNorm reductions
        // (NB_STATS) it is worth 0.16 ms per update-step (14.15 vs 14.31 ms, same box) and is the default.  CDRL_FIN_ON_LOAD=0 -> stand-alone
        // finalize launches.
        const char* e6 = cdrl_getenv("CDRL_FIN_ON_LOAD");
        fin_on_load_ = !(e6 && atoi(e6) == 0);
        const char* e4 = cdrl_getenv("CDRL_FUSED_BWD");     // 0 -> backward-data (critical stream) + filter gradient (side stream) as two kernels
        fused_bwd_ = !(e4 && atoi(e4) == 0);
        // bf16 storage: its two-kernel form is cheap already (one plane, half the bytes); fused-on vs fused-off measured
        // 12.41 vs 12.20 ms per update-step at B = 256, 18.67 vs 18.74 at B = 512, 31.38 vs 32.62 at B = 1024 -> from B = 512
        if (at_ && !(e4 && atoi(e4) == 1) && B < 512) fused_bwd_ = false;
    }

    // ---- stem (core/architectures.py:159-161)
    {
        const int at = at_;
        Tens y = tens_a(N * Hs * Ws, c.stem, false);
        PRef w = param(M_TRUNK, "img.stem.conv.w", {3, 3, 3, c.stem}, true);
        PRef b = param(M_TRUNK, "img.stem.conv.b", {c.stem}, true);
        note_scratch(0, 0, (size_t)N * Hs * Ws * c.stem, 0, (size_t)stem_bwd_part_elems(B, T, c.H, c.W, c.stem));
        Op op;
        const int H = c.H, W = c.W, Cs = c.stem;
        // forward: BN statistics in the conv's epilogue (training only; inference uses the moving statistics)
        const bool stem_fstats = stem_fwd_stats_supported(Cs) && !(cdrl_getenv("CDRL_FUSED_STEM") && atoi(cdrl_getenv("CDRL_FUSED_STEM")) == 0);
        const int nb_stem = stem_fstats ? stem_fwd_stats_nb(B, T, H, W, Cs) : 0;
        if (at && !stem_fstats) build_fail("bf16 activation storage needs the fused stem forward (stem channels %d, CDRL_FUSED_STEM)", Cs);
        op.fwd = [=](hipStream_t st, int training) -> int {
            // (bf16 storage: the statistics form is the one with a bf16 store; its partials are simply unused in inference)
            if (stem_fstats && (training || at)) return stem_fwd_stats(in_image_, w.p, b.p, y.p, scr_main_.part, B, T, H, W, Cs, st, at);
            return stem_fwd(in_image_, w.p, b.p, y.p, B, T, H, W, Cs, st);
        };
        const bool stem_fused = stem_bwd_fused_supported(Cs) && !(cdrl_getenv("CDRL_FUSED_STEM") && atoi(cdrl_getenv("CDRL_FUSED_STEM")) == 0);
        // blocks of the stem BatchNorm (allocated here: the stem conv's backward consumes them in the fused form)
        float* stem_stats = alloc((size_t)4 * T * Cs);
        float* stem_coef = alloc((size_t)3 * T * Cs);
        const int Hp0 = same_out_h(Hs, 2), Wp0 = same_out_h(Ws, 2);
        Tens pool = tens_a(N * Hp0 * Wp0, c.stem);
        if (at && !stem_fused) build_fail("bf16 activation storage needs the fused stem backward (stem channels %d, CDRL_FUSED_STEM)", Cs);
        uint8_t* argmax = reinterpret_cast<uint8_t*>(alloc(((size_t)N * Hp0 * Wp0 * c.stem + 3) / 4));
        note_named("img.stem.bn.stats", stem_stats, (size_t)4 * T * Cs * sizeof(float));
        note_named("img.stem.bn.x", y.p, (size_t)N * Hs * Ws * Cs * esz());
        note_named("img.stem.pool.argmax", argmax, (size_t)N * Hp0 * Wp0 * Cs);
        note_named("img.stem.pool.out", pool.p, (size_t)N * Hp0 * Wp0 * Cs * esz());
        note_named("img.stem.pool.out.g", pool.g, (size_t)N * Hp0 * Wp0 * Cs * esz());
        op.bwd = [=](hipStream_t st) -> int {
            if (stem_fused) {
                CDRL_TRY(next_slot(st));
                // the last kernel of the backward: nothing is left on the critical stream to run beside it, so the hand-over to the side
                // stream and back only costs its two event bubbles (CDRL_STEM_BWD_MAIN=0 -> side stream as in rounds 1-4)
                static const bool on_main = true;
                hipStream_t side = on_main ? st : fork_side(st);
                PoolSrc ps = make_pool_src(argmax, pool.g, Hs, Ws);
                static const bool diag_skip = cdrl_getenv("CDRL_DIAG_SKIP_STEMF") && atoi(cdrl_getenv("CDRL_DIAG_SKIP_STEMF")) == 1;    // timing diagnostics only (no stem filter gradient)
                if (!diag_skip)
                    CDRL_TRY(stem_bwd_filter_fused(in_image_, ps, y.p, stem_stats, stem_coef, w.g, b.g, B, T, H, W, Cs, fparts_[slot_], side, at));
                return done_side(side);
            }
            hipStream_t side = fork_side(st);
            CDRL_TRY(stem_bwd_filter(in_image_, dys_[slot_], w.g, b.g, B, T, H, W, Cs, fparts_[slot_], side));
            return done_side(side);
        };
        ops.push_back(op);
        // stem BN + ReLU6 + max-pool as one fused block: the BN op only produces statistics in the forward
        // (no apply), the pool kernel applies scale/shift/ReLU6 on the raw conv output while pooling, and the
        // BN backward gathers its incoming gradient straight from the pooled gradient through the argmax.
        const int Hp = Hp0, Wp = Wp0;
        {
            const int G = T, Mg = B * Hs * Ws, C = c.stem;
            PRef gamma = param(M_TRUNK, "img.stem.bn.gamma", {C}, true);
            PRef beta = param(M_TRUNK, "img.stem.bn.beta", {C}, true);
            PRef mm = param(M_TRUNK, "img.stem.bn.moving_mean", {C}, false);
            PRef mv = param(M_TRUNK, "img.stem.bn.moving_var", {C}, false);
            float* stats = stem_stats;
            note_bn_inference(gamma.p, beta.p, mm.p, mv.p, stats, G, C);
            float* coef = stem_coef;
            const int nb = vcol_geom(Mg, C).nb;
            const int nb_pool = vcol_geom(B * Hp * Wp, C).nb;
            note_scratch((size_t)G * std::max(std::max(nb, nb_pool), nb_stem) * 2 * C, (size_t)G * nb * C, 0, 0);
            View yv = y.v();
            Op bn;
            bn.fwd = [=](hipStream_t st, int training) -> int {
                if (training && !stem_fstats) CDRL_TRY(colstats(yv, G, Mg, C, scr_main_.part, st));
                if (training)       // (inference: statistics block from the batched launch)
                    CDRL_TRY(bn_finalize(scr_main_.part, stem_fstats ? nb_stem : nb, G, Mg, C, gamma.p, beta.p, mm.p, mv.p, 1, training, stats,
                                         st));
                return maxpool_bn_fwd(y.p, stats, G, B, pool.p, argmax, N, Hs, Ws, C, st, at);
            };
            bn.bwd = [=](hipStream_t st) -> int {
                PoolSrc ps = make_pool_src(argmax, pool.g, Hs, Ws);
                View none{nullptr, 0, 0};
                if (stem_fused) {       // sums in scatter form over the pooled gradient; the apply happens inside the stem filter-gradient GEMM
                    ps.pa = pool.p;                 // mask and xhat from the pooled activated output, no gather (bf16 storage: xhat from the ROUNDED pooled value)
                    CDRL_TRY(pool_bn_bwd_reduce(ps, y.p, G, B, C, stats, scr_main_.part, st, at));
                    return bn_bwd_finalize(scr_main_.part, nb_pool, G, Mg, C, stats, gamma.g, beta.g, coef, st);
                }
                CDRL_TRY(bn_bwd_reduce(none, 0, yv, G, Mg, C, stats, ACT_RELU6, scr_main_.part, st, &ps));
                CDRL_TRY(bn_bwd_finalize(scr_main_.part, nb, G, Mg, C, stats, gamma.g, beta.g, coef, st));
                CDRL_TRY(next_slot(st));
                return bn_bwd_apply(none, 0, yv, G, Mg, C, stats, coef, ACT_RELU6, dys_[slot_], part2s_[slot_], st, &ps);
            };
            ops.push_back(bn);
        }

        // ---- stages (core/architectures.py:120-151,164-167)
        Tens X = pool;
        int curH = Hp, curW = Wp, curC = c.stem;
        for (int s = 0; s < 3; ++s) {
            for (int u = 0; u < c.stage_n[s]; ++u) {
                const int stride = u == 0 ? 2 : 1;
                const int C = c.stage_c[s];
                const int sc_c = stride == 2 ? curC : curC / 2;
                const int main_in = stride == 2 ? curC : curC - sc_c;
                const int main_off = stride == 2 ? 0 : sc_c;
                const int mid = C / 2, main_out = C - sc_c;
                const int Ho = stride == 2 ? same_out_h(curH, 2) : curH, Wo = stride == 2 ? same_out_h(curW, 2) : curW;
                const int rows_in = N * curH * curW, rows_out = N * Ho * Wo;
                const int Mg_in = B * curH * curW, Mg_out = B * Ho * Wo;
                const std::string pre = "img.s" + std::to_string(s) + ".u" + std::to_string(u);
                Tens out = tens_a(rows_out, C);
                note_named(pre + ".out", out.p, (size_t)rows_out * C * esz());        // unit output / its gradient (per-unit parity tests)
                note_named(pre + ".out.g", out.g, (size_t)rows_out * C * esz());
                // stride-2 units: the shortcut branch (dw3x3/s2 -> BN -> 1x1 -> BN+ReLU6) only depends on the unit input; in the
                // FORWARD pass (where the side stream is idle) it runs on the side stream next to the main branch
                static const bool sc_overlap_env = true;
                const bool sc_overlap = sc_overlap_env && stride == 2;
                const int sc_ev = s;
                if (sc_overlap) {
                    Op fk;
                    fk.fwd = [=](hipStream_t st, int) -> int {
                        if (!side_enabled_) return 0;
                        hipEvent_t evf = nullptr;
                        CDRL_TRY(mark_stream(st, ev_sc_fork_[sc_ev], &evf));
                        CDRL_HIP(hipStreamWaitEvent(side_, evf, 0));
                        return 0;
                    };
                    fk.bwd = [](hipStream_t) -> int { return 0; };
                    ops.push_back(fk);
                }
                Passthrough pass;
                static const bool fuse_pass = !(cdrl_getenv("CDRL_FUSED_PASS") && atoi(cdrl_getenv("CDRL_FUSED_PASS")) == 0);
                if (stride == 1 && fuse_pass && sc_c == C - sc_c) {
                    // identity half: carried by the unit's last BatchNorm op (same channel count as the main half)
                    pass.fsrc = X.v(0);
                    pass.fdst = out.v(0);
                    pass.gsrc = out.gv(0);
                    pass.gdst = X.gv(0);
                } else if (stride == 1) {
                    if (at) build_fail("%s: the stand-alone identity-half copy has no bf16-storage form (CDRL_FUSED_PASS=0 is set)", pre.c_str());
                    Op cp;                      // shortcut half: identity through concat + shuffle
                    View src = X.v(0), dst = out.v(0), gsrc = out.gv(0), gdst = X.gv(0);
                    cp.fwd = [=](hipStream_t st, int) -> int {
                        return bn_apply(src, 1, rows_in, sc_c, nullptr, ACT_NONE, dst, C, st);
                    };
                    cp.bwd = [=](hipStream_t st) -> int { return gather_view(gsrc, C, rows_in, sc_c, gdst, 0, st); };
                    ops.push_back(cp);
                }
                Tens y1 = tens_a(rows_in, mid, false);
                Tens y2 = tens_a(rows_out, mid, false), a2 = tens_a(rows_out, mid);
                Tens y3 = tens_a(rows_out, main_out, false);
                // BatchNorm work folded into the 1x1-conv GEMMs (K, N <= 128: stages 0 and 1)
                // (the K, N = 232 variants of stage 2 run at one workgroup per CU -- 116 W-fragment VGPRs per wave; slower than the
                //  tiled GEMM + separate BN passes at v19 (+0.15 ms), faster at v29 (-0.24 ms/update-step): CDRL_FUSED_PW=1 -> off)
                const bool fpw = fused_dw_ && fused_pw_ && (fused_pw_wide_ || (mid <= 128 && main_in <= 128 && main_out <= 128)) &&
                                 pw_nn_supported(X.v(main_off), mid, main_in) &&
                                 pw_nn_supported(y2.v(), main_out, mid) && pw_nn_supported(y3.v(), mid, main_out) &&
                                 pw_nn_supported(y1.v(), main_in, mid);
                if (fpw) {
                    // BN-backward apply as GEMM operand prologue (needs the filter-gradient GEMM's fixed column mapping)
                    // (bit 2 of CDRL_FUSED_BB: also for N <= 32 outputs of the backward-data GEMM, i.e. the first unit's 24 input
                    //  channels; see the default above)
                    // (also for the wide stage-2 units: without the prologues there 19.55 vs 18.82 ms/update-step;
                    //  CU-masking the side stream re-measured at v33: 224 / 192 / 128 CUs -> 21.2 / 21.2 / 23.5 vs 18.7 ms)
                    const bool bb1 = (fused_bb_ & 1) && gemm_tn_dpro_supported(mid) && (main_in > 32 || (fused_bb_ & 4));
                    const bool bb3 = (fused_bb_ & 2) && gemm_tn_dpro_supported(main_out);
                    float* stats1 = alloc((size_t)4 * T * mid);
                    float* coef1 = alloc((size_t)3 * T * mid);
                    PwFuse f1;
                    f1.fwd_pw = true;
                    f1.epi_stats = true;
                    f1.bwd_pw = true;
                    if (bb1) {
                        f1.bb = true;
                        f1.bb_stats = stats1;
                        f1.bb_coef = coef1;          // dz = masked gradient in the current scratch slot (written by the dw op)
                    }
                    std::shared_ptr<bool> bn1_fin = std::make_shared<bool>(false);
                    if (bb1 && fused_dw_) {     // BN1's backward sums come out of the depthwise backward: sc->part, dwf_geom rows
                        // (parameter order = the reference's variable order: pw1's own parameters are registered first, as add_pw would)
                        (void)param(M_TRUNK, pre + ".pw1.w", {1, 1, main_in, mid}, true);
                        (void)param(M_TRUNK, pre + ".pw1.b", {mid}, true);
                        PRef g1 = param(M_TRUNK, pre + ".bn1.gamma", {mid}, true), b1 = param(M_TRUNK, pre + ".bn1.beta", {mid}, true);
                        f1.bb_fin = true;
                        f1.bb_fin_part = &build_scr_->part;
                        f1.bb_fin_nb = dwf_geom(B, T, curH, curW, mid, stride).nb_bwd;
                        f1.bb_dgamma = g1.g;
                        f1.bb_dbeta = b1.g;
                        f1.bb_fin_done = bn1_fin;
                    }
                    add_pw(ops, pre + ".pw1", X.v(main_off), rows_in, main_in, mid, y1.p, X.gv(main_off), stride == 2 ? 1 : 0,
                           bnrec(T, Mg_in, mid), f1);
                    const int nb1 = pw_fwd_nbpg(T, Mg_in, mid, main_in);
                    const int nbb = pw_bwd_nbpg(T, Mg_out, mid, main_out);        // pw2 backward-data epilogue rows
                    float* coef2 = nullptr;
                    std::shared_ptr<bool> bn2_done = std::make_shared<bool>(false);
                    float* stats2 = add_dw_block(ops, pre, "bn1", "dw", "bn2", y1.p, curH, curW, mid, stride, y2.p, a2.v(), a2.gv(),
                                                 View{nullptr, 0, 0}, nb1, false, nbb, stats1, coef1, bb1, &coef2, bn2_done, bn1_fin);
                    // BN3's statistics / coefficient blocks are allocated by add_bn below; bump-allocate them here first so
                    // that pw2 (which precedes bn3 in the op list) can reference them
                    PwFuse f2;
                    f2.fwd_pw = true;
                    f2.pro_stats = stats2;
                    f2.epi_stats = true;
                    f2.bwd_pw = true;
                    f2.bwd_ey = y2.p;
                    f2.bwd_epi_stats = stats2;
                    {   // BN2's blocks for the fused backward of pw2 (same arena slots as add_dw_block's lookups)
                        PRef g2 = param(M_TRUNK, pre + ".bn2.gamma", {mid}, true), b2 = param(M_TRUNK, pre + ".bn2.beta", {mid}, true);
                        f2.a_gamma = g2.p;
                        f2.a_beta = b2.p;
                        f2.a_dgamma = g2.g;
                        f2.a_dbeta = b2.g;
                        f2.a_coef = coef2;
                        f2.a_bn = true;
                    }
                    const size_t pw2_at = ops.size();
                    add_pw(ops, pre + ".pw2", y2.v(), rows_out, mid, main_out, y3.p, a2.gv(), 0, bnrec(T, Mg_out, main_out), f2);
                    BnRec r3 = add_bn(ops, M_TRUNK, pre + ".bn3", y3.v(), T, Mg_out, main_out, true, ACT_RELU6, out.v(sc_c), C,
                                      out.gv(sc_c), C, nullptr, pw_fwd_nbpg(T, Mg_out, main_out, mid), bb3, pass);
                    if (bb3) {      // rebuild pw2 with the BN3 blocks known (same parameters -> same arena slots)
                        f2.bb = true;
                        f2.bb_stats = r3.stats;
                        f2.bb_coef = r3.coef;
                        f2.bb_dz = out.gv(sc_c);
                        f2.bb_shuffle = C;
                        f2.bb_act = ACT_RELU6;
                        f2.bb_claim_slot = true;
                        f2.a_bn_done = bn2_done;
                        f2.bb_fin = true;               // BN3's sums: bn_bwd_reduce(_shuf) into *r3.part_ptr, r3.nb rows per group
                        f2.bb_fin_part = r3.part_ptr;
                        f2.bb_fin_nb = r3.nb;
                        f2.bb_dgamma = r3.dgamma;
                        f2.bb_dbeta = r3.dbeta;
                        f2.bb_fin_done = r3.fin_by_consumer;
                        std::vector<Op> tmp;
                        add_pw(tmp, pre + ".pw2", y2.v(), rows_out, mid, main_out, y3.p, a2.gv(), 0, bnrec(T, Mg_out, main_out), f2);
                        ops[pw2_at] = tmp[0];
                    }
                } else {
                    add_pw(ops, pre + ".pw1", X.v(main_off), rows_in, main_in, mid, y1.p, X.gv(main_off), stride == 2 ? 1 : 0,
                           bnrec(T, Mg_in, mid));
                    if (fused_dw_) {
                        add_dw_block(ops, pre, "bn1", "dw", "bn2", y1.p, curH, curW, mid, stride, y2.p, a2.v(), a2.gv(),
                                     View{nullptr, 0, 0});
                    } else {
                        Tens a1 = tens_a(rows_in, mid);
                        BnRec r1 = add_bn(ops, M_TRUNK, pre + ".bn1", y1.v(), T, Mg_in, mid, true, ACT_RELU6, a1.v(), 0, a1.gv(), 0,
                                          nullptr);
                        add_dw(ops, pre + ".dw", a1.v(), N, curH, curW, mid, stride, y2.p, a1.gv(), 0, &r1);
                        add_bn(ops, M_TRUNK, pre + ".bn2", y2.v(), T, Mg_out, mid, true, ACT_NONE, a2.v(), 0, a2.gv(), 0, nullptr);
                    }
                    add_pw(ops, pre + ".pw2", a2.v(), rows_out, mid, main_out, y3.p, a2.gv(), 0, bnrec(T, Mg_out, main_out));
                    add_bn(ops, M_TRUNK, pre + ".bn3", y3.v(), T, Mg_out, main_out, true, ACT_RELU6, out.v(sc_c), C, out.gv(sc_c), C,
                           nullptr, 0, false, pass);
                }
                if (stride == 2) {
                    const size_t sc_begin = ops.size();
                    if (sc_overlap) build_scr_ = &scr_sc_;
                    Tens ys1 = tens_a(rows_out, sc_c, false), b1 = tens_a(rows_out, sc_c);
                    // Round 6: the shortcut branch dw3x3/s2 -> BN -> 1x1 -> BN+ReLU6 (core/architectures.py:133-137) is the second half of a
                    // main branch and runs on the same fused ops: BN-apply of sc_bn1 on the conv's operand load (its output is never
                    // written), statistics of sc_bn2 in the conv's epilogue, and in the backward the BatchNorm-backward of sc_bn2 as the
                    // conv's operand prologue, sc_bn1's backward sums out of the conv backward (fused conv backward: stages 0 / 1; the
                    // BatchNorm-sum epilogue of the wide kernel: stage 2).  Before: 3 more launches per stride-2 unit on the critical
                    // stream of every backward (apply of sc_bn2, reduce + finalize of sc_bn1) and two on the forward's side stream.
                    static const bool sc_fused_env = !(cdrl_getenv("CDRL_FUSED_SC") && atoi(cdrl_getenv("CDRL_FUSED_SC")) == 0);
                    const bool sc_fpw = sc_fused_env && fused_dw_ && fused_pw_ && (fused_pw_wide_ || sc_c <= 128) && (fused_bb_ & 2) &&
                                        gemm_tn_dpro_supported(sc_c) && pw_nn_supported(ys1.v(), sc_c, sc_c);
                    if (sc_fpw) {
                        const int nbb_sc = pw_bwd_nbpg(T, Mg_out, sc_c, sc_c);      // conv backward-data epilogue rows (BatchNorm sums of sc_bn1)
                        float* coef_s1 = nullptr;
                        std::shared_ptr<bool> s1_done = std::make_shared<bool>(false);
                        float* stats_s1 = add_dw_block(ops, pre, nullptr, "sc_dw", "sc_bn1", X.p, curH, curW, sc_c, 2, ys1.p, b1.v(), b1.gv(),
                                                       X.gv(0), 0, false, nbb_sc, nullptr, nullptr, false, &coef_s1, s1_done, nullptr);
                        Tens ys2 = tens_a(rows_out, sc_c, false);
                        PwFuse fs;
                        fs.fwd_pw = true;
                        fs.pro_stats = stats_s1;
                        fs.epi_stats = true;
                        fs.bwd_pw = true;
                        fs.bwd_ey = ys1.p;
                        fs.bwd_epi_stats = stats_s1;
                        {
                            PRef g1 = param(M_TRUNK, pre + ".sc_bn1.gamma", {sc_c}, true), bt1 = param(M_TRUNK, pre + ".sc_bn1.beta", {sc_c}, true);
                            fs.a_gamma = g1.p;
                            fs.a_beta = bt1.p;
                            fs.a_dgamma = g1.g;
                            fs.a_dbeta = bt1.g;
                            fs.a_coef = coef_s1;
                            fs.a_bn = true;
                        }
                        const size_t scpw_at = ops.size();
                        add_pw(ops, pre + ".sc_pw", ys1.v(), rows_out, sc_c, sc_c, ys2.p, b1.gv(), 0, bnrec(T, Mg_out, sc_c), fs);
                        BnRec rs = add_bn(ops, M_TRUNK, pre + ".sc_bn2", ys2.v(), T, Mg_out, sc_c, true, ACT_RELU6, out.v(0), C, out.gv(0), C,
                                          nullptr, pw_fwd_nbpg(T, Mg_out, sc_c, sc_c), true);
                        fs.bb = true;
                        fs.bb_stats = rs.stats;
                        fs.bb_coef = rs.coef;
                        fs.bb_dz = out.gv(0);
                        fs.bb_shuffle = C;
                        fs.bb_act = ACT_RELU6;
                        fs.bb_claim_slot = true;
                        fs.a_bn_done = s1_done;
                        fs.bb_fin = true;
                        fs.bb_fin_part = rs.part_ptr;
                        fs.bb_fin_nb = rs.nb;
                        fs.bb_dgamma = rs.dgamma;
                        fs.bb_dbeta = rs.dbeta;
                        fs.bb_fin_done = rs.fin_by_consumer;
                        std::vector<Op> tmp;
                        add_pw(tmp, pre + ".sc_pw", ys1.v(), rows_out, sc_c, sc_c, ys2.p, b1.gv(), 0, bnrec(T, Mg_out, sc_c), fs);
                        ops[scpw_at] = tmp[0];
                    } else {
                    if (fused_dw_) {
                        add_dw_block(ops, pre, nullptr, "sc_dw", "sc_bn1", X.p, curH, curW, sc_c, 2, ys1.p, b1.v(), b1.gv(),
                                     X.gv(0));
                    } else {
                        add_dw(ops, pre + ".sc_dw", X.v(0), N, curH, curW, sc_c, 2, ys1.p, X.gv(0), 0);
                        add_bn(ops, M_TRUNK, pre + ".sc_bn1", ys1.v(), T, Mg_out, sc_c, true, ACT_NONE, b1.v(), 0, b1.gv(), 0,
                               nullptr);
                    }
                    Tens ys2 = tens_a(rows_out, sc_c, false);
                    add_pw(ops, pre + ".sc_pw", b1.v(), rows_out, sc_c, sc_c, ys2.p, b1.gv(), 0, bnrec(T, Mg_out, sc_c));
                    add_bn(ops, M_TRUNK, pre + ".sc_bn2", ys2.v(), T, Mg_out, sc_c, true, ACT_RELU6, out.v(0), C, out.gv(0),
                           C, nullptr);
                    }
                    if (sc_overlap) {
                        build_scr_ = &scr_main_;
                        for (size_t i = sc_begin; i < ops.size(); ++i) {        // forward of the shortcut ops -> side stream
                            auto f = ops[i].fwd;
                            ops[i].fwd = [=](hipStream_t st, int training) -> int { return f(side_enabled_ ? side_ : st, training); };
                        }
                        Op jn;
                        jn.fwd = [=](hipStream_t st, int) -> int {
                            if (!side_enabled_) return 0;
                            CDRL_HIP(hipEventRecord(ev_sc_done_[sc_ev], side_));
                            CDRL_HIP(hipStreamWaitEvent(st, ev_sc_done_[sc_ev], 0));
                            const uint64_t js = note_side_record(ev_sc_done_[sc_ev]);
                            if (st == main_) main_waited_ = js;
                            return 0;
                        };
                        jn.bwd = [](hipStream_t) -> int { return 0; };
                        ops.push_back(jn);
                    }
                }
                X = out;
                curH = Ho;
                curW = Wo;
                curC = C;
            }
        }
        // ---- head conv + GAP (core/architectures.py:170-172)
        const int P = curH * curW, rows = N * P;
        Tens yh = tens_a(rows, c.last, false);
        add_pw(ops, "img.head.conv", X.v(), rows, curC, c.last, yh.p, X.gv(), 0, bnrec(T, B * P, c.last));
        feat_ = tens(N, c.last);
        static const bool gap_fused = true;
        if (gap_fused) {
            // BatchNorm + ReLU6 + GlobalAveragePooling2D as one op: the 12288 x 768 activated tensor and its gradient are never
            // written -- the forward pools on the fly, the backward reads the pooled gradient broadcast over the frame's pixels
            Passthrough hp;
            hp.gap_out = feat_.p;
            hp.gap_dout = feat_.g;
            hp.gap_rows = P;
            add_bn(ops, M_TRUNK, "img.head.bn", yh.v(), T, B * P, c.last, true, ACT_RELU6, View{nullptr, 0, 0}, 0, View{nullptr, 0, 0}, 0,
                   nullptr, 0, false, hp);
        } else {
            if (at) build_fail("bf16 activation storage needs the fused head pool (CDRL_FUSED_GAP=0 is set)");
            Tens ah = tens(rows, c.last);
            add_bn(ops, M_TRUNK, "img.head.bn", yh.v(), T, B * P, c.last, true, ACT_RELU6, ah.v(), 0, ah.gv(), 0, nullptr);
            Tens feat = feat_;
            const int Cl = c.last;
            Op gp;
            gp.fwd = [=](hipStream_t st, int) -> int { return gap_fwd(ah.p, feat.p, N, P, Cl, st); };
            gp.bwd = [=](hipStream_t st) -> int { return gap_bwd(feat.g, ah.g, N, P, Cl, st); };
            ops.push_back(gp);
        }
    }

    // every trunk parameter registered from here on is a TAIL tensor (feature nets, GRUs, concat BN + Dense): their gradients
    // are final at the `mark` op below, those of the tower above only at the end of the backward
    if (!table_frozen_) tail_off_ = tr_size_[M_TRUNK];
    // ---- feature nets (core/architectures.py:9-27)
    const char* fnames[3] = {"road", "vehicle", "navigation"};
    const int fdims[3] = {c.road, c.vehicle, c.navigation};
    Tens fout[3];
    for (int i = 0; i < 3; ++i) {
        const std::string nm = fnames[i];
        const int D = fdims[i];
        Tens xin = tens(N, D, false);
        Op pin;
        const int which = i;
        pin.fwd = [=](hipStream_t st, int) -> int {
            const float* src = which == 0 ? in_road_ : (which == 1 ? in_vehicle_ : in_navigation_);
            return permute_bt(src, xin.p, B, T, D, st);
        };
        pin.bwd = [](hipStream_t) -> int { return 0; };
        build_scr_ = &scr_aux_;
        aux_ops_.push_back(pin);
        Tens a0 = tens(N, c.feat), n0 = tens(N, c.feat), a1 = tens(N, c.feat), n1 = tens(N, c.feat);
        add_dense(aux_ops_, M_TRUNK, nm + ".fc0", xin.v(), N, D, c.feat, ACT_RELU6, a0.v(), a0.gv(), View{nullptr, 0, 0}, 0,
                  false, "glorot");
        add_bn(aux_ops_, M_TRUNK, nm + ".bn0", a0.v(), T, B, c.feat, false, ACT_NONE, n0.v(), 0, n0.gv(), 0, a0.g);
        add_dense(aux_ops_, M_TRUNK, nm + ".fc1", n0.v(), N, c.feat, c.feat, ACT_RELU6, a1.v(), a1.gv(), n0.gv(), 0, true,
                  "glorot");
        add_bn(aux_ops_, M_TRUNK, nm + ".bn1", a1.v(), T, B, c.feat, false, ACT_NONE, n1.v(), 0, n1.gv(), 0, a1.g);
        build_scr_ = &scr_main_;
        fout[i] = n1;
    }

    // ---- GRUs + concat + BN + Dense (core/networks.py:44-56)
    const int catC = c.rnn_image + 3 * c.rnn_small;
    Tens cat = tens(B, catC);
    {   // backward: everything behind this point of the (reversed) op list -- heads, concat BN + Dense, the small-modality
        // nets on the aux stream, the image GRU -- has its gradient work enqueued: hand the "tail gradients final" point to
        // the caller's communication stream (DataParallelLearner: early all-reduce bucket under the tower's backward)
        Op mark;
        mark.fwd = [](hipStream_t, int) -> int { return 0; };
        mark.bwd = [=](hipStream_t st) -> int {
            // under hipGraph capture the external communication stream must not be pulled into the capture (it would never be
            // joined, and a replay would never release it): the caller falls back to the single post-pass all-reduce
            if (!comm_ || graphs_enabled_) return 0;
            CDRL_TRY(aux_wait());
            CDRL_HIP(hipEventRecord(ev_tail_main_, st));          // (recorded, with its system-scope fence: the collectives read these bytes)
            CDRL_HIP(hipStreamWaitEvent(comm_, ev_tail_main_, 0));
            if (side_enabled_) {
                CDRL_TRY(flush_side(st));       // (queued side jobs go out BEHIND an event of the critical stream, as everywhere else)
                CDRL_HIP(hipEventRecord(ev_tail_side_, side_));
                CDRL_HIP(hipStreamWaitEvent(comm_, ev_tail_side_, 0));
                if (aux_pending_) CDRL_HIP(hipStreamWaitEvent(comm_, ev_aux_done_, 0));
            }
            return 0;
        };
        ops.push_back(mark);
    }
    add_gru(ops, "gru_image", feat_, c.last, c.rnn_image, cat.v(0), cat.gv(0), true);
    build_scr_ = &scr_aux_;
    for (int i = 0; i < 3; ++i)
        add_gru(aux_ops_, std::string("gru_") + fnames[i], fout[i], c.feat, c.rnn_small, cat.v(c.rnn_image + i * c.rnn_small),
                cat.gv(c.rnn_image + i * c.rnn_small), true);
    build_scr_ = &scr_main_;
    add_aux_join(ops);
    Tens ncat = tens(B, catC);
    add_bn(ops, M_TRUNK, "dyn.bn", cat.v(), 1, B, catC, false, ACT_NONE, ncat.v(), 0, ncat.gv(), 0, cat.g);
    dyn_ = tens(B, c.dyn);
    add_dense(ops, M_TRUNK, "dyn.fc", ncat.v(), B, catC, c.dyn, ACT_NONE, dyn_.v(), dyn_.gv(), ncat.gv(), 0, true, "glorot");
}

void Learner::build_head(std::vector<Op>& ops, int model, const std::string& prefix, Tens& lin, int nheads,
                         const int* head_dims, const char* const* head_names) {
    const Config& c = cfg_;
    const int B = c.B;
    Tens n0 = tens(B, c.dyn), a0 = tens(B, c.head), n1 = tens(B, c.head), a1 = tens(B, c.head);
    add_bn(ops, model, prefix + ".bn0", dyn_.v(), 1, B, c.dyn, false, ACT_NONE, n0.v(), 0, n0.gv(), 0, dyn_.g);
    add_dense(ops, model, prefix + ".fc0", n0.v(), B, c.dyn, c.head, ACT_SWISH6, a0.v(), a0.gv(), n0.gv(), 0, true, "glorot");
    add_bn(ops, model, prefix + ".bn1", a0.v(), 1, B, c.head, false, ACT_NONE, n1.v(), 0, n1.gv(), 0, a0.g);
    add_dense(ops, model, prefix + ".fc1", n1.v(), B, c.head, c.head, ACT_SWISH6, a1.v(), a1.gv(), n1.gv(), 0, true, "glorot");
    int L = 0;
    for (int i = 0; i < nheads; ++i) L += head_dims[i];
    lin = tens(B, L);
    static const bool fused_heads = true;
    if (fused_heads && nheads <= HEADS_MAX && L <= HEADS_MAX_OUT) {
        // all linear heads of the branch in one launch per direction (heads.hip); same parameter names / order as add_dense
        HeadSet hs{};
        hs.nheads = nheads;
        int off = 0;
        for (int i = 0; i < nheads; ++i) {
            PRef w = param(model, prefix + "." + head_names[i] + ".w", {c.head, head_dims[i]}, true);
            PRef b = param(model, prefix + "." + head_names[i] + ".b", {head_dims[i]}, true);
            hs.n[i] = head_dims[i];
            hs.off[i] = off;
            hs.w[i] = w.p;
            hs.b[i] = b.p;
            hs.gw[i] = w.g;
            hs.gb[i] = b.g;
            off += head_dims[i];
        }
        const int K = c.head;
        Tens a1c = a1, linc = lin;
        Op op;
        op.fwd = [=](hipStream_t st, int) -> int { return heads_fwd(a1c.p, K, hs, linc.p, L, B, K, st); };
        op.bwd = [=](hipStream_t st) -> int { return heads_bwd(a1c.p, K, hs, linc.g, L, a1c.g, K, B, K, st); };
        ops.push_back(op);
        return;
    }
    int off = 0;
    for (int i = 0; i < nheads; ++i) {
        add_dense(ops, model, prefix + "." + head_names[i], a1.v(), B, c.head, head_dims[i], ACT_NONE, lin.v(off),
                  lin.gv(off), a1.gv(), i < nheads - 1 ? 1 : 0, true, "glorot");
        off += head_dims[i];
    }
}

void Learner::build(bool dry) {
    dry_ = dry;
    ws_off_ = 0;
    guard_off_.clear();
    trunk_ops_.clear();
    policy_ops_.clear();
    value_ops_.clear();
    old_policy_ops_.clear();
    if (!dry) {
        scr_main_.part = alloc_d(max_part_);
        scr_main_.part2 = alloc_d(max_part2_);
        scr_main_.tn = alloc(max_tn_);
        scr_aux_.part = alloc_d(max_part_);
        scr_aux_.part2 = alloc_d(max_part2_);
        scr_aux_.tn = alloc(max_tn_);
        scr_sc_.part = alloc_d(max_part_);
        scr_sc_.part2 = alloc_d(max_part2_);
        scr_sc_.tn = scr_aux_.tn;           // (unused by the shortcut ops)
        for (int i = 0; i < NSLOT; ++i) {
            dys_[i] = alloc((max_dy_ * esz() + 3) / 4);      // tower gradients: activation-typed
            part2s_[i] = alloc_d(max_part2_);
            tns_[i] = alloc(max_tn_);
            fparts_[i] = alloc_d(max_fpart_);
        }
        for (int i = 0; i < NQ; ++i) {
            qparts_[i] = alloc(max_qpart_);
            dbparts_[i] = alloc_d(max_dbpart_);
            fintots_[i] = alloc_d((size_t)8 * 2 * 128);
        }
    }
    h_pwt_.clear();
    pwt_by_name_.clear();
    pwt_tiles_ = 0;
    zero_once_.clear();
    h_pack_.clear();
    h_pack3_.clear();
    h_gpack_.clear();
    h_bninf_.clear();
    bninf_max_c_ = 0;
    build_trunk(trunk_ops_);
    const int A = cfg_.A;
    const int pdims[4] = {A, A, 1, 1};
    const char* const pnames[4] = {"alpha", "beta", "similarity", "speed"};
    build_head(policy_ops_, M_POLICY, "pi", lin_p_, 4, pdims, pnames);
    const int vdims[4] = {1, 1, 1, 1};
    const char* const vnames[4] = {"base", "exp", "speed", "similarity"};
    build_head(value_ops_, M_VALUE, "v", lin_v_, 4, vdims, vnames);
    build_head(old_policy_ops_, M_OLD_POLICY, "pi", lin_old_, 4, pdims, pnames);
    // device copies of the per-pass tables (packed weights, inference statistics, transposes): sized AFTER the heads are built, so that an
    // op of a control branch may register entries too (round 6: a small-M GEMM that did so overflowed the table sized behind the trunk)
    d_bninf_ = reinterpret_cast<BnInfEntry*>(alloc((h_bninf_.size() + 1) * sizeof(BnInfEntry) / sizeof(float) + 4));
    d_pack_ = reinterpret_cast<PwPack*>(alloc((h_pack_.size() + 1) * sizeof(PwPack) / sizeof(float) + 4));
    d_pack3_ = reinterpret_cast<PwX3Pack*>(alloc((h_pack3_.size() + 1) * sizeof(PwX3Pack) / sizeof(float) + 4));
    d_gpack_ = reinterpret_cast<GemmX3Pack*>(alloc((h_gpack_.size() + 1) * sizeof(GemmX3Pack) / sizeof(float) + 4));
    d_pwt_ = reinterpret_cast<PwTranspose*>(alloc((h_pwt_.size() + 1) * sizeof(PwTranspose) / sizeof(float) + 4));
    metrics_p_ = alloc(16);
    metrics_v_ = alloc(16);
    aux_p_ = alloc((size_t)cfg_.B * 4 * A);
    aux_v_ = alloc((size_t)cfg_.B * 2);
    sample_u_ = alloc((size_t)cfg_.B * A);
    sample_da_ = alloc((size_t)cfg_.B * A);
    sample_db_ = alloc((size_t)cfg_.B * A);
    note_named("sample.u", sample_u_, (size_t)cfg_.B * A * sizeof(float));
    note_named("sample.du_dalpha", sample_da_, (size_t)cfg_.B * A * sizeof(float));
    note_named("sample.du_dbeta", sample_db_, (size_t)cfg_.B * A * sizeof(float));
    hp_dev_ = reinterpret_cast<DevHP*>(alloc(sizeof(DevHP) / sizeof(float) + 4));
    note_named("hparams", hp_dev_, sizeof(DevHP));      // 10 floats (lr x3, clip, entropy, clip norms x2, beta1, beta2, eps), 3 int step counters
    // optimiser tables
    for (int m = 1; m <= 2; ++m) {
        SegTable& s = seg_[m];
        int nt = 0;
        int64_t nch = 0;
        for (const ParamInfo& pi : infos_[m])
            if (pi.trainable) {
                ++nt;
                nch += (pi.numel + 1023) / 1024;
            }
        s.segs = reinterpret_cast<TensorSeg*>(alloc((size_t)nt * sizeof(TensorSeg) / sizeof(float)));
        s.chunk_tensor = reinterpret_cast<int*>(alloc((size_t)nch));
        s.chunk_off = reinterpret_cast<int64_t*>(alloc((size_t)nch * 2));
        s.chunk_part = alloc_d((size_t)nch);
        s.sqnorms = alloc((size_t)nt);
    }
    if (guard_) {       // band table (device) behind everything else; no band behind it
        const bool g = guard_;
        guard_ = false;
        guard_tab_ = reinterpret_cast<int64_t*>(alloc_d(GUARD_TABLE_MAX + 2));
        guard_ = g;
        if (!dry && guard_off_.size() > GUARD_TABLE_MAX) build_fail("CDRL_GUARD: %zu bands exceed the table", guard_off_.size());
    }
    if (dry) {
        // scratch goes first in the real layout; account for it here
        ws_off_ += 512 + 2 * (align_up(max_part_ * sizeof(double), 256) + align_up(max_part2_ * sizeof(double), 256) +
                        align_up(max_tn_ * sizeof(float), 256)) +
                   align_up(max_part_ * sizeof(double), 256) + align_up(max_part2_ * sizeof(double), 256) +
                   NSLOT * (align_up((max_dy_ * esz() + 3) / 4 * sizeof(float), 256) + align_up(max_part2_ * sizeof(double), 256) +
                            align_up(max_tn_ * sizeof(float), 256) + align_up(max_fpart_ * sizeof(double), 256)) +
                   NQ * (align_up(max_qpart_ * sizeof(float), 256) + align_up(max_dbpart_ * sizeof(double), 256) +
                         align_up((size_t)8 * 2 * 128 * sizeof(double), 256));
        if (guard_) ws_off_ += (size_t)(8 + 4 * NSLOT + 3 * NQ) * GUARD_BYTES;      // one band per scratch allocation above
        ws_bytes_ = ws_off_ + 4096;
    }
}

void Learner::build_seg_tables() {
    for (int m = 1; m <= 2; ++m) {
        SegTable& s = seg_[m];
        s.h_segs.clear();
        s.h_chunk_tensor.clear();
        s.h_chunk_off.clear();
        for (const ParamInfo& pi : infos_[m]) {
            if (!pi.trainable) continue;
            TensorSeg seg;
            seg.off = pi.off;
            seg.n = pi.numel;
            seg.first_chunk = (int)s.h_chunk_tensor.size();
            seg.nchunks = (int)((pi.numel + 1023) / 1024);
            for (int k = 0; k < seg.nchunks; ++k) {
                s.h_chunk_tensor.push_back((int)s.h_segs.size());
                s.h_chunk_off.push_back(pi.off + (int64_t)k * 1024);
            }
            s.h_segs.push_back(seg);
        }
        s.ntensors = (int)s.h_segs.size();
        s.nchunks = (int)s.h_chunk_tensor.size();
    }
}

int Learner::upload_seg_tables() {
    if (!h_pwt_.empty())
        CDRL_HIP(hipMemcpy(d_pwt_, h_pwt_.data(), h_pwt_.size() * sizeof(PwTranspose), hipMemcpyHostToDevice));
    if (!h_pack_.empty())
        CDRL_HIP(hipMemcpy(d_pack_, h_pack_.data(), h_pack_.size() * sizeof(PwPack), hipMemcpyHostToDevice));
    if (!h_bninf_.empty())
        CDRL_HIP(hipMemcpy(d_bninf_, h_bninf_.data(), h_bninf_.size() * sizeof(BnInfEntry), hipMemcpyHostToDevice));
    if (!h_pack3_.empty())
        CDRL_HIP(hipMemcpy(d_pack3_, h_pack3_.data(), h_pack3_.size() * sizeof(PwX3Pack), hipMemcpyHostToDevice));
    if (!h_gpack_.empty())
        CDRL_HIP(hipMemcpy(d_gpack_, h_gpack_.data(), h_gpack_.size() * sizeof(GemmX3Pack), hipMemcpyHostToDevice));
    for (int m = 1; m <= 2; ++m) {
        SegTable& s = seg_[m];
        CDRL_HIP(hipMemcpy(s.segs, s.h_segs.data(), s.h_segs.size() * sizeof(TensorSeg), hipMemcpyHostToDevice));
        CDRL_HIP(hipMemcpy(s.chunk_tensor, s.h_chunk_tensor.data(), s.h_chunk_tensor.size() * sizeof(int),
                           hipMemcpyHostToDevice));
        CDRL_HIP(hipMemcpy(s.chunk_off, s.h_chunk_off.data(), s.h_chunk_off.size() * sizeof(int64_t),
                           hipMemcpyHostToDevice));
    }
    return 0;
}

int Learner::bind(const Buffers& b) {
    if (!b.params || !b.grads || !b.adam_m || !b.adam_v || !b.workspace) {
        set_error("bind: null buffer");
        return -1;
    }
    if (b.workspace_bytes < ws_bytes_) {
        set_error("bind: workspace too small (%zu < %zu)", b.workspace_bytes, ws_bytes_);
        return -1;
    }
    drop_graphs();
    named_.clear();
    buf_ = b;
    ws_base_ = reinterpret_cast<char*>(b.workspace);
    build(false);
    if (ws_off_ > b.workspace_bytes) {
        set_error("bind: internal workspace overflow");
        return -1;
    }
    CDRL_TRY(upload_seg_tables());
    for (auto& z : zero_once_) CDRL_HIP(hipMemset(z.first, 0, z.second));
    if (guard_) {
        if (!build_err_.empty()) {
            set_error("bind: %s", build_err_.c_str());
            return -1;
        }
        CDRL_HIP(hipMemcpy(guard_tab_, guard_off_.data(), guard_off_.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(guard_fill_kernel, dim3((unsigned)guard_off_.size()), dim3(256), 0, nullptr, ws_base_, guard_tab_, (int)(GUARD_BYTES / 4));
        CDRL_LAUNCH_CHECK();
        CDRL_HIP(hipDeviceSynchronize());
    }
    if (!side_) {
        const char* env = cdrl_getenv("CDRL_SIDE_STREAM");
        side_enabled_ = !(env && atoi(env) == 0);
        // hipGraph replay is OFF by default: measured on MI355X / ROCm 7.2 at B=256 the captured update-step
        // (4 graphs of ~500-1000 kernel nodes over two streams) replays in 32.7 ms vs 31.0 ms eager, and the
        // host still spends 15 ms per step inside hipGraphLaunch (22 ms for eager launches): neither mode is
        // host-bound.  CDRL_GRAPH=1 enables it (parity suite passes in both modes).
        const char* genv = cdrl_getenv("CDRL_GRAPH");
        graphs_enabled_ = genv && atoi(genv) != 0;
        // All three streams at the default priority.  Giving the main stream (the dependent chain) the highest and the side /
        // aux streams the lowest priority (measured in round 2) changes nothing for the update-step (15.98 vs 16.01 ms) but costs
        // rollout inference 4.4 ms per call: whenever the high-priority queue sits on a barrier that waits for a kernel of a
        // low-priority queue (the small-modality nets on the aux stream: at 1..128 environments the main stream reaches the join
        // first), the command processor comes back to the low-priority queue only after milliseconds -- predict() 5.5 ms instead
        // of 1.1 ms at E = 1 (tools/bench_rollout_rows.py).
        const int prio_lo = 0, prio_hi = 0;
        CDRL_HIP(hipStreamCreateWithPriority(&main_, hipStreamNonBlocking, prio_hi));
        // Events between the engine's own streams carry no system-scope fence (hipEventDisableSystemFence): the marker packet of a default
        // event writes the L2 back and invalidates it, in the middle of the critical stream.  The events that hand over to the communication
        // stream (collectives: other devices read what they cover) keep it.  CDRL_EVENT_FENCE=1 -> default events everywhere (rounds 1-5).
        const char* fe = cdrl_getenv("CDRL_EVENT_FENCE");
        const unsigned evf_int = hipEventDisableTiming | ((fe && atoi(fe) == 1) ? 0u : (unsigned)hipEventDisableSystemFence);
        CDRL_HIP(hipEventCreateWithFlags(&ev_in_, evf_int));      // (caller's stream: same device; what the host reads afterwards goes through a copy with its own fences)
        CDRL_HIP(hipEventCreateWithFlags(&ev_out_, evf_int));
        CDRL_HIP(hipEventCreateWithFlags(&ev_out_sys_, hipEventDisableTiming));
        CDRL_HIP(hipEventCreateWithFlags(&ev_in_sys_, hipEventDisableTiming));
        CDRL_HIP(hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, prio_lo));
        for (int i = 0; i < NSLOT; ++i) {
            CDRL_HIP(hipEventCreateWithFlags(&ev_main_[i], evf_int));
            CDRL_HIP(hipEventCreateWithFlags(&ev_side_[i], evf_int));
        }
        CDRL_HIP(hipEventCreateWithFlags(&ev_join_, evf_int));
        for (int i = 0; i < NQ; ++i) CDRL_HIP(hipEventCreateWithFlags(&ev_q_[i], evf_int));
        {
            // CDRL_SIDE_LAG=-1 -> every claim of a scratch slot waits for its own event (rounds 1-5); n >= 0: see wait_side_record
            const char* le = cdrl_getenv("CDRL_SIDE_LAG");
            side_lag_ = le ? atoi(le) : 4;
        }
        {
            // CDRL_TAIL_EVENTS=0 -> forks by hipEventRecord on the critical stream (rounds 1-5)
            const char* te = cdrl_getenv("CDRL_TAIL_EVENTS");
            const bool tail_on = !(te && atoi(te) == 0) && !graphs_enabled_;
            tail_.stream = main_;
            tail_.n = 0;
            if (tail_on) {
                for (int i = 0; i < 32; ++i) CDRL_HIP(hipEventCreateWithFlags(&tail_.ring[i], evf_int));
                tail_.n = 32;
            }
        }
        CDRL_HIP(hipEventCreateWithFlags(&ev_tail_main_, hipEventDisableTiming));
        CDRL_HIP(hipEventCreateWithFlags(&ev_tail_side_, hipEventDisableTiming));
        for (int i = 0; i < 3; ++i) {
            CDRL_HIP(hipEventCreateWithFlags(&ev_sc_fork_[i], evf_int));
            CDRL_HIP(hipEventCreateWithFlags(&ev_sc_done_[i], evf_int));
        }
        CDRL_HIP(hipStreamCreateWithPriority(&aux_, hipStreamNonBlocking, prio_lo));
        CDRL_HIP(hipEventCreateWithFlags(&ev_aux_fork_, evf_int));
        CDRL_HIP(hipEventCreateWithFlags(&ev_aux_done_, evf_int));
        const char* tenv = cdrl_getenv("CDRL_AUX_THREAD");
        // opt-in (CDRL_AUX_THREAD=1): measured 20.76 vs 20.83 ms/update-step at B=256 -- the host is 8 ms per step ahead of
        // the GPU in steady state, so the second enqueue thread only pays off for small images (host-bound below ~45x60)
        if (side_enabled_ && !graphs_enabled_ && tenv && atoi(tenv) == 1) {
            int dev = 0;
            CDRL_HIP(hipGetDevice(&dev));
            aux_worker_.reset(new AuxWorker(dev));
            side_lag_ = -1;         // (two enqueue threads: the record bookkeeping is single-threaded)
        }
    }
    if (!hp_stage_) CDRL_HIP(hipHostMalloc(reinterpret_cast<void**>(&hp_stage_), sizeof(DevHP), 0));
    CDRL_HIP(hipMemcpy(hp_dev_, &hp_host_, sizeof(DevHP), hipMemcpyHostToDevice));
    return 0;
}

int Learner::upload_hp(hipStream_t st) {
    // only the float block (lr / clip / entropy / betas); the Adam counters stay device-resident
    memcpy(hp_stage_, &hp_host_, sizeof(DevHP));
    CDRL_HIP(hipMemcpyAsync(hp_dev_, hp_stage_, offsetof(DevHP, t_policy), hipMemcpyHostToDevice, st));
    tail_invalidate(st);
    return 0;
}

int Learner::reset_counters(hipStream_t st) {
    CDRL_HIP(hipMemsetAsync(reinterpret_cast<char*>(hp_dev_) + offsetof(DevHP, t_policy), 0, 3 * sizeof(int), st));
    tail_invalidate(st);
    return 0;
}

// ------------------------------------------------------------------------------------------
// execution
// ------------------------------------------------------------------------------------------
int Learner::run_fwd(std::vector<Op>& ops, hipStream_t st, int training) {
    for (Op& op : ops) CDRL_TRY(op.fwd(st, training));
    return 0;
}

int Learner::run_bwd(std::vector<Op>& ops, hipStream_t st) {
    for (size_t i = ops.size(); i-- > 0;) CDRL_TRY(ops[i].bwd(st));
    return 0;
}

int Learner::set_inputs(const float* image, const float* road, const float* vehicle, const float* navigation) {
    if (!ws_base_) {
        set_error("learner not bound");
        return -1;
    }
    if (!image || !road || !vehicle || !navigation) {
        set_error("null state input");
        return -1;
    }
    in_image_ = image;
    in_road_ = road;
    in_vehicle_ = vehicle;
    in_navigation_ = navigation;
    return 0;
}

int Learner::trunk_forward_train(const float* image, const float* road, const float* vehicle, const float* navigation,
                                 hipStream_t caller) {
    return launch(caller, {}, false, [&](hipStream_t st) -> int {
        CDRL_TRY(set_inputs(image, road, vehicle, navigation));
        return run_trunk_fwd(st, 1);
    });
}

int Learner::policy_forward(const float* image, const float* road, const float* vehicle, const float* navigation,
                            hipStream_t caller) {
    return launch(caller, {}, false, [&](hipStream_t st) -> int { return policy_forward_impl(image, road, vehicle, navigation, st); });
}

int Learner::policy_forward_impl(const float* image, const float* road, const float* vehicle, const float* navigation,
                                 hipStream_t st) {
    CDRL_TRY(set_inputs(image, road, vehicle, navigation));
    CDRL_TRY(run_trunk_fwd(st, 1));
    CDRL_TRY(run_fwd(policy_ops_, st, 1));
    // alpha, beta (+ mean, std) of the CURRENT policy for the Beta re-sampling
    return policy_dist(lin_p_.p, aux_p_, cfg_.B, cfg_.A, st);
}

int Learner::policy_backward(const PolicyBatch& b, float inv_world, hipStream_t caller) {
    if (inv_world != 1.0f) dp_hint_ = true;
    return launch(caller, {}, false, [&](hipStream_t st) -> int { return policy_backward_impl(b, inv_world, st); });
}

int Learner::policy_backward_impl(const PolicyBatch& b, float inv_world, hipStream_t st) {
    PolicyLossArgs a;
    a.lin = lin_p_.p;
    a.adv = b.adv;
    a.old_logp = b.old_logp;
    a.speed = b.speed;
    a.similarity = b.similarity;
    a.u = b.u;
    a.du_da = b.du_da;
    a.du_db = b.du_db;
    a.hp = reinterpret_cast<const float*>(hp_dev_);
    a.dlin = lin_p_.g;
    a.metrics = metrics_p_;
    a.aux = aux_p_;
    a.B = cfg_.B;
    a.A = cfg_.A;
    a.inv_world = inv_world;
    CDRL_TRY(policy_loss(a, st));
    CDRL_TRY(run_bwd(policy_ops_, st));
    if (!packs_have_wt_) CDRL_TRY(transpose_many(d_pwt_, (int)h_pwt_.size(), pwt_tiles_, st));      // W^T of the pointwise convs (normally packed with the forward's operands)
    CDRL_TRY(run_bwd(trunk_ops_, st));
    return join_side(st);
}

int Learner::policy_forward_backward_resample(const PolicyBatch& b, uint64_t seed, uint64_t offset, float inv_world,
                                              hipStream_t caller) {
    if (inv_world != 1.0f) dp_hint_ = true;
    // (seed, offset) are kernel arguments that change every call -> eager, not graph-replayed
    return launch(caller, {}, false, [&](hipStream_t st) -> int {
        CDRL_TRY(policy_forward_impl(b.image, b.road, b.vehicle, b.navigation, st));
        const int A = cfg_.A;
        CDRL_TRY(beta_sample(aux_p_, aux_p_ + A, cfg_.B, A, 4 * A, seed, offset, sample_u_, sample_da_, sample_db_, st));
        PolicyBatch r = b;
        r.u = sample_u_;
        r.du_da = sample_da_;
        r.du_db = sample_db_;
        return policy_backward_impl(r, inv_world, st);
    });
}

static inline uint64_t K(const void* p) { return (uint64_t)reinterpret_cast<uintptr_t>(p); }
static inline uint64_t Kf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

int Learner::policy_forward_backward(const PolicyBatch& b, float inv_world, hipStream_t caller) {
    if (inv_world != 1.0f) dp_hint_ = true;
    std::vector<uint64_t> key = {1, K(b.image), K(b.road), K(b.vehicle), K(b.navigation), K(b.adv), K(b.old_logp), K(b.speed),
                                 K(b.similarity), K(b.u), K(b.du_da), K(b.du_db), Kf(inv_world)};
    return launch(caller, key, true, [&](hipStream_t st) -> int {
        CDRL_TRY(set_inputs(b.image, b.road, b.vehicle, b.navigation));
        CDRL_TRY(run_trunk_fwd(st, 1));
        CDRL_TRY(run_fwd(policy_ops_, st, 1));
        return policy_backward_impl(b, inv_world, st);
    });
}

int Learner::value_forward_backward(const ValueBatch& b, float inv_world, hipStream_t caller) {
    if (inv_world != 1.0f) dp_hint_ = true;
    std::vector<uint64_t> key = {2, K(b.image), K(b.road), K(b.vehicle), K(b.navigation), K(b.returns), K(b.speed),
                                 K(b.similarity), Kf(inv_world)};
    return launch(caller, key, true, [&](hipStream_t st) -> int { return value_forward_backward_impl(b, inv_world, st); });
}

int Learner::value_forward_backward_impl(const ValueBatch& b, float inv_world, hipStream_t st) {
    CDRL_TRY(set_inputs(b.image, b.road, b.vehicle, b.navigation));
    CDRL_TRY(run_trunk_fwd(st, 1));
    CDRL_TRY(run_fwd(value_ops_, st, 1));
    ValueLossArgs a;
    a.lin = lin_v_.p;
    a.returns = b.returns;
    a.speed = b.speed;
    a.similarity = b.similarity;
    a.dlin = lin_v_.g;
    a.metrics = metrics_v_;
    a.values = aux_v_;
    a.B = cfg_.B;
    a.exp_scale = cfg_.exp_scale;
    a.inv_world = inv_world;
    CDRL_TRY(value_loss(a, st));
    CDRL_TRY(run_bwd(value_ops_, st));
    if (!packs_have_wt_) CDRL_TRY(transpose_many(d_pwt_, (int)h_pwt_.size(), pwt_tiles_, st));
    CDRL_TRY(run_bwd(trunk_ops_, st));
    return join_side(st);
}

int Learner::update_old_policy(hipStream_t caller) {
    return launch(caller, {}, false, [&](hipStream_t st) -> int { return update_old_policy_impl(st); });
}

int Learner::update_old_policy_impl(hipStream_t st) {
    // old_policy.set_weights(policy.get_weights()): all weights incl. BN moving statistics
    // (reference core/networks.py:281-285)
    return copy_two(buf_.params + tr_offset(M_OLD_POLICY), buf_.params + tr_offset(M_POLICY), tr_size_[M_POLICY],
                    buf_.params + st_offset(M_OLD_POLICY), buf_.params + st_offset(M_POLICY), st_size_[M_POLICY], st);
}

int Learner::policy_apply(hipStream_t caller) {
    return launch(caller, {3}, true, [&](hipStream_t st) -> int { return policy_apply_impl(st); });
}

int Learner::policy_apply_impl(hipStream_t st) {
    // order: trunk Adam (unclipped, F9) -> clip -> old_policy <- policy -> policy Adam
    const int64_t to = tr_offset(M_TRUNK), po = tr_offset(M_POLICY);
    CDRL_TRY(clip_adam(buf_.params + to, buf_.grads + to, buf_.adam_m + to, buf_.adam_v + to, tr_size_[M_TRUNK], nullptr,
                       nullptr, 0, nullptr, nullptr, hp_dev_, 2, st));
    SegTable& s = seg_[M_POLICY];
    // four launches instead of seven (round 6): the chunk kernel of the norms also advances the trunk's step counter (its update ran in front)
    // and the policy's (its update runs behind and is told so); the per-tensor fold of the chunk partials happens inside clip_adam
    CDRL_TRY(tensor_sqnorms(buf_.grads + po, s.segs, s.ntensors, s.chunk_tensor, s.chunk_off, s.nchunks, s.chunk_part,
                            s.sqnorms, st, hp_dev_, 4 | 1, true));
    CDRL_TRY(update_old_policy_impl(st));
    return clip_adam(buf_.params + po, buf_.grads + po, buf_.adam_m + po, buf_.adam_v + po, tr_size_[M_POLICY], s.chunk_tensor,
                     s.chunk_off, s.nchunks, s.segs, nullptr, hp_dev_, 0, st, s.chunk_part, 1);
}

int Learner::value_apply(hipStream_t caller) {
    return launch(caller, {4}, true, [&](hipStream_t st) -> int { return value_apply_impl(st); });
}

int Learner::value_apply_impl(hipStream_t st) {
    const int64_t to = tr_offset(M_TRUNK), vo = tr_offset(M_VALUE);
    CDRL_TRY(clip_adam(buf_.params + to, buf_.grads + to, buf_.adam_m + to, buf_.adam_v + to, tr_size_[M_TRUNK], nullptr,
                       nullptr, 0, nullptr, nullptr, hp_dev_, 2, st));
    SegTable& s = seg_[M_VALUE];
    CDRL_TRY(tensor_sqnorms(buf_.grads + vo, s.segs, s.ntensors, s.chunk_tensor, s.chunk_off, s.nchunks, s.chunk_part,
                            s.sqnorms, st, hp_dev_, 4 | 2, true));       // (see policy_apply_impl)
    return clip_adam(buf_.params + vo, buf_.grads + vo, buf_.adam_m + vo, buf_.adam_v + vo, tr_size_[M_VALUE], s.chunk_tensor,
                     s.chunk_off, s.nchunks, s.segs, nullptr, hp_dev_, 1, st, s.chunk_part, 1);
}

int Learner::predict(const float* image, const float* road, const float* vehicle, const float* navigation,
                     float* dist_out, float* value_out, float* dyn_out, hipStream_t caller) {
    std::vector<uint64_t> key = {5, K(image), K(road), K(vehicle), K(navigation), K(dist_out), K(value_out), K(dyn_out)};
    return launch(caller, key, true, [&](hipStream_t st) -> int {
        return predict_impl(image, road, vehicle, navigation, dist_out, value_out, dyn_out, st);
    });
}

int Learner::predict_impl(const float* image, const float* road, const float* vehicle, const float* navigation,
                          float* dist_out, float* value_out, float* dyn_out, hipStream_t st) {
    CDRL_TRY(set_inputs(image, road, vehicle, navigation));
    CDRL_TRY(run_trunk_fwd(st, 0));
    CDRL_TRY(run_fwd(old_policy_ops_, st, 0));
    CDRL_TRY(policy_dist(lin_old_.p, dist_out, cfg_.B, cfg_.A, st));
    CDRL_TRY(run_fwd(value_ops_, st, 0));
    CDRL_TRY(value_act(lin_v_.p, value_out, cfg_.B, cfg_.exp_scale, st));
    if (dyn_out)
        CDRL_HIP(hipMemcpyAsync(dyn_out, dyn_.p, (size_t)cfg_.B * cfg_.dyn * sizeof(float), hipMemcpyDeviceToDevice, st));
    tail_invalidate(st);
    return 0;
}

}  // namespace cdrl
