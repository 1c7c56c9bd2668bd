// extern "C" surface of libcdrl_hip.so (declared in include/cdrl.h).
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/cdrl.h"
#include "engine.h"

using namespace cdrl;

struct cdrl_learner {
    Learner* impl;
};

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define CHECK_L(l)                         \
    if (!(l) || !(l)->impl) {              \
        cdrl::set_error("null learner");   \
        return -1;                         \
    }

// act_type of the op-level entry points below: element type of their ACTIVATION tensors -- 0 float32, 1 bf16 (configuration 3's storage)
static inline bool bad_act_type(int at) {
    if (at == 0 || at == 1) return false;
    cdrl::set_error("act_type: 0 (float32) or 1 (bf16)");
    return true;
}

extern "C" {

const char* cdrl_last_error(void) { return cdrl::last_error(); }
int cdrl_version(void) { return CDRL_VERSION; }
int cdrl_env_overrides(char* buf, int cap) { return cdrl::env_overrides(buf, cap); }
int cdrl_diag_active(void) { return cdrl::diag_active(); }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host: checksums of the TF-checkpoint-V2
// writer (tf_checkpoint.py): every SSTable block and every tensor entry carries one
namespace {
struct Crc32cTable {
    uint32_t t[8][256];
    Crc32cTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};
}  // namespace

uint32_t cdrl_crc32c(uint32_t crc, const void* data, size_t n) {
    // function-local static: initialised exactly once, thread-safe (ctypes releases the GIL around the call)
    static const Crc32cTable table;
    const uint32_t (*T)[256] = table.t;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^ T[4][lo >> 24] ^ T[3][hi & 0xff] ^
            T[2][(hi >> 8) & 0xff] ^ T[1][(hi >> 16) & 0xff] ^ T[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xff];
    return ~c;
}

void cdrl_config_default(cdrl_config* c) {
    if (!c) return;
    Config d;
    c->B = d.B; c->T = d.T; c->H = d.H; c->W = d.W;
    c->road = d.road; c->vehicle = d.vehicle; c->navigation = d.navigation; c->A = d.A;
    c->stem = d.stem;
    for (int i = 0; i < 3; ++i) { c->stage_c[i] = d.stage_c[i]; c->stage_n[i] = d.stage_n[i]; }
    c->last = d.last; c->feat = d.feat; c->rnn_image = d.rnn_image; c->rnn_small = d.rnn_small;
    c->dyn = d.dyn; c->head = d.head; c->exp_scale = d.exp_scale;
    c->compute = d.compute;
}

int cdrl_learner_create(const cdrl_config* c, cdrl_learner** out) {
    if (!c || !out) {
        cdrl::set_error("cdrl_learner_create: null argument");
        return -1;
    }
    if (c->B < 1 || c->T < 1 || c->H < 35 || c->W < 35 || c->A < 1 || c->A > 8) {
        cdrl::set_error("cdrl_learner_create: bad geometry B=%d T=%d H=%d W=%d A=%d", c->B, c->T, c->H, c->W, c->A);
        return -1;
    }
    for (int i = 0; i < 3; ++i)
        if (c->stage_c[i] % 4 != 0 || c->stage_n[i] < 1) {
            cdrl::set_error("cdrl_learner_create: stage channels must be multiples of 4");
            return -1;
        }
    Config d;
    d.B = c->B; d.T = c->T; d.H = c->H; d.W = c->W;
    d.road = c->road; d.vehicle = c->vehicle; d.navigation = c->navigation; d.A = c->A;
    d.stem = c->stem;
    for (int i = 0; i < 3; ++i) { d.stage_c[i] = c->stage_c[i]; d.stage_n[i] = c->stage_n[i]; }
    d.last = c->last; d.feat = c->feat; d.rnn_image = c->rnn_image; d.rnn_small = c->rnn_small;
    d.dyn = c->dyn; d.head = c->head; d.exp_scale = c->exp_scale;
    if (c->compute != CDRL_COMPUTE_F32 && c->compute != CDRL_COMPUTE_BF16_OPERANDS && c->compute != CDRL_COMPUTE_BF16_STORAGE) {
        cdrl::set_error("cdrl_learner_create: unknown compute mode %d", c->compute);
        return -1;
    }
    d.compute = c->compute;
    if (cdrl::diag_active()) {      // loud, every time: results of this learner are WRONG by request (timing diagnostics)
        char ov[2048];
        cdrl::env_overrides(ov, (int)sizeof(ov));
        fprintf(stderr, "[cdrl] WARNING: CDRL_DIAG=1 with %d wrong-result diagnostic switch(es) active -- environment: %s\n",
                cdrl::diag_active(), ov);
    }
    cdrl_learner* l = new (std::nothrow) cdrl_learner;
    if (!l) return -3;
    l->impl = new (std::nothrow) Learner(d);
    if (!l->impl) {
        delete l;
        return -3;
    }
    if (!l->impl->build_error().empty()) {
        cdrl::set_error("cdrl_learner_create: %s", l->impl->build_error().c_str());
        delete l->impl;
        delete l;
        return -1;
    }
    *out = l;
    return 0;
}

void cdrl_learner_destroy(cdrl_learner* l) {
    if (!l) return;
    delete l->impl;
    delete l;
}

int cdrl_learner_param_count(const cdrl_learner* l, int model) {
    CHECK_L(l);
    if (model == CDRL_OLD_POLICY) model = CDRL_POLICY;
    if (model < 0 || model > 2) return -1;
    return (int)l->impl->params(model).size();
}

int cdrl_learner_param_info(const cdrl_learner* l, int model, int index, cdrl_param_info* out) {
    CHECK_L(l);
    if (model == CDRL_OLD_POLICY) model = CDRL_POLICY;
    if (model < 0 || model > 2 || !out) return -1;
    const auto& v = l->impl->params(model);
    if (index < 0 || index >= (int)v.size()) {
        cdrl::set_error("param index %d out of range", index);
        return -1;
    }
    const ParamInfo& p = v[index];
    memset(out, 0, sizeof(*out));
    strncpy(out->name, p.name.c_str(), sizeof(out->name) - 1);
    for (int i = 0; i < 4; ++i) out->shape[i] = p.shape[i];
    out->ndim = p.ndim;
    out->trainable = p.trainable;
    out->numel = p.numel;
    out->offset = p.off;
    return 0;
}

int64_t cdrl_learner_region_offset(const cdrl_learner* l, int model, int trainable) {
    CHECK_L(l);
    return trainable ? l->impl->tr_offset(model) : l->impl->st_offset(model);
}

int64_t cdrl_learner_region_elems(const cdrl_learner* l, int model, int trainable) {
    CHECK_L(l);
    if (model == CDRL_OLD_POLICY) model = CDRL_POLICY;
    return trainable ? l->impl->trainable_elems(model) : l->impl->state_elems(model);
}

int64_t cdrl_learner_params_total(const cdrl_learner* l) {
    CHECK_L(l);
    return l->impl->params_total();
}

int64_t cdrl_learner_grads_total(const cdrl_learner* l) {
    CHECK_L(l);
    return l->impl->grads_total();
}

size_t cdrl_learner_workspace_bytes(const cdrl_learner* l) {
    if (!l || !l->impl) return 0;
    return l->impl->workspace_bytes();
}

int cdrl_learner_bind(cdrl_learner* l, float* params, float* grads, float* adam_m, float* adam_v, void* workspace,
                      size_t workspace_bytes) {
    CHECK_L(l);
    Buffers b;
    b.params = params;
    b.grads = grads;
    b.adam_m = adam_m;
    b.adam_v = adam_v;
    b.workspace = workspace;
    b.workspace_bytes = workspace_bytes;
    return l->impl->bind(b);
}

int cdrl_learner_set_hparams(cdrl_learner* l, const cdrl_hparams* hp, void* stream) {
    CHECK_L(l);
    if (!hp) return -1;
    DevHP* h = l->impl->host_hp();
    h->lr_policy = hp->policy_lr;
    h->lr_value = hp->value_lr;
    h->lr_dynamics = hp->dynamics_lr;
    h->clip_ratio = hp->clip_ratio;
    h->entropy_coef = hp->entropy_coef;
    h->clip_norm_policy = hp->clip_norm_policy;
    h->clip_norm_value = hp->clip_norm_value;
    h->beta1 = hp->beta1;
    h->beta2 = hp->beta2;
    h->eps = hp->eps;
    return l->impl->upload_hp(S(stream));
}

int cdrl_learner_share_hparams(cdrl_learner* l, const cdrl_learner* owner) {
    CHECK_L(l);
    if (!owner || !owner->impl || !owner->impl->dev_hp() || !l->impl->dev_hp()) {
        cdrl::set_error("cdrl_learner_share_hparams: both learners must be bound");
        return -1;
    }
    l->impl->share_hp(*owner->impl);
    return 0;
}

int cdrl_learner_set_comm_stream(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    l->impl->set_comm_stream(S(stream));
    return 0;
}

int64_t cdrl_learner_tail_offset(const cdrl_learner* l) {
    if (!l) return -1;
    // with hipGraph replay the communication stream is never released mid-pass: no early bucket (everything is "tower")
    if (l->impl->graphs_enabled()) return l->impl->trainable_elems(cdrl::M_TRUNK);
    return l->impl->tail_offset();
}

int cdrl_learner_reset_optimizer_steps(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->reset_counters(S(stream));
}

int cdrl_learner_policy_forward_backward(cdrl_learner* l, const cdrl_policy_batch* b, float grad_scale, void* stream) {
    CHECK_L(l);
    if (!b) return -1;
    PolicyBatch pb{b->image, b->road, b->vehicle, b->navigation, b->advantages, b->old_log_prob,
                   b->speed, b->similarity, b->u, b->du_dalpha, b->du_dbeta};
    if (!pb.adv || !pb.old_logp || !pb.speed || !pb.similarity || !pb.u) {
        cdrl::set_error("policy batch: null tensor");
        return -1;
    }
    return l->impl->policy_forward_backward(pb, grad_scale, S(stream));
}

int cdrl_learner_policy_forward(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                                const float* navigation, void* stream) {
    CHECK_L(l);
    return l->impl->policy_forward(image, road, vehicle, navigation, S(stream));
}

int cdrl_learner_policy_backward(cdrl_learner* l, const cdrl_policy_batch* b, float grad_scale, void* stream) {
    CHECK_L(l);
    if (!b) return -1;
    PolicyBatch pb{b->image, b->road, b->vehicle, b->navigation, b->advantages, b->old_log_prob,
                   b->speed, b->similarity, b->u, b->du_dalpha, b->du_dbeta};
    if (!pb.adv || !pb.old_logp || !pb.speed || !pb.similarity || !pb.u) {
        cdrl::set_error("policy batch: null tensor");
        return -1;
    }
    return l->impl->policy_backward(pb, grad_scale, S(stream));
}

int cdrl_learner_policy_forward_backward_resample(cdrl_learner* l, const cdrl_policy_batch* b, uint64_t seed,
                                                 uint64_t offset, float grad_scale, void* stream) {
    CHECK_L(l);
    if (!b) return -1;
    PolicyBatch pb{b->image, b->road, b->vehicle, b->navigation, b->advantages, b->old_log_prob,
                   b->speed, b->similarity, nullptr, nullptr, nullptr};
    if (!pb.adv || !pb.old_logp || !pb.speed || !pb.similarity) {
        cdrl::set_error("policy batch: null tensor");
        return -1;
    }
    return l->impl->policy_forward_backward_resample(pb, seed, offset, grad_scale, S(stream));
}

int cdrl_beta_sample_logp(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                          float* u, float* log_prob, void* stream) {
    if (!alpha || !beta || !u || !log_prob) {
        cdrl::set_error("cdrl_beta_sample_logp: null argument");
        return -1;
    }
    return beta_sample(alpha, beta, rows, A, ld, seed, offset, u, nullptr, nullptr, S(stream), log_prob);
}

int cdrl_beta_sample(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                     float* u, float* du_dalpha, float* du_dbeta, void* stream) {
    if (!alpha || !beta || !u) {
        cdrl::set_error("cdrl_beta_sample: null argument");
        return -1;
    }
    return beta_sample(alpha, beta, rows, A, ld, seed, offset, u, du_dalpha, du_dbeta, S(stream));
}

int cdrl_gamma_implicit_grad(const double* a, const double* g, int n, double* out, void* stream) {
    return gamma_implicit_grad(a, g, n, out, S(stream));
}

int cdrl_beta_sample_gammas(const float* alpha, const float* beta, int rows, int A, int ld, uint64_t seed, uint64_t offset,
                            double* gammas, void* stream) {
    if (!alpha || !beta || !gammas) {
        cdrl::set_error("cdrl_beta_sample_gammas: null argument");
        return -1;
    }
    return beta_sample(alpha, beta, rows, A, ld, seed, offset, nullptr, nullptr, nullptr, S(stream), nullptr, gammas);
}

int cdrl_philox_words(uint64_t seed, uint64_t offset, uint64_t idx0, int n, int nblocks, uint32_t* out, void* stream) {
    if (!out) {
        cdrl::set_error("cdrl_philox_words: null output");
        return -1;
    }
    return philox_words(seed, offset, idx0, n, nblocks, out, S(stream));
}

int cdrl_learner_sequence_begin(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->sequence_begin(S(stream));
}

int cdrl_learner_sequence_end(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->sequence_end(S(stream));
}

int cdrl_learner_policy_apply(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->policy_apply(S(stream));
}

int cdrl_learner_value_forward_backward(cdrl_learner* l, const cdrl_value_batch* b, float grad_scale, void* stream) {
    CHECK_L(l);
    if (!b) return -1;
    ValueBatch vb{b->image, b->road, b->vehicle, b->navigation, b->returns, b->speed, b->similarity};
    if (!vb.returns || !vb.speed || !vb.similarity) {
        cdrl::set_error("value batch: null tensor");
        return -1;
    }
    return l->impl->value_forward_backward(vb, grad_scale, S(stream));
}

int cdrl_learner_value_apply(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->value_apply(S(stream));
}

int cdrl_learner_update_old_policy(cdrl_learner* l, void* stream) {
    CHECK_L(l);
    return l->impl->update_old_policy(S(stream));
}

int cdrl_learner_predict(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                         const float* navigation, float* dist_out, float* value_out, float* dynamics_out, void* stream) {
    CHECK_L(l);
    if (!dist_out || !value_out) return -1;
    return l->impl->predict(image, road, vehicle, navigation, dist_out, value_out, dynamics_out, S(stream));
}

int cdrl_learner_trunk_forward_train(cdrl_learner* l, const float* image, const float* road, const float* vehicle,
                                     const float* navigation, void* stream) {
    CHECK_L(l);
    return l->impl->trunk_forward_train(image, road, vehicle, navigation, S(stream));
}

int cdrl_learner_get_buffer(const cdrl_learner* l, int which, float** ptr, int64_t* elems) {
    CHECK_L(l);
    if (!ptr || !elems) return -1;
    const Config& c = l->impl->config();
    switch (which) {
        case CDRL_BUF_DYNAMICS: *ptr = l->impl->dyn_out(); *elems = (int64_t)c.B * c.dyn; break;
        case CDRL_BUF_IMG_FEAT: *ptr = l->impl->img_feat(); *elems = (int64_t)c.B * c.T * c.last; break;
        case CDRL_BUF_METRICS_P: *ptr = l->impl->metrics_policy(); *elems = 16; break;
        case CDRL_BUF_METRICS_V: *ptr = l->impl->metrics_value(); *elems = 16; break;
        case CDRL_BUF_AUX_P: *ptr = l->impl->policy_aux(); *elems = (int64_t)c.B * 4 * c.A; break;
        case CDRL_BUF_AUX_V: *ptr = l->impl->value_aux(); *elems = (int64_t)c.B * 2; break;
        case CDRL_BUF_LIN_P: *ptr = l->impl->policy_lin(); *elems = (int64_t)c.B * (2 * c.A + 2); break;
        case CDRL_BUF_LIN_V: *ptr = l->impl->value_lin(); *elems = (int64_t)c.B * 4; break;
        case CDRL_BUF_SAMPLE: *ptr = l->impl->sample_buffer(); *elems = (int64_t)c.B * c.A; break;
        default: cdrl::set_error("unknown buffer id %d", which); return -1;
    }
    return 0;
}

int cdrl_learner_named_buffer(const cdrl_learner* l, const char* name, void** ptr, int64_t* bytes) {
    CHECK_L(l);
    if (!name || !ptr || !bytes) return -1;
    if (!l->impl->named_buffer(name, ptr, bytes)) {
        cdrl::set_error("unknown named buffer %s (or learner not bound)", name);
        return -1;
    }
    return 0;
}

int cdrl_learner_check_guards(cdrl_learner* l, void* stream, int64_t* bad_bands, int64_t* first_bad_offset) {
    CHECK_L(l);
    return l->impl->check_guards(S(stream), bad_bands, first_bad_offset);
}

int cdrl_gae_returns(const float* rewards, const float* values_be, int N, double gamma, double lambda, float scale,
                     float* returns, float* returns_be, float* adv_raw, float* adv, double* scratch, void* stream) {
    if (!rewards || !values_be || !returns || !returns_be || !adv_raw || !adv || !scratch) {
        cdrl::set_error("cdrl_gae_returns: null argument");
        return -1;
    }
    return gae_returns(rewards, values_be, N, gamma, lambda, scale, returns, returns_be, adv_raw, adv, scratch, S(stream));
}

int64_t cdrl_pwconv_x3_packed_bytes(int K) { return pw_x3_packed_bytes(K); }
int64_t cdrl_pwconv_x3_packed_bytes_n(int K, int N) { return pw_x3_packed_bytes_n(K, N); }
int cdrl_pwconv_x3_partial_rows(int G, int Mg, int N, int K) { return pw_x3_partial_rows(G, Mg, N, K); }

int cdrl_pwconv_x3_pack(const float* W, int K, int N, int sbk, int sbn, void* packed, void* stream) {
    if (!W || !packed) return -1;
    // one-entry table through a temporary device copy (test / tooling entry point; the engine packs all layers in one launch)
    PwX3Pack e = pw_x3_pack_entry(W, packed, K, N, sbk, sbn);
    PwX3Pack* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(e)) != hipSuccess) return -2;
    int rc = hipMemcpy(d, &e, sizeof(e), hipMemcpyHostToDevice) == hipSuccess ? pw_x3_pack_many(d, 1, S(stream)) : -2;
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(d);
    return rc;
}

int cdrl_pwconv_x3(const float* A, int lda, int a_coff, const float* pro_stats, const void* W_packed, const float* bias, float* C,
                   int ldc, int c_coff, int G, int Mg, int N, int K, double* part, void* stream) {
    if (!A || !W_packed || !C) {
        cdrl::set_error("cdrl_pwconv_x3: null argument");
        return -1;
    }
    return pw_x3(make_view(const_cast<float*>(A), lda, a_coff), pro_stats, W_packed, bias, make_view(C, ldc, c_coff), G, Mg, N, K,
                 part, S(stream));
}

int cdrl_pwconv_x3_wide_bwd_rows(int Mg) { return pw_x3_wide_bwd_rows(Mg); }

int cdrl_pwconv_x3_wide_bwd(const float* dz, int ld_dz, int dz_coff, int dz_shuffle, int act, const float* y, const float* stats,
                            const float* coef, const void* W_packed, float* da, int ldda, int da_coff, int accumulate, int G, int Mg,
                            int Cin, int Cout, double* part2, const float* ey, const float* epi_stats, double* part, void* stream) {
    if (!dz || !y || !stats || !coef || !W_packed || !da) {
        cdrl::set_error("cdrl_pwconv_x3_wide_bwd: null argument");
        return -1;
    }
    PwBnBwd bb{y, stats, coef, dz_shuffle, act ? ACT_RELU6 : ACT_NONE, part2};
    return pw_x3_wide_bwd(make_view(const_cast<float*>(dz), ld_dz, dz_coff), bb, W_packed, make_view(da, ldda, da_coff), accumulate, G, Mg,
                          Cin, Cout, ey, epi_stats, part, S(stream));
}

int64_t cdrl_pwconv_bwd_fused_workspace(int G, int Mg, int N, int K, int which, int act_type) {
    return which == 0 ? pw_bwd_fused_qpart_elems(G, Mg, N, K, act_type) : pw_bwd_fused_dbpart_elems(G, Mg, N, K, act_type);
}

int cdrl_pwconv_bwd_fused(const float* dz, int ld_dz, int dz_coff, int dz_shuffle, int act, const float* y, const float* stats,
                          const float* coef, const float* a, int lda, int a_coff, const float* a_stats, const float* a_gamma,
                          const float* a_beta, float* a_dgamma, float* a_dbeta, float* a_coef, const float* W, const void* W_packed,
                          float* da, int ldda, int da_coff, int accumulate, float* dW, float* db, float* qpart, double* dbpart, int G,
                          int Mg, int N, int K, int act_type, void* stream) {
    if (!dz || !y || !stats || !coef || !a || !W || !W_packed || !da || !dW || !db || !qpart || !dbpart) {
        cdrl::set_error("cdrl_pwconv_bwd_fused: null argument");
        return -1;
    }
    PwBwdFused f;
    f.dz = make_view(const_cast<float*>(dz), ld_dz, dz_coff);
    f.dz_shuffle = dz_shuffle;
    f.act = act;
    f.y = y;
    f.stats = stats;
    f.coef = coef;
    f.a = make_view(const_cast<float*>(a), lda, a_coff);
    f.a_stats = a_stats;
    f.a_gamma = a_gamma;
    f.a_beta = a_beta;
    f.a_dgamma = a_dgamma;
    f.a_dbeta = a_dbeta;
    f.a_coef = a_coef;
    f.W = W;
    f.Wp = W_packed;
    f.da = make_view(da, ldda, da_coff);
    f.accumulate = accumulate;
    f.dW = dW;
    f.db = db;
    f.qpart = qpart;
    f.dbpart = dbpart;
    f.G = G;
    f.Mg = Mg;
    f.N = N;
    f.K = K;
    f.at = act_type;
    CDRL_TRY(pw_bwd_fused(f, S(stream)));
    return pw_bwd_fused_reduce(f, S(stream));
}

int64_t cdrl_gemm_x3_packed_bytes(int N, int K) { return gemm_x3_packed_bytes(N, K); }

int cdrl_gemm_x3_pack(const float* B, int K, int N, int sbk, int sbn, void* packed, void* stream) {
    if (!B || !packed) return -1;
    GemmX3Pack e = gemm_x3_pack_entry(B, packed, K, N, sbk, sbn);
    GemmX3Pack* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(e)) != hipSuccess) return -2;      // tooling entry point; the engine packs in one launch
    int rc = hipMemcpy(d, &e, sizeof(e), hipMemcpyHostToDevice) == hipSuccess ? gemm_x3_pack_many(d, 1, S(stream)) : -2;
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(d);
    return rc;
}

int cdrl_gemm_x3(const float* A, int lda, int a_coff, const void* B_packed, const float* bias, float* C, int ldc, int c_coff, int M,
                 int N, int K, int accumulate, int act_type, void* stream) {
    if (!A || !B_packed || !C) {
        cdrl::set_error("cdrl_gemm_x3: null argument");
        return -1;
    }
    return gemm_x3(make_view(const_cast<float*>(A), lda, a_coff), B_packed, bias, make_view(C, ldc, c_coff), M, N, K, accumulate,
                   S(stream), act_type != 0, act_type);
}

int cdrl_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y) return -1;
    return f32_to_bf16(x, y, n, S(stream));
}

int cdrl_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return -1;
    return bf16_to_f32(x, y, n, S(stream));
}

int cdrl_pwconv_bf16_partial_rows(int G, int Mg, int N, int K) { return pw_bf16_partial_rows(G, Mg, N, K); }

int64_t cdrl_pwconv_bf16_packed_elems(int K) { return pw_bf16_packed_elems(K); }

int cdrl_pwconv_bf16_pack(const float* W, int K, int N, void* packed, void* stream) {
    if (!W || !packed) return -1;
    return pw_bf16_pack(W, K, N, packed, S(stream));
}

int cdrl_pwconv_bf16(const void* A, int lda, int a_coff, const float* pro_stats, const float* W, const void* W_packed,
                     const float* bias, void* C, int ldc, int c_coff, int G, int Mg, int N, int K, double* part, void* stream) {
    if (!A || (!W && !W_packed) || !C) {
        cdrl::set_error("cdrl_pwconv_bf16: null argument");
        return -1;
    }
    return pw_bf16(A, lda, a_coff, pro_stats, W, bias, C, ldc, c_coff, G, Mg, N, K, part, S(stream), W_packed);
}

int cdrl_gru_step_fwd(const float* xp, const float* hprev, const float* R, const float* b1, float* z, float* r, float* hh,
                      float* hp, float* hnew, int B, int u, void* stream) {
    if (!xp || !hprev || !R || !b1 || !z || !r || !hh || !hp || !hnew) {
        cdrl::set_error("cdrl_gru_step_fwd: null argument");
        return -1;
    }
    return gru_step_fwd(xp, hprev, R, b1, z, r, hh, hp, hnew, View{nullptr, 0, 0}, B, u, S(stream));
}

int cdrl_gru_step_bwd(const float* dh, int ld_dh, const float* z, const float* r, const float* hh, const float* hp,
                      const float* hprev, const float* RT, float* dxp, float* dhp, float* dhprev, int B, int u, void* stream) {
    if (!dh || !z || !r || !hh || !hp || !hprev || !RT || !dxp || !dhp) {
        cdrl::set_error("cdrl_gru_step_bwd: null argument");
        return -1;
    }
    return gru_step_bwd(make_view(const_cast<float*>(dh), ld_dh), z, r, hh, hp, hprev, RT, dxp, dhp, dhprev, B, u, S(stream));
}

int cdrl_gather_rows(const float* src, const int32_t* idx, float* dst, int nrows, int64_t row_elems, void* stream) {
    if (!src || !idx || !dst) {
        cdrl::set_error("cdrl_gather_rows: null argument");
        return -1;
    }
    return gather_rows(src, idx, dst, nrows, row_elems, S(stream));
}

int cdrl_gemm_nn(const float* A, int lda, int a_coff, const float* B, int sbk, int sbn, const float* bias, float* C,
                 int ldc, int c_coff, int M, int N, int K, int accumulate, void* stream) {
    return gemm_nn(make_view(const_cast<float*>(A), lda, a_coff), B, sbk, sbn, bias, make_view(C, ldc, c_coff), M, N, K,
                   accumulate, S(stream));
}

int64_t cdrl_gemm_tn_workspace_elems(int M, int N, int K) { return gemm_tn_part_elems(M, N, K); }

int cdrl_gemm_tn(const float* A, int lda, int a_coff, const float* D, int ldd, int d_coff, float* out, int M, int N,
                 int K, float* workspace, int accumulate, int act_type, void* stream) {
    return gemm_tn(make_view(const_cast<float*>(A), lda, a_coff), make_view(const_cast<float*>(D), ldd, d_coff), out, M, N,
                   K, workspace, accumulate, S(stream), 1, nullptr, nullptr, act_type != 0, act_type);
}

int cdrl_stem_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int H, int W, int Cout,
                  void* stream) {
    return stem_fwd(x, w, bias, y, B, T, H, W, Cout, S(stream));
}

int cdrl_stem_fwd_stats_rows(int B, int T, int H, int W, int Cout) { return stem_fwd_stats_nb(B, T, H, W, Cout); }

int cdrl_stem_fwd_stats(const float* x, const float* w, const float* bias, float* y, double* part, int B, int T, int H, int W, int Cout,
                        int act_type, void* stream) {
    return stem_fwd_stats(x, w, bias, y, part, B, T, H, W, Cout, S(stream), act_type);
}

int64_t cdrl_stem_bwd_workspace_doubles(int B, int T, int H, int W, int Cout) {
    return stem_bwd_part_elems(B, T, H, W, Cout);
}

int cdrl_stem_bwd_filter(const float* x, const float* dy, float* dw, float* db, int B, int T, int H, int W, int Cout,
                         double* workspace, void* stream) {
    return stem_bwd_filter(x, dy, dw, db, B, T, H, W, Cout, workspace, S(stream));
}

int cdrl_dwconv_fwd(const float* a, const float* w, const float* bias, float* y, int N, int H, int W, int C, int stride,
                    void* stream) {
    return dw_fwd(make_view(const_cast<float*>(a), C), w, bias, y, N, H, W, C, stride, S(stream));
}

int cdrl_dwconv_bwd_data(const float* dy, const float* w, float* da, int N, int H, int W, int C, int stride, void* stream) {
    return dw_bwd_data(dy, w, make_view(da, C), N, H, W, C, stride, 0, S(stream));
}

int64_t cdrl_dwconv_bwd_workspace_doubles(int N, int H, int W, int C, int stride) {
    return dw_bwd_part_elems(N, H, W, C, stride);
}

int cdrl_dwconv_bwd_filter(const float* a, const float* dy, float* dw, float* db, int N, int H, int W, int C, int stride,
                           double* workspace, void* stream) {
    return dw_bwd_filter(make_view(const_cast<float*>(a), C), dy, dw, db, N, H, W, C, stride, workspace, S(stream));
}

int64_t cdrl_augment_workspace_floats(int T, int H, int W) { return (int64_t)2 * T * H * W * 3 + 5 * T; }

int cdrl_augment_images(const float* in, float* out, int T, int H, int W, const cdrl_aug_plan* plan, float* workspace,
                        void* stream) {
    static_assert(sizeof(cdrl_aug_plan) == sizeof(AugPlan), "cdrl_aug_plan / AugPlan layout mismatch");
    if (!plan || in == out) {
        set_error("cdrl_augment_images: null plan or aliased in/out");
        return -1;
    }
    AugPlan p;
    memcpy(&p, plan, sizeof(p));
    return augment_images(in, out, T, H, W, p, workspace, S(stream));
}

int64_t cdrl_stem_block_bwd_workspace_doubles(int B, int T, int H, int W, int Cout) {
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const int Hp = same_out(Ho, 2), Wp = same_out(Wo, 2);
    return (int64_t)T * vcol_geom(B * Hp * Wp, Cout).nb * 2 * Cout + stem_bwd_part_elems(B, T, H, W, Cout);
}

int cdrl_stem_block_bwd(const float* x, const float* y, const float* stats, const uint8_t* argmax, const float* dp, int B, int T,
                        int H, int W, int Cout, float* dgamma, float* dbeta, float* coef, float* dw, float* db,
                        double* workspace, int act_type, void* stream) {
    return cdrl_stem_block_bwd_pooled(x, y, stats, argmax, dp, nullptr, B, T, H, W, Cout, dgamma, dbeta, coef, dw, db, workspace, act_type, stream);
}

int cdrl_stem_block_bwd_pooled(const float* x, const float* y, const float* stats, const uint8_t* argmax, const float* dp,
                               const float* pooled, int B, int T, int H, int W, int Cout, float* dgamma, float* dbeta, float* coef,
                               float* dw, float* db, double* workspace, int act_type, void* stream) {
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const int Hp = same_out(Ho, 2), Wp = same_out(Wo, 2);
    hipStream_t st = S(stream);
    PoolSrc ps = make_pool_src(argmax, dp, Ho, Wo);
    ps.pa = pooled;
    const int nb = vcol_geom(B * Hp * Wp, Cout).nb;
    double* part = workspace;
    double* fpart = workspace + (int64_t)T * nb * 2 * Cout;
    CDRL_TRY(pool_bn_bwd_reduce(ps, y, T, B, Cout, stats, part, st, act_type));
    CDRL_TRY(bn_bwd_finalize(part, nb, T, B * Ho * Wo, Cout, stats, dgamma, dbeta, coef, st));
    return stem_bwd_filter_fused(x, ps, y, stats, coef, dw, db, B, T, H, W, Cout, fpart, st, act_type);
}

int cdrl_pwconv_fused_partial_rows(int G, int Mg, int N, int K) { return pw_nn_plan(G, Mg, N, K).nbpg; }

int cdrl_pwconv_fused(const float* a, int lda, int a_coff, const float* pro_stats, const float* w, int sbk, int sbn,
                      const float* bias, float* c, int ldc, int c_coff, int accumulate, int G, int Mg, int N, int K,
                      int epilogue, const float* epi_y, const float* epi_stats, double* part, void* stream) {
    return pw_nn(make_view(const_cast<float*>(a), lda, a_coff), pro_stats, w, sbk, sbn, bias, make_view(c, ldc, c_coff),
                 accumulate, G, Mg, N, K, epilogue, epi_y, epi_stats, part, S(stream));
}

int64_t cdrl_pwconv_pack_elems(int N, int K) { return pw_packed_elems(N, K); }

int cdrl_pwconv_pack(const float* W, int K, int N, int sbk, int sbn, float* packed, int bf16, void* stream) {
    if (!W || !packed) return -1;
    // one-entry table through a temporary device copy (test / tooling entry point; the engine packs all layers in one launch)
    PwPack e = pw_pack_entry(W, packed, K, N, sbk, sbn, bf16 != 0);
    PwPack* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(e)) != hipSuccess) return -2;
    int rc = hipMemcpy(d, &e, sizeof(e), hipMemcpyHostToDevice) == hipSuccess ? pw_pack_many(d, 1, S(stream)) : -2;
    (void)hipStreamSynchronize(S(stream));
    (void)hipFree(d);
    return rc;
}

int cdrl_pwconv_fused_packed(const float* a, int lda, int a_coff, const float* pro_stats, const float* w, int sbk, int sbn,
                             const float* bias, float* c, int ldc, int c_coff, int accumulate, int G, int Mg, int N, int K,
                             int epilogue, const float* epi_y, const float* epi_stats, double* part, const float* w_packed,
                             int packed_bf16, int act_type, void* stream) {
    if (!w_packed) {
        cdrl::set_error("cdrl_pwconv_fused_packed: null packed weights");
        return -1;
    }
    return pw_nn(make_view(const_cast<float*>(a), lda, a_coff), pro_stats, w, sbk, sbn, bias, make_view(c, ldc, c_coff),
                 accumulate, G, Mg, N, K, epilogue, epi_y, epi_stats, part, S(stream), nullptr, w_packed, packed_bf16 != 0, act_type);
}

static int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }

int64_t cdrl_pwconv_bn_bwd_workspace_bytes(int G, int Mg, int N, int K) {
    const int64_t nbr = vcol_geom(Mg, N).nb, nbp = std::max(pw_nn_plan(G, Mg, K, N).nbpg, pw_x3_wide_bwd_rows(Mg));
    return al256((int64_t)G * nbr * 2 * N * 8) + al256((int64_t)G * nbp * N * 8) + al256(gemm_tn_part_elems(G * Mg, N, K, G) * 4);
}

static int pwconv_bn_bwd_impl(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, int relu6, const float* y,
                              const float* stats, const float* x, int x_ld, int x_coff, const float* x_pro_stats, const float* w,
                              int G, int Mg, int N, int K, float* dgamma, float* dbeta, float* coef, float* dx, int dx_ld,
                              int dx_coff, int accumulate, float* dw, float* db, void* workspace, const float* wt_packed,
                              int packed_bf16, int act_type, void* stream) {
    hipStream_t st = S(stream);
    const int act = relu6 ? ACT_RELU6 : ACT_NONE;
    const int nbr = vcol_geom(Mg, N).nb, nbp = pw_nn_plan(G, Mg, K, N).nbpg;
    char* ws = static_cast<char*>(workspace);
    double* part = reinterpret_cast<double*>(ws);
    ws += al256((int64_t)G * nbr * 2 * N * 8);
    double* part2 = reinterpret_cast<double*>(ws);
    ws += al256((int64_t)G * std::max(nbp, pw_x3_wide_bwd_rows(Mg)) * N * 8);
    float* tn = reinterpret_cast<float*>(ws);
    View vd = make_view(const_cast<float*>(dout), dout_ld, dout_coff), vy = make_view(const_cast<float*>(y), N);
    View vx = make_view(const_cast<float*>(x), x_ld, x_coff);
    CDRL_TRY(bn_bwd_reduce(vd, shuffle_ctot, vy, G, Mg, N, stats, act, part, st, nullptr, nullptr, nullptr, 0, act_type));
    CDRL_TRY(bn_bwd_finalize(part, nbr, G, Mg, N, stats, dgamma, dbeta, coef, st));
    PwBnBwd bb{y, stats, coef, shuffle_ctot, act, part2};
    // dx[m,k] = sum_n dy[m,n] W[k,n]: GEMM with "K" = N (reduction over the conv outputs) and "N" = K
    const View vdx = make_view(dx, dx_ld, dx_coff);
    if (!act_type && !packed_bf16 && pw_x3_wide_bwd_supported(vd, vdx, K, N, shuffle_ctot)) {
        // 232-channel shapes, float32: the one-tile-per-workgroup split-precision kernel the engine runs (gemm_pw_x3.hip); the packed
        // operand is built here through a temporary device buffer (test / tooling entry point)
        void* wp = nullptr;
        PwX3Pack* d = nullptr;
        if (hipMalloc(&wp, (size_t)pw_x3_packed_bytes_n(N, K)) != hipSuccess) return -2;
        if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(PwX3Pack)) != hipSuccess) {
            (void)hipFree(wp);
            return -2;
        }
        PwX3Pack e = pw_x3_pack_entry(w, wp, N, K, 1, N);
        int rc = hipMemcpy(d, &e, sizeof(e), hipMemcpyHostToDevice) == hipSuccess ? pw_x3_pack_many(d, 1, st) : -2;
        if (rc == 0) rc = pw_x3_wide_bwd(vd, bb, wp, vdx, accumulate, G, Mg, K, N, nullptr, nullptr, nullptr, st);
        if (rc == 0) rc = reduce_partials(part2, G * pw_x3_wide_bwd_rows(Mg), N, N, db, 0, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(d);
        (void)hipFree(wp);
        if (rc != 0) return rc;
    } else {
        CDRL_TRY(pw_nn(vd, nullptr, w, 1, N, nullptr, vdx, accumulate, G, Mg, K, N, 0, nullptr, nullptr, nullptr,
                       st, &bb, wt_packed, packed_bf16 != 0, act_type));
        CDRL_TRY(reduce_partials(part2, G * nbp, N, N, db, 0, st));
    }
    TnBnBwd tb{y, stats, coef, shuffle_ctot, act};
    return gemm_tn(vx, vd, dw, G * Mg, N, K, tn, 0, st, G, x_pro_stats, &tb, wt_packed && packed_bf16 != 0, act_type);
}

int cdrl_pwconv_bn_bwd(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, int relu6, const float* y,
                       const float* stats, const float* x, int x_ld, int x_coff, const float* x_pro_stats, const float* w,
                       int G, int Mg, int N, int K, float* dgamma, float* dbeta, float* coef, float* dx, int dx_ld,
                       int dx_coff, int accumulate, float* dw, float* db, void* workspace, int act_type, void* stream) {
    return pwconv_bn_bwd_impl(dout, dout_ld, dout_coff, shuffle_ctot, relu6, y, stats, x, x_ld, x_coff, x_pro_stats, w, G, Mg, N, K,
                              dgamma, dbeta, coef, dx, dx_ld, dx_coff, accumulate, dw, db, workspace, nullptr, 0, act_type, stream);
}

int cdrl_pwconv_bn_bwd_packed(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, int relu6, const float* y,
                              const float* stats, const float* x, int x_ld, int x_coff, const float* x_pro_stats, const float* w,
                              int G, int Mg, int N, int K, float* dgamma, float* dbeta, float* coef, float* dx, int dx_ld,
                              int dx_coff, int accumulate, float* dw, float* db, void* workspace, const float* wt_packed,
                              int packed_bf16, int act_type, void* stream) {
    if (!wt_packed) {
        cdrl::set_error("cdrl_pwconv_bn_bwd_packed: null packed weights");
        return -1;
    }
    return pwconv_bn_bwd_impl(dout, dout_ld, dout_coff, shuffle_ctot, relu6, y, stats, x, x_ld, x_coff, x_pro_stats, w, G, Mg, N, K,
                              dgamma, dbeta, coef, dx, dx_ld, dx_coff, accumulate, dw, db, workspace, wt_packed, packed_bf16, act_type, stream);
}

int64_t cdrl_dwconv_bn_workspace_doubles(int G, int B, int H, int W, int C, int stride) {
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    const int64_t nb_out = vcol_geom(B * Ho * Wo, C).nb, nb_in = vcol_geom(B * H * W, C).nb;
    const int64_t nbm = nb_out > nb_in ? nb_out : nb_in;
    return dwf_stats_part_elems(B, G, H, W, C, stride) + dwf_filter_part_elems(B, G, H, W, C, stride) + (int64_t)G * nbm * 3 * C;
}

int cdrl_dwconv_bn_fwd(const float* x, const float* pre_stats, const float* w, const float* bias, float* y, int G, int B,
                       int H, int W, int C, int stride, const float* gamma, const float* beta, float* moving_mean,
                       float* moving_var, int bessel, float* post_stats, double* workspace, int act_type, void* stream) {
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    CDRL_TRY(dwf_fwd(x, pre_stats, w, bias, y, workspace, G, B, H, W, C, stride, S(stream), act_type));
    return bn_finalize(workspace, dwf_geom(B, G, H, W, C, stride).nb, G, B * Ho * Wo, C, gamma, beta, moving_mean, moving_var,
                       bessel, 1, post_stats, S(stream));
}

int cdrl_dwconv_bn_bwd(const float* x, const float* pre_stats, const float* dout, const float* y, const float* post_stats,
                       const float* w, int G, int B, int H, int W, int C, int stride, float* dx, float* dw, float* db,
                       float* dgamma_post, float* dbeta_post, float* coef_post, float* dgamma_pre, float* dbeta_pre,
                       float* coef_pre, double* workspace, int act_type, void* stream) {
    const int Ho = same_out(H, stride), Wo = same_out(W, stride);
    const int Mo = B * Ho * Wo, Mi = B * H * W;
    const DwfGeom g = dwf_geom(B, G, H, W, C, stride);
    double* part_bn = workspace;
    double* part_w = part_bn + dwf_stats_part_elems(B, G, H, W, C, stride);
    double* part_r = part_w + dwf_filter_part_elems(B, G, H, W, C, stride);
    hipStream_t st = S(stream);
    View vd = make_view(const_cast<float*>(dout), C), vy = make_view(const_cast<float*>(y), C);
    CDRL_TRY(bn_bwd_reduce(vd, 0, vy, G, Mo, C, post_stats, ACT_NONE, part_r, st, nullptr, nullptr, nullptr, 0, act_type));
    CDRL_TRY(bn_bwd_finalize(part_r, vcol_geom(Mo, C).nb, G, Mo, C, post_stats, dgamma_post, dbeta_post, coef_post, st));
    CDRL_TRY(dwf_bwd(x, pre_stats, dout, y, post_stats, coef_post, w, make_view(dx, C), part_bn, part_w, G, B, H, W, C, stride, st, act_type));
    CDRL_TRY(reduce_partials(part_w, G * g.nb_bwd, 9 * C, (int64_t)10 * C, dw, 0, st));
    CDRL_TRY(reduce_partials(part_w + 9 * C, G * g.nb_bwd, C, (int64_t)10 * C, db, 0, st));
    if (pre_stats) {
        CDRL_TRY(bn_bwd_finalize(part_bn, g.nb_bwd, G, Mi, C, pre_stats, dgamma_pre, dbeta_pre, coef_pre, st));
        View vx = make_view(const_cast<float*>(x), C);
        CDRL_TRY(bn_bwd_apply(make_view(dx, C), 0, vx, G, Mi, C, pre_stats, coef_pre, ACT_NONE, dx, part_r, st, nullptr, 0, act_type));
    }
    return 0;
}

int cdrl_maxpool_fwd(const float* a, float* p, uint8_t* argmax, int N, int H, int W, int C, void* stream) {
    return maxpool_fwd(a, p, argmax, N, H, W, C, S(stream));
}

int cdrl_maxpool_bwd(const uint8_t* argmax, const float* dp, float* da, int N, int H, int W, int C, void* stream) {
    return maxpool_bwd(argmax, dp, da, N, H, W, C, S(stream));
}

int cdrl_bn_train_fwd(const float* y, int G, int Mg, int C, const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, int bessel, int relu6, float* out, int out_ld, int out_coff, int shuffle_ctot,
                      float* stats, double* workspace, int act_type, void* stream) {
    View yv = make_view(const_cast<float*>(y), C);
    const int nb = vcol_geom(Mg, C).nb;
    CDRL_TRY(colstats(yv, G, Mg, C, workspace, S(stream), act_type));
    CDRL_TRY(bn_finalize(workspace, nb, G, Mg, C, gamma, beta, moving_mean, moving_var, bessel, 1, stats, S(stream)));
    return bn_apply(yv, G, Mg, C, stats, relu6 ? ACT_RELU6 : ACT_NONE, make_view(out, out_ld, out_coff), shuffle_ctot,
                    S(stream), nullptr, nullptr, act_type);
}

int cdrl_bn_train_bwd(const float* dout, int dout_ld, int dout_coff, int shuffle_ctot, const float* y, int G, int Mg,
                      int C, const float* stats, int relu6, float* dgamma, float* dbeta, float* dy, float* coef,
                      double* workspace, int act_type, void* stream) {
    View yv = make_view(const_cast<float*>(y), C);
    View dv = make_view(const_cast<float*>(dout), dout_ld, dout_coff);
    const int nb = vcol_geom(Mg, C).nb;
    const int act = relu6 ? ACT_RELU6 : ACT_NONE;
    CDRL_TRY(bn_bwd_reduce(dv, shuffle_ctot, yv, G, Mg, C, stats, act, workspace, S(stream), nullptr, nullptr, nullptr, 0, act_type));
    CDRL_TRY(bn_bwd_finalize(workspace, nb, G, Mg, C, stats, dgamma, dbeta, coef, S(stream)));
    return bn_bwd_apply(dv, shuffle_ctot, yv, G, Mg, C, stats, coef, act, dy, workspace, S(stream), nullptr, 0, act_type);
}

int cdrl_bn_small_fwd(const float* y, int M, int C, const float* gamma, const float* beta, float* moving_mean,
                      float* moving_var, float* stats, float* out, void* stream) {
    return bn_small_fwd(make_view(const_cast<float*>(y), C), M, C, gamma, beta, moving_mean, moving_var, stats, make_view(out, C),
                        S(stream));
}

int cdrl_bn_small_bwd(const float* dout, const float* y, int M, int C, const float* stats, float* dgamma, float* dbeta,
                      float* coef, float* dx, void* stream) {
    return bn_small_bwd(make_view(const_cast<float*>(dout), C), make_view(const_cast<float*>(y), C), M, C, stats, dgamma, dbeta,
                        coef, dx, S(stream));
}

static int make_head_set(HeadSet& hs, int nheads, const int* n, const float* const* w, const float* const* b, float* const* dw,
                         float* const* db) {
    if (nheads < 1 || nheads > HEADS_MAX) {
        set_error("linear_heads: %d heads not supported", nheads);
        return -1;
    }
    memset(&hs, 0, sizeof(hs));
    hs.nheads = nheads;
    int off = 0;
    for (int h = 0; h < nheads; ++h) {
        hs.n[h] = n[h];
        hs.off[h] = off;
        hs.w[h] = w[h];
        hs.b[h] = b ? b[h] : nullptr;
        hs.gw[h] = dw ? dw[h] : nullptr;
        hs.gb[h] = db ? db[h] : nullptr;
        off += n[h];
    }
    return off;
}

int cdrl_linear_heads_fwd(const float* a, int nheads, const int* n, const float* const* w, const float* const* b, float* lin,
                          int B, int K, void* stream) {
    HeadSet hs;
    const int L = make_head_set(hs, nheads, n, w, b, nullptr, nullptr);
    if (L < 0) return L;
    return heads_fwd(a, K, hs, lin, L, B, K, S(stream));
}

int cdrl_linear_heads_bwd(const float* a, int nheads, const int* n, const float* const* w, const float* dlin, float* da,
                          float* const* dw, float* const* db, int B, int K, void* stream) {
    HeadSet hs;
    const int L = make_head_set(hs, nheads, n, w, nullptr, dw, db);
    if (L < 0) return L;
    return heads_bwd(a, K, hs, dlin, L, da, K, B, K, S(stream));
}

int cdrl_maxpool_bn_fwd(const float* y, const float* stats, int G, int frames_per_group, float* p, uint8_t* argmax,
                        int N, int H, int W, int C, int act_type, void* stream) {
    return maxpool_bn_fwd(y, stats, G, frames_per_group, p, argmax, N, H, W, C, S(stream), act_type);
}

int cdrl_bn_train_bwd_pooled(const uint8_t* argmax, const float* dp, int H, int W, const float* y, int G, int Mg, int C,
                             const float* stats, float* dgamma, float* dbeta, float* dy, float* coef, double* workspace,
                             void* stream) {
    View yv = make_view(const_cast<float*>(y), C);
    View none = make_view(nullptr, 0);
    PoolSrc ps = make_pool_src(argmax, dp, H, W);
    const int nb = vcol_geom(Mg, C).nb;
    CDRL_TRY(bn_bwd_reduce(none, 0, yv, G, Mg, C, stats, ACT_RELU6, workspace, S(stream), &ps));
    CDRL_TRY(bn_bwd_finalize(workspace, nb, G, Mg, C, stats, dgamma, dbeta, coef, S(stream)));
    return bn_bwd_apply(none, 0, yv, G, Mg, C, stats, coef, ACT_RELU6, dy, workspace, S(stream), &ps);
}

int cdrl_beta_ppo_loss(const float* lin, const float* adv, const float* old_logp, const float* speed,
                       const float* similarity, const float* u, const float* du_da, const float* du_db, float clip_ratio,
                       float entropy_coef, int B, int A, float grad_scale, float* dlin, float* metrics, float* aux,
                       float* hp_scratch16, void* stream) {
    DevHP h;
    memset(&h, 0, sizeof(h));
    h.clip_ratio = clip_ratio;
    h.entropy_coef = entropy_coef;
    CDRL_HIP(hipMemcpyAsync(hp_scratch16, &h, sizeof(h), hipMemcpyHostToDevice, S(stream)));
    PolicyLossArgs a;
    a.lin = lin; a.adv = adv; a.old_logp = old_logp; a.speed = speed; a.similarity = similarity;
    a.u = u; a.du_da = du_da; a.du_db = du_db; a.hp = hp_scratch16; a.dlin = dlin; a.metrics = metrics; a.aux = aux;
    a.B = B; a.A = A; a.inv_world = grad_scale;
    return policy_loss(a, S(stream));
}

int cdrl_value_loss(const float* lin, const float* returns, const float* speed, const float* similarity, int B,
                    float exp_scale, float grad_scale, float* dlin, float* metrics, float* values, void* stream) {
    ValueLossArgs a;
    a.lin = lin; a.returns = returns; a.speed = speed; a.similarity = similarity; a.dlin = dlin; a.metrics = metrics;
    a.values = values; a.B = B; a.exp_scale = exp_scale; a.inv_world = grad_scale;
    return value_loss(a, S(stream));
}

}  // extern "C"
