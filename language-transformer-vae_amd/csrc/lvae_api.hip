// extern "C" entry points of liblvae.so (see include/lvae.h for the contract of each).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pair_plan.h"

// ---- launchers implemented in the kernel files --------------------------------------------------
int lvae_launch_colsum(const float* mu, int64_t n, int D, PairHeader* hdr, hipStream_t st, double* zero5 = nullptr);
bool lvae_colsum_single_ok(int64_t n, int D);
int lvae_launch_colsum_single(const float* mu, int64_t n, int D, PairHeader* hdr, size_t clear_bytes, hipStream_t st,
                              double* zero5 = nullptr);
int lvae_launch_prep(int src, bool frag, const float* x, const float* spread, int spread_kind,
                     int64_t n, int64_t npad, int D, int DP, PairHeader* hdr, int64_t ngauss,
                     float* out, float* bias, float* dparams, int DC, hipStream_t st,
                     float cond_floor = -1.f);
int lvae_launch_prep_dparams(const float* mu, const float* spread, int spread_kind, int64_t n, int D,
                             int DC, float* dparams, const PairHeader* gate_hdr, float cond_limit,
                             hipStream_t st);
int lvae_launch_prep_pair(const float* mu, const float* spread, int spread_kind, int64_t n, int64_t npad_r,
                          const float* z, int64_t m, int64_t npad_c, int D, int DP, PairHeader* hdr, float* rfrag,
                          float* rbias, float* dparams, int DC, float* cop, float cond_floor, hipStream_t st,
                          bool want_norm = false);
int lvae_launch_finalize_small(const float2* partial_a, int nsplit_a, const float2* partial_b, int nsplit_b,
                               int64_t cpad, int64_t C, float log_norm, float* out, float* out_max, float* out_sum,
                               const PairHeader* hdr, float cond_limit, const float* z, const float* mu,
                               const float* logvar, int D, double* sums, hipStream_t st);
int lvae_launch_finalize(const float2* partial, int nsplit, int64_t cpad, int64_t C,
                         const float* colbias, float log_norm, float* out, float* out_max,
                         float* out_sum, const PairHeader* hdr, float cond_limit, int use_if_above,
                         int use_ref, int* flags_out, const int* flags_in, int group_cols, hipStream_t st);
int lvae_launch_lse_shift_bias(float* rbias, int64_t n, int64_t n_valid, PairHeader* hdr, hipStream_t st,
                               const float* dp_mu = nullptr, const float* dp_spread = nullptr, int spread_kind = 0,
                               int64_t dp_n = 0, int D = 0, int DC = 0, float* dparams = nullptr, float cond_limit = 0.f);
int lvae_launch_finalize_pm(const float2* partial, int nsplit, int64_t cpad, int64_t C, int D, int DC,
                            float log_norm, float* tmp, float* out, hipStream_t st);
int lvae_launch_pair_lse_mfma(const PairPlan& p, const float* rfrag, const float* rbias,
                              const float* cop, float2* partial, const PairHeader* hdr,
                              float cond_limit, int use_if_above, bool excl, int64_t excl_offset,
                              hipStream_t st);
int lvae_launch_pair_direct(const PairPlan& p, bool joint, bool pm, const float* z,
                            const float* dparams, float2* partial, float2* partial_pm, int* redo_flags,
                            const PairHeader* hdr, float cond_limit, int use_if_above, int excl,
                            int64_t excl_offset, hipStream_t st, bool pm_fast_done = false,
                            float pm_cond_limit = -1.0f);
bool lvae_pm_bx_supported(int DC);
int lvae_launch_pm_bx(const PairPlan& p, const float* z, const float* mu, const float* spread, int spread_kind,
                      const PairHeader* hdr, void* aop, void* bop, float2* partial_pm, int* redo_flags,
                      float cond_limit, hipStream_t st);

int lvae_launch_reduce(int mode, const float* x, int64_t n, double scale, double* out, double* part,
                       hipStream_t st);
int lvae_launch_sample_iw(bool from_eps, const float* mu, const float* logvar, const float* src,
                          const float* log_p_x_z, int64_t B, int64_t K, int D, float* z_out,
                          float* log_q_z_x, float* log_p_z, float* iw_ll, hipStream_t st);
int lvae_launch_sample_ll_elem(const float* mu, const float* logvar, const float* z, int64_t B,
                               int64_t K, int D, float* out, hipStream_t st);
int lvae_launch_iw_lse(const float* a, const float* b, const float* c, int64_t B, int64_t K, float* out,
                       hipStream_t st);

int lvae_launch_latent_sums(const float* z, const float* mu, const float* logvar, const float* log_qz,
                            const float* log_qz_pm, int64_t M, int D, double* sums, hipStream_t st,
                            bool sums_are_zero = false, const LateMerge* late = nullptr);
bool lvae_latent_sums_vec(int D);

int lvae_launch_pair_mfma(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                          float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                          bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym, const float* colnorm,
                          hipStream_t st);
int lvae_launch_pair_bf16x3(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                          float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                          bool excl, int64_t excl_offset, int epi, const int* redo_flags, int sym, const float* colnorm,
                          hipStream_t st);
int lvae_launch_finalize_ksum(const double* partial, int nsplit, int64_t cpad, int64_t C, double* out,
                              hipStream_t st);
int lvae_launch_kl(const float* mu, const float* logvar, int64_t n, int D, float lambda, float* kl_elem,
                   float* hinge_elem, float* kl_row, float* hinge_row, hipStream_t st);
int lvae_launch_colmean(const float* x, int64_t n, int64_t c, float* out, double* acc, hipStream_t st);
void lvae_kde_plan(int64_t n, int64_t m, int* nsplit, int64_t* rows);
int lvae_launch_kde(const float* X, const float* Y, int64_t n, int64_t n_norm, int64_t m, int d, float bw, int stabilise,
                    float* out, float* ws, hipStream_t st);
int lvae_launch_pair_matrix(const float* a, const float* spread, int spread_kind, const float* b,
                            int64_t na, int64_t nb, int D, int mode, const PairHeader* hdr, float* out,
                            hipStream_t st);
int lvae_launch_diag_ll(const float* z, const float* mu, const float* spread, int spread_kind, int64_t n,
                        int D, float* out, hipStream_t st);
int lvae_launch_bounds_finalize(const float* diag, const float* lse_all, const float* lse_off, int64_t n,
                                double* out, hipStream_t st);
int lvae_launch_set_ksum(PairHeader* hdr, const KsumParams& kp, hipStream_t st);
int lvae_launch_ksum_columns(const double* partial, int nsplit, int64_t cpad, int64_t C, int64_t R, int excl,
                             int64_t excl_offset, int zero_alphas, double* colsum, double* total,
                             hipStream_t st);
int lvae_launch_pair_mfma_perm(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                               float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                               int64_t excl_offset, int epi, const PermArgs& pa, int sym, const float* colnorm,
                               hipStream_t st);
int lvae_launch_prep_labels(const unsigned char* labels, int P, int64_t n, int64_t nblk, void* out,
                            hipStream_t st);
int lvae_launch_perm_lr(const double* partial, const double* diag_partial, int nsplit, int64_t cpad, int64_t C,
                        const unsigned char* labels, int64_t n, int64_t col_offset, int P, int sym,
                        const double* qperm, double* out, hipStream_t st);
int lvae_launch_finalize_pm_lse(const float2* partial, int nsplit, int64_t cpad, int64_t C, int DC,
                                float* lse2, hipStream_t st);
int lvae_launch_grad_records(const float* z, const float* log_qz, float log_norm, const float* g,
                             const float* h, const float* lse2, int64_t M, int D, int DC, float* rec,
                             hipStream_t st);
int lvae_launch_grad_sample(const float* z, const float* dparams, const float* log_qz, float log_norm,
                            const float* g, const float* h, const float* lse2, int64_t M, int64_t N, int D,
                            int DC, int nsplit, int64_t rows_per_split, float* partial_dz, hipStream_t st,
                            const PairHeader* gate_hdr = nullptr, float cond_limit = 0.f);
int lvae_launch_grad_posterior(const float* dparams, const float* rec, bool pm, int64_t M, int64_t N, int DC,
                               int nsplit, int64_t rows_per_split, float* partial_dp, hipStream_t st,
                               const PairHeader* gate_hdr = nullptr, float cond_limit = 0.f);
int lvae_launch_grad_reduce(const float* partial, int nsplit, int64_t rows, int stride, int off, int D,
                            float* out, hipStream_t st, const PairHeader* gate_hdr = nullptr, float cond_limit = 0.f);
int lvae_launch_sample_ll_bwd(const float* z, const float* mu, const float* logvar, const float* g, int g_elem,
                              int64_t B, int64_t K, int D, float* d_z, float* d_mu, float* d_logvar,
                              hipStream_t st);
int lvae_launch_reparam_bwd(const float* logvar, const float* eps, const float* g, int64_t B, int64_t K, int D,
                            float* d_mu, float* d_logvar, hipStream_t st);
int lvae_launch_kl_bwd(const float* mu, const float* logvar, const float* g_kl, const float* g_hinge,
                       int row_mode, int64_t n, int D, float lambda, float* d_mu, float* d_logvar,
                       hipStream_t st);
int lvae_launch_lse_combine(const float* max_a, const float* sum_a, const float* max_b, const float* sum_b,
                            int64_t n, float log_norm, float* out, float* out_max, float* out_sum, hipStream_t st);
int lvae_launch_lse_rebase(const float* max_a, const float* sum_a, const float* new_max, int64_t n, float* out_sum,
                           hipStream_t st);
int lvae_launch_grad_wide(const float* z, const float* dparams, const float* P, const float* log_qz, float log_norm,
                          const float* g, const float* h, const float* lse2, int64_t M, int64_t N, int D, int DC,
                          float* d_z, float* d_mu, float* d_logvar, hipStream_t st,
                          const PairHeader* gate_hdr = nullptr, float cond_limit = 0.f);
int lvae_launch_bwd_mfma(const BwdMfmaLayout& L, void* ws, const float* z, const float* mu, const float* logvar,
                         int64_t M, int64_t N, int D, float log_norm, const float* log_qz, const float* g,
                         PairHeader* hdr, float cond_limit, float* d_z, float* d_mu, float* d_logvar, hipStream_t st,
                         float poison_limit = -1.f);
BwdMfmaLayout lvae_bwd_mfma_layout(int64_t M, int64_t N, int D);
int lvae_launch_ksum_wide(const float* a, const float* b, int64_t na, int64_t nb, int D, const PairHeader* hdr,
                          bool excl, int64_t excl_offset, int nsplit, int64_t cpad, double* partial,
                          hipStream_t st);
size_t lvae_cat_bounds_workspace(int64_t K);
int lvae_launch_cat_bounds(const float* probs, const int64_t* samples, int64_t K, int64_t V, double* out, void* ws,
                           hipStream_t st);
size_t lvae_ks_workspace(int64_t n1, int64_t n2);
int lvae_launch_ks_2samp(const float* a, int64_t n1, const float* b, int64_t n2, double* out, void* ws,
                         hipStream_t st);
bool lvae_bf16x3_supported(int DP);
int lvae_launch_pair_bf16x3_perm(const PairPlan& p, const float* rfrag, const float* rbias, const float* cop,
                                 float2* partial, const PairHeader* hdr, float cond_limit, int use_if_above,
                                 int64_t excl_offset, int epi, const PermArgs& pa, int sym, const float* colnorm,
                                 hipStream_t st);
int lvae_launch_prep_bf16x3(int src, bool frag, const float* x, const float* spread, int spread_kind,
                            int64_t n, int64_t npad, int D, int DP, const PairHeader* hdr, int64_t ngauss,
                            void* out, hipStream_t st);
int lvae_launch_gauss_ll_masked(const float* sample, const float* mu, const float* var, const float* mask, int ndim,
                                const int64_t* shape, const int64_t* st_s, const int64_t* st_mu,
                                const int64_t* st_var, const int64_t* st_mask, int dim, int64_t n_out, float* out,
                                hipStream_t st);
bool lvae_encode_tail_tiny_ok(int64_t B, int64_t K, int D);
int lvae_launch_encode_tail_tiny(const float* mu, const float* logvar, const float* eps, const float* prior, int64_t B,
                                 int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z,
                                 float* log_qz, float* log_qz_pm, float* scalars, hipStream_t st);
int lvae_launch_encode_tail_tiny_bwd(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                                     float log_norm, const float* g_z, const float* g_lqx, const float* g_lpz,
                                     const float* g_lq, const float* g_pm, float* d_mu, float* d_lv, hipStream_t st);
bool lvae_encode_tail_mid_ok(int64_t B, int64_t K, int D);
int64_t lvae_encode_tail_mid_workspace(int64_t B, int D);
int lvae_launch_encode_tail_mid(const float* mu, const float* logvar, const float* eps, const float* prior, int64_t B,
                                int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z,
                                float* log_qz, float* log_qz_pm, float* scalars, void* ws, hipStream_t st);
int lvae_launch_encode_tail_mid_bwd(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                                    float log_norm, const float* g_z, const float* g_lqx, const float* g_lpz,
                                    const float* g_lq, const float* g_pm, float* d_mu, float* d_lv, void* ws,
                                    hipStream_t st);
bool lvae_pair_multi_supported(int DP, int epi);
int lvae_launch_pair_multi(int DP, const PairJobs& jobs, int njobs, const PairHeader* hdr, bool excl, int epi,
                           hipStream_t st);
int lvae_launch_prep_ksum_multi(const float* const* x, const int64_t* n, const int64_t* npad, float* const* out,
                                float* const* bias, const int* point, int njobs, int D, int DP, PairHeader* hdr,
                                hipStream_t st);
int lvae_launch_finalize_ksum_multi(const double* const* partial, const int* nsplit, const int64_t* cpad, const int64_t* C,
                                    int nsets, double* out, hipStream_t st);
int lvae_launch_ksum_init(PairHeader* hdr, const KsumParams& kp, double* zero, int nzero, hipStream_t st);
int lvae_launch_mmd_stat(const double* sums, int64_t n1, int64_t n2, int unbiased, int zero_alphas,
                         double* out, hipStream_t st);

// ---- error text, per calling thread --------------------------------------------------------------
static thread_local char g_err[512] = "";

int lvae_set_error(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

extern "C" const char* lvae_last_error(void) { return g_err; }

// ---- development switches: the environment is read once (std::call_once), never on a product call --------------
#include <mutex>
static LvaeSwitches g_switches;
static std::once_flag g_switches_once;
static void load_switches() {
    auto num = [](const char* name, long long unset) -> long long {
        const char* v = getenv(name);
        return v ? atoll(v) : unset;
    };
    LvaeSwitches s{};
    s.nsplit = num("LVAE_NSPLIT", 0);
    s.dsplit = num("LVAE_DSPLIT", 0);
    s.small_pairs = getenv("LVAE_SMALL_PAIRS") ? atof(getenv("LVAE_SMALL_PAIRS")) : -1.0;
    s.no_small_path = getenv("LVAE_NO_SMALL_PATH") != nullptr;
    s.prep_rb = getenv("LVAE_PREP_RB") ? (int)num("LVAE_PREP_RB", 0) : 0;
    if (getenv("LVAE_PREP_RB") && s.prep_rb <= 0) s.prep_rb = -1;      // set but not a count: separate launches, built-in blocks
    s.no_fused_tail = getenv("LVAE_NO_FUSED_TAIL") != nullptr;
    s.no_fused_mmd = getenv("LVAE_NO_FUSED_MMD") != nullptr;
    const char* bp = getenv("LVAE_BWD_PATH");
    s.bwd_path = !bp ? 0 : (!strcmp(bp, "valu") ? 1 : (!strcmp(bp, "mfma") ? 2 : 0));
    s.iw_pipe = getenv("LVAE_IW_PIPE") ? (num("LVAE_IW_PIPE", 0) != 0 ? 1 : 0) : -1;
    s.iw_u = (int)num("LVAE_IW_U", 0);
    s.iw_tpb = (int)num("LVAE_IW_TPB", 0);
    s.iw_no_stage = getenv("LVAE_IW_NO_STAGE") != nullptr;
    s.iw_plain_stores = getenv("LVAE_IW_PLAIN_STORES") != nullptr;
    s.mmd_bx_min_pairs = num("LVAE_MMD_BX_MIN_PAIRS", -1);
    g_switches = s;
}
const LvaeSwitches& lvae_switches() {
    std::call_once(g_switches_once, load_switches);
    return g_switches;
}
// Re-reads the environment (sweep tools and tests that change a switch inside one process); not for use while other
// threads are inside the library.
extern "C" int lvae_dev_reload_env(void) {
    std::call_once(g_switches_once, load_switches);
    load_switches();
    return LVAE_OK;
}
extern "C" int lvae_version(void) { return 100; }

// ---- optional per-launch timing of the dominant (pair) kernels ------------------------------------
// HIP events recorded on the caller's stream right around the pair kernels, so a benchmark can price
// the dominant kernel alone while timing whole calls.  Off by default; not thread-safe by design
// (a measurement facility, one measuring thread).
#include <atomic>
#include <vector>
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_prof_ev;   // start/stop pairs
static int g_prof_used = 0;

extern "C" int lvae_profile_begin(int max_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    LVAE_CHECK_ARG(max_launches >= 1 && max_launches <= 65536, "max_launches out of range");
    while ((int)g_prof_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        LVAE_HIP(hipEventCreate(&e));
        g_prof_ev.push_back(e);
    }
    g_prof_used = 0;
    g_prof_on = true;
    return LVAE_OK;
}

extern "C" int lvae_profile_end(double* total_ms, int* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    LVAE_CHECK_ARG(total_ms && launches, "NULL output");
    g_prof_on = false;
    double tot = 0.0;
    for (int i = 0; i < g_prof_used; ++i) {
        LVAE_HIP(hipEventSynchronize(g_prof_ev[2 * i + 1]));
        float ms = 0.f;
        LVAE_HIP(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = g_prof_used;
    g_prof_used = 0;
    return LVAE_OK;
}

static hipEvent_t prof_mark(hipStream_t st, hipEvent_t open) {
    // open == nullptr: start a bracket, returns the stop event to record later (or nullptr when off)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (open) {
        (void)hipEventRecord(open, st);
        return nullptr;
    }
    if (!g_prof_on || 2 * (g_prof_used + 1) > (int)g_prof_ev.size()) return nullptr;
    hipEvent_t a = g_prof_ev[2 * g_prof_used], b = g_prof_ev[2 * g_prof_used + 1];
    ++g_prof_used;
    (void)hipEventRecord(a, st);
    return b;
}

// Conditioning estimate above which the auto path leaves the MFMA expansion for the direct form.
// prep computes  cond = max_j sum_d 0.5 * (2|mu_jd|/sigma_jd + 3)^2  -- the magnitude of the terms
// that cancel in the expansion for a sample within ~3 sigma of posterior j (the only pairs that carry
// weight in the logsumexp).  One fp32 ulp of that is the absolute error scale in nats:
// 2^-23 * cond <= ~2.4e-3  <=>  cond <= 2e4.
#define LVAE_COND_LIMIT 2.0e4f
// Smallest N*M from which the AUTO path runs the prod-marginals polynomial on the bf16 matrix pipe (two more
// operand-prep launches; measured faster from 2048 x 2048 up, launch-bound calls keep the single VALU sweep).
#define LVAE_PM_BX_MIN_PAIRS 4194304.0

static int check_ws(const void* ws, int64_t ws_bytes, size_t need) {
    if (!ws) return lvae_set_error(LVAE_ERR_WORKSPACE, "workspace pointer is NULL (need %zu bytes)", need);
    if (((uintptr_t)ws & 255) != 0)
        return lvae_set_error(LVAE_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (ws_bytes < 0 || (size_t)ws_bytes < need)
        return lvae_set_error(LVAE_ERR_WORKSPACE, "workspace too small: %lld < %zu bytes",
                              (long long)ws_bytes, need);
    return LVAE_OK;
}

// Clears the header region (clear_bytes from hdr on) and accumulates the column sums of mu into it: one launch for
// small inputs, a fill + the multi-workgroup kernel otherwise.
static int clear_and_colsum(const float* mu, int64_t N, int D, PairHeader* hdr, size_t clear_bytes, hipStream_t st,
                            double* zero5 = nullptr) {
    if (lvae_colsum_single_ok(N, D) && clear_bytes <= 65536)
        return lvae_launch_colsum_single(mu, N, D, hdr, (clear_bytes + 3) & ~(size_t)3, st, zero5);
    LVAE_HIP(hipMemsetAsync(hdr, 0, clear_bytes, st));
    return lvae_launch_colsum(mu, N, D, hdr, st, zero5);
}

static int check_pair_shapes(const float* z, const float* mu, const float* spread, int spread_kind,
                             int64_t M, int64_t N, int D) {
    LVAE_CHECK_ARG(z && mu && spread, "z, mu and spread must be non-NULL");
    LVAE_CHECK_ARG(M >= 1 && N >= 1, "need M >= 1 and N >= 1 (got M=%lld N=%lld)", (long long)M, (long long)N);
    LVAE_CHECK_ARG(D >= 1 && D <= 4096, "need 1 <= D <= 4096 (got %d)", D);
    LVAE_CHECK_ARG(M < (1LL << 31) && N < (1LL << 31), "M and N must be below 2^31");
    LVAE_CHECK_ARG(spread_kind == LVAE_SPREAD_LOGVAR || spread_kind == LVAE_SPREAD_VAR,
                   "spread_kind must be LVAE_SPREAD_LOGVAR or LVAE_SPREAD_VAR");
    return LVAE_OK;
}

extern "C" int lvae_log_qz_workspace_bytes(int64_t M, int64_t N, int D, int with_prod_marg,
                                           int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(M >= 1 && N >= 1 && D >= 1 && D <= 4096, "bad shape M=%lld N=%lld D=%d",
                   (long long)M, (long long)N, D);
    // (the layout depends on the contraction the call will choose: cover both)
    const size_t a = lvae_log_qz_plan(N, M, D, with_prod_marg != 0, false).total;
    const size_t b = lvae_log_qz_plan(N, M, D, with_prod_marg != 0, true).total;
    *out_bytes = (int64_t)(a > b ? a : b);
    return LVAE_OK;
}

// Optional tail of the driver below: the float64 sums of lvae_latent_sums_f32 for samples z_i drawn from
// posterior (mu[i], logvar[i]) -- fused into the merge kernel on the small path.
struct StatsTail {
    const float* mu;
    const float* logvar;
    double* sums;
};

// Largest N * M served by the small path (five launches, running-maximum epilogue in one pass: its lower MFMA rate
// costs less than the extra launches of the fixed-reference form -- nine since the launch-chain fusions -- up to
// ~8,000^2; measured, tools/sweep_small_threshold.py, D = 32, small vs general: 48 vs 60 us at 4,096^2, 78 vs 81 at
// 6,144^2, 110 vs 110 at 8,192^2, 213 vs 200 at 12,288^2, 328 vs 303 at 16,384^2), and the most samples its merge
// handles.
#define LVAE_SMALL_PAIRS 60000000.0
#define LVAE_SMALL_MAX_M 16384

// Shared driver: samples on the kept axis, posteriors on the reduced axis.
static int run_log_qz(const float* z, const float* mu, const float* spread, int spread_kind,
                      int64_t M, int64_t N, int D, float log_norm, int exclude_diag,
                      int64_t diag_offset, int path, float* log_qz, float* log_qz_pm, float* row_max,
                      float* row_sum, void* ws, int64_t ws_bytes, hipStream_t st,
                      const StatsTail* tail = nullptr) {
    int rc = check_pair_shapes(z, mu, spread, spread_kind, M, N, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(path == LVAE_PATH_AUTO || path == LVAE_PATH_MFMA || path == LVAE_PATH_DIRECT ||
                       path == LVAE_PATH_MFMA_BF16X3,
                   "unknown path %d", path);
    const bool bx = path == LVAE_PATH_MFMA_BF16X3;
    if (bx) path = LVAE_PATH_MFMA;          // same control flow as the forced fp32 MFMA path, other pair kernel
    const bool with_pm = log_qz_pm != nullptr;
    const PairPlan p = lvae_log_qz_plan(N, M, D, with_pm, bx, exclude_diag != 0);
    rc = check_ws(ws, ws_bytes, p.total);
    if (rc) return rc;
    if (path == LVAE_PATH_MFMA && p.DP == 0)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "MFMA path needs D <= 128 (got D=%d)", D);
    if (bx && !lvae_bf16x3_supported(p.DP))
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the bf16x3 path needs D <= 64 (got D=%d)", D);
    const auto launch_pair = bx ? lvae_launch_pair_bf16x3 : lvae_launch_pair_mfma;

    char* base = (char*)ws;
    PairHeader* hdr = (PairHeader*)(base + p.off_hdr);
    float* rfrag = (float*)(base + p.off_rfrag);
    float* rbias = (float*)(base + p.off_rbias);
    float* cop = (float*)(base + p.off_cop);
    float2* partial = (float2*)(base + p.off_partial);
    float* dparams = (float*)(base + p.off_dparams);
    float2* dpartial = (float2*)(base + p.off_dpartial);
    float2* dpartial_pm = (float2*)(base + p.off_dpartial_pm);
    const bool want_joint = log_qz || row_max;

    // The prod-marginals sweep is a kernel of its own (packed fp32 VALU + one exponential per term);
    // the joint density goes through the MFMA kernel as usual, which costs a fraction of that sweep.
    int joint_path = path;
    if (p.DP == 0) joint_path = LVAE_PATH_DIRECT;
    const bool need_mfma = want_joint && joint_path != LVAE_PATH_DIRECT;
    const bool need_direct = with_pm || (want_joint && joint_path != LVAE_PATH_MFMA);

    // ---- small problems: memset, column sums, ONE operand-prep launch, the pair kernel(s) with the running-maximum
    //      epilogue, ONE merge (+ statistics) launch.  Same kernels, same device-side conditioning gate.
    const double pairs = (double)N * (double)M;
    const LvaeSwitches& sw = lvae_switches();              // (threshold sweeps: tools/sweep_small_threshold.py)
    const double small_pairs = sw.small_pairs >= 0.0 ? sw.small_pairs : LVAE_SMALL_PAIRS;
    const bool small = p.DP != 0 && !bx && want_joint && pairs <= small_pairs && M <= LVAE_SMALL_MAX_M &&
                       !(with_pm && pairs >= LVAE_PM_BX_MIN_PAIRS) && !sw.no_small_path;
    if (small) {
        const bool gated_s = joint_path == LVAE_PATH_AUTO;
        const float limit_s = gated_s ? LVAE_COND_LIMIT : -1.0f;
        const bool djoint = joint_path != LVAE_PATH_MFMA;
        rc = clear_and_colsum(mu, N, D, hdr, sizeof(PairHeader), st, tail ? tail->sums : nullptr);
        if (rc) return rc;
        rc = lvae_launch_prep_pair(mu, spread, spread_kind, N, p.nblk * 32, z, M, p.cpad, D, p.DP, hdr, rfrag, rbias,
                                   need_direct ? dparams : nullptr, p.DC, cop, gated_s ? LVAE_COND_LIMIT : INFINITY, st);
        if (rc) return rc;
        hipEvent_t prof_stop_s = prof_mark(st, nullptr);
        if (need_mfma) {
            rc = lvae_launch_pair_mfma(p, rfrag, rbias, cop, partial, hdr, limit_s, 0, exclude_diag != 0, diag_offset, 0,
                                       nullptr, 0, nullptr, st);
            if (rc) return rc;
        }
        if (need_direct) {
            rc = lvae_launch_pair_direct(p, djoint, with_pm, z, dparams, dpartial, dpartial_pm,
                                         (int*)(base + p.off_dflags), hdr, limit_s, 1, exclude_diag, diag_offset, st);
            if (rc) return rc;
        }
        if (prof_stop_s) prof_mark(st, prof_stop_s);
        rc = lvae_launch_finalize_small(need_mfma ? partial : nullptr, p.nsplit, djoint ? dpartial : nullptr, p.dsplit,
                                        p.cpad, M, log_norm, log_qz, row_max, row_sum, hdr, limit_s, z,
                                        tail ? tail->mu : nullptr, tail ? tail->logvar : nullptr, D,
                                        tail ? tail->sums : nullptr, st);
        if (rc) return rc;
        if (with_pm) {
            rc = lvae_launch_finalize_pm(dpartial_pm, p.dsplit, p.cpad, M, D, p.DC, log_norm,
                                         (float*)(base + p.off_pmtmp), log_qz_pm, st);
            if (rc) return rc;
        }
        return LVAE_OK;
    }

    // Direct-form parameters (25 MB at C2) are only written when something will read them: always with
    // prod-marginals or a forced direct path; in AUTO mode by a separate kernel that runs iff the device-side
    // conditioning test selects the direct path.
    const bool auto_gate = want_joint && joint_path == LVAE_PATH_AUTO;
    const bool dparams_now = with_pm || (want_joint && joint_path == LVAE_PATH_DIRECT);
    // (the redo flags of the fixed-reference pass sit right behind the header: one fill for both)
    // (AUTO path with the fixed-reference pass: the gated direct-form parameters ride on the bias-shift launch)
    const bool dparams_with_shift = need_direct && !dparams_now && auto_gate && need_mfma && N <= (1LL << 24) && p.DP;
    const size_t clear_bytes = need_mfma ? p.off_redo - p.off_hdr + (size_t)(p.cpad / 128 + 1) * sizeof(int)
                                         : sizeof(PairHeader);
    if (p.DP) {
        rc = clear_and_colsum(mu, N, D, hdr, clear_bytes, st, tail ? tail->sums : nullptr);   // (also zeroes the statistics)
        if (rc) return rc;
        // (gated calls only compare the conditioning estimate with the limit: workgroups below it publish nothing)
        // fp32 contraction: both operands in ONE launch (one 32-row block per workgroup, posterior rows first)
        const bool prep_both = need_mfma && !bx && sw.prep_rb == 0;
        if (prep_both)
            rc = lvae_launch_prep_pair(mu, spread, spread_kind, N, p.nblk * 32, z, M, p.cpad, D, p.DP, hdr, rfrag, rbias,
                                       dparams_now ? dparams : nullptr, p.DC, cop, auto_gate ? LVAE_COND_LIMIT : -1.f, st,
                                       true);
        else
            rc = lvae_launch_prep(0, true, mu, spread, spread_kind, N, p.nblk * 32, D, p.DP, hdr, N,
                                  bx ? nullptr : rfrag, rbias, dparams_now ? dparams : nullptr, p.DC, st,
                                  auto_gate ? LVAE_COND_LIMIT : -1.f);
        if (rc) return rc;
        if (bx && need_mfma) {   // limb operands into the same regions (the fp32 prep above still owns bias + header)
            rc = lvae_launch_prep_bf16x3(0, true, mu, spread, spread_kind, N, p.nblk * 32, D, p.DP, hdr, N,
                                         rfrag, st);
            if (rc) return rc;
            rc = lvae_launch_prep_bf16x3(1, false, z, nullptr, 0, M, p.cpad, D, p.DP, hdr, N, cop, st);
            if (rc) return rc;
        }
        if (need_direct && !dparams_now && !dparams_with_shift) {
            rc = lvae_launch_prep_dparams(mu, spread, spread_kind, N, D, p.DC, dparams,
                                          auto_gate ? hdr : nullptr, LVAE_COND_LIMIT, st);
            if (rc) return rc;
        }
        if (need_mfma && !bx && !prep_both) {
            rc = lvae_launch_prep(1, false, z, nullptr, 0, M, p.cpad, D, p.DP, hdr, N, cop, nullptr,
                                  nullptr, p.DC, st);
            if (rc) return rc;
        }
    } else {
        LVAE_HIP(hipMemsetAsync(hdr, 0, clear_bytes, st));
        rc = lvae_launch_prep_dparams(mu, spread, spread_kind, N, D, p.DC, dparams, nullptr, 0.f, st);
        if (rc) return rc;
    }

    const bool gated = want_joint && joint_path == LVAE_PATH_AUTO;
    const float limit = gated ? LVAE_COND_LIMIT : -1.0f;
    // MFMA path: a first pass against a fixed reference (no running maximum: every term is bounded by
    // the largest row bias), then -- for the rare columns whose total underflowed that reference -- a
    // redo of their workgroups with the running-maximum kernel.  All decided on the device.
    const bool fixed_ref = need_mfma && N <= (1LL << 24);
    // The statistics call folds the redo merge, the direct-form merge and the float64 sums into one launch
    // (LateMerge): three launches fewer, two of them gates that exit at once on ordinary inputs.
    // Without the statistics (lvae_log_qz_f32) the same kernel runs as the late merge alone: one launch for both merges.
    const bool fuse_tail = log_qz && !row_max && fixed_ref && (tail ? !with_pm : true) && lvae_latent_sums_vec(D) &&
                           !sw.no_fused_tail;
    int* redo = (int*)(base + p.off_redo);
    const int group_cols = 128 * p.NCB;
    if (need_mfma) {
        if (fixed_ref) {
            if (dparams_with_shift)
                rc = lvae_launch_lse_shift_bias(rbias, p.nblk * 32, N, hdr, st, mu, spread, spread_kind, N, D, p.DC,
                                                dparams, LVAE_COND_LIMIT);
            else
                rc = lvae_launch_lse_shift_bias(rbias, p.nblk * 32, N, hdr, st);
            if (rc) return rc;
        }
    }
    hipEvent_t prof_stop = prof_mark(st, nullptr);
    if (need_mfma) {
        rc = launch_pair(p, rfrag, rbias, cop, partial, hdr, limit, 0, exclude_diag != 0, diag_offset,
                         fixed_ref ? 4 : 0, nullptr, 0, nullptr, st);
        if (rc) return rc;
    }
    if (need_direct) {
        const bool djoint = want_joint && joint_path != LVAE_PATH_MFMA;
        // The prod-marginals polynomial on the bf16 matrix pipe (D <= 64; three bf16 limbs per value, fp32-equivalent),
        // always behind the device-side conditioning test (its expansion cancels like the joint one; the packed-fp32
        // direct form runs with the opposite gate): on the opt-in bf16x3 path at every size, in AUTO mode from
        // LVAE_PM_BX_MIN_PAIRS on.
        const bool pm_bx_ok = with_pm && p.DP && lvae_pm_bx_supported(p.DC);
        const bool pm_bx = pm_bx_ok && (bx || (gated && need_mfma && (double)N * (double)M >= LVAE_PM_BX_MIN_PAIRS));
        if (pm_bx) {
            rc = lvae_launch_pm_bx(p, z, mu, spread, spread_kind, hdr, base + p.off_pmbx_a, base + p.off_pmbx_b,
                                   dpartial_pm, (int*)(base + p.off_dflags), LVAE_COND_LIMIT, st);
            if (rc) return rc;
        }
        rc = lvae_launch_pair_direct(p, djoint, with_pm, z, dparams, dpartial, dpartial_pm,
                                     (int*)(base + p.off_dflags), hdr, limit, 1,
                                     exclude_diag, diag_offset, st, false, pm_bx ? LVAE_COND_LIMIT : -1.0f);
        if (rc) return rc;
    }
    if (prof_stop) prof_mark(st, prof_stop);
    if (want_joint) {
        // gated: both pair kernels were launched and exactly one ran; the finalize of the other one
        // exits on the same device-side test, so the outputs are written once.
        if (joint_path != LVAE_PATH_DIRECT) {
            rc = lvae_launch_finalize(partial, p.nsplit, p.cpad, M, nullptr, log_norm, log_qz, row_max,
                                      row_sum, hdr, limit, 0, fixed_ref ? 1 : 0, fixed_ref ? redo : nullptr,
                                      nullptr, group_cols, st);
            if (rc) return rc;
            if (fixed_ref) {
                rc = launch_pair(p, rfrag, rbias, cop, partial, hdr, limit, 0, exclude_diag != 0, diag_offset, 0,
                                 redo, 0, nullptr, st);
                if (rc) return rc;
                if (!fuse_tail) {
                    rc = lvae_launch_finalize(partial, p.nsplit, p.cpad, M, nullptr, log_norm, log_qz, row_max,
                                              row_sum, hdr, limit, 0, 1, nullptr, redo, group_cols, st);
                    if (rc) return rc;
                }
            }
        }
        if (joint_path != LVAE_PATH_MFMA && !fuse_tail) {
            rc = lvae_launch_finalize(dpartial, p.dsplit, p.cpad, M, nullptr, log_norm, log_qz, row_max,
                                      row_sum, hdr, limit, 1, 0, nullptr, nullptr, group_cols, st);
            if (rc) return rc;
        }
    }
    if (with_pm) {
        rc = lvae_launch_finalize_pm(dpartial_pm, p.dsplit, p.cpad, M, D, p.DC, log_norm,
                                     (float*)(base + p.off_pmtmp), log_qz_pm, st);
        if (rc) return rc;
    }
    if (fuse_tail) {
        LateMerge lm{};
        lm.partial_a = partial;
        lm.nsplit_a = p.nsplit;
        lm.partial_b = joint_path != LVAE_PATH_MFMA ? dpartial : nullptr;
        lm.nsplit_b = p.dsplit;
        lm.hdr = hdr;
        lm.flags = redo;
        lm.out = log_qz;
        lm.cpad = p.cpad;
        lm.group_cols = group_cols;
        lm.use_ref_a = 1;
        lm.cond_limit = limit;
        lm.log_norm = log_norm;
        if (!tail) return lvae_launch_latent_sums(nullptr, nullptr, nullptr, log_qz, nullptr, M, D, nullptr, st, true, &lm);
        return lvae_launch_latent_sums(z, tail->mu, tail->logvar, log_qz, nullptr, M, D, tail->sums, st, true, &lm);
    }
    if (tail) return lvae_launch_latent_sums(z, tail->mu, tail->logvar, log_qz, log_qz_pm, M, D, tail->sums, st, p.DP != 0);
    return LVAE_OK;
}

extern "C" int lvae_latent_stats_f32(const float* z, const float* mu_all, const float* logvar_all, int64_t M, int64_t N,
                                     int D, float log_norm, int path, const float* mu_local,
                                     const float* logvar_local, float* log_qz, double* sums, void* ws,
                                     int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(log_qz && sums && mu_local && logvar_local, "log_qz, sums, mu_local and logvar_local must be non-NULL");
    const StatsTail tail{mu_local, logvar_local, sums};
    return run_log_qz(z, mu_all, logvar_all, LVAE_SPREAD_LOGVAR, M, N, D, log_norm, 0, 0, path, log_qz, nullptr, nullptr,
                      nullptr, ws, ws_bytes, (hipStream_t)stream, &tail);
}

extern "C" int lvae_plan_info(int64_t M, int64_t N, int D, int prod_marginals, int64_t* out) {
    LVAE_CHECK_ARG(out, "out is NULL");
    LVAE_CHECK_ARG(M >= 1 && N >= 1 && D >= 1, "need M, N, D >= 1");
    const PairPlan p = lvae_log_qz_plan(N, M, D, prod_marginals != 0, false);
    out[0] = p.DP; out[1] = p.NCB; out[2] = p.cpad; out[3] = p.nsplit; out[4] = p.blk_per_split;
    out[5] = p.dsplit; out[6] = p.nblk; out[7] = (int64_t)p.total;
    return LVAE_OK;
}

extern "C" int lvae_log_qz_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                               int64_t M, int64_t N, int D, float log_norm, int exclude_diag,
                               int64_t diag_offset, int path, float* log_qz, float* log_qz_prod_marg,
                               void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(log_qz || log_qz_prod_marg, "at least one of log_qz / log_qz_prod_marg is required");
    return run_log_qz(z, mu, spread, spread_kind, M, N, D, log_norm, exclude_diag, diag_offset, path,
                      log_qz, log_qz_prod_marg, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

// ---- allocation-free variant of lvae_log_qz_f32 -----------------------------------------------------------
// For callers that cannot allocate device memory (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the workspace is this file-scope device array instead of a caller-provided buffer.  Every GPU the
// library runs on holds its own copy at its own address until the process ends.  In host code the array's NAME
// is only a stand-in for the symbol: the device address is looked up once per device (hipGetSymbolAddress with
// that device current) and only that pointer is ever handed to kernels or memsets.
static __device__ __align__(256) unsigned char scratch[LVAE_NOALLOC_SCRATCH_BYTES];
#define LVAE_NOALLOC_MAX_DEVICES 64
static void* g_scratch_ptr[LVAE_NOALLOC_MAX_DEVICES];     // device address of `scratch`, per device ordinal
static std::mutex g_scratch_mu;

static int scratch_of_current_device(void** out) {
    int dev = -1;
    LVAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= LVAE_NOALLOC_MAX_DEVICES)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, LVAE_NOALLOC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        LVAE_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return lvae_set_error(LVAE_ERR_HIP, "hipGetSymbolAddress returned a NULL address for the scratch array");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return LVAE_OK;
}

extern "C" int lvae_log_qz_noalloc_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                                       int64_t M, int64_t N, int D, float log_norm, int exclude_diag,
                                       int64_t diag_offset, int path, float* log_qz, float* log_qz_prod_marg,
                                       void* stream) {
    LVAE_CHECK_ARG(log_qz || log_qz_prod_marg, "at least one of log_qz / log_qz_prod_marg is required");
    int rc = check_pair_shapes(z, mu, spread, spread_kind, M, N, D);
    if (rc) return rc;
    const PairPlan p = lvae_log_qz_plan(N, M, D, log_qz_prod_marg != nullptr, path == LVAE_PATH_MFMA_BF16X3,
                                        exclude_diag != 0);
    if (p.total > (size_t)LVAE_NOALLOC_SCRATCH_BYTES)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED,
                              "M=%lld, N=%lld, D=%d%s needs %zu bytes of scratch; the allocation-free entry point holds "
                              "%zu per GPU (use lvae_log_qz_f32 with a caller-provided workspace)",
                              (long long)M, (long long)N, D, log_qz_prod_marg ? " with prod-marginals" : "", p.total,
                              (size_t)LVAE_NOALLOC_SCRATCH_BYTES);
    void* ws = nullptr;
    rc = scratch_of_current_device(&ws);
    if (rc) return rc;
    // run_log_qz clears what must start at zero (header, redo flags) at the start of every call; everything else
    // in the workspace is written before it is read, so the previous call's contents do no harm.
    return run_log_qz(z, mu, spread, spread_kind, M, N, D, log_norm, exclude_diag, diag_offset, path, log_qz,
                      log_qz_prod_marg, nullptr, nullptr, ws, (int64_t)LVAE_NOALLOC_SCRATCH_BYTES, (hipStream_t)stream);
}

// ---- backward of the element-wise estimators -------------------------------------------------------------
extern "C" int lvae_sample_ll_bwd_f32(const float* z, const float* mu, const float* logvar, int64_t B, int64_t K,
                                      int D, const float* g, int g_per_element, float* d_z, float* d_mu,
                                      float* d_logvar, void* stream) {
    LVAE_CHECK_ARG(z && g, "z and g must be non-NULL");
    LVAE_CHECK_ARG((mu == nullptr) == (logvar == nullptr), "mu and logvar must be given together");
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1");
    LVAE_CHECK_ARG(d_z || d_mu || d_logvar, "at least one output gradient is required");
    LVAE_CHECK_ARG(mu || (!d_mu && !d_logvar), "the prior branch has no mu / logvar gradients");
    return lvae_launch_sample_ll_bwd(z, mu, logvar, g, g_per_element != 0, B, K, D, d_z, d_mu, d_logvar,
                                     (hipStream_t)stream);
}

extern "C" int lvae_reparam_bwd_f32(const float* logvar, const float* eps, const float* g_z, int64_t B, int64_t K,
                                   int D, float* d_mu, float* d_logvar, void* stream) {
    LVAE_CHECK_ARG(logvar && eps && g_z, "logvar, eps and g_z must be non-NULL");
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1");
    LVAE_CHECK_ARG(d_mu || d_logvar, "at least one output gradient is required");
    return lvae_launch_reparam_bwd(logvar, eps, g_z, B, K, D, d_mu, d_logvar, (hipStream_t)stream);
}

extern "C" int lvae_kl_diag_bwd_f32(const float* mu, const float* logvar, int64_t N, int D, float lambda,
                                    const float* g_kl, const float* g_hinge, int g_per_row, float* d_mu,
                                    float* d_logvar, void* stream) {
    LVAE_CHECK_ARG(mu && logvar, "mu and logvar must be non-NULL");
    LVAE_CHECK_ARG(N >= 1 && D >= 1, "need N, D >= 1");
    LVAE_CHECK_ARG(g_kl || g_hinge, "at least one upstream gradient is required");
    LVAE_CHECK_ARG(d_mu || d_logvar, "at least one output gradient is required");
    return lvae_launch_kl_bwd(mu, logvar, g_kl, g_hinge, g_per_row != 0, N, D, lambda, d_mu, d_logvar,
                              (hipStream_t)stream);
}

// ---- backward of log q(z) / prod-marginals -------------------------------------------------------------
struct GradLayout {
    PairPlan p;
    int nsA, nsB;                 // splits of the looped axis: posteriors (sample sweep), samples (posterior sweep)
    int64_t rowsA, rowsB;
    BwdMfmaLayout mf;             // matrix-core sweeps (joint density only, D <= 128)
    size_t off_lse, off_rec, off_pdz, off_pdp, off_pmat, off_mfma, off_lqacc, total;
};
#define LVAE_GRAD_WIDE_MAX_PAIRS (1LL << 28)     // D > 64: the explicit N x M pair matrix is capped at 1 GiB
// Largest conditioning estimate (pair_grad_mfma.hip: the largest eight-dimension group of cancelling terms, times
// sqrt(groups)) for which the matrix-core backward holds 1e-4 relative on the gradients.  Measured against float64
// (tools/diag_bwd.py): d z and d mu stay below 3e-6 up to an estimate of 1e5; d logvar, whose second-moment sums
// cancel like the expansion itself, reaches 2e-5 at 1e4-3e4 and 1e-4..2e-4 at 6e4-2e5.
#define LVAE_BWD_COND_LIMIT 2.0e4f
static int grad_splits(int64_t kept, int64_t looped, int64_t* rows) {
    const int64_t waves = lvae_ceil_div(kept, 64);
    int64_t ns = lvae_ceil_div(2048, waves);
    const int64_t cap = lvae_ceil_div(looped, 16);
    if (ns > cap) ns = cap;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    *rows = lvae_ceil_div(looped, ns);
    return (int)lvae_ceil_div(looped, *rows);
}

static GradLayout grad_layout(int64_t M, int64_t N, int D, bool pm) {
    GradLayout g;
    g.p = lvae_make_pair_plan(N, M, D, pm);
    g.nsA = grad_splits(M, N, &g.rowsA);
    g.nsB = grad_splits(N, M, &g.rowsB);
    const int DC = g.p.DC;
    size_t o = g.p.total;
    g.off_lse = o; o = lvae_align256(o + (size_t)M * DC * sizeof(float));
    g.off_rec = o; o = lvae_align256(o + (size_t)M * (2 * DC + 4) * sizeof(float));
    g.off_pdz = o; o = lvae_align256(o + (size_t)g.nsA * M * DC * sizeof(float));
    g.off_pdp = o; o = lvae_align256(o + (size_t)g.nsB * N * 2 * DC * sizeof(float));
    g.off_lqacc = o; o = lvae_align256(o + (size_t)g.p.cpad * sizeof(float));
    g.off_pmat = o;
    if (D > 64 && M * N <= LVAE_GRAD_WIDE_MAX_PAIRS) o = lvae_align256(o + (size_t)M * N * sizeof(float));
    g.off_mfma = o;
    g.mf = BwdMfmaLayout{};
    if (!pm) {
        g.mf = lvae_bwd_mfma_layout(M, N, D);
        if (g.mf.DP) o = lvae_align256(o + g.mf.total);
    }
    g.total = o;
    return g;
}

extern "C" int lvae_log_qz_bwd_workspace_bytes(int64_t M, int64_t N, int D, int prod_marginals,
                                               int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(M >= 1 && N >= 1 && D >= 1, "need M, N, D >= 1");
    // (a workspace sized for the prod-marginals gradient also serves a call that passes g_log_qz alone)
    size_t need = grad_layout(M, N, D, false).total;
    if (prod_marginals) {
        const size_t with_pm = grad_layout(M, N, D, true).total;
        if (with_pm > need) need = with_pm;
    }
    *out_bytes = (int64_t)need;
    return LVAE_OK;
}

extern "C" int lvae_log_qz_bwd_f32(const float* z, const float* mu, const float* logvar, int64_t M, int64_t N,
                                   int D, float log_norm, const float* log_qz, const float* g_log_qz,
                                   const float* g_prod_marg, float* d_z, float* d_mu, float* d_logvar,
                                   void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_pair_shapes(z, mu, logvar, LVAE_SPREAD_LOGVAR, M, N, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(g_log_qz || g_prod_marg, "at least one upstream gradient is required");
    LVAE_CHECK_ARG(!g_log_qz || log_qz, "log_qz (the forward result) is required with g_log_qz");
    LVAE_CHECK_ARG(d_z || d_mu || d_logvar, "at least one output gradient is required");
    const bool pm = g_prod_marg != nullptr;
    // Path: the joint-density gradient runs on the matrix cores (D <= 128), behind a device-side conditioning gate
    // when the vector-ALU form can serve the shape as well (both are launched, one exits); gradients through the
    // prod-marginals term take the vector-ALU form.  LVAE_BWD_PATH = valu | mfma forces one (tests, benchmarks;
    // mfma: ungated).
    const bool mfma_ok = !pm && lvae_pad_dim(D) != 0;
    const bool valu_ok = D <= 64 || M * N <= LVAE_GRAD_WIDE_MAX_PAIRS;
    const int forced = lvae_switches().bwd_path;
    bool use_mfma = mfma_ok;      // (at every size: 50 vs 185 us at 512^2, D = 32; 86 vs 335 us at 2,048^2)
    bool use_valu = !use_mfma || valu_ok;
    if (forced == 1) { use_mfma = false; use_valu = true; }
    if (forced == 2 && mfma_ok) { use_mfma = true; use_valu = false; }
    if (use_valu && !valu_ok)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED,
                              "the backward pass for D > 64 with prod-marginals gradients (or D > 128) keeps an explicit "
                              "N x M pair matrix and is sized for training batches: M * N <= 2^28 (got M=%lld, N=%lld)",
                              (long long)M, (long long)N);
    const GradLayout g = grad_layout(M, N, D, pm);
    rc = check_ws(ws, ws_bytes, g.total);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const PairPlan& p = g.p;
    char* base = (char*)ws;
    PairHeader* hdr = (PairHeader*)(base + p.off_hdr);
    const PairHeader* gate = nullptr;        // non-NULL: the vector-ALU kernels run iff cond > LVAE_BWD_COND_LIMIT
    if (use_mfma) {
        rc = clear_and_colsum(mu, N, D, hdr, sizeof(PairHeader), st);
        if (rc) return rc;
        // (64 < D <= 128 beyond 2^28 pairs: the vector-ALU form cannot serve the shape, so there is no branch for an
        //  estimate past the limit -- the gradients then come out as NaN rather than silently below the 1e-4 bar;
        //  LVAE_BWD_PATH=mfma, a test switch, runs it ungated and unpoisoned)
        const float poison = (!use_valu && forced != 2) ? LVAE_BWD_COND_LIMIT : -1.0f;
        rc = lvae_launch_bwd_mfma(g.mf, base + g.off_mfma, z, mu, logvar, M, N, D, log_norm, log_qz, g_log_qz, hdr,
                                  use_valu ? LVAE_BWD_COND_LIMIT : -1.0f, d_z, d_mu, d_logvar, st, poison);
        if (rc || !use_valu) return rc;
        gate = hdr;
    }
    float* dparams = (float*)(base + p.off_dparams);
    float* lse2 = pm ? (float*)(base + g.off_lse) : nullptr;
    float* rec = (float*)(base + g.off_rec);
    float* pdz = (float*)(base + g.off_pdz);
    float* pdp = (float*)(base + g.off_pdp);
    rc = lvae_launch_prep_dparams(mu, logvar, LVAE_SPREAD_LOGVAR, N, D, p.DC, dparams, gate, LVAE_BWD_COND_LIMIT, st);
    if (rc) return rc;
    // One forward sweep in the direct form (exact differences) ahead of the gradient sweeps: the per-(sample, dimension)
    // logsumexp the prod-marginals weights need, and the joint logsumexp L_i again -- the caller's log_qz may come
    // from the matrix-core expansion, whose absolute error (a few 1e-4 at D ~ 100) would be the RELATIVE error of
    // every weight exp(v_ij - L_i); recomputed here the weights of a sample sum to one to fp32 rounding.
    const float gate_limit = gate ? LVAE_BWD_COND_LIMIT : -1.0f;
    float* lq_acc = (float*)(base + g.off_lqacc);
    rc = lvae_launch_pair_direct(p, g_log_qz != nullptr, pm, z, dparams, (float2*)(base + p.off_dpartial),
                                 (float2*)(base + p.off_dpartial_pm), (int*)(base + p.off_dflags), hdr, gate_limit, 1,
                                 0, 0, st);
    if (rc) return rc;
    if (pm) {
        rc = lvae_launch_finalize_pm_lse((const float2*)(base + p.off_dpartial_pm), p.dsplit, p.cpad, M, p.DC,
                                         lse2, st);
        if (rc) return rc;
    }
    if (g_log_qz) {
        rc = lvae_launch_finalize((const float2*)(base + p.off_dpartial), p.dsplit, p.cpad, M, nullptr, log_norm, lq_acc,
                                  nullptr, nullptr, hdr, gate_limit, 1, 0, nullptr, nullptr, 128 * p.NCB, st);
        if (rc) return rc;
    }
    const float* gq = g_log_qz;
    const float* lq = g_log_qz ? lq_acc : nullptr;
    if (D > 64) {
        float* pmat = (float*)(base + g.off_pmat);
        if (gq) {   // (no gate of its own: a gated wide problem pays for the M x N writes whichever path is selected)
            rc = lvae_launch_pair_matrix(mu, logvar, LVAE_SPREAD_LOGVAR, z, N, M, D, 0, nullptr, pmat, st);
            if (rc) return rc;
        }
        return lvae_launch_grad_wide(z, dparams, pmat, lq, log_norm, gq, g_prod_marg, lse2, M, N, D, p.DC, d_z, d_mu,
                                     d_logvar, st, gate, LVAE_BWD_COND_LIMIT);
    }
    if (d_z) {
        rc = lvae_launch_grad_sample(z, dparams, lq, log_norm, gq, g_prod_marg, lse2, M, N, D, p.DC, g.nsA,
                                     g.rowsA, pdz, st, gate, LVAE_BWD_COND_LIMIT);
        if (rc) return rc;
        rc = lvae_launch_grad_reduce(pdz, g.nsA, M, p.DC, 0, D, d_z, st, gate, LVAE_BWD_COND_LIMIT);
        if (rc) return rc;
    }
    if (d_mu || d_logvar) {
        rc = lvae_launch_grad_records(z, lq, log_norm, gq, g_prod_marg, lse2, M, D, p.DC, rec, st);
        if (rc) return rc;
        rc = lvae_launch_grad_posterior(dparams, rec, pm, M, N, p.DC, g.nsB, g.rowsB, pdp, st, gate, LVAE_BWD_COND_LIMIT);
        if (rc) return rc;
        if (d_mu) {
            rc = lvae_launch_grad_reduce(pdp, g.nsB, N, 2 * p.DC, 0, D, d_mu, st, gate, LVAE_BWD_COND_LIMIT);
            if (rc) return rc;
        }
        if (d_logvar) {
            rc = lvae_launch_grad_reduce(pdp, g.nsB, N, 2 * p.DC, p.DC, D, d_logvar, st, gate, LVAE_BWD_COND_LIMIT);
            if (rc) return rc;
        }
    }
    return LVAE_OK;
}

extern "C" int lvae_log_qz_partial_f32(const float* z, const float* mu, const float* spread,
                                       int spread_kind, int64_t M, int64_t N, int D, int path,
                                       float* row_max, float* row_sum, void* ws, int64_t ws_bytes,
                                       void* stream) {
    LVAE_CHECK_ARG(row_max && row_sum, "row_max and row_sum must be non-NULL");
    return run_log_qz(z, mu, spread, spread_kind, M, N, D, 0.f, 0, 0, path, nullptr, nullptr, row_max,
                      row_sum, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int lvae_lse_combine_f32(const float* max_a, const float* sum_a, const float* max_b,
                                    const float* sum_b, int64_t n, float log_norm, float* out_log_qz,
                                    float* out_max, float* out_sum, void* stream) {
    LVAE_CHECK_ARG(max_a && sum_a, "max_a and sum_a must be non-NULL");
    LVAE_CHECK_ARG((max_b == nullptr) == (sum_b == nullptr), "max_b and sum_b come together");
    LVAE_CHECK_ARG((out_max == nullptr) == (out_sum == nullptr), "out_max and out_sum come together");
    LVAE_CHECK_ARG(out_log_qz || out_max, "nothing to write");
    LVAE_CHECK_ARG(n >= 1, "need n >= 1");
    return lvae_launch_lse_combine(max_a, sum_a, max_b, sum_b, n, log_norm, out_log_qz, out_max, out_sum,
                                   (hipStream_t)stream);
}

extern "C" int lvae_lse_rebase_f32(const float* max_a, const float* sum_a, const float* new_max, int64_t n,
                                   float* out_sum, void* stream) {
    LVAE_CHECK_ARG(max_a && sum_a && new_max && out_sum, "NULL pointer");
    LVAE_CHECK_ARG(n >= 1, "need n >= 1");
    return lvae_launch_lse_rebase(max_a, sum_a, new_max, n, out_sum, (hipStream_t)stream);
}

// Roles swapped: posteriors on the kept axis (one output each), samples on the reduced axis.
extern "C" int lvae_qz_estimate_f32(const float* z, const float* mu, const float* spread,
                                    int spread_kind, int64_t M, int64_t N, int D, float log_norm,
                                    float* out, void* ws, int64_t ws_bytes, void* stream) {
    int rc = check_pair_shapes(z, mu, spread, spread_kind, M, N, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(out, "out is NULL");
    hipStream_t st = (hipStream_t)stream;
    const PairPlan p = lvae_make_pair_plan(M, N, D, false, 0, false, true);
    rc = check_ws(ws, ws_bytes, p.total);
    if (rc) return rc;
    if (p.DP == 0)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "q_z_estimate needs D <= 128 (got D=%d)", D);
    char* base = (char*)ws;
    PairHeader* hdr = (PairHeader*)(base + p.off_hdr);
    float* rfrag = (float*)(base + p.off_rfrag);
    float* rbias = (float*)(base + p.off_rbias);
    float* cop = (float*)(base + p.off_cop);
    float* colbias = (float*)(base + p.off_colbias);
    float2* partial = (float2*)(base + p.off_partial);
    rc = clear_and_colsum(mu, N, D, hdr, sizeof(PairHeader), st);
    if (rc) return rc;
    rc = lvae_launch_prep(1, true, z, nullptr, 0, M, p.nblk * 32, D, p.DP, hdr, N, rfrag, rbias,
                          nullptr, p.DC, st);
    if (rc) return rc;
    rc = lvae_launch_prep(0, false, mu, spread, spread_kind, N, p.cpad, D, p.DP, hdr, N, cop, colbias,
                          nullptr, p.DC, st);
    if (rc) return rc;
    rc = lvae_launch_pair_lse_mfma(p, rfrag, rbias, cop, partial, hdr, -1.f, 0, false, 0, st);
    if (rc) return rc;
    return lvae_launch_finalize(partial, p.nsplit, p.cpad, N, colbias, log_norm, out, nullptr, nullptr,
                                hdr, -1.f, 0, 0, nullptr, nullptr, 1, st);
}

// ---- per-sample densities / sampling / IW ----------------------------------------------------------
static int check_bkd(int64_t B, int64_t K, int D) {
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1 (got B=%lld K=%lld D=%d)", (long long)B,
                   (long long)K, D);
    LVAE_CHECK_ARG(B < (1LL << 31), "B must be below 2^31");
    return LVAE_OK;
}

extern "C" int lvae_sample_ll_f32(const float* z, const float* mu, const float* logvar, int64_t B,
                                  int64_t K, int D, float* log_q_z_x, float* log_p_z, void* stream) {
    int rc = check_bkd(B, K, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(z, "z is NULL");
    LVAE_CHECK_ARG(log_q_z_x || log_p_z, "no output requested");
    LVAE_CHECK_ARG(!log_q_z_x || (mu && logvar), "log_q_z_x needs mu and logvar");
    LVAE_CHECK_ARG((mu == nullptr) == (logvar == nullptr), "mu and logvar must come together");
    return lvae_launch_sample_iw(false, mu, logvar, z, nullptr, B, K, D, nullptr, log_q_z_x, log_p_z,
                                 nullptr, (hipStream_t)stream);
}

extern "C" int lvae_sample_ll_elem_f32(const float* z, const float* mu, const float* logvar, int64_t B,
                                       int64_t K, int D, float* out, void* stream) {
    int rc = check_bkd(B, K, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(z && out, "z and out must be non-NULL");
    LVAE_CHECK_ARG((mu == nullptr) == (logvar == nullptr), "mu and logvar must come together");
    return lvae_launch_sample_ll_elem(mu, logvar, z, B, K, D, out, (hipStream_t)stream);
}

extern "C" int lvae_reparam_iw_f32(const float* mu, const float* logvar, const float* eps,
                                   const float* log_p_x_z, int64_t B, int64_t K, int D, float* z,
                                   float* log_q_z_x, float* log_p_z, float* iw_ll, void* stream) {
    int rc = check_bkd(B, K, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(mu && logvar && eps, "mu, logvar and eps must be non-NULL");
    LVAE_CHECK_ARG(z || log_q_z_x || log_p_z || iw_ll, "no output requested");
    hipEvent_t prof_stop = prof_mark((hipStream_t)stream, nullptr);
    rc = lvae_launch_sample_iw(true, mu, logvar, eps, log_p_x_z, B, K, D, z, log_q_z_x, log_p_z, iw_ll,
                               (hipStream_t)stream);
    if (prof_stop) prof_mark((hipStream_t)stream, prof_stop);
    return rc;
}

// ---- the whole training-time tail of encode() for one small batch, one launch each way --------------------------
extern "C" int lvae_encode_tail_supported(int64_t B, int64_t K, int D) { return lvae_encode_tail_tiny_ok(B, K, D) ? 1 : 0; }

extern "C" int lvae_encode_tail_f32(const float* mu, const float* logvar, const float* eps, const float* prior_sample,
                                    int64_t B, int64_t K, int D, float log_norm, float* z, float* log_q_z_x,
                                    float* log_p_z, float* log_qz, float* log_qz_prod_marg, float* scalars,
                                    void* stream) {
    LVAE_CHECK_ARG(mu && logvar && eps, "mu, logvar and eps must be non-NULL");
    LVAE_CHECK_ARG(z || log_q_z_x || log_p_z || log_qz || log_qz_prod_marg || scalars, "no output requested");
    if (!lvae_encode_tail_tiny_ok(B, K, D))
        return lvae_set_error(LVAE_ERR_UNSUPPORTED,
                              "the one-launch encode tail serves B <= 128, B * D <= 4096, B * K * D <= 2^20 "
                              "(got B=%lld K=%lld D=%d); use the general entry points", (long long)B, (long long)K, D);
    return lvae_launch_encode_tail_tiny(mu, logvar, eps, prior_sample, B, K, D, log_norm, z, log_q_z_x, log_p_z, log_qz,
                                        log_qz_prod_marg, scalars, (hipStream_t)stream);
}

extern "C" int lvae_encode_tail_bwd_f32(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K,
                                        int D, float log_norm, const float* g_z, const float* g_log_q_z_x,
                                        const float* g_log_p_z, const float* g_log_qz_mean,
                                        const float* g_prod_marg_mean, float* d_mu, float* d_logvar, void* stream) {
    LVAE_CHECK_ARG(mu && logvar && eps, "mu, logvar and eps must be non-NULL");
    LVAE_CHECK_ARG(d_mu || d_logvar, "at least one output gradient is required");
    if (!lvae_encode_tail_tiny_ok(B, K, D))
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "outside the envelope of the one-launch encode tail (B=%lld K=%lld D=%d)",
                              (long long)B, (long long)K, D);
    return lvae_launch_encode_tail_tiny_bwd(mu, logvar, eps, B, K, D, log_norm, g_z, g_log_q_z_x, g_log_p_z,
                                            g_log_qz_mean, g_prod_marg_mean, d_mu, d_logvar, (hipStream_t)stream);
}

// ---- the same tail for any training batch: one workgroup up to B = LVAE_TAIL_ONE_WG_MAX, B / 4 workgroups beyond ----
// mode: 0 = by size, 1 = the one-workgroup form (its envelope), 2 = the multi-workgroup form
#define LVAE_TAIL_ONE_WG_MAX 32
static int tail_form(int mode, int64_t B, int64_t K, int D) {      // 1 / 2, or 0: neither serves the shape
    const bool t = lvae_encode_tail_tiny_ok(B, K, D), m = lvae_encode_tail_mid_ok(B, K, D);
    if (mode == 1) return t ? 1 : 0;
    if (mode == 2) return m ? 2 : 0;
    if (t && (B <= LVAE_TAIL_ONE_WG_MAX || !m)) return 1;
    return m ? 2 : 0;
}

extern "C" int lvae_encode_tail_ws_supported(int64_t B, int64_t K, int D) { return tail_form(0, B, K, D) ? 1 : 0; }

extern "C" int lvae_encode_tail_workspace_bytes(int64_t B, int64_t K, int D, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1");
    *out_bytes = lvae_encode_tail_mid_ok(B, K, D) ? lvae_encode_tail_mid_workspace(B, D) : 0;
    return LVAE_OK;
}

extern "C" int lvae_encode_tail_ws_f32(const float* mu, const float* logvar, const float* eps, const float* prior_sample,
                                       int64_t B, int64_t K, int D, float log_norm, float* z, float* log_q_z_x,
                                       float* log_p_z, float* log_qz, float* log_qz_prod_marg, float* scalars, int mode,
                                       void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(mu && logvar && eps, "mu, logvar and eps must be non-NULL");
    LVAE_CHECK_ARG(z || log_q_z_x || log_p_z || log_qz || log_qz_prod_marg || scalars, "no output requested");
    LVAE_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (by size), 1 (one workgroup) or 2 (multi-workgroup)");
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1");
    const int form = tail_form(mode, B, K, D);
    if (form == 1)
        return lvae_launch_encode_tail_tiny(mu, logvar, eps, prior_sample, B, K, D, log_norm, z, log_q_z_x, log_p_z,
                                            log_qz, log_qz_prod_marg, scalars, (hipStream_t)stream);
    if (form == 0)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED,
                              "the fused encode tail serves D <= 64, B <= 4096, B * K * D <= 2^26 (one workgroup: B <= 128, "
                              "B * D <= 4096, B * K * D <= 2^20); got B=%lld K=%lld D=%d mode=%d",
                              (long long)B, (long long)K, D, mode);
    const int64_t need = lvae_encode_tail_mid_workspace(B, D);
    if (!ws || ws_bytes < need)
        return lvae_set_error(LVAE_ERR_WORKSPACE, "encode tail: workspace of %lld bytes needed, %lld given",
                              (long long)need, (long long)(ws ? ws_bytes : 0));
    return lvae_launch_encode_tail_mid(mu, logvar, eps, prior_sample, B, K, D, log_norm, z, log_q_z_x, log_p_z, log_qz,
                                       log_qz_prod_marg, scalars, ws, (hipStream_t)stream);
}

extern "C" int lvae_encode_tail_bwd_ws_f32(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K,
                                           int D, float log_norm, const float* g_z, const float* g_log_q_z_x,
                                           const float* g_log_p_z, const float* g_log_qz_mean,
                                           const float* g_prod_marg_mean, float* d_mu, float* d_logvar, int mode,
                                           void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(mu && logvar && eps, "mu, logvar and eps must be non-NULL");
    LVAE_CHECK_ARG(d_mu || d_logvar, "at least one output gradient is required");
    LVAE_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (by size), 1 (one workgroup) or 2 (multi-workgroup)");
    LVAE_CHECK_ARG(B >= 1 && K >= 1 && D >= 1, "need B, K, D >= 1");
    const int form = tail_form(mode, B, K, D);
    if (form == 1)
        return lvae_launch_encode_tail_tiny_bwd(mu, logvar, eps, B, K, D, log_norm, g_z, g_log_q_z_x, g_log_p_z,
                                                g_log_qz_mean, g_prod_marg_mean, d_mu, d_logvar, (hipStream_t)stream);
    if (form == 0)
        return lvae_set_error(LVAE_ERR_UNSUPPORTED, "outside the envelope of the fused encode tail (B=%lld K=%lld D=%d mode=%d)",
                              (long long)B, (long long)K, D, mode);
    const int64_t need = lvae_encode_tail_mid_workspace(B, D);
    if (!ws || ws_bytes < need)
        return lvae_set_error(LVAE_ERR_WORKSPACE, "encode tail backward: workspace of %lld bytes needed, %lld given",
                              (long long)need, (long long)(ws ? ws_bytes : 0));
    return lvae_launch_encode_tail_mid_bwd(mu, logvar, eps, B, K, D, log_norm, g_z, g_log_q_z_x, g_log_p_z, g_log_qz_mean,
                                           g_prod_marg_mean, d_mu, d_logvar, ws, (hipStream_t)stream);
}

extern "C" int lvae_iw_lse_f32(const float* log_p_x_z, const float* log_p_z, const float* log_q_z_x,
                               int64_t B, int64_t K, float* out, void* stream) {
    LVAE_CHECK_ARG(B >= 1 && K >= 1, "need B, K >= 1");
    LVAE_CHECK_ARG(log_p_x_z && log_p_z && log_q_z_x && out, "NULL pointer");
    return lvae_launch_iw_lse(log_p_x_z, log_p_z, log_q_z_x, B, K, out, (hipStream_t)stream);
}

extern "C" int lvae_reduce_f32(const float* x, int64_t n, int mode, double scale, double* out, void* ws,
                               int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(x && out, "NULL pointer");
    LVAE_CHECK_ARG(n >= 1, "need n >= 1");
    LVAE_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 (sum) or 1 (sum exp)");
    int rc = check_ws(ws, ws_bytes, LVAE_REDUCE_WS_BYTES);
    if (rc) return rc;
    return lvae_launch_reduce(mode, x, n, scale, out, (double*)ws, (hipStream_t)stream);
}

extern "C" int lvae_latent_sums_f32(const float* z, const float* mu, const float* logvar,
                                    const float* log_qz, const float* log_qz_prod_marg, int64_t M, int D,
                                    double* sums, void* stream) {
    LVAE_CHECK_ARG(z && mu && logvar && sums, "z, mu, logvar and sums must be non-NULL");
    LVAE_CHECK_ARG(M >= 1 && D >= 1, "need M >= 1 and D >= 1");
    return lvae_launch_latent_sums(z, mu, logvar, log_qz, log_qz_prod_marg, M, D, sums, (hipStream_t)stream);
}

// ---- MI bounds ---------------------------------------------------------------------------------------
static size_t bounds_tail_bytes(int64_t K) { return lvae_align256((size_t)K * 3 * sizeof(float)); }

// The two log q(z) calls of the bounds (all pairs / diagonal excluded) plan differently at D in (32, 64] -- two column
// blocks per wave without the excluded diagonal, one with it or on the split-bf16 contraction --: the region is sized
// for the larger plan.  (It used to be sized for the excluded-diagonal plan alone: K = 20,000 at D = 64 was refused
// with "workspace too small" by the first call.)
static size_t bounds_plan_bytes(int64_t K, int D) {
    const size_t a = lvae_log_qz_plan(K, K, D, false, false, false).total;
    const size_t b = lvae_log_qz_plan(K, K, D, false, true, true).total;
    return a > b ? a : b;
}

extern "C" int lvae_pair_bounds_workspace_bytes(int64_t K, int D, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(K >= 2 && D >= 1 && D <= 4096, "need K >= 2 and 1 <= D <= 4096");
    *out_bytes = (int64_t)(bounds_plan_bytes(K, D) + bounds_tail_bytes(K));
    return LVAE_OK;
}

extern "C" int lvae_pair_bounds_f32(const float* z, const float* mu, const float* var, int64_t K, int D,
                                    int path, double* out, void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(out, "out is NULL");
    LVAE_CHECK_ARG(K >= 2, "the MBU bound needs K >= 2 (got %lld)", (long long)K);
    int rc = check_pair_shapes(z, mu, var, LVAE_SPREAD_VAR, K, K, D);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t plan_bytes = bounds_plan_bytes(K, D);
    rc = check_ws(ws, ws_bytes, plan_bytes + bounds_tail_bytes(K));
    if (rc) return rc;
    float* tail = (float*)((char*)ws + plan_bytes);
    float *diag = tail, *lse_all = tail + K, *lse_off = tail + 2 * K;
    rc = run_log_qz(z, mu, var, LVAE_SPREAD_VAR, K, K, D, logf((float)K), 0, 0, path, lse_all, nullptr,
                    nullptr, nullptr, ws, (int64_t)plan_bytes, st);
    if (rc) return rc;
    rc = run_log_qz(z, mu, var, LVAE_SPREAD_VAR, K, K, D, logf((float)(K - 1)), 1, 0, path, lse_off, nullptr,
                    nullptr, nullptr, ws, (int64_t)plan_bytes, st);
    if (rc) return rc;
    rc = lvae_launch_diag_ll(z, mu, var, LVAE_SPREAD_VAR, K, D, diag, st);
    if (rc) return rc;
    return lvae_launch_bounds_finalize(diag, lse_all, lse_off, K, out, st);
}

extern "C" int lvae_log_prob_pairs_f32(const float* z, const float* mu, const float* spread,
                                       int spread_kind, int64_t M, int64_t N, int D, float* out,
                                       void* stream) {
    int rc = check_pair_shapes(z, mu, spread, spread_kind, M, N, D);
    if (rc) return rc;
    LVAE_CHECK_ARG(out, "out is NULL");
    return lvae_launch_pair_matrix(mu, spread, spread_kind, z, N, M, D, 0, nullptr, out, (hipStream_t)stream);
}

extern "C" int lvae_gauss_ll_masked_f32(const float* sample, const float* mu, const float* var, const float* mask,
                                        int ndim, const int64_t* shape, const int64_t* st_sample,
                                        const int64_t* st_mu, const int64_t* st_var, const int64_t* st_mask, int dim,
                                        float* out, void* stream) {
    LVAE_CHECK_ARG(sample && mask && out && shape && st_sample && st_mask, "sample, mask, out, shape and strides must be non-NULL");
    LVAE_CHECK_ARG(ndim >= 1 && ndim <= 6, "need 1 <= ndim <= 6 (got %d)", ndim);
    LVAE_CHECK_ARG(dim >= 0 && dim < ndim, "dim must name an axis of the broadcast shape");
    LVAE_CHECK_ARG((!mu || st_mu) && (!var || st_var), "strides missing for mu / var");
    int64_t n_out = 1;
    for (int k = 0; k < ndim; ++k) {
        LVAE_CHECK_ARG(shape[k] >= 1 && shape[k] < (1LL << 31), "bad extent on axis %d", k);
        LVAE_CHECK_ARG(st_sample[k] >= 0 && st_mask[k] >= 0 && (!mu || st_mu[k] >= 0) && (!var || st_var[k] >= 0),
                       "strides must be non-negative");
        if (k != dim) n_out *= shape[k];
        LVAE_CHECK_ARG(n_out < (1LL << 40), "output too large");
    }
    return lvae_launch_gauss_ll_masked(sample, mu, var, mask, ndim, shape, st_sample, st_mu, st_var, st_mask, dim,
                                       n_out, out, (hipStream_t)stream);
}

// ---- KL, column means, KDE --------------------------------------------------------------------------
extern "C" int lvae_kl_diag_f32(const float* mu, const float* logvar, int64_t N, int D, float lambda,
                                float* kl_elem, float* hinge_elem, float* kl_row, float* hinge_row,
                                void* stream) {
    LVAE_CHECK_ARG(mu && logvar, "mu and logvar must be non-NULL");
    LVAE_CHECK_ARG(N >= 1 && D >= 1, "need N >= 1 and D >= 1");
    LVAE_CHECK_ARG(kl_elem || hinge_elem || kl_row || hinge_row, "no output requested");
    return lvae_launch_kl(mu, logvar, N, D, lambda, kl_elem, hinge_elem, kl_row, hinge_row,
                          (hipStream_t)stream);
}

extern "C" int lvae_colmean_f32(const float* x, int64_t n, int64_t c, float* out, void* ws,
                                int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(x && out, "NULL pointer");
    LVAE_CHECK_ARG(n >= 1 && c >= 1, "need n >= 1 and c >= 1");
    int rc = check_ws(ws, ws_bytes, (size_t)c * sizeof(double));
    if (rc) return rc;
    return lvae_launch_colmean(x, n, c, out, (double*)ws, (hipStream_t)stream);
}

extern "C" int lvae_kde_workspace_bytes(int64_t n, int64_t m, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(n >= 1 && m >= 1, "need n, m >= 1");
    int nsplit;
    int64_t rows;
    lvae_kde_plan(n, m, &nsplit, &rows);
    *out_bytes = (int64_t)(((size_t)nsplit + 1) * m * sizeof(float));
    return LVAE_OK;
}

extern "C" int lvae_kde_score_f32(const float* X, const float* Y, int64_t n, int64_t n_norm, int64_t m, int d,
                                  float bw, int stabilise, float* out, void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(X && Y && out, "NULL pointer");
    LVAE_CHECK_ARG(n >= 1 && m >= 1 && d >= 1, "need n, m, d >= 1");
    LVAE_CHECK_ARG(n_norm >= 0, "n_norm must be >= 0 (0: normalise by n)");
    if (n_norm == 0) n_norm = n;
    LVAE_CHECK_ARG(bw > 0.f, "bandwidth must be positive");
    int nsplit;
    int64_t rows;
    lvae_kde_plan(n, m, &nsplit, &rows);
    int rc = check_ws(ws, ws_bytes, ((size_t)nsplit + 1) * m * sizeof(float));
    if (rc) return rc;
    return lvae_launch_kde(X, Y, n, n_norm, m, d, bw, stabilise, out, (float*)ws, (hipStream_t)stream);
}

// ---- Categorical MI bounds, KS statistic -------------------------------------------------------------------
extern "C" int lvae_categorical_bounds_workspace_bytes(int64_t K, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(K >= 2, "the MBU bound needs K >= 2 (got %lld)", (long long)K);
    *out_bytes = (int64_t)lvae_align256(lvae_cat_bounds_workspace(K));
    return LVAE_OK;
}

extern "C" int lvae_categorical_bounds_f32(const float* probs, const int64_t* samples, int64_t K, int64_t V,
                                           double* out, void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(probs && samples && out, "probs, samples and out must be non-NULL");
    LVAE_CHECK_ARG(K >= 2 && K < (1LL << 31), "the MBU bound needs 2 <= K < 2^31 (got %lld)", (long long)K);
    LVAE_CHECK_ARG(V >= 1, "need V >= 1");
    int rc = check_ws(ws, ws_bytes, lvae_align256(lvae_cat_bounds_workspace(K)));
    if (rc) return rc;
    return lvae_launch_cat_bounds(probs, samples, K, V, out, ws, (hipStream_t)stream);
}

extern "C" int lvae_ks_2samp_workspace_bytes(int64_t n1, int64_t n2, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(n1 >= 1 && n2 >= 1 && n1 + n2 <= (1LL << 30), "need n1, n2 >= 1 and n1 + n2 <= 2^30");
    *out_bytes = (int64_t)lvae_ks_workspace(n1, n2);
    return LVAE_OK;
}

extern "C" int lvae_ks_2samp_f32(const float* a, int64_t n1, const float* b, int64_t n2, double* out, void* ws,
                                 int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(a && b && out, "a, b and out must be non-NULL");
    LVAE_CHECK_ARG(n1 >= 1 && n2 >= 1 && n1 + n2 <= (1LL << 30), "need n1, n2 >= 1 and n1 + n2 <= 2^30");
    int rc = check_ws(ws, ws_bytes, lvae_ks_workspace(n1, n2));
    if (rc) return rc;
    return lvae_launch_ks_2samp(a, n1, b, n2, out, ws, (hipStream_t)stream);
}

// The kernel sums contract two dimensions per MFMA step (norms enter as biases): their plans are made for D / 2.
static inline int ksum_plan_dim(int D) { return (D + 1) / 2; }

// ---- MMD -----------------------------------------------------------------------------------------------
// Splits the bandwidth list into (count of zeros, KsumParams for the positive ones).  If the positive
// alphas are all integer multiples (<= LVAE_KSUM_MAX) of the smallest one the sum of exponentials is a
// polynomial in t = exp(-base*v).
static int build_ksum(const float* alphas, int n_alpha, float eps, KsumParams* kp, int* zeros, float* alpha_max = nullptr) {
    LVAE_CHECK_ARG(alphas, "alphas is NULL");
    LVAE_CHECK_ARG(n_alpha >= 1 && n_alpha <= LVAE_KSUM_MAX, "need 1 <= n_alpha <= %d", LVAE_KSUM_MAX);
    memset(kp, 0, sizeof(*kp));
    kp->eps = eps;
    *zeros = 0;
    float base = 0.f, amax = 0.f;
    for (int k = 0; k < n_alpha; ++k) {
        LVAE_CHECK_ARG(alphas[k] >= 0.f && alphas[k] < 1e30f, "alphas must be finite and non-negative");
        if (alphas[k] == 0.f) ++*zeros;
        else if (base == 0.f || alphas[k] < base) base = alphas[k];
        if (alphas[k] > amax) amax = alphas[k];
    }
    if (alpha_max) *alpha_max = amax;
    if (*zeros == n_alpha) {      // nothing but alpha = 0: every pair contributes n_alpha
        kp->poly = 1; kp->n = 0; kp->nbase = 0.f; kp->c[0] = 0.f;
        return LVAE_OK;
    }
    bool poly = true;
    int mmax = 0;
    for (int k = 0; k < n_alpha && poly; ++k) {
        if (alphas[k] == 0.f) continue;
        const double ratio = (double)alphas[k] / (double)base;
        const double rr = floor(ratio + 0.5);
        if (fabs(ratio - rr) > 1e-4 * ratio || rr > LVAE_KSUM_MAX) poly = false;
        else if ((int)rr > mmax) mmax = (int)rr;
    }
    if (poly) {
        kp->poly = 1;
        kp->n = mmax;
        kp->nbase = -base * LVAE_LOG2E;
        for (int k = 0; k < n_alpha; ++k)
            if (alphas[k] != 0.f) kp->c[(int)floor((double)alphas[k] / (double)base + 0.5)] += 1.0f;
    } else {
        kp->poly = 0;
        int n = 0;
        for (int k = 0; k < n_alpha; ++k)
            if (alphas[k] != 0.f) kp->c[n++] = -alphas[k] * LVAE_LOG2E;
        kp->n = n;
    }
    return LVAE_OK;
}

// Workspace of one kernel-sum pass (kept = a [na], reduced = b [nb]).  The split search of the plan is not
// monotone in the shape or in `sym`, so every size below is the extent of the plan that will actually be laid
// out (or the maximum over the plans a call may lay out), never a "largest case" guess.
static size_t ksum_gate_extra(const PairPlan& p);
static size_t ksum_pass_extent(int64_t na, int64_t nb, int D, bool sym) {
    const PairPlan p = lvae_make_pair_plan(nb, na, ksum_plan_dim(D), false, 0, sym);
    return p.total + ksum_gate_extra(p);          // (the gated AUTO layout is the largest one of a pass)
}

// lvae_mmd_rbf_f32 runs three passes in one workspace: (x, x) and (y, y) symmetric, (x, y) plain.
static size_t mmd_pair_extent(int64_t n1, int64_t n2, int D) {
    size_t e = ksum_pass_extent(n1, n1, D, true);
    const size_t e2 = ksum_pass_extent(n2, n2, D, true), e3 = ksum_pass_extent(n1, n2, D, false);
    if (e2 > e) e = e2;
    if (e3 > e) e = e3;
    return e;
}

// Fused form of lvae_mmd_rbf_f32 (D <= 128, fp32 contraction, single-exponential epilogues): the POINT and SAMPLE
// operands of both samples live side by side and the three kernel matrices are ONE launch (pair_lse_multi_kernel):
//   header | x POINT frag + bias | y POINT frag + bias | x SAMPLE rows + norms | y SAMPLE rows + norms |
//   partials (x,y) | partials (x,x) | partials (y,y) | 4 float64
struct MmdFusedLayout {
    PairPlan pxy, pxx, pyy;
    size_t off_xr, off_xrb, off_yr, off_yrb, off_xc, off_xn, off_yc, off_yn, off_pxy, off_pxx, off_pyy, off_sums, total;
};
static MmdFusedLayout mmd_fused_layout(int64_t n1, int64_t n2, int D) {
    MmdFusedLayout L;
    const int dp = ksum_plan_dim(D);
    L.pxy = lvae_make_pair_plan(n2, n1, dp, false, 0, false);
    L.pxx = lvae_make_pair_plan(n1, n1, dp, false, 0, true);
    L.pyy = lvae_make_pair_plan(n2, n2, dp, false, 0, true);
    const int DP = L.pxy.DP;
    size_t o = lvae_align256(sizeof(PairHeader));
    auto take = [&](size_t bytes) { const size_t at = o; o = lvae_align256(o + bytes); return at; };
    L.off_xr = take((size_t)L.pxx.nblk * DP * 64 * sizeof(float));
    L.off_xrb = take(((size_t)L.pxx.nblk * 32 + 64) * sizeof(float));
    L.off_yr = take((size_t)L.pyy.nblk * DP * 64 * sizeof(float));
    L.off_yrb = take(((size_t)L.pyy.nblk * 32 + 64) * sizeof(float));
    L.off_xc = take((size_t)L.pxx.cpad * 2 * DP * sizeof(float));
    L.off_xn = take((size_t)L.pxx.cpad * sizeof(float));
    L.off_yc = take((size_t)L.pyy.cpad * 2 * DP * sizeof(float));
    L.off_yn = take((size_t)L.pyy.cpad * sizeof(float));
    L.off_pxy = take((size_t)L.pxy.nsplit * L.pxy.cpad * sizeof(double));
    L.off_pxx = take((size_t)L.pxx.nsplit * L.pxx.cpad * sizeof(double));
    L.off_pyy = take((size_t)L.pyy.nsplit * L.pyy.cpad * sizeof(double));
    L.off_sums = take(4 * sizeof(double));
    L.total = o;
    return L;
}
static bool mmd_fused_dims(int D) { return ksum_plan_dim(D) <= 64; }

extern "C" int lvae_mmd_workspace_bytes(int64_t n1, int64_t n2, int D, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(n1 >= 2 && n2 >= 2 && D >= 1, "need n1, n2 >= 2 and D >= 1");
    size_t need = mmd_pair_extent(n1, n2, D) + 256;           // one pass at a time (wide latents, split-bf16 mode, ...)
    if (mmd_fused_dims(D)) {
        const size_t f = mmd_fused_layout(n1, n2, D).total;   // all three passes side by side
        if (f > need) need = f;
    }
    *out_bytes = (int64_t)need;
    return LVAE_OK;
}

// Kernel-matrix sum over (kept = a [na], reduced = b [nb]) into *out (float64, device).
static int ksum_epi(const KsumParams& kp) {
    if (!kp.poly) return 3;
    if (kp.n <= 1) return 1;
    if (kp.n > 16) return 3;
    // coefficients in pairs (c0 = 0, c[2j+1] == c[2j+2]): the cheaper factored form, e.g. the reference's
    // bandwidths 0, 0.1 .. 1.4 (loss_and_optimisation.py:598)
    bool paired = kp.c[0] == 0.f && kp.n <= 14;
    for (int j = 0; j < 7 && paired; ++j) paired = kp.c[2 * j + 1] == kp.c[2 * j + 2];
    return paired ? 5 : 2;
}

// Contraction of one kernel-sum pass, chosen per call from the entry point's `path` argument:
//   wide   direct differences on the vector ALU (LVAE_PATH_DIRECT, and every path at D > 256);
//   bx     the cross term as three bf16 limbs per value on the bf16 matrix pipe (LVAE_PATH_MFMA_BF16X3, ungated);
//   gated  LVAE_PATH_AUTO on problems of at least LVAE_MMD_BX_MIN_PAIRS pairs: BOTH contractions are prepared and
//          launched, the operand preparation publishes max |point|^2 when it exceeds gate_limit, and on the device the
//          split-bf16 kernel runs iff it did not, the fp32 MFMA kernel iff it did (same test as the prod-marginals gate).
// The quantity that cancels in d2 = |x|^2 + |y|^2 - 2 x.y is at most 4 max |p|^2; one fp32 ulp of it times the largest
// bandwidth is the absolute error of the exponent, so the gate keeps alpha_max * 4 max|p|^2 <= LVAE_COND_LIMIT
// (<= 2.4e-3 in the exponent, the bound the log-density gate uses).  Below it the six limb products kept of nine
// differ from the fp32 contraction by <= 3 * 2^-24 of sum_d |x_d y_d| (measured: tools/accuracy_paths.py).
// Measured crossover (tools/bench_bf16x3.py: tts statistic at n = 16,384, 0.52 -> 0.45 ms): 2^27 pairs.
#define LVAE_MMD_BX_MIN_PAIRS 134217728.0
struct KsumMode {
    bool wide, bx, gated;
    float gate_limit;        // on max |p|^2 (gated only)
};
static int ksum_mode(int path, const PairPlan& p, int64_t na, int64_t nb, float alpha_max, bool perm, KsumMode* m) {
    LVAE_CHECK_ARG(path == LVAE_PATH_AUTO || path == LVAE_PATH_MFMA || path == LVAE_PATH_DIRECT ||
                       path == LVAE_PATH_MFMA_BF16X3, "unknown path %d", path);
    *m = KsumMode{false, false, false, -1.f};
    if (path == LVAE_PATH_DIRECT || p.DP == 0) {
        if (perm) return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the permutation test has no direct-difference form (D <= 256, paths AUTO / MFMA / MFMA_BF16X3)");
        if (p.DP == 0 && (path == LVAE_PATH_MFMA || path == LVAE_PATH_MFMA_BF16X3))
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the matrix-core kernel sums need D <= 256");
        m->wide = true;
        return LVAE_OK;
    }
    if (path == LVAE_PATH_MFMA_BF16X3) {
        if (!lvae_bf16x3_supported(p.DP))
            return lvae_set_error(LVAE_ERR_UNSUPPORTED, "the split-bf16 kernel sums need D <= 128");
        m->bx = true;
        return LVAE_OK;
    }
    if (path == LVAE_PATH_AUTO && lvae_bf16x3_supported(p.DP) && alpha_max > 0.f) {
        const long long forced = lvae_switches().mmd_bx_min_pairs;
        const double min_pairs = forced >= 0 ? (double)forced : LVAE_MMD_BX_MIN_PAIRS;
        if ((double)na * (double)nb >= min_pairs) {
            m->bx = m->gated = true;
            m->gate_limit = LVAE_COND_LIMIT / (4.0f * alpha_max);
        }
    }
    return LVAE_OK;
}
// gated passes keep the fp32 operands of the other branch behind the plan's own regions
static size_t ksum_gate_extra(const PairPlan& p) {
    if (!lvae_bf16x3_supported(p.DP)) return 0;
    return lvae_align256((size_t)p.nblk * p.DP * 64 * sizeof(float)) + lvae_align256((size_t)p.cpad * 2 * p.DP * sizeof(float));
}
struct KsumRegions {
    PairHeader* hdr;
    float *rfrag, *rbias, *cop, *colnorm, *rfrag32, *cop32;
    float2* partial;
};
static KsumRegions ksum_regions(const PairPlan& p, void* ws) {
    char* base = (char*)ws;
    KsumRegions r;
    r.hdr = (PairHeader*)(base + p.off_hdr);
    r.rfrag = (float*)(base + p.off_rfrag);
    r.rbias = (float*)(base + p.off_rbias);
    r.cop = (float*)(base + p.off_cop);
    r.colnorm = (float*)(base + p.off_colbias);
    r.partial = (float2*)(base + p.off_partial);
    r.rfrag32 = (float*)(base + p.total);
    r.cop32 = (float*)(base + p.total + lvae_align256((size_t)p.nblk * p.DP * 64 * sizeof(float)));
    return r;
}

// Operands of a kernel-sum pass: reduced axis = points b (linear-only POINT rows -2b, fragment-major, + |b|^2
// biases), kept axis = points a (linear-only SAMPLE rows, + |a|^2 into colnorm).  bx: the limb operands go to the
// plan's regions (the fp32 prep still writes the biases and the norms); gated: the fp32 operands as well, behind them.
static int prep_ksum_operands(const PairPlan& p, const KsumMode& m, const float* a, int64_t na, const float* b, int64_t nb,
                              int D, const KsumRegions& r, hipStream_t st) {
    float* rf32 = m.gated ? r.rfrag32 : (m.bx ? nullptr : r.rfrag);
    float* cop32 = m.gated ? r.cop32 : (m.bx ? nullptr : r.cop);
    const float floor = m.gated ? m.gate_limit : -1.f;
    int rc = lvae_launch_prep(3, true, b, nullptr, 0, nb, p.nblk * 32, D, p.DP, r.hdr, 1, rf32, r.rbias, nullptr, p.DC, st,
                              floor);
    if (rc) return rc;
    rc = lvae_launch_prep(4, false, a, nullptr, 0, na, p.cpad, D, p.DP, r.hdr, 1, cop32, r.colnorm, nullptr, p.DC, st,
                          floor);
    if (rc || !m.bx) return rc;
    rc = lvae_launch_prep_bf16x3(3, true, b, nullptr, 0, nb, p.nblk * 32, D, p.DP, r.hdr, 1, r.rfrag, st);
    if (rc) return rc;
    return lvae_launch_prep_bf16x3(4, false, a, nullptr, 0, na, p.cpad, D, p.DP, r.hdr, 1, r.cop, st);
}

// The pair launch(es) of one kernel-sum pass under its mode.
static int launch_ksum_pair(const PairPlan& p, const KsumMode& m, const KsumRegions& r, bool excl, int64_t excl_offset,
                            int epi, int sym, hipStream_t st) {
    if (!m.bx)
        return lvae_launch_pair_mfma(p, r.rfrag, r.rbias, r.cop, r.partial, r.hdr, -1.f, 0, excl, excl_offset, epi, nullptr,
                                     sym, r.colnorm, st);
    int rc = lvae_launch_pair_bf16x3(p, r.rfrag, r.rbias, r.cop, r.partial, r.hdr, m.gated ? m.gate_limit : -1.f, 0, excl,
                                     excl_offset, epi, nullptr, sym, r.colnorm, st);
    if (rc || !m.gated) return rc;
    return lvae_launch_pair_mfma(p, r.rfrag32, r.rbias, r.cop32, r.partial, r.hdr, m.gate_limit, 1, excl, excl_offset, epi,
                                 nullptr, sym, r.colnorm, st);
}

static int ksum_pass(const float* a, int64_t na, const float* b, int64_t nb, int D, bool excl, int epi,
                     const KsumMode& m, void* ws, size_t ws_extent, double* out, hipStream_t st) {
    const PairPlan p = lvae_make_pair_plan(nb, na, ksum_plan_dim(D), false, 0, a == b);
    int rc;
    if (p.total + (m.gated ? ksum_gate_extra(p) : 0) > ws_extent)
        return lvae_set_error(LVAE_ERR_WORKSPACE, "internal: kernel-sum pass needs %zu bytes, %zu reserved", p.total,
                              ws_extent);
    const KsumRegions r = ksum_regions(p, ws);
    if (m.wide) {         // direct differences on the vector ALU (no symmetric walk)
        rc = lvae_launch_ksum_wide(a, b, na, nb, D, r.hdr, excl, 0, p.nsplit, p.cpad, (double*)r.partial, st);
    } else {
        rc = prep_ksum_operands(p, m, a, na, b, nb, D, r, st);
        if (rc) return rc;
        rc = launch_ksum_pair(p, m, r, excl, 0, epi, a == b ? 1 : 0, st);
    }
    if (rc) return rc;
    return lvae_launch_finalize_ksum((const double*)r.partial, p.nsplit, p.cpad, na, out, st);
}

extern "C" int lvae_mmd_rbf_f32(const float* x, const float* y, int64_t n1, int64_t n2, int D,
                                const float* alphas, int n_alpha, float dist_eps, int unbiased, int path, double* out,
                                void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(x && y && out, "x, y and out must be non-NULL");
    LVAE_CHECK_ARG(n1 >= 2 && n2 >= 2, "need n1, n2 >= 2");
    LVAE_CHECK_ARG(D >= 1 && D <= 4096, "the MMD path needs 1 <= D <= 4096 (got %d)", D);
    LVAE_CHECK_ARG(n1 < (1LL << 31) && n2 < (1LL << 31), "n1, n2 must be below 2^31");
    KsumParams kp;
    int zeros = 0;
    float alpha_max = 0.f;
    int rc = build_ksum(alphas, n_alpha, dist_eps, &kp, &zeros, &alpha_max);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool all_zero = kp.n == 0 && kp.c[0] == 0.f;
    // the fused fp32 form serves LVAE_PATH_MFMA, and LVAE_PATH_AUTO wherever AUTO stays on fp32 (small problems)
    KsumMode mxy;
    rc = ksum_mode(path, lvae_make_pair_plan(n2, n1, ksum_plan_dim(D), false, 0, false), n1, n2, alpha_max, false, &mxy);
    if (rc) return rc;
    if (!all_zero && mmd_fused_dims(D) && !mxy.bx && !mxy.wide &&
        lvae_pair_multi_supported(lvae_pad_dim(ksum_plan_dim(D)), ksum_epi(kp)) && !lvae_switches().no_fused_mmd) {
        // ---- fused form: five launches (init, one operand prep, one pair launch for the three matrices, one total,
        //      the statistic) instead of eighteen
        const MmdFusedLayout L = mmd_fused_layout(n1, n2, D);
        rc = check_ws(ws, ws_bytes, L.total);
        if (rc) return rc;
        char* base = (char*)ws;
        PairHeader* hdr = (PairHeader*)base;
        double* sums = (double*)(base + L.off_sums);
        const int DP = L.pxy.DP;
        rc = lvae_launch_ksum_init(hdr, kp, sums, 4, st);
        if (rc) return rc;
        float* xr = (float*)(base + L.off_xr);  float* xrb = (float*)(base + L.off_xrb);
        float* yr = (float*)(base + L.off_yr);  float* yrb = (float*)(base + L.off_yrb);
        float* xc = (float*)(base + L.off_xc);  float* xn = (float*)(base + L.off_xn);
        float* yc = (float*)(base + L.off_yc);  float* yn = (float*)(base + L.off_yn);
        {
            const float* src[4] = {x, y, x, y};
            const int64_t nn[4] = {n1, n2, n1, n2};
            const int64_t npad[4] = {L.pxx.nblk * 32, L.pyy.nblk * 32, L.pxx.cpad, L.pyy.cpad};
            float* outp[4] = {xr, yr, xc, yc};
            float* bias[4] = {xrb, yrb, xn, yn};
            const int point[4] = {1, 1, 0, 0};
            rc = lvae_launch_prep_ksum_multi(src, nn, npad, outp, bias, point, 4, D, DP, hdr, st);
            if (rc) return rc;
        }
        const bool excl = unbiased != 0;
        const int64_t no_diag = (int64_t)1 << 40;             // the cross term has no diagonal to leave out
        PairJobs jobs{};
        auto fill = [&](PairJob& J, const PairPlan& p, const float* rf, const float* rb, const float* cp, const float* cn,
                        size_t off_partial, int sym, int64_t excl_offset) {
            J.rfrag = rf; J.rbias = rb; J.cop = cp; J.colnorm = cn; J.partial = (float2*)(base + off_partial);
            J.nblk = p.nblk; J.blk_per_split = p.blk_per_split; J.cpad = p.cpad; J.excl_offset = excl_offset;
            J.sym = sym; J.gx = (int)(p.cpad / (128 * p.NCB)); J.gy = p.nsplit;
        };
        fill(jobs.j[0], L.pxy, yr, yrb, xc, xn, L.off_pxy, 0, no_diag);     // kept = x, reduced = y: the fullest first
        fill(jobs.j[1], L.pxx, xr, xrb, xc, xn, L.off_pxx, 1, 0);
        fill(jobs.j[2], L.pyy, yr, yrb, yc, yn, L.off_pyy, 1, 0);
        rc = lvae_launch_pair_multi(DP, jobs, 3, hdr, excl, ksum_epi(kp), st);
        if (rc) return rc;
        {
            const double* parts[3] = {(const double*)(base + L.off_pxx), (const double*)(base + L.off_pyy),
                                      (const double*)(base + L.off_pxy)};          // sums = {xx, yy, xy}
            const int ns[3] = {L.pxx.nsplit, L.pyy.nsplit, L.pxy.nsplit};
            const int64_t cpads[3] = {L.pxx.cpad, L.pyy.cpad, L.pxy.cpad};
            const int64_t cs[3] = {n1, n2, n1};
            rc = lvae_launch_finalize_ksum_multi(parts, ns, cpads, cs, 3, sums, st);
            if (rc) return rc;
        }
        return lvae_launch_mmd_stat(sums, n1, n2, unbiased, zeros, out, st);
    }
    const size_t extent = mmd_pair_extent(n1, n2, D);
    rc = check_ws(ws, ws_bytes, extent + 256);
    if (rc) return rc;
    PairHeader* hdr = (PairHeader*)ws;
    double* sums = (double*)((char*)ws + extent);            // 3 float64 behind the pair workspace
    LVAE_HIP(hipMemsetAsync(hdr, 0, sizeof(PairHeader), st));   // colsum = 0 -> no centering shift
    rc = lvae_launch_set_ksum(hdr, kp, st);
    if (rc) return rc;
    const bool excl = unbiased != 0;
    if (kp.n == 0 && kp.c[0] == 0.f) {
        LVAE_HIP(hipMemsetAsync(sums, 0, 3 * sizeof(double), st));
    } else {
        const int epi = ksum_epi(kp);
        // (one mode for the three passes, the cross term's: a statistic is one contraction throughout, and the gate's
        //  estimate -- the largest squared norm of either sample, kept in the header across the passes -- is shared)
        rc = ksum_pass(x, n1, x, n1, D, excl, epi, mxy, ws, extent, sums + 0, st);
        if (rc) return rc;
        rc = ksum_pass(y, n2, y, n2, D, excl, epi, mxy, ws, extent, sums + 1, st);
        if (rc) return rc;
        rc = ksum_pass(x, n1, y, n2, D, false, epi, mxy, ws, extent, sums + 2, st);
        if (rc) return rc;
    }
    return lvae_launch_mmd_stat(sums, n1, n2, unbiased, zeros, out, st);
}

extern "C" int lvae_kernel_sum_workspace_bytes(int64_t na, int64_t nb, int D, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(na >= 1 && nb >= 1 && D >= 1, "need na, nb, D >= 1");
    const size_t e0 = ksum_pass_extent(na, nb, D, false), e1 = ksum_pass_extent(na, nb, D, true);
    *out_bytes = (int64_t)(e0 > e1 ? e0 : e1);     // serves either mode (neither plan is always the larger)
    return LVAE_OK;
}

extern "C" int lvae_kernel_sum_f32(const float* a, const float* b, int64_t na, int64_t nb, int D,
                                   const float* alphas, int n_alpha, float dist_eps, int exclude_diag,
                                   int64_t diag_offset, int symmetric, int path, double* out_total, double* out_colsum,
                                   void* ws, int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(a && b, "a and b must be non-NULL");
    LVAE_CHECK_ARG(out_total || out_colsum, "at least one of out_total / out_colsum is required");
    LVAE_CHECK_ARG(na >= 1 && nb >= 1, "need na, nb >= 1");
    LVAE_CHECK_ARG(D >= 1 && D <= 4096, "the kernel-sum path needs 1 <= D <= 4096 (got %d)", D);
    LVAE_CHECK_ARG(na < (1LL << 31) && nb < (1LL << 31), "na, nb must be below 2^31");
    LVAE_CHECK_ARG(!symmetric || (diag_offset >= 0 && diag_offset % 32 == 0 && diag_offset + na <= nb && !out_colsum),
                   "symmetric evaluation: a must be rows diag_offset.. of b, diag_offset %% 32 == 0, totals only");
    KsumParams kp;
    int zeros = 0;
    float alpha_max = 0.f;
    int rc = build_ksum(alphas, n_alpha, dist_eps, &kp, &zeros, &alpha_max);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int sym = symmetric != 0;
    const PairPlan p = lvae_make_pair_plan(nb, na, ksum_plan_dim(D), false, 0, sym);
    KsumMode m;
    rc = ksum_mode(path, p, na, nb, alpha_max, false, &m);
    if (rc) return rc;
    rc = check_ws(ws, ws_bytes, p.total + (m.gated ? ksum_gate_extra(p) : 0));
    if (rc) return rc;
    const KsumRegions r = ksum_regions(p, ws);
    LVAE_HIP(hipMemsetAsync(r.hdr, 0, sizeof(PairHeader), st));   // colsum = 0 -> no centering shift
    rc = lvae_launch_set_ksum(r.hdr, kp, st);
    if (rc) return rc;
    if (kp.n == 0 && kp.c[0] == 0.f) {      // only alpha = 0 bandwidths: no pair kernel needed
        LVAE_HIP(hipMemsetAsync(r.partial, 0, (size_t)p.nsplit * p.cpad * sizeof(double), st));
    } else if (m.wide) {
        // direct differences on the vector ALU; every pair of the owned rows is evaluated (a symmetric request is
        // served as a plain one: the owners' totals still add up)
        rc = lvae_launch_ksum_wide(a, b, na, nb, D, r.hdr, exclude_diag != 0, diag_offset, p.nsplit, p.cpad,
                                   (double*)r.partial, st);
        if (rc) return rc;
    } else {
        rc = prep_ksum_operands(p, m, a, na, b, nb, D, r, st);
        if (rc) return rc;
        rc = launch_ksum_pair(p, m, r, exclude_diag != 0, diag_offset, ksum_epi(kp), sym, st);
        if (rc) return rc;
    }
    return lvae_launch_ksum_columns((const double*)r.partial, p.nsplit, p.cpad, na, nb, exclude_diag != 0,
                                    diag_offset, zeros, out_colsum, out_total, st);
}

// ---- MMD permutation test ------------------------------------------------------------------------------
// workspace: pair plan | label fragments | diagonal-tile column sums [nsplit][cpad] | 256 float64
static size_t perm_labfrag_bytes(const PairPlan& p) { return lvae_align256((size_t)p.nblk * 512 * 16); }
static size_t perm_diag_bytes(const PairPlan& p) { return lvae_align256((size_t)p.nsplit * p.cpad * sizeof(double)); }
static PairPlan perm_plan(int64_t nc, int64_t n, int D, bool sym) {
    return lvae_make_pair_plan(n, nc, ksum_plan_dim(D), false, 1, sym);
}
// (label fragments, diagonal sums and the q_p accumulators sit behind the largest -- gated -- operand layout)
static size_t perm_tail_offset(const PairPlan& p) { return p.total + ksum_gate_extra(p); }
static size_t perm_extent(const PairPlan& p) {
    return perm_tail_offset(p) + perm_labfrag_bytes(p) + perm_diag_bytes(p) + 256 * sizeof(double);
}

extern "C" int lvae_mmd_perm_workspace_bytes(int64_t nc, int64_t n, int D, int64_t* out_bytes) {
    LVAE_CHECK_ARG(out_bytes, "out_bytes is NULL");
    LVAE_CHECK_ARG(nc >= 1 && n >= 2 && D >= 1, "need nc >= 1, n >= 2, D >= 1");
    const size_t e0 = perm_extent(perm_plan(nc, n, D, false)), e1 = perm_extent(perm_plan(nc, n, D, true));
    *out_bytes = (int64_t)(e0 > e1 ? e0 : e1);     // serves either mode
    return LVAE_OK;
}

extern "C" int lvae_mmd_perm_f32(const float* cols, int64_t nc, int64_t col_offset, const float* pooled,
                                 int64_t n, int D, const unsigned char* labels, int P, const float* alphas,
                                 int n_alpha, float dist_eps, int symmetric, int path, double* out, void* ws,
                                 int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(cols && pooled && labels && out, "cols, pooled, labels and out must be non-NULL");
    LVAE_CHECK_ARG(n >= 2 && nc >= 1 && col_offset >= 0 && col_offset + nc <= n,
                   "need n >= 2 and 0 <= col_offset, col_offset + nc <= n");
    LVAE_CHECK_ARG(D >= 1 && D <= 256, "the permutation test needs 1 <= D <= 256 (got %d)", D);
    LVAE_CHECK_ARG(n < (1LL << 31), "n must be below 2^31");
    LVAE_CHECK_ARG(P >= 1 && P <= 128, "need 1 <= P <= 128 labelings per call (got %d)", P);
    LVAE_CHECK_ARG(!symmetric || col_offset % 32 == 0, "symmetric evaluation needs col_offset %% 32 == 0");
    KsumParams kp;
    int zeros = 0;
    float alpha_max = 0.f;
    int rc = build_ksum(alphas, n_alpha, dist_eps, &kp, &zeros, &alpha_max);
    if (rc) return rc;
    (void)zeros;      // alpha == 0 bandwidths are left out of q, lr and T (lvae.h): they cancel in the statistic
    hipStream_t st = (hipStream_t)stream;
    const int sym = symmetric != 0;
    const PairPlan p = perm_plan(nc, n, D, sym);
    KsumMode m;
    rc = ksum_mode(path, p, nc, n, alpha_max, true, &m);
    if (rc) return rc;
    const size_t lab_bytes = perm_labfrag_bytes(p);
    const size_t diag_bytes = perm_diag_bytes(p);
    rc = check_ws(ws, ws_bytes, perm_extent(p));      // the extent of the layout used below, not of another mode's plan
    if (rc) return rc;
    char* base = (char*)ws;
    const KsumRegions r = ksum_regions(p, ws);
    const size_t tail = perm_tail_offset(p);
    void* labfrag = base + tail;
    double* diag_partial = (double*)(base + tail + lab_bytes);
    double* qperm = (double*)(base + tail + lab_bytes + diag_bytes);
    LVAE_HIP(hipMemsetAsync(r.hdr, 0, sizeof(PairHeader), st));
    LVAE_HIP(hipMemsetAsync(qperm, 0, 256 * sizeof(double), st));
    rc = lvae_launch_set_ksum(r.hdr, kp, st);
    if (rc) return rc;
    if (kp.n == 0 && kp.c[0] == 0.f) {      // only alpha = 0 bandwidths: the statistic is identically 0
        LVAE_HIP(hipMemsetAsync(out, 0, (2 * (size_t)P + 1) * sizeof(double), st));
        return LVAE_OK;
    }
    rc = lvae_launch_prep_labels(labels, P, n, p.nblk, labfrag, st);
    if (rc) return rc;
    rc = prep_ksum_operands(p, m, cols, nc, pooled, n, D, r, st);
    if (rc) return rc;
    PermArgs pa{};
    pa.labfrag = labfrag;
    pa.labels = labels;
    pa.qperm = qperm;
    pa.diag_partial = diag_partial;
    pa.n_lab = n;
    pa.col_offset = col_offset;
    pa.C = nc;
    pa.P = P;
    const int epi = ksum_epi(kp);
    hipEvent_t prof_stop = prof_mark(st, nullptr);
    if (!m.bx) {
        rc = lvae_launch_pair_mfma_perm(p, r.rfrag, r.rbias, r.cop, r.partial, r.hdr, -1.f, 0, col_offset, epi, pa, sym,
                                        r.colnorm, st);
    } else {
        // gated (LVAE_PATH_AUTO): both forms are launched, the device-side test on the operand preparation's estimate
        // lets exactly one of them run
        rc = lvae_launch_pair_bf16x3_perm(p, r.rfrag, r.rbias, r.cop, r.partial, r.hdr, m.gated ? m.gate_limit : -1.f, 0,
                                          col_offset, epi, pa, sym, r.colnorm, st);
        if (!rc && m.gated)
            rc = lvae_launch_pair_mfma_perm(p, r.rfrag32, r.rbias, r.cop32, r.partial, r.hdr, m.gate_limit, 1, col_offset,
                                            epi, pa, sym, r.colnorm, st);
    }
    if (rc) return rc;
    if (prof_stop) prof_mark(st, prof_stop);
    return lvae_launch_perm_lr((const double*)r.partial, diag_partial, p.nsplit, p.cpad, nc, labels, n, col_offset,
                               P, sym, qperm, out, st);
}

// Diagnostic, host only: the extent of the workspace layout an entry point uses for one mode.
extern "C" int lvae_layout_info(int kind, int64_t na, int64_t nb, int D, int symmetric, int64_t* out) {
    LVAE_CHECK_ARG(out, "out is NULL");
    LVAE_CHECK_ARG(na >= 1 && nb >= 1 && D >= 1, "need na, nb, D >= 1");
    const bool sym = symmetric != 0;
    if (kind == 0) {
        const PairPlan p = lvae_make_pair_plan(nb, na, ksum_plan_dim(D), false, 0, sym);
        out[0] = (int64_t)(p.total + ksum_gate_extra(p)); out[1] = p.nsplit; out[2] = p.cpad;
    } else if (kind == 1) {
        const PairPlan p = perm_plan(na, nb, D, sym);
        out[0] = (int64_t)perm_extent(p); out[1] = p.nsplit; out[2] = p.cpad;
    } else if (kind == 2) {
        // the pass of lvae_mmd_rbf_f32 selected by `symmetric`: 0 = (x, y), 1 = (x, x), 2 = (y, y)
        const PairPlan p = symmetric == 0 ? lvae_make_pair_plan(nb, na, ksum_plan_dim(D), false, 0, false)
                         : symmetric == 1 ? lvae_make_pair_plan(na, na, ksum_plan_dim(D), false, 0, true)
                                          : lvae_make_pair_plan(nb, nb, ksum_plan_dim(D), false, 0, true);
        out[0] = (int64_t)(p.total + ksum_gate_extra(p)) + 256; out[1] = p.nsplit; out[2] = p.cpad;
    } else {
        return lvae_set_error(LVAE_ERR_INVALID, "unknown layout kind %d", kind);
    }
    return LVAE_OK;
}

extern "C" int lvae_kernel_matrix_f32(const float* x, const float* y, int64_t n1, int64_t n2, int D,
                                      const float* alphas, int n_alpha, float dist_eps, float* out, void* ws,
                                      int64_t ws_bytes, void* stream) {
    LVAE_CHECK_ARG(x && y && out, "x, y and out must be non-NULL");
    LVAE_CHECK_ARG(n1 >= 1 && n2 >= 1 && D >= 1, "need n1, n2, D >= 1");
    KsumParams kp;
    int zeros = 0;
    int rc = build_ksum(alphas, n_alpha, dist_eps, &kp, &zeros);
    if (rc) return rc;
    if (kp.poly) kp.c[0] += (float)zeros;     // alpha = 0 terms: +1 per pair, a constant coefficient
    else {
        LVAE_CHECK_ARG(kp.n + zeros <= LVAE_KSUM_MAX, "too many bandwidths");
        for (int k = 0; k < zeros; ++k) kp.c[kp.n++] = 0.f;
    }
    rc = check_ws(ws, ws_bytes, LVAE_KMAT_WS_BYTES);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    PairHeader* hdr = (PairHeader*)ws;
    // (only the kernel-sum parameters are touched here; the normaliser slots at the end of the header belong to log q(z))
    static_assert(offsetof(PairHeader, slot) <= LVAE_KMAT_WS_BYTES, "header larger than the kernel-matrix workspace");
    rc = lvae_launch_set_ksum(hdr, kp, st);
    if (rc) return rc;
    return lvae_launch_pair_matrix(x, nullptr, 0, y, n1, n2, D, 1, hdr, out, st);
}
