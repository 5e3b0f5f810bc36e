/*
 * lvae.h -- C ABI of the MI355X-native latent-analysis library (liblvae.so, gfx950).
 *
 * The reference (ClaartjeBarkhof/language-transformer-vae) has no FFI: its boundary for this path is a
 * set of plain Python functions on torch tensors.  Each entry point below is
 * what a ctypes binding behind one of those functions calls; the comment on each names the reference
 * function it replaces (path:line under NewsVAE/).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is DEVICE memory, float32, row-major contiguous, owned by the caller, unless the entry
 *    point says otherwise (bandwidth lists live on the host; labels are uint8, class indices int64, scalar
 *    statistics float64);
 *  - the library never allocates user-visible memory: scratch comes in through `ws` / `ws_bytes`
 *    (size from the matching *_workspace_bytes query; 256-byte aligned);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on it;
 *  - return value: LVAE_OK (0) or a negative lvae_status; lvae_last_error() gives the text for the
 *    calling thread.  No exceptions, no aborts;
 *  - all log-densities are natural logs.
 */
#ifndef LVAE_H
#define LVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LVAE_OK = 0,
    LVAE_ERR_INVALID = -1,     /* bad shape / null pointer / bad flag          */
    LVAE_ERR_WORKSPACE = -2,   /* workspace missing, misaligned or too small   */
    LVAE_ERR_HIP = -3,         /* a HIP runtime call failed                    */
    LVAE_ERR_UNSUPPORTED = -4  /* valid request this build cannot serve        */
} lvae_status;

/* how the posterior spread array is parameterised */
#define LVAE_SPREAD_LOGVAR 0   /* loss_and_optimisation.py: logvar              */
#define LVAE_SPREAD_VAR 1      /* mutual_information.py: var / covariance diag  */

/* which kernel family evaluates the pair matrix */
#define LVAE_PATH_AUTO 0       /* fp32 MFMA unless the conditioning estimate says otherwise; the prod-marginals
                                * sweep (D <= 128, from 2^22 pairs) then evaluates its per-dimension polynomial on
                                * the bf16 matrix pipe (three limbs per value, fp32-equivalent), else on the VALU */
#define LVAE_PATH_MFMA 1       /* fp32 MFMA quadratic expansion                */
#define LVAE_PATH_DIRECT 2     /* VALU direct form (z-mu)^2 p, no cancellation */
#define LVAE_PATH_MFMA_BF16X3 3 /* the same expansion as three bf16 limbs per value (six limb products, fp32
                                * accumulate, fp32-equivalent accuracy); D <= 64 (with prod-marginals: their
                                * polynomial too, ungated).  For the log-density matrix (lvae_log_qz_f32 and
                                * friends) it is opt-in: the north star asks for the fp32 MFMA contraction there
                                * and LVAE_PATH_AUTO never picks this form.  The kernel sums of the MMD entry points
                                * (lvae_mmd_rbf_f32, lvae_kernel_sum_f32, lvae_mmd_perm_f32; D <= 128) take it under
                                * LVAE_PATH_AUTO from 2^27 pairs on, behind a device-side gate on the largest squared
                                * norm of the points (alpha_max * 4 max|p|^2 <= 2e4; beyond it the fp32 MFMA kernel
                                * runs): 1.55-1.7x the fp32 rate.  For those entry points LVAE_PATH_MFMA is the fp32
                                * contraction, LVAE_PATH_DIRECT exact differences on the vector ALU (any D; the only
                                * form at D > 256; not for the permutation sums). */

const char* lvae_last_error(void);
int lvae_version(void);

/* ---------------------------------------------------------------------------------------------
 * Aggregate-posterior log-density:  for every sample i
 *     log_qz[i]           = LSE_j  sum_d log N(z_id; mu_jd, var_jd)            - log_norm
 *     log_qz_prod_marg[i] = sum_d ( LSE_j log N(z_id; mu_jd, var_jd) - log_norm )   (optional)
 * Replaces approximate_log_q_z (loss_and_optimisation.py:139-221) and its [M,N,D] temporary; the
 * host passes log_norm = log(N*dataset_size) ("chen", :207) or log(N) (:215).
 * With exclude_diag != 0 the pair j == i + diag_offset is left out of the first reduction
 * (mi_upper_bound's masked logsumexp, mutual_information.py:207-208).
 * z [M,D]; mu, spread [N,D]; log_qz [M]; log_qz_prod_marg [M] or NULL.
 * ------------------------------------------------------------------------------------------- */
int lvae_log_qz_workspace_bytes(int64_t M, int64_t N, int D, int with_prod_marg, int64_t* out_bytes);
int lvae_log_qz_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                    int64_t M, int64_t N, int D, float log_norm,
                    int exclude_diag, int64_t diag_offset, int path,
                    float* log_qz, float* log_qz_prod_marg,
                    void* ws, int64_t ws_bytes, void* stream);

/* The same computation as lvae_log_qz_f32 WITHOUT a workspace argument and without any allocation: no
 * counterpart in the reference; for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  The scratch is a static device array inside the
 * library, LVAE_NOALLOC_SCRATCH_BYTES (64 MiB) on every GPU the library runs on, held until the process ends;
 * shapes whose plan needs more (lvae_log_qz_workspace_bytes) are rejected with LVAE_ERR_UNSUPPORTED -- e.g.
 * M = N = 4,096 with prod-marginals or M = N = 32,768 without fit at D = 32.  The call issues no hipMalloc /
 * hipFree.  Its FIRST call on a device looks the array's address up once (hipGetSymbolAddress) and makes the runtime
 * load the kernels: under HIP graph capture make one ordinary, un-captured warm-up call per device first, then capture.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * serialise them. */
#define LVAE_NOALLOC_SCRATCH_BYTES (64u << 20)
int lvae_log_qz_noalloc_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                            int64_t M, int64_t N, int D, float log_norm,
                            int exclude_diag, int64_t diag_offset, int path,
                            float* log_qz, float* log_qz_prod_marg, void* stream);

/* Backward of lvae_log_qz_f32 (the reference differentiates approximate_log_q_z with autograd inside its
 * training objectives, loss_and_optimisation.py:925-1018 via modules/encoder.py:102-107).
 * Inputs: the forward inputs (logvar only), log_norm, the forward result log_qz [M] (needed with g_log_qz),
 * and the upstream gradients g_log_qz [M] (on log_qz) and/or g_prod_marg [M] (on log_qz_prod_marg); either
 * may be NULL.  Outputs (each may be NULL): d_z [M,D], d_mu [N,D], d_logvar [N,D].  Nothing of size M x N is
 * stored for D <= 128 when only g_log_qz is given: the softmax weights are recomputed tile by tile on the fp32
 * matrix cores and contracted with thin feature matrices in registers (pair_grad_mfma.hip; any M, N), behind a
 * device-side conditioning gate with the vector-ALU sweeps (exact differences, D <= 64: weights recomputed in
 * registers) as the other branch.  The weights of a sample are renormalised by their own sum, so the caller's
 * log_qz only has to be good to ~1e-2.  With g_prod_marg the vector-ALU sweeps run.  D > 64 on the vector ALU
 * (g_prod_marg given, or D > 128) reads the joint weights from an explicit N x M matrix of pair densities held
 * in the workspace (M * N <= 2^28, LVAE_ERR_UNSUPPORTED beyond).  One shape has no second branch: 64 < D <= 128 with
 * M * N > 2^28 and g_log_qz alone runs on the matrix cores only; if its conditioning estimate exceeds the gate's
 * limit (2e4: d logvar would be good to 1e-4..2e-4 only) every requested gradient is written as NaN -- a loud
 * result instead of one silently below the parity bar. */
int lvae_log_qz_bwd_workspace_bytes(int64_t M, int64_t N, int D, int prod_marginals, int64_t* out_bytes);
int lvae_log_qz_bwd_f32(const float* z, const float* mu, const float* logvar, int64_t M, int64_t N, int D,
                        float log_norm, const float* log_qz, const float* g_log_qz, const float* g_prod_marg,
                        float* d_z, float* d_mu, float* d_logvar, void* ws, int64_t ws_bytes, void* stream);

/* Backward of lvae_sample_ll_f32 / lvae_sample_ll_elem_f32 (sample_log_likelihood under autograd,
 * loss_and_optimisation.py:79-110): g is the upstream gradient per (b,k) [B,K] (g_per_element == 0) or per
 * element [B,K,D]; mu == NULL selects the prior branch (only d_z).  Outputs may be NULL: d_z [B,K,D],
 * d_mu, d_logvar [B,D]. */
int lvae_sample_ll_bwd_f32(const float* z, const float* mu, const float* logvar, int64_t B, int64_t K, int D,
                           const float* g, int g_per_element, float* d_z, float* d_mu, float* d_logvar,
                           void* stream);

/* Backward of the sampling step of lvae_reparam_iw_f32 (reparameterize, modules/encoder.py:148-157):
 * g_z [B,K,D] -> d_mu, d_logvar [B,D] (either may be NULL). */
int lvae_reparam_bwd_f32(const float* logvar, const float* eps, const float* g_z, int64_t B, int64_t K, int D,
                         float* d_mu, float* d_logvar, void* stream);

/* Backward of lvae_kl_diag_f32 (kl_divergence under autograd, :14-43): g_kl / g_hinge are the upstream
 * gradients of the two outputs, per element [N,D] (g_per_row == 0) or per row [N]; either may be NULL. */
int lvae_kl_diag_bwd_f32(const float* mu, const float* logvar, int64_t N, int D, float lambda,
                         const float* g_kl, const float* g_hinge, int g_per_row, float* d_mu, float* d_logvar,
                         void* stream);

/* Diagnostic (no counterpart in the reference): the launch plan of lvae_log_qz_f32 for these sizes.
 * out[8] = {padded latent dim of the MFMA path (0: unsupported), column blocks per wave, padded M, splits of
 * the posterior axis, 32-row blocks per split, splits of the direct path, 32-row blocks, workspace bytes}.
 * Host only; used by the tests of the grid quantisation. */
int lvae_plan_info(int64_t M, int64_t N, int D, int prod_marginals, int64_t* out);

/* Diagnostic (no counterpart in the reference), host only: the extent in bytes of the workspace layout that an
 * MMD entry point lays out for ONE mode, so that tests can check it against the matching *_workspace_bytes
 * query (which must cover every mode).  kind 0: lvae_kernel_sum_f32(a [na], b [nb], symmetric); kind 1:
 * lvae_mmd_perm_f32(cols [na], pooled [nb], symmetric); kind 2: the pass of lvae_mmd_rbf_f32(x [na], y [nb])
 * named by `symmetric`: 0 = (x, y), 1 = (x, x), 2 = (y, y).  out[3] = {extent, reduced-axis splits, padded
 * kept axis}. */
int lvae_layout_info(int kind, int64_t na, int64_t nb, int D, int symmetric, int64_t* out);

/* Column-sharded variant: the same reduction over the posteriors handed in, returned un-normalised as
 * (row_max, row_sum) with log_qz[i] = row_max[i] + log(row_sum[i]) - log_norm, so that partial results
 * from several devices combine with all-reduce(max) + all-reduce(sum).
 * No counterpart in the reference (it never shards this reduction). */
int lvae_log_qz_partial_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                            int64_t M, int64_t N, int D, int path,
                            float* row_max, float* row_sum,
                            void* ws, int64_t ws_bytes, void* stream);

/* Merge of partial logsumexps over disjoint sets of posteriors (the pairs lvae_log_qz_partial_f32 writes):
 * (max, sum) = (max_a, sum_a) (+) (max_b, sum_b), written to out_max / out_sum (renormalised so that the sum
 * lies in [1, 2)), and / or the finished log_qz[i] = max + log(sum) - log_norm.  max_b / sum_b may be NULL
 * (finalise one partial).  In-place use (out_* aliasing an input pair) is allowed.  No counterpart in the
 * reference; used to overlap the all-gather of posterior parameters with the local block of the pair matrix
 * (distributed.sharded_latent_stats). */
int lvae_lse_combine_f32(const float* max_a, const float* sum_a, const float* max_b, const float* sum_b,
                         int64_t n, float log_norm, float* out_log_qz, float* out_max, float* out_sum,
                         void* stream);

/* The all-reduce form of the same merge (column-sharded posteriors): once the owners have
 * all-reduced their row_max into new_max (>= max_a), out_sum = sum_a * exp(max_a - new_max) re-expresses this
 * owner's sums against it, so that all-reduce(sum) of out_sum followed by lvae_lse_combine_f32(new_max, sum, NULL,
 * NULL, ..) finishes log_qz.  out_sum may alias sum_a.  No counterpart in the reference. */
int lvae_lse_rebase_f32(const float* max_a, const float* sum_a, const float* new_max, int64_t n, float* out_sum,
                        void* stream);

/* q_z_estimate (mutual_information.py:44-52) including quirk Q2: the reduction runs over the SAMPLE
 * axis:  out[j] = LSE_i log N(z_i; mu_j, var_j) - log_norm,  out [N].  Workspace: query with the
 * roles swapped, lvae_log_qz_workspace_bytes(N, M, D, 0, ..). */
int lvae_qz_estimate_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                         int64_t M, int64_t N, int D, float log_norm, float* out,
                         void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-sample densities, reparameterised sampling and the importance-weighted reduction.
 * ------------------------------------------------------------------------------------------- */

/* sample_log_likelihood with reduce_latent_dim=True (loss_and_optimisation.py:79-110):
 * z [B,K,D]; mu, logvar [B,D] (broadcast over K, :86-90).  log_q_z_x[b,k] = log N(z_bk; mu_b, var_b)
 * (needs mu/logvar), log_p_z[b,k] = log N(z_bk; 0, I).  Either output may be NULL. */
int lvae_sample_ll_f32(const float* z, const float* mu, const float* logvar, int64_t B, int64_t K,
                       int D, float* log_q_z_x, float* log_p_z, void* stream);

/* The same with reduce_latent_dim=False: out [B,K,D]; mu == NULL selects the prior branch (:99-101). */
int lvae_sample_ll_elem_f32(const float* z, const float* mu, const float* logvar, int64_t B, int64_t K,
                            int D, float* out, void* stream);

/* Fused EncoderNewsVAE.reparameterize (modules/encoder.py:130-159, noise injected instead of drawn at
 * :156) + the two densities of encode() (:93-98) + iw_log_p_x (loss_and_optimisation.py:783-794; twin
 * utils_evaluation.py:176-180).  eps [B,K,D]; log_p_x_z [B,K] or NULL (treated as 0);
 * outputs (each may be NULL): z [B,K,D], log_q_z_x, log_p_z [B,K], iw_ll [B]. */
int lvae_reparam_iw_f32(const float* mu, const float* logvar, const float* eps, const float* log_p_x_z,
                        int64_t B, int64_t K, int D, float* z, float* log_q_z_x, float* log_p_z,
                        float* iw_ll, void* stream);

/* The hot-path tail of EncoderNewsVAE.encode for ONE training batch (modules/encoder.py:79-107) as one launch:
 * reparameterize (:148-157, noise injected), log q(z|x) and log p(z) per (b, k) (:93-98), approximate_log_q_z(method
 * ="chen", prod_marginals=True) on the FIRST sample of every row with its two means (:102-104; log_norm = log(B *
 * dataset_size), loss_and_optimisation.py:207) and maximum_mean_discrepancy of the first samples against
 * prior_sample [B,D] (:107; NULL: no MMD).  The reference runs this every training step at B = 32, where it is
 * launch-bound: the general entry points need ~40 launches for it.  Direct form throughout (no expansion).
 * Outputs (each may be NULL): z [B,K,D], log_q_z_x / log_p_z [B,K], log_qz / log_qz_prod_marg [B] (per sample),
 * scalars[3] = {mean log_qz, mean log_qz_prod_marg, mmd}.  Envelope (lvae_encode_tail_supported, host only, 1/0):
 * B <= 128, B * D <= 4096, B * K * D <= 2^20; LVAE_ERR_UNSUPPORTED outside. */
int lvae_encode_tail_supported(int64_t B, int64_t K, int D);
int lvae_encode_tail_f32(const float* mu, const float* logvar, const float* eps, const float* prior_sample, int64_t B,
                         int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z, float* log_qz,
                         float* log_qz_prod_marg, float* scalars, void* stream);

/* Backward of lvae_encode_tail_f32 in one launch (what loss.backward() sends through encode(), loss_and_optimisation
 * .py:925-1018): upstream gradients (each may be NULL) g_z [B,K,D], g_log_q_z_x / g_log_p_z [B,K], and one float each
 * in device memory on the two MEANS (g_log_qz_mean, g_prod_marg_mean); the MMD carries no gradient (quirk Q5).
 * Outputs: d_mu, d_logvar [B,D] (everything chained through z = mu + eps * exp(logvar / 2)).  Deterministic. */
int lvae_encode_tail_bwd_f32(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                             float log_norm, const float* g_z, const float* g_log_q_z_x, const float* g_log_p_z,
                             const float* g_log_qz_mean, const float* g_prod_marg_mean, float* d_mu, float* d_logvar,
                             void* stream);

/* The same tail and its backward for ANY training batch (modules/encoder.py:79-107; loss_and_optimisation.py:925-1018
 * for the gradients): up to B = 32 the one-workgroup form above, beyond it B / 4 workgroups (a wave per sample, the
 * other axis streamed through LDS; two launches forward, two backward, deterministic).  Same arguments as
 * lvae_encode_tail_f32 / lvae_encode_tail_bwd_f32, plus `mode` (0: by size, 1: one workgroup, 2: multi-workgroup)
 * and a caller-provided workspace of lvae_encode_tail_workspace_bytes(B, K, D) bytes (256-byte aligned; the backward
 * uses it for the [B, B] softmax weights).  Envelope (lvae_encode_tail_ws_supported, host only, 1/0): D <= 64,
 * B <= 4096, B * K * D <= 2^26, or the one-workgroup envelope; LVAE_ERR_UNSUPPORTED outside, LVAE_ERR_WORKSPACE when
 * the multi-workgroup form is given too little. */
int lvae_encode_tail_ws_supported(int64_t B, int64_t K, int D);
int lvae_encode_tail_workspace_bytes(int64_t B, int64_t K, int D, int64_t* out_bytes);
int lvae_encode_tail_ws_f32(const float* mu, const float* logvar, const float* eps, const float* prior_sample, int64_t B,
                            int64_t K, int D, float log_norm, float* z, float* log_q_z_x, float* log_p_z, float* log_qz,
                            float* log_qz_prod_marg, float* scalars, int mode, void* ws, int64_t ws_bytes, void* stream);
int lvae_encode_tail_bwd_ws_f32(const float* mu, const float* logvar, const float* eps, int64_t B, int64_t K, int D,
                                float log_norm, const float* g_z, const float* g_log_q_z_x, const float* g_log_p_z,
                                const float* g_log_qz_mean, const float* g_prod_marg_mean, float* d_mu, float* d_logvar,
                                int mode, void* ws, int64_t ws_bytes, void* stream);

/* iw_log_p_x alone (loss_and_optimisation.py:783-794): three [B,K] inputs -> [B]. */
int lvae_iw_lse_f32(const float* log_p_x_z, const float* log_p_z, const float* log_q_z_x, int64_t B,
                    int64_t K, float* out, void* stream);

/* Fused tail of the aggregate-posterior statistics for samples z_i drawn from posterior i (M rows):
 * evaluates log p(z_i) and log q(z_i|x_i) on the fly and returns, in device memory,
 *   sums[5] = { sum log_qz, sum log_qz_prod_marg, sum log p(z), sum log q(z|x), M }   (float64)
 * i.e. the four means that assemble_loss turns into MI / TC / dim-KL / marginal-KL
 * (loss_and_optimisation.py:816-824).  log_qz / log_qz_prod_marg [M] may be NULL (sum = 0). */
int lvae_latent_sums_f32(const float* z, const float* mu, const float* logvar, const float* log_qz,
                         const float* log_qz_prod_marg, int64_t M, int D, double* sums, void* stream);

/* lvae_log_qz_f32 (joint density, logvar parameterisation) followed by lvae_latent_sums_f32 as ONE call: log_qz
 * [M] and sums[5] for samples z_i drawn from posterior (mu_local[i], logvar_local[i]) (this rank's rows of the
 * N posteriors), i.e. everything the consumer lines loss_and_optimisation.py:816-824 need for marginal KL and MI.
 * On small problems (N * M <= 2^26) the statistics ride on the kernel that merges the split partials.
 * Workspace: lvae_log_qz_workspace_bytes(M, N, D, 0, ..). */
int lvae_latent_stats_f32(const float* z, const float* mu_all, const float* logvar_all, int64_t M, int64_t N, int D,
                          float log_norm, int path, const float* mu_local, const float* logvar_local, float* log_qz,
                          double* sums, void* ws, int64_t ws_bytes, void* stream);

/* Deterministic reduction to one float64 in device memory: mode 0: scale*sum(x), mode 1:
 * scale*sum(exp(x)).  Used for the `.mean()` the reference applies to per-sample vectors
 * (loss_and_optimisation.py:108,:217-219).  Workspace: LVAE_REDUCE_WS_BYTES. */
#define LVAE_REDUCE_WS_BYTES 8192
int lvae_reduce_f32(const float* x, int64_t n, int mode, double scale, double* out, void* ws,
                    int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MI bounds, explicit pair matrices, KL, KDE, MMD.
 * ------------------------------------------------------------------------------------------- */

/* calc_upper_and_lower_bound_representational_mi (mutual_information.py:218-248) for K diagonal
 * Gaussians N(mu_j, diag(var_j)) and their samples z_j:  out[0] = InfoNCE lower bound
 * (mi_lower_bound :135-173), out[1] = MBU upper bound (mi_upper_bound :176-215, the diagonal masked out
 * of the logsumexp, :207).  `var` is the covariance diagonal as the reference passes it (quirk Q3: the
 * host hands over std).  out: 2 float64 in device memory. */
int lvae_pair_bounds_workspace_bytes(int64_t K, int D, int64_t* out_bytes);
int lvae_pair_bounds_f32(const float* z, const float* mu, const float* var, int64_t K, int D, int path,
                         double* out, void* ws, int64_t ws_bytes, void* stream);

/* log_prob_pairs (mutual_information.py:108-132) / _gaussian_log_likelihood full branch (:41):
 * out[j, i] = log N(z_i; mu_j, diag(spread_j)),  out [N, M] (rows = distributions). */
int lvae_log_prob_pairs_f32(const float* z, const float* mu, const float* spread, int spread_kind,
                            int64_t M, int64_t N, int D, float* out, void* stream);

/* _gaussian_log_likelihood for any mask / axis (mutual_information.py:26-41): over the broadcast shape
 * `shape[ndim]` of (sample, mu, var) the element-wise term log(2 pi var) + (sample - mu)^2 / var (mu == NULL:
 * 0, var == NULL: 1, :32-41) is weighted with `mask` -- already unsqueezed at `dim` by the caller -- and summed
 * over axis `dim`; out = -0.5 * that, contiguous over the shape without `dim`.  shape / st_* are HOST arrays;
 * st_*[k] is the element stride of that operand along axis k of the broadcast shape (0 where it broadcasts).
 * ndim <= 6. */
int lvae_gauss_ll_masked_f32(const float* sample, const float* mu, const float* var, const float* mask, int ndim,
                             const int64_t* shape, const int64_t* st_sample, const int64_t* st_mu,
                             const int64_t* st_var, const int64_t* st_mask, int dim, float* out, void* stream);

/* kl_divergence (loss_and_optimisation.py:14-43): kl = 0.5*(mu^2 + e^logvar - logvar - 1), hinge keeps
 * kl > lambda (:27-30; lambda <= 0: hinge == kl).  Outputs (each may be NULL): element-wise [N,D] and
 * latent-summed [N]. */
int lvae_kl_diag_f32(const float* mu, const float* logvar, int64_t N, int D, float lambda,
                     float* kl_elem, float* hinge_elem, float* kl_row, float* hinge_row, void* stream);

/* Mean over the rows of x [n, c] -> out [c] (the `.mean(dim=0)` of loss_and_optimisation.py:40-41,:108).
 * Workspace: 8*c bytes. */
int lvae_colmean_f32(const float* x, int64_t n, int64_t c, float* out, void* ws, int64_t ws_bytes,
                     void* stream);

/* GaussianKDE.score_samples (utils_latent_analysis_optimisation.py:33-60): X [n,d] fit points, Y [m,d]
 * queries, bandwidth bw; out[j] = log(bw^-d * sum_i exp(logN((X_i - Y_j)/bw)) / n_norm), float32.  n_norm is
 * the size of the FIT set (the reference divides by self.n, :58, also when score_samples is handed a batch X
 * of another size); 0 means n.
 * stabilise == 0: plain sum like the reference (quirk Q6: -inf when every term underflows);
 * stabilise != 0 (no counterpart in the reference; for dataset-scale use): the sum is taken relative to the
 * largest term (one extra pass without exponentials), finite for any query. */
int lvae_kde_workspace_bytes(int64_t n, int64_t m, int64_t* out_bytes);
int lvae_kde_score_f32(const float* X, const float* Y, int64_t n, int64_t n_norm, int64_t m, int d, float bw,
                       int stabilise, float* out, void* ws, int64_t ws_bytes, void* stream);

/* Gaussian-kernel MMD two-sample statistic with a sum of bandwidths:
 *     k(a, b) = sum_k exp(-alphas[k] * (dist_eps + |a - b|^2))
 * unbiased != 0: torch_two_sample.statistics_diff.MMDStatistic.__call__ as used at
 *   loss_and_optimisation.py:617-619 (dist_eps = 1e-5, diagonals excluded, 1/(n(n-1)) weights) --
 *   third-party source absent from the reference tree: PARITY UNPINNED, restated from its algorithm;
 * unbiased == 0: maximum_mean_discrepancy (loss_and_optimisation.py:46-60) with alphas = {1/D^2},
 *   dist_eps = 0: mean(Kxx) + mean(Kyy) - 2 mean(Kxy), diagonals included.
 * x [n1,D], y [n2,D] device; alphas: HOST array of n_alpha >= 1 non-negative floats (n_alpha <= 32).
 * D <= 256: the cross term -2 x.y on the matrix cores (two dimensions per MFMA step, the squared norms enter
 * as biases); 256 < D <= 4096 (wide latents at training-time batch sizes): direct differences on the vector
 * ALU, O(n1 n2 D).  `path`: LVAE_PATH_AUTO / _MFMA / _DIRECT / _MFMA_BF16X3 as described at their definitions
 * (per call: there is no process-wide mode).
 * out: 4 float64 in device memory = {statistic, sum Kxx, sum Kyy, sum Kxy}. */
int lvae_mmd_workspace_bytes(int64_t n1, int64_t n2, int D, int64_t* out_bytes);
int lvae_mmd_rbf_f32(const float* x, const float* y, int64_t n1, int64_t n2, int D, const float* alphas,
                     int n_alpha, float dist_eps, int unbiased, int path, double* out, void* ws, int64_t ws_bytes,
                     void* stream);

/* One block of the kernel-matrix sums above: sum over i < na, j < nb of k(a_i, b_j), same kernel k as
 * lvae_mmd_rbf_f32, optionally skipping the pairs j == i + diag_offset (exclude_diag != 0).  This is the
 * multi-GPU building block of the MMD (a = this rank's rows, b = the gathered sample, diag_offset = the
 * rank's first row; loss_and_optimisation.py:46-60,:617-619 evaluate the whole matrix in one process).
 * out_total: 1 float64 (may be NULL); out_colsum: [na] float64, the per-a_i sums over j (may be NULL).
 * symmetric != 0: a must be the rows diag_offset .. diag_offset+na-1 of b (diag_offset % 32 == 0); only the
 * tiles on and above the diagonal are evaluated and counted twice where due, so that the totals of all owners
 * still add up to the sum over the full symmetric matrix (out_colsum must be NULL). */
int lvae_kernel_sum_workspace_bytes(int64_t na, int64_t nb, int D, int64_t* out_bytes);
int lvae_kernel_sum_f32(const float* a, const float* b, int64_t na, int64_t nb, int D, const float* alphas,
                        int n_alpha, float dist_eps, int exclude_diag, int64_t diag_offset, int symmetric, int path,
                        double* out_total, double* out_colsum, void* ws, int64_t ws_bytes, void* stream);

/* Label-permutation sums for the MMD permutation test (torch_two_sample MMDStatistic.pval; the C4
 * benchmark configuration; never called by the reference itself -- PARITY UNPINNED like the statistic).
 * pooled [n,D]: both samples stacked; cols [nc,D]: the rows col_offset .. col_offset+nc-1 of it that this
 * call (this rank) owns; labels [P][n] uint8 in device memory, labels[p][i] != 0 <=> under labeling p element
 * i belongs to sample 1 (P <= 128 per call).  With K the pooled kernel matrix of lvae_mmd_rbf_f32 restricted to
 * the alpha > 0 bandwidths (an alpha = 0 bandwidth adds 1 to every pair, which cancels exactly in the statistic,
 * and is LEFT OUT of q, lr and T here -- unlike lvae_kernel_sum_f32, which counts it; do not mix the two),
 * diagonal excluded, L_p the indicator
 * vector of labeling p and r = K 1, out receives 2P+1 float64:
 *     out[p]      = sum over owned columns c of L_p[c] * (K L_p)[c]      (q_p; all columns: L_p^T K L_p)
 *     out[P + p]  = sum over owned columns c of L_p[c] * r[c]
 *     out[2P]     = sum over owned columns of r[c]                       (T)
 * Summed over all owners: k11 = q_p, k12 = lr_p - q_p, k22 = T - 2 lr_p + q_p and
 * statistic_p = k11/(n1(n1-1)) + k22/(n2(n2-1)) - 2 k12/(n1 n2).  The kernel values enter the label
 * contraction as two bf16 limbs (relative error <= 2^-17 per entry, unbiased); bandwidths must be integer
 * multiples (<= 16) of the smallest positive one.
 * symmetric != 0 (needs col_offset % 32 == 0): only the tiles on and above the diagonal of the pooled matrix
 * are evaluated for the owned columns (about half the work for col_offset = 0, nc = n); the three outputs then
 * hold this call's share of the same totals -- they still add up over owners to L^T K L, L^T K 1 and 1^T K 1
 * -- but owners of low columns do more work than owners of high ones. */
int lvae_mmd_perm_workspace_bytes(int64_t nc, int64_t n, int D, int64_t* out_bytes);
int lvae_mmd_perm_f32(const float* cols, int64_t nc, int64_t col_offset, const float* pooled, int64_t n, int D,
                      const unsigned char* labels, int P, const float* alphas, int n_alpha, float dist_eps,
                      int symmetric, int path, double* out, void* ws, int64_t ws_bytes, void* stream);

/* The kernel matrix itself (gaussian_kernel, loss_and_optimisation.py:63-76; MMDStatistic ret_matrix):
 * out[i, j] = k(x_i, y_j), out [n1, n2]; alphas on the HOST as above.  Workspace: LVAE_KMAT_WS_BYTES. */
#define LVAE_KMAT_WS_BYTES 4096
int lvae_kernel_matrix_f32(const float* x, const float* y, int64_t n1, int64_t n2, int D,
                           const float* alphas, int n_alpha, float dist_eps, float* out, void* ws,
                           int64_t ws_bytes, void* stream);

/* calc_upper_and_lower_bound_generative_mi (mutual_information.py:251-278): the same two bounds with
 * Categorical critics, P[j, i] = Categorical(probs = probs[j, :]).log_prob(samples[i]) =
 * log(clamp(probs[j, s_i] / sum_v probs[j, v], eps, 1 - eps)), eps = 2^-23 (torch's probs_to_logits).
 * probs [K, V] fp32 (un-normalised rows allowed, as torch allows), samples [K] int64 in [0, V) -- an index
 * outside that range makes both outputs NaN (nothing is read out of bounds).  out[0] = InfoNCE lower bound
 * (:135-173), out[1] = MBU upper bound (:176-215); 2 float64 in device memory.  K >= 2. */
int lvae_categorical_bounds_workspace_bytes(int64_t K, int64_t* out_bytes);
int lvae_categorical_bounds_f32(const float* probs, const int64_t* samples, int64_t K, int64_t V, double* out,
                                void* ws, int64_t ws_bytes, void* stream);

/* Two-sample Kolmogorov-Smirnov statistic of two fp32 vectors: the `statistic` of
 * scipy.stats.ks_2samp(a, b) (two-sided) as called on the importance-weighted log-likelihood vectors at
 * utils_train.py:496-497 (scipy is a third-party dependency of the reference; 1.15.3 in this image).
 * out: 3 float64 in device memory: D = max(D+, D-), D+ = max_v(F_a(v) - F_b(v)), D- = max_v(F_b(v) - F_a(v))
 * clipped to [0, 1].  A NaN in either input makes all three NaN (scipy's nan_policy="propagate").
 * n1, n2 >= 1, n1 + n2 <= 2^30. */
int lvae_ks_2samp_workspace_bytes(int64_t n1, int64_t n2, int64_t* out_bytes);
int lvae_ks_2samp_f32(const float* a, int64_t n1, const float* b, int64_t n2, double* out, void* ws,
                      int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference): HIP events recorded on the caller's stream
 * around the pair kernels of lvae_log_qz_f32 / lvae_log_qz_partial_f32 / lvae_latent_stats_f32, the pair launch of
 * lvae_mmd_perm_f32 and the kernel of lvae_reparam_iw_f32, so a benchmark can time the dominant kernel
 * alone.  lvae_profile_end synchronises on those events.
 * ------------------------------------------------------------------------------------------- */
int lvae_profile_begin(int max_launches);
int lvae_profile_end(double* total_ms, int* launches);

/* Development aid (no counterpart in the reference): the library reads its LVAE_* environment switches (DESIGN.md
 * section 8: forced split counts, path thresholds, ...) once per process; a sweep or a test that changes one inside a
 * running process calls this to have them read again.  Not for use while other threads are inside the library. */
int lvae_dev_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* LVAE_H */
