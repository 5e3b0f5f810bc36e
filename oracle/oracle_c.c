/* CPU oracle, plain C (double precision) -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates approximate_log_q_z (NewsVAE/loss_and_optimisation.py:139-221) and the importance-weighted
 * reduction (loss_and_optimisation.py:783-794 with modules/encoder.py:148-157 and
 * loss_and_optimisation.py:93-101) as straight loops, so that parity tests at sizes NumPy would take
 * minutes for finish in seconds.  Pinned to the reference through tests/test_oracle_golden.py (the
 * same golden vectors as oracle_np.py).  Only tests/, smoke() and bench.py's cpu_baseline leg may load
 * this library; the product never does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define LOG_2PI 1.8378770664093454835606594728112

/* log_dens[i][j][d] = -0.5*(logvar_jd + log 2pi + (z_id - mu_jd)^2 / exp(logvar_jd))      (:96)
 * log_q_z[i]      = logsumexp_j sum_d log_dens - log_norm                                  (:198,:207,:215)
 * log_q_z_pm[i]   = sum_d (logsumexp_j log_dens[i][j][d] - log_norm)                       (:195,:205,:213) */
void oracle_log_qz(const float* z, const float* mu, const float* logvar, int64_t M, int64_t N, int D,
                   double log_norm, double* log_q_z, double* log_q_z_pm) {
    double* inv_var = (double*)malloc(sizeof(double) * (size_t)N * D);
    double* base = (double*)malloc(sizeof(double) * (size_t)N * D);
    for (int64_t k = 0; k < N * D; ++k) {
        inv_var[k] = exp(-(double)logvar[k]);
        base[k] = -0.5 * ((double)logvar[k] + LOG_2PI);
    }
#pragma omp parallel
    {
        double* md = (double*)malloc(sizeof(double) * D);
        double* sd = (double*)malloc(sizeof(double) * D);
#pragma omp for schedule(dynamic, 8)
        for (int64_t i = 0; i < M; ++i) {
            double mj = -INFINITY, sj = 0.0;
            for (int d = 0; d < D; ++d) { md[d] = -INFINITY; sd[d] = 0.0; }
            for (int64_t j = 0; j < N; ++j) {
                double v = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double diff = (double)z[i * D + d] - (double)mu[j * D + d];
                    const double t = base[j * D + d] - 0.5 * diff * diff * inv_var[j * D + d];
                    v += t;
                    if (log_q_z_pm) {
                        if (t > md[d]) { sd[d] = sd[d] * exp(md[d] - t) + 1.0; md[d] = t; }
                        else sd[d] += exp(t - md[d]);
                    }
                }
                if (v > mj) { sj = sj * exp(mj - v) + 1.0; mj = v; }
                else sj += exp(v - mj);
            }
            if (log_q_z) log_q_z[i] = mj + log(sj) - log_norm;
            if (log_q_z_pm) {
                double tot = 0.0;
                for (int d = 0; d < D; ++d) tot += md[d] + log(sd[d]) - log_norm;
                log_q_z_pm[i] = tot;
            }
        }
        free(md);
        free(sd);
    }
    free(inv_var);
    free(base);
}

/* logsumexp_j over j != i + diag_offset of sum_d log_dens[i][j][d], minus log_norm: the masked logsumexp of
 * mi_upper_bound (NewsVAE/mutual_information.py:204-213, the diagonal set to -inf at :207). */
void oracle_log_qz_excl(const float* z, const float* mu, const float* logvar, int64_t M, int64_t N, int D,
                        double log_norm, int64_t diag_offset, double* out) {
    double* inv_var = (double*)malloc(sizeof(double) * (size_t)N * D);
    double* base = (double*)malloc(sizeof(double) * (size_t)N * D);
    for (int64_t k = 0; k < N * D; ++k) {
        inv_var[k] = exp(-(double)logvar[k]);
        base[k] = -0.5 * ((double)logvar[k] + LOG_2PI);
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < M; ++i) {
        double mj = -INFINITY, sj = 0.0;
        for (int64_t j = 0; j < N; ++j) {
            if (j == i + diag_offset) continue;
            double v = 0.0;
            for (int d = 0; d < D; ++d) {
                const double diff = (double)z[i * D + d] - (double)mu[j * D + d];
                v += base[j * D + d] - 0.5 * diff * diff * inv_var[j * D + d];
            }
            if (v > mj) { sj = sj * exp(mj - v) + 1.0; mj = v; }
            else sj += exp(v - mj);
        }
        out[i] = mj + log(sj) - log_norm;
    }
    free(inv_var);
    free(base);
}

/* z = mu + eps*std in float32 like the reference (encoder.py:149,157); densities and the
 * importance-weighted logsumexp in double.  Any output pointer may be NULL. */
void oracle_reparam_iw(const float* mu, const float* logvar, const float* eps, const float* log_p_x_z,
                       int64_t B, int64_t K, int D, float* z_out, double* log_q_z_x, double* log_p_z,
                       double* iw_ll) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        double mx = -INFINITY, s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            double lq = 0.0, lp = 0.0;
            for (int d = 0; d < D; ++d) {
                const float std32 = expf(0.5f * logvar[b * D + d]);
                const float prod = eps[(b * K + k) * D + d] * std32;
                const float zz = mu[b * D + d] + prod;
                if (z_out) z_out[(b * K + k) * D + d] = zz;
                const double diff = (double)zz - (double)mu[b * D + d];
                lq += -0.5 * ((double)logvar[b * D + d] + LOG_2PI + diff * diff / exp((double)logvar[b * D + d]));
                lp += -0.5 * (LOG_2PI + (double)zz * (double)zz);
            }
            if (log_q_z_x) log_q_z_x[b * K + k] = lq;
            if (log_p_z) log_p_z[b * K + k] = lp;
            if (iw_ll) {
                const double w = (log_p_x_z ? (double)log_p_x_z[b * K + k] : 0.0) + lp - lq;
                if (w > mx) { s = s * exp(mx - w) + 1.0; mx = w; }
                else s += exp(w - mx);
            }
        }
        if (iw_ll) iw_ll[b] = mx + log(s) - log((double)K);
    }
}
