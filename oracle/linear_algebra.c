/* linear_algebra.c — restatement of src/linear_algebra.rs:1-299 (build_restriction_matrix,
 * multigrid_solve, iterative_solve).  Test infrastructure (see oracle.h).
 * Quirks kept on purpose: SURVEY.md §8a Q4 (nested re-scaling), Q5 (coarse levels re-solve the
 * restricted RHS), Q6 (row-index pairing, weights of 2), Q7 (BiCGSTAB ignores relaxation and
 * threshold; the Jacobi convergence test skips sweeps 0 and 1).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "vec3.h"

#define MULTIGRID_SMOOTHER ORC_SOLVER_BICGSTAB /* linear_algebra.rs:9 */
#define MULTIGRID_COARSENING_LEVELS 3          /* linear_algebra.rs:10 */

static int64_t g_last_jacobi_sweeps = 0;
int64_t or_last_jacobi_sweeps(void) { return g_last_jacobi_sweeps; }

/* linear_algebra.rs:12-63 */
OrCsr *or_build_restriction_matrix(const OrCsr *a, int injection) {
    int64_t nc = a->ncols;
    int64_t n = nc / 2 + nc % 2;
    int64_t cap = 2 * nc + 4, cnt = 0;
    int64_t *ri = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t *ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    double *vv = (double *)malloc(sizeof(double) * (size_t)cap);
#define PUSH(r, c, x) do { ri[cnt] = (r); ci[cnt] = (c); vv[cnt] = (x); cnt++; } while (0)
    if (injection) { /* :16-26 (unused by the reference: :293 hard-codes Strongest) */
        for (int64_t row = 0; row + 1 < n; row++) { PUSH(row, 2 * row, 1.); PUSH(row, 2 * row + 1, 1.); }
        PUSH(n - 1, 2 * (n - 1), 1.);
        if (2 * (n - 1) + 1 < nc) PUSH(n - 1, 2 * (n - 1) + 1, 1.);
    } else { /* :30-60 Strongest */
        char *combined = (char *)calloc((size_t)(nc > 0 ? nc : 1), 1); /* HashSet<usize> is only a membership set */
        for (int64_t i = 0; i < a->nrows; i++) {
            double strongest_coeff = 1.7976931348623157e308; /* Float::MAX */
            int64_t strongest = -1;                           /* usize::MAX */
            for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
                int64_t j = a->col[q];
                if (combined[j] || i == j) continue;
                double coeff = a->val[q]; /* a.get(i, j) */
                if (coeff < strongest_coeff) { strongest_coeff = coeff; strongest = j; }
            }
            if (strongest >= 0) {
                combined[strongest] = 1;
                PUSH(i / 2, i, 1.0);
                PUSH(i / 2, strongest, 1.0);
            }
        }
        free(combined);
    }
#undef PUSH
    OrCsr *r = or_csr_from_coo(n, nc, cnt, ri, ci, vv);
    free(ri); free(ci); free(vv);
    return r;
}

/* linear_algebra.rs:66-141 */
static int multigrid_solve(const OrCsr *a, const double *r, uint64_t level, uint64_t max_levels, int smooth_method,
                           uint64_t smooth_iter_count, double smooth_relaxation, double smooth_threshold,
                           int preconditioner, double *out /* a->ncols */) {
    OrCsr *R = or_build_restriction_matrix(a, 0);          /* :80 */
    int64_t nc = R->nrows;
    double *r_prime = (double *)malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1));
    or_spmv(R, r, r_prime);                                  /* :82 */
    OrCsr *Rt = or_transpose(R);
    OrCsr *RA = or_spgemm(R, a);
    OrCsr *a_prime = or_spgemm(RA, Rt);                      /* :84  (R * a) * R^T */
    or_csr_free(RA);
    double *e_prime = (double *)calloc((size_t)(nc > 0 ? nc : 1), sizeof(double)); /* :86 */
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(nc > 0 ? nc : 1));
    int st = or_iterative_solve(a_prime, r_prime, e_prime, smooth_iter_count, smooth_method, smooth_relaxation,
                                smooth_threshold, preconditioner); /* :87-96 */
    if (st == ORC_OK) {
        or_spmv(a_prime, e_prime, tmp);
        for (int64_t i = 0; i < nc; i++) tmp[i] = r_prime[i] - tmp[i];
        double error_magnitude = or_norm(tmp, nc);           /* :97 */
        if (isnan(error_magnitude)) st = ORC_ERR_MULTIGRID_DIVERGED; /* :103-105 */
    }
    if (st == ORC_OK && level < max_levels && a_prime->nrows > 16) { /* :109 */
        double *corr = (double *)malloc(sizeof(double) * (size_t)nc);
        st = multigrid_solve(a_prime, r_prime, level + 1, max_levels, smooth_method, smooth_iter_count,
                             smooth_relaxation, smooth_threshold, preconditioner, corr); /* :110-121 — r_prime, not the residual (Q5) */
        if (st == ORC_OK) {
            for (int64_t i = 0; i < nc; i++) e_prime[i] += corr[i];
            st = or_iterative_solve(a_prime, r_prime, e_prime, smooth_iter_count, smooth_method, smooth_relaxation,
                                    smooth_threshold / 10., preconditioner); /* :123-132 */
        }
        free(corr);
    }
    if (st == ORC_OK) or_spmv(Rt, e_prime, out);             /* :140 */
    free(r_prime); free(e_prime); free(tmp);
    or_csr_free(R); or_csr_free(Rt); or_csr_free(a_prime);
    return st;
}

/* linear_algebra.rs:144-299 */
int or_iterative_solve(const OrCsr *a, const double *b, double *x, uint64_t iteration_count, int method,
                       double relaxation_factor, double convergence_threshold, int preconditioner) {
    int64_t n = a->nrows;
    OrCsr *a_tmp = NULL;
    double *b_tmp = NULL;
    const OrCsr *ap = a;
    const double *bp = b;
    if (preconditioner == ORC_PRECOND_JACOBI) { /* :159-167 */
        /* p_inv = diagonal_as_csr() with v -> 1/v; a_tmp = p_inv * a; b_tmp = p_inv * b.
         * Row i of the product is (1/a_ii) * a_ij (one product per entry); rows whose diagonal is
         * not stored come out empty and their b entry becomes 0. */
        a_tmp = or_csr_alloc(a->nrows, a->ncols, a->nnz);
        b_tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        int64_t out = 0;
        for (int64_t i = 0; i < n; i++) {
            int64_t d = (i < a->ncols) ? or_csr_find(a, i, i) : -1;
            if (d >= 0) {
                double pinv = 1. / a->val[d];
                for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
                    a_tmp->col[out] = a->col[q];
                    a_tmp->val[out] = 0. + pinv * a->val[q];
                    out++;
                }
                b_tmp[i] = 0. + pinv * b[i];
            } else {
                b_tmp[i] = 0.;
            }
            a_tmp->row_ptr[i + 1] = out;
        }
        a_tmp->nnz = out;
        ap = a_tmp; bp = b_tmp;
    } else if (preconditioner != ORC_PRECOND_NONE) {
        return ORC_ERR_BAD_ARGUMENT;
    }
    int st = ORC_OK;
    double initial_residual = 0.; /* :170 */
    switch (method) {
    case ORC_SOLVER_JACOBI: { /* :172-218 */
        OrCsr *a_prime = or_csr_clone(ap);
        double *b_prime = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        for (int64_t i = 0; i < n && st == ORC_OK; i++) {
            int64_t d = or_csr_find(ap, i, i);
            if (d < 0) { st = ORC_ERR_STRUCTURAL_ZERO; break; } /* a_preconditioned.get(i,i) */
            double aii = ap->val[d];
            for (int64_t q = ap->row_ptr[i]; q < ap->row_ptr[i + 1]; q++)
                a_prime->val[q] = (ap->col[q] == i) ? 0. : ap->val[q] / aii; /* :174-180 */
            b_prime[i] = bp[i] / aii;                                           /* :181-187 */
        }
        g_last_jacobi_sweeps = 0;
        for (uint64_t iter_num = 0; iter_num < iteration_count && st == ORC_OK; iter_num++) {
            for (int64_t i = 0; i < n; i++) if (isnan(x[i])) { st = ORC_ERR_JACOBI_NAN; break; } /* :192-196 */
            if (st != ORC_OK) break;
            or_spmv(a_prime, x, tmp);
            for (int64_t i = 0; i < n; i++) /* :199-200 */
                x[i] = relaxation_factor * (b_prime[i] - tmp[i]) + x[i] * (1. - relaxation_factor);
            or_spmv(ap, x, tmp);
            for (int64_t i = 0; i < n; i++) tmp[i] = bp[i] - tmp[i];
            double r = or_norm(tmp, n); /* :202 */
            double max_abs_val = 0.;    /* :203-207 max_by(|a|.total_cmp(|b|)): the last maximal element wins */
            if (n > 0) {
                double best = x[0];
                for (int64_t i = 1; i < n; i++) if (f64_total_cmp(fabs(x[i]), fabs(best)) >= 0) best = x[i];
                max_abs_val = fabs(best);
            }
            g_last_jacobi_sweeps++;
            if (iter_num == 1) initial_residual = r;                              /* :208-209 */
            else if (r / initial_residual < convergence_threshold) break;        /* :210-213 */
            if (max_abs_val > 1e10) { st = ORC_ERR_JACOBI_TOO_LARGE; break; }     /* :214-216 */
        }
        or_csr_free(a_prime); free(b_prime); free(tmp);
        break;
    }
    case ORC_SOLVER_GAUSS_SEIDEL: { /* :219-246 */
        /* Dense row scan through get(i, j): the first structural zero panics (lib.rs:664-666);
         * a fully dense matrix would sweep and then hit panic!("Gauss-Seidel out for maintenance"). */
        st = ORC_ERR_GS_MAINTENANCE;
        for (uint64_t it = 0; it < iteration_count && st == ORC_ERR_GS_MAINTENANCE; it++) {
            for (int64_t i = 0; i < n; i++) {
                double sum = 0.;
                int bad = 0;
                for (int64_t j = 0; j < n; j++) {
                    if (i != j) {
                        int64_t q = or_csr_find(ap, i, j);
                        if (q < 0) { bad = 1; break; }
                        sum += ap->val[q] * x[j];
                    } else sum += 0.;
                }
                if (bad) { st = ORC_ERR_STRUCTURAL_ZERO; break; }
                int64_t d = or_csr_find(ap, i, i);
                if (d < 0) { st = ORC_ERR_STRUCTURAL_ZERO; break; }
                x[i] = x[i] * (1. - relaxation_factor) + relaxation_factor * (bp[i] - sum) / ap->val[d];
                if (isnan(x[i])) { st = ORC_ERR_SOLUTION_DIVERGED; break; } /* :240-242 */
            }
        }
        break;
    }
    case ORC_SOLVER_BICGSTAB: { /* :247-269 */
        size_t nn = (size_t)(n > 0 ? n : 1);
        double *r = (double *)malloc(sizeof(double) * nn), *r_hat_0 = (double *)malloc(sizeof(double) * nn);
        double *pv = (double *)malloc(sizeof(double) * nn), *nu = (double *)malloc(sizeof(double) * nn);
        double *h = (double *)malloc(sizeof(double) * nn), *s = (double *)malloc(sizeof(double) * nn);
        double *t = (double *)malloc(sizeof(double) * nn);
        or_spmv(ap, x, r);
        for (int64_t i = 0; i < n; i++) { r[i] = bp[i] - r[i]; r_hat_0[i] = 1.; } /* :250-252 */
        double rho = or_dot(r, r_hat_0, n);                                       /* :253 */
        memcpy(pv, r, sizeof(double) * (size_t)n);                                /* :254 */
        for (uint64_t it = 0; it < iteration_count; it++) {
            or_spmv(ap, pv, nu);                                                  /* :256 */
            double alpha = rho / or_dot(r_hat_0, nu, n);                          /* :257 */
            for (int64_t i = 0; i < n; i++) h[i] = x[i] + alpha * pv[i];          /* :258 */
            for (int64_t i = 0; i < n; i++) s[i] = r[i] - alpha * nu[i];          /* :259 */
            or_spmv(ap, s, t);                                                    /* :260 */
            double omega = or_dot(t, s, n) / or_dot(t, t, n);                     /* :261 */
            for (int64_t i = 0; i < n; i++) x[i] = h[i] + omega * s[i];           /* :262 */
            for (int64_t i = 0; i < n; i++) r[i] = s[i] - omega * t[i];           /* :263 */
            double rho_prev = rho;                                                /* :264 */
            rho = or_dot(r_hat_0, r, n);                                          /* :265 */
            double beta = rho / rho_prev * alpha / omega;                         /* :266 */
            for (int64_t i = 0; i < n; i++) pv[i] = r[i] + beta * (pv[i] - omega * nu[i]); /* :267 */
        }
        free(r); free(r_hat_0); free(pv); free(nu); free(h); free(s); free(t);
        break;
    }
    case ORC_SOLVER_MULTIGRID: { /* :270-296 */
        st = or_iterative_solve(ap, bp, x, iteration_count, MULTIGRID_SMOOTHER, relaxation_factor,
                                convergence_threshold, preconditioner); /* :273-282 — scales the scaled system again (Q4) */
        if (st == ORC_OK) {
            size_t nn = (size_t)(n > 0 ? n : 1);
            double *r = (double *)malloc(sizeof(double) * nn), *corr = (double *)malloc(sizeof(double) * nn);
            or_spmv(ap, x, r);
            for (int64_t i = 0; i < n; i++) r[i] = bp[i] - r[i]; /* :283 */
            st = multigrid_solve(ap, r, 1, MULTIGRID_COARSENING_LEVELS, MULTIGRID_SMOOTHER, iteration_count,
                                 relaxation_factor, convergence_threshold, preconditioner, corr); /* :284-295 */
            if (st == ORC_OK) for (int64_t i = 0; i < n; i++) x[i] += corr[i];
            free(r); free(corr);
        }
        break;
    }
    default:
        st = ORC_ERR_UNSUPPORTED_SOLVER; /* :297 */
    }
    or_csr_free(a_tmp);
    free(b_tmp);
    return st;
}
