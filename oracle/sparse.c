/* sparse.c — restatement of the nalgebra 0.32.4 / nalgebra-sparse 0.9.0 arithmetic that
 * ORC calls on its hot path.  Test infrastructure (see oracle.h).
 *
 * The crates are not vendored under /root/reference (Cargo.lock:326-327, 353-354) and cannot
 * be fetched; what is restated here is their published algorithm:
 *   - CsrMatrix::from(&CooMatrix)   (nalgebra-sparse convert::serial::convert_coo_csr):
 *       rows bucketed in push order, each row sorted by column, duplicates summed.
 *       Call sites: discretization.rs:130,445,471; linear_algebra.rs:62.
 *   - &CsrMatrix * &DVector         (ops::serial::spmm_csr_dense, beta = 0, alpha = 1):
 *       y_i = sum_k a_ik * x_k accumulated from 0.0 in stored (ascending-column) order.
 *       Call sites: linear_algebra.rs:82,97,199,202,250,256,260,283.
 *   - &CsrMatrix * &CsrMatrix       (ops::serial::spmm_csr_prealloc after spmm_csr_pattern):
 *       c_ij = sum_k a_ik * b_kj, k in row-i order of A, j in row-k order of B, from 0.0;
 *       the pattern is the structural product (explicit zeros kept), columns sorted.
 *       Call sites: linear_algebra.rs:84,164.
 *   - CsrMatrix::transpose          (linear_algebra.rs:84,140): CSR of the transpose, sorted.
 *   - DVector::dot / norm           (nalgebra base/blas.rs `dotx`): eight running
 *       accumulators over blocks of 8, folded as res += (acc0+acc4); (acc1+acc5);
 *       (acc2+acc6); (acc3+acc7); then the tail left to right.  norm = sqrt(dot(v,v)).
 *       Call sites: linear_algebra.rs:97,202,253,257,261,265; solver.rs:1226.
 *   - iter().sum::<f64>()           plain left-to-right sum (solver.rs:206-208).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle.h"

OrCsr *or_csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz) {
    OrCsr *a = (OrCsr *)calloc(1, sizeof(OrCsr));
    a->nrows = nrows; a->ncols = ncols; a->nnz = nnz;
    a->row_ptr = (int64_t *)calloc((size_t)nrows + 1, sizeof(int64_t));
    a->col = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    a->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    return a;
}

void or_csr_free(OrCsr *a) {
    if (!a) return;
    free(a->row_ptr); free(a->col); free(a->val); free(a);
}

OrCsr *or_csr_clone(const OrCsr *a) {
    OrCsr *c = or_csr_alloc(a->nrows, a->ncols, a->nnz);
    memcpy(c->row_ptr, a->row_ptr, sizeof(int64_t) * (size_t)(a->nrows + 1));
    memcpy(c->col, a->col, sizeof(int64_t) * (size_t)a->nnz);
    memcpy(c->val, a->val, sizeof(double) * (size_t)a->nnz);
    return c;
}

OrCsr *or_csr_from_arrays(int64_t nrows, int64_t ncols, const int64_t *row_ptr, const int64_t *col, const double *val) {
    OrCsr *c = or_csr_alloc(nrows, ncols, row_ptr[nrows]);
    memcpy(c->row_ptr, row_ptr, sizeof(int64_t) * (size_t)(nrows + 1));
    memcpy(c->col, col, sizeof(int64_t) * (size_t)c->nnz);
    memcpy(c->val, val, sizeof(double) * (size_t)c->nnz);
    return c;
}

/* convert_coo_csr: bucket by row (stable), sort each row by column (stable insertion sort —
 * rows are short), sum duplicates in that order. */
OrCsr *or_csr_from_coo(int64_t nrows, int64_t ncols, int64_t n, const int64_t *ri, const int64_t *ci, const double *v) {
    int64_t *cnt = (int64_t *)calloc((size_t)nrows + 1, sizeof(int64_t));
    for (int64_t k = 0; k < n; k++) cnt[ri[k] + 1]++;
    for (int64_t i = 0; i < nrows; i++) cnt[i + 1] += cnt[i];
    int64_t *tc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    double *tv = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
    memcpy(pos, cnt, sizeof(int64_t) * (size_t)(nrows + 1));
    for (int64_t k = 0; k < n; k++) {
        int64_t q = pos[ri[k]]++;
        tc[q] = ci[k]; tv[q] = v[k];
    }
    /* sort + compress */
    int64_t *row_ptr = (int64_t *)calloc((size_t)nrows + 1, sizeof(int64_t));
    int64_t out = 0;
    for (int64_t i = 0; i < nrows; i++) {
        int64_t b = cnt[i], e = cnt[i + 1];
        for (int64_t q = b + 1; q < e; q++) {
            int64_t c = tc[q]; double x = tv[q]; int64_t r = q - 1;
            while (r >= b && tc[r] > c) { tc[r + 1] = tc[r]; tv[r + 1] = tv[r]; r--; }
            tc[r + 1] = c; tv[r + 1] = x;
        }
        int64_t row_start = out;
        for (int64_t q = b; q < e; q++) {
            if (out > row_start && tc[out - 1] == tc[q]) tv[out - 1] += tv[q];
            else { tc[out] = tc[q]; tv[out] = tv[q]; out++; }
        }
        row_ptr[i + 1] = out;
    }
    OrCsr *a = or_csr_alloc(nrows, ncols, out);
    memcpy(a->row_ptr, row_ptr, sizeof(int64_t) * (size_t)(nrows + 1));
    memcpy(a->col, tc, sizeof(int64_t) * (size_t)out);
    memcpy(a->val, tv, sizeof(double) * (size_t)out);
    free(cnt); free(tc); free(tv); free(pos); free(row_ptr);
    return a;
}

/* CsrMatrix::get_entry — binary search in the row (lib.rs:657-668) */
int64_t or_csr_find(const OrCsr *a, int64_t i, int64_t j) {
    int64_t lo = a->row_ptr[i], hi = a->row_ptr[i + 1] - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a->col[mid] == j) return mid;
        if (a->col[mid] < j) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

void or_spmv(const OrCsr *a, const double *x, double *y) {
    for (int64_t i = 0; i < a->nrows; i++) {
        double dot = 0.;
        for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) dot += a->val[q] * x[a->col[q]];
        y[i] = dot;
    }
}

OrCsr *or_spgemm(const OrCsr *a, const OrCsr *b) {
    int64_t n = a->nrows, m = b->ncols;
    int64_t *marker = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    double *acc = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (int64_t j = 0; j < m; j++) marker[j] = -1;
    int64_t cap = a->nnz * 2 + 16, out = 0;
    int64_t *cc = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    double *cv = (double *)malloc(sizeof(double) * (size_t)cap);
    int64_t *row_ptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) {
        int64_t row_start = out;
        /* pattern */
        for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
            int64_t k = a->col[q];
            for (int64_t t = b->row_ptr[k]; t < b->row_ptr[k + 1]; t++) {
                int64_t j = b->col[t];
                if (marker[j] != i) {
                    marker[j] = i;
                    if (out == cap) {
                        cap *= 2;
                        cc = (int64_t *)realloc(cc, sizeof(int64_t) * (size_t)cap);
                        cv = (double *)realloc(cv, sizeof(double) * (size_t)cap);
                    }
                    cc[out++] = j;
                    acc[j] = 0.;
                }
            }
        }
        /* sort the row's columns */
        for (int64_t q = row_start + 1; q < out; q++) {
            int64_t c = cc[q], r = q - 1;
            while (r >= row_start && cc[r] > c) { cc[r + 1] = cc[r]; r--; }
            cc[r + 1] = c;
        }
        /* values: k ascending in A's row, then B's row order */
        for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
            int64_t k = a->col[q];
            double aik = a->val[q];
            for (int64_t t = b->row_ptr[k]; t < b->row_ptr[k + 1]; t++) acc[b->col[t]] += aik * b->val[t];
        }
        for (int64_t q = row_start; q < out; q++) cv[q] = acc[cc[q]];
        row_ptr[i + 1] = out;
    }
    OrCsr *c = or_csr_alloc(n, m, out);
    memcpy(c->row_ptr, row_ptr, sizeof(int64_t) * (size_t)(n + 1));
    memcpy(c->col, cc, sizeof(int64_t) * (size_t)out);
    memcpy(c->val, cv, sizeof(double) * (size_t)out);
    free(marker); free(acc); free(cc); free(cv); free(row_ptr);
    return c;
}

OrCsr *or_transpose(const OrCsr *a) {
    OrCsr *t = or_csr_alloc(a->ncols, a->nrows, a->nnz);
    for (int64_t q = 0; q < a->nnz; q++) t->row_ptr[a->col[q] + 1]++;
    for (int64_t i = 0; i < a->ncols; i++) t->row_ptr[i + 1] += t->row_ptr[i];
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(a->ncols + 1));
    memcpy(pos, t->row_ptr, sizeof(int64_t) * (size_t)(a->ncols + 1));
    for (int64_t i = 0; i < a->nrows; i++)
        for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
            int64_t d = pos[a->col[q]]++;
            t->col[d] = i; t->val[d] = a->val[q];
        }
    free(pos);
    return t;
}

/* Diagnostic switch (tests only): 1 = pairwise (tree) association instead of nalgebra's.  Used to
 * measure how sensitive a reference run is to the association of its dot products, which is the
 * only freedom a parallel reduction takes. */
static int g_dot_mode = 0;
void or_set_dot_mode(int mode) { g_dot_mode = mode; }
static double dot_pairwise(const double *a, const double *b, int64_t n) {
    if (n <= 8) {
        double s = 0.;
        for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
        return s;
    }
    int64_t h = n / 2;
    return dot_pairwise(a, b, h) + dot_pairwise(a + h, b + h, n - h);
}

double or_dot(const double *a, const double *b, int64_t n) {
    if (g_dot_mode == 1) return dot_pairwise(a, b, n);
    double res = 0.;
    double acc0 = 0., acc1 = 0., acc2 = 0., acc3 = 0., acc4 = 0., acc5 = 0., acc6 = 0., acc7 = 0.;
    int64_t i = 0;
    while (n - i >= 8) {
        acc0 += a[i] * b[i];
        acc1 += a[i + 1] * b[i + 1];
        acc2 += a[i + 2] * b[i + 2];
        acc3 += a[i + 3] * b[i + 3];
        acc4 += a[i + 4] * b[i + 4];
        acc5 += a[i + 5] * b[i + 5];
        acc6 += a[i + 6] * b[i + 6];
        acc7 += a[i + 7] * b[i + 7];
        i += 8;
    }
    res += acc0 + acc4;
    res += acc1 + acc5;
    res += acc2 + acc6;
    res += acc3 + acc7;
    for (int64_t k = i; k < n; k++) res += a[k] * b[k];
    return res;
}

double or_norm(const double *a, int64_t n) { return sqrt(or_dot(a, a, n)); }

double or_sum(const double *a, int64_t n) {
    double s = 0.;
    for (int64_t i = 0; i < n; i++) s += a[i];
    return s;
}
