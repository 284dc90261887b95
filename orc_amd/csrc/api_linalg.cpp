// api_linalg.cpp — C ABI for linear_algebra::iterative_solve (linear_algebra.rs:144-153).
#include <algorithm>

#include "linalg.hpp"

namespace orc {
int amg_debug_coarsen(const MatView &A, Arena &arena, std::vector<int> &choice_h, std::vector<int64_t> &row_ptr_h,
                      std::vector<int64_t> &col_h, std::vector<double> &val_h, int *rounds);
int gs_debug_coloring(const SellDev &P, std::vector<int> &colors, int *n_colors);
static SolveStats g_last_stats;
SolveStats &last_stats() { return g_last_stats; }
}  // namespace orc

extern "C" {

int orc_iterative_solve(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, const double *b,
                        double *solution_vector, uint64_t iteration_count, int method, double relaxation_factor,
                        double convergence_threshold, int preconditioner) {
    using namespace orc;
    ORC_TRY(ensure_init());
    if (n < 0 || !row_ptr || (!col_idx && n > 0) || (!values && n > 0) || !b || !solution_vector)
        return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SellMatrix pat;
    ORC_TRY(sell_from_csr_host(n, n, row_ptr, col_idx, pat));
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    DevBuf<double> csr_vals, vals, db, dx;
    ORC_TRY(csr_vals.upload(values, (size_t)pat.nnz));
    ORC_TRY(vals.alloc((size_t)std::max<int64_t>(pat.padded, 1)));
    ORC_TRY(sell_import_values(pat, csr_vals.p, vals.p));
    ORC_TRY(db.alloc(nn));
    ORC_TRY(dx.alloc(nn));
    ORC_TRY(db.upload(b, (size_t)n));
    ORC_TRY(dx.upload(solution_vector, (size_t)n));
    MatView A;
    A.P = pat.dev();
    A.val = vals.p;
    A.symmetric = pat.symmetric;
    Arena arena;
    g_last_stats = SolveStats();
    int st = iterative_solve_dev(A, db.p, dx.p, iteration_count, method, relaxation_factor, convergence_threshold, preconditioner,
                                 arena, &g_last_stats);
    // the reference mutates solution_vector in place up to the panic; hand back what was computed
    int st2 = dx.download(solution_vector, (size_t)n);
    return st != ORC_OK ? st : st2;
}

int orc_csr_spmv(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, const double *x, double *y,
                 int reps, double *avg_ms) {
    using namespace orc;
    ORC_TRY(ensure_init());
    if (n < 0 || !row_ptr || !x || !y) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SellMatrix pat;
    ORC_TRY(sell_from_csr_host(n, n, row_ptr, col_idx, pat));
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    DevBuf<double> csr_vals, vals, dx, dy;
    ORC_TRY(csr_vals.upload(values, (size_t)pat.nnz));
    ORC_TRY(vals.alloc((size_t)std::max<int64_t>(pat.padded, 1)));
    ORC_TRY(sell_import_values(pat, csr_vals.p, vals.p));
    ORC_TRY(dx.alloc(nn));
    ORC_TRY(dy.alloc(nn));
    ORC_TRY(dx.upload(x, (size_t)n));
    MatView A;
    A.P = pat.dev();
    A.val = vals.p;
    if (reps < 1) reps = 1;
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    ORC_TRY(spmv_dev(A, dx.p, dy.p));
    ORC_HIP(hipEventRecord(e0, ctx().stream));
    for (int i = 1; i < reps; ++i) ORC_TRY(spmv_dev(A, dx.p, dy.p));
    ORC_HIP(hipEventRecord(e1, ctx().stream));
    ORC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    ORC_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = reps > 1 ? (double)ms / (reps - 1) : 0.;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return dy.download(y, (size_t)n);
}

int orc_amg_coarsen(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, int64_t *partner,
                    int64_t *out_n_coarse, int64_t *out_nnz, int64_t *out_row_ptr, int64_t *out_col, double *out_val, int *rounds) {
    using namespace orc;
    ORC_TRY(ensure_init());
    SellMatrix pat;
    ORC_TRY(sell_from_csr_host(n, n, row_ptr, col_idx, pat));
    DevBuf<double> csr_vals, vals;
    ORC_TRY(csr_vals.upload(values, (size_t)pat.nnz));
    ORC_TRY(vals.alloc((size_t)std::max<int64_t>(pat.padded, 1)));
    ORC_TRY(sell_import_values(pat, csr_vals.p, vals.p));
    MatView A;
    A.P = pat.dev();
    A.val = vals.p;
    A.symmetric = pat.symmetric;
    Arena arena;
    std::vector<int> choice;
    std::vector<int64_t> rp, ci;
    std::vector<double> v;
    ORC_TRY(amg_debug_coarsen(A, arena, choice, rp, ci, v, rounds));
    if (partner) for (int64_t i = 0; i < n; ++i) partner[i] = choice[(size_t)i];
    if (out_n_coarse) *out_n_coarse = (int64_t)rp.size() - 1;
    if (out_nnz) *out_nnz = (int64_t)ci.size();
    if (out_col) {
        std::copy(rp.begin(), rp.end(), out_row_ptr);
        std::copy(ci.begin(), ci.end(), out_col);
        std::copy(v.begin(), v.end(), out_val);
    }
    return ORC_OK;
}

int orc_debug_coloring(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, int32_t *colors, int32_t *n_colors) {
    using namespace orc;
    ORC_TRY(ensure_init());
    SellMatrix pat;
    ORC_TRY(sell_from_csr_host(n, n, row_ptr, col_idx, pat));
    std::vector<int> c;
    int nc = 0;
    ORC_TRY(gs_debug_coloring(pat.dev(), c, &nc));
    for (int64_t i = 0; i < n; ++i) colors[i] = c[(size_t)i];
    if (n_colors) *n_colors = nc;
    return ORC_OK;
}

int64_t orc_last_jacobi_sweeps(void) { return orc::g_last_stats.jacobi_sweeps; }

}  // extern "C"
