// api_linalg.cpp — C ABI for linear_algebra::iterative_solve (linear_algebra.rs:144-153).
#include <algorithm>

#include "linalg.hpp"

namespace orc {
int amg_debug_coarsen(const MatView &A, Arena &arena, std::vector<int> &choice_h, std::vector<int64_t> &row_ptr_h,
                      std::vector<int64_t> &col_h, std::vector<double> &val_h, int *rounds, const double *x_h = nullptr, double *y_h = nullptr,
                      int scaled = 0, int *mirror_out = nullptr);
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

// Three systems on one pattern (the u, v, w momentum matrices): values, b and x per system, contiguous, in CSR (ORC) order.
int orc_iterative_solve3(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *const values[3], const double *const b[3],
                         double *const solution_vectors[3], uint64_t iteration_count, int method, double relaxation_factor,
                         double convergence_threshold, int preconditioner, int status_out[3]) {
    using namespace orc;
    ORC_TRY(ensure_init());
    if (n < 0 || !row_ptr || (!col_idx && n > 0) || !values || !b || !solution_vectors) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    if (method != ORC_SOLVER_BICGSTAB && method != ORC_SOLVER_MULTIGRID)
        return set_error(ORC_ERR_BAD_ARGUMENT, "orc_iterative_solve3: BiCGSTAB and Multigrid arms only (method %d)", method);
    SellMatrix pat;
    ORC_TRY(sell_from_csr_host(n, n, row_ptr, col_idx, pat));
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    DevBuf<double> csr_vals, vals[3], db[3], dx[3];
    MatView3 A3;
    A3.P = pat.dev();
    for (int k = 0; k < 3; ++k) {
        ORC_TRY(csr_vals.upload(values[k], (size_t)pat.nnz));
        ORC_TRY(vals[k].alloc((size_t)std::max<int64_t>(pat.padded, 1)));
        ORC_TRY(sell_import_values(pat, csr_vals.p, vals[k].p));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
        ORC_TRY(db[k].alloc(nn));
        ORC_TRY(dx[k].alloc(nn));
        ORC_TRY(db[k].upload(b[k], (size_t)n));
        ORC_TRY(dx[k].upload(solution_vectors[k], (size_t)n));
        A3.val[k] = vals[k].p;
    }
    Arena arena;
    int st = ORC_OK;
    int st3[3] = {ORC_OK, ORC_OK, ORC_OK};
    if (!triple_supported()) {
        // reference reduction order (one wavefront per dot product) or a multi-rank context: one solve per system, same results
        for (int k = 0; k < 3; ++k) {
            MatView A;
            A.P = pat.dev();
            A.val = vals[k].p;
            A.symmetric = pat.symmetric;
            SolveStats stats;
            st3[k] = iterative_solve_dev(A, db[k].p, dx[k].p, iteration_count, method, relaxation_factor, convergence_threshold, preconditioner, arena, &stats);
            // a solver VERDICT (the reference's "Multigrid diverged" panic) is per system; anything else — a HIP error, a failed
            // arena allocation — is the call's own failure and must not disappear behind ORC_OK
            if (st == ORC_OK && st3[k] != ORC_OK && st3[k] != ORC_ERR_MULTIGRID_DIVERGED) st = st3[k];
        }
    } else if (method == ORC_SOLVER_BICGSTAB) {
        double *b3, *x3;
        ORC_TRY(arena.alloc(3 * nn, &b3));
        ORC_TRY(arena.alloc(3 * nn, &x3));
        ORC_TRY(interleave3_dev(db[0].p, db[1].p, db[2].p, b3, n));
        ORC_TRY(interleave3_dev(dx[0].p, dx[1].p, dx[2].p, x3, n));
        st = bicgstab3_dev(A3, b3, x3, iteration_count, preconditioner, arena);
        if (st == ORC_OK) st = deinterleave3_dev(x3, dx[0].p, dx[1].p, dx[2].p, n);
    } else {
        TripleLane lanes[3];
        Arena hier[3], vec[3], scratch[3];  // (scratch: the set-ups' transient storage, as in the solver's lanes — and what the shared Galerkin pass needs)
        SiblingPairing sibling;
        hipStream_t streams[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 6 && st == ORC_OK; ++k)
            st = stream_create(&streams[k], kPlainStream, k, "solve3");
        for (int k = 0; k < 3; ++k) {
            lanes[k].setup_stream = streams[2 * k]; lanes[k].solve_stream = streams[2 * k + 1];
            lanes[k].hier_arena = &hier[k]; lanes[k].vec_arena = &vec[k]; lanes[k].scratch_arena = &scratch[k];
            lanes[k].symmetric = pat.symmetric;
        }
        const double *bb[3] = {db[0].p, db[1].p, db[2].p};
        double *xx[3] = {dx[0].p, dx[1].p, dx[2].p};
        if (st == ORC_OK)
            st = multigrid_arm3_dev(A3, bb, xx, iteration_count, relaxation_factor, convergence_threshold, preconditioner, arena, lanes, &sibling, st3);
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 6; ++k)
            if (streams[k]) stream_destroy(streams[k]);
    }
    for (int k = 0; k < 3; ++k) {
        const int st2 = dx[k].download(solution_vectors[k], (size_t)n);
        if (st == ORC_OK) st = st2;
        if (status_out) status_out[k] = st3[k];
    }
    if (!status_out && st == ORC_OK)  // nobody to tell per system: the first verdict is the call's
        for (int k = 0; k < 3 && st == ORC_OK; ++k) st = st3[k];
    return st;
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

int orc_debug_amg_coarse_product(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, int scaled, const double *x,
                                 double *y, int *has_window_mirror) {
    using namespace orc;
    ORC_TRY(ensure_init());
    if (n < 1 || !row_ptr || !col_idx || !values || !x || !y) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
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
    return amg_debug_coarsen(A, arena, choice, rp, ci, v, nullptr, x, y, scaled, has_window_mirror);
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
