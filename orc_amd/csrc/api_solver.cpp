// api_solver.cpp — C ABI for mesh::Mesh, discretization::* and solver::solve_steady.
#include <algorithm>
#include <chrono>
#include <memory>

#include "assembly.hpp"

using namespace orc;

namespace orc {
void gs_forget_pattern(const void *col_ptr);  // gs.hip
}

namespace {

int upload_csr_values(OrcMesh &m, const double *host_vals, DevBuf<double> &sell, DevBuf<double> &tmp) {
    ORC_TRY(tmp.upload(host_vals, (size_t)m.pat.nnz));
    return sell_import_values(m.pat, tmp.p, sell.p);
}

int download_csr_values(OrcMesh &m, const DevBuf<double> &sell, double *host_vals, DevBuf<double> &tmp) {
    ORC_TRY(tmp.ensure((size_t)m.pat.nnz));
    ORC_TRY(sell_export_values(m.pat, sell.p, tmp.p));
    return tmp.download(host_vals, (size_t)m.pat.nnz);
}

}  // namespace

namespace orc {
int bench_gs_sweep_dev(const MatView &A, const double *b, double *x, int reps, Arena &arena, float *ms_per_sweep, int *n_colors);  // gs.hip
int bench_gs_sweep0_dev(const MatView A[3], const double *const b[3], int reps, Arena &arena, float ms[2], int *n_colors);         // gs.hip
}

extern "C" {

OrcMesh *orc_mesh_create(int64_t n_cells, int64_t n_faces, int32_t n_zones, const int64_t *face_c0, const int64_t *face_c1,
                         const int32_t *face_zone, const double *face_area, const double *face_normal, const double *face_centroid,
                         const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                         const int64_t *cell_faces, const int32_t *zone_type, const double *zone_scalar, const double *zone_vector,
                         int *status) {
    int st = ensure_init();
    OrcMesh *m = nullptr;
    if (st == ORC_OK) {
        m = new OrcMesh();
        st = mesh_upload(*m, n_cells, n_cells, n_faces, n_zones, face_c0, face_c1, face_zone, face_area, face_normal, face_centroid,
                         cell_centroid, cell_volume, cell_face_ptr, cell_faces, zone_type, zone_scalar, zone_vector);
        if (st != ORC_OK) { delete m; m = nullptr; }
    }
    if (status) *status = st;
    return m;
}

OrcMesh *orc_mesh_create_partitioned(int64_t n_owned, int64_t n_cells, int64_t n_cells_global, int64_t n_faces, int32_t n_zones,
                                     const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone, const double *face_area,
                                     const double *face_normal, const double *face_centroid, const double *cell_centroid,
                                     const double *cell_volume, const int64_t *cell_face_ptr, const int64_t *cell_faces,
                                     const int32_t *zone_type, const double *zone_scalar, const double *zone_vector, int32_t n_peers,
                                     const int32_t *peers, const int64_t *send_ptr, const int64_t *send_idx, const int64_t *recv_ptr,
                                     int *status) {
    int st = ensure_init();
    OrcMesh *m = nullptr;
    if (st == ORC_OK) {
        m = new OrcMesh();
        m->n_global = n_cells_global;
        st = mesh_upload(*m, n_owned, n_cells, n_faces, n_zones, face_c0, face_c1, face_zone, face_area, face_normal, face_centroid,
                         cell_centroid, cell_volume, cell_face_ptr, cell_faces, zone_type, zone_scalar, zone_vector);
        if (st == ORC_OK && n_peers > 0) {
            HaloPlan &H = m->halo;
            H.n_own = n_owned;
            H.n_ghost = n_cells - n_owned;
            H.n_send = send_ptr[n_peers];
            if (recv_ptr[n_peers] != H.n_ghost) st = set_error(ORC_ERR_BAD_ARGUMENT, "ghost blocks (%lld) do not cover the ghost cells (%lld)",
                                                                (long long)recv_ptr[n_peers], (long long)H.n_ghost);
            std::vector<int32_t> idx((size_t)H.n_send);
            for (int64_t i = 0; i < H.n_send && st == ORC_OK; ++i) {
                if (send_idx[i] < 0 || send_idx[i] >= n_owned) st = set_error(ORC_ERR_BAD_ARGUMENT, "send index out of the owned range");
                idx[(size_t)i] = (int32_t)send_idx[i];
            }
            for (int q = 0; q < n_peers; ++q) {
                H.peers.push_back(peers[q]);
                H.send_off.push_back(send_ptr[q]); H.send_cnt.push_back(send_ptr[q + 1] - send_ptr[q]);
                H.recv_off.push_back(recv_ptr[q]); H.recv_cnt.push_back(recv_ptr[q + 1] - recv_ptr[q]);
            }
            if (st == ORC_OK) st = H.send_idx.upload(idx.data(), idx.size());
            if (st == ORC_OK) {  // the longest run of 64-row slices whose rows have no ghost neighbour
                const int64_t n_slices = (n_owned + 63) / 64;
                std::vector<unsigned char> touches((size_t)n_slices, 0);
                for (int64_t f = 0; f < n_faces; ++f) {
                    const int64_t a = face_c0[f], b = face_c1[f];
                    if (b < 0) continue;
                    if (a < n_owned && b >= n_owned) touches[(size_t)(a >> 6)] = 1;
                    if (b < n_owned && a >= n_owned) touches[(size_t)(b >> 6)] = 1;
                }
                int64_t best_lo = 0, best_len = 0, run_lo = 0;
                for (int64_t sl = 0; sl <= n_slices; ++sl) {
                    if (sl == n_slices || touches[(size_t)sl]) {
                        if (sl - run_lo > best_len) { best_len = sl - run_lo; best_lo = run_lo; }
                        run_lo = sl + 1;
                    }
                }
                H.interior_lo = (int32_t)best_lo;
                H.interior_hi = (int32_t)(best_lo + best_len);
            }
        }
        if (st != ORC_OK) { delete m; m = nullptr; }
    }
    if (status) *status = st;
    return m;
}

// The whole mesh renumbered by `ordering` (orc_mesh_partition with one rank): rows sorted by RCM for coalesced reads when
// the generator's numbering is arbitrary.  Fields still go in and out in ORC order.
OrcMesh *orc_mesh_create_reordered(int64_t n_cells, int64_t n_faces, int32_t n_zones, const int64_t *face_c0, const int64_t *face_c1,
                                   const int32_t *face_zone, const double *face_area, const double *face_normal, const double *face_centroid,
                                   const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                                   const int64_t *cell_faces, const int32_t *zone_type, const double *zone_scalar, const double *zone_vector,
                                   int32_t ordering, int *status) {
    int st = ORC_OK;
    OrcPartition *P = orc_mesh_partition(n_cells, n_faces, face_c0, face_c1, face_zone, face_area, face_normal, face_centroid, cell_centroid,
                                         cell_volume, cell_face_ptr, cell_faces, 1, 0, ordering, &st);
    OrcMesh *m = nullptr;
    if (P) {
        m = orc_partition_upload(P, n_zones, zone_type, zone_scalar, zone_vector, &st);
        if (m) {
            m->h_global_ids.resize((size_t)n_cells);
            st = orc_partition_arrays(P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, m->h_global_ids.data(),
                                      nullptr, nullptr, nullptr, nullptr, nullptr);
            bool identity = true;
            for (int64_t c = 0; c < n_cells && identity; ++c) identity = m->h_global_ids[(size_t)c] == c;
            if (identity) m->h_global_ids.clear();
        }
        orc_partition_destroy(P);
    }
    if (status) *status = st;
    return m;
}

int orc_mesh_cell_order(const OrcMesh *m, int64_t *global_ids) {
    if (!m || !global_ids) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    for (int64_t c = 0; c < m->n_cells; ++c) global_ids[c] = m->h_global_ids.empty() ? c : m->h_global_ids[(size_t)c];
    return ORC_OK;
}

int orc_mesh_update_zones(OrcMesh *m, const int32_t *zone_type, const double *zone_scalar, const double *zone_vector) {
    if (!m) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    ORC_TRY(m->ztype.upload(zone_type, (size_t)m->n_zones));
    ORC_TRY(m->zscal.upload(zone_scalar, (size_t)m->n_zones));
    ORC_TRY(m->zvec.upload(zone_vector, (size_t)3 * m->n_zones));
    return ORC_OK;
}

void orc_mesh_destroy(OrcMesh *m) {
    if (m) orc::gs_forget_pattern((const void *)m->pat.col.p);
    delete m;
}
int64_t orc_mesh_n_cells(const OrcMesh *m) { return m ? m->n_cells : 0; }
int64_t orc_mesh_n_owned(const OrcMesh *m) { return m ? m->n_own : 0; }
int64_t orc_mesh_nnz(const OrcMesh *m) { return m ? m->pat.nnz : 0; }

int orc_mesh_matrix_pattern(const OrcMesh *m, int64_t *row_ptr, int64_t *col_idx) {
    if (!m) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    std::copy(m->h_row_ptr.begin(), m->h_row_ptr.end(), row_ptr);
    std::copy(m->h_col.begin(), m->h_col.end(), col_idx);
    return ORC_OK;
}

// ---------------------------------------------------------------- device-resident solver
OrcSolver *orc_solver_create(OrcMesh *m, const OrcSettings *settings, double rho, double mu, int *status) {
    int st = ORC_OK;
    OrcSolver *s = nullptr;
    if (!m || !settings) st = set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    if (st == ORC_OK) {
        s = new OrcSolver();
        st = solver_init(s->st, m, settings, rho, mu);
        if (st != ORC_OK) { delete s; s = nullptr; }
    }
    if (status) *status = st;
    return s;
}

void orc_solver_destroy(OrcSolver *s) { delete s; }

int orc_solver_set_fields(OrcSolver *s, const double *u, const double *v, const double *w, const double *p) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    const size_t n = (size_t)s->st.n;
    const std::vector<int64_t> &g = s->st.mesh->h_global_ids;  // reordered mesh: internal cell c holds ORC cell g[c]
    const double *src[4] = {u, v, w, p};
    DevBuf<double> *dst[4] = {&s->st.u, &s->st.v, &s->st.w, &s->st.p};
    std::vector<double> tmp;
    for (int k = 0; k < 4; ++k) {
        if (g.empty()) { ORC_TRY(dst[k]->upload(src[k], n)); continue; }
        tmp.resize(n);
        for (size_t c = 0; c < n; ++c) tmp[c] = src[k][g[c]];
        ORC_TRY(dst[k]->upload(tmp.data(), n));
    }
    return ORC_OK;
}

int orc_solver_get_fields(OrcSolver *s, double *u, double *v, double *w, double *p) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    const size_t n = (size_t)s->st.n;
    const std::vector<int64_t> &g = s->st.mesh->h_global_ids;
    double *dst[4] = {u, v, w, p};
    DevBuf<double> *src[4] = {&s->st.u, &s->st.v, &s->st.w, &s->st.p};
    std::vector<double> tmp;
    for (int k = 0; k < 4; ++k) {
        if (g.empty()) { ORC_TRY(src[k]->download(dst[k], n)); continue; }
        tmp.resize(n);
        ORC_TRY(src[k]->download(tmp.data(), n));
        for (size_t c = 0; c < n; ++c) dst[k][g[c]] = tmp[c];
    }
    return ORC_OK;
}

namespace {
// one device vector of the mesh's internal cell order -> the caller's array in ORC cell order (orc_mesh_create_reordered)
int download_in_orc_order(const OrcMesh &m, const DevBuf<double> &src, double *dst, size_t n) {
    const std::vector<int64_t> &g = m.h_global_ids;
    if (g.empty()) return src.download(dst, n);
    std::vector<double> tmp(n);
    ORC_TRY(src.download(tmp.data(), n));
    for (size_t c = 0; c < n; ++c) dst[g[c]] = tmp[c];
    return ORC_OK;
}
}  // namespace

namespace {
// A drop-in for solver::solve_steady must not hide that the breakdown guard (default on) kept a solve alive where the
// reference divides 0/0 and panics "solution diverged" (solver.rs:217-221): an ORC_OK return then carries a note.
struct GuardNote {
    int64_t before;
    GuardNote() : before(orc_breakdown_guard_events(0)) {}
    void leave(int status) const {
        if (status != ORC_OK) return;
        const int64_t fired = orc_breakdown_guard_events(0) - before;
        if (fired > 0) set_error(ORC_OK, "breakdown guard fired in %lld solve(s): the reference would have returned NaN (linear_algebra.rs:255-268)", (long long)fired);
        else ctx().last_error.clear();
    }
};
}  // namespace

int orc_solver_iterate(OrcSolver *s, uint64_t iterations, double *report) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    GuardNote note;
    const int st = solver_iterate(s->st, iterations, report);
    note.leave(st);
    return st;
}

int orc_solver_snapshot(OrcSolver *s) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    DevBuf<double> *src[7] = {&t.u, &t.v, &t.w, &t.p, &t.du, &t.dv, &t.dw};
    for (int k = 0; k < 7; ++k) {
        ORC_TRY(t.snap[k].ensure((size_t)t.n));
        ORC_TRY(vec_copy(t.snap[k].p, src[k]->p, t.n));
    }
    t.snap_iterations = t.iterations_done;
    t.has_snapshot = true;
    return ORC_OK;
}

int orc_solver_restore(OrcSolver *s) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    if (!t.has_snapshot) return set_error(ORC_ERR_BAD_ARGUMENT, "orc_solver_restore without orc_solver_snapshot");
    DevBuf<double> *dst[7] = {&t.u, &t.v, &t.w, &t.p, &t.du, &t.dv, &t.dw};
    for (int k = 0; k < 7; ++k) ORC_TRY(vec_copy(dst[k]->p, t.snap[k].p, t.n));  // asynchronous, library stream
    t.iterations_done = t.snap_iterations;
    return ORC_OK;
}

int orc_solver_assemble_momentum(OrcSolver *s, double *a_u, double *a_v, double *a_w, double *b_u, double *b_v, double *b_w,
                                 double peclet[3]) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    const bool tvd = t.settings.momentum >= ORC_MOMENTUM_TVD_LUD;
    ORC_TRY(k_gradients(t, tvd));
    ORC_TRY(k_face_flux(t, true));
    double pe[3];
    ORC_TRY(k_momentum(t, pe));
    if (peclet) { peclet[0] = pe[0]; peclet[1] = pe[1]; peclet[2] = pe[2]; }
    DevBuf<double> tmp;
    const size_t n = (size_t)t.n;
    if (a_u) ORC_TRY(download_csr_values(*t.mesh, t.a_u, a_u, tmp));
    if (a_v) ORC_TRY(download_csr_values(*t.mesh, t.a_v, a_v, tmp));
    if (a_w) ORC_TRY(download_csr_values(*t.mesh, t.a_w, a_w, tmp));
    if (b_u) ORC_TRY(t.b_u.download(b_u, n));
    if (b_v) ORC_TRY(t.b_v.download(b_v, n));
    if (b_w) ORC_TRY(t.b_w.download(b_w, n));
    return fetch_status(t);
}

int orc_solver_assemble_pressure(OrcSolver *s, double *a_p, double *b_p) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    ORC_TRY(k_gradients(t, false));
    ORC_TRY(k_pressure_correction(t));
    DevBuf<double> tmp;
    if (a_p) ORC_TRY(download_csr_values(*t.mesh, t.a_p, a_p, tmp));
    if (b_p) ORC_TRY(t.b_p.download(b_p, (size_t)t.n));
    return fetch_status(t);
}

// ---------------------------------------------------------------- discretization::* with host arrays
int orc_build_momentum_diffusion_matrix(const OrcMesh *m, int diffusion_scheme, double mu, double *a_values, double *b_u,
                                        double *b_v, double *b_w) {
    if (!m) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    OrcSettings st;
    orc_settings_default(&st);
    st.diffusion = diffusion_scheme;
    auto s = std::make_unique<OrcSolver>();
    ORC_TRY(solver_init(s->st, const_cast<OrcMesh *>(m), &st, 1.0, mu));
    DevBuf<double> tmp;
    ORC_TRY(download_csr_values(*s->st.mesh, s->st.a_di, a_values, tmp));
    const size_t n = (size_t)m->n_cells;
    ORC_TRY(s->st.b_u_di.download(b_u, n));
    ORC_TRY(s->st.b_v_di.download(b_v, n));
    ORC_TRY(s->st.b_w_di.download(b_w, n));
    return ORC_OK;
}

int orc_initialize_momentum_matrix(const OrcMesh *m, double *a_values) {
    if (!m) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    OrcSettings st;
    orc_settings_default(&st);
    auto s = std::make_unique<OrcSolver>();
    ORC_TRY(solver_init(s->st, const_cast<OrcMesh *>(m), &st, 1.0, 1.0));
    DevBuf<double> tmp;
    return download_csr_values(*s->st.mesh, s->st.a_u, a_values, tmp);
}

int orc_build_momentum_advection_matrices(const OrcMesh *m, double *a_u_values, double *a_v_values, double *a_w_values, double *b_u,
                                          double *b_v, double *b_w, const double *a_di_values, const double *u, const double *v,
                                          const double *w, const double *p, const OrcSettings *settings, double rho,
                                          double peclet[3]) {
    if (!m || !settings) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    auto s = std::make_unique<OrcSolver>();
    SolverState &t = s->st;
    ORC_TRY(solver_init(t, const_cast<OrcMesh *>(m), settings, rho, 1.0));
    DevBuf<double> tmp;
    // a_di, and the incoming a_u/a_v/a_w whose diagonals Rhie-Chow reads (solver.rs:1068-1081)
    ORC_TRY(upload_csr_values(*t.mesh, a_di_values, t.a_di, tmp));
    ORC_TRY(upload_csr_values(*t.mesh, a_u_values, t.a_u, tmp));
    ORC_TRY(upload_csr_values(*t.mesh, a_v_values, t.a_v, tmp));
    ORC_TRY(upload_csr_values(*t.mesh, a_w_values, t.a_w, tmp));
    std::vector<double> d((size_t)t.n);
    const double *mats[3] = {a_u_values, a_v_values, a_w_values};
    DevBuf<double> *diags[3] = {&t.du, &t.dv, &t.dw};
    for (int k = 0; k < 3; ++k) {
        for (int64_t c = 0; c < t.n_own; ++c) {
            const int64_t *row = m->h_col.data() + m->h_row_ptr[(size_t)c];
            const int64_t len = m->h_row_ptr[(size_t)c + 1] - m->h_row_ptr[(size_t)c];
            d[(size_t)c] = mats[k][m->h_row_ptr[(size_t)c] + (std::lower_bound(row, row + len, c) - row)];
        }
        ORC_TRY(diags[k]->upload(d.data(), (size_t)t.n));
    }
    // the diffusion RHS is added by the caller in the reference (solver.rs:80-82): return b without it
    ORC_TRY(t.b_u_di.zero()); ORC_TRY(t.b_v_di.zero()); ORC_TRY(t.b_w_di.zero());
    ORC_TRY(orc_solver_set_fields(s.get(), u, v, w, p));
    return orc_solver_assemble_momentum(s.get(), a_u_values, a_v_values, a_w_values, b_u, b_v, b_w, peclet);
}

int orc_build_pressure_correction_matrices(const OrcMesh *m, const double *u, const double *v, const double *w, const double *p,
                                           const double *a_u_values, const double *a_v_values, const double *a_w_values,
                                           const OrcSettings *settings, double rho, double *a_values, double *b) {
    if (!m || !settings) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    auto s = std::make_unique<OrcSolver>();
    SolverState &t = s->st;
    ORC_TRY(solver_init(t, const_cast<OrcMesh *>(m), settings, rho, 1.0));
    std::vector<double> d((size_t)t.n);
    const double *mats[3] = {a_u_values, a_v_values, a_w_values};
    DevBuf<double> *diags[3] = {&t.du, &t.dv, &t.dw};
    for (int k = 0; k < 3; ++k) {
        for (int64_t c = 0; c < t.n_own; ++c) {
            const int64_t *row = m->h_col.data() + m->h_row_ptr[(size_t)c];
            const int64_t len = m->h_row_ptr[(size_t)c + 1] - m->h_row_ptr[(size_t)c];
            d[(size_t)c] = mats[k][m->h_row_ptr[(size_t)c] + (std::lower_bound(row, row + len, c) - row)];
        }
        ORC_TRY(diags[k]->upload(d.data(), (size_t)t.n));
    }
    ORC_TRY(orc_solver_set_fields(s.get(), u, v, w, p));
    return orc_solver_assemble_pressure(s.get(), a_values, b);
}

int orc_calculate_gradients(const OrcMesh *m, const double *u, const double *v, const double *w, const double *p,
                            const OrcSettings *settings, double *grad_p, double *grad_u) {
    if (!m || !settings) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    auto s = std::make_unique<OrcSolver>();
    SolverState &t = s->st;
    ORC_TRY(solver_init(t, const_cast<OrcMesh *>(m), settings, 1.0, 1.0));
    ORC_TRY(orc_solver_set_fields(s.get(), u, v, w, p));
    ORC_TRY(k_gradients(t, grad_u != nullptr));
    const size_t n = (size_t)t.n;
    if (grad_p) {
        std::vector<double> g(3 * n);
        ORC_TRY(t.gp.download(g.data(), 3 * n));
        for (size_t c = 0; c < n; ++c)
            for (int k = 0; k < 3; ++k) grad_p[3 * c + k] = g[k * n + c];
    }
    if (grad_u) {
        std::vector<double> g(9 * n);
        ORC_TRY(t.gu.download(g.data(), 9 * n));
        for (size_t c = 0; c < n; ++c)
            for (int k = 0; k < 9; ++k) grad_u[9 * c + k] = g[k * n + c];
    }
    return fetch_status(t);
}

// ---------------------------------------------------------------- solver::initialize_* (solver.rs:246-509)
namespace {
// the scheme triple initialize_flow hard-codes (solver.rs:301-304); q1_compat / breakdown_guard follow the caller
OrcSettings initializer_settings(const OrcSettings *settings) {
    OrcSettings t;
    orc_settings_default(&t);
    if (settings) { t.q1_compat = settings->q1_compat; t.breakdown_guard = settings->breakdown_guard; t.reduction_order = settings->reduction_order; }
    t.momentum = ORC_MOMENTUM_UD;
    t.velocity_interpolation = ORC_VINTERP_LINEAR_WEIGHTED;
    t.pressure_interpolation = ORC_PINTERP_LINEAR_WEIGHTED;
    return t;
}
}  // namespace

int orc_check_boundary_conditions(const OrcMesh *m, int *constraint_type) {
    if (!m) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    const int kind = check_boundary_conditions(*m);
    if (kind < 0) return set_error(-kind, "You must set boundary conditions.");
    if (constraint_type) *constraint_type = kind;
    return ORC_OK;
}

int orc_initialize_pressure_field(const OrcMesh *m, double *p) {
    if (!m || !p) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    const OrcSettings t = initializer_settings(nullptr);
    auto s = std::make_unique<OrcSolver>();
    ORC_TRY(solver_init(s->st, const_cast<OrcMesh *>(m), &t, 1.0, 1.0));
    {  // the incoming p (the reference overwrites it) in the mesh's internal order, the result back in ORC order
        const std::vector<int64_t> &g = m->h_global_ids;
        const size_t n = (size_t)s->st.n;
        if (g.empty()) ORC_TRY(s->st.p.upload(p, n));
        else {
            std::vector<double> tmp(n);
            for (size_t c = 0; c < n; ++c) tmp[c] = p[g[c]];
            ORC_TRY(s->st.p.upload(tmp.data(), n));
        }
    }
    int st = initialize_pressure_field_dev(s->st);
    ORC_TRY(download_in_orc_order(*m, s->st.p, p, (size_t)s->st.n));
    return st;
}

int orc_initialize_flow(const OrcMesh *m, double mu, double rho, uint64_t iteration_count, const OrcSettings *settings, double *u, double *v,
                        double *w, double *p) {
    if (!m || !u || !v || !w || !p) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    const OrcSettings t = initializer_settings(settings);
    auto s = std::make_unique<OrcSolver>();
    ORC_TRY(solver_init(s->st, const_cast<OrcMesh *>(m), &t, rho, mu));  // fields start at zero (:273-277)
    int st = initialize_flow_dev(s->st, iteration_count);
    int st2 = orc_solver_get_fields(s.get(), u, v, w, p);
    return st != ORC_OK ? st : st2;
}

// initialize_velocity_field (solver.rs:511-696): u, v, w from the potential psi; psi (optional) is what the reference
// writes to ./examples/psi.csv — the files themselves are left to the caller (orc_write_data / orc_write_gradients).
int orc_initialize_velocity_field(const OrcMesh *m, const OrcSettings *settings, double *u, double *v, double *w, double *psi) {
    if (!m || !u || !v || !w) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    const OrcSettings t = initializer_settings(settings);
    auto s = std::make_unique<OrcSolver>();
    ORC_TRY(solver_init(s->st, const_cast<OrcMesh *>(m), &t, 1.0, 1.0));
    int st = initialize_velocity_field_dev(s->st);
    const size_t n = (size_t)s->st.n;
    int st2 = download_in_orc_order(*m, s->st.u, u, n);
    if (st2 == ORC_OK) st2 = download_in_orc_order(*m, s->st.v, v, n);
    if (st2 == ORC_OK) st2 = download_in_orc_order(*m, s->st.w, w, n);
    if (st2 == ORC_OK && psi) st2 = download_in_orc_order(*m, s->st.p_prime, psi, n);
    return st != ORC_OK ? st : st2;
}

// initialize_flow_new (solver.rs:354-410): the match has overlapping arms, so Hybrid takes the first one — PressureOnly |
// Hybrid -> initialize_pressure_field (velocities stay zero), VelocityOnly -> initialize_velocity_field (pressure stays zero).
int orc_initialize_flow_new(const OrcMesh *m, double mu, double rho, uint64_t iteration_count, double *u, double *v, double *w, double *p) {
    (void)mu; (void)rho; (void)iteration_count;
    if (!m || !u || !v || !w || !p) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    int kind = 0;
    ORC_TRY(orc_check_boundary_conditions(m, &kind));
    const size_t n = (size_t)m->n_cells;
    std::fill(u, u + n, 0.); std::fill(v, v + n, 0.); std::fill(w, w + n, 0.); std::fill(p, p + n, 0.);
    if (kind == 1) return orc_initialize_velocity_field(m, nullptr, u, v, w, nullptr);
    return orc_initialize_pressure_field(m, p);
}

// ---------------------------------------------------------------- solver::solve_steady (solver.rs:26-244)
int orc_solve_steady(OrcMesh *m, double *u, double *v, double *w, double *p, const OrcSettings *settings, double rho, double mu,
                     uint64_t iteration_count, uint64_t reporting_interval, OrcReportFn report_cb, void *user) {
    if (!m || !settings || !u || !v || !w || !p) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    int st = ORC_OK;
    OrcSolver *s = orc_solver_create(m, settings, rho, mu, &st);
    if (!s) return st;
    std::unique_ptr<OrcSolver> guard(s);
    ORC_TRY(orc_solver_set_fields(s, u, v, w, p));
    GuardNote note;
    auto start = std::chrono::steady_clock::now();
    for (uint64_t it = 1; it <= iteration_count && st == ORC_OK; ++it) {
        double rep[8];
        const bool want = report_cb && reporting_interval > 0 && it % reporting_interval == 0;
        st = solver_iterate(s->st, 1, want ? rep : nullptr);
        if (want && st == ORC_OK) {  // solver.rs:209-216
            auto now = std::chrono::steady_clock::now();
            double ms = std::chrono::duration<double, std::milli>(now - start).count() / (double)reporting_interval;
            start = now;
            report_cb(it, rep, rep + 3, rep[6], rep[7], ms, user);
        }
    }
    if (st == ORC_OK) st = post_loop_gradients_dev(s->st);  // solver.rs:227-242
    int st2 = orc_solver_get_fields(s, u, v, w, p);  // fields are mutated in place up to a panic
    note.leave(st != ORC_OK ? st : st2);
    return st != ORC_OK ? st : st2;
}

int orc_bench_spmv(OrcSolver *s, int reps, double *avg_ms, double *checksum) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    MatView A;
    A.P = t.mesh->pat.dev();
    A.val = t.a_u.p;
    A.persistent_pattern = true;
    Arena::Mark mk = t.arena.mark();
    double *y;
    ORC_TRY(t.arena.alloc((size_t)t.n, &y));
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    ORC_TRY(spmv_dev(A, t.u.p, y));
    ORC_HIP(hipEventRecord(e0, ctx().stream));
    for (int i = 0; i < reps; ++i) ORC_TRY(spmv_dev(A, t.u.p, y));
    ORC_HIP(hipEventRecord(e1, ctx().stream));
    ORC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    ORC_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = (double)ms / std::max(reps, 1);
    if (checksum) {
        std::vector<double> h((size_t)t.n);
        ORC_HIP(hipMemcpy(h.data(), y, sizeof(double) * (size_t)t.n, hipMemcpyDeviceToHost));
        double c = 0.;
        for (double x : h) c += x;
        *checksum = c;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    t.arena.release(mk);
    return ORC_OK;
}

// The level-0 products of the momentum system a_u exactly as the solver's BiCGSTAB iterations launch them inside the Multigrid
// arm: through the arm's Jacobi scaling and the smoother's nested one, with the reduction epilogues.
// avg_ms[0], [1]: spmv_uniform_k<EpiStoreSum, false, true>, <EpiTs, false, true> (one system: what the p' solve and any one-system
// solve run); avg_ms[2], [3]: spmv3_uniform_k<EpiStoreSum3, 4, true>, <EpiTs3, 4, true> (u, v, w in one launch).
// the template arguments the two launches above are made with, as rocprofv3 prints them: "<narrow>, <scaled>" of
// spmv_uniform_k<Epi, false, true, narrow, scaled> (and of spmv3_uniform_k<Epi3, 4, true, narrow, scaled>)
static char g_inloop_variant[32] = "false, true, false";
const char *orc_bench_inloop_variant(void) { return g_inloop_variant; }

int orc_bench_inloop_products(OrcSolver *s, int reps, double avg_ms[4]) {
    if (!s || !avg_ms) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SolverState &t = s->st;
    if (reps < 1) reps = 1;
    ArenaScope scope(t.arena);
    const size_t n = (size_t)std::max<int64_t>(t.n, 1);
    MatView A;
    A.P = t.mesh->pat.dev();
    A.val = t.a_u.p;
    A.symmetric = t.mesh->pat.symmetric;
    A.persistent_pattern = true;
    double *d1, *d2, *y, *partials;
    ORC_TRY(t.arena.alloc(n, &d1));
    ORC_TRY(t.arena.alloc(n, &d2));
    ORC_TRY(t.arena.alloc(n, &y));
    ORC_TRY(t.arena.alloc((size_t)6 * kMaxPartials, &partials));
    const bool jac = t.settings.preconditioner == ORC_PRECOND_JACOBI;
    if (jac) {
        ORC_TRY(diag_inverse_dev(A, d1));
        A.s1 = d1;
        ORC_TRY(diag_inverse_dev(A, d2));
        A.s2 = d2;
    }
    ORC_TRY(materialize_scaled_view(A, t.settings.iterations, t.arena));  // as a smoothing solve of the configured length does
    {  // narrow columns, scalings on the fly, non-temporal matrix loads: the last three template arguments of the kernels just timed
        const bool narrow = A.P.col16 != nullptr, scaled = A.s1 || A.s2;
        snprintf(g_inloop_variant, sizeof(g_inloop_variant), "%s, %s, %s", narrow ? "true" : "false", scaled ? "true" : "false",
                 (!scaled && matview_stream_nt(A)) ? "true" : "false");
    }
    float ms[2];
    ORC_TRY(bench_inloop_products_dev(A, t.u.p, y, partials, reps, ms));
    avg_ms[0] = ms[0]; avg_ms[1] = ms[1];
    avg_ms[2] = avg_ms[3] = 0.;
    if (triple_supported()) {
        MatView3 A3;
        A3.P = t.mesh->pat.dev();
        A3.val[0] = t.a_u.p; A3.val[1] = t.a_v.p; A3.val[2] = t.a_w.p;
        A3.mesh_pattern = true;
        double *e1, *e2, *x3, *y3;
        ORC_TRY(t.arena.alloc(3 * n, &e1));
        ORC_TRY(t.arena.alloc(3 * n, &e2));
        ORC_TRY(t.arena.alloc(3 * n, &x3));
        ORC_TRY(t.arena.alloc(3 * n, &y3));
        ORC_TRY(interleave3_dev(t.u.p, t.v.p, t.w.p, x3, t.n));
        if (jac) {
            ORC_TRY(diag_inverse3_dev(A3, e1));
            A3.s1 = e1;
            ORC_TRY(diag_inverse3_dev(A3, e2));
            A3.s2 = e2;
        }
        ORC_TRY(materialize_scaled_view3(A3, t.settings.iterations, t.arena));
        ORC_TRY(bench_inloop_products3_dev(A3, x3, y3, partials, reps, ms));
        avg_ms[2] = ms[0]; avg_ms[3] = ms[1];
    }
    return ORC_OK;
}

// One multicolour Gauss-Seidel sweep over a_u (the preconditioner application of the GS-preconditioned BiCGSTAB, BASELINE configs[2]):
// n_colors launches of gs_color_sorted_k; average ms per SWEEP over `reps` sweeps.
int orc_bench_gs_sweep(OrcSolver *s, int reps, double *avg_ms, int *n_colors) {
    if (!s || !avg_ms) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SolverState &t = s->st;
    MatView A;
    A.P = t.mesh->pat.dev();
    A.val = t.a_u.p;
    A.symmetric = t.mesh->pat.symmetric;
    A.persistent_pattern = true;
    ArenaScope scope(t.arena);
    double *x;
    ORC_TRY(t.arena.alloc((size_t)std::max<int64_t>(t.n, 1), &x));
    float ms = 0.f;
    int nc = 0;
    ORC_TRY(bench_gs_sweep_dev(A, t.b_u.p, x, std::max(reps, 1), t.arena, &ms, &nc));
    *avg_ms = ms;
    if (n_colors) *n_colors = nc;
    return ORC_OK;
}

// [r04] the same application as the slot-space GS-BiCGSTAB launches it (from zero, no fill): avg_ms[0] one system, avg_ms[1] u, v, w per launch
int orc_bench_gs_sweep0(OrcSolver *s, int reps, double avg_ms[2], int *n_colors) {
    if (!s || !avg_ms) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SolverState &t = s->st;
    MatView A[3];
    const double *vals[3] = {t.a_u.p, t.a_v.p, t.a_w.p};
    const double *b[3] = {t.b_u.p, t.b_v.p, t.b_w.p};
    for (int k = 0; k < 3; ++k) {
        A[k].P = t.mesh->pat.dev();
        A[k].val = vals[k];
        A[k].symmetric = t.mesh->pat.symmetric;
        A[k].persistent_pattern = true;
    }
    float ms[2] = {0.f, 0.f};
    int nc = 0;
    ORC_TRY(bench_gs_sweep0_dev(A, b, std::max(reps, 1), t.arena, ms, &nc));
    avg_ms[0] = ms[0]; avg_ms[1] = ms[1];
    if (n_colors) *n_colors = nc;
    return ORC_OK;
}

int orc_bench_bicgstab_iteration(OrcSolver *s, int reps, double *avg_ms) {
    if (!s) return set_error(ORC_ERR_BAD_ARGUMENT, "null solver");
    SolverState &t = s->st;
    MatView A;
    A.P = t.mesh->pat.dev();
    A.val = t.a_u.p;
    Arena::Mark mk = t.arena.mark();
    double *x;
    ORC_TRY(t.arena.alloc((size_t)t.n, &x));
    ORC_TRY(vec_copy(x, t.u.p, t.n));
    float ms = 0.f;
    ORC_TRY(bench_bicgstab_dev(A, t.b_u.p, x, reps, t.arena, &ms));
    if (avg_ms) *avg_ms = ms;
    t.arena.release(mk);
    return ORC_OK;
}

int orc_bench_amg_levels(OrcSolver *s, int reps, int64_t *rows, int64_t *nnz, int64_t *padded, double *avg_ms, int *n_levels) {
    if (!s || !rows || !nnz || !padded || !avg_ms || !n_levels) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    SolverState &t = s->st;
    if (reps < 1) reps = 1;
    Arena::Mark mk = t.arena.mark();
    MatView A;
    A.P = t.mesh->pat.dev();
    A.val = t.a_u.p;
    A.symmetric = t.mesh->pat.symmetric;
    A.persistent_pattern = true;
    const int pre = t.settings.preconditioner;
    AmgHierarchy H;
    ORC_TRY(multigrid_prepare_dev(A, pre, t.arena, H));
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    int count = 0;
    for (int l = 0; l <= H.n_levels && l < 4; ++l) {
        MatView V;
        int64_t n_l, padded_l;
        if (l == 0) { V = A; n_l = t.mesh->pat.n; padded_l = t.mesh->pat.padded; }
        else {
            const AmgHierarchy::Level &h = H.level[l - 1];
            V.P = h.P; V.val = h.val; V.pk = h.pk; V.xw = h.xw; V.symmetric = A.symmetric;
            n_l = h.n; padded_l = h.padded;
            if (cfg().debug_xwin && h.xw.wsize) {  // window statistics of the level (measurement runs only)
                const int64_t nb = ((int64_t)h.P.n_slices + 3) / 4;
                std::vector<int32_t> ws((size_t)nb);
                ORC_HIP(hipStreamSynchronize(ctx().stream));
                ORC_HIP(hipMemcpy(ws.data(), h.xw.wsize, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
                int64_t neg = 0, mx = 0;
                double sum = 0.;
                for (int32_t w : ws) { if (w < 0) ++neg; else { sum += w; mx = std::max<int64_t>(mx, w); } }
                fprintf(stderr, "[orc xwin] level %d: %lld blocks, %lld without a window, mean window %.0f, max %lld\n", l, (long long)nb, (long long)neg,
                        nb > neg ? sum / (double)(nb - neg) : 0., (long long)mx);
            }
        }
        // the scalings the solver's products carry: the arm's Jacobi preconditioner on level 0, and the smoother's own
        // on every level
        double *d1 = nullptr, *d2 = nullptr, *x, *y;
        const size_t nn = (size_t)std::max<int64_t>(n_l, 1);
        ORC_TRY(t.arena.alloc(nn, &x));
        ORC_TRY(t.arena.alloc(nn, &y));
        ORC_TRY(vec_fill(x, 1., n_l));
        if (pre == ORC_PRECOND_JACOBI) {
            ORC_TRY(t.arena.alloc(nn, &d1));
            ORC_TRY(diag_inverse_dev(V, d1));
            V.s1 = d1;
            if (l == 0) {
                ORC_TRY(t.arena.alloc(nn, &d2));
                ORC_TRY(diag_inverse_dev(V, d2));
                V.s2 = d2;
            }
        }
        ORC_TRY(materialize_scaled_view(V, t.settings.iterations, t.arena));  // levels 0 and 1, as their smoothing solves do
        ORC_TRY(spmv_dev(V, x, y));
        ORC_HIP(hipEventRecord(e0, ctx().stream));
        for (int i = 0; i < reps; ++i) ORC_TRY(spmv_dev(V, x, y));
        ORC_HIP(hipEventRecord(e1, ctx().stream));
        ORC_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ORC_HIP(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int32_t> len((size_t)n_l);
        ORC_HIP(hipMemcpy(len.data(), V.P.row_len, sizeof(int32_t) * (size_t)n_l, hipMemcpyDeviceToHost));
        int64_t nz = 0;
        for (int32_t q : len) nz += q;
        rows[count] = n_l; nnz[count] = nz; padded[count] = padded_l; avg_ms[count] = (double)ms / reps;
        ++count;
    }
    *n_levels = count;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    t.arena.release(mk);
    return ORC_OK;
}

int orc_profile_report(char *buf, int64_t buf_len) {
    if (buf && buf_len > 0) buf[0] = 0;
    return ORC_OK;
}

}  // extern "C"
