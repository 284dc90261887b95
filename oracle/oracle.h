/* oracle.h — CPU restatement of ORC's per-SIMPLE-iteration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liborc_oracle.so; orc_amd/ never does.
 *
 * It restates, single-threaded and in f64, the algorithms of the reference
 * (/root/reference, ORC v0.3.0): every function cites the file:line it follows, including
 * the quirks listed in SURVEY.md §8a (Q1..Q10).  Arithmetic that the reference delegates to
 * nalgebra 0.32.4 / nalgebra-sparse 0.9.0 (not vendored, Cargo.lock:326-327,353-354) is
 * restated from those crates' published algorithms — see sparse.c.
 *
 * Parity pin: the reference has one #[test] on this path (linear_algebra.rs:309-378) and
 * printed-only validations (tests.rs:111-151, main.rs:150-172,304-326); tests/test_oracle_*.py
 * checks the oracle against all of them.  The Multigrid arm and the assembly have no
 * reference-held vectors: "parity unpinned" for those.
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/orc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } Vec3;
typedef struct { Vec3 x, y, z; } Tensor3;

/* nalgebra_sparse::CsrMatrix<f64>: sorted column indices per row, duplicates summed. */
typedef struct OrCsr {
    int64_t nrows, ncols, nnz;
    int64_t *row_ptr; /* nrows + 1 */
    int64_t *col;     /* nnz */
    double *val;      /* nnz */
} OrCsr;

/* mesh::FaceZone (mesh.rs:12-17) */
typedef struct OrZone {
    uint64_t id;
    int32_t zone_type; /* OrcFaceConditionType */
    double scalar_value;
    Vec3 vector_value;
    char name[64];
} OrZone;

/* mesh::Mesh (mesh.rs:181-187) flattened: Face (mesh.rs:140-149), Cell (mesh.rs:164-169). */
typedef struct OrMesh {
    int32_t dimensions;
    int64_t n_vertices, n_faces, n_cells, n_zones;
    Vec3 *vertices;
    /* faces */
    int32_t *face_zone;   /* index into zones[] (not the TGRID id) */
    int64_t *face_c0;     /* cell_indices[0] */
    int64_t *face_c1;     /* cell_indices[1] or -1 when cell_indices.len()==1 */
    int64_t *face_node_ptr; /* n_faces+1 */
    int64_t *face_nodes;
    double *face_area;
    Vec3 *face_centroid;
    Vec3 *face_normal;    /* unit, outward from cell_indices[0] (mesh.rs:216-222) */
    /* cells */
    int64_t *cell_face_ptr; /* n_cells+1 */
    int64_t *cell_faces;    /* ascending face id (io.rs:404-410) */
    double *cell_volume;
    Vec3 *cell_centroid;
    OrZone *zones;
} OrMesh;

/* ---- sparse.c : nalgebra / nalgebra-sparse restatements ---- */
OrCsr *or_csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz);
OrCsr *or_csr_clone(const OrCsr *a);
void or_csr_free(OrCsr *a);
OrCsr *or_csr_from_coo(int64_t nrows, int64_t ncols, int64_t n, const int64_t *ri, const int64_t *ci, const double *v);
OrCsr *or_csr_from_arrays(int64_t nrows, int64_t ncols, const int64_t *row_ptr, const int64_t *col, const double *val);
int64_t or_csr_find(const OrCsr *a, int64_t i, int64_t j); /* position or -1 (lib.rs:657-668 get_entry) */
void or_spmv(const OrCsr *a, const double *x, double *y);
OrCsr *or_spgemm(const OrCsr *a, const OrCsr *b);
OrCsr *or_transpose(const OrCsr *a);
double or_dot(const double *a, const double *b, int64_t n);
double or_norm(const double *a, int64_t n);
double or_sum(const double *a, int64_t n);
void or_set_dot_mode(int mode); /* diagnostic: 1 = pairwise association (tests only) */

/* ---- mesh_io.c ---- */
OrMesh *or_read_mesh(const char *path);  /* io.rs:32-515 */
OrMesh *or_mesh_from_arrays(int32_t dimensions, int64_t n_cells, int64_t n_faces, int32_t n_zones,
                            const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                            const double *face_area, const double *face_normal, const double *face_centroid,
                            const double *cell_centroid, const double *cell_volume,
                            const int64_t *cell_face_ptr, const int64_t *cell_faces,
                            const int32_t *zone_type, const double *zone_scalar, const double *zone_vector);
void or_mesh_free(OrMesh *m);
int or_mesh_set_zone(OrMesh *m, const char *name, int32_t zone_type, double scalar, double vx, double vy, double vz); /* mesh.rs:189-195 + tests.rs:60-76 */
int or_mesh_zone_index(const OrMesh *m, const char *name);

/* ---- linear_algebra.c ---- */
OrCsr *or_build_restriction_matrix(const OrCsr *a, int injection); /* linear_algebra.rs:12-63 */
int or_iterative_solve(const OrCsr *a, const double *b, double *x, uint64_t iteration_count, int method,
                       double relaxation_factor, double convergence_threshold, int preconditioner); /* linear_algebra.rs:144-299 */
/* statistics of the last Jacobi-arm call: sweeps executed (linear_algebra.rs:188-217) */
int64_t or_last_jacobi_sweeps(void);

/* ---- discretization.c ---- */
int or_build_momentum_diffusion_matrix(const OrMesh *m, int diffusion_scheme, double mu, OrCsr **a_out,
                                       double *b_u, double *b_v, double *b_w); /* discretization.rs:39-131 */
OrCsr *or_initialize_momentum_matrix(const OrMesh *m); /* discretization.rs:450-472 */
int or_build_momentum_advection_matrices(OrCsr *a_u, OrCsr *a_v, OrCsr *a_w, double *b_u, double *b_v, double *b_w,
                                         const OrCsr *a_di, const OrMesh *m, const double *u, const double *v,
                                         const double *w, const double *p, const OrcSettings *s, double rho,
                                         double peclet_out[3]); /* discretization.rs:134-356 */
int or_build_pressure_correction_matrices(const OrMesh *m, const double *u, const double *v, const double *w,
                                          const double *p, const OrCsr *a_u, const OrCsr *a_v, const OrCsr *a_w,
                                          const OrcSettings *s, double rho, OrCsr **a_out, double *b_out); /* discretization.rs:359-448 */

/* ---- solver.c ---- */
int or_calculate_pressure_gradient(const OrMesh *m, const double *p, int64_t cell, int scheme, int q1, Vec3 *out);   /* solver.rs:874-950 */
int or_calculate_velocity_gradient(const OrMesh *m, const double *u, const double *v, const double *w, int64_t cell, int scheme, Tensor3 *out); /* solver.rs:774-872 */
int or_get_face_velocity(const OrMesh *m, const double *u, const double *v, const double *w, int64_t face, int scheme, Vec3 *out); /* solver.rs:952-1003 */
int or_get_face_pressure(const OrMesh *m, const double *p, int64_t face, int interp, int grad_scheme, int q1, double *out); /* solver.rs:1104-1150 */
int or_get_face_flux(const OrMesh *m, const double *u, const double *v, const double *w, const double *p, int64_t face,
                     int64_t cell, int vinterp, int grad_scheme, int q1, const double *diag_u, const double *diag_v,
                     const double *diag_w, double *out); /* solver.rs:1007-1102; diag_* = a_{u,v,w}.get(i,i) */
int or_apply_pressure_correction(const OrMesh *m, const double *diag_u, const double *diag_v, const double *diag_w,
                                 const double *p_prime, double *u, double *v, double *w, double *p,
                                 const OrcSettings *s, double out_norms[2]); /* solver.rs:1170-1227 */
/* report[6*iter + k]: u_avg, v_avg, w_avg, peclet_avg, vel_corr, p_corr (solver.rs:206-216); may be NULL */
int or_solve_steady(const OrMesh *m, double *u, double *v, double *w, double *p, const OrcSettings *s, double rho,
                    double mu, uint64_t iteration_count, double *report); /* solver.rs:26-244 */
int or_initialize_flow(const OrMesh *m, double mu, double rho, uint64_t iteration_count, int q1_compat,
                       double *u, double *v, double *w, double *p); /* solver.rs:246-352 */
int or_initialize_velocity_field(const OrMesh *m, double *u, double *v, double *w, double *psi_out);
int or_initialize_pressure_field(const OrMesh *m, double *p); /* solver.rs:414-509 */

void or_settings_default(OrcSettings *s); /* lib.rs:58-86 */
const char *or_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif
