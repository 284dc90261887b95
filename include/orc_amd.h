/* orc_amd.h — C ABI of liborc_amd.so: ORC's per-SIMPLE-iteration hot path on MI355X (gfx950).
 *
 * ORC (reidprichard/ORC v0.3.0) has no plugin/FFI layer; the drop-in boundary is its ordinary
 * `pub fn` surface.  Each entry point below replaces one of those functions and
 * cites it (file:line into the reference).  A Rust `extern "C"` shim that keeps ORC's own
 * signatures on top of this header is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; every array is caller-owned host memory unless the name
 *    says `_dev`; the library copies in/out and keeps its own device allocations.
 *  - fields and matrices are in ORC cell order; indices are 0-based int64 (`usize`).
 *  - return value: OrcStatus (include/orc_types.h). ORC panics where we return non-zero; the
 *    shim turns non-zero into panic!(orc_status_string(code)).
 *  - there is NO CPU fallback: without a HIP device every compute entry returns ORC_ERR_NO_DEVICE.
 *  - CSR matrices use ORC's pattern: per cell the diagonal plus one entry per interior face,
 *    columns ascending (what CsrMatrix::from(&CooMatrix) yields at discretization.rs:130,445,471).
 */
#ifndef ORC_AMD_H
#define ORC_AMD_H

#include <stdint.h>
#include "orc_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcMesh OrcMesh;     /* device-resident SoA image of mesh::Mesh (mesh.rs:181-187) */
typedef struct OrcSolver OrcSolver; /* device-resident state of one solve_steady call (solver.rs:39-49) */

/* ---------- runtime ---------- */
int orc_init(int device_ordinal);              /* hipSetDevice + stream; idempotent */
int orc_device_count(void);                    /* 0 when no HIP device is visible */
int orc_synchronize(void);                     /* hipStreamSynchronize on the library stream */
const char *orc_status_string(int status);     /* the reference's panic text for the code */
const char *orc_last_error(void);              /* detail of the last ORC_ERR_HIP / BAD_ARGUMENT */
/* The library reads its environment switches (ORC_*: INTEGRATION.md lists them; none changes a result) ONCE, in the first orc_init; a caller
 * that changes one afterwards — the tests do — says so here.  Never called by the library itself. */
int orc_reload_environment(void);
int orc_device_memory(int64_t *free_bytes, int64_t *total_bytes); /* hipMemGetInfo of the library's device */
/* "<device name> | pci <domain:bus:device.function> | ordinal <n> | <CUs> CUs" of the library's device: a multi-GPU run reports it per rank,
 * so that "did N ranks run on N different cards" is answered by the result itself.  Returns ORC_ERR_BAD_ARGUMENT when cap is too small. */
int orc_device_info(char *buf, int cap);
void orc_settings_default(OrcSettings *s);     /* NumericalSettings::default() + MatrixSolverSettings::default(), lib.rs:58-86 */

/* ---------- mesh::Mesh (mesh.rs:140-187) ---------- */
/* face_c0/face_c1: Face.cell_indices[0], [1] (-1 when the face has one cell, io.rs:332-337);
 * face_normal: unit normal outward from face_c0 (mesh.rs:216-222); face_zone: index into the
 * zone arrays; zone_type: OrcFaceConditionType per zone (FaceZone.zone_type, mesh.rs:12-17);
 * cell_faces: Cell.face_indices, ascending face id (io.rs:404-410). */
OrcMesh *orc_mesh_create(int64_t n_cells, int64_t n_faces, int32_t n_zones,
                         const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                         const double *face_area, const double *face_normal /*[3F]*/, const double *face_centroid /*[3F]*/,
                         const double *cell_centroid /*[3n]*/, const double *cell_volume,
                         const int64_t *cell_face_ptr /*[n+1]*/, const int64_t *cell_faces,
                         const int32_t *zone_type, const double *zone_scalar, const double *zone_vector /*[3Z]*/,
                         int *status);
/* The same mesh with its cells renumbered internally by `ordering` (OrcCellOrdering; ORC_ORDER_RCM = the north star's
 * "rows sorted by RCM"): for meshes whose generator numbered the cells arbitrarily.  orc_solver_set_fields /
 * orc_solver_get_fields / orc_solve_steady keep taking and returning fields in ORC order; matrices, patterns and the
 * per-cell arrays of the assembly entry points are in the internal order, which orc_mesh_cell_order reports
 * (global_ids[c] = ORC index of internal cell c).  A renumbered run is the reference's algorithm on a renumbered mesh: its
 * iterates differ from the ORC-order run wherever the reference depends on the numbering (pairwise aggregation by row
 * index), its converged fields do not. */
OrcMesh *orc_mesh_create_reordered(int64_t n_cells, int64_t n_faces, int32_t n_zones,
                                   const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                                   const double *face_area, const double *face_normal, const double *face_centroid,
                                   const double *cell_centroid, const double *cell_volume,
                                   const int64_t *cell_face_ptr, const int64_t *cell_faces,
                                   const int32_t *zone_type, const double *zone_scalar, const double *zone_vector,
                                   int32_t ordering, int *status);
int orc_mesh_cell_order(const OrcMesh *m, int64_t *global_ids /*[n]*/);
/* One rank's part of a cell-partitioned mesh: local cells are numbered owned first [0, n_owned), then one
 * contiguous block of ghost cells per peer (recv_ptr, in peer order); cell_face_ptr gives ghost cells empty face lists;
 * faces are those touching an owned cell, in ascending global id with the global c0/c1 orientation.
 * send_idx[send_ptr[q] .. send_ptr[q+1]) are the owned cells whose values peer q needs, in the order of q's ghost block.
 * Requires orc_comm_init (RCCL) or orc_comm_set_host_transport first when n_peers > 0. */
OrcMesh *orc_mesh_create_partitioned(int64_t n_owned, int64_t n_cells, int64_t n_cells_global, int64_t n_faces, int32_t n_zones,
                                     const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone, const double *face_area,
                                     const double *face_normal, const double *face_centroid, const double *cell_centroid,
                                     const double *cell_volume, const int64_t *cell_face_ptr, const int64_t *cell_faces,
                                     const int32_t *zone_type, const double *zone_scalar, const double *zone_vector, int32_t n_peers,
                                     const int32_t *peers, const int64_t *send_ptr, const int64_t *send_idx, const int64_t *recv_ptr,
                                     int *status);
int64_t orc_mesh_n_owned(const OrcMesh *m);
/* Cell partitioning of any mesh (host only, no device needed): the cells are put in `ordering` (OrcCellOrdering) and cut
 * into n_ranks contiguous blocks; the result holds rank `rank`'s arrays exactly as orc_mesh_create_partitioned takes them
 * (owned cells in block order, ghost blocks per peer sorted by position in the order, faces touching an owned cell in
 * ascending global id).  Every rank calls it on the same global arrays (those of orc_mesh_create / orc_mesh_data_arrays);
 * my send list to a peer and that peer's ghost block of my cells coincide by construction.  global_ids[n_local] maps a
 * local cell to its ORC index: fields are scattered / gathered through it, so the permutation stays internal.
 * n_ranks = 1 gives the whole mesh renumbered (e.g. RCM row ordering on one GPU). */
typedef struct OrcPartition OrcPartition;
OrcPartition *orc_mesh_partition(int64_t n_cells, int64_t n_faces, const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                                 const double *face_area, const double *face_normal, const double *face_centroid,
                                 const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                                 const int64_t *cell_faces, int32_t n_ranks, int32_t rank, int32_t ordering, int *status);
/* [r04] The same for a mesh a rank generated or read FOR ITSELF (its share of the cells plus ghost layers): the caller names the
 * owner of every cell (cell_owner[n_cells]: a rank, or -1 = nobody's — e.g. the outer of two generated ghost layers; such a cell
 * may not touch a cell of `rank`), the cells keep their order, n_global is the cell count of the whole mesh (< 0: n_cells).
 * No process ever holds the whole mesh: BASELINE configs[4] (40 M mixed cells on 8 GPUs) generates 1/8 + two block layers per
 * side on every rank.  The two ranks of a cut must number the cells they share in the same relative order (generators that
 * number layer by layer do; tests/test_partition_cpu.py exchanges global ids over gloo to check it).  global_ids are indices
 * into the caller's own arrays. */
OrcPartition *orc_mesh_partition_owner(int64_t n_cells, int64_t n_faces, const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                                       const double *face_area, const double *face_normal, const double *face_centroid,
                                       const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                                       const int64_t *cell_faces, const int32_t *cell_owner, int32_t n_ranks, int32_t rank, int64_t n_global,
                                       int *status);
void orc_partition_destroy(OrcPartition *p);
int orc_partition_sizes(const OrcPartition *p, int64_t *n_owned, int64_t *n_local, int64_t *n_global, int64_t *n_faces, int64_t *n_cell_faces,
                        int32_t *n_peers, int64_t *n_send);
/* any pointer may be NULL; sizes from orc_partition_sizes (send_ptr / recv_ptr: n_peers + 1) */
int orc_partition_arrays(const OrcPartition *p, int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area, double *face_normal,
                         double *face_centroid, double *cell_centroid, double *cell_volume, int64_t *cell_face_ptr, int64_t *cell_faces,
                         int64_t *global_ids, int64_t *global_face_ids, int32_t *peers, int64_t *send_ptr, int64_t *send_idx, int64_t *recv_ptr);
/* = orc_mesh_create_partitioned on the partition's arrays and the (global) zone tables */
OrcMesh *orc_partition_upload(const OrcPartition *p, int32_t n_zones, const int32_t *zone_type, const double *zone_scalar,
                              const double *zone_vector, int *status);
/* mesh.get_face_zone(name).zone_type / scalar_value / vector_value = ... (tests.rs:60-76) */
int orc_mesh_update_zones(OrcMesh *m, const int32_t *zone_type, const double *zone_scalar, const double *zone_vector);
void orc_mesh_destroy(OrcMesh *m);
int64_t orc_mesh_n_cells(const OrcMesh *m);
int64_t orc_mesh_nnz(const OrcMesh *m);
/* CSR pattern shared by a_di, a_u, a_v, a_w and the pressure-correction matrix */
int orc_mesh_matrix_pattern(const OrcMesh *m, int64_t *row_ptr /*[n+1]*/, int64_t *col_idx /*[nnz]*/);

/* ---------- io::read_mesh (io.rs:32-515) and Mesh::get_face_zone (mesh.rs:189-195): host only, no device needed ----------
 * TGRID / Fluent ASCII .msh -> flat host image of mesh::Mesh with the reference's numbering and geometry rules
 * (normal (n2-n1)x(n1-n0) normalised, flipped when cell 0 is absent; face centroid = node mean; area = triangle fan
 * about the centroid; cell centroid = mean of face centroids; volume = sum A |(fc-cc).n| / dim; Cell.face_indices in
 * ascending face id).  Reader quirks are kept: section items are read as hexadecimal (io.rs:47-54), zone-0 declaration
 * sections are skipped (io.rs:24-30), a zone takes the last word of the preceding "(0 ...)" comment as its name
 * (io.rs:83-90), the node count of a face line is "tokens - 2" also for mixed/polygonal zones (io.rs:232).
 * Every expect()/panic! of the reference is ORC_ERR_MESH_FORMAT (ORC_ERR_IO when the file cannot be opened). */
typedef struct OrcMeshData OrcMeshData;
OrcMeshData *orc_read_mesh(const char *mesh_path, int *status);
void orc_mesh_data_destroy(OrcMeshData *d);
int orc_mesh_data_sizes(const OrcMeshData *d, int32_t *dimensions, int64_t *n_vertices, int64_t *n_cells, int64_t *n_faces,
                        int64_t *n_cell_faces, int64_t *n_face_nodes, int32_t *n_zones);
/* the argument arrays of orc_mesh_create; any pointer may be NULL */
int orc_mesh_data_arrays(const OrcMeshData *d, int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area,
                         double *face_normal, double *face_centroid, double *cell_centroid, double *cell_volume,
                         int64_t *cell_face_ptr, int64_t *cell_faces);
/* Mesh.vertices and Face.node_indices (0-based) */
int orc_mesh_data_nodes(const OrcMeshData *d, double *vertices /*[3V]*/, int64_t *face_node_ptr /*[F+1]*/, int64_t *face_nodes);
/* FaceZone k (order of first appearance in the file): TGRID zone id, type, values, name */
int orc_mesh_data_zone(const OrcMeshData *d, int32_t k, uint64_t *zone_id, int32_t *zone_type, double *scalar_value,
                       double *vector_value /*[3]*/, char *name, int64_t name_len);
int orc_mesh_data_zone_index(const OrcMeshData *d, const char *name); /* -1 when absent */
/* `let z = mesh.get_face_zone(name); z.zone_type = ..; z.scalar_value = ..; z.vector_value = ..` (tests.rs:60-76);
 * ORC_ERR_ZONE_NOT_FOUND where the reference panics */
int orc_mesh_data_set_zone(OrcMeshData *d, const char *name, int32_t zone_type, double scalar_value, const double *vector_value /*[3] or NULL*/);
OrcMesh *orc_mesh_upload(const OrcMeshData *d, int *status);    /* = orc_mesh_create on the arrays above */
int orc_mesh_sync_zones(OrcMesh *m, const OrcMeshData *d);      /* = orc_mesh_update_zones from d's zone table */

/* ---------- io::read_data / write_data / write_data_with_precision / write_gradients (io.rs:519-662) ----------
 * One line per cell: "{centroid}\t({u}, {v}, {w})\t{p}", centroid as Vector's Display "({:.2e}, {:.2e}, {:.2e})"
 * (lib.rs:551-555), values in Rust LowerExp form (no '+', no exponent padding): shortest round-trip digits when
 * decimal_precision < 0 (write_data's "{:.e}"), else that many fraction digits (write_data_with_precision). */
int orc_write_data(const char *output_file_name, int64_t n_cells, const double *cell_centroid /*[3n]*/, const double *u,
                   const double *v, const double *w, const double *p, int decimal_precision);
/* Fills at most `capacity` rows, returns the row count of the file in n_read (capacity 0: count only).
 * ORC_ERR_IO = the reference's Err("could not read data file"). */
int orc_read_data(const char *data_file_path, int64_t capacity, double *u, double *v, double *w, double *p, int64_t *n_read);
/* "{centroid}\t({9 velocity-gradient entries, row-major, each followed by ', '})\t({3 pressure-gradient entries, same})":
 * the trailing ", " inside the parentheses is the reference's (its strip_suffix result is dropped, io.rs:644,654).
 * Gradients come from the device Green-Gauss kernels (orc_calculate_gradients). */
int orc_write_gradients(const OrcMesh *m, const double *cell_centroid, const double *u, const double *v, const double *w,
                        const double *p, const char *output_file_name, int decimal_precision, const OrcSettings *settings);

/* ---------- linear_algebra::iterative_solve (linear_algebra.rs:144-153) ---------- */
int orc_iterative_solve(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values,
                        const double *b, double *solution_vector /*in/out*/, uint64_t iteration_count,
                        int method /*OrcSolutionMethod*/, double relaxation_factor, double convergence_threshold,
                        int preconditioner /*OrcPreconditionMethod*/);
/* The same for THREE systems that share one sparsity pattern — what solver::solve_steady's three momentum solves are
 * (solver.rs:99-136: a_u, a_v, a_w come from one initialize_momentum_matrix pattern, discretization.rs:450-472).  One
 * column stream and one 24-byte gather per entry serve the three value streams; the fixed iteration count of the BiCGSTAB
 * arm (linear_algebra.rs:255) keeps the systems in lock-step.  method: ORC_SOLVER_BICGSTAB or ORC_SOLVER_MULTIGRID (tree
 * reductions, single GPU).  Each system's result is bit-identical to orc_iterative_solve on that system alone;
 * status_out[k] is system k's verdict (e.g. ORC_ERR_MULTIGRID_DIVERGED), the return value the call's own. */
int orc_iterative_solve3(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *const values[3],
                         const double *const b[3], double *const solution_vectors[3] /*in/out*/, uint64_t iteration_count,
                         int method, double relaxation_factor, double convergence_threshold, int preconditioner,
                         int status_out[3]);
/* Process-wide default of OrcSettings.breakdown_guard for orc_iterative_solve (whose signature has no
 * settings argument, like the reference's).  1 (default) = guard on; 0 = NaN like the reference. */
int orc_set_breakdown_guard(int on);
/* Number of BiCGSTAB solves (process-wide, since the last call with reset != 0) in which the breakdown guard fired, i.e.
 * in which the reference (linear_algebra.rs:255-268, no test at all) would have divided 0/0 and returned NaN — a caller
 * that replaces solver::solve_steady (solver.rs:217-221 "solution diverged") can tell that the guard kept it alive.
 * orc_solve_steady / orc_solver_iterate also leave "breakdown guard fired in N solve(s)" in orc_last_error() when they
 * return ORC_OK after such an iteration. */
int64_t orc_breakdown_guard_events(int reset);
/* Process-wide default of OrcSettings.reduction_order for orc_iterative_solve (OrcReductionOrder). */
int orc_set_reduction_order(int order);
/* sweeps the Jacobi arm executed in the last orc_iterative_solve (linear_algebra.rs:188-217) */
int64_t orc_last_jacobi_sweeps(void);
/* y = A x: the `&CsrMatrix * &DVector` product the reference takes from nalgebra-sparse
 * (linear_algebra.rs:256,260); row sums accumulate in ascending-column order from 0.0, so y is
 * bit-identical to the CPU product.  `reps` > 1 repeats the launch (timing); avg_ms may be NULL. */
int orc_csr_spmv(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, const double *x,
                 double *y, int reps, double *avg_ms);

/* Test hooks for the two private functions under the Multigrid arm (linear_algebra.rs:12-63, :80-84):
 * partner[i] = strongest_unmerged_neighbor of row i (-1 = none) of build_restriction_matrix(Strongest), and
 * a' = (R a) R^T as CSR.  Call with out_col == NULL to get the sizes (out_n_coarse, out_nnz). */
int orc_amg_coarsen(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, int64_t *partner /*[n]*/,
                    int64_t *out_n_coarse, int64_t *out_nnz, int64_t *out_row_ptr, int64_t *out_col, double *out_val,
                    int *rounds);

/* Test hook for the multicolour Gauss-Seidel extension: the distance-1 colouring (Jones-Plassmann, first fit) the device
 * uses for a pattern; rows of one colour share no entry. */
int orc_debug_coloring(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, int32_t *colors /*[n]*/, int32_t *n_colors);

/* ---------- discretization::* ---------- */
/* build_momentum_diffusion_matrix (discretization.rs:39-48): values in pattern order + 3 RHS */
int orc_build_momentum_diffusion_matrix(const OrcMesh *m, int diffusion_scheme, double mu,
                                        double *a_values /*[nnz]*/, double *b_u, double *b_v, double *b_w);
/* initialize_momentum_matrix (discretization.rs:450) */
int orc_initialize_momentum_matrix(const OrcMesh *m, double *a_values /*[nnz]*/);
/* build_momentum_advection_matrices (discretization.rs:134-152).  a_u/a_v/a_w values are in/out:
 * their diagonals on entry are what Rhie-Chow reads (frozen-diagonal semantics).
 * peclet: (avg, min, max) as returned by the reference (:355). */
int orc_build_momentum_advection_matrices(const OrcMesh *m, double *a_u_values, double *a_v_values, double *a_w_values,
                                          double *b_u, double *b_v, double *b_w, const double *a_di_values,
                                          const double *u, const double *v, const double *w, const double *p,
                                          const OrcSettings *settings, double rho, double peclet[3]);
/* build_pressure_correction_matrices (discretization.rs:359-370) -> LinearSystem{a,b} */
int orc_build_pressure_correction_matrices(const OrcMesh *m, const double *u, const double *v, const double *w,
                                           const double *p, const double *a_u_values, const double *a_v_values,
                                           const double *a_w_values, const OrcSettings *settings, double rho,
                                           double *a_values /*[nnz]*/, double *b /*[n]*/);

/* ---------- solver::* ---------- */
/* calculate_pressure_gradient / calculate_velocity_gradient (solver.rs:774-950) evaluated for every cell with
 * settings->gradient_reconstruction — GreenGauss(CellBased) or LeastSquares: grad_p[3n], grad_u[9n] (row = velocity
 * component).  ORC_ERR_SINGULAR_MATRIX where the reference's try_inverse().unwrap() panics. */
int orc_calculate_gradients(const OrcMesh *m, const double *u, const double *v, const double *w, const double *p,
                            const OrcSettings *settings, double *grad_p, double *grad_u);
/* check_boundary_conditions (solver.rs:710-772): constraint_type 0 PressureOnly, 1 VelocityOnly, 2 Hybrid;
 * ORC_ERR_NO_BOUNDARY_CONDITIONS for its "You must set boundary conditions." panic */
int orc_check_boundary_conditions(const OrcMesh *m, int *constraint_type);
/* initialize_pressure_field (solver.rs:414-509): Laplace system on the mesh pattern, 10 Jacobi sweeps; p in/out */
int orc_initialize_pressure_field(const OrcMesh *m, double *p /*[n]*/);
/* initialize_flow (solver.rs:246-352): pressure initialisation, UD / LinearWeighted momentum assembly at zero velocity,
 * then the six-step diffusion -> advection ramp of Jacobi-preconditioned BiCGSTAB solves (iteration_count each).
 * settings may be NULL; only q1_compat and breakdown_guard are read from it.  Outputs u, v, w, p [n]. */
int orc_initialize_flow(const OrcMesh *m, double mu, double rho, uint64_t iteration_count, const OrcSettings *settings,
                        double *u, double *v, double *w, double *p);
/* initialize_velocity_field (solver.rs:511-696): potential-flow system for psi (velocity inlets as sources, a pressure
 * outlet as psi = 0), ten Jacobi-preconditioned BiCGSTAB iterations, then u, v, w = least-squares gradient of psi over
 * the interior neighbours with all-zero columns dropped.  settings may be NULL (q1_compat, breakdown_guard and
 * reduction_order are read).  psi [n] is optional: the field the reference writes to ./examples/psi.csv. */
int orc_initialize_velocity_field(const OrcMesh *m, const OrcSettings *settings, double *u, double *v, double *w, double *psi);
/* initialize_flow_new (solver.rs:354-410): PressureOnly | Hybrid -> initialize_pressure_field, VelocityOnly ->
 * initialize_velocity_field */
int orc_initialize_flow_new(const OrcMesh *m, double mu, double rho, uint64_t iteration_count,
                            double *u, double *v, double *w, double *p);
/* solve_steady (solver.rs:26-37). report_cb is called every reporting_interval iterations with
 * what the reference prints (solver.rs:209-216): iteration, mean u/v/w, Peclet avg/min/max,
 * velocity- and pressure-correction norms, ms/iter.  May be NULL. */
typedef void (*OrcReportFn)(uint64_t iteration, const double mean_velocity[3], const double peclet[3],
                            double velocity_correction, double pressure_correction, double ms_per_iter, void *user);
int orc_solve_steady(OrcMesh *m, double *u, double *v, double *w, double *p, const OrcSettings *settings,
                     double rho, double mu, uint64_t iteration_count, uint64_t reporting_interval,
                     OrcReportFn report_cb, void *user);

/* ---------- device-resident solver (what orc_solve_steady is made of) ---------- */
OrcSolver *orc_solver_create(OrcMesh *m, const OrcSettings *settings, double rho, double mu, int *status); /* solver.rs:39-49 */
void orc_solver_destroy(OrcSolver *s);
int orc_solver_set_fields(OrcSolver *s, const double *u, const double *v, const double *w, const double *p);
int orc_solver_get_fields(OrcSolver *s, double *u, double *v, double *w, double *p);
/* runs `iterations` SIMPLE iterations (solver.rs:60-222); report, if not NULL, receives 8 doubles per
 * iteration: u_avg, v_avg, w_avg, peclet_avg, peclet_min, peclet_max, vel_corr, p_corr */
int orc_solver_iterate(OrcSolver *s, uint64_t iterations, double *report);
/* Device-side copy of the state a SIMPLE iteration starts from (u, v, w, p and the momentum diagonals that Rhie-Chow
 * reads, solver.rs:1068-1081) and its restoration (asynchronous device-to-device copies on the library stream):
 * bench.py times the same iteration repeatedly, so that every step does identical work. */
int orc_solver_snapshot(OrcSolver *s);
int orc_solver_restore(OrcSolver *s);
/* individual phases, for the parity tests: matrices come back in pattern order */
int orc_solver_assemble_momentum(OrcSolver *s, double *a_u, double *a_v, double *a_w, double *b_u, double *b_v, double *b_w, double peclet[3]);
int orc_solver_assemble_pressure(OrcSolver *s, double *a_p, double *b_p);

/* ---------- measurement hooks (bench.py): HIP-event timed launches of single kernels ---------- */
/* y = A x with the momentum matrix a_u of the solver, `reps` launches; returns average ms per launch */
int orc_bench_spmv(OrcSolver *s, int reps, double *avg_ms, double *checksum);
/* one BiCGSTAB iteration body (linear_algebra.rs:255-268) repeated `reps` times on a_u */
int orc_bench_bicgstab_iteration(OrcSolver *s, int reps, double *avg_ms);
/* The Multigrid hierarchy of the momentum system a_u as the solver builds it (Jacobi-scaled operator, levels 0..3):
 * per level rows, stored non-zeros, padded SELL-64 entries and the HIP-event average of `reps` products y = A x.
 * Arrays hold up to 4 entries; *n_levels receives the count (level 0 = the mesh-pattern matrix). */
/* bench.py's headline roofline: the level-0 products of a_u as the solver's BiCGSTAB iterations launch them inside the
 * Multigrid arm (two Jacobi scalings, reduction epilogues), timed with HIP events on the library stream.  avg_ms[0], [1]: one
 * system per launch (nu = A p with sum(nu); t = A s with t.s, t.t); avg_ms[2], [3]: u, v, w in one launch (0 when unsupported). */
int orc_bench_inloop_products(OrcSolver *s, int reps, double avg_ms[4]);
/* "<narrow>, <scaled>": the last two template arguments of the kernels orc_bench_inloop_products has just launched
 * (spmv_uniform_k<Epi, false, true, narrow, scaled>), so that bench.py names the kernel as a kernel trace does */
const char *orc_bench_inloop_variant(void);
/* one multicolour Gauss-Seidel sweep over a_u (extension; BASELINE configs[2]): n_colors launches of the colour-sorted
 * kernel per sweep, average ms per sweep over `reps` sweeps */
int orc_bench_gs_sweep(OrcSolver *s, int reps, double *avg_ms, int *n_colors);
/* [r04] the preconditioner application x^ = M^-1 b as the slot-space GS-BiCGSTAB launches it (one sweep from zero without a zero fill,
 * n_colors launches): avg_ms[0] one system (the p' solve), avg_ms[1] the u, v, w momentum systems per launch */
int orc_bench_gs_sweep0(OrcSolver *s, int reps, double avg_ms[2], int *n_colors);
int orc_bench_amg_levels(OrcSolver *s, int reps, int64_t *rows, int64_t *nnz, int64_t *padded, double *avg_ms, int *n_levels);
/* Test hook: how many level-0 products of partitioned operators this thread has run in the overlapped form (interior rows
 * on a second stream beside the halo exchange, rows along the cuts after it) since orc_init. */
long long orc_debug_halo_overlaps(void);
/* Which of the library's streams still hold work — hipStreamQuery, never blocks; meant for a watchdog thread while the calling thread of a
 * solve waits: one line per stream, "<name>[<lane>] priority <p> busy|idle".  Returns the number of busy streams (-1: buffer too short). */
int orc_debug_stream_report(char *buf, int cap);
/* [r04] test hooks.  orc_debug_clamp_partials_grid: the ONE function through which every launcher sizes a grid whose workgroups
 * write per-workgroup partial sums (<= orc_debug_max_partials(), whatever the CU count or a measurement switch asks for); host only.
 * orc_debug_amg_certification: out[0] = aggregations whose asynchronous cascades were followed by the certifying lock-step
 * rounds, out[1] = the rounds those took in total; equal means no certification changed a pairing.  orc_debug_xwin_counters:
 * out[0] = 256-row blocks of coarse operators given an LDS x window description since the last reset, out[1] = of those
 * without a window because it would exceed the cap (ORC_XWIN_CAP <= 5000 entries), out[2] = because the block's column span
 * exceeds the bitmap (ORC_XWIN_BITWORDS <= 8192 words of 32 columns): the products of such blocks gather from global memory. */
/* orc_debug_amg_coarse_product: one level of the Multigrid set-up on (A) — pairing, (R a) R^T, and where the coarse rows are
 * long enough (ORC_SPMV_XWIN_MIN_NNZ, 24 entries per row) the packed mirror with its LDS x windows — then y = Ac x launched as
 * the solves launch it (linear_algebra.rs:82-97: `a_prime * e_prime`); x, y hold ceil(n / 2) doubles.  scaled != 0: the Jacobi
 * scaling 1 / diag materialised into the streamed values, as inside a smoothing solve (:159-166).  Ac itself comes from
 * orc_amg_coarsen (same kernels), so a test can evaluate the same product on the oracle. */
int orc_debug_amg_coarse_product(int64_t n, const int64_t *row_ptr, const int64_t *col_idx, const double *values, int scaled, const double *x,
                                 double *y, int *has_window_mirror);
/* collectives this process has issued since the last reset — halo exchanges (one grouped ncclSend/ncclRecv launch each) and
 * all-reduces, status agreements included: the latency-bound messages of a partitioned SIMPLE iteration */
long long orc_debug_collectives(int reset);
int orc_debug_clamp_partials_grid(long long requested);
int orc_debug_max_partials(void);
int orc_debug_amg_certification(long long out[2], int reset);
int orc_debug_xwin_counters(long long out[3], int reset);
/* [r04] coarse operators of SIBLING systems built by a shared Galerkin pass since the last reset (the u, v, w momentum matrices of a
 * SIMPLE iteration share their pattern; when v's and w's fine pairings verify as u's, ONE symbolic pass carries the three value sets:
 * linear_algebra.rs:80-84 per system, bit-identical).  ORC_AMG_SHARED_GALERKIN=0 switches the shared pass off. */
long long orc_debug_shared_galerkin(int reset);
/* kernel-level timers accumulated inside orc_solver_iterate when enabled: name/ms pairs */
int orc_profile_enable(int on);
int orc_profile_report(char *buf, int64_t buf_len);

/* ---------- synthetic meshes: host-side generator, ORC numbering rules ---------- */
/* Structured hex channel nx*ny*nz cells on [0,lx]x[0,ly]x[0,lz]: interior faces first, then zones
 * INLET(x-min) OUTLET(x-max) PERIODIC_-Z PERIODIC_+Z TOP_WALL BOTTOM_WALL — the zone order of
 * couette_flow_128x64x1.msh; all boundary zones type 3 (wall) like the reference's fixtures.
 * Call with NULL arrays to get sizes. */
int orc_hex_channel_sizes(int64_t nx, int64_t ny, int64_t nz, int64_t *n_cells, int64_t *n_faces, int64_t *n_cell_faces);
int orc_hex_channel_generate(int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                             int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area,
                             double *face_normal, double *face_centroid, double *cell_centroid, double *cell_volume,
                             int64_t *cell_face_ptr, int64_t *cell_faces);
/* writes the same mesh as an ASCII TGRID .msh that ORC's read_mesh (io.rs:32) accepts */
int orc_hex_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz);

/* BASELINE config 5 workload: a box of nx x ny x nz blocks (nx >= 20) cut along x into regions of hexahedra, columns of
 * triangular prisms, Kuhn tetrahedra (six per block) and one-block transition layers of pyramids, conforming
 * throughout; about 3 cells per block, matrix rows of 5 / 6 / 7 entries.  Written as a TGRID .msh with triangular and
 * quadrilateral faces in separate zones (the reference's reader cannot parse mixed sections, io.rs:232); zones FLUID,
 * INLET, OUTLET, WALL, PERIODIC_-Z, PERIODIC_+Z and their "_TRI" twins, boundary zones of type 3 (wall). */
int orc_mixed_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                                int64_t *n_cells, int64_t *n_faces);
/* The same with a region of POLYHEDRAL cells between x = lx/20 and lx/4 (BASELINE config 5: "tet/hex/poly"): blocks
 * alternate between a hexahedron and six pyramids about the block centre, and every pyramid is agglomerated into the
 * hexahedron behind its base — rhombic dodecahedra of 12 planar quadrilateral faces (13 matrix entries per row), with
 * partial cells and left-over pyramids along the region's sides.  Polyhedral CELLS written through triangular and
 * quadrilateral face sections: the subset of TGRID the reference's reader parses correctly (io.rs:232-233 misreads the
 * per-line node count of face_type 0 / 5 sections), no two cells sharing more than one face (discretization.rs:312-322). */
int orc_poly_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                               int64_t *n_cells, int64_t *n_faces);
/* [r05] The same mesh (polyhedra = 0: orc_mixed_channel_write_msh's, != 0: orc_poly_channel_write_msh's) built IN MEMORY: a handle as from
 * orc_read_mesh whose every array equals, bit for bit, what reading the written file gives (same numbering, zones, geometry code) — without
 * the 580 MB temporary file per rank that BASELINE configs[4] cost in r04.  NULL + *status on failure. */
OrcMeshData *orc_mixed_channel_generate(int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz, int polyhedra, int *status);

/* ---------- multi-GPU (one process per GPU, RCCL over xGMI) ---------- */
#define ORC_COMM_ID_BYTES 128
int orc_comm_get_unique_id(unsigned char id[ORC_COMM_ID_BYTES]);          /* rank 0, then broadcast by the host launcher */
int orc_comm_init(const unsigned char id[ORC_COMM_ID_BYTES], int rank, int world_size);
int orc_comm_finalize(void);
/* One-GPU self-check of the RCCL data path (single-rank communicator, rank 0 as its own neighbour): halo exchange of
 * two fields, sum/max all-reduce, status agreement.  Needs an uninitialised communicator. */
int orc_comm_selftest(void);
/* Debug/test: a persistent single-rank RCCL communicator posing as a two-rank world whose only peer is this rank.  A
 * partitioned mesh whose halo plan names peer 0 everywhere then runs the whole partitioned path (RCCL halos, all-reduces,
 * status agreement, the products that overlap their halo exchange) on ONE GPU; the run is self-coupled (ghost values are
 * the rank's own cells), so it is compared with itself under other switches.  Ended by orc_comm_finalize. */
int orc_comm_init_self_loop(void);
/* Debug transport for tests where ranks share one GPU (RCCL refuses duplicate devices): halo exchange and all-reduce
 * are staged through host memory and carried by the caller's callbacks (e.g. torch.distributed/gloo).
 * exchange_fn: void(int n_peers, const int* peers, const double* send, const int64_t* send_off, const int64_t* send_cnt,
 *                   double* recv, const int64_t* recv_off, const int64_t* recv_cnt, void* user)
 * allreduce_fn: void(double* values, int n, int op (0 sum, 1 max), void* user).  Call orc_comm_init(NULL, rank, world) first. */
int orc_comm_set_host_transport(void *exchange_fn, void *allreduce_fn, void *user);

#ifdef __cplusplus
}
#endif
#endif /* ORC_AMD_H */
