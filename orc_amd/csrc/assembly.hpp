// assembly.hpp — device mesh (SoA image of mesh::Mesh) and the SIMPLE-iteration state.
#pragma once
#include <memory>

#include "linalg.hpp"

namespace orc {

// Device pointers of the mesh, passed to kernels by value.  AoS `Vec<Face>` / `Vec<Cell>` with heap
// adjacency (mesh.rs:140-187) becomes structure-of-arrays so face- and cell-parallel kernels read
// coalesced; the zone HashMap lookup (solver.rs:1113) becomes a per-face zone index into three
// tiny tables.
struct MeshDev {
    int64_t n_cells = 0, n_faces = 0;  // n_cells = owned + ghost cells (array length / SoA stride)
    int64_t n_own = 0;                 // cells this rank assembles and solves: [0, n_own)
    int32_t n_zones = 0;
    const int32_t *c0 = nullptr, *c1 = nullptr, *fzone = nullptr;
    const double *area = nullptr, *nx = nullptr, *ny = nullptr, *nz = nullptr;
    const double *fcx = nullptr, *fcy = nullptr, *fcz = nullptr;
    const double *ccx = nullptr, *ccy = nullptr, *ccz = nullptr, *vol = nullptr;
    const int32_t *cfp = nullptr;    // [n+1] cell -> face list (Cell.face_indices, ascending face id)
    const int32_t *cf = nullptr;     // face ids
    const int32_t *cfpos = nullptr;  // SELL element offset of A(cell, neighbour) for that face, -1 on boundary faces
    const int32_t *ztype = nullptr;  // OrcFaceConditionType per zone
    const double *zscal = nullptr;   // FaceZone.scalar_value
    const double *zvec = nullptr;    // FaceZone.vector_value [3Z]
};

}  // namespace orc

struct OrcMesh {
    int64_t n_cells = 0, n_faces = 0, n_cell_faces = 0;
    int64_t n_own = 0;         // == n_cells on a single GPU
    int64_t n_global = 0;      // cells of the whole mesh (report averages, solver.rs:206-208)
    orc::HaloPlan halo;        // empty on a single GPU
    int32_t n_zones = 0;
    orc::DevBuf<int32_t> c0, c1, fzone, cfp, cf, cfpos, ztype;
    orc::DevBuf<double> area, nx, ny, nz, fcx, fcy, fcz, ccx, ccy, ccz, vol, zscal, zvec;
    orc::SellMatrix pat;  // shared sparsity of a_di, a_u, a_v, a_w, A_p
    std::vector<int64_t> h_row_ptr, h_col;  // the same pattern in CSR (ORC order) for the C ABI
    std::vector<int64_t> h_global_ids;      // orc_mesh_create_reordered: ORC index of every internal cell (empty = identity);
                                            // orc_solver_set_fields / get_fields / orc_solve_steady permute through it
    orc::MeshDev dev() const;
};

namespace orc {

struct Fields {
    double *u, *v, *w, *p;
};

// Everything solve_steady allocates before its loop (solver.rs:39-49) plus per-iteration scratch.
struct SolverState {
    OrcMesh *mesh = nullptr;
    OrcSettings settings;
    double rho = 0., mu = 0.;
    int64_t n = 0;      // local array length (owned + ghost)
    int64_t n_own = 0;  // rows
    DevBuf<double> u, v, w, p, p_prime;
    DevBuf<double> a_di, a_u, a_v, a_w, a_p;            // SELL value arrays (mesh pattern)
    DevBuf<double> b_u_di, b_v_di, b_w_di, b_u, b_v, b_w, b_p;
    DevBuf<double> du, dv, dw;                           // a_{u,v,w}.get(i,i): what Rhie-Chow reads
    // frozen_diagonals = 0 (the reference's in-place reads): last iteration's diagonals, per-cell Peclet terms and
    // the level schedule of the cell order (level_cells[level_ptr[l] .. level_ptr[l+1]) = cells of level l, ascending)
    DevBuf<double> du_old, dv_old, dw_old, pe;
    DevBuf<int32_t> level_cells;
    std::vector<int64_t> level_ptr;
    DevBuf<double> gp;                                   // grad p  [3][n]
    DevBuf<double> gu;                                   // grad U  [9][n]  (TVD only)
    DevBuf<double> flux, pf, coef;                       // per face: outward (from c0) flux, face pressure, p' coefficient
    DevBuf<double> partials, scal;
    DevBuf<int> dev_status;
    Arena arena;
    SolveStats stats;
    SiblingPairing sibling;  // u's pairing of this iteration as the starting state of v's and w's (linalg.hpp)
    // The u, v and w systems of an iteration are independent (solver.rs:99-136 solves them one after the other, none
    // reads another's result): each gets its own stream, arena and host thread, so the latency-bound set-up rounds of
    // one hierarchy overlap the bandwidth-bound products of another.  Same kernels, same order per system: same bits.
    struct Lane {
        hipStream_t stream = nullptr;
        Arena arena;
        SolveStats stats;
        SolveSide side;   // second stream of the lane's Multigrid solves (set-up beside smoothing, linalg.hpp)
        Arena side_arena;
        Arena scratch_arena;  // transient storage of a hierarchy set-up ahead of its solve (multigrid_prepare_dev's `scratch`)
        AmgHierarchy hierarchy;        // partitioned runs: built by the lane while the library stream does the level-0 work
        hipEvent_t level0_done = nullptr;
    } lanes[3];
    // The three momentum systems solved in lock-step on their shared pattern (MatView3, linalg.hpp): one column stream and one
    // 24-byte gather per entry for three value streams on level 0 and, when the pairings coincide, on level 1.  Uses the
    // lanes' streams and arenas for the hierarchy set-ups and the per-system coarse levels.  ORC_TRIPLE_MOMENTUM=0: off.
    TripleLane triple[3];
    bool triple_momentum = true;
    // The pressure-correction matrix depends on the momentum diagonals and the geometry only (discretization.rs:401-438;
    // the new velocities enter its RHS), so its Multigrid hierarchy is built on a stream of its own while the momentum
    // systems are being solved, and the p' solve finds it ready.
    AmgHierarchy p_hierarchy;
    Arena hier_arena, hier_scratch;
    hipStream_t prep_stream = nullptr;
    bool early_p_hierarchy = true;
    bool p_scratch_shared = false;  // this iteration's p' set-up borrows lane 0's scratch arena (it starts when the momentum set-ups are through)
    bool sibling_pairing = true;  // v and w take u's fine-level pairing when it verifies as theirs (ORC_AMG_SIBLING=0: off)
    SolveSide side;       // the same for the solves on the library stream (p', or all four when the lanes are off)
    Arena side_arena;
    bool two_stream_multigrid = true;
    bool concurrent_momentum = true;
    ~SolverState();
    uint64_t iterations_done = 0;
    // orc_solver_snapshot / orc_solver_restore: the state one SIMPLE iteration starts from (u, v, w, p and the momentum
    // diagonals Rhie-Chow reads), kept device-side so that a benchmark can time the SAME iteration repeatedly
    DevBuf<double> snap[7];
    uint64_t snap_iterations = 0;
    bool has_snapshot = false;
};

int mesh_upload(OrcMesh &m, int64_t n_own, int64_t n_cells, int64_t n_faces, int32_t n_zones, const int64_t *face_c0, const int64_t *face_c1,
                const int32_t *face_zone, const double *face_area, const double *face_normal, const double *face_centroid,
                const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr, const int64_t *cell_faces,
                const int32_t *zone_type, const double *zone_scalar, const double *zone_vector);

int solver_init(SolverState &s, OrcMesh *m, const OrcSettings *settings, double rho, double mu);
// kernels (all asynchronous on the library stream)
int k_diffusion(SolverState &s);                 // K14  discretization.rs:39-131
int k_init_momentum(SolverState &s);             // discretization.rs:450-472
int k_gradients(SolverState &s, bool need_gu);   // K9   solver.rs:774-802, 874-902
int k_face_flux(SolverState &s, bool with_pf);   // K10  solver.rs:1007-1150
int k_momentum(SolverState &s, double *peclet_host /*3, may be null*/);  // K11  discretization.rs:134-356
int k_pressure_correction(SolverState &s);       // K12  discretization.rs:359-448
int k_apply_correction(SolverState &s, double *sums_host /*5: p'^2, |dU|^2, sum u, sum v, sum w*/);  // K13 solver.rs:1170-1227
int solver_iterate(SolverState &s, uint64_t iterations, double *report);
int fetch_status(SolverState &s);
int check_boundary_conditions(const OrcMesh &m);       // solver.rs:710-772: 0 PressureOnly, 1 VelocityOnly, 2 Hybrid, < 0 = -status
int initialize_pressure_field_dev(SolverState &s);     // solver.rs:414-509
int initialize_flow_dev(SolverState &s, uint64_t iteration_count);  // solver.rs:246-352
int initialize_velocity_field_dev(SolverState &s);     // solver.rs:511-696 (psi in s.p_prime, velocities in s.u/v/w)
int post_loop_gradients_dev(SolverState &s);           // solver.rs:227-242

}  // namespace orc

struct OrcSolver {
    orc::SolverState st;
};
