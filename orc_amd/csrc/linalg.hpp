// linalg.hpp — device sparse matrix format and the iterative solvers (K1-K8 of SURVEY §2.1).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include "common.hpp"
#include "halo.hpp"

namespace orc {

// SELL-64: rows are grouped in slices of 64 (one wavefront); inside a slice entry k of every row
// is stored contiguously: element (row r, k) lives at slice_ptr[r>>6] + k*64 + (r&63).
// A thread-per-row SpMV therefore reads values and column indices fully coalesced while each row
// is still accumulated in ascending-column order from 0.0 — the summation order of
// nalgebra-sparse's CSR product, which makes y = A x bit-identical to the CPU oracle.
struct SellDev {
    int64_t n = 0;      // rows (owned cells)
    int64_t ncols = 0;  // vector length: rows + ghost columns of a partitioned matrix (== n otherwise)
    int32_t n_slices = 0;
    int32_t ragged = 0;  // padding > 8 % of the stored entries (coarse AMG levels): 1 = long rows, 2 = short rows (< 24 slots per row); picks the product variant
    int64_t padded = 0;  // stored slots (slice_ptr[n_slices]) when the host knows it, else 0
    const int64_t *slice_ptr = nullptr;  // [n_slices+1], element offsets (multiples of 64)
    const int32_t *row_len = nullptr;    // [n]
    const int32_t *col = nullptr;        // [padded]
    const int32_t *diag_pos = nullptr;   // [n] element offset of the stored diagonal, -1 if absent
    // Narrow column image [r03] (host-built patterns: the mesh pattern and user matrices): the columns of one depth of one slice
    // lie within a few tens of each other on a mesh (neighbours of 64 consecutive cells); a pattern whose every slice and depth spans
    // < 65 536 stores them as colbase[slot / 64] + col16[slot]: 2 instead of 4 bytes per entry in the product's stream — the
    // level-0 product is bandwidth-bound at the copy ceiling, bytes are what is left.  Other kernels keep `col`.
    const uint16_t *col16 = nullptr;     // [padded]; present only if EVERY depth of EVERY slice spans < 65 536 (all or nothing: no per-slice branch)
    const int32_t *colbase = nullptr;    // [padded / 64]: smallest column of that depth of that slice
    // [r04] host-built patterns also keep their CSR form as the PATTERN half of a row-contiguous mirror (RowsDev): entry k of row r at
    // rows_base[r >> 6] + rows_intra[r] + k.  The set-up kernels walk single rows (aggregation, Galerkin merge); in SELL a row's entries sit in
    // different cache lines (7 lines of columns + 7 of values for an 84-byte hex row); multigrid_prepare_dev adds the values per solve.
    const long long *rows_base = nullptr;  // [n_slices]
    const int32_t *rows_intra = nullptr;   // [n]
    const int32_t *rows_col = nullptr;     // [nnz]
    const int64_t *csr_row_ptr = nullptr;  // [n + 1] (value export into the mirror)
    int64_t nnz = 0;                       // stored entries when the host knows them (host-built patterns), else 0
};

// Zero-padding mirror of a SELL-64 matrix for the wave-cooperative product (coarse AMG levels, whose rows are ragged:
// 6-47 % of a padded SELL image is padding).  Row r keeps lane r & 63 of slice r >> 6; at depth k only the lanes whose
// row is longer than k own an entry, and those entries are stored back to back in lane order:
//     pos(r, k) = ptr[r >> 6] + sum_{k' < k} count(k') + |{lanes l < (r & 63) : len(l) > k}|
// which the wavefront evaluates with a ballot, a population count and v_mbcnt per depth.  Loads stay contiguous over the
// active lanes, every row is still summed in ascending-column order (bit-identical to the padded product), and the
// 64 rows of a wavefront remain neighbours (no length sorting: the x gathers keep their locality).  A lone thread
// cannot address an entry, so the padded image stays beside it for the set-up kernels that walk single rows.
struct PackedDev {
    const int64_t *ptr = nullptr;  // [n_slices+1] element offsets, multiples of 16 (128-byte aligned values)
    const int32_t *col = nullptr;
    const double *val = nullptr;
    int64_t total = 0;             // stored entries (ptr[n_slices]) when the host knows it, else 0
};

// LDS-staged x tiles for the packed mirror (BASELINE north star: "LDS-staged x-vector tiles").  Rows are taken in
// blocks of kXWinRows (one workgroup); the distinct columns a block references — its window — are listed once, ascending,
// in wcol, and every entry carries the 16-bit position of its column inside that window.  A product loads the window
// from x into LDS with (mostly contiguous) coalesced reads, then every gather is an LDS read: the per-lane divergent
// global gathers of a coarse level (13 cache lines per wave-instruction, each re-fetched ~100x per product) leave the
// vector-memory pipe, which is what bounds those levels once their loads are branch-free.  Entries cost 8 + 2 bytes
// instead of 8 + 4.  Same values, same order, same sums.  A block whose window does not fit (wsize < 0) keeps the
// global gathers.
constexpr int kXWinRows = 256;   // rows per block = 4 slices = one workgroup
constexpr int kXWinCap = 5000;   // window entries per block: 40 KB of LDS, four workgroups per CU (r02: 4080 "for five" — four were resident)
struct XWinDev {
    const int32_t *wcol = nullptr;   // [n_blocks * kXWinCap]
    const int32_t *wsize = nullptr;  // [n_blocks], -1 = no window for this block
    const uint16_t *lidx = nullptr;  // [packed entries] window position of the entry's column
    // [r05] the LDS entries a product of THIS level provides per workgroup (<= kXWinCap): the smallest of a few sizes that leaves <= 1 % of the
    // level's blocks without a window.  The compiled worst case (5 000 entries = 40 KB) held a CU to four workgroups whatever the level needed
    // (the channel's level 2: 2 200 per block); blocks whose window is larger gather from global memory, as blocks without a window always did.
    int32_t cap = kXWinCap;
    // [r05] one workgroup per block: the workgroups' partial sums (2 x the launch's workgroups) and the arrival counter of the launch in flight — one
    // product of a level at a time (a level's products follow each other on one stream; sibling systems have scratch of their own)
    double *fold_scratch = nullptr;
    unsigned *fold_counter = nullptr;
};

// Row-contiguous mirror (the Galerkin product's scratch rows, kept alive with their level): entry k of row r sits at
// slice_base[r >> 6] + intra_off[r] + k.  The set-up kernels that walk single, scattered rows (aggregation rounds) read
// a row's columns and values from two or three cache lines here instead of one line per entry in the SELL image.
struct RowsDev {
    const long long *slice_base = nullptr;  // [n_slices]
    const int32_t *intra_off = nullptr;     // [n]
    const int32_t *col = nullptr;
    const double *val = nullptr;
};

// A matrix seen through up to two explicit left (row) scalings: value(i,j) = s2[i]*(s1[i]*val).
// This is how the reference's Jacobi preconditioner `p_inv * a` (linear_algebra.rs:159-166) and
// its nested re-application are evaluated without materialising a_tmp.
struct MatView {
    SellDev P;
    const double *val = nullptr;
    const double *s1 = nullptr;
    const double *s2 = nullptr;
    int nt = 0;             // set by the launch: the matrix streams are loaded with the non-temporal hint (launch_spmv)
    PackedDev pk;           // optional packed mirror (same pattern, same values): what the product streams when present
    XWinDev xw;             // optional LDS x-window description of the packed mirror
    RowsDev rows;           // optional row-contiguous mirror for single-row walks
    bool symmetric = true;  // structural symmetry of the pattern (aggregation fast path)
    bool persistent_pattern = false;  // the pattern outlives the solve (mesh pattern): derived data such as a colouring may be cached
    HaloPlan *halo = nullptr;  // partitioned level-0 operator: x's ghost entries are refreshed before every product,
                               // reductions are summed over ranks; coarse AMG levels are per-rank (halo == nullptr)
    // A product restricted to the slices [slice_lo, slice_hi) (slice_hi < 0: all of them) whose partial sums go to
    // partials[q * part_stride + part_base + blockIdx.x] (part_stride 0: the grid size): the pieces of a product that
    // overlaps its halo exchange (launch_spmv) are folded together by one reduce_partials.
    int32_t slice_lo = 0, slice_hi = -1, part_stride = 0, part_base = 0;
};

// Three systems on ONE sparsity pattern — the u, v and w momentum matrices of a SIMPLE iteration (they share the mesh
// pattern, discretization.rs:450-472, and on the first coarse level the pattern of (R a) R^T whenever their fine-level
// pairings coincide).  One column stream serves three value streams, and the three vectors are INTERLEAVED
// (x3[3 * i + s] = entry i of system s) so that one 24-byte gather fetches what three 8-byte gathers did: per entry
// 28 bytes and five vector-memory instructions instead of 36 and nine.  Every row of every system is still summed in
// ascending-column order from 0.0, every reduction keeps the thread -> element map and the fold of the one-system
// kernels, so each system's results are bit-identical to its own one-system solve.
struct MatView3 {
    SellDev P;
    const double *val[3] = {nullptr, nullptr, nullptr};
    const double *s1 = nullptr, *s2 = nullptr;  // row scalings, interleaved [3 n] (MatView::s1 / s2 per system)
    bool mesh_pattern = false;                  // level 0 (kernel-name tag only, see spmv_uniform_k's kMesh)
    int nt = 0;                                 // set by the launch (MatView::nt)
    // [r04] partitioned level-0 operator (MatView::halo): the interleaved input vector holds 3 * P.ncols doubles, its ghost entries are
    // refreshed by ONE exchange of 24 bytes per cell before every product, the reductions are summed over the ranks in one all-reduce
    // of 3 (or 6) scalars — a third of the collectives of three one-system solves (solver.rs:99-136 on a cell-partitioned mesh)
    HaloPlan *halo = nullptr;
    // the pieces of a product that overlaps its halo exchange (MatView::slice_lo / slice_hi / part_stride / part_base)
    int32_t slice_lo = 0, slice_hi = -1, part_stride = 0, part_base = 0;
};

// Owning pattern (built on the host from CSR, e.g. the mesh pattern or a user matrix).
struct SellMatrix {
    int64_t n = 0, ncols = 0, nnz = 0, padded = 0;
    int32_t n_slices = 0;
    bool symmetric = true;
    int ragged = 0;
    DevBuf<int64_t> slice_ptr;
    DevBuf<int32_t> row_len, col, diag_pos, colbase;
    DevBuf<uint16_t> col16;
    DevBuf<int64_t> csr_row_ptr;  // for value import/export in CSR (ORC) order
    DevBuf<long long> rows_base;  // pattern half of the level-0 row mirror (SellDev)
    DevBuf<int32_t> rows_intra, rows_col;
    SellDev dev() const {
        SellDev d;
        d.n = n; d.ncols = ncols; d.n_slices = n_slices; d.ragged = ragged; d.padded = padded; d.slice_ptr = slice_ptr.p; d.row_len = row_len.p; d.col = col.p; d.diag_pos = diag_pos.p;
        d.col16 = col16.p; d.colbase = colbase.p;
        d.rows_base = rows_base.p; d.rows_intra = rows_intra.p; d.rows_col = rows_col.p; d.csr_row_ptr = csr_row_ptr.p; d.nnz = nnz;
        return d;
    }
};

int sell_from_csr_host(int64_t n, int64_t ncols, const int64_t *row_ptr, const int64_t *col, SellMatrix &out);
// values: CSR order (device) <-> SELL order (device)
int sell_import_values(const SellMatrix &m, const double *csr_vals_dev, double *sell_vals_dev);
int sell_export_values(const SellMatrix &m, const double *sell_vals_dev, double *csr_vals_dev);
int sell_rows_values_dev(const SellDev &P, const double *sell_vals_dev, double *rows_vals_dev);  // -> the value half of the level-0 row mirror

// Optional second stream of a solve (Multigrid arm): the hierarchy set-up of level l+1 — host-synchronised rounds,
// latency-bound — needs only the level-l matrix, while the smoothing solve of level l — 100 products, bandwidth-bound,
// no host interaction — needs the matrix and the restricted residual.  Set-up work stays on ctx().stream, everything
// that touches vectors goes to `stream` with its temporaries in `arena`; two events order the hand-overs.
struct SolveSide {
    hipStream_t stream = nullptr;
    Arena *arena = nullptr;
    hipEvent_t ev_setup = nullptr, ev_solve = nullptr;
};

// A Multigrid hierarchy built ahead of the solve that uses it (multigrid_prepare_dev): per level the pairing of the
// finer rows and the Galerkin operator.  It is a function of the matrix values alone, so it can be built as soon as
// they exist — for the pressure correction that is before the momentum solves, whose results only enter its RHS.
struct AmgHierarchy {
    struct Level {
        int *choice = nullptr, *chooser = nullptr;  // partner of / chosen-by, per row of the finer level
        SellDev P;                                  // coarse operator
        double *val = nullptr;
        PackedDev pk;
        XWinDev xw;
        RowsDev rows;
        int64_t n = 0, padded = 0;
        int rounds = 0;
    } level[4];
    int n_levels = 0;
    int64_t n_fine = 0;
};

// The u, v and w momentum systems of one SIMPLE iteration share their pattern and differ in a few coefficients (the TVD
// limiter per component), and their greedy pairings come out the same for all but a handful of rows (measured: 100.00 %
// on level 0, 99.9 % on level 1 of the 400 x 40 x 40 channel).  The pairing is a unique fixed point that the device reaches
// from ANY starting state and certifies, so the first system to finish a level (the leader, u) publishes its pairing and
// the others start from it instead of from the unconstrained arg-min: their first lock-step round then finds almost
// nothing to change.  Nothing is carried from one SIMPLE iteration to the next.
struct SiblingPairing {
    static constexpr int kLevels = 8;
    DevBuf<int> buf[kLevels];          // the leader's pairing per level (a copy: the leader's own array is arena memory)
    int64_t n[kLevels] = {0};
    hipEvent_t ready[kLevels] = {nullptr};
    bool published[kLevels] = {false};
    bool leader_done = true;           // no leader at work: followers do not wait
    std::mutex mu;
    std::condition_variable cv;
    ~SiblingPairing();
    void begin(bool leader_will_run);  // before the systems of an iteration start
    int publish(int level, const int *choice, int64_t rows, hipStream_t stream);  // leader, after its aggregation of `level`
    const int *wait(int level, int64_t rows, hipStream_t stream);                 // follower; null = start from scratch
    void finish();                     // leader, on every exit from its solve

    // [r04] One Galerkin pass for the systems that share the fine pairing (multigrid_prepare_dev; amg.hip: GalerkinSibling).  A follower
    // OFFERS its fine view and arenas and blocks; the leader, through with its fine aggregation, COLLECTS the offers, checks its pairing
    // against every offered matrix (agg_verify_k), builds the first coarse operators of all that agree in one pass and ANSWERS; a follower
    // whose answer is "adopted" continues with the second level, any other one falls back to wait() and its own product.  All hand-overs
    // go through mu / cv, so the leader may allocate in a blocked follower's arenas.
    struct Offer {
        bool made = false, ok = false;          // made: the follower has spoken (ok = false: it withdrew)
        const MatView *view = nullptr;
        Arena *arena = nullptr, *rows_arena = nullptr;
        hipEvent_t view_ready = nullptr;        // recorded on the follower's stream once its view (scalings) is complete
        bool answered = false, adopted = false;
        void *level = nullptr;                  // CoarseLevel the leader fills (owned by the follower's stack frame)
    } offer[2];
    int expected_offers = 0;                    // followers that WILL speak (set before the leader starts: set_expected)
    hipEvent_t ops_ready = nullptr;             // recorded on the leader's stream behind the shared pass
    const int *lead_choice = nullptr, *lead_chooser = nullptr;  // the leader's own arrays (alive as long as its hierarchy)
    void set_expected(int n);
    int make_offer(int slot, const MatView *view, Arena *arena, Arena *rows_arena, void *level, hipStream_t stream);  // follower
    void withdraw(int slot);                                                                                          // follower, instead of an offer
    bool wait_answer(int slot, hipStream_t stream);  // follower: true = its level was built by the leader (stream waits for it)
    int collect_offers(Offer *out[2]);               // leader: blocks until every expected follower has spoken; returns how many offered
    int answer(const bool adopted[2], const int *choice, const int *chooser, hipStream_t stream);  // leader
};

struct SolveStats {
    SiblingPairing *sibling = nullptr;  // optional, owned by the caller (shared by the momentum systems of an iteration)
    int sibling_role = 0;               // 1 = leader (publishes), 2 = follower (starts from the leader's pairing)
    SolveSide *side = nullptr;  // optional, owned by the caller
    const AmgHierarchy *hierarchy = nullptr;  // optional: a hierarchy prepared for exactly this matrix
    int64_t jacobi_sweeps = 0;
    int amg_levels = 0;
    int64_t amg_rows[8] = {0};
    int64_t amg_nnz[8] = {0};
    int amg_rounds[8] = {0};
};

// linear_algebra::iterative_solve on device-resident data (linear_algebra.rs:144-299).
// b, x: device vectors of A.P.n doubles; x is in/out.  Temporaries come from `arena`.
// Returns OrcStatus (after synchronising the stream once to fetch the sticky status word).
int iterative_solve_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, int method,
                        double relaxation_factor, double convergence_threshold, int preconditioner, Arena &arena,
                        SolveStats *stats);

// Builds the hierarchy the Multigrid arm of iterative_solve_dev(A, ..., preconditioner) would build for itself (same
// kernels, same results); all of its memory comes from `arena` and stays valid until the caller releases it.
// `scratch` (optional): a second arena for what is dead once a level is complete (aggregation work lists, symbolic bounds, the
// Galerkin product's scratch rows); only what the solve reads stays in `arena` — about 10 GB instead of 23 GB at 10.24 M rows.
int multigrid_prepare_dev(const MatView &A, int preconditioner, Arena &arena, AmgHierarchy &H, SiblingPairing *sibling = nullptr,
                          int sibling_role = 0, Arena *scratch = nullptr);
// dinv[i] = 1 / A(i,i) through the view (the Jacobi preconditioner's p_inv, linear_algebra.rs:159-166)
int diag_inverse_dev(const MatView &A, double *dinv);
// out = 0 + s * b  (p_inv * b, linear_algebra.rs:165)
int scale_vec_dev(const double *s, const double *b, double *out, int64_t n);
// The rank-local remainder of the Multigrid arm once its level-0 smoothing and residual are done (:284-295): the
// V-recursion from level 1 on a prepared hierarchy (stats->hierarchy), x += correction.  No rank-to-rank traffic, no
// host synchronisation; *dev_status collects ORC_ERR_MULTIGRID_DIVERGED.
int multigrid_coarse_part_dev(const MatView &A, const double *r, double *x, uint64_t iteration_count, double relaxation_factor,
                              double convergence_threshold, int preconditioner, Arena &arena, SolveStats *stats, int *dev_status);

// y = A x (K1).  Exposed for bench/tests.
int spmv_dev(const MatView &A, const double *x, double *y);
// one BiCGSTAB iteration body repeated reps times (bench)
int bench_bicgstab_dev(const MatView &A, const double *b, double *x, int reps, Arena &arena, float *ms);

// r = b - A x ;  out[0] = sum((b - A x)^2) (NaN iff the reference's .norm() is NaN)
int residual_dev(const MatView &A, const double *b, const double *x, double *r);
// r_scratch (n doubles, optional): where the residual is kept when the norm is to be summed in the reference's order
int residual_norm2_dev(const MatView &A, const double *b, const double *x, double *partials, double *out, double *r_scratch = nullptr);
// out[0] = a . b in nalgebra's dotx association (a == nullptr: the all-ones vector); one wavefront, verification mode only
int dot_reference(const double *a, const double *b, int64_t n, double *out, const double *skip_flags);
// out[0] = the left-to-right sum of a (nalgebra's sum(): solver.rs:206-208; the running sums of the reference's cell loops); verification mode
int sum_reference(const double *a, int64_t n, double *out);

// ---- three systems in lock-step (MatView3): interleaved vectors of 3 n doubles
int interleave3_dev(const double *a, const double *b, const double *c, double *out3, int64_t n);
int deinterleave3_dev(const double *in3, double *a, double *b, double *c, int64_t n);
int diag_inverse3_dev(const MatView3 &A, double *dinv3);
int spmv3_dev(const MatView3 &A, const double *x3, double *y3);
int residual3_dev(const MatView3 &A, const double *b3, double *x3, double *r3);
// out[s] = sum((b - A x)^2) of system s
// (a partitioned operator: x3 holds 3 * P.ncols doubles and its ghost entries are refreshed first; out3 is summed over the ranks)
int residual_norm2_3_dev(const MatView3 &A, const double *b3, double *x3, double *partials /* 3 * kMaxPartials */, double *out3);
// iterative_solve's BiCGSTAB arm (linear_algebra.rs:247-269) with its Jacobi preconditioner (:159-167) for the three systems at
// once; fixed iteration count, so the three stay in lock-step; the breakdown guard acts per system.  Tree reductions only.
int bicgstab3_dev(const MatView3 &A, const double *b3, double *x3, uint64_t iteration_count, int preconditioner, Arena &arena);
// bench.py: the level-0 products exactly as a BiCGSTAB iteration launches them (scalings of the view, reduction epilogues):
// ms[0] = nu = A p with sum(nu) (EpiStoreSum), ms[1] = t = A s with t.s, t.t (EpiTs); average per launch over `reps`
int bench_inloop_products_dev(const MatView &A, const double *x, double *y, double *partials, int reps, float ms[2]);
int bench_inloop_products3_dev(const MatView3 &A, const double *x3, double *y3, double *partials, int reps, float ms[2]);
// A BiCGSTAB solve of at least ORC_MATERIALIZE_SCALING (4) iterations evaluates the view's row scalings once, into values of its own
// from `arena`: A.val = s2 * (s1 * val), A.s1 = A.s2 = null — the same numbers the product kernels would form entry by entry, 16
// (one system) / 48 (three systems) fewer bytes per row and product.  No-op when the view has no scaling or no known size.
int matview_stream_nt(const MatView &A);  // the cache policy launch_spmv picks for this view's matrix streams (MatView::nt)
int materialize_scaled_view(MatView &A, uint64_t iteration_count, Arena &arena);
int materialize_scaled_view3(MatView3 &A, uint64_t iteration_count, Arena &arena);
// A matrix seen through one more Jacobi scaling (linear_algebra.rs:159-166), prepared ONCE for several BiCGSTAB solves on it: the
// inverse diagonal and, from ORC_MATERIALIZE_SCALING iterations on, the scaled values.  multigrid_solve's two smoothing solves of a
// level (:87-96, :123-132) scale the same coarse matrix the same way; the second one reuses what the first built — the same
// kernels on the same numbers, so nothing changes in any bit.  Everything is allocated from `arena` in the CALLER's scope.
struct ScaledOperator {
    MatView A;
    const double *dinv = nullptr;
    uint64_t iterations = 0;
};
struct ScaledOperator3 {
    MatView3 A;
    const double *dinv3 = nullptr;
    uint64_t iterations = 0;
};
int jacobi_scaling_prepare_dev(const MatView &A, uint64_t iteration_count, Arena &arena, ScaledOperator &S);
int jacobi_scaling_prepare3_dev(const MatView3 &A, uint64_t iteration_count, Arena &arena, ScaledOperator3 &S);
// iterative_solve(A, b, x, S.iterations, BiCGSTAB, Jacobi) on the prepared operator: b is scaled here (:165), the rest is :247-269
int bicgstab_scaled_dev(const ScaledOperator &S, const double *b, double *x, Arena &arena);
int bicgstab3_scaled_dev(const ScaledOperator3 &S, const double *b3, double *x3, Arena &arena);
// is the triple path usable in the calling context (tree reductions; since r04 also on a partitioned mesh)?
bool triple_supported();

// One system's share of a three-system Multigrid solve: the streams its hierarchy set-up and its per-system coarse levels
// run on, their arenas, and the hierarchy itself (owned by the caller so that it outlives the call's arenas' unwinding).
struct TripleLane {
    hipStream_t setup_stream = nullptr, solve_stream = nullptr;
    Arena *hier_arena = nullptr, *vec_arena = nullptr;
    Arena *scratch_arena = nullptr;  // optional: transient set-up storage (multigrid_prepare_dev's `scratch`)
    AmgHierarchy hierarchy;
    SolveStats stats;
    bool symmetric = true;
};
// The Multigrid arm (linear_algebra.rs:270-296, BiCGSTAB smoother) for three systems on one pattern: b[k], x[k] are the
// systems' own contiguous vectors (x in/out); status_out[k] = ORC_OK or ORC_ERR_MULTIGRID_DIVERGED per system.
int multigrid_arm3_dev(const MatView3 &A3, const double *const b[3], double *const x[3], uint64_t iteration_count, double relaxation_factor,
                       double convergence_threshold, int preconditioner, Arena &arena, TripleLane lanes[3], SiblingPairing *sibling, int status_out[3],
                       const std::function<void()> &on_hierarchies_built = nullptr);  // called once, from the calling thread, when the three set-ups are through

// [r04] multicolour-GS-preconditioned BiCGSTAB (extension, gs.hip) for the three momentum systems in lock-step on the mesh pattern
// (BASELINE configs[2]); b[k], x[k] in row order, x in / out; single GPU.  gs_slot_space_enabled: ORC_GS_SLOTSPACE != 0.
int gs_bicgstab3_dev(const MatView A[3], const double *const b[3], double *const x[3], uint64_t iteration_count, Arena &arena);
bool gs_slot_space_enabled();

// plain vector helpers used by the SIMPLE driver
int vec_fill(double *x, double v, int64_t n);
int vec_copy(double *dst, const double *src, int64_t n);

}  // namespace orc
