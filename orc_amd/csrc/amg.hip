// amg.hip — the reference's Multigrid arm on the device.
// Reference: src/linear_algebra.rs:12-63 (build_restriction_matrix, Strongest), :66-141
// (multigrid_solve), :270-296 (Multigrid arm of iterative_solve).
//
// What has to be reproduced: a SEQUENTIAL greedy pairing — row i takes the
// most negative off-diagonal a_ij whose column j no earlier row has taken; rows 2k and 2k+1 both
// land in coarse row k, so R is not a partition and carries weights of 2 — followed by the
// Galerkin product (R A) R^T, a fixed-count BiCGSTAB on every level, and a recursion that
// re-solves the restricted right-hand side.
//
// The greedy pairing is a triangular fixed point: choice(i) depends only on choice(k), k < i, within
// two hops.  The device iterates it Jacobi-style from the unconstrained arg-min: each round
// re-evaluates only the rows whose inputs changed (work list + flags), reading the previous round's
// choices, so the only synchronisation is the kernel boundary.  The fixed point is unique, hence
// identical to the sequential result, whatever the number of rounds (<= longest dependency chain,
// ~nx/2 on a structured channel; later rounds touch a few thousand rows).
// The Galerkin product is evaluated per coarse row in LDS with sorted insert-accumulate lists in
// the reference's summation order (i ascending inside T = R A, j ascending inside T R^T), so the
// coarse operator is bit-identical to nalgebra-sparse's.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>

#include <thread>

#include "linalg_kernels.hpp"

namespace orc {


// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ int64_t sell_pos(const SellDev &P, int64_t row, int k) { return P.slice_ptr[row >> 6] + (int64_t)k * 64 + (row & 63); }

struct AggCounters {
    int changed;  // rows whose choice changed in the current sweep
    int rounds;
};

// ---- exact greedy pairing as a fixed point -------------------------------------------------
// State: choice[i] (partner of row i or -1).  A column j is "in combined_cells when row i is visited"
// (linear_algebra.rs:41) iff some row m < i chose it, i.e. iff taken_by[j] = min{m : choice[m] = j} < i.
// Every round rebuilds taken_by from choice (reset + atomicMin scatter), then sweeps: one thread
// walks one 64-row slice IN ORDER, like the reference's loop, seeing its own updates immediately
// (Gauss-Seidel inside the slice) and the other slices' as they land (chaotic relaxation).  A
// sweep that changes nothing has evaluated every row against a taken_by that is exact for the
// final choice, so the state is the unique solution of the triangular system = the sequential
// result.  Chains inside a slice resolve in one sweep; a chain that crosses k slices needs ~k
// sweeps (an x-line of 400 cells: ~7).  Slices that cannot be affected by the last sweep's changes
// are skipped.
__global__ void agg_reset_k(int *__restrict__ taken_by, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) taken_by[i] = 0x7fffffff;
}
__global__ void agg_scatter_k(const int *__restrict__ choice, int *__restrict__ taken_by, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (choice[i] >= 0) atomicMin(&taken_by[choice[i]], (int)i);
}

// arg-min over j != i of a_ij among columns not taken by an earlier row (strict <, first wins: :37-52)
__device__ __forceinline__ int agg_eval_row(const MatView &A, const int *__restrict__ taken_by, int64_t i, bool constrained) {
    const int len = A.P.row_len[i];
    const int64_t base = A.P.slice_ptr[i >> 6] + (i & 63);
    double best = 1.7976931348623157e308;  // Float::MAX
    int bj = -1;
    for (int k = 0; k < len; ++k) {
        const int64_t pos = base + (int64_t)k * 64;
        const int j = A.P.col[pos];
        if (j == i || j >= A.P.n) continue;  // ghost columns (partitioned level 0) are never partners: aggregates stay on the rank
        if (constrained && taken_by[j] < i) continue;
        const double a = view_value(A, i, pos);
        if (a < best) { best = a; bj = j; }
    }
    return bj;
}

__global__ void agg_init_k(MatView A, int *__restrict__ choice) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x) choice[i] = agg_eval_row(A, nullptr, i, false);
}

// Preference lists: a row's order of preference (value ascending, position ascending: the strict <, first-wins scan of linear_algebra.rs:37-52)
// does not change while a pairing is sought; da_first_k keeps every row's kPrefs most preferred columns (-1 = fewer; bit 30 of the last entry =
// the row has more candidates than listed).  Four: r04 measured 8 / 4 / 2 / 3 (lists of three straddle cache lines).
constexpr int kPrefs = 4;
constexpr int kPrefMore = 1 << 30;

// one thread = one slice, rows in ascending order (the fallback of aggregate(): works on any pattern, symmetric or not)
__global__ void agg_sweep_k(MatView A, int *__restrict__ choice, int *__restrict__ taken_by, AggCounters *C) {
    const int64_t n = A.P.n;
    int changed = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < A.P.n_slices; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = s * 64, hi = lo + 64 < n ? lo + 64 : n;
        for (int64_t i = lo; i < hi; ++i) {
            const int old = choice[i];
            const int nv = agg_eval_row(A, taken_by, i, true);
            if (nv == old) continue;
            ++changed;
            choice[i] = nv;
            if (nv >= 0) atomicMin(&taken_by[nv], (int)i);
            // `old` may still be taken by another row; leaving taken_by[old] <= i is only ever too pessimistic for rows > i and is repaired
            // by the next sweep's rebuild, which cannot be skipped because this sweep counted a change.
        }
    }
    if (changed) atomicAdd(&C->changed, changed);
}

__global__ void agg_rotate_k(AggCounters *C, int *snapshot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *snapshot = C->changed;
        C->changed = 0;
        C->rounds += 1;
    }
}

// One atomicAdd per wavefront instead of one per lane: the lanes that want a slot are counted with a ballot, the
// lowest of them reserves the block of slots and every lane takes its rank inside it.  The work lists of a round hold
// tens of thousands of rows; their appends all hit ONE counter, and same-address atomics retire at ~12 ns each on
// this chip — that serialisation, not the row work, was most of a round's 50-60 us.
__device__ __forceinline__ int wave_append_slot(int *counter, bool want) {
    const unsigned long long m = __ballot(want);
    if (!want) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

__global__ void chooser_k(const int *__restrict__ choice, int *__restrict__ chooser, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (choice[i] >= 0) chooser[choice[i]] = (int)i;  // a column is taken at most once (:41, :55)
}

// Row I of R (linear_algebra.rs:53-58 after COO->CSR): up to 4 (fine index, weight) pairs, ascending, duplicates summed.
struct RRow {
    int idx[4];
    double w[4];
    int n;
};
// Every array index below is a compile-time constant (fixed sorting network, unrolled merges), so the row lives in
// registers; a dynamically indexed idx[]/w[] would be spilled to scratch memory.
// (pair0, pair1 = choice[2I], choice[2I + 1], -1 where the fine row does not exist: loaded by the caller, ahead of time)
__device__ __forceinline__ RRow restriction_row_from(int64_t I, int pair0, int pair1) {
    constexpr int kNone = 0x7fffffff;
    int v[4] = {kNone, kNone, kNone, kNone};
    if (pair0 >= 0) { v[0] = (int)(2 * I); v[1] = pair0; }
    if (pair1 >= 0) { v[2] = (int)(2 * I + 1); v[3] = pair1; }
    // sorting network for 4 keys (absent entries sort last)
#define ORC_CSWAP(x, y) { const int lo__ = min(v[x], v[y]), hi__ = max(v[x], v[y]); v[x] = lo__; v[y] = hi__; }
    ORC_CSWAP(0, 1) ORC_CSWAP(2, 3) ORC_CSWAP(0, 2) ORC_CSWAP(1, 3) ORC_CSWAP(1, 2)
#undef ORC_CSWAP
    // merge equal indices: out[m-1] absorbs a repeat (weights 1 -> 2; a fine row can appear at most twice)
    RRow r;
    r.n = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.idx[q] = 0; r.w[q] = 0.; }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const bool present = v[a] != kNone;
        const bool repeat = present && a > 0 && v[a] == v[a > 0 ? a - 1 : 0];
        if (present && !repeat) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == r.n) { r.idx[q] = v[a]; r.w[q] = 1.; }
            r.n++;
        } else if (repeat) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == r.n - 1) r.w[q] += 1.;
        }
    }
    return r;
}
__device__ __forceinline__ RRow restriction_row(const int *__restrict__ choice, int64_t I, int64_t n_fine) {
    const int pair0 = 2 * I < n_fine ? choice[2 * I] : -1;
    const int pair1 = 2 * I + 1 < n_fine ? choice[2 * I + 1] : -1;
    return restriction_row_from(I, pair0, pair1);
}

// r' = R r (:82)
__global__ void restrict_k(const int *__restrict__ choice, int64_t n_fine, int64_t n_coarse, const double *__restrict__ r, double *__restrict__ rc) {
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < n_coarse; I += (int64_t)gridDim.x * blockDim.x) {
        const RRow R = restriction_row(choice, I, n_fine);
        double acc = 0.;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < R.n) acc += R.w[a] * r[R.idx[a]];
        rc[I] = acc;
    }
}

// Row j of R^T: (coarse index, weight) pairs, ascending
__device__ __forceinline__ int rt_row(const int *__restrict__ choice, const int *__restrict__ chooser, int j, int J[2], double W[2]) {
    const int own = choice[j] >= 0 ? (j >> 1) : -1;
    const int oth = chooser[j] >= 0 ? (chooser[j] >> 1) : -1;
    int n = 0;
    if (own >= 0 && oth >= 0) {
        if (own == oth) { J[0] = own; W[0] = 2.; n = 1; }
        else if (own < oth) { J[0] = own; W[0] = 1.; J[1] = oth; W[1] = 1.; n = 2; }
        else { J[0] = oth; W[0] = 1.; J[1] = own; W[1] = 1.; n = 2; }
    } else if (own >= 0) { J[0] = own; W[0] = 1.; n = 1; }
    else if (oth >= 0) { J[0] = oth; W[0] = 1.; n = 1; }
    return n;
}

// out = R^T e (:140); when `add_to` is set: add_to += R^T e (x += multigrid_solve(...), :284)
__global__ void prolong_k(const int *__restrict__ choice, const int *__restrict__ chooser, int64_t n_fine, const double *__restrict__ e,
                          double *__restrict__ out, double *__restrict__ add_to) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_fine; j += (int64_t)gridDim.x * blockDim.x) {
        int J[2];
        double W[2];
        const int n = rt_row(choice, chooser, (int)j, J, W);
        double acc = 0.;
        for (int a = 0; a < n; ++a) acc += W[a] * e[J[a]];
        if (out) out[j] = acc;
        if (add_to) add_to[j] += acc;
    }
}

__global__ void vec_add_k(double *__restrict__ x, const double *__restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] += y[i];
}

// ---- the same for three systems that share a pairing (interleaved vectors, linalg.hpp MatView3): one row of R / R^T per thread,
// applied to the three systems in the one-system order of additions
__global__ void restrict3_k(const int *__restrict__ choice, int64_t n_fine, int64_t n_coarse, const double *__restrict__ r3, double *__restrict__ rc3) {
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < n_coarse; I += (int64_t)gridDim.x * blockDim.x) {
        const RRow R = restriction_row(choice, I, n_fine);
        double a0 = 0., a1 = 0., a2 = 0.;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < R.n) {
                const int64_t e = 3 * (int64_t)R.idx[a];
                a0 += R.w[a] * r3[e];
                a1 += R.w[a] * r3[e + 1];
                a2 += R.w[a] * r3[e + 2];
            }
        rc3[3 * I] = a0; rc3[3 * I + 1] = a1; rc3[3 * I + 2] = a2;
    }
}
// add_to3 += R^T e3
__global__ void prolong3_k(const int *__restrict__ choice, const int *__restrict__ chooser, int64_t n_fine, const double *__restrict__ e3,
                           double *__restrict__ add_to3) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_fine; j += (int64_t)gridDim.x * blockDim.x) {
        int J[2];
        double W[2];
        const int n = rt_row(choice, chooser, (int)j, J, W);
        double a0 = 0., a1 = 0., a2 = 0.;
        for (int a = 0; a < n; ++a) {
            const int64_t e = 3 * (int64_t)J[a];
            a0 += W[a] * e3[e];
            a1 += W[a] * e3[e + 1];
            a2 += W[a] * e3[e + 2];
        }
        add_to3[3 * j] += a0; add_to3[3 * j + 1] += a1; add_to3[3 * j + 2] += a2;
    }
}
// x3[3 i + s] += y_s[i]  (the corrections of the coarser levels, one contiguous vector per system)
__global__ void vec_add3_k(double *__restrict__ x3, const double *__restrict__ y0, const double *__restrict__ y1, const double *__restrict__ y2, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x3[3 * i] += y0[i]; x3[3 * i + 1] += y1[i]; x3[3 * i + 2] += y2[i];
    }
}
// counts the entries in which two int arrays differ (pairings / row lengths of sibling systems)
__global__ void count_diff_k(const int *__restrict__ a, const int *__restrict__ b, int64_t n, int *__restrict__ counter) {
    int d = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(counter, d);
}
__global__ void nan_to_status3_k(const double *__restrict__ value3, int *status3, int code) {
    if (blockIdx.x == 0 && threadIdx.x < 3 && isnan(value3[threadIdx.x])) atomicCAS(status3 + threadIdx.x, 0, code);
}

// ------------------------------------------------------------------ Galerkin product (R A) R^T
// (R A)[I,:] is the sum of <= 4 fine rows, R^T spreads every fine column over <= 2 coarse columns; coarse rows reach a few hundred
// candidate products on the deeper levels.  Every sum must associate exactly like nalgebra-sparse's spmm_csr (c += a_ik * b_kj, k in row
// order) to keep the coarse operators bit-identical to the CPU oracle.  galerkin_bound_k sorts the rows into LDS tiers (list capacity
// 64 << t) by their candidate bound, one launch of galerkin_merge_k per non-empty tier; output goes to row-contiguous scratch at offsets
// from a scan of per-row bounds; galerkin_pack_fused_k then writes the SELL-64 image and the packed mirror.
// exclusive prefix sum of one int per lane across the wavefront
__device__ __forceinline__ int wave_excl_scan(int v, int &total) {
    const int lane = threadIdx.x & 63;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// The candidates of T = (R A)[I,:] are <= 4 fine rows whose columns ascend, so T is a MERGE: every candidate finds its
// place by binary searches in the other lists (equal columns keep the order of the fine rows, i ascending, which is the
// reference's order of accumulation).  And (T R^T)[I,J] = sum_j T_j R_Jj is a sum over the <= 4 fine indices of row J
// of R (restriction_row(J), ascending j — again the reference's order): once the distinct J are known each output lane
// looks its <= 4 terms up in T.  The distinct J need no sort either: a fine column j reaches J = j >> 1 (when j has a
// partner) and J' = chooser[j] >> 1 (when it was chosen), and whether an earlier T entry reaches the same coarse column is
// decided by O(1) look-ups (the sibling 2J+1 / the sibling's partner).  The survivors are ranked by counting.  (Round 1 sorted
// every coarse row twice with a bitonic network: dozens of LDS passes with a barrier each, twice the LDS; same bits.)
__device__ __forceinline__ int lds_lower_bound(const int *p, int len, int key) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (p[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int lds_upper_bound(const int *p, int len, int key) {
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (p[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// exclusive prefix sum of one int per lane across a group of G lanes
template <int G>
__device__ __forceinline__ int group_excl_scan(int v, int &total) {
    const int gl = threadIdx.x & (G - 1);
    int x = v;
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        const int y = __shfl_up(x, off, G);
        if (gl >= off) x += y;
    }
    total = __shfl(x, G - 1, G);
    return x - v;
}

// G lanes per coarse row (64 / G rows per wavefront): the passes of a narrow row (<= 32 candidates on the first coarse
// level) fill half a wavefront.
// [r03] Step 1 used to walk the <= 4 fine rows one after the other — descriptor -> columns -> values, twelve dependent global
// round trips per coarse row, 43 % of the kernel on the widest tier (a kernel cut short
// after step 1 and after step 4 measured it: 3.6 of 8.4 ms).  Now the descriptors of all fine rows are requested together, then their
// first G entries together, from the row-contiguous mirror where the matrix has one, and the next coarse row's index and
// pairing travel while the current row is merged: step 1 3.6 -> 1.5 ms, the kernel 8.4 -> 7.7 ms (the later steps slow down
// as the first one stops pacing them: the kernel as a whole moves ~800 scattered cache-line requests per coarse row).
// Batching the pairing look-ups of step 4 (speculative second-level loads) and running the LDS searches of a step side by
// side were measured too: +1.2 ms and +2.1 ms.  Same LDS passes, same order of every sum and output entry: bit-identical.
// [r04] S systems on ONE fine pattern and ONE pairing (the u, v, w momentum matrices whenever v's and w's fine-level pairings verify as u's:
// SiblingPairing) share everything symbolic — which candidates there are, where each one merges to, which coarse columns come out and in
// which order — so one pass carries S value sets through the same LDS passes (MergeSiblings: the values, scalings and outputs of systems
// 1 .. S-1; system 0 travels in the ordinary arguments).  Per system the products and the order of every sum are those of its own pass:
// bit-identical (tests/test_gpu_triple.py).  The kernel waits for scattered look-ups most of its time; those are now paid once for three.
struct MergeSiblings {
    const double *val[2] = {nullptr, nullptr};    // fine values, addressed like system 0's (SELL image or row-contiguous mirror)
    const double *s1[2] = {nullptr, nullptr}, *s2[2] = {nullptr, nullptr};  // the views' row scalings (null where system 0 has none)
    double *s_val[2] = {nullptr, nullptr};        // scratch rows, same offsets as system 0's
};

template <int G, int S = 1>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S == 1 ? 6 : 4))) void galerkin_merge_k(MatView A, const int *__restrict__ choice, const int *__restrict__ chooser, int64_t n_coarse,
                                                        int cap /* power of two >= 2 * candidates */, int *__restrict__ row_len_c,
                                                        const long long *__restrict__ slice_base, const int *__restrict__ intra_off, int *__restrict__ s_col,
                                                        double *__restrict__ s_val, const int *__restrict__ list, const int *__restrict__ list_count,
                                                        MergeSiblings X) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kRows = 64 / G;  // coarse rows in flight per wavefront
    constexpr int kX = S - 1;      // sibling systems
    const int h = cap >> 1;
    const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    double *src_val = reinterpret_cast<double *>(smem + (size_t)grp * (size_t)cap * (size_t)(12 + 8 * kX));  // candidates in generation order, later T's values
    double *m_val = src_val + h;                          // merged candidates
    int *src_col = reinterpret_cast<int *>(m_val + h);    // ... later T's columns
    int *m_col = src_col + h;
    // [cap] distinct coarse columns, unsorted: written by step 4, when the merged candidates (last read by step 3) are dead — the list lives
    // in their values' place [r04: 12 instead of 16 bytes of LDS per list slot]
    int *U = reinterpret_cast<int *>(m_val);
    double *x_src = reinterpret_cast<double *>(m_col + h);  // siblings: [kX][h] candidates / T values, then [kX][h] merged candidates
    double *x_m = x_src + (size_t)(kX > 0 ? kX : 1) * h;
    const int n_fine = (int)A.P.n;
    const int64_t total_rows = list ? (int64_t)*list_count : n_coarse;
    const int64_t it_step = (int64_t)gridDim.x * kRows;
    // entry k of fine row i: the row-contiguous mirror where the matrix has one (RowWalk)
    const bool mirror = A.rows.col != nullptr;
    const int32_t *colp = mirror ? A.rows.col : A.P.col;
    const double *valp = mirror ? A.rows.val : A.val;
    const int64_t stride = mirror ? 1 : 64;
    // this iteration's row, loaded one iteration ahead
    int64_t it0 = (int64_t)blockIdx.x * kRows;
    bool active = it0 + grp < total_rows;
    int64_t I = active ? (list ? (int64_t)list[it0 + grp] : it0 + grp) : 0;
    int pair0 = (active && 2 * I < n_fine) ? choice[2 * I] : -1;
    int pair1 = (active && 2 * I + 1 < n_fine) ? choice[2 * I + 1] : -1;
    for (; it0 < total_rows; it0 += it_step) {
        // (the barriers below are reached by every group the same number of times)
        const int64_t it_n = it0 + it_step + grp;
        const bool active_n = it_n < total_rows;
        const int64_t I_n = active_n ? (list ? (int64_t)list[it_n] : it_n) : 0;  // in flight during steps 1-3
        RRow R;
        R.n = 0;
        if (active) R = restriction_row_from(I, pair0, pair1);
        // ---- 1. candidates, list after list (ghost columns dropped: coarse levels are per rank)
        int b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        {
            int len[4];
            int64_t rb[4];
            double sc1[4], sc2[4];
            double xs1[kX > 0 ? kX : 1][4], xs2[kX > 0 ? kX : 1][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {  // descriptors of all fine rows at once
                const bool on = a < R.n;
                const int i = on ? R.idx[a] : 0;
                len[a] = on ? A.P.row_len[i] : 0;
                rb[a] = mirror ? (int64_t)A.rows.slice_base[i >> 6] + A.rows.intra_off[i] : A.P.slice_ptr[i >> 6] + (i & 63);
                sc1[a] = A.s1 ? A.s1[i] : 1.;
                sc2[a] = A.s2 ? A.s2[i] : 1.;
#pragma unroll
                for (int x = 0; x < kX; ++x) {
                    xs1[x][a] = X.s1[x] ? X.s1[x][i] : 1.;
                    xs2[x][a] = X.s2[x] ? X.s2[x][i] : 1.;
                }
            }
            int c0[4];
            double v0[4];
            double xv0[kX > 0 ? kX : 1][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {  // their first G entries at once
                const bool in = lane < len[a];
                const int64_t pos = rb[a] + (int64_t)(in ? lane : 0) * stride;
                c0[a] = in ? colp[pos] : -1;
                v0[a] = in ? valp[pos] : 0.;
#pragma unroll
                for (int x = 0; x < kX; ++x) xv0[x][a] = in ? X.val[x][pos] : 0.;
            }
            int base = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < R.n) {
                    const double w = R.w[a];
                    for (int k0 = 0; k0 < len[a]; k0 += G) {
                        const int k = k0 + lane;
                        int c = -1;
                        double v = 0.;
                        double xv[kX > 0 ? kX : 1];
#pragma unroll
                        for (int x = 0; x < kX; ++x) xv[x] = 0.;
                        if (k0 == 0) {
                            c = c0[a]; v = v0[a];
#pragma unroll
                            for (int x = 0; x < kX; ++x) xv[x] = xv0[x][a];
                        } else if (k < len[a]) {
                            const int64_t pos = rb[a] + (int64_t)k * stride;
                            c = colp[pos]; v = valp[pos];
#pragma unroll
                            for (int x = 0; x < kX; ++x) xv[x] = X.val[x][pos];
                        }
                        const int valid = (c >= 0 && c < n_fine) ? 1 : 0;
                        int tot;
                        const int slot = base + group_excl_scan<G>(valid, tot);
                        if (valid) {
                            if (A.s1) v = sc1[a] * v;  // view_value's order
                            if (A.s2) v = sc2[a] * v;
                            src_col[slot] = c;
                            src_val[slot] = w * v;
#pragma unroll
                            for (int x = 0; x < kX; ++x) {
                                double t = xv[x];
                                if (X.s1[x]) t = xs1[x][a] * t;
                                if (X.s2[x]) t = xs2[x][a] * t;
                                x_src[(size_t)x * h + slot] = w * t;
                            }
                        }
                        base += tot;
                    }
                }
                if (a == 0) b1 = base;
                if (a == 1) b2 = base;
                if (a == 2) b3 = base;
                if (a == 3) b4 = base;
            }
        }
        const int cnt = b4;
        __syncthreads();
        // ---- 2. merge: rank = own position + entries of earlier lists with column <= c + entries of later lists with column < c
        for (int e = lane; e < cnt; e += G) {
            const int a = (e >= b1) + (e >= b2) + (e >= b3);
            const int c = src_col[e];
            const int own_base = a == 0 ? 0 : (a == 1 ? b1 : (a == 2 ? b2 : b3));
            int rank = e - own_base;
            if (a != 0 && b1 > 0) rank += lds_upper_bound(src_col, b1, c);
            if (a != 1 && b2 > b1) rank += a > 1 ? lds_upper_bound(src_col + b1, b2 - b1, c) : lds_lower_bound(src_col + b1, b2 - b1, c);
            if (a != 2 && b3 > b2) rank += a > 2 ? lds_upper_bound(src_col + b2, b3 - b2, c) : lds_lower_bound(src_col + b2, b3 - b2, c);
            if (a != 3 && b4 > b3) rank += lds_lower_bound(src_col + b3, b4 - b3, c);
            m_col[rank] = c;
            m_val[rank] = src_val[e];
#pragma unroll
            for (int x = 0; x < kX; ++x) x_m[(size_t)x * h + rank] = x_src[(size_t)x * h + e];
        }
        __syncthreads();
        // ---- 3. runs of equal j -> T (sorted by j) into src_col / src_val
        int cntT = 0;
        for (int b0 = 0; b0 < cnt; b0 += G) {
            const int e = b0 + lane;
            int head = 0;
            if (e < cnt) head = (e == 0) || (m_col[e] != m_col[e - 1]);
            int tot;
            const int slot = cntT + group_excl_scan<G>(head, tot);
            if (head) {
                const int j = m_col[e];
                double acc = 0. + m_val[e];
                double xacc[kX > 0 ? kX : 1];
#pragma unroll
                for (int x = 0; x < kX; ++x) xacc[x] = 0. + x_m[(size_t)x * h + e];
                for (int q = e + 1; q < cnt && m_col[q] == j; ++q) {
                    acc += m_val[q];
#pragma unroll
                    for (int x = 0; x < kX; ++x) xacc[x] += x_m[(size_t)x * h + q];
                }
                src_col[slot] = j;
                src_val[slot] = acc;
#pragma unroll
                for (int x = 0; x < kX; ++x) x_src[(size_t)x * h + slot] = xacc[x];
            }
            cntT += tot;
        }
        __syncthreads();
        // the next row's pairing (its index has arrived by now): in flight during steps 4-5
        const int pair0_n = (active_n && 2 * I_n < n_fine) ? choice[2 * I_n] : -1;
        const int pair1_n = (active_n && 2 * I_n + 1 < n_fine) ? choice[2 * I_n + 1] : -1;
        const int *tj = src_col;
        const double *tv = src_val;
        // ---- 4. the distinct coarse columns, first occurrence only
        int nU = 0;
        for (int b0 = 0; b0 < cntT; b0 += G) {
            const int e = b0 + lane;
            int k0 = -1, k1 = -1;
            if (e < cntT) {
                const int j = tj[e];
                const int cj = choice[j], mj = chooser[j];
                if (cj >= 0) {  // J = j >> 1; its other fine row 2J comes first when it is here too
                    const bool dup = (j & 1) && e > 0 && tj[e - 1] == j - 1 && choice[j - 1] >= 0;
                    if (!dup) k0 = j >> 1;
                }
                if (mj >= 0) {  // J' = chooser[j] >> 1
                    const int Jp = mj >> 1;
                    bool drop = cj >= 0 && Jp == (j >> 1);
                    if (!drop) {  // reached through its own fine rows 2J', 2J'+1 (with a partner) by some T entry?
                        const int p = lds_lower_bound(tj, cntT, 2 * Jp);
                        const bool has_even = p < cntT && tj[p] == 2 * Jp;
                        if (has_even && choice[2 * Jp] >= 0) drop = true;
                        else {
                            const int q = has_even ? p + 1 : p;
                            if (q < cntT && tj[q] == 2 * Jp + 1 && choice[2 * Jp + 1] >= 0) drop = true;
                        }
                    }
                    if (!drop) {  // the sibling of chooser[j] chose an earlier T entry: that one keeps J'
                        const int sib = mj ^ 1;
                        if (sib < n_fine) {
                            const int js = choice[sib];
                            if (js >= 0 && js < j) {
                                const int p = lds_lower_bound(tj, cntT, js);
                                if (p < cntT && tj[p] == js) drop = true;
                            }
                        }
                    }
                    if (!drop) k1 = Jp;
                }
            }
            int tot;
            int slot = nU + group_excl_scan<G>((k0 >= 0 ? 1 : 0) + (k1 >= 0 ? 1 : 0), tot);
            if (k0 >= 0) U[slot++] = k0;
            if (k1 >= 0) U[slot] = k1;
            nU += tot;
        }
        __syncthreads();
        // ---- 5. every distinct J: position by counting, value from the <= 4 fine indices of row J of R (ascending)
        const long long off = active ? slice_base[I >> 6] + intra_off[I] : 0;
        for (int e = lane; e < nU; e += G) {
            const int u = U[e];
            int rank = 0;
            for (int q = 0; q < nU; ++q) rank += U[q] < u ? 1 : 0;
            const RRow RJ = restriction_row(choice, u, A.P.n);
            double acc = 0.;
            double xacc[kX > 0 ? kX : 1];
#pragma unroll
            for (int x = 0; x < kX; ++x) xacc[x] = 0.;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < RJ.n) {
                    const int p = lds_lower_bound(tj, cntT, RJ.idx[a]);
                    if (p < cntT && tj[p] == RJ.idx[a]) {
                        acc += tv[p] * RJ.w[a];
#pragma unroll
                        for (int x = 0; x < kX; ++x) xacc[x] += x_src[(size_t)x * h + p] * RJ.w[a];
                    }
                }
            }
            s_col[off + rank] = u;
            s_val[off + rank] = acc;
#pragma unroll
            for (int x = 0; x < kX; ++x) X.s_val[x][off + rank] = xacc[x];
        }
        if (active && lane == 0) row_len_c[I] = nU;
        __syncthreads();
        active = active_n; I = I_n; pair0 = pair0_n; pair1 = pair1_n;
    }
}

// scratch rows -> SELL-64 (columns, values, diagonal offsets, padding) and, when asked for, the packed mirror, in one
// pass.  The scratch rows are contiguous per ROW, the images are interleaved per SLICE: a thread copying its own row reads
// 64 different cache lines per instruction (the texture path serialises them: 3 ms per image at 5 M rows).  Here one
// wavefront moves one slice through an LDS tile of 16 depths x 64 rows: four rows at a time are read with 16 consecutive
// lanes each (a handful of lines per instruction), the tile is read back depth by depth with lane = row, and both images
// are written with full-width stores.
// kValuesOnly [r04]: a sibling system on the same coarse pattern (MergeSiblings) — only val_c / pk_val are written.
constexpr int kPackDepth = 16;
template <bool kValuesOnly = false>
__global__ __launch_bounds__(64) void galerkin_pack_fused_k(SellDev Pc, const long long *__restrict__ slice_base, const int *__restrict__ intra_off,
                                                            const int *__restrict__ s_col, const double *__restrict__ s_val, int *__restrict__ col_c,
                                                            double *__restrict__ val_c, int *__restrict__ diag_c, const int64_t *__restrict__ pk_ptr,
                                                            int *__restrict__ pk_col, double *__restrict__ pk_val) {
    __shared__ int t_col[kPackDepth * 65];
    __shared__ double t_val[kPackDepth * 65];
    const int lane = threadIdx.x;
    const int kk = lane & (kPackDepth - 1), rr = lane / kPackDepth;  // gather phase: depth inside the tile, row inside the group of 4
    for (int64_t slice = blockIdx.x; slice < Pc.n_slices; slice += gridDim.x) {
        const int64_t I = slice * 64 + lane;
        const bool live = I < Pc.n;
        const int len = live ? Pc.row_len[I] : 0;
        const long long src = live ? slice_base[slice] + intra_off[I] : 0;
        const int src_lo = (int)(unsigned)(src & 0xffffffffll), src_hi = (int)(src >> 32);
        const int64_t base = Pc.slice_ptr[slice];
        const int width = (int)((Pc.slice_ptr[slice + 1] - base) >> 6);
        int64_t pk_off = pk_ptr ? pk_ptr[slice] : 0;
        int d = -1;
        for (int kc = 0; kc < width; kc += kPackDepth) {
            // ---- gather: rows 4 rb + rr, depths kc + kk
#pragma unroll 4
            for (int rb = 0; rb < 64 / (64 / kPackDepth); ++rb) {
                const int row = rb * (64 / kPackDepth) + rr;
                const int rlen = __shfl(len, row, 64);
                const long long rsrc = ((long long)__shfl(src_hi, row, 64) << 32) | (long long)(unsigned)__shfl(src_lo, row, 64);
                const int k = kc + kk;
                if (k < rlen) {
                    if (!kValuesOnly) t_col[kk * 65 + row] = s_col[rsrc + k];
                    t_val[kk * 65 + row] = s_val[rsrc + k];
                }
            }
            __syncthreads();
            // ---- scatter: depth by depth, lane = row
            const int kend = min(kPackDepth, width - kc);
            for (int q = 0; q < kend; ++q) {
                const int k = kc + q;
                const bool in = k < len;
                const int c = (in && !kValuesOnly) ? t_col[q * 65 + lane] : (int)I;
                const double v = in ? t_val[q * 65 + lane] : 0.;
                const int64_t pos = base + (int64_t)k * 64 + lane;
                if (live) {
                    if (!kValuesOnly) col_c[pos] = c;
                    val_c[pos] = v;
                    if (!kValuesOnly && in && c == (int)I) d = (int)pos;
                }
                if (pk_ptr) {
                    const unsigned long long m = __ballot(in);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (in) {
                        if (!kValuesOnly) pk_col[pk_off + rank] = c;
                        pk_val[pk_off + rank] = v;
                    }
                    pk_off += __popcll(m);
                }
            }
            __syncthreads();
        }
        if (live && !kValuesOnly) diag_c[I] = d;
    }
}

// Per coarse row: candidate count c = sum of the lengths of its (<= 4) fine rows.  2c bounds the row's coarse
// entries (every candidate spawns <= 2 products), so the scratch offset of row I is the exclusive prefix sum of 2c:
// computed here per 64-row slice (wave scan) + slice totals, finished by scan_excl_dev.  No allocator atomics.
constexpr int kGalerkinTiers = 7;  // LDS list capacities 64 << t, t = 0..6 (2 KB .. 128 KB per wavefront)
// [r04] The tier lists are appended to behind one counter per tier, and same-address atomics retire at 11.4 ns per wave-instruction
// (scripts/microbench/atomic_rate.hip): one append per slice and tier was the kernel — 80 000 slices on the first level, 0.74-1.06 ms of a
// pass that moves 100 MB.  Now a wavefront keeps the tiers of all its slices in LDS (kBoundIters of them at most: the launch is sized for
// that), counts them, reserves ONE range per tier and writes its rows there in a second walk over the LDS bytes.
constexpr int kBoundIters = 32;
__global__ __launch_bounds__(64) void galerkin_bound_k(SellDev P, const int *__restrict__ choice, int64_t n_coarse, int *__restrict__ out_max,
                                                       unsigned long long *__restrict__ out_sum, int *__restrict__ intra_off,
                                                       long long *__restrict__ slice_tot, int *__restrict__ tier_count, int *__restrict__ tier_list, int min_tier) {
    __shared__ signed char tiers[kBoundIters][64];
    const int lane = threadIdx.x;
    const int64_t n_slices = (n_coarse + 63) / 64;
    int mx = 0;
    unsigned long long sm = 0;
    int cnt[kGalerkinTiers];
#pragma unroll
    for (int t = 0; t < kGalerkinTiers; ++t) cnt[t] = 0;
    // (a wavefront takes ADJACENT slices, so that the tier lists come out in row order: neighbouring list entries — what concurrent merge
    // wavefronts work on — are neighbouring coarse rows)
    const int64_t per_wave = (n_slices + gridDim.x - 1) / gridDim.x;  // <= kBoundIters by the launch's size
    const int64_t s_lo = (int64_t)blockIdx.x * per_wave, s_hi = s_lo + per_wave < n_slices ? s_lo + per_wave : n_slices;
    int it = 0;
    for (int64_t s = s_lo; s < s_hi && it < kBoundIters; ++s, ++it) {
        const int64_t I = s * 64 + lane;
        int c = 0;
        if (I < n_coarse) {
            const RRow R = restriction_row(choice, I, P.n);
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < R.n) c += P.row_len[R.idx[a]];
        }
        int tot;
        const int ex = wave_excl_scan(2 * c, tot);
        if (I < n_coarse) intra_off[I] = ex;
        if (lane == 0) slice_tot[s] = tot;
        mx = max(mx, c);
        sm += (unsigned long long)c;
        // the row's list never exceeds 2c entries: it goes to the narrowest tier that holds them (no overflow passes)
        int tier = -1;
        if (I < n_coarse) {
            tier = min_tier;  // sorting kernel: 128 slots (4 KB) at least, 16 wavefronts per CU already saturate the narrow rows
            while (tier < kGalerkinTiers - 1 && (64 << tier) < 2 * c) ++tier;
        }
        tiers[it][lane] = (signed char)tier;
#pragma unroll
        for (int t = 0; t < kGalerkinTiers; ++t) cnt[t] += __popcll(__ballot(tier == t));  // wave-uniform
    }
    int base[kGalerkinTiers];
#pragma unroll
    for (int t = 0; t < kGalerkinTiers; ++t) {
        base[t] = 0;
        if (cnt[t] > 0) {  // wave-uniform
            if (lane == 0) base[t] = atomicAdd(&tier_count[t], cnt[t]);
            base[t] = __shfl(base[t], 0, 64);
        }
    }
    it = 0;
    for (int64_t s = s_lo; s < s_hi && it < kBoundIters; ++s, ++it) {
        const int64_t I = s * 64 + lane;
        const int tier = tiers[it][lane];
#pragma unroll
        for (int t = 0; t < kGalerkinTiers; ++t) {
            const unsigned long long m = __ballot(tier == t);
            if (m == 0ull) continue;
            if (tier == t) tier_list[(int64_t)t * n_coarse + base[t] + __popcll(m & ((1ull << lane) - 1ull))] = (int)I;
            base[t] += __popcll(m);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = max(mx, __shfl_down(mx, off, 64));
        sm += __shfl_down(sm, off, 64);
    }
    if (lane == 0) { atomicMax(out_max, mx); atomicAdd(out_sum, sm); }
}

// exclusive scan of one value per thread across a workgroup of kScanThreads (the building block of scan_excl_dev below).  256 threads, not
// 1024: beside the products of another stream a 16-wavefront workgroup waits for a CU with four free slots on every SIMD (100 us alone,
// 0.6-1.2 ms in the concurrent schedule, most of it before its first instruction).
constexpr int kScanThreads = 256;
__device__ __forceinline__ long long block_excl_scan(long long v, long long *buf /*[kScanThreads]*/, long long &total) {
    const int t = threadIdx.x;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {
        const long long x = t >= off ? buf[t - off] : 0;
        __syncthreads();
        buf[t] += x;
        __syncthreads();
    }
    total = buf[kScanThreads - 1];
    const long long r = buf[t] - v;
    __syncthreads();
    return r;
}

// slice widths -> slice_ptr and packed sizes -> pk_ptr.
// One wavefront per slice reduces its 64 row lengths (SELL width * 64 and the packed size rounded up to 16 elements =
// 128 bytes), then both tables are scanned (scan_excl_dev; a single workgroup reading all n row lengths itself took 0.9 + 1.4 ms
// per level at 5 M rows).
__global__ __launch_bounds__(kBlock) void slice_sizes_k(const int *__restrict__ row_len, int64_t n, int n_slices, int64_t *__restrict__ w_sell,
                                                        int64_t *__restrict__ w_pk) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t s = wave; s < n_slices; s += waves) {
        const int64_t r = s * 64 + lane;
        const int len = r < n ? row_len[r] : 0;
        int mx = len, sum = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = max(mx, __shfl_xor(mx, off, 64));
            sum += __shfl_xor(sum, off, 64);
        }
        if (lane == 0) { w_sell[s] = (int64_t)mx * 64; w_pk[s] = ((int64_t)sum + 15) & ~(int64_t)15; }
    }
}
// [r04] The same scans over the whole chip: one workgroup walking 80 000 slice totals took 0.1-0.2 ms (its threads' contiguous shares are
// 64 different cache lines per load instruction) — on the set-up's dependent chain, in front of a host read, six times per hierarchy.  Three
// small launches instead: per-chunk sums (coalesced), a scan of the <= kScanBlocks sums, per-chunk scans from their bases.  Integer sums: exact.
constexpr int kScanBlocks = 128;
__global__ __launch_bounds__(kScanThreads) void scan_part_k(const long long *__restrict__ in_a, const long long *__restrict__ in_b, int64_t n, int64_t chunk,
                                                            long long *__restrict__ part /* [2][kScanBlocks] sums, then [2] totals */) {
    __shared__ long long buf[kScanThreads];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = std::min<int64_t>(n, lo + chunk);
    long long sa = 0, sb = 0;
    for (int64_t e = lo + threadIdx.x; e < hi; e += kScanThreads) { sa += in_a[e]; if (in_b) sb += in_b[e]; }
    long long ta, tb;
    (void)block_excl_scan(sa, buf, ta);
    (void)block_excl_scan(sb, buf, tb);
    if (threadIdx.x == 0) { part[blockIdx.x] = ta; part[kScanBlocks + blockIdx.x] = tb; }
}
__global__ __launch_bounds__(kScanThreads) void scan_mid_k(long long *__restrict__ part, int n_blocks) {
    __shared__ long long buf[kScanThreads];
    const int t = threadIdx.x;
    const long long va = t < n_blocks ? part[t] : 0, vb = t < n_blocks ? part[kScanBlocks + t] : 0;
    long long ta, tb;
    const long long ra = block_excl_scan(va, buf, ta);
    const long long rb = block_excl_scan(vb, buf, tb);
    if (t < n_blocks) { part[t] = ra; part[kScanBlocks + t] = rb; }
    if (t == 0) { part[2 * kScanBlocks] = ta; part[2 * kScanBlocks + 1] = tb; }
}
__global__ __launch_bounds__(kScanThreads) void scan_write_k(const long long *__restrict__ in_a, const long long *__restrict__ in_b, int64_t n, int64_t chunk,
                                                             const long long *__restrict__ part, long long *__restrict__ out_a, long long *__restrict__ out_b,
                                                             int write_totals /* out[n] = total */) {
    __shared__ long long buf[kScanThreads];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = std::min<int64_t>(n, lo + chunk);
    long long base_a = part[blockIdx.x], base_b = part[kScanBlocks + blockIdx.x];
    for (int64_t t0 = lo; t0 < hi; t0 += kScanThreads) {  // (workgroup-uniform trip count)
        const int64_t e = t0 + threadIdx.x;
        const long long va = e < hi ? in_a[e] : 0, vb = (in_b && e < hi) ? in_b[e] : 0;
        long long ta, tb = 0;
        const long long ra = block_excl_scan(va, buf, ta);
        long long rb = 0;
        if (in_b) rb = block_excl_scan(vb, buf, tb);
        if (e < hi) { out_a[e] = base_a + ra; if (in_b) out_b[e] = base_b + rb; }
        base_a += ta;
        base_b += tb;
    }
    if (write_totals && blockIdx.x == 0 && threadIdx.x == 0) { out_a[n] = part[2 * kScanBlocks]; if (in_b) out_b[n] = part[2 * kScanBlocks + 1]; }
}
// out_a (and out_b) = exclusive prefix sums of in_a (in_b; null: one table); write_totals: out[n] = the sum.  `part`: 2 * kScanBlocks + 2 words.
static int scan_excl_dev(const long long *in_a, const long long *in_b, int64_t n, long long *out_a, long long *out_b, bool write_totals, long long *part, hipStream_t st) {
    const int64_t n1 = std::max<int64_t>(n, 1);
    int64_t chunk = (n1 + kScanBlocks - 1) / kScanBlocks;
    chunk = ((chunk + kScanThreads - 1) / kScanThreads) * kScanThreads;
    const int n_blocks = (int)((n1 + chunk - 1) / chunk);
    hipLaunchKernelGGL(scan_part_k, dim3(n_blocks), dim3(kScanThreads), 0, st, in_a, in_b, n, chunk, part);
    hipLaunchKernelGGL(scan_mid_k, dim3(1), dim3(kScanThreads), 0, st, part, n_blocks);
    hipLaunchKernelGGL(scan_write_k, dim3(n_blocks), dim3(kScanThreads), 0, st, in_a, in_b, n, chunk, (const long long *)part, out_a, out_b, write_totals ? 1 : 0);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}


// ---- packed mirror of the coarse operator (PackedDev, linalg.hpp): what the level's ~200 products stream
// ---- LDS x windows of the packed mirror (XWinDev, linalg.hpp): per block of 256 rows the ascending list of distinct
// columns and, per packed entry, the 16-bit position of its column in that list.  One workgroup per block: the columns
// set bits in an LDS bitmap over the block's column span, a prefix of the word population counts turns a bit into its
// rank.  A block whose span exceeds the bitmap or whose window exceeds kXWinCap gets wsize = -1 (global gathers).
constexpr int kXBitWords = 8192;  // 262144 columns of span
// [r04] The two limits are run-time arguments bounded by the compiled LDS sizes (win_cap <= kXWinCap, bit_words <= kXBitWords;
// ORC_XWIN_CAP / ORC_XWIN_BITWORDS, read per set-up): at bench size 1 % of level 3's blocks take the no-window path of the
// product and none the span branch, on test-sized meshes none at all — the tests shrink the limits to drive a chosen share of
// the blocks through both branches and compare with the oracle (tests/test_gpu_window_fallback.py).  g_xwin_counters: blocks
// built / without a window because of the cap / because of the span, since the last reset (orc_debug_xwin_counters).
__device__ unsigned long long g_xwin_counters[3];
// [r04] Two passes: the first with a bitmap of kXBitWordsSmall words (25 KB of LDS: six workgroups per CU instead of three) takes every block whose
// columns span at most 131 072 and marks the others pending (wsize = -2); the second, with the full bitmap, runs only if any block is pending and
// looks at those only.  (ORC_AMG_TRACE "[amg windows]": with 2 048 words half of the channel's level-2 / 3 blocks were left to the second pass.)
constexpr int kXBitWordsSmall = 4096;
template <int kWords, bool kSecond>
__global__ __launch_bounds__(kBlock) void xwin_build_k(SellDev P, PackedDev pk, int *__restrict__ wcol, int *__restrict__ wsize,
                                                       unsigned short *__restrict__ lidx, int64_t n_blocks, int win_cap, int bit_words, int pass_words,
                                                       int *__restrict__ pending /* blocks the first pass left to the second */) {
    if (kSecond && *pending == 0) return;
    __shared__ unsigned bits[kWords];
    __shared__ unsigned short wpre[kWords];  // exclusive prefix of the word population counts (windows hold <= 4096)
    __shared__ int s_min, s_max, s_part[kBlock];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long n_built = 0, n_capped = 0, n_spanned = 0;  // thread 0's tallies: ONE atomic per counter and workgroup at the end
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        if (kSecond && wsize[b] != -2) continue;  // workgroup-uniform: done by the first pass
        const int64_t row = b * kXWinRows + tid;
        const bool live = row < P.n;
        const int len = live ? P.row_len[row] : 0;
        const int64_t rbase = live ? P.slice_ptr[row >> 6] + (row & 63) : 0;
        if (tid == 0) { s_min = 0x7fffffff; s_max = -1; }
        __syncthreads();
        if (len > 0) {  // columns ascend within a row
            atomicMin(&s_min, P.col[rbase]);
            atomicMax(&s_max, P.col[rbase + (int64_t)(len - 1) * 64]);
        }
        __syncthreads();
        const int cmin = s_min, span = s_max - s_min + 1;
        const int words = (span + 31) >> 5;
        if (s_max < 0) {  // empty block
            if (tid == 0) wsize[b] = 0;
            __syncthreads();
            continue;
        }
        if (!kSecond && words <= bit_words && words > pass_words) {  // the full bitmap's business (pass_words <= kWords: ORC_XWIN_SMALL_BITWORDS, a test hook)
            if (tid == 0) { wsize[b] = -2; atomicAdd(pending, 1); }
            __syncthreads();
            continue;
        }
        ++n_built;
        if (words > bit_words || words > kWords) {
            if (tid == 0) wsize[b] = -1;
            ++n_spanned;
            __syncthreads();
            continue;
        }
        for (int w = tid; w < words; w += kBlock) bits[w] = 0u;
        __syncthreads();
        for (int k = 0; k < len; ++k) {
            const int c = P.col[rbase + (int64_t)k * 64] - cmin;
            atomicOr(&bits[c >> 5], 1u << (c & 31));
        }
        __syncthreads();
        // prefix over the words: thread t owns words [t * per, (t + 1) * per)
        const int per = (words + kBlock - 1) / kBlock;
        int local = 0;
        for (int w = tid * per; w < words && w < (tid + 1) * per; ++w) local += __popc(bits[w]);
        s_part[tid] = local;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {  // Hillis-Steele inclusive scan of the 256 partial sums
            const int t = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        const int total = s_part[kBlock - 1];
        if (total > win_cap) {
            __syncthreads();
            if (tid == 0) wsize[b] = -1;
            ++n_capped;
            __syncthreads();
            continue;
        }
        int run = s_part[tid] - local;  // exclusive
        int *wc = wcol + b * kXWinCap;
        for (int w = tid * per; w < words && w < (tid + 1) * per; ++w) {
            wpre[w] = (unsigned short)run;
            unsigned m = bits[w];
            while (m) {
                const int bit = __ffs(m) - 1;
                wc[run++] = cmin + (w << 5) + bit;
                m &= m - 1;
            }
        }
        if (tid == 0) wsize[b] = total;
        __syncthreads();
        // window positions of the packed entries: wave per slice, depth by depth (the packed order of galerkin_pack_packed_k)
        const int64_t slice = b * 4 + wave;
        if (slice < P.n_slices) {
            const int64_t sbase = P.slice_ptr[slice];
            const int width = (int)((P.slice_ptr[slice + 1] - sbase) >> 6);
            int64_t off = pk.ptr[slice];
            for (int q = 0; q < width; ++q) {
                const bool in = q < len;
                const unsigned long long m = __ballot(in);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (in) {
                    const int c = P.col[sbase + (int64_t)q * 64 + lane] - cmin;
                    lidx[off + rank] = (unsigned short)(wpre[c >> 5] + __popc(bits[c >> 5] & ((1u << (c & 31)) - 1u)));
                }
                off += __popcll(m);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (n_built) atomicAdd(&g_xwin_counters[0], n_built);
        if (n_capped) atomicAdd(&g_xwin_counters[1], n_capped);
        if (n_spanned) atomicAdd(&g_xwin_counters[2], n_spanned);
    }
}

// ORC_XWIN_STATS=1 (measurement): how the windows of a level are made up — entries, maximal runs of consecutive columns, runs of
// eight or more, blocks whose columns span fewer than 65 536
// candidates for a level's LDS share (XWinDev::cap), ascending; over[q] = blocks whose window holds more than kXWinCapSize[q] entries
constexpr int kXWinCapSizes = 5;
__device__ __constant__ int kXWinCapSizeDev[kXWinCapSizes] = {2048, 2560, 3200, 4000, kXWinCap};
static const int kXWinCapSize[kXWinCapSizes] = {2048, 2560, 3200, 4000, kXWinCap};
__global__ __launch_bounds__(kBlock) void xwin_cap_k(const int *__restrict__ wsize, int64_t n_blocks, int *__restrict__ over) {
    int c[kXWinCapSizes] = {0, 0, 0, 0, 0};
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (int64_t)gridDim.x * blockDim.x) {
        const int ws = wsize[b];
#pragma unroll
        for (int q = 0; q < kXWinCapSizes; ++q) c[q] += ws > kXWinCapSizeDev[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < kXWinCapSizes; ++q)
        if (c[q]) atomicAdd(over + q, c[q]);
}

__global__ __launch_bounds__(kBlock) void xwin_stats_k(const int *__restrict__ wcol, const int *__restrict__ wsize, int64_t n_blocks, unsigned long long *__restrict__ out) {
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const int ws = wsize[b];
        if (ws <= 0) { if (threadIdx.x == 0 && ws < 0) atomicAdd(out + 4, 1ull); continue; }
        const int *wc = wcol + b * kXWinCap;
        unsigned long long runs = 0, long_entries = 0;
        for (int j = threadIdx.x; j < ws; j += kBlock) {
            if (j == 0 || wc[j] != wc[j - 1] + 1) {
                ++runs;
                int e = j + 1;
                while (e < ws && wc[e] == wc[e - 1] + 1) ++e;
                if (e - j >= 8) long_entries += (unsigned long long)(e - j);
            }
        }
        atomicAdd(out + 1, runs);
        atomicAdd(out + 2, long_entries);
        if (threadIdx.x == 0) {
            atomicAdd(out + 0, (unsigned long long)ws);
            if (wc[ws - 1] - wc[0] < 65536) atomicAdd(out + 3, 1ull);
        }
    }
}

// ---- narrow column image of a coarse operator (SellDev::col16 / colbase, linalg.hpp): one wavefront per slice; per depth the
// smallest column among the rows that reach it and 16-bit offsets from it; *too_wide is raised if a depth spans 65 536 or more
__global__ __launch_bounds__(64) void narrow_build_k(SellDev P, unsigned short *__restrict__ col16, int *__restrict__ colbase, int *__restrict__ too_wide) {
    const int lane = threadIdx.x;
    for (int64_t slice = blockIdx.x; slice < P.n_slices; slice += gridDim.x) {
        const int64_t row = slice * 64 + lane;
        const int64_t sb = P.slice_ptr[slice];
        const int width = (int)((P.slice_ptr[slice + 1] - sb) >> 6);
        const int len = row < P.n ? P.row_len[row] : 0;
        for (int k = 0; k < width; ++k) {
            const bool in = k < len;
            const int c = in ? P.col[sb + (int64_t)k * 64 + lane] : 0;
            int lo = in ? c : 0x7fffffff, hi = in ? c : -1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo = min(lo, __shfl_xor(lo, off, 64));
                hi = max(hi, __shfl_xor(hi, off, 64));
            }
            if (hi < 0) lo = 0;  // nobody reaches this depth
            if (lane == 0) {
                colbase[(sb >> 6) + k] = lo;
                if (hi >= 0 && hi - lo > 65535) atomicOr(too_wide, 1);
            }
            col16[sb + (int64_t)k * 64 + lane] = in ? (unsigned short)(c - lo) : (unsigned short)0;
        }
    }
}

__global__ void nan_to_status_k(const double *__restrict__ value, int *status, int code) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && isnan(value[0])) atomicCAS(status, 0, code);
}

// ------------------------------------------------------------------ host drivers
struct CoarseLevel {
    SellDev P;
    double *val = nullptr;
    PackedDev pk;
    XWinDev xw;
    RowsDev rows;
    bool rows_transient = false;  // `rows` lives in the set-up's companion arena: valid until the next level has been built
    int64_t n = 0, padded = 0;
    int *choice = nullptr, *chooser = nullptr;  // of the FINE level this was built from
    int rounds = 0;
};

// Is `choice` the fixed point?  Every row is evaluated against the exact first-taker table of `choice` itself; a state in
// which no row would choose differently is the sequential greedy pairing (the fixed point is unique).  Thread per row:
// coalesced reads of the interleaved image.  Counts the rows that would change.
__global__ void agg_verify_k(MatView A, const int *__restrict__ choice, const int *__restrict__ taken_by, AggCounters *C) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x)
        if (agg_eval_row(A, taken_by, i, true) != choice[i]) ++bad;
    if (bad) atomicAdd(&C->changed, bad);  // rare: a verified pairing has none
}

// ------------------------------------------------------------------ the pairing by deferred acceptance [r05]
// The reference's loop (linear_algebra.rs:30-60) is a SERIAL DICTATORSHIP: row i takes the column it prefers most — value ascending, position
// ascending: the strict <, first-wins scan, with the diagonal, NaNs and Float::MAX left out — among the columns no EARLIER row holds.  Give every
// column the same priority order over the rows (the lower index wins) and that allocation is the unique stable matching, which row-proposing
// deferred acceptance reaches from ANY order of proposals: a row proposes down its list; a column keeps the lowest row that ever proposed to it;
// a row that loses a column (rejected at once, or displaced later by a lower row) goes on to ITS next preference and never back up — a column,
// once held, is held by ever lower rows.  The whole mutable state is holder[j] = the lowest row that has proposed to column j, and one
// returning atomicMin IS a proposal: it returns a lower row -> rejected; a higher one -> accepted, and that row is displaced and becomes the
// lane's next proposer; nobody -> the chain ends.  A displaced row's next column needs no per-row state: it is the preference after the column
// just lost.  One lane per row starts a chain; the chains of the channel (a triangle of displaced rows at the end of every grid line: DESIGN §6)
// are followed by the lanes that run into them, thousands at a time, each step two dependent accesses (the atomic, the next row's preference
// list) instead of r02-r04's versioned first-taker repairs (three returning atomics + two store completions per step, 38 GB of look-ups per
// hierarchy).  No fences: holder is touched by device-scope atomics only, everything else is read-only while the kernel runs.
// r02-r04's machinery (slice sweeps, lock-step rounds, cascades) stays as the fallback for a pairing the verification pass rejects or a
// chain that exceeds the step budget — neither has been seen (scripts/analysis/deferred_acceptance.py: the argument, checked on the CPU).
struct DaCounters {
    int overflow;  // chains cut off by the step budget (the fallback then runs)
    int steps;     // proposals made by da_chase_k: statistics
    int list;      // rows the first pass left to the chains
    int scans;     // (statistics: proposals found by a scan of the row, the list having run out)
    int longest;   // (statistics: proposals of the longest chain)
};

// A look at the holder table before proposing: holder[j] only ever DECREASES, so a value below r — however stale the copy a load returns —
// means the proposal would be rejected, and the row passes the column by without an atomic.  (A stale copy errs towards "free": the atomic
// that follows is the authority.)  Relaxed agent-scope loads: past the vector L1, which would never show another CU's atomics.
__device__ __forceinline__ int da_peek(const int *holder, int j) { return __hip_atomic_load(holder + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// First pass: one thread per row, the SELL image (lane = row: coalesced).  The row's most preferred column that no lower row is seen to hold
// (agg_eval_row's scan with the holder table as the first-taker table) gets the row's proposal.  Whoever loses — the row itself, when a lower
// row got in between the look and the atomic; or the higher row it displaces — goes on the list of the chains: (row, the column it lost on).
// The scan also leaves the row's kPrefs most preferred columns behind (agg_init_prefs_k's lists: value ascending, position ascending; bit 30 of the
// last entry: the row has more candidates): a chain that displaces the row finds its next proposal in ONE 16-byte line at a known address.
__global__ __launch_bounds__(kBlock) void da_first_k(MatView A, int *holder, int2 *__restrict__ list, DaCounters *C, int *__restrict__ prefs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int loser = -1, lost_on = -1;
    if (i < A.P.n) {
        const int len = A.P.row_len[i];
        const int64_t base = A.P.slice_ptr[i >> 6] + (i & 63);
        double bv[kPrefs];
        int bj[kPrefs];
#pragma unroll
        for (int q = 0; q < kPrefs; ++q) { bv[q] = 1.7976931348623157e308; bj[q] = -1; }
        double best = 1.7976931348623157e308;  // Float::MAX
        int fj = -1, cand = 0;
        for (int k = 0; k < len; ++k) {
            const int64_t pos = base + (int64_t)k * 64;
            const int j = A.P.col[pos];
            if (j == i || j >= A.P.n) continue;  // ghost columns (partitioned level 0) are never partners
            const double a = view_value(A, i, pos);
            if (!(a < 1.7976931348623157e308)) continue;  // never chosen (nor a NaN)
            ++cand;
            int pos_q = kPrefs;  // its place in the list: in front of the first listed entry it is STRICTLY smaller than
#pragma unroll
            for (int q = kPrefs - 1; q >= 0; --q)
                if (a < bv[q]) pos_q = q;
#pragma unroll
            for (int q = kPrefs - 1; q >= 1; --q)
                if (q > pos_q) { bv[q] = bv[q - 1]; bj[q] = bj[q - 1]; }
#pragma unroll
            for (int q = 0; q < kPrefs; ++q)
                if (q == pos_q) { bv[q] = a; bj[q] = j; }
            if (!(a < best)) continue;
            if (da_peek(holder, j) < (int)i) continue;
            best = a; fj = j;
        }
        int4 pl;
        pl.x = bj[0]; pl.y = bj[1]; pl.z = bj[2];
        pl.w = (cand > kPrefs && bj[3] >= 0) ? (bj[3] | kPrefMore) : bj[3];
        reinterpret_cast<int4 *>(prefs)[i] = pl;
        if (fj >= 0) {
            const int old = atomicMin(&holder[fj], (int)i);
            if (old < (int)i) { loser = (int)i; lost_on = fj; }
            else if (old != 0x7fffffff) { loser = old; lost_on = fj; }
        }
    }
    const int slot = wave_append_slot(&C->list, loser >= 0);
    if (loser >= 0) list[slot] = make_int2(loser, lost_on);
}

// The chains: a group of G lanes takes a listed row and follows what its proposals set off — propose; if a higher row is displaced, go on as
// that row — until a proposal meets a free column or a row runs out of candidates; then it takes the next listed row.  No state but the holder
// table; no group waits for another.
//
// ONE flat loop per wavefront, the same straight-line code for its 64 / G groups in every pass: a group that is through with its chain takes
// its next row in the same pass in which the others make their next step (nested "for every listed row { follow the chain }" leaves a group
// whose chain has ended masked off until the longest chain of its wavefront ends).  Loads are unconditional at clamped addresses (a
// conditional load is a branch with a wait of its own behind it), the results masked.  A pass:
//   1. the row's LIST (da_first_k: its kPrefs most preferred columns): lanes 0-3 look at the holders of the listed columns behind `after`;
//   2. only if some group of the wavefront found them all taken and its row has more candidates: the SCAN of the row — descriptor, then
//      columns and values (<= kDaRegs entries per lane in registers, only the slots the wavefront's longest scanned row needs; longer rows:
//      two sweeps), then every candidate's holder, two reductions;
//   3. the proposal (one returning atomicMin per group) — beside it, already on its way, the list of the row it will most likely displace
//      (the holder just seen).
// Two dependent round trips per step where the list reaches (the channel's fine level: 25 600 chains of ~160 steps, 1.4 ms), five where the
// row is scanned.  What bounds the coarse levels is the LONGEST chain times those trips, not the number of proposals (measured: more groups
// per wavefront, fewer looks per scan — in order of preference, lane by lane — and fewer memory instructions per pass all leave 5-8 ms).
constexpr int kDaRegs = 8;
template <int G>
__global__ __launch_bounds__(kBlock) void da_chase_k(MatView A, int *holder, const int2 *__restrict__ list, DaCounters *C, int max_steps, const int *__restrict__ prefs) {
    const int gl = threadIdx.x & (G - 1);
    const int shift = (threadIdx.x & 63) & ~(G - 1);
    const unsigned long long gmask = G == 64 ? ~0ull : (((1ull << G) - 1ull) << shift);
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / G;
    const int count = C->list;
    const int4 *pl4 = reinterpret_cast<const int4 *>(prefs);
    const int n = (int)A.P.n;
    const bool mirror = A.rows.col != nullptr;
    const int32_t *colp = mirror ? A.rows.col : A.P.col;
    const double *valp = mirror ? A.rows.val : A.val;
    const int64_t stride = mirror ? 1 : 64;
    int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    bool have = e < count;
    int r = 0, after = -1;
    int4 pl = make_int4(-1, -1, -1, -1);
    if (have) {
        const int2 it = list[e];
        r = it.x; after = it.y;
        pl = pl4[r];
    }
    int steps = 0, scans = 0, chain_steps = 0, longest = 0;
    bool cut = false;
    while (__ballot(have) != 0ull) {
        // ---- 1. the list
        const int p3 = pl.w >= 0 ? (pl.w & ~kPrefMore) : -1;
        const bool more = pl.w >= 0 && (pl.w & kPrefMore) != 0;
        const int pq = gl == 0 ? pl.x : (gl == 1 ? pl.y : (gl == 2 ? pl.z : (gl == 3 ? p3 : -1)));
        const int first = after == pl.x ? 1 : (after == pl.y ? 2 : (after == pl.z ? 3 : 4));  // the first listed preference behind `after`
        const bool beyond = first == 4 && after != p3;                                         // `after` lies behind the whole list already
        const bool look = have && !beyond && gl < kPrefs && gl >= first && pq >= 0;
        int h = da_peek(holder, look ? pq : 0);
        h = look ? h : -1;
        const unsigned long long free_all = __ballot(look && h >= r);
        const unsigned free_mine = (unsigned)((free_all & gmask) >> shift);
        int cand = -1, seen = 0x7fffffff;
        if (free_mine) {
            const int q = __ffs((int)free_mine) - 1;
            cand = __shfl(pq, q, G);
            seen = __shfl(h, q, G);
        }
        const bool need_scan = have && (beyond || (!free_mine && more));
        const int scan_after = beyond ? after : p3;  // (a row's unlisted candidates all rank behind its last listed one)
        // ---- 2. the scan, for the groups whose list ran out (wave-uniform branch; inside, every lane runs the same code)
        if (__ballot(need_scan) != 0ull) {
            const int rs = need_scan ? r : 0;
            int len = A.P.row_len[rs];
            const int64_t base = mirror ? (int64_t)A.rows.slice_base[rs >> 6] + A.rows.intra_off[rs] : A.P.slice_ptr[rs >> 6] + (rs & 63);
            const double s1 = A.s1 ? A.s1[rs] : 1., s2 = A.s2 ? A.s2[rs] : 1.;
            len = need_scan ? len : 0;
            int k_c = -1;
            double v_c = 0.;
            double best = 1.7976931348623157e308;  // Float::MAX
            int bk = 0x7fffffff, bj = -1, bh = 0x7fffffff;
            int len_max = len;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) len_max = max(len_max, __shfl_xor(len_max, off, 64));
            if (len_max <= G * kDaRegs) {
                const int u_max = (len_max + G - 1) / G;  // wave-uniform
                int cj[kDaRegs], hp[kDaRegs];
                double cv[kDaRegs];
#pragma unroll
                for (int u = 0; u < kDaRegs; ++u) {
                    cj[u] = -1;
                    if (u < u_max) {
                        const int k = gl + u * G;
                        const int c = colp[base + (int64_t)(k < len ? k : 0) * stride];
                        cj[u] = k < len ? c : -1;
                    }
                }
#pragma unroll
                for (int u = 0; u < kDaRegs; ++u) {
                    cv[u] = 0.;
                    if (u < u_max) {
                        const int k = gl + u * G;
                        cv[u] = valp[base + (int64_t)(k < len ? k : 0) * stride];
                    }
                }
#pragma unroll
                for (int u = 0; u < kDaRegs; ++u) {
                    hp[u] = -1;
                    if (u >= u_max) continue;
                    const bool ok = cj[u] >= 0 && cj[u] != rs && cj[u] < n;
                    const int hh = da_peek(holder, ok ? cj[u] : 0);
                    hp[u] = ok ? hh : -1;  // (-1: not a candidate)
                    double t = cv[u];
                    if (A.s1) t = s1 * t;  // RowWalk::value's order
                    if (A.s2) t = s2 * t;
                    cv[u] = t;
                    if (cj[u] == scan_after && cj[u] >= 0) { k_c = gl + u * G; v_c = t; }
                }
#pragma unroll
                for (int off = G / 2; off > 0; off >>= 1) {
                    const int ok = __shfl_xor(k_c, off, G);
                    const double ov = __shfl_xor(v_c, off, G);
                    if (ok > k_c) { k_c = ok; v_c = ov; }
                }
#pragma unroll
                for (int u = 0; u < kDaRegs; ++u) {  // (a lane's positions ascend with u: strict < keeps the earlier one)
                    const int k = gl + u * G;
                    const double a = cv[u];
                    if (hp[u] < rs) continue;                                         // not a candidate, or held by a lower row
                    if (k_c >= 0 && !(a > v_c || (a == v_c && k > k_c))) continue;    // at or before `scan_after`: refused already
                    if (a < best) { best = a; bk = k; bj = cj[u]; bh = hp[u]; }
                }
            } else {  // rows beyond G * kDaRegs entries: the same in two sweeps over the row
                const int sweeps = (len_max + G - 1) / G;
                for (int u = 0; u < sweeps; ++u) {
                    const int k = gl + u * G;
                    const int64_t pos = base + (int64_t)(k < len ? k : 0) * stride;
                    const int c = colp[pos];
                    double t = valp[pos];
                    if (A.s1) t = s1 * t;
                    if (A.s2) t = s2 * t;
                    if (k < len && c == scan_after) { k_c = k; v_c = t; }
                }
#pragma unroll
                for (int off = G / 2; off > 0; off >>= 1) {
                    const int ok = __shfl_xor(k_c, off, G);
                    const double ov = __shfl_xor(v_c, off, G);
                    if (ok > k_c) { k_c = ok; v_c = ov; }
                }
                for (int u = 0; u < sweeps; ++u) {
                    const int k = gl + u * G;
                    const int64_t pos = base + (int64_t)(k < len ? k : 0) * stride;
                    const int c = colp[pos];
                    double t = valp[pos];
                    if (A.s1) t = s1 * t;
                    if (A.s2) t = s2 * t;
                    const bool ok = k < len && c != rs && c < n;
                    const int hh = da_peek(holder, ok ? c : 0);
                    if (!ok || hh < rs) continue;
                    if (k_c >= 0 && !(t > v_c || (t == v_c && k > k_c))) continue;
                    if (t < best) { best = t; bk = k; bj = c; bh = hh; }
                }
            }
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, G);
                const int ok = __shfl_xor(bk, off, G);
                const int oj = __shfl_xor(bj, off, G);
                const int oh = __shfl_xor(bh, off, G);
                if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; bj = oj; bh = oh; }
            }
            if (need_scan) { cand = bj; seen = bh; ++scans; }
        }
        // ---- 3. the proposal; beside it the list of the row it will most likely displace
        const bool propose = have && cand >= 0;
        int old = 0;
        if (gl == 0 && propose) old = atomicMin(&holder[cand], r);
        const int guess = (propose && seen != 0x7fffffff && seen > r) ? seen : r;
        const int4 pl_guess = pl4[guess];
        old = __shfl(old, 0, G);
        bool done = have && !propose;  // the row has no candidate left: unmatched, the chain ends
        if (propose) {
            ++steps;
            ++chain_steps;
            after = cand;
            if (old > r) {
                if (old == 0x7fffffff) done = true;  // a free column: the chain ends
                else {
                    r = old;                         // accepted; `old` is displaced and goes on from the column it lost
                    pl = old == guess ? pl_guess : pl4[old];
                }
            }
            if (!done && chain_steps >= max_steps) { cut = true; done = true; }
        }
        // ---- the next listed row, in the same pass
        if (done) {
            longest = max(longest, chain_steps);
            e += groups;
            have = e < count;
            chain_steps = 0;
            if (have) {
                const int2 it = list[e];
                r = it.x; after = it.y;
                pl = pl4[r];
            }
        }
    }
    if (gl == 0) {
        if (cut) atomicAdd(&C->overflow, 1);
        if (steps) atomicAdd(&C->steps, steps);
        if (scans) atomicAdd(&C->scans, scans);
        if (longest) atomicMax(&C->longest, longest);
    }
}

// holder -> the pairing: chooser[j] = the row that holds column j (-1: nobody), choice[that row] = j (choice preset to -1)
__global__ void da_finish_k(const int *__restrict__ holder, int *__restrict__ choice, int *__restrict__ chooser, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const int h = holder[j];
        chooser[j] = h == 0x7fffffff ? -1 : h;
        if (h != 0x7fffffff) choice[h] = (int)j;  // a row holds one column at most
    }
}

SiblingPairing::~SiblingPairing() {
    for (auto &e : ready)
        if (e) (void)hipEventDestroy(e);
    for (auto &o : offer)
        if (o.view_ready) (void)hipEventDestroy(o.view_ready);
    if (ops_ready) (void)hipEventDestroy(ops_ready);
}
void SiblingPairing::begin(bool leader_will_run) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &p : published) p = false;
    leader_done = !leader_will_run;
    expected_offers = 0;
    for (auto &o : offer) { o.made = o.ok = o.answered = o.adopted = false; o.view = nullptr; o.arena = o.rows_arena = nullptr; o.level = nullptr; }
    lead_choice = lead_chooser = nullptr;
}
int SiblingPairing::publish(int level, const int *choice, int64_t rows, hipStream_t stream) {
    if (level < 0 || level >= kLevels) return ORC_OK;
    int st = ORC_OK;
    if (buf[level].n < (size_t)std::max<int64_t>(rows, 1)) st = buf[level].alloc((size_t)std::max<int64_t>(rows, 1));
    if (st == ORC_OK && !ready[level] && hipEventCreateWithFlags(&ready[level], hipEventDisableTiming) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventCreate failed");
    if (st == ORC_OK && rows > 0 && hipMemcpyAsync(buf[level].p, choice, sizeof(int) * (size_t)rows, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        st = set_error(ORC_ERR_HIP, "hipMemcpyAsync failed");
    if (st == ORC_OK && hipEventRecord(ready[level], stream) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    {
        std::lock_guard<std::mutex> lk(mu);
        if (st == ORC_OK) { n[level] = rows; published[level] = true; }
        else leader_done = true;  // nobody waits for a level that will not come
    }
    cv.notify_all();
    return st;
}
const int *SiblingPairing::wait(int level, int64_t rows, hipStream_t stream) {
    if (level < 0 || level >= kLevels) return nullptr;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return published[level] || leader_done; });
    if (!published[level] || n[level] != rows) return nullptr;
    if (hipStreamWaitEvent(stream, ready[level], 0) != hipSuccess) return nullptr;
    return buf[level].p;
}
void SiblingPairing::finish() {
    {
        std::lock_guard<std::mutex> lk(mu);
        leader_done = true;
    }
    cv.notify_all();
}
void SiblingPairing::set_expected(int n) {
    std::lock_guard<std::mutex> lk(mu);
    expected_offers = n;
}
int SiblingPairing::make_offer(int slot, const MatView *view, Arena *arena, Arena *rows_arena, void *level, hipStream_t stream) {
    if (slot < 0 || slot > 1) return ORC_OK;
    int st = ORC_OK;
    Offer &o = offer[slot];
    if (!o.view_ready && hipEventCreateWithFlags(&o.view_ready, hipEventDisableTiming) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventCreate failed");
    if (st == ORC_OK && hipEventRecord(o.view_ready, stream) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    {
        std::lock_guard<std::mutex> lk(mu);
        o.made = true;
        o.ok = st == ORC_OK;
        o.view = view; o.arena = arena; o.rows_arena = rows_arena; o.level = level;
    }
    cv.notify_all();
    return st;
}
void SiblingPairing::withdraw(int slot) {
    if (slot < 0 || slot > 1) return;
    {
        std::lock_guard<std::mutex> lk(mu);
        offer[slot].made = true;
        offer[slot].ok = false;
    }
    cv.notify_all();
}
bool SiblingPairing::wait_answer(int slot, hipStream_t stream) {
    if (slot < 0 || slot > 1) return false;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return offer[slot].answered || leader_done; });
    if (!offer[slot].answered || !offer[slot].adopted) return false;
    return hipStreamWaitEvent(stream, ops_ready, 0) == hipSuccess;
}
int SiblingPairing::collect_offers(Offer *out[2]) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return (int)offer[0].made + (int)offer[1].made >= expected_offers; });
    int n = 0;
    for (int q = 0; q < 2; ++q)
        if (offer[q].made && offer[q].ok) out[n++] = &offer[q];
    return n;
}
int SiblingPairing::answer(const bool adopted[2], const int *choice, const int *chooser, hipStream_t stream) {
    int st = ORC_OK;
    if (!ops_ready && hipEventCreateWithFlags(&ops_ready, hipEventDisableTiming) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventCreate failed");
    if (st == ORC_OK && hipEventRecord(ops_ready, stream) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    {
        std::lock_guard<std::mutex> lk(mu);
        lead_choice = choice; lead_chooser = chooser;
        for (int q = 0; q < 2; ++q) {
            offer[q].answered = true;
            offer[q].adopted = st == ORC_OK && adopted[q];
        }
    }
    cv.notify_all();
    return st;
}

// orc_debug_amg_certification: aggregations whose pairing was certified, and how many certifying passes / sweeps that took in total (equal =
// every certification found nothing to change: the deferred-acceptance chains had reached the fixed point by themselves)
static std::atomic<long long> g_cert_aggregations{0}, g_cert_rounds{0};
void debug_amg_certification(long long out[2], bool reset) {
    out[0] = g_cert_aggregations.load(std::memory_order_relaxed);
    out[1] = g_cert_rounds.load(std::memory_order_relaxed);
    if (reset) { g_cert_aggregations.store(0); g_cert_rounds.store(0); }
}
int debug_xwin_counters(long long out[3], bool reset) {
    unsigned long long h[3] = {0, 0, 0};
    ORC_HIP(hipDeviceSynchronize());
    ORC_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xwin_counters), sizeof(h)));
    for (int i = 0; i < 3; ++i) out[i] = (long long)h[i];
    if (reset) {
        const unsigned long long z[3] = {0, 0, 0};
        ORC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_xwin_counters), z, sizeof(z)));
    }
    return ORC_OK;
}

// build_restriction_matrix's pairing (linear_algebra.rs:30-60) for the matrix behind `A`: choice[i] = the column row i takes (-1: none),
// chooser[j] = the row that took column j (-1: nobody).
//   warm (optional): a sibling system's pairing of THIS iteration — taken if it IS this matrix's fixed point (agg_verify_k: one pass, nothing
//     to iterate), dropped otherwise (r02 measured: a pairing that is off in a few per cent of the rows is a worse start than none).
//   1. deferred acceptance (da_first_k, da_chase_k): the pairing, read off the holder table, certified by ONE verification pass and ONE host read;
//   2. only if that pass finds a row that would choose differently, or a chain was cut by the step budget (ORC_AMG_DA_STEPS: a test hook) —
//      never seen otherwise —, or with ORC_AMG_DA=0: slice-sequential sweeps against the rebuilt first-taker table until a sweep changes
//      nothing.  Slow (one sweep per slice a chain crosses) and as simple as the reference's loop: a fallback, not a path to tune.  r02-r04's
//      lock-step rounds and asynchronous cascades (HISTORY.md) are gone with the round that made them unnecessary.
static int aggregate(const MatView &A, Arena &arena, int *choice, int *chooser, int *rounds_out, const int *warm = nullptr) {
    const int64_t n = A.P.n;
    const int g = grid_for(n);
    const int gs = grid_for(A.P.n_slices, 64);  // one thread per slice, 64-thread workgroups spread the slices over the CUs
    int *taken_by, *snap;
    AggCounters *C;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &taken_by));
    ORC_TRY(arena.alloc((size_t)1, &C));
    ORC_TRY(arena.alloc((size_t)64, &snap));
    hipStream_t st = ctx().stream;
    const bool trace = cfg().amg_trace;
    ORC_HIP(hipMemsetAsync(C, 0, sizeof(AggCounters), st));
    if (n == 0) { if (rounds_out) *rounds_out = 0; return ORC_OK; }
    auto finish_from_choice = [&]() -> int {
        ORC_HIP(hipMemsetAsync(chooser, 0xff, sizeof(int) * (size_t)n, st));
        hipLaunchKernelGGL(chooser_k, dim3(g), dim3(kBlock), 0, st, (const int *)choice, chooser, n);
        ORC_HIP(hipGetLastError());
        return ORC_OK;
    };
    if (warm) {  // a sibling's pairing: this matrix's too?
        AggCounters hc;
        ORC_HIP(hipMemcpyAsync(choice, warm, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(agg_reset_k, dim3(g), dim3(kBlock), 0, st, taken_by, n);
        hipLaunchKernelGGL(agg_scatter_k, dim3(g), dim3(kBlock), 0, st, (const int *)choice, taken_by, n);
        hipLaunchKernelGGL(agg_verify_k, dim3(g), dim3(kBlock), 0, st, A, (const int *)choice, (const int *)taken_by, C);
        ORC_HIP(hipGetLastError());
        ORC_HIP(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        if (trace) fprintf(stderr, "[amg sibling n=%lld] rows that would change: %d\n", (long long)n, hc.changed);
        if (hc.changed == 0) {
            if (rounds_out) *rounds_out = 1;
            return finish_from_choice();
        }
        ORC_HIP(hipMemsetAsync(C, 0, sizeof(AggCounters), st));  // not this matrix's pairing: from scratch
    }
    if (cfg().amg_da) {
        ArenaScope da_scope(arena);  // the list is dead when the pairing is known
        DaCounters *D;
        int2 *list;
        int *da_prefs;
        ORC_TRY(arena.alloc((size_t)1, &D));
        ORC_TRY(arena.alloc((size_t)n, &list));
        ORC_TRY(arena.alloc((size_t)n * kPrefs, &da_prefs));
        ORC_HIP(hipMemsetAsync(D, 0, sizeof(DaCounters), st));
        hipLaunchKernelGGL(agg_reset_k, dim3(g), dim3(kBlock), 0, st, taken_by, n);  // holder = taken_by: nobody
        hipLaunchKernelGGL(da_first_k, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, A, taken_by, list, D, da_prefs);
        // lanes per chain: the list needs four; a scan reads the row G entries at a time (kDaRegs slots per lane in registers).  Measured per level
        // of the channel (one hierarchy, us): 7 entries per row: 4 / 8 / 16 lanes 1 620 / 1 454 / 1 475; 15: 4 767 / 6 303 / 8 274; 33: 9 048 / 8 047 / 8 596
        const double da_avg = (double)A.P.padded / (double)n;
        const int da_group = cfg().amg_da_group > 0 ? cfg().amg_da_group : (da_avg <= 24. ? 4 : 8);
        const int da_steps = cfg().amg_da_steps;
        if (da_group == 4) hipLaunchKernelGGL(da_chase_k<4>, dim3(kMaxGrid), dim3(kBlock), 0, st, A, taken_by, (const int2 *)list, D, da_steps, (const int *)da_prefs);
        else if (da_group == 8) hipLaunchKernelGGL(da_chase_k<8>, dim3(kMaxGrid), dim3(kBlock), 0, st, A, taken_by, (const int2 *)list, D, da_steps, (const int *)da_prefs);
        else hipLaunchKernelGGL(da_chase_k<16>, dim3(kMaxGrid), dim3(kBlock), 0, st, A, taken_by, (const int2 *)list, D, da_steps, (const int *)da_prefs);
        ORC_HIP(hipMemsetAsync(choice, 0xff, sizeof(int) * (size_t)n, st));
        hipLaunchKernelGGL(da_finish_k, dim3(g), dim3(kBlock), 0, st, (const int *)taken_by, choice, chooser, n);
        // is it the fixed point?  every row against the exact first-taker table (= holder): the sequential pairing is the only state that passes
        hipLaunchKernelGGL(agg_verify_k, dim3(g), dim3(kBlock), 0, st, A, (const int *)choice, (const int *)taken_by, C);
        ORC_HIP(hipGetLastError());
        AggCounters hc;
        DaCounters hd;
        ORC_HIP(hipMemcpyAsync(&hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipMemcpyAsync(&hd, D, sizeof(hd), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        if (trace) fprintf(stderr, "[amg da n=%lld] rows left to the chains %d, their proposals %d (%d by a scan of the row), longest chain %d, chains cut %d, rows that would change %d\n",
                           (long long)n, hd.list, hd.steps, hd.scans, hd.longest, hd.overflow, hc.changed);
        if (hd.overflow == 0 && hc.changed == 0) {
            g_cert_aggregations.fetch_add(1, std::memory_order_relaxed);  // certified by one pass that changed nothing
            g_cert_rounds.fetch_add(1, std::memory_order_relaxed);
            if (rounds_out) *rounds_out = 1;
            return ORC_OK;
        }
        ORC_HIP(hipMemsetAsync(C, 0, sizeof(AggCounters), st));
    }
    // ---- the fallback: sweeps from the unconstrained arg-min state, four per host read
    hipLaunchKernelGGL(agg_init_k, dim3(g), dim3(kBlock), 0, st, A, choice);
    constexpr int kBulk = 4;
    int rounds = 0;
    for (bool done = false; !done;) {
        for (int b = 0; b < kBulk; ++b) {
            hipLaunchKernelGGL(agg_reset_k, dim3(g), dim3(kBlock), 0, st, taken_by, n);
            hipLaunchKernelGGL(agg_scatter_k, dim3(g), dim3(kBlock), 0, st, choice, taken_by, n);
            hipLaunchKernelGGL(agg_sweep_k, dim3(gs), dim3(64), 0, st, A, choice, taken_by, C);
            hipLaunchKernelGGL(agg_rotate_k, dim3(1), dim3(1), 0, st, C, snap + b);
        }
        ORC_HIP(hipGetLastError());
        int h[kBulk];
        ORC_HIP(hipMemcpyAsync(h, snap, sizeof(int) * kBulk, hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < kBulk; ++b) {
            ++rounds;
            if (h[b] == 0) { done = true; break; }  // a sweep that changed nothing has evaluated every row against the exact table
        }
        if (rounds > 8 * 1000 * 1000) return set_error(ORC_ERR_BAD_ARGUMENT, "aggregation did not reach its fixed point");
    }
    if (trace) fprintf(stderr, "[amg fallback n=%lld] %d sweeps\n", (long long)n, rounds);
    g_cert_aggregations.fetch_add(1, std::memory_order_relaxed);
    g_cert_rounds.fetch_add(rounds, std::memory_order_relaxed);
    if (rounds_out) *rounds_out = rounds;
    return finish_from_choice();
}

// Row-contiguous mirror, compacted: the product's scratch rows (reserved at twice the candidate count per row: the bound of the
// symbolic step, about 3.7 times what the rows really hold) copied to exact size — entry k of coarse row r at
// new_base[r >> 6] + new_intra[r] + k — so that the scratch can be handed back.  One wavefront per slice of 64 coarse rows.
__global__ __launch_bounds__(64) void rows_compact_k(const int *__restrict__ row_len, int64_t n_rows, int n_slices, const long long *__restrict__ old_base,
                                                     const int *__restrict__ old_intra, const int *__restrict__ s_col, const double *__restrict__ s_val,
                                                     const int64_t *__restrict__ new_base, int *__restrict__ new_intra, int *__restrict__ out_col,
                                                     double *__restrict__ out_val) {
    const int lane = threadIdx.x;
    for (int64_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        const int64_t row = slice * 64 + lane;
        const bool live = row < n_rows;
        const int len = live ? row_len[row] : 0;
        int incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        const int excl = incl - len;
        if (live) new_intra[row] = excl;
        const long long ob = old_base[slice];
        const int64_t nb = new_base[slice];
        const int oi = live ? old_intra[row] : 0;
        for (int r = 0; r < 64; ++r) {
            const int n = __shfl(len, r, 64);
            const long long src = ob + __shfl(oi, r, 64);
            const int64_t dst = nb + __shfl(excl, r, 64);
            for (int e = lane; e < n; e += 64) {
                out_col[dst + e] = s_col[src + e];
                out_val[dst + e] = s_val[src + e];
            }
        }
    }
}

// `scratch` (optional): a second arena for everything that is dead when the level is complete — the symbolic bounds, the tier
// lists and the product's scratch rows, 11 GB of a 10.24 M-row hierarchy's 23 GB — released before returning; the row-contiguous
// mirror is then a compacted copy in `arena`.  Without it the scratch rows themselves stay alive as the mirror (round 2).
// `sib` / n_sib [r04]: sibling systems on A's pattern that verified THIS pairing as theirs (SiblingPairing): their coarse operators share
// the pattern this product builds — one symbolic pass, n_sib + 1 value sets (MergeSiblings) — and get their own values in their own
// arenas; every kernel runs on the calling thread's stream.  Needs `scratch` (the mirrors are then compacted copies, one per system).
struct GalerkinSibling {
    const MatView *A = nullptr;  // same pattern (and mirror structure) as the leader's view, its own values and scalings
    Arena *arena = nullptr;      // the sibling's hierarchy arena: its coarse values
    Arena *rows_arena = nullptr; // the companion of the sibling's scratch arena: its copy of the transient row-contiguous mirror
    CoarseLevel *L = nullptr;
};
static std::atomic<long long> g_shared_galerkin{0};  // sibling operators built by a shared pass (orc_debug_shared_galerkin)
long long debug_shared_galerkin(bool reset) {
    const long long v = g_shared_galerkin.load(std::memory_order_relaxed);
    if (reset) g_shared_galerkin.store(0, std::memory_order_relaxed);
    return v;
}

static int galerkin(const MatView &A, const int *choice, const int *chooser, Arena &arena, CoarseLevel &L, Arena *scratch = nullptr, bool last_level = false,
                    const GalerkinSibling *sib = nullptr, int n_sib = 0) {
    if (n_sib < 0 || n_sib > 2 || (n_sib > 0 && (!scratch || !sib))) return set_error(ORC_ERR_BAD_ARGUMENT, "galerkin: bad sibling arguments");
    Arena &tmp = scratch ? *scratch : arena;
    ArenaScope tmp_scope(tmp);  // with `scratch`: unwinds it on every exit; without: re-marked below so that nothing is released
    const int64_t n = A.P.n, nc = n / 2 + n % 2;  // :13
    hipStream_t st = ctx().stream;
    int *row_len, *diag, *flags, *intra_off;  // flags[0] = max candidates, [1] = overflow (cannot happen: rows are pre-sorted into tiers)
    long long *slice_tot, *slice_base;
    unsigned long long *counters;  // [1] = sum of candidates
    int64_t *slice_ptr;
    const int n_slices = (int)((nc + 63) / 64);
    const size_t ncs = (size_t)std::max<int64_t>(nc, 1);
    const bool trace_t = cfg().amg_trace;
    double t_mark = 0.;
    auto lap = [&](const char *what) {  // trace only: wall time of the phase that just ended (drains the stream)
        if (!trace_t) return;
        (void)hipStreamSynchronize(st);
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (what) fprintf(stderr, "[amg phase n=%lld] %s %.3f ms\n", (long long)n, what, now - t_mark);
        t_mark = now;
    };
    lap(nullptr);
    ORC_TRY(arena.alloc(ncs, &row_len));
    ORC_TRY(arena.alloc(ncs, &diag));
    ORC_TRY(tmp.alloc(ncs, &intra_off));
    ORC_TRY(tmp.alloc((size_t)n_slices + 1, &slice_tot));
    ORC_TRY(tmp.alloc((size_t)n_slices + 1, &slice_base));
    ORC_TRY(arena.alloc((size_t)n_slices + 1, &slice_ptr));
    ORC_TRY(tmp.alloc((size_t)4, &flags));
    ORC_TRY(tmp.alloc((size_t)2, &counters));
    ORC_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), st));
    ORC_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(unsigned long long), st));
    // lanes per coarse row by LDS tier (list capacity 64 << t); ORC_GALERKIN_GROUPS="g0,g1,..." overrides
    const std::array<int, kGalerkinTiers> tier_group = [] {
        std::array<int, kGalerkinTiers> g = {16, 16, 32, 64, 64, 64, 64};  // measured at 10.24 M fine rows (levels of 7 / 15 / 34 entries per row)
        if (!cfg().galerkin_groups.empty()) {
            const char *e = cfg().galerkin_groups.c_str();
            int t = 0;
            for (const char *q = e; *q && t < kGalerkinTiers; ++t) {
                const int v = atoi(q);
                if (v == 16 || v == 32 || v == 64) g[t] = v;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        return g;
    }();
    int *tier_count, *tier_list;
    ORC_TRY(tmp.alloc((size_t)kGalerkinTiers + 1, &tier_count));
    ORC_TRY(tmp.alloc((size_t)kGalerkinTiers * ncs, &tier_list));
    ORC_HIP(hipMemsetAsync(tier_count, 0, (kGalerkinTiers + 1) * sizeof(int), st));
    // (a wavefront walks at most kBoundIters slices: the grid grows with the level beyond 8 192 x kBoundIters slices = 16.8 M coarse rows)
    const int64_t bound_grid = std::max<int64_t>(std::min<int64_t>(std::max<int64_t>(n_slices, 1), 8192), ((int64_t)n_slices + kBoundIters - 1) / kBoundIters);
    hipLaunchKernelGGL(galerkin_bound_k, dim3((unsigned)bound_grid), dim3(64), 0, st, A.P, choice, nc, flags, counters + 1, intra_off,
                       slice_tot, tier_count, tier_list, 0);
    long long *scan_part;
    ORC_TRY(tmp.alloc((size_t)2 * kScanBlocks + 2, &scan_part));
    ORC_TRY(scan_excl_dev(slice_tot, nullptr, (int64_t)n_slices, slice_base, nullptr, false, scan_part, st));
    int hflags[4];
    unsigned long long hcount[2];
    ORC_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    int htier[kGalerkinTiers];
    ORC_HIP(hipMemcpyAsync(hcount, counters, sizeof(hcount), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(htier, tier_count, sizeof(htier), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    const int max_cand = std::max(hflags[0], 1);
    const long long scratch_cap = (long long)std::max<unsigned long long>(2ull * hcount[1], 64ull);
    int *s_col;
    double *s_val;
    ORC_TRY(tmp.alloc((size_t)scratch_cap, &s_col));
    ORC_TRY(tmp.alloc((size_t)scratch_cap, &s_val));
    MergeSiblings X;
    const bool fine_mirror = A.rows.col != nullptr;
    for (int x = 0; x < n_sib; ++x) {
        const MatView &B = *sib[x].A;
        if (B.P.n != n || B.P.col != A.P.col || (B.rows.col != nullptr) != fine_mirror || (fine_mirror && B.rows.col != A.rows.col))
            return set_error(ORC_ERR_BAD_ARGUMENT, "galerkin: a sibling system does not share the leader's pattern");
        X.val[x] = fine_mirror ? B.rows.val : B.val;
        X.s1[x] = B.s1;
        X.s2[x] = B.s2;
        ORC_TRY(tmp.alloc((size_t)scratch_cap, &X.s_val[x]));
    }
    lap("galerkin bounds");
    static std::once_flag attr_once;  // several lane threads reach this concurrently
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_merge_k<16, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    // LDS tiers (32 B per list slot): every row was assigned to the narrowest list that is guaranteed to hold it
    if ((size_t)2 * max_cand > (size_t)(64 << (kGalerkinTiers - 1))) return set_error(ORC_ERR_BAD_ARGUMENT, "Galerkin row too long for LDS (%d candidates)", max_cand);
    for (int t = 0; t < kGalerkinTiers; ++t) {
        if (htier[t] == 0) continue;
        const int cap = 64 << t;
        int G = tier_group[t];  // narrow rows: two or four coarse rows per wavefront
        // a wavefront's lists must fit the LDS of one workgroup: wider groups (fewer rows per wavefront) where the sibling value sets would not
        while (G < 64 && (size_t)cap * (size_t)(12 + 8 * n_sib) * (size_t)(64 / G) > (size_t)150 * 1024) G <<= 1;
        const int rows_per_wave = 64 / G;
        const size_t smem = (size_t)cap * (size_t)(12 + 8 * n_sib) * (size_t)rows_per_wave;
        if (smem > (size_t)160 * 1024) return set_error(ORC_ERR_BAD_ARGUMENT, "Galerkin row too long for a shared pass (%d candidates, %d systems)", max_cand, n_sib + 1);
        // 83-88 VGPRs: five wavefronts per SIMD are resident.  Measured (one stream, all tiers of six SIMPLE iterations): 16 per CU 407 ms,
        // 20: 372 ms, 24: 410 ms (the launch no longer fits and its tail runs alone); round 2's kernel at 16: 438 ms
        // [r04] 80 VGPRs (amdgpu_waves_per_eu(6): 12-24 bytes of scratch per lane) and 12 instead of 16 bytes of LDS per list slot: six wavefronts per
        // SIMD are resident, 24 per CU (r03: 83-88 VGPRs, five per SIMD; 16 per CU 407 ms over six iterations, 20: 372 ms, 24: 410 ms as the launch
        // no longer fitted); the shared pass for sibling systems carries three value sets in 128 VGPRs: four per SIMD
        const int merge_waves = n_sib == 0 ? 24 : 16;
        const int waves_per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)merge_waves, (size_t)(150 * 1024) / smem));
        const int g = (int)std::min<int64_t>(((int64_t)htier[t] + rows_per_wave - 1) / rows_per_wave, (int64_t)256 * waves_per_cu);
        const int *tl = tier_list + (int64_t)t * nc, *tc = tier_count + t;
#define ORC_MERGE(GG, NS) hipLaunchKernelGGL(HIP_KERNEL_NAME(galerkin_merge_k<GG, NS>), dim3(g), dim3(64), smem, st, A, choice, chooser, nc, cap, row_len, slice_base, intra_off, s_col, s_val, tl, tc, X)
        if (n_sib == 0) {
            if (G == 16) ORC_MERGE(16, 1);
            else if (G == 32) ORC_MERGE(32, 1);
            else ORC_MERGE(64, 1);
        } else if (n_sib == 1) {
            if (G == 16) ORC_MERGE(16, 2);
            else if (G == 32) ORC_MERGE(32, 2);
            else ORC_MERGE(64, 2);
        } else {
            if (G == 16) ORC_MERGE(16, 3);
            else if (G == 32) ORC_MERGE(32, 3);
            else ORC_MERGE(64, 3);
        }
#undef ORC_MERGE
    }
    lap("galerkin product");
    int64_t *pk_ptr, *w_sell, *w_pk;
    ORC_TRY(arena.alloc((size_t)n_slices + 1, &pk_ptr));
    ORC_TRY(tmp.alloc((size_t)n_slices + 1, &w_sell));
    ORC_TRY(tmp.alloc((size_t)n_slices + 1, &w_pk));
    hipLaunchKernelGGL(slice_sizes_k, dim3((unsigned)std::min<int64_t>(((int64_t)n_slices + 3) / 4, 4096)), dim3(kBlock), 0, st, row_len, nc, n_slices, w_sell, w_pk);
    static_assert(sizeof(long long) == sizeof(int64_t), "64-bit tables");
    ORC_TRY(scan_excl_dev(reinterpret_cast<const long long *>(w_sell), reinterpret_cast<const long long *>(w_pk), (int64_t)n_slices, reinterpret_cast<long long *>(slice_ptr),
                          reinterpret_cast<long long *>(pk_ptr), true, scan_part, st));
    ORC_HIP(hipGetLastError());
    int64_t padded = 0, packed_total = 0;
    ORC_HIP(hipMemcpyAsync(&packed_total, pk_ptr + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(&padded, slice_ptr + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    if (hflags[1]) return set_error(ORC_ERR_BAD_ARGUMENT, "Galerkin overflow (%d)", hflags[1]);
    if (padded >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "coarse matrix too large for 32-bit offsets");
    int *col;
    double *val;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &col));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &val));
    SellDev Pc;
    Pc.n = nc; Pc.ncols = nc; Pc.n_slices = n_slices; Pc.ragged = padded < 24 * nc ? 2 : 1; Pc.padded = padded; Pc.slice_ptr = slice_ptr; Pc.row_len = row_len; Pc.col = col; Pc.diag_pos = diag;
    // Packed mirror + LDS x windows for the levels whose rows are long enough for a window to be re-used (measured at
    // 10.24 M fine rows: 33 entries per row +2 %, 70 entries per row +17 % against the padded product; 15 entries per row
    // -10 %, so that level keeps the padded image).  ORC_SPMV_XWIN_MIN_NNZ < 0 switches the mirror off.
    // [r05] ... unless the padded image wastes what the windows cost: config 5's level 1 (17 entries per row, rows of a tet / hex / polyhedral mesh paired:
    // 29 % of its SELL image is padding) 165 -> 126 us with the mirror, the whole iteration 516 -> 479 ms (scripts/archive/gpu_r05_r.sh); the channel's level 1
    // (5.7 % padding) 198 -> 222 us.  From 15 % padding on, a level of at least half the entries per row takes the mirror too.
    const int xwin_min = cfg().spmv_xwin_min_nnz;
    const bool long_rows = packed_total >= (int64_t)xwin_min * nc;
    const bool ragged_rows = 2 * packed_total >= (int64_t)xwin_min * nc && (double)padded >= 1.15 * (double)packed_total;
    const bool mirror = xwin_min >= 0 && packed_total > 0 && (long_rows || ragged_rows);
    int *pk_col = nullptr;
    double *pk_val = nullptr;
    if (mirror) {
        ORC_TRY(arena.alloc((size_t)packed_total, &pk_col));
        ORC_TRY(arena.alloc((size_t)packed_total, &pk_val));
    }
    hipLaunchKernelGGL(galerkin_pack_fused_k<false>, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_slices, 256 * 12))), dim3(64), 0, st, Pc, slice_base, intra_off,
                       s_col, s_val, col, val, diag, mirror ? (const int64_t *)pk_ptr : (const int64_t *)nullptr, pk_col, pk_val);
    ORC_HIP(hipGetLastError());
    double *x_val[2] = {nullptr, nullptr}, *x_pk_val[2] = {nullptr, nullptr};
    for (int x = 0; x < n_sib; ++x) {  // the siblings' values on the same images
        ORC_TRY(sib[x].arena->alloc((size_t)std::max<int64_t>(padded, 1), &x_val[x]));
        if (mirror) ORC_TRY(sib[x].arena->alloc((size_t)packed_total, &x_pk_val[x]));
        hipLaunchKernelGGL(galerkin_pack_fused_k<true>, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_slices, 256 * 12))), dim3(64), 0, st, Pc, slice_base, intra_off,
                           s_col, (const double *)X.s_val[x], (int *)nullptr, x_val[x], (int *)nullptr, mirror ? (const int64_t *)pk_ptr : (const int64_t *)nullptr, (int *)nullptr,
                           x_pk_val[x]);
        ORC_HIP(hipGetLastError());
    }
    lap("galerkin pack");
    // narrow column image for the levels the uniform kernels multiply (no packed mirror: the first coarse level): 2-byte columns in
    // their products' stream; all or nothing, decided on the host (the kernel variant is a template argument)
    const bool narrow_on = cfg().spmv_narrow_cols;
    if (narrow_on && !mirror && padded > 0) {
        unsigned short *c16;
        int *cbase, *wide;
        ORC_TRY(arena.alloc((size_t)padded, &c16));
        ORC_TRY(arena.alloc((size_t)(padded / 64) + 1, &cbase));
        ORC_TRY(tmp.alloc((size_t)1, &wide));
        ORC_HIP(hipMemsetAsync(wide, 0, sizeof(int), st));
        hipLaunchKernelGGL(narrow_build_k, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_slices, 256 * 32))), dim3(64), 0, st, Pc, c16, cbase, wide);
        ORC_HIP(hipGetLastError());
        int h_wide = 1;
        ORC_HIP(hipMemcpyAsync(&h_wide, wide, sizeof(int), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        if (!h_wide) { Pc.col16 = c16; Pc.colbase = cbase; }
    }
    L.P = Pc; L.val = val; L.n = nc; L.padded = padded;
    L.pk = PackedDev();
    L.xw = XWinDev();
    L.rows = RowsDev();
    for (int x = 0; x < n_sib; ++x) {
        CoarseLevel &Lx = *sib[x].L;
        Lx = CoarseLevel();
        Lx.P = Pc; Lx.val = x_val[x]; Lx.n = nc; Lx.padded = padded; Lx.rounds = L.rounds;
        Lx.choice = L.choice; Lx.chooser = L.chooser;
    }
    if (!scratch) { L.rows.slice_base = slice_base; L.rows.intra_off = intra_off; L.rows.col = s_col; L.rows.val = s_val; }
    if (scratch && packed_total > 0 && !last_level) {  // exact-size copy; the slices start where the packed mirror's do (pk_ptr); the last level is never aggregated
        // Only the NEXT level's aggregation and Galerkin product walk it, and the mirror of the level below (A.rows) is dead now
        // that this product's kernels are queued (same stream): both take turns in the scratch arena's companion, so a hierarchy
        // keeps no mirror once it is built (2 GB of 10.3 GB per hierarchy at 10.24 M rows).
        Arena &rows_arena = scratch->companion();
        rows_arena.release(Arena::Mark{0, 0});
        int *r_col, *r_intra;
        double *r_val;
        ORC_TRY(rows_arena.alloc((size_t)packed_total, &r_col));
        ORC_TRY(rows_arena.alloc((size_t)packed_total, &r_val));
        ORC_TRY(rows_arena.alloc(ncs, &r_intra));
        L.rows_transient = true;
        hipLaunchKernelGGL(rows_compact_k, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_slices, 256 * 16))), dim3(64), 0, st, (const int *)row_len, nc, n_slices,
                           (const long long *)slice_base, (const int *)intra_off, (const int *)s_col, (const double *)s_val, (const int64_t *)pk_ptr, r_intra, r_col, r_val);
        ORC_HIP(hipGetLastError());
        L.rows.slice_base = reinterpret_cast<const long long *>(pk_ptr); L.rows.intra_off = r_intra; L.rows.col = r_col; L.rows.val = r_val;
        for (int x = 0; x < n_sib; ++x) {  // a whole copy per sibling: the leader's is gone when ITS next level is built
            Arena &xa = sib[x].rows_arena ? *sib[x].rows_arena : *sib[x].arena;
            if (sib[x].rows_arena) xa.release(Arena::Mark{0, 0});
            int *xr_col, *xr_intra;
            double *xr_val;
            ORC_TRY(xa.alloc((size_t)packed_total, &xr_col));
            ORC_TRY(xa.alloc((size_t)packed_total, &xr_val));
            ORC_TRY(xa.alloc(ncs, &xr_intra));
            hipLaunchKernelGGL(rows_compact_k, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_slices, 256 * 16))), dim3(64), 0, st, (const int *)row_len, nc, n_slices,
                               (const long long *)slice_base, (const int *)intra_off, (const int *)s_col, (const double *)X.s_val[x], (const int64_t *)pk_ptr, xr_intra, xr_col, xr_val);
            ORC_HIP(hipGetLastError());
            CoarseLevel &Lx = *sib[x].L;
            Lx.rows.slice_base = reinterpret_cast<const long long *>(pk_ptr); Lx.rows.intra_off = xr_intra; Lx.rows.col = xr_col; Lx.rows.val = xr_val;
            Lx.rows_transient = sib[x].rows_arena != nullptr;
        }
    }
    if (mirror) {
        L.pk.ptr = pk_ptr; L.pk.col = pk_col; L.pk.val = pk_val; L.pk.total = packed_total;
        const int64_t n_blocks = ((int64_t)n_slices + 3) / 4;
        int *wcol, *wsize;
        unsigned short *lidx;
        ORC_TRY(arena.alloc((size_t)n_blocks * kXWinCap, &wcol));
        ORC_TRY(arena.alloc((size_t)n_blocks, &wsize));
        ORC_TRY(arena.alloc((size_t)packed_total, &lidx));
        const int win_cap = cfg().xwin_cap > 0 ? std::min(kXWinCap, cfg().xwin_cap) : kXWinCap;  // (test hooks: forced fallbacks)
        const int bit_words = cfg().xwin_bitwords > 0 ? std::min(kXBitWords, cfg().xwin_bitwords) : kXBitWords;
        const int small_words = cfg().xwin_small_bitwords > 0 ? std::min(kXBitWordsSmall, cfg().xwin_small_bitwords) : kXBitWordsSmall;
        int *pending;
        ORC_TRY(tmp.alloc((size_t)1, &pending));
        ORC_HIP(hipMemsetAsync(pending, 0, sizeof(int), st));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(xwin_build_k<kXBitWordsSmall, false>), dim3((unsigned)std::min<int64_t>(n_blocks, 2048)), dim3(kBlock), 0, st, Pc, L.pk, wcol, wsize, lidx,
                           n_blocks, win_cap, bit_words, small_words, pending);
        if (bit_words > small_words)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(xwin_build_k<kXBitWords, true>), dim3((unsigned)std::min<int64_t>(n_blocks, 2048)), dim3(kBlock), 0, st, Pc, L.pk, wcol, wsize, lidx,
                               n_blocks, win_cap, bit_words, kXBitWords, pending);
        if (trace_t) {
            int hp = 0;
            ORC_HIP(hipMemcpyAsync(&hp, pending, sizeof(int), hipMemcpyDeviceToHost, st));
            ORC_HIP(hipStreamSynchronize(st));
            fprintf(stderr, "[amg windows n=%lld] %lld blocks, %d left to the full bitmap\n", (long long)nc, (long long)n_blocks, hp);
        }
        ORC_HIP(hipGetLastError());
        L.xw.wcol = wcol; L.xw.wsize = wsize; L.xw.lidx = lidx;
        {   // [r05] this level's LDS share per workgroup: the smallest of a few sizes that leaves <= 1 % of the blocks without a window
            // (+ whatever had none to begin with); one small kernel and one host read per level with windows
            int *over;
            ORC_TRY(tmp.alloc((size_t)kXWinCapSizes, &over));
            ORC_HIP(hipMemsetAsync(over, 0, kXWinCapSizes * sizeof(int), st));
            hipLaunchKernelGGL(xwin_cap_k, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_blocks + kBlock - 1) / kBlock, 64))), dim3(kBlock), 0, st, (const int *)wsize, n_blocks, over);
            int h_over[kXWinCapSizes];
            ORC_HIP(hipMemcpyAsync(h_over, over, sizeof(h_over), hipMemcpyDeviceToHost, st));
            ORC_HIP(hipStreamSynchronize(st));
            int cap = win_cap;
            for (int q = 0; q < kXWinCapSizes && cfg().xwin_level_cap; ++q)
                if (kXWinCapSize[q] <= win_cap && (int64_t)h_over[q] * 100 <= n_blocks) { cap = kXWinCapSize[q]; break; }
            L.xw.cap = cap;
            if (trace_t) fprintf(stderr, "[amg windows n=%lld] LDS share %d entries per workgroup (%d of %lld blocks have larger windows)\n", (long long)nc, cap,
                                 cap < win_cap ? h_over[std::find(kXWinCapSize, kXWinCapSize + kXWinCapSizes, cap) - kXWinCapSize] : 0, (long long)n_blocks);
        }
        {   // [r05] scratch of the in-launch fold of this level's products (spmv_xwin_k): two sums per workgroup of a one-workgroup-per-block launch
            const size_t n_wg = (size_t)((n_blocks + 7) / 8 * 8);
            double *fs;
            unsigned *fc;
            ORC_TRY(arena.alloc(2 * n_wg, &fs));
            ORC_TRY(arena.alloc((size_t)4, &fc));
            ORC_HIP(hipMemsetAsync(fc, 0, 4 * sizeof(unsigned), st));
            L.xw.fold_scratch = fs; L.xw.fold_counter = fc;
        }
        for (int x = 0; x < n_sib; ++x) {  // same structure, own values (and scratch of their own: the systems' products run side by side)
            CoarseLevel &Lx = *sib[x].L;
            Lx.pk = L.pk;
            Lx.pk.val = x_pk_val[x];
            Lx.xw = L.xw;
            const size_t n_wg = (size_t)((n_blocks + 7) / 8 * 8);
            double *fs;
            unsigned *fc;
            ORC_TRY(sib[x].arena->alloc(2 * n_wg, &fs));
            ORC_TRY(sib[x].arena->alloc((size_t)4, &fc));
            ORC_HIP(hipMemsetAsync(fc, 0, 4 * sizeof(unsigned), st));
            Lx.xw.fold_scratch = fs; Lx.xw.fold_counter = fc;
        }
        const bool xwin_stats = cfg().debug_xwin;
        if (xwin_stats) {
            unsigned long long *d_st, h_st[5];
            ORC_TRY(tmp.alloc((size_t)5, &d_st));
            ORC_HIP(hipMemsetAsync(d_st, 0, sizeof(h_st), st));
            hipLaunchKernelGGL(xwin_stats_k, dim3((unsigned)std::min<int64_t>(n_blocks, 4096)), dim3(kBlock), 0, st, (const int *)wcol, (const int *)wsize, n_blocks, d_st);
            ORC_HIP(hipMemcpyAsync(h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, st));
            ORC_HIP(hipStreamSynchronize(st));
            fprintf(stderr, "[orc xwin] rows %lld nnz %lld blocks %lld: window entries %llu (%.1f per row), runs %llu (%.1f entries per run), in runs >= 8: %.1f %%, span < 65536: %llu blocks, no window: %llu\n",
                    (long long)nc, (long long)packed_total, (long long)n_blocks, h_st[0], (double)h_st[0] / (double)nc, h_st[1], (double)h_st[0] / (double)std::max<unsigned long long>(h_st[1], 1),
                    100. * (double)h_st[2] / (double)std::max<unsigned long long>(h_st[0], 1), h_st[3], h_st[4]);
        }
    }
    lap("galerkin mirrors");
    if (!scratch) tmp_scope.mark = tmp.mark();  // the scratch rows ARE the mirror: everything stays
    if (n_sib > 0) g_shared_galerkin.fetch_add(n_sib, std::memory_order_relaxed);
    return ORC_OK;
}

struct MgParams {
    uint64_t max_levels, iters;
    int smoother, preconditioner;
    double relaxation, threshold;
};

// Runs the enclosed calls on the solve-side stream (see SolveSide) by making it the context's current stream.
struct StreamSwitch {
    hipStream_t saved;
    bool on;
    StreamSwitch(SolveSide *side) : saved(ctx().stream), on(side != nullptr) { if (on) ctx().stream = side->stream; }
    ~StreamSwitch() { if (on) ctx().stream = saved; }
};
// B waits for what A has queued so far / A waits for what B has queued so far
static int side_wait_setup(SolveSide *side, hipStream_t setup_stream) {
    if (!side) return ORC_OK;
    ORC_HIP(hipEventRecord(side->ev_setup, setup_stream));
    ORC_HIP(hipStreamWaitEvent(side->stream, side->ev_setup, 0));
    return ORC_OK;
}
static int setup_wait_side(SolveSide *side, hipStream_t setup_stream) {
    if (!side) return ORC_OK;
    ORC_HIP(hipEventRecord(side->ev_solve, side->stream));
    ORC_HIP(hipStreamWaitEvent(setup_stream, side->ev_solve, 0));
    return ORC_OK;
}

// Leaves a level of the Multigrid arm on every exit path: what the side stream still reads (pairing, coarse matrix,
// vectors) must outlive it, so the set-up stream first waits for the side stream, then both arenas unwind.
struct SideScope {
    SolveSide *side;
    hipStream_t setup_stream;
    Arena &arena, &varena;
    Arena::Mark mk, vmk;
    bool release_varena;
    SideScope(SolveSide *sd, hipStream_t st, Arena &a, Arena &va, bool rel_v)
        : side(sd), setup_stream(st), arena(a), varena(va), mk(a.mark()), vmk(va.mark()), release_varena(rel_v) {}
    ~SideScope() {
        if (side) (void)setup_wait_side(side, setup_stream);
        if (release_varena) varena.release(vmk);
        arena.release(mk);
    }
    SideScope(const SideScope &) = delete;
    SideScope &operator=(const SideScope &) = delete;
};

// linear_algebra.rs:66-141.  `add_to`: the fine vector the prolonged correction is added to.
// With a SolveSide the vector work of a level (restriction, smoothing solves, residual check, prolongation) is queued
// on the side stream in exactly the order below, and the recursion's set-up overlaps this level's smoothing.
static int multigrid_solve_dev(const MatView &A, const double *r, uint64_t level, const MgParams &mp, double threshold, Arena &arena,
                               SolveStats *stats, int *dev_status, double *out, double *add_to, SolveSide *side) {
    const int64_t n = A.P.n;
    hipStream_t st = ctx().stream;
    Arena &varena = side ? *side->arena : arena;  // vectors and solver work space
    SideScope scope(side, st, arena, varena, side != nullptr);  // runs on every return below
    auto leave = [](int code) { return code; };
    int *choice, *chooser;
    CoarseLevel L;
    const AmgHierarchy *hier = stats ? stats->hierarchy : nullptr;
    if (hier && (int)level <= hier->n_levels && (level == 1 ? hier->n_fine : hier->level[level - 2].n) == n) {
        // :80, :84 were done ahead of time (multigrid_prepare_dev) for exactly this matrix
        const AmgHierarchy::Level &h = hier->level[level - 1];
        choice = h.choice; chooser = h.chooser;
        L.P = h.P; L.val = h.val; L.pk = h.pk; L.xw = h.xw; L.rows = h.rows; L.n = h.n; L.padded = h.padded; L.rounds = h.rounds;
    } else {
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &choice));
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &chooser));
        const int *warm = nullptr;
        SiblingPairing *sib = stats ? stats->sibling : nullptr;
        if (sib && stats->sibling_role == 2 && level == 1) warm = sib->wait((int)level, n, st);  // the fine level only: there the systems share their pattern
        const int agg_st = aggregate(A, arena, choice, chooser, &L.rounds, warm);  // :80 (scratch is released with the level)
        if (sib && stats->sibling_role == 1 && level == 1) {
            if (agg_st == ORC_OK) ORC_TRY(sib->publish((int)level, choice, n, st));
            sib->finish();  // nothing more will be published
        }
        ORC_TRY(agg_st);
        ORC_TRY(galerkin(A, choice, chooser, arena, L));  // :84
    }
    const int64_t nc = L.n;
    if (stats && level < 8) {
        stats->amg_levels = std::max(stats->amg_levels, (int)level);
        stats->amg_rows[level] = nc;
        stats->amg_nnz[level] = L.padded;
        stats->amg_rounds[level] = L.rounds;
    }
    ORC_TRY(side_wait_setup(side, st));  // the coarse matrix and the pairing are complete
    MatView Ac;
    Ac.P = L.P;
    Ac.val = L.val;
    Ac.pk = L.pk;
    Ac.xw = L.xw;
    Ac.rows = L.rows;
    Ac.symmetric = A.symmetric;  // halo stays null: coarse levels are solved per rank
    double *r_prime, *e_prime, *partials, *scal;
    const bool shared_scaling = cfg().amg_shared_scaling && mp.smoother == ORC_SOLVER_BICGSTAB && mp.preconditioner == ORC_PRECOND_JACOBI;
    ScaledOperator scaled;
    {
        StreamSwitch sw(side);
        hipStream_t vs = ctx().stream;
        ORC_TRY(varena.alloc((size_t)std::max<int64_t>(nc, 1), &r_prime));
        ORC_TRY(varena.alloc((size_t)std::max<int64_t>(nc, 1), &e_prime));
        ORC_TRY(varena.alloc((size_t)kMaxPartials, &partials));
        ORC_TRY(varena.alloc((size_t)4, &scal));
        double *r_check = nullptr;  // reference-order norm (verification mode): the residual is materialised
        if (ctx().reduction_order == ORC_REDUCTION_REFERENCE) ORC_TRY(varena.alloc((size_t)std::max<int64_t>(nc, 1), &r_check));
        hipLaunchKernelGGL(restrict_k, dim3(grid_for(nc)), dim3(kBlock), 0, vs, choice, n, nc, r, r_prime);  // :82
        ORC_HIP(hipGetLastError());
        ORC_TRY(vec_fill(e_prime, 0., nc));  // :86
        // both smoothing solves of this level scale Ac the same way (:159-166): prepared once, held in varena until the level unwinds
        if (shared_scaling) ORC_TRY(jacobi_scaling_prepare_dev(Ac, mp.iters, varena, scaled));
        int stt = shared_scaling ? bicgstab_scaled_dev(scaled, r_prime, e_prime, varena)
                                 : iterative_solve_dev(Ac, r_prime, e_prime, mp.iters, mp.smoother, mp.relaxation, threshold, mp.preconditioner, varena, stats);  // :87-96
        if (stt != ORC_OK) return leave(stt);
        // :97-105  |r' - a' e'| is NaN -> "Multigrid diverged"
        ORC_TRY(residual_norm2_dev(Ac, r_prime, e_prime, partials, scal, r_check));
        hipLaunchKernelGGL(nan_to_status_k, dim3(1), dim3(64), 0, vs, scal, dev_status, (int)ORC_ERR_MULTIGRID_DIVERGED);
    }
    if (level < mp.max_levels && nc > 16) {  // :109
        // :110-121 — the recursion receives r', not the residual; its set-up runs beside the smoothing above
        int stt = multigrid_solve_dev(Ac, r_prime, level + 1, mp, threshold, arena, stats, dev_status, nullptr, e_prime, side);
        if (stt != ORC_OK) return leave(stt);
        StreamSwitch sw(side);
        stt = shared_scaling ? bicgstab_scaled_dev(scaled, r_prime, e_prime, varena)
                             : iterative_solve_dev(Ac, r_prime, e_prime, mp.iters, mp.smoother, mp.relaxation, threshold / 10., mp.preconditioner, varena, stats);  // :123-132
        if (stt != ORC_OK) return leave(stt);
    }
    {
        StreamSwitch sw(side);
        hipLaunchKernelGGL(prolong_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, choice, chooser, n, e_prime, out, add_to);  // :140
        ORC_HIP(hipGetLastError());
    }
    return leave(ORC_OK);
}

// The set-up half of the Multigrid arm on its own: levels 1..3 of the hierarchy for `A_in` seen through the arm's
// preconditioner (linear_algebra.rs:159-166 then :80, :84 per level, recursion rule of :109).
int multigrid_prepare_dev(const MatView &A_in, int preconditioner, Arena &arena, AmgHierarchy &H, SiblingPairing *sibling, int sibling_role, Arena *scratch) {
    H = AmgHierarchy();
    const int64_t n = A_in.P.n;
    H.n_fine = n;
    // a follower (roles 2, 3) SPEAKS on every path — an offer or a withdrawal — because the leader waits for every follower it expects
    struct SpeakGuard {
        SiblingPairing *s;
        int slot;
        bool spoken = false;
        ~SpeakGuard() { if (s && !spoken) s->withdraw(slot); }
    } speak{(sibling && sibling_role >= 2) ? sibling : nullptr, sibling_role - 2};
    if (n == 0) return ORC_OK;
    if (scratch) {  // nothing of an earlier set-up's mirrors is alive
        scratch->companion().release(Arena::Mark{0, 0});
        ORC_TRY(scratch->companion().reset());
    }
    MatView views[4];
    views[0] = A_in;
    // [r04] level 0 gets a row-contiguous mirror too (its pattern half is the mesh pattern's CSR form, built at mesh creation unless
    // ORC_AMG_L0_MIRROR=0; the values are exported here, one coalesced-read pass): the aggregation and the first Galerkin product walk single
    // rows, and in SELL every entry of a row is a cache line of its own.  No gain on one stream, -10 ... -14 ms in the concurrent iteration
    // (sell_from_csr_host).
    if (!views[0].rows.col && A_in.P.rows_col && A_in.P.rows_base && A_in.val && A_in.P.csr_row_ptr && A_in.P.nnz > 0) {
        if (cfg().amg_l0_mirror) {
            double *rv;
            ORC_TRY(arena.alloc((size_t)A_in.P.nnz, &rv));
            ORC_TRY(sell_rows_values_dev(A_in.P, A_in.val, rv));
            views[0].rows.slice_base = A_in.P.rows_base; views[0].rows.intra_off = A_in.P.rows_intra; views[0].rows.col = A_in.P.rows_col; views[0].rows.val = rv;
        }
    }
    if (preconditioner == ORC_PRECOND_JACOBI) {
        double *dinv;
        ORC_TRY(arena.alloc((size_t)n, &dinv));
        ORC_TRY(diag_inverse_dev(A_in, dinv));
        if (!views[0].s1) views[0].s1 = dinv;
        else if (!views[0].s2) views[0].s2 = dinv;
        else return set_error(ORC_ERR_BAD_ARGUMENT, "more than two nested Jacobi scalings");
    }
    const uint64_t max_levels = 3;  // MULTIGRID_COARSENING_LEVELS, :10
    // [r04] one Galerkin pass for the momentum systems that share the fine pairing (SiblingPairing::make_offer ...; ORC_AMG_SHARED_GALERKIN=0:
    // every system multiplies for itself, r03).  Read per call: the tests compare the two forms.
    const bool share_on = cfg().amg_shared_galerkin && scratch != nullptr;
    for (uint64_t level = 1; level <= max_levels; ++level) {
        const MatView &A = views[level - 1];
        const int64_t nf = A.P.n;
        AmgHierarchy::Level &h = H.level[level - 1];
        CoarseLevel L;
        bool adopted = false;
        if (sibling && sibling_role >= 2 && level == 1) {  // a follower always speaks: the leader waits for every expected follower
            const int slot = sibling_role - 2;
            speak.spoken = true;
            if (share_on && sibling->make_offer(slot, &views[0], &arena, scratch ? &scratch->companion() : nullptr, &L, ctx().stream) == ORC_OK) {
                adopted = sibling->wait_answer(slot, ctx().stream);
            } else {
                sibling->withdraw(slot);
            }
        }
        if (adopted) {  // the leader has built this level on ITS pattern with this system's values (L) and vouches for the pairing
            h.choice = const_cast<int *>(sibling->lead_choice);
            h.chooser = const_cast<int *>(sibling->lead_chooser);
            L.rounds = 1;
        } else {
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nf, 1), &h.choice));
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nf, 1), &h.chooser));
        const int *warm = nullptr;
        if (sibling && sibling_role >= 2 && level == 1) warm = sibling->wait((int)level, nf, ctx().stream);
        // the aggregation's work lists (48 bytes per row) are dead when it returns: they live in `scratch` when there is one
        Arena &agg_arena = scratch ? *scratch : arena;
        const Arena::Mark agg_mark = agg_arena.mark();
        const int agg_st = aggregate(A, agg_arena, h.choice, h.chooser, &L.rounds, warm);
        if (scratch) scratch->release(agg_mark);
        GalerkinSibling gs[2];
        int n_sib = 0;
        bool took[2] = {false, false};
        const bool leads = sibling && sibling_role == 1 && level == 1;
        if (leads) {
            if (agg_st == ORC_OK) ORC_TRY(sibling->publish((int)level, h.choice, nf, ctx().stream));
            if (agg_st == ORC_OK && share_on) {
                SiblingPairing::Offer *offers[2] = {nullptr, nullptr};
                const int n_off = sibling->collect_offers(offers);
                if (n_off > 0 && nf > 0) {  // is this pairing the fixed point of the offered matrices too?  (agg_verify_k: one pass each)
                    hipStream_t st = ctx().stream;
                    ArenaScope vscope(*scratch);
                    int *taken_by;
                    AggCounters *C;
                    ORC_TRY(scratch->alloc((size_t)nf, &taken_by));
                    ORC_TRY(scratch->alloc((size_t)2, &C));
                    ORC_HIP(hipMemsetAsync(C, 0, 2 * sizeof(AggCounters), st));
                    const int g = grid_for(nf);
                    hipLaunchKernelGGL(agg_reset_k, dim3(g), dim3(kBlock), 0, st, taken_by, nf);
                    hipLaunchKernelGGL(agg_scatter_k, dim3(g), dim3(kBlock), 0, st, (const int *)h.choice, taken_by, nf);
                    for (int q = 0; q < n_off; ++q) {
                        const MatView &B = *offers[q]->view;
                        if (B.P.n != nf || B.P.col != A.P.col) continue;
                        ORC_HIP(hipStreamWaitEvent(st, offers[q]->view_ready, 0));
                        hipLaunchKernelGGL(agg_verify_k, dim3(g), dim3(kBlock), 0, st, B, (const int *)h.choice, (const int *)taken_by, C + q);
                    }
                    ORC_HIP(hipGetLastError());
                    AggCounters hc[2];
                    ORC_HIP(hipMemcpyAsync(hc, C, sizeof(hc), hipMemcpyDeviceToHost, st));
                    ORC_HIP(hipStreamSynchronize(st));
                    const bool trace_v = cfg().amg_trace;
                    for (int q = 0; q < n_off; ++q) {
                        const MatView &B = *offers[q]->view;
                        const bool same = B.P.n == nf && B.P.col == A.P.col && hc[q].changed == 0;
                        if (trace_v) fprintf(stderr, "[amg sibling n=%lld] offered system %d: rows that would change: %d\n", (long long)nf, (int)(offers[q] - sibling->offer), hc[q].changed);
                        if (!same) continue;
                        gs[n_sib].A = offers[q]->view;
                        gs[n_sib].arena = offers[q]->arena;
                        gs[n_sib].rows_arena = offers[q]->rows_arena;
                        gs[n_sib].L = static_cast<CoarseLevel *>(offers[q]->level);
                        took[offers[q] - sibling->offer] = true;
                        ++n_sib;
                    }
                }
            }
        }
        int gal_st = agg_st;
        if (gal_st == ORC_OK) gal_st = galerkin(A, h.choice, h.chooser, arena, L, scratch, level == max_levels, gs, n_sib);
        if (leads) {
            const bool none[2] = {false, false};
            const int ans_st = sibling->answer(gal_st == ORC_OK ? took : none, h.choice, h.chooser, ctx().stream);
            sibling->finish();
            if (gal_st == ORC_OK) gal_st = ans_st;
        }
        ORC_TRY(gal_st);
        }
        // a mirror in the companion arena lives until the next level is built: the hierarchy does not carry it
        h.P = L.P; h.val = L.val; h.pk = L.pk; h.xw = L.xw; h.rows = L.rows_transient ? RowsDev() : L.rows; h.n = L.n; h.padded = L.padded; h.rounds = L.rounds;
        H.n_levels = (int)level;
        if (!(level < max_levels && L.n > 16)) break;  // :109
        MatView Ac;
        Ac.P = L.P;
        Ac.val = L.val;
        Ac.pk = L.pk;
        Ac.xw = L.xw;
            Ac.rows = L.rows;
        Ac.symmetric = A.symmetric;
        views[level] = Ac;
    }
    return ORC_OK;
}

int multigrid_coarse_part_dev(const MatView &A, const double *r, double *x, uint64_t iteration_count, double relaxation_factor,
                              double convergence_threshold, int preconditioner, Arena &arena, SolveStats *stats, int *dev_status) {
    if (A.P.n == 0) return ORC_OK;
    if (!stats || !stats->hierarchy || stats->hierarchy->n_levels < 1 || stats->hierarchy->n_fine != A.P.n)
        return set_error(ORC_ERR_BAD_ARGUMENT, "the coarse part of the Multigrid arm needs a hierarchy prepared for this matrix");
    MgParams mp{3 /* MULTIGRID_COARSENING_LEVELS, :10 */, iteration_count, ORC_SOLVER_BICGSTAB, preconditioner, relaxation_factor, convergence_threshold};
    return multigrid_solve_dev(A, r, 1, mp, convergence_threshold, arena, stats, dev_status, nullptr, x, nullptr);
}

// Multigrid arm of iterative_solve (:270-296); A and b are already the preconditioned system.
int multigrid_arm_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor,
                      double convergence_threshold, int preconditioner, Arena &arena, SolveStats *stats, int smoother) {
    const int64_t n = A.P.n;
    if (n == 0) return ORC_OK;
    hipStream_t st = ctx().stream;
    // two streams only where no host-synchronised smoother (colouring) is involved; on a partitioned operator the
    // level-0 work (halo exchanges, all-reduces) stays on the library stream — every RCCL call keeps its stream — and
    // only the rank-local coarse levels use the side stream
    SolveSide *side = (stats && stats->side && stats->side->stream && smoother == ORC_SOLVER_BICGSTAB && !stats->hierarchy) ? stats->side : nullptr;
    SolveSide *side0 = A.halo ? nullptr : side;
    Arena &varena = side0 ? *side0->arena : arena;
    SideScope scope(side, st, arena, varena, side0 != nullptr);  // runs on every return below
    auto leave = [](int code) { return code; };
    ORC_TRY(side_wait_setup(side0, st));  // the preconditioned system (scaling vectors, b) was prepared on the set-up stream
    double *r;
    int *dev_status;
    {
        StreamSwitch sw(side0);
        // :273-282 — the smoother is called with the same preconditioner: the scaled system is scaled again (Q4)
        int stt = iterative_solve_dev(A, b, x, iteration_count, smoother, relaxation_factor, convergence_threshold, preconditioner, varena, stats);
        if (stt != ORC_OK) return leave(stt);
        ORC_TRY(varena.alloc((size_t)n, &r));
        ORC_TRY(varena.alloc((size_t)1, &dev_status));
        ORC_HIP(hipMemsetAsync(dev_status, 0, sizeof(int), ctx().stream));
        ORC_TRY(residual_dev(A, b, x, r));  // :283
    }
    MgParams mp{3 /* MULTIGRID_COARSENING_LEVELS, :10 */, iteration_count, smoother, preconditioner, relaxation_factor, convergence_threshold};
    int stt = multigrid_solve_dev(A, r, 1, mp, convergence_threshold, arena, stats, dev_status, nullptr, x, side);  // :284-295
    if (stt == ORC_OK) {
        ORC_TRY(setup_wait_side(side, st));
        int h = 0;
        ORC_HIP(hipMemcpyAsync(&h, dev_status, sizeof(int), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        stt = h;
    }
    return leave(stt);
}


// ------------------------------------------------------------------ the Multigrid arm for three systems on one pattern
// linear_algebra.rs:270-296 for the u, v and w momentum systems of one SIMPLE iteration at once (MatView3, linalg.hpp).
// Per system the operations and their order are those of multigrid_arm_dev / multigrid_solve_dev, so each system's result
// is bit-identical to its own solve; what changes is who shares a kernel:
//   * level 0 (the mesh pattern) — smoothing solve and residual for the three systems in lock-step (bicgstab3_dev);
//   * level 1 — whenever v's and w's fine-level pairings equal u's (the normal case: SiblingPairing), the three Galerkin
//     operators share their pattern as well and level 1 is solved in lock-step too;
//   * levels 2 and 3 — the level-1 pairings differ in a few rows, so these stay per system, queued on one stream each;
//   * the three hierarchies are built by one host thread each (their rounds synchronise their stream) beside the level-0 solve.
// Anything that does not fit (pairings differ, a level too small) falls back to the per-system coarse part.
int multigrid_arm3_dev(const MatView3 &A3, const double *const b[3], double *const x[3], uint64_t iteration_count, double relaxation_factor,
                       double convergence_threshold, int preconditioner, Arena &arena, TripleLane lanes[3], SiblingPairing *sibling, int status_out[3],
                       const std::function<void()> &on_hierarchies_built) {
    const int64_t n = A3.P.n;
    for (int k = 0; k < 3; ++k) status_out[k] = ORC_OK;
    if (n == 0) return ORC_OK;
    if (!triple_supported()) return set_error(ORC_ERR_BAD_ARGUMENT, "three-system solve: tree reductions only");
    Ctx &g = ctx();
    hipStream_t st = g.stream;
    ArenaScope scope(arena);
    const size_t n3 = (size_t)3 * (size_t)n;
    // [r04] partitioned mesh (A3.halo): level 0 exchanges the interleaved iterate's ghost entries and all-reduces its sums (every RCCL
    // call on the library stream, issued by this thread); the hierarchies and every coarse level are rank-local as in the
    // one-system path (ghost columns are never partners and are dropped from the Galerkin products).  x[k] hold ncols entries.
    const size_t ncols3 = (size_t)3 * (size_t)std::max<int64_t>(A3.P.ncols, n);
    const MgParams mp{3 /* MULTIGRID_COARSENING_LEVELS, :10 */, iteration_count, ORC_SOLVER_BICGSTAB, preconditioner, relaxation_factor, convergence_threshold};

    // ---- hierarchies: one thread per system, from now on (they need the matrices only)
    MatView plain[3];
    for (int k = 0; k < 3; ++k) {
        plain[k].P = A3.P;
        plain[k].val = A3.val[k];
        plain[k].symmetric = lanes[k].symmetric;
        plain[k].persistent_pattern = true;
        plain[k].halo = A3.halo;  // (nothing below exchanges through it: the set-up and the coarse parts are rank-local)
    }
    ORC_HIP(hipStreamSynchronize(st));  // the assembled matrices are complete before other streams read them
    Ctx local[3];
    std::thread th[3];
    int st_prep[3] = {ORC_OK, ORC_OK, ORC_OK};
    bool prepared[3] = {false, false, false};
    if (sibling) sibling->begin(true);
    auto prepare = [&](int k) {
        CtxScope cs(&local[k]);
        if (hipSetDevice(local[k].device) != hipSuccess) { st_prep[k] = set_error(ORC_ERR_HIP, "hipSetDevice failed in a set-up thread"); return; }
        lanes[k].hier_arena->release(Arena::Mark{0, 0});
        int stp = lanes[k].hier_arena->empty() ? lanes[k].hier_arena->reset() : ORC_OK;
        if (stp == ORC_OK && lanes[k].scratch_arena) {
            lanes[k].scratch_arena->release(Arena::Mark{0, 0});
            stp = lanes[k].scratch_arena->reset();  // nothing of the previous set-up is alive: a fragmented reservation becomes one chunk
        }
        // roles: 1 = leader (u), 2 / 3 = followers (slots 0 / 1 of the shared Galerkin pass)
        if (stp == ORC_OK) stp = multigrid_prepare_dev(plain[k], preconditioner, *lanes[k].hier_arena, lanes[k].hierarchy, sibling, k == 0 ? 1 : k + 1, lanes[k].scratch_arena);
        else if (k > 0 && sibling) sibling->withdraw(k - 1);  // the leader waits for every follower it was told to expect
        if (k == 0 && sibling) sibling->finish();  // whatever happened to u: v and w must not wait for a level that will not come
        if (hipStreamSynchronize(local[k].stream) != hipSuccess && stp == ORC_OK) stp = set_error(ORC_ERR_HIP, "stream synchronisation failed in a set-up thread");
        // test hook (tests/mp_worker.py, mode gpu_lane_error): ORC_DEBUG_INJECT_LANE_ERROR="rank:lane" fails that rank's set-up thread
        // locally — the level-0 collectives of every rank still complete and the caller's status agreement tells all of them
        if (!cfg().inject_lane_error.empty()) {
            int r_ = -1, k_ = -1;
            if (sscanf(cfg().inject_lane_error.c_str(), "%d:%d", &r_, &k_) == 2 && r_ == local[k].rank && k_ == k && stp == ORC_OK)
                stp = set_error(ORC_ERR_HIP, "injected lane error (rank %d, lane %d)", r_, k_);
        }
        st_prep[k] = stp;
        prepared[k] = true;
    };
    struct Joiner {  // no exit path may leave a thread running
        std::thread *t;
        ~Joiner() { for (int k = 0; k < 3; ++k) if (t[k].joinable()) t[k].join(); }
    } joiner{th};
    // followers first: the leader is told how many of them will speak (a follower without a thread runs after the join, when the leader
    // is through, and multiplies for itself)
    int n_follower_threads = 0;
    for (int k = 2; k >= 0; --k) {
        local[k] = g;
        local[k].stream = lanes[k].setup_stream;
        local[k].last_error.clear();
        if (k == 0 && sibling) sibling->set_expected(n_follower_threads);
        try { th[k] = std::thread(prepare, k); if (k > 0) ++n_follower_threads; } catch (...) { /* no thread to be had: prepared below, before the join */ }
    }

    auto join_hierarchies = [&] {
        for (int k = 0; k < 3; ++k) {
            if (th[k].joinable()) th[k].join();
            if (!prepared[k]) prepare(k);
        }
        if (on_hierarchies_built) on_hierarchies_built();
    };

    // ---- level 0 in lock-step
    double *b3, *x3, *r3;
    int *dev_status;
    ORC_TRY(arena.alloc(n3, &b3));
    ORC_TRY(arena.alloc(ncols3, &x3));
    ORC_TRY(arena.alloc(n3, &r3));
    ORC_TRY(arena.alloc((size_t)4, &dev_status));
    ORC_HIP(hipMemsetAsync(dev_status, 0, 4 * sizeof(int), st));
    ORC_TRY(interleave3_dev(b[0], b[1], b[2], b3, n));
    ORC_TRY(interleave3_dev(x[0], x[1], x[2], x3, n));
    MatView3 V = A3;
    const double *bp3 = b3;
    if (preconditioner == ORC_PRECOND_JACOBI) {  // iterative_solve's own scaling of the system the arm sees (:159-166)
        double *dinv3, *bt3;
        ORC_TRY(arena.alloc(n3, &dinv3));
        ORC_TRY(arena.alloc(n3, &bt3));
        ORC_TRY(diag_inverse3_dev(A3, dinv3));
        ORC_TRY(scale_vec_dev(dinv3, b3, bt3, (int64_t)n3));
        V.s1 = dinv3;
        bp3 = bt3;
    }
    ORC_TRACE("arm3: level-0 solve");
    ORC_TRY(bicgstab3_dev(V, bp3, x3, iteration_count, preconditioner, arena));  // :273-282 (scaled again inside: Q4)
    ORC_TRY(residual3_dev(V, bp3, x3, r3));                                       // :283
    ORC_TRACE("arm3: level-0 solve queued; joining the hierarchies");

    // ---- the hierarchies
    join_hierarchies();
    for (int k = 0; k < 3; ++k)
        if (st_prep[k] != ORC_OK) { g.last_error = local[k].last_error; return st_prep[k]; }
    const AmgHierarchy *H[3] = {&lanes[0].hierarchy, &lanes[1].hierarchy, &lanes[2].hierarchy};
    bool shared = H[0]->n_levels >= 1 && H[1]->n_levels == H[0]->n_levels && H[2]->n_levels == H[0]->n_levels;
    const int64_t nc = shared ? H[0]->level[0].n : 0;
    if (shared) {
        for (int k = 1; k < 3; ++k) shared = shared && H[k]->level[0].n == nc && H[k]->level[0].padded == H[0]->level[0].padded;
    }
    // [r05] A first coarse level with a packed mirror (ragged rows: config 5) is multiplied by spmv_xwin_k when a system is solved alone — one workgroup per
    // 256-row block, i.e. another thread -> row map and other partial sums than the SELL walk the lock-step kernels share with spmv_uniform_k: in
    // lock-step its dot products would round differently from the one-system solve's.  Such a level is solved per system (the lanes below).
    if (shared && H[0]->level[0].pk.ptr) shared = false;
    if (shared) {  // same pairing and same coarse row lengths => same coarse pattern (the symbolic part of the product depends on nothing else)
        int *diff;
        ORC_TRY(arena.alloc((size_t)1, &diff));
        ORC_HIP(hipMemsetAsync(diff, 0, sizeof(int), st));
        for (int k = 1; k < 3; ++k) {
            hipLaunchKernelGGL(count_diff_k, dim3(grid_for(n)), dim3(kBlock), 0, st, (const int *)H[0]->level[0].choice, (const int *)H[k]->level[0].choice, n, diff);
            hipLaunchKernelGGL(count_diff_k, dim3(grid_for(nc)), dim3(kBlock), 0, st, H[0]->level[0].P.row_len, H[k]->level[0].P.row_len, nc, diff);
        }
        int hd = 0;
        ORC_HIP(hipMemcpyAsync(&hd, diff, sizeof(int), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        shared = hd == 0;
    }
    const bool trace = cfg().amg_trace;
    if (trace) fprintf(stderr, "[amg triple n=%lld] level 1 %s\n", (long long)n, shared ? "in lock-step" : "per system");
    ORC_TRACE("arm3: hierarchies joined, level 1 %s", shared ? "in lock-step" : "per system");

    hipEvent_t ev_main = nullptr, ev_lane[3] = {nullptr, nullptr, nullptr};
    struct Events {
        hipEvent_t *m, *l;
        ~Events() { if (*m) (void)hipEventDestroy(*m); for (int k = 0; k < 3; ++k) if (l[k]) (void)hipEventDestroy(l[k]); }
    } events{&ev_main, ev_lane};
    ORC_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
    for (int k = 0; k < 3; ++k) ORC_HIP(hipEventCreateWithFlags(&ev_lane[k], hipEventDisableTiming));
    // every lane stream is drained before its arena is unwound, whatever happens below
    struct Drain {
        TripleLane *l;
        ~Drain() { for (int k = 0; k < 3; ++k) { (void)hipStreamSynchronize(l[k].solve_stream); l[k].vec_arena->release(Arena::Mark{0, 0}); } }
    } drain{lanes};
    for (int k = 0; k < 3; ++k) {
        lanes[k].vec_arena->release(Arena::Mark{0, 0});
        if (lanes[k].vec_arena->empty()) ORC_TRY(lanes[k].vec_arena->reset());
        lanes[k].stats = SolveStats();
        lanes[k].stats.hierarchy = &lanes[k].hierarchy;
    }
    // queues `fn` on lane k's solve stream (the calling thread keeps issuing; nothing below synchronises with the host)
    auto on_lane = [&](int k, auto &&fn) {
        hipStream_t saved = g.stream;
        g.stream = lanes[k].solve_stream;
        const int r = fn();
        g.stream = saved;
        return r;
    };

    if (!shared) {
        // per-system coarse parts (multigrid_coarse_part_dev) side by side: r and x per system, contiguous
        double *rk[3];
        for (int k = 0; k < 3; ++k) ORC_TRY(arena.alloc((size_t)n, &rk[k]));
        ORC_TRY(deinterleave3_dev(r3, rk[0], rk[1], rk[2], n));
        ORC_TRY(deinterleave3_dev(x3, x[0], x[1], x[2], n));
        ORC_HIP(hipEventRecord(ev_main, st));
        for (int k = 0; k < 3; ++k) {
            ORC_HIP(hipStreamWaitEvent(lanes[k].solve_stream, ev_main, 0));
            ORC_TRY(on_lane(k, [&] {
                return multigrid_coarse_part_dev(plain[k], rk[k], x[k], iteration_count, relaxation_factor, convergence_threshold, preconditioner,
                                                 *lanes[k].vec_arena, &lanes[k].stats, dev_status + k);
            }));
            ORC_HIP(hipEventRecord(ev_lane[k], lanes[k].solve_stream));
            ORC_HIP(hipStreamWaitEvent(st, ev_lane[k], 0));
        }
    } else {
        // ---- level 1 in lock-step (multigrid_solve_dev, level 1)
        const AmgHierarchy::Level &L0 = H[0]->level[0];
        MatView3 Ac3;
        Ac3.P = L0.P;
        for (int k = 0; k < 3; ++k) Ac3.val[k] = H[k]->level[0].val;
        const size_t nc3 = (size_t)3 * (size_t)nc;
        double *r1, *e1, *partials, *norm3;
        ORC_TRY(arena.alloc(nc3, &r1));
        ORC_TRY(arena.alloc(nc3, &e1));
        ORC_TRY(arena.alloc((size_t)3 * kMaxPartials, &partials));
        ORC_TRY(arena.alloc((size_t)4, &norm3));
        const int dbg_mask = cfg().debug_sync;  // debugging aid: drain the stream after chosen steps
        int step_no = 0;
        auto step = [&](const char *what) { if (dbg_mask & (1 << step_no)) { (void)hipStreamSynchronize(st); ORC_TRACE("arm3 level 1: %s done", what); } ++step_no; };
        const bool dbg_sync = (dbg_mask & 64) != 0;
        step("level 0");
        hipLaunchKernelGGL(restrict3_k, dim3(grid_for(nc)), dim3(kBlock), 0, st, (const int *)L0.choice, n, nc, (const double *)r3, r1);  // :82
        ORC_HIP(hipGetLastError());
        ORC_TRY(vec_fill(e1, 0., (int64_t)nc3));                                                                                      // :86
        step("restriction");
        const bool shared_scaling = cfg().amg_shared_scaling && preconditioner == ORC_PRECOND_JACOBI;  // as in multigrid_solve_dev
        ScaledOperator3 scaled3;
        if (shared_scaling) ORC_TRY(jacobi_scaling_prepare3_dev(Ac3, iteration_count, arena, scaled3));
        ORC_TRY(shared_scaling ? bicgstab3_scaled_dev(scaled3, r1, e1, arena) : bicgstab3_dev(Ac3, r1, e1, iteration_count, preconditioner, arena));  // :87-96
        step("pre-smoothing");
        ORC_TRY(residual_norm2_3_dev(Ac3, r1, e1, partials, norm3));                                                                  // :97-105
        step("residual norm");
        hipLaunchKernelGGL(nan_to_status3_k, dim3(1), dim3(64), 0, st, (const double *)norm3, dev_status, (int)ORC_ERR_MULTIGRID_DIVERGED);
        ORC_HIP(hipGetLastError());
        for (int k = 0; k < 3; ++k) {
            lanes[k].stats.amg_levels = 1;
            lanes[k].stats.amg_rows[1] = nc;
            lanes[k].stats.amg_nnz[1] = H[k]->level[0].padded;
            lanes[k].stats.amg_rounds[1] = H[k]->level[0].rounds;
        }
        if (1 < mp.max_levels && nc > 16) {  // :109
            // :110-121 — levels 2.. per system (their level-1 pairings differ), each on its own stream; the recursion receives r'
            double *rk[3], *ck[3];
            for (int k = 0; k < 3; ++k) {
                ORC_TRY(arena.alloc((size_t)nc, &rk[k]));
                ORC_TRY(arena.alloc((size_t)nc, &ck[k]));
            }
            ORC_TRY(deinterleave3_dev(r1, rk[0], rk[1], rk[2], nc));
            ORC_HIP(hipEventRecord(ev_main, st));
            for (int k = 0; k < 3; ++k) {
                MatView Ak;
                Ak.P = H[k]->level[0].P; Ak.val = H[k]->level[0].val; Ak.pk = H[k]->level[0].pk; Ak.xw = H[k]->level[0].xw;
                Ak.rows = H[k]->level[0].rows;
                Ak.symmetric = plain[k].symmetric;
                ORC_HIP(hipStreamWaitEvent(lanes[k].solve_stream, ev_main, 0));
                ORC_TRY(on_lane(k, [&] {
                    return multigrid_solve_dev(Ak, rk[k], 2, mp, convergence_threshold, *lanes[k].vec_arena, &lanes[k].stats, dev_status + k, ck[k], nullptr, nullptr);
                }));
                ORC_HIP(hipEventRecord(ev_lane[k], lanes[k].solve_stream));
                ORC_HIP(hipStreamWaitEvent(st, ev_lane[k], 0));
            }
            if (dbg_sync) for (int k = 0; k < 3; ++k) { (void)hipStreamSynchronize(lanes[k].solve_stream); ORC_TRACE("arm3 level 1: lane %d levels 2.. done", k); }
            hipLaunchKernelGGL(vec_add3_k, dim3(grid_for(nc)), dim3(kBlock), 0, st, e1, (const double *)ck[0], (const double *)ck[1], (const double *)ck[2], nc);  // e' += ...
            ORC_HIP(hipGetLastError());
            step("corrections added");
            ORC_TRY(shared_scaling ? bicgstab3_scaled_dev(scaled3, r1, e1, arena) : bicgstab3_dev(Ac3, r1, e1, iteration_count, preconditioner, arena));  // :123-132
            step("post-smoothing");
        }
        hipLaunchKernelGGL(prolong3_k, dim3(grid_for(n)), dim3(kBlock), 0, st, (const int *)L0.choice, (const int *)L0.chooser, n, (const double *)e1, x3);  // :140, :284
        ORC_HIP(hipGetLastError());
        ORC_TRY(deinterleave3_dev(x3, x[0], x[1], x[2], n));
    }
    int h[4] = {0, 0, 0, 0};
    ORC_TRACE("arm3: coarse parts queued; waiting for the status words");
    ORC_HIP(hipMemcpyAsync(h, dev_status, sizeof(h), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    ORC_TRACE("arm3: done (%d %d %d)", h[0], h[1], h[2]);
    for (int k = 0; k < 3; ++k) status_out[k] = h[k];
    return ORC_OK;
}

}  // namespace orc

// ------------------------------------------------------------------ test hooks (private fns of the reference made observable)
namespace orc {

// x_h / y_h (optional, [ceil(n / 2)]): y = Ac x with the coarse operator AS THE SOLVES MULTIPLY IT — launch_spmv on the level's view,
// i.e. the packed mirror + LDS window product wherever galerkin() built one.  scaled != 0: the view a smoothing solve launches
// (Jacobi scaling 1 / diag materialised into the streamed values: spmv_xwin_k<Epi, 0, false>), else the plain values
// (spmv_xwin_k<Epi>, what the residual checks launch).  *mirror_out: did the level get a window mirror at all?
int amg_debug_coarsen(const MatView &A, Arena &arena, std::vector<int> &choice_h, std::vector<int64_t> &row_ptr_h,
                      std::vector<int64_t> &col_h, std::vector<double> &val_h, int *rounds, const double *x_h, double *y_h, int scaled, int *mirror_out) {
    const int64_t n = A.P.n;
    hipStream_t st = ctx().stream;
    Arena::Mark mk = arena.mark();
    int *choice, *chooser;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &choice));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &chooser));
    CoarseLevel L;
    ORC_TRY(aggregate(A, arena, choice, chooser, &L.rounds));
    if (rounds) *rounds = L.rounds;
    ORC_TRY(galerkin(A, choice, chooser, arena, L));
    choice_h.resize((size_t)n);
    ORC_HIP(hipMemcpyAsync(choice_h.data(), choice, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    const int64_t nc = L.n;
    if (mirror_out) *mirror_out = (L.pk.ptr && L.xw.lidx) ? 1 : 0;
    if (x_h && y_h && nc > 0) {
        MatView V;
        V.P = L.P; V.val = L.val; V.pk = L.pk; V.xw = L.xw; V.symmetric = A.symmetric;
        double *x, *y, *d1;
        ORC_TRY(arena.alloc((size_t)nc, &x));
        ORC_TRY(arena.alloc((size_t)nc, &y));
        ORC_HIP(hipMemcpyAsync(x, x_h, sizeof(double) * (size_t)nc, hipMemcpyHostToDevice, st));
        if (scaled) {
            ORC_TRY(arena.alloc((size_t)nc, &d1));
            ORC_TRY(diag_inverse_dev(V, d1));
            V.s1 = d1;
            ORC_TRY(materialize_scaled_view(V, 50, arena));
        }
        ORC_TRY(spmv_dev(V, x, y));
        ORC_HIP(hipMemcpyAsync(y_h, y, sizeof(double) * (size_t)nc, hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
    }
    std::vector<int> row_len((size_t)nc), col((size_t)std::max<int64_t>(L.padded, 1));
    std::vector<int64_t> slice_ptr((size_t)L.P.n_slices + 1);
    std::vector<double> val((size_t)std::max<int64_t>(L.padded, 1));
    ORC_HIP(hipMemcpyAsync(row_len.data(), L.P.row_len, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(slice_ptr.data(), L.P.slice_ptr, sizeof(int64_t) * slice_ptr.size(), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(col.data(), L.P.col, sizeof(int) * (size_t)L.padded, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(val.data(), L.val, sizeof(double) * (size_t)L.padded, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    row_ptr_h.assign((size_t)nc + 1, 0);
    for (int64_t I = 0; I < nc; ++I) row_ptr_h[(size_t)I + 1] = row_ptr_h[(size_t)I] + row_len[(size_t)I];
    col_h.resize((size_t)row_ptr_h[(size_t)nc]);
    val_h.resize((size_t)row_ptr_h[(size_t)nc]);
    for (int64_t I = 0; I < nc; ++I) {
        const int64_t base = slice_ptr[(size_t)(I >> 6)] + (I & 63);
        for (int k = 0; k < row_len[(size_t)I]; ++k) {
            col_h[(size_t)(row_ptr_h[(size_t)I] + k)] = col[(size_t)(base + (int64_t)k * 64)];
            val_h[(size_t)(row_ptr_h[(size_t)I] + k)] = val[(size_t)(base + (int64_t)k * 64)];
        }
    }
    arena.release(mk);
    return ORC_OK;
}

}  // namespace orc
