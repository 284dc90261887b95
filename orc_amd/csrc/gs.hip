// gs.hip — multicolour Gauss-Seidel: NEW-BUILD EXTENSION, no reference counterpart.
// ORC's Gauss-Seidel arm scans every (i, j) through get() — which panics on the first structural zero of a sparse
// matrix — and ends in panic!("Gauss-Seidel out for maintenance :)") (linear_algebra.rs:219-246); its
// PreconditionMethod knows only None | Jacobi (lib.rs:181-185).  BASELINE configs 3/4 nevertheless ask for
// "multicolour GS-preconditioned BiCGSTAB" and an "AMG V-cycle (GS smoother)", so this file provides
//   ORC_SOLVER_MULTICOLOR_GS        iteration_count sweeps of the reference's row update (:225-239) in colour order,
//   ORC_SOLVER_BICGSTAB_GS_PRECOND  the reference's BiCGSTAB recurrences, right-preconditioned by one GS sweep,
//   ORC_SOLVER_MULTIGRID_GS         the Multigrid arm with GS sweeps as its smoother (amg.hip).
// Rows of one colour share no matrix entry, so a colour is one fully parallel kernel and the sweep is a true
// Gauss-Seidel in colour order; the row sum runs in ascending-column order like everything else.
// Colouring: speculative first-fit with conflict resolution by a deterministic hash priority (64-bit colour mask), on
// the device; the one-class-per-round Jones-Plassmann variant is kept behind ORC_GS_JONES_PLASSMANN=1.
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>

#include "linalg_kernels.hpp"

namespace orc {

// Colour-sorted copy of a pattern (persistent patterns only: it is built on the host once): the rows of a colour are
// contiguous, every colour starts on a slice boundary, columns keep their original numbering.  A sweep over one colour
// is then a product-like pass over a contiguous slice range with fully coalesced matrix reads; with the rows left in
// mesh order every colour's kernel touches every cache line of the matrix (n_colors times the traffic).
struct ColorSell {
    bool built = false;
    int64_t n_slots = 0, padded = 0;
    int n_slices = 0;
    DevBuf<int64_t> slice_ptr;          // [n_slices + 1]
    DevBuf<int> row_len, rowid, diag_off;  // per slot: entries, original row (-1: padding slot), element offset of the diagonal (-1: none)
    DevBuf<int> col;                    // [padded] original column indices
    DevBuf<int> cslot;                  // [padded] [r04] the slot of that column's row: what the slot-space solver gathers through
    DevBuf<int> slot_of_row;            // [n]
    std::vector<int> color_slice;       // [n_colors + 1] first slice of every colour
};

struct Coloring {
    int n_colors = 0;
    DevBuf<int> color;       // [n]   (owned when the colouring is cached; per-solve colourings live in the solve's arena)
    DevBuf<int> rows;        // rows grouped by colour
    int *color_p = nullptr, *rows_p = nullptr;
    bool has_rows = false;
    std::vector<int> start;  // [n_colors + 1]
    ColorSell sorted;
};

// what a sorted sweep reads: the cached ColorSell of a persistent pattern plus per-solve values, or a layout built on the
// device in the solve's arena (coarse AMG levels)
struct SortedView {
    const int64_t *sp = nullptr;
    const int *row_len = nullptr, *rowid = nullptr, *diag_off = nullptr, *col = nullptr;
    const int *cslot = nullptr;  // [r04] columns as slots (slot-space solver)
    int n_slices = 0;
    const double *val = nullptr;
    std::vector<int> color_slice;
    bool ok() const { return val != nullptr; }
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool prio_greater(int a, int b) {  // strict total order on rows
    const unsigned ha = hash32((unsigned)a), hb = hash32((unsigned)b);
    return ha > hb || (ha == hb && a > b);
}

__device__ __forceinline__ int64_t row_base_sell(const SellDev &P, int64_t i) { return P.slice_ptr[i >> 6] + (i & 63); }

// Speculative first-fit colouring with conflict resolution (Gebremedhin-Manne), two kernels per round, Jacobi style so
// that the result does not depend on scheduling: every uncoloured row takes the smallest colour none of its neighbours
// holds in the previous round's state (tentative), then a row that shares its tentative colour with a neighbour coloured
// in the same round gives it up if the neighbour has the higher priority.  A handful of rounds whatever the row
// lengths, where one Jones-Plassmann colour per round needs ~160 rounds on the coarse AMG levels (150 neighbours).
__global__ void spec_assign_k(SellDev P, const int *__restrict__ color_in, int *__restrict__ color_out, int *__restrict__ overflow) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = color_in[i];
        if (ci >= 0) { color_out[i] = ci; continue; }
        const int len = P.row_len[i];
        const int64_t base = row_base_sell(P, i);
        unsigned long long used = 0ull;
        for (int k = 0; k < len; ++k) {
            const int j = P.col[base + (int64_t)k * 64];
            if (j == i || j >= P.n) continue;
            const int cj = color_in[j];
            if (cj >= 0) used |= 1ull << cj;
        }
        const unsigned long long freec = ~used;
        if (freec == 0ull) { atomicExch(overflow, 1); color_out[i] = -1; continue; }
        color_out[i] = __ffsll((long long)freec) - 1;
    }
}
// color_prev: state before this round (-1 = was uncoloured), color_tent: after the tentative assignment; the outcome
// goes to a third array, so every row sees the same tentative state whatever the scheduling.
__global__ void spec_resolve_k(SellDev P, const int *__restrict__ color_prev, const int *__restrict__ color_tent, int *__restrict__ color_out,
                               int *__restrict__ remaining) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = color_tent[i];
        if (color_prev[i] >= 0 || ci < 0) { color_out[i] = ci; if (ci < 0) atomicAdd(remaining, 1); continue; }
        const int len = P.row_len[i];
        const int64_t base = row_base_sell(P, i);
        bool lose = false;
        for (int k = 0; k < len && !lose; ++k) {
            const int j = P.col[base + (int64_t)k * 64];
            if (j == i || j >= P.n) continue;
            if (color_prev[j] < 0 && color_tent[j] == ci && prio_greater(j, (int)i)) lose = true;
        }
        color_out[i] = lose ? -1 : ci;
        if (lose) atomicAdd(remaining, 1);
    }
}

__global__ void color_count_k(const int *__restrict__ color, int64_t n, int *__restrict__ counts) {
    __shared__ int local[64];  // per-workgroup histogram: 64 global atomics per workgroup instead of one per row
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&local[color[i]], 1);
    __syncthreads();
    if (threadIdx.x < 64 && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
}
__global__ void color_fill_k(const int *__restrict__ color, int64_t n, int *__restrict__ cursor, int *__restrict__ rows) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) rows[atomicAdd(&cursor[color[i]], 1)] = (int)i;
}

static int build_coloring(const SellDev &P, Coloring &C, Arena *arena = nullptr, bool want_row_lists = true) {
    const int64_t n = P.n;
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    hipStream_t st = ctx().stream;
    DevBuf<int> tmp_b, tent_b, flags_b, counts_b;
    int *tmp, *tent, *flags, *counts;
    if (arena) {  // per-solve colouring: no hipMalloc / hipFree inside the SIMPLE loop
        ORC_TRY(arena->alloc(nn, &C.color_p));
        ORC_TRY(arena->alloc(nn, &C.rows_p));
        ORC_TRY(arena->alloc(nn, &tmp));
        ORC_TRY(arena->alloc(nn, &tent));
        ORC_TRY(arena->alloc((size_t)2, &flags));
        ORC_TRY(arena->alloc((size_t)128, &counts));
    } else {
        ORC_TRY(C.color.alloc(nn));
        ORC_TRY(C.rows.alloc(nn));
        ORC_TRY(tmp_b.alloc(nn));
        ORC_TRY(tent_b.alloc(nn));
        ORC_TRY(flags_b.alloc(2));
        ORC_TRY(counts_b.alloc(128));
        C.color_p = C.color.p; C.rows_p = C.rows.p; tmp = tmp_b.p; tent = tent_b.p; flags = flags_b.p; counts = counts_b.p;
    }
    ORC_HIP(hipMemsetAsync(C.color_p, 0xff, sizeof(int) * (size_t)n, st));
    int *in = C.color_p, *out = tmp;
    const int g = grid_for(n);
    for (int round = 0; round < 10000; ++round) {
        ORC_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(spec_assign_k, dim3(g), dim3(kBlock), 0, st, P, in, tent, flags + 1);
        hipLaunchKernelGGL(spec_resolve_k, dim3(g), dim3(kBlock), 0, st, P, in, tent, out, flags);
        ORC_HIP(hipGetLastError());
        int h[2];
        ORC_HIP(hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        std::swap(in, out);
        if (h[1]) return set_error(ORC_ERR_BAD_ARGUMENT, "colouring needs more than 64 colours");
        if (h[0] == 0) break;
    }
    if (in != C.color_p) ORC_HIP(hipMemcpyAsync(C.color_p, in, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    ORC_HIP(hipMemsetAsync(counts, 0, 128 * sizeof(int), st));
    hipLaunchKernelGGL(color_count_k, dim3(g), dim3(kBlock), 0, st, C.color_p, n, counts);
    int hc[64];
    ORC_HIP(hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    C.n_colors = 0;
    for (int c = 0; c < 64; ++c) if (hc[c] > 0) C.n_colors = c + 1;
    C.start.assign((size_t)C.n_colors + 1, 0);
    for (int c = 0; c < C.n_colors; ++c) C.start[(size_t)c + 1] = C.start[(size_t)c] + hc[c];
    if (want_row_lists) {  // only the unsorted sweeps walk row lists
        ORC_HIP(hipMemcpyAsync(counts + 64, C.start.data(), sizeof(int) * (size_t)C.n_colors, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(color_fill_k, dim3(g), dim3(kBlock), 0, st, C.color_p, n, counts + 64, C.rows_p);
        ORC_HIP(hipGetLastError());
        ORC_HIP(hipStreamSynchronize(st));
        C.has_rows = true;
    }
    return ORC_OK;
}

// Row update of the reference's Gauss-Seidel arm (linear_algebra.rs:225-239) for the rows of one colour:
//   x_i = x_i (1 - w) + w (b_i - sum_{j != i} a_ij x_j) / a_ii ,  the j == i term contributing the literal 0.
__global__ void gs_color_k(MatView A, const double *__restrict__ b, double *__restrict__ x, const int *__restrict__ rows, int count, double omega,
                           int *__restrict__ status) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const int i = rows[idx];
        const int len = A.P.row_len[i];
        const int64_t base = A.P.slice_ptr[i >> 6] + (i & 63);
        const int d = A.P.diag_pos[i];
        if (d < 0) {
            // get(i, i) panics (lib.rs:664); an EMPTY row (coarse AMG row whose fine rows found no partner) has nothing to relax
            if (len > 0) atomicCAS(status, 0, (int)ORC_ERR_STRUCTURAL_ZERO);
            continue;
        }
        double sum = 0.;
        for (int k = 0; k < len; ++k) {
            const int64_t pos = base + (int64_t)k * 64;
            const int j = A.P.col[pos];
            sum += (j == i) ? 0. : view_value(A, i, pos) * x[j];
        }
        const double xi = x[i] * (1. - omega) + omega * (b[i] - sum) / view_value(A, i, d);
        x[i] = xi;
        if (xi != xi) atomicCAS(status, 0, (int)ORC_ERR_SOLUTION_DIVERGED);  // :240-242
    }
}

// ---- colour-sorted layout
static int build_color_sell(const SellDev &P, Coloring &C) {
    const int64_t n = P.n;
    hipStream_t st = ctx().stream;
    std::vector<int> color((size_t)n), row_len((size_t)n), diag((size_t)n);
    std::vector<int64_t> sp((size_t)P.n_slices + 1);
    ORC_HIP(hipMemcpyAsync(color.data(), C.color_p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(row_len.data(), P.row_len, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(diag.data(), P.diag_pos, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(sp.data(), P.slice_ptr, sizeof(int64_t) * sp.size(), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    const int64_t padded_in = sp[(size_t)P.n_slices];
    std::vector<int> col((size_t)std::max<int64_t>(padded_in, 1));
    ORC_HIP(hipMemcpy(col.data(), P.col, sizeof(int) * (size_t)padded_in, hipMemcpyDeviceToHost));
    ColorSell &S = C.sorted;
    // slots: colour after colour, rows ascending inside a colour, every colour padded to whole slices
    std::vector<int64_t> cursor((size_t)C.n_colors + 1, 0);
    S.color_slice.assign((size_t)C.n_colors + 1, 0);
    for (int c = 0; c < C.n_colors; ++c) {
        const int64_t cnt = C.start[(size_t)c + 1] - C.start[(size_t)c];
        S.color_slice[(size_t)c + 1] = S.color_slice[(size_t)c] + (int)((cnt + 63) / 64);
        cursor[(size_t)c] = (int64_t)S.color_slice[(size_t)c] * 64;
    }
    S.n_slices = S.color_slice[(size_t)C.n_colors];
    S.n_slots = (int64_t)S.n_slices * 64;
    std::vector<int> slot_of_row((size_t)n), rowid((size_t)std::max<int64_t>(S.n_slots, 1), -1), len_p((size_t)std::max<int64_t>(S.n_slots, 1), 0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t slot = cursor[(size_t)color[(size_t)i]]++;
        slot_of_row[(size_t)i] = (int)slot;
        rowid[(size_t)slot] = (int)i;
        len_p[(size_t)slot] = row_len[(size_t)i];
    }
    std::vector<int64_t> sp_p((size_t)S.n_slices + 1, 0);
    for (int sidx = 0; sidx < S.n_slices; ++sidx) {
        int w = 0;
        for (int l = 0; l < 64; ++l) w = std::max(w, len_p[(size_t)sidx * 64 + l]);
        sp_p[(size_t)sidx + 1] = sp_p[(size_t)sidx] + (int64_t)w * 64;
    }
    S.padded = sp_p[(size_t)S.n_slices];
    if (S.padded >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "colour-sorted matrix too large for 32-bit offsets");
    std::vector<int> col_p((size_t)std::max<int64_t>(S.padded, 1), 0), diag_p((size_t)std::max<int64_t>(S.n_slots, 1), -1);
    std::vector<int> cslot_p((size_t)std::max<int64_t>(S.padded, 1), 0);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t slot = slot_of_row[(size_t)i];
        const int64_t src = sp[(size_t)(i >> 6)] + (i & 63), dst = sp_p[(size_t)(slot >> 6)] + (slot & 63);
        for (int k = 0; k < row_len[(size_t)i]; ++k) {
            col_p[(size_t)(dst + (int64_t)k * 64)] = col[(size_t)(src + (int64_t)k * 64)];
            const int cj = col[(size_t)(src + (int64_t)k * 64)];
            cslot_p[(size_t)(dst + (int64_t)k * 64)] = cj < n ? slot_of_row[(size_t)cj] : -1;  // (ghost columns: the slot-space solver is single-GPU)
            if (diag[(size_t)i] == (int)(src + (int64_t)k * 64)) diag_p[(size_t)slot] = (int)(dst + (int64_t)k * 64);
        }
    }
    ORC_TRY(S.slice_ptr.upload(sp_p.data(), sp_p.size()));
    ORC_TRY(S.row_len.upload(len_p.data(), len_p.size()));
    ORC_TRY(S.rowid.upload(rowid.data(), rowid.size()));
    ORC_TRY(S.diag_off.upload(diag_p.data(), diag_p.size()));
    ORC_TRY(S.col.upload(col_p.data(), col_p.size()));
    ORC_TRY(S.cslot.upload(cslot_p.data(), cslot_p.size()));
    ORC_TRY(S.slot_of_row.upload(slot_of_row.data(), slot_of_row.size()));
    S.built = true;
    return ORC_OK;
}

// values of the matrix view -> colour-sorted storage: thread per original row (coalesced reads), the scalings applied
__global__ void gs_permute_values_k(MatView A, const int *__restrict__ slot_of_row, const int64_t *__restrict__ sp_p, double *__restrict__ val_p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int len = A.P.row_len[i];
        const int64_t src = A.P.slice_ptr[i >> 6] + (i & 63);
        const int64_t slot = slot_of_row[i];
        const int64_t dst = sp_p[slot >> 6] + (slot & 63);
        for (int k = 0; k < len; ++k) val_p[dst + (int64_t)k * 64] = view_value(A, i, src + (int64_t)k * 64);
    }
}

// the row update of gs_color_k for the slices [slice_lo, slice_hi) of one colour: one lane per slot
__global__ __launch_bounds__(kBlock) void gs_color_sorted_k(const int64_t *__restrict__ sp, const int *__restrict__ row_len, const int *__restrict__ rowid,
                                                            const int *__restrict__ diag_off, const int *__restrict__ col, const double *__restrict__ val,
                                                            const double *__restrict__ b, double *__restrict__ x, int slice_lo, int slice_hi, double omega,
                                                            int *__restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    for (int sidx = slice_lo + blockIdx.x * waves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); sidx < slice_hi; sidx += gridDim.x * waves) {  // (scalar: see SliceWalk)
        const int64_t slot = (int64_t)sidx * 64 + lane;
        const int i = rowid[slot];
        const int len = i >= 0 ? row_len[slot] : 0;
        const int64_t base = sp[sidx] + lane;
        const int width = (int)((sp[sidx + 1] - sp[sidx]) >> 6);
        double sum = 0.;
        // chunks of 8 like the product: column and value loads first, then the x gathers, then the ordered additions
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[8], xv[8];
            const int64_t p0 = base + (int64_t)k0 * 64;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = k0 + u < len;
                c[u] = in ? col[p0 + (int64_t)u * 64] : 0;
                v[u] = in ? val[p0 + (int64_t)u * 64] : 0.;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = (k0 + u < len) ? x[c[u]] : 0.;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < len) sum += (c[u] == i) ? 0. : v[u] * xv[u];
        }
        if (i < 0) continue;
        const int d = diag_off[slot];
        if (d < 0) {
            if (len > 0) atomicCAS(status, 0, (int)ORC_ERR_STRUCTURAL_ZERO);
            continue;
        }
        const double xi = x[i] * (1. - omega) + omega * (b[i] - sum) / val[d];
        x[i] = xi;
        if (xi != xi) atomicCAS(status, 0, (int)ORC_ERR_SOLUTION_DIVERGED);  // :240-242
    }
}

static int gs_sweep_sorted(const Coloring &C, const SortedView &V, const double *b, double *x, double omega, int *status) {
    for (int c = 0; c < C.n_colors; ++c) {
        const int lo = V.color_slice[(size_t)c], hi = V.color_slice[(size_t)c + 1];
        if (hi <= lo) continue;
        const int g = std::min(kMaxGrid, (hi - lo + 3) / 4);
        hipLaunchKernelGGL(gs_color_sorted_k, dim3(g), dim3(kBlock), 0, ctx().stream, V.sp, V.row_len, V.rowid, V.diag_off, V.col, V.val, b, x, lo, hi, omega,
                           status);
    }
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// ---- the same layout built on the device, in the solve's arena, for patterns that live for one solve (coarse AMG
// levels).  Rank of a row inside its colour = rows of that colour in earlier 64-row blocks (column scan of a
// [blocks x colours] count table) + rows of that colour before it in its own block (ballot).
__global__ __launch_bounds__(64) void gs_block_counts_k(const int *__restrict__ color, int64_t n, int n_colors, int *__restrict__ table,
                                                        int *__restrict__ intra) {
    const int lane = threadIdx.x;
    const int64_t n_blocks = (n + 63) / 64;
    for (int64_t bidx = blockIdx.x; bidx < n_blocks; bidx += gridDim.x) {
        const int64_t i = bidx * 64 + lane;
        const int c = i < n ? color[i] : -1;
        for (int q = 0; q < n_colors; ++q) {
            const unsigned long long m = __ballot(c == q);
            if (c == q) intra[i] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) table[bidx * n_colors + q] = __popcll(m);
        }
    }
}
// exclusive scan down every column of the table: one workgroup per colour
__global__ __launch_bounds__(1024) void gs_column_scan_k(int *__restrict__ table, int64_t n_blocks, int n_colors) {
    __shared__ int carry;
    __shared__ int buf[1024];
    const int q = blockIdx.x;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t e = base + threadIdx.x;
        const int v = e < n_blocks ? table[e * n_colors + q] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (e < n_blocks) table[e * n_colors + q] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}
__global__ void gs_assign_slots_k(SellDev P, const int *__restrict__ color, const int *__restrict__ table, const int *__restrict__ intra, int n_colors,
                                  const int *__restrict__ color_first_slot, int *__restrict__ slot_of_row, int *__restrict__ rowid, int *__restrict__ len_p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = color[i];
        const int slot = color_first_slot[c] + table[(i >> 6) * n_colors + c] + intra[i];
        slot_of_row[i] = slot;
        rowid[slot] = (int)i;
        len_p[slot] = P.row_len[i];
    }
}
__global__ __launch_bounds__(1024) void gs_slice_ptr_k(const int *__restrict__ len_p, int n_slices, int64_t *__restrict__ slice_ptr) {
    __shared__ long long carry;
    __shared__ long long buf[1024];
    if (threadIdx.x == 0) { carry = 0; slice_ptr[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n_slices; base += 1024) {
        const int sidx = base + threadIdx.x;
        long long w = 0;
        if (sidx < n_slices) {
            int mx = 0;
            for (int l = 0; l < 64; ++l) mx = max(mx, len_p[(int64_t)sidx * 64 + l]);
            w = (long long)mx * 64;
        }
        buf[threadIdx.x] = w;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const long long t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (sidx < n_slices) slice_ptr[sidx + 1] = carry + buf[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}
// pattern and values of the view -> colour-sorted storage (thread per original row), diagonal offsets on the way
__global__ void gs_permute_all_k(MatView A, const int *__restrict__ slot_of_row, const int64_t *__restrict__ sp_p, int *__restrict__ col_p,
                                 double *__restrict__ val_p, int *__restrict__ diag_p, int *__restrict__ cslot_p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int len = A.P.row_len[i];
        const int64_t src = A.P.slice_ptr[i >> 6] + (i & 63);
        const int64_t slot = slot_of_row[i];
        const int64_t dst = sp_p[slot >> 6] + (slot & 63);
        int d = -1;
        for (int k = 0; k < len; ++k) {
            const int j = A.P.col[src + (int64_t)k * 64];
            col_p[dst + (int64_t)k * 64] = j;
            cslot_p[dst + (int64_t)k * 64] = j < A.P.n ? slot_of_row[j] : -1;
            val_p[dst + (int64_t)k * 64] = view_value(A, i, src + (int64_t)k * 64);
            if (j == i) d = (int)(dst + (int64_t)k * 64);
        }
        diag_p[slot] = d;
    }
}

static int build_sorted_on_device(const MatView &A, const Coloring &C, Arena &arena, SortedView &V) {
    const int64_t n = A.P.n;
    hipStream_t st = ctx().stream;
    const int nc = C.n_colors;
    V.color_slice.assign((size_t)nc + 1, 0);
    std::vector<int> first_slot((size_t)std::max(nc, 1), 0);
    for (int c = 0; c < nc; ++c) {
        const int64_t cnt = C.start[(size_t)c + 1] - C.start[(size_t)c];
        first_slot[(size_t)c] = V.color_slice[(size_t)c] * 64;
        V.color_slice[(size_t)c + 1] = V.color_slice[(size_t)c] + (int)((cnt + 63) / 64);
    }
    const int n_slices = V.color_slice[(size_t)nc];
    const int64_t n_slots = (int64_t)n_slices * 64, n_blocks = (n + 63) / 64;
    if (n_slices == 0) return ORC_OK;
    int *table, *intra, *first_dev, *slot_of_row, *rowid, *len_p, *diag_p, *col_p;
    int64_t *sp_p;
    double *val_p;
    ORC_TRY(arena.alloc((size_t)(n_blocks * nc), &table));
    ORC_TRY(arena.alloc((size_t)n, &intra));
    ORC_TRY(arena.alloc((size_t)nc, &first_dev));
    ORC_TRY(arena.alloc((size_t)n, &slot_of_row));
    ORC_TRY(arena.alloc((size_t)n_slots, &rowid));
    ORC_TRY(arena.alloc((size_t)n_slots, &len_p));
    ORC_TRY(arena.alloc((size_t)n_slots, &diag_p));
    ORC_TRY(arena.alloc((size_t)n_slices + 1, &sp_p));
    ORC_HIP(hipMemcpyAsync(first_dev, first_slot.data(), sizeof(int) * (size_t)nc, hipMemcpyHostToDevice, st));
    ORC_HIP(hipMemsetAsync(rowid, 0xff, sizeof(int) * (size_t)n_slots, st));
    ORC_HIP(hipMemsetAsync(len_p, 0, sizeof(int) * (size_t)n_slots, st));
    ORC_HIP(hipMemsetAsync(diag_p, 0xff, sizeof(int) * (size_t)n_slots, st));
    hipLaunchKernelGGL(gs_block_counts_k, dim3((unsigned)std::min<int64_t>(n_blocks, 8192)), dim3(64), 0, st, C.color_p, n, nc, table, intra);
    hipLaunchKernelGGL(gs_column_scan_k, dim3(nc), dim3(1024), 0, st, table, n_blocks, nc);
    hipLaunchKernelGGL(gs_assign_slots_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A.P, C.color_p, table, intra, nc, first_dev, slot_of_row, rowid, len_p);
    hipLaunchKernelGGL(gs_slice_ptr_k, dim3(1), dim3(1024), 0, st, len_p, n_slices, sp_p);
    ORC_HIP(hipGetLastError());
    int64_t padded = 0;
    ORC_HIP(hipMemcpyAsync(&padded, sp_p + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    if (padded >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "colour-sorted matrix too large for 32-bit offsets");
    int *cslot_p;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &col_p));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &cslot_p));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &val_p));
    hipLaunchKernelGGL(gs_permute_all_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A, slot_of_row, sp_p, col_p, val_p, diag_p, cslot_p);
    ORC_HIP(hipGetLastError());
    V.sp = sp_p; V.row_len = len_p; V.rowid = rowid; V.diag_off = diag_p; V.col = col_p; V.val = val_p; V.cslot = cslot_p; V.n_slices = n_slices;
    return ORC_OK;
}

static int gs_sweep(const MatView &A, const Coloring &C, const double *b, double *x, double omega, int *status, const SortedView *sorted = nullptr) {
    if (sorted && sorted->ok()) return gs_sweep_sorted(C, *sorted, b, x, omega, status);
    if (!C.has_rows) return set_error(ORC_ERR_BAD_ARGUMENT, "colouring without row lists: the colour-sorted layout is missing");
    for (int c = 0; c < C.n_colors; ++c) {
        const int cnt = C.start[(size_t)c + 1] - C.start[(size_t)c];
        if (cnt == 0) continue;
        hipLaunchKernelGGL(gs_color_k, dim3(grid_for(cnt)), dim3(kBlock), 0, ctx().stream, A, b, x, C.rows_p + C.start[(size_t)c], cnt, omega, status);
    }
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// colourings of persistent patterns (the mesh pattern of a solver) are kept; AMG levels are coloured per solve
static std::map<const void *, std::unique_ptr<Coloring>> &color_cache() {
    static std::map<const void *, std::unique_ptr<Coloring>> c;
    return c;
}
void gs_forget_pattern(const void *col_ptr) { color_cache().erase(col_ptr); }

static int get_coloring(const MatView &A, std::unique_ptr<Coloring> &owned, const Coloring **out, Arena *arena) {
    if (A.persistent_pattern) {
        static std::mutex cache_mutex;  // the momentum lanes solve on the same (mesh) pattern at the same time
        std::lock_guard<std::mutex> lock(cache_mutex);
        auto &cache = color_cache();
        auto it = cache.find((const void *)A.P.col);
        if (it == cache.end()) {
            auto c = std::make_unique<Coloring>();
            ORC_TRY(build_coloring(A.P, *c));
            ORC_TRY(build_color_sell(A.P, *c));
            it = cache.emplace((const void *)A.P.col, std::move(c)).first;
        }
        *out = it->second.get();
        return ORC_OK;
    }
    owned = std::make_unique<Coloring>();
    ORC_TRY(build_coloring(A.P, *owned, arena, false));
    *out = owned.get();
    return ORC_OK;
}

// Right-preconditioned recurrences with the same breakdown guard as the reference arm (linalg.hip): scal[5] / scal[6]
// are the frozen flags (5: set by kernels that react with a no-op, 6: by the x/r update).
__device__ __forceinline__ bool pre_frozen(const double *scal, int guard) { return guard && (scal[5] != 0. || scal[6] != 0.); }
__device__ __forceinline__ bool fin_nz(double v) { return v != 0. && isfinite(v); }

// x = (x + alpha p^) + omega s^ ; r = s - omega t ; partial sum(r)
__global__ __launch_bounds__(kBlock) void bicg_xr_pre_k(double *__restrict__ scal, int rho_idx, int i_sum_nu, int i_ts, int i_tt, double *__restrict__ x,
                                                        const double *__restrict__ ph, const double *__restrict__ sh, const double *__restrict__ s,
                                                        const double *__restrict__ t, double *__restrict__ r, int64_t n, double *__restrict__ partials, int guard) {
    __shared__ double lds[8];
    if (guard && scal[5] != 0.) return;
    const double alpha = scal[rho_idx] / scal[i_sum_nu];
    double omega = scal[i_ts] / scal[i_tt];
    const bool bad = guard && !(fin_nz(scal[i_tt]) && isfinite(omega));
    if (bad) omega = 0.;
    double acc = 0.;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double h = x[i] + alpha * ph[i];
        x[i] = bad ? h : h + omega * sh[i];
        const double ri = bad ? s[i] : s[i] - omega * t[i];
        r[i] = ri;
        acc += ri;
    }
    if (bad && blockIdx.x == 0 && threadIdx.x == 0) scal[6] = 1.;
    const double tsum = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = tsum;
}
__global__ void bicg_s_pre_k(double *__restrict__ scal, int rho_idx, int i_sum_nu, const double *__restrict__ r, const double *__restrict__ nu,
                             double *__restrict__ s, int64_t n, int guard) {
    if (pre_frozen(scal, guard)) return;
    const double alpha = scal[rho_idx] / scal[i_sum_nu];
    if (guard && !(fin_nz(scal[rho_idx]) && fin_nz(scal[i_sum_nu]) && isfinite(alpha))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[5] = 1.;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s[i] = r[i] - alpha * nu[i];
}
__global__ void bicg_p_pre_k(double *__restrict__ scal, int rho_prev_idx, int rho_idx, int i_sum_nu, int i_ts, int i_tt, const double *__restrict__ r,
                             const double *__restrict__ nu, double *__restrict__ p, int64_t n, int guard) {
    if (pre_frozen(scal, guard)) return;
    const double rho_prev = scal[rho_prev_idx], rho = scal[rho_idx];
    const double alpha = rho_prev / scal[i_sum_nu];
    const double omega = scal[i_ts] / scal[i_tt];
    const double beta = rho / rho_prev * alpha / omega;
    if (guard && !(fin_nz(omega) && isfinite(beta))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[5] = 1.;
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = r[i] + beta * (p[i] - omega * nu[i]);
}
struct EpiSum {  // y = A x ; partial sum(y)
    static constexpr int kReductions = 1;
    double *y;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &) const { y[row] = acc; r0 += acc; }
};
struct EpiTsPre {  // t = A s^ ; partials t.s, t.t  (s is the un-preconditioned residual)
    static constexpr int kReductions = 2;
    const double *s;
    double *t;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &r1) const { t[row] = acc; r0 += acc * s[row]; r1 += acc * acc; }
};
struct EpiRes {  // r = b - A x ; p = r ; partial sum(r)
    static constexpr int kReductions = 1;
    const double *b;
    double *r, *p;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &) const { const double v = b[row] - acc; r[row] = v; p[row] = v; r0 += v; }
};

template <class Epi>
static int spmv_launch(const MatView &A, const double *x, const Epi &epi, double *partials, int *grid_out, const double *skip = nullptr) {
    int64_t g = ((int64_t)A.P.n_slices + 3) / 4;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g >= 8) g = (g / 8) * 8;
    if (g < 1) g = 1;
    *grid_out = (int)g;
    if (A.halo) ORC_TRY(A.halo->exchange(const_cast<double *>(x)));  // C1: refresh the ghost entries of x
    hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_k<Epi>), dim3((unsigned)g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// ------------------------------------------------------------------ [r04] GS-preconditioned BiCGSTAB in SLOT SPACE, S systems in lock-step
// BASELINE configs[2] (1.03 M cells): r03's solve spent 54 % of its kernel time in 10-microsecond colour launches at 0.2 of peak,
// another 12 % in one-workgroup folds and zero fills, and ran u, v, w as three such chains.  Here the whole solve lives in the
// colour-sorted numbering ("slots"): the vectors are permuted once on the way in and out, products and sweeps gather through the
// slot of a column (ColorSell::cslot) — still adding a row's entries in their original ascending-column order, so every row sum has
// its bits — and
//   * the preconditioner application x^ = M^-1 b (one sweep from zero, omega = 1) needs no zero fill: a colour's kernel knows that
//     every column of its own or a later colour still holds the zero of the start (a slot compare against the colour's first slot),
//     skips those gathers, and their products — exact zeros that leave the running sum unchanged — with them;
//   * the three sums of an iteration are folded by the kernels that consume them (fold_partials_block, as in the reference arm);
//   * S = 3: the momentum systems share the mesh pattern, so one launch serves u, v and w — interleaved vectors x[3 slot + s], three
//     value streams, one column stream: a third of the launches, three times the work per launch of these latency-bound kernels.
// Per system the arithmetic of S = 3 is that of S = 1 (same grids, same thread -> slot map, same folds): bit-identical (tests).
template <int S> struct SlotMat {
    const int64_t *sp;
    const int *row_len, *rowid, *diag_off, *cslot;
    const double *val[S];
    int n_slices;
    int64_t n_slots;
};
template <int S> struct CVecs { const double *p[S]; };
template <int S> struct MVecs { double *p[S]; };
enum { GX_RHO0 = 0, GX_RHO1 = 1, GX_SUM_NU = 2, GX_TS = 3, GX_TT = 4, GX_FROZEN = 5, GX_FROZEN2 = 6, GX_STRIDE = 8 };

template <int S>
__global__ void gsx_in_k(const int *__restrict__ rowid, int64_t n_slots, CVecs<S> src, double *__restrict__ dst) {
    for (int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < n_slots; slot += (int64_t)gridDim.x * blockDim.x) {
        const int i = rowid[slot];
#pragma unroll
        for (int s = 0; s < S; ++s) dst[S * slot + s] = i >= 0 ? src.p[s][i] : 0.;
    }
}
template <int S>
__global__ void gsx_out_k(const int *__restrict__ rowid, int64_t n_slots, const double *__restrict__ src, MVecs<S> dst) {
    for (int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < n_slots; slot += (int64_t)gridDim.x * blockDim.x) {
        const int i = rowid[slot];
        if (i < 0) continue;
#pragma unroll
        for (int s = 0; s < S; ++s) dst.p[s][i] = src[S * slot + s];
    }
}

// kMode 0: r = b - A x, p = r, sum(r)   1: y = A x, sum(y)   2: t = A x, t.aux, t.t      (linear_algebra.rs:250-261)
// partial sums: quantity q of system s at partials[(s * kRed + q) * gridDim.x + blockIdx.x]
// (r04, measured and dropped: skipping the value stream and the products of a system the guard has frozen — per-system predicates in these
// two kernels cost 10-25 % on the unfrozen path (config 3: 28.9 -> 31.3 ms per iteration) and no system of that run is ever frozen)
template <int S, int kMode>
__global__ __launch_bounds__(kBlock) void gsx_spmv_k(SlotMat<S> M, const double *__restrict__ x, const double *__restrict__ aux, double *__restrict__ y,
                                                     double *__restrict__ y2, double *__restrict__ partials) {
    __shared__ double lds[8];
    constexpr int kRed = kMode == 2 ? 2 : 1;
    const int lane = threadIdx.x & 63;
    double red[S][2];
#pragma unroll
    for (int s = 0; s < S; ++s) red[s][0] = red[s][1] = 0.;
    SliceWalk w(M.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t slot = slice * 64 + lane;
        const int i = M.rowid[slot];
        const int len = i >= 0 ? M.row_len[slot] : 0;
        const int64_t base = M.sp[slice] + lane;
        const int width = (int)((M.sp[slice + 1] - M.sp[slice]) >> 6);
        double acc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.;
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[S][8];
            const int64_t p0 = base + (int64_t)k0 * 64;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = k0 + u < len;
                c[u] = in ? M.cslot[p0 + (int64_t)u * 64] : 0;
#pragma unroll
                for (int s = 0; s < S; ++s) v[s][u] = in ? M.val[s][p0 + (int64_t)u * 64] : 0.;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < len && c[u] >= 0) {  // (c < 0: a ghost column — never in a slot-space solve, which is single-GPU)
#pragma unroll
                    for (int s = 0; s < S; ++s) acc[s] += v[s][u] * x[(int64_t)S * c[u] + s];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int64_t e = S * slot + s;
            if (kMode == 0) {
                const double r = i >= 0 ? aux[e] - acc[s] : 0.;
                y[e] = r; y2[e] = r;
                red[s][0] += r;
            } else if (kMode == 1) {
                const double o = i >= 0 ? acc[s] : 0.;
                y[e] = o;
                red[s][0] += o;
            } else {
                const double o = i >= 0 ? acc[s] : 0.;
                y[e] = o;
                if (i >= 0) { red[s][0] += o * aux[e]; red[s][1] += o * o; }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int q = 0; q < kRed; ++q) {
            const double t = block_sum(red[s][q], lds);
            if (threadIdx.x == 0) partials[(size_t)(s * kRed + q) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// x^ = M^-1 b for the slots [slice_lo, slice_hi) of one colour: gs_color_sorted_k's row update (linear_algebra.rs:225-239, omega = 1)
// from a zero start — the entries whose column belongs to this or a later colour (cslot >= first slot of the colour) multiply the
// zero of the start: skipped with their gathers (an exact zero added to the running sum leaves it as it is)
template <int S>
__global__ __launch_bounds__(kBlock) void gsx_sweep0_k(SlotMat<S> M, const double *__restrict__ b, double *__restrict__ xh, int slice_lo, int slice_hi,
                                                       int *__restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int first_slot = slice_lo * 64;
    for (int sidx = slice_lo + blockIdx.x * waves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); sidx < slice_hi; sidx += gridDim.x * waves) {  // (scalar: see SliceWalk)
        const int64_t slot = (int64_t)sidx * 64 + lane;
        const int i = M.rowid[slot];
        const int len = i >= 0 ? M.row_len[slot] : 0;
        const int64_t base = M.sp[sidx] + lane;
        const int width = (int)((M.sp[sidx + 1] - M.sp[sidx]) >> 6);
        double sum[S];
#pragma unroll
        for (int s = 0; s < S; ++s) sum[s] = 0.;
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[S][8];
            const int64_t p0 = base + (int64_t)k0 * 64;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = k0 + u < len;
                c[u] = in ? M.cslot[p0 + (int64_t)u * 64] : first_slot;
                const bool prev = c[u] >= 0 && c[u] < first_slot;  // a column swept by an earlier colour: the only ones that are not zero yet
#pragma unroll
                for (int s = 0; s < S; ++s) v[s][u] = prev ? M.val[s][p0 + (int64_t)u * 64] : 0.;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (c[u] >= 0 && c[u] < first_slot) {
#pragma unroll
                    for (int s = 0; s < S; ++s) sum[s] += v[s][u] * xh[(int64_t)S * c[u] + s];
                }
            }
        }
        const int d = i >= 0 ? M.diag_off[slot] : -1;
        if (i >= 0 && d < 0 && len > 0) atomicCAS(status, 0, (int)ORC_ERR_STRUCTURAL_ZERO);  // get(i, i) panics (lib.rs:664)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            double xi = 0.;
            if (d >= 0) xi = 0. * (1. - 1.) + 1. * (b[S * slot + s] - sum[s]) / M.val[s][d];  // x_i (1 - w) + w (b_i - sum) / a_ii with x_i = 0, w = 1
            xh[S * slot + s] = xi;
        }
    }
}

__device__ __forceinline__ bool gx_frozen(const double *sc, int guard) { return guard && (sc[GX_FROZEN] != 0. || sc[GX_FROZEN2] != 0.); }

// rho_0 = sum(r) of every system from the residual product's partial sums (one workgroup)
template <int S>
__global__ __launch_bounds__(kBlock) void gsx_rho0_k(double *__restrict__ scal, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[S * 16];
    double rho[S];
    fold_partials_multi<S>(fold, fold_count, lds, rho);
    if (threadIdx.x == 0)
        for (int s = 0; s < S; ++s) scal[GX_STRIDE * s + GX_RHO0] = rho[s];
}
// The vector kernels work on the interleaved vectors as FLAT arrays (element e = S slot + s belongs to system e % S): fully coalesced
// 8-byte accesses — a thread that handled "its slot's S values" touched 24-byte strides, 12 cache lines per wave instruction instead
// of 4 (r04's first form: 1.9 TB/s).  Element-wise results do not depend on who computes them; the one reduction (sum(r) in gsx_xr_k)
// hands its r values back to the slot's own thread through LDS, so the partial sums keep the thread -> slot map of S = 1.
// s = r - alpha nu, alpha = rho / sum(nu)                                         (:257, :259)
template <int S>
__global__ __launch_bounds__(kBlock) void gsx_s_k(double *__restrict__ scal, int rho_idx, const double *__restrict__ r, const double *__restrict__ nu,
                                                  double *__restrict__ sv, int64_t n, int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[S * 16];
    double alpha[S], sum_nu[S];
    bool act[S];
    fold_partials_multi<S>(fold, fold_count, lds, sum_nu);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double *sc = scal + GX_STRIDE * s;
        const bool frz = gx_frozen(sc, guard);
        const double rho = sc[rho_idx];
        alpha[s] = rho / sum_nu[s];
        const bool bad = guard && !(fin_nz(rho) && fin_nz(sum_nu[s]) && isfinite(alpha[s]));
        act[s] = !frz && !bad;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !frz) {
            sc[GX_SUM_NU] = sum_nu[s];
            if (bad) sc[GX_FROZEN] = 1.;
        }
    }
    const int64_t total = (int64_t)S * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int s = (int)(e % S);
        double a = alpha[0];
        bool on = act[0];
#pragma unroll
        for (int q = 1; q < S; ++q) { if (s == q) { a = alpha[q]; on = act[q]; } }
        if (on) sv[e] = r[e] - a * nu[e];
    }
}
// x = (x + alpha p^) + omega s^ ; r = s - omega t ; partial sum(r)                (:258, :261-265, right-preconditioned)
template <int S>
__global__ __launch_bounds__(kBlock) void gsx_xr_k(double *__restrict__ scal, int rho_idx, double *__restrict__ x, const double *__restrict__ ph,
                                                   const double *__restrict__ sh, const double *__restrict__ sv, const double *__restrict__ t,
                                                   double *__restrict__ r, const int *__restrict__ rowid, int64_t n, double *__restrict__ partials,
                                                   int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[8];
    __shared__ double lds_f[2 * S * 16];
    __shared__ double r_tile[S * kBlock];
    double alpha[S], omega[S], acc[S], tsq[2 * S];
    int state[S];  // 0 = normal, 1 = t vanished or overflowed (x = h, r = s, stop), 2 = frozen (no-op)
    fold_partials_multi<2 * S>(fold, fold_count, lds_f, tsq);  // (t.s, t.t) of system s at 2 s, 2 s + 1
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double *sc = scal + GX_STRIDE * s;
        const bool frz = guard && sc[GX_FROZEN] != 0.;
        const double ts = tsq[2 * s], tt = tsq[2 * s + 1];
        alpha[s] = sc[rho_idx] / sc[GX_SUM_NU];
        omega[s] = ts / tt;
        const bool bad = guard && !(fin_nz(tt) && isfinite(omega[s]));
        if (bad) omega[s] = 0.;
        state[s] = frz ? 2 : (bad ? 1 : 0);
        acc[s] = 0.;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !frz) {
            sc[GX_TS] = ts; sc[GX_TT] = tt;
            if (bad) sc[GX_FROZEN2] = 1.;
        }
    }
    // a workgroup takes chunks of kBlock consecutive slots (= S kBlock consecutive doubles of every vector); thread tid owns slot
    // chunk * kBlock + tid for the sums — S = 1's map — and computes the flat elements chunk * S kBlock + tid + j kBlock, j < S
    const int64_t n_chunks = (n + kBlock - 1) / kBlock, total = (int64_t)S * n;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t base = chunk * (int64_t)S * kBlock;
        double xv[S], pv[S], hv[S], sv_[S], tv[S];
#pragma unroll
        for (int j = 0; j < S; ++j) {  // all loads of the chunk first: S x 5 requests in flight per lane
            const int64_t e = base + threadIdx.x + (int64_t)j * kBlock;
            const bool in = e < total;
            xv[j] = in ? x[e] : 0.; pv[j] = in ? ph[e] : 0.; hv[j] = in ? sh[e] : 0.; sv_[j] = in ? sv[e] : 0.; tv[j] = in ? t[e] : 0.;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int64_t e = base + threadIdx.x + (int64_t)j * kBlock;
            double ri = 0.;
            if (e < total) {
                const int s = (int)(e % S);
                double a = alpha[0], o = omega[0];
                int stt = state[0];
#pragma unroll
                for (int q = 1; q < S; ++q) { if (s == q) { a = alpha[q]; o = omega[q]; stt = state[q]; } }
                if (stt != 2) {
                    const double h = xv[j] + a * pv[j];
                    x[e] = stt == 1 ? h : h + o * hv[j];
                    ri = stt == 1 ? sv_[j] : sv_[j] - o * tv[j];
                    r[e] = ri;
                }
            }
            r_tile[threadIdx.x + j * kBlock] = ri;
        }
        __syncthreads();
        const int64_t slot = chunk * kBlock + threadIdx.x;
        if (slot < n && rowid[slot] >= 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s] += r_tile[S * threadIdx.x + s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const double tsum = block_sum(acc[s], lds);
        if (threadIdx.x == 0 && state[s] != 2) partials[(size_t)s * gridDim.x + blockIdx.x] = tsum;
    }
}
// beta = rho / rho_prev * alpha / omega ; p = r + beta (p - omega nu)             (:266-267)
template <int S>
__global__ __launch_bounds__(kBlock) void gsx_p_k(double *__restrict__ scal, int rho_prev_idx, int rho_idx, const double *__restrict__ r,
                                                  const double *__restrict__ nu, double *__restrict__ p, int64_t n, int guard,
                                                  const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[S * 16];
    double beta[S], omega[S], rho[S];
    bool act[S];
    fold_partials_multi<S>(fold, fold_count, lds, rho);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double *sc = scal + GX_STRIDE * s;
        const bool frz = gx_frozen(sc, guard);
        const double rho_prev = sc[rho_prev_idx];
        const double alpha = rho_prev / sc[GX_SUM_NU];
        omega[s] = sc[GX_TS] / sc[GX_TT];
        beta[s] = rho[s] / rho_prev * alpha / omega[s];
        const bool bad = guard && !(fin_nz(omega[s]) && isfinite(beta[s]));
        act[s] = !frz && !bad;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (!frz) {
                sc[rho_idx] = rho[s];
                if (bad) sc[GX_FROZEN] = 1.;
            } else if (sc[GX_FROZEN2] != 0.) {
                sc[GX_FROZEN] = 1.;  // the x / r update took x = h, r = s: promote, or the next one would add alpha p^ again (see bicg_p_k)
            }
        }
    }
    const int64_t total = (int64_t)S * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int s = (int)(e % S);
        double bt = beta[0], o = omega[0];
        bool on = act[0];
#pragma unroll
        for (int q = 1; q < S; ++q) { if (s == q) { bt = beta[q]; o = omega[q]; on = act[q]; } }
        if (on) p[e] = r[e] + bt * (p[e] - o * nu[e]);
    }
}
template <int S>
__global__ void gsx_guard_event_k(const double *__restrict__ scal, int *__restrict__ counter) {
    int c = 0;
    for (int s = 0; s < S; ++s)
        if (scal[GX_STRIDE * s + GX_FROZEN] != 0. || scal[GX_STRIDE * s + GX_FROZEN2] != 0.) ++c;
    if (c) atomicAdd(counter, c);
}

static inline int gsx_grid(int64_t work_groups) {  // <= 1024 workgroups: every consumer folds the partial sums of its producer
    int64_t g = std::min<int64_t>(work_groups, 1024);
    if (g >= 8) g = (g / 8) * 8;
    return clamp_partials_grid(g);
}

// linear_algebra.rs:247-269 right-preconditioned by one coloured Gauss-Seidel sweep (extension), S systems on one pattern.
// M: colour-sorted storage with this solve's values; b[s], x[s]: the systems' vectors in ROW order (x in / out).
template <int S>
static int gsx_bicgstab(const SlotMat<S> &M, const std::vector<int> &color_slice, const CVecs<S> &b, const MVecs<S> &x, uint64_t iteration_count, Arena &arena,
                        int *status) {
    hipStream_t st = ctx().stream;
    const int64_t ns = M.n_slots;
    const size_t len = (size_t)S * (size_t)std::max<int64_t>(ns, 1);
    double *bs, *xs, *r, *p, *nu, *sv, *t, *ph, *sh, *partials, *partials2, *scal;
    ORC_TRY(arena.alloc(len, &bs)); ORC_TRY(arena.alloc(len, &xs)); ORC_TRY(arena.alloc(len, &r)); ORC_TRY(arena.alloc(len, &p));
    ORC_TRY(arena.alloc(len, &nu)); ORC_TRY(arena.alloc(len, &sv)); ORC_TRY(arena.alloc(len, &t)); ORC_TRY(arena.alloc(len, &ph));
    ORC_TRY(arena.alloc(len, &sh));
    ORC_TRY(arena.alloc((size_t)2 * S * kMaxPartials, &partials));
    ORC_TRY(arena.alloc((size_t)S * kMaxPartials, &partials2));
    ORC_TRY(arena.alloc((size_t)S * GX_STRIDE, &scal));
    ORC_HIP(hipMemsetAsync(scal, 0, S * GX_STRIDE * sizeof(double), st));
    const int guard = ctx().breakdown_guard ? 1 : 0;
    const int vg = gsx_grid((ns + kBlock - 1) / kBlock), g = gsx_grid(((int64_t)M.n_slices + 3) / 4);
    CVecs<S> xc;
    for (int s = 0; s < S; ++s) xc.p[s] = x.p[s];
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_in_k<S>), dim3(vg), dim3(kBlock), 0, st, M.rowid, ns, b, bs);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_in_k<S>), dim3(vg), dim3(kBlock), 0, st, M.rowid, ns, xc, xs);
    auto sweep = [&](const double *rhs, double *out) {
        for (size_t c = 0; c + 1 < color_slice.size(); ++c) {
            const int lo = color_slice[c], hi = color_slice[c + 1];
            if (hi <= lo) continue;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_sweep0_k<S>), dim3(std::min(kMaxGrid, (hi - lo + 3) / 4)), dim3(kBlock), 0, st, M, rhs, out, lo, hi, status);
        }
    };
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_spmv_k<S, 0>), dim3(g), dim3(kBlock), 0, st, M, (const double *)xs, (const double *)bs, r, p, partials);  // r = b - A x ; p = r
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_rho0_k<S>), dim3(1), dim3(kBlock), 0, st, scal, (const double *)partials, g);
    for (uint64_t it = 0; it < iteration_count; ++it) {
        const int cur = (int)(it & 1), nxt = cur ^ 1;
        sweep(p, ph);                                                                                                               // p^ = M^-1 p
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_spmv_k<S, 1>), dim3(g), dim3(kBlock), 0, st, M, (const double *)ph, (const double *)nullptr, nu, (double *)nullptr, partials);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_s_k<S>), dim3(vg), dim3(kBlock), 0, st, scal, GX_RHO0 + cur, (const double *)r, (const double *)nu, sv, ns, guard,
                           (const double *)partials, g);
        sweep(sv, sh);                                                                                                              // s^ = M^-1 s
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_spmv_k<S, 2>), dim3(g), dim3(kBlock), 0, st, M, (const double *)sh, (const double *)sv, t, (double *)nullptr, partials);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_xr_k<S>), dim3(vg), dim3(kBlock), 0, st, scal, GX_RHO0 + cur, xs, (const double *)ph, (const double *)sh, (const double *)sv,
                           (const double *)t, r, M.rowid, ns, partials2, guard, (const double *)partials, g);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_p_k<S>), dim3(vg), dim3(kBlock), 0, st, scal, GX_RHO0 + cur, GX_RHO0 + nxt, (const double *)r, (const double *)nu, p, ns, guard,
                           (const double *)partials2, vg);
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_out_k<S>), dim3(vg), dim3(kBlock), 0, st, M.rowid, ns, (const double *)xs, x);
    ORC_HIP(hipGetLastError());
    if (guard && ctx().guard_events) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_guard_event_k<S>), dim3(1), dim3(1), 0, st, (const double *)scal, ctx().guard_events);
        ORC_HIP(hipGetLastError());
    }
    return ORC_OK;
}

// is the slot-space solver used?  (ORC_GS_SLOTSPACE=0: r03's row-space recurrences; read per solve: the tests compare the two)
static inline bool gsx_enabled() { return cfg().gs_slotspace; }

// The u, v, w momentum systems of a SIMPLE iteration (solver.rs:99-136) under ORC_SOLVER_BICGSTAB_GS_PRECOND, in lock-step: the views
// share the mesh pattern (persistent: its colouring and colour-sorted layout are cached); b[k], x[k] in row order.  Single GPU.
int gs_bicgstab3_dev(const MatView A[3], const double *const b[3], double *const x[3], uint64_t iteration_count, Arena &arena) {
    const int64_t n = A[0].P.n;
    if (n == 0) return ORC_OK;
    if (!A[0].persistent_pattern || A[0].halo) return set_error(ORC_ERR_BAD_ARGUMENT, "gs_bicgstab3_dev: mesh-pattern systems on one GPU only");
    hipStream_t st = ctx().stream;
    std::unique_ptr<Coloring> owned;
    const Coloring *C = nullptr;
    ArenaScope scope(arena);
    ORC_TRY(get_coloring(A[0], owned, &C, &arena));
    if (!C->sorted.built) return set_error(ORC_ERR_BAD_ARGUMENT, "gs_bicgstab3_dev: no colour-sorted layout (ORC_GS_SORTED=0?)");
    int *status;
    ORC_TRY(arena.alloc((size_t)1, &status));
    ORC_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    SlotMat<3> M;
    M.sp = C->sorted.slice_ptr.p; M.row_len = C->sorted.row_len.p; M.rowid = C->sorted.rowid.p; M.diag_off = C->sorted.diag_off.p; M.cslot = C->sorted.cslot.p;
    M.n_slices = C->sorted.n_slices; M.n_slots = C->sorted.n_slots;
    CVecs<3> bv;
    MVecs<3> xv;
    for (int k = 0; k < 3; ++k) {
        double *vals;
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(C->sorted.padded, 1), &vals));
        hipLaunchKernelGGL(gs_permute_values_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A[k], C->sorted.slot_of_row.p, C->sorted.slice_ptr.p, vals);
        M.val[k] = vals; bv.p[k] = b[k]; xv.p[k] = x[k];
    }
    ORC_HIP(hipGetLastError());
    ORC_TRY(gsx_bicgstab<3>(M, C->sorted.color_slice, bv, xv, iteration_count, arena, status));
    int h = 0;
    ORC_HIP(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    return h == ORC_ERR_STRUCTURAL_ZERO ? h : ORC_OK;
}
bool gs_slot_space_enabled() { return gsx_enabled(); }

int gs_arm_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor, int method, Arena &arena) {
    const int64_t n = A.P.n;
    if (n == 0) return ORC_OK;
    // Partitioned operator: every rank colours and sweeps its own rows; the ghost entries of the swept vector are refreshed
    // once per sweep, so rows along a cut see their remote neighbours as of the previous sweep (processor-block
    // Gauss-Seidel: Gauss-Seidel inside a rank, Jacobi across the cuts).  Reductions are summed over the ranks.
    const bool global = A.halo != nullptr && ctx().world > 1;
    hipStream_t st = ctx().stream;
    std::unique_ptr<Coloring> owned;
    const Coloring *C = nullptr;
    Arena::Mark mk = arena.mark();
    ORC_TRY(get_coloring(A, owned, &C, &arena));
    int *status;
    ORC_TRY(arena.alloc((size_t)1, &status));
    ORC_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    // the matrix in colour-sorted storage: cached layout + this solve's values, or all of it built now (coarse levels)
    constexpr bool sorted_enabled = true;  // (colour-sorted storage; r02's row-list kernel gs_color_k serves what has no sorted image)
    SortedView view;
    if (C->sorted.built) {
        double *vals;
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(C->sorted.padded, 1), &vals));
        hipLaunchKernelGGL(gs_permute_values_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A, C->sorted.slot_of_row.p, C->sorted.slice_ptr.p, vals);
        ORC_HIP(hipGetLastError());
        view.sp = C->sorted.slice_ptr.p; view.row_len = C->sorted.row_len.p; view.rowid = C->sorted.rowid.p; view.diag_off = C->sorted.diag_off.p;
        view.col = C->sorted.col.p; view.val = vals; view.color_slice = C->sorted.color_slice;
        view.cslot = C->sorted.cslot.p; view.n_slices = C->sorted.n_slices;
    } else if (sorted_enabled && !A.persistent_pattern) {
        ORC_TRY(build_sorted_on_device(A, *C, arena, view));
    }
    const SortedView *sv = &view;
    if (method == ORC_SOLVER_MULTICOLOR_GS) {
        for (uint64_t it = 0; it < iteration_count; ++it) {
            if (A.halo) ORC_TRY(A.halo->exchange(x));
            ORC_TRY(gs_sweep(A, *C, b, x, relaxation_factor, status, sv));
        }
    } else if (!global && sv->ok() && sv->cslot && gsx_enabled()) {  // [r04] the same recurrences in slot space (single GPU)
        SlotMat<1> M;
        M.sp = sv->sp; M.row_len = sv->row_len; M.rowid = sv->rowid; M.diag_off = sv->diag_off; M.cslot = sv->cslot; M.val[0] = sv->val;
        M.n_slices = sv->n_slices; M.n_slots = (int64_t)sv->n_slices * 64;
        CVecs<1> bv;
        MVecs<1> xv;
        bv.p[0] = b; xv.p[0] = x;
        ORC_TRY(gsx_bicgstab<1>(M, sv->color_slice, bv, xv, iteration_count, arena, status));
    } else {  // ORC_SOLVER_BICGSTAB_GS_PRECOND: linear_algebra.rs:247-269 with p^ = M^-1 p, s^ = M^-1 s, M^-1 = one GS sweep from 0
        const size_t nn = (size_t)std::max(A.P.ncols, n);  // products gather ghost entries of p^ and s^
        double *r, *p, *nu, *s, *t, *ph, *sh, *partials, *scal;
        ORC_TRY(arena.alloc(nn, &r)); ORC_TRY(arena.alloc(nn, &p)); ORC_TRY(arena.alloc(nn, &nu)); ORC_TRY(arena.alloc(nn, &s));
        ORC_TRY(arena.alloc(nn, &t)); ORC_TRY(arena.alloc(nn, &ph)); ORC_TRY(arena.alloc(nn, &sh));
        ORC_TRY(arena.alloc((size_t)2 * kMaxPartials, &partials));
        ORC_TRY(arena.alloc((size_t)8, &scal));
        enum { RHO0 = 0, RHO1 = 1, SUM_NU = 2, TS = 3, TT = 4 };
        ORC_HIP(hipMemsetAsync(scal, 0, 8 * sizeof(double), st));
        const int guard = ctx().breakdown_guard ? 1 : 0;
        const double *skip = guard ? scal + 5 : nullptr;
        const int vg = grid_for(n);
        int g = 0;
        ORC_TRY(spmv_launch(A, x, EpiRes{b, r, p}, partials, &g));
        ORC_TRY(reduce_partials(partials, g, 1, scal + RHO0, global));
        for (uint64_t it = 0; it < iteration_count; ++it) {
            const int cur = (int)(it & 1), nxt = cur ^ 1;
            ORC_TRY(vec_fill(ph, 0., (int64_t)nn));  // ghost entries too: the preconditioner is rank-local
            ORC_TRY(gs_sweep(A, *C, p, ph, 1.0, status, sv));
            ORC_TRY(spmv_launch(A, ph, EpiSum{nu}, partials, &g, skip));
            ORC_TRY(reduce_partials(partials, g, 1, scal + SUM_NU, global));
            hipLaunchKernelGGL(bicg_s_pre_k, dim3(vg), dim3(kBlock), 0, st, scal, RHO0 + cur, SUM_NU, r, nu, s, n, guard);
            ORC_TRY(vec_fill(sh, 0., (int64_t)nn));
            ORC_TRY(gs_sweep(A, *C, s, sh, 1.0, status, sv));
            ORC_TRY(spmv_launch(A, sh, EpiTsPre{s, t}, partials, &g, skip));
            ORC_TRY(reduce_partials(partials, g, 2, scal + TS, global));
            hipLaunchKernelGGL(bicg_xr_pre_k, dim3(vg), dim3(kBlock), 0, st, scal, RHO0 + cur, SUM_NU, TS, TT, x, ph, sh, s, t, r, n, partials, guard);
            ORC_TRY(reduce_partials(partials, vg, 1, scal + RHO0 + nxt, global));
            hipLaunchKernelGGL(bicg_p_pre_k, dim3(vg), dim3(kBlock), 0, st, scal, RHO0 + cur, RHO0 + nxt, SUM_NU, TS, TT, r, nu, p, n, guard);
            ORC_HIP(hipGetLastError());
        }
    }
    int h = 0;
    ORC_HIP(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    arena.release(mk);
    return method == ORC_SOLVER_MULTICOLOR_GS ? h : (h == ORC_ERR_STRUCTURAL_ZERO ? h : ORC_OK);
}

// bench.py (BASELINE configs[2]): one multicolour sweep = C->n_colors launches of gs_color_sorted_k over the colour-sorted matrix,
// `reps` sweeps between two HIP events on the library stream; x = M^-1 b from zero, as the preconditioner application does it.
int bench_gs_sweep_dev(const MatView &A, const double *b, double *x, int reps, Arena &arena, float *ms_per_sweep, int *n_colors) {
    const int64_t n = A.P.n;
    if (n == 0) return ORC_OK;
    hipStream_t st = ctx().stream;
    std::unique_ptr<Coloring> owned;
    const Coloring *C = nullptr;
    ArenaScope scope(arena);
    ORC_TRY(get_coloring(A, owned, &C, &arena));
    int *status;
    ORC_TRY(arena.alloc((size_t)1, &status));
    ORC_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    SortedView view;
    if (C->sorted.built) {
        double *vals;
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(C->sorted.padded, 1), &vals));
        hipLaunchKernelGGL(gs_permute_values_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A, C->sorted.slot_of_row.p, C->sorted.slice_ptr.p, vals);
        ORC_HIP(hipGetLastError());
        view.sp = C->sorted.slice_ptr.p; view.row_len = C->sorted.row_len.p; view.rowid = C->sorted.rowid.p; view.diag_off = C->sorted.diag_off.p;
        view.col = C->sorted.col.p; view.val = vals; view.color_slice = C->sorted.color_slice;
    } else if (!A.persistent_pattern) {
        ORC_TRY(build_sorted_on_device(A, *C, arena, view));
    }
    if (n_colors) *n_colors = C->n_colors;
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    ORC_TRY(vec_fill(x, 0., std::max(A.P.ncols, n)));
    int rc = gs_sweep(A, *C, b, x, 1.0, status, &view);  // warm
    if (rc == ORC_OK && hipEventRecord(e0, st) != hipSuccess) rc = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    for (int i = 0; i < reps && rc == ORC_OK; ++i) rc = gs_sweep(A, *C, b, x, 1.0, status, &view);
    if (rc == ORC_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    if (rc == ORC_OK && hipEventElapsedTime(ms_per_sweep, e0, e1) != hipSuccess) rc = set_error(ORC_ERR_HIP, "hipEventElapsedTime failed");
    if (rc == ORC_OK) *ms_per_sweep /= (float)std::max(reps, 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// [r04] the preconditioner application as the slot-space solver launches it: x^ = M^-1 b from zero (gsx_sweep0_k, n_colors launches, no
// zero fill), one system (ms[0]) and u, v, w per launch (ms[1]); A[k]: the three momentum views on the mesh pattern
int bench_gs_sweep0_dev(const MatView A[3], const double *const b[3], int reps, Arena &arena, float ms[2], int *n_colors) {
    const int64_t n = A[0].P.n;
    ms[0] = ms[1] = 0.f;
    if (n == 0) return ORC_OK;
    hipStream_t st = ctx().stream;
    std::unique_ptr<Coloring> owned;
    const Coloring *C = nullptr;
    ArenaScope scope(arena);
    ORC_TRY(get_coloring(A[0], owned, &C, &arena));
    if (!C->sorted.built) return set_error(ORC_ERR_BAD_ARGUMENT, "no colour-sorted layout");
    if (n_colors) *n_colors = C->n_colors;
    int *status;
    ORC_TRY(arena.alloc((size_t)1, &status));
    ORC_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    SlotMat<3> M3;
    M3.sp = C->sorted.slice_ptr.p; M3.row_len = C->sorted.row_len.p; M3.rowid = C->sorted.rowid.p; M3.diag_off = C->sorted.diag_off.p; M3.cslot = C->sorted.cslot.p;
    M3.n_slices = C->sorted.n_slices; M3.n_slots = C->sorted.n_slots;
    CVecs<3> bv;
    for (int k = 0; k < 3; ++k) {
        double *vals;
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(C->sorted.padded, 1), &vals));
        hipLaunchKernelGGL(gs_permute_values_k, dim3(grid_for(n)), dim3(kBlock), 0, st, A[k], C->sorted.slot_of_row.p, C->sorted.slice_ptr.p, vals);
        M3.val[k] = vals; bv.p[k] = b[k];
    }
    SlotMat<1> M1;
    M1.sp = M3.sp; M1.row_len = M3.row_len; M1.rowid = M3.rowid; M1.diag_off = M3.diag_off; M1.cslot = M3.cslot; M1.val[0] = M3.val[0];
    M1.n_slices = M3.n_slices; M1.n_slots = M3.n_slots;
    CVecs<1> b1;
    b1.p[0] = b[0];
    const int64_t ns = M3.n_slots;
    double *bs3, *xh3, *bs1, *xh1;
    ORC_TRY(arena.alloc((size_t)3 * ns, &bs3)); ORC_TRY(arena.alloc((size_t)3 * ns, &xh3));
    ORC_TRY(arena.alloc((size_t)ns, &bs1)); ORC_TRY(arena.alloc((size_t)ns, &xh1));
    const int vg = gsx_grid((ns + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_in_k<3>), dim3(vg), dim3(kBlock), 0, st, M3.rowid, ns, bv, bs3);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_in_k<1>), dim3(vg), dim3(kBlock), 0, st, M1.rowid, ns, b1, bs1);
    ORC_HIP(hipGetLastError());
    const std::vector<int> &cs = C->sorted.color_slice;
    auto timed = [&](auto &&sweep, float *out) -> int {
        hipEvent_t e0, e1;
        ORC_HIP(hipEventCreate(&e0));
        ORC_HIP(hipEventCreate(&e1));
        sweep();  // warm
        int rc = hipEventRecord(e0, st) == hipSuccess ? ORC_OK : set_error(ORC_ERR_HIP, "hipEventRecord failed");
        for (int i = 0; i < reps && rc == ORC_OK; ++i) sweep();
        if (rc == ORC_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = set_error(ORC_ERR_HIP, "hipEventRecord failed");
        if (rc == ORC_OK && hipEventElapsedTime(out, e0, e1) != hipSuccess) rc = set_error(ORC_ERR_HIP, "hipEventElapsedTime failed");
        if (rc == ORC_OK) *out /= (float)std::max(reps, 1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    };
    ORC_TRY(timed([&] {
        for (size_t c = 0; c + 1 < cs.size(); ++c)
            if (cs[c + 1] > cs[c])
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_sweep0_k<1>), dim3(std::min(kMaxGrid, (cs[c + 1] - cs[c] + 3) / 4)), dim3(kBlock), 0, st, M1, (const double *)bs1, xh1, cs[c], cs[c + 1], status);
    }, &ms[0]));
    ORC_TRY(timed([&] {
        for (size_t c = 0; c + 1 < cs.size(); ++c)
            if (cs[c + 1] > cs[c])
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_sweep0_k<3>), dim3(std::min(kMaxGrid, (cs[c + 1] - cs[c] + 3) / 4)), dim3(kBlock), 0, st, M3, (const double *)bs3, xh3, cs[c], cs[c + 1], status);
    }, &ms[1]));
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// test hook: the colouring of a pattern (host arrays out)
int gs_debug_coloring(const SellDev &P, std::vector<int> &colors, int *n_colors) {
    Coloring C;
    ORC_TRY(build_coloring(P, C));
    colors.resize((size_t)P.n);
    ORC_HIP(hipMemcpy(colors.data(), C.color_p, sizeof(int) * (size_t)P.n, hipMemcpyDeviceToHost));
    *n_colors = C.n_colors;
    return ORC_OK;
}

}  // namespace orc
