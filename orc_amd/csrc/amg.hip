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
#include <cmath>

#include "linalg_kernels.hpp"

namespace orc {

int comm_allreduce_sum(double *dev, int n);

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ int64_t sell_pos(const SellDev &P, int64_t row, int k) { return P.slice_ptr[row >> 6] + (int64_t)k * 64 + (row & 63); }

struct AggCounters {
    int cur;      // rows to evaluate in this round
    int changed;  // rows whose choice changed in this round
    int next;     // rows activated for the next round
    int rounds;
};

// arg-min over j != i of a_ij among columns not taken by an earlier row (strict <, first wins: :37-52).
// `constrained` = 0 evaluates the unconstrained arg-min (round 0).
__device__ __forceinline__ int agg_eval_row(const MatView &A, const int *__restrict__ choice, int64_t i, int constrained) {
    const int len = A.P.row_len[i];
    double best = 1.7976931348623157e308;  // Float::MAX
    int bj = -1;
    for (int k = 0; k < len; ++k) {
        const int64_t pos = sell_pos(A.P, i, k);
        const int j = A.P.col[pos];
        if (j == i) continue;
        if (constrained) {
            // j in combined_cells when row i is visited  <=>  some row m < i chose j.  Rows that can
            // choose j hold j in their pattern = (structural symmetry) the columns of row j.
            bool taken = false;
            const int lj = A.P.row_len[j];
            for (int kk = 0; kk < lj; ++kk) {
                const int m = A.P.col[sell_pos(A.P, j, kk)];
                if (m < i && choice[m] == j) { taken = true; break; }
            }
            if (taken) continue;
        }
        const double a = view_value(A, i, pos);
        if (a < best) { best = a; bj = j; }
    }
    return bj;
}

__global__ void agg_init_k(MatView A, int *__restrict__ choice, int *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x) {
        choice[i] = agg_eval_row(A, choice, i, 0);
        flag[i] = 0;
    }
}

// evaluate the rows of the work list (list == nullptr: all rows) against the committed choices
__global__ void agg_eval_k(MatView A, const int *__restrict__ choice, const int *__restrict__ list, AggCounters *C,
                           int *__restrict__ flag, int *__restrict__ changed_rows, int *__restrict__ changed_vals) {
    const int count = C->cur;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const int i = list ? list[idx] : idx;
        flag[i] = 0;
        const int nv = agg_eval_row(A, choice, i, 1);
        if (nv != choice[i]) {
            const int slot = atomicAdd(&C->changed, 1);
            changed_rows[slot] = i;
            changed_vals[slot] = nv;
        }
    }
}

// commit the changes and activate every later row that can see them
__global__ void agg_commit_k(MatView A, int *__restrict__ choice, AggCounters *C, int *__restrict__ flag,
                             const int *__restrict__ changed_rows, const int *__restrict__ changed_vals, int *__restrict__ next_list) {
    const int count = C->changed;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const int i = changed_rows[idx];
        const int old = choice[i], nv = changed_vals[idx];
        choice[i] = nv;
        const int js[2] = {old, nv};
        for (int t = 0; t < 2; ++t) {
            const int j = js[t];
            if (j < 0) continue;
            const int lj = A.P.row_len[j];
            for (int kk = 0; kk < lj; ++kk) {
                const int m = A.P.col[sell_pos(A.P, j, kk)];
                if (m > i && atomicExch(&flag[m], 1) == 0) next_list[atomicAdd(&C->next, 1)] = m;
            }
        }
    }
}

__global__ void agg_rotate_k(AggCounters *C) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        C->cur = C->next;
        C->next = 0;
        C->changed = 0;
        C->rounds += 1;
    }
}

// General (structurally asymmetric) fall-back: full passes; "taken before row i" via the smallest
// chooser of each column, rebuilt every round with atomicMin.
__global__ void agg_fc_reset_k(int *__restrict__ fc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) fc[i] = 0x7fffffff;
}
__global__ void agg_fc_scatter_k(const int *__restrict__ choice, int *__restrict__ fc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (choice[i] >= 0) atomicMin(&fc[choice[i]], (int)i);
}
__global__ void agg_fc_eval_k(MatView A, const int *__restrict__ choice, const int *__restrict__ fc, int *__restrict__ choice_new,
                              AggCounters *C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.P.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int len = A.P.row_len[i];
        double best = 1.7976931348623157e308;
        int bj = -1;
        for (int k = 0; k < len; ++k) {
            const int64_t pos = sell_pos(A.P, i, k);
            const int j = A.P.col[pos];
            if (j == i || fc[j] < i) continue;
            const double a = view_value(A, i, pos);
            if (a < best) { best = a; bj = j; }
        }
        choice_new[i] = bj;
        if (bj != choice[i]) atomicAdd(&C->changed, 1);
    }
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
__device__ __forceinline__ RRow restriction_row(const int *__restrict__ choice, int64_t I, int64_t n_fine) {
    RRow r;
    r.n = 0;
    for (int t = 0; t < 2; ++t) {
        const int64_t i = 2 * I + t;
        if (i >= n_fine) break;
        const int c = choice[i];
        if (c < 0) continue;
        r.idx[r.n] = (int)i; r.w[r.n] = 1.; r.n++;
        r.idx[r.n] = c; r.w[r.n] = 1.; r.n++;
    }
    // insertion sort (<= 4), then merge equal indices
    for (int a = 1; a < r.n; ++a) {
        const int key = r.idx[a];
        int b = a - 1;
        while (b >= 0 && r.idx[b] > key) { r.idx[b + 1] = r.idx[b]; --b; }
        r.idx[b + 1] = key;
    }
    int m = 0;
    for (int a = 0; a < r.n; ++a) {
        if (m > 0 && r.idx[m - 1] == r.idx[a]) r.w[m - 1] += 1.;
        else { r.idx[m] = r.idx[a]; r.w[m] = 1.; m++; }
    }
    r.n = m;
    return r;
}

// r' = R r (:82)
__global__ void restrict_k(const int *__restrict__ choice, int64_t n_fine, int64_t n_coarse, const double *__restrict__ r, double *__restrict__ rc) {
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < n_coarse; I += (int64_t)gridDim.x * blockDim.x) {
        const RRow R = restriction_row(choice, I, n_fine);
        double acc = 0.;
        for (int a = 0; a < R.n; ++a) acc += R.w[a] * r[R.idx[a]];
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

// ------------------------------------------------------------------ Galerkin product (R A) R^T per coarse row
// Sorted insert-accumulate into an LDS list laid out [slot][thread] (conflict-free when lanes
// touch the same slot).  Returns false when the list is full.
__device__ __forceinline__ bool list_add(int *__restrict__ keys, double *__restrict__ vals, int &len, int cap, int tpb, int key, double v) {
    const int t = threadIdx.x;
    int pos = len;
    while (pos > 0 && keys[(pos - 1) * tpb + t] > key) --pos;
    if (pos > 0 && keys[(pos - 1) * tpb + t] == key) {
        vals[(pos - 1) * tpb + t] += v;
        return true;
    }
    if (len >= cap) return false;
    for (int q = len; q > pos; --q) {
        keys[q * tpb + t] = keys[(q - 1) * tpb + t];
        vals[q * tpb + t] = vals[(q - 1) * tpb + t];
    }
    keys[pos * tpb + t] = key;
    vals[pos * tpb + t] = 0. + v;
    ++len;
    return true;
}

// NUMERIC = false: row lengths only; true: write columns / values / diagonal offsets of the coarse SELL matrix.
template <bool NUMERIC>
__global__ void galerkin_k(MatView A, const int *__restrict__ choice, const int *__restrict__ chooser, int64_t n_coarse, int capT,
                           int capO, int *__restrict__ row_len_c, SellDev Pc, int *__restrict__ col_c, double *__restrict__ val_c,
                           int *__restrict__ diag_c, int *__restrict__ overflow) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tpb = blockDim.x;
    double *Tval = reinterpret_cast<double *>(smem);
    double *Oval = Tval + (size_t)capT * tpb;
    int *Tkey = reinterpret_cast<int *>(Oval + (size_t)capO * tpb);
    int *Okey = Tkey + (size_t)capT * tpb;
    const int t = threadIdx.x;
    for (int64_t I0 = (int64_t)blockIdx.x * tpb; I0 < n_coarse; I0 += (int64_t)gridDim.x * tpb) {
        const int64_t I = I0 + t;
        if (I >= n_coarse) continue;
        const RRow R = restriction_row(choice, I, A.P.n);
        int lenT = 0, lenO = 0;
        bool ok = true;
        // T = (R A)[I, :]   — i ascending, then A's row order (spmm_csr: c_Ij += R_Ii * a_ij)
        for (int a = 0; a < R.n && ok; ++a) {
            const int i = R.idx[a];
            const double w = R.w[a];
            const int len = A.P.row_len[i];
            for (int k = 0; k < len; ++k) {
                const int64_t pos = sell_pos(A.P, i, k);
                ok = list_add(Tkey, Tval, lenT, capT, tpb, A.P.col[pos], w * view_value(A, i, pos));
                if (!ok) break;
            }
        }
        // A'[I, :] = T R^T  — j ascending, then R^T's row order (c_IJ += T_Ij * R^T_jJ)
        for (int q = 0; q < lenT && ok; ++q) {
            const int j = Tkey[q * tpb + t];
            const double tv = Tval[q * tpb + t];
            int J[2];
            double W[2];
            const int nj = rt_row(choice, chooser, j, J, W);
            for (int a = 0; a < nj; ++a) {
                ok = list_add(Okey, Oval, lenO, capO, tpb, J[a], tv * W[a]);
                if (!ok) break;
            }
        }
        if (!ok) { atomicExch(overflow, 1); continue; }
        if (!NUMERIC) {
            row_len_c[I] = lenO;
        } else {
            const int64_t base = Pc.slice_ptr[I >> 6] + (I & 63);
            const int width = (int)((Pc.slice_ptr[(I >> 6) + 1] - Pc.slice_ptr[I >> 6]) >> 6);
            int d = -1;
            for (int q = 0; q < width; ++q) {
                const int64_t pos = base + (int64_t)q * 64;
                if (q < lenO) {
                    const int J = Okey[q * tpb + t];
                    col_c[pos] = J;
                    val_c[pos] = Oval[q * tpb + t];
                    if (J == I) d = (int)pos;
                } else {
                    col_c[pos] = (int)I;
                    val_c[pos] = 0.;
                }
            }
            diag_c[I] = d;
        }
    }
}

// capacity bound for the T list: sum of the lengths of the (<= 4) fine rows of each coarse row
__global__ __launch_bounds__(kBlock) void galerkin_bound_k(SellDev P, const int *__restrict__ choice, int64_t n_coarse, int *__restrict__ out_max) {
    __shared__ double lds[8];
    double mx = 0.;
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < n_coarse; I += (int64_t)gridDim.x * blockDim.x) {
        const RRow R = restriction_row(choice, I, P.n);
        int s = 0;
        for (int a = 0; a < R.n; ++a) s += P.row_len[R.idx[a]];
        mx = fmax(mx, (double)s);
    }
    const double m = block_max(mx, lds);
    if (threadIdx.x == 0) atomicMax(out_max, (int)m);
}

// slice widths -> slice_ptr (single workgroup scan; n_slices is n/64)
__global__ __launch_bounds__(1024) void slice_ptr_k(const int *__restrict__ row_len, int64_t n, int n_slices, int64_t *__restrict__ slice_ptr) {
    __shared__ long long carry;
    __shared__ long long buf[1024];
    if (threadIdx.x == 0) { carry = 0; slice_ptr[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n_slices; base += 1024) {
        const int s = base + threadIdx.x;
        long long w = 0;
        if (s < n_slices) {
            const int64_t lo = (int64_t)s * 64, hi = lo + 64 < n ? lo + 64 : n;
            int mx = 0;
            for (int64_t r = lo; r < hi; ++r) mx = max(mx, row_len[r]);
            w = (long long)mx * 64;
        }
        buf[threadIdx.x] = w;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            long long v = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += v;
            __syncthreads();
        }
        if (s < n_slices) slice_ptr[s + 1] = carry + buf[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
}

__global__ void nan_to_status_k(const double *__restrict__ value, int *status, int code) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && isnan(value[0])) atomicCAS(status, 0, code);
}

// ------------------------------------------------------------------ host drivers
struct CoarseLevel {
    SellDev P;
    double *val = nullptr;
    int64_t n = 0, padded = 0;
    int *choice = nullptr, *chooser = nullptr;  // of the FINE level this was built from
    int rounds = 0;
};

static int aggregate(const MatView &A, Arena &arena, int *choice, int *chooser, int *rounds_out) {
    const int64_t n = A.P.n;
    const int g = grid_for(n);
    int *flag, *listA, *listB, *changed_rows, *changed_vals;
    AggCounters *C;
    ORC_TRY(arena.alloc((size_t)n, &flag));
    ORC_TRY(arena.alloc((size_t)n, &listA));
    ORC_TRY(arena.alloc((size_t)n, &listB));
    ORC_TRY(arena.alloc((size_t)n, &changed_rows));
    ORC_TRY(arena.alloc((size_t)n, &changed_vals));
    ORC_TRY(arena.alloc((size_t)1, &C));
    hipStream_t st = ctx().stream;
    hipLaunchKernelGGL(agg_init_k, dim3(g), dim3(kBlock), 0, st, A, choice, flag);
    AggCounters h{(int)n, 0, 0, 0};
    ORC_HIP(hipMemcpyAsync(C, &h, sizeof(h), hipMemcpyHostToDevice, st));
    int rounds = 0;
    if (A.symmetric) {
        // round 1: every row; later rounds: work lists
        const int *cur_list = nullptr;
        int *next_list = listA;
        bool first = true;
        while (true) {
            const int batch = first ? 1 : 16;
            for (int b = 0; b < batch; ++b) {
                const int ge = first ? g : 512;
                hipLaunchKernelGGL(agg_eval_k, dim3(ge), dim3(kBlock), 0, st, A, choice, cur_list, C, flag, changed_rows, changed_vals);
                hipLaunchKernelGGL(agg_commit_k, dim3(ge), dim3(kBlock), 0, st, A, choice, C, flag, changed_rows, changed_vals, next_list);
                hipLaunchKernelGGL(agg_rotate_k, dim3(1), dim3(1), 0, st, C);
                cur_list = next_list;
                next_list = (next_list == listA) ? listB : listA;
                first = false;
            }
            ORC_HIP(hipGetLastError());
            ORC_HIP(hipMemcpyAsync(&h, C, sizeof(h), hipMemcpyDeviceToHost, st));
            ORC_HIP(hipStreamSynchronize(st));
            rounds = h.rounds;
            if (h.cur == 0) break;
            if (rounds > 4 * 1000 * 1000) return set_error(ORC_ERR_BAD_ARGUMENT, "aggregation did not reach its fixed point");
        }
    } else {
        int *fc = flag;           // reuse
        int *choice_new = listA;  // reuse
        int *cur = choice;
        while (true) {
            hipLaunchKernelGGL(agg_fc_reset_k, dim3(g), dim3(kBlock), 0, st, fc, n);
            hipLaunchKernelGGL(agg_fc_scatter_k, dim3(g), dim3(kBlock), 0, st, cur, fc, n);
            hipLaunchKernelGGL(agg_fc_eval_k, dim3(g), dim3(kBlock), 0, st, A, cur, fc, choice_new, C);
            ORC_HIP(hipGetLastError());
            ORC_HIP(hipMemcpyAsync(&h, C, sizeof(h), hipMemcpyDeviceToHost, st));
            ORC_HIP(hipStreamSynchronize(st));
            std::swap(cur, choice_new);
            ++rounds;
            if (h.changed == 0) break;
            h.changed = 0;
            ORC_HIP(hipMemcpyAsync(C, &h, sizeof(h), hipMemcpyHostToDevice, st));
            if (rounds > (int)n + 2) return set_error(ORC_ERR_BAD_ARGUMENT, "aggregation did not reach its fixed point");
        }
        if (cur != choice) ORC_HIP(hipMemcpyAsync(choice, cur, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    ORC_HIP(hipMemsetAsync(chooser, 0xff, sizeof(int) * (size_t)n, st));
    hipLaunchKernelGGL(chooser_k, dim3(g), dim3(kBlock), 0, st, choice, chooser, n);
    ORC_HIP(hipGetLastError());
    if (rounds_out) *rounds_out = rounds;
    return ORC_OK;
}

static int galerkin(const MatView &A, const int *choice, const int *chooser, Arena &arena, CoarseLevel &L) {
    const int64_t n = A.P.n, nc = n / 2 + n % 2;  // :13
    hipStream_t st = ctx().stream;
    int *row_len, *diag, *flags;  // flags[0] = max T length, flags[1] = overflow
    int64_t *slice_ptr;
    const int n_slices = (int)((nc + 63) / 64);
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nc, 1), &row_len));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nc, 1), &diag));
    ORC_TRY(arena.alloc((size_t)n_slices + 1, &slice_ptr));
    ORC_TRY(arena.alloc((size_t)2, &flags));
    ORC_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(galerkin_bound_k, dim3(grid_for(nc)), dim3(kBlock), 0, st, A.P, choice, nc, flags);
    int hflags[2];
    ORC_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    int capT = std::max(hflags[0], 1);
    int capO = 2 * capT;
    if ((int64_t)capO > nc) capO = (int)std::max<int64_t>(nc, 1);
    int tpb = 64;
    const size_t budget = 150 * 1024;
    while (tpb > 1 && (size_t)tpb * (capT + capO) * 12 > budget) tpb >>= 1;
    if ((size_t)tpb * (capT + capO) * 12 > budget) return set_error(ORC_ERR_BAD_ARGUMENT, "Galerkin row too long for LDS (%d entries)", capT);
    const size_t smem = (size_t)tpb * (capT + capO) * 12;
    const int g = (int)std::min<int64_t>((nc + tpb - 1) / tpb, 4096);
    SellDev Pc;  // filled below
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&galerkin_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(galerkin_k<false>), dim3(g), dim3(tpb), smem, st, A, choice, chooser, nc, capT, capO, row_len, Pc, nullptr,
                       nullptr, nullptr, flags + 1);
    hipLaunchKernelGGL(slice_ptr_k, dim3(1), dim3(1024), 0, st, row_len, nc, n_slices, slice_ptr);
    ORC_HIP(hipGetLastError());
    int64_t padded = 0;
    ORC_HIP(hipMemcpyAsync(&padded, slice_ptr + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, st));
    ORC_HIP(hipStreamSynchronize(st));
    if (hflags[1]) return set_error(ORC_ERR_BAD_ARGUMENT, "Galerkin LDS list overflow");
    if (padded >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "coarse matrix too large for 32-bit offsets");
    int *col;
    double *val;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &col));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(padded, 1), &val));
    Pc.n = nc; Pc.n_slices = n_slices; Pc.slice_ptr = slice_ptr; Pc.row_len = row_len; Pc.col = col; Pc.diag_pos = diag;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(galerkin_k<true>), dim3(g), dim3(tpb), smem, st, A, choice, chooser, nc, capT, capO, row_len, Pc, col, val,
                       diag, flags + 1);
    ORC_HIP(hipGetLastError());
    L.P = Pc; L.val = val; L.n = nc; L.padded = padded;
    return ORC_OK;
}

struct MgParams {
    uint64_t max_levels, iters;
    int smoother, preconditioner;
    double relaxation, threshold;
};

// linear_algebra.rs:66-141.  `add_to`: the fine vector the prolonged correction is added to.
static int multigrid_solve_dev(const MatView &A, const double *r, uint64_t level, const MgParams &mp, double threshold, Arena &arena,
                               SolveStats *stats, int *dev_status, double *out, double *add_to) {
    const int64_t n = A.P.n;
    hipStream_t st = ctx().stream;
    Arena::Mark mk = arena.mark();
    int *choice, *chooser;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &choice));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &chooser));
    CoarseLevel L;
    ORC_TRY(aggregate(A, arena, choice, chooser, &L.rounds));  // :80 (scratch is released with the level)
    ORC_TRY(galerkin(A, choice, chooser, arena, L));  // :84
    const int64_t nc = L.n;
    if (stats && level < 8) {
        stats->amg_levels = std::max(stats->amg_levels, (int)level);
        stats->amg_rows[level] = nc;
        stats->amg_nnz[level] = L.padded;
        stats->amg_rounds[level] = L.rounds;
    }
    double *r_prime, *e_prime, *partials, *scal;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nc, 1), &r_prime));
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(nc, 1), &e_prime));
    ORC_TRY(arena.alloc((size_t)kMaxPartials, &partials));
    ORC_TRY(arena.alloc((size_t)4, &scal));
    hipLaunchKernelGGL(restrict_k, dim3(grid_for(nc)), dim3(kBlock), 0, st, choice, n, nc, r, r_prime);  // :82
    ORC_HIP(hipGetLastError());
    ORC_TRY(vec_fill(e_prime, 0., nc));  // :86
    MatView Ac;
    Ac.P = L.P;
    Ac.val = L.val;
    Ac.symmetric = A.symmetric;
    int stt = iterative_solve_dev(Ac, r_prime, e_prime, mp.iters, mp.smoother, mp.relaxation, threshold, mp.preconditioner, arena, stats);  // :87-96
    if (stt != ORC_OK) { arena.release(mk); return stt; }
    // :97-105  |r' - a' e'| is NaN -> "Multigrid diverged"
    ORC_TRY(residual_norm2_dev(Ac, r_prime, e_prime, partials, scal));
    hipLaunchKernelGGL(nan_to_status_k, dim3(1), dim3(64), 0, st, scal, dev_status, (int)ORC_ERR_MULTIGRID_DIVERGED);
    if (level < mp.max_levels && nc > 16) {  // :109
        // :110-121 — the recursion receives r', not the residual
        stt = multigrid_solve_dev(Ac, r_prime, level + 1, mp, threshold, arena, stats, dev_status, nullptr, e_prime);
        if (stt != ORC_OK) { arena.release(mk); return stt; }
        stt = iterative_solve_dev(Ac, r_prime, e_prime, mp.iters, mp.smoother, mp.relaxation, threshold / 10., mp.preconditioner, arena, stats);  // :123-132
        if (stt != ORC_OK) { arena.release(mk); return stt; }
    }
    hipLaunchKernelGGL(prolong_k, dim3(grid_for(n)), dim3(kBlock), 0, st, choice, chooser, n, e_prime, out, add_to);  // :140
    ORC_HIP(hipGetLastError());
    arena.release(mk);
    return ORC_OK;
}

// Multigrid arm of iterative_solve (:270-296); A and b are already the preconditioned system.
int multigrid_arm_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor,
                      double convergence_threshold, int preconditioner, Arena &arena, SolveStats *stats, int smoother) {
    const int64_t n = A.P.n;
    if (n == 0) return ORC_OK;
    hipStream_t st = ctx().stream;
    Arena::Mark mk = arena.mark();
    // :273-282 — the smoother is called with the same preconditioner: the scaled system is scaled again (Q4)
    int stt = iterative_solve_dev(A, b, x, iteration_count, smoother, relaxation_factor, convergence_threshold, preconditioner, arena, stats);
    if (stt != ORC_OK) { arena.release(mk); return stt; }
    double *r;
    int *dev_status;
    ORC_TRY(arena.alloc((size_t)n, &r));
    ORC_TRY(arena.alloc((size_t)1, &dev_status));
    ORC_HIP(hipMemsetAsync(dev_status, 0, sizeof(int), st));
    ORC_TRY(residual_dev(A, b, x, r));  // :283
    MgParams mp{3 /* MULTIGRID_COARSENING_LEVELS, :10 */, iteration_count, smoother, preconditioner, relaxation_factor, convergence_threshold};
    stt = multigrid_solve_dev(A, r, 1, mp, convergence_threshold, arena, stats, dev_status, nullptr, x);  // :284-295
    if (stt == ORC_OK) {
        int h = 0;
        ORC_HIP(hipMemcpyAsync(&h, dev_status, sizeof(int), hipMemcpyDeviceToHost, st));
        ORC_HIP(hipStreamSynchronize(st));
        stt = h;
    }
    arena.release(mk);
    return stt;
}

}  // namespace orc

// ------------------------------------------------------------------ test hooks (private fns of the reference made observable)
namespace orc {

int amg_debug_coarsen(const MatView &A, Arena &arena, std::vector<int> &choice_h, std::vector<int64_t> &row_ptr_h,
                      std::vector<int64_t> &col_h, std::vector<double> &val_h, int *rounds) {
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
