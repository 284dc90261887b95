// linalg.hip — SELL-64 SpMV, wave reductions, BiCGSTAB, Jacobi and the iterative_solve driver
// (SURVEY §2.1 K1-K5).  Reference: src/linear_algebra.rs:144-299.
//
// Everything is HBM-bound fp64 (AI ~ 0.13 flop/B): no MFMA, the levers are coalescing (SELL-64),
// XCD-local x-vector reuse, fused vector updates and no host round-trips inside a solve.
// Compiled with -ffp-contract=off: rustc never fuses a*b+c, and bit-parity of y = A x with the
// CPU oracle depends on that.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>

#include "linalg_kernels.hpp"

namespace orc {


// ------------------------------------------------------------------ reductions
__global__ __launch_bounds__(1024) void reduce_partials_k(const double *__restrict__ partials, int count, int nq, double *__restrict__ out) {
    __shared__ double lds[16];
    for (int q = 0; q < nq; ++q) {
        double v = 0.;
        for (int i = threadIdx.x; i < count; i += blockDim.x) v += partials[(size_t)q * count + i];
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = 0.;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += lds[i];
            out[q] = r;
        }
    }
}

int reduce_partials(const double *partials, int count, int nq, double *out, bool global) {
    hipLaunchKernelGGL(reduce_partials_k, dim3(1), dim3(1024), 0, ctx().stream, partials, count, nq, out);
    ORC_HIP(hipGetLastError());
    if (global && ctx().world > 1) ORC_TRY(comm_allreduce_sum(out, nq));
    return ORC_OK;
}

// ------------------------------------------------------------------ reference-order reductions (verification mode)
// OrcSettings.reduction_order = ORC_REDUCTION_REFERENCE: every dot product / norm of the solvers is evaluated in the
// association of nalgebra 0.32.4's `dotx` (base/blas.rs): eight running accumulators
// over blocks of 8, folded as res += (acc0+acc4); (acc1+acc5); (acc2+acc6); (acc3+acc7), then the tail left to right.
// Lane k of one wavefront owns accumulator k and walks its elements in order — n/8 dependent additions, so this is a
// slow path (milliseconds per ten million rows); it exists so that a device solve can be compared with the reference's
// arithmetic BIT FOR BIT at any iteration count, instead of through tolerances that the unguarded r_hat_0 = 1
// BiCGSTAB (linear_algebra.rs:252) amplifies.  a == nullptr stands for the all-ones r_hat_0 (1.0 * b[i] == b[i]).
// [r04] The n/8 dependent additions per accumulator are the floor (about 4 ms for 10.24 M elements); r02/r03's kernel paid a
// global-memory round trip per eight blocks on top of it (0.2 s per dot product at that size: ten minutes per SIMPLE iteration of
// the benchmark in this mode).  Now the products a[i] * b[i] are formed by fifteen loader wavefronts, coalesced, into a double-
// buffered LDS tile (the multiplication is element-wise: who performs it changes nothing), while lanes 0-7 of wavefront 0 walk the
// previous tile in order.  Same accumulators, same order of additions, same final fold: every bit as before.
constexpr int kDotTile = 4096;  // elements per LDS tile (2 x 32 KB)
__global__ __launch_bounds__(1024) void dot_reference_k(const double *__restrict__ a, const double *__restrict__ b, int64_t n,
                                                        double *__restrict__ out, const double *__restrict__ skip_flags) {
    __shared__ double tile[2][kDotTile];
    if (skip_flags && (skip_flags[0] != 0. || skip_flags[1] != 0.)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n8 = (n >> 3) << 3;  // elements in whole blocks of eight
    const int64_t n_tiles = (n8 + kDotTile - 1) / kDotTile;
    auto load = [&](int64_t t, int first, int stride) {  // products of tile t into tile[t & 1]
        double *dst = tile[t & 1];
        const int64_t base = t * kDotTile;
        for (int e = first; e < kDotTile; e += stride) {
            const int64_t i = base + e;
            if (i < n8) dst[e] = (a ? a[i] : 1.) * b[i];
        }
    };
    double acc = 0.;
    if (n_tiles > 0) load(0, tid, 1024);
    __syncthreads();
    for (int64_t t = 0; t < n_tiles; ++t) {
        if (wave == 0) {
            if (lane < 8) {
                const double *src = tile[t & 1] + lane;
                const int64_t left = n8 - t * kDotTile;
                const int cnt = (int)((left < kDotTile ? left : kDotTile) >> 3);  // blocks in this tile
                int j = 0;
                for (; j + 16 <= cnt; j += 16) {
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = src[(j + q) << 3];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc += v[q];
                }
                for (; j < cnt; ++j) acc += src[j << 3];
            }
        } else if (t + 1 < n_tiles) {
            load(t + 1, tid - 64, 960);
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // lane k < 4 forms acc_k + acc_{k+4}; lane 0 adds the four pairs and the tail in order
    const double hi = __shfl_down(acc, 4, 64);
    const double pair = acc + hi;
    const double p1 = __shfl(pair, 1, 64), p2 = __shfl(pair, 2, 64), p3 = __shfl(pair, 3, 64);
    if (lane == 0) {
        double res = 0.;
        res += pair;
        res += p1;
        res += p2;
        res += p3;
        for (int64_t k = n8; k < n; ++k) res += (a ? a[k] : 1.) * b[k];
        out[0] = res;
    }
}

int dot_reference(const double *a, const double *b, int64_t n, double *out, const double *skip_flags) {
    hipLaunchKernelGGL(dot_reference_k, dim3(1), dim3(1024), 0, ctx().stream, a, b, n, out, skip_flags);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// out[0] = ((0 + a[0]) + a[1]) + ...: the plain left-to-right fold behind nalgebra's `sum()` / `mean()` (solver.rs:206-208) and the
// running sums of the reference's cell loops (solver.rs:1224, discretization.rs:338) — ONE chain of n dependent additions (about
// 35 ms for 10.24 M elements): lane 0 of wavefront 0 walks LDS tiles the other fifteen wavefronts fill.  Verification mode only.
__global__ __launch_bounds__(1024) void sum_reference_k(const double *__restrict__ a, int64_t n, double *__restrict__ out) {
    __shared__ double tile[2][kDotTile];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t n_tiles = (n + kDotTile - 1) / kDotTile;
    auto load = [&](int64_t t, int first, int stride) {
        double *dst = tile[t & 1];
        const int64_t base = t * kDotTile;
        for (int e = first; e < kDotTile; e += stride)
            if (base + e < n) dst[e] = a[base + e];
    };
    double acc = 0.;
    if (n_tiles > 0) load(0, tid, 1024);
    __syncthreads();
    for (int64_t t = 0; t < n_tiles; ++t) {
        if (wave == 0) {
            if (tid == 0) {
                const double *src = tile[t & 1];
                const int64_t left = n - t * kDotTile;
                const int cnt = (int)(left < kDotTile ? left : kDotTile);
                int j = 0;
                for (; j + 16 <= cnt; j += 16) {
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = src[j + q];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc += v[q];
                }
                for (; j < cnt; ++j) acc += src[j];
            }
        } else if (t + 1 < n_tiles) {
            load(t + 1, tid - 64, 960);
        }
        __syncthreads();
    }
    if (tid == 0) out[0] = acc;
}
int sum_reference(const double *a, int64_t n, double *out) {
    hipLaunchKernelGGL(sum_reference_k, dim3(1), dim3(1024), 0, ctx().stream, a, n, out);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}


static inline bool reference_order(const MatView &A) { return ctx().reduction_order == ORC_REDUCTION_REFERENCE && A.halo == nullptr; }

// Non-temporal matrix loads pay where the matrix streams through the caches (10.24 M cells: ~1 GB per product; in-loop level-0
// product 198 -> 178 us) and cost where it lives in the 256 MB Infinity Cache (1.03 M cells, 62 MB: 0.65 -> 0.60 of peak).
// ORC_SPMV_NT=0 / 1 forces the policy.
static inline int stream_nt(int64_t stream_bytes) {
    const int forced = cfg().spmv_nt;
    if (forced >= 0) return forced != 0;
    return stream_bytes > ((int64_t)128 << 20);
}

int matview_stream_nt(const MatView &A) { return stream_nt(A.pk.ptr && A.xw.lidx ? A.pk.total * 10 : A.P.padded * (A.P.col16 ? 10 : 12)); }

static inline int spmv_grid(int32_t n_slices) {
    int64_t g = ((int64_t)n_slices + 3) / 4;  // 4 waves (slices) per workgroup
    const int cap = cfg().spmv_grid > 0 ? std::max(8, cfg().spmv_grid) : kMaxGrid;  // (ORC_SPMV_GRID: a test hook of the partial-sum bound)
    if (g > cap) g = cap;
    if (g >= 8) g = (g / 8) * 8;  // multiple of 8 for the XCD-aware walk
    return clamp_partials_grid(g);
}

// ------------------------------------------------------------------ SELL build / import / export
int sell_from_csr_host(int64_t n, int64_t ncols, const int64_t *row_ptr, const int64_t *col, SellMatrix &out) {
    if (ncols < n) ncols = n;
    if (n < 0) return set_error(ORC_ERR_BAD_ARGUMENT, "negative row count");
    const int64_t nnz = n > 0 ? row_ptr[n] : 0;
    const int32_t n_slices = (int32_t)((n + 63) / 64);
    std::vector<int64_t> slice_ptr((size_t)n_slices + 1, 0);
    std::vector<int32_t> row_len((size_t)std::max<int64_t>(n, 1));
    for (int32_t s = 0; s < n_slices; ++s) {
        int64_t w = 0;
        for (int64_t r = (int64_t)s * 64; r < std::min<int64_t>(n, (int64_t)s * 64 + 64); ++r) w = std::max(w, row_ptr[r + 1] - row_ptr[r]);
        slice_ptr[s + 1] = slice_ptr[s] + w * 64;
    }
    const int64_t padded = slice_ptr[n_slices];
    if (padded >= (int64_t)1 << 31) return set_error(ORC_ERR_BAD_ARGUMENT, "matrix too large for 32-bit element offsets (%lld)", (long long)padded);
    std::vector<int32_t> scol((size_t)std::max<int64_t>(padded, 1), 0), diag((size_t)std::max<int64_t>(n, 1), -1);
    bool symmetric = true;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t b = row_ptr[r], e = row_ptr[r + 1];
        row_len[r] = (int32_t)(e - b);
        const int64_t base = slice_ptr[r >> 6] + (r & 63);
        for (int64_t k = 0; k < e - b; ++k) {
            const int64_t c = col[b + k];
            if (c < 0 || c >= ncols) return set_error(ORC_ERR_BAD_ARGUMENT, "column index out of range");
            if (k > 0 && col[b + k - 1] >= c) return set_error(ORC_ERR_BAD_ARGUMENT, "CSR columns must be strictly ascending per row");
            scol[base + k * 64] = (int32_t)c;
            if (c == r) diag[r] = (int32_t)(base + k * 64);
            if (symmetric && c != r && c < n) {
                const int64_t *lo = col + row_ptr[c], *hi = col + row_ptr[c + 1];
                const int64_t *it = std::lower_bound(lo, hi, r);
                if (it == hi || *it != r) symmetric = false;
            }
        }
        // padding slots point at the row itself (never dereferenced: guarded by row_len)
        const int64_t width = (slice_ptr[(r >> 6) + 1] - slice_ptr[r >> 6]) >> 6;
        for (int64_t k = e - b; k < width; ++k) scol[base + k * 64] = (int32_t)r;
    }
    out.n = n; out.ncols = ncols; out.nnz = nnz; out.padded = padded; out.n_slices = n_slices; out.symmetric = symmetric;
    out.ragged = (double)padded > 1.08 * (double)std::max<int64_t>(nnz, 1) ? (padded < 24 * std::max<int64_t>(n, 1) ? 2 : 1) : 0;
    ORC_TRY(out.slice_ptr.upload(slice_ptr.data(), slice_ptr.size()));
    ORC_TRY(out.row_len.upload(row_len.data(), (size_t)n));
    ORC_TRY(out.col.upload(scol.data(), (size_t)padded));
    // narrow column image (SellDev): per slice and depth the smallest column among the rows that reach that depth + 16-bit offsets
    const bool narrow_on = cfg().spmv_narrow_cols;
    if (narrow_on && padded > 0) {
        std::vector<uint16_t> c16((size_t)padded, 0);
        std::vector<int32_t> cbase((size_t)(padded / 64), 0);
        bool all_fit = true;
        int64_t wide_slices = 0;
        const bool count_wide = cfg().trace;
        for (int32_t s_ = 0; s_ < n_slices && (all_fit || count_wide); ++s_) {
            const int64_t sb = slice_ptr[s_], w = (slice_ptr[s_ + 1] - sb) / 64;
            const int64_t r0 = (int64_t)s_ * 64, r1 = std::min<int64_t>(n, r0 + 64);
            bool fits = true;
            for (int64_t k = 0; k < w && fits; ++k) {
                int64_t lo = INT64_MAX, hi = -1;
                for (int64_t r = r0; r < r1; ++r)
                    if (k < row_len[r]) { const int64_t c = scol[sb + k * 64 + (r - r0)]; lo = std::min(lo, c); hi = std::max(hi, c); }
                if (hi < 0) { cbase[(size_t)(sb / 64 + k)] = 0; continue; }
                if (hi - lo > 65535) { fits = false; break; }
                cbase[(size_t)(sb / 64 + k)] = (int32_t)lo;
                for (int64_t r = r0; r < r1; ++r)
                    if (k < row_len[r]) c16[(size_t)(sb + k * 64 + (r - r0))] = (uint16_t)(scol[sb + k * 64 + (r - r0)] - lo);
            }
            all_fit = all_fit && fits;
            if (!fits) ++wide_slices;
        }
        if (count_wide && wide_slices) fprintf(stderr, "[orc sell] narrow column image: %lld of %d slices have a depth that spans more than 65 535 columns\n", (long long)wide_slices, n_slices);
        if (all_fit) {  // all or nothing: the product kernels have no per-slice branch (scalar registers, see spmv_uniform_k)
            ORC_TRY(out.col16.upload(c16.data(), c16.size()));
            ORC_TRY(out.colbase.upload(cbase.data(), cbase.size()));
        }
    }
    ORC_TRY(out.diag_pos.upload(diag.data(), (size_t)n));
    ORC_TRY(out.csr_row_ptr.upload(row_ptr, (size_t)n + 1));
    // the pattern half of the row-contiguous mirror: CSR itself, addressed per slice (SellDev::rows_*)
    // (r04, measured twice at 10.24 M cells.  First half of the round: not a millisecond in any set-up phase on one stream — the walks are
    // latency-bound either way — and 2.8 GB more: off.  End of the round, with the set-up's counters and launches out of the way, in the CONCURRENT
    // iteration: 788.3 / 780.9 -> 774.0 / 771.4 ms on one box — a row is 2 cache lines instead of 15, and the fine level's sweeps and cascades
    // stop taking ~150 GB per iteration from the products beside them.  ON by default; ORC_AMG_L0_MIRROR=0 leaves it out.)
    const bool l0_mirror = cfg().amg_l0_mirror;
    if (l0_mirror && n > 0 && nnz > 0 && nnz < ((int64_t)1 << 31)) {
        std::vector<long long> rb((size_t)n_slices);
        std::vector<int32_t> ri((size_t)n), rc((size_t)nnz);
        for (int32_t s_ = 0; s_ < n_slices; ++s_) rb[(size_t)s_] = (long long)row_ptr[(int64_t)s_ * 64];
        for (int64_t r = 0; r < n; ++r) ri[(size_t)r] = (int32_t)(row_ptr[r] - row_ptr[(r >> 6) << 6]);
        for (int64_t q = 0; q < nnz; ++q) rc[(size_t)q] = (int32_t)col[q];
        ORC_TRY(out.rows_base.upload(rb.data(), rb.size()));
        ORC_TRY(out.rows_intra.upload(ri.data(), ri.size()));
        ORC_TRY(out.rows_col.upload(rc.data(), rc.size()));
    }
    return ORC_OK;
}

__global__ void sell_import_k(SellDev P, const int64_t *__restrict__ row_ptr, const double *__restrict__ csr, double *__restrict__ sell) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < P.n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = P.slice_ptr[r >> 6] + (r & 63), b = row_ptr[r];
        const int len = P.row_len[r];
        const int width = (int)((P.slice_ptr[(r >> 6) + 1] - P.slice_ptr[r >> 6]) >> 6);
        for (int k = 0; k < width; ++k) sell[base + (int64_t)k * 64] = k < len ? csr[b + k] : 0.;
    }
}
__global__ void sell_export_k(SellDev P, const int64_t *__restrict__ row_ptr, const double *__restrict__ sell, double *__restrict__ csr) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < P.n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = P.slice_ptr[r >> 6] + (r & 63), b = row_ptr[r];
        const int len = P.row_len[r];
        for (int k = 0; k < len; ++k) csr[b + k] = sell[base + (int64_t)k * 64];
    }
}

int sell_import_values(const SellMatrix &m, const double *csr_vals_dev, double *sell_vals_dev) {
    if (m.n == 0) return ORC_OK;
    hipLaunchKernelGGL(sell_import_k, dim3(grid_for(m.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.csr_row_ptr.p, csr_vals_dev, sell_vals_dev);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}
// values of a view's padded image -> row-contiguous (CSR) order: the VALUE half of the level-0 row mirror (SellDev::rows_*)
int sell_rows_values_dev(const SellDev &P, const double *sell_vals_dev, double *rows_vals_dev) {
    if (P.n == 0 || !P.csr_row_ptr) return ORC_OK;
    hipLaunchKernelGGL(sell_export_k, dim3(grid_for(P.n)), dim3(kBlock), 0, ctx().stream, P, P.csr_row_ptr, sell_vals_dev, rows_vals_dev);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}
int sell_export_values(const SellMatrix &m, const double *sell_vals_dev, double *csr_vals_dev) {
    if (m.n == 0) return ORC_OK;
    hipLaunchKernelGGL(sell_export_k, dim3(grid_for(m.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.csr_row_ptr.p, sell_vals_dev, csr_vals_dev);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// ------------------------------------------------------------------ vector kernels
__global__ void fill_k(double *x, double v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
int vec_fill(double *x, double v, int64_t n) {
    if (n == 0) return ORC_OK;
    hipLaunchKernelGGL(fill_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, x, v, n);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}
int vec_copy(double *dst, const double *src, int64_t n) {
    if (n) ORC_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx().stream));
    return ORC_OK;
}

// dinv[i] = 1 / A(i,i) through the view; 0 where the diagonal is not stored (the reference's
// p_inv row is then empty: linear_algebra.rs:160-165)
__global__ void diag_inverse_k(MatView A, double *__restrict__ dinv) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.P.n; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = A.P.diag_pos[r];
        dinv[r] = d >= 0 ? 1. / view_value(A, r, d) : 0.;
    }
}
int diag_inverse_dev(const MatView &A, double *dinv) {
    if (A.P.n == 0) return ORC_OK;
    hipLaunchKernelGGL(diag_inverse_k, dim3(grid_for(A.P.n)), dim3(kBlock), 0, ctx().stream, A, dinv);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}
// out = 0 + s * b   (p_inv * b as a one-entry-per-row SpMV, linear_algebra.rs:165)
__global__ void scale_vec_k(const double *__restrict__ s, const double *__restrict__ b, double *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0. + s[i] * b[i];
}

int scale_vec_dev(const double *sv, const double *b, double *out, int64_t n) {
    if (n == 0) return ORC_OK;
    hipLaunchKernelGGL(scale_vec_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, sv, b, out, n);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// ------------------------------------------------------------------ SpMV epilogues
struct EpiStore {  // y = A x
    static constexpr int kReductions = 0;
    double *y;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &, double &) const { y[row] = acc; }
};
struct EpiStoreSum {  // y = A x ; partial sum(y)          (nu = A p, r_hat_0 . nu : linear_algebra.rs:256-257)
    static constexpr int kReductions = 1;
    double *y;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &) const { y[row] = acc; r0 += acc; }
};
struct EpiResidual {  // r = b - A x ; p = r ; partial sum(r)  (linear_algebra.rs:250-254)
    static constexpr int kReductions = 1;
    const double *b;
    double *r, *p;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &) const {
        const double v = b[row] - acc;
        r[row] = v;
        if (p) p[row] = v;
        r0 += v;
    }
};
struct EpiResidualNorm {  // partial sum((b - A x)^2)       (linear_algebra.rs:97, :202)
    static constexpr int kReductions = 1;
    const double *b;
    double *r;  // optional
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &) const {
        const double v = b[row] - acc;
        if (r) r[row] = v;
        r0 += v * v;
    }
};
struct EpiTs {  // t = A s ; partials t.s, t.t            (linear_algebra.rs:260-261)
    static constexpr int kReductions = 2;
    const double *s;
    double *t;
    __device__ __forceinline__ void apply(int64_t row, double acc, double &r0, double &r1) const {
        t[row] = acc;
        r0 += acc * s[row];
        r1 += acc * acc;
    }
};

template <class Epi>
static int launch_spmv(const MatView &A_in, const double *x, const Epi &epi, double *partials, int *grid_out, const double *skip_flags = nullptr) {
    MatView A = A_in;
    A.nt = matview_stream_nt(A);
    int g = spmv_grid(A.P.n_slices);
    const bool xwin = A.xw.lidx != nullptr && A.pk.ptr != nullptr;
    if (xwin) {
        // One workgroup per 256-row block.  The blocks differ in cost (row lengths; blocks without a window gather from global
        // memory), and a workgroup's share is fixed, so MORE workgroups than are resident balance better: r02's 5 per CU left the
        // chip at 10 of 20 waves per CU on average (profiles/r03_pmc_products.csv: SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE; 4 are resident
        // with 32.7 KB of LDS and 92-96 VGPRs each); 8 per CU = the 2048 partial sums a product may write (kMaxPartials) measured
        // level 2: 251 -> 245 us, level 3: 289 -> 270 us.
        const int per_cu = cfg().xwin_wgs_per_cu;  // (8; a test hook sweeps it far past the partial-sum bound)
        static const int n_cu = [] {
            hipDeviceProp_t prop;
            int dev = 0;
            return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }();
        int64_t gb = ((int64_t)A.P.n_slices + 3) / 4;
        // [r05] ONE workgroup per block, dispatched by the hardware as wave slots free up: with 2 048 persistent workgroups of 4-5 (level 2) or 2-3 blocks
        // (level 3) each, a third of a CU's wave slots stood empty on average (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE: 13 of 20) — the plain product 230-235 ->
        // 215-217 us on level 2, 250 -> 243 us on level 3 (scripts/archive/gpu_r05_z.sh).  Products with reductions fold their sums inside the launch
        // (spmv_xwin_k, XWinDev::fold_scratch): the consumers see one sum per quantity.
        // Blocks of short rows (config 5's level 1: 4 350 entries per block) are over before the ticket of the fold has paid for itself (its iteration
        // +5 ... +10 ms with one block per workgroup): such a level's workgroups take up to four blocks.  Pulling blocks from per-XCD queues with persistent workgroups (one ticket per workgroup
        // instead of one per block) was measured too: the returning atomic at the head of a wavefront's in-order memory queue holds up its stream —
        // levels 2-3 0.54 / 0.58, the iteration 768-771 against 736-743 ms.  Two to four blocks per workgroup where blocks are short (12 000 / 18 000 / 30 000
        // entries per workgroup): 727 / 724, 729 / 733, 742 / 735 against 726 / 730 ms; config 5 475 / 472, 478 / 473, 488 / 491 against 478 / 479.
        const bool one_per_block = cfg().xwin_wg_per_block && (Epi::kReductions == 0 || A.xw.fold_scratch != nullptr) && A.pk.total > 0;
        if (one_per_block) {
            // about 12 000 entries per workgroup: one block on the channel's levels 2-3 (8 450 / 18 000 entries per block), three on config 5's level 1 (4 350)
            const int64_t per_block = std::max<int64_t>(1, A.pk.total / std::max<int64_t>(gb, 1));
            const int64_t blocks_per_wg = std::min<int64_t>(4, std::max<int64_t>(1, (12000 + per_block / 2) / per_block));
            gb = (gb + blocks_per_wg - 1) / blocks_per_wg;
            gb = (gb + 7) / 8 * 8;
        } else {
            A.xw.fold_scratch = nullptr;  // the kernel writes one partial sum per workgroup
            if (gb > (int64_t)n_cu * per_cu) gb = (int64_t)n_cu * per_cu;
            gb = clamp_partials_grid(gb);  // whatever the CU count (304 on gfx942) and the switch: the epilogue writes partials[blockIdx.x]
        }
        if (gb >= 8) gb = (gb / 8) * 8;
        g = (int)std::max<int64_t>(gb, 1);
    }
    const bool overlap_on = cfg().halo_overlap;
    // Partitioned level-0 operator: the rows without a ghost column (a contiguous run of slices, HaloPlan::interior_*) are
    // multiplied on a second stream while the exchange travels; the rows along the cuts follow it on the library stream.
    HaloPlan *H = A.halo;
    const bool uniform_kernel = !A.pk.ptr && A.P.ragged != 1;
    if (H && overlap_on && uniform_kernel && A.slice_hi < 0 && ctx().world > 1 && g >= 64 &&
        (int64_t)(H->interior_hi - H->interior_lo) * 2 >= (int64_t)A.P.n_slices) {
        const int g_b = std::max(8, (g / 8 / 8) * 8), g_i = std::max(8, ((g - 2 * g_b) / 8) * 8);
        const int total = g_i + 2 * g_b;
        if (grid_out) *grid_out = total;
        if (!H->aux_stream) {
            hipStream_t st2;
            hipEvent_t e1, e2;
            ORC_TRY(stream_create(&st2, kSolveStream, 0, "halo-overlap"));
            ORC_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            ORC_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            H->aux_stream = st2; H->ev_ready = e1; H->ev_done = e2;
        }
        hipStream_t lib = ctx().stream, aux = (hipStream_t)H->aux_stream;
        ORC_HIP(hipEventRecord((hipEvent_t)H->ev_ready, lib));  // x and whatever the epilogue reads are complete
        ORC_HIP(hipStreamWaitEvent(aux, (hipEvent_t)H->ev_ready, 0));
        MatView V = A;
        V.part_stride = total;
        V.slice_lo = H->interior_lo; V.slice_hi = H->interior_hi; V.part_base = 0;
        // RCCL: the exchange is queued first, so that its kernels are resident before the interior product fills the CUs.
        // The debug transport blocks this thread inside exchange(): there the interior product is launched first.
        const bool exchange_first = !comm_host_transport_active();
        if (exchange_first) ORC_TRY(H->exchange(const_cast<double *>(x)));  // C1 on the library stream (every RCCL call stays there)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false>), dim3(g_i), dim3(kBlock), 0, aux, V, x, epi, partials, skip_flags);
        ORC_HIP(hipEventRecord((hipEvent_t)H->ev_done, aux));
        if (!exchange_first) {
            const int ex = H->exchange(const_cast<double *>(x));
            if (ex != ORC_OK) {  // the interior product is in flight: the library stream must not run ahead of it
                (void)hipStreamWaitEvent(lib, (hipEvent_t)H->ev_done, 0);
                return ex;
            }
        }
        V.slice_lo = 0; V.slice_hi = H->interior_lo; V.part_base = g_i;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false>), dim3(g_b), dim3(kBlock), 0, lib, V, x, epi, partials, skip_flags);
        V.slice_lo = H->interior_hi; V.slice_hi = A.P.n_slices; V.part_base = g_i + g_b;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false>), dim3(g_b), dim3(kBlock), 0, lib, V, x, epi, partials, skip_flags);
        ORC_HIP(hipStreamWaitEvent(lib, (hipEvent_t)H->ev_done, 0));
        ORC_HIP(hipGetLastError());
        ctx().halo_overlaps += 1;
        return ORC_OK;
    }
    if (grid_out) *grid_out = (xwin && Epi::kReductions > 0 && A.xw.fold_scratch) ? 1 : g;  // (folded inside the launch: one sum per quantity)
    if (A.P.n == 0) return ORC_OK;
    if (A.halo) ORC_TRY(A.halo->exchange(const_cast<double *>(x)));  // C1: refresh the ghost entries of x
    if (xwin) {
        const size_t xwin_smem = sizeof(double) * (size_t)std::max(1, std::min(A.xw.cap, kXWinCap));
        if (!A.s1 && !A.s2 && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_xwin_k<Epi, false, true>), dim3(g), dim3(kBlock), xwin_smem, ctx().stream, A, x, epi, partials, skip_flags);
        else if (!A.s1 && !A.s2) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_xwin_k<Epi, false>), dim3(g), dim3(kBlock), xwin_smem, ctx().stream, A, x, epi, partials, skip_flags);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_xwin_k<Epi>), dim3(g), dim3(kBlock), xwin_smem, ctx().stream, A, x, epi, partials, skip_flags);
        ORC_HIP(hipGetLastError());
        return ORC_OK;
    }
    if (A.pk.ptr)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_k<Epi, kSpmvPacked>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    else if (A.P.ragged == 1)  // long ragged rows without a mirror: every slot clamped, nothing skipped
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_k<Epi, kSpmvRagged>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    else if (A.persistent_pattern) {  // mesh-pattern matrices (level 0): wave-uniform loads, predicated gathers
        const bool narrow = A.P.col16 != nullptr, scaled = A.s1 || A.s2;
        if (narrow && !scaled && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true, true, false, true>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else if (narrow && !scaled) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true, true, false>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else if (narrow) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true, true, true>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else if (!scaled && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true, false, false, true>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else if (!scaled) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true, false, false>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, true>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    }
    else if (!(A.s1 || A.s2) && A.P.col16) {  // first coarse level, scaled values materialised, narrow column image
        if (A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, false, true, false, true>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, false, true, false>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    } else if (!(A.s1 || A.s2))  // short ragged rows (first coarse level): the same kernel under its own name; scaled values materialised
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false, false, false, false>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_uniform_k<Epi, false>), dim3(g), dim3(kBlock), 0, ctx().stream, A, x, epi, partials, skip_flags);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int spmv_dev(const MatView &A, const double *x, double *y) {
    EpiStore e{y};
    return launch_spmv(A, x, e, nullptr, nullptr);
}

int residual_dev(const MatView &A, const double *b, const double *x, double *r) {
    int g = 0;
    static double *const dummy = [] {  // thread-safe one-time allocation (concurrent solves)
        double *p = nullptr;
        return hipMalloc((void **)&p, sizeof(double) * kMaxPartials) == hipSuccess ? p : nullptr;
    }();
    if (!dummy) return set_error(ORC_ERR_HIP, "hipMalloc of the residual scratch failed");
    return launch_spmv(A, x, EpiResidual{b, r, nullptr}, dummy, &g);
}

int residual_norm2_dev(const MatView &A, const double *b, const double *x, double *partials, double *out, double *r_scratch) {
    int g = 0;
    const bool ref = reference_order(A) && r_scratch != nullptr;
    ORC_TRY(launch_spmv(A, x, EpiResidualNorm{b, ref ? r_scratch : nullptr}, partials, &g));
    if (ref) return dot_reference(r_scratch, r_scratch, A.P.n, out, nullptr);
    return reduce_partials(partials, g, 1, out, A.halo != nullptr);
}

// ------------------------------------------------------------------ BiCGSTAB (linear_algebra.rs:247-269)
// scal[] layout (device doubles):
enum { S_RHO0 = 0, S_RHO1 = 1, S_SUM_NU = 2, S_TS = 3, S_TT = 4, S_FROZEN = 5, S_FROZEN2 = 6, S_COUNT = 8 };

// Breakdown guard (OrcSettings.breakdown_guard, new-build extension).  The reference iterates a fixed
// count with no test at all (:255-268); when rho, r_hat.nu, t.t or omega is exactly 0 (a cancelling
// tree sum, a zero right-hand side, a converged start) it divides 0/0 and the SIMPLE loop panics
// with "solution diverged".  With the guard the solve freezes instead: x keeps its last finite
// value and the remaining iterations are no-ops.  Nothing changes when no denominator is 0.
// S_FROZEN is written only by kernels whose reaction to a breakdown is "do nothing" (so a block
// that starts late and sees the flag behaves like one that evaluated the test itself); the x/r
// update kernel reacts with x = h, r = s and therefore publishes through S_FROZEN2, which it does
// not read.
__device__ __forceinline__ bool bicg_frozen(const double *__restrict__ scal, int guard) {
    return guard && (scal[S_FROZEN] != 0. || scal[S_FROZEN2] != 0.);
}
__device__ __forceinline__ bool finite_nonzero(double v) { return v != 0. && isfinite(v); }

// s = r - alpha*nu, alpha = rho / (r_hat_0 . nu)                     (:257, :259)
// fold (null: scal[S_SUM_NU] is there already): the product's partial sums of nu, folded by every workgroup here
__global__ __launch_bounds__(kBlock) void bicg_s_k(double *__restrict__ scal, int rho_idx, const double *__restrict__ r, const double *__restrict__ nu,
                                                   double *__restrict__ s, int64_t n, int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds16[16];
    if (bicg_frozen(scal, guard)) return;
    // 16-byte accesses: two consecutive elements per lane (arena vectors are 256-byte aligned).  The first pair of every
    // thread is requested BEFORE the fold, so that its round trip and the fold's overlap.
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    const double2 *r2 = reinterpret_cast<const double2 *>(r), *nu2 = reinterpret_cast<const double2 *>(nu);
    double2 *s2 = reinterpret_cast<double2 *>(s);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 a = make_double2(0., 0.), b = make_double2(0., 0.);
    if (i < n2) { a = r2[i]; b = nu2[i]; }
    double sum_nu;
    if (fold) {
        sum_nu = fold_partials_block(fold, fold_count, lds16);
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_SUM_NU] = sum_nu;  // the later kernels of the iteration read it
    } else {
        sum_nu = scal[S_SUM_NU];
    }
    const double alpha = scal[rho_idx] / sum_nu;
    if (guard && !(finite_nonzero(scal[rho_idx]) && finite_nonzero(sum_nu) && isfinite(alpha))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_FROZEN] = 1.;
        return;
    }
    while (i < n2) {
        const int64_t nx = i + stride;
        double2 an = make_double2(0., 0.), bn = make_double2(0., 0.);
        if (nx < n2) { an = r2[nx]; bn = nu2[nx]; }
        s2[i] = make_double2(a.x - alpha * b.x, a.y - alpha * b.y);
        a = an; b = bn; i = nx;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s[n - 1] = r[n - 1] - alpha * nu[n - 1];
}
// h = x + alpha p ; x = h + omega s ; r = s - omega t ; partial sum(r)   (:258, :261-263, :265)
__global__ __launch_bounds__(kBlock) void bicg_xr_k(double *__restrict__ scal, int rho_idx, double *__restrict__ x,
                                                    const double *__restrict__ p, const double *__restrict__ s,
                                                    const double *__restrict__ t, double *__restrict__ r, int64_t n,
                                                    double *__restrict__ partials, int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[8];
    __shared__ double lds16[32];
    if (guard && scal[S_FROZEN] != 0.) return;
    // the first pairs of every thread are requested before the folds (their round trips overlap)
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    double2 *x2 = reinterpret_cast<double2 *>(x), *r2 = reinterpret_cast<double2 *>(r);
    const double2 *p2 = reinterpret_cast<const double2 *>(p), *s2 = reinterpret_cast<const double2 *>(s), *t2 = reinterpret_cast<const double2 *>(t);
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 xv = make_double2(0., 0.), pv = xv, sv = xv, tv = xv;
    if (i0 < n2) { xv = x2[i0]; pv = p2[i0]; sv = s2[i0]; tv = t2[i0]; }
    double ts, tt;
    if (fold) {  // t.s and t.t from the product's two partial arrays (fold != partials: this kernel writes its own sums)
        double both[2];
        fold_partials_multi<2>(fold, fold_count, lds16, both);  // [r04] the two folds' loads in flight together, two barriers instead of four: the same bits
        ts = both[0]; tt = both[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) { scal[S_TS] = ts; scal[S_TT] = tt; }
    } else {
        ts = scal[S_TS]; tt = scal[S_TT];
    }
    const double alpha = scal[rho_idx] / scal[S_SUM_NU];
    double omega = ts / tt;
    const bool bad = guard && !(finite_nonzero(tt) && isfinite(omega));
    double acc = 0.;
    if (bad) {
        // t = A s vanished (s is already the zero residual) or overflowed: take x = h, r = s and stop
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            x[i] = x[i] + alpha * p[i];
            const double si = s[i];
            r[i] = si;
            acc += si;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_FROZEN2] = 1.;
    } else {
        int64_t i = i0;
        while (i < n2) {
            const int64_t nx = i + stride;
            double2 xn = make_double2(0., 0.), pn = xn, sn = xn, tn = xn;
            if (nx < n2) { xn = x2[nx]; pn = p2[nx]; sn = s2[nx]; tn = t2[nx]; }
            const double h0 = xv.x + alpha * pv.x, h1 = xv.y + alpha * pv.y;
            x2[i] = make_double2(h0 + omega * sv.x, h1 + omega * sv.y);
            const double q0 = sv.x - omega * tv.x, q1 = sv.y - omega * tv.y;
            r2[i] = make_double2(q0, q1);
            acc += q0;
            acc += q1;
            xv = xn; pv = pn; sv = sn; tv = tn; i = nx;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t i = n - 1;
            const double h = x[i] + alpha * p[i];
            const double si = s[i];
            x[i] = h + omega * si;
            const double ri = si - omega * t[i];
            r[i] = ri;
            acc += ri;
        }
    }
    const double tsum = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = tsum;
}
// beta = rho/rho_prev * alpha/omega ; p = r + beta (p - omega nu)       (:266-267)
__global__ __launch_bounds__(kBlock) void bicg_p_k(double *__restrict__ scal, int rho_prev_idx, int rho_idx, const double *__restrict__ r,
                                                   const double *__restrict__ nu, double *__restrict__ p, int64_t n, int guard,
                                                   const double *__restrict__ fold, int fold_count) {
    __shared__ double lds16[16];
    if (bicg_frozen(scal, guard)) {
        // bicg_xr_k took x = h, r = s and published through S_FROZEN2, which it does not read itself: promote it, or the next
        // iteration's bicg_xr_k would add alpha p once more (every workgroup of THIS launch returns here either way)
        if (guard && scal[S_FROZEN2] != 0. && blockIdx.x == 0 && threadIdx.x == 0) scal[S_FROZEN] = 1.;
        return;
    }
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    double2 *p2 = reinterpret_cast<double2 *>(p);
    const double2 *r2 = reinterpret_cast<const double2 *>(r), *nu2 = reinterpret_cast<const double2 *>(nu);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 rv = make_double2(0., 0.), pv = rv, nv = rv;
    if (i < n2) { rv = r2[i]; pv = p2[i]; nv = nu2[i]; }  // requested before the fold
    double rho;
    if (fold) {  // rho = sum(r) from bicg_xr_k's partial sums
        rho = fold_partials_block(fold, fold_count, lds16);
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[rho_idx] = rho;
    } else {
        rho = scal[rho_idx];
    }
    const double rho_prev = scal[rho_prev_idx];
    const double alpha = rho_prev / scal[S_SUM_NU];
    const double omega = scal[S_TS] / scal[S_TT];
    const double beta = rho / rho_prev * alpha / omega;
    if (guard && !(finite_nonzero(omega) && isfinite(beta))) {
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_FROZEN] = 1.;
        return;
    }
    while (i < n2) {
        const int64_t nx = i + stride;
        double2 rn = make_double2(0., 0.), pn = rn, nn = rn;
        if (nx < n2) { rn = r2[nx]; pn = p2[nx]; nn = nu2[nx]; }
        p2[i] = make_double2(rv.x + beta * (pv.x - omega * nv.x), rv.y + beta * (pv.y - omega * nv.y));
        rv = rn; pv = pn; nv = nn; i = nx;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = r[n - 1] + beta * (p[n - 1] - omega * nu[n - 1]);
}

// one more solve in which the guard fired (orc_breakdown_guard_events): a drop-in caller must be able to tell that the
// reference would have produced NaN here
__global__ void guard_event_k(const double *__restrict__ scal, int *__restrict__ counter) {
    if (scal[S_FROZEN] != 0. || scal[S_FROZEN2] != 0.) atomicAdd(counter, 1);
}

struct BicgWork {
    double *r, *p, *nu, *s, *t, *partials, *partials2, *scal;  // partials2: bicg_xr_k's sums while it still folds the product's
};

static int bicg_alloc(Arena &arena, int64_t n, BicgWork &w) {  // n = vector length incl. ghost entries
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    ORC_TRY(arena.alloc(nn, &w.r));
    ORC_TRY(arena.alloc(nn, &w.p));
    ORC_TRY(arena.alloc(nn, &w.nu));
    ORC_TRY(arena.alloc(nn, &w.s));
    ORC_TRY(arena.alloc(nn, &w.t));
    ORC_TRY(arena.alloc((size_t)2 * kMaxPartials, &w.partials));
    ORC_TRY(arena.alloc((size_t)kMaxPartials, &w.partials2));
    ORC_TRY(arena.alloc((size_t)S_COUNT, &w.scal));
    ORC_HIP(hipMemsetAsync(w.scal, 0, S_COUNT * sizeof(double), ctx().stream));
    return ORC_OK;
}

static int bicg_iteration(const MatView &A, double *x, const BicgWork &w, uint64_t it, int guard) {
    const int64_t n = A.P.n;
    const int vg = grid_for((n + 1) / 2);  // two elements per lane
    const int cur = (int)(it & 1), nxt = cur ^ 1;
    const double *skip = guard ? w.scal + S_FROZEN : nullptr;  // frozen solves skip their SpMVs too
    int g = 0;
    const bool ref = reference_order(A);  // dot products in nalgebra's association (verification mode)
    // Single GPU, tree reductions: the three sums of the iteration are folded by the kernels that consume them (every
    // workgroup folds, workgroup 0 stores the scalar for the later kernels) instead of by one-workgroup launches in between.
    const bool fused = !ref && A.halo == nullptr;
    ORC_TRY(launch_spmv(A, w.p, EpiStoreSum{w.nu}, w.partials, &g, skip));     // nu = A p, sum(nu)
    if (ref) ORC_TRY(dot_reference(nullptr, w.nu, n, w.scal + S_SUM_NU, skip));            // r_hat_0 . nu  (:257)
    else if (!fused) ORC_TRY(reduce_partials(w.partials, g, 1, w.scal + S_SUM_NU, A.halo != nullptr));
    hipLaunchKernelGGL(bicg_s_k, dim3(vg), dim3(kBlock), 0, ctx().stream, w.scal, S_RHO0 + cur, w.r, w.nu, w.s, n, guard,
                       fused ? (const double *)w.partials : (const double *)nullptr, g);
    ORC_TRY(launch_spmv(A, w.s, EpiTs{w.s, w.t}, w.partials, &g, skip));       // t = A s, t.s, t.t
    if (ref) {
        ORC_TRY(dot_reference(w.t, w.s, n, w.scal + S_TS, skip));                          // t . s, t . t  (:261)
        ORC_TRY(dot_reference(w.t, w.t, n, w.scal + S_TT, skip));
    } else if (!fused) ORC_TRY(reduce_partials(w.partials, g, 2, w.scal + S_TS, A.halo != nullptr));
    double *xr_partials = fused ? w.partials2 : w.partials;
    hipLaunchKernelGGL(bicg_xr_k, dim3(vg), dim3(kBlock), 0, ctx().stream, w.scal, S_RHO0 + cur, x, w.p, w.s, w.t, w.r, n, xr_partials, guard,
                       fused ? (const double *)w.partials : (const double *)nullptr, g);
    if (ref) ORC_TRY(dot_reference(nullptr, w.r, n, w.scal + S_RHO0 + nxt, skip));  // rho = r_hat_0 . r  (:265)
    else if (!fused) ORC_TRY(reduce_partials(w.partials, vg, 1, w.scal + S_RHO0 + nxt, A.halo != nullptr));  // rho = r_hat_0 . r
    hipLaunchKernelGGL(bicg_p_k, dim3(vg), dim3(kBlock), 0, ctx().stream, w.scal, S_RHO0 + cur, S_RHO0 + nxt, w.r, w.nu, w.p, n, guard,
                       fused ? (const double *)w.partials2 : (const double *)nullptr, vg);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// Jacobi-scaled values, materialised: out[p] = s2[row] * (s1[row] * val[p]) — the product kernels' own expression, evaluated once per
// solve instead of once per product.  The reference materialises `p_inv * a` too (linear_algebra.rs:159-166); on the device the
// point is bytes: a level-0 product is bandwidth-bound at 5.6 TB/s of real traffic (profiles/r03_pmc_products.csv) and the two
// scaling vectors are 16 of its ~125 bytes per row — read 101 times per smoothing solve, against one extra pass over the values.
__global__ __launch_bounds__(kBlock) void scale_values_k(MatView A, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const double s1 = (A.s1 && live) ? A.s1[row] : 1.;
        const double s2 = (A.s2 && live) ? A.s2[row] : 1.;
        for (int k = 0; k < width; ++k) {
            const int64_t p = base + (int64_t)k * 64 + lane;
            double t = A.val[p];
            if (A.s1) t = s1 * t;
            if (A.s2) t = s2 * t;
            out[p] = t;
        }
    }
}
// from how many iterations on a solve materialises its scaled values (ORC_MATERIALIZE_SCALING=0: never)
static inline bool materialize_scaling(uint64_t iteration_count) {
    const int min_its = cfg().materialize_scaling;
    return min_its > 0 && iteration_count >= (uint64_t)min_its;
}

// the same over a packed mirror (PackedDev): entries of depth k of a slice sit back to back in lane order
__global__ __launch_bounds__(kBlock) void scale_packed_k(MatView A, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int width = (int)((A.P.slice_ptr[slice + 1] - A.P.slice_ptr[slice]) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        const double s1 = (A.s1 && live) ? A.s1[row] : 1.;
        const double s2 = (A.s2 && live) ? A.s2[row] : 1.;
        int64_t off = A.pk.ptr[slice];
        for (int k = 0; k < width; ++k) {
            const bool in = k < len;
            const unsigned long long m = __ballot(in);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (in) {
                double t = A.pk.val[off + rank];
                if (A.s1) t = s1 * t;
                if (A.s2) t = s2 * t;
                out[off + rank] = t;
            }
            off += __popcll(m);
        }
    }
}

int materialize_scaled_view(MatView &A, uint64_t iteration_count, Arena &arena) {
    if (!((A.s1 || A.s2) && A.P.n > 0 && materialize_scaling(iteration_count))) return ORC_OK;
    if (A.pk.ptr) {
        // levels with a packed mirror + LDS windows: their products stream pk.val only (launch_spmv: production variant)
        if (!(A.xw.lidx && A.pk.total > 0)) return ORC_OK;
        double *scaled;
        ORC_TRY(arena.alloc((size_t)A.pk.total, &scaled));
        hipLaunchKernelGGL(scale_packed_k, dim3(spmv_grid(A.P.n_slices)), dim3(kBlock), 0, ctx().stream, A, scaled);
        ORC_HIP(hipGetLastError());
        A.pk.val = scaled;
        A.val = nullptr;  // the padded image keeps the unscaled values: nothing may read it through this view
        A.s1 = A.s2 = nullptr;
        return ORC_OK;
    }
    if (A.P.padded <= 0) return ORC_OK;
    double *scaled;
    ORC_TRY(arena.alloc((size_t)A.P.padded, &scaled));
    hipLaunchKernelGGL(scale_values_k, dim3(spmv_grid(A.P.n_slices)), dim3(kBlock), 0, ctx().stream, A, scaled);
    ORC_HIP(hipGetLastError());
    A.val = scaled;
    A.s1 = A.s2 = nullptr;
    return ORC_OK;
}

// linear_algebra.rs:247-269 on a view whose scalings are final (materialised or carried as s1 / s2)
static int bicgstab_run(const MatView &A, const double *b, double *x, uint64_t iteration_count, Arena &arena) {
    const int64_t n = A.P.n;
    if (n == 0) return ORC_OK;
    ArenaScope scope(arena);
    BicgWork w;
    ORC_TRY(bicg_alloc(arena, std::max(A.P.ncols, n), w));
    const int guard = ctx().breakdown_guard ? 1 : 0;
    int g = 0;
    ORC_TRY(launch_spmv(A, x, EpiResidual{b, w.r, w.p}, w.partials, &g));      // r = b - A x ; p = r ; rho = sum(r)
    if (reference_order(A)) ORC_TRY(dot_reference(nullptr, w.r, n, w.scal + S_RHO0, nullptr));  // r . r_hat_0  (:253)
    else ORC_TRY(reduce_partials(w.partials, g, 1, w.scal + S_RHO0, A.halo != nullptr));
    for (uint64_t it = 0; it < iteration_count; ++it) ORC_TRY(bicg_iteration(A, x, w, it, guard));
    if (guard && ctx().guard_events) {
        hipLaunchKernelGGL(guard_event_k, dim3(1), dim3(1), 0, ctx().stream, w.scal, ctx().guard_events);
        ORC_HIP(hipGetLastError());
    }
    return ORC_OK;
}

static int bicgstab_dev(const MatView &A_in, const double *b, double *x, uint64_t iteration_count, Arena &arena) {
    if (A_in.P.n == 0) return ORC_OK;
    ArenaScope scope(arena);
    MatView A = A_in;
    ORC_TRY(materialize_scaled_view(A, iteration_count, arena));
    return bicgstab_run(A, b, x, iteration_count, arena);
}

int jacobi_scaling_prepare_dev(const MatView &A_in, uint64_t iteration_count, Arena &arena, ScaledOperator &S) {
    ORC_TRY(ensure_init());
    const int64_t n = A_in.P.n;
    S = ScaledOperator();
    S.A = A_in;
    S.iterations = iteration_count;
    double *dinv;
    ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &dinv));
    if (n) {
        hipLaunchKernelGGL(diag_inverse_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, A_in, dinv);
        ORC_HIP(hipGetLastError());
    }
    if (!S.A.s1) S.A.s1 = dinv;
    else if (!S.A.s2) S.A.s2 = dinv;
    else return set_error(ORC_ERR_BAD_ARGUMENT, "more than two nested Jacobi scalings");
    S.dinv = dinv;
    return materialize_scaled_view(S.A, iteration_count, arena);
}

int bicgstab_scaled_dev(const ScaledOperator &S, const double *b, double *x, Arena &arena) {
    const int64_t n = S.A.P.n;
    if (n == 0) return ORC_OK;
    ArenaScope scope(arena);
    double *b_tmp;
    ORC_TRY(arena.alloc((size_t)n, &b_tmp));
    hipLaunchKernelGGL(scale_vec_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, S.dinv, b, b_tmp, n);  // :165
    ORC_HIP(hipGetLastError());
    return bicgstab_run(S.A, b_tmp, x, S.iterations, arena);
}

template <class Fn>
static int time_launches(int reps, float *ms, Fn &&launch) {
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    int st = launch();  // warm
    if (st == ORC_OK && hipEventRecord(e0, ctx().stream) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    for (int i = 0; i < reps && st == ORC_OK; ++i) st = launch();
    if (st == ORC_OK && (hipEventRecord(e1, ctx().stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) st = set_error(ORC_ERR_HIP, "hipEventRecord failed");
    if (st == ORC_OK && hipEventElapsedTime(ms, e0, e1) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipEventElapsedTime failed");
    if (st == ORC_OK) *ms /= (float)std::max(reps, 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return st;
}
int bench_inloop_products_dev(const MatView &A, const double *x, double *y, double *partials, int reps, float ms[2]) {
    int g = 0;
    ORC_TRY(time_launches(reps, &ms[0], [&] { return launch_spmv(A, x, EpiStoreSum{y}, partials, &g, nullptr); }));
    ORC_TRY(time_launches(reps, &ms[1], [&] { return launch_spmv(A, x, EpiTs{x, y}, partials, &g, nullptr); }));
    return ORC_OK;
}

int bench_bicgstab_dev(const MatView &A, const double *b, double *x, int reps, Arena &arena, float *ms) {
    const int64_t n = A.P.n;
    Arena::Mark mk = arena.mark();
    BicgWork w;
    ORC_TRY(bicg_alloc(arena, std::max(A.P.ncols, n), w));
    int g = 0;
    ORC_TRY(launch_spmv(A, x, EpiResidual{b, w.r, w.p}, w.partials, &g));
    ORC_TRY(reduce_partials(w.partials, g, 1, w.scal + S_RHO0, A.halo != nullptr));
    hipEvent_t e0, e1;
    ORC_HIP(hipEventCreate(&e0));
    ORC_HIP(hipEventCreate(&e1));
    ORC_TRY(bicg_iteration(A, x, w, 0, 0));  // warm; guard off so every timed launch does full work
    ORC_HIP(hipEventRecord(e0, ctx().stream));
    for (int it = 1; it <= reps; ++it) ORC_TRY(bicg_iteration(A, x, w, (uint64_t)it, 0));
    ORC_HIP(hipEventRecord(e1, ctx().stream));
    ORC_HIP(hipEventSynchronize(e1));
    ORC_HIP(hipEventElapsedTime(ms, e0, e1));
    *ms /= (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    arena.release(mk);
    return ORC_OK;
}


// ------------------------------------------------------------------ three systems in lock-step (MatView3, linalg.hpp)
// The u, v and w momentum systems of an iteration: one pattern, three value arrays, interleaved vectors.  Every kernel
// below keeps, per system, the thread -> element map, the order of the additions and the fold of its one-system
// counterpart above, so a system solved here and the same system solved alone agree in every bit
// (tests/test_gpu_triple.py).  Scalars of system s: scal3[idx * 3 + s].
bool triple_supported() { return ctx().reduction_order != ORC_REDUCTION_REFERENCE; }

struct EpiStore3 {
    static constexpr int kReductions = 0;
    double *y3;
    __device__ __forceinline__ void apply(int64_t row, const double (&acc)[3], double (&)[3][2]) const {
        reinterpret_cast<Vec3d *>(y3)[row] = Vec3d{acc[0], acc[1], acc[2]};
    }
};
struct EpiStoreSum3 {
    static constexpr int kReductions = 1;
    double *y3;
    __device__ __forceinline__ void apply(int64_t row, const double (&acc)[3], double (&red)[3][2]) const {
        reinterpret_cast<Vec3d *>(y3)[row] = Vec3d{acc[0], acc[1], acc[2]};
        red[0][0] += acc[0]; red[1][0] += acc[1]; red[2][0] += acc[2];
    }
};
struct EpiResidual3 {
    static constexpr int kReductions = 1;
    const double *b3;
    double *r3, *p3;
    __device__ __forceinline__ void apply(int64_t row, const double (&acc)[3], double (&red)[3][2]) const {
        const Vec3d b = reinterpret_cast<const Vec3d *>(b3)[row];
        const Vec3d v = {b.a - acc[0], b.b - acc[1], b.c - acc[2]};
        reinterpret_cast<Vec3d *>(r3)[row] = v;
        if (p3) reinterpret_cast<Vec3d *>(p3)[row] = v;
        red[0][0] += v.a; red[1][0] += v.b; red[2][0] += v.c;
    }
};
struct EpiResidualNorm3 {
    static constexpr int kReductions = 1;
    const double *b3;
    __device__ __forceinline__ void apply(int64_t row, const double (&acc)[3], double (&red)[3][2]) const {
        const Vec3d b = reinterpret_cast<const Vec3d *>(b3)[row];
        const double v0 = b.a - acc[0], v1 = b.b - acc[1], v2 = b.c - acc[2];
        red[0][0] += v0 * v0; red[1][0] += v1 * v1; red[2][0] += v2 * v2;
    }
};
struct EpiTs3 {
    static constexpr int kReductions = 2;
    const double *s3;
    double *t3;
    __device__ __forceinline__ void apply(int64_t row, const double (&acc)[3], double (&red)[3][2]) const {
        const Vec3d sv = reinterpret_cast<const Vec3d *>(s3)[row];
        reinterpret_cast<Vec3d *>(t3)[row] = Vec3d{acc[0], acc[1], acc[2]};
        red[0][0] += acc[0] * sv.a; red[0][1] += acc[0] * acc[0];
        red[1][0] += acc[1] * sv.b; red[1][1] += acc[1] * acc[1];
        red[2][0] += acc[2] * sv.c; red[2][1] += acc[2] * acc[2];
    }
};

template <class Epi3>
static int launch_spmv3(const MatView3 &A_in, const double *x3, const Epi3 &epi, double *partials, int *grid_out) {
    MatView3 A = A_in;
    A.nt = stream_nt(A.P.padded * (A.P.col16 ? 26 : 28));
    const int g = spmv_grid(A.P.n_slices);  // the one-system grid: same walk, same partial sums
    if (grid_out) *grid_out = g;
    if (A.P.n == 0) return ORC_OK;
    // Partitioned level-0 operator [r04]: as launch_spmv does for one system, the rows without a ghost column (HaloPlan::interior_*) are
    // multiplied on a second stream while the exchange of the interleaved iterate travels; the rows along the cuts follow it on the library
    // stream.  Same slice ranges, same grids and same layout of the partial sums as the one-system form: per system the same bits.
    HaloPlan *H = A.halo;
    const bool overlap_on = cfg().halo_overlap;
    const bool plain_kernel = A.mesh_pattern && A.P.col16 != nullptr && !(A.s1 || A.s2);  // the variant the solves launch (materialised, narrow columns)
    if (H && overlap_on && plain_kernel && ctx().world > 1 && g >= 64 && (int64_t)(H->interior_hi - H->interior_lo) * 2 >= (int64_t)A.P.n_slices) {
        const int g_b = std::max(8, (g / 8 / 8) * 8), g_i = std::max(8, ((g - 2 * g_b) / 8) * 8);
        const int total = g_i + 2 * g_b;
        if (grid_out) *grid_out = total;
        if (!H->aux_stream) {
            hipStream_t st2;
            hipEvent_t e1, e2;
            ORC_TRY(stream_create(&st2, kSolveStream, 0, "halo-overlap"));
            ORC_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            ORC_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            H->aux_stream = st2; H->ev_ready = e1; H->ev_done = e2;
        }
        hipStream_t lib = ctx().stream, aux = (hipStream_t)H->aux_stream;
        ORC_HIP(hipEventRecord((hipEvent_t)H->ev_ready, lib));
        ORC_HIP(hipStreamWaitEvent(aux, (hipEvent_t)H->ev_ready, 0));
        MatView3 V = A;
        V.part_stride = total;
        V.slice_lo = H->interior_lo; V.slice_hi = H->interior_hi; V.part_base = 0;
        const bool exchange_first = !comm_host_transport_active();
        if (exchange_first) ORC_TRY(H->exchange_interleaved(const_cast<double *>(x3), 3));
        auto launch = [&](const MatView3 &W, int gg, hipStream_t s) {
            if (W.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, true, false, true>), dim3(gg), dim3(kBlock), 0, s, W, x3, epi, partials);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, true, false>), dim3(gg), dim3(kBlock), 0, s, W, x3, epi, partials);
        };
        launch(V, g_i, aux);
        ORC_HIP(hipEventRecord((hipEvent_t)H->ev_done, aux));
        if (!exchange_first) {
            const int ex = H->exchange_interleaved(const_cast<double *>(x3), 3);
            if (ex != ORC_OK) {  // the interior product is in flight: the library stream must not run ahead of it
                (void)hipStreamWaitEvent(lib, (hipEvent_t)H->ev_done, 0);
                return ex;
            }
        }
        V.slice_lo = 0; V.slice_hi = H->interior_lo; V.part_base = g_i;
        launch(V, g_b, lib);
        V.slice_lo = H->interior_hi; V.slice_hi = A.P.n_slices; V.part_base = g_i + g_b;
        launch(V, g_b, lib);
        ORC_HIP(hipStreamWaitEvent(lib, (hipEvent_t)H->ev_done, 0));
        ORC_HIP(hipGetLastError());
        ctx().halo_overlaps += 1;
        return ORC_OK;
    }
    if (A.halo) ORC_TRY(A.halo->exchange_interleaved(const_cast<double *>(x3), 3));  // C1: the ghost entries of the three systems in one message per peer
    const int g_launch = g;
    if (A.mesh_pattern) {
        const bool narrow = A.P.col16 != nullptr, scaled = A.s1 || A.s2;
        if (narrow && !scaled && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, true, false, true>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
        else if (narrow && !scaled) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, true, false>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
        else if (narrow) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, true, true>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
        else if (!scaled && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, false, false, true>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
        else if (!scaled) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true, false, false>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, true>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
    } else if (!(A.s1 || A.s2) && A.P.col16 && A.nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, false, true, false, true>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
    else if (!(A.s1 || A.s2) && A.P.col16) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, false, true, false>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
    else if (!(A.s1 || A.s2)) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4, false, false, false>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv3_uniform_k<Epi3, 4>), dim3(g_launch), dim3(kBlock), 0, ctx().stream, A, x3, epi, partials);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int spmv3_dev(const MatView3 &A, const double *x3, double *y3) { return launch_spmv3(A, x3, EpiStore3{y3}, nullptr, nullptr); }

static double *triple_scratch() {  // partial sums nobody reads (residual3_dev); one allocation per process
    static double *const p = [] {
        double *q = nullptr;
        return hipMalloc((void **)&q, sizeof(double) * 3 * kMaxPartials) == hipSuccess ? q : nullptr;
    }();
    return p;
}
int residual3_dev(const MatView3 &A, const double *b3, double *x3, double *r3) {
    double *dummy = triple_scratch();
    if (!dummy) return set_error(ORC_ERR_HIP, "hipMalloc of the residual scratch failed");
    return launch_spmv3(A, x3, EpiResidual3{b3, r3, nullptr}, dummy, nullptr);
}
int residual_norm2_3_dev(const MatView3 &A, const double *b3, double *x3, double *partials, double *out3) {
    int g = 0;
    ORC_TRY(launch_spmv3(A, x3, EpiResidualNorm3{b3}, partials, &g));
    return reduce_partials(partials, g, 3, out3, A.halo != nullptr);
}

__global__ void interleave3_k(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c, double *__restrict__ out3, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<Vec3d *>(out3)[i] = Vec3d{a[i], b[i], c[i]};
}
__global__ void deinterleave3_k(const double *__restrict__ in3, double *__restrict__ a, double *__restrict__ b, double *__restrict__ c, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const Vec3d v = reinterpret_cast<const Vec3d *>(in3)[i];
        if (a) a[i] = v.a;
        if (b) b[i] = v.b;
        if (c) c[i] = v.c;
    }
}
int interleave3_dev(const double *a, const double *b, const double *c, double *out3, int64_t n) {
    if (n == 0) return ORC_OK;
    hipLaunchKernelGGL(interleave3_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, a, b, c, out3, n);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}
int deinterleave3_dev(const double *in3, double *a, double *b, double *c, int64_t n) {
    if (n == 0) return ORC_OK;
    hipLaunchKernelGGL(deinterleave3_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, in3, a, b, c, n);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

__global__ void diag_inverse3_k(MatView3 A, double *__restrict__ dinv3) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < A.P.n; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = A.P.diag_pos[r];
        double o[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double v = 0.;
            if (d >= 0) {  // view_value per system
                v = A.val[s][d];
                if (A.s1) v = A.s1[3 * r + s] * v;
                if (A.s2) v = A.s2[3 * r + s] * v;
                v = 1. / v;
            }
            o[s] = v;
        }
        reinterpret_cast<Vec3d *>(dinv3)[r] = Vec3d{o[0], o[1], o[2]};
    }
}
int diag_inverse3_dev(const MatView3 &A, double *dinv3) {
    if (A.P.n == 0) return ORC_OK;
    hipLaunchKernelGGL(diag_inverse3_k, dim3(grid_for(A.P.n)), dim3(kBlock), 0, ctx().stream, A, dinv3);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

// scale_values_k for system `sys` of a MatView3 (scalings interleaved)
__global__ __launch_bounds__(kBlock) void scale_values3_k(MatView3 A, int sys, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const double *__restrict__ val = A.val[sys];
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const double s1 = (A.s1 && live) ? A.s1[3 * row + sys] : 1.;
        const double s2 = (A.s2 && live) ? A.s2[3 * row + sys] : 1.;
        for (int k = 0; k < width; ++k) {
            const int64_t p = base + (int64_t)k * 64 + lane;
            double t = val[p];
            if (A.s1) t = s1 * t;
            if (A.s2) t = s2 * t;
            out[p] = t;
        }
    }
}

int materialize_scaled_view3(MatView3 &A, uint64_t iteration_count, Arena &arena) {
    if (!((A.s1 || A.s2) && A.P.padded > 0 && A.P.n > 0 && materialize_scaling(iteration_count))) return ORC_OK;
    for (int s = 0; s < 3; ++s) {
        double *scaled;
        ORC_TRY(arena.alloc((size_t)A.P.padded, &scaled));
        hipLaunchKernelGGL(scale_values3_k, dim3(spmv_grid(A.P.n_slices)), dim3(kBlock), 0, ctx().stream, A, s, scaled);
        A.val[s] = scaled;  // the kernel reads A.val[sys] only; the scalings are dropped once all three are through
    }
    ORC_HIP(hipGetLastError());
    A.s1 = A.s2 = nullptr;
    return ORC_OK;
}

#define SC3(idx, s) ((idx) * 3 + (s))
__device__ __forceinline__ bool bicg_frozen3(const double *__restrict__ scal3, int s, int guard) {
    return guard && (scal3[SC3(S_FROZEN, s)] != 0. || scal3[SC3(S_FROZEN2, s)] != 0.);
}

// bicg_s_k for three systems: s = r - alpha nu, alpha = rho / sum(nu); fold: the product's partial sums, system s at fold + s * fold_count
__global__ __launch_bounds__(kBlock) void bicg_s3_k(double *__restrict__ scal3, int rho_idx, const double *__restrict__ r3, const double *__restrict__ nu3,
                                                    double *__restrict__ s3, int64_t n, int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds16[3 * 16];
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    const double2 *r2 = reinterpret_cast<const double2 *>(r3), *nu2 = reinterpret_cast<const double2 *>(nu3);
    double2 *s2 = reinterpret_cast<double2 *>(s3);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 a0, a1, a2, b0, b1, b2;
    a0 = a1 = a2 = b0 = b1 = b2 = make_double2(0., 0.);
    if (i < n2) { a0 = r2[3 * i]; a1 = r2[3 * i + 1]; a2 = r2[3 * i + 2]; b0 = nu2[3 * i]; b1 = nu2[3 * i + 1]; b2 = nu2[3 * i + 2]; }
    double alpha[3];
    bool act[3];
    // fold_count == 0 (partitioned operator): fold holds the sums themselves — folded by reduce_partials_k, summed over the ranks
    double folded[3] = {0., 0., 0.};
    if (fold_count) fold_partials_multi<3>(fold, fold_count, lds16, folded);  // [r04] the three folds' loads in flight together, two barriers
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const bool frz = bicg_frozen3(scal3, s, guard);
        const double sum_nu = fold_count ? folded[s] : fold[s];
        const double rho = scal3[SC3(rho_idx, s)];
        alpha[s] = rho / sum_nu;
        const bool bad = guard && !(finite_nonzero(rho) && finite_nonzero(sum_nu) && isfinite(alpha[s]));
        act[s] = !frz && !bad;
        if (blockIdx.x == 0 && threadIdx.x == 0 && !frz) {
            scal3[SC3(S_SUM_NU, s)] = sum_nu;
            if (bad) scal3[SC3(S_FROZEN, s)] = 1.;
        }
    }
    if (act[0] && act[1] && act[2]) {
        // a pair of rows = six consecutive doubles: systems (0,1) (2,0) (1,2)
        while (i < n2) {
            const int64_t nx = i + stride;
            double2 an0, an1, an2, bn0, bn1, bn2;
            an0 = an1 = an2 = bn0 = bn1 = bn2 = make_double2(0., 0.);
            if (nx < n2) { an0 = r2[3 * nx]; an1 = r2[3 * nx + 1]; an2 = r2[3 * nx + 2]; bn0 = nu2[3 * nx]; bn1 = nu2[3 * nx + 1]; bn2 = nu2[3 * nx + 2]; }
            s2[3 * i] = make_double2(a0.x - alpha[0] * b0.x, a0.y - alpha[1] * b0.y);
            s2[3 * i + 1] = make_double2(a1.x - alpha[2] * b1.x, a1.y - alpha[0] * b1.y);
            s2[3 * i + 2] = make_double2(a2.x - alpha[1] * b2.x, a2.y - alpha[2] * b2.y);
            a0 = an0; a1 = an1; a2 = an2; b0 = bn0; b1 = bn1; b2 = bn2; i = nx;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) s3[3 * (n - 1) + s] = r3[3 * (n - 1) + s] - alpha[s] * nu3[3 * (n - 1) + s];
        }
    } else {  // a system broke down or is frozen: element by element, the others as usual
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (act[s]) s3[3 * e + s] = r3[3 * e + s] - alpha[s] * nu3[3 * e + s];
        }
    }
}

// bicg_xr_k for three systems.  partials: system s at partials + s * gridDim.x; fold: the product's sums, (t.s, t.t) of system s
// at fold + (2 s) * fold_count and fold + (2 s + 1) * fold_count
__global__ __launch_bounds__(kBlock) void bicg_xr3_k(double *__restrict__ scal3, int rho_idx, double *__restrict__ x3, const double *__restrict__ p3,
                                                     const double *__restrict__ s3, const double *__restrict__ t3, double *__restrict__ r3, int64_t n,
                                                     double *__restrict__ partials, int guard, const double *__restrict__ fold, int fold_count) {
    __shared__ double lds[8];
    __shared__ double lds16[6 * 16];
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double2 *x2 = reinterpret_cast<double2 *>(x3), *r2 = reinterpret_cast<double2 *>(r3);
    const double2 *p2 = reinterpret_cast<const double2 *>(p3), *s2 = reinterpret_cast<const double2 *>(s3), *t2 = reinterpret_cast<const double2 *>(t3);
    double alpha[3], omega[3];
    int state[3];  // 0 = normal, 1 = t = A s vanished or overflowed (x = h, r = s, stop), 2 = frozen (no-op)
    double folded[6] = {0., 0., 0., 0., 0., 0.};
    if (fold_count) fold_partials_multi<6>(fold, fold_count, lds16, folded);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const bool frz = guard && scal3[SC3(S_FROZEN, s)] != 0.;
        const double ts = fold_count ? folded[2 * s] : fold[2 * s];
        const double tt = fold_count ? folded[2 * s + 1] : fold[2 * s + 1];
        alpha[s] = scal3[SC3(rho_idx, s)] / scal3[SC3(S_SUM_NU, s)];
        omega[s] = ts / tt;
        const bool bad = guard && !(finite_nonzero(tt) && isfinite(omega[s]));
        state[s] = frz ? 2 : (bad ? 1 : 0);
        if (blockIdx.x == 0 && threadIdx.x == 0 && !frz) { scal3[SC3(S_TS, s)] = ts; scal3[SC3(S_TT, s)] = tt; }
    }
    double acc[3] = {0., 0., 0.};
    if (state[0] == 0 && state[1] == 0 && state[2] == 0) {
        int64_t i = i0;
        while (i < n2) {
            const double2 xa = x2[3 * i], xb = x2[3 * i + 1], xc = x2[3 * i + 2];
            const double2 pa = p2[3 * i], pb = p2[3 * i + 1], pc = p2[3 * i + 2];
            const double2 sa = s2[3 * i], sb = s2[3 * i + 1], sc = s2[3 * i + 2];
            const double2 ta = t2[3 * i], tb = t2[3 * i + 1], tc = t2[3 * i + 2];
            // row 2i: (xa.x, xa.y, xb.x) = systems 0, 1, 2; row 2i + 1: (xb.y, xc.x, xc.y)
            const double h00 = xa.x + alpha[0] * pa.x, h01 = xa.y + alpha[1] * pa.y, h02 = xb.x + alpha[2] * pb.x;
            const double h10 = xb.y + alpha[0] * pb.y, h11 = xc.x + alpha[1] * pc.x, h12 = xc.y + alpha[2] * pc.y;
            x2[3 * i] = make_double2(h00 + omega[0] * sa.x, h01 + omega[1] * sa.y);
            x2[3 * i + 1] = make_double2(h02 + omega[2] * sb.x, h10 + omega[0] * sb.y);
            x2[3 * i + 2] = make_double2(h11 + omega[1] * sc.x, h12 + omega[2] * sc.y);
            const double q00 = sa.x - omega[0] * ta.x, q01 = sa.y - omega[1] * ta.y, q02 = sb.x - omega[2] * tb.x;
            const double q10 = sb.y - omega[0] * tb.y, q11 = sc.x - omega[1] * tc.x, q12 = sc.y - omega[2] * tc.y;
            r2[3 * i] = make_double2(q00, q01);
            r2[3 * i + 1] = make_double2(q02, q10);
            r2[3 * i + 2] = make_double2(q11, q12);
            acc[0] += q00; acc[0] += q10;
            acc[1] += q01; acc[1] += q11;
            acc[2] += q02; acc[2] += q12;
            i += stride;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int64_t e = 3 * (n - 1) + s;
                const double h = x3[e] + alpha[s] * p3[e];
                const double si = s3[e];
                x3[e] = h + omega[s] * si;
                const double ri = si - omega[s] * t3[e];
                r3[e] = ri;
                acc[s] += ri;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (state[s] == 1) {  // bicg_xr_k's breakdown branch: one element per step of the grid-stride loop
                for (int64_t e = i0; e < n; e += stride) {
                    x3[3 * e + s] = x3[3 * e + s] + alpha[s] * p3[3 * e + s];
                    const double si = s3[3 * e + s];
                    r3[3 * e + s] = si;
                    acc[s] += si;
                }
                if (blockIdx.x == 0 && threadIdx.x == 0) scal3[SC3(S_FROZEN2, s)] = 1.;
            } else if (state[s] == 0) {  // bicg_xr_k's pair loop, this system's entries only
                for (int64_t i = i0; i < n2; i += stride) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int64_t e = 3 * (2 * i + h2) + s;
                        const double h = x3[e] + alpha[s] * p3[e];
                        const double si = s3[e];
                        x3[e] = h + omega[s] * si;
                        const double q = si - omega[s] * t3[e];
                        r3[e] = q;
                        acc[s] += q;
                    }
                }
                if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
                    const int64_t e = 3 * (n - 1) + s;
                    const double h = x3[e] + alpha[s] * p3[e];
                    const double si = s3[e];
                    x3[e] = h + omega[s] * si;
                    const double ri = si - omega[s] * t3[e];
                    r3[e] = ri;
                    acc[s] += ri;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double tsum = block_sum(acc[s], lds);
        if (threadIdx.x == 0 && state[s] != 2) partials[(size_t)s * gridDim.x + blockIdx.x] = tsum;
    }
}

// bicg_p_k for three systems; fold: bicg_xr3_k's partial sums, system s at fold + s * fold_count
__global__ __launch_bounds__(kBlock) void bicg_p3_k(double *__restrict__ scal3, int rho_prev_idx, int rho_idx, const double *__restrict__ r3,
                                                    const double *__restrict__ nu3, double *__restrict__ p3, int64_t n, int guard,
                                                    const double *__restrict__ fold, int fold_count) {
    __shared__ double lds16[3 * 16];
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    double2 *p2 = reinterpret_cast<double2 *>(p3);
    const double2 *r2 = reinterpret_cast<const double2 *>(r3), *nu2 = reinterpret_cast<const double2 *>(nu3);
    double beta[3], omega[3];
    bool act[3];
    double folded[3] = {0., 0., 0.};
    if (fold_count) fold_partials_multi<3>(fold, fold_count, lds16, folded);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const bool frz = bicg_frozen3(scal3, s, guard);
        const double rho = fold_count ? folded[s] : fold[s];
        const double rho_prev = scal3[SC3(rho_prev_idx, s)];
        const double alpha = rho_prev / scal3[SC3(S_SUM_NU, s)];
        omega[s] = scal3[SC3(S_TS, s)] / scal3[SC3(S_TT, s)];
        beta[s] = rho / rho_prev * alpha / omega[s];
        const bool bad = guard && !(finite_nonzero(omega[s]) && isfinite(beta[s]));
        act[s] = !frz && !bad;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (!frz) {
                scal3[SC3(rho_idx, s)] = rho;
                if (bad) scal3[SC3(S_FROZEN, s)] = 1.;
            } else if (scal3[SC3(S_FROZEN2, s)] != 0.) {
                scal3[SC3(S_FROZEN, s)] = 1.;  // see bicg_p_k
            }
        }
    }
    if (act[0] && act[1] && act[2]) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
            const double2 ra = r2[3 * i], rb = r2[3 * i + 1], rc = r2[3 * i + 2];
            const double2 pa = p2[3 * i], pb = p2[3 * i + 1], pc = p2[3 * i + 2];
            const double2 na = nu2[3 * i], nb = nu2[3 * i + 1], nc = nu2[3 * i + 2];
            p2[3 * i] = make_double2(ra.x + beta[0] * (pa.x - omega[0] * na.x), ra.y + beta[1] * (pa.y - omega[1] * na.y));
            p2[3 * i + 1] = make_double2(rb.x + beta[2] * (pb.x - omega[2] * nb.x), rb.y + beta[0] * (pb.y - omega[0] * nb.y));
            p2[3 * i + 2] = make_double2(rc.x + beta[1] * (pc.x - omega[1] * nc.x), rc.y + beta[2] * (pc.y - omega[2] * nc.y));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int64_t e = 3 * (n - 1) + s;
                p3[e] = r3[e] + beta[s] * (p3[e] - omega[s] * nu3[e]);
            }
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (act[s]) p3[3 * e + s] = r3[3 * e + s] + beta[s] * (p3[3 * e + s] - omega[s] * nu3[3 * e + s]);
        }
    }
}

__global__ void guard_event3_k(const double *__restrict__ scal3, int *__restrict__ counter) {
    int c = 0;
    for (int s = 0; s < 3; ++s)
        if (scal3[SC3(S_FROZEN, s)] != 0. || scal3[SC3(S_FROZEN2, s)] != 0.) ++c;
    if (c) atomicAdd(counter, c);
}

int jacobi_scaling_prepare3_dev(const MatView3 &A_in, uint64_t iteration_count, Arena &arena, ScaledOperator3 &S) {
    const int64_t n = A_in.P.n;
    S = ScaledOperator3();
    S.A = A_in;
    S.iterations = iteration_count;
    if (n == 0) return ORC_OK;
    double *dinv3;
    ORC_TRY(arena.alloc((size_t)3 * (size_t)n, &dinv3));
    ORC_TRY(diag_inverse3_dev(A_in, dinv3));
    if (!S.A.s1) S.A.s1 = dinv3;
    else if (!S.A.s2) S.A.s2 = dinv3;
    else return set_error(ORC_ERR_BAD_ARGUMENT, "more than two nested Jacobi scalings");
    S.dinv3 = dinv3;
    return materialize_scaled_view3(S.A, iteration_count, arena);
}

int bicgstab3_scaled_dev(const ScaledOperator3 &S, const double *b3, double *x3, Arena &arena) {
    const int64_t n = S.A.P.n;
    if (n == 0) return ORC_OK;
    ArenaScope scope(arena);
    double *bt3;
    ORC_TRY(arena.alloc((size_t)3 * (size_t)n, &bt3));
    ORC_TRY(scale_vec_dev(S.dinv3, b3, bt3, 3 * n));  // :165
    return bicgstab3_dev(S.A, bt3, x3, S.iterations, ORC_PRECOND_NONE, arena);
}

int bicgstab3_dev(const MatView3 &A_in, const double *b3_in, double *x3, uint64_t iteration_count, int preconditioner, Arena &arena) {
    const int64_t n = A_in.P.n;
    if (n == 0) return ORC_OK;
    if (!triple_supported()) return set_error(ORC_ERR_BAD_ARGUMENT, "three-system solve: tree reductions only");
    ArenaScope scope(arena);
    const size_t n3 = (size_t)3 * (size_t)n;
    // partitioned operator (A.halo): the vectors that ENTER a product (x3 — the caller's —, p3, s3) carry their ghost entries:
    // 3 * ncols doubles; the three sums of an iteration are folded by one-workgroup launches and summed over the ranks by one
    // all-reduce each (reduce_partials: 3, 6 and 3 scalars) instead of being folded by their consumers
    const bool part = A_in.halo != nullptr && ctx().world > 1;
    const size_t nc3 = (size_t)3 * (size_t)std::max<int64_t>(A_in.P.ncols, n);
    MatView3 A = A_in;
    const double *b3 = b3_in;
    if (preconditioner == ORC_PRECOND_JACOBI) {  // :159-167, as iterative_solve_body does it
        double *dinv3, *bt3;
        ORC_TRY(arena.alloc(n3, &dinv3));
        ORC_TRY(arena.alloc(n3, &bt3));
        ORC_TRY(diag_inverse3_dev(A_in, dinv3));
        ORC_TRY(scale_vec_dev(dinv3, b3_in, bt3, (int64_t)n3));
        if (!A.s1) A.s1 = dinv3;
        else if (!A.s2) A.s2 = dinv3;
        else return set_error(ORC_ERR_BAD_ARGUMENT, "more than two nested Jacobi scalings");
        b3 = bt3;
    } else if (preconditioner != ORC_PRECOND_NONE) {
        return set_error(ORC_ERR_BAD_ARGUMENT, "unknown preconditioner %d", preconditioner);
    }
    ORC_TRY(materialize_scaled_view3(A, iteration_count, arena));
    double *r3, *p3, *nu3, *s3, *t3, *partials, *partials2, *scal3;
    double *sums;  // partitioned: the folded and all-reduced sums of the launch before (6 doubles)
    ORC_TRY(arena.alloc(n3, &r3));
    ORC_TRY(arena.alloc(nc3, &p3));
    ORC_TRY(arena.alloc(n3, &nu3));
    ORC_TRY(arena.alloc(nc3, &s3));
    ORC_TRY(arena.alloc(n3, &t3));
    ORC_TRY(arena.alloc((size_t)6 * kMaxPartials, &partials));
    ORC_TRY(arena.alloc((size_t)3 * kMaxPartials, &partials2));
    ORC_TRY(arena.alloc((size_t)3 * S_COUNT, &scal3));
    ORC_TRY(arena.alloc((size_t)8, &sums));
    hipStream_t st = ctx().stream;
    ORC_HIP(hipMemsetAsync(scal3, 0, 3 * S_COUNT * sizeof(double), st));
    const int guard = ctx().breakdown_guard ? 1 : 0;
    const int vg = grid_for((n + 1) / 2);
    int g = 0;
    ORC_TRY(launch_spmv3(A, x3, EpiResidual3{b3, r3, p3}, partials, &g));  // r = b - A x ; p = r ; rho = sum(r)   (:250-254)
    ORC_TRY(reduce_partials(partials, g, 3, scal3 + SC3(S_RHO0, 0), part));
    for (uint64_t it = 0; it < iteration_count; ++it) {
        const int cur = (int)(it & 1), nxt = cur ^ 1;
        ORC_TRY(launch_spmv3(A, p3, EpiStoreSum3{nu3}, partials, &g));                                   // nu = A p, sum(nu)   (:256-257)
        if (part) ORC_TRY(reduce_partials(partials, g, 3, sums, true));                                  // C2: one all-reduce for the three systems
        hipLaunchKernelGGL(bicg_s3_k, dim3(vg), dim3(kBlock), 0, st, scal3, S_RHO0 + cur, (const double *)r3, (const double *)nu3, s3, n, guard,
                           part ? (const double *)sums : (const double *)partials, part ? 0 : g);        // s = r - alpha nu    (:259)
        ORC_TRY(launch_spmv3(A, s3, EpiTs3{s3, t3}, partials, &g));                                      // t = A s, t.s, t.t   (:260-261)
        if (part) ORC_TRY(reduce_partials(partials, g, 6, sums, true));
        hipLaunchKernelGGL(bicg_xr3_k, dim3(vg), dim3(kBlock), 0, st, scal3, S_RHO0 + cur, x3, (const double *)p3, (const double *)s3, (const double *)t3, r3,
                           n, partials2, guard, part ? (const double *)sums : (const double *)partials, part ? 0 : g);  // x, r, sum(r)  (:258, :262-265)
        if (part) ORC_TRY(reduce_partials(partials2, vg, 3, sums, true));
        hipLaunchKernelGGL(bicg_p3_k, dim3(vg), dim3(kBlock), 0, st, scal3, S_RHO0 + cur, S_RHO0 + nxt, (const double *)r3, (const double *)nu3, p3, n,
                           guard, part ? (const double *)sums : (const double *)partials2, part ? 0 : vg);  // p                   (:266-267)
    }
    ORC_HIP(hipGetLastError());
    if (guard && ctx().guard_events) {
        hipLaunchKernelGGL(guard_event3_k, dim3(1), dim3(1), 0, st, (const double *)scal3, ctx().guard_events);
        ORC_HIP(hipGetLastError());
    }
    return ORC_OK;
}
#undef SC3
int bench_inloop_products3_dev(const MatView3 &A, const double *x3, double *y3, double *partials, int reps, float ms[2]) {
    int g = 0;
    ORC_TRY(time_launches(reps, &ms[0], [&] { return launch_spmv3(A, x3, EpiStoreSum3{y3}, partials, &g); }));
    ORC_TRY(time_launches(reps, &ms[1], [&] { return launch_spmv3(A, x3, EpiTs3{x3, y3}, partials, &g); }));
    return ORC_OK;
}

// ------------------------------------------------------------------ Jacobi arm (linear_algebra.rs:172-218)
struct JacobiCtrl {
    int done;           // convergence break taken (:210-213)
    int status;         // sticky OrcStatus
    long long sweeps;   // sweeps executed
    double initial_residual;
    long long iter_num;
};

// x_new = omega * (b'_i - sum_j a'_ij x_j) + x_i (1 - omega), a' = offdiag(A)/diag(A), b' = b/diag(A);
// also flags NaN in the incoming x (:192-196)
__global__ __launch_bounds__(kBlock) void jacobi_sweep_k(MatView A, const double *__restrict__ b, const double *__restrict__ x,
                                                         double *__restrict__ x_new, double omega, JacobiCtrl *ctrl) {
    if (ctrl->done || ctrl->status) return;
    const int lane = threadIdx.x & 63;
    int saw_nan = 0;
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        double aii = 1.;
        if (live) {
            const int32_t d = A.P.diag_pos[row];
            if (d < 0) { atomicCAS(&ctrl->status, 0, (int)ORC_ERR_STRUCTURAL_ZERO); aii = 1.; }
            else aii = view_value(A, row, d);
        }
        double acc = 0.;
        for (int k = 0; k < width; ++k) {
            if (k < len) {
                const int64_t pos = base + (int64_t)k * 64 + lane;
                const int c = A.P.col[pos];
                const double v = (c == row) ? 0. : view_value(A, row, pos) / aii;  // :174-180
                acc += v * x[c];
            }
        }
        if (live) {
            const double xi = x[row];
            if (xi != xi) saw_nan = 1;
            const double bp = b[row] / aii;  // :181-187
            x_new[row] = omega * (bp - acc) + xi * (1. - omega);  // :199-200
        }
    }
    if (saw_nan) atomicCAS(&ctrl->status, 0, (int)ORC_ERR_JACOBI_NAN);
}

// partial sum((b - A x)^2) and max |x|  (:202-207)
__global__ __launch_bounds__(kBlock) void jacobi_residual_k(MatView A, const double *__restrict__ b, const double *__restrict__ x,
                                                            double *__restrict__ partials, JacobiCtrl *ctrl) {
    __shared__ double lds[8];
    if (ctrl->done || ctrl->status) return;
    const int lane = threadIdx.x & 63;
    double r2 = 0., mx = 0.;
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        double acc = 0.;
        for (int k = 0; k < width; ++k) {
            if (k < len) {
                const int64_t pos = base + (int64_t)k * 64 + lane;
                acc += view_value(A, row, pos) * x[A.P.col[pos]];
            }
        }
        if (live) {
            const double v = b[row] - acc;
            r2 += v * v;
            mx = max_nan(mx, fabs(x[row]));
        }
    }
    const double t = block_sum(r2, lds);
    const double m = block_max(mx, lds);
    if (threadIdx.x == 0) { partials[blockIdx.x] = t; partials[gridDim.x + blockIdx.x] = m; }
}

// fold the per-workgroup maxima (second partial array of jacobi_residual_k)
__global__ __launch_bounds__(1024) void reduce_max_k(const double *__restrict__ partials, int count, double *__restrict__ out) {
    __shared__ double lds[16];
    double v = 0.;
    for (int i = threadIdx.x; i < count; i += blockDim.x) v = max_nan(v, partials[i]);
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = max_nan(r, lds[i]);
        out[0] = r;
    }
}

// one thread: the reference's per-sweep bookkeeping (:208-216); red[0] = sum((b - A x)^2), red[1] = max |x| — NaN when x
// holds one: max_by(total_cmp) (:203-207) ranks NaN above everything, `NaN > 1e10` is false, and the next sweep's
// NaN check (:192-196) is what panics
__global__ void jacobi_control_k(const double *__restrict__ red, double threshold, JacobiCtrl *ctrl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctrl->done || ctrl->status) return;
    const double r = sqrt(red[0]), mx = red[1];
    ctrl->sweeps += 1;
    const long long it = ctrl->iter_num;
    ctrl->iter_num = it + 1;
    if (it == 1) ctrl->initial_residual = r;
    else if (r / ctrl->initial_residual < threshold) { ctrl->done = 1; return; }
    if (mx > 1e10) ctrl->status = (int)ORC_ERR_JACOBI_TOO_LARGE;
}

static int jacobi_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor,
                      double threshold, Arena &arena, SolveStats *stats, int *status_out) {
    const int64_t n = A.P.n;
    *status_out = ORC_OK;
    if (n == 0 || iteration_count == 0) return ORC_OK;
    ArenaScope scope(arena);
    double *x2, *partials, *red;
    JacobiCtrl *ctrl;
    const bool global = A.halo != nullptr;
    ORC_TRY(arena.alloc((size_t)std::max(A.P.ncols, n), &x2));
    ORC_TRY(arena.alloc((size_t)2 * kMaxPartials, &partials));
    ORC_TRY(arena.alloc((size_t)2, &red));
    ORC_TRY(arena.alloc((size_t)1, &ctrl));
    ORC_HIP(hipMemsetAsync(ctrl, 0, sizeof(JacobiCtrl), ctx().stream));
    const bool ref = reference_order(A);
    double *rvec = nullptr, *ref_partials = nullptr;
    if (ref) {
        ORC_TRY(arena.alloc((size_t)n, &rvec));
        ORC_TRY(arena.alloc((size_t)kMaxPartials, &ref_partials));
    }
    const int g = spmv_grid(A.P.n_slices);
    // Sweeps alternate x -> x2 -> x.  A sweep that is skipped (done/status set) leaves both
    // buffers untouched, so the newest iterate is in x2 iff the executed sweep count is odd.
    double *cur = x, *nxt = x2;
    for (uint64_t it = 0; it < iteration_count; ++it) {
        if (global) ORC_TRY(A.halo->exchange(cur));
        hipLaunchKernelGGL(jacobi_sweep_k, dim3(g), dim3(kBlock), 0, ctx().stream, A, b, cur, nxt, relaxation_factor, ctrl);
        if (global) ORC_TRY(A.halo->exchange(nxt));
        hipLaunchKernelGGL(jacobi_residual_k, dim3(g), dim3(kBlock), 0, ctx().stream, A, b, nxt, partials, ctrl);
        ORC_TRY(reduce_partials(partials, g, 1, red, global));
        if (ref) {  // |b - A x|^2 in nalgebra's association (:202); a sweep past the break recomputes a value nobody reads
            int g2 = 0;
            ORC_TRY(launch_spmv(A, nxt, EpiResidualNorm{b, rvec}, ref_partials, &g2));
            ORC_TRY(dot_reference(rvec, rvec, n, red, nullptr));
        }
        hipLaunchKernelGGL(reduce_max_k, dim3(1), dim3(1024), 0, ctx().stream, partials + g, g, red + 1);
        if (global) ORC_TRY(comm_allreduce_max(red + 1, 1));
        hipLaunchKernelGGL(jacobi_control_k, dim3(1), dim3(1), 0, ctx().stream, red, threshold, ctrl);
        std::swap(cur, nxt);
    }
    ORC_HIP(hipGetLastError());
    JacobiCtrl h;
    ORC_HIP(hipMemcpyAsync(&h, ctrl, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
    ORC_HIP(hipStreamSynchronize(ctx().stream));
    // the NaN check of the reference runs at the top of a sweep: a NaN seen by sweep k means
    // sweep k itself was still executed by the kernel above, but the reference panics before it.
    // Either way the call fails with "diverged"; the iterate is not observable after a panic.
    if (stats) stats->jacobi_sweeps = h.sweeps;
    // sweeps executed = h.sweeps, except that a sweep launched after a status was raised inside
    // jacobi_sweep_k (NaN / structural zero) has no matching control step.
    const bool newest_in_x2 = (h.sweeps & 1) != 0;
    if (newest_in_x2) ORC_TRY(vec_copy(x, x2, n));
    *status_out = h.status;
    return ORC_OK;
}

// ------------------------------------------------------------------ iterative_solve (linear_algebra.rs:144-299)
int multigrid_arm_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor,
                      double convergence_threshold, int preconditioner, Arena &arena, SolveStats *stats, int smoother);  // amg.hip
int gs_arm_dev(const MatView &A, const double *b, double *x, uint64_t iteration_count, double relaxation_factor, int method,
               Arena &arena);  // gs.hip (extension)

static int iterative_solve_body(const MatView &A_in, const double *b_in, double *x, uint64_t iteration_count, int method,
                                double relaxation_factor, double convergence_threshold, int preconditioner, Arena &arena,
                                SolveStats *stats) {
    const int64_t n = A_in.P.n;
    ArenaScope scope(arena);
    MatView A = A_in;
    const double *b = b_in;
    if (preconditioner == ORC_PRECOND_JACOBI) {  // :159-167
        double *dinv, *b_tmp;
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &dinv));
        ORC_TRY(arena.alloc((size_t)std::max<int64_t>(n, 1), &b_tmp));
        if (n) {
            hipLaunchKernelGGL(diag_inverse_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, A_in, dinv);
            hipLaunchKernelGGL(scale_vec_k, dim3(grid_for(n)), dim3(kBlock), 0, ctx().stream, dinv, b_in, b_tmp, n);
            ORC_HIP(hipGetLastError());
        }
        if (!A.s1) A.s1 = dinv;
        else if (!A.s2) A.s2 = dinv;
        else return set_error(ORC_ERR_BAD_ARGUMENT, "more than two nested Jacobi scalings");
        b = b_tmp;
    } else if (preconditioner != ORC_PRECOND_NONE) {
        return set_error(ORC_ERR_BAD_ARGUMENT, "unknown preconditioner %d", preconditioner);
    }
    int st = ORC_OK;
    switch (method) {
    case ORC_SOLVER_JACOBI: {
        int jst = ORC_OK;
        st = jacobi_dev(A, b, x, iteration_count, relaxation_factor, convergence_threshold, arena, stats, &jst);
        if (st == ORC_OK) st = jst;
        break;
    }
    case ORC_SOLVER_BICGSTAB:
        st = bicgstab_dev(A, b, x, iteration_count, arena);
        break;
    case ORC_SOLVER_MULTIGRID:
        st = multigrid_arm_dev(A, b, x, iteration_count, relaxation_factor, convergence_threshold, preconditioner, arena, stats, ORC_SOLVER_BICGSTAB);
        break;
    case ORC_SOLVER_MULTIGRID_GS:
        st = multigrid_arm_dev(A, b, x, iteration_count, relaxation_factor, convergence_threshold, preconditioner, arena, stats, ORC_SOLVER_MULTICOLOR_GS);
        break;
    case ORC_SOLVER_MULTICOLOR_GS:
    case ORC_SOLVER_BICGSTAB_GS_PRECOND:
        st = gs_arm_dev(A, b, x, iteration_count, relaxation_factor, method, arena);
        break;
    case ORC_SOLVER_GAUSS_SEIDEL:
        // The reference's arm scans every (i, j) through get(), which panics on the first
        // structural zero of a sparse matrix, and otherwise ends in
        // panic!("Gauss-Seidel out for maintenance :)") (linear_algebra.rs:219-246).
        st = ORC_ERR_GS_MAINTENANCE;
        break;
    default:
        st = ORC_ERR_UNSUPPORTED_SOLVER;  // :297
    }
    return st;
}

int iterative_solve_dev(const MatView &A_in, const double *b_in, double *x, uint64_t iteration_count, int method,
                        double relaxation_factor, double convergence_threshold, int preconditioner, Arena &arena,
                        SolveStats *stats) {
    int st = ensure_init();
    if (st == ORC_OK)
        st = iterative_solve_body(A_in, b_in, x, iteration_count, method, relaxation_factor, convergence_threshold, preconditioner, arena, stats);
    // partitioned operator: whatever happened locally (an early error return included), every rank takes part in the
    // status agreement and leaves with the same verdict — a rank that skipped it would strand its peers in RCCL
    if (A_in.halo) st = comm_global_status(st);
    return st;
}

}  // namespace orc
