// linalg_kernels.hpp — device-side building blocks shared by linalg.hip and amg.hip.
#pragma once
#include "linalg.hpp"

namespace orc {

__device__ __forceinline__ double view_value(const MatView &A, int64_t row, int64_t pos) {
    double v = A.val[pos];
    if (A.s1) v = A.s1[row] * v;
    if (A.s2) v = A.s2[row] * v;
    return v;
}

// XCD-aware slice walk: workgroups b and b+8 share an XCD (MI355X_MICROARCH "Workgroup dispatch"),
// so XCD g = blockIdx%8 sweeps the contiguous slice range [g*spx, (g+1)*spx): the x-vector
// window a row block needs (i+-1, i+-nx, i+-nx*ny) then stays inside one XCD's 4 MiB L2 instead
// of being fetched by all eight.  Pure speed: any placement gives the same result.
struct SliceWalk {
    int64_t begin, end, step;
    __device__ __forceinline__ SliceWalk(int32_t n_slices) {
        const int waves = blockDim.x >> 6;
        const int wave = threadIdx.x >> 6;
        if ((gridDim.x & 7) == 0 && gridDim.x >= 8) {
            const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nb = gridDim.x >> 3;
            const int64_t spx = ((int64_t)n_slices + 7) / 8;
            const int64_t lo = (int64_t)xcd * spx;
            int64_t hi = lo + spx;
            if (hi > n_slices) hi = n_slices;
            begin = lo + (int64_t)bl * waves + wave;
            end = hi;
            step = (int64_t)nb * waves;
        } else {
            begin = (int64_t)blockIdx.x * waves + wave;
            end = n_slices;
            step = (int64_t)gridDim.x * waves;
        }
    }
};

// Generic thread-per-row SELL-64 SpMV.  Epi::apply(row, acc, r0, r1) consumes the row result and
// may accumulate up to two per-thread reduction terms; partial sums per workgroup go to
// partials[q * gridDim.x + blockIdx.x] and are folded by reduce_partials_k.
template <class Epi, bool kRagged = false>
__global__ __launch_bounds__(kBlock) void spmv_k(MatView A, const double *__restrict__ x, Epi epi, double *__restrict__ partials,
                                                 const double *__restrict__ skip_flags /* 2 doubles or null: non-zero = no-op */) {
    __shared__ double lds[8];
    if (skip_flags && (skip_flags[0] != 0. || skip_flags[1] != 0.)) return;
    const int lane = threadIdx.x & 63;
    double r0 = 0., r1 = 0.;
    SliceWalk w(A.P.n_slices);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        const double s1 = (A.s1 && live) ? A.s1[row] : 1.;
        const double s2 = (A.s2 && live) ? A.s2[row] : 1.;
        double acc = 0.;
        // chunks of 8 entries: all column and value loads are issued first, then the dependent x gathers, then the
        // products are added in ascending k — the association of the CPU product, just with the loads in flight together.
        // Padding slots hold a valid column (the row itself) and are masked out of the sum.
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[8], xv[8];
            const int64_t p0 = base + (int64_t)k0 * 64 + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // kRagged (coarse AMG levels, 11-49 % padding): padding slots are not fetched, so a cache line whose lanes
                // are all past their rows' ends stays in HBM; otherwise the loads stay wave-uniform (cheaper to issue)
                const bool in = k0 + u < (kRagged ? len : width);
                c[u] = in ? A.P.col[p0 + (int64_t)u * 64] : 0;
                v[u] = in ? A.val[p0 + (int64_t)u * 64] : 0.;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = (k0 + u < len) ? x[c[u]] : 0.;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < len) {
                    double t = v[u];
                    if (A.s1) t = s1 * t;
                    if (A.s2) t = s2 * t;
                    acc += t * xv[u];
                }
            }
        }
        if (live) epi.apply(row, acc, r0, r1);
    }
    if (Epi::kReductions > 0) {
        double t = block_sum(r0, lds);
        if (threadIdx.x == 0) partials[blockIdx.x] = t;
    }
    if (Epi::kReductions > 1) {
        double t = block_sum(r1, lds);
        if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = t;
    }
}

// Folds nq partial arrays of `count` entries each (fixed order => reproducible) into out[q].
// One workgroup; launched after every kernel that produces partials.  In a multi-GPU run the
// caller follows it with an RCCL all-reduce of out[0..nq).
__global__ __launch_bounds__(1024) void reduce_partials_k(const double *__restrict__ partials, int count, int nq, double *__restrict__ out);

int reduce_partials(const double *partials, int count, int nq, double *out, bool global = false);

}  // namespace orc
