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

// The matrix streams of a product (values, columns, window positions) are read once per launch and are many times the size of
// the caches: loaded with the non-temporal hint they leave L2 and the Infinity Cache to the vectors.  Measured on the level-0
// shape (scripts/microbench/stream_floor.hip, 10.24 M rows): plain read 5.2 -> 5.6-6.0 TB/s, value stream 5.05 -> 5.54 TB/s, the
// whole product 215 -> 194 us.
// A matrix that fits the Infinity Cache (a 1 M-cell mesh: 62 MB) is served from it launch after launch and loses with the hint
// (in-loop product 0.65 -> 0.60 of peak at 1.03 M cells): the launch decides (MatView::nt, launch_spmv) between two instantiations
// (a scalar branch per chunk inside ONE kernel cost 3 % — it fences the scheduler).
template <bool NT, class T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// The same fold done by EVERY workgroup of the kernel that consumes the sum (256 threads): four passes over
// reduce_partials_k's 16 virtual wavefronts, so the association — and therefore every bit — is that of the one-workgroup
// kernel.  A BiCGSTAB iteration has three such sums; as separate one-workgroup launches they sit between the big kernels
// of their stream and, when other streams fill the chip, wait for a slot each time (84 us on average in the concurrent
// schedule against 4.8 us alone).  Returns the sum to every thread.
__device__ __forceinline__ double fold_partials_block(const double *__restrict__ partials, int count, double *lds16 /* 16 doubles */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // blockDim.x == 256
    double a[4], b[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // virtual thread vt of 1024 adds partials[vt] and partials[vt + 1024] (count <= 2048)
        const int vt = (w + 4 * p) * 64 + lane;
        a[p] = vt < count ? partials[vt] : 0.;
        b[p] = vt + 1024 < count ? partials[vt + 1024] : 0.;
    }
    __syncthreads();  // lds16 may still be read from a previous fold
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int vt = (w + 4 * p) * 64 + lane;
        double v = 0.;
        if (vt < count) v += a[p];
        if (vt + 1024 < count) v += b[p];
        v = wave_sum(v);
        if (lane == 0) lds16[w + 4 * p] = v;
    }
    __syncthreads();
    double r = 0.;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += lds16[i];
    return r;
}

// [r04] NQ such folds at once: quantity q's partial sums at partials + q * count.  All loads are issued first and the workgroup meets
// at two barriers instead of 2 NQ — a three-system kernel folds 3 or 6 sums before it can start (six dependent round trips and twelve
// barriers in a row were 10 % of a 1 M-cell vector kernel).  Per quantity the additions are fold_partials_block's: the same bits.
template <int NQ>
__device__ __forceinline__ void fold_partials_multi(const double *__restrict__ partials, int count, double *lds /* NQ * 16 doubles */, double (&out)[NQ]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // blockDim.x == 256
    double a[NQ][4], b[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int vt = (w + 4 * p) * 64 + lane;
            a[q][p] = vt < count ? partials[(size_t)q * count + vt] : 0.;
            b[q][p] = vt + 1024 < count ? partials[(size_t)q * count + vt + 1024] : 0.;
        }
    }
    __syncthreads();  // lds may still be read from a previous fold
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int vt = (w + 4 * p) * 64 + lane;
            double v = 0.;
            if (vt < count) v += a[q][p];
            if (vt + 1024 < count) v += b[q][p];
            v = wave_sum(v);
            if (lane == 0) lds[q * 16 + w + 4 * p] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double r = 0.;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += lds[q * 16 + i];
        out[q] = r;
    }
}

// XCD-aware slice walk: workgroups b and b+8 share an XCD (MI355X_MICROARCH "Workgroup dispatch"),
// so XCD g = blockIdx%8 sweeps the contiguous slice range [g*spx, (g+1)*spx): the x-vector
// window a row block needs (i+-1, i+-nx, i+-nx*ny) then stays inside one XCD's 4 MiB L2 instead
// of being fetched by all eight.  Pure speed: any placement gives the same result.
struct SliceWalk {
    int64_t begin, end, step;
    // block / grid: the workgroup this walk belongs to — the launch's own (default) or a VIRTUAL one: a launch of fewer, resident workgroups walks
    // the shares of a larger grid one after the other and writes that grid's partial sums (spmv3_uniform_k: MatView3::vgrid)
    __device__ __forceinline__ SliceWalk(int32_t n_slices, int32_t first = 0) : SliceWalk(n_slices, first, (int)blockIdx.x, (int)gridDim.x) {}
    __device__ __forceinline__ SliceWalk(int32_t n_slices, int32_t first, int block, int grid) {  // slices [first, n_slices)
        const int waves = blockDim.x >> 6;
        // (the wavefront's index is the same in all its lanes; said so explicitly, the slice index, the slice's base and width and the narrow
        // column image's per-depth bases live in scalar registers and are fetched through the scalar cache instead of by 64-lane vector loads)
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if ((grid & 7) == 0 && grid >= 8) {
            const int xcd = block & 7, bl = block >> 3, nb = grid >> 3;
            const int64_t spx = ((int64_t)(n_slices - first) + 7) / 8;
            const int64_t lo = (int64_t)first + (int64_t)xcd * spx;
            int64_t hi = lo + spx;
            if (hi > n_slices) hi = n_slices;
            begin = lo + (int64_t)bl * waves + wave;
            end = hi;
            step = (int64_t)nb * waves;
        } else {
            begin = (int64_t)first + (int64_t)block * waves + wave;
            end = n_slices;
            step = (int64_t)grid * waves;
        }
    }
};

// Generic thread-per-row SELL-64 SpMV.  Epi::apply(row, acc, r0, r1) consumes the row result and
// may accumulate up to two per-thread reduction terms; partial sums per workgroup go to
// partials[q * gridDim.x + blockIdx.x] and are folded by reduce_partials_k.
// kLayout: 0 = padded SELL-64, wave-uniform loads (mesh-pattern matrices, < 8 % padding);
//          1 = padded SELL-64, padding slots not fetched (ragged matrices without a packed mirror);
//          2 = packed mirror (PackedDev): no padding in memory at all.
enum { kSpmvPlain = 0, kSpmvRagged = 1, kSpmvPacked = 2 };
// kNoGather (diagnostic, orc_debug_set_spmv_variant): x[row] instead of x[col] — the matrix stream without the gathers
// kGuard: slots at or beyond the slice width are skipped by wave-uniform branches instead of re-reading the last slot
template <class Epi, int kLayout = kSpmvPlain, bool kNoGather = false, bool kGuard = false>
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
        // Every load below is UNCONDITIONAL and branch-free: a lane past its row's end re-reads an entry that is being
        // read anyway (same cache line or its own previous entry) and its product is dropped by a select.  Per-lane
        // predicated loads compile into one exec-masked branch each with an s_waitcnt behind every column load, i.e.
        // eight serialised round trips per chunk — that, not padding or the x gathers, is what held the ragged coarse
        // levels at 36-48 % of peak.
        int64_t pk_off = kLayout == kSpmvPacked ? A.pk.ptr[slice] : 0;  // wave-uniform running offset of depth k0
        const int last = (kLayout == kSpmvRagged ? (len > 0 ? len : 1) : width) - 1;  // deepest slot this lane may touch
        // lanes past the last row of the matrix (last slice only) re-read the last row's slots: their own were never
        // written by the device-side pack kernels, and an unconditional gather through a garbage column would fault
        const int lane_c = live ? lane : (int)((A.P.n - 1) & 63);
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kLayout == kSpmvPacked) {
                    // entries of depth k0 + u: one per lane whose row is long enough, back to back in lane order
                    const bool in = k0 + u < len;
                    const unsigned long long m = __ballot(in);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    const int64_t p = in ? pk_off + rank : (m != 0ull ? pk_off : pk_off - 1);  // idle lanes: a stored neighbour
                    c[u] = A.pk.col[p];
                    v[u] = A.pk.val[p];
                    pk_off += __popcll(m);
                } else if (kGuard) {
                    if (k0 + u < width) {  // wave-uniform
                        // padded layout: the slot exists for every lane, so its address is the chunk base plus a compile-time
                        // offset (folded into the load instruction); only the predicated layout has to clamp per lane
                        const int kk = kLayout == kSpmvPlain ? k0 + u : (k0 + u < last ? k0 + u : last);
                        const int64_t p = kLayout == kSpmvPlain ? (base + (int64_t)k0 * 64 + lane_c) + (int64_t)u * 64 : base + (int64_t)kk * 64 + lane_c;
                        c[u] = A.P.col[p];
                        v[u] = A.val[p];
                    } else { c[u] = 0; v[u] = 0.; }
                } else {
                    const int kk = k0 + u < last ? k0 + u : last;
                    const int64_t p = base + (int64_t)kk * 64 + lane_c;
                    c[u] = A.P.col[p];
                    v[u] = A.val[p];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kGuard) { if (k0 + u < width) xv[u] = x[c[u]]; else xv[u] = 0.; }
                else xv[u] = x[kNoGather ? (int)(live ? row : 0) + (c[u] & 0) : c[u]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double t = v[u];
                if (A.s1) t = s1 * t;
                if (A.s2) t = s2 * t;
                const double next = acc + t * xv[u];
                acc = (k0 + u < len) ? next : acc;
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

// Product kernel for matrices with (nearly) uniform row lengths — the mesh-pattern matrices (width 5-7, 0.1 % padding) and the
// first coarse level (15 entries per row, 6 % padding): wave-uniform matrix loads at compile-time offsets from one chunk base,
// per-lane predicated gathers.  Measured 4-5 % faster on those than the clamped branch-free spmv_k below, which wins where
// rows are long and ragged (33-70 entries per row: +7-10 %), where this kernel's kRagged form serialises its loads.
// Generic thread-per-row SELL-64 SpMV.  Epi::apply(row, acc, r0, r1) consumes the row result and
// may accumulate up to two per-thread reduction terms; partial sums per workgroup go to
// partials[q * gridDim.x + blockIdx.x] and are folded by reduce_partials_k.
// kMesh: a tag, no code — products on the mesh pattern (level 0: a_u, a_v, a_w, A_p) get a kernel name of their own, so that a
// kernel trace separates them from the first coarse level's (profiles/: the roofline of bench.py is about level 0).
// kNarrow: the pattern has a narrow column image for EVERY slice (SellDev::col16) and the product streams that; kScaled = false:
// the view carries no row scaling (materialised), so neither the scaling vectors nor the multiplications are compiled in.
// Both keep the kernel's scalar-register count under 81 — at 81-96 only seven wavefronts per SIMD are resident, and a launch of
// eight workgroups per CU then runs a second, nearly empty round (measured: 213 -> 273 us).
// kNT: the matrix streams are loaded with the non-temporal hint (launch_spmv decides by the size of the stream).
template <class Epi, bool kRagged = false, bool kMesh = false, bool kNarrow = false, bool kScaled = true, bool kNT = false>
__global__ __launch_bounds__(kBlock) void spmv_uniform_k(MatView A, const double *__restrict__ x, Epi epi, double *__restrict__ partials,
                                                 const double *__restrict__ skip_flags /* 2 doubles or null: non-zero = no-op */) {
    __shared__ double lds[8];
    if (skip_flags && (skip_flags[0] != 0. || skip_flags[1] != 0.)) return;
    const int lane = threadIdx.x & 63;
    double r0 = 0., r1 = 0.;
    SliceWalk w(A.slice_hi >= 0 ? A.slice_hi : A.P.n_slices, A.slice_lo);
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        const double s1 = (kScaled && A.s1 && live) ? A.s1[row] : 1.;
        const double s2 = (kScaled && A.s2 && live) ? A.s2[row] : 1.;
        double acc = 0.;
        // chunks of 8 entries: all column and value loads are issued first, then the dependent x gathers, then the
        // products are added in ascending k — the association of the CPU product, just with the loads in flight together.
        // Padding slots hold a valid column (the row itself) and are masked out of the sum.
        const int32_t *__restrict__ cb = kNarrow ? A.P.colbase + (base >> 6) : nullptr;  // narrow column image: the 32-bit columns are never touched
        for (int k0 = 0; k0 < width; k0 += 8) {
            int c[8];
            double v[8], xv[8];
            const int64_t p0 = base + (int64_t)k0 * 64 + lane;
            if (kNarrow) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = k0 + u < width;
                    c[u] = in ? cb[k0 + u] + (int)ld_stream<kNT>(A.P.col16 + p0 + (int64_t)u * 64) : 0;
                    v[u] = in ? ld_stream<kNT>(A.val + p0 + (int64_t)u * 64) : 0.;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // kRagged (coarse AMG levels, 11-49 % padding): padding slots are not fetched, so a cache line whose lanes
                    // are all past their rows' ends stays in HBM; otherwise the loads stay wave-uniform (cheaper to issue)
                    const bool in = k0 + u < (kRagged ? len : width);
                    c[u] = in ? ld_stream<kNT>(A.P.col + p0 + (int64_t)u * 64) : 0;
                    v[u] = in ? ld_stream<kNT>(A.val + p0 + (int64_t)u * 64) : 0.;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = (k0 + u < len) ? x[c[u]] : 0.;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < len) {
                    double t = v[u];
                    if (kScaled && A.s1) t = s1 * t;
                    if (kScaled && A.s2) t = s2 * t;
                    acc += t * xv[u];
                }
            }
        }
        if (live) epi.apply(row, acc, r0, r1);
    }
    const int pstride = A.part_stride > 0 ? A.part_stride : (int)gridDim.x;
    if (Epi::kReductions > 0) {
        double t = block_sum(r0, lds);
        if (threadIdx.x == 0) partials[A.part_base + blockIdx.x] = t;
    }
    if (Epi::kReductions > 1) {
        double t = block_sum(r1, lds);
        if (threadIdx.x == 0) partials[pstride + A.part_base + blockIdx.x] = t;
    }
}


// Three systems on one pattern (MatView3): one column load and one 24-byte gather per entry serve three value streams.
// Epi3::apply(row, acc[3], r[3][2]) consumes the three row results; partial sums of reduction q of system s go to
// partials[(s * Epi3::kReductions + q) * gridDim.x + blockIdx.x].  Same grid, same SliceWalk and same per-thread order as
// spmv_uniform_k, so every system's row sums AND partial sums are those of its own one-system product.
struct __attribute__((aligned(8))) Vec3d { double a, b, c; };

template <class Epi3, int kChunk = 4, bool kMesh = false, bool kNarrow = false, bool kScaled = true, bool kNT = false>
__global__ __launch_bounds__(kBlock) void spmv3_uniform_k(MatView3 A, const double *__restrict__ x3, Epi3 epi, double *__restrict__ partials) {
    __shared__ double lds[8];
    const int lane = threadIdx.x & 63;
    const Vec3d *__restrict__ xv3 = reinterpret_cast<const Vec3d *>(x3);
    const Vec3d *__restrict__ s1v = reinterpret_cast<const Vec3d *>(A.s1);
    const Vec3d *__restrict__ s2v = reinterpret_cast<const Vec3d *>(A.s2);
    // (the walk is written for a VIRTUAL grid — r04 measured fewer, resident workgroups walking the one-system grid's shares: the product
    // alone gained 3 %, the iteration lost 14 ms, HISTORY.md — and runs with the launch's own grid: one share per workgroup)
    const int vgrid = (int)gridDim.x;
    for (int vb = blockIdx.x; vb < vgrid; vb += gridDim.x) {
    double red[3][2] = {{0., 0.}, {0., 0.}, {0., 0.}};
    SliceWalk w(A.slice_hi >= 0 ? A.slice_hi : A.P.n_slices, A.slice_lo, vb, vgrid);  // a slice range when the product overlaps its halo exchange (launch_spmv3)
    for (int64_t slice = w.begin; slice < w.end; slice += w.step) {
        const int64_t row = slice * 64 + lane;
        const int64_t base = A.P.slice_ptr[slice];
        const int width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        const bool live = row < A.P.n;
        const int len = live ? A.P.row_len[row] : 0;
        Vec3d s1 = {1., 1., 1.}, s2 = {1., 1., 1.};
        if (kScaled && A.s1 && live) s1 = s1v[row];
        if (kScaled && A.s2 && live) s2 = s2v[row];
        double acc0 = 0., acc1 = 0., acc2 = 0.;
        const int32_t *__restrict__ cb = kNarrow ? A.P.colbase + (base >> 6) : nullptr;  // narrow column image (SellDev)
        for (int k0 = 0; k0 < width; k0 += kChunk) {
            int c[kChunk];
            double v0[kChunk], v1[kChunk], v2[kChunk];
            Vec3d xv[kChunk];
            const int64_t p0 = base + (int64_t)k0 * 64 + lane;
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {  // wave-uniform matrix loads at compile-time offsets from one chunk base
                const bool in = k0 + u < width;
                if (kNarrow) c[u] = in ? cb[k0 + u] + (int)ld_stream<kNT>(A.P.col16 + p0 + (int64_t)u * 64) : 0;
                else c[u] = in ? ld_stream<kNT>(A.P.col + p0 + (int64_t)u * 64) : 0;
                v0[u] = in ? ld_stream<kNT>(A.val[0] + p0 + (int64_t)u * 64) : 0.;
                v1[u] = in ? ld_stream<kNT>(A.val[1] + p0 + (int64_t)u * 64) : 0.;
                v2[u] = in ? ld_stream<kNT>(A.val[2] + p0 + (int64_t)u * 64) : 0.;
            }
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
                if (k0 + u < len) xv[u] = xv3[c[u]];
                else xv[u] = Vec3d{0., 0., 0.};
            }
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
                if (k0 + u < len) {
                    double t0 = v0[u], t1 = v1[u], t2 = v2[u];
                    if (kScaled && A.s1) { t0 = s1.a * t0; t1 = s1.b * t1; t2 = s1.c * t2; }
                    if (kScaled && A.s2) { t0 = s2.a * t0; t1 = s2.b * t1; t2 = s2.c * t2; }
                    acc0 += t0 * xv[u].a;
                    acc1 += t1 * xv[u].b;
                    acc2 += t2 * xv[u].c;
                }
            }
        }
        if (live) {
            const double acc[3] = {acc0, acc1, acc2};
            epi.apply(row, acc, red);
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int q = 0; q < Epi3::kReductions; ++q) {
            const double t = block_sum(red[s][q], lds);
            if (threadIdx.x == 0) partials[(size_t)(s * Epi3::kReductions + q) * (A.part_stride > 0 ? A.part_stride : vgrid) + A.part_base + vb] = t;
        }
    }
    }
}

// The same product software-pipelined across chunks AND slices: the column/value loads of the next chunk (of the same
// slice or of the wave's next slice) are issued right behind the current chunk's x gathers, so a wave's HBM round trip
// overlaps its gather round trip instead of following it (vmcnt counts in order: gathers first, then the younger
// prefetches, so the wait for the gathers leaves the prefetches in flight).  Same rows, same order, same sums.
template <int kLayout>
struct SpmvSliceMeta {
    int64_t base, row, pk_off;
    int width, len, last;
    bool live;
    double s1, s2;
    int lane_c;
    __device__ __forceinline__ void load(const MatView &A, int64_t slice, int lane) {
        row = slice * 64 + lane;
        lane_c = row < A.P.n ? lane : (int)((A.P.n - 1) & 63);
        base = A.P.slice_ptr[slice];
        width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
        live = row < A.P.n;
        len = live ? A.P.row_len[row] : 0;
        s1 = (A.s1 && live) ? A.s1[row] : 1.;
        s2 = (A.s2 && live) ? A.s2[row] : 1.;
        pk_off = kLayout == kSpmvPacked ? A.pk.ptr[slice] : 0;
        last = (kLayout == kSpmvRagged ? (len > 0 ? len : 1) : width) - 1;
    }
    // issues the 16 loads of chunk k0 (branch-free, see spmv_k); advances the packed offset
    __device__ __forceinline__ void issue(const MatView &A, int k0, int lane, int (&c)[8], double (&v)[8]) {
        if (width <= 0) {  // wave-uniform: an all-empty slice owns no storage
#pragma unroll
            for (int u = 0; u < 8; ++u) { c[u] = 0; v[u] = 0.; }
            return;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kLayout == kSpmvPacked) {
                const bool in = k0 + u < len;
                const unsigned long long m = __ballot(in);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int64_t p = in ? pk_off + rank : (m != 0ull ? pk_off : pk_off - 1);
                c[u] = A.pk.col[p];
                v[u] = A.pk.val[p];
                pk_off += __popcll(m);
            } else {
                const int kk = k0 + u < last ? k0 + u : last;
                const int64_t p = base + (int64_t)kk * 64 + lane_c;
                c[u] = A.P.col[p];
                v[u] = A.val[p];
            }
        }
    }
};

// Product on the packed mirror with LDS-staged x windows (XWinDev): one workgroup per block of 256 rows, its 4 waves on
// the block's 4 slices.  Workgroups b and b + 8 share an XCD, so XCD g walks a contiguous eighth of the blocks.
// kScaled = false: the view carries no row scaling (a smoothing solve has materialised its scaled values, materialize_scaled_view):
// the two scaling multiplications per entry and their selects are not compiled in — the stream loop of this kernel is bound by
// instruction issue as much as by memory (39 vector instructions per entry and wavefront, profiles/r03_pmc_products.csv)
// kC: entries per lane and chunk of the stream (two chunks in flight): 8 = 92-94 VGPRs, five workgroups per CU where the level's LDS share
// (XWinDev::cap) allows them.  [r05] measured: 4 (58 VGPRs, up to 24 wavefronts per CU on the level whose windows fit 25 KB) is no faster — 233.1 /
// 226.6 against 226.2 / 227.5 us on that level, 226.4 / 219.7 against 224.9 / 218.6 on the last (scripts/archive/gpu_r05_j.sh): like the per-level LDS
// share itself (16 -> 20 wavefronts per CU: -1 ... -3 %), occupancy is not what holds this product at 4.6-4.8 TB/s
template <class Epi, bool kScaled = true, bool kNT = false, int kC = 8>
__global__ __launch_bounds__(kBlock) void spmv_xwin_k(MatView A, const double *__restrict__ x, Epi epi, double *__restrict__ partials,
                                                      const double *__restrict__ skip_flags) {
    __shared__ double lds[8];
    extern __shared__ __align__(16) double xs[];  // A.xw.cap entries (dynamic: sized per level by the launch)
    if (skip_flags && (skip_flags[0] != 0. || skip_flags[1] != 0.)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: the slice's descriptors go through the scalar cache)
    double r0 = 0., r1 = 0.;
    const int64_t n_blocks = ((int64_t)A.P.n_slices + 3) >> 2;
    int64_t b_begin, b_end, b_step;
    if ((gridDim.x & 7) == 0 && gridDim.x >= 8) {
        const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const int64_t per = (n_blocks + 7) / 8;
        b_begin = (int64_t)xcd * per + bl;
        b_end = (int64_t)(xcd + 1) * per < n_blocks ? (int64_t)(xcd + 1) * per : n_blocks;
        b_step = nb;
    } else {
        b_begin = blockIdx.x; b_end = n_blocks; b_step = gridDim.x;
    }
    for (int64_t b = b_begin; b < b_end; b += b_step) {
        const int ws_built = A.xw.wsize[b];                  // workgroup-uniform
        const int ws = ws_built <= A.xw.cap ? ws_built : -1;  // a window larger than this level's LDS share: global gathers, like a block without one
        // the slice's own stream does not depend on the window: its descriptors are requested before the window is loaded (requesting its
        // first chunk there too was measured: neutral)
        const int64_t slice = b * 4 + wave;
        const bool has_slice = slice < A.P.n_slices;
        const int64_t row = slice * 64 + lane;
        const bool live = has_slice && row < A.P.n;
        int width = 0, len = 0;
        int64_t pk_off = 0;
        if (has_slice) {
            const int64_t base = A.P.slice_ptr[slice];
            width = (int)((A.P.slice_ptr[slice + 1] - base) >> 6);
            len = live ? A.P.row_len[row] : 0;
            pk_off = A.pk.ptr[slice];
        }
        // Addresses are a wave-uniform slice base (scalar registers) plus a 32-bit in-slice offset: no 64-bit vector arithmetic
        // per entry.
        const int64_t sb = __builtin_amdgcn_readfirstlane((int)(pk_off & 0xffffffff)) | ((int64_t)__builtin_amdgcn_readfirstlane((int)(pk_off >> 32)) << 32);
        const unsigned short *s_lidx = A.xw.lidx + sb;
        const double *s_val = A.pk.val + sb;
        int off32 = 0;  // wave-uniform running offset of the chunk inside the slice
        int c[kC], cn[kC];
        double v[kC], vn[kC];
        auto issue = [&](int k0, int (&cc)[kC], double (&vv)[kC]) {
#pragma unroll
            for (int u = 0; u < kC; ++u) {  // branch-free loads, see spmv_k
                const bool in = k0 + u < len;
                const unsigned long long m = __ballot(in);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int p = in ? off32 + rank : (m != 0ull ? off32 : off32 - 1);
                cc[u] = (int)ld_stream<kNT>(s_lidx + p);
                vv[u] = ld_stream<kNT>(s_val + p);
                off32 += __popcll(m);
            }
        };
        if (ws > 0) {
            // window -> LDS: all column loads of a pass are issued before the x gathers, those before the LDS writes
            // (a rolled loop would cost two dependent round trips per element)
            const int32_t *wc = A.xw.wcol + b * kXWinCap;
            for (int j0 = 0; j0 < ws; j0 += 8 * kBlock) {
                int wj[8];
                double xw[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = j0 + q * kBlock + (int)threadIdx.x;
                    wj[q] = ld_stream<kNT>(wc + (j < ws ? j : 0));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) xw[q] = x[wj[q]];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = j0 + q * kBlock + (int)threadIdx.x;
                    if (j < ws) xs[j] = xw[q];
                }
            }
        }
        __syncthreads();
        if (has_slice) {
            const double s1 = (kScaled && A.s1 && live) ? A.s1[row] : 1.;
            const double s2 = (kScaled && A.s2 && live) ? A.s2[row] : 1.;
            double acc = 0.;
            if (ws >= 0) {
                // two chunks in flight: the loads of chunk k0 + 8 are issued before chunk k0 is consumed (with 40 KB of LDS
                // per workgroup only 16 waves fit a CU, so each has to keep more bytes in flight).
                // [r04] ... and they have to STAY in flight: r03's loop copied the second register set into the first at the end of every
                // pass, and a copy reads its source — `s_waitcnt vmcnt(0)` in front of the copies (listing), i.e. every pass ended by
                // waiting for the chunk it had just requested.  Now the two register sets take turns (the loop is unrolled by two): a pass
                // waits for the OLDER chunk only (vmcnt retires in order), the younger one travels while it is multiplied.
                auto consume = [&](int k0, const int (&cc)[kC], const double (&vv)[kC]) {
                    double xv[kC];
#pragma unroll
                    for (int u = 0; u < kC; ++u) xv[u] = xs[cc[u]];
#pragma unroll
                    for (int u = 0; u < kC; ++u) {
                        double t = vv[u];
                        if (kScaled && A.s1) t = s1 * t;
                        if (kScaled && A.s2) t = s2 * t;
                        const double next = acc + t * xv[u];
                        acc = (k0 + u < len) ? next : acc;
                    }
                };
                if (width > 0) issue(0, c, v);
                int k0 = 0;
                for (; k0 + kC < width; k0 += 2 * kC) {  // at the top: chunk k0 is on its way into (c, v)
                    issue(k0 + kC, cn, vn);
                    consume(k0, c, v);
                    if (k0 + 2 * kC < width) issue(k0 + 2 * kC, c, v);
                    consume(k0 + kC, cn, vn);
                }
                if (k0 < width) consume(k0, c, v);
            } else {
                for (int k0 = 0; k0 < width; k0 += 8) {
                    int cg[8];
                    double vg[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool in = k0 + u < len;
                        const unsigned long long m = __ballot(in);
                        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        const int64_t p = in ? pk_off + rank : (m != 0ull ? pk_off : pk_off - 1);
                        cg[u] = ld_stream<kNT>(A.pk.col + p);
                        vg[u] = ld_stream<kNT>(A.pk.val + p);
                        pk_off += __popcll(m);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = x[cg[u]];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        double t = vg[u];
                        if (kScaled && A.s1) t = s1 * t;
                        if (kScaled && A.s2) t = s2 * t;
                        const double next = acc + t * xv[u];
                        acc = (k0 + u < len) ? next : acc;
                    }
                }
            }
            if (live) epi.apply(row, acc, r0, r1);
        }
        __syncthreads();  // the next block overwrites the window
    }
    if (Epi::kReductions > 0 && A.xw.fold_scratch) {
        // [r05] One workgroup per block (launch_spmv): more workgroups than a consumer could fold.  Every workgroup leaves its sums in the level's
        // scratch; the one that arrives LAST — whichever it is — adds them up in index order with a fixed tree, so the result does not depend on the
        // order of arrival, and leaves ONE sum per quantity in partials[q] (the consumers see a fold count of 1).
        __shared__ int s_last;
        const int nwg = (int)gridDim.x;
        const double t0 = block_sum(r0, lds);
        const double t1 = Epi::kReductions > 1 ? block_sum(r1, lds) : 0.;
        // No fences: a device-scope fence writes back / invalidates the XCD's L2 on this chip (10 000 of them per launch cost twice the product).
        // The sums travel as agent-scope atomic stores and loads (they bypass the XCD-local L2), the ticket is an agent-scope atomic, and a workgroup
        // takes its ticket only once its stores are acknowledged (s_waitcnt): whoever draws the last ticket finds every sum in place.
        if (threadIdx.x == 0) {
            __hip_atomic_store(A.xw.fold_scratch + blockIdx.x, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (Epi::kReductions > 1) __hip_atomic_store(A.xw.fold_scratch + nwg + blockIdx.x, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            s_last = __hip_atomic_fetch_add(A.xw.fold_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nwg - 1);
        }
        __syncthreads();
        if (s_last) {
            const double *fs = A.xw.fold_scratch;
#pragma unroll
            for (int q = 0; q < Epi::kReductions; ++q) {
                double v = 0.;
                for (int i0 = threadIdx.x; i0 < nwg; i0 += 8 * kBlock) {  // eight loads in flight per thread, added in index order
                    double t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * kBlock;
                        t[u] = i < nwg ? __hip_atomic_load(fs + (size_t)q * nwg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) v += t[u];
                }
                v = block_sum(v, lds);
                if (threadIdx.x == 0) partials[q] = v;
            }
            if (threadIdx.x == 0) __hip_atomic_store(A.xw.fold_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next launch on this level follows on the same stream)
        }
        return;
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
