// stream_floor.hip — what a SELL-64 product of the level-0 shape (10.24 M rows x 7 entries, 7-point hex stencil) costs per
// wavefront-entry on gfx950 when its parts are switched on one by one: value stream only, + 2-byte column stream, + x gathers
// (= the product), with the chunk depth and the workgroups per CU as parameters, next to a plain read of the same bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off stream_floor.hip -o stream_floor ; run: ./stream_floor [nx ny nz]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kW = 8;  // depths per slice (7 entries + 1 padding, as the library's chunk of 8)

// PARTS: 1 = values, 2 = + columns, 3 = + gathers.  DEPTH: slices in flight per wavefront (loads of the next DEPTH-1 slices are
// requested before the current one is consumed).
template <int PARTS, int DEPTH, bool NT = false>
__global__ __launch_bounds__(256) void product(const double *__restrict__ val, const unsigned short *__restrict__ col16, const int *__restrict__ colbase,
                                               const double *__restrict__ x, double *__restrict__ y, int n_slices, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    // XCD-aware: workgroups b and b + 8 share an XCD; XCD g walks a contiguous eighth of the slices
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, waves_per_xcd = waves >> 3;
    const int per = (n_slices + 7) / 8;
    const int begin = xcd * per + wg_in_xcd * 4 + (threadIdx.x >> 6), end = min(n_slices, (xcd + 1) * per);
    (void)wave;
    double v[DEPTH][kW];
    int c[DEPTH][kW];
    auto issue = [&](int s, int slot) {
        const int64_t base = (int64_t)s * kW * 64 + lane;
#pragma unroll
        for (int k = 0; k < kW; ++k) {
            v[slot][k] = NT ? __builtin_nontemporal_load(val + base + k * 64) : val[base + k * 64];
            if (PARTS >= 2) c[slot][k] = colbase[s * kW + k] + (int)(NT ? __builtin_nontemporal_load(col16 + base + k * 64) : col16[base + k * 64]);
            else c[slot][k] = 0;
        }
    };
    int s = begin;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (s + d * waves_per_xcd < end) issue(s + d * waves_per_xcd, d);
    while (s < end) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {  // slots are compile-time constants: the buffers stay in registers
            if (s < end) {
                const int ahead = s + (DEPTH - 1) * waves_per_xcd;
                if (ahead < end) issue(ahead, (d + DEPTH - 1) % DEPTH);
                double xv[kW];
#pragma unroll
                for (int k = 0; k < kW; ++k) xv[k] = PARTS >= 3 ? x[c[d][k]] : (PARTS == 2 ? (double)c[d][k] : 1.0);
                double acc = 0.;
#pragma unroll
                for (int k = 0; k < kW; ++k) acc = acc + v[d][k] * xv[k];
                const int64_t row = (int64_t)s * 64 + lane;
                if (row < n) { if (NT) __builtin_nontemporal_store(acc, y + row); else y[row] = acc; }
                s += waves_per_xcd;
            }
        }
    }
}

// three systems on one pattern: three value streams, one column stream, interleaved x (24 bytes per column), CHUNK entries per pass
struct __attribute__((aligned(8))) V3 { double a, b, c; };
template <int CHUNK, bool NT, bool NTX, bool SOA = false>
__global__ __launch_bounds__(256) void product3(const double *__restrict__ v0p, const double *__restrict__ v1p, const double *__restrict__ v2p,
                                                const unsigned short *__restrict__ col16, const int *__restrict__ colbase, const double *__restrict__ x3,
                                                double *__restrict__ y3, int n_slices, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, waves_per_xcd = waves >> 3;
    const int per = (n_slices + 7) / 8;
    const int begin = xcd * per + wg_in_xcd * 4 + (threadIdx.x >> 6), end = min(n_slices, (xcd + 1) * per);
    const V3 *xv3 = reinterpret_cast<const V3 *>(x3);
    for (int s = begin; s < end; s += waves_per_xcd) {
        double a0 = 0., a1 = 0., a2 = 0.;
        for (int k0 = 0; k0 < kW; k0 += CHUNK) {
            int c[CHUNK];
            double v0[CHUNK], v1[CHUNK], v2[CHUNK];
            V3 xv[CHUNK];
            const int64_t base = ((int64_t)s * kW + k0) * 64 + lane;
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                const bool in = k0 + u < kW;
                c[u] = in ? colbase[s * kW + k0 + u] + (int)(NT ? __builtin_nontemporal_load(col16 + base + u * 64) : col16[base + u * 64]) : 0;
                v0[u] = in ? (NT ? __builtin_nontemporal_load(v0p + base + u * 64) : v0p[base + u * 64]) : 0.;
                v1[u] = in ? (NT ? __builtin_nontemporal_load(v1p + base + u * 64) : v1p[base + u * 64]) : 0.;
                v2[u] = in ? (NT ? __builtin_nontemporal_load(v2p + base + u * 64) : v2p[base + u * 64]) : 0.;
            }
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                if (SOA) { xv[u].a = x3[c[u]]; xv[u].b = x3[(int64_t)n_slices * 64 + c[u]]; xv[u].c = x3[(int64_t)n_slices * 128 + c[u]]; }  // three vectors of n
                else xv[u] = xv3[c[u]];
            }
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                if (k0 + u < kW) { a0 = a0 + v0[u] * xv[u].a; a1 = a1 + v1[u] * xv[u].b; a2 = a2 + v2[u] * xv[u].c; }
            }
        }
        const int64_t row = (int64_t)s * 64 + lane;
        if (row < n && SOA) { y3[row] = a0; y3[(int64_t)n_slices * 64 + row] = a1; y3[(int64_t)n_slices * 128 + row] = a2; }
        else if (row < n) {
            if (NTX) { __builtin_nontemporal_store(a0, y3 + 3 * row); __builtin_nontemporal_store(a1, y3 + 3 * row + 1); __builtin_nontemporal_store(a2, y3 + 3 * row + 2); }
            else { y3[3 * row] = a0; y3[3 * row + 1] = a1; y3[3 * row + 2] = a2; }
        }
    }
}

// plain read of `bytes` (16 bytes per lane and instruction, eight in flight), grid-stride
template <bool NT>
__global__ __launch_bounds__(256) void plain_read(const double2 *__restrict__ p, int64_t n16, double *out) {
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.;
    for (int64_t i = tid; i < n16; i += 8 * stride) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t[u] = double2{0., 0.};
            if (i + u * stride < n16) {
                if (NT) { t[u].x = __builtin_nontemporal_load(&p[i + u * stride].x); t[u].y = __builtin_nontemporal_load(&p[i + u * stride].y); }
                else t[u] = p[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u].x + t[u].y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int nx = argc > 3 ? atoi(argv[1]) : 400, ny = argc > 3 ? atoi(argv[2]) : 160, nz = argc > 3 ? atoi(argv[3]) : 160;
    const int64_t n = (int64_t)nx * ny * nz;
    const int n_slices = (int)((n + 63) / 64);
    const int64_t padded = (int64_t)n_slices * kW * 64;
    std::vector<double> h_val((size_t)padded, 0.);
    std::vector<unsigned short> h_c16((size_t)padded, 0);
    std::vector<int> h_base((size_t)n_slices * kW, 0);
    std::vector<int> cols((size_t)padded, 0);
    std::vector<int> len((size_t)n_slices * 64, 0);
    for (int64_t i = 0; i < n; ++i) {
        const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((int64_t)nx * ny));
        int64_t nb[7];
        int m = 0;
        if (iz > 0) nb[m++] = i - (int64_t)nx * ny;
        if (iy > 0) nb[m++] = i - nx;
        if (ix > 0) nb[m++] = i - 1;
        nb[m++] = i;
        if (ix < nx - 1) nb[m++] = i + 1;
        if (iy < ny - 1) nb[m++] = i + nx;
        if (iz < nz - 1) nb[m++] = i + (int64_t)nx * ny;
        const int64_t s = i >> 6, l = i & 63;
        len[(size_t)i] = m;
        for (int k = 0; k < kW; ++k) {
            const int64_t pos = (s * kW + k) * 64 + l;
            cols[(size_t)pos] = k < m ? (int)nb[k] : (int)i;
            h_val[(size_t)pos] = k < m ? 1.0 / (1 + k) : 0.;
        }
    }
    for (int64_t s = 0; s < n_slices; ++s)
        for (int k = 0; k < kW; ++k) {
            int lo = 0x7fffffff;
            for (int l = 0; l < 64; ++l) lo = std::min(lo, cols[(size_t)((s * kW + k) * 64 + l)]);
            h_base[(size_t)(s * kW + k)] = lo;
            for (int l = 0; l < 64; ++l) {
                const int d = cols[(size_t)((s * kW + k) * 64 + l)] - lo;
                if (d >= 65536) { printf("depth too wide\n"); return 1; }
                h_c16[(size_t)((s * kW + k) * 64 + l)] = (unsigned short)d;
            }
        }
    double *val, *x, *y, *out;
    unsigned short *c16;
    int *base;
    CK(hipMalloc(&val, padded * 8)); CK(hipMalloc(&c16, padded * 2)); CK(hipMalloc(&base, (size_t)n_slices * kW * 4));
    CK(hipMalloc(&x, n_slices * 64 * 8)); CK(hipMalloc(&y, n_slices * 64 * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(val, h_val.data(), padded * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c16, h_c16.data(), padded * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(base, h_base.data(), (size_t)n_slices * kW * 4, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, n_slices * 64 * 8));
    const double wave_entries = (double)n_slices * kW;
    printf("rows %lld, slices %d, stored entries %lld (8 per row)\n", (long long)n, n_slices, (long long)padded);
    auto report = [&](const char *name, float ms, double bytes) {
        const double cyc = ms * 1e-3 * 2.4e9 / (wave_entries / 256.);
        printf("%-46s %8.1f us  %6.0f GB/s of its own bytes  %6.1f CU-cycles per wavefront-entry\n", name, ms * 1e3, bytes / ms / 1e6, cyc);
    };
    for (int wgs : {8}) {
        const int grid = 256 * wgs;
        printf("-- %d workgroups per CU\n", wgs);
#define RUN(P, D, label, bytes) report(label, time_ms([&] { hipLaunchKernelGGL((product<P, D>), dim3(grid), dim3(256), 0, 0, val, c16, base, x, y, n_slices, n); }, 20), bytes)
#define RUNNT(P, D, label, bytes) report(label, time_ms([&] { hipLaunchKernelGGL((product<P, D, true>), dim3(grid), dim3(256), 0, 0, val, c16, base, x, y, n_slices, n); }, 20), bytes)
        const double b1 = padded * 8. + n * 8., b2 = b1 + padded * 2. + n_slices * kW * 4., b3 = b2 + n * 8.;
        RUN(1, 1, "values, 1 slice in flight", b1);
        RUN(1, 2, "values, 2 slices in flight", b1);
        RUN(1, 3, "values, 3 slices in flight", b1);
        RUN(2, 2, "values + columns, 2 slices in flight", b2);
        RUN(3, 1, "product, 1 slice in flight", b3);
        RUN(3, 2, "product, 2 slices in flight", b3);
        RUN(3, 3, "product, 3 slices in flight", b3);
        RUNNT(1, 2, "values (non-temporal), 2 slices in flight", b1);
        RUNNT(2, 2, "values + columns (non-temporal), 2 in flight", b2);
        RUNNT(3, 1, "product (non-temporal matrix), 1 in flight", b3);
        RUNNT(3, 2, "product (non-temporal matrix), 2 in flight", b3);
    }
    {
        double *v1, *v2, *x3, *y3;
        CK(hipMalloc(&v1, padded * 8)); CK(hipMalloc(&v2, padded * 8)); CK(hipMalloc(&x3, (size_t)n_slices * 64 * 24)); CK(hipMalloc(&y3, (size_t)n_slices * 64 * 24));
        CK(hipMemcpy(v1, val, padded * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(v2, val, padded * 8, hipMemcpyDeviceToDevice));
        CK(hipMemset(x3, 0, (size_t)n_slices * 64 * 24));
        const double b = 3. * padded * 8. + padded * 2. + n_slices * kW * 4. + 2. * n * 24.;
        for (int wgs : {5, 8}) {
            const int grid = 256 * wgs;
            printf("-- three systems, %d workgroups per CU\n", wgs);
#define RUN3(C, NT, NTX, label) report(label, time_ms([&] { hipLaunchKernelGGL((product3<C, NT, NTX>), dim3(grid), dim3(256), 0, 0, val, v1, v2, c16, base, x3, y3, n_slices, n); }, 20), b)
            RUN3(4, false, false, "three systems, chunks of 4");
            RUN3(4, true, false, "three systems, chunks of 4, non-temporal matrix");
            RUN3(4, true, true, "three systems, chunks of 4, nt matrix + nt y");
            report("three systems, separate x / y vectors, nt matrix", time_ms([&] { hipLaunchKernelGGL((product3<4, true, false, true>), dim3(grid), dim3(256), 0, 0, val, v1, v2, c16, base, x3, y3, n_slices, n); }, 20), b);
            report("three systems, separate x / y vectors", time_ms([&] { hipLaunchKernelGGL((product3<4, false, false, true>), dim3(grid), dim3(256), 0, 0, val, v1, v2, c16, base, x3, y3, n_slices, n); }, 20), b);
        }
    }
    for (int wgs : {8, 32}) {
        const double bytes = padded * 8.;
        char name[64];
        snprintf(name, sizeof name, "plain read of the values, %d workgroups per CU", wgs);
        report(name, time_ms([&] { hipLaunchKernelGGL(plain_read<false>, dim3(256 * wgs), dim3(256), 0, 0, (const double2 *)val, padded / 2, out); }, 20), bytes);
        snprintf(name, sizeof name, "plain read, non-temporal, %d workgroups per CU", wgs);
        report(name, time_ms([&] { hipLaunchKernelGGL(plain_read<true>, dim3(256 * wgs), dim3(256), 0, 0, (const double2 *)val, padded / 2, out); }, 20), bytes);
    }
    return 0;
}
