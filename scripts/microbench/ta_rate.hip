// ta_rate.hip — how many cycles a vector-memory wave-instruction costs on gfx950 by per-lane width and address pattern,
// from a cache-resident buffer (so the answer is the L1/TA path, not HBM).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class T, int MODE>
__global__ __launch_bounds__(256) void rd(const T *__restrict__ p, size_t n_elems, int iters, double *out, const int *__restrict__ perm) {
    // MODE 0: lane-contiguous; 1: random element per lane (within the buffer) via perm; 2: stride-16-elements (one line per lane)
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    double acc = 0.;
    size_t base = (wave * 64 * 8) & (n_elems - 1);
    for (int it = 0; it < iters; ++it) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            size_t idx;
            if (MODE == 0) idx = (base + u * 64 + lane) & (n_elems - 1);
            else if (MODE == 1) idx = (size_t)perm[(base + u * 64 + lane) & (n_elems - 1)];
            else idx = (base + (size_t)(u * 64 + lane) * 16) & (n_elems - 1);
            v[u] = p[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char *b = reinterpret_cast<const char *>(&v[u]);
            acc += (double)b[0];
        }
        base = (base + 64 * 8 * 977) & (n_elems - 1);
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <class T, int MODE>
int run(const char *name, const void *buf, size_t bytes, const int *perm, double *out) {
    const size_t n = bytes / sizeof(T);
    const int iters = 200, grid = 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rd<T, MODE>), dim3(grid), dim3(256), 0, 0, (const T *)buf, n, 10, out, perm);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((rd<T, MODE>), dim3(grid), dim3(256), 0, 0, (const T *)buf, n, iters, out, perm);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)grid * 4 * iters * 8;
    const double per_cu_ns = ms * 1e6 / (winstr / 256.);
    printf("%-34s buffer %6.1f MB: %7.3f ms, %6.1f ns per wave-instr per CU (~%5.1f cyc @2.4GHz), %7.1f GB/s useful\n", name, bytes / 1e6, ms, per_cu_ns,
           per_cu_ns * 2.4, winstr * 64 * sizeof(T) / ms / 1e6);
    return 0;
}

int main() {
    const size_t big = 64u << 20;
    void *buf; int *perm; double *out;
    CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    CK(hipMalloc(&out, 64));
    for (size_t bytes : {(size_t)1 << 20, (size_t)16 << 20}) {   // 1 MB: L2 (+L1) resident; 16 MB: L2/MALL
        std::vector<int> h(bytes / 4);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (bytes / 16)); }
        CK(hipMalloc(&perm, h.size() * 4));
        CK(hipMemcpy(perm, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        if (run<int, 0>("dword   contiguous", buf, bytes, perm, out)) return 1;
        if (run<double, 0>("dwordx2 contiguous", buf, bytes, perm, out)) return 1;
        if (run<double2, 0>("dwordx4 contiguous", buf, bytes, perm, out)) return 1;
        if (run<double, 2>("dwordx2 one 128B line per lane", buf, bytes, perm, out)) return 1;
        if (run<double, 1>("dwordx2 random (perm load incl.)", buf, bytes, perm, out)) return 1;
        if (run<int, 1>("dword   random (perm load incl.)", buf, bytes, perm, out)) return 1;
        CK(hipFree(perm));
    }
    return 0;
}
