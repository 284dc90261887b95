// atomic_rate.hip — what one shared counter costs on gfx950: M atomicAdd on ONE address from a chip-wide launch, as the aggregation's work
// queues (tail_chase_k: one returning atomicAdd per claimed row), its change counters (agg_sweep_group_k: one per group that changed a row)
// and the Galerkin bounds (galerkin_bound_k: one per slice and tier) issue them.  Variants: returning / non-returning, one per 16-lane group /
// one per wavefront / one per workgroup (the rest reduced first), one address / one address per XCD-sized bucket.
// Build: hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every `every`-th lane issues `per` atomics on counter[(blockIdx.x % buckets) * 32]; RET: the next one depends on the value returned
template <bool RET>
__global__ __launch_bounds__(256) void hammer(int *counter, int every, int per, int buckets, int *sink) {
    int *c = counter + (blockIdx.x % buckets) * 32;
    if ((threadIdx.x % every) != 0) return;
    int acc = 0;
    for (int k = 0; k < per; ++k) {
        if (RET) acc += atomicAdd(c, 1 + (acc & 0));
        else atomicAdd(c, 1);
    }
    if (acc == -12345) sink[0] = acc;
}

int main() {
    int *counter, *sink;
    CK(hipMalloc(&counter, 4096 * sizeof(int)));
    CK(hipMalloc(&sink, sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 2048;
    struct Case { const char *name; bool ret; int every, per, buckets; };
    const Case cases[] = {
        {"returning, one per 16 lanes, x8, one address", true, 16, 8, 1},
        {"returning, one per wavefront, x8, one address", true, 64, 8, 1},
        {"returning, one per workgroup, x8, one address", true, 256, 8, 1},
        {"returning, one per 16 lanes, x8, 8 addresses", true, 16, 8, 8},
        {"returning, one per 16 lanes, x8, 64 addresses", true, 16, 8, 64},
        {"non-returning, one per 16 lanes, x8, one address", false, 16, 8, 1},
        {"non-returning, one per wavefront, x8, one address", false, 64, 8, 1},
        {"non-returning, one per 16 lanes, x8, 64 addresses", false, 16, 8, 64},
        {"non-returning, every lane, x1, one address", false, 1, 1, 1},
    };
    for (const Case &c : cases) {
        CK(hipMemset(counter, 0, 4096 * sizeof(int)));
        for (int rep = 0; rep < 2; ++rep) {  // second launch timed
            CK(hipEventRecord(e0));
            if (c.ret) hipLaunchKernelGGL(hammer<true>, dim3(grid), dim3(256), 0, 0, counter, c.every, c.per, c.buckets, sink);
            else hipLaunchKernelGGL(hammer<false>, dim3(grid), dim3(256), 0, 0, counter, c.every, c.per, c.buckets, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)grid * (256 / c.every) * c.per;
        printf("%-52s %9.0f atomics in %8.3f ms = %7.1f ns per atomic (chip-wide)\n", c.name, n, ms, ms * 1e6 / n);
    }
    return 0;
}
