// lds_gather.hip — cycles of the LDS pipe per 64-lane gather on gfx950, by element width and position pattern: what the
// x-window product (spmv_xwin_k) pays per entry.  Window of 4080 doubles as in the product; 16 waves per CU (the product's
// occupancy with 32 KB of LDS per workgroup).  Build: hipcc --offload-arch=gfx950 -O3 lds_gather.hip -o lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kCap = 4096;  // (the product's window holds 4080)

// MODE 0: one 8-byte read per lane; 1: two 4-byte reads (low / high words in separate arrays)
template <int MODE>
__global__ __launch_bounds__(256) void gather(const unsigned short *__restrict__ idx, int sets, int iters, double *out) {
    __shared__ double xs[kCap];
    __shared__ unsigned lo[MODE == 1 ? kCap : 1], hi[MODE == 1 ? kCap : 1];
    for (int j = threadIdx.x; j < kCap; j += 256) {
        xs[j] = (double)j;
        if (MODE == 1) { lo[j] = (unsigned)j; hi[j] = 0x40000000u; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned short my[8];
    for (int u = 0; u < 8; ++u) my[u] = idx[(((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + u) % sets) * 64 + lane];
    double acc = 0.;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = (my[u] + it * 17) & (kCap - 1);  // every lane shifts alike: the pattern stays, the compiler cannot hoist the read
            if (MODE == 0) v[u] = xs[p];
            else v[u] = __hiloint2double((int)hi[p], (int)lo[p]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int MODE>
int run(const char *name, const unsigned short *idx, int sets, double *out) {
    const int iters = 2000, grid = 256 * 4;  // 4 workgroups = 16 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather<MODE>, dim3(grid), dim3(256), 0, 0, idx, sets, 10, out);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gather<MODE>, dim3(grid), dim3(256), 0, 0, idx, sets, iters, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gathers_per_cu = (double)grid * 4 * iters * 8 / 256.;
    printf("%-44s %8.3f ms  %6.1f cycles of a CU per 64-lane gather (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / gathers_per_cu);
    return 0;
}

int main() {
    const int sets = 4096;
    std::vector<unsigned short> seq(sets * 64), rnd(sets * 64), loc(sets * 64), str(sets * 64);
    srand(7);
    for (int s = 0; s < sets; ++s) {
        const int base = rand() % (kCap - 64);
        for (int l = 0; l < 64; ++l) {
            seq[s * 64 + l] = (unsigned short)(base + l);                       // consecutive positions
            rnd[s * 64 + l] = (unsigned short)(rand() % kCap);                  // anywhere in the window
            loc[s * 64 + l] = (unsigned short)((base + l + rand() % 32 - 16 + kCap) % kCap);  // consecutive +- 16 (ragged neighbours)
            str[s * 64 + l] = (unsigned short)((base % 64 + l * 16) % kCap);    // stride 16 doubles: one bank pair
        }
    }
    unsigned short *d; double *out;
    CK(hipMalloc(&d, seq.size() * 2)); CK(hipMalloc(&out, 64));
    struct { const char *n; std::vector<unsigned short> *v; } pats[] = {{"consecutive", &seq}, {"consecutive +- 16", &loc}, {"random in 4080", &rnd}, {"stride 16 (same bank pair)", &str}};
    for (auto &p : pats) {
        CK(hipMemcpy(d, p.v->data(), p.v->size() * 2, hipMemcpyHostToDevice));
        char name[128];
        snprintf(name, sizeof name, "8-byte reads, %s", p.n);
        if (run<0>(name, d, sets, out)) return 1;
        snprintf(name, sizeof name, "2 x 4-byte reads, %s", p.n);
        if (run<1>(name, d, sets, out)) return 1;
    }
    return 0;
}
