// common.hpp — runtime plumbing shared by the HIP translation units of liborc_amd.so.
// gfx950 only: 64-lane wavefronts are hard-coded.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/orc_amd.h"
#include "config.hpp"

namespace orc {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kMaxGrid = 2048;     // 256 CUs x 8 workgroups: grid-stride above this
constexpr int kMaxPartials = kMaxGrid;

struct Ctx {
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;
    std::string last_error;
    bool profile = false;
    bool breakdown_guard = true;  // OrcSettings.breakdown_guard of the running solve
    int reduction_order = 0;      // OrcReductionOrder of the running solve: 0 = wave trees, 1 = the reference's (nalgebra) association
    long long halo_overlaps = 0;  // level-0 products that ran beside their halo exchange (orc_debug_halo_overlaps)
    int *guard_events = nullptr;  // device counter: BiCGSTAB solves in which the breakdown guard fired (orc_breakdown_guard_events)
    // multi-GPU (comm.cpp)
    int rank = 0, world = 1;
    void *nccl_comm = nullptr;
};
Ctx &ctx();  // the process-wide context, or the calling thread's private copy while a CtxScope is alive
// A worker thread that drives its own stream (concurrent momentum solves, assembly.hip) works on a private copy of
// the context: every kernel launch, copy and error text of the code it calls goes through ctx().
struct CtxScope {
    explicit CtxScope(Ctx *local);
    ~CtxScope();
};

// A solver runs its solves with ITS guard and reduction order; the process-wide defaults (orc_set_breakdown_guard,
// orc_set_reduction_order: what orc_iterative_solve uses) come back on every exit.
struct CtxDefaultsScope {
    Ctx &c;
    bool guard;
    int order;
    explicit CtxDefaultsScope(Ctx &cc) : c(cc), guard(cc.breakdown_guard), order(cc.reduction_order) {}
    ~CtxDefaultsScope() { c.breakdown_guard = guard; c.reduction_order = order; }
    CtxDefaultsScope(const CtxDefaultsScope &) = delete;
    CtxDefaultsScope &operator=(const CtxDefaultsScope &) = delete;
};

int set_error(int code, const char *fmt, ...);
// ORC_DEBUG_TRACE=1: progress markers of the SIMPLE driver and the solves on stderr (rank, thread-agnostic): where does a run stand?
#define ORC_TRACE(...)                                                                     \
    do {                                                                                   \
        if (orc::cfg().trace) { fprintf(stderr, "[orc trace r%d] ", orc::ctx().rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } \
    } while (0)

#define ORC_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return orc::set_error(ORC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#define ORC_TRY(call)                 \
    do {                              \
        int st__ = (call);            \
        if (st__ != ORC_OK) return st__; \
    } while (0)

int ensure_init();

// Every stream the library creates goes through these two (runtime.cpp) — one place decides the priority class, and a registry knows what
// exists: orc_debug_stream_report tells a watchdog which streams still hold work (a stall must name its stream).
// role: the set-up streams carry dependent rounds of small kernels, the solve streams the bandwidth-bound products (classes: DESIGN §6).
enum StreamRole { kSetupStream = 0, kSolveStream = 100, kPlainStream = 200 };
int stream_create(hipStream_t *out, int role, int lane, const char *name);
void stream_destroy(hipStream_t st);
int stream_role(hipStream_t st);  // the role a stream was created with (kPlainStream for a stream the library did not create)
// true when several ranks of this job share ONE device (the host-staged rehearsal transport: comm.cpp): every stream then lives in one
// priority class, see stream_create
bool device_shared_between_ranks();

// Device buffer with explicit lifetime (no hipMalloc inside timed loops: see Arena).
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return ORC_OK;
        ORC_HIP(hipMalloc((void **)&p, count * sizeof(T)));
        return ORC_OK;
    }
    int ensure(size_t count) { return count <= n ? ORC_OK : alloc(count); }
    int upload(const T *h, size_t count) {
        ORC_TRY(ensure(count));
        if (count) ORC_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
        return ORC_OK;
    }
    int download(T *h, size_t count) const {
        if (count) ORC_HIP(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
        return ORC_OK;
    }
    int zero() {
        if (n) ORC_HIP(hipMemsetAsync(p, 0, n * sizeof(T), ctx().stream));
        return ORC_OK;
    }
};

// Stack arena for per-solve temporaries (AMG hierarchies change size every solve; the SIMPLE
// loop must not call hipMalloc).  Chunks are kept until destroy; mark()/release() unwind.
class Arena {
  public:
    ~Arena();
    struct Mark { size_t chunk, off, live = 0; };
    Mark mark() const { return {cur_, off_, live_}; }
    void release(Mark m) { cur_ = m.chunk; off_ = m.off; live_ = m.live; }
    // Everything the arena holds is dead (no stream reads it any more): unwind to empty and, when the reservation has
    // become fragmented (several chunks, or far more than was ever in use at once), replace it by ONE chunk of the
    // high-water size.  hipFree synchronises the device, so this happens in the first iterations only: once one chunk
    // holds a whole iteration nothing is freed or allocated again.
    int reset();
    bool empty() const { return cur_ == 0 && off_ == 0; }
    int alloc_bytes(size_t bytes, void **out);
    template <class T>
    int alloc(size_t count, T **out) { return alloc_bytes(count * sizeof(T), (void **)out); }
    size_t reserved() const;
    // A second stack with its own life cycle that travels with this one (created on first use): a hierarchy set-up keeps the
    // row-contiguous mirror of the level it is aggregating there — needed until the next level's Galerkin product has run, no
    // longer — while its other transient storage unwinds level by level on this stack.
    Arena &companion() { if (!companion_) companion_.reset(new Arena()); return *companion_; }

  private:
    std::unique_ptr<Arena> companion_;
    struct Chunk { char *p; size_t size; };
    std::vector<Chunk> chunks_;
    size_t cur_ = 0, off_ = 0;
    size_t live_ = 0;  // bytes handed out and not released (what ONE chunk would have to hold: the skipped tails of exhausted chunks do not count)
    size_t high_ = 0;  // most live bytes at once since the last reset
    size_t peak_ = 0;  // the largest high_ of any cycle so far
};

// Unwinds an arena to where it stood when the scope was entered, on EVERY exit path (ORC_TRY / ORC_HIP return early).
struct ArenaScope {
    Arena &arena;
    Arena::Mark mark;
    explicit ArenaScope(Arena &a) : arena(a), mark(a.mark()) {}
    ~ArenaScope() { arena.release(mark); }
    ArenaScope(const ArenaScope &) = delete;
    ArenaScope &operator=(const ArenaScope &) = delete;
};

// THE bound on every grid whose workgroups write per-workgroup partial sums (partials[q * gridDim.x + blockIdx.x]): the partial
// arrays hold kMaxPartials entries per quantity, whatever the device's CU count and whatever a measurement switch asks for.
// r03 found out the hard way (ORC_XWIN_WGS_PER_CU = 12 wrote past the arrays and froze the solves); since r04 every launcher
// sizes such a grid through this one function (grid_for, spmv_grid, the window product's balanced grid, the cascade grids) and
// tests/test_abi_cpu.py + tests/test_gpu_grid_switches.py pin it.
static_assert(kMaxGrid <= kMaxPartials, "a grid-stride grid may write one partial sum per workgroup");
constexpr int clamp_partials_grid(int64_t g) { return g < 1 ? 1 : (g > (int64_t)kMaxPartials ? kMaxPartials : (int)g); }

inline int grid_for(int64_t work_items, int per_block = kBlock) {
    return clamp_partials_grid((work_items + per_block - 1) / per_block);
}

// ---------------- device helpers ----------------
#ifdef __HIPCC__
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// max that keeps a NaN (fmax drops it): the reference's max_by(total_cmp) ranks NaN above every number
__device__ __forceinline__ double max_nan(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmax(a, b); }
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max_nan(v, __shfl_down(v, off, 64));
    return v;
}
// Sum over the workgroup (256 threads = 4 waves); result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *lds /*>=4 doubles*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = 0.;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) r += lds[i];
    }
    return r;
}
__device__ __forceinline__ double block_max(double v, double *lds) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = 0.;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        r = lds[0];
        for (int i = 1; i < nw; i++) r = max_nan(r, lds[i]);
    }
    return r;
}
#endif

}  // namespace orc
