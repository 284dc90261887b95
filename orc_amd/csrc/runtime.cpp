// runtime.cpp — process-wide context (device, stream, error text), arena allocator.
#include <algorithm>
#include <cstdarg>
#include <mutex>

#include "common.hpp"
#include "config.hpp"

namespace orc {
void debug_amg_certification(long long out[2], bool reset);  // amg.hip
long long debug_shared_galerkin(bool reset);                   // amg.hip
int debug_xwin_counters(long long out[3], bool reset);       // amg.hip

static thread_local Ctx *t_ctx_override = nullptr;

Ctx &ctx() {
    static Ctx c;
    return t_ctx_override ? *t_ctx_override : c;
}

CtxScope::CtxScope(Ctx *local) { t_ctx_override = local; }
CtxScope::~CtxScope() { t_ctx_override = nullptr; }

int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx().last_error = buf;
    return code;
}


// ------------------------------------------------------------------ environment switches (config.hpp)
static Config g_cfg;
static std::mutex g_cfg_mu;
static bool g_cfg_loaded = false;
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
static bool env_on(const char *name, bool dflt) { const char *e = getenv(name); return (e && *e) ? atoi(e) != 0 : dflt; }
static bool env_set(const char *name) { return getenv(name) != nullptr; }
static std::string env_str(const char *name) { const char *e = getenv(name); return e ? std::string(e) : std::string(); }

void config_reload() {
    Config c;
    c.triple_momentum = env_on("ORC_TRIPLE_MOMENTUM", true);
    c.concurrent_momentum = env_on("ORC_CONCURRENT_MOMENTUM", true);
    c.two_stream_multigrid = env_on("ORC_TWO_STREAM_MULTIGRID", true);
    c.early_p_hierarchy = env_on("ORC_EARLY_P_HIERARCHY", true);
    c.stream_priorities = env_int("ORC_STREAM_PRIORITIES", 3);
    c.halo_overlap = env_on("ORC_HALO_OVERLAP", true);
    c.amg_da = env_on("ORC_AMG_DA", true);
    c.amg_da_steps = std::max(1, env_int("ORC_AMG_DA_STEPS", 1 << 22));
    c.amg_da_group = env_int("ORC_AMG_DA_GROUP", 0);
    c.amg_sibling = env_on("ORC_AMG_SIBLING", true);
    c.amg_shared_scaling = env_on("ORC_AMG_SHARED_SCALING", true);
    c.amg_shared_galerkin = env_on("ORC_AMG_SHARED_GALERKIN", true);
    c.amg_l0_mirror = env_on("ORC_AMG_L0_MIRROR", true);
    c.galerkin_groups = env_str("ORC_GALERKIN_GROUPS");
    c.spmv_nt = env_int("ORC_SPMV_NT", -1);
    c.spmv_narrow_cols = env_on("ORC_SPMV_NARROW_COLS", true);
    c.materialize_scaling = env_int("ORC_MATERIALIZE_SCALING", 4);
    c.spmv_xwin_min_nnz = env_int("ORC_SPMV_XWIN_MIN_NNZ", 24);
    c.spmv_grid = env_int("ORC_SPMV_GRID", 0);
    c.xwin_wgs_per_cu = std::max(1, env_int("ORC_XWIN_WGS_PER_CU", 8));
    c.xwin_cap = env_int("ORC_XWIN_CAP", 0);
    c.xwin_bitwords = env_int("ORC_XWIN_BITWORDS", 0);
    c.xwin_small_bitwords = env_int("ORC_XWIN_SMALL_BITWORDS", 0);
    c.xwin_wg_per_block = env_on("ORC_XWIN_WG_PER_BLOCK", true);
    c.xwin_level_cap = env_on("ORC_XWIN_LEVEL_CAP", true);
    c.gs_slotspace = env_on("ORC_GS_SLOTSPACE", true);
    c.trace = env_set("ORC_DEBUG_TRACE");
    c.amg_trace = env_set("ORC_AMG_TRACE");
    c.arena_trace = env_set("ORC_ARENA_TRACE");
    c.debug_nan = env_set("ORC_DEBUG_NAN");
    c.debug_xwin = env_set("ORC_DEBUG_XWIN") || env_on("ORC_XWIN_STATS", false);
    c.debug_sync = env_int("ORC_DEBUG_SYNC", 0);
    c.inject_lane_error = env_str("ORC_DEBUG_INJECT_LANE_ERROR");
    c.keep_priority_classes = env_on("ORC_DEBUG_KEEP_PRIORITY_CLASSES", false);
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    g_cfg = c;
    g_cfg_loaded = true;
}

const Config &cfg() {
    if (!g_cfg_loaded) config_reload();  // (before orc_init: host-only entries)
    return g_cfg;
}

// ------------------------------------------------------------------ streams
// Priority classes, and the one rule about them [r05].  HIP keeps one pool of hardware queues PER PRIORITY CLASS (GPU_MAX_HW_QUEUES = 4 each);
// a stream that waits for another stream's event is a barrier packet PARKED at the head of its hardware queue.  r04 put the solve streams in the
// highest class, the set-up streams in the lowest and left the library stream — which carries the level-0 / level-1 solves and records the
// events the solve lanes wait for — in the default class BETWEEN them: the lock-step schedule (multigrid_arm3_dev) parks three top-class
// queues on an event that sits behind some hundred kernels of a LOWER class.  One process per card got away with it; two processes on one
// card (the host-transport rehearsal) stalled for good: eight parked top-class queues and the command processor never came round to the
// default-class queue that would release them.  Established by experiment (scripts/archive/gpu_r05_b.sh, gpu_r05_c.sh; 40x26x16 slabs, two ranks):
//   classes as in r04, 4 queues per class: stalls (3 of 3 runs; also with the p' set-up moved) — 3 or 2 queues per class: runs;
//   set-up class ABOVE the solve class, or classes by lane: runs;  ONE class with 4, 8, 16 queues per process: runs.
// So it is neither the number of queues nor of streams, it is the DIRECTION of the wait: a higher class parked on a lower one — priority
// inversion, with the hardware scheduler as the party that never yields.  The rule: a stream only ever waits (hipStreamWaitEvent) for streams
// of its own class or a higher one.  The library stream and the halo-overlap stream therefore live in the SOLVE class (they are solve
// streams); the set-up streams, which are joined by their host threads and wait for nobody, stay below.  And ranks that share one card get
// ONE class altogether: the rehearsal gains nothing from classes.
struct StreamEntry { hipStream_t st; std::string name; int priority; int role; };
static std::mutex g_streams_mu;
static std::vector<StreamEntry> g_streams;

static void stream_register(hipStream_t st, const char *name, int priority, int role) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    g_streams.push_back(StreamEntry{st, name ? name : "?", priority, role});
}

int stream_role(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_streams_mu);
    for (const auto &e : g_streams)
        if (e.st == st) return e.role;
    return kPlainStream;
}

int stream_create(hipStream_t *out, int role, int lane, const char *name) {
    // ORC_STREAM_PRIORITIES: 3 (default) = solve streams above set-up streams — since the round-2 set-up rework the solves are the critical
    // path of the momentum phase (0.863-0.879 s per iteration against 0.903-0.909 s with 2 = set-up above solve and 0.926-0.957 s with 0 =
    // no classes, r02); 1 = one class per lane
    // ORC_DEBUG_KEEP_PRIORITY_CLASSES=1 (scripts/archive/gpu_r05_b.sh only): the r04 behaviour — classes even when ranks share the card — to reproduce the stall
    const int prio_mode = (role == kPlainStream || (device_shared_between_ranks() && !cfg().keep_priority_classes)) ? 0 : cfg().stream_priorities;
    int least = 0, greatest = 0, prio = 0;
    bool with_prio = false;
    if (prio_mode != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        int klass = prio_mode == 1 ? lane : (role == kSolveStream ? 2 : 0);
        if (prio_mode == 3) klass = role == kSolveStream ? 0 : 2;  // products first
        prio = klass == 0 ? greatest : (klass == 2 ? least : (least + greatest) / 2);
        with_prio = true;
    }
    if (with_prio) ORC_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio));
    else ORC_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    char full[96];
    snprintf(full, sizeof(full), "%s[%d]", name ? name : "stream", lane);
    stream_register(*out, full, with_prio ? prio : 0, role);
    return ORC_OK;
}

void stream_destroy(hipStream_t st) {
    if (!st) return;
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i].st == st) { g_streams.erase(g_streams.begin() + (long)i); break; }
    }
    (void)hipStreamDestroy(st);
}

int ensure_init() {
    if (ctx().inited) return ORC_OK;
    return orc_init(-1);
}

Arena::~Arena() {
    if (cfg().arena_trace && !chunks_.empty())
        fprintf(stderr, "[orc arena %p] %zu chunk(s), reserved %.2f GB, high water %.2f GB\n", (void *)this, chunks_.size(), (double)reserved() / 1e9, (double)high_ / 1e9);
    for (auto &c : chunks_)
        if (c.p) (void)hipFree(c.p);
}

int Arena::reset() {
    cur_ = 0;
    off_ = 0;
    live_ = 0;
    // the most any cycle since the last compaction needed: an arena that serves solves of very different footprints in turn
    // (u, v, w, p' on one stream) must be sized for the largest, not for whichever came last — or it would shrink after a
    // small solve, spill into extra chunks in the next large one and compact again, freeing and allocating (hipFree
    // synchronises the device) in every iteration
    peak_ = std::max(peak_, high_);
    const size_t used = peak_;
    high_ = 0;
    if (chunks_.empty() || used == 0) return ORC_OK;
    // a single chunk is never shrunk: only a fragmented reservation (several chunks) is folded into one
    if (chunks_.size() == 1) return ORC_OK;
    const size_t want = used + used / 16 + ((size_t)16 << 20);  // what the largest cycle needed, plus slack for the next one's drift
    if (cfg().arena_trace) fprintf(stderr, "[orc arena %p] compacting %zu chunk(s), reserved %.2f GB -> %.2f GB\n", (void *)this, chunks_.size(), (double)reserved() / 1e9, (double)want / 1e9);
    for (auto &c : chunks_)
        if (c.p) (void)hipFree(c.p);
    chunks_.clear();
    Chunk c{nullptr, want};
    hipError_t e = hipMalloc((void **)&c.p, want);
    if (e != hipSuccess) return set_error(ORC_ERR_HIP, "arena hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    chunks_.push_back(c);
    return ORC_OK;
}

size_t Arena::reserved() const {
    size_t t = 0;
    for (auto &c : chunks_) t += c.size;
    return t;
}

int Arena::alloc_bytes(size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    while (true) {
        if (cur_ < chunks_.size()) {
            Chunk &c = chunks_[cur_];
            if (off_ + bytes <= c.size) {
                *out = c.p + off_;
                off_ += bytes;
                live_ += bytes;
                if (live_ > high_) high_ = live_;
                return ORC_OK;
            }
            // current chunk exhausted: move on (the tail stays unused until release())
            ++cur_;
            off_ = 0;
            continue;
        }
        // new chunk: at least 64 MiB, at least the request, doubling with the reservation
        size_t want = std::max<size_t>(bytes, (size_t)64 << 20);
        want = std::max(want, reserved() / 2);
        Chunk c{nullptr, want};
        hipError_t e = hipMalloc((void **)&c.p, want);
        if (e != hipSuccess) return set_error(ORC_ERR_HIP, "arena hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        chunks_.push_back(c);
    }
}

}  // namespace orc

extern "C" {

int orc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orc_init(int device_ordinal) {
    orc::Ctx &c = orc::ctx();
    int n = orc_device_count();
    if (n <= 0) return orc::set_error(ORC_ERR_NO_DEVICE, "no HIP device visible: liborc_amd has no CPU fallback");
    if (!c.inited) orc::config_reload();  // the environment switches: read here and by orc_reload_environment(), nowhere else
    if (c.inited && (device_ordinal < 0 || device_ordinal == c.device)) return ORC_OK;
    if (device_ordinal < 0) device_ordinal = 0;
    if (device_ordinal >= n) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "device %d out of range (%d visible)", device_ordinal, n);
    ORC_HIP(hipSetDevice(device_ordinal));
    if (c.stream) { orc::stream_destroy(c.stream); c.stream = nullptr; }
    ORC_TRY(orc::stream_create(&c.stream, orc::kSolveStream, 0, "library"));  // a solve stream: stream_create, "the one rule"
    c.device = device_ordinal;
    if (c.guard_events) { (void)hipFree(c.guard_events); c.guard_events = nullptr; }
    ORC_HIP(hipMalloc((void **)&c.guard_events, sizeof(int)));
    ORC_HIP(hipMemset(c.guard_events, 0, sizeof(int)));
    c.inited = true;
    return ORC_OK;
}

int orc_synchronize(void) {
    if (!orc::ctx().inited) return ORC_OK;
    ORC_HIP(hipStreamSynchronize(orc::ctx().stream));
    return ORC_OK;
}

const char *orc_last_error(void) { return orc::ctx().last_error.c_str(); }

int orc_device_memory(int64_t *free_bytes, int64_t *total_bytes) {
    ORC_TRY(orc::ensure_init());
    size_t f = 0, t = 0;
    ORC_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return ORC_OK;
}

int orc_device_info(char *buf, int cap) {
    ORC_TRY(orc::ensure_init());
    if (!buf || cap <= 0) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    hipDeviceProp_t prop;
    char bus[64] = "?";
    const int dev = orc::ctx().device;
    ORC_HIP(hipGetDeviceProperties(&prop, dev));
    (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev);
    const int w = snprintf(buf, (size_t)cap, "%s | pci %s | ordinal %d | %d CUs", prop.name, bus, dev, prop.multiProcessorCount);
    if (w < 0 || w >= cap) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "orc_device_info: buffer of %d bytes is too small", cap);
    return ORC_OK;
}

const char *orc_status_string(int st) {
    switch (st) {
    case ORC_OK: return "ok";
    case ORC_ERR_SOLUTION_DIVERGED: return "solution diverged";                    // solver.rs:220
    case ORC_ERR_MULTIGRID_DIVERGED: return "Multigrid diverged";                  // linear_algebra.rs:104
    case ORC_ERR_JACOBI_NAN: return "diverged";                                    // linear_algebra.rs:194
    case ORC_ERR_JACOBI_TOO_LARGE: return "Diverged - max solution value > 10^10";  // linear_algebra.rs:215
    case ORC_ERR_GS_MAINTENANCE: return "Gauss-Seidel out for maintenance :)";     // linear_algebra.rs:245
    case ORC_ERR_STRUCTURAL_ZERO: return "Tried to access CsrMatrix element that hasn't been stored yet.";  // lib.rs:665
    case ORC_ERR_UNSUPPORTED_BC: return "BC not supported";                         // discretization.rs:116
    case ORC_ERR_UNSUPPORTED_SCHEME: return "unsupported scheme";
    case ORC_ERR_UNSUPPORTED_SOLVER: return "unsupported solution method";         // linear_algebra.rs:297
    case ORC_ERR_BAD_ARGUMENT: return "bad argument";
    case ORC_ERR_NO_DEVICE: return "no HIP device (liborc_amd has no CPU fallback)";
    case ORC_ERR_HIP: return "HIP runtime error";
    case ORC_ERR_IO: return "I/O error";
    case ORC_ERR_COMM: return "RCCL error";
    case ORC_ERR_MESH_FORMAT: return "malformed mesh or data file";                // io.rs:32-571
    case ORC_ERR_NO_BOUNDARY_CONDITIONS: return "You must set boundary conditions."; // solver.rs:770
    case ORC_ERR_ZONE_NOT_FOUND: return "face zone should exist in mesh";          // mesh.rs:194
    case ORC_ERR_SINGULAR_MATRIX: return "called `Option::unwrap()` on a `None` value";  // solver.rs:850,943 (try_inverse)
    default: return "unknown status";
    }
}

void orc_settings_default(OrcSettings *s) {  // lib.rs:58-86
    memset(s, 0, sizeof(*s));
    s->momentum = ORC_MOMENTUM_CD1;
    s->diffusion = ORC_DIFFUSION_CD;
    s->pressure_interpolation = ORC_PINTERP_SECOND_ORDER;
    s->velocity_interpolation = ORC_VINTERP_RHIE_CHOW;
    s->gradient_reconstruction = ORC_GRAD_GREEN_GAUSS_CELL;
    s->pressure_relaxation = 0.01;
    s->momentum_relaxation = 0.5;
    s->solver_type = ORC_SOLVER_MULTIGRID;
    s->iterations = 50;
    s->relaxation = 0.5;
    s->relative_convergence_threshold = 1e-3;
    s->preconditioner = ORC_PRECOND_JACOBI;
    s->q1_compat = 1;
    s->breakdown_guard = 1;
    s->frozen_diagonals = 1;  // the device evaluates all Rhie-Chow diagonals from the previous iteration
    s->reduction_order = ORC_REDUCTION_TREE;
}

int orc_set_breakdown_guard(int on) {
    orc::ctx().breakdown_guard = on != 0;
    return ORC_OK;
}

int orc_set_reduction_order(int order) {
    if (order != ORC_REDUCTION_TREE && order != ORC_REDUCTION_REFERENCE) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "unknown reduction order %d", order);
    orc::ctx().reduction_order = order;
    return ORC_OK;
}

int64_t orc_breakdown_guard_events(int reset) {
    orc::Ctx &c = orc::ctx();
    if (!c.inited || !c.guard_events) return 0;
    int h = 0;
    if (hipStreamSynchronize(c.stream) != hipSuccess) return -1;
    if (hipMemcpy(&h, c.guard_events, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset && hipMemset(c.guard_events, 0, sizeof(int)) != hipSuccess) return -1;
    return h;
}

long long orc_debug_halo_overlaps(void) { return orc::ctx().halo_overlaps; }
int orc_debug_clamp_partials_grid(long long requested) { return orc::clamp_partials_grid(requested); }
int orc_debug_max_partials(void) { return orc::kMaxPartials; }
int orc_debug_amg_certification(long long out[2], int reset) {
    if (!out) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    orc::debug_amg_certification(out, reset != 0);
    return ORC_OK;
}
long long orc_debug_shared_galerkin(int reset) { return orc::debug_shared_galerkin(reset != 0); }
int orc_debug_xwin_counters(long long out[3], int reset) {
    if (!out) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    ORC_TRY(orc::ensure_init());
    return orc::debug_xwin_counters(out, reset != 0);
}

// Which of the library's streams still hold work (hipStreamQuery: never blocks), one line each: "<name> priority <p> busy|idle".  Meant for a
// watchdog thread while the calling thread of a solve hangs in a synchronisation; returns the number of busy streams, -1 on a short buffer.
int orc_debug_stream_report(char *buf, int cap) {
    if (!buf || cap <= 0) return -1;
    std::vector<orc::StreamEntry> copy;
    {
        std::lock_guard<std::mutex> lk(orc::g_streams_mu);
        copy = orc::g_streams;
    }
    int busy = 0, off = 0;
    for (const auto &e : copy) {
        const hipError_t q = hipStreamQuery(e.st);
        const char *state = q == hipSuccess ? "idle" : (q == hipErrorNotReady ? "busy" : hipGetErrorString(q));
        if (q == hipErrorNotReady) ++busy;
        const int w = snprintf(buf + off, (size_t)(cap - off), "%s priority %d %s\n", e.name.c_str(), e.priority, state);
        if (w < 0 || w >= cap - off) return -1;
        off += w;
    }
    return busy;
}

int orc_reload_environment(void) {
    orc::config_reload();
    return ORC_OK;
}

int orc_profile_enable(int on) {
    orc::ctx().profile = on != 0;
    return ORC_OK;
}

}  // extern "C"
