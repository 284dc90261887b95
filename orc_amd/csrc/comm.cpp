// comm.cpp — RCCL plumbing for the cell-partitioned multi-GPU path.
// One process per GPU; the host launcher (bench.py under torch.distributed.run, or a Rust driver) broadcasts the
// 128-byte unique id and every rank calls orc_comm_init.  Collectives are issued on the library stream so they
// order with the kernels around them.  xGMI is point-to-point: a slab partition's two neighbours are two direct
// links, and the halo of a 10M-cell slab (64 000 doubles = 512 KB per field) is latency-, not link-bound.
#include <rccl/rccl.h>

#include <atomic>

#include "halo.hpp"

namespace orc {

static_assert(sizeof(ncclUniqueId) == ORC_COMM_ID_BYTES, "ncclUniqueId size");

#define ORC_NCCL(call)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r__ = (call);                                                                                  \
        if (r__ != ncclSuccess) return orc::set_error(ORC_ERR_COMM, "%s failed: %s", #call, ncclGetErrorString(r__)); \
    } while (0)

static std::atomic<long long> g_collectives{0};
long long comm_collectives(bool reset) {
    const long long v = g_collectives.load(std::memory_order_relaxed);
    if (reset) g_collectives.store(0, std::memory_order_relaxed);
    return v;
}

static HostExchangeFn g_host_ex = nullptr;
static HostAllreduceFn g_host_ar = nullptr;
static void *g_host_user = nullptr;

void comm_set_host_transport(HostExchangeFn ex, HostAllreduceFn ar, void *user) {
    g_host_ex = ex; g_host_ar = ar; g_host_user = user;
}

static int host_allreduce(double *dev, int n, int op) {
    std::vector<double> h((size_t)n);
    ORC_HIP(hipMemcpyAsync(h.data(), dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx().stream));
    ORC_HIP(hipStreamSynchronize(ctx().stream));
    g_host_ar(h.data(), n, op, g_host_user);
    ORC_HIP(hipMemcpyAsync(dev, h.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx().stream));
    ORC_HIP(hipStreamSynchronize(ctx().stream));
    return ORC_OK;
}

// C2: a few f64 scalars (BiCGSTAB dot products, report sums) summed across ranks.
int comm_allreduce_sum(double *dev, int n) {
    Ctx &c = ctx();
    if (c.world <= 1) return ORC_OK;
    g_collectives.fetch_add(1, std::memory_order_relaxed);
    if (g_host_ar) return host_allreduce(dev, n, 0);
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

int comm_allreduce_max(double *dev, int n) {
    Ctx &c = ctx();
    if (c.world <= 1) return ORC_OK;
    g_collectives.fetch_add(1, std::memory_order_relaxed);
    if (g_host_ar) return host_allreduce(dev, n, 1);
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclMax, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

int comm_global_status(int status) {
    Ctx &c = ctx();
    if (c.world <= 1) return status;
    static DevBuf<double> slot;
    if (!slot.p && slot.alloc(1) != ORC_OK) return status ? status : ORC_ERR_HIP;
    const double v = (double)status;
    if (hipMemcpyAsync(slot.p, &v, sizeof(double), hipMemcpyHostToDevice, c.stream) != hipSuccess) return status ? status : ORC_ERR_HIP;
    int st = comm_allreduce_max(slot.p, 1);
    if (st != ORC_OK) return st;
    double out = 0.;
    if (hipMemcpyAsync(&out, slot.p, sizeof(double), hipMemcpyDeviceToHost, c.stream) != hipSuccess ||
        hipStreamSynchronize(c.stream) != hipSuccess)
        return status ? status : ORC_ERR_HIP;
    return (int)out;
}

__global__ void halo_pack_k(const double *__restrict__ x, const int32_t *__restrict__ idx, double *__restrict__ buf, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

// C1: neighbour halo exchange of k fields.
bool comm_host_transport_active() { return g_host_ex != nullptr; }
// The host-staged transport exists for one thing: several ranks of a job on ONE card (RCCL refuses duplicate devices).
bool device_shared_between_ranks() { return g_host_ex != nullptr; }

HaloPlan::~HaloPlan() {
    if (ev_ready) (void)hipEventDestroy((hipEvent_t)ev_ready);
    if (ev_done) (void)hipEventDestroy((hipEvent_t)ev_done);
    if (aux_stream) stream_destroy((hipStream_t)aux_stream);
}

int HaloPlan::exchange(double *const *xs, int k) {
    Ctx &c = ctx();
    if (!active() || c.world <= 1) return ORC_OK;
    g_collectives.fetch_add(1, std::memory_order_relaxed);
    ORC_TRY(send_buf.ensure((size_t)k * (size_t)n_send));
    for (int f = 0; f < k; ++f)
        hipLaunchKernelGGL(halo_pack_k, dim3(grid_for(n_send)), dim3(kBlock), 0, c.stream, xs[f], send_idx.p, send_buf.p + (size_t)f * n_send, n_send);
    ORC_HIP(hipGetLastError());
    const int np = (int)peers.size();
    if (g_host_ex) {
        h_send.resize((size_t)n_send);
        h_recv.resize((size_t)n_ghost);
        for (int f = 0; f < k; ++f) {
            ORC_HIP(hipMemcpyAsync(h_send.data(), send_buf.p + (size_t)f * n_send, sizeof(double) * (size_t)n_send, hipMemcpyDeviceToHost, c.stream));
            ORC_HIP(hipStreamSynchronize(c.stream));
            g_host_ex(np, peers.data(), h_send.data(), send_off.data(), send_cnt.data(), h_recv.data(), recv_off.data(), recv_cnt.data(), g_host_user);
            ORC_HIP(hipMemcpyAsync(xs[f] + n_own, h_recv.data(), sizeof(double) * (size_t)n_ghost, hipMemcpyHostToDevice, c.stream));
            ORC_HIP(hipStreamSynchronize(c.stream));
        }
        return ORC_OK;
    }
    ORC_NCCL(ncclGroupStart());
    for (int f = 0; f < k; ++f)
        for (int q = 0; q < np; ++q) {
            if (send_cnt[q] > 0)
                ORC_NCCL(ncclSend(send_buf.p + (size_t)f * n_send + send_off[q], (size_t)send_cnt[q], ncclDouble, peers[q], (ncclComm_t)c.nccl_comm, c.stream));
            if (recv_cnt[q] > 0)
                ORC_NCCL(ncclRecv(xs[f] + n_own + recv_off[q], (size_t)recv_cnt[q], ncclDouble, peers[q], (ncclComm_t)c.nccl_comm, c.stream));
        }
    ORC_NCCL(ncclGroupEnd());
    return ORC_OK;
}

__global__ void halo_pack_w_k(const double *__restrict__ x, const int32_t *__restrict__ idx, double *__restrict__ buf, int64_t n, int w) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * w; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = i / w;
        buf[i] = x[(int64_t)idx[cell] * w + (i - cell * w)];
    }
}

int HaloPlan::exchange_interleaved(double *xw, int w) {
    Ctx &c = ctx();
    if (!active() || c.world <= 1) return ORC_OK;
    if (w < 1) return set_error(ORC_ERR_BAD_ARGUMENT, "exchange_interleaved: width %d", w);
    g_collectives.fetch_add(1, std::memory_order_relaxed);
    ORC_TRY(send_buf.ensure((size_t)w * (size_t)n_send));
    hipLaunchKernelGGL(halo_pack_w_k, dim3(grid_for(n_send * w)), dim3(kBlock), 0, c.stream, xw, send_idx.p, send_buf.p, n_send, w);
    ORC_HIP(hipGetLastError());
    const int np = (int)peers.size();
    if (g_host_ex) {
        h_send.resize((size_t)w * (size_t)n_send);
        h_recv.resize((size_t)w * (size_t)n_ghost);
        std::vector<int64_t> so(send_off), sc(send_cnt), ro(recv_off), rc(recv_cnt);
        for (int q = 0; q < np; ++q) { so[q] *= w; sc[q] *= w; ro[q] *= w; rc[q] *= w; }
        ORC_HIP(hipMemcpyAsync(h_send.data(), send_buf.p, sizeof(double) * h_send.size(), hipMemcpyDeviceToHost, c.stream));
        ORC_HIP(hipStreamSynchronize(c.stream));
        g_host_ex(np, peers.data(), h_send.data(), so.data(), sc.data(), h_recv.data(), ro.data(), rc.data(), g_host_user);
        ORC_HIP(hipMemcpyAsync(xw + (size_t)w * n_own, h_recv.data(), sizeof(double) * h_recv.size(), hipMemcpyHostToDevice, c.stream));
        ORC_HIP(hipStreamSynchronize(c.stream));
        return ORC_OK;
    }
    ORC_NCCL(ncclGroupStart());
    for (int q = 0; q < np; ++q) {
        if (send_cnt[q] > 0)
            ORC_NCCL(ncclSend(send_buf.p + (size_t)w * send_off[q], (size_t)w * send_cnt[q], ncclDouble, peers[q], (ncclComm_t)c.nccl_comm, c.stream));
        if (recv_cnt[q] > 0)
            ORC_NCCL(ncclRecv(xw + (size_t)w * (n_own + recv_off[q]), (size_t)w * recv_cnt[q], ncclDouble, peers[q], (ncclComm_t)c.nccl_comm, c.stream));
    }
    ORC_NCCL(ncclGroupEnd());
    return ORC_OK;
}

}  // namespace orc

extern "C" {

long long orc_debug_collectives(int reset) { return orc::comm_collectives(reset != 0); }

int orc_comm_get_unique_id(unsigned char id[ORC_COMM_ID_BYTES]) {
    ncclUniqueId uid;
    ORC_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, &uid, ORC_COMM_ID_BYTES);
    return ORC_OK;
}

int orc_comm_init(const unsigned char id[ORC_COMM_ID_BYTES], int rank, int world_size) {
    ORC_TRY(orc::ensure_init());
    orc::Ctx &c = orc::ctx();
    if (world_size < 1 || rank < 0 || rank >= world_size) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "bad rank/world");
    if (c.nccl_comm) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "communicator already initialised");
    c.rank = rank;
    c.world = world_size;
    if (world_size == 1 || id == nullptr) return ORC_OK;  // id == NULL: host transport only (orc_comm_set_host_transport)
    ncclUniqueId uid;
    memcpy(&uid, id, ORC_COMM_ID_BYTES);
    ncclComm_t comm;
    ORC_NCCL(ncclCommInitRank(&comm, world_size, uid, rank));
    c.nccl_comm = comm;
    return ORC_OK;
}

int orc_comm_set_host_transport(void *exchange_fn, void *allreduce_fn, void *user) {
    orc::comm_set_host_transport((orc::HostExchangeFn)exchange_fn, (orc::HostAllreduceFn)allreduce_fn, user);
    return ORC_OK;
}

// One-GPU check of the RCCL data path: a single-rank communicator, rank 0 posing as its own neighbour.  Runs the
// real HaloPlan::exchange (pack kernel + grouped ncclSend/ncclRecv, two fields), both all-reduces and the status
// agreement on the library stream and verifies the values.  The multi-rank wiring itself needs >= 2 GPUs.
int orc_comm_selftest(void) {
    ORC_TRY(orc::ensure_init());
    orc::Ctx &c = orc::ctx();
    if (c.nccl_comm || c.world != 1) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "self-test needs an uninitialised communicator");
    ncclUniqueId uid;
    ORC_NCCL(ncclGetUniqueId(&uid));
    ncclComm_t comm;
    ORC_NCCL(ncclCommInitRank(&comm, 1, uid, 0));
    c.nccl_comm = comm;
    c.world = 2;  // makes exchange()/all-reduce take the RCCL branch; the only peer is rank 0 itself
    int st = ORC_OK;
    {
        orc::HaloPlan H;
        H.n_own = 8; H.n_ghost = 4; H.n_send = 4;
        H.peers = {0}; H.send_off = {0}; H.send_cnt = {4}; H.recv_off = {0}; H.recv_cnt = {4};
        const int32_t idx[4] = {1, 3, 5, 7};
        double h[2][12], out[2][12];
        for (int f = 0; f < 2; ++f)
            for (int i = 0; i < 12; ++i) h[f][i] = i < 8 ? 10. * f + i : -1.;
        orc::DevBuf<double> x0, x1, sc;
        st = H.send_idx.upload(idx, 4);
        if (st == ORC_OK) st = x0.upload(h[0], 12);
        if (st == ORC_OK) st = x1.upload(h[1], 12);
        double *xs[2] = {x0.p, x1.p};
        if (st == ORC_OK) st = H.exchange(xs, 2);
        if (st == ORC_OK) st = x0.download(out[0], 12);
        if (st == ORC_OK) st = x1.download(out[1], 12);
        for (int f = 0; f < 2 && st == ORC_OK; ++f)
            for (int q = 0; q < 4; ++q)
                if (out[f][8 + q] != h[f][idx[q]]) st = orc::set_error(ORC_ERR_COMM, "halo self-exchange returned %g, expected %g", out[f][8 + q], h[f][idx[q]]);
        const double v[2] = {1.5, -2.5};
        double r[2] = {0., 0.};
        if (st == ORC_OK) st = sc.upload(v, 2);
        if (st == ORC_OK) st = orc::comm_allreduce_sum(sc.p, 2);
        if (st == ORC_OK) st = orc::comm_allreduce_max(sc.p, 2);
        if (st == ORC_OK) st = sc.download(r, 2);
        if (st == ORC_OK && (r[0] != 1.5 || r[1] != -2.5)) st = orc::set_error(ORC_ERR_COMM, "single-rank all-reduce changed the values");
        if (st == ORC_OK && orc::comm_global_status(3) != 3) st = orc::set_error(ORC_ERR_COMM, "status agreement failed");
        (void)hipStreamSynchronize(c.stream);
    }
    ncclCommDestroy(comm);
    c.nccl_comm = nullptr;
    c.world = 1;
    c.rank = 0;
    return st;
}

// Debug/test: a persistent single-rank RCCL communicator that poses as a two-rank world in which every peer is this rank (what
// orc_comm_selftest sets up for the duration of one call).  A partitioned mesh whose halo plan names peer 0 everywhere then
// runs the WHOLE partitioned path — grouped ncclSend/ncclRecv halos, ncclAllReduce, the status agreement, the momentum lanes
// and the level-0 products that overlap their exchange (the `exchange_first` branch, which the host-staged transport cannot
// take) — on one GPU.  Its ghost values are the rank's own cells, so the run is self-coupled, not a cut of a larger mesh: it
// is compared with itself under ORC_HALO_OVERLAP=0, not with a single-rank run.  orc_comm_finalize ends it.
int orc_comm_init_self_loop(void) {
    ORC_TRY(orc::ensure_init());
    orc::Ctx &c = orc::ctx();
    if (c.nccl_comm || c.world != 1) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "the self-loop communicator needs an uninitialised communicator");
    ncclUniqueId uid;
    ORC_NCCL(ncclGetUniqueId(&uid));
    ncclComm_t comm;
    ORC_NCCL(ncclCommInitRank(&comm, 1, uid, 0));
    c.nccl_comm = comm;
    c.rank = 0;
    c.world = 2;
    return ORC_OK;
}

int orc_comm_finalize(void) {
    orc::Ctx &c = orc::ctx();
    if (c.nccl_comm) {
        (void)hipStreamSynchronize(c.stream);
        ncclCommDestroy((ncclComm_t)c.nccl_comm);
        c.nccl_comm = nullptr;
    }
    orc::comm_set_host_transport(nullptr, nullptr, nullptr);
    c.world = 1;
    c.rank = 0;
    return ORC_OK;
}

}  // extern "C"
