// comm.cpp — RCCL plumbing for the cell-partitioned multi-GPU path.
// One process per GPU; the host launcher (bench.py under torch.distributed.run, or a Rust driver) broadcasts the
// 128-byte unique id and every rank calls orc_comm_init.  Collectives are issued on the library stream so they
// order with the kernels around them.  xGMI is point-to-point: a slab partition's two neighbours are two direct
// links, and the halo of a 10M-cell slab (64 000 doubles = 512 KB per field) is latency-, not link-bound.
#include <rccl/rccl.h>

#include "halo.hpp"

namespace orc {

static_assert(sizeof(ncclUniqueId) == ORC_COMM_ID_BYTES, "ncclUniqueId size");

#define ORC_NCCL(call)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r__ = (call);                                                                                  \
        if (r__ != ncclSuccess) return orc::set_error(ORC_ERR_COMM, "%s failed: %s", #call, ncclGetErrorString(r__)); \
    } while (0)

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
    if (g_host_ar) return host_allreduce(dev, n, 0);
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

int comm_allreduce_max(double *dev, int n) {
    Ctx &c = ctx();
    if (c.world <= 1) return ORC_OK;
    if (g_host_ar) return host_allreduce(dev, n, 1);
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclMax, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

__global__ void halo_pack_k(const double *__restrict__ x, const int32_t *__restrict__ idx, double *__restrict__ buf, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

// C1: neighbour halo exchange of k fields.
int HaloPlan::exchange(double *const *xs, int k) {
    Ctx &c = ctx();
    if (!active() || c.world <= 1) return ORC_OK;
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

}  // namespace orc

extern "C" {

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
