// comm.cpp — RCCL plumbing for the cell-partitioned multi-GPU path.
// One process per GPU; the host launcher (bench.py under torch.distributed.run, or a Rust
// driver) broadcasts the 128-byte unique id and every rank calls orc_comm_init.
// Collectives are issued on the library stream so they order with the kernels around them.
#include <rccl/rccl.h>

#include "common.hpp"

namespace orc {

static_assert(sizeof(ncclUniqueId) == ORC_COMM_ID_BYTES, "ncclUniqueId size");

#define ORC_NCCL(call)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r__ = (call);                                                                                  \
        if (r__ != ncclSuccess) return orc::set_error(ORC_ERR_COMM, "%s failed: %s", #call, ncclGetErrorString(r__)); \
    } while (0)

// C2: sum of a few f64 scalars (BiCGSTAB dot products, report sums) across ranks.
int comm_allreduce_sum(double *dev, int n) {
    Ctx &c = ctx();
    if (c.world <= 1) return ORC_OK;
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

int comm_allreduce_max(double *dev, int n) {
    Ctx &c = ctx();
    if (c.world <= 1) return ORC_OK;
    ORC_NCCL(ncclAllReduce(dev, dev, (size_t)n, ncclDouble, ncclMax, (ncclComm_t)c.nccl_comm, c.stream));
    return ORC_OK;
}

// C1: neighbour halo exchange. For each peer q: send `send_count[q]` doubles starting at
// send_buf + send_off[q], receive recv_count[q] doubles into recv_buf + recv_off[q].
// Grouped point-to-point = one fused launch; xGMI is point-to-point so a slab partition's two
// neighbours map to two direct links.
int comm_halo_exchange(const double *send_buf, const int64_t *send_off, const int64_t *send_count, double *recv_buf,
                       const int64_t *recv_off, const int64_t *recv_count, const int *peers, int n_peers) {
    Ctx &c = ctx();
    if (c.world <= 1 || n_peers == 0) return ORC_OK;
    ORC_NCCL(ncclGroupStart());
    for (int k = 0; k < n_peers; ++k) {
        if (send_count[k] > 0) ORC_NCCL(ncclSend(send_buf + send_off[k], (size_t)send_count[k], ncclDouble, peers[k], (ncclComm_t)c.nccl_comm, c.stream));
        if (recv_count[k] > 0) ORC_NCCL(ncclRecv(recv_buf + recv_off[k], (size_t)recv_count[k], ncclDouble, peers[k], (ncclComm_t)c.nccl_comm, c.stream));
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
    if (world_size == 1) return ORC_OK;
    ncclUniqueId uid;
    memcpy(&uid, id, ORC_COMM_ID_BYTES);
    ncclComm_t comm;
    ORC_NCCL(ncclCommInitRank(&comm, world_size, uid, rank));
    c.nccl_comm = comm;
    return ORC_OK;
}

int orc_comm_finalize(void) {
    orc::Ctx &c = orc::ctx();
    if (c.nccl_comm) {
        (void)hipStreamSynchronize(c.stream);
        ncclCommDestroy((ncclComm_t)c.nccl_comm);
        c.nccl_comm = nullptr;
    }
    c.world = 1;
    c.rank = 0;
    return ORC_OK;
}

}  // extern "C"
