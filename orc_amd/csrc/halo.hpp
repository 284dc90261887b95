// halo.hpp — ghost-cell exchange plan of one rank.
// Local numbering: owned cells [0, n_own), then one contiguous ghost block per peer.  An exchange packs
// x[send_idx] per peer, runs grouped ncclSend/ncclRecv on the library stream and lands the peer's values directly
// in x[n_own + recv_off ...].  A debug transport (host-staged through a caller-supplied callback, e.g.
// torch.distributed/gloo) lets two processes share ONE GPU in tests, where RCCL refuses duplicate devices.
#pragma once
#include "common.hpp"

namespace orc {

typedef void (*HostExchangeFn)(int n_peers, const int *peers, const double *send, const int64_t *send_off, const int64_t *send_cnt,
                               double *recv, const int64_t *recv_off, const int64_t *recv_cnt, void *user);
typedef void (*HostAllreduceFn)(double *values, int n, int op /*0 sum, 1 max*/, void *user);

struct HaloPlan {
    int64_t n_own = 0, n_ghost = 0, n_send = 0;
    std::vector<int> peers;
    std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;  // per peer, in doubles
    DevBuf<int32_t> send_idx;  // [n_send] owned cells to pack, peer after peer
    DevBuf<double> send_buf;   // [k * n_send]
    DevBuf<double> recv_buf;   // [k * n_ghost] staging for multi-field exchanges
    std::vector<double> h_send, h_recv;  // debug transport
    // Slices [interior_lo, interior_hi) of the mesh pattern hold no row with a ghost column (the longest such run: after a
    // slab cut or a contiguous-block partition the rows along the cuts sit at the ends of the owned range).  A level-0
    // product runs them on `aux_stream` while the exchange travels on the library stream.
    int32_t interior_lo = 0, interior_hi = 0;
    void *aux_stream = nullptr, *ev_ready = nullptr, *ev_done = nullptr;  // hipStream_t / hipEvent_t, created on first use
    ~HaloPlan();  // comm.cpp: the second stream and its events
    HaloPlan() = default;
    HaloPlan(const HaloPlan &) = delete;
    HaloPlan &operator=(const HaloPlan &) = delete;
    bool active() const { return !peers.empty(); }
    // exchange the ghost entries of k vectors (each n_own + n_ghost long) in one grouped launch
    int exchange(double *const *xs, int k);
    int exchange(double *x) { return exchange(&x, 1); }
    // [r04] ONE vector of w interleaved doubles per cell (x[w * cell + s]: the lock-step momentum solve's u, v, w iterates,
    // MatView3): one pack launch, one message per peer of w times the cells — a third of the exchanges of three one-system solves
    int exchange_interleaved(double *xw, int w);
};
// collectives issued by this process since the last reset (halo exchanges, all-reduces incl. status agreements): what a SIMPLE
// iteration costs in latency-bound messages (orc_debug_collectives)
long long comm_collectives(bool reset);

int comm_allreduce_sum(double *dev, int n);
int comm_allreduce_max(double *dev, int n);
// The status word of a partitioned operation must be the same on every rank, or the ranks part ways at the next
// collective: max over ranks of a host-side OrcStatus (no-op on a single rank).
int comm_global_status(int status);
void comm_set_host_transport(HostExchangeFn ex, HostAllreduceFn ar, void *user);
bool comm_host_transport_active();  // the debug transport blocks the calling thread inside exchange()

}  // namespace orc
