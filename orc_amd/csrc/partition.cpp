// partition.cpp — cell partitioning of ANY mesh behind the C ABI, host only.
//
// The reference is single-process; its Mesh (mesh.rs:140-187) arrives here as the flat arrays of orc_mesh_create.  Cells
// are put in an order — ORC's own, reverse Cuthill-McKee over the face-neighbour graph, or sorted along the longest
// extent of the domain — and cut into n_ranks contiguous blocks of that order.  One rank's part follows the rules of
// orc_mesh_create_partitioned: owned cells first (block order), then one ghost block per peer (peers ascending, a
// block sorted by position in the order), faces = those touching an owned cell in ascending global id with the
// global c0/c1 orientation, ghost cells with empty face lists.  A peer's ghost block of my cells and my send list to
// that peer are the same set in the same order by construction, so no negotiation is needed between ranks: every
// rank evaluates this function on the same global arrays.
// With n_ranks = 1 the result is the whole mesh renumbered by the chosen order (the RCM row ordering of the north
// star: fields go in and out in ORC order through global_ids).
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>
#include <queue>

#include "common.hpp"

struct OrcPartition {
    int64_t n_owned = 0, n_local = 0, n_global = 0;
    std::vector<int64_t> face_c0, face_c1, cfp, cf, global_ids, global_face_ids, send_ptr, send_idx, recv_ptr;
    std::vector<int32_t> face_zone, peers;
    std::vector<double> area, normal, fcent, ccent, vol;
};

namespace {

using orc::set_error;

// adjacency (CSR) of the cells through interior faces, neighbours in ascending face id
void build_adjacency(int64_t n, int64_t F, const int64_t *c0, const int64_t *c1, std::vector<int64_t> &ptr, std::vector<int64_t> &adj) {
    ptr.assign((size_t)n + 1, 0);
    for (int64_t f = 0; f < F; ++f)
        if (c1[f] >= 0) { ++ptr[(size_t)c0[f] + 1]; ++ptr[(size_t)c1[f] + 1]; }
    for (int64_t i = 0; i < n; ++i) ptr[(size_t)i + 1] += ptr[(size_t)i];
    adj.resize((size_t)ptr[(size_t)n]);
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t f = 0; f < F; ++f)
        if (c1[f] >= 0) {
            adj[(size_t)cur[(size_t)c0[f]]++] = c1[f];
            adj[(size_t)cur[(size_t)c1[f]]++] = c0[f];
        }
}

// reverse Cuthill-McKee: breadth-first from a minimum-degree cell of every component, neighbours by ascending degree
// (ties by id), the whole sequence reversed
void rcm_order(int64_t n, const std::vector<int64_t> &ptr, const std::vector<int64_t> &adj, std::vector<int64_t> &seq) {
    seq.clear();
    seq.reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    std::vector<int64_t> by_degree((size_t)n);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    auto deg = [&](int64_t c) { return ptr[(size_t)c + 1] - ptr[(size_t)c]; };
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int64_t a, int64_t b) { return deg(a) < deg(b); });
    std::vector<int64_t> nb;
    size_t head = 0;
    for (int64_t start : by_degree) {
        if (seen[(size_t)start]) continue;
        seen[(size_t)start] = 1;
        seq.push_back(start);
        while (head < seq.size()) {
            const int64_t c = seq[head++];
            nb.clear();
            for (int64_t q = ptr[(size_t)c]; q < ptr[(size_t)c + 1]; ++q)
                if (!seen[(size_t)adj[(size_t)q]]) { seen[(size_t)adj[(size_t)q]] = 1; nb.push_back(adj[(size_t)q]); }
            std::sort(nb.begin(), nb.end(), [&](int64_t a, int64_t b) { return deg(a) != deg(b) ? deg(a) < deg(b) : a < b; });
            seq.insert(seq.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(seq.begin(), seq.end());
}

void geometric_order(int64_t n, const double *cc, std::vector<int64_t> &seq) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t c = 0; c < n; ++c)
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], cc[3 * c + k]); hi[k] = std::max(hi[k], cc[3 * c + k]); }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
        if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    seq.resize((size_t)n);
    std::iota(seq.begin(), seq.end(), 0);
    std::stable_sort(seq.begin(), seq.end(), [&](int64_t a, int64_t b) { return cc[3 * a + axis] < cc[3 * b + axis]; });
}

// One rank's part of a mesh whose cells are put in the order `seq` (pos = its inverse) and given to ranks by `owner` (per CELL;
// -1 = nobody's: such a cell may not touch a cell of `rank`).  Owned cells first in the order, then one ghost block per peer
// (peers ascending, a block sorted by position), faces touching an owned cell in ascending id with the global orientation.
OrcPartition *build_part(int64_t n, int64_t F, const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone, const double *face_area,
                         const double *face_normal, const double *face_centroid, const double *cell_centroid, const double *cell_volume,
                         const int64_t *cell_face_ptr, const int64_t *cell_faces, const std::vector<int64_t> &ptr, const std::vector<int64_t> &adj,
                         const std::vector<int64_t> &seq, const std::vector<int64_t> &pos, const std::vector<int32_t> &owner, int32_t n_ranks, int32_t rank,
                         int64_t n_global, int *status) {
    auto fail = [&](int code, const char *msg) -> OrcPartition * {
        if (status) *status = set_error(code, "%s", msg);
        return nullptr;
    };
    auto P = std::make_unique<OrcPartition>();
    P->n_global = n_global;
    // ---- owned cells in the order; ghost cells per peer (sorted by position) and my send lists
    std::vector<int64_t> own_pos;  // positions of my cells, ascending
    for (int64_t p = 0; p < n; ++p)
        if (owner[(size_t)seq[(size_t)p]] == rank) own_pos.push_back(p);
    P->n_owned = (int64_t)own_pos.size();
    std::vector<int64_t> own_local((size_t)n, -1);  // cell -> local id of an owned cell
    for (int64_t l = 0; l < P->n_owned; ++l) own_local[(size_t)seq[(size_t)own_pos[(size_t)l]]] = l;
    std::vector<std::vector<int64_t>> ghost((size_t)n_ranks), send((size_t)n_ranks);  // positions / local ids
    for (int64_t l = 0; l < P->n_owned; ++l) {
        const int64_t c = seq[(size_t)own_pos[(size_t)l]];
        for (int64_t q = ptr[(size_t)c]; q < ptr[(size_t)c + 1]; ++q) {
            const int64_t nb = adj[(size_t)q];
            const int o = owner[(size_t)nb];
            if (o == rank) continue;
            if (o < 0 || o >= n_ranks) return fail(ORC_ERR_BAD_ARGUMENT, "a cell of this rank touches a cell that belongs to no rank");
            ghost[(size_t)o].push_back(pos[(size_t)nb]);
            send[(size_t)o].push_back(l);
        }
    }
    P->recv_ptr.push_back(0);
    P->send_ptr.push_back(0);
    std::vector<std::pair<int64_t, int64_t>> ghost_local;  // (position, local id), sorted by position
    int64_t next_local = P->n_owned;
    for (int r = 0; r < n_ranks; ++r) {
        auto &g = ghost[(size_t)r];
        auto &s = send[(size_t)r];
        std::sort(g.begin(), g.end());
        g.erase(std::unique(g.begin(), g.end()), g.end());
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        if (g.empty() && s.empty()) continue;
        if (g.empty() != s.empty()) return fail(ORC_ERR_BAD_ARGUMENT, "asymmetric cell adjacency");  // cannot happen: faces are two-sided
        P->peers.push_back(r);
        for (int64_t pg : g) ghost_local.emplace_back(pg, next_local++);
        P->recv_ptr.push_back(next_local - P->n_owned);
        for (int64_t ls : s) P->send_idx.push_back(ls);
        P->send_ptr.push_back((int64_t)P->send_idx.size());
    }
    P->n_local = next_local;
    std::sort(ghost_local.begin(), ghost_local.end());
    auto local_id = [&](int64_t cell) -> int64_t {
        if (own_local[(size_t)cell] >= 0) return own_local[(size_t)cell];
        const int64_t p = pos[(size_t)cell];
        auto it = std::lower_bound(ghost_local.begin(), ghost_local.end(), std::make_pair(p, (int64_t)-1));
        return (it != ghost_local.end() && it->first == p) ? it->second : -1;
    };
    // ---- cells
    P->global_ids.resize((size_t)P->n_local);
    for (int64_t l = 0; l < P->n_owned; ++l) P->global_ids[(size_t)l] = seq[(size_t)own_pos[(size_t)l]];
    for (auto &gl : ghost_local) P->global_ids[(size_t)gl.second] = seq[(size_t)gl.first];
    P->ccent.resize((size_t)3 * P->n_local);
    P->vol.resize((size_t)P->n_local);
    for (int64_t l = 0; l < P->n_local; ++l) {
        const int64_t c = P->global_ids[(size_t)l];
        for (int k = 0; k < 3; ++k) P->ccent[(size_t)(3 * l + k)] = cell_centroid[3 * c + k];
        P->vol[(size_t)l] = cell_volume[c];
    }
    // ---- faces touching an owned cell, ascending global id, global orientation
    std::vector<int64_t> local_face((size_t)F, -1);
    auto owned = [&](int64_t cell) { return own_local[(size_t)cell] >= 0; };
    for (int64_t f = 0; f < F; ++f) {
        if (!(owned(face_c0[f]) || (face_c1[f] >= 0 && owned(face_c1[f])))) continue;
        const int64_t l0 = local_id(face_c0[f]), l1 = face_c1[f] >= 0 ? local_id(face_c1[f]) : -1;
        if (l0 < 0 || (face_c1[f] >= 0 && l1 < 0)) return fail(ORC_ERR_BAD_ARGUMENT, "a face of an owned cell has a neighbour outside the ghost layer");
        local_face[(size_t)f] = (int64_t)P->face_c0.size();
        P->global_face_ids.push_back(f);
        P->face_c0.push_back(l0);
        P->face_c1.push_back(l1);
        P->face_zone.push_back(face_zone[f]);
        P->area.push_back(face_area[f]);
        for (int k = 0; k < 3; ++k) { P->normal.push_back(face_normal[3 * f + k]); P->fcent.push_back(face_centroid[3 * f + k]); }
    }
    // ---- face lists of the owned cells (Cell.face_indices: ascending face id, kept by the order-preserving renumbering)
    P->cfp.assign((size_t)P->n_local + 1, 0);
    for (int64_t l = 0; l < P->n_owned; ++l) {
        const int64_t c = P->global_ids[(size_t)l];
        for (int64_t q = cell_face_ptr[c]; q < cell_face_ptr[c + 1]; ++q) {
            const int64_t lf = local_face[(size_t)cell_faces[q]];
            if (lf < 0) return fail(ORC_ERR_BAD_ARGUMENT, "cell_faces names a face that does not touch its cell");
            P->cf.push_back(lf);
        }
        P->cfp[(size_t)l + 1] = (int64_t)P->cf.size();
    }
    for (int64_t l = P->n_owned; l < P->n_local; ++l) P->cfp[(size_t)l + 1] = (int64_t)P->cf.size();
    return P.release();
}

}  // namespace

extern "C" {

OrcPartition *orc_mesh_partition(int64_t n_cells, int64_t n_faces, const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                                 const double *face_area, const double *face_normal, const double *face_centroid,
                                 const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                                 const int64_t *cell_faces, int32_t n_ranks, int32_t rank, int32_t ordering, int *status) {
    auto fail = [&](int code, const char *msg) -> OrcPartition * {
        if (status) *status = set_error(code, "%s", msg);
        return nullptr;
    };
    if (status) *status = ORC_OK;
    if (n_cells < 0 || n_faces < 0 || !face_c0 || !face_c1 || !face_zone || !face_area || !face_normal || !face_centroid || !cell_centroid ||
        !cell_volume || !cell_face_ptr || !cell_faces)
        return fail(ORC_ERR_BAD_ARGUMENT, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ORC_ERR_BAD_ARGUMENT, "bad rank / n_ranks");
    if (ordering < ORC_ORDER_ORC || ordering > ORC_ORDER_GEOMETRIC) return fail(ORC_ERR_BAD_ARGUMENT, "unknown ordering");
    const int64_t n = n_cells, F = n_faces;
    for (int64_t f = 0; f < F; ++f)
        if (face_c0[f] < 0 || face_c0[f] >= n || face_c1[f] >= n) return fail(ORC_ERR_BAD_ARGUMENT, "face cell index out of range");
    std::vector<int64_t> ptr, adj, seq;
    build_adjacency(n, F, face_c0, face_c1, ptr, adj);
    if (ordering == ORC_ORDER_RCM) rcm_order(n, ptr, adj, seq);
    else if (ordering == ORC_ORDER_GEOMETRIC) geometric_order(n, cell_centroid, seq);
    else { seq.resize((size_t)n); std::iota(seq.begin(), seq.end(), 0); }
    std::vector<int64_t> pos((size_t)n);
    for (int64_t i = 0; i < n; ++i) pos[(size_t)seq[(size_t)i]] = i;
    // n_ranks contiguous blocks of the order
    std::vector<int32_t> owner((size_t)n);
    for (int r = 0; r < n_ranks; ++r) {
        const int64_t lo = (int64_t)((__int128)n * r / n_ranks), hi = (int64_t)((__int128)n * (r + 1) / n_ranks);
        for (int64_t p = lo; p < hi; ++p) owner[(size_t)seq[(size_t)p]] = r;
    }
    return build_part(n, F, face_c0, face_c1, face_zone, face_area, face_normal, face_centroid, cell_centroid, cell_volume, cell_face_ptr, cell_faces, ptr, adj,
                      seq, pos, owner, n_ranks, rank, n, status);
}

// [r04] The same with the owner of every cell given by the caller (-1: nobody's — a cell this rank's mesh carries only because
// its generator made it, e.g. the outer of two ghost layers; it may not touch an owned cell) and the cells kept in their own order.
// This is what a rank calls on a mesh it generated or read FOR ITSELF — its share plus ghost layers — so that no process ever holds
// the whole mesh (BASELINE configs[4]: 40 M cells on 8 GPUs).  A peer's ghost block of my cells and my send list to that peer are
// the same cells in the same order provided the two ranks number the cells they share in the same relative order (generators that
// number layer by layer do); parallel.py's world-N tests exchange global ids to check exactly that.  n_global: the cell count of
// the whole mesh (the ranks' owned cells summed), for the report means.
OrcPartition *orc_mesh_partition_owner(int64_t n_cells, int64_t n_faces, const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                                       const double *face_area, const double *face_normal, const double *face_centroid,
                                       const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr,
                                       const int64_t *cell_faces, const int32_t *cell_owner, int32_t n_ranks, int32_t rank, int64_t n_global, int *status) {
    auto fail = [&](int code, const char *msg) -> OrcPartition * {
        if (status) *status = set_error(code, "%s", msg);
        return nullptr;
    };
    if (status) *status = ORC_OK;
    if (n_cells < 0 || n_faces < 0 || !face_c0 || !face_c1 || !face_zone || !face_area || !face_normal || !face_centroid || !cell_centroid ||
        !cell_volume || !cell_face_ptr || !cell_faces || !cell_owner)
        return fail(ORC_ERR_BAD_ARGUMENT, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ORC_ERR_BAD_ARGUMENT, "bad rank / n_ranks");
    const int64_t n = n_cells, F = n_faces;
    for (int64_t f = 0; f < F; ++f)
        if (face_c0[f] < 0 || face_c0[f] >= n || face_c1[f] >= n) return fail(ORC_ERR_BAD_ARGUMENT, "face cell index out of range");
    std::vector<int64_t> ptr, adj, seq((size_t)n), pos((size_t)n);
    build_adjacency(n, F, face_c0, face_c1, ptr, adj);
    std::iota(seq.begin(), seq.end(), 0);
    std::iota(pos.begin(), pos.end(), 0);
    std::vector<int32_t> owner(cell_owner, cell_owner + n);
    for (int32_t o : owner)
        if (o < -1 || o >= n_ranks) return fail(ORC_ERR_BAD_ARGUMENT, "cell owner out of range");
    return build_part(n, F, face_c0, face_c1, face_zone, face_area, face_normal, face_centroid, cell_centroid, cell_volume, cell_face_ptr, cell_faces, ptr, adj,
                      seq, pos, owner, n_ranks, rank, n_global < 0 ? n : n_global, status);
}

void orc_partition_destroy(OrcPartition *p) { delete p; }

int orc_partition_sizes(const OrcPartition *p, int64_t *n_owned, int64_t *n_local, int64_t *n_global, int64_t *n_faces, int64_t *n_cell_faces,
                        int32_t *n_peers, int64_t *n_send) {
    if (!p) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "null partition");
    if (n_owned) *n_owned = p->n_owned;
    if (n_local) *n_local = p->n_local;
    if (n_global) *n_global = p->n_global;
    if (n_faces) *n_faces = (int64_t)p->face_c0.size();
    if (n_cell_faces) *n_cell_faces = (int64_t)p->cf.size();
    if (n_peers) *n_peers = (int32_t)p->peers.size();
    if (n_send) *n_send = (int64_t)p->send_idx.size();
    return ORC_OK;
}

int orc_partition_arrays(const OrcPartition *p, int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area, double *face_normal,
                         double *face_centroid, double *cell_centroid, double *cell_volume, int64_t *cell_face_ptr, int64_t *cell_faces,
                         int64_t *global_ids, int64_t *global_face_ids, int32_t *peers, int64_t *send_ptr, int64_t *send_idx, int64_t *recv_ptr) {
    if (!p) return orc::set_error(ORC_ERR_BAD_ARGUMENT, "null partition");
    auto put = [](auto *dst, const auto &src) { if (dst) std::copy(src.begin(), src.end(), dst); };
    put(face_c0, p->face_c0); put(face_c1, p->face_c1); put(face_zone, p->face_zone); put(face_area, p->area);
    put(face_normal, p->normal); put(face_centroid, p->fcent); put(cell_centroid, p->ccent); put(cell_volume, p->vol);
    put(cell_face_ptr, p->cfp); put(cell_faces, p->cf); put(global_ids, p->global_ids); put(global_face_ids, p->global_face_ids);
    put(peers, p->peers); put(send_ptr, p->send_ptr); put(send_idx, p->send_idx); put(recv_ptr, p->recv_ptr);
    return ORC_OK;
}

OrcMesh *orc_partition_upload(const OrcPartition *p, int32_t n_zones, const int32_t *zone_type, const double *zone_scalar, const double *zone_vector,
                              int *status) {
    if (!p) {
        if (status) *status = orc::set_error(ORC_ERR_BAD_ARGUMENT, "null partition");
        return nullptr;
    }
    return orc_mesh_create_partitioned(p->n_owned, p->n_local, p->n_global, (int64_t)p->face_c0.size(), n_zones, p->face_c0.data(), p->face_c1.data(),
                                       p->face_zone.data(), p->area.data(), p->normal.data(), p->fcent.data(), p->ccent.data(), p->vol.data(),
                                       p->cfp.data(), p->cf.data(), zone_type, zone_scalar, zone_vector, (int32_t)p->peers.size(), p->peers.data(),
                                       p->send_ptr.data(), p->send_idx.data(), p->recv_ptr.data(), status);
}

}  // extern "C"
