// AddressSanitizer / UBSan driver for the product's HOST-side io.rs mirror (orc_amd/csrc/mesh_io.cpp): the TGRID reader
// on well-formed and malformed files, zone assignment, write_data / read_data.  The four device-side entry points the
// file refers to are stubbed — nothing here touches a GPU (GPU sanitizers are not available on the pool).
// Built and run by tests/test_product_reader_sanitizers.py with g++ -fsanitize=address,undefined.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "orc_amd.h"

namespace orc {
static std::string g_err;
int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace orc
extern "C" {
OrcMesh *orc_mesh_create(int64_t, int64_t, int32_t, const int64_t *, const int64_t *, const int32_t *, const double *, const double *, const double *,
                         const double *, const double *, const int64_t *, const int64_t *, const int32_t *, const double *, const double *, int *status) {
    if (status) *status = ORC_ERR_NO_DEVICE;
    return nullptr;
}
int orc_mesh_update_zones(OrcMesh *, const int32_t *, const double *, const double *) { return ORC_ERR_NO_DEVICE; }
int orc_calculate_gradients(const OrcMesh *, const double *, const double *, const double *, const double *, const OrcSettings *, double *, double *) {
    return ORC_ERR_NO_DEVICE;
}
OrcMesh *orc_mesh_create_partitioned(int64_t, int64_t, int64_t, int64_t, int32_t, const int64_t *, const int64_t *, const int32_t *, const double *,
                                     const double *, const double *, const double *, const double *, const int64_t *, const int64_t *, const int32_t *,
                                     const double *, const double *, int32_t, const int32_t *, const int64_t *, const int64_t *, const int64_t *, int *status) {
    if (status) *status = ORC_ERR_NO_DEVICE;
    return nullptr;
}
}

static int run_file(const char *path, bool expect_ok) {
    int st = 0;
    OrcMeshData *d = orc_read_mesh(path, &st);
    if (expect_ok != (d != nullptr)) {
        fprintf(stderr, "%s: status %d (%s), expected %s\n", path, st, orc::g_err.c_str(), expect_ok ? "ok" : "failure");
        orc_mesh_data_destroy(d);
        return 1;
    }
    if (!d) return 0;
    int32_t dims, nz;
    int64_t nv, nc, nf, ncf, nfn;
    orc_mesh_data_sizes(d, &dims, &nv, &nc, &nf, &ncf, &nfn, &nz);
    std::vector<int64_t> c0((size_t)nf), c1((size_t)nf), cfp((size_t)nc + 1), cf((size_t)ncf), fnp((size_t)nf + 1), fn((size_t)nfn);
    std::vector<int32_t> fz((size_t)nf);
    std::vector<double> area((size_t)nf), nrm((size_t)3 * nf), fc((size_t)3 * nf), cc((size_t)3 * nc), vol((size_t)nc), vert((size_t)3 * nv);
    orc_mesh_data_arrays(d, c0.data(), c1.data(), fz.data(), area.data(), nrm.data(), fc.data(), cc.data(), vol.data(), cfp.data(), cf.data());
    orc_mesh_data_nodes(d, vert.data(), fnp.data(), fn.data());
    for (int k = 0; k < nz; ++k) {
        char name[64];
        uint64_t id; int32_t zt; double sc, vec[3];
        orc_mesh_data_zone(d, k, &id, &zt, &sc, vec, name, sizeof(name));
        const double v3[3] = {1e-3, 0., 0.};
        orc_mesh_data_set_zone(d, name, ORC_BC_WALL, 0.5, v3);
    }
    orc_mesh_data_set_zone(d, "no such zone", ORC_BC_WALL, 0., nullptr);
    // checkpoint formats
    std::vector<double> u((size_t)nc), v((size_t)nc), w((size_t)nc), p((size_t)nc), u2((size_t)nc), v2((size_t)nc), w2((size_t)nc), p2((size_t)nc);
    for (int64_t i = 0; i < nc; ++i) { u[(size_t)i] = 1e-3 * (double)i / 7.; v[(size_t)i] = -1. / (double)(i + 1); w[(size_t)i] = i % 3 ? 0. : 5e-324; p[(size_t)i] = 1e300 / (double)(i + 1); }
    const std::string out = std::string(path) + ".asan.csv";
    int rc = orc_write_data(out.c_str(), nc, cc.data(), u.data(), v.data(), w.data(), p.data(), -1);
    int64_t nread = 0;
    if (rc == 0) rc = orc_read_data(out.c_str(), nc, u2.data(), v2.data(), w2.data(), p2.data(), &nread);
    remove(out.c_str());
    if (rc != 0 || nread != nc || (nc > 0 && (memcmp(u.data(), u2.data(), sizeof(double) * (size_t)nc) || memcmp(p.data(), p2.data(), sizeof(double) * (size_t)nc)))) {
        fprintf(stderr, "%s: checkpoint round trip failed (%d, %lld rows)\n", path, rc, (long long)nread);
        orc_mesh_data_destroy(d);
        return 1;
    }
    // orc_mesh_partition (partition.cpp) in every ordering and at 1, 2 and 5 ranks: sizes, arrays, the (stubbed) upload
    if (dims == 3 && nc > 0) {
        bool rejected = false;  // a mesh the reader accepts but orc_mesh_create would not (a face without cells): a clean error, once
        for (int ordering = 0; ordering < 3 && !rejected; ++ordering)
            for (int ranks : {1, 2, 5}) {
                int64_t owned_total = 0;
                for (int r = 0; r < ranks; ++r) {
                    int pst = 0;
                    OrcPartition *P = orc_mesh_partition(nc, nf, c0.data(), c1.data(), fz.data(), area.data(), nrm.data(), fc.data(), cc.data(), vol.data(),
                                                         cfp.data(), cf.data(), ranks, r, ordering, &pst);
                    if (!P && pst == ORC_ERR_BAD_ARGUMENT && ordering == 0 && ranks == 1) { rejected = true; break; }
                    if (!P) { fprintf(stderr, "%s: partition failed (%d: %s)\n", path, pst, orc::g_err.c_str()); orc_mesh_data_destroy(d); return 1; }
                    int64_t no, nl, ng, pf, pcf, ns;
                    int32_t np;
                    orc_partition_sizes(P, &no, &nl, &ng, &pf, &pcf, &np, &ns);
                    std::vector<int64_t> a0((size_t)pf), a1((size_t)pf), pcfp((size_t)nl + 1), pcfa((size_t)pcf), gid((size_t)nl), gfid((size_t)pf), sp((size_t)np + 1),
                        si((size_t)ns), rp((size_t)np + 1);
                    std::vector<int32_t> az((size_t)pf), peers((size_t)np);
                    std::vector<double> pa((size_t)pf), pn((size_t)3 * pf), pfc((size_t)3 * pf), pcc((size_t)3 * nl), pv((size_t)nl);
                    orc_partition_arrays(P, a0.data(), a1.data(), az.data(), pa.data(), pn.data(), pfc.data(), pcc.data(), pv.data(), pcfp.data(), pcfa.data(),
                                         gid.data(), gfid.data(), peers.data(), sp.data(), si.data(), rp.data());
                    owned_total += no;
                    int ust = 0;
                    if (orc_partition_upload(P, 0, nullptr, nullptr, nullptr, &ust) != nullptr || ust != ORC_ERR_NO_DEVICE) { orc_partition_destroy(P); orc_mesh_data_destroy(d); return 1; }
                    orc_partition_destroy(P);
                }
                if (rejected) break;
                if (owned_total != nc) { fprintf(stderr, "%s: %d ranks own %lld of %lld cells\n", path, ranks, (long long)owned_total, (long long)nc); orc_mesh_data_destroy(d); return 1; }
            }
    }
    int stu = 0;
    if (orc_mesh_upload(d, &stu) != nullptr || stu != ORC_ERR_NO_DEVICE) { orc_mesh_data_destroy(d); return 1; }  // goes through the (stubbed) device entry
    orc_mesh_data_destroy(d);
    return 0;
}

int main(int argc, char **argv) {
    int bad = 0;
    {   // the mixed tet / pyramid / prism / hex generator (mesh_gen.cpp) and the reader on its output
        const std::string path = std::string(argv[argc - 1] + (argv[argc - 1][0] == '!' ? 1 : 0)) + ".mixed.msh";
        int64_t gc = 0, gf = 0;
        if (orc_mixed_channel_write_msh(path.c_str(), 20, 3, 2, 0.002, 0.001, 2e-4, &gc, &gf) != ORC_OK || gc <= 0) { fprintf(stderr, "mixed generator failed\n"); bad++; }
        else bad += run_file(path.c_str(), true);
        if (orc_mixed_channel_write_msh(path.c_str(), 5, 3, 2, 0.002, 0.001, 2e-4, &gc, &gf) != ORC_ERR_BAD_ARGUMENT) bad++;  // nx < 20
        remove(path.c_str());
    }
    for (int a = 1; a < argc; ++a) {
        const bool expect_ok = argv[a][0] != '!';
        bad += run_file(argv[a] + (expect_ok ? 0 : 1), expect_ok);
    }
    if (!bad) printf("sanitize_reader ok: %d files\n", argc - 1);
    return bad ? 1 : 0;
}
