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
    int stu = 0;
    if (orc_mesh_upload(d, &stu) != nullptr || stu != ORC_ERR_NO_DEVICE) { orc_mesh_data_destroy(d); return 1; }  // goes through the (stubbed) device entry
    orc_mesh_data_destroy(d);
    return 0;
}

int main(int argc, char **argv) {
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        const bool expect_ok = argv[a][0] != '!';
        bad += run_file(argv[a] + (expect_ok ? 0 : 1), expect_ok);
    }
    if (!bad) printf("sanitize_reader ok: %d files\n", argc - 1);
    return bad ? 1 : 0;
}
