// mesh_gen.cpp — synthetic structured hex channel in ORC's mesh conventions (host side).
//
// Not a restatement of reference code: ORC ships no generator.  What is mirrored is the DATA
// FORMAT the hot path consumes: face/cell numbering "interior faces first, then
// zones", the zone order of examples/couette_flow_128x64x1.msh, the unit normal outward from
// cell_indices[0] (mesh.rs:216-222) and the geometry rules read_mesh applies to a TGRID file
// (io.rs:322-326 normal, :338-342 face centroid, :375-397 triangle-fan area, :412,419 cell
// centroid = mean of face centroids, :430-433 volume) so that generate() and
// write_msh() -> read_mesh() describe the same Mesh.
#include <algorithm>
#include <array>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/orc_amd.h"

namespace {

struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct Grid {
    int64_t nx, ny, nz;
    double lx, ly, lz;
    int64_t fx, fy, fz, n_int;                 // interior face counts
    int64_t b_inlet, b_outlet, b_mz, b_pz, b_top, b_bot;  // first face id of each boundary zone
    int64_t n_faces, n_cells;
    Grid(int64_t nx_, int64_t ny_, int64_t nz_, double lx_, double ly_, double lz_)
        : nx(nx_), ny(ny_), nz(nz_), lx(lx_), ly(ly_), lz(lz_) {
        fx = (nx - 1) * ny * nz;
        fy = nx * (ny - 1) * nz;
        fz = nx * ny * (nz - 1);
        n_int = fx + fy + fz;
        b_inlet = n_int;
        b_outlet = b_inlet + ny * nz;
        b_mz = b_outlet + ny * nz;
        b_pz = b_mz + nx * ny;
        b_top = b_pz + nx * ny;
        b_bot = b_top + nx * nz;
        n_faces = b_bot + nx * nz;
        n_cells = nx * ny * nz;
    }
    int64_t cell(int64_t i, int64_t j, int64_t k) const { return i + nx * (j + ny * k); }
    int64_t node(int64_t i, int64_t j, int64_t k) const { return i + (nx + 1) * (j + (ny + 1) * k); }
    V3 pos(int64_t i, int64_t j, int64_t k) const {
        return {lx * (double)i / (double)nx, ly * (double)j / (double)ny, lz * (double)k / (double)nz};
    }
    // interior face ids
    int64_t xface(int64_t i, int64_t j, int64_t k) const { return i + (nx - 1) * (j + ny * k); }              // between (i,j,k),(i+1,j,k)
    int64_t yface(int64_t i, int64_t j, int64_t k) const { return fx + i + nx * (j + (ny - 1) * k); }         // between (i,j,k),(i,j+1,k)
    int64_t zface(int64_t i, int64_t j, int64_t k) const { return fx + fy + i + nx * (j + ny * k); }          // between (i,j,k),(i,j,k+1)
};

// Node quadruple (grid indices) of a face; order chosen so that (n2-n1)x(n1-n0) is the outward
// normal of the owner cell c0 (io.rs:322-326 with c0 present).
struct FaceDesc {
    std::array<std::array<int64_t, 3>, 4> n;
    int64_t c0, c1;
    int32_t zone;
};

// axis: 0 x, 1 y, 2 z; sign: +1 / -1 outward direction from the owner; (i,j,k) owner cell.
inline void face_nodes(int axis, int sign, int64_t i, int64_t j, int64_t k, std::array<std::array<int64_t, 3>, 4> &n) {
    int64_t x0 = i, x1 = i + 1, y0 = j, y1 = j + 1, z0 = k, z1 = k + 1;
    if (axis == 0) {
        int64_t x = sign > 0 ? x1 : x0;
        if (sign > 0) n = {{{x, y0, z0}, {x, y0, z1}, {x, y1, z1}, {x, y1, z0}}};
        else n = {{{x, y0, z0}, {x, y1, z0}, {x, y1, z1}, {x, y0, z1}}};
    } else if (axis == 1) {
        int64_t y = sign > 0 ? y1 : y0;
        if (sign > 0) n = {{{x0, y, z0}, {x1, y, z0}, {x1, y, z1}, {x0, y, z1}}};
        else n = {{{x0, y, z0}, {x0, y, z1}, {x1, y, z1}, {x1, y, z0}}};
    } else {
        int64_t z = sign > 0 ? z1 : z0;
        if (sign > 0) n = {{{x0, y0, z}, {x0, y1, z}, {x1, y1, z}, {x1, y0, z}}};
        else n = {{{x0, y0, z}, {x1, y0, z}, {x1, y1, z}, {x0, y1, z}}};
    }
}

// Enumerate faces in id order. F(face_id, desc)
template <class Fn>
void for_each_face(const Grid &g, Fn &&fn) {
    FaceDesc d;
    int64_t id = 0;
    d.zone = 0;
    for (int64_t k = 0; k < g.nz; k++)
        for (int64_t j = 0; j < g.ny; j++)
            for (int64_t i = 0; i + 1 < g.nx; i++) {
                face_nodes(0, +1, i, j, k, d.n); d.c0 = g.cell(i, j, k); d.c1 = g.cell(i + 1, j, k); fn(id++, d);
            }
    for (int64_t k = 0; k < g.nz; k++)
        for (int64_t j = 0; j + 1 < g.ny; j++)
            for (int64_t i = 0; i < g.nx; i++) {
                face_nodes(1, +1, i, j, k, d.n); d.c0 = g.cell(i, j, k); d.c1 = g.cell(i, j + 1, k); fn(id++, d);
            }
    for (int64_t k = 0; k + 1 < g.nz; k++)
        for (int64_t j = 0; j < g.ny; j++)
            for (int64_t i = 0; i < g.nx; i++) {
                face_nodes(2, +1, i, j, k, d.n); d.c0 = g.cell(i, j, k); d.c1 = g.cell(i, j, k + 1); fn(id++, d);
            }
    d.c1 = -1;
    d.zone = 1;  // INLET x-min
    for (int64_t k = 0; k < g.nz; k++) for (int64_t j = 0; j < g.ny; j++) { face_nodes(0, -1, 0, j, k, d.n); d.c0 = g.cell(0, j, k); fn(id++, d); }
    d.zone = 2;  // OUTLET x-max
    for (int64_t k = 0; k < g.nz; k++) for (int64_t j = 0; j < g.ny; j++) { face_nodes(0, +1, g.nx - 1, j, k, d.n); d.c0 = g.cell(g.nx - 1, j, k); fn(id++, d); }
    d.zone = 3;  // PERIODIC_-Z
    for (int64_t j = 0; j < g.ny; j++) for (int64_t i = 0; i < g.nx; i++) { face_nodes(2, -1, i, j, 0, d.n); d.c0 = g.cell(i, j, 0); fn(id++, d); }
    d.zone = 4;  // PERIODIC_+Z
    for (int64_t j = 0; j < g.ny; j++) for (int64_t i = 0; i < g.nx; i++) { face_nodes(2, +1, i, j, g.nz - 1, d.n); d.c0 = g.cell(i, j, g.nz - 1); fn(id++, d); }
    d.zone = 5;  // TOP_WALL y-max
    for (int64_t k = 0; k < g.nz; k++) for (int64_t i = 0; i < g.nx; i++) { face_nodes(1, +1, i, g.ny - 1, k, d.n); d.c0 = g.cell(i, g.ny - 1, k); fn(id++, d); }
    d.zone = 6;  // BOTTOM_WALL y-min
    for (int64_t k = 0; k < g.nz; k++) for (int64_t i = 0; i < g.nx; i++) { face_nodes(1, -1, i, 0, k, d.n); d.c0 = g.cell(i, 0, k); fn(id++, d); }
}

}  // namespace

extern "C" int orc_hex_channel_sizes(int64_t nx, int64_t ny, int64_t nz, int64_t *n_cells, int64_t *n_faces, int64_t *n_cell_faces) {
    if (nx < 1 || ny < 1 || nz < 1) return ORC_ERR_BAD_ARGUMENT;
    Grid g(nx, ny, nz, 1, 1, 1);
    if (n_cells) *n_cells = g.n_cells;
    if (n_faces) *n_faces = g.n_faces;
    if (n_cell_faces) *n_cell_faces = 6 * g.n_cells;
    return ORC_OK;
}

extern "C" int orc_hex_channel_generate(int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                                        int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area,
                                        double *face_normal, double *face_centroid, double *cell_centroid,
                                        double *cell_volume, int64_t *cell_face_ptr, int64_t *cell_faces) {
    if (nx < 1 || ny < 1 || nz < 1) return ORC_ERR_BAD_ARGUMENT;
    Grid g(nx, ny, nz, lx, ly, lz);
    const int64_t n = g.n_cells;
    for (int64_t c = 0; c <= n; c++) cell_face_ptr[c] = 6 * c;
    std::vector<uint8_t> fill((size_t)n, 0);
    for (int64_t c = 0; c < 3 * n; c++) cell_centroid[c] = 0.;
    // faces in ascending id: geometry per io.rs rules, cell face lists fill in ascending order
    for_each_face(g, [&](int64_t f, const FaceDesc &d) {
        V3 p[4];
        for (int q = 0; q < 4; q++) p[q] = g.pos(d.n[q][0], d.n[q][1], d.n[q][2]);
        V3 nr = cross(sub(p[2], p[1]), sub(p[1], p[0]));
        double len = norm(nr);
        nr = {nr.x / len, nr.y / len, nr.z / len};
        V3 cen = {0., 0., 0.};
        for (int q = 0; q < 4; q++) cen = add(cen, p[q]);
        cen = {cen.x / 4., cen.y / 4., cen.z / 4.};
        double area = 0.;
        for (int q = 0; q + 1 < 4; q++) area = area + std::fabs(norm(cross(sub(p[q], cen), sub(p[q + 1], cen)))) / 2.;
        area = area + std::fabs(norm(cross(sub(p[0], cen), sub(p[3], cen)))) / 2.;
        face_c0[f] = d.c0; face_c1[f] = d.c1; face_zone[f] = d.zone; face_area[f] = area;
        face_normal[3 * f] = nr.x; face_normal[3 * f + 1] = nr.y; face_normal[3 * f + 2] = nr.z;
        face_centroid[3 * f] = cen.x; face_centroid[3 * f + 1] = cen.y; face_centroid[3 * f + 2] = cen.z;
        int64_t cs[2] = {d.c0, d.c1};
        for (int q = 0; q < 2; q++) {
            int64_t c = cs[q];
            if (c < 0) continue;
            cell_faces[6 * c + fill[(size_t)c]++] = f;
            cell_centroid[3 * c] += cen.x; cell_centroid[3 * c + 1] += cen.y; cell_centroid[3 * c + 2] += cen.z;
        }
    });
    for (int64_t c = 0; c < n; c++) {
        cell_centroid[3 * c] /= 6.; cell_centroid[3 * c + 1] /= 6.; cell_centroid[3 * c + 2] /= 6.;
        V3 cc = {cell_centroid[3 * c], cell_centroid[3 * c + 1], cell_centroid[3 * c + 2]};
        double vol = 0.;
        for (int q = 0; q < 6; q++) {
            int64_t f = cell_faces[6 * c + q];
            V3 fc = {face_centroid[3 * f], face_centroid[3 * f + 1], face_centroid[3 * f + 2]};
            V3 nr = {face_normal[3 * f], face_normal[3 * f + 1], face_normal[3 * f + 2]};
            vol = vol + face_area[f] * std::fabs(dot(sub(fc, cc), nr)) / 3.;
        }
        cell_volume[c] = vol;
    }
    return ORC_OK;
}

extern "C" int orc_hex_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz) {
    if (nx < 1 || ny < 1 || nz < 1) return ORC_ERR_BAD_ARGUMENT;
    Grid g(nx, ny, nz, lx, ly, lz);
    FILE *fp = std::fopen(path, "w");
    if (!fp) return ORC_ERR_IO;
    const int64_t nv = (nx + 1) * (ny + 1) * (nz + 1);
    std::fprintf(fp, "(0 \"Created by: orc_amd hex channel generator\")\n(0 \"Units: Meters\")\n(2 3)\n(0 \"Node Section\")\n");
    std::fprintf(fp, "(10 (0 1 %" PRIx64 " 0 3))\n(10 (1 1 %" PRIx64 " 1 3)\n(\n", (uint64_t)nv, (uint64_t)nv);
    for (int64_t k = 0; k <= nz; k++)
        for (int64_t j = 0; j <= ny; j++)
            for (int64_t i = 0; i <= nx; i++) {
                V3 p = g.pos(i, j, k);
                std::fprintf(fp, "%.17g %.17g %.17g\n", p.x, p.y, p.z);
            }
    std::fprintf(fp, "))\n(12 (0 1 %" PRIx64 " 0 0))\n(12 (2 1 %" PRIx64 " 1 4))\n(13 (0 1 %" PRIx64 " 0 0))\n", (uint64_t)g.n_cells,
                 (uint64_t)g.n_cells, (uint64_t)g.n_faces);
    const char *names[7] = {"FLUID", "INLET", "OUTLET", "PERIODIC_-Z", "PERIODIC_+Z", "TOP_WALL", "BOTTOM_WALL"};
    const int64_t first[8] = {0, g.b_inlet, g.b_outlet, g.b_mz, g.b_pz, g.b_top, g.b_bot, g.n_faces};
    int cur_zone = -1;
    for_each_face(g, [&](int64_t f, const FaceDesc &d) {
        if (d.zone != cur_zone) {
            if (cur_zone >= 0) std::fprintf(fp, ")\n)\n");
            cur_zone = d.zone;
            std::fprintf(fp, "(0 \"%s of zone %s\")\n", cur_zone == 0 ? "Interior faces" : "Faces", names[cur_zone]);
            std::fprintf(fp, "(13 (%x %" PRIx64 " %" PRIx64 " %x 4)(\n", (unsigned)(cur_zone + 3), (uint64_t)(first[cur_zone] + 1),
                         (uint64_t)first[cur_zone + 1], cur_zone == 0 ? 2u : 3u);
        }
        std::fprintf(fp, "%" PRIx64 " %" PRIx64 " %" PRIx64 " %" PRIx64 " %" PRIx64 " %" PRIx64 "\n",
                     (uint64_t)(g.node(d.n[0][0], d.n[0][1], d.n[0][2]) + 1), (uint64_t)(g.node(d.n[1][0], d.n[1][1], d.n[1][2]) + 1),
                     (uint64_t)(g.node(d.n[2][0], d.n[2][1], d.n[2][2]) + 1), (uint64_t)(g.node(d.n[3][0], d.n[3][1], d.n[3][2]) + 1),
                     (uint64_t)(d.c0 + 1), (uint64_t)(d.c1 + 1));
        (void)f;
    });
    std::fprintf(fp, ")\n)\n(0 \"Zone Sections\")\n");
    std::fclose(fp);
    return ORC_OK;
}
