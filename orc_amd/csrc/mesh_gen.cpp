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
#include <unordered_map>
#include <vector>

#include "../../include/orc_amd.h"
#include "mesh_raw.hpp"

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

// ------------------------------------------------------------------ mixed tetrahedron / pyramid / prism / hexahedron channel (BASELINE config 5)
// Not a restatement: the reference ships neither a generator nor a mixed-element fixture.  A box of nx x ny x nz blocks is
// cut along x into regions — hexahedra | columns of triangular prisms | hexahedra | transition | Kuhn tetrahedra |
// transition | hexahedra — that stay conforming: prism columns keep quadrilateral sides, the six Kuhn tetrahedra of a
// block share its main diagonal (translation invariant, so neighbouring blocks agree on every face diagonal), and a
// transition block is six pyramids about its centre whose pyramid facing the tetrahedra is itself split along the
// Kuhn diagonal of that face.  Rows of the resulting matrices hold 5 (tet), 6 (pyramid, prism) or 7 (hex) entries.
// The file is the TGRID subset ORC reads (io.rs:32-515): triangular and quadrilateral faces in SEPARATE zones, because
// the reference's reader takes "tokens - 2" as the node count and cannot parse mixed sections (io.rs:232).  Faces are
// oriented so that the reader's normal (n2 - n1) x (n1 - n0) points out of cell 0.
namespace {

struct FaceRec {
    int64_t key[4];  // sorted node ids (key[3] = -1 for triangles)
    int64_t nodes[4];
    int64_t cell;
    int nn;
    bool agglomerated;  // face of a cell that was put together from several blocks' pieces (polyhedral region)
};

struct MixedBuilder {
    int64_t nx, ny, nz;
    double lx, ly, lz;
    std::vector<V3> coords;
    std::vector<FaceRec> recs;
    std::vector<V3> cell_sum;
    std::vector<int> cell_cnt;
    int64_t n_cells = 0;
    int64_t grid_nodes;
    int64_t node(int64_t i, int64_t j, int64_t k) const { return i + (nx + 1) * (j + (ny + 1) * k); }
    int64_t new_cell() {
        cell_sum.push_back({0., 0., 0.});
        cell_cnt.push_back(0);
        return n_cells++;
    }
    void face(int64_t cell, std::initializer_list<int64_t> ns, bool agglomerated = false) {
        FaceRec r;
        r.agglomerated = agglomerated;
        r.nn = (int)ns.size();
        int q = 0;
        for (int64_t v : ns) { r.nodes[q] = v; r.key[q] = v; ++q; }
        if (r.nn == 3) { r.nodes[3] = -1; r.key[3] = -1; }
        std::sort(r.key, r.key + r.nn);
        r.cell = cell;
        recs.push_back(r);
        for (int t = 0; t < r.nn; ++t) { cell_sum[(size_t)cell] = add(cell_sum[(size_t)cell], coords[(size_t)r.nodes[t]]); ++cell_cnt[(size_t)cell]; }
    }
    void tet(int64_t a, int64_t b, int64_t c, int64_t d) {
        const int64_t t = new_cell();
        face(t, {a, b, c}); face(t, {a, b, d}); face(t, {a, c, d}); face(t, {b, c, d});
    }
    // pyramid over the quadrilateral q[0..3] (cyclic order) with apex p; split = two tetrahedra along the diagonal q[s] - q[s+2]
    void pyramid(int64_t p, const int64_t q[4], int split) {
        if (split < 0) {
            const int64_t c = new_cell();
            face(c, {q[0], q[1], q[2], q[3]});
            for (int e = 0; e < 4; ++e) face(c, {q[e], q[(e + 1) & 3], p});
        } else {
            tet(p, q[split], q[(split + 1) & 3], q[(split + 2) & 3]);
            tet(p, q[split], q[(split + 2) & 3], q[(split + 3) & 3]);
        }
    }
};

enum BlockKind { kHexBlock, kPrismBlock, kTetBlock, kTransitionBlock, kPolyBlock };

// Polyhedral region (`polyhedra` != 0; BASELINE config 5: "tet/hex/poly").  True polygonal face sections (face_type 5) are
// out of reach — the reference's reader takes "tokens - 2" as the node count of every face line (io.rs:232) — but a
// polyhedral CELL only needs many faces, and those may be triangles and quadrilaterals.  The blocks of the region alternate
// like a checkerboard between a hexahedron and six pyramids about the block centre; every pyramid is AGGLOMERATED into the
// hexahedron behind its base.  The cell that results is a rhombic dodecahedron: the pyramid sides of the two blocks that
// meet along a block edge are coplanar and separate the same two cells, so they are written as ONE planar rhombus
// (edge end, block centre, edge end, block centre) — 12 quadrilateral faces, 14 vertices, 13 matrix entries per row.
// A pyramid whose base looks out of the region (or at the wall) stays a pyramid, and next to it the cells keep single
// triangular faces, so that no two cells ever share more than one face (one matrix entry per interior face,
// discretization.rs:312-322).
int write_mixed_channel(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz, int polyhedra,
                        int64_t *n_cells_out, int64_t *n_faces_out, OrcMeshData *into = nullptr);

}  // namespace

// [r05] The same mesh WITHOUT the file: the generator's nodes and faces go straight into the reader's raw form (mesh_raw.hpp) and through the
// reader's own geometry step — every array equals, bit for bit, what orc_read_mesh returns for the file orc_mixed_channel_write_msh /
// orc_poly_channel_write_msh would have written (node coordinates are written with 17 significant digits; tests/test_io_cpu.py compares).
// BASELINE configs[4]: a rank's 5.4 M-cell box cost 11.4 s through a 580 MB temporary file (r04).  Handle as from orc_read_mesh.
extern "C" OrcMeshData *orc_mixed_channel_generate(int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz, int polyhedra, int *status) {
    OrcMeshData *d = new OrcMeshData();
    const int st = write_mixed_channel(nullptr, nx, ny, nz, lx, ly, lz, polyhedra, nullptr, nullptr, d);
    if (status) *status = st;
    if (st != ORC_OK) { delete d; return nullptr; }
    return d;
}

extern "C" int orc_mixed_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                                           int64_t *n_cells_out, int64_t *n_faces_out) {
    return write_mixed_channel(path, nx, ny, nz, lx, ly, lz, 0, n_cells_out, n_faces_out);
}

extern "C" int orc_poly_channel_write_msh(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz,
                                          int64_t *n_cells_out, int64_t *n_faces_out) {
    return write_mixed_channel(path, nx, ny, nz, lx, ly, lz, 1, n_cells_out, n_faces_out);
}

namespace {

int write_mixed_channel(const char *path, int64_t nx, int64_t ny, int64_t nz, double lx, double ly, double lz, int polyhedra,
                        int64_t *n_cells_out, int64_t *n_faces_out, OrcMeshData *into) {
    if ((!path && !into) || nx < 20 || ny < 1 || nz < 1) return ORC_ERR_BAD_ARGUMENT;
    MixedBuilder B;
    B.nx = nx; B.ny = ny; B.nz = nz; B.lx = lx; B.ly = ly; B.lz = lz;
    B.grid_nodes = (nx + 1) * (ny + 1) * (nz + 1);
    const int64_t iq0 = nx / 20 + 1, iq1 = nx / 4;                // polyhedra (hexahedra either side)
    const int64_t ip0 = nx * 3 / 10, ip1 = nx * 45 / 100;         // prism columns
    const int64_t it0 = nx / 2 + 1, it1 = nx * 85 / 100;          // Kuhn tetrahedra, one transition block either side
    auto kind = [&](int64_t i) {
        if (polyhedra && i >= iq0 && i < iq1) return kPolyBlock;
        if (i >= ip0 && i < ip1) return kPrismBlock;
        if (i >= it0 && i < it1) return kTetBlock;
        if (i == it0 - 1 || i == it1) return kTransitionBlock;
        return kHexBlock;
    };
    // cell of the hexahedron block (i, j, k) of the polyhedral region, created on first use (a pyramid block may come first)
    std::vector<int64_t> poly_cell;
    if (polyhedra) poly_cell.assign((size_t)((iq1 - iq0) * ny * nz), -1);
    auto poly_hex_cell = [&](int64_t i, int64_t j, int64_t k) {
        int64_t &c = poly_cell[(size_t)((i - iq0) + (iq1 - iq0) * (j + ny * k))];
        if (c < 0) c = B.new_cell();
        return c;
    };
    B.coords.reserve((size_t)B.grid_nodes + (size_t)(2 * ny * nz));
    for (int64_t k = 0; k <= nz; ++k)
        for (int64_t j = 0; j <= ny; ++j)
            for (int64_t i = 0; i <= nx; ++i) B.coords.push_back({lx * (double)i / (double)nx, ly * (double)j / (double)ny, lz * (double)k / (double)nz});
    for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) {
                // corners: v[dx + 2 dy + 4 dz]
                int64_t v[8];
                for (int c = 0; c < 8; ++c) v[c] = B.node(i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1));
                const BlockKind kd = kind(i);
                if (kd == kHexBlock) {
                    const int64_t c = B.new_cell();
                    B.face(c, {v[0], v[1], v[3], v[2]}); B.face(c, {v[4], v[5], v[7], v[6]});
                    B.face(c, {v[0], v[1], v[5], v[4]}); B.face(c, {v[2], v[3], v[7], v[6]});
                    B.face(c, {v[0], v[2], v[6], v[4]}); B.face(c, {v[1], v[3], v[7], v[5]});
                } else if (kd == kPolyBlock && ((i + j + k) & 1) == 0) {  // the hexahedron its neighbours' pyramids are merged into
                    const int64_t c = poly_hex_cell(i, j, k);
                    B.face(c, {v[0], v[1], v[3], v[2]}, true); B.face(c, {v[4], v[5], v[7], v[6]}, true);
                    B.face(c, {v[0], v[1], v[5], v[4]}, true); B.face(c, {v[2], v[3], v[7], v[6]}, true);
                    B.face(c, {v[0], v[2], v[6], v[4]}, true); B.face(c, {v[1], v[3], v[7], v[5]}, true);
                } else if (kd == kPolyBlock) {  // six pyramids about the centre, each handed to the hexahedron behind its base
                    const int64_t p = (int64_t)B.coords.size();
                    V3 s = {0., 0., 0.};
                    for (int c = 0; c < 8; ++c) s = add(s, B.coords[(size_t)v[c]]);
                    B.coords.push_back({s.x / 8., s.y / 8., s.z / 8.});
                    const int64_t quads[6][4] = {{v[0], v[2], v[6], v[4]}, {v[1], v[3], v[7], v[5]}, {v[0], v[1], v[5], v[4]},
                                                 {v[2], v[3], v[7], v[6]}, {v[0], v[1], v[3], v[2]}, {v[4], v[5], v[7], v[6]}};
                    const int64_t di[6] = {-1, 1, 0, 0, 0, 0}, dj[6] = {0, 0, -1, 1, 0, 0}, dk[6] = {0, 0, 0, 0, -1, 1};
                    for (int q = 0; q < 6; ++q) {
                        const int64_t ni = i + di[q], nj = j + dj[q], nk = k + dk[q];
                        const bool merge = ni >= iq0 && ni < iq1 && nj >= 0 && nj < ny && nk >= 0 && nk < nz;
                        const int64_t c = merge ? poly_hex_cell(ni, nj, nk) : B.new_cell();
                        B.face(c, {quads[q][0], quads[q][1], quads[q][2], quads[q][3]}, true);  // inside a merged cell: dropped below
                        for (int e = 0; e < 4; ++e) B.face(c, {quads[q][e], quads[q][(e + 1) & 3], p}, true);
                    }
                } else if (kd == kPrismBlock) {  // two prisms along the diagonal v0 - v3 of the base
                    const int64_t tri[2][3] = {{0, 1, 3}, {0, 3, 2}};
                    for (int t = 0; t < 2; ++t) {
                        const int64_t c = B.new_cell();
                        const int64_t a = tri[t][0], b = tri[t][1], d = tri[t][2];
                        B.face(c, {v[a], v[b], v[d]});
                        B.face(c, {v[a + 4], v[b + 4], v[d + 4]});
                        B.face(c, {v[a], v[b], v[b + 4], v[a + 4]});
                        B.face(c, {v[b], v[d], v[d + 4], v[b + 4]});
                        B.face(c, {v[d], v[a], v[a + 4], v[d + 4]});
                    }
                } else if (kd == kTetBlock) {  // Kuhn: one tetrahedron per ordering of the axes, all through v0 - v7
                    const int perm[6][3] = {{1, 2, 4}, {1, 4, 2}, {2, 1, 4}, {2, 4, 1}, {4, 1, 2}, {4, 2, 1}};
                    for (int t = 0; t < 6; ++t) B.tet(v[0], v[perm[t][0]], v[perm[t][0] + perm[t][1]], v[7]);
                } else {  // six pyramids about the block centre
                    const int64_t p = (int64_t)B.coords.size();
                    V3 s = {0., 0., 0.};
                    for (int c = 0; c < 8; ++c) s = add(s, B.coords[(size_t)v[c]]);
                    B.coords.push_back({s.x / 8., s.y / 8., s.z / 8.});
                    const int64_t quads[6][4] = {{v[0], v[2], v[6], v[4]}, {v[1], v[3], v[7], v[5]}, {v[0], v[1], v[5], v[4]},
                                                 {v[2], v[3], v[7], v[6]}, {v[0], v[1], v[3], v[2]}, {v[4], v[5], v[7], v[6]}};
                    // the x- / x+ quadrilateral is listed from its lowest corner, so its Kuhn diagonal is q[0] - q[2]
                    B.pyramid(p, quads[0], (i > 0 && kind(i - 1) == kTetBlock) ? 0 : -1);
                    B.pyramid(p, quads[1], (i + 1 < nx && kind(i + 1) == kTetBlock) ? 0 : -1);
                    for (int q = 2; q < 6; ++q) B.pyramid(p, quads[q], -1);
                }
            }
    // pair the face records
    auto key_less = [](const FaceRec &a, const FaceRec &b) {
        if (a.nn != b.nn) return a.nn < b.nn;
        for (int q = 0; q < 4; ++q)
            if (a.key[q] != b.key[q]) return a.key[q] < b.key[q];
        return a.cell < b.cell;
    };
    std::sort(B.recs.begin(), B.recs.end(), key_less);
    struct OutFace { int64_t n[4]; int64_t c0, c1; int nn; };
    // zones: 0 FLUID (quad), 1 FLUID_TRI, then per location quad / tri
    const char *zone_names[12] = {"FLUID", "FLUID_TRI", "INLET", "INLET_TRI", "OUTLET", "OUTLET_TRI", "PERIODIC_+Z", "PERIODIC_+Z_TRI",
                                  "PERIODIC_-Z", "PERIODIC_-Z_TRI", "WALL", "WALL_TRI"};
    std::vector<OutFace> zones[12];
    const double eps = 1e-12 * std::max(lx, std::max(ly, lz));
    // orientation: (n2 - n1) x (n1 - n0) out of c0 (io.rs:322-326, mesh.rs:216-222)
    auto orient = [&](OutFace &o) {
        const V3 p0 = B.coords[(size_t)o.n[0]], p1 = B.coords[(size_t)o.n[1]], p2 = B.coords[(size_t)o.n[2]];
        const V3 nrm = cross(sub(p2, p1), sub(p1, p0));
        V3 fc = {0., 0., 0.};
        for (int q = 0; q < o.nn; ++q) fc = add(fc, B.coords[(size_t)o.n[q]]);
        fc = {fc.x / o.nn, fc.y / o.nn, fc.z / o.nn};
        const V3 cs = B.cell_sum[(size_t)o.c0];
        const double cnt = (double)B.cell_cnt[(size_t)o.c0];
        const V3 cc = {cs.x / cnt, cs.y / cnt, cs.z / cnt};
        if (dot(sub(fc, cc), nrm) < 0.) std::reverse(o.n, o.n + o.nn);
    };
    // interior triangles of agglomerated cells, by cell pair: the second triangle between the same two cells closes a rhombus
    std::unordered_map<int64_t, OutFace> open_tri;
    for (size_t r = 0; r < B.recs.size();) {
        const FaceRec &a = B.recs[r];
        const bool paired = r + 1 < B.recs.size() && B.recs[r + 1].nn == a.nn && std::equal(a.key, a.key + 4, B.recs[r + 1].key);
        if (paired && B.recs[r + 1].cell == a.cell) { r += 2; continue; }  // inside an agglomerated cell
        OutFace o;
        o.nn = a.nn;
        o.c0 = a.cell;
        o.c1 = paired ? B.recs[r + 1].cell : -1;
        for (int q = 0; q < 4; ++q) o.n[q] = a.nodes[q];
        if (paired && o.nn == 3 && a.agglomerated && B.recs[r + 1].agglomerated) {
            const int64_t lo = std::min(o.c0, o.c1), hi = std::max(o.c0, o.c1);
            const int64_t pair_key = lo * B.n_cells + hi;
            auto it = open_tri.find(pair_key);
            if (it == open_tri.end()) {
                open_tri.emplace(pair_key, o);
            } else {  // (e0, e1, a) + (e0, e1, b) -> the planar rhombus (e0, a, e1, b)
                const OutFace &t = it->second;
                int64_t shared[2], mine = -1, theirs = -1;
                int ns = 0;
                for (int q = 0; q < 3; ++q) {
                    bool in = false;
                    for (int w = 0; w < 3; ++w) in = in || t.n[w] == o.n[q];
                    if (in) { if (ns < 2) shared[ns] = o.n[q]; ++ns; }
                    else mine = o.n[q];
                }
                for (int w = 0; w < 3; ++w)
                    if (t.n[w] != shared[0] && t.n[w] != shared[1]) theirs = t.n[w];
                if (ns != 2 || mine < 0 || theirs < 0) return ORC_ERR_BAD_ARGUMENT;  // two cells sharing two unrelated faces
                OutFace qd;
                qd.nn = 4; qd.c0 = t.c0; qd.c1 = t.c1;
                qd.n[0] = shared[0]; qd.n[1] = theirs; qd.n[2] = shared[1]; qd.n[3] = mine;
                orient(qd);
                zones[0].push_back(qd);
                open_tri.erase(it);
            }
            r += 2;
            continue;
        }
        orient(o);
        int z;
        if (paired) z = 0;
        else {
            auto all_on = [&](int axis, double val) {
                for (int q = 0; q < o.nn; ++q) {
                    const V3 p = B.coords[(size_t)o.n[q]];
                    const double c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
                    if (std::fabs(c - val) > eps) return false;
                }
                return true;
            };
            if (all_on(2, 0.)) z = 8;
            else if (all_on(2, lz)) z = 6;
            else if (all_on(0, 0.)) z = 2;
            else if (all_on(0, lx)) z = 4;
            else if (all_on(1, 0.) || all_on(1, ly)) z = 10;
            else return ORC_ERR_BAD_ARGUMENT;  // an unpaired face inside the box: the decomposition is not conforming
        }
        zones[z + (o.nn == 3 ? 1 : 0)].push_back(o);
        r += paired ? 2 : 1;
    }
    {  // single triangles between agglomerated cells (next to a pyramid that stayed on its own), in a reproducible order
        std::vector<OutFace> rest;
        rest.reserve(open_tri.size());
        for (auto &kv : open_tri) rest.push_back(kv.second);
        std::sort(rest.begin(), rest.end(), [](const OutFace &a, const OutFace &b) {
            return std::lexicographical_compare(a.n, a.n + 3, b.n, b.n + 3);
        });
        for (OutFace &o : rest) { orient(o); zones[1].push_back(o); }
    }
    int64_t n_faces = 0;
    for (auto &z : zones) n_faces += (int64_t)z.size();
    if (into) {  // what the reader would hold after reading the file written below: same numbering, same zones, same node lists
        orc::RawMesh R;
        R.dims = 3;
        R.n_vert = (int64_t)B.coords.size();
        R.vert.reserve(B.coords.size());
        for (const V3 &p : B.coords) R.vert.push_back({p.x, p.y, p.z});
        R.vert_present.assign(B.coords.size(), 1);
        R.faces.reserve((size_t)n_faces);
        R.node_pool.reserve((size_t)n_faces * 4);
        into->cell_zones.emplace_back((uint64_t)2, (uint64_t)1);  // "(12 (2 1 n 1 0))"
        int zone_id = 3;
        for (int z = 0; z < 12; ++z) {
            if (zones[z].empty()) continue;
            const int zi = (int)into->zones.size();
            into->zones.push_back({(uint64_t)zone_id++, z < 2 ? 2 : 3, 0., {0., 0., 0.}, zone_names[z]});
            for (const OutFace &o : zones[z]) {
                orc::RawFace rf;
                rf.node_begin = (int64_t)R.node_pool.size();
                rf.n_nodes = o.nn;
                rf.zone = zi;
                for (int q = 0; q < o.nn; ++q) R.node_pool.push_back(o.n[q]);
                rf.c[0] = o.c0;
                rf.c[1] = o.c1;
                R.faces.push_back(rf);
            }
        }
        R.n_face = (int64_t)R.faces.size();
        if (n_cells_out) *n_cells_out = B.n_cells;
        if (n_faces_out) *n_faces_out = n_faces;
        return orc::mesh_finalize_geometry("<generated mixed channel>", 0, R, *into);
    }
    FILE *fp = std::fopen(path, "w");
    if (!fp) return ORC_ERR_IO;
    const int64_t nv = (int64_t)B.coords.size();
    std::fprintf(fp, "(0 \"Created by: orc_amd mixed tet/pyramid/prism/hex channel generator\")\n(0 \"Units: Meters\")\n(2 3)\n(0 \"Node Section\")\n");
    std::fprintf(fp, "(10 (0 1 %" PRIx64 " 0 3))\n(10 (1 1 %" PRIx64 " 1 3)(\n", (uint64_t)nv, (uint64_t)nv);
    for (const V3 &p : B.coords) std::fprintf(fp, "%.17g %.17g %.17g\n", p.x, p.y, p.z);
    std::fprintf(fp, "))\n(12 (0 1 %" PRIx64 " 0))\n(12 (2 1 %" PRIx64 " 1 0))\n(13 (0 1 %" PRIx64 " 0))\n", (uint64_t)B.n_cells, (uint64_t)B.n_cells,
                 (uint64_t)n_faces);
    int64_t start = 1;
    int zone_id = 3;
    for (int z = 0; z < 12; ++z) {
        if (zones[z].empty()) continue;
        std::fprintf(fp, "(0 \"Faces of zone %s\")\n", zone_names[z]);
        std::fprintf(fp, "(13 (%x %" PRIx64 " %" PRIx64 " %x %x)(\n", (unsigned)zone_id++, (uint64_t)start, (uint64_t)(start + (int64_t)zones[z].size() - 1),
                     z < 2 ? 2u : 3u, (z & 1) ? 3u : 4u);
        for (const OutFace &o : zones[z]) {
            for (int q = 0; q < o.nn; ++q) std::fprintf(fp, "%" PRIx64 " ", (uint64_t)(o.n[q] + 1));
            std::fprintf(fp, "%" PRIx64 " %" PRIx64 "\n", (uint64_t)(o.c0 + 1), (uint64_t)(o.c1 + 1));
        }
        std::fprintf(fp, "))\n");
        start += (int64_t)zones[z].size();
    }
    std::fclose(fp);
    if (n_cells_out) *n_cells_out = B.n_cells;
    if (n_faces_out) *n_faces_out = n_faces;
    return ORC_OK;
}

}  // namespace
