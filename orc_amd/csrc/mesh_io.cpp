// mesh_io.cpp — host side of SURVEY §8(f): io::read_mesh (io.rs:32-515) into a flat host image of mesh::Mesh,
// the zone lookup of Mesh::get_face_zone (mesh.rs:189-195), and the text checkpoint formats read_data / write_data /
// write_data_with_precision / write_gradients (io.rs:519-662).  No device work here: orc_mesh_upload hands the arrays
// to mesh_upload (assembly.hip).  Compiled with -ffp-contract=off like the kernels, so the geometry (io.rs:289-438)
// rounds exactly as the reference's operator chains do.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "assembly.hpp"
#include "common.hpp"
#include "orc_amd.h"

#include "mesh_raw.hpp"

namespace {

using orc::set_error;

using V3 = orc::MeshV3;
// numerical_types::Vector operators (lib.rs:240-273, 356-447, 529-538), one rounding per written operation
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 over(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 unit(V3 a) { return over(a, norm(a)); }

// ---- line cursor: BufRead::lines() semantics ('\n' terminated, one trailing '\r' dropped, last line may lack '\n')
struct LineCursor {
    const char *p, *end;
    bool next(std::string_view &out) {
        if (p >= end) return false;
        const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *stop = q ? q : end;
        const char *e = stop;
        if (q && e > p && e[-1] == '\r') --e;
        out = std::string_view(p, (size_t)(e - p));
        p = q ? q + 1 : end;
        return true;
    }
};

inline bool is_ascii_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\x0C' || c == '\r'; }

// split_ascii_whitespace
void split_ws(std::string_view s, std::vector<std::string_view> &out) {
    out.clear();
    size_t i = 0, n = s.size();
    while (i < n) {
        while (i < n && is_ascii_ws(s[i])) ++i;
        size_t b = i;
        while (i < n && !is_ascii_ws(s[i])) ++i;
        if (i > b) out.push_back(s.substr(b, i - b));
    }
}

// usize::from_str_radix(s, 16): optional '+', then hex digits only
bool parse_hex(std::string_view s, uint64_t &v) {
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size() || s.size() - i > 16) return false;
    uint64_t r = 0;
    for (; i < s.size(); ++i) {
        char c = s[i];
        unsigned d;
        if (c >= '0' && c <= '9') d = (unsigned)(c - '0');
        else if (c >= 'a' && c <= 'f') d = (unsigned)(c - 'a') + 10;
        else if (c >= 'A' && c <= 'F') d = (unsigned)(c - 'A') + 10;
        else return false;
        r = (r << 4) | d;
    }
    v = r;
    return true;
}

// str::parse::<f64>(): correctly rounded decimal; no surrounding whitespace, no hex floats
bool parse_f64(std::string_view s, double &v) {
    char buf[128];
    if (s.empty() || s.size() >= sizeof(buf)) return false;
    for (char c : s)
        if (c == 'x' || c == 'X' || c == '(' || is_ascii_ws(c)) return false;
    memcpy(buf, s.data(), s.size());
    buf[s.size()] = 0;
    char *endp = nullptr;
    errno = 0;
    v = strtod(buf, &endp);
    return endp == buf + s.size();
}

// read_section_header_common (io.rs:47-54): every maximal run of [0-9a-z], read as hexadecimal
bool header_items(std::string_view line, std::vector<uint64_t> &items) {
    items.clear();
    size_t i = 0, n = line.size();
    auto in_class = [](char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z'); };
    while (i < n) {
        while (i < n && !in_class(line[i])) ++i;
        size_t b = i;
        while (i < n && in_class(line[i])) ++i;
        if (i > b) {
            uint64_t v;
            if (!parse_hex(line.substr(b, i - b), v)) return false;  // .expect("valid hex")
            items.push_back(v);
        }
    }
    return true;
}

bool valid_bc(uint64_t t) {  // FaceConditionTypes::try_from (mesh.rs:51-66)
    switch (t) {
    case 2: case 3: case 4: case 5: case 7: case 8: case 9: case 10: case 12: case 14: case 20: case 24: case 31: case 36: case 37: return true;
    default: return false;
    }
}

using orc::RawFace;

int fail(const char *path, int64_t line_no, const char *what) {
    return set_error(ORC_ERR_MESH_FORMAT, "%s:%lld: %s", path, (long long)line_no, what);
}

int parse_msh(const char *path, const std::string &text, OrcMeshData &d) {
    LineCursor cur{text.data(), text.data() + text.size()};
    int64_t line_no = 0;
    auto next = [&](std::string_view &l) {
        bool ok = cur.next(l);
        if (ok) ++line_no;
        return ok;
    };
    std::vector<std::string_view> blk;
    std::vector<uint64_t> items;
    orc::RawMesh R;
    std::vector<V3> &vert = R.vert;
    std::vector<char> &vert_present = R.vert_present;
    std::vector<RawFace> &faces = R.faces;
    std::vector<int64_t> &node_pool = R.node_pool;
    int64_t &n_vert = R.n_vert, &n_face = R.n_face;
    std::string zone_name;
    int &dims = R.dims;

    std::string_view header;
    if (!next(header)) return fail(path, 0, "mesh is at least one line long");  // io.rs:75-77
    for (;;) {
        split_ws(header, blk);
        if (blk.empty()) return fail(path, line_no, "empty line where a section header is expected");  // io.rs:82 index panic
        std::string_view b0 = blk[0];
        if (b0 == "(0") {  // comment: the word after the last space names the zone that follows (io.rs:83-90)
            size_t sp = header.rfind(' ');
            if (sp == std::string_view::npos) return fail(path, line_no, "comment has a space");
            std::string_view nm = header.substr(sp + 1);
            while (nm.size() >= 2 && nm[nm.size() - 2] == '"' && nm[nm.size() - 1] == ')') nm.remove_suffix(2);
            zone_name.assign(nm);
        } else if (b0 == "(2") {  // io.rs:92-104
            if (blk.size() < 2) return fail(path, line_no, "dimensions section should have two items");
            std::string_view t = blk[1];
            if (t.empty() || t.back() != ')') return fail(path, line_no, "second item ends with )");
            t.remove_suffix(1);
            uint64_t v = 0;
            size_t i = (!t.empty() && t[0] == '+') ? 1 : 0;
            if (i >= t.size()) return fail(path, line_no, "second item is integer dimension count");
            for (; i < t.size(); ++i) {
                if (t[i] < '0' || t[i] > '9' || v > 255) return fail(path, line_no, "second item is integer dimension count");
                v = v * 10 + (uint64_t)(t[i] - '0');
            }
            if (v != 2 && v != 3) return fail(path, line_no, "Mesh is not 2D or 3D.");
            dims = (int)v;
        } else if (b0 == "(10" || b0 == "(12" || b0 == "(13") {
            if (blk.size() < 2) return fail(path, line_no, "section header has no zone item");  // skip_zone_zero! index panic
            if (blk[1] != "(0") {                                                              // io.rs:24-30
                if (!header_items(header, items)) return fail(path, line_no, "valid hex");
                if (items.size() != 6) return fail(path, line_no, "section header has six items");
                if (b0 == "(12") {  // io.rs:180-193
                    bool seen = false;
                    for (auto &cz : d.cell_zones) seen |= cz.first == items[1];
                    if (!seen) d.cell_zones.emplace_back(items[1], items[4]);
                } else if (b0 == "(10") {  // io.rs:105-175
                    int64_t node_number = (int64_t)items[2];
                    std::string_view l;
                    if (!next(l)) return fail(path, line_no, "node section shouldn't be empty");
                    for (;;) {
                        if (l == "(") {
                            if (!next(l)) return fail(path, line_no, "node section ends after '('");
                            continue;
                        }
                        if (!l.empty() && l[0] == ')') break;
                        split_ws(l, blk);
                        if ((int)blk.size() == dims && dims > 0) {
                            V3 v{0., 0., 0.};
                            if (!parse_f64(blk[0], v.x) || !parse_f64(blk[1], v.y) || (dims == 3 && !parse_f64(blk[2], v.z)))
                                return fail(path, line_no, "should be a string representation of a float");
                            int64_t idx = node_number - 1;
                            if (idx < 0) return fail(path, line_no, "node number 0");
                            if ((size_t)idx >= vert.size()) {
                                size_t ns = std::max<size_t>((size_t)idx + 1, vert.size() * 2);
                                vert.resize(ns);
                                vert_present.resize(ns, 0);
                            }
                            vert[(size_t)idx] = v;
                            if (!vert_present[(size_t)idx]) { vert_present[(size_t)idx] = 1; ++n_vert; }
                        }
                        if (!next(l)) break;
                        ++node_number;  // every line advances the number, parsed or not (io.rs:167-171)
                    }
                } else {  // "(13": io.rs:194-274
                    uint64_t zone_id = items[1], start_index = items[2], boundary_type = items[4], face_type = items[5];
                    int zi = -1;
                    for (size_t k = 0; k < d.zones.size(); ++k)
                        if (d.zones[k].id == zone_id) zi = (int)k;
                    if (zi < 0) {  // entry().or_insert(): try_from is evaluated before the lookup in the reference; a known zone
                                   // id with a bad type code would panic there too
                        zi = (int)d.zones.size();
                        d.zones.push_back({zone_id, (int32_t)boundary_type, 0., {0., 0., 0.}, zone_name});
                    }
                    if (!valid_bc(boundary_type)) return fail(path, line_no, "valid BC type");
                    int64_t face_number = (int64_t)start_index;
                    std::string_view l;
                    if (!next(l)) return fail(path, line_no, "face section has contents");
                    for (;;) {
                        if (l == "(") {
                            if (!next(l)) return fail(path, line_no, "face section ends after '('");
                            continue;
                        }
                        if (!l.empty() && l[0] == ')') break;
                        split_ws(l, blk);
                        if (blk.size() < 2) break;
                        size_t node_count = blk.size() - 2;  // also for face types 0 and 5, whose lines lead with a count (io.rs:232)
                        if (face_type != 0 && face_type != 5 && face_type != node_count) break;
                        int64_t idx = face_number - 1;
                        if (idx < 0) return fail(path, line_no, "face number 0");
                        if ((size_t)idx >= faces.size()) faces.resize(std::max<size_t>((size_t)idx + 1, faces.size() * 2));
                        RawFace &rf = faces[(size_t)idx];
                        if (rf.node_begin < 0) ++n_face;
                        rf.node_begin = (int64_t)node_pool.size();
                        rf.n_nodes = (int32_t)node_count;
                        rf.zone = zi;
                        for (size_t k = 0; k < node_count; ++k) {
                            uint64_t v;
                            if (!parse_hex(blk[k], v)) return fail(path, line_no, "face node is not hexadecimal");
                            node_pool.push_back(v > 0 ? (int64_t)v - 1 : -1);
                        }
                        for (int k = 0; k < 2; ++k) {
                            uint64_t v;
                            if (!parse_hex(blk[node_count + (size_t)k], v)) return fail(path, line_no, "face cell is not hexadecimal");
                            rf.c[k] = v > 0 ? (int64_t)v - 1 : -1;  // usize::MAX in the reference
                        }
                        if (!next(l)) break;
                        ++face_number;
                    }
                }
            }
        }
        // "(1", "(18", "(58", "(59", "(61" and everything else: nothing the Mesh keeps (io.rs:91, 176-179, 275-278)
        if (!next(header)) break;
    }
    if (dims == 0) return fail(path, line_no, "no dimension section before the geometry");
    return orc::mesh_finalize_geometry(path, line_no, R, d);
}

}  // namespace

// ---- io.rs:289-438: faces in ascending number, then the cells (shared with the generators' in-memory form, mesh_raw.hpp)
int orc::mesh_finalize_geometry(const char *path, int64_t line_no, orc::RawMesh &R, OrcMeshData &d) {
    const std::vector<V3> &vert = R.vert;
    const std::vector<char> &vert_present = R.vert_present;
    const std::vector<RawFace> &faces = R.faces;
    const std::vector<int64_t> &node_pool = R.node_pool;
    const int64_t n_vert = R.n_vert, n_face = R.n_face;
    const int dims = R.dims;
    d.dimensions = dims;
    d.n_vertices = n_vert;
    d.n_faces = n_face;
    for (int64_t i = 0; i < n_vert; ++i)
        if (!vert_present[(size_t)i]) return fail(path, line_no, "node numbers are not contiguous");
    d.vertex.resize((size_t)3 * n_vert);
    for (int64_t i = 0; i < n_vert; ++i) {
        d.vertex[3 * i] = vert[(size_t)i].x;
        d.vertex[3 * i + 1] = vert[(size_t)i].y;
        d.vertex[3 * i + 2] = vert[(size_t)i].z;
    }
    const size_t F = (size_t)n_face;
    d.face_c0.resize(F); d.face_c1.resize(F); d.face_zone.resize(F); d.face_area.resize(F);
    d.face_normal.resize(3 * F); d.face_centroid.resize(3 * F); d.face_node_ptr.assign(F + 1, 0);
    int64_t n_cell_ids = 0, max_cell = -1;
    std::vector<int64_t> cell_count;
    auto node = [&](const RawFace &rf, int k) -> const V3 & { return vert[(size_t)node_pool[(size_t)rf.node_begin + (size_t)k]]; };
    for (size_t f = 0; f < F; ++f) {
        if (f >= faces.size() || faces[f].node_begin < 0) return fail(path, line_no, "face numbers are not contiguous");
        const RawFace &rf = faces[f];
        if (rf.n_nodes < dims) return fail(path, line_no, "face has too few nodes");
        for (int k = 0; k < rf.n_nodes; ++k) {
            int64_t nd = node_pool[(size_t)rf.node_begin + (size_t)k];
            if (nd < 0 || nd >= (int64_t)vert.size() || !vert_present[(size_t)nd]) return fail(path, line_no, "nodes should have all been read");
            d.face_nodes.push_back(nd);
        }
        d.face_node_ptr[f + 1] = (int64_t)d.face_nodes.size();
        V3 nrm;
        if (dims == 2) {  // io.rs:305-321
            V3 t = sub(node(rf, 1), node(rf, 0));
            nrm = t.x == 0. ? unit(V3{1., -t.x / t.y, 0.}) : unit(V3{-t.y / t.x, 1., 0.});
        } else {  // io.rs:322-326
            nrm = unit(cross(sub(node(rf, 2), node(rf, 1)), sub(node(rf, 1), node(rf, 0))));
        }
        int64_t c0 = rf.c[0], c1 = rf.c[1];
        if (c0 < 0) {  // io.rs:332-337: the TGRID normal points at cell 0; without one, flip and keep the other cell
            nrm = neg(nrm);
            c0 = c1;
            c1 = -1;
        }
        V3 cen{0., 0., 0.};  // io.rs:338-342
        for (int k = 0; k < rf.n_nodes; ++k) cen = add(cen, node(rf, k));
        cen = over(cen, (double)rf.n_nodes);
        double area;
        if (rf.n_nodes == 2) {  // io.rs:345-349
            if (dims != 2) return fail(path, line_no, "two-node face in a 3-D mesh");
            area = norm(sub(node(rf, 1), node(rf, 0)));
        } else {  // io.rs:375-397: triangle fan about the centroid, closing triangle last
            auto tri = [&](const V3 &a, const V3 &b) { return std::fabs(norm(cross(sub(a, cen), sub(b, cen)))) / 2.; };
            area = 0.;
            for (int k = 0; k + 1 < rf.n_nodes; ++k) area = area + tri(node(rf, k), node(rf, k + 1));
            area = area + tri(node(rf, 0), node(rf, rf.n_nodes - 1));
        }
        d.face_c0[f] = c0; d.face_c1[f] = c1; d.face_zone[f] = rf.zone; d.face_area[f] = area;
        d.face_normal[3 * f] = nrm.x; d.face_normal[3 * f + 1] = nrm.y; d.face_normal[3 * f + 2] = nrm.z;
        d.face_centroid[3 * f] = cen.x; d.face_centroid[3 * f + 1] = cen.y; d.face_centroid[3 * f + 2] = cen.z;
        for (int64_t c : {c0, c1}) {
            if (c < 0) continue;
            if (c > max_cell) max_cell = c;
            if ((size_t)c >= cell_count.size()) cell_count.resize(std::max<size_t>((size_t)c + 1, cell_count.size() * 2), 0);
            if (cell_count[(size_t)c]++ == 0) ++n_cell_ids;
        }
    }
    if (max_cell + 1 != n_cell_ids) return fail(path, line_no, "cell numbers are not contiguous");  // io.rs:418 unwrap
    const size_t N = (size_t)n_cell_ids;
    d.n_cells = n_cell_ids;
    d.cell_face_ptr.assign(N + 1, 0);
    for (size_t c = 0; c < N; ++c) d.cell_face_ptr[c + 1] = d.cell_face_ptr[c] + cell_count[c];
    d.cell_faces.resize((size_t)d.cell_face_ptr[N]);
    d.cell_centroid.assign(3 * N, 0.);
    d.cell_volume.assign(N, 0.);
    std::vector<int64_t> fill(d.cell_face_ptr.begin(), d.cell_face_ptr.end() - 1);
    for (size_t f = 0; f < F; ++f) {  // io.rs:404-414: ascending face id; centroid sums face centroids in that order
        for (int64_t c : {d.face_c0[f], d.face_c1[f]}) {
            if (c < 0) continue;
            d.cell_faces[(size_t)fill[(size_t)c]++] = (int64_t)f;
            for (int k = 0; k < 3; ++k) d.cell_centroid[3 * (size_t)c + k] = d.cell_centroid[3 * (size_t)c + k] + d.face_centroid[3 * f + k];
        }
    }
    for (size_t c = 0; c < N; ++c) {  // io.rs:417-438
        int64_t nf = d.cell_face_ptr[c + 1] - d.cell_face_ptr[c];
        V3 cc = over(V3{d.cell_centroid[3 * c], d.cell_centroid[3 * c + 1], d.cell_centroid[3 * c + 2]}, (double)nf);
        d.cell_centroid[3 * c] = cc.x; d.cell_centroid[3 * c + 1] = cc.y; d.cell_centroid[3 * c + 2] = cc.z;
        if (nf < dims + 1) return fail(path, line_no, "cell has too few faces");
        double vol = 0.;
        for (int64_t q = d.cell_face_ptr[c]; q < d.cell_face_ptr[c + 1]; ++q) {
            size_t f = (size_t)d.cell_faces[(size_t)q];
            V3 fc{d.face_centroid[3 * f], d.face_centroid[3 * f + 1], d.face_centroid[3 * f + 2]};
            V3 fn{d.face_normal[3 * f], d.face_normal[3 * f + 1], d.face_normal[3 * f + 2]};
            vol = vol + d.face_area[f] * std::fabs(dot(sub(fc, cc), fn)) / (double)dims;
        }
        d.cell_volume[c] = vol;
    }
    return ORC_OK;
}

namespace {

bool slurp(const char *path, std::string &out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    bool ok = false;
    if (fseek(fp, 0, SEEK_END) == 0) {
        long sz = ftell(fp);
        if (sz >= 0 && fseek(fp, 0, SEEK_SET) == 0) {
            out.resize((size_t)sz);
            ok = sz == 0 || fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
        }
    }
    fclose(fp);
    return ok;
}

// ---- Rust float formatting (core::fmt::LowerExp): shortest round-trip digits when no precision is given
// ("{:e}"; "{:.e}" of io.rs:587 carries an empty precision and formats the same way), else exactly `precision`
// fraction digits, round-half-even on the exact binary value; exponent without '+' or leading zeros.
void append_lower_exp(std::string &out, double v, int precision) {
    if (std::isnan(v)) { out += "NaN"; return; }
    if (std::isinf(v)) { out += v < 0 ? "-inf" : "inf"; return; }
    char buf[64];
    int len;
    if (precision < 0) {
        // shortest digits: the first %.{p}e that reads back exactly (p <= 16 always suffices for f64)
        len = 0;
        for (int p = 0; p <= 16; ++p) {
            len = snprintf(buf, sizeof(buf), "%.*e", p, v);
            if (strtod(buf, nullptr) == v) break;
        }
    } else {
        len = snprintf(buf, sizeof(buf), "%.*e", precision, v);
    }
    char *e = (char *)memchr(buf, 'e', (size_t)len);
    out.append(buf, (size_t)(e - buf));
    out += 'e';
    ++e;
    if (*e == '-') out += '-';
    ++e;  // sign
    while (*e == '0' && e[1]) ++e;
    out += e;
}

void append_vector_2e(std::string &out, const double *c) {  // Display for Vector (lib.rs:551-555)
    out += '(';
    append_lower_exp(out, c[0], 2);
    out += ", ";
    append_lower_exp(out, c[1], 2);
    out += ", ";
    append_lower_exp(out, c[2], 2);
    out += ')';
}

int write_text(const char *path, const std::string &s) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return set_error(ORC_ERR_IO, "cannot create %s", path);
    bool ok = fwrite(s.data(), 1, s.size(), fp) == s.size();
    ok = (fclose(fp) == 0) && ok;
    return ok ? ORC_OK : set_error(ORC_ERR_IO, "short write to %s", path);
}

}  // namespace

extern "C" {

OrcMeshData *orc_read_mesh(const char *path, int *status) {
    std::string text;
    int st = ORC_OK;
    OrcMeshData *d = nullptr;
    if (!path || !slurp(path, text)) st = set_error(ORC_ERR_IO, "Unable to open mesh file for reading. (%s)", path ? path : "null");  // io.rs:286
    else {
        d = new OrcMeshData();
        st = parse_msh(path, text, *d);
        if (st != ORC_OK) { delete d; d = nullptr; }
    }
    if (status) *status = st;
    return d;
}

void orc_mesh_data_destroy(OrcMeshData *d) { delete d; }

int orc_mesh_data_sizes(const OrcMeshData *d, int32_t *dimensions, int64_t *n_vertices, int64_t *n_cells, int64_t *n_faces,
                        int64_t *n_cell_faces, int64_t *n_face_nodes, int32_t *n_zones) {
    if (!d) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh data");
    if (dimensions) *dimensions = d->dimensions;
    if (n_vertices) *n_vertices = d->n_vertices;
    if (n_cells) *n_cells = d->n_cells;
    if (n_faces) *n_faces = d->n_faces;
    if (n_cell_faces) *n_cell_faces = (int64_t)d->cell_faces.size();
    if (n_face_nodes) *n_face_nodes = (int64_t)d->face_nodes.size();
    if (n_zones) *n_zones = (int32_t)d->zones.size();
    return ORC_OK;
}

#define COPY_OUT(dst, src) \
    if (dst) std::copy((src).begin(), (src).end(), dst)

int orc_mesh_data_arrays(const OrcMeshData *d, int64_t *face_c0, int64_t *face_c1, int32_t *face_zone, double *face_area,
                         double *face_normal, double *face_centroid, double *cell_centroid, double *cell_volume,
                         int64_t *cell_face_ptr, int64_t *cell_faces) {
    if (!d) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh data");
    COPY_OUT(face_c0, d->face_c0); COPY_OUT(face_c1, d->face_c1); COPY_OUT(face_zone, d->face_zone); COPY_OUT(face_area, d->face_area);
    COPY_OUT(face_normal, d->face_normal); COPY_OUT(face_centroid, d->face_centroid); COPY_OUT(cell_centroid, d->cell_centroid);
    COPY_OUT(cell_volume, d->cell_volume); COPY_OUT(cell_face_ptr, d->cell_face_ptr); COPY_OUT(cell_faces, d->cell_faces);
    return ORC_OK;
}

int orc_mesh_data_nodes(const OrcMeshData *d, double *vertices, int64_t *face_node_ptr, int64_t *face_nodes) {
    if (!d) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh data");
    COPY_OUT(vertices, d->vertex); COPY_OUT(face_node_ptr, d->face_node_ptr); COPY_OUT(face_nodes, d->face_nodes);
    return ORC_OK;
}

int orc_mesh_data_zone(const OrcMeshData *d, int32_t k, uint64_t *zone_id, int32_t *zone_type, double *scalar_value, double *vector_value,
                       char *name, int64_t name_len) {
    if (!d || k < 0 || (size_t)k >= d->zones.size()) return set_error(ORC_ERR_BAD_ARGUMENT, "zone index out of range");
    const auto &z = d->zones[(size_t)k];
    if (zone_id) *zone_id = z.id;
    if (zone_type) *zone_type = z.type;
    if (scalar_value) *scalar_value = z.scalar;
    if (vector_value) { vector_value[0] = z.vec[0]; vector_value[1] = z.vec[1]; vector_value[2] = z.vec[2]; }
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", z.name.c_str());
    return ORC_OK;
}

int orc_mesh_data_zone_index(const OrcMeshData *d, const char *name) {
    if (!d || !name) return -1;
    for (size_t k = 0; k < d->zones.size(); ++k)
        if (d->zones[k].name == name) return (int)k;
    return -1;
}

int orc_mesh_data_set_zone(OrcMeshData *d, const char *name, int32_t zone_type, double scalar_value, const double *vector_value) {
    int k = orc_mesh_data_zone_index(d, name);
    if (k < 0) return set_error(ORC_ERR_ZONE_NOT_FOUND, "face zone '%s' should exist in mesh", name ? name : "");  // mesh.rs:194
    if (!valid_bc((uint64_t)zone_type)) return set_error(ORC_ERR_BAD_ARGUMENT, "Invalid boundary condition value.");  // mesh.rs:66
    auto &z = d->zones[(size_t)k];
    z.type = zone_type;
    z.scalar = scalar_value;
    for (int i = 0; i < 3; ++i) z.vec[i] = vector_value ? vector_value[i] : 0.;
    return ORC_OK;
}

OrcMesh *orc_mesh_upload(const OrcMeshData *d, int *status) {
    int st = ORC_OK;
    OrcMesh *m = nullptr;
    if (!d) st = set_error(ORC_ERR_BAD_ARGUMENT, "null mesh data");
    else {
        std::vector<int32_t> zt;
        std::vector<double> zs, zv;
        for (const auto &z : d->zones) {
            zt.push_back(z.type);
            zs.push_back(z.scalar);
            zv.insert(zv.end(), z.vec, z.vec + 3);
        }
        m = orc_mesh_create(d->n_cells, d->n_faces, (int32_t)d->zones.size(), d->face_c0.data(), d->face_c1.data(), d->face_zone.data(),
                            d->face_area.data(), d->face_normal.data(), d->face_centroid.data(), d->cell_centroid.data(),
                            d->cell_volume.data(), d->cell_face_ptr.data(), d->cell_faces.data(), zt.data(), zs.data(), zv.data(), &st);
    }
    if (status) *status = st;
    return m;
}

int orc_mesh_sync_zones(OrcMesh *m, const OrcMeshData *d) {
    if (!m || !d) return set_error(ORC_ERR_BAD_ARGUMENT, "null mesh");
    std::vector<int32_t> zt;
    std::vector<double> zs, zv;
    for (const auto &z : d->zones) {
        zt.push_back(z.type);
        zs.push_back(z.scalar);
        zv.insert(zv.end(), z.vec, z.vec + 3);
    }
    if ((int32_t)zt.size() != m->n_zones) return set_error(ORC_ERR_BAD_ARGUMENT, "zone count differs from the uploaded mesh");
    return orc_mesh_update_zones(m, zt.data(), zs.data(), zv.data());
}

// ---- io.rs:573-620
int orc_write_data(const char *path, int64_t n_cells, const double *cell_centroid, const double *u, const double *v, const double *w,
                   const double *p, int decimal_precision) {
    if (!path || n_cells < 0 || (n_cells > 0 && (!cell_centroid || !u || !v || !w || !p))) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    std::string out;
    out.reserve((size_t)n_cells * 96);
    for (int64_t i = 0; i < n_cells; ++i) {
        append_vector_2e(out, cell_centroid + 3 * i);
        out += "\t(";
        append_lower_exp(out, u[i], decimal_precision);
        out += ", ";
        append_lower_exp(out, v[i], decimal_precision);
        out += ", ";
        append_lower_exp(out, w[i], decimal_precision);
        out += ")\t";
        append_lower_exp(out, p[i], decimal_precision);
        out += '\n';
    }
    return write_text(path, out);
}

// ---- io.rs:519-571.  Returns the number of rows through n_read; fills at most `capacity` rows (pass 0 to count).
int orc_read_data(const char *path, int64_t capacity, double *u, double *v, double *w, double *p, int64_t *n_read) {
    std::string text;
    if (!path || !slurp(path, text)) return set_error(ORC_ERR_IO, "could not read data file");  // io.rs:569
    LineCursor cur{text.data(), text.data() + text.size()};
    std::string_view line;
    int64_t row = 0;
    while (cur.next(line)) {
        // splitn(3, '\t'): [centroid (ignored)] [ (u, v, w) ] [ p ]
        size_t t1 = line.find('\t');
        if (t1 != std::string_view::npos) {
            size_t t2 = line.find('\t', t1 + 1);
            std::string_view vec = t2 == std::string_view::npos ? line.substr(t1 + 1) : line.substr(t1 + 1, t2 - t1 - 1);
            // Vector::parse (lib.rs:319-333): "(x, y, z)", splitn(3, ", ")
            double x[3];
            if (vec.size() < 2 || vec.front() != '(' || vec.back() != ')') return set_error(ORC_ERR_MESH_FORMAT, "%s:%lld: vector is not parenthesised", path, (long long)row + 1);
            vec = vec.substr(1, vec.size() - 2);
            size_t a = vec.find(", ");
            size_t b = a == std::string_view::npos ? a : vec.find(", ", a + 2);
            if (b == std::string_view::npos || !parse_f64(vec.substr(0, a), x[0]) || !parse_f64(vec.substr(a + 2, b - a - 2), x[1]) ||
                !parse_f64(vec.substr(b + 2), x[2]))
                return set_error(ORC_ERR_MESH_FORMAT, "%s:%lld: vector does not parse", path, (long long)row + 1);
            double pv = 0.;
            bool has_p = t2 != std::string_view::npos;
            if (has_p && !parse_f64(line.substr(t2 + 1), pv)) return set_error(ORC_ERR_MESH_FORMAT, "%s:%lld: pressure does not parse", path, (long long)row + 1);
            if (!has_p) return set_error(ORC_ERR_MESH_FORMAT, "%s:%lld: three tab-separated columns expected", path, (long long)row + 1);
            if (row < capacity) {
                if (u) u[row] = x[0];
                if (v) v[row] = x[1];
                if (w) w[row] = x[2];
                if (p) p[row] = pv;
            }
            ++row;
        }
        // a line without a tab has only column 0: nothing is pushed (io.rs:545-546)
    }
    if (n_read) *n_read = row;
    return ORC_OK;
}

// ---- io.rs:623-662: gradients of every cell with the device Green-Gauss kernels; the reference's
// strip_suffix(", ") result is discarded (io.rs:644, 654), so each list keeps its trailing ", ".
int orc_write_gradients(const OrcMesh *m, const double *cell_centroid, const double *u, const double *v, const double *w, const double *p,
                        const char *path, int decimal_precision, const OrcSettings *settings) {
    if (!m || !path || !cell_centroid) return set_error(ORC_ERR_BAD_ARGUMENT, "null argument");
    const int64_t n = m->n_cells;
    std::vector<double> gp((size_t)3 * n), gu((size_t)9 * n);
    ORC_TRY(orc_calculate_gradients(m, u, v, w, p, settings, gp.data(), gu.data()));
    std::string out;
    out.reserve((size_t)n * 200);
    for (int64_t i = 0; i < n; ++i) {
        append_vector_2e(out, cell_centroid + 3 * i);
        out += "\t(";
        for (int k = 0; k < 9; ++k) {  // Tensor::flatten: rows x, y, z (lib.rs:600-605)
            append_lower_exp(out, gu[(size_t)9 * i + k], decimal_precision);
            out += ", ";
        }
        out += ")\t(";
        for (int k = 0; k < 3; ++k) {
            append_lower_exp(out, gp[(size_t)3 * i + k], decimal_precision);
            out += ", ";
        }
        out += ")\n";
    }
    return write_text(path, out);
}

}  // extern "C"
