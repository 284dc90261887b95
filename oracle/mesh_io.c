/* mesh_io.c — restatement of io::read_mesh (src/io.rs:32-515): ASCII TGRID/Fluent .msh
 * parser and the face/cell geometry rules, plus mesh::Mesh::get_face_zone (mesh.rs:189-195).
 * Test infrastructure (see oracle.h).
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "vec3.h"

typedef struct { char **line; int64_t n, pos; } Lines;

static Lines read_lines(const char *path) {
    Lines L = {NULL, 0, 0};
    FILE *f = fopen(path, "rb");
    if (!f) return L;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return L; }
    fclose(f);
    buf[sz] = 0;
    int64_t cap = 1024;
    L.line = (char **)malloc(sizeof(char *) * (size_t)cap);
    char *s = buf;
    /* BufRead::lines(): split on '\n', strip a trailing '\r'; a final empty piece is not a line */
    while (*s) {
        char *e = strchr(s, '\n');
        if (L.n == cap) { cap *= 2; L.line = (char **)realloc(L.line, sizeof(char *) * (size_t)cap); }
        L.line[L.n++] = s;
        if (!e) break;
        *e = 0;
        if (e > s && e[-1] == '\r') e[-1] = 0;
        s = e + 1;
    }
    return L; /* buf is owned by L.line[0] */
}

static void free_lines(Lines *L) {
    if (L->line) { if (L->n > 0) free(L->line[0]); free(L->line); }
}

static char *next_line(Lines *L) { return L->pos < L->n ? L->line[L->pos++] : NULL; }

/* split_ascii_whitespace into at most cap blocks (copies pointers into a scratch copy) */
static int split_ws(const char *line, char *scratch, size_t scratch_sz, char **blocks, int cap) {
    strncpy(scratch, line, scratch_sz - 1);
    scratch[scratch_sz - 1] = 0;
    int n = 0;
    char *s = scratch;
    while (*s) {
        while (*s && isspace((unsigned char)*s)) s++;
        if (!*s) break;
        if (n < cap) blocks[n] = s;
        n++;
        while (*s && !isspace((unsigned char)*s)) s++;
        if (*s) *s++ = 0;
    }
    return n;
}

/* read_section_header_common (io.rs:47-54): every maximal run of [0-9a-z] parsed as hex */
static int header_items(const char *line, uint64_t *items, int cap) {
    int n = 0;
    const char *s = line;
    while (*s) {
        while (*s && !((*s >= '0' && *s <= '9') || (*s >= 'a' && *s <= 'z'))) s++;
        if (!*s) break;
        const char *b = s;
        while ((*s >= '0' && *s <= '9') || (*s >= 'a' && *s <= 'z')) s++;
        char tok[32];
        size_t len = (size_t)(s - b);
        if (len >= sizeof(tok)) return -1;
        memcpy(tok, b, len);
        tok[len] = 0;
        char *end;
        uint64_t v = strtoull(tok, &end, 16);
        if (*end) return -1; /* .expect("valid hex") */
        if (n < cap) items[n] = v;
        n++;
    }
    return n;
}

static int valid_bc(uint64_t t) {
    switch (t) { /* mesh.rs:51-65 */
    case 2: case 3: case 4: case 5: case 7: case 8: case 9: case 10: case 12: case 14: case 20: case 24: case 31: case 36: case 37: return 1;
    default: return 0;
    }
}

void or_mesh_free(OrMesh *m) {
    if (!m) return;
    free(m->vertices); free(m->face_zone); free(m->face_c0); free(m->face_c1); free(m->face_node_ptr);
    free(m->face_nodes); free(m->face_area); free(m->face_centroid); free(m->face_normal);
    free(m->cell_face_ptr); free(m->cell_faces); free(m->cell_volume); free(m->cell_centroid); free(m->zones);
    free(m);
}

typedef struct { int64_t *nodes; int nn; int64_t c[2]; int nc; int zone; int present; } RawFace;

OrMesh *or_read_mesh(const char *path) {
    Lines L = read_lines(path);
    if (!L.line) return NULL; /* io.rs:285-287 */
    int dimensions = 0;
    char zone_name[64] = "";
    char scratch[4096];
    char *blocks[64];
    uint64_t items[16];

    int64_t vcap = 1024, nv = 0;
    Vec3 *verts = (Vec3 *)malloc(sizeof(Vec3) * (size_t)vcap);
    char *vpresent = (char *)calloc((size_t)vcap, 1);
    int64_t fcap = 1024, nf = 0;
    RawFace *faces = (RawFace *)calloc((size_t)fcap, sizeof(RawFace));
    int zcap = 16, nz = 0;
    OrZone *zones = (OrZone *)calloc((size_t)zcap, sizeof(OrZone));
    int ok = 1;

    char *header = next_line(&L);
    while (header && ok) {
        int nb = split_ws(header, scratch, sizeof(scratch), blocks, 64);
        if (nb > 0) {
            const char *b0 = blocks[0];
            if (strcmp(b0, "(0") == 0) {
                /* io.rs:83-90: text after the last space, minus trailing `")` characters */
                const char *sp = strrchr(header, ' ');
                if (sp) {
                    strncpy(zone_name, sp + 1, sizeof(zone_name) - 1);
                    zone_name[sizeof(zone_name) - 1] = 0;
                    size_t len = strlen(zone_name);
                    while (len >= 2 && zone_name[len - 2] == '"' && zone_name[len - 1] == ')') { zone_name[len - 2] = 0; len -= 2; }
                }
            } else if (strcmp(b0, "(2") == 0) {
                /* io.rs:92-104 */
                if (nb < 2) { ok = 0; break; }
                dimensions = atoi(blocks[1]);
                if (dimensions != 2 && dimensions != 3) { ok = 0; break; }
            } else if (strcmp(b0, "(10") == 0) {
                /* io.rs:105-175 */
                if (!(nb > 1 && strcmp(blocks[1], "(0") == 0)) {
                    int ni = header_items(header, items, 16);
                    if (ni != 6) { ok = 0; break; }
                    int64_t node_number = (int64_t)items[2];
                    char *cur = next_line(&L);
                    while (cur) {
                        if (strcmp(cur, "(") == 0) { cur = next_line(&L); continue; }
                        if (cur[0] == ')') break;
                        char sc2[512];
                        char *bl[8];
                        int n2 = split_ws(cur, sc2, sizeof(sc2), bl, 8);
                        if (n2 == dimensions) {
                            double x = strtod(bl[0], NULL), y = strtod(bl[1], NULL), z = 0.;
                            if (dimensions == 3) z = strtod(bl[2], NULL);
                            int64_t idx = node_number - 1;
                            while (idx >= vcap) {
                                verts = (Vec3 *)realloc(verts, sizeof(Vec3) * (size_t)vcap * 2);
                                vpresent = (char *)realloc(vpresent, (size_t)vcap * 2);
                                memset(vpresent + vcap, 0, (size_t)vcap);
                                vcap *= 2;
                            }
                            verts[idx] = v3(x, y, z);
                            if (!vpresent[idx]) { vpresent[idx] = 1; nv++; }
                        }
                        cur = next_line(&L);
                        if (!cur) break;
                        node_number++;
                    }
                }
            } else if (strcmp(b0, "(13") == 0) {
                /* io.rs:194-274 */
                if (!(nb > 1 && strcmp(blocks[1], "(0") == 0)) {
                    int ni = header_items(header, items, 16);
                    if (ni != 6) { ok = 0; break; }
                    uint64_t zone_id = items[1], start_index = items[2], boundary_type = items[4], face_type = items[5];
                    int zi = -1;
                    for (int k = 0; k < nz; k++) if (zones[k].id == zone_id) zi = k;
                    if (zi < 0) {
                        if (!valid_bc(boundary_type)) { ok = 0; break; } /* .expect("valid BC type") */
                        if (nz == zcap) { zcap *= 2; zones = (OrZone *)realloc(zones, sizeof(OrZone) * (size_t)zcap); }
                        zi = nz++;
                        memset(&zones[zi], 0, sizeof(OrZone));
                        zones[zi].id = zone_id;
                        zones[zi].zone_type = (int32_t)boundary_type;
                        strncpy(zones[zi].name, zone_name, sizeof(zones[zi].name) - 1);
                    }
                    int64_t face_number = (int64_t)start_index;
                    char *cur = next_line(&L);
                    while (cur) {
                        if (strcmp(cur, "(") == 0) { cur = next_line(&L); continue; }
                        if (cur[0] == ')') break;
                        char sc2[1024];
                        char *bl[64];
                        int n2 = split_ws(cur, sc2, sizeof(sc2), bl, 64);
                        if (n2 < 2) break;
                        int node_count = n2 - 2; /* io.rs:232 — also for face_type 0/5 lines (SURVEY C5 reader hazard) */
                        if (face_type != 0 && face_type != 5 && face_type != (uint64_t)node_count) break;
                        int64_t idx = face_number - 1;
                        while (idx >= fcap) {
                            faces = (RawFace *)realloc(faces, sizeof(RawFace) * (size_t)fcap * 2);
                            memset(faces + fcap, 0, sizeof(RawFace) * (size_t)fcap);
                            fcap *= 2;
                        }
                        RawFace *rf = &faces[idx];
                        if (!rf->present) nf++;
                        free(rf->nodes);
                        rf->present = 1;
                        rf->zone = zi;
                        rf->nn = node_count;
                        rf->nodes = (int64_t *)malloc(sizeof(int64_t) * (size_t)(node_count > 0 ? node_count : 1));
                        for (int k = 0; k < node_count; k++) {
                            uint64_t nn = strtoull(bl[k], NULL, 16);
                            rf->nodes[k] = nn > 0 ? (int64_t)nn - 1 : -1;
                        }
                        rf->nc = 2;
                        for (int k = 0; k < 2; k++) {
                            uint64_t cn = strtoull(bl[node_count + k], NULL, 16);
                            rf->c[k] = cn > 0 ? (int64_t)cn - 1 : -1; /* usize::MAX in the reference */
                        }
                        cur = next_line(&L);
                        if (!cur) break;
                        face_number++;
                    }
                }
            }
            /* "(1", "(12", "(18", "(58", "(59", "(61", anything else: no effect on the Mesh used by the hot path */
        }
        header = next_line(&L);
    }
    OrMesh *m = NULL;
    if (ok && nf > 0 && dimensions != 0) {
        m = (OrMesh *)calloc(1, sizeof(OrMesh));
        m->dimensions = dimensions;
        m->n_vertices = nv; m->n_faces = nf; m->n_zones = nz;
        m->vertices = (Vec3 *)malloc(sizeof(Vec3) * (size_t)(nv > 0 ? nv : 1));
        memcpy(m->vertices, verts, sizeof(Vec3) * (size_t)nv);
        m->zones = (OrZone *)malloc(sizeof(OrZone) * (size_t)(nz > 0 ? nz : 1));
        memcpy(m->zones, zones, sizeof(OrZone) * (size_t)nz);
        m->face_zone = (int32_t *)malloc(sizeof(int32_t) * (size_t)nf);
        m->face_c0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nf);
        m->face_c1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nf);
        m->face_node_ptr = (int64_t *)calloc((size_t)nf + 1, sizeof(int64_t));
        m->face_area = (double *)malloc(sizeof(double) * (size_t)nf);
        m->face_centroid = (Vec3 *)malloc(sizeof(Vec3) * (size_t)nf);
        m->face_normal = (Vec3 *)malloc(sizeof(Vec3) * (size_t)nf);
        int64_t tot_nodes = 0, max_cell = -1;
        for (int64_t f = 0; f < nf && ok; f++) {
            if (!faces[f].present) { ok = 0; break; } /* faces_hashmap.get_mut(&face_index).unwrap() */
            tot_nodes += faces[f].nn;
            m->face_node_ptr[f + 1] = tot_nodes;
        }
        m->face_nodes = (int64_t *)malloc(sizeof(int64_t) * (size_t)(tot_nodes > 0 ? tot_nodes : 1));
        /* io.rs:289-415 face geometry */
        for (int64_t f = 0; f < nf && ok; f++) {
            RawFace *rf = &faces[f];
            if (rf->nn < dimensions) { ok = 0; break; } /* "face has too few nodes" */
            for (int k = 0; k < rf->nn; k++) {
                if (rf->nodes[k] < 0 || rf->nodes[k] >= nv) { ok = 0; break; }
                m->face_nodes[m->face_node_ptr[f] + k] = rf->nodes[k];
            }
            if (!ok) break;
            Vec3 n0 = verts[rf->nodes[0]], n1 = verts[rf->nodes[1]];
            Vec3 normal;
            if (dimensions == 2) { /* io.rs:305-321 */
                Vec3 t = v_sub(n1, n0);
                if (t.x == 0.) normal = v_unit(v3(1., -t.x / t.y, 0.));
                else normal = v_unit(v3(-t.y / t.x, 1., 0.));
            } else { /* io.rs:322-326 */
                Vec3 n2 = verts[rf->nodes[2]];
                normal = v_unit(v_cross(v_sub(n2, n1), v_sub(n1, n0)));
            }
            int64_t c0 = rf->c[0], c1 = rf->c[1];
            if (c0 < 0) { normal = v_neg(normal); c0 = c1; c1 = -1; } /* io.rs:332-337 */
            m->face_c0[f] = c0; m->face_c1[f] = c1;
            m->face_normal[f] = normal;
            m->face_zone[f] = rf->zone;
            Vec3 cen = v_zero(); /* io.rs:338-342 */
            for (int k = 0; k < rf->nn; k++) cen = v_add(cen, verts[rf->nodes[k]]);
            cen = v_divs(cen, (double)rf->nn);
            m->face_centroid[f] = cen;
            double area;
            if (rf->nn == 2) { /* io.rs:345-349 */
                if (dimensions != 2) { ok = 0; break; }
                area = v_norm(v_sub(n1, n0));
            } else { /* io.rs:375-397 triangle fan about the centroid */
                area = 0.;
                for (int k = 0; k + 1 < rf->nn; k++) {
                    Vec3 a = verts[rf->nodes[k]], b = verts[rf->nodes[k + 1]];
                    area = area + fabs(v_norm(v_cross(v_sub(a, cen), v_sub(b, cen)))) / 2.;
                }
                Vec3 first = verts[rf->nodes[0]], last = verts[rf->nodes[rf->nn - 1]];
                area = area + fabs(v_norm(v_cross(v_sub(first, cen), v_sub(last, cen)))) / 2.;
            }
            m->face_area[f] = area;
            if (c0 > max_cell) max_cell = c0;
            if (c1 > max_cell) max_cell = c1;
        }
        if (ok) {
            int64_t nc = max_cell + 1;
            m->n_cells = nc;
            m->cell_face_ptr = (int64_t *)calloc((size_t)nc + 1, sizeof(int64_t));
            m->cell_volume = (double *)calloc((size_t)(nc > 0 ? nc : 1), sizeof(double));
            m->cell_centroid = (Vec3 *)calloc((size_t)(nc > 0 ? nc : 1), sizeof(Vec3));
            for (int64_t f = 0; f < nf; f++) {
                m->cell_face_ptr[m->face_c0[f] + 1]++;
                if (m->face_c1[f] >= 0) m->cell_face_ptr[m->face_c1[f] + 1]++;
            }
            for (int64_t c = 0; c < nc; c++) {
                if (m->cell_face_ptr[c + 1] == 0) ok = 0; /* cells_hashmap.get_mut(&cell_index).unwrap() */
                m->cell_face_ptr[c + 1] += m->cell_face_ptr[c];
            }
            m->cell_faces = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->cell_face_ptr[nc] > 0 ? m->cell_face_ptr[nc] : 1));
            int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nc + 1));
            memcpy(pos, m->cell_face_ptr, sizeof(int64_t) * (size_t)(nc + 1));
            /* io.rs:404-414: faces visited in ascending id; centroid accumulates face centroids in that order */
            for (int64_t f = 0; f < nf; f++) {
                int64_t cs[2] = {m->face_c0[f], m->face_c1[f]};
                for (int k = 0; k < 2; k++) {
                    if (cs[k] < 0) continue;
                    m->cell_faces[pos[cs[k]]++] = f;
                    m->cell_centroid[cs[k]] = v_add(m->cell_centroid[cs[k]], m->face_centroid[f]);
                }
            }
            free(pos);
            /* io.rs:417-438 */
            for (int64_t c = 0; c < nc && ok; c++) {
                int64_t nfc = m->cell_face_ptr[c + 1] - m->cell_face_ptr[c];
                m->cell_centroid[c] = v_divs(m->cell_centroid[c], (double)nfc);
                if (nfc < dimensions + 1) { ok = 0; break; } /* "cell has too few faces" */
                double vol = 0.;
                for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
                    int64_t f = m->cell_faces[q];
                    vol = vol + m->face_area[f] * fabs(v_dot(v_sub(m->face_centroid[f], m->cell_centroid[c]), m->face_normal[f])) / (double)dimensions;
                }
                m->cell_volume[c] = vol;
            }
        }
        if (!ok) { or_mesh_free(m); m = NULL; }
    }
    for (int64_t f = 0; f < fcap; f++) free(faces[f].nodes);
    free(faces); free(verts); free(vpresent); free(zones);
    free_lines(&L);
    return m;
}

OrMesh *or_mesh_from_arrays(int32_t dimensions, int64_t n_cells, int64_t n_faces, int32_t n_zones,
                            const int64_t *face_c0, const int64_t *face_c1, const int32_t *face_zone,
                            const double *face_area, const double *face_normal, const double *face_centroid,
                            const double *cell_centroid, const double *cell_volume,
                            const int64_t *cell_face_ptr, const int64_t *cell_faces,
                            const int32_t *zone_type, const double *zone_scalar, const double *zone_vector) {
    OrMesh *m = (OrMesh *)calloc(1, sizeof(OrMesh));
    m->dimensions = dimensions; m->n_cells = n_cells; m->n_faces = n_faces; m->n_zones = n_zones;
    size_t F = (size_t)n_faces, C = (size_t)n_cells;
    m->face_zone = (int32_t *)malloc(sizeof(int32_t) * F); memcpy(m->face_zone, face_zone, sizeof(int32_t) * F);
    m->face_c0 = (int64_t *)malloc(sizeof(int64_t) * F); memcpy(m->face_c0, face_c0, sizeof(int64_t) * F);
    m->face_c1 = (int64_t *)malloc(sizeof(int64_t) * F); memcpy(m->face_c1, face_c1, sizeof(int64_t) * F);
    m->face_area = (double *)malloc(sizeof(double) * F); memcpy(m->face_area, face_area, sizeof(double) * F);
    m->face_normal = (Vec3 *)malloc(sizeof(Vec3) * F); memcpy(m->face_normal, face_normal, sizeof(Vec3) * F);
    m->face_centroid = (Vec3 *)malloc(sizeof(Vec3) * F); memcpy(m->face_centroid, face_centroid, sizeof(Vec3) * F);
    m->cell_centroid = (Vec3 *)malloc(sizeof(Vec3) * C); memcpy(m->cell_centroid, cell_centroid, sizeof(Vec3) * C);
    m->cell_volume = (double *)malloc(sizeof(double) * C); memcpy(m->cell_volume, cell_volume, sizeof(double) * C);
    m->cell_face_ptr = (int64_t *)malloc(sizeof(int64_t) * (C + 1)); memcpy(m->cell_face_ptr, cell_face_ptr, sizeof(int64_t) * (C + 1));
    size_t ncf = (size_t)cell_face_ptr[n_cells];
    m->cell_faces = (int64_t *)malloc(sizeof(int64_t) * (ncf ? ncf : 1)); memcpy(m->cell_faces, cell_faces, sizeof(int64_t) * ncf);
    m->zones = (OrZone *)calloc((size_t)(n_zones > 0 ? n_zones : 1), sizeof(OrZone));
    for (int k = 0; k < n_zones; k++) {
        m->zones[k].id = (uint64_t)k;
        m->zones[k].zone_type = zone_type[k];
        m->zones[k].scalar_value = zone_scalar[k];
        m->zones[k].vector_value = v3(zone_vector[3 * k], zone_vector[3 * k + 1], zone_vector[3 * k + 2]);
        snprintf(m->zones[k].name, sizeof(m->zones[k].name), "zone%d", k);
    }
    m->face_node_ptr = (int64_t *)calloc(F + 1, sizeof(int64_t));
    return m;
}

int or_mesh_zone_index(const OrMesh *m, const char *name) {
    for (int k = 0; k < m->n_zones; k++) if (strcmp(m->zones[k].name, name) == 0) return k;
    return -1;
}

/* mesh.get_face_zone(name).zone_type = ...; .scalar_value = ...; .vector_value = ... (tests.rs:60-76) */
int or_mesh_set_zone(OrMesh *m, const char *name, int32_t zone_type, double scalar, double vx, double vy, double vz) {
    int k = or_mesh_zone_index(m, name);
    if (k < 0) return ORC_ERR_BAD_ARGUMENT; /* mesh.rs:194 panics */
    m->zones[k].zone_type = zone_type;
    m->zones[k].scalar_value = scalar;
    m->zones[k].vector_value = v3(vx, vy, vz);
    return ORC_OK;
}
