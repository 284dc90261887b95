/* discretization.c — restatement of src/discretization.rs (all four pub fns).
 * Test infrastructure (see oracle.h).  Cell loop and face loop run in the reference's order
 * (cells ascending, cell.face_indices ascending face id), expression order follows the Rust
 * operator chains, including the `f64 * Vector` bug and the in-place diagonal reads
 * of the Rhie-Chow flux (`frozen_diagonals` selects the all-old variant).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "vec3.h"

static inline Vec3 outward_normal(const OrMesh *m, int64_t face, int64_t cell) {
    return cell == m->face_c0[face] ? m->face_normal[face] : v_neg(m->face_normal[face]);
}

/* discretization.rs:39-131 */
int or_build_momentum_diffusion_matrix(const OrMesh *m, int diffusion_scheme, double mu, OrCsr **a_out,
                                       double *b_u, double *b_v, double *b_w) {
    *a_out = NULL;
    if (diffusion_scheme != ORC_DIFFUSION_CD) return ORC_ERR_UNSUPPORTED_SCHEME; /* :49-51 */
    int64_t n = m->n_cells;
    int64_t cap = m->cell_face_ptr[n] + n + 1, cnt = 0;
    int64_t *ri = (int64_t *)malloc(8 * (size_t)cap), *ci = (int64_t *)malloc(8 * (size_t)cap);
    double *vv = (double *)malloc(8 * (size_t)cap);
    for (int64_t i = 0; i < n; i++) { b_u[i] = 0.; b_v[i] = 0.; b_w[i] = 0.; }
    int st = ORC_OK;
    for (int64_t c = 0; c < n && st == ORC_OK; c++) {
        double a_p = 0.;
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            const OrZone *z = &m->zones[m->face_zone[f]];
            double d_i;
            int64_t nb = -1;
            switch (z->zone_type) {
            case ORC_BC_WALL: case ORC_BC_VELOCITY_INLET: { /* :70-79 */
                d_i = mu * m->face_area[f] / v_norm(v_sub(m->face_centroid[f], m->cell_centroid[c]));
                Vec3 sc = v_muls(z->vector_value, d_i);
                b_u[c] += sc.x; b_v[c] += sc.y; b_w[c] += sc.z;
                break;
            }
            case ORC_BC_PRESSURE_INLET: case ORC_BC_PRESSURE_OUTLET: case ORC_BC_SYMMETRY: /* :80-88 */
                d_i = 0.;
                break;
            case ORC_BC_INTERIOR: { /* :89-113 */
                nb = m->face_c0[f];
                if (nb == c) nb = m->face_c1[f];
                Vec3 e_xi = v_sub(m->cell_centroid[nb], m->cell_centroid[c]);
                d_i = mu * m->face_area[f] / v_norm(e_xi);
                break;
            }
            default: st = ORC_ERR_UNSUPPORTED_BC; d_i = 0.; break; /* :114-117 */
            }
            if (st) break;
            a_p += d_i;
            if (nb >= 0) { ri[cnt] = c; ci[cnt] = nb; vv[cnt] = -d_i; cnt++; } /* :123-126 */
        }
        ri[cnt] = c; ci[cnt] = c; vv[cnt] = a_p; cnt++; /* :128 */
    }
    if (st == ORC_OK) *a_out = or_csr_from_coo(n, n, cnt, ri, ci, vv);
    free(ri); free(ci); free(vv);
    return st;
}

/* discretization.rs:450-472 */
OrCsr *or_initialize_momentum_matrix(const OrMesh *m) {
    int64_t n = m->n_cells;
    int64_t cap = m->cell_face_ptr[n] + n + 1, cnt = 0;
    int64_t *ri = (int64_t *)malloc(8 * (size_t)cap), *ci = (int64_t *)malloc(8 * (size_t)cap);
    double *vv = (double *)malloc(8 * (size_t)cap);
    for (int64_t c = 0; c < n; c++) {
        ri[cnt] = c; ci[cnt] = c; vv[cnt] = 1.; cnt++;
        double nf = (double)(m->cell_face_ptr[c + 1] - m->cell_face_ptr[c]);
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            if (m->face_c1[f] >= 0) {
                int64_t nb = m->face_c0[f] == c ? m->face_c1[f] : m->face_c0[f];
                ri[cnt] = c; ci[cnt] = nb; vv[cnt] = -1. / nf; cnt++;
            }
        }
    }
    OrCsr *a = or_csr_from_coo(n, n, cnt, ri, ci, vv);
    free(ri); free(ci); free(vv);
    return a;
}

/* TVD limiter psi(r) (lib.rs:107-118).  f64::min/max ignore NaN, like fmin/fmax. */
static double psi_eval(int momentum, double r) {
    switch (momentum) {
    case ORC_MOMENTUM_TVD_UD: return 0.;
    case ORC_MOMENTUM_TVD_CD1: return 1.;
    case ORC_MOMENTUM_TVD_LUD: return r;
    case ORC_MOMENTUM_TVD_QUICK: return (3. + r) / 4.;
    case ORC_MOMENTUM_TVD_UMIST: {
        double acc = INFINITY;
        acc = fmin(acc, 2. * r);
        acc = fmin(acc, (1. + 3. * r) / 4.);
        acc = fmin(acc, (3. + r) / 4.);
        acc = fmin(acc, 2.);
        return fmax(0., acc);
    }
    default: return NAN;
    }
}

static int is_tvd(int momentum) {
    return momentum == ORC_MOMENTUM_TVD_LUD || momentum == ORC_MOMENTUM_TVD_QUICK || momentum == ORC_MOMENTUM_TVD_UMIST ||
           momentum == ORC_MOMENTUM_TVD_UD || momentum == ORC_MOMENTUM_TVD_CD1;
}

/* discretization.rs:134-356 */
int or_build_momentum_advection_matrices(OrCsr *a_u, OrCsr *a_v, OrCsr *a_w, double *b_u, double *b_v, double *b_w,
                                         const OrCsr *a_di, const OrMesh *m, const double *u, const double *v,
                                         const double *w, const double *p, const OrcSettings *s, double rho,
                                         double peclet_out[3]) {
    int64_t n = m->n_cells;
    int q1 = s->q1_compat;
    if (!(s->momentum == ORC_MOMENTUM_UD || s->momentum == ORC_MOMENTUM_CD1 || is_tvd(s->momentum)))
        return ORC_ERR_UNSUPPORTED_SCHEME; /* :287 */
    size_t nn = (size_t)(n > 0 ? n : 1);
    /* Rhie-Chow reads a_{u,v,w}.get(i,i) / get(j,j) while the loop overwrites them (Q2).
     * du/dv/dw mirror what get() would return: live (reference) or frozen (all-old). */
    double *du = (double *)malloc(8 * nn), *dv = (double *)malloc(8 * nn), *dw = (double *)malloc(8 * nn);
    for (int64_t i = 0; i < n; i++) {
        du[i] = a_u->val[or_csr_find(a_u, i, i)];
        dv[i] = a_v->val[or_csr_find(a_v, i, i)];
        dw[i] = a_w->val[or_csr_find(a_w, i, i)];
    }
    double min_pe = INFINITY, max_pe = -INFINITY, avg_pe = 0.;
    int st = ORC_OK;
    for (int64_t c = 0; c < n && st == ORC_OK; c++) {
        Vec3 s_u = v_zero(); /* get_momentum_source_term (solver.rs:698-701) */
        Vec3 s_u_dc = v_zero(), s_d_cross = v_zero();
        int64_t dpos = or_csr_find(a_di, c, c);
        if (dpos < 0) { st = ORC_ERR_STRUCTURAL_ZERO; break; }
        double a_ii_di = a_di->val[dpos]; /* :176 */
        Vec3 a_p = v_zero();
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            double face_flux;
            st = or_get_face_flux(m, u, v, w, p, f, c, s->velocity_interpolation, s->gradient_reconstruction, q1, du, dv, dw, &face_flux); /* :184-197 */
            if (st) break;
            Vec3 n_out = outward_normal(m, f, c);
            double f_i = face_flux * m->face_area[f] * rho; /* :202 */
            double face_pressure;
            st = or_get_face_pressure(m, p, f, s->pressure_interpolation, s->gradient_reconstruction, q1, &face_pressure); /* :203-209 */
            if (st) break;
            int64_t nb = (m->face_c1[f] < 0) ? -1 : (m->face_c0[f] == c ? m->face_c1[f] : m->face_c0[f]); /* :210-220 */
            Vec3 a_nb;
            if (s->momentum == ORC_MOMENTUM_UD) {
                a_nb = s_mulv(fmin(f_i, 0.), v_ones(), q1); /* :226 */
            } else if (s->momentum == ORC_MOMENTUM_CD1) {
                a_nb = v_divs(s_mulv(f_i, v_ones(), q1), 2.); /* :231 */
            } else { /* TVD :233-286 */
                if (nb < 0) {
                    a_nb = s_mulv(fmin(f_i, 0.), v_ones(), q1); /* :238 */
                } else {
                    int64_t down = f_i > 0. ? nb : c; /* :244-248 */
                    Vec3 dvel = v3(u[down], v[down], w[down]);
                    Vec3 vel = v3(u[c], v[c], w[c]);
                    if (v_norm(v_sub(dvel, vel)) == 0.) {
                        a_nb = v_divs(s_mulv(f_i, v_ones(), q1), 2.); /* :264 */
                    } else {
                        Tensor3 g;
                        st = or_calculate_velocity_gradient(m, u, v, w, c, s->gradient_reconstruction, &g); /* :266-273 */
                        if (st) break;
                        Vec3 r_pa = v_sub(m->cell_centroid[nb], m->cell_centroid[c]);
                        /* :276-278  2. * grad.inner(r_pa) / (down - vel) - 1.  ==  ((f64*Vector) / Vector) - f64 */
                        Vec3 r = v_subs(v_div(s_mulv(2., t_inner(g, r_pa), q1), v_sub(dvel, vel)), 1.);
                        Vec3 ps = v3(psi_eval(s->momentum, r.x), psi_eval(s->momentum, r.y), psi_eval(s->momentum, r.z));
                        a_nb = v_divs(s_mulv(f_i, ps, q1), 2.); /* :279-283 */
                    }
                }
            }
            a_p = v_add(a_p, v_adds(v_neg(a_nb), f_i));                                   /* :290 */
            s_u = v_add(s_u, v_muls(v_muls(v_neg(n_out), face_pressure), m->face_area[f])); /* :291 */
            if (nb < 0) { /* :294-307 */
                const OrZone *z = &m->zones[m->face_zone[f]];
                if (z->zone_type == ORC_BC_WALL || z->zone_type == ORC_BC_VELOCITY_INLET)
                    s_u = v_add(s_u, v3((a_nb.x - f_i) * z->vector_value.x, (a_nb.y - f_i) * z->vector_value.y, (a_nb.z - f_i) * z->vector_value.z));
                else
                    s_u = v_add(s_u, v_zero());
            } else { /* :308-324 */
                int64_t t = or_csr_find(a_di, c, nb);
                int64_t pu = or_csr_find(a_u, c, nb), pv = or_csr_find(a_v, c, nb), pw = or_csr_find(a_w, c, nb);
                if (t < 0 || pu < 0 || pv < 0 || pw < 0) { st = ORC_ERR_STRUCTURAL_ZERO; break; }
                double a_ij_di = a_di->val[t];
                a_u->val[pu] = a_nb.x + a_ij_di;
                a_v->val[pv] = a_nb.y + a_ij_di;
                a_w->val[pw] = a_nb.z + a_ij_di;
            }
        }
        if (st) break;
        Vec3 total = v_add(v_add(s_u, s_u_dc), s_d_cross); /* :326 */
        b_u[c] = total.x; b_v[c] = total.y; b_w[c] = total.z;
        double pe[3] = {a_p.x / a_ii_di, a_p.y / a_ii_di, a_p.z / a_ii_di}; /* :331-333 */
        /* :336-337 max_by/min_by with total_cmp over [running, x, y, z]: max_by keeps the last of
         * equals, min_by the first */
        for (int k = 0; k < 3; k++) {
            if (f64_total_cmp(pe[k], max_pe) >= 0) max_pe = pe[k];
            if (f64_total_cmp(pe[k], min_pe) < 0) min_pe = pe[k];
        }
        avg_pe += (((0. + pe[0]) + pe[1]) + pe[2]) / 3.; /* :338 */
        int64_t pu = or_csr_find(a_u, c, c), pv = or_csr_find(a_v, c, c), pw = or_csr_find(a_w, c, c);
        a_u->val[pu] = a_p.x + a_ii_di; /* :340-351 */
        a_v->val[pv] = a_p.y + a_ii_di;
        a_w->val[pw] = a_p.z + a_ii_di;
        if (!s->frozen_diagonals) { du[c] = a_u->val[pu]; dv[c] = a_v->val[pv]; dw[c] = a_w->val[pw]; }
    }
    if (peclet_out) { peclet_out[0] = avg_pe / (double)n; peclet_out[1] = min_pe; peclet_out[2] = max_pe; } /* :355 */
    free(du); free(dv); free(dw);
    return st;
}

/* discretization.rs:359-448 */
int or_build_pressure_correction_matrices(const OrMesh *m, const double *u, const double *v, const double *w,
                                          const double *p, const OrCsr *a_u, const OrCsr *a_v, const OrCsr *a_w,
                                          const OrcSettings *s, double rho, OrCsr **a_out, double *b_out) {
    *a_out = NULL;
    int64_t n = m->n_cells;
    size_t nn = (size_t)(n > 0 ? n : 1);
    double *du = (double *)malloc(8 * nn), *dv = (double *)malloc(8 * nn), *dw = (double *)malloc(8 * nn);
    for (int64_t i = 0; i < n; i++) {
        du[i] = a_u->val[or_csr_find(a_u, i, i)];
        dv[i] = a_v->val[or_csr_find(a_v, i, i)];
        dw[i] = a_w->val[or_csr_find(a_w, i, i)];
    }
    int64_t cap = m->cell_face_ptr[n] + n + 1, cnt = 0;
    int64_t *ri = (int64_t *)malloc(8 * (size_t)cap), *ci = (int64_t *)malloc(8 * (size_t)cap);
    double *vv = (double *)malloc(8 * (size_t)cap);
    int st = ORC_OK;
    for (int64_t c = 0; c < n && st == ORC_OK; c++) {
        double a_p = 0., b_p = 0.;
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            double flux;
            st = or_get_face_flux(m, u, v, w, p, f, c, s->velocity_interpolation, s->gradient_reconstruction, s->q1_compat, du, dv, dw, &flux); /* :383-396 */
            if (st) break;
            Vec3 n_in = v_muls(outward_normal(m, f, c), -1.); /* mesh.rs:224-226 */
            b_p += rho * (-flux) * m->face_area[f]; /* :399 */
            if (m->face_c1[f] >= 0) { /* :401-424 */
                int64_t nb = m->face_c0[f] != c ? m->face_c0[f] : m->face_c1[f];
                /* get_face_normal_momentum_coefficient! (:25-34): 0.5 * |((a_u[ii]+a_u[jj]) n_x, ...)| */
                double a_int = 0.5 * v_norm(v3((du[c] + du[nb]) * n_in.x, (dv[c] + dv[nb]) * n_in.y, (dw[c] + dw[nb]) * n_in.z));
                double a_nb = rho * (m->face_area[f] * m->face_area[f]) / a_int; /* :422 */
                ri[cnt] = c; ci[cnt] = nb; vv[cnt] = -a_nb; cnt++;
                a_p += a_nb;
            } else { /* :425-436 */
                double a_ii_norm = v_norm(v3(du[c] * n_in.x, dv[c] * n_in.y, dw[c] * n_in.z));
                double a_nb = rho * (m->face_area[f] * m->face_area[f]) / a_ii_norm;
                a_p += a_nb / 2.;
            }
        }
        ri[cnt] = c; ci[cnt] = c; vv[cnt] = a_p; cnt++; /* :438 */
        b_out[c] = b_p;
    }
    if (st == ORC_OK) *a_out = or_csr_from_coo(n, n, cnt, ri, ci, vv);
    free(ri); free(ci); free(vv); free(du); free(dv); free(dw);
    return st;
}
