/* solver.c — restatement of src/solver.rs: solve_steady (:26-244), initialize_flow (:246-352),
 * initialize_pressure_field (:414-509), Green-Gauss gradients (:774-802, :874-902), face
 * velocity / flux / pressure (:952-1150), apply_pressure_correction (:1170-1227).
 * Test infrastructure (see oracle.h).  Least-squares arms are out of scope.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "vec3.h"

void or_settings_default(OrcSettings *s) { /* lib.rs:58-86 */
    memset(s, 0, sizeof(*s));
    s->momentum = ORC_MOMENTUM_CD1;
    s->diffusion = ORC_DIFFUSION_CD;
    s->pressure_interpolation = ORC_PINTERP_SECOND_ORDER;
    s->velocity_interpolation = ORC_VINTERP_RHIE_CHOW;
    s->gradient_reconstruction = ORC_GRAD_GREEN_GAUSS_CELL;
    s->pressure_relaxation = 0.01;
    s->momentum_relaxation = 0.5;
    s->solver_type = ORC_SOLVER_MULTIGRID;
    s->iterations = 50;
    s->relaxation = 0.5;
    s->relative_convergence_threshold = 1e-3;
    s->preconditioner = ORC_PRECOND_JACOBI;
    s->q1_compat = 1;
    s->frozen_diagonals = 0;
    s->breakdown_guard = 0; /* the oracle IS the reference: no guard */
}

const char *or_status_string(int st) {
    switch (st) {
    case ORC_OK: return "ok";
    case ORC_ERR_SOLUTION_DIVERGED: return "solution diverged";
    case ORC_ERR_MULTIGRID_DIVERGED: return "Multigrid diverged";
    case ORC_ERR_JACOBI_NAN: return "diverged";
    case ORC_ERR_JACOBI_TOO_LARGE: return "Diverged - max solution value > 10^10";
    case ORC_ERR_GS_MAINTENANCE: return "Gauss-Seidel out for maintenance :)";
    case ORC_ERR_STRUCTURAL_ZERO: return "Tried to access CsrMatrix element that hasn't been stored yet.";
    case ORC_ERR_UNSUPPORTED_BC: return "BC not supported";
    case ORC_ERR_UNSUPPORTED_SCHEME: return "unsupported scheme";
    case ORC_ERR_UNSUPPORTED_SOLVER: return "unsupported solution method";
    case ORC_ERR_BAD_ARGUMENT: return "bad argument";
    default: return "error";
    }
}

static inline Vec3 outward_normal(const OrMesh *m, int64_t face, int64_t cell) { /* mesh.rs:216-222 */
    return cell == m->face_c0[face] ? m->face_normal[face] : v_neg(m->face_normal[face]);
}
static inline const OrZone *zone_of(const OrMesh *m, int64_t face) { return &m->zones[m->face_zone[face]]; }

/* solver.rs:952-1003 */
int or_get_face_velocity(const OrMesh *m, const double *u, const double *v, const double *w, int64_t face, int scheme, Vec3 *out) {
    const OrZone *z = zone_of(m, face);
    int64_t c0 = m->face_c0[face];
    switch (z->zone_type) {
    case ORC_BC_WALL: case ORC_BC_VELOCITY_INLET: *out = z->vector_value; return ORC_OK;
    case ORC_BC_PRESSURE_INLET: case ORC_BC_PRESSURE_OUTLET: case ORC_BC_SYMMETRY: *out = v3(u[c0], v[c0], w[c0]); return ORC_OK;
    case ORC_BC_INTERIOR: {
        int64_t c1 = m->face_c1[face];
        Vec3 vel0 = v3(u[c0], v[c0], w[c0]), vel1 = v3(u[c1], v[c1], w[c1]);
        if (scheme == ORC_VINTERP_LINEAR) { *out = v_divs(v_add(vel0, vel1), 2.); return ORC_OK; } /* :987 */
        if (scheme == ORC_VINTERP_LINEAR_WEIGHTED) { /* :988-992 */
            double dx0 = v_norm(v_sub(m->cell_centroid[c0], m->face_centroid[face]));
            double dx1 = v_norm(v_sub(m->cell_centroid[c1], m->face_centroid[face]));
            *out = v_add(vel0, v_divs(v_muls(v_sub(vel1, vel0), dx0), dx0 + dx1));
            return ORC_OK;
        }
        return ORC_ERR_UNSUPPORTED_SCHEME; /* :993-998 */
    }
    default: return ORC_ERR_UNSUPPORTED_BC; /* :1001 */
    }
}

/* solver.rs:874-902 (Green-Gauss cell-based arm) */
/* nalgebra 0.32.4 dense arithmetic of the least-squares arms (crate not vendored, Cargo.lock:326-327; restated from its
 * published source):
 *   - `&a.transpose() * &b`, `&a.transpose() * &a`, `a_inv * b` (base/blas.rs gemm): an operand dimension <= 5 keeps
 *     matrixmultiply out and the product is one gemv per output column — y = (1 * col_0) * x_0, then
 *     y += (1 * col_j) * x_j — i.e. every output entry is a left-to-right sum over the inner index, no fma.
 *   - DMatrix::try_inverse (linalg/inverse.rs): closed forms for dimensions 1, 2 and 3 (minors / determinant), a zero
 *     determinant yields None.
 * Returns 1 when the matrix was inverted. */
static int nalgebra_try_inverse(int dim, double a[3][3]) {
    if (dim == 0) return 1;
    if (dim == 1) {
        if (a[0][0] == 0.) return 0;
        a[0][0] = 1. / a[0][0];
        return 1;
    }
    if (dim == 2) {
        double m11 = a[0][0], m12 = a[0][1], m21 = a[1][0], m22 = a[1][1];
        double determinant = m11 * m22 - m21 * m12;
        if (determinant == 0.) return 0;
        a[0][0] = m22 / determinant; a[0][1] = -m12 / determinant;
        a[1][0] = -m21 / determinant; a[1][1] = m11 / determinant;
        return 1;
    }
    double m11 = a[0][0], m12 = a[0][1], m13 = a[0][2], m21 = a[1][0], m22 = a[1][1], m23 = a[1][2], m31 = a[2][0], m32 = a[2][1], m33 = a[2][2];
    double minor_m12_m23 = m22 * m33 - m32 * m23;
    double minor_m11_m23 = m21 * m33 - m31 * m23;
    double minor_m11_m22 = m21 * m32 - m31 * m22;
    double determinant = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (determinant == 0.) return 0;
    a[0][0] = minor_m12_m23 / determinant;
    a[0][1] = (m13 * m32 - m33 * m12) / determinant;
    a[0][2] = (m12 * m23 - m22 * m13) / determinant;
    a[1][0] = -minor_m11_m23 / determinant;
    a[1][1] = (m11 * m33 - m31 * m13) / determinant;
    a[1][2] = (m13 * m21 - m23 * m11) / determinant;
    a[2][0] = minor_m11_m22 / determinant;
    a[2][1] = (m12 * m31 - m32 * m11) / determinant;
    a[2][2] = (m11 * m22 - m21 * m12) / determinant;
    return 1;
}
/* A^T A and A^T b_q for the n x dim matrix whose rows are x[f][cols[.]], q < nb right-hand sides */
static void normal_equations(int64_t n, const double (*x)[3], int dim, const int *cols, int nb, const double (*b)[3], double ata[3][3], double atb[3][3]) {
    for (int i = 0; i < dim; i++) {
        for (int j = 0; j < dim; j++) {
            double y = 0.;
            for (int64_t f = 0; f < n; f++) {
                double t = (1. * x[f][cols[i]]) * x[f][cols[j]];
                y = (f == 0) ? t : t + 1. * y;
            }
            ata[i][j] = y;
        }
        for (int q = 0; q < nb; q++) {
            double y = 0.;
            for (int64_t f = 0; f < n; f++) {
                double t = (1. * x[f][cols[i]]) * b[f][q];
                y = (f == 0) ? t : t + 1. * y;
            }
            atb[q][i] = y;
        }
    }
}
/* a_inv * b (gemv over the columns of a_inv) */
static void inv_times(int dim, double ainv[3][3], const double b[3], double out[3]) {
    for (int i = 0; i < dim; i++) {
        double y = 0.;
        for (int j = 0; j < dim; j++) {
            double t = (1. * ainv[i][j]) * b[j];
            y = (j == 0) ? t : t + 1. * y;
        }
        out[i] = y;
    }
}

/* solver.rs:874-950 */
int or_calculate_pressure_gradient(const OrMesh *m, const double *p, int64_t cell, int scheme, int q1, Vec3 *out) {
    if (scheme == ORC_GRAD_LEAST_SQUARES) { /* :903-947 */
        int64_t n = m->cell_face_ptr[cell + 1] - m->cell_face_ptr[cell];
        double (*x)[3] = malloc(sizeof(double[3]) * (size_t)(n > 0 ? n : 1));
        double (*b)[3] = malloc(sizeof(double[3]) * (size_t)(n > 0 ? n : 1));
        Vec3 cc = m->cell_centroid[cell];
        int st = ORC_OK;
        for (int64_t k = 0; k < n && st == ORC_OK; k++) {
            int64_t f = m->cell_faces[m->cell_face_ptr[cell] + k];
            const OrZone *z = zone_of(m, f);
            Vec3 d;
            double val;
            if (z->zone_type == ORC_BC_INTERIOR) {
                int64_t nb = m->face_c0[f] == cell ? m->face_c1[f] : m->face_c0[f];
                d = v_sub(m->cell_centroid[nb], cc);
                val = p[nb] - p[cell];
            } else { /* :925-934 the face VALUE, not a difference */
                d = v_sub(m->face_centroid[f], cc);
                st = or_get_face_pressure(m, p, f, ORC_PINTERP_NONE, ORC_GRAD_NONE, q1, &val);
            }
            x[k][0] = d.x; x[k][1] = d.y; x[k][2] = d.z;
            b[k][0] = val;
        }
        if (st == ORC_OK) {
            const int cols[3] = {0, 1, 2};
            double ata[3][3], atb[3][3], g[3];
            normal_equations(n, x, 3, cols, 1, b, ata, atb);
            if (!nalgebra_try_inverse(3, ata)) st = ORC_ERR_SINGULAR_MATRIX; /* :943 unwrap() */
            else { inv_times(3, ata, atb[0], g); *out = v3(g[0], g[1], g[2]); }
        }
        free(x); free(b);
        return st;
    }
    if (scheme != ORC_GRAD_GREEN_GAUSS_CELL) return ORC_ERR_UNSUPPORTED_SCHEME;
    Vec3 acc = v_zero();
    for (int64_t q = m->cell_face_ptr[cell]; q < m->cell_face_ptr[cell + 1]; q++) {
        int64_t f = m->cell_faces[q];
        double face_value;
        int st = or_get_face_pressure(m, p, f, ORC_PINTERP_LINEAR, scheme, q1, &face_value);
        if (st) return st;
        /* :896-898  face_value * (area/volume) * outward_normal  ==  (f64 * f64) * Vector  -> Q1 */
        acc = v_add(acc, s_mulv(face_value * (m->face_area[f] / m->cell_volume[cell]), outward_normal(m, f, cell), q1));
    }
    *out = acc;
    return ORC_OK;
}

/* solver.rs:774-802 (Green-Gauss arm) */
int or_calculate_velocity_gradient(const OrMesh *m, const double *u, const double *v, const double *w, int64_t cell, int scheme, Tensor3 *out) {
    if (scheme == ORC_GRAD_LEAST_SQUARES) { /* :803-869 */
        int64_t n = m->cell_face_ptr[cell + 1] - m->cell_face_ptr[cell];
        double (*x)[3] = malloc(sizeof(double[3]) * (size_t)(n > 0 ? n : 1));
        double (*b)[3] = malloc(sizeof(double[3]) * (size_t)(n > 0 ? n : 1));
        Vec3 cc = m->cell_centroid[cell];
        int st = ORC_OK;
        for (int64_t k = 0; k < n && st == ORC_OK; k++) {
            int64_t f = m->cell_faces[m->cell_face_ptr[cell] + k];
            const OrZone *z = zone_of(m, f);
            Vec3 d;
            if (z->zone_type == ORC_BC_INTERIOR) {
                int64_t nb = m->face_c0[f] == cell ? m->face_c1[f] : m->face_c0[f];
                d = v_sub(m->cell_centroid[nb], cc);
                b[k][0] = u[nb] - u[cell]; b[k][1] = v[nb] - v[cell]; b[k][2] = w[nb] - w[cell];
            } else { /* :830-838 the face VELOCITY, not a difference */
                Vec3 fv;
                st = or_get_face_velocity(m, u, v, w, f, ORC_VINTERP_NONE, &fv);
                d = v_sub(m->face_centroid[f], cc);
                b[k][0] = fv.x; b[k][1] = fv.y; b[k][2] = fv.z;
            }
            x[k][0] = d.x; x[k][1] = d.y; x[k][2] = d.z;
        }
        if (st == ORC_OK) {
            const int cols[3] = {0, 1, 2};
            double ata[3][3], atb[3][3], g[3][3];
            normal_equations(n, x, 3, cols, 3, b, ata, atb);
            if (!nalgebra_try_inverse(3, ata)) st = ORC_ERR_SINGULAR_MATRIX; /* :850 unwrap() */
            else {
                for (int q = 0; q < 3; q++) inv_times(3, ata, atb[q], g[q]);
                out->x = v3(g[0][0], g[0][1], g[0][2]); out->y = v3(g[1][0], g[1][1], g[1][2]); out->z = v3(g[2][0], g[2][1], g[2][2]);
            }
        }
        free(x); free(b);
        return st;
    }
    if (scheme != ORC_GRAD_GREEN_GAUSS_CELL && scheme != ORC_GRAD_GREEN_GAUSS_NODE) return ORC_ERR_UNSUPPORTED_SCHEME; /* GreenGauss(_) */
    Tensor3 acc = t_zero();
    for (int64_t q = m->cell_face_ptr[cell]; q < m->cell_face_ptr[cell + 1]; q++) {
        int64_t f = m->cell_faces[q];
        Vec3 fv;
        int st = or_get_face_velocity(m, u, v, w, f, ORC_VINTERP_LINEAR, &fv);
        if (st) return st;
        acc = t_add(acc, v_outer(fv, v_muls(outward_normal(m, f, cell), m->face_area[f] / m->cell_volume[cell])));
    }
    *out = acc;
    return ORC_OK;
}

/* solver.rs:1104-1150 */
int or_get_face_pressure(const OrMesh *m, const double *p, int64_t face, int interp, int grad_scheme, int q1, double *out) {
    const OrZone *z = zone_of(m, face);
    switch (z->zone_type) {
    case ORC_BC_SYMMETRY: case ORC_BC_WALL: case ORC_BC_VELOCITY_INLET: *out = p[m->face_c0[face]]; return ORC_OK;
    case ORC_BC_PRESSURE_INLET: case ORC_BC_PRESSURE_OUTLET: *out = z->scalar_value; return ORC_OK;
    case ORC_BC_INTERIOR: {
        int64_t c0 = m->face_c0[face], c1 = m->face_c1[face];
        switch (interp) {
        case ORC_PINTERP_LINEAR: *out = (p[c0] + p[c1]) * 0.5; return ORC_OK; /* :1128 */
        case ORC_PINTERP_LINEAR_WEIGHTED: { /* :1129-1133 */
            double x0 = v_norm(v_sub(m->cell_centroid[c0], m->face_centroid[face]));
            double x1 = v_norm(v_sub(m->cell_centroid[c1], m->face_centroid[face]));
            *out = p[c0] + (p[c1] - p[c0]) * x0 / (x0 + x1);
            return ORC_OK;
        }
        case ORC_PINTERP_SECOND_ORDER: { /* :1138-1144 */
            Vec3 g0, g1;
            int st = or_calculate_pressure_gradient(m, p, c0, grad_scheme, q1, &g0);
            if (st) return st;
            st = or_calculate_pressure_gradient(m, p, c1, grad_scheme, q1, &g1);
            if (st) return st;
            Vec3 r0 = v_sub(m->face_centroid[face], m->cell_centroid[c0]);
            Vec3 r1 = v_sub(m->face_centroid[face], m->cell_centroid[c1]);
            *out = 0.5 * ((p[c0] + p[c1]) + (v_dot(g0, r0) + v_dot(g1, r1)));
            return ORC_OK;
        }
        default: return ORC_ERR_UNSUPPORTED_SCHEME; /* :1134-1137, :1145 */
        }
    }
    default: return ORC_ERR_UNSUPPORTED_BC; /* :1148 */
    }
}

/* get_normal_momentum_coefficient! (discretization.rs:14-23) */
static inline double normal_momentum_coefficient(int64_t i, const double *du, const double *dv, const double *dw, Vec3 n) {
    return v_norm(v3(du[i] * n.x, dv[i] * n.y, dw[i] * n.z));
}

/* solver.rs:1007-1102 */
int or_get_face_flux(const OrMesh *m, const double *u, const double *v, const double *w, const double *p, int64_t face,
                     int64_t cell, int vinterp, int grad_scheme, int q1, const double *du, const double *dv,
                     const double *dw, double *out) {
    Vec3 n = outward_normal(m, face, cell);
    const OrZone *z = zone_of(m, face);
    Vec3 fv;
    int st;
    switch (z->zone_type) {
    case ORC_BC_WALL: case ORC_BC_SYMMETRY: *out = 0.; return ORC_OK;
    case ORC_BC_VELOCITY_INLET: case ORC_BC_PRESSURE_INLET: case ORC_BC_PRESSURE_OUTLET:
        st = or_get_face_velocity(m, u, v, w, face, ORC_VINTERP_NONE, &fv);
        if (st) return st;
        *out = v_dot(n, fv);
        return ORC_OK;
    case ORC_BC_INTERIOR:
        if (vinterp == ORC_VINTERP_LINEAR || vinterp == ORC_VINTERP_LINEAR_WEIGHTED) {
            st = or_get_face_velocity(m, u, v, w, face, vinterp, &fv);
            if (st) return st;
            *out = v_dot(n, fv);
            return ORC_OK;
        }
        if (vinterp == ORC_VINTERP_RHIE_CHOW) { /* :1051-1095 */
            int64_t nb = m->face_c0[face];
            if (nb == cell) nb = m->face_c1[face];
            Vec3 vel_i = v3(u[cell], v[cell], w[cell]), vel_j = v3(u[nb], v[nb], w[nb]);
            Vec3 ccv = v_sub(m->cell_centroid[nb], m->cell_centroid[cell]);
            double a_i = normal_momentum_coefficient(cell, du, dv, dw, n);
            double a_j = normal_momentum_coefficient(nb, du, dv, dw, n);
            Vec3 g_i, g_j;
            st = or_calculate_pressure_gradient(m, p, cell, grad_scheme, q1, &g_i);
            if (st) return st;
            st = or_calculate_pressure_gradient(m, p, nb, grad_scheme, q1, &g_j);
            if (st) return st;
            double vol_i = m->cell_volume[cell], vol_j = m->cell_volume[nb];
            double term_1 = v_dot(v_add(vel_i, vel_j), n);
            double term_2 = (vol_i / a_i + vol_j / a_j) * (p[cell] - p[nb]) / v_norm(ccv);
            double term_3 = v_dot(v_add(s_mulv(vol_i / a_i, g_i, q1), s_mulv(vol_j / a_j, g_j, q1)), v_unit(ccv));
            *out = 0.5 * (term_1 + term_2 - term_3);
            return ORC_OK;
        }
        return ORC_ERR_UNSUPPORTED_SCHEME; /* :1096-1098 */
    default: return ORC_ERR_UNSUPPORTED_BC; /* :1100 */
    }
}

/* solver.rs:1170-1227 */
int or_apply_pressure_correction(const OrMesh *m, const double *du, const double *dv, const double *dw,
                                 const double *p_prime, double *u, double *v, double *w, double *p,
                                 const OrcSettings *s, double out_norms[2]) {
    double velocity_corr_sum = 0.;
    for (int64_t c = 0; c < m->n_cells; c++) {
        p[c] += s->pressure_relaxation * p_prime[c];
        Vec3 acc = v_zero();
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            const OrZone *z = zone_of(m, f);
            Vec3 n = outward_normal(m, f, c);
            double pp_nb;
            switch (z->zone_type) {
            case ORC_BC_WALL: case ORC_BC_SYMMETRY: case ORC_BC_VELOCITY_INLET: pp_nb = p_prime[c]; break;
            case ORC_BC_PRESSURE_INLET: case ORC_BC_PRESSURE_OUTLET: pp_nb = 0.; break;
            case ORC_BC_INTERIOR: pp_nb = p_prime[m->face_c0[f] == c ? m->face_c1[f] : m->face_c0[f]]; break;
            default: return ORC_ERR_UNSUPPORTED_BC; /* :1209-1212 */
            }
            Vec3 scaled = v3(n.x / du[c], n.y / dv[c], n.z / dw[c]);
            acc = v_add(acc, v_muls(v_muls(scaled, p_prime[c] - pp_nb), m->face_area[f]));
        }
        u[c] += acc.x * s->momentum_relaxation;
        v[c] += acc.y * s->momentum_relaxation;
        w[c] += acc.z * s->momentum_relaxation;
        double nn = v_norm(acc);
        velocity_corr_sum += nn * nn; /* .norm().powi(2) */
    }
    if (out_norms) { out_norms[0] = or_norm(p_prime, m->n_cells); out_norms[1] = sqrt(velocity_corr_sum); }
    return ORC_OK;
}

static void csr_diag(const OrCsr *a, double *d) {
    for (int64_t i = 0; i < a->nrows; i++) d[i] = a->val[or_csr_find(a, i, i)];
}

/* solver.rs:26-244.  The reference keeps a_u/a_v/a_w across iterations (initialised at :43-45)
 * so Rhie-Chow reads diag = 1.0 in iteration 1. */
int or_solve_steady(const OrMesh *m, double *u, double *v, double *w, double *p, const OrcSettings *s, double rho,
                    double mu, uint64_t iteration_count, double *report) {
    int64_t n = m->n_cells;
    OrCsr *a_di = NULL;
    size_t nn = (size_t)(n > 0 ? n : 1);
    double *b_u_di = (double *)calloc(nn, 8), *b_v_di = (double *)calloc(nn, 8), *b_w_di = (double *)calloc(nn, 8);
    double *b_u = (double *)calloc(nn, 8), *b_v = (double *)calloc(nn, 8), *b_w = (double *)calloc(nn, 8);
    double *p_prime = (double *)calloc(nn, 8), *b_p = (double *)calloc(nn, 8);
    double *du = (double *)malloc(8 * nn), *dv = (double *)malloc(8 * nn), *dw = (double *)malloc(8 * nn);
    int st = or_build_momentum_diffusion_matrix(m, s->diffusion, mu, &a_di, b_u_di, b_v_di, b_w_di); /* :41-42 */
    OrCsr *a_u = NULL, *a_v = NULL, *a_w = NULL;
    if (st == ORC_OK) {
        a_u = or_initialize_momentum_matrix(m); a_v = or_initialize_momentum_matrix(m); a_w = or_initialize_momentum_matrix(m); /* :43-45 */
    }
    for (uint64_t iter = 1; iter <= iteration_count && st == ORC_OK; iter++) { /* :60 */
        double peclet[3];
        st = or_build_momentum_advection_matrices(a_u, a_v, a_w, b_u, b_v, b_w, a_di, m, u, v, w, p, s, rho, peclet); /* :61-79 */
        if (st) break;
        for (int64_t i = 0; i < n; i++) { b_u[i] += b_u_di[i]; b_v[i] += b_v_di[i]; b_w[i] += b_w_di[i]; } /* :80-82 */
        st = or_iterative_solve(a_u, b_u, u, s->iterations, s->solver_type, s->relaxation, s->relative_convergence_threshold, s->preconditioner); /* :99-110 */
        if (st) break;
        st = or_iterative_solve(a_v, b_v, v, s->iterations, s->solver_type, s->relaxation, s->relative_convergence_threshold, s->preconditioner); /* :112-123 */
        if (st) break;
        st = or_iterative_solve(a_w, b_w, w, s->iterations, s->solver_type, s->relaxation, s->relative_convergence_threshold, s->preconditioner); /* :125-136 */
        if (st) break;
        OrCsr *a_p = NULL;
        st = or_build_pressure_correction_matrices(m, u, v, w, p, a_u, a_v, a_w, s, rho, &a_p, b_p); /* :137-148 */
        if (st) { or_csr_free(a_p); break; }
        for (int64_t i = 0; i < n; i++) p_prime[i] *= 0.; /* :167 (NaN stays NaN, as in the reference) */
        st = or_iterative_solve(a_p, b_p, p_prime, s->iterations, s->solver_type, s->relaxation, s->relative_convergence_threshold, s->preconditioner); /* :168-179 */
        or_csr_free(a_p);
        if (st) break;
        csr_diag(a_u, du); csr_diag(a_v, dv); csr_diag(a_w, dw);
        double norms[2];
        st = or_apply_pressure_correction(m, du, dv, dw, p_prime, u, v, w, p, s, norms); /* :193-204 */
        if (st) break;
        double u_avg = or_sum(u, n) / (double)n, v_avg = or_sum(v, n) / (double)n, w_avg = or_sum(w, n) / (double)n; /* :206-208 */
        if (report) {
            double *r = report + 6 * (iter - 1);
            r[0] = u_avg; r[1] = v_avg; r[2] = w_avg; r[3] = peclet[0]; r[4] = norms[1]; r[5] = norms[0];
        }
        if (isnan(u_avg) || isnan(v_avg) || isnan(w_avg)) st = ORC_ERR_SOLUTION_DIVERGED; /* :217-221 */
    }
    /* :227-242: one gradient evaluation per cell after the loop; the accumulated |gradients| are never printed, but the
     * pass still panics on an unsupported scheme (:870,:901,:948) or a singular least-squares matrix (:850,:943) */
    for (int64_t i = 0; i < n && st == ORC_OK; i++) {
        Vec3 gp;
        Tensor3 gu;
        st = or_calculate_pressure_gradient(m, p, i, s->gradient_reconstruction, s->q1_compat, &gp);
        if (st == ORC_OK) st = or_calculate_velocity_gradient(m, u, v, w, i, s->gradient_reconstruction, &gu);
    }
    or_csr_free(a_di); or_csr_free(a_u); or_csr_free(a_v); or_csr_free(a_w);
    free(b_u_di); free(b_v_di); free(b_w_di); free(b_u); free(b_v); free(b_w); free(p_prime); free(b_p);
    free(du); free(dv); free(dw);
    return st;
}

/* solver.rs:414-509 */
int or_initialize_pressure_field(const OrMesh *m, double *p) {
    int64_t n = m->n_cells;
    int64_t cap = m->cell_face_ptr[n] + n + 1, cnt = 0;
    int64_t *ri = (int64_t *)malloc(8 * (size_t)cap), *ci = (int64_t *)malloc(8 * (size_t)cap);
    double *vv = (double *)malloc(8 * (size_t)cap);
    double *b = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t c = 0; c < n; c++) {
        double a_p = 0.;
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            Vec3 nrm = outward_normal(m, f, c);
            const OrZone *z = zone_of(m, f);
            double a_nb = 0., source = 0.;
            int64_t nb = -1;
            if (z->zone_type == ORC_BC_INTERIOR) { /* :451-466 */
                nb = m->face_c0[f] == c ? m->face_c1[f] : m->face_c0[f];
                a_nb = v_dot(v_reciprocal(v_sub(m->cell_centroid[c], m->cell_centroid[nb])), nrm) * (m->face_area[f] / m->cell_volume[c]);
            } else if (z->zone_type == ORC_BC_PRESSURE_INLET || z->zone_type == ORC_BC_PRESSURE_OUTLET) { /* :467-474 */
                a_nb = v_dot(v_reciprocal(v_sub(m->cell_centroid[c], m->face_centroid[f])), nrm) * (m->face_area[f] / m->cell_volume[c]);
                source = a_nb * z->scalar_value;
            }
            if (nb >= 0) { ri[cnt] = c; ci[cnt] = nb; vv[cnt] = -a_nb; cnt++; }
            b[c] += source;
            a_p += a_nb;
        }
        ri[cnt] = c; ci[cnt] = c; vv[cnt] = a_p; cnt++;
    }
    OrCsr *a = or_csr_from_coo(n, n, cnt, ri, ci, vv);
    int st = or_iterative_solve(a, b, p, 10, ORC_SOLVER_JACOBI, 0.1, 1e-6, ORC_PRECOND_JACOBI); /* :498-507 */
    or_csr_free(a); free(ri); free(ci); free(vv); free(b);
    return st;
}

/* solver.rs:511-696.  psi_out (optional, n doubles): the potential the reference writes to ./examples/psi.csv */
int or_initialize_velocity_field(const OrMesh *m, double *u, double *v, double *w, double *psi_out) {
    int64_t n = m->n_cells;
    int64_t cap = m->cell_face_ptr[n] + n + 1, cnt = 0;
    int64_t *ri = (int64_t *)malloc(8 * (size_t)cap), *ci = (int64_t *)malloc(8 * (size_t)cap);
    double *vv = (double *)malloc(8 * (size_t)cap);
    double *b = (double *)calloc((size_t)(n > 0 ? n : 1), 8), *psi = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    for (int64_t c = 0; c < n; c++) {
        double a_p = 0.;
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            Vec3 nrm = outward_normal(m, f, c);
            const OrZone *z = zone_of(m, f);
            double a_nb = 0., source = 0.;
            int64_t nb = -1;
            if (z->zone_type == ORC_BC_INTERIOR) { /* :536-552 */
                nb = m->face_c0[f] == c ? m->face_c1[f] : m->face_c0[f];
                a_nb = v_dot(v_reciprocal(v_sub(m->cell_centroid[c], m->cell_centroid[nb])), nrm) * (m->face_area[f] / m->cell_volume[c]);
            } else if (z->zone_type == ORC_BC_VELOCITY_INLET) { /* :553-560 */
                source = -v_dot(z->vector_value, nrm);
            } else if (z->zone_type == ORC_BC_PRESSURE_OUTLET) { /* :565-575: no area/volume factor */
                a_nb = v_dot(v_reciprocal(v_sub(m->cell_centroid[c], m->face_centroid[f])), nrm);
            }
            if (nb >= 0) { ri[cnt] = c; ci[cnt] = nb; vv[cnt] = -a_nb; cnt++; }
            b[c] += source;
            a_p += a_nb;
        }
        ri[cnt] = c; ci[cnt] = c; vv[cnt] = a_p; cnt++;
    }
    OrCsr *a = or_csr_from_coo(n, n, cnt, ri, ci, vv);
    int st = or_iterative_solve(a, b, psi, 10, ORC_SOLVER_BICGSTAB, 0.1, 1e-6, ORC_PRECOND_JACOBI); /* :595-604 */
    or_csr_free(a); free(ri); free(ci); free(vv); free(b);
    if (psi_out) memcpy(psi_out, psi, 8 * (size_t)n);
    /* :606-618 write_data / write_gradients of psi into ./examples: files, not part of the returned fields */
    for (int64_t c = 0; c < n && st == ORC_OK; c++) { /* :622-692 least-squares gradient of psi over the interior neighbours */
        int64_t nf = m->cell_face_ptr[c + 1] - m->cell_face_ptr[c];
        double (*x)[3] = malloc(sizeof(double[3]) * (size_t)(nf > 0 ? nf : 1));
        double (*rhs)[3] = malloc(sizeof(double[3]) * (size_t)(nf > 0 ? nf : 1));
        int64_t k = 0;
        for (int64_t q = m->cell_face_ptr[c]; q < m->cell_face_ptr[c + 1]; q++) {
            int64_t f = m->cell_faces[q];
            if (m->face_c1[f] < 0) continue; /* cell_indices.len() == 2 */
            int64_t nb = m->face_c0[f] != c ? m->face_c0[f] : m->face_c1[f];
            Vec3 d = v_sub(m->cell_centroid[nb], m->cell_centroid[c]);
            x[k][0] = d.x; x[k][1] = d.y; x[k][2] = d.z;
            rhs[k][0] = psi[nb] - psi[c];
            k++;
        }
        int cols[3], dim = 0;
        for (int j = 0; j < 3; j++) { /* :648-654 columns with a non-zero minimum or maximum */
            double mn = 0., mx = 0.;
            for (int64_t r = 0; r < k; r++) { if (r == 0 || x[r][j] < mn) mn = x[r][j]; if (r == 0 || x[r][j] > mx) mx = x[r][j]; }
            if (mn != 0. || mx != 0.) cols[dim++] = j;
        }
        double ata[3][3], atb[3][3], cv[3] = {0., 0., 0.};
        normal_equations(k, x, dim, cols, 1, rhs, ata, atb);
        if (nalgebra_try_inverse(dim, ata)) inv_times(dim, ata, atb[0], cv); /* None: "Could not invert. Skipping." */
        double comp[3] = {0., 0., 0.};
        for (int j = 0; j < dim; j++) comp[cols[j]] = cv[j];
        u[c] = isnan(comp[0]) ? 0. : comp[0];
        v[c] = isnan(comp[1]) ? 0. : comp[1];
        w[c] = isnan(comp[2]) ? 0. : comp[2];
        free(x); free(rhs);
    }
    free(psi);
    return st;
}

/* &a * (1 - f) + &a_di * f  (solver.rs:319,329,339): CSR scale then CSR add over the union
 * pattern; both operands share one pattern here so the sum is entry-wise. */
static OrCsr *blend(const OrCsr *a, const OrCsr *a_di, double f) {
    OrCsr *r = or_csr_clone(a);
    for (int64_t i = 0; i < a->nrows; i++)
        for (int64_t q = a->row_ptr[i]; q < a->row_ptr[i + 1]; q++) {
            int64_t t = or_csr_find(a_di, i, a->col[q]);
            double x = a->val[q] * (1. - f);
            r->val[q] = (t >= 0) ? x + a_di->val[t] * f : x;
        }
    return r;
}

/* solver.rs:246-352 */
int or_initialize_flow(const OrMesh *m, double mu, double rho, uint64_t iteration_count, int q1_compat,
                       double *u, double *v, double *w, double *p) {
    int64_t n = m->n_cells;
    size_t nn = (size_t)(n > 0 ? n : 1);
    memset(u, 0, 8 * (size_t)n); memset(v, 0, 8 * (size_t)n); memset(w, 0, 8 * (size_t)n); memset(p, 0, 8 * (size_t)n);
    OrCsr *a_di = NULL;
    double *b_u_di = (double *)calloc(nn, 8), *b_v_di = (double *)calloc(nn, 8), *b_w_di = (double *)calloc(nn, 8);
    double *b_u = (double *)calloc(nn, 8), *b_v = (double *)calloc(nn, 8), *b_w = (double *)calloc(nn, 8);
    int st = or_build_momentum_diffusion_matrix(m, ORC_DIFFUSION_CD, mu, &a_di, b_u_di, b_v_di, b_w_di);
    OrCsr *a_u = NULL, *a_v = NULL, *a_w = NULL;
    if (st == ORC_OK) {
        a_u = or_initialize_momentum_matrix(m); a_v = or_initialize_momentum_matrix(m); a_w = or_initialize_momentum_matrix(m);
        st = or_initialize_pressure_field(m, p); /* :287 */
    }
    if (st == ORC_OK) {
        OrcSettings s;
        or_settings_default(&s);
        s.momentum = ORC_MOMENTUM_UD;
        s.velocity_interpolation = ORC_VINTERP_LINEAR_WEIGHTED;
        s.pressure_interpolation = ORC_PINTERP_LINEAR_WEIGHTED;
        s.q1_compat = q1_compat;
        double peclet[3];
        st = or_build_momentum_advection_matrices(a_u, a_v, a_w, b_u, b_v, b_w, a_di, m, u, v, w, p, &s, rho, peclet); /* :288-306 */
    }
    if (st == ORC_OK) {
        for (int64_t i = 0; i < n; i++) { b_u[i] += b_u_di[i]; b_v[i] += b_v_di[i]; b_w[i] += b_w_di[i]; }
        double diffusion_fraction = 1.;
        while (diffusion_fraction >= 0. && st == ORC_OK) { /* :316-349 */
            OrCsr *mu_ = blend(a_u, a_di, diffusion_fraction);
            st = or_iterative_solve(mu_, b_u, u, iteration_count, ORC_SOLVER_BICGSTAB, 0.5, 1e-6, ORC_PRECOND_JACOBI);
            or_csr_free(mu_);
            if (st) break;
            OrCsr *mv_ = blend(a_v, a_di, diffusion_fraction);
            st = or_iterative_solve(mv_, b_v, v, iteration_count, ORC_SOLVER_BICGSTAB, 0.5, 1e-6, ORC_PRECOND_JACOBI);
            or_csr_free(mv_);
            if (st) break;
            OrCsr *mw_ = blend(a_w, a_di, diffusion_fraction);
            st = or_iterative_solve(mw_, b_w, w, iteration_count, ORC_SOLVER_BICGSTAB, 0.5, 1e-6, ORC_PRECOND_JACOBI);
            or_csr_free(mw_);
            diffusion_fraction -= 0.2;
        }
    }
    or_csr_free(a_di); or_csr_free(a_u); or_csr_free(a_v); or_csr_free(a_w);
    free(b_u_di); free(b_v_di); free(b_w_di); free(b_u); free(b_v); free(b_w);
    return st;
}
