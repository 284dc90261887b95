// assembly.hip — face-loop assembly of the SIMPLE iteration on the device.
// Reference: src/discretization.rs (all), src/solver.rs:774-802, 874-902, 952-1227.
//
// The reference walks cells, and inside each (cell, face) visit recomputes two or four
// Green-Gauss pressure gradients (solver.rs:1082-1084, 1139-1140).  p, u, v, w are immutable
// during an assembly, so here every gradient is evaluated once per cell (K9), every face flux and
// face pressure once per face (K10 — the two sides of a face see exactly opposite fluxes once the
// Rhie-Chow diagonals are frozen), and the cell kernel (K11/K12) only gathers.
// The cell kernels visit a cell's faces in ascending face id and use the reference's operator
// order, so with -ffp-contract=off every assembled coefficient is bit-identical to the CPU
// oracle in frozen-diagonal mode.  All kernels are HBM-bound gathers; no atomics are needed
// because each thread owns its matrix row.
#include <algorithm>
#include <cmath>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "assembly.hpp"
#include "linalg_kernels.hpp"

namespace orc {


struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { return {x, y, z}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vneg(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 vmuls(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }   // Vector * Float (lib.rs:479-492)
__device__ __forceinline__ V3 vdivs(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 vadds(V3 a, double s) { return {a.x + s, a.y + s, a.z + s}; }
__device__ __forceinline__ double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double vnorm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// Float * Vector (lib.rs:540-548): z := rhs.y * self when q1
__device__ __forceinline__ V3 smulv(double s, V3 a, int q1) { return {a.x * s, a.y * s, (q1 ? a.y : a.z) * s}; }

__device__ __forceinline__ V3 face_normal(const MeshDev &M, int f) { return mk(M.nx[f], M.ny[f], M.nz[f]); }
__device__ __forceinline__ V3 cell_centroid(const MeshDev &M, int c) { return mk(M.ccx[c], M.ccy[c], M.ccz[c]); }
__device__ __forceinline__ V3 face_centroid(const MeshDev &M, int f) { return mk(M.fcx[f], M.fcy[f], M.fcz[f]); }
__device__ __forceinline__ V3 zone_vec(const MeshDev &M, int z) { return mk(M.zvec[3 * z], M.zvec[3 * z + 1], M.zvec[3 * z + 2]); }

__device__ __forceinline__ void raise(int *status, int code) { atomicCAS(status, 0, code); }

#define GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ K14: diffusion matrix (once)
// discretization.rs:39-131
__global__ void diffusion_k(MeshDev M, SellDev P, double mu, double *__restrict__ a_di, double *__restrict__ b_u,
                            double *__restrict__ b_v, double *__restrict__ b_w, int *status) {
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        double a_p = 0., bu = 0., bv = 0., bw = 0.;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            double d_i = 0.;
            if (zt == ORC_BC_WALL || zt == ORC_BC_VELOCITY_INLET) {  // :70-79
                d_i = mu * M.area[f] / vnorm(vsub(face_centroid(M, f), cc));
                const V3 sc = vmuls(zone_vec(M, z), d_i);
                bu += sc.x; bv += sc.y; bw += sc.z;
            } else if (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET || zt == ORC_BC_SYMMETRY) {  // :80-88
                d_i = 0.;
            } else if (zt == ORC_BC_INTERIOR) {  // :89-113
                const int nb = (M.c0[f] == c) ? M.c1[f] : M.c0[f];
                d_i = mu * M.area[f] / vnorm(vsub(cell_centroid(M, nb), cc));
                a_di[M.cfpos[q]] = -d_i;  // :125
            } else {
                raise(status, ORC_ERR_UNSUPPORTED_BC);  // :114-117
            }
            a_p += d_i;
        }
        a_di[P.diag_pos[c]] = a_p;  // :128
        b_u[c] = bu; b_v[c] = bv; b_w[c] = bw;
    }
}

// discretization.rs:450-472
__global__ void init_momentum_k(MeshDev M, SellDev P, double *__restrict__ a) {
    GRID_STRIDE(c, M.n_own) {
        a[P.diag_pos[c]] = 1.;
        const double nf = (double)(M.cfp[c + 1] - M.cfp[c]);
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q)
            if (M.cfpos[q] >= 0) a[M.cfpos[q]] = -1. / nf;
    }
}

__global__ void extract_diag_k(SellDev P, const double *__restrict__ a, double *__restrict__ d) {
    GRID_STRIDE(c, P.n) d[c] = a[P.diag_pos[c]];
}

// ------------------------------------------------------------------ initialize_pressure_field's Laplace system
// solver.rs:436-492 on the mesh pattern: interior faces couple the two cells with
// a_nb = (cc - c_nb).reciprocal() . n_out * (A / V); pressure boundaries add a_nb to the diagonal and a_nb * P_bc to b;
// every other zone type contributes nothing.  Sums run in Cell.face_indices order from 0.0 like the reference's.
__device__ __forceinline__ V3 vreciprocal(V3 a) {  // lib.rs:246-252
    return mk(a.x != 0. ? 1. / a.x : 0., a.y != 0. ? 1. / a.y : 0., a.z != 0. ? 1. / a.z : 0.);
}

__global__ void laplace_p_k(MeshDev M, SellDev P, double *__restrict__ a, double *__restrict__ b, int *status) {
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        const double vol = M.vol[c];
        double a_p = 0., src = 0.;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            V3 nrm = mk(M.nx[f], M.ny[f], M.nz[f]);
            if (M.c0[f] != c) nrm = vneg(nrm);  // get_outward_face_normal (mesh.rs:216-222)
            double a_nb = 0., source = 0.;
            if (zt == ORC_BC_INTERIOR) {  // :451-466
                if (M.cfpos[q] < 0) { raise(status, ORC_ERR_BAD_ARGUMENT); continue; }  // cell_indices[1] out of bounds in the reference
                const int nb = (M.c0[f] == c) ? M.c1[f] : M.c0[f];
                a_nb = vdot(vreciprocal(vsub(cc, cell_centroid(M, nb))), nrm) * (M.area[f] / vol);
                a[M.cfpos[q]] = -a_nb;  // :486
            } else if (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET) {  // :467-474
                a_nb = vdot(vreciprocal(vsub(cc, face_centroid(M, f))), nrm) * (M.area[f] / vol);
                source = a_nb * M.zscal[z];
            }
            src += source;  // :488
            a_p += a_nb;    // :489
        }
        a[P.diag_pos[c]] = a_p;  // :491
        b[c] = src;
    }
}

// &a * (1. - f) + &a_di * f (solver.rs:319, 329, 339): CSR scale, CSR scale, CSR add — entry-wise on the shared pattern
__global__ void blend_k(int64_t len, const double *__restrict__ a, const double *__restrict__ a_di, double one_minus_f, double f,
                        double *__restrict__ out) {
    GRID_STRIDE(i, len) out[i] = a[i] * one_minus_f + a_di[i] * f;
}

// ------------------------------------------------------------------ K9: Green-Gauss gradients
// get_face_pressure with PressureInterpolation::Linear (solver.rs:1114-1128)
__device__ __forceinline__ double face_pressure_linear(const MeshDev &M, const double *__restrict__ p, int f, int zt, int z) {
    if (zt == ORC_BC_INTERIOR) return (p[M.c0[f]] + p[M.c1[f]]) * 0.5;
    if (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET) return M.zscal[z];
    return p[M.c0[f]];  // Symmetry | Wall | VelocityInlet
}

__device__ __forceinline__ bool bc_supported(int zt) {
    return zt == ORC_BC_INTERIOR || zt == ORC_BC_WALL || zt == ORC_BC_SYMMETRY || zt == ORC_BC_VELOCITY_INLET ||
           zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET;
}

// calculate_pressure_gradient, GreenGauss(CellBased) (solver.rs:883-900); returns (gx, gy, gy) under Q1
__global__ void grad_p_k(MeshDev M, const double *__restrict__ p, double *__restrict__ gp, int q1, int *status) {
    const int64_t n = M.n_cells;
    GRID_STRIDE(c, M.n_own) {
        const double vol = M.vol[c];
        V3 acc = mk(0., 0., 0.);
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            if (!bc_supported(zt)) { raise(status, ORC_ERR_UNSUPPORTED_BC); continue; }  // solver.rs:1148
            const double fv = face_pressure_linear(M, p, f, zt, z);
            V3 nrm = face_normal(M, f);
            if (M.c0[f] != c) nrm = vneg(nrm);
            acc = vadd(acc, smulv(fv * (M.area[f] / vol), nrm, q1));  // :896-898
        }
        gp[c] = acc.x; gp[n + c] = acc.y; gp[2 * n + c] = acc.z;
    }
}

// get_face_velocity(…, Linear) (solver.rs:952-987)
__device__ __forceinline__ V3 face_velocity_linear(const MeshDev &M, const double *__restrict__ u, const double *__restrict__ v,
                                                   const double *__restrict__ w, int f, int zt, int z) {
    const int a = M.c0[f];
    if (zt == ORC_BC_WALL || zt == ORC_BC_VELOCITY_INLET) return zone_vec(M, z);
    if (zt == ORC_BC_INTERIOR) {
        const int b = M.c1[f];
        return vdivs(vadd(mk(u[a], v[a], w[a]), mk(u[b], v[b], w[b])), 2.);
    }
    return mk(u[a], v[a], w[a]);
}

// calculate_velocity_gradient, GreenGauss arm (solver.rs:784-801): row = velocity component
__global__ void grad_u_k(MeshDev M, const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                         double *__restrict__ gu, int *status) {
    const int64_t n = M.n_cells;
    GRID_STRIDE(c, M.n_own) {
        const double vol = M.vol[c];
        V3 tx = mk(0., 0., 0.), ty = tx, tz = tx;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            if (!bc_supported(zt)) { raise(status, ORC_ERR_UNSUPPORTED_BC); continue; }  // solver.rs:1001
            const V3 fv = face_velocity_linear(M, u, v, w, f, zt, z);
            V3 nrm = face_normal(M, f);
            if (M.c0[f] != c) nrm = vneg(nrm);
            const V3 nn = vmuls(nrm, M.area[f] / vol);  // :799
            tx = vadd(tx, mk(fv.x * nn.x, fv.x * nn.y, fv.x * nn.z));  // outer (lib.rs:275-293)
            ty = vadd(ty, mk(fv.y * nn.x, fv.y * nn.y, fv.y * nn.z));
            tz = vadd(tz, mk(fv.z * nn.x, fv.z * nn.y, fv.z * nn.z));
        }
        gu[0 * n + c] = tx.x; gu[1 * n + c] = tx.y; gu[2 * n + c] = tx.z;
        gu[3 * n + c] = ty.x; gu[4 * n + c] = ty.y; gu[5 * n + c] = ty.z;
        gu[6 * n + c] = tz.x; gu[7 * n + c] = tz.y; gu[8 * n + c] = tz.z;
    }
}

// ------------------------------------------------------------------ least-squares gradients (solver.rs:803-869, 903-947)
// One thread per cell: the rows of the n x 3 system are the cell's faces in Cell.face_indices order — neighbour centroid
// minus cell centroid with the value DIFFERENCE on interior faces, face centroid minus cell centroid with the face VALUE
// itself on boundary faces (the reference's own formulation, solver.rs:830-838, 925-934) — and the normal equations are
// accumulated face by face in nalgebra's small-matrix product order (one gemv per output column: left-to-right sums,
// the first term assigned, (1 * a) * b per term), inverted with its closed 3 x 3 form (linalg/inverse.rs) and applied
// by one more gemv.  A zero determinant is the reference's `try_inverse().unwrap()` panic.
struct Lsq3 {
    double ata[3][3], atb[3][3];
    bool first = true;
    __device__ __forceinline__ void add_row(const double x[3], const double *b, int nb) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double t = (1. * x[i]) * x[j];
                ata[i][j] = first ? t : t + 1. * ata[i][j];
            }
            for (int q = 0; q < nb; ++q) {
                const double t = (1. * x[i]) * b[q];
                atb[q][i] = first ? t : t + 1. * atb[q][i];
            }
        }
        first = false;
    }
};
// nalgebra try_inverse, dimension 3, in place; false = singular
__device__ __forceinline__ bool inverse3(double a[3][3]) {
    const double m11 = a[0][0], m12 = a[0][1], m13 = a[0][2], m21 = a[1][0], m22 = a[1][1], m23 = a[1][2], m31 = a[2][0], m32 = a[2][1], m33 = a[2][2];
    const double minor_m12_m23 = m22 * m33 - m32 * m23;
    const double minor_m11_m23 = m21 * m33 - m31 * m23;
    const double minor_m11_m22 = m21 * m32 - m31 * m22;
    const double determinant = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (determinant == 0.) return false;
    a[0][0] = minor_m12_m23 / determinant;
    a[0][1] = (m13 * m32 - m33 * m12) / determinant;
    a[0][2] = (m12 * m23 - m22 * m13) / determinant;
    a[1][0] = -minor_m11_m23 / determinant;
    a[1][1] = (m11 * m33 - m31 * m13) / determinant;
    a[1][2] = (m13 * m21 - m23 * m11) / determinant;
    a[2][0] = minor_m11_m22 / determinant;
    a[2][1] = (m12 * m31 - m32 * m11) / determinant;
    a[2][2] = (m11 * m22 - m21 * m12) / determinant;
    return true;
}
__device__ __forceinline__ void inv_times3(const double ainv[3][3], const double b[3], double out[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double y = (1. * ainv[i][0]) * b[0];
        y = (1. * ainv[i][1]) * b[1] + 1. * y;
        y = (1. * ainv[i][2]) * b[2] + 1. * y;
        out[i] = y;
    }
}

__global__ void grad_p_lsq_k(MeshDev M, const double *__restrict__ p, double *__restrict__ gp, int *status) {
    const int64_t n = M.n_cells;
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        Lsq3 L;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            if (!bc_supported(zt)) { raise(status, ORC_ERR_UNSUPPORTED_BC); continue; }
            V3 d;
            double val;
            if (zt == ORC_BC_INTERIOR) {
                const int nb = (M.c0[f] == c) ? M.c1[f] : M.c0[f];
                d = vsub(cell_centroid(M, nb), cc);
                val = p[nb] - p[c];
            } else {  // get_face_pressure(…, None, None): zone scalar on pressure zones, cell-0 value elsewhere
                d = vsub(face_centroid(M, f), cc);
                val = (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET) ? M.zscal[z] : p[M.c0[f]];
            }
            const double x[3] = {d.x, d.y, d.z};
            L.add_row(x, &val, 1);
        }
        double g[3] = {0., 0., 0.};
        if (L.first || !inverse3(L.ata)) raise(status, ORC_ERR_SINGULAR_MATRIX);  // a cell without faces: 0 x 0 ... the reference cannot get there either
        else inv_times3(L.ata, L.atb[0], g);
        gp[c] = g[0]; gp[n + c] = g[1]; gp[2 * n + c] = g[2];
    }
}

__global__ void grad_u_lsq_k(MeshDev M, const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                             double *__restrict__ gu, int *status) {
    const int64_t n = M.n_cells;
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        Lsq3 L;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            if (!bc_supported(zt)) { raise(status, ORC_ERR_UNSUPPORTED_BC); continue; }
            V3 d;
            double b[3];
            if (zt == ORC_BC_INTERIOR) {
                const int nb = (M.c0[f] == c) ? M.c1[f] : M.c0[f];
                d = vsub(cell_centroid(M, nb), cc);
                b[0] = u[nb] - u[c]; b[1] = v[nb] - v[c]; b[2] = w[nb] - w[c];
            } else {  // get_face_velocity(…, None): zone vector on walls / velocity inlets, cell-0 velocity elsewhere
                d = vsub(face_centroid(M, f), cc);
                const int a0 = M.c0[f];
                const V3 fv = (zt == ORC_BC_WALL || zt == ORC_BC_VELOCITY_INLET) ? zone_vec(M, z) : mk(u[a0], v[a0], w[a0]);
                b[0] = fv.x; b[1] = fv.y; b[2] = fv.z;
            }
            const double x[3] = {d.x, d.y, d.z};
            L.add_row(x, b, 3);
        }
        double g[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
        if (L.first || !inverse3(L.ata)) raise(status, ORC_ERR_SINGULAR_MATRIX);
        else {
            inv_times3(L.ata, L.atb[0], g[0]);
            inv_times3(L.ata, L.atb[1], g[1]);
            inv_times3(L.ata, L.atb[2], g[2]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) gu[(3 * r + k) * n + c] = g[r][k];
    }
}

// ------------------------------------------------------------------ initialize_velocity_field (solver.rs:511-696)
// Potential-flow system for psi on the mesh pattern: interior coupling (cc - cc_nb).reciprocal() . n * A / V, velocity inlets
// put -U . n on the right-hand side, a pressure outlet adds (cc - fc).reciprocal() . n to the diagonal (no A / V: :565-575).
__global__ void psi_system_k(MeshDev M, SellDev P, double *__restrict__ a, double *__restrict__ b, int *status) {
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        const double vol = M.vol[c];
        double a_p = 0., src = 0.;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int z = M.fzone[f];
            const int zt = M.ztype[z];
            V3 nrm = mk(M.nx[f], M.ny[f], M.nz[f]);
            if (M.c0[f] != c) nrm = vneg(nrm);
            double a_nb = 0., source = 0.;
            if (zt == ORC_BC_INTERIOR) {
                if (M.cfpos[q] < 0) { raise(status, ORC_ERR_BAD_ARGUMENT); continue; }
                const int nb = (M.c0[f] == c) ? M.c1[f] : M.c0[f];
                a_nb = vdot(vreciprocal(vsub(cc, cell_centroid(M, nb))), nrm) * (M.area[f] / vol);
                a[M.cfpos[q]] = -a_nb;
            } else if (zt == ORC_BC_VELOCITY_INLET) {
                source = -vdot(zone_vec(M, z), nrm);
            } else if (zt == ORC_BC_PRESSURE_OUTLET) {
                a_nb = vdot(vreciprocal(vsub(cc, face_centroid(M, f))), nrm);
            }
            src += source;
            a_p += a_nb;
        }
        a[P.diag_pos[c]] = a_p;
        b[c] = src;
    }
}

// velocity = least-squares gradient of psi over the interior neighbours, all-zero columns dropped (:622-692)
__global__ void psi_velocity_k(MeshDev M, const double *__restrict__ psi, double *__restrict__ u, double *__restrict__ v, double *__restrict__ w) {
    GRID_STRIDE(c, M.n_own) {
        const V3 cc = cell_centroid(M, (int)c);
        // pass 1: which columns have a non-zero minimum or maximum
        double mn[3] = {0., 0., 0.}, mx[3] = {0., 0., 0.};
        int rows = 0;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            if (M.c1[f] < 0) continue;
            const int nb = (M.c0[f] != c) ? M.c0[f] : M.c1[f];
            const V3 d = vsub(cell_centroid(M, nb), cc);
            const double x[3] = {d.x, d.y, d.z};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (rows == 0 || x[j] < mn[j]) mn[j] = x[j];
                if (rows == 0 || x[j] > mx[j]) mx[j] = x[j];
            }
            ++rows;
        }
        int cols[3], dim = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (mn[j] != 0. || mx[j] != 0.) cols[dim++] = j;
        // pass 2: normal equations of the selected columns, face by face
        double ata[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}}, atb[3] = {0., 0., 0.};
        bool first = true;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            if (M.c1[f] < 0) continue;
            const int nb = (M.c0[f] != c) ? M.c0[f] : M.c1[f];
            const V3 d = vsub(cell_centroid(M, nb), cc);
            const double x[3] = {d.x, d.y, d.z};
            const double dpsi = psi[nb] - psi[c];
            for (int i = 0; i < dim; ++i) {
                for (int j = 0; j < dim; ++j) {
                    const double t = (1. * x[cols[i]]) * x[cols[j]];
                    ata[i][j] = first ? t : t + 1. * ata[i][j];
                }
                const double t = (1. * x[cols[i]]) * dpsi;
                atb[i] = first ? t : t + 1. * atb[i];
            }
            first = false;
        }
        double cv[3] = {0., 0., 0.};
        bool ok = true;
        if (dim == 1) {
            if (ata[0][0] == 0.) ok = false;
            else cv[0] = (1. * (1. / ata[0][0])) * atb[0];
        } else if (dim == 2) {
            const double m11 = ata[0][0], m12 = ata[0][1], m21 = ata[1][0], m22 = ata[1][1];
            const double determinant = m11 * m22 - m21 * m12;
            if (determinant == 0.) ok = false;
            else {
                const double i00 = m22 / determinant, i01 = -m12 / determinant, i10 = -m21 / determinant, i11 = m11 / determinant;
                cv[0] = (1. * i01) * atb[1] + 1. * ((1. * i00) * atb[0]);
                cv[1] = (1. * i11) * atb[1] + 1. * ((1. * i10) * atb[0]);
            }
        } else if (dim == 3) {
            if (!inverse3(ata)) ok = false;
            else inv_times3(ata, atb, cv);
        }
        double comp[3] = {0., 0., 0.};
        if (ok)
            for (int j = 0; j < dim; ++j) comp[cols[j]] = cv[j];
        u[c] = comp[0] != comp[0] ? 0. : comp[0];
        v[c] = comp[1] != comp[1] ? 0. : comp[1];
        w[c] = comp[2] != comp[2] ? 0. : comp[2];
    }
}

// ------------------------------------------------------------------ K10: face flux / face pressure
struct FaceArgs {
    const double *u, *v, *w, *p, *gp, *du, *dv, *dw;
    int vinterp, pinterp, q1;
    double rho;
};

// get_face_flux seen from cell_indices[0] (solver.rs:1007-1102); the other side is the exact negative GIVEN the same two
// momentum diagonals.  d_i / d_j: a_{u,v,w}.get(i,i) / get(j,j) as the visiting cell sees them (both from the previous
// iteration with frozen diagonals; the reference's in-place mix otherwise, momentum_k<true>).
__device__ __forceinline__ double face_flux_c0(const MeshDev &M, const FaceArgs &A, int f, int zt, int z, int *status, V3 d_i, V3 d_j) {
    const int i = M.c0[f];
    const V3 n = face_normal(M, f);
    if (zt == ORC_BC_WALL || zt == ORC_BC_SYMMETRY) return 0.;  // :1026
    if (zt == ORC_BC_VELOCITY_INLET) return vdot(n, zone_vec(M, z));  // :1027-1039
    if (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET) return vdot(n, mk(A.u[i], A.v[i], A.w[i]));
    if (zt != ORC_BC_INTERIOR) { raise(status, ORC_ERR_UNSUPPORTED_BC); return 0.; }  // :1100
    const int j = M.c1[f];
    const V3 vel_i = mk(A.u[i], A.v[i], A.w[i]), vel_j = mk(A.u[j], A.v[j], A.w[j]);
    if (A.vinterp == ORC_VINTERP_LINEAR) return vdot(n, vdivs(vadd(vel_i, vel_j), 2.));  // :987
    if (A.vinterp == ORC_VINTERP_LINEAR_WEIGHTED) {  // :988-992
        const V3 fc = face_centroid(M, f);
        const double dx0 = vnorm(vsub(cell_centroid(M, i), fc)), dx1 = vnorm(vsub(cell_centroid(M, j), fc));
        return vdot(n, vadd(vel_i, vdivs(vmuls(vsub(vel_j, vel_i), dx0), dx0 + dx1)));
    }
    // Rhie-Chow (:1051-1095)
    const int64_t nc = M.n_cells;
    const V3 ccv = vsub(cell_centroid(M, j), cell_centroid(M, i));
    const double a_i = vnorm(mk(d_i.x * n.x, d_i.y * n.y, d_i.z * n.z));  // discretization.rs:14-23
    const double a_j = vnorm(mk(d_j.x * n.x, d_j.y * n.y, d_j.z * n.z));
    const V3 g_i = mk(A.gp[i], A.gp[nc + i], A.gp[2 * nc + i]), g_j = mk(A.gp[j], A.gp[nc + j], A.gp[2 * nc + j]);
    const double vol_i = M.vol[i], vol_j = M.vol[j];
    const double len = vnorm(ccv);
    const double term_1 = vdot(vadd(vel_i, vel_j), n);
    const double term_2 = (vol_i / a_i + vol_j / a_j) * (A.p[i] - A.p[j]) / len;
    const V3 unit = mk(ccv.x / len, ccv.y / len, ccv.z / len);
    const double term_3 = vdot(vadd(smulv(vol_i / a_i, g_i, A.q1), smulv(vol_j / a_j, g_j, A.q1)), unit);
    return 0.5 * (term_1 + term_2 - term_3);
}

// get_face_pressure (solver.rs:1104-1150)
__device__ __forceinline__ double face_pressure(const MeshDev &M, const FaceArgs &A, int f, int zt, int z) {
    if (zt != ORC_BC_INTERIOR) return face_pressure_linear(M, A.p, f, zt, z);
    const int c0 = M.c0[f], c1 = M.c1[f];
    const double p0 = A.p[c0], p1 = A.p[c1];
    if (A.pinterp == ORC_PINTERP_LINEAR) return (p0 + p1) * 0.5;
    const V3 fc = face_centroid(M, f);
    if (A.pinterp == ORC_PINTERP_LINEAR_WEIGHTED) {  // :1129-1133
        const double x0 = vnorm(vsub(cell_centroid(M, c0), fc)), x1 = vnorm(vsub(cell_centroid(M, c1), fc));
        return p0 + (p1 - p0) * x0 / (x0 + x1);
    }
    // SecondOrder (:1138-1144)
    const int64_t nc = M.n_cells;
    const V3 g0 = mk(A.gp[c0], A.gp[nc + c0], A.gp[2 * nc + c0]), g1 = mk(A.gp[c1], A.gp[nc + c1], A.gp[2 * nc + c1]);
    const V3 r0 = vsub(fc, cell_centroid(M, c0)), r1 = vsub(fc, cell_centroid(M, c1));
    return 0.5 * ((p0 + p1) + (vdot(g0, r0) + vdot(g1, r1)));
}

// MODE 0: momentum phase (flux + face pressure); MODE 1: pressure-correction phase (flux + coefficient)
template <int MODE>
__global__ void face_k(MeshDev M, FaceArgs A, double *__restrict__ flux, double *__restrict__ pf, double *__restrict__ coef, int *status) {
    GRID_STRIDE(f, M.n_faces) {
        const int z = M.fzone[f];
        const int zt = M.ztype[z];
        {
            const int i0 = M.c0[f], j0 = M.c1[f] >= 0 ? M.c1[f] : M.c0[f];
            flux[f] = face_flux_c0(M, A, (int)f, zt, z, status, mk(A.du[i0], A.dv[i0], A.dw[i0]), mk(A.du[j0], A.dv[j0], A.dw[j0]));
        }
        if (MODE == 0) {
            pf[f] = bc_supported(zt) ? face_pressure(M, A, (int)f, zt, z) : 0.;
        } else {
            // discretization.rs:401-436: rho * A^2 / a~ ; a~ from the (sign-insensitive) inward normal
            const V3 n = face_normal(M, (int)f);
            const int i = M.c0[f], j = M.c1[f];
            const double ar = M.area[f];
            if (j >= 0) {
                const double a_int = 0.5 * vnorm(mk((A.du[i] + A.du[j]) * -n.x, (A.dv[i] + A.dv[j]) * -n.y, (A.dw[i] + A.dw[j]) * -n.z));
                coef[f] = A.rho * (ar * ar) / a_int;
            } else {
                const double a_ii = vnorm(mk(A.du[i] * -n.x, A.dv[i] * -n.y, A.dw[i] * -n.z));
                coef[f] = A.rho * (ar * ar) / a_ii;
            }
        }
    }
}

// the coefficient half of face_k<1> alone: what the p' MATRIX needs (no velocity is read)
__global__ void face_coef_k(MeshDev M, const double *__restrict__ du, const double *__restrict__ dv, const double *__restrict__ dw, double rho,
                            double *__restrict__ coef) {
    GRID_STRIDE(f, M.n_faces) {
        const V3 n = face_normal(M, (int)f);
        const int i = M.c0[f], j = M.c1[f];
        const double ar = M.area[f];
        if (j >= 0) {
            const double a_int = 0.5 * vnorm(mk((du[i] + du[j]) * -n.x, (dv[i] + dv[j]) * -n.y, (dw[i] + dw[j]) * -n.z));
            coef[f] = rho * (ar * ar) / a_int;
        } else {
            const double a_ii = vnorm(mk(du[i] * -n.x, dv[i] * -n.y, dw[i] * -n.z));
            coef[f] = rho * (ar * ar) / a_ii;
        }
    }
}

// ------------------------------------------------------------------ K11: momentum matrices
__device__ __forceinline__ double psi_eval(int momentum, double r) {  // lib.rs:107-118
    switch (momentum) {
    case ORC_MOMENTUM_TVD_UD: return 0.;
    case ORC_MOMENTUM_TVD_CD1: return 1.;
    case ORC_MOMENTUM_TVD_LUD: return r;
    case ORC_MOMENTUM_TVD_QUICK: return (3. + r) / 4.;
    default: {  // UMIST; f64::min / max ignore NaN like fmin / fmax
        double acc = INFINITY;
        acc = fmin(acc, 2. * r);
        acc = fmin(acc, (1. + 3. * r) / 4.);
        acc = fmin(acc, (3. + r) / 4.);
        acc = fmin(acc, 2.);
        return fmax(0., acc);
    }
    }
}

struct MomentumArgs {
    const double *u, *v, *w, *gu, *flux, *pf, *a_di, *b_u_di, *b_v_di, *b_w_di;
    double *a_u, *a_v, *a_w, *b_u, *b_v, *b_w, *du, *dv, *dw;
    int momentum, q1;
    double rho;
};

// The reference's in-place diagonal reads (discretization.rs:182-197, 340-351; frozen_diagonals = 0): while cell
// c is assembled, Rhie-Chow sees THIS iteration's diagonal of a neighbour j < c (its row is already rewritten) and LAST
// iteration's of c itself and of every j > c.  New diagonals thus form a triangular recurrence over the cell order; it
// is evaluated level by level (level(c) = 1 + max level of the neighbours below c): the cells of one level only read new
// diagonals of lower levels, so a level is one parallel launch over its cell list.
struct InplaceArgs {
    const int32_t *cells;  // cells of this level (ascending)
    int64_t count;
    const double *du_old, *dv_old, *dw_old;  // diagonals of the previous iteration
    double *pe;                              // [3][n] per-cell Peclet terms, folded after the last level
    FaceArgs F;                              // what get_face_flux reads
    int *status;
};

// discretization.rs:134-356, one thread per cell (= per matrix row)
template <bool kInplace>
__global__ __launch_bounds__(kBlock) void momentum_k(MeshDev M, SellDev P, MomentumArgs A, double *__restrict__ partials, InplaceArgs I) {
    __shared__ double lds[8];
    const int64_t n = M.n_cells;
    double pe_sum = 0., pe_min = INFINITY, pe_max = -INFINITY;
    const bool tvd = A.momentum >= ORC_MOMENTUM_TVD_LUD;
    // [r04] Cells in blocks of one workgroup; XCD g (workgroups g, g + 8, ...) walks a contiguous eighth of the blocks.  With the plain grid
    // stride neighbouring blocks — the two cells of every y- and z-face — sit on different XCDs, each XCD's L2 fetches the face's geometry, flux
    // and the neighbour's fields for itself, and the kernel moved 3.2x its algorithmic bytes (18.3 GB against 5.8 GB).
    const int64_t n_items = kInplace ? I.count : M.n_own;
    const int64_t n_blk = (n_items + blockDim.x - 1) / blockDim.x;
    int64_t vb = blockIdx.x, vb_end = n_blk, vb_step = gridDim.x;
    if (!kInplace && (gridDim.x & 7) == 0 && gridDim.x >= 8) {
        const int64_t per = (n_blk + 7) / 8;
        const int xcd = blockIdx.x & 7;
        vb = (int64_t)xcd * per + (blockIdx.x >> 3);
        vb_end = (int64_t)(xcd + 1) * per < n_blk ? (int64_t)(xcd + 1) * per : n_blk;
        vb_step = gridDim.x >> 3;
    }
    for (; vb < vb_end; vb += vb_step) {
        const int64_t idx = vb * blockDim.x + threadIdx.x;
        if (idx >= n_items) break;
        const int64_t c = kInplace ? (int64_t)I.cells[idx] : idx;
        V3 s_u = mk(0., 0., 0.);  // get_momentum_source_term (solver.rs:698-701)
        const int dpos = P.diag_pos[c];
        const double a_ii_di = A.a_di[dpos];  // :176
        V3 a_p = mk(0., 0., 0.);
        const V3 vel = mk(A.u[c], A.v[c], A.w[c]);
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const bool side0 = M.c0[f] == c;
            V3 n_out = face_normal(M, f);
            if (!side0) n_out = vneg(n_out);
            double face_flux = side0 ? A.flux[f] : -A.flux[f];
            if (kInplace && M.c1[f] >= 0 && I.F.vinterp == ORC_VINTERP_RHIE_CHOW) {
                // the visiting cell's own mix of old and new diagonals (get() on the matrices being rewritten)
                const int i0 = M.c0[f], j0 = M.c1[f];
                const int other = side0 ? j0 : i0;
                const V3 d_self = mk(I.du_old[c], I.dv_old[c], I.dw_old[c]);
                const V3 d_other = other < c ? mk(A.du[other], A.dv[other], A.dw[other]) : mk(I.du_old[other], I.dv_old[other], I.dw_old[other]);
                const double fl = face_flux_c0(M, I.F, f, ORC_BC_INTERIOR, M.fzone[f], I.status, side0 ? d_self : d_other, side0 ? d_other : d_self);
                face_flux = side0 ? fl : -fl;
            }
            const double ar = M.area[f];
            const double f_i = face_flux * ar * A.rho;  // :202
            const double face_pressure = A.pf[f];
            const int c1 = M.c1[f];
            const int nb = c1 < 0 ? -1 : (side0 ? c1 : M.c0[f]);
            V3 a_nb;
            if (A.momentum == ORC_MOMENTUM_UD || (tvd && nb < 0)) {
                a_nb = smulv(fmin(f_i, 0.), mk(1., 1., 1.), A.q1);  // :226, :238
            } else if (A.momentum == ORC_MOMENTUM_CD1) {
                a_nb = vdivs(smulv(f_i, mk(1., 1., 1.), A.q1), 2.);  // :231
            } else {
                const int down = f_i > 0. ? nb : (int)c;  // :244-248
                const V3 dvel = mk(A.u[down], A.v[down], A.w[down]);
                const V3 dv = vsub(dvel, vel);
                if (vnorm(dv) == 0.) {
                    a_nb = vdivs(smulv(f_i, mk(1., 1., 1.), A.q1), 2.);  // :264
                } else {
                    const V3 r_pa = vsub(cell_centroid(M, nb), cell_centroid(M, (int)c));
                    const V3 gx = mk(A.gu[c], A.gu[n + c], A.gu[2 * n + c]);
                    const V3 gy = mk(A.gu[3 * n + c], A.gu[4 * n + c], A.gu[5 * n + c]);
                    const V3 gz = mk(A.gu[6 * n + c], A.gu[7 * n + c], A.gu[8 * n + c]);
                    const V3 inner = mk(vdot(gx, r_pa), vdot(gy, r_pa), vdot(gz, r_pa));  // lib.rs:584-590
                    const V3 two = smulv(2., inner, A.q1);                                   // :276
                    const V3 r = vadds(mk(two.x / dv.x, two.y / dv.y, two.z / dv.z), -1.);  // :277-278
                    const V3 ps = mk(psi_eval(A.momentum, r.x), psi_eval(A.momentum, r.y), psi_eval(A.momentum, r.z));
                    a_nb = vdivs(smulv(f_i, ps, A.q1), 2.);  // :279-283
                }
            }
            a_p = vadd(a_p, vadds(vneg(a_nb), f_i));                           // :290
            s_u = vadd(s_u, vmuls(vmuls(vneg(n_out), face_pressure), ar));     // :291
            if (nb < 0) {  // :294-307
                const int z = M.fzone[f];
                const int zt = M.ztype[z];
                if (zt == ORC_BC_WALL || zt == ORC_BC_VELOCITY_INLET) {
                    const V3 vv = zone_vec(M, z);
                    s_u = vadd(s_u, mk((a_nb.x - f_i) * vv.x, (a_nb.y - f_i) * vv.y, (a_nb.z - f_i) * vv.z));
                } else {
                    s_u = vadd(s_u, mk(0., 0., 0.));
                }
            } else {  // :308-324
                const int pos = M.cfpos[q];
                const double a_ij_di = A.a_di[pos];
                A.a_u[pos] = a_nb.x + a_ij_di;
                A.a_v[pos] = a_nb.y + a_ij_di;
                A.a_w[pos] = a_nb.z + a_ij_di;
            }
        }
        const V3 total = vadd(vadd(s_u, mk(0., 0., 0.)), mk(0., 0., 0.));  // :326
        A.b_u[c] = total.x + A.b_u_di[c];                                   // :327-329 then solver.rs:80-82
        A.b_v[c] = total.y + A.b_v_di[c];
        A.b_w[c] = total.z + A.b_w_di[c];
        const double px = a_p.x / a_ii_di, py = a_p.y / a_ii_di, pz = a_p.z / a_ii_di;  // :331-333
        if (kInplace) { I.pe[c] = px; I.pe[n + c] = py; I.pe[2 * n + c] = pz; }
        else if (I.pe) I.pe[c] = (((0. + px) + py) + pz) / 3.;  // [r05] reference reduction order with frozen diagonals: the cell's term of :338
        pe_max = fmax(pe_max, fmax(px, fmax(py, pz)));
        pe_min = fmin(pe_min, fmin(px, fmin(py, pz)));
        pe_sum += (((0. + px) + py) + pz) / 3.;  // :338
        const double d_u = a_p.x + a_ii_di, d_v = a_p.y + a_ii_di, d_w = a_p.z + a_ii_di;  // :340-351
        A.a_u[dpos] = d_u; A.a_v[dpos] = d_v; A.a_w[dpos] = d_w;
        A.du[c] = d_u; A.dv[c] = d_v; A.dw[c] = d_w;
    }
    if (kInplace) return;  // the statistics are folded by peclet_stats_k once every level is done
    const double t = block_sum(pe_sum, lds);
    const double mn = -block_max(-pe_min, lds);
    const double mx = block_max(pe_max, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        partials[gridDim.x + blockIdx.x] = mn;
        partials[2 * gridDim.x + blockIdx.x] = mx;
    }
}

// pe[c] <- (((0 + pe_x) + pe_y) + pe_z) / 3: the term cell c adds to the reference's running mean (discretization.rs:338)
__global__ void peclet_cell_term_k(double *__restrict__ pe, int64_t n) {
    GRID_STRIDE(c, n) pe[c] = (((0. + pe[c]) + pe[n + c]) + pe[2 * n + c]) / 3.;
}

// the Peclet statistics of momentum_k from the stored per-cell terms (same per-workgroup partial layout)
__global__ __launch_bounds__(kBlock) void peclet_stats_k(const double *__restrict__ pe, int64_t n_own, int64_t n, double *__restrict__ partials) {
    __shared__ double lds[8];
    double pe_sum = 0., pe_min = INFINITY, pe_max = -INFINITY;
    GRID_STRIDE(c, n_own) {
        const double px = pe[c], py = pe[n + c], pz = pe[2 * n + c];
        pe_max = fmax(pe_max, fmax(px, fmax(py, pz)));
        pe_min = fmin(pe_min, fmin(px, fmin(py, pz)));
        pe_sum += (((0. + px) + py) + pz) / 3.;
    }
    const double t = block_sum(pe_sum, lds);
    const double mn = -block_max(-pe_min, lds);
    const double mx = block_max(pe_max, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        partials[gridDim.x + blockIdx.x] = mn;
        partials[2 * gridDim.x + blockIdx.x] = mx;
    }
}

// ------------------------------------------------------------------ K12: pressure-correction system
// discretization.rs:359-448 with the fixed pattern (no COO build / sort per iteration)
__global__ void pressure_k(MeshDev M, SellDev P, const double *__restrict__ flux, const double *__restrict__ coef, double rho,
                           double *__restrict__ a_p_mat, double *__restrict__ b_p) {
    GRID_STRIDE(c, M.n_own) {
        double a_p = 0., b = 0.;
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const double out_flux = (M.c0[f] == c) ? flux[f] : -flux[f];
            b += rho * (-out_flux) * M.area[f];  // :399
            const double a_nb = coef[f];
            const int pos = M.cfpos[q];
            if (pos >= 0) {
                a_p_mat[pos] = -a_nb;  // :423
                a_p += a_nb;
            } else {
                a_p += a_nb / 2.;  // :435
            }
        }
        a_p_mat[P.diag_pos[c]] = a_p;  // :438
        b_p[c] = b;
    }
}

// ------------------------------------------------------------------ K13: velocity / pressure correction
// solver.rs:1170-1227 fused with the report sums of solver.rs:206-208
__global__ __launch_bounds__(kBlock) void correction_k(MeshDev M, const double *__restrict__ du, const double *__restrict__ dv,
                                                       const double *__restrict__ dw, const double *__restrict__ pp,
                                                       double *__restrict__ u, double *__restrict__ v, double *__restrict__ w,
                                                       double *__restrict__ p, double alpha_p, double alpha_u,
                                                       double *__restrict__ partials, int *status, double *__restrict__ corr_cell) {
    __shared__ double lds[8];
    double s_pp = 0., s_corr = 0., s_u = 0., s_v = 0., s_w = 0.;
    GRID_STRIDE(c, M.n_own) {
        const double ppc = pp[c];
        p[c] += alpha_p * ppc;  // :1186
        V3 acc = mk(0., 0., 0.);
        const double iu = du[c], iv = dv[c], iw = dw[c];
        for (int q = M.cfp[c]; q < M.cfp[c + 1]; ++q) {
            const int f = M.cf[q];
            const int zt = M.ztype[M.fzone[f]];
            V3 n = face_normal(M, f);
            const bool side0 = M.c0[f] == c;
            if (!side0) n = vneg(n);
            double pp_nb;
            if (zt == ORC_BC_INTERIOR) pp_nb = pp[side0 ? M.c1[f] : M.c0[f]];
            else if (zt == ORC_BC_PRESSURE_INLET || zt == ORC_BC_PRESSURE_OUTLET) pp_nb = 0.;
            else if (zt == ORC_BC_WALL || zt == ORC_BC_SYMMETRY || zt == ORC_BC_VELOCITY_INLET) pp_nb = ppc;
            else { raise(status, ORC_ERR_UNSUPPORTED_BC); pp_nb = ppc; }  // :1209-1212
            const V3 scaled = mk(n.x / iu, n.y / iv, n.z / iw);
            acc = vadd(acc, vmuls(vmuls(scaled, ppc - pp_nb), M.area[f]));  // :1219
        }
        const double un = u[c] + acc.x * alpha_u, vn = v[c] + acc.y * alpha_u, wn = w[c] + acc.z * alpha_u;  // :1221-1223
        u[c] = un; v[c] = vn; w[c] = wn;
        const double nn = vnorm(acc);
        s_corr += nn * nn;  // :1224
        if (corr_cell) corr_cell[c] = nn * nn;  // reference reduction order: summed cell by cell afterwards (k_apply_correction)
        s_pp += ppc * ppc;
        s_u += un; s_v += vn; s_w += wn;
    }
    const double t0 = block_sum(s_pp, lds), t1 = block_sum(s_corr, lds), t2 = block_sum(s_u, lds), t3 = block_sum(s_v, lds),
                 t4 = block_sum(s_w, lds);
    if (threadIdx.x == 0) {
        const int g = gridDim.x;
        partials[blockIdx.x] = t0; partials[g + blockIdx.x] = t1; partials[2 * g + blockIdx.x] = t2;
        partials[3 * g + blockIdx.x] = t3; partials[4 * g + blockIdx.x] = t4;
    }
}

__global__ void negate_k(double *x) { if (threadIdx.x == 0 && blockIdx.x == 0) *x = -*x; }

__global__ void reduce_minmax_k(const double *__restrict__ partials, int count, double *__restrict__ out) {
    // out[0] = sum(partials[0..count)), out[1] = min(partials[count..2count)), out[2] = max(partials[2count..3count))
    if (threadIdx.x != 0) return;
    double s = 0., mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < count; ++i) { s += partials[i]; mn = fmin(mn, partials[count + i]); mx = fmax(mx, partials[2 * count + i]); }
    out[0] = s; out[1] = mn; out[2] = mx;
}

}  // namespace orc

// ====================================================================== host side
orc::MeshDev OrcMesh::dev() const {
    orc::MeshDev d;
    d.n_cells = n_cells; d.n_own = n_own; d.n_faces = n_faces; d.n_zones = n_zones;
    d.c0 = c0.p; d.c1 = c1.p; d.fzone = fzone.p; d.area = area.p; d.nx = nx.p; d.ny = ny.p; d.nz = nz.p;
    d.fcx = fcx.p; d.fcy = fcy.p; d.fcz = fcz.p; d.ccx = ccx.p; d.ccy = ccy.p; d.ccz = ccz.p; d.vol = vol.p;
    d.cfp = cfp.p; d.cf = cf.p; d.cfpos = cfpos.p; d.ztype = ztype.p; d.zscal = zscal.p; d.zvec = zvec.p;
    return d;
}

namespace orc {

template <class T, class S>
static int upload_as(DevBuf<T> &dst, const S *src, size_t n, size_t stride = 1, size_t off = 0) {
    std::vector<T> tmp(std::max<size_t>(n, 1));
    for (size_t i = 0; i < n; ++i) tmp[i] = (T)src[i * stride + off];
    return dst.upload(tmp.data(), n);
}

int mesh_upload(OrcMesh &m, int64_t n_own, int64_t n_cells, int64_t n_faces, int32_t n_zones, const int64_t *face_c0, const int64_t *face_c1,
                const int32_t *face_zone, const double *face_area, const double *face_normal, const double *face_centroid,
                const double *cell_centroid, const double *cell_volume, const int64_t *cell_face_ptr, const int64_t *cell_faces,
                const int32_t *zone_type, const double *zone_scalar, const double *zone_vector) {
    if (n_cells < 1 || n_faces < 1 || n_zones < 1 || n_own < 1 || n_own > n_cells) return set_error(ORC_ERR_BAD_ARGUMENT, "empty mesh");
    if (n_faces >= ((int64_t)1 << 31) || n_cells >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "mesh too large for 32-bit device indices");
    const int64_t ncf = cell_face_ptr[n_cells];
    if (ncf >= ((int64_t)1 << 31)) return set_error(ORC_ERR_BAD_ARGUMENT, "mesh too large for 32-bit device indices");
    for (int64_t f = 0; f < n_faces; ++f) {
        if (face_c0[f] < 0 || face_c0[f] >= n_cells || face_c1[f] >= n_cells || face_zone[f] < 0 || face_zone[f] >= n_zones)
            return set_error(ORC_ERR_BAD_ARGUMENT, "face %lld: index out of range", (long long)f);
    }
    m.n_cells = n_cells; m.n_own = n_own; m.n_faces = n_faces; m.n_zones = n_zones; m.n_cell_faces = ncf;
    if (m.n_global == 0) m.n_global = n_own;
    // matrix pattern: diagonal + one entry per interior face, ascending columns
    // (CsrMatrix::from(&CooMatrix) at discretization.rs:130,445,471)
    m.h_row_ptr.assign((size_t)n_own + 1, 0);
    std::vector<int64_t> nbr((size_t)ncf);
    for (int64_t c = 0; c < n_own; ++c) {
        int64_t cnt = 1;
        for (int64_t q = cell_face_ptr[c]; q < cell_face_ptr[c + 1]; ++q) {
            const int64_t f = cell_faces[q];
            if (f < 0 || f >= n_faces) return set_error(ORC_ERR_BAD_ARGUMENT, "cell %lld: face index out of range", (long long)c);
            int64_t nb = -1;
            if (face_c1[f] >= 0) nb = face_c0[f] == c ? face_c1[f] : face_c0[f];
            nbr[(size_t)q] = nb;
            if (nb >= 0) cnt++;
        }
        m.h_row_ptr[(size_t)c + 1] = m.h_row_ptr[(size_t)c] + cnt;
    }
    m.h_col.assign((size_t)m.h_row_ptr[(size_t)n_own], 0);
    for (int64_t c = 0; c < n_own; ++c) {
        int64_t *row = m.h_col.data() + m.h_row_ptr[(size_t)c];
        int64_t k = 0;
        row[k++] = c;
        for (int64_t q = cell_face_ptr[c]; q < cell_face_ptr[c + 1]; ++q)
            if (nbr[(size_t)q] >= 0) row[k++] = nbr[(size_t)q];
        std::sort(row, row + k);
        // two cells sharing two faces would give duplicate columns (summed by the reference's
        // COO->CSR); not supported by the fixed-pattern assembly
        for (int64_t t = 1; t < k; ++t)
            if (row[t] == row[t - 1]) return set_error(ORC_ERR_BAD_ARGUMENT, "cells %lld and %lld share more than one face", (long long)c, (long long)row[t]);
    }
    ORC_TRY(sell_from_csr_host(n_own, n_cells, m.h_row_ptr.data(), m.h_col.data(), m.pat));
    // SELL offsets of the neighbour entries
    std::vector<int32_t> cfpos((size_t)ncf, -1);
    {
        std::vector<int64_t> slice_ptr((size_t)m.pat.n_slices + 1);
        ORC_TRY(m.pat.slice_ptr.download(slice_ptr.data(), slice_ptr.size()));
        for (int64_t c = 0; c < n_own; ++c) {
            const int64_t *row = m.h_col.data() + m.h_row_ptr[(size_t)c];
            const int64_t len = m.h_row_ptr[(size_t)c + 1] - m.h_row_ptr[(size_t)c];
            const int64_t base = slice_ptr[(size_t)(c >> 6)] + (c & 63);
            for (int64_t q = cell_face_ptr[c]; q < cell_face_ptr[c + 1]; ++q) {
                if (nbr[(size_t)q] < 0) continue;
                const int64_t k = std::lower_bound(row, row + len, nbr[(size_t)q]) - row;
                cfpos[(size_t)q] = (int32_t)(base + k * 64);
            }
        }
    }
    ORC_TRY(upload_as(m.c0, face_c0, (size_t)n_faces));
    ORC_TRY(upload_as(m.c1, face_c1, (size_t)n_faces));
    ORC_TRY(m.fzone.upload(face_zone, (size_t)n_faces));
    ORC_TRY(m.area.upload(face_area, (size_t)n_faces));
    ORC_TRY(upload_as(m.nx, face_normal, (size_t)n_faces, 3, 0));
    ORC_TRY(upload_as(m.ny, face_normal, (size_t)n_faces, 3, 1));
    ORC_TRY(upload_as(m.nz, face_normal, (size_t)n_faces, 3, 2));
    ORC_TRY(upload_as(m.fcx, face_centroid, (size_t)n_faces, 3, 0));
    ORC_TRY(upload_as(m.fcy, face_centroid, (size_t)n_faces, 3, 1));
    ORC_TRY(upload_as(m.fcz, face_centroid, (size_t)n_faces, 3, 2));
    ORC_TRY(upload_as(m.ccx, cell_centroid, (size_t)n_cells, 3, 0));
    ORC_TRY(upload_as(m.ccy, cell_centroid, (size_t)n_cells, 3, 1));
    ORC_TRY(upload_as(m.ccz, cell_centroid, (size_t)n_cells, 3, 2));
    ORC_TRY(m.vol.upload(cell_volume, (size_t)n_cells));
    ORC_TRY(upload_as(m.cfp, cell_face_ptr, (size_t)n_cells + 1));
    ORC_TRY(upload_as(m.cf, cell_faces, (size_t)ncf));
    ORC_TRY(m.cfpos.upload(cfpos.data(), (size_t)ncf));
    ORC_TRY(m.ztype.upload(zone_type, (size_t)n_zones));
    ORC_TRY(m.zscal.upload(zone_scalar, (size_t)n_zones));
    ORC_TRY(m.zvec.upload(zone_vector, (size_t)3 * n_zones));
    return ORC_OK;
}

static bool is_tvd(int momentum) { return momentum >= ORC_MOMENTUM_TVD_LUD && momentum <= ORC_MOMENTUM_TVD_CD1; }

static int validate_settings(const OrcSettings &s) {
    if (!(s.momentum == ORC_MOMENTUM_UD || s.momentum == ORC_MOMENTUM_CD1 || is_tvd(s.momentum)))
        return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "unsupported momentum scheme");  // discretization.rs:287
    if (s.diffusion != ORC_DIFFUSION_CD) return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "unsupported diffusion scheme");  // :50
    if (s.gradient_reconstruction != ORC_GRAD_GREEN_GAUSS_CELL && s.gradient_reconstruction != ORC_GRAD_LEAST_SQUARES)
        return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "unsupported gradient scheme");  // solver.rs:870,901,948
    if (!(s.pressure_interpolation == ORC_PINTERP_LINEAR || s.pressure_interpolation == ORC_PINTERP_LINEAR_WEIGHTED ||
          s.pressure_interpolation == ORC_PINTERP_SECOND_ORDER))
        return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "unsupported pressure interpolation");  // solver.rs:1136,1145
    if (!(s.velocity_interpolation == ORC_VINTERP_LINEAR || s.velocity_interpolation == ORC_VINTERP_LINEAR_WEIGHTED ||
          s.velocity_interpolation == ORC_VINTERP_RHIE_CHOW))
        return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "unsupported velocity interpolation");  // solver.rs:994,1097
    if (s.frozen_diagonals != 0 && s.frozen_diagonals != 1) return set_error(ORC_ERR_BAD_ARGUMENT, "frozen_diagonals must be 0 or 1");
    if (s.reduction_order != ORC_REDUCTION_TREE && s.reduction_order != ORC_REDUCTION_REFERENCE)
        return set_error(ORC_ERR_BAD_ARGUMENT, "unknown reduction order %d", s.reduction_order);
    return ORC_OK;
}

int fetch_status(SolverState &s) {
    int h = 0;
    ORC_HIP(hipMemcpyAsync(&h, s.dev_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
    ORC_HIP(hipStreamSynchronize(ctx().stream));
    return h;
}

int k_diffusion(SolverState &s) {
    OrcMesh &m = *s.mesh;
    ORC_TRY(s.a_di.zero());
    hipLaunchKernelGGL(diffusion_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), s.mu, s.a_di.p,
                       s.b_u_di.p, s.b_v_di.p, s.b_w_di.p, s.dev_status.p);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int k_init_momentum(SolverState &s) {
    OrcMesh &m = *s.mesh;
    DevBuf<double> *mats[3] = {&s.a_u, &s.a_v, &s.a_w};
    for (auto *a : mats) {
        ORC_TRY(a->zero());
        hipLaunchKernelGGL(init_momentum_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), a->p);
    }
    ORC_HIP(hipGetLastError());
    ORC_TRY(vec_fill(s.du.p, 1., s.n));  // diag = 1.0: what Rhie-Chow reads in iteration 1
    ORC_TRY(vec_fill(s.dv.p, 1., s.n));
    ORC_TRY(vec_fill(s.dw.p, 1., s.n));
    return ORC_OK;
}

int solver_init(SolverState &s, OrcMesh *m, const OrcSettings *settings, double rho, double mu) {
    ORC_TRY(ensure_init());
    s.mesh = m;
    s.settings = *settings;
    s.rho = rho; s.mu = mu; s.n = m->n_cells; s.n_own = m->n_own;
    // the schedule of THIS solver: fixed at its creation (config.hpp; a solver never changes schedule between iterations)
    s.concurrent_momentum = cfg().concurrent_momentum;
    s.two_stream_multigrid = cfg().two_stream_multigrid;
    s.triple_momentum = cfg().triple_momentum;
    s.early_p_hierarchy = cfg().early_p_hierarchy;
    s.sibling_pairing = cfg().amg_sibling;
    ORC_TRY(validate_settings(s.settings));
    const size_t n = (size_t)s.n, pad = (size_t)std::max<int64_t>(m->pat.padded, 1), F = (size_t)m->n_faces;
    DevBuf<double> *nvec[] = {&s.u, &s.v, &s.w, &s.p, &s.p_prime, &s.b_u_di, &s.b_v_di, &s.b_w_di, &s.b_u, &s.b_v, &s.b_w, &s.b_p, &s.du, &s.dv, &s.dw};
    for (auto *b : nvec) { ORC_TRY(b->alloc(n)); ORC_TRY(b->zero()); }
    DevBuf<double> *mats[] = {&s.a_di, &s.a_u, &s.a_v, &s.a_w, &s.a_p};
    for (auto *b : mats) { ORC_TRY(b->alloc(pad)); ORC_TRY(b->zero()); }
    ORC_TRY(s.gp.alloc(3 * n));
    ORC_TRY(s.gu.alloc(9 * n));
    ORC_TRY(s.flux.alloc(F));
    ORC_TRY(s.pf.alloc(F));
    ORC_TRY(s.coef.alloc(F));
    if (s.settings.frozen_diagonals == 0) {
        // the order dependence is defined on ONE process's cell order; a partitioned mesh has no such order across ranks
        if (m->halo.active()) return set_error(ORC_ERR_UNSUPPORTED_SCHEME, "frozen_diagonals = 0 (in-place diagonals) is not available on a partitioned mesh");
        ORC_TRY(s.du_old.alloc(n));
        ORC_TRY(s.dv_old.alloc(n));
        ORC_TRY(s.dw_old.alloc(n));
        ORC_TRY(s.pe.alloc(3 * n));
        // level(c) = 1 + max level(j), j < c a neighbour of c (the matrix pattern is the face-neighbour graph + the diagonal)
        std::vector<int32_t> level((size_t)s.n_own, 0);
        int32_t n_levels = 0;
        for (int64_t c = 0; c < s.n_own; ++c) {
            int32_t lv = 0;
            for (int64_t q = m->h_row_ptr[(size_t)c]; q < m->h_row_ptr[(size_t)c + 1]; ++q) {
                const int64_t j = m->h_col[(size_t)q];
                if (j < c) lv = std::max(lv, level[(size_t)j] + 1);
            }
            level[(size_t)c] = lv;
            n_levels = std::max(n_levels, lv + 1);
        }
        s.level_ptr.assign((size_t)n_levels + 1, 0);
        for (int64_t c = 0; c < s.n_own; ++c) ++s.level_ptr[(size_t)level[(size_t)c] + 1];
        for (int32_t l = 0; l < n_levels; ++l) s.level_ptr[(size_t)l + 1] += s.level_ptr[(size_t)l];
        std::vector<int32_t> cells((size_t)std::max<int64_t>(s.n_own, 1));
        std::vector<int64_t> cur(s.level_ptr.begin(), s.level_ptr.end() - 1);
        for (int64_t c = 0; c < s.n_own; ++c) cells[(size_t)cur[(size_t)level[(size_t)c]]++] = (int32_t)c;
        ORC_TRY(s.level_cells.upload(cells.data(), cells.size()));
    } else if (s.settings.reduction_order == ORC_REDUCTION_REFERENCE && !m->halo.active()) {
        ORC_TRY(s.pe.alloc(n));  // [r05] per-cell terms of the report's sums in the reference's association (k_momentum, k_apply_correction)
    }
    ORC_TRY(s.partials.alloc((size_t)8 * kMaxPartials));
    ORC_TRY(s.scal.alloc(16));
    ORC_TRY(s.dev_status.alloc(1));
    ORC_TRY(s.dev_status.zero());
    ORC_TRY(k_diffusion(s));       // solver.rs:41-42
    ORC_TRY(k_init_momentum(s));   // solver.rs:43-45
    int st = fetch_status(s);
    if (st) return set_error(st, "unsupported boundary condition in mesh zones");
    return ORC_OK;
}

int k_gradients(SolverState &s, bool need_gu) {
    OrcMesh &m = *s.mesh;
    if (s.settings.gradient_reconstruction == ORC_GRAD_LEAST_SQUARES) {
        hipLaunchKernelGGL(grad_p_lsq_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), s.p.p, s.gp.p, s.dev_status.p);
        if (need_gu)
            hipLaunchKernelGGL(grad_u_lsq_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), s.u.p, s.v.p, s.w.p, s.gu.p, s.dev_status.p);
        ORC_HIP(hipGetLastError());
        return ORC_OK;
    }
    hipLaunchKernelGGL(grad_p_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), s.p.p, s.gp.p, s.settings.q1_compat, s.dev_status.p);
    if (need_gu)
        hipLaunchKernelGGL(grad_u_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), s.u.p, s.v.p, s.w.p, s.gu.p, s.dev_status.p);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int k_face_flux(SolverState &s, bool momentum_phase) {
    OrcMesh &m = *s.mesh;
    FaceArgs A{s.u.p, s.v.p, s.w.p, s.p.p, s.gp.p, s.du.p, s.dv.p, s.dw.p, s.settings.velocity_interpolation,
               s.settings.pressure_interpolation, s.settings.q1_compat, s.rho};
    const int g = grid_for(m.n_faces);
    if (momentum_phase)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(face_k<0>), dim3(g), dim3(kBlock), 0, ctx().stream, m.dev(), A, s.flux.p, s.pf.p, s.coef.p, s.dev_status.p);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(face_k<1>), dim3(g), dim3(kBlock), 0, ctx().stream, m.dev(), A, s.flux.p, s.pf.p, s.coef.p, s.dev_status.p);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int k_momentum(SolverState &s, double *peclet_host) {
    OrcMesh &m = *s.mesh;
    MomentumArgs A{s.u.p, s.v.p, s.w.p, s.gu.p, s.flux.p, s.pf.p, s.a_di.p, s.b_u_di.p, s.b_v_di.p, s.b_w_di.p,
                   s.a_u.p, s.a_v.p, s.a_w.p, s.b_u.p, s.b_v.p, s.b_w.p, s.du.p, s.dv.p, s.dw.p, s.settings.momentum,
                   s.settings.q1_compat, s.rho};
    const int g = grid_for(s.n);
    bool ref_pe = false;  // the mean Peclet number as the reference's running sum (discretization.rs:338), reference reduction order only
    if (s.settings.frozen_diagonals == 0) {
        // in-place diagonals (the reference's own mode): level by level over the cell order
        ORC_TRY(vec_copy(s.du_old.p, s.du.p, s.n));
        ORC_TRY(vec_copy(s.dv_old.p, s.dv.p, s.n));
        ORC_TRY(vec_copy(s.dw_old.p, s.dw.p, s.n));
        InplaceArgs I;
        I.du_old = s.du_old.p; I.dv_old = s.dv_old.p; I.dw_old = s.dw_old.p;
        I.pe = s.pe.p;
        I.F = FaceArgs{s.u.p, s.v.p, s.w.p, s.p.p, s.gp.p, s.du.p, s.dv.p, s.dw.p, s.settings.velocity_interpolation,
                       s.settings.pressure_interpolation, s.settings.q1_compat, s.rho};
        I.status = s.dev_status.p;
        for (size_t l = 0; l + 1 < s.level_ptr.size(); ++l) {
            I.cells = s.level_cells.p + s.level_ptr[l];
            I.count = s.level_ptr[l + 1] - s.level_ptr[l];
            if (I.count == 0) continue;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(momentum_k<true>), dim3(grid_for(I.count)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), A, s.partials.p, I);
        }
        hipLaunchKernelGGL(peclet_stats_k, dim3(g), dim3(kBlock), 0, ctx().stream, s.pe.p, s.n_own, s.n, s.partials.p);
        ref_pe = s.settings.reduction_order == ORC_REDUCTION_REFERENCE && !s.mesh->halo.active() && peclet_host != nullptr;
        if (ref_pe) hipLaunchKernelGGL(peclet_cell_term_k, dim3(g), dim3(kBlock), 0, ctx().stream, s.pe.p, s.n);  // pe[c] = its term of :338
    } else {
        // [r05] frozen diagonals in the reference's reduction order (a verification mode too: the frozen oracle bit for bit, report included)
        InplaceArgs I{};
        ref_pe = s.settings.reduction_order == ORC_REDUCTION_REFERENCE && !s.mesh->halo.active() && peclet_host != nullptr && s.pe.p != nullptr;
        I.pe = ref_pe ? s.pe.p : nullptr;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(momentum_k<false>), dim3(g), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), A, s.partials.p, I);
    }
    hipLaunchKernelGGL(reduce_minmax_k, dim3(1), dim3(64), 0, ctx().stream, s.partials.p, g, s.scal.p + 8);
    ORC_HIP(hipGetLastError());
    if (ref_pe) ORC_TRY(sum_reference(s.pe.p, s.n, s.scal.p + 8));
    if (peclet_host) {
        if (s.mesh->halo.active()) {  // statistics over the whole mesh: sum; max of (-min, max)
            hipLaunchKernelGGL(negate_k, dim3(1), dim3(1), 0, ctx().stream, s.scal.p + 9);
            ORC_TRY(comm_allreduce_sum(s.scal.p + 8, 1));
            ORC_TRY(comm_allreduce_max(s.scal.p + 9, 2));
            hipLaunchKernelGGL(negate_k, dim3(1), dim3(1), 0, ctx().stream, s.scal.p + 9);
        }
        ORC_HIP(hipMemcpyAsync(peclet_host, s.scal.p + 8, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
        peclet_host[0] /= (double)s.mesh->n_global;  // discretization.rs:355
    }
    return ORC_OK;
}

int k_pressure_correction(SolverState &s) {
    OrcMesh &m = *s.mesh;
    ORC_TRY(k_face_flux(s, false));
    hipLaunchKernelGGL(pressure_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), s.flux.p, s.coef.p, s.rho,
                       s.a_p.p, s.b_p.p);
    ORC_HIP(hipGetLastError());
    return ORC_OK;
}

int k_apply_correction(SolverState &s, double *sums_host) {
    OrcMesh &m = *s.mesh;
    const int g = grid_for(s.n);
    // [r04] The reference's own mode (in-place diagonals + nalgebra's reduction order, one GPU): the REPORT's sums are taken in the
    // reference's association too — p'.norm() through dotx (solver.rs:1226), the velocity-correction sum cell by cell (:1224), the
    // means as left-to-right folds (:206-208) — so that not only the fields but the six report doubles of an iteration are the
    // oracle's bit for bit (scripts/reference_mode_fullsize.py compares exactly those at 10.24 M cells).  s.pe is free by now.
    const bool ref_report = s.settings.reduction_order == ORC_REDUCTION_REFERENCE && !s.mesh->halo.active() && s.pe.p;
    hipLaunchKernelGGL(correction_k, dim3(g), dim3(kBlock), 0, ctx().stream, m.dev(), s.du.p, s.dv.p, s.dw.p, s.p_prime.p, s.u.p, s.v.p,
                       s.w.p, s.p.p, s.settings.pressure_relaxation, s.settings.momentum_relaxation, s.partials.p, s.dev_status.p,
                       ref_report ? s.pe.p : (double *)nullptr);
    ORC_HIP(hipGetLastError());
    if (ref_report) {
        ORC_TRY(dot_reference(s.p_prime.p, s.p_prime.p, s.n, s.scal.p + 0, nullptr));
        ORC_TRY(sum_reference(s.pe.p, s.n, s.scal.p + 1));
        ORC_TRY(sum_reference(s.u.p, s.n, s.scal.p + 2));
        ORC_TRY(sum_reference(s.v.p, s.n, s.scal.p + 3));
        ORC_TRY(sum_reference(s.w.p, s.n, s.scal.p + 4));
    } else {
        ORC_TRY(reduce_partials(s.partials.p, g, 5, s.scal.p, s.mesh->halo.active()));
    }
    if (sums_host) {
        ORC_HIP(hipMemcpyAsync(sums_host, s.scal.p, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
    }
    return ORC_OK;
}

// Streams of the concurrent solves: set-up streams (dependent rounds of tiny kernels, each waited for by the host) and solve streams (the
// bandwidth-bound products) get different priority classes — stream_create (runtime.cpp) decides, and knows when it must not.
// [r04] measurement: ORC_SETUP_CU_MASK (a 32-bit hex pattern repeated over the device's CU mask words) confined the set-up streams to a share
// of the CUs: all settings within +-1 %; removed in r05 with the other measured-and-dropped switches.
static int create_stream(hipStream_t *out, int role, int lane) {
    return stream_create(out, role, lane, role == kSolveStream ? "solve" : "set-up");
}

static int solve_field_on(SolverState &s, DevBuf<double> &a, DevBuf<double> &b, DevBuf<double> &x, int eq, Arena &arena, SolveStats &stats,
                          SolveSide *side, Arena *side_arena, const AmgHierarchy *prepared = nullptr) {
    MatView A;
    A.P = s.mesh->pat.dev();
    A.val = a.p;
    A.symmetric = s.mesh->pat.symmetric;
    A.halo = s.mesh->halo.active() ? &s.mesh->halo : nullptr;
    A.persistent_pattern = true;
    const OrcSettings &t = s.settings;
    if (arena.empty()) ORC_TRY(arena.reset());  // nothing of the previous solve is alive: a fragmented reservation is folded into one chunk
    if (side_arena && side_arena->empty()) ORC_TRY(side_arena->reset());
    CtxDefaultsScope restore_defaults(ctx());  // this solver's guard and reduction order for the solve only
    ctx().breakdown_guard = t.breakdown_guard != 0;
    ctx().reduction_order = t.reduction_order;
    stats.side = nullptr;
    // the momentum systems share their pairing's starting state (the caller has opened the exchange: SiblingPairing::begin)
    stats.sibling = (s.sibling_pairing && eq < 3) ? &s.sibling : nullptr;
    stats.sibling_role = eq == 0 ? 1 : 2;
    stats.hierarchy = prepared ? prepared : ((eq == 3 && s.p_hierarchy.n_levels > 0) ? &s.p_hierarchy : nullptr);
    if (side && s.two_stream_multigrid && t.solver_type == ORC_SOLVER_MULTIGRID) {
        if (!side->stream) {
            // the side stream and the stream of this solve wait for each other's events level by level (side_wait_setup / setup_wait_side): they
            // share a class — "a stream only waits for its own class or a higher one" (runtime.cpp, stream_create)
            const int role = stream_role(ctx().stream) == kSetupStream ? kSetupStream : kSolveStream;
            ORC_TRY(create_stream(&side->stream, role, eq < 3 ? eq : 1));
            ORC_HIP(hipEventCreateWithFlags(&side->ev_setup, hipEventDisableTiming));
            ORC_HIP(hipEventCreateWithFlags(&side->ev_solve, hipEventDisableTiming));
            side->arena = side_arena;
        }
        stats.side = side;
    }
    return iterative_solve_dev(A, b.p, x.p, t.iterations, t.solver_type, t.relaxation, t.relative_convergence_threshold,
                               t.preconditioner, arena, &stats);
}

static int solve_field(SolverState &s, DevBuf<double> &a, DevBuf<double> &b, DevBuf<double> &x, int eq) {
    return solve_field_on(s, a, b, x, eq, s.arena, s.stats, &s.side, &s.side_arena);
}

static void destroy_side(SolveSide &d) {
    if (d.stream) stream_destroy(d.stream);
    if (d.ev_setup) (void)hipEventDestroy(d.ev_setup);
    if (d.ev_solve) (void)hipEventDestroy(d.ev_solve);
    d = SolveSide();
}

SolverState::~SolverState() {
    for (auto &l : lanes) {
        if (l.stream) stream_destroy(l.stream);
        if (l.level0_done) (void)hipEventDestroy(l.level0_done);
        destroy_side(l.side);
    }
    destroy_side(side);
    if (prep_stream) stream_destroy(prep_stream);
}

// Runs in a host thread of its own beside the momentum solves: the p' matrix from the fresh momentum diagonals (the
// RHS it also writes is recomputed after the solves), then the Multigrid hierarchy for it.  No rank-to-rank traffic.
static int prepare_p_hierarchy(SolverState &s) {
    OrcMesh &m = *s.mesh;
    hipStream_t st = ctx().stream;
    hipLaunchKernelGGL(face_coef_k, dim3(grid_for(m.n_faces)), dim3(kBlock), 0, st, m.dev(), s.du.p, s.dv.p, s.dw.p, s.rho, s.coef.p);
    hipLaunchKernelGGL(pressure_k, dim3(grid_for(s.n)), dim3(kBlock), 0, st, m.dev(), m.pat.dev(), s.flux.p, s.coef.p, s.rho, s.a_p.p, s.b_p.p);
    ORC_HIP(hipGetLastError());
    MatView A;
    A.P = m.pat.dev();
    A.val = s.a_p.p;
    A.symmetric = m.pat.symmetric;
    A.halo = m.halo.active() ? &m.halo : nullptr;  // level 1 pairs owned rows only; nothing is exchanged
    A.persistent_pattern = true;
    ORC_TRY(s.hier_arena.reset());
    // transient set-up storage: its own arena; when the momentum set-ups are through (lock-step schedule: this thread starts
    // after them) lane 0's is free and big enough
    Arena &scratch = s.p_scratch_shared ? s.lanes[0].scratch_arena : s.hier_scratch;
    scratch.release(Arena::Mark{0, 0});
    ORC_TRY(scratch.reset());
    ORC_TRY(multigrid_prepare_dev(A, s.settings.preconditioner, s.hier_arena, s.p_hierarchy, nullptr, 0, &scratch));
    ORC_HIP(hipStreamSynchronize(st));
    return ORC_OK;
}

struct PrepareThread {
    std::thread th;
    Ctx local;
    int status = ORC_OK;
    bool running = false;
    void start(SolverState &s) {
        Ctx &g = ctx();
        local = g;
        local.stream = s.prep_stream;
        local.last_error.clear();
        auto work = [this, &s] {
            CtxScope scope(&local);
            if (hipSetDevice(local.device) != hipSuccess) { status = set_error(ORC_ERR_HIP, "hipSetDevice failed in the set-up thread"); return; }
            status = prepare_p_hierarchy(s);
        };
        try {
            th = std::thread(work);
            running = true;
        } catch (...) {  // no thread to be had: build the hierarchy right here
            work();
        }
    }
    int join() {
        if (running) {
            th.join();
            running = false;
        }
        if (status != ORC_OK) ctx().last_error = local.last_error;
        return status;
    }
    ~PrepareThread() { if (running) th.join(); }
};

static int join_prepare(PrepareThread *prep) { return prep->join(); }

// Partitioned runs (one rank of several): the same overlap with every RCCL call left where it was — on the library
// stream, issued by the calling thread, in an order that is the same on all ranks.  Per momentum system the Multigrid
// arm falls into (1) the Jacobi scaling, the level-0 smoothing solve and the residual, which exchange halos and
// all-reduce scalars, (2) the hierarchy set-up, which needs the matrix only, (3) the V-recursion from level 1 on, and
// (4) the status agreement.  (2) and (3) are rank-local: a lane (thread + streams) per system builds the hierarchy from
// the start and runs (3) as soon as the library stream has finished that system's (1); the calling thread does
// (1) for u, v, w in turn and (4) for u, v, w at the end.  Per system the operations and their order are those of the
// sequential path, so the fields are identical.
static int solve_momentum_partitioned(SolverState &s) {
    Ctx &g = ctx();
    const OrcSettings &t = s.settings;
    OrcMesh &m = *s.mesh;
    const int64_t n = m.n_own;
    DevBuf<double> *mats[3] = {&s.a_u, &s.a_v, &s.a_w}, *rhs[3] = {&s.b_u, &s.b_v, &s.b_w}, *sol[3] = {&s.u, &s.v, &s.w};
    ArenaScope arena_scope(s.arena);  // unwound on every exit (ORC_TRY / ORC_HIP return early)
    MatView plain[3], view[3];
    const double *b_used[3];
    double *r[3];
    int *dev_status;
    ORC_TRY(s.arena.alloc((size_t)4, &dev_status));
    ORC_HIP(hipMemsetAsync(dev_status, 0, 4 * sizeof(int), g.stream));
    for (int k = 0; k < 3; ++k) {
        MatView A;
        A.P = m.pat.dev();
        A.val = mats[k]->p;
        A.symmetric = m.pat.symmetric;
        A.halo = &m.halo;
        A.persistent_pattern = true;
        plain[k] = view[k] = A;
        b_used[k] = rhs[k]->p;
        ORC_TRY(s.arena.alloc((size_t)std::max<int64_t>(n, 1), &r[k]));
        if (t.preconditioner == ORC_PRECOND_JACOBI) {  // linear_algebra.rs:159-166, as iterative_solve_dev does it
            double *dinv, *bt;
            ORC_TRY(s.arena.alloc((size_t)std::max<int64_t>(n, 1), &dinv));
            ORC_TRY(s.arena.alloc((size_t)std::max<int64_t>(n, 1), &bt));
            ORC_TRY(diag_inverse_dev(A, dinv));
            ORC_TRY(scale_vec_dev(dinv, rhs[k]->p, bt, n));
            view[k].s1 = dinv;
            b_used[k] = bt;
        }
        if (!s.lanes[k].stream) ORC_TRY(create_stream(&s.lanes[k].stream, kSetupStream, k));
        if (!s.lanes[k].side.stream) {
            ORC_TRY(create_stream(&s.lanes[k].side.stream, kSolveStream, k));
            ORC_HIP(hipEventCreateWithFlags(&s.lanes[k].side.ev_setup, hipEventDisableTiming));
            ORC_HIP(hipEventCreateWithFlags(&s.lanes[k].side.ev_solve, hipEventDisableTiming));
            s.lanes[k].side.arena = &s.lanes[k].side_arena;
        }
        if (!s.lanes[k].level0_done) ORC_HIP(hipEventCreateWithFlags(&s.lanes[k].level0_done, hipEventDisableTiming));
    }
    ORC_HIP(hipStreamSynchronize(g.stream));  // the assembled systems and the scalings are complete
    g.breakdown_guard = t.breakdown_guard != 0;
    g.reduction_order = ORC_REDUCTION_TREE;  // partitioned operators always reduce as trees + all-reduce

    std::mutex mu;
    std::condition_variable cv;
    bool ready[3] = {false, false, false};
    bool abort_all = false;
    int st_lane[3] = {ORC_OK, ORC_OK, ORC_OK};
    Ctx local[3];
    std::thread th[3];
    for (int k = 0; k < 3; ++k) {
        local[k] = g;
        local[k].stream = s.lanes[k].stream;
        local[k].last_error.clear();
    }
    auto lane_work = [&](int k) {
        SolverState::Lane &L = s.lanes[k];
        CtxScope scope(&local[k]);
        if (hipSetDevice(local[k].device) != hipSuccess) { st_lane[k] = set_error(ORC_ERR_HIP, "hipSetDevice failed in a solve thread"); return; }
        L.arena.release(Arena::Mark{0, 0});
        L.scratch_arena.release(Arena::Mark{0, 0});
        (void)L.scratch_arena.reset();
        int st = multigrid_prepare_dev(plain[k], t.preconditioner, L.arena, L.hierarchy, nullptr, 0, &L.scratch_arena);  // (2)
        if (st == ORC_OK && hipStreamSynchronize(local[k].stream) != hipSuccess) st = set_error(ORC_ERR_HIP, "stream synchronisation failed in a solve thread");
        // test hook (tests/mp_worker.py, mode gpu_lane_error): ORC_DEBUG_INJECT_LANE_ERROR="rank:lane" makes that rank's lane fail
        // locally after its set-up — every rank must still leave the solve with the same verdict and nobody may hang
        if (!cfg().inject_lane_error.empty()) {
            int r_ = -1, k_ = -1;
            if (sscanf(cfg().inject_lane_error.c_str(), "%d:%d", &r_, &k_) == 2 && r_ == local[k].rank && k_ == k && st == ORC_OK)
                st = set_error(ORC_ERR_HIP, "injected lane error (rank %d, lane %d)", r_, k_);
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[k] || abort_all; });
            if (abort_all) { st_lane[k] = st; return; }
        }
        if (st == ORC_OK) {  // (3) on the lane's solve stream, behind the library stream's level-0 work for this system
            local[k].stream = L.side.stream;
            if (hipStreamWaitEvent(L.side.stream, L.level0_done, 0) != hipSuccess) st = set_error(ORC_ERR_HIP, "hipStreamWaitEvent failed");
            L.stats = SolveStats();
            L.stats.hierarchy = &L.hierarchy;
            L.side_arena.release(Arena::Mark{0, 0});
            if (st == ORC_OK)
                st = multigrid_coarse_part_dev(view[k], r[k], sol[k]->p, t.iterations, t.relaxation, t.relative_convergence_threshold, t.preconditioner,
                                               L.side_arena, &L.stats, dev_status + k);
            if (hipStreamSynchronize(L.side.stream) != hipSuccess && st == ORC_OK) st = set_error(ORC_ERR_HIP, "stream synchronisation failed in a solve thread");
        }
        st_lane[k] = st;
    };
    for (int k = 0; k < 3; ++k) {
        try {
            th[k] = std::thread(lane_work, k);
        } catch (...) {
            // without a thread the lane's work is done further down, after the level-0 section of its system
        }
    }
    // (1) on the library stream, one system after the other
    int st_main = ORC_OK;
    for (int k = 0; k < 3 && st_main == ORC_OK; ++k) {
        s.stats.side = nullptr; s.stats.hierarchy = nullptr;
        st_main = iterative_solve_dev(view[k], b_used[k], sol[k]->p, t.iterations, ORC_SOLVER_BICGSTAB, t.relaxation, t.relative_convergence_threshold,
                                      t.preconditioner, s.arena, &s.stats);  // :273-282 (nested scaling, Q4)
        if (st_main == ORC_OK) st_main = residual_dev(view[k], b_used[k], sol[k]->p, r[k]);  // :283
        if (st_main == ORC_OK && hipEventRecord(s.lanes[k].level0_done, g.stream) != hipSuccess) st_main = set_error(ORC_ERR_HIP, "hipEventRecord failed");
        {
            std::lock_guard<std::mutex> lk(mu);
            if (st_main == ORC_OK) ready[k] = true;
            else abort_all = true;
        }
        cv.notify_all();
        if (st_main == ORC_OK && !th[k].joinable()) lane_work(k);  // no thread was to be had for this lane
    }
    if (st_main != ORC_OK) {
        { std::lock_guard<std::mutex> lk(mu); abort_all = true; }
        cv.notify_all();
    }
    for (int k = 0; k < 3; ++k)
        if (th[k].joinable()) th[k].join();
    // (4) the verdicts, agreed between the ranks in u, v, w order (a rank whose lane failed locally still takes part)
    // iterative_solve_dev has already agreed st_main between the ranks (its own status agreement), so either every rank
    // is here with st_main == ORC_OK or none is; a local HIP failure below still enters the three agreements.
    int result = st_main;
    if (st_main == ORC_OK) {
        int h[4] = {0, 0, 0, 0};
        int copy_st = ORC_OK;
        if (hipMemcpyAsync(h, dev_status, sizeof(h), hipMemcpyDeviceToHost, g.stream) != hipSuccess || hipStreamSynchronize(g.stream) != hipSuccess)
            copy_st = set_error(ORC_ERR_HIP, "fetching the Multigrid status words failed");
        for (int k = 0; k < 3; ++k) {
            int stk = st_lane[k] != ORC_OK ? st_lane[k] : (copy_st != ORC_OK ? copy_st : h[k]);
            if (st_lane[k] != ORC_OK) g.last_error = local[k].last_error;
            stk = comm_global_status(stk);
            if (result == ORC_OK && stk != ORC_OK) result = stk;
        }
    }
    s.stats = s.lanes[0].stats;
    return result;
}

// The three momentum solves of one iteration on three streams, one host thread each (the set-up phases synchronise
// their stream every few rounds).  Returns the first non-zero status in u, v, w order, like the sequential loop.
struct PrepareThread;
static int join_prepare(PrepareThread *prep);

// The three momentum solves of one iteration on three streams, one host thread each (the set-up phases synchronise
// their stream every few rounds).  Returns the first non-zero status in u, v, w order, like the sequential loop.
// setup_first (Multigrid arm): phase 1 builds the three hierarchies side by side (beside the p' hierarchy of `prep`) with
// no product running — the set-up rounds are chains of tiny dependent kernels, and behind other streams' 2048-workgroup
// products every one of them waits for a free wave slot (tail kernels: 0.48 s of kernel time per iteration alone, 1.5 s
// summed when stretched by that contention) — phase 2 runs the three solves on the prepared hierarchies.  Same kernels on
// the same data in the same order per system: same bits.
static int solve_momentum_concurrently(SolverState &s, bool setup_first, PrepareThread *prep) {
    Ctx &g = ctx();
    ORC_HIP(hipStreamSynchronize(g.stream));  // the assembled systems are complete
    DevBuf<double> *mats[3] = {&s.a_u, &s.a_v, &s.a_w}, *rhs[3] = {&s.b_u, &s.b_v, &s.b_w}, *sol[3] = {&s.u, &s.v, &s.w};
    int st[3] = {ORC_OK, ORC_OK, ORC_OK};
    Ctx local[3];
    for (int k = 0; k < 3; ++k) {
        if (!s.lanes[k].stream) ORC_TRY(create_stream(&s.lanes[k].stream, kSetupStream, k));
        local[k] = g;
        local[k].stream = s.lanes[k].stream;
        local[k].last_error.clear();
    }
    auto run_lanes = [&](auto &&work) {
        std::thread th[3];
        for (int k = 0; k < 3; ++k) {
            try {
                th[k] = std::thread(work, k);
            } catch (...) {  // no thread to be had: this lane runs here, after the ones already started (nothing crosses the C ABI)
                work(k);
            }
        }
        for (int k = 0; k < 3; ++k)
            if (th[k].joinable()) th[k].join();
    };
    if (setup_first) {
        auto prepare = [&](int k) {
            SolverState::Lane &L = s.lanes[k];
            CtxScope scope(&local[k]);
            if (hipSetDevice(local[k].device) != hipSuccess) { st[k] = set_error(ORC_ERR_HIP, "hipSetDevice failed in a set-up thread"); return; }
            MatView A;
            A.P = s.mesh->pat.dev();
            A.val = mats[k]->p;
            A.symmetric = s.mesh->pat.symmetric;
            A.persistent_pattern = true;
            L.arena.release(Arena::Mark{0, 0});
            L.scratch_arena.release(Arena::Mark{0, 0});
            (void)L.scratch_arena.reset();
            st[k] = multigrid_prepare_dev(A, s.settings.preconditioner, L.arena, L.hierarchy, nullptr, 0, &L.scratch_arena);
            if (hipStreamSynchronize(local[k].stream) != hipSuccess && st[k] == ORC_OK) st[k] = set_error(ORC_ERR_HIP, "stream synchronisation failed in a set-up thread");
        };
        run_lanes(prepare);
        const int pst = prep ? join_prepare(prep) : ORC_OK;  // the p' hierarchy was being built beside them
        for (int k = 0; k < 3; ++k)
            if (st[k] != ORC_OK) { g.last_error = local[k].last_error; return st[k]; }
        if (pst != ORC_OK) return pst;
    }
    auto work = [&](int k) {
        SolverState::Lane &L = s.lanes[k];
        CtxScope scope(&local[k]);
        if (hipSetDevice(local[k].device) != hipSuccess) { st[k] = set_error(ORC_ERR_HIP, "hipSetDevice failed in a solve thread"); return; }
        if (setup_first) {
            L.side_arena.release(Arena::Mark{0, 0});
            st[k] = solve_field_on(s, *mats[k], *rhs[k], *sol[k], k, L.side_arena, L.stats, nullptr, nullptr, &L.hierarchy);
        } else {
            st[k] = solve_field_on(s, *mats[k], *rhs[k], *sol[k], k, L.arena, L.stats, &L.side, &L.side_arena);
        }
        if (k == 0) s.sibling.finish();  // whatever happened to u: v and w must not wait for a level that will not come
        if (hipStreamSynchronize(local[k].stream) != hipSuccess && st[k] == ORC_OK) st[k] = set_error(ORC_ERR_HIP, "stream synchronisation failed in a solve thread");
    };
    s.sibling.begin(!setup_first);  // with the hierarchies prepared ahead nobody aggregates inside the solves
    run_lanes(work);
    s.stats = s.lanes[0].stats;
    for (int k = 0; k < 3; ++k)
        if (st[k] != ORC_OK) {
            g.last_error = local[k].last_error;
            return st[k];
        }
    return ORC_OK;
}

// The three momentum systems in lock-step (linalg.hpp MatView3 / multigrid_arm3_dev, bicgstab3_dev): single GPU, tree
// reductions, Multigrid or BiCGSTAB solver.  Returns the first non-zero status in u, v, w order, like the sequential loop.
static int solve_momentum_triple(SolverState &s, const std::function<void()> &on_hierarchies_built) {
    Ctx &g = ctx();
    const OrcSettings &t = s.settings;
    g.breakdown_guard = t.breakdown_guard != 0;
    g.reduction_order = t.reduction_order;
    MatView3 A3;
    A3.P = s.mesh->pat.dev();
    A3.val[0] = s.a_u.p; A3.val[1] = s.a_v.p; A3.val[2] = s.a_w.p;
    A3.mesh_pattern = true;
    if (s.mesh->halo.active()) {  // [r04] cell-partitioned mesh: one halo exchange and one all-reduce per step for the three systems
        A3.halo = &s.mesh->halo;
        g.reduction_order = ORC_REDUCTION_TREE;  // partitioned operators always reduce as trees + all-reduce
    }
    const double *b[3] = {s.b_u.p, s.b_v.p, s.b_w.p};
    double *x[3] = {s.u.p, s.v.p, s.w.p};
    if (s.arena.empty()) ORC_TRY(s.arena.reset());
    if (t.solver_type == ORC_SOLVER_BICGSTAB_GS_PRECOND) {  // [r04] BASELINE configs[2]: u, v, w per colour in ONE launch (gs.hip, slot space)
        ArenaScope scope(s.arena);
        MatView V[3];
        const double *bb[3];
        const size_t nn = (size_t)std::max<int64_t>(s.n_own, 1);
        for (int k = 0; k < 3; ++k) {
            V[k].P = s.mesh->pat.dev();
            V[k].val = A3.val[k];
            V[k].symmetric = s.mesh->pat.symmetric;
            V[k].persistent_pattern = true;
            bb[k] = b[k];
            if (t.preconditioner == ORC_PRECOND_JACOBI) {  // linear_algebra.rs:159-166, as iterative_solve_dev does it per system
                double *dinv, *bt;
                ORC_TRY(s.arena.alloc(nn, &dinv));
                ORC_TRY(s.arena.alloc(nn, &bt));
                ORC_TRY(diag_inverse_dev(V[k], dinv));
                ORC_TRY(scale_vec_dev(dinv, b[k], bt, s.n_own));
                V[k].s1 = dinv;
                bb[k] = bt;
            } else if (t.preconditioner != ORC_PRECOND_NONE) {
                return set_error(ORC_ERR_BAD_ARGUMENT, "unknown preconditioner %d", t.preconditioner);
            }
        }
        ORC_TRY(gs_bicgstab3_dev(V, bb, x, t.iterations, s.arena));
        ORC_HIP(hipStreamSynchronize(g.stream));
        return ORC_OK;
    }
    if (t.solver_type == ORC_SOLVER_BICGSTAB) {
        ArenaScope scope(s.arena);
        const size_t n3 = (size_t)3 * (size_t)s.n_own;
        double *b3, *x3;
        ORC_TRY(s.arena.alloc(n3, &b3));
        ORC_TRY(s.arena.alloc((size_t)3 * (size_t)s.n, &x3));  // with its ghost entries on a partitioned mesh
        ORC_TRY(interleave3_dev(b[0], b[1], b[2], b3, s.n_own));
        ORC_TRY(interleave3_dev(x[0], x[1], x[2], x3, s.n_own));
        ORC_TRY(bicgstab3_dev(A3, b3, x3, t.iterations, t.preconditioner, s.arena));
        ORC_TRY(deinterleave3_dev(x3, x[0], x[1], x[2], s.n_own));
        ORC_HIP(hipStreamSynchronize(g.stream));
        return ORC_OK;
    }
    for (int k = 0; k < 3; ++k) {
        SolverState::Lane &L = s.lanes[k];
        if (!L.stream) ORC_TRY(create_stream(&L.stream, kSetupStream, k));
        if (!L.side.stream) {
            ORC_TRY(create_stream(&L.side.stream, kSolveStream, k));
            ORC_HIP(hipEventCreateWithFlags(&L.side.ev_setup, hipEventDisableTiming));
            ORC_HIP(hipEventCreateWithFlags(&L.side.ev_solve, hipEventDisableTiming));
            L.side.arena = &L.side_arena;
        }
        s.triple[k].setup_stream = L.stream;
        s.triple[k].solve_stream = L.side.stream;
        s.triple[k].hier_arena = &L.arena;
        s.triple[k].vec_arena = &L.side_arena;
        s.triple[k].scratch_arena = &L.scratch_arena;
        s.triple[k].symmetric = s.mesh->pat.symmetric;
    }
    int st3[3] = {ORC_OK, ORC_OK, ORC_OK};
    ORC_TRY(multigrid_arm3_dev(A3, b, x, t.iterations, t.relaxation, t.relative_convergence_threshold, t.preconditioner, s.arena, s.triple,
                               s.sibling_pairing ? &s.sibling : nullptr, st3, on_hierarchies_built));
    s.stats = s.triple[0].stats;
    for (int k = 0; k < 3; ++k)
        if (st3[k] != ORC_OK) return st3[k];
    return ORC_OK;
}

static void debug_field(SolverState &s, const char *name, const DevBuf<double> &f) {
    std::vector<double> h((size_t)s.n);
    (void)f.download(h.data(), (size_t)s.n);
    int nn = 0;
    double mx = 0.;
    for (double x : h) { if (std::isnan(x)) nn++; else mx = std::max(mx, std::fabs(x)); }
    fprintf(stderr, "[orc debug] it %llu %s: nan=%d max=%.17g\n", (unsigned long long)s.iterations_done, name, nn, mx);
}

// One pass of solver.rs:60-222 per iteration.  In a partitioned run (mesh.halo active) the ghost entries of every
// field a face kernel reads are refreshed first (C1); the solves exchange their own work vectors.
int solver_iterate(SolverState &s, uint64_t iterations, double *report) {
    // the solves below run with THIS solver's guard and reduction order; the process-wide defaults (orc_set_breakdown_guard,
    // orc_set_reduction_order: what orc_iterative_solve uses) are put back on every exit
    CtxDefaultsScope restore_defaults(ctx());
    const bool tvd = is_tvd(s.settings.momentum);
    const bool dbg = cfg().debug_nan;
    HaloPlan &H = s.mesh->halo;
    const int64_t n = s.n;
    for (uint64_t it = 0; it < iterations; ++it) {
        double peclet[3] = {0., 0., 0.};
        if (s.arena.empty()) ORC_TRY(s.arena.reset());  // between iterations nothing in the solver's own arena is alive
        ORC_TRACE("iteration %llu: start", (unsigned long long)s.iterations_done);
        if (H.active()) { double *f[4] = {s.u.p, s.v.p, s.w.p, s.p.p}; ORC_TRY(H.exchange(f, 4)); }
        ORC_TRY(k_gradients(s, tvd));
        if (H.active()) { double *g3[3] = {s.gp.p, s.gp.p + n, s.gp.p + 2 * n}; ORC_TRY(H.exchange(g3, 3)); }
        ORC_TRY(k_face_flux(s, true));
        ORC_TRY(k_momentum(s, report ? peclet : nullptr));       // :61-82
        if (H.active()) { double *d3[3] = {s.du.p, s.dv.p, s.dw.p}; ORC_TRY(H.exchange(d3, 3)); }
        if (dbg) { debug_field(s, "b_u", s.b_u); debug_field(s, "b_v", s.b_v); debug_field(s, "b_w", s.b_w); }
        const int method = s.settings.solver_type;
        PrepareThread prep;
        s.p_hierarchy.n_levels = 0;
        const bool early_p = s.early_p_hierarchy && (method == ORC_SOLVER_MULTIGRID || method == ORC_SOLVER_MULTIGRID_GS) && !dbg && !ctx().profile;
        if (early_p && !s.prep_stream) ORC_TRY(create_stream(&s.prep_stream, kSetupStream, 2));
        const bool lanes_ok = s.concurrent_momentum && !H.active() && !dbg && !ctx().profile &&
                              (method == ORC_SOLVER_MULTIGRID || method == ORC_SOLVER_BICGSTAB || method == ORC_SOLVER_JACOBI ||
                               method == ORC_SOLVER_MULTIGRID_GS || method == ORC_SOLVER_BICGSTAB_GS_PRECOND || method == ORC_SOLVER_MULTICOLOR_GS);
        const bool lanes_partitioned = s.concurrent_momentum && H.active() && !dbg && !ctx().profile && method == ORC_SOLVER_MULTIGRID;
        // [r04] the lock-step schedule also on a partitioned mesh (tree reductions there by construction): N > 1 runs the schedule of the
        // N = 1 headline, with a third of the momentum phase's halo exchanges and all-reduces (ORC_TRIPLE_MOMENTUM=0: one system per solve)
        // (the schedule — hence the SEQUENCE OF COLLECTIVES of a rank — is decided from this solver's settings and the mesh alone, never from
        // the thread context's reduction order, which is whatever the last solve left there: ADVICE r04.  A partitioned mesh reduces as
        // trees whatever the settings say; solve_momentum_triple puts that into the context itself.)
        const bool triple_part = s.concurrent_momentum && H.active() && !dbg && !ctx().profile && s.triple_momentum &&
                                 (method == ORC_SOLVER_MULTIGRID || method == ORC_SOLVER_BICGSTAB);
        const bool triple_ok = (lanes_ok && s.triple_momentum && s.settings.reduction_order != ORC_REDUCTION_REFERENCE &&
                                (method == ORC_SOLVER_MULTIGRID || method == ORC_SOLVER_BICGSTAB ||
                                 (method == ORC_SOLVER_BICGSTAB_GS_PRECOND && gs_slot_space_enabled()))) || triple_part;
        // The p' hierarchy is needed after the momentum solves.  Beside the per-system lanes it is built from the start; in the
        // lock-step schedule the momentum set-ups are the critical path of the first phase (nothing bandwidth-bound but the
        // level-0 solve runs beside them), so it starts when they are through and runs beside the bandwidth-bound coarse levels.
        const bool p_late = early_p && triple_ok && method == ORC_SOLVER_MULTIGRID;
        s.p_scratch_shared = p_late;
        if (early_p && !p_late) {
            ORC_HIP(hipStreamSynchronize(ctx().stream));  // the diagonals (and their ghosts) are in place
            prep.start(s);
        }
        if (triple_ok) {
            // :99-136, the three systems in lock-step on their shared pattern
            ORC_TRACE("momentum: lock-step solve");
            int st3 = solve_momentum_triple(s, p_late ? std::function<void()>([&] { prep.start(s); }) : std::function<void()>());
            ORC_TRACE("momentum: lock-step solve returned %d", st3);
            // partitioned: a rank whose set-up thread or coarse level failed locally has still taken part in every level-0 collective;
            // the verdict must be the same on every rank or they part ways at the next one
            if (H.active()) st3 = comm_global_status(st3);
            ORC_TRY(st3);
        } else if (lanes_ok) {
            ORC_TRY(solve_momentum_concurrently(s, false, prep.running ? &prep : nullptr));  // :99-136, the three systems side by side
        } else if (lanes_partitioned) {
            ORC_TRY(solve_momentum_partitioned(s));                     // the same with every RCCL call on the library stream
        } else {
            s.sibling.begin(true);
            const int st_u = solve_field(s, s.a_u, s.b_u, s.u, 0);      // :99-110
            s.sibling.finish();
            ORC_TRY(st_u);
            if (dbg) debug_field(s, "u", s.u);
            ORC_TRY(solve_field(s, s.a_v, s.b_v, s.v, 1));              // :112-123
            if (dbg) debug_field(s, "v", s.v);
            ORC_TRY(solve_field(s, s.a_w, s.b_w, s.w, 2));              // :125-136
            if (dbg) debug_field(s, "w", s.w);
        }
        ORC_TRACE("momentum done; joining the p' set-up");
        ORC_TRY(prep.join());
        ORC_TRACE("p' set-up joined");
        if (H.active()) { double *f[3] = {s.u.p, s.v.p, s.w.p}; ORC_TRY(H.exchange(f, 3)); }
        ORC_TRY(k_pressure_correction(s));                       // :137-148 (the matrix comes out as in the early pass)
        ORC_TRY(vec_fill(s.p_prime.p, 0., s.n));                 // :167
        if (dbg) debug_field(s, "b_p", s.b_p);
        ORC_TRACE("p' solve");
        ORC_TRY(solve_field(s, s.a_p, s.b_p, s.p_prime, 3));        // :168-179
        ORC_TRACE("p' solve done");
        if (dbg) debug_field(s, "p_prime", s.p_prime);
        if (H.active()) ORC_TRY(H.exchange(s.p_prime.p));
        double sums[5];
        ORC_TRY(k_apply_correction(s, sums));                    // :193-208
        s.iterations_done++;
        const double ng = (double)s.mesh->n_global;
        const double u_avg = sums[2] / ng, v_avg = sums[3] / ng, w_avg = sums[4] / ng;
        if (report) {
            double *r = report + 8 * it;
            r[0] = u_avg; r[1] = v_avg; r[2] = w_avg; r[3] = peclet[0]; r[4] = peclet[1]; r[5] = peclet[2];
            r[6] = std::sqrt(sums[1]); r[7] = std::sqrt(sums[0]);
        }
        int st = 0;
        ORC_HIP(hipMemcpyAsync(&st, s.dev_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
        ORC_HIP(hipStreamSynchronize(ctx().stream));
        if (H.active()) st = comm_global_status(st);
        if (st) return st;
        if (std::isnan(u_avg) || std::isnan(v_avg) || std::isnan(w_avg)) return ORC_ERR_SOLUTION_DIVERGED;  // :217-221
    }
    return ORC_OK;
}

// ------------------------------------------------------------------ solver::initialize_* (solver.rs:246-509, 710-772)
static MatView mesh_view(SolverState &s, double *values) {
    MatView A;
    A.P = s.mesh->pat.dev();
    A.val = values;
    A.symmetric = s.mesh->pat.symmetric;
    A.halo = s.mesh->halo.active() ? &s.mesh->halo : nullptr;
    A.persistent_pattern = true;
    return A;
}

// check_boundary_conditions (solver.rs:710-772).  Its angle tolerance is 5*180/pi radians (:713), which no angle
// exceeds, so the two "tangent" panics cannot fire; what remains is the count of velocity- and pressure-type zones.
// A moving wall adds one count per mesh face (:722-723, a u16 that a release build wraps; not reproduced).
// Returns 0 PressureOnly, 1 VelocityOnly, 2 Hybrid, or ORC_ERR_NO_BOUNDARY_CONDITIONS as a negative number.
int check_boundary_conditions(const OrcMesh &m) {
    std::vector<int32_t> zt((size_t)m.n_zones);
    std::vector<double> zv((size_t)3 * m.n_zones);
    if (m.ztype.download(zt.data(), zt.size()) != ORC_OK || m.zvec.download(zv.data(), zv.size()) != ORC_OK) return -ORC_ERR_HIP;
    uint64_t pressure = 0, velocity = 0;
    for (int z = 0; z < m.n_zones; ++z) {
        const double *v = &zv[(size_t)3 * z];
        if (zt[z] == ORC_BC_WALL) {
            if (std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 0.) velocity += (uint64_t)m.n_faces;
        } else if (zt[z] == ORC_BC_VELOCITY_INLET) {
            velocity += 1;
        } else if (zt[z] == ORC_BC_PRESSURE_INLET || zt[z] == ORC_BC_PRESSURE_OUTLET) {
            pressure += 1;
        }
    }
    if (velocity > 0) return pressure > 1 ? 2 : 1;
    if (pressure > 0) return 0;
    return -ORC_ERR_NO_BOUNDARY_CONDITIONS;  // "You must set boundary conditions."
}

// initialize_pressure_field (solver.rs:414-509): Laplace system in s.a_p / s.b_p, 10 Jacobi sweeps (omega 0.1,
// Jacobi-preconditioned) on s.p.
int initialize_pressure_field_dev(SolverState &s) {
    OrcMesh &m = *s.mesh;
    ORC_TRY(s.a_p.zero());
    hipLaunchKernelGGL(laplace_p_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), s.a_p.p, s.b_p.p, s.dev_status.p);
    ORC_HIP(hipGetLastError());
    ORC_TRY(iterative_solve_dev(mesh_view(s, s.a_p.p), s.b_p.p, s.p.p, 10, ORC_SOLVER_JACOBI, 0.1, 1e-6, ORC_PRECOND_JACOBI, s.arena, &s.stats));
    return fetch_status(s);
}

// initialize_velocity_field (solver.rs:511-696): potential psi from ten Jacobi-preconditioned BiCGSTAB iterations on the
// psi system (assembled in s.a_p / s.b_p, solved into s.p_prime), then u, v, w = least-squares gradient of psi.
int initialize_velocity_field_dev(SolverState &s) {
    OrcMesh &m = *s.mesh;
    ORC_TRY(s.a_p.zero());
    hipLaunchKernelGGL(psi_system_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), m.pat.dev(), s.a_p.p, s.b_p.p, s.dev_status.p);
    ORC_HIP(hipGetLastError());
    ORC_TRY(vec_fill(s.p_prime.p, 0., s.n));
    CtxDefaultsScope restore_defaults(ctx());
    ctx().breakdown_guard = s.settings.breakdown_guard != 0;
    ctx().reduction_order = s.settings.reduction_order;
    ORC_TRY(iterative_solve_dev(mesh_view(s, s.a_p.p), s.b_p.p, s.p_prime.p, 10, ORC_SOLVER_BICGSTAB, 0.1, 1e-6, ORC_PRECOND_JACOBI, s.arena, &s.stats));  // :595-604
    if (m.halo.active()) ORC_TRY(m.halo.exchange(s.p_prime.p));
    hipLaunchKernelGGL(psi_velocity_k, dim3(grid_for(s.n)), dim3(kBlock), 0, ctx().stream, m.dev(), s.p_prime.p, s.u.p, s.v.p, s.w.p);
    ORC_HIP(hipGetLastError());
    return fetch_status(s);
}

// the gradient pass solve_steady runs after its loop (solver.rs:227-242): the accumulated values are never used, but an
// unsupported scheme or a singular least-squares matrix still panics there
int post_loop_gradients_dev(SolverState &s) {
    HaloPlan &H = s.mesh->halo;
    if (H.active()) { double *f[4] = {s.u.p, s.v.p, s.w.p, s.p.p}; ORC_TRY(H.exchange(f, 4)); }
    ORC_TRY(k_gradients(s, true));
    int st = fetch_status(s);
    if (H.active()) st = comm_global_status(st);
    return st;
}

// initialize_flow (solver.rs:246-352).  `s` must have been set up with UD / LinearWeighted / LinearWeighted
// (:301-303) and zero fields; the blended matrix lives in s.a_p, which the pressure initialisation no longer needs.
int initialize_flow_dev(SolverState &s, uint64_t iteration_count) {
    const int kind = check_boundary_conditions(*s.mesh);  // :272 (result unused there, panics kept)
    if (kind < 0) return set_error(-kind, "You must set boundary conditions.");
    HaloPlan &H = s.mesh->halo;
    const int64_t n = s.n;
    ORC_TRY(initialize_pressure_field_dev(s));  // :287
    if (H.active()) { double *f[4] = {s.u.p, s.v.p, s.w.p, s.p.p}; ORC_TRY(H.exchange(f, 4)); }
    ORC_TRY(k_gradients(s, false));
    if (H.active()) { double *g3[3] = {s.gp.p, s.gp.p + n, s.gp.p + 2 * n}; ORC_TRY(H.exchange(g3, 3)); }
    ORC_TRY(k_face_flux(s, true));
    ORC_TRY(k_momentum(s, nullptr));  // :288-309 (b += b_di inside)
    CtxDefaultsScope restore_defaults(ctx());
    ctx().breakdown_guard = s.settings.breakdown_guard != 0;
    ctx().reduction_order = s.settings.reduction_order;
    const int64_t len = std::max<int64_t>(s.mesh->pat.padded, 1);
    double diffusion_fraction = 1.;
    while (diffusion_fraction >= 0.) {  // :316-349
        DevBuf<double> *mats[3] = {&s.a_u, &s.a_v, &s.a_w}, *rhs[3] = {&s.b_u, &s.b_v, &s.b_w}, *x[3] = {&s.u, &s.v, &s.w};
        for (int k = 0; k < 3; ++k) {
            hipLaunchKernelGGL(blend_k, dim3(grid_for(len)), dim3(kBlock), 0, ctx().stream, len, mats[k]->p, s.a_di.p, 1. - diffusion_fraction,
                               diffusion_fraction, s.a_p.p);
            ORC_HIP(hipGetLastError());
            ORC_TRY(iterative_solve_dev(mesh_view(s, s.a_p.p), rhs[k]->p, x[k]->p, iteration_count, ORC_SOLVER_BICGSTAB, 0.5, 1e-6,
                                        ORC_PRECOND_JACOBI, s.arena, &s.stats));
        }
        diffusion_fraction -= 0.2;
    }
    return fetch_status(s);
}

}  // namespace orc
