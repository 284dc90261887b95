/* vec3.h — restatement of numerical_types::Vector / Tensor (src/lib.rs:216-648).
 * Test infrastructure (see oracle.h). Operation order follows the Rust operator impls so the
 * rounding path is the reference's. Compile with -ffp-contract=off (rustc never fuses).
 */
#ifndef ORC_ORACLE_VEC3_H
#define ORC_ORACLE_VEC3_H
#include <math.h>
#include <string.h>
#include "oracle.h"

static inline Vec3 v3(double x, double y, double z) { Vec3 r = {x, y, z}; return r; }
static inline Vec3 v_zero(void) { return v3(0., 0., 0.); }           /* lib.rs:224-230 */
static inline Vec3 v_ones(void) { return v3(1., 1., 1.); }           /* lib.rs:232-238 */
static inline double v_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* lib.rs:240-242 */
static inline Vec3 v_cross(Vec3 a, Vec3 b) {                          /* lib.rs:254-260 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double v_norm(Vec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } /* lib.rs:262-264 (powi(2) == x*x) */
static inline Vec3 v_unit(Vec3 a) { double l = v_norm(a); return v3(a.x / l, a.y / l, a.z / l); } /* lib.rs:266-273 */
static inline Vec3 v_reciprocal(Vec3 a) {                             /* lib.rs:246-252 */
    return v3(a.x != 0. ? 1. / a.x : 0., a.y != 0. ? 1. / a.y : 0., a.z != 0. ? 1. / a.z : 0.);
}
static inline Vec3 v_add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* lib.rs:368-378 */
static inline Vec3 v_adds(Vec3 a, double s) { return v3(a.x + s, a.y + s, a.z + s); }       /* lib.rs:356-366 */
static inline Vec3 v_sub(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* lib.rs:402-412 */
static inline Vec3 v_subs(Vec3 a, double s) { return v3(a.x - s, a.y - s, a.z - s); }       /* lib.rs:390-400 */
static inline Vec3 v_neg(Vec3 a) { return v3(-a.x, -a.y, -a.z); }                           /* lib.rs:529-538 */
static inline Vec3 v_muls(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }       /* Vector * Float, lib.rs:479-492 (correct) */
static inline Vec3 v_divs(Vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }       /* lib.rs:429-447 */
static inline Vec3 v_div(Vec3 a, Vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }    /* lib.rs:450-459 element-wise */
static inline Vec3 v_abs(Vec3 a) { return v3(fabs(a.x), fabs(a.y), fabs(a.z)); }
/* Float * Vector, lib.rs:540-548: the reference sets z: rhs.y * self.
 * q1 != 0 reproduces it; q1 == 0 is the mathematically intended product. */
static inline Vec3 s_mulv(double s, Vec3 a, int q1) { return v3(a.x * s, a.y * s, (q1 ? a.y : a.z) * s); }

static inline Tensor3 t_zero(void) { Tensor3 t; t.x = v_zero(); t.y = v_zero(); t.z = v_zero(); return t; }
static inline Tensor3 v_outer(Vec3 a, Vec3 b) {                       /* lib.rs:275-293 */
    Tensor3 t;
    t.x = v3(a.x * b.x, a.x * b.y, a.x * b.z);
    t.y = v3(a.y * b.x, a.y * b.y, a.y * b.z);
    t.z = v3(a.z * b.x, a.z * b.y, a.z * b.z);
    return t;
}
static inline Tensor3 t_add(Tensor3 a, Tensor3 b) { Tensor3 t; t.x = v_add(a.x, b.x); t.y = v_add(a.y, b.y); t.z = v_add(a.z, b.z); return t; } /* lib.rs:608-617 */
static inline Vec3 t_inner(Tensor3 t, Vec3 v) { return v3(v_dot(t.x, v), v_dot(t.y, v), v_dot(t.z, v)); } /* lib.rs:584-590 */

/* f64::total_cmp (used at discretization.rs:336-337, linear_algebra.rs:205) */
static inline int f64_total_cmp(double a, double b) {
    int64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    x ^= (int64_t)(((uint64_t)(x >> 63)) >> 1);
    y ^= (int64_t)(((uint64_t)(y >> 63)) >> 1);
    return (x > y) - (x < y);
}
#endif
