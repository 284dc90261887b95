/* sanitize_main.c — AddressSanitizer / UBSan driver for the CPU oracle (test infrastructure, see oracle.h).
 * GPU sanitizers are not available on the pool, so the memory-safety check runs on the restatement:
 * read a reference mesh, set the boundary conditions of tests.rs:60-76, initialise the flow, run a few SIMPLE
 * iterations with every solver arm, free everything.  Built and run by tests/test_oracle_sanitizers.py:
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all sanitize_main.c <oracle sources> -lm
 * usage: sanitize_main mesh.msh wall_zone_a [wall_zone_b] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int st__ = (call);                                                 \
        if (st__ != 0) { fprintf(stderr, "%s -> %d\n", #call, st__); return 2; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s mesh.msh wall_zone...\n", argv[0]); return 64; }
    OrMesh *m = or_read_mesh(argv[1]);
    if (!m) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    for (int a = 2; a < argc; ++a) CHECK(or_mesh_set_zone(m, argv[a], ORC_BC_WALL, 0., a == 2 ? 5e-4 : 0., 0., 0.));
    CHECK(or_mesh_set_zone(m, "INLET", ORC_BC_PRESSURE_INLET, -0.01, 0., 0., 0.));
    CHECK(or_mesh_set_zone(m, "OUTLET", ORC_BC_PRESSURE_OUTLET, 0., 0., 0., 0.));
    CHECK(or_mesh_set_zone(m, "PERIODIC_-Z", ORC_BC_SYMMETRY, 0., 0., 0., 0.));
    CHECK(or_mesh_set_zone(m, "PERIODIC_+Z", ORC_BC_SYMMETRY, 0., 0., 0., 0.));
    const int64_t n = m->n_cells;
    double *f = (double *)calloc((size_t)(4 * n), sizeof(double));
    double *u = f, *v = f + n, *w = f + 2 * n, *p = f + 3 * n;
    CHECK(or_initialize_flow(m, 1e-3, 1000., 20, 1, u, v, w, p));
    const int solvers[3] = {ORC_SOLVER_JACOBI, ORC_SOLVER_BICGSTAB, ORC_SOLVER_MULTIGRID};
    const int schemes[3] = {ORC_MOMENTUM_UD, ORC_MOMENTUM_CD1, ORC_MOMENTUM_TVD_UMIST};
    double checksum = 0.;
    for (int k = 0; k < 3; ++k) {
        OrcSettings s;
        or_settings_default(&s);
        s.solver_type = solvers[k];
        s.momentum = schemes[k];
        s.iterations = 8;
        double *g = (double *)malloc(sizeof(double) * (size_t)(4 * n));
        memcpy(g, f, sizeof(double) * (size_t)(4 * n));
        double report[6 * 3];
        CHECK(or_solve_steady(m, g, g + n, g + 2 * n, g + 3 * n, &s, 1000., 1e-3, 3, report));
        for (int64_t i = 0; i < 4 * n; ++i) checksum += g[i];
        free(g);
    }
    {   /* round 2: least-squares gradient arms inside the SIMPLE loop + the post-loop pass, and the velocity initialisation */
        OrcSettings s;
        or_settings_default(&s);
        s.solver_type = ORC_SOLVER_BICGSTAB;
        s.momentum = ORC_MOMENTUM_TVD_UMIST;
        s.gradient_reconstruction = ORC_GRAD_LEAST_SQUARES;
        s.iterations = 8;
        double *g = (double *)malloc(sizeof(double) * (size_t)(4 * n));
        memcpy(g, f, sizeof(double) * (size_t)(4 * n));
        double report[6 * 2];
        CHECK(or_solve_steady(m, g, g + n, g + 2 * n, g + 3 * n, &s, 1000., 1e-3, 2, report));
        for (int64_t i = 0; i < 4 * n; ++i) checksum += g[i];
        CHECK(or_mesh_set_zone(m, "INLET", ORC_BC_VELOCITY_INLET, 0., 1e-3, 0., 0.));
        CHECK(or_initialize_velocity_field(m, g, g + n, g + 2 * n, g + 3 * n));
        for (int64_t i = 0; i < 4 * n; ++i) checksum += g[i];
        free(g);
    }
    free(f);
    or_mesh_free(m);
    printf("sanitize_main ok: %lld cells, checksum %.17g\n", (long long)n, checksum);
    return 0;
}
