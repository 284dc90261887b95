#!/usr/bin/env python3
"""The oracle's (oracle/: CPU restatement of ORC) version of rust/dump_golden.rs: the same calls on the same inputs, the same
file layout.  Used by rust/compare_with_oracle.py (bit-for-bit comparison with a real ORC dump) and by the CPU suite, which
round-trips the format (tests/test_golden_cpu.py).
    python rust/oracle_dump.py rust/inputs /tmp/oracle_dump"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

UD, CD1, UMIST = 0, 1, 5
JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3
ASSEMBLY_CASES = (("3x3_cube", "3x3_cube.msh", H.cube_bcs, UMIST), ("3x3_cube_mixed", "3x3_cube.msh", H.cube_bcs_mixed, CD1),
                  ("channel_flow", "channel_flow.msh", H.channel_bcs, CD1))
SOLVE_CASES = (("channel_flow", "channel_flow.msh", H.channel_bcs), ("3x3_cube", "3x3_cube.msh", H.cube_bcs))
SOLVE_VARIANTS = (("multigrid_cd1_5it", CD1, MULTIGRID, 50), ("multigrid_umist_5it", UMIST, MULTIGRID, 50),
                  ("multigrid_umist_20inner_5it", UMIST, MULTIGRID, 20), ("bicgstab_umist_5it", UMIST, BICGSTAB, 50), ("jacobi_ud_5it", UD, JACOBI, 50))


def rd(path, dtype):
    return np.fromfile(path, dtype=dtype)


def wr(path, a, dtype="<f8"):
    np.ascontiguousarray(a, dtype=dtype).tofile(path)


def status_text(st):
    return "ok" if st == 0 else po.status_string(st)


def fields(inputs, case):
    return [rd(os.path.join(inputs, case, k + ".f64"), "<f8").copy() for k in ("u0", "v0", "w0", "p0")]


def main(inputs, out):
    po.build()
    rho, mu = 1000.0, 1e-3
    for case, mesh_file, bcs, momentum in ASSEMBLY_CASES:
        d = os.path.join(out, case)
        os.makedirs(d, exist_ok=True)
        om = bcs(po.Mesh.read(os.path.join(inputs, "meshes", mesh_file)))
        u, v, w, p = fields(inputs, case)
        s = po.default_settings(momentum=momentum, frozen_diagonals=0, breakdown_guard=0)  # the reference's own mode
        a_di, bu_di, bv_di, bw_di = po.build_momentum_diffusion_matrix(om, mu)
        rp, ci, val = a_di.arrays()
        wr(os.path.join(d, "pattern_row_ptr.i64"), rp, "<i8")
        wr(os.path.join(d, "pattern_col.i64"), ci, "<i8")
        wr(os.path.join(d, "a_di.f64"), val)
        for nm, b in (("b_u_di", bu_di), ("b_v_di", bv_di), ("b_w_di", bw_di)):
            wr(os.path.join(d, nm + ".f64"), b)
        mats = [po.initialize_momentum_matrix(om) for _ in range(3)]
        wr(os.path.join(d, "a_init.f64"), mats[0].arrays()[2])
        for it in (1, 2):
            bu, bv, bw, pe = po.build_momentum_advection_matrices(mats[0], mats[1], mats[2], a_di, om, u, v, w, p, s, rho)
            for nm, m in zip("uvw", mats):
                wr(os.path.join(d, "a_%s_it%d.f64" % (nm, it)), m.arrays()[2])
            for nm, b in zip("uvw", (bu, bv, bw)):
                wr(os.path.join(d, "b_%s_it%d.f64" % (nm, it)), b)
            wr(os.path.join(d, "peclet_it%d.f64" % it), np.array(pe))
        a_p, b_p = po.build_pressure_correction_matrices(om, u, v, w, p, mats[0], mats[1], mats[2], s, rho)
        rp, ci, val = a_p.arrays()
        wr(os.path.join(d, "p_pattern_row_ptr.i64"), rp, "<i8")
        wr(os.path.join(d, "p_pattern_col.i64"), ci, "<i8")
        wr(os.path.join(d, "a_p.f64"), val)
        wr(os.path.join(d, "b_p.f64"), b_p)
    for case, mesh_file, bcs in SOLVE_CASES:
        d = os.path.join(out, case)
        os.makedirs(d, exist_ok=True)
        for tag, momentum, solver, inner in SOLVE_VARIANTS:
            om = bcs(po.Mesh.read(os.path.join(inputs, "meshes", mesh_file)))
            f = fields(inputs, case)
            s = po.default_settings(momentum=momentum, solver_type=solver, iterations=inner, frozen_diagonals=0, breakdown_guard=0)
            st, _ = po.solve_steady(om, *f, s, rho, mu, 5)
            open(os.path.join(d, "solve_steady_%s_status.txt" % tag), "w").write(status_text(st))
            if st == 0:
                for nm, x in zip("uvwp", f):
                    wr(os.path.join(d, "solve_steady_%s_%s.f64" % (tag, nm)), x)
    om = H.channel_bcs(po.Mesh.read(os.path.join(inputs, "meshes", "channel_flow.msh")))
    st, u, v, w, p = po.initialize_flow(om, mu, rho, 40)
    d = os.path.join(out, "channel_flow")
    open(os.path.join(d, "initialize_flow_status.txt"), "w").write(status_text(st))
    if st == 0:
        for nm, x in zip("uvwp", (u, v, w, p)):
            wr(os.path.join(d, "initialize_flow_%s.f64" % nm), x)
    sysdir = os.path.join(inputs, "systems")
    for name in sorted(os.listdir(sysdir)) if os.path.isdir(sysdir) else []:
        src, d = os.path.join(sysdir, name), os.path.join(out, "systems", name)
        os.makedirs(d, exist_ok=True)
        rp, ci = rd(os.path.join(src, "row_ptr.i64"), "<i8"), rd(os.path.join(src, "col.i64"), "<i8")
        val, b, x0 = (rd(os.path.join(src, k + ".f64"), "<f8") for k in ("values", "b", "x0"))
        thr = float(rd(os.path.join(src, "threshold.f64"), "<f8")[0])
        import scipy.sparse as sp
        A = po.Csr.from_scipy(sp.csr_matrix((val, ci, rp), shape=(len(b), len(b))))
        for method, mname in ((JACOBI, "jacobi"), (BICGSTAB, "bicgstab"), (MULTIGRID, "multigrid")):
            for pre, pname in ((0, "none"), (1, "jacobi")):
                x = x0.copy()
                st = po.iterative_solve(A, b, x, 50, method, 0.5, thr, pre)
                tag = "%s_pre_%s" % (mname, pname)
                open(os.path.join(d, tag + "_status.txt"), "w").write(status_text(st))
                if st == 0:
                    wr(os.path.join(d, tag + "_x.f64"), x)
        x = x0.copy()
        if po.iterative_solve(A, b, x, 50, JACOBI, 0.5, thr, 1) == 0:
            xj = x.copy()
            if po.iterative_solve(A, b, x, 50, BICGSTAB, 0.5, thr, 1) == 0:
                wr(os.path.join(d, "chained_x_after_jacobi.f64"), xj)
                wr(os.path.join(d, "chained_x_after_bicgstab.f64"), x)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
