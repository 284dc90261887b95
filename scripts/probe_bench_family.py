#!/usr/bin/env python3
"""Measurement behind tests/test_gpu_bench_family.py's tolerances: the device's product default against the oracle in frozen and
in the reference's in-place mode on the bench-family channel (report norms per iteration, field differences)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import orc_amd
from oracle import pyoracle as po
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver
orc_amd.init(0)
KW = dict(momentum=5, solver_type=2, iterations=50, momentum_relaxation=0.1, pressure_relaxation=0.001)
for shape in [(40, 16, 16), (80, 32, 32)]:
    a = set_channel_bcs(hex_channel(*shape))
    om = po.Mesh.from_arrays(a)
    f0 = bench.initial_fields(np.asarray(a["cell_centroid"]))
    its = 4
    refs = {}
    for frozen in (1, 0):
        ref = [np.ascontiguousarray(x).copy() for x in f0]
        st, rep = po.solve_steady(om, *ref, po.default_settings(frozen_diagonals=frozen, breakdown_guard=0, **KW), 1000.0, 1e-3, its, report=True)
        refs[frozen] = (st, rep, ref)
    s = Solver(Mesh(a), NumericalSettings.default(**KW), 1000.0, 1e-3)
    s.set_fields(*f0)
    std, rep_d = s.iterate(its, report=True, raise_on_error=False)
    fd = s.get_fields()
    print(shape, "status", std, refs[1][0], refs[0][0])
    for it in range(its):
        print(" it", it + 1, "device vc/pc %.9e %.9e | oracle frozen %.9e %.9e | oracle in-place %.9e %.9e" %
              (rep_d[it, 6], rep_d[it, 7], refs[1][1][it, 4], refs[1][1][it, 5], refs[0][1][it, 4], refs[0][1][it, 5]))
        print("      u_mean device %.12e frozen %.12e in-place %.12e" % (rep_d[it, 0], refs[1][1][it, 0], refs[0][1][it, 0]))
    for name, x, yf, yi in zip("uvwp", fd, refs[1][2], refs[0][2]):
        un = np.linalg.norm(refs[1][2][0]) if name in "uvw" else np.linalg.norm(yf)
        print("  %s: device vs oracle-frozen %.3e  device vs oracle-in-place %.3e (relative to |u| or |p|)" % (name, np.linalg.norm(x - yf) / un, np.linalg.norm(x - yi) / un))
