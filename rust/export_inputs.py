#!/usr/bin/env python3
"""Writes the inputs rust/dump_golden.rs reads (raw little-endian arrays): the reference's mesh fixtures this repository
ships, the seeded start fields of tests/golden/*.npz (u0, v0, w0, p0), and the linear systems of the parity tests.
    python rust/export_inputs.py rust/inputs"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import fv_like_matrix, splitmix64_uniform, unit_test_system  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rust", "inputs")
os.makedirs(os.path.join(out, "meshes"), exist_ok=True)
for name in ("3x3_cube.msh", "channel_flow.msh", "couette_flow_8x8x1.msh"):
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "meshes", name), os.path.join(out, "meshes", name))
for case in ("3x3_cube", "3x3_cube_mixed", "channel_flow"):
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    d = os.path.join(out, case)
    os.makedirs(d, exist_ok=True)
    for k in ("u0", "v0", "w0", "p0"):
        np.ascontiguousarray(g[k], dtype="<f8").tofile(os.path.join(d, k + ".f64"))


def export_system(name, a, b, x0, threshold):
    d = os.path.join(out, "systems", name)
    os.makedirs(d, exist_ok=True)
    a = a.tocsr()
    a.sort_indices()
    np.ascontiguousarray(a.indptr, dtype="<i8").tofile(os.path.join(d, "row_ptr.i64"))
    np.ascontiguousarray(a.indices, dtype="<i8").tofile(os.path.join(d, "col.i64"))
    np.ascontiguousarray(a.data, dtype="<f8").tofile(os.path.join(d, "values.f64"))
    np.ascontiguousarray(b, dtype="<f8").tofile(os.path.join(d, "b.f64"))
    np.ascontiguousarray(x0, dtype="<f8").tofile(os.path.join(d, "x0.f64"))
    np.array([threshold], dtype="<f8").tofile(os.path.join(d, "threshold.f64"))


a, b, sol = unit_test_system()  # linear_algebra.rs:313-337
export_system("unit_test", a, b, np.zeros(len(b)), 1e-3 / len(b) ** 3)
for shape in ((7, 5, 3), (20, 17, 9), (33, 9, 4)):
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    export_system("fv_%dx%dx%d" % shape, a, a @ splitmix64_uniform(n, 7), 0.1 * splitmix64_uniform(n, 8), 1e-3)
print("inputs written to", out)
