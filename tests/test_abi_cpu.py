"""The C-ABI library: loads without a GPU, exports every symbol include/orc_amd.h declares, shares the settings layout
with the oracle, fails loudly (no CPU fallback) when no HIP device is visible, and never touches oracle/.  CPU only —
no compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "orc_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orc_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    import orc_amd
    if not os.path.exists(orc_amd._lib.LIB_PATH):
        orc_amd.build()
    return orc_amd._lib.lib()


def test_library_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_settings_default_layout_matches_reference_defaults(lib, oracle):
    from orc_amd.settings import NumericalSettings
    s = NumericalSettings.default()
    # lib.rs:58-86
    assert (s.momentum, s.diffusion, s.pressure_interpolation, s.velocity_interpolation, s.gradient_reconstruction) == (1, 0, 3, 2, 0)
    assert (s.solver_type, s.preconditioner, s.iterations) == (2, 1, 50)
    assert (s.momentum_relaxation, s.pressure_relaxation, s.relaxation, s.relative_convergence_threshold) == (0.5, 0.01, 0.5, 1e-3)
    assert s.q1_compat == 1 and s.frozen_diagonals == 1 and s.breakdown_guard == 1 and s.reduction_order == 0
    o = oracle.default_settings()
    assert C.sizeof(s) == C.sizeof(o) == 88
    assert o.frozen_diagonals == 0 and o.breakdown_guard == 0  # the oracle defaults to the reference's own behaviour
    for f, _ in s._fields_[:13]:
        assert getattr(s, f) == getattr(o, f), f
    # compiled size of the C struct
    src = '#include <stdio.h>\n#include "orc_types.h"\nint main(){printf("%zu", sizeof(OrcSettings));return 0;}'
    exe = os.path.join(ROOT, "tests", ".sizeof_tmp")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src.encode(), check=True)
    try:
        assert int(subprocess.check_output([exe])) == 88
    finally:
        os.remove(exe)


def test_no_cpu_fallback_without_device(lib):
    import orc_amd
    from orc_amd import OrcError
    if orc_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    import scipy.sparse as sp
    from orc_amd.linear_algebra import csr_spmv, iterative_solve
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    with pytest.raises(OrcError) as e:
        iterative_solve(sp.identity(4, format="csr"), np.ones(4), np.zeros(4), 1, 3, 0.5, 1e-3, 1)
    assert e.value.status == 11
    with pytest.raises(OrcError) as e:
        csr_spmv(sp.identity(4, format="csr"), np.ones(4))
    assert e.value.status == 11
    with pytest.raises(OrcError) as e:
        Mesh(set_channel_bcs(hex_channel(3, 2, 2)))
    assert e.value.status == 11
    assert lib.orc_status_string(C.c_int(1)) == b"solution diverged"  # solver.rs:220
    assert lib.orc_status_string(C.c_int(2)) == b"Multigrid diverged"  # linear_algebra.rs:104


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liborc_oracle" not in txt and "oracle/" not in txt.replace("// oracle", ""), f


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 4, 3), (9, 2, 1)])
def test_hex_channel_generator_equals_reader_of_its_own_msh(oracle, tmp_path, shape):
    """orc_hex_channel_generate (arrays) and orc_hex_channel_write_msh -> ORC's read_mesh rules (oracle) describe the same
    Mesh bit for bit: numbering, adjacency, zone order, geometry."""
    from orc_amd.mesh import ZONE_NAMES, hex_channel, write_hex_channel_msh
    nx, ny, nz = shape
    a = hex_channel(nx, ny, nz)
    p = str(tmp_path / "m.msh")
    write_hex_channel_msh(p, nx, ny, nz)
    om = oracle.Mesh.read(p)
    oa = om.arrays()
    assert a.n_cells == nx * ny * nz and a.n_faces == (nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1) + 2 * (ny * nz + nx * ny + nx * nz)
    for k in ("face_c0", "face_c1", "cell_face_ptr", "cell_faces"):
        assert np.array_equal(np.asarray(a[k]), oa[k]), k
    for k in ("face_area", "face_normal", "face_centroid", "cell_centroid", "cell_volume"):
        assert np.array_equal(np.asarray(a[k]), oa[k]), k
    names = om.zone_names()
    present = [ZONE_NAMES[z] for z in sorted(set(np.asarray(a["face_zone"]).tolist()))]
    assert names == present
    assert np.allclose(a["cell_volume"].sum(), 0.002 * 0.001 * 1e-4 * nz)


def test_hex_channel_c4_sizes():
    """400x160x160 -> 10 240 000 cells, 30 873 600 faces (30 566 400 interior)."""
    from orc_amd._lib import check, lib
    nc, nf, ncf = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib().orc_hex_channel_sizes(C.c_int64(400), C.c_int64(160), C.c_int64(160), C.byref(nc), C.byref(nf), C.byref(ncf)))
    assert (nc.value, nf.value, ncf.value) == (10240000, 30873600, 61440000)


def test_partial_sum_grids_have_one_hard_bound(lib):
    """VERDICT r03 weak #7 / ADVICE r03: every launcher whose workgroups write partials[q * gridDim.x + blockIdx.x] sizes its grid
    through ONE function (common.hpp: clamp_partials_grid; grid_for, spmv_grid, the window product's balanced grid and the
    cascade grids all end in it), so no CU count and no measurement switch can ask for more workgroups than the partial-sum
    arrays (kMaxPartials entries per quantity) hold.  Host-only: no device needed."""
    cap = lib.orc_debug_max_partials()
    assert cap == 2048
    f = lib.orc_debug_clamp_partials_grid
    f.argtypes = [C.c_longlong]
    assert f(0) == 1 and f(-7) == 1 and f(1) == 1
    assert f(cap - 1) == cap - 1 and f(cap) == cap
    assert f(cap + 1) == cap and f(304 * 8) == cap and f(256 * 40) == cap and f(1 << 40) == cap
