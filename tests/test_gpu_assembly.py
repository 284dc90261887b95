"""HIP face-loop assembly (K9-K14) vs the CPU oracle in frozen-diagonal mode: bit-exact."""
import itertools

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

UD, CD1, CD2, LUD, QUICK, UMIST = 0, 1, 2, 3, 4, 5


def make(oracle, mesh_path, name):
    """-> (oracle mesh, device mesh, arrays)"""
    from orc_amd.mesh import Mesh, MeshArrays, hex_channel, set_channel_bcs
    if name == "hex6x5x4":
        a = set_channel_bcs(hex_channel(6, 5, 4), top_wall_velocity=5e-4)
        om = oracle.Mesh.from_arrays(a)
        return om, Mesh(a), a
    om = oracle.Mesh.read(mesh_path(name))
    {"channel_flow": H.channel_bcs, "3x3_cube": H.cube_bcs, "3x3_cube_mixed": H.cube_bcs_mixed, "3D_1x3": H.line_bcs}[name](om)
    a = MeshArrays(om.arrays())
    return om, Mesh(a), a


MESHES = ["3D_1x3", "3x3_cube", "hex6x5x4", "channel_flow"]


_MIXED_CACHE = {}


def mixed_mesh_file(kind):
    """prism/hex fixture written once per session (tests/meshgen.py): "prism_hex" alternates columns, "prism_all" is all prisms"""
    import os
    import tempfile

    import meshgen
    if kind not in _MIXED_CACHE:
        path = os.path.join(tempfile.mkdtemp(prefix="orc_mixed_"), kind + ".msh")
        info = meshgen.write_mixed_channel_msh(path, 6, 4, 3, split="checker" if kind == "prism_hex" else "all")
        _MIXED_CACHE[kind] = (path, info)
    return _MIXED_CACHE[kind]


def make_named(oracle, mesh_path, name):
    if name in ("prism_hex", "prism_all"):
        import meshgen
        from orc_amd.mesh import Mesh, MeshArrays
        path, info = mixed_mesh_file(name)
        om = oracle.Mesh.read(path)
        meshgen.mixed_channel_bcs(om.set_zone, info["zone_names"], top_wall_velocity=5e-4)
        a = MeshArrays(om.arrays())
        return om, Mesh(a), a
    if name == "3x3_cube_mixed":
        om = oracle.Mesh.read(mesh_path("3x3_cube"))
        H.cube_bcs_mixed(om)
        from orc_amd.mesh import Mesh, MeshArrays
        a = MeshArrays(om.arrays())
        return om, Mesh(a), a
    return make(oracle, mesh_path, name)


@pytest.mark.parametrize("name", MESHES + ["3x3_cube_mixed", "prism_hex", "prism_all"])
def test_pattern_diffusion_and_init_matrix(gpu, oracle, mesh_path, name):
    from orc_amd import discretization as D
    om, dm, a = make_named(oracle, mesh_path, name)
    A_di, bu, bv, bw = oracle.build_momentum_diffusion_matrix(om, 1e-3)
    rp, ci, val = A_di.arrays()
    drp, dci = dm.matrix_pattern()
    assert np.array_equal(rp, drp) and np.array_equal(ci, dci)
    dv, dbu, dbv, dbw = D.build_momentum_diffusion_matrix(dm, 1e-3)
    assert np.array_equal(dv, val)
    assert np.array_equal(dbu, bu) and np.array_equal(dbv, bv) and np.array_equal(dbw, bw)
    init = oracle.initialize_momentum_matrix(om)
    assert np.array_equal(D.initialize_momentum_matrix(dm), init.arrays()[2])


@pytest.mark.parametrize("q1", [1, 0])
@pytest.mark.parametrize("name", ["3x3_cube", "hex6x5x4", "3x3_cube_mixed", "prism_hex"])
def test_green_gauss_gradients(gpu, oracle, mesh_path, name, q1):
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import calculate_gradients
    om, dm, a = make_named(oracle, mesh_path, name)
    u, v, w, p = H.seeded_fields(a, seed=5)
    s = NumericalSettings.default(q1_compat=q1)
    gp, gu = calculate_gradients(dm, u, v, w, p, s)
    ref = oracle.pressure_gradient(om, p, q1=q1)
    assert np.array_equal(gp, ref)
    if q1:  # the reference returns (gx, gy, gy)
        assert np.array_equal(gp[:, 2], gp[:, 1])
    # velocity gradient vs an independent numpy Green-Gauss sum (tolerance: different association)
    F = len(a["face_area"])
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    zt = np.asarray(a["zone_type"])[np.asarray(a["face_zone"])]
    zv = np.asarray(a["zone_vector"])[np.asarray(a["face_zone"])]
    vel = np.stack([u, v, w], axis=1)
    fv = np.where((c1 >= 0)[:, None], 0.5 * (vel[c0] + vel[np.maximum(c1, 0)]), vel[c0])
    wall = (zt == 3) | (zt == 10)
    fv[wall] = zv[wall]
    nA = np.asarray(a["face_normal"]) * np.asarray(a["face_area"])[:, None]
    g = np.zeros((len(u), 3, 3))
    np.add.at(g, c0, fv[:, :, None] * nA[:, None, :])
    m = c1 >= 0
    np.add.at(g, c1[m], -fv[m][:, :, None] * nA[m][:, None, :])
    g /= np.asarray(a["cell_volume"])[:, None, None]
    assert np.allclose(gu, g, rtol=1e-11, atol=1e-9 * np.abs(g).max())


SCHEMES = [(UD, 2, 3), (CD1, 2, 3), (QUICK, 2, 3), (UMIST, 2, 3), (LUD, 0, 0), (UMIST, 1, 1), (CD1, 0, 1), (UD, 1, 0)]


@pytest.mark.parametrize("q1", [1, 0])
@pytest.mark.parametrize("momentum,vinterp,pinterp", SCHEMES)
@pytest.mark.parametrize("name", ["3x3_cube", "hex6x5x4", "3x3_cube_mixed", "channel_flow", "prism_hex", "prism_all"])
def test_momentum_and_pressure_assembly_bit_exact(gpu, oracle, mesh_path, name, momentum, vinterp, pinterp, q1):
    """build_momentum_advection_matrices + build_pressure_correction_matrices, every scheme combination,
    vs the oracle with frozen diagonals: identical bits (NaNs included, e.g. QUICK where a velocity
    component is constant: SURVEY Q10)."""
    from orc_amd import discretization as D
    from orc_amd.settings import NumericalSettings
    if name == "channel_flow" and (q1 == 0 or (momentum, vinterp, pinterp) not in SCHEMES[:4]):
        pytest.skip("large fixture: default interpolation only")
    om, dm, a = make_named(oracle, mesh_path, name)
    rho = 1000.0
    u, v, w, p = H.seeded_fields(a, seed=3, w_zero=(name == "channel_flow"))
    kw = dict(momentum=momentum, velocity_interpolation=vinterp, pressure_interpolation=pinterp, q1_compat=q1, frozen_diagonals=1)
    s_dev = NumericalSettings.default(**kw)
    s_orc = oracle.default_settings(**kw)
    A_di, *_ = oracle.build_momentum_diffusion_matrix(om, 1e-3)
    # incoming matrices with non-trivial diagonals (iteration >= 2 of the SIMPLE loop)
    n = dm.n_cells
    diag_scale = 1.0 + 0.5 * np.abs(np.stack([np.sin(np.arange(n) + k) for k in range(3)]))
    mats_o = []
    mats_d = []
    for k in range(3):
        M = oracle.initialize_momentum_matrix(om)
        rp, ci, val = M.arrays()
        rows = np.repeat(np.arange(n), np.diff(rp))
        val[rows == ci] = diag_scale[k] * 1e-6
        mats_o.append(M)
        mats_d.append(val.copy())
    bu, bv, bw, pe = oracle.build_momentum_advection_matrices(mats_o[0], mats_o[1], mats_o[2], A_di, om, u, v, w, p, s_orc, rho)
    dbu, dbv, dbw, dpe = D.build_momentum_advection_matrices(dm, mats_d[0], mats_d[1], mats_d[2], A_di.arrays()[2], u, v, w, p, s_dev, rho)
    for k in range(3):
        assert np.array_equal(mats_d[k], mats_o[k].arrays()[2], equal_nan=True), "a_%s" % "uvw"[k]
    assert np.array_equal(dbu, bu, equal_nan=True) and np.array_equal(dbv, bv, equal_nan=True) and np.array_equal(dbw, bw, equal_nan=True)
    if not np.isnan(np.array(pe)).any():
        assert np.isclose(dpe[0], pe[0], rtol=1e-10) and dpe[1] == pe[1] and dpe[2] == pe[2]
    if np.isnan(mats_d[2]).any() or np.isnan(mats_d[1]).any():
        return
    # pressure correction with the freshly assembled diagonals
    A_p, b_p = oracle.build_pressure_correction_matrices(om, u, v, w, p, mats_o[0], mats_o[1], mats_o[2], s_orc, rho)
    da, db = D.build_pressure_correction_matrices(dm, u, v, w, p, mats_d[0], mats_d[1], mats_d[2], s_dev, rho)
    assert np.array_equal(da, A_p.arrays()[2]) and np.array_equal(db, b_p)


def test_unsupported_settings_and_bcs_return_reference_panics(gpu, oracle, mesh_path):
    from orc_amd import OrcError
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om = oracle.Mesh.read(mesh_path("3x3_cube"))
    a = MeshArrays(om.arrays())  # INLET/OUTLET are type 24 (Interface) in the file: unsupported BC
    dm = Mesh(a)
    with pytest.raises(OrcError) as e:
        Solver(dm, NumericalSettings.default(), 1000.0, 1e-3)
    assert e.value.status == 7  # "BC not supported" (discretization.rs:114-117)
    H.cube_bcs(om)
    dm = Mesh(MeshArrays(om.arrays()))
    for kw in (dict(momentum=CD2), dict(pressure_interpolation=2), dict(velocity_interpolation=3), dict(gradient_reconstruction=1),
               dict(gradient_reconstruction=3)):
        with pytest.raises(OrcError) as e:
            Solver(dm, NumericalSettings.default(**kw), 1000.0, 1e-3)
        assert e.value.status == 8
