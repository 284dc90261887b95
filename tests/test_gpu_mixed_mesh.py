"""BASELINE config 5 in miniature: a skewed mixed prism/hex mesh (tests/meshgen.py) through the whole path — product
reader, device SIMPLE loop with every solver arm, exact aggregation on rows of unequal length — against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import helpers as H
import meshgen

pytestmark = pytest.mark.gpu

JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3


@pytest.fixture(scope="module")
def mixed(tmp_path_factory, oracle):
    from orc_amd import io as orc_io
    path = str(tmp_path_factory.mktemp("mixed") / "mixed.msh")
    info = meshgen.write_mixed_channel_msh(path, 10, 6, 4, skew=0.2)
    om = oracle.Mesh.read(path)
    meshgen.mixed_channel_bcs(om.set_zone, info["zone_names"], top_wall_velocity=5e-4)
    d = orc_io.read_mesh(path)
    meshgen.mixed_channel_bcs(d.set_zone, info["zone_names"], top_wall_velocity=5e-4)
    return om, d, d.upload(), d.arrays()


def test_mixed_mesh_shape(gpu, mixed):
    om, d, dm, a = mixed
    nf = np.diff(a["cell_face_ptr"])
    assert set(np.unique(nf)) == {5, 6}                      # prisms and hexahedra
    rp, ci = dm.matrix_pattern()
    assert set(np.unique(np.diff(rp))) <= {4, 5, 6, 7} and len(set(np.diff(rp))) >= 3   # ragged rows
    ortho = np.einsum("ij,ij->i", a["face_normal"], a["cell_centroid"][np.maximum(a["face_c1"], 0)] - a["cell_centroid"][a["face_c0"]])
    dist = np.linalg.norm(a["cell_centroid"][np.maximum(a["face_c1"], 0)] - a["cell_centroid"][a["face_c0"]], axis=1)
    inner = a["face_c1"] >= 0
    assert (ortho[inner] / dist[inner]).min() < 0.995        # genuinely non-orthogonal faces


def test_simple_iterations_jacobi_bit_exact(gpu, oracle, mixed):
    """No reduction in the Jacobi data path: ten SIMPLE iterations, TVD-UMIST + Rhie-Chow, identical bits."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, d, dm, a = mixed
    kw = dict(momentum=5, solver_type=JACOBI, frozen_diagonals=1)
    u, v, w, p = H.seeded_fields(a, seed=9)
    uo, vo, wo, po = (x.copy() for x in (u, v, w, p))
    st, _ = oracle.solve_steady(om, uo, vo, wo, po, oracle.default_settings(**kw), 1000.0, 1e-3, 10)
    assert st == 0
    solve_steady(dm, u, v, w, p, NumericalSettings.default(**kw), 1000.0, 1e-3, 10)
    for x, y in ((u, uo), (v, vo), (w, wo), (p, po)):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("solver,tol", [(BICGSTAB, 1e-9), (MULTIGRID, 1e-7)])
def test_simple_iteration_reduction_solvers(gpu, oracle, mixed, solver, tol):
    """One SIMPLE iteration with short inner solves (beyond that the reference's BiCGSTAB is chaotic in the dot-product
    association, tests/test_oracle_sensitivity.py): bit-exact assembly + SpMV + hierarchy, ulps from the reductions."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, d, dm, a = mixed
    kw = dict(momentum=4, solver_type=solver, iterations=4, frozen_diagonals=1)
    u, v, w, p = H.seeded_fields(a, seed=4)
    uo, vo, wo, po = (x.copy() for x in (u, v, w, p))
    st, _ = oracle.solve_steady(om, uo, vo, wo, po, oracle.default_settings(**kw), 1000.0, 1e-3, 1)
    assert st == 0
    solve_steady(dm, u, v, w, p, NumericalSettings.default(**kw), 1000.0, 1e-3, 1)
    for x, y in ((u, uo), (v, vo), (w, wo), (p, po)):
        assert H.rel_l2(x, y) < tol, H.rel_l2(x, y)


def test_aggregation_and_galerkin_on_ragged_rows(gpu, oracle, mixed):
    """The momentum matrix of the mixed mesh (4 to 7 entries per row, unsymmetric values): device pairing == the
    reference's sequential greedy pairing, Galerkin product == (R a) R^T bit for bit."""
    from orc_amd import discretization as D
    from orc_amd.linear_algebra import amg_coarsen
    from orc_amd.settings import NumericalSettings
    om, d, dm, a = mixed
    u, v, w, p = H.seeded_fields(a, seed=6)
    s = NumericalSettings.default(momentum=5)
    a_di, *_ = D.build_momentum_diffusion_matrix(dm, 1e-3)
    mats = [D.initialize_momentum_matrix(dm) for _ in range(3)]
    D.build_momentum_advection_matrices(dm, mats[0], mats[1], mats[2], a_di, u, v, w, p, s, 1000.0)
    A = dm.csr(mats[0])
    n = A.shape[0]
    partner, ac, rounds = amg_coarsen(A)
    Ao = oracle.Csr.from_scipy(A)
    R = oracle.build_restriction_matrix(Ao)
    rows, cols = [], []
    for i in range(n):
        if partner[i] >= 0:
            rows += [i // 2, i // 2]
            cols += [i, int(partner[i])]
    Rd = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=((n + 1) // 2, n)).tocsr()
    Rd.sum_duplicates()
    assert abs(Rd - R.to_scipy()).max() == 0
    ref = R.matmul(Ao).matmul(R.transpose())
    rp, ci, val = ref.arrays()
    assert np.array_equal(ac.indptr, rp) and np.array_equal(ac.indices, ci) and np.array_equal(ac.data, val)


def test_initialize_flow_on_mixed_mesh(gpu, oracle, mixed):
    from orc_amd.solver import initialize_flow
    om, d, dm, a = mixed
    st, uo, vo, wo, po = oracle.initialize_flow(om, 1e-3, 1000.0, 60)
    assert st == 0 and not np.isnan(uo).any()
    u, v, w, p = initialize_flow(dm, 1e-3, 1000.0, 60)
    assert np.array_equal(p, po)
    assert H.rel_l2(u, uo) < 1e-6 and H.rel_l2(v, vo) < 1e-5, (H.rel_l2(u, uo), H.rel_l2(v, vo))
