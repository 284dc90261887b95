"""Shared builders for the parity tests (mesh fixtures with the reference's BC set-ups, seeded fields)."""
import numpy as np

from conftest import splitmix64_uniform

BC_INTERIOR, BC_WALL, BC_PINLET, BC_POUTLET, BC_SYMMETRY, BC_VINLET = 2, 3, 4, 5, 7, 10


def channel_bcs(om, top_wall_velocity=0.0, dp_dx=5.0):
    """tests.rs:60-76 on an oracle mesh (names as in the fixture)."""
    names = om.zone_names()
    if "TOP_WALL" in names:
        om.set_zone("TOP_WALL", BC_WALL, 0.0, (top_wall_velocity, 0.0, 0.0))
        om.set_zone("BOTTOM_WALL", BC_WALL)
    else:
        om.set_zone("WALL", BC_WALL)
    om.set_zone("INLET", BC_PINLET, -dp_dx * 0.002)
    om.set_zone("OUTLET", BC_POUTLET, 0.0)
    om.set_zone("PERIODIC_-Z", BC_SYMMETRY)
    om.set_zone("PERIODIC_+Z", BC_SYMMETRY)
    return om


def cube_bcs(om):
    """main.rs:282-293 (test_3d_3x3): pressure inlet/outlet, walls elsewhere."""
    om.set_zone("INLET", BC_PINLET, 1.0)
    om.set_zone("OUTLET", BC_POUTLET, 0.0)
    om.set_zone("PERIODIC_-Z", BC_WALL)
    om.set_zone("PERIODIC_+Z", BC_WALL)
    return om


def cube_bcs_mixed(om):
    """every supported BC type at once on the 3x3 cube: velocity inlet, pressure outlet, moving wall, symmetry."""
    om.set_zone("INLET", BC_VINLET, 0.0, (0.3, 0.02, -0.01))
    om.set_zone("OUTLET", BC_POUTLET, 0.25)
    om.set_zone("PERIODIC_-Z", BC_SYMMETRY)
    om.set_zone("PERIODIC_+Z", BC_WALL, 0.0, (0.1, 0.05, 0.0))
    return om


def line_bcs(om):
    """main.rs:196-211 (test_3d_1x3)."""
    om.set_zone("INLET", BC_VINLET, 0.0, (1.0, 0.0, 0.0))
    om.set_zone("OUTLET", BC_POUTLET, 0.0)
    om.set_zone("WALL", BC_WALL)
    return om


def seeded_fields(arrays, seed=1, scale_u=1e-3, scale_p=1e-2, w_zero=False):
    """smooth profile + splitmix64 noise: nothing is exactly zero or exactly equal between neighbours."""
    cc = np.asarray(arrays["cell_centroid"])
    n = len(cc)
    span = cc.max(axis=0) - cc.min(axis=0) + 1e-300
    x, y, z = ((cc - cc.min(axis=0)) / span).T
    r = [splitmix64_uniform(n, seed + k) for k in range(4)]
    u = scale_u * (4 * y * (1 - y) + 0.3 * r[0])
    v = scale_u * (0.2 * np.sin(3 * x) + 0.1 * r[1])
    w = np.zeros(n) if w_zero else scale_u * (0.1 * np.cos(2 * z + x) + 0.1 * r[2])
    p = scale_p * (1 - x + 0.05 * r[3])
    return tuple(np.ascontiguousarray(a) for a in (u, v, w, p))


def rel_l2(a, b):
    d = np.linalg.norm(np.asarray(a) - np.asarray(b))
    return d / max(np.linalg.norm(b), 1e-300)


def analytical_poiseuille(y, h=0.001, mu=1e-3, dp_dx=5.0, u_top=0.0):
    """tests.rs:26-29"""
    return u_top * y / h + 1.0 / (2.0 * mu) * dp_dx * (y * y - h * y)


def rough_fields(cc, dp_dx=5.0, h=1e-3, mu=1e-3, lx=0.002, seed=0x524F55):
    """The bench's Poiseuille start (bench.initial_fields) with PER-CENT-level cell-to-cell noise on u and p and v, w of a per cent of
    the bulk velocity: successive velocity differences change sign from cell to cell, so every branch of the TVD limiters
    (lib.rs:107-118), both upwind directions and the zero-difference fallback's neighbourhood are live in one assembly.
    A formula, so that tests/golden/make_golden_bench_midsize.py --frozen and the GPU test build the same doubles."""
    n = len(cc)
    y = cc[:, 1]
    r = [splitmix64_uniform(n, seed + k) for k in range(4)]
    ub = 1.0 / (2.0 * mu) * dp_dx * (y * y - h * y)
    u = ub * (1.0 + 0.05 * r[0])
    v = 0.02 * np.abs(ub).max() * r[1]
    w = 0.02 * np.abs(ub).max() * r[2]
    p = -dp_dx * lx * (1.0 - cc[:, 0] / lx) * (1.0 + 0.03 * r[3])
    return tuple(np.ascontiguousarray(a) for a in (u, v, w, p))
