"""io.rs on the product side: orc_read_mesh against the oracle's reader on every reference mesh and on
the synthetic TGRID writer's output, bit for bit; the reader hazards SURVEY lists; the text checkpoint formats.  Host
code only — runs without a GPU."""
import math
import os

import numpy as np
import pytest

import orc_amd
from orc_amd import io as orc_io
from orc_amd._lib import OrcError

GEOMETRY_KEYS = ["face_c0", "face_c1", "face_zone", "face_area", "face_normal", "face_centroid", "cell_centroid", "cell_volume",
                 "cell_face_ptr", "cell_faces", "zone_type"]
REFERENCE_MESHES = ["2D_2x4", "2D_3x6", "3D_1x3", "3x3_cube", "couette_flow_8x8x1", "channel_flow", "couette_flow_128x64x1"]


def assert_same_mesh(a, b):
    for k in GEOMETRY_KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, k
        assert np.array_equal(x, y), (k, np.abs(x.astype(float) - y.astype(float)).max())
    assert list(a["zone_names"]) == list(b["zone_names"])


@pytest.mark.parametrize("stem", REFERENCE_MESHES)
def test_read_mesh_matches_oracle_bitwise(oracle, mesh_path, stem):
    path = mesh_path(stem)
    d = orc_io.read_mesh(path)
    o = oracle.Mesh.read(path)
    assert_same_mesh(d.arrays(), o.arrays())
    assert d.dimensions == (2 if stem.startswith("2D") else 3)


def test_reference_mesh_inventory(mesh_path):
    """Counts quoted by the reference's own checks (main.rs:150-172: 3x3 cube has 27 cells, 108 faces, 64 nodes)."""
    d = orc_io.read_mesh(mesh_path("3x3_cube"))
    assert (d.n_cells, d.n_faces, d.n_vertices) == (27, 108, 64)
    a = d.arrays()
    assert np.allclose(a["cell_volume"].sum(), 27.0 * a["cell_volume"][0])
    # closed cells: sum of outward area vectors vanishes
    s = np.zeros((d.n_cells, 3))
    av = a["face_area"][:, None] * a["face_normal"]
    np.add.at(s, a["face_c0"], av)
    inner = a["face_c1"] >= 0
    np.add.at(s, a["face_c1"][inner], -av[inner])
    assert np.abs(s).max() < 1e-12 * a["face_area"].max()


def test_synthetic_tgrid_round_trip(oracle, tmp_path):
    """The generator's arrays == what the reader makes of the generator's own .msh (numbering + geometry rules)."""
    from orc_amd.mesh import hex_channel, write_hex_channel_msh
    path = str(tmp_path / "chan.msh")
    write_hex_channel_msh(path, 7, 5, 3)
    d = orc_io.read_mesh(path)
    a = d.arrays()
    assert_same_mesh(a, oracle.Mesh.read(path).arrays())
    g = hex_channel(7, 5, 3)
    for k in ["face_c0", "face_c1", "face_zone", "cell_face_ptr", "cell_faces"]:
        assert np.array_equal(a[k], g[k]), k
    for k in ["face_area", "face_normal", "face_centroid", "cell_centroid", "cell_volume"]:
        assert np.allclose(a[k], g[k], rtol=1e-12, atol=1e-18), k
    assert a["zone_names"] == g["zone_names"]


@pytest.mark.parametrize("split", ["checker", "all", "none"])
def test_mixed_prism_hex_mesh_matches_oracle_bitwise(oracle, tmp_path, split):
    """Triangular + quadrilateral faces, prisms + hexahedra, skewed nodes (tests/meshgen.py)."""
    import meshgen
    path = str(tmp_path / "mixed.msh")
    info = meshgen.write_mixed_channel_msh(path, 5, 4, 3, split=split, skew=0.2)
    d = orc_io.read_mesh(path)
    assert (d.n_cells, d.n_faces, d.n_vertices) == (info["n_cells"], info["n_faces"], info["n_nodes"])
    a = d.arrays()
    assert_same_mesh(a, oracle.Mesh.read(path).arrays())
    assert a["zone_names"] == info["zone_names"]
    out = a["face_centroid"] - a["cell_centroid"][a["face_c0"]]
    assert np.all(np.einsum("ij,ij->i", out, a["face_normal"]) > 0)
    assert np.isclose(a["cell_volume"].sum(), 0.002 * 0.001 * 3e-4, rtol=1e-12)


def test_zone_lookup_and_assignment(mesh_path):
    d = orc_io.read_mesh(mesh_path("couette_flow_8x8x1"))
    names = [z[4] for z in d.zones()]
    assert "INLET" in names and "OUTLET" in names
    from orc_amd.settings import FaceConditionTypes as T
    d.set_zone("INLET", T.PressureInlet, -0.01)
    d.set_zone("TOP_WALL" if "TOP_WALL" in names else "WALL", T.Wall, 0.0, (5e-4, 0.0, 0.0))
    z = {t[4]: t for t in d.zones()}
    assert z["INLET"][1] == T.PressureInlet and z["INLET"][2] == -0.01
    with pytest.raises(KeyError):
        d.get_face_zone("NO_SUCH_ZONE")
    with pytest.raises(OrcError) as e:  # mesh.rs:194 panic
        d.set_zone("NO_SUCH_ZONE", T.Wall)
    assert e.value.status == 16 and "NO_SUCH_ZONE" in str(e.value)


CUBE = """(0 "one hexahedron")
(2 3)
(10 (0 1 8 0 3))
(10 (1 1 8 1 3)(
0 0 0
1 0 0
1 1 0
0 1 0
0 0 1
1 0 1
1 1 1
0 1 1
))
(12 (0 1 2 0))
(12 (2 1 2 1 4))
(13 (0 1 b 0))
(0 "Interior faces of zone FLUID")
(13 (3 1 1 2 4)(
8 7 6 5 1 2
))
(0 "Faces of zone OUTER")
(13 (a 2 b 3 4)(
2 3 4 1 1 0
5 6 2 1 1 0
6 7 3 2 1 0
7 8 4 3 1 0
8 5 1 4 1 0
c b a 9 2 0
9 a 6 5 2 0
a b 7 6 2 0
b c 8 7 2 0
c 9 5 8 2 0
))
"""


def cube_file(tmp_path, text=CUBE, extra_nodes=True):
    if extra_nodes:  # second layer of nodes 9..c for the upper cell
        text = text.replace("(10 (0 1 8 0 3))\n(10 (1 1 8 1 3)(", "(10 (0 1 c 0 3))\n(10 (1 1 c 1 3)(").replace(
            "0 1 1\n))", "0 1 1\n0 0 2\n1 0 2\n1 1 2\n0 1 2\n))")
    p = tmp_path / "cube.msh"
    p.write_text(text)
    return str(p)


def test_reader_hexadecimal_items_comments_and_zone_zero(oracle, tmp_path):
    """io.rs:47-54 (items are hex: zone id 'a' = 10, faces 2..b), io.rs:83-90 (zone named by the comment's last word),
    io.rs:24-30 (zone-0 declarations skipped)."""
    path = cube_file(tmp_path)
    d = orc_io.read_mesh(path)
    assert (d.n_cells, d.n_faces, d.n_vertices) == (2, 11, 12)
    z = d.zones()
    assert [(t[0], t[1], t[4]) for t in z] == [(3, 2, "FLUID"), (10, 3, "OUTER")]
    a = d.arrays()
    assert_same_mesh(a, oracle.Mesh.read(path).arrays())
    assert np.allclose(a["cell_volume"], 1.0) and np.allclose(a["face_area"], 1.0)
    assert a["face_c1"][0] == 1 and np.all(a["face_c1"][1:] == -1)
    # boundary faces: normal outward from their only cell; the interior face: outward from c0
    out = a["face_centroid"] - a["cell_centroid"][a["face_c0"]]
    assert np.all(np.einsum("ij,ij->i", out, a["face_normal"]) > 0)


def test_reader_crlf_and_missing_final_newline(tmp_path):
    path = cube_file(tmp_path)
    ref = orc_io.read_mesh(path).arrays()
    txt = open(path).read().replace("\n", "\r\n").rstrip("\r\n")
    p2 = tmp_path / "crlf.msh"
    p2.write_bytes(txt.encode())
    assert_same_mesh(orc_io.read_mesh(str(p2)).arrays(), ref)


def test_reader_polygonal_zone_keeps_count_prefix_as_node(oracle, tmp_path):
    """io.rs:232: for face type 0/5 the per-line node count is read as a node (SURVEY §8f-1 hazard) — mirrored, not fixed."""
    txt = CUBE.replace("(13 (a 2 b 3 4)(", "(13 (a 2 b 3 5)(")
    for ln in ["2 3 4 1 1 0", "5 6 2 1 1 0", "6 7 3 2 1 0", "7 8 4 3 1 0", "8 5 1 4 1 0", "c b a 9 2 0", "9 a 6 5 2 0", "a b 7 6 2 0",
               "b c 8 7 2 0", "c 9 5 8 2 0"]:
        txt = txt.replace(ln + "\n", "4 " + ln + "\n")
    path = cube_file(tmp_path, txt)
    d = orc_io.read_mesh(path)
    _, ptr, idx = d.nodes()
    assert np.all(np.diff(ptr)[1:] == 5) and np.all(idx[ptr[1:-1]] == 3)  # "4" - 1
    assert_same_mesh(d.arrays(), oracle.Mesh.read(path).arrays())


@pytest.mark.parametrize("mutation, message", [
    (("(2 3)", "(2 4)"), "Mesh is not 2D or 3D."),
    (("(13 (a 2 b 3 4)(", "(13 (a 2 b 6 4)("), "valid BC type"),
    (("(13 (a 2 b 3 4)(", "(13 (a 2 b 3 4 1)("), "six items"),
    (("(13 (a 2 b 3 4)(", "(13 (g 2 b 3 4)("), "valid hex"),
    (("1 0 0\n", "1 zero 0\n"), "float"),
    (("8 7 6 5 1 2", "5 6 7 d 1 2"), "nodes should have all been read"),
    (("(13 (a 2 b 3 4)(", "(13 (a 3 c 3 4)("), "not contiguous"),  # second zone starts at face 3: face 2 never defined
    (("(0 \"one hexahedron\")", "(0"), "comment has a space"),
])
def test_reader_panic_sites_become_status(tmp_path, mutation, message):
    txt = CUBE
    assert mutation[0] in txt
    txt = txt.replace(*mutation)
    path = cube_file(tmp_path, txt)
    with pytest.raises(OrcError) as e:
        orc_io.read_mesh(path)
    assert e.value.status == 15, str(e.value)
    assert message in str(e.value), str(e.value)


def test_reader_missing_file_is_io_error(tmp_path):
    with pytest.raises(OrcError) as e:
        orc_io.read_mesh(str(tmp_path / "absent.msh"))
    assert e.value.status == 13 and "Unable to open mesh file for reading." in str(e.value)


# ------------------------------------------------------------------ text checkpoint formats
def rust_lower_exp(x, precision=None):
    """Independent restatement of core::fmt::LowerExp for f64 built on Python's shortest repr / correctly rounded %e."""
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "-inf" if x < 0 else "inf"
    if precision is None:
        r = repr(float(x))
        sign = "-" if r.startswith("-") else ""
        r = r.lstrip("-")
        mant, _, ex = r.partition("e")
        ex = int(ex) if ex else 0
        ip, _, fp = mant.partition(".")
        digits = (ip + fp)
        point = len(ip)  # digits[:point] . digits[point:]
        stripped = digits.lstrip("0")
        if not stripped:
            return sign + "0e0"
        lead = len(digits) - len(stripped)
        exp10 = ex + point - lead - 1
        stripped = stripped.rstrip("0") or "0"
        body = stripped[0] + ("." + stripped[1:] if len(stripped) > 1 else "")
        return "%s%se%d" % (sign, body, exp10)
    s = "%.*e" % (precision, x)
    mant, ex = s.split("e")
    return "%se%d" % (mant, int(ex))


SAMPLES = [0.0, -0.0, 1.0, -1.0, 1.5, 0.1, 1e-5, 1.2345678901234567e-8, 123456.789, 1e21, 1e22, 5e-324, 1.7976931348623157e308,
           2.5e-3, -3.0000000000000004e-4, 1 / 3, 100.0, 1e15, 1e16, 123e-20, float("inf"), float("-inf"), float("nan")]


def test_write_data_is_rust_lower_exp(tmp_path):
    n = len(SAMPLES)
    rng = np.random.default_rng(7)
    cc = rng.uniform(-1e-3, 2e-3, (n, 3))
    cc[0] = [0.0, 1.125e-3, -9.995e-4]
    u = np.array(SAMPLES)
    v, w, p = u[::-1].copy(), rng.standard_normal(n) * 1e-4, rng.standard_normal(n) * 10 ** rng.uniform(-12, 12, n)
    path = str(tmp_path / "out.csv")
    orc_io.write_data(cc, u, v, w, p, path)
    lines = open(path).read().split("\n")
    assert lines[-1] == "" and len(lines) == n + 1
    for i in range(n):
        want = "(%s, %s, %s)\t(%s, %s, %s)\t%s" % tuple(
            [rust_lower_exp(c, 2) for c in cc[i]] + [rust_lower_exp(x) for x in (u[i], v[i], w[i], p[i])])
        assert lines[i] == want, (i, lines[i], want)
    assert lines[1].count("\t") == 2
    # literal spot checks of the format (no '+', no exponent padding, integer mantissa without '.')
    assert lines[0].split("\t")[1].startswith("(0e0, ")
    assert lines[2].split("\t")[1].startswith("(1e0, ")
    orc_io.write_data(cc, u, v, w, p, path, decimal_precision=4)
    first = open(path).readline().rstrip("\n")
    assert first.split("\t")[1].startswith("(0.0000e0, ") and first.endswith(rust_lower_exp(p[0], 4))


def test_write_then_read_data_round_trips_exactly(tmp_path):
    rng = np.random.default_rng(3)
    n = 500
    cc = rng.uniform(0, 1, (n, 3))
    u, v, w = (rng.standard_normal(n) * 10 ** rng.uniform(-8, 3, n) for _ in range(3))
    p = rng.standard_normal(n)
    path = str(tmp_path / "state.csv")
    orc_io.write_data(cc, u, v, w, p, path)
    back = orc_io.read_data(path)
    for a, b in zip((u, v, w, p), back):
        assert np.array_equal(a, b)  # shortest round-trip digits: resume is lossless (SURVEY §5 checkpoint/resume)
    orc_io.write_data(cc, u, v, w, p, path, decimal_precision=3)
    for a, b in zip((u, v, w, p), orc_io.read_data(path)):
        assert np.allclose(a, b, rtol=6e-4, atol=0)


def test_read_data_accepts_reference_style_lines_and_rejects_garbage(tmp_path):
    path = tmp_path / "d.csv"
    path.write_text("(1.00e-3, 2.00e-3, 0.00e0)\t(1e0, -2.5e-3, 0e0)\t1.25e2\n"
                    "(1.00e-3, 2.00e-3, 0.00e0)\t(inf, NaN, 3)\t-7\n"
                    "no tab on this line\n")
    u, v, w, p = orc_io.read_data(str(path))
    assert u[0] == 1.0 and v[0] == -2.5e-3 and w[0] == 0.0 and p[0] == 125.0
    assert np.isinf(u[1]) and np.isnan(v[1]) and w[1] == 3.0 and p[1] == -7.0 and len(u) == 2
    path.write_text("(0, 0, 0)\t(1, 2)\t3\n")
    with pytest.raises(OrcError):
        orc_io.read_data(str(path))
    with pytest.raises(OrcError) as e:
        orc_io.read_data(str(tmp_path / "absent.csv"))
    assert e.value.status == 13 and "could not read data file" in str(e.value)


def test_mixed_channel_generator_is_conforming_and_readable(oracle, tmp_path):
    """orc_mixed_channel_write_msh (BASELINE config 5 workload): tetrahedra, pyramids, prisms and hexahedra in one conforming
    mesh — every interior face has exactly two cells (the writer rejects an unpaired one), every cell is closed, the volumes
    fill the box, triangular and quadrilateral faces sit in separate zones, and the product reader and the oracle's reader
    build identical meshes from the file."""
    from orc_amd import io as orc_io
    from orc_amd.mesh import write_mixed_channel_msh
    path = str(tmp_path / "mixed.msh")
    nx, ny, nz = 24, 5, 4
    nc, nf = write_mixed_channel_msh(path, nx, ny, nz)
    d = orc_io.read_mesh(path)
    a = d.arrays()
    om = oracle.Mesh.read(path)
    ao = om.arrays()
    assert len(a["cell_volume"]) == nc == om.n_cells and len(a["face_area"]) == nf == om.n_faces
    for k in ("face_c0", "face_c1", "face_zone", "face_area", "face_normal", "face_centroid", "cell_centroid", "cell_volume", "cell_face_ptr", "cell_faces"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(ao[k])), k
    nfc = np.diff(a["cell_face_ptr"])
    assert set(np.unique(nfc).tolist()) == {4, 5, 6}
    vol = np.asarray(a["cell_volume"])
    assert vol.min() > 0 and abs(vol.sum() - 0.002 * 0.001 * 1e-4 * nz) < 1e-13 * vol.sum()
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    An = np.asarray(a["face_normal"]) * np.asarray(a["face_area"])[:, None]
    S = np.zeros((nc, 3))
    np.add.at(S, c0, An)
    m = c1 >= 0
    np.add.at(S, c1[m], -An[m])
    assert np.abs(S).max() < 1e-20
    names = a["zone_names"]
    assert {"FLUID", "FLUID_TRI", "INLET", "OUTLET", "WALL", "WALL_TRI", "PERIODIC_-Z", "PERIODIC_-Z_TRI", "PERIODIC_+Z", "PERIODIC_+Z_TRI"} <= set(names)
    # faces of a "_TRI" zone have three nodes, the others four (the reference cannot parse mixed zones, io.rs:232)
    fz = np.asarray(a["face_zone"])
    _, fnp, _ = d.nodes()
    nn = np.diff(fnp)
    for zi, name in enumerate(a["zone_names"]):
        sel = nn[fz == zi]
        assert len(sel) == 0 or set(np.unique(sel).tolist()) == ({3} if name.endswith("_TRI") else {4}), name
    with pytest.raises(Exception):
        write_mixed_channel_msh(path, 10, 3, 2)  # fewer than 20 blocks along x: the regions would collapse


def test_poly_channel_generator_agglomerated_polyhedra(oracle, tmp_path):
    """orc_poly_channel_write_msh (BASELINE config 5: "tet/hex/poly"): the mixed channel with a region of agglomerated
    polyhedral cells — rhombic dodecahedra of 12 quadrilateral faces in the region's interior — written through triangular
    and quadrilateral face sections only, which is what the reference's reader parses (io.rs:232-233).  Both readers agree
    bit for bit, every cell is closed, the volumes fill the box, and no two cells share more than one face (the matrix
    pattern is "diagonal + one entry per interior face", discretization.rs:312-322)."""
    from orc_amd import io as orc_io
    from orc_amd.mesh import write_mixed_channel_msh
    path = str(tmp_path / "poly.msh")
    nx, ny, nz = 40, 6, 5
    nc, nf = write_mixed_channel_msh(path, nx, ny, nz, polyhedra=True)
    d = orc_io.read_mesh(path)
    a = d.arrays()
    om = oracle.Mesh.read(path)
    ao = om.arrays()
    assert len(a["cell_volume"]) == nc == om.n_cells and len(a["face_area"]) == nf == om.n_faces
    for k in ("face_c0", "face_c1", "face_zone", "face_area", "face_normal", "face_centroid", "cell_centroid", "cell_volume", "cell_face_ptr", "cell_faces"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(ao[k])), k
    nfc = np.diff(a["cell_face_ptr"])
    counts = dict(zip(*[x.tolist() for x in np.unique(nfc, return_counts=True)]))
    # full polyhedra (12 rhombi), partial ones along the region's sides and the walls (a quadrilateral + triangles instead of
    # the rhombi of the missing pyramids: 13 faces), and the four classic shapes
    assert counts.get(12, 0) > 0 and max(counts) == 13 and {4, 5, 6} <= set(counts)
    vol = np.asarray(a["cell_volume"])
    assert vol.min() > 0 and abs(vol.sum() - 0.002 * 0.001 * 1e-4 * nz) < 1e-13 * vol.sum()
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    An = np.asarray(a["face_normal"]) * np.asarray(a["face_area"])[:, None]
    S = np.zeros((nc, 3))
    np.add.at(S, c0, An)
    m = c1 >= 0
    np.add.at(S, c1[m], -An[m])
    assert np.abs(S).max() < 1e-20
    pairs = np.stack([np.minimum(c0[m], c1[m]), np.maximum(c0[m], c1[m])], axis=1)
    assert len(np.unique(pairs, axis=0)) == len(pairs)  # one face per pair of neighbours
    # the 12-face cells: all faces planar quadrilaterals of equal area (rhombi), volume = two blocks' worth
    fz = np.asarray(a["face_zone"])
    _, fnp, _ = d.nodes()
    nn = np.diff(fnp)
    cfp, cf = np.asarray(a["cell_face_ptr"]), np.asarray(a["cell_faces"])
    block = 0.002 / nx * 0.001 / ny * 1e-4
    full = 0
    for c in np.nonzero(nfc == 12)[0]:
        faces = cf[cfp[c]:cfp[c + 1]]
        if set(nn[faces].tolist()) == {4} and np.all(c1[faces] >= 0):  # a whole rhombic dodecahedron: hexahedron + six pyramids
            full += 1
            assert abs(vol[c] - 2 * block) < 1e-12 * block
    assert full > 0
    for zi, name in enumerate(a["zone_names"]):
        sel = nn[fz == zi]
        assert len(sel) == 0 or set(np.unique(sel).tolist()) == ({3} if name.endswith("_TRI") else {4}), name


@pytest.mark.parametrize("polyhedra", [False, True])
def test_mixed_channel_generated_in_memory_equals_the_written_and_read_file(tmp_path, polyhedra):
    """[r05] orc_mixed_channel_generate (what bench.py --workload config5 uses per rank): the generator's nodes and faces handed to the reader's
    own geometry step, no file — every array of the handle (adjacency, zones, areas, normals, centroids, volumes, cell face lists) has the bits
    of orc_read_mesh on the file orc_mixed_channel_write_msh / orc_poly_channel_write_msh writes for the same arguments."""
    from orc_amd import io as orc_io
    from orc_amd.mesh import write_mixed_channel_msh
    nx, ny, nz = 60, 12, 10
    path = str(tmp_path / "mixed.msh")
    write_mixed_channel_msh(path, nx, ny, nz, polyhedra=polyhedra)
    a = orc_io.read_mesh(path).arrays()
    d = orc_io.MeshData.mixed_channel(nx, ny, nz, polyhedra=polyhedra)
    b = d.arrays()
    assert set(a) == set(b)
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k
    assert d.n_vertices > (nx + 1) * (ny + 1) * (nz + 1)  # (block-centre nodes of the pyramid blocks)
