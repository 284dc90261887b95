"""Mixed-cell TGRID fixtures for the parity tests (BASELINE config 5 in miniature): an extruded channel whose columns are
hexahedra or pairs of triangular prisms, optionally with skewed (non-orthogonal) node positions.  Written as a Fluent
ASCII .msh in the subset ORC reads (io.rs:32-515): one face zone per (location, face type) because the reference's
reader cannot parse mixed zones (io.rs:232).  Faces are oriented so that the reader's normal points out of cell 0."""
import numpy as np


def write_mixed_channel_msh(path, nx=6, ny=4, nz=3, lx=0.002, ly=0.001, lz=3e-4, skew=0.15, seed=11, split="checker"):
    """Returns a dict of counts.  split: "checker" (alternate columns are prisms), "all" (prisms only), "none" (hex only)."""
    rng = np.random.default_rng(seed)
    dx, dy, dz = lx / nx, ly / ny, lz / nz
    # nodes: jitter in x, y identical on every z level so that every face stays planar
    jx = np.zeros((nx + 1, ny + 1))
    jy = np.zeros((nx + 1, ny + 1))
    jx[1:-1, 1:-1] = rng.uniform(-skew, skew, (nx - 1, ny - 1)) * dx
    jy[1:-1, 1:-1] = rng.uniform(-skew, skew, (nx - 1, ny - 1)) * dy

    def nid(i, j, k):
        return i + (nx + 1) * (j + (ny + 1) * k)

    coords = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    for k in range(nz + 1):
        for j in range(ny + 1):
            for i in range(nx + 1):
                coords[nid(i, j, k)] = (i * dx + jx[i, j], j * dy + jy[i, j], k * dz)

    def is_split(i, j):
        return split == "all" or (split == "checker" and (i + j) % 2 == 0)

    faces = {}  # key: sorted node tuple -> [node list (as first seen), [cells]]
    n_cells = 0

    def add_face(nodes, cell):
        key = tuple(sorted(nodes))
        if key in faces:
            faces[key][1].append(cell)
        else:
            faces[key] = [list(nodes), [cell]]

    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                b = [nid(i, j, k), nid(i + 1, j, k), nid(i + 1, j + 1, k), nid(i, j + 1, k)]
                t = [nid(i, j, k + 1), nid(i + 1, j, k + 1), nid(i + 1, j + 1, k + 1), nid(i, j + 1, k + 1)]
                if is_split(i, j):
                    for tri in ((0, 1, 2), (0, 2, 3)):
                        c = n_cells
                        n_cells += 1
                        add_face([b[q] for q in tri], c)
                        add_face([t[q] for q in tri], c)
                        for e in range(3):
                            p, q = tri[e], tri[(e + 1) % 3]
                            add_face([b[p], b[q], t[q], t[p]], c)
                else:
                    c = n_cells
                    n_cells += 1
                    add_face(b, c)
                    add_face(t, c)
                    for e in range(4):
                        p, q = e, (e + 1) % 4
                        add_face([b[p], b[q], t[q], t[p]], c)

    # provisional cell centres (mean of the cell's nodes) to orient the faces
    csum = np.zeros((n_cells, 3))
    ccnt = np.zeros(n_cells)
    for nodes, cells in faces.values():
        for c in cells:
            csum[c] += coords[nodes].sum(axis=0)
            ccnt[c] += len(nodes)
    ccen = csum / ccnt[:, None]

    eps = 1e-12

    def location(nodes):
        p = coords[nodes]
        if np.all(p[:, 2] < eps):
            return "PERIODIC_-Z"
        if np.all(p[:, 2] > lz - eps):
            return "PERIODIC_+Z"
        if np.all(np.abs(p[:, 0]) < eps):
            return "INLET"
        if np.all(np.abs(p[:, 0] - lx) < eps):
            return "OUTLET"
        if np.all(np.abs(p[:, 1]) < eps) or np.all(np.abs(p[:, 1] - ly) < eps):
            return "WALL"
        return None

    zones = {}  # (name, n_nodes) -> list of (nodes, c0, c1)
    for nodes, cells in faces.values():
        p = coords[nodes]
        nrm = np.cross(p[2] - p[1], p[1] - p[0])  # io.rs:323-325
        c0 = cells[0]
        fc = p.mean(axis=0)
        if np.dot(fc - ccen[c0], nrm) < 0:  # make the reader's normal point out of c0
            nodes = nodes[::-1]
        if len(cells) == 2:
            name, c1 = "FLUID", cells[1]
        else:
            name, c1 = location(nodes), -1
            assert name is not None, "boundary face not on the box"
        if len(nodes) == 3 and name != "FLUID":
            name = name + "_TRI"
        if len(nodes) == 3 and name == "FLUID":
            name = "FLUID_TRI"
        zones.setdefault((name, len(nodes)), []).append((nodes, c0, c1))

    order = sorted(zones, key=lambda z: (0 if z[0].startswith("FLUID") else 1, z[0]))
    n_faces = sum(len(v) for v in zones.values())
    with open(path, "w") as f:
        f.write('(0 "mixed prism/hex channel for the parity tests")\n(2 3)\n')
        f.write("(10 (0 1 %x 0 3))\n(10 (1 1 %x 1 3)(\n" % (len(coords), len(coords)))
        for x, y, z in coords:
            f.write("%.17g %.17g %.17g\n" % (x, y, z))
        f.write("))\n")
        f.write("(12 (0 1 %x 0))\n(12 (2 1 %x 1 0))\n" % (n_cells, n_cells))
        f.write("(13 (0 1 %x 0))\n" % n_faces)
        start = 1
        for zi, key in enumerate(order):
            name, nn = key
            recs = zones[key]
            bc = 2 if name.startswith("FLUID") else 3
            f.write('(0 "Faces of zone %s")\n' % name)
            f.write("(13 (%x %x %x %x %x)(\n" % (zi + 3, start, start + len(recs) - 1, bc, nn))
            for nodes, c0, c1 in recs:
                f.write(" ".join("%x" % (n + 1) for n in nodes) + " %x %x\n" % (c0 + 1, c1 + 1))
            f.write("))\n")
            start += len(recs)
    return dict(n_cells=n_cells, n_faces=n_faces, n_nodes=len(coords), zone_names=[k[0] for k in order])


def mixed_channel_bcs(set_zone, zone_names, top_wall_velocity=0.0, dp=0.01):
    """tests.rs:60-76 on the fixture's zones; set_zone(name, type, scalar, vector)."""
    for name in zone_names:
        base = name[:-4] if name.endswith("_TRI") else name
        if base == "FLUID":
            continue
        if base == "WALL":
            set_zone(name, 3, 0.0, (top_wall_velocity, 0.0, 0.0))
        elif base == "INLET":
            set_zone(name, 4, dp, (0.0, 0.0, 0.0))
        elif base == "OUTLET":
            set_zone(name, 5, 0.0, (0.0, 0.0, 0.0))
        else:
            set_zone(name, 7, 0.0, (0.0, 0.0, 0.0))


def shuffle_cells(a, seed=1):
    """MeshArrays with the cells renumbered by a seeded random permutation (an "arbitrarily numbered" mesh)."""
    from orc_amd.mesh import renumber_cells
    n = len(a["cell_volume"])
    return renumber_cells(a, np.random.default_rng(seed).permutation(n))
