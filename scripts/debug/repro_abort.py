import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import orc_amd
from oracle import pyoracle as po
from orc_amd.mesh import Mesh, MeshArrays
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver
orc_amd.init(0)
om = po.Mesh.read(os.path.join(ROOT, "tests/golden/meshes/channel_flow.msh"))
H.channel_bcs(om)
a = MeshArrays(om.arrays())
dm = Mesh(a)
u, v, w, p = H.seeded_fields(a, seed=4, scale_u=4e-4)
for k in range(2):
    s = Solver(dm, NumericalSettings.default(iterations=10), 1000.0, 1e-3)
    s.set_fields(u, v, w, p)
    for it in range(4):
        st = s.iterate(1, raise_on_error=False)
        print("run", k, "iteration", it, "status", st, flush=True)
print("done")
