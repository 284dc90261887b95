"""Debug probe: one lock-step Multigrid solve of three near-identical systems with ORC_AMG_TRACE=1; prints the shared-pass counter."""
import os, sys
os.environ["ORC_AMG_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from test_gpu_triple import three_systems
import orc_amd
from orc_amd.linear_algebra import iterative_solve3, shared_galerkin
orc_amd.init(0)
mats, bs, xs = three_systems((20, 17, 9))
shared_galerkin(reset=True)
st, st3 = iterative_solve3(mats, bs, [x.copy() for x in xs], 6, 2, 0.5, 1e-3, 1)
print("status", st, st3, "shared operators", shared_galerkin())
