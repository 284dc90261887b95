"""Which ORC_XWIN_CAP / ORC_XWIN_BITWORDS values send what share of the test-size coarse blocks through the no-window branches
(used once to pick the parameters of tests/test_gpu_window_fallback.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import orc_amd
from conftest import fv_like_matrix, splitmix64_uniform
from orc_amd.linear_algebra import amg_coarse_product, amg_coarsen, xwin_counters

orc_amd.init(0)
a = fv_like_matrix(64, 40, 12)
levels = [a]
for _ in range(3):
    levels.append(amg_coarsen(levels[-1])[1])
for lv in (1, 2):
    fine, coarse = levels[lv], levels[lv + 1]
    print("level", lv + 1, "rows", coarse.shape[0], "nnz/row %.1f" % (coarse.nnz / coarse.shape[0]))
    x = splitmix64_uniform(coarse.shape[0], 3)
    for env, vals in (("ORC_XWIN_CAP", [5000, 2000, 1500, 1000, 700, 500, 300, 100, 40]), ("ORC_XWIN_BITWORDS", [8192, 200, 100, 60, 40, 20, 10])):
        for v in vals:
            os.environ[env] = str(v)
            xwin_counters(reset=True)
            y, m = amg_coarse_product(fine, x)
            print("   %s=%d mirror=%s counters=%s" % (env, v, m, xwin_counters()))
        del os.environ[env]
