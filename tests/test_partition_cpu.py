"""The N > 1 path's host logic on CPU: world_size-2 and -3 gloo runs of tests/mp_worker.py (slab construction, halo plan,
the gloo exchange that doubles as liborc_amd's debug transport, distributed SpMV == global SpMV)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(nproc, mode, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "mp_worker.py"), mode]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("world", [2, 3])
def test_slab_partition_and_halo_exchange_gloo(world):
    r = launch(world, "cpu")
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_slab_arrays_single_rank_is_the_whole_mesh():
    import numpy as np
    from orc_amd.mesh import hex_channel
    from orc_amd.parallel import slab_arrays
    a, halo, gids = slab_arrays(5, 4, 3, 0, 1)
    g = hex_channel(5, 4, 3)
    assert halo["n_owned"] == 60 and len(halo["peers"]) == 0 and np.array_equal(gids, np.arange(60))
    for k in ("face_c0", "face_c1", "face_zone", "cell_face_ptr", "cell_faces", "face_area", "cell_volume"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(g[k])), k
