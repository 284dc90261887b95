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
    # (the GPU modes put several ranks on ONE card through the host-staged transport; r04 capped their hardware queues here — the library now
    # keeps such ranks' streams in one priority class by itself: runtime.cpp stream_create)
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


@pytest.mark.parametrize("world", [2, 4])
def test_general_partitioner_on_read_meshes_gloo(world):
    """orc_mesh_partition (C ABI, host only) on channel_flow.msh and the prism + hexahedron mesh, ORC / RCM / geometric
    orderings, world sizes 2 and 4: tiling, face lists, halo symmetry (real gloo exchange), distributed product."""
    r = launch(world, "cpu_general")
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_general_partitioner_matches_the_slab_generator():
    """On the structured channel in ORC order the general partitioner reproduces slab_arrays' topology and halo plan."""
    import numpy as np
    from orc_amd import parallel
    from orc_amd.mesh import hex_channel
    g = hex_channel(6, 5, 9)
    for rank in range(3):
        a, halo, gids = parallel.partition_arrays(g, 3, rank)
        b, hb, gb = parallel.slab_arrays(6, 5, 3, rank, 3)
        assert np.array_equal(gids, gb)
        for k in ("face_c0", "face_c1", "face_zone", "cell_face_ptr", "cell_faces"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        for k in ("peers", "send_ptr", "send_idx", "recv_ptr"):
            assert np.array_equal(halo[k], hb[k]), k


def test_rcm_ordering_reduces_the_bandwidth_of_a_shuffled_mesh():
    """n_ranks = 1 with ORC_ORDER_RCM = the north star's RCM row ordering: on a randomly renumbered channel the matrix
    bandwidth max|i - j| comes back to the order of the structured numbering's."""
    import numpy as np
    from orc_amd import parallel
    from orc_amd.mesh import hex_channel
    from meshgen import shuffle_cells
    g = hex_channel(12, 10, 8)
    sh = shuffle_cells(g, seed=5)
    def bandwidth(a):
        c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
        m = c1 >= 0
        return int(np.abs(c0[m] - c1[m]).max())
    a, halo, gids = parallel.partition_arrays(sh, 1, 0, parallel.ORDER_RCM)
    assert sorted(gids.tolist()) == list(range(len(gids))) and halo["n_owned"] == len(gids) and len(halo["peers"]) == 0
    assert bandwidth(sh) > 5 * bandwidth(g)
    assert bandwidth(a) <= 2 * bandwidth(g)


@pytest.mark.parametrize("world", [2, 8])
def test_rank_local_generation_of_the_mixed_poly_channel_gloo(world):
    """BASELINE configs[4] (mixed tet / hex / poly, partitioned) without the whole mesh in any process: every rank generates its
    slab + two ghost block layers per inner side and cuts its part out with orc_mesh_partition_owner.  World sizes 2 and 8 (the
    config's own) on a small mesh: tiling, complete ghost geometry, face lists, halo symmetry by really exchanging global ids over
    gloo, distributed product == global product."""
    r = launch(world, "cpu_mixed_slabs")
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
