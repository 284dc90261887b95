"""Two ranks sharing cuda:0 through the host-staged debug transport: partitioned SIMPLE iterations vs the single-rank run
(Jacobi solver: bit-exact; BiCGSTAB: 1e-9; Multigrid with per-rank coarse levels: same answer to a few percent)."""
import pytest

from test_partition_cpu import launch

pytestmark = pytest.mark.gpu


def test_two_ranks_match_single_rank(gpu):
    r = launch(2, "gpu", timeout=900)
    print(r.stdout[-1500:])
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_general_partitioner_two_ranks_match_single_rank(gpu):
    """orc_mesh_partition (geometric and RCM orderings) on the read prism + hexahedron mesh: two ranks on one GPU."""
    r = launch(2, "gpu_general", timeout=900)
    print(r.stdout[-1500:])
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_overlapped_level0_product_two_ranks(gpu):
    """Slabs of 260 slices per rank: the level-0 products run their interior slices beside the halo exchange; fields against
    the same partitioned run without the overlap (1e-11 / 1e-8: only the layout of the partial sums differs), against the
    single-rank run, and the overlap counter > 0."""
    r = launch(2, "gpu_overlap", timeout=900)
    print(r.stdout[-1500:])
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_two_ranks_converged_default_stack_matches_the_oracle(gpu):
    """channel_flow.msh cut in two by orc_mesh_partition (ORC order and RCM), the reference's default stack run partitioned to
    convergence: u, v, w, p within 1e-6 rel-L2 of the oracle's converged fields (the north-star criterion at N = 2)."""
    r = launch(2, "gpu_converged", timeout=1500)
    print(r.stdout[-1500:])
    assert "MP_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_rccl_selftest_single_rank(gpu):
    """The RCCL branch of HaloPlan::exchange / all-reduce / status agreement on one GPU: a single-rank communicator
    with rank 0 as its own neighbour.  (Two real ranks need two GPUs: the driver's scaling run.)"""
    from orc_amd._lib import check, lib
    check(lib().orc_comm_selftest())
