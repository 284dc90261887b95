"""AddressSanitizer + UBSan over the CPU oracle (no GPU sanitizers on this pool, so the memory-safety check
runs on the restatement): mesh reader, flow initialisation and SIMPLE iterations with every solver arm."""
import os
import subprocess

import pytest

from conftest import ROOT

ORACLE = os.path.join(ROOT, "oracle")
SRC = ["sanitize_main.c", "sparse.c", "mesh_io.c", "linear_algebra.c", "discretization.c", "solver.c"]


@pytest.fixture(scope="module")
def sanitized_binary(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("asan") / "sanitize_main")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-o", exe] + [os.path.join(ORACLE, f) for f in SRC] + ["-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("libasan is not installed in this image")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("stem, walls", [("couette_flow_8x8x1", ["WALL"]), ("channel_flow", ["WALL"])])
def test_oracle_under_asan_ubsan(sanitized_binary, mesh_path, stem, walls):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sanitized_binary, mesh_path(stem)] + walls, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitize_main ok" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
