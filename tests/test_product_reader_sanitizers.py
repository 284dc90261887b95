"""AddressSanitizer + UBSan over the product's host-side code (orc_amd/csrc/mesh_io.cpp, partition.cpp, mesh_gen.cpp,
compiled with g++ and the device entry points stubbed): every reference mesh, truncated and corrupted variants, the
checkpoint formats, the partitioner in every ordering at 1 / 2 / 5 ranks, the mixed-element generator."""
import os
import subprocess

import pytest

from conftest import ROOT
from test_io_cpu import CUBE, REFERENCE_MESHES, cube_file


@pytest.fixture(scope="module")
def reader_binary(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("asan_reader") / "sanitize_reader")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "orc_amd", "csrc"), os.path.join(ROOT, "tests", "sanitize_reader_main.cpp"),
           os.path.join(ROOT, "orc_amd", "csrc", "mesh_io.cpp"), os.path.join(ROOT, "orc_amd", "csrc", "partition.cpp"),
           os.path.join(ROOT, "orc_amd", "csrc", "mesh_gen.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_reader_and_checkpoint_formats_under_asan_ubsan(reader_binary, mesh_path, tmp_path):
    good = [mesh_path(m) for m in REFERENCE_MESHES] + [cube_file(tmp_path)]
    bad = []
    text = open(cube_file(tmp_path)).read()
    variants = {
        "truncated_nodes": text[: text.index("1 1 0")],
        "truncated_faces": text[: text.index("6 7 3 2")],
        "no_dimension": text.replace("(2 3)\n", ""),
        "huge_index": text.replace("8 7 6 5 1 2", "8 7 6 7fffffff 1 2"),
        "zero_cells": text.replace("8 7 6 5 1 2", "8 7 6 5 0 0"),
        "garbage": "(((((\n" + "\x00\xff" * 50 + "\n(13 (",
        "empty": "",
    }
    # like the reference, the reader accepts a file that ends inside its node section (a mesh without faces or cells) and
    # a face whose two cell numbers are both 0 (it belongs to no cell): io.rs:289-415 has no panic for either
    accepted = {"truncated_nodes", "zero_cells"}
    for name, body in variants.items():
        p = tmp_path / (name + ".msh")
        p.write_bytes(body.encode("latin-1"))
        (good if name in accepted else bad).append(str(p) if name in accepted else "!" + str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([reader_binary] + good + bad, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sanitize_reader ok" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
