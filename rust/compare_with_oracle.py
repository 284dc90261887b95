#!/usr/bin/env python3
"""Compares a dump written by rust/dump_golden.rs (real ORC) with the oracle's dump of the same inputs, file by file, BIT FOR BIT
(NaNs must sit in the same places; status texts must agree on ok / not ok).  Exit code 0 = every file present on both sides is
identical.  This is the check that turns DESIGN.md's "parity unpinned" (assembly, Multigrid arm, solve_steady) into a pin.
    python rust/compare_with_oracle.py tests/golden/reference_dump [rust/inputs]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust"))


def compare_dirs(ref_dir, oracle_dir, out=sys.stdout):
    bad, seen, missing = [], 0, []
    for base, _, files in os.walk(oracle_dir):
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(base, f), oracle_dir)
            rp = os.path.join(ref_dir, rel)
            if not os.path.exists(rp):
                missing.append(rel)
                continue
            seen += 1
            if f.endswith(".txt"):
                a, b = open(rp).read().strip(), open(os.path.join(base, f)).read().strip()
                same = (a == "ok") == (b == "ok")
                detail = "reference: %r  oracle: %r" % (a, b)
            else:
                dt = "<i8" if f.endswith(".i64") else "<u8"  # doubles compared as bit patterns
                a, b = np.fromfile(rp, dtype=dt), np.fromfile(os.path.join(base, f), dtype=dt)
                if f.endswith(".f64"):
                    fa, fb = a.view("<f8"), b.view("<f8")
                    na, nb = np.isnan(fa), np.isnan(fb)
                    same = len(a) == len(b) and np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb])
                    detail = "" if same else ("max |diff| %.3e of %d" % (np.nanmax(np.abs(fa - fb)) if len(a) == len(b) else float("nan"), len(a)))
                else:
                    same = np.array_equal(a, b)
                    detail = ""
            if not same:
                bad.append((rel, detail))
    for rel, detail in bad:
        print("DIFFERENT  %-60s %s" % (rel, detail), file=out)
    print("%d files compared, %d different, %d only in the oracle's dump" % (seen, len(bad), len(missing)), file=out)
    return len(bad) == 0 and seen > 0


def main():
    ref_dir = sys.argv[1]
    inputs = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "rust", "inputs")
    import oracle_dump
    with tempfile.TemporaryDirectory() as tmp:
        oracle_dump.main(inputs, tmp)
        ok = compare_dirs(ref_dir, tmp)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
