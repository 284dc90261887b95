#!/bin/bash
# r04 call n: full GPU suite on the current tree, smoke(), default bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout=900 -x > gpurun_out/r04n_suite.log 2>&1
rc=$?
tail -5 gpurun_out/r04n_suite.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04n_smoke.log 2>&1 && tail -2 gpurun_out/r04n_smoke.log &&
timeout -k 10 400 python bench.py > gpurun_out/r04n_bench.json 2> gpurun_out/r04n_bench.err && cut -c1-600 gpurun_out/r04n_bench.json
