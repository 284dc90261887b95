#!/bin/bash
# r04 call aq: level-0 row mirror on by default: full GPU suite, smoke, both full-size reference-mode runs against the oracle, bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_aq
O=gpurun_out/r04_aq
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout=900 -x > $O/suite.log 2>&1
rc=$?; tail -3 $O/suite.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/suite.log | head -20; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log || exit 1
bash scripts/gpu_final_r04_f.sh || exit 1
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --levels-csv $O/levels.csv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
