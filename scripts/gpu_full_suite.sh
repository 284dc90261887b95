#!/bin/bash
# the whole GPU suite (what the driver runs at round end), then smoke() and one short bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/full_suite.log 2>&1; rc=$?; tail -28 gpurun_out/full_suite.log; echo "suite rc=$rc"
[ $rc = 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/full_suite_bench.json 2> gpurun_out/full_suite_bench.err && python -c "import json; d=json.load(open('gpurun_out/full_suite_bench.json')); print('ms_per_step %.1f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], [round(l['frac_of_peak'],3) for l in d['amg_levels']], 'hbm', d['config']['hbm_used_gb'])"
