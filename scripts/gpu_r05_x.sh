#!/bin/bash
# r05 x (evidence 4, final tree): the bench line (10 steps, CPU baseline, per-level table), config 5's bench line and per-level table, kernel statistics of
# bench.py in the concurrent schedule
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_x
O=gpurun_out/r05_x
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --levels-csv $O/levels.csv > $O/bench_10steps.json 2> $O/bench_10steps.err; echo "bench rc=$?"
python3 -c "import json; d=json.load(open('$O/bench_10steps.json')); print('hex ms_per_step %.1f frac %.3f traffic %s cpu %s hbm %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['config']['hbm_used_gb']), [round(x) for x in d['step_ms']])"; cat $O/levels.csv
timeout -k 10 400 python3 bench.py --workload config5 --steps 5 --warmup 1 --levels-csv $O/config5_levels.csv > $O/config5_bench.json 2> $O/config5.err; echo "config5 rc=$?"
python3 -c "import json; d=json.load(open('$O/config5_bench.json')); print('config5 ms_per_step %.1f setup_s %s frac %.3f hbm %s gen %s' % (d['ms_per_step'], d['config']['setup_s'], d['roofline']['frac'], d['config']['hbm_used_gb'], d['config']['mixed_mesh']['generation_s']), d['config']['momentum_solves'])"; cat $O/config5_levels.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1
cp $O/conc/*/*kernel_stats.csv $O/bench_multigrid_concurrent_3steps_kernel_stats.csv; rm -rf $O/conc; echo "concurrent profile done"
