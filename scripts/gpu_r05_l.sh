#!/bin/bash
# r05 l (evidence 2): BASELINE configs[4]'s per-GPU slab — counters (bytes) + kernel statistics of its products and set-up, its bench line with the per-level
# table; BASELINE configs[2]; configs[3] as its text reads (GS smoother)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_l
O=gpurun_out/r05_l
bash scripts/gpu_pmc.sh r05_config5_pmc 6 bytes --workload config5 > $O/pmc.log 2>&1; tail -4 $O/pmc.log
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py --workload config5 --steps 5 --warmup 1 --levels-csv $O/config5_levels.csv > $O/config5_bench.json 2> $O/config5.err; echo "config5 rc=$?"; python3 -c "import json; d=json.load(open('$O/config5_bench.json')); print('config5 ms_per_step %.1f setup_s %s frac %.3f hbm %s' % (d['ms_per_step'], d['config']['setup_s'], d['roofline']['frac'], d['config']['hbm_used_gb']), d['config']['mixed_mesh']['generation_s'])"
timeout -k 10 300 python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 > $O/config3_bench.json 2> $O/config3.err; echo "config3 rc=$?"; python3 -c "import json; d=json.load(open('$O/config3_bench.json')); print('configs[2] ms_per_step %.2f' % d['ms_per_step'])"
timeout -k 10 300 python3 bench.py --solver multigrid_gs --steps 3 --warmup 1 --no-cpu-baseline > $O/config4_gs_bench.json 2> $O/config4_gs.err; echo "gs rc=$?"; python3 -c "import json; d=json.load(open('$O/config4_gs_bench.json')); print('configs[3] GS smoother ms_per_step %.1f hbm %s' % (d['ms_per_step'], d['config']['hbm_used_gb']))"
