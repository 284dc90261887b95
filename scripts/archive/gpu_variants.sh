#!/bin/bash
# usage: scripts/gpu_variants.sh "ENV1=a ENV2=b" "ENV3=c" ...   — one short bench.py run per variant, ms_per_step + product times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/variant_$i.log 2>&1
  rc=$?
  echo "[$v] rc=$rc $(tail -1 gpurun_out/variant_$i.log | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    r = d['roofline']
    t = r.get('three_systems_per_launch') or {}
    print('ms_per_step %.1f  inloop %.1f us  triple %.1f us  levels %s  hbm %.0f GB' % (d['ms_per_step'], 1e3 * r['avg_launch_ms'], 1e3 * t.get('avg_launch_ms', 0), [round(L['us_per_product']) for L in d['amg_levels']], d['config']['hbm_used_gb']))
except Exception as e:
    print('no json', e)
")"
done
