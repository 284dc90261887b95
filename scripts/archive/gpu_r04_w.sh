#!/bin/bash
# A/B on one box — liborc_amd.so (the new build) against liborc_amd_alt.so (the build before), bench runs in the order new old old new new old: a box
# drifts up by 3-5 ms per run as it warms, and a fixed "new first" order hands the new build that much (caveat)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_w
O=gpurun_out/r04_w
cp orc_amd/liborc_amd.so $O/new.so; cp orc_amd/liborc_amd_alt.so $O/old.so
r=0; for v in new old old new new old; do r=$((r+1)); round=$r;
  cp $O/$v.so orc_amd/liborc_amd.so
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python - "$O/bench_${v}_$round.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'ms_per_step %.1f' % d['ms_per_step'], 'L0 x1 %.1f us, x3 %.1f us' % (1e3*r['avg_launch_ms'], 1e3*r['three_systems_per_launch']['avg_launch_ms']), 'levels', [round(l['us_per_product'],1) for l in d['amg_levels']])
PY
done
rm -f $O/new.so $O/old.so
