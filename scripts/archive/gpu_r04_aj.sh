#!/bin/bash
# r04 call aj: momentum_k with an XCD-aware walk over the cells: assembly tests, then time and memory-side bytes of the kernel in both builds
# (scripts/profile_kernels.py: one momentum assembly at 10.24 M cells; --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE: separate passes)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_aj
O=gpurun_out/r04_aj
timeout -k 10 800 python -m pytest tests/test_gpu_assembly.py tests/test_gpu_solve_steady.py tests/test_gpu_golden.py tests/test_gpu_reference_order.py -q -x --timeout=700 > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
cp orc_amd/liborc_amd.so $O/new.so; cp orc_amd/liborc_amd_alt.so $O/old.so
for v in new old; do
  cp $O/$v.so orc_amd/liborc_amd.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/k_$v --output-format csv -- python3 scripts/profile_kernels.py > $O/k_$v.log 2>&1
  cp $O/k_$v/*/*kernel_stats.csv $O/stats_$v.csv; rm -rf $O/k_$v
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c -d $O/p_${v}_$c --output-format csv -- python3 scripts/profile_kernels.py > $O/p_${v}_$c.log 2>&1
    cp $O/p_${v}_$c/*/*counter_collection.csv $O/pmc_${v}_$c.csv; rm -rf $O/p_${v}_$c
  done
done
cp $O/new.so orc_amd/liborc_amd.so; rm -f $O/new.so $O/old.so
python3 - <<'PY'
import csv
O='gpurun_out/r04_aj'
for v in ('new','old'):
    t=[r for r in csv.DictReader(open(f'{O}/stats_{v}.csv')) if 'momentum_k' in r['Name'] and 'init' not in r['Name']]
    us=float(t[0]['AverageNs'])/1e3 if t else 0
    b={}
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        tot=0.
        for r in csv.DictReader(open(f'{O}/pmc_{v}_{c}.csv')):
            if 'momentum_k' in r['Kernel_Name'] and 'init' not in r['Kernel_Name'] and r['Counter_Name']==c: tot+=float(r['Counter_Value'])
        b[c]=tot
    rd=b['FETCH_SIZE']*1024*2; wr=b['WRITE_SIZE']*1024
    print(v, 'momentum_k %.1f us  read %.2f GB  written %.2f GB' % (us, rd/1e9, wr/1e9))
PY
