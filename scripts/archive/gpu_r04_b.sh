#!/bin/bash
# r04 call b: full GPU suite, then the full-size reference-mode run, then a bench line.  A step that is killed or times out ends the call.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/r04b_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r04b_tests.log
if [ $rc -gt 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
timeout -k 10 600 python scripts/reference_mode_fullsize.py --iterations 3 --out gpurun_out/r04_reference_mode_400x160x160.json > gpurun_out/r04b_refmode.log 2>&1
rc2=$?
tail -5 gpurun_out/r04b_refmode.log
if [ $rc2 -gt 1 ]; then echo "reference-mode run ended with $rc2: stopping"; exit $rc2; fi
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
rc3=$?
cut -c 1-400 gpurun_out/r04b_bench.json
exit $(( rc > rc3 ? rc : rc3 ))
