#!/bin/bash
# the whole d3q19-short suite, both executables, one part per gpurun call:  bash tools/regression_sweep.sh a 0 56
part=$1; first=$2; last=$3
rm -f gpurun_out/regression_sweep_$part.jsonl
timeout -k 10 1050 python tools/regression_sweep.py run --first $first --last $last --limit 150 \
  --out gpurun_out/regression_sweep_$part.jsonl > gpurun_out/regression_sweep_$part.txt 2>&1
rc=$?
tail -n 4 gpurun_out/regression_sweep_$part.txt
exit $rc
