#!/bin/bash
# Run bench.py against each ablation build in tools/variants (same box, same
# process order); prints variant, hydro flag, MLUPS, kernel ms.
cd "$(dirname "$0")/.."
for lib in tools/variants/liblbmi_*.so; do
  name=$(basename $lib .so); name=${name#liblbmi_}
  for hydro in ${HYDRO:-0 1}; do
    LBMI_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-baseline 0 --hydro $hydro ${EXTRA} 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s hydro=%s  %9.1f MLUPS  kernel %.4f ms  algo %.0f GB/s' % ('$name', '$hydro', d['value'], r['avg_launch_ms'], r['achieved']))"
  done
done
