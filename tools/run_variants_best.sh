#!/bin/bash
# best of REPS separate processes per variant (process-to-process placement
# varies by +-4 %): variant, hydro, best kernel ms, GB/s
cd "$(dirname "$0")/.."
for lib in tools/variants/liblbmi_*.so; do
  name=$(basename $lib .so); name=${name#liblbmi_}
  for hydro in ${HYDRO:-0 1}; do
    for rep in $(seq 1 ${REPS:-3}); do
      LBMI_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-baseline 0 --hydro $hydro ${EXTRA} 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s hydro=%s %.4f %.0f' % ('$name', '$hydro', r['avg_launch_ms'], r['achieved']))"
    done
  done
done | sort | awk '{k=$1" "$2; if (!(k in b) || $3 < b[k]) {b[k]=$3; g[k]=$4} all[k]=all[k]" "$3} END {for (k in b) print k, "best", b[k], "ms", g[k], "GB/s  all:" all[k]}' | sort
