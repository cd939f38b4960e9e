#!/bin/bash
# bench.py over execution mode x hydro traffic on one GPU (D3Q19 M10 256^3): one line per case.
# usage: bash tools/mode_matrix.sh [outfile]
out=${1:-gpurun_out/mode_matrix.txt}
: > $out
for mode in eager inplace fused_halo fused_soa fused; do
  for hydro in 0 1 lazy; do
    timeout -k 10 200 python bench.py --mode $mode --hydro $hydro --steps 60 --warmup 10 --cpu-baseline 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%-11s hydro=%-4s %8.1f MLUPS %7.4f ms/step  roofline GB/s %s' % ('$mode', '$hydro', d['value'], d['ms_per_step'], json.dumps((d.get('roofline') or {}).get('achieved'))))
" >> $out || exit 1
  done
done
cat $out
