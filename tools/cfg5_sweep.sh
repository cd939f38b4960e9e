#!/bin/bash
# BASELINE config 5 per GPU (D3Q27 M10 64x512x256): launch tuning of the fused kernel, interleaved on one box.
#   bash tools/cfg5_sweep.sh  -> gpurun_out/cfg5_sweep.txt
out=gpurun_out/cfg5_sweep.txt
: > $out
for rep in 1 2; do
for t in "" "xcd_group=16" "xcd_group=64" "lds_cap=32768" "lds_cap=49152" "lds_cap=98304" "xcd_group=16,lds_cap=49152" "xcd_group=64,lds_cap=49152"; do
  python3 bench.py --config 5 --steps 100 --warmup 10 --cpu-baseline 0 ${t:+--tune $t} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline') or {}
        print('%-28s %8.1f MLUPS  %.5f ms/step  kernel %.5f ms  frac %s' % ('${t:-default}', d['value'], d['ms_per_step'], r.get('avg_launch_ms', 0), r.get('frac')))
" >> $out
done
done
cat $out
