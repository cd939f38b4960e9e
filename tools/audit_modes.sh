#!/bin/bash
# every execution mode x collision scheme x velocity set through bench.py: looking for cliffs
run() { timeout -k 10 300 python bench.py --cpu-baseline 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline'] or {}; print('%-66s %9.1f MLUPS  %.5f ms/step  kernel %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r.get('avg_launch_ms')))" "$@"; }
for nvel in 19 27; do
  for mode in fused eager inplace fused_halo; do
    for s in m10 bgk trt; do
      if [ $nvel = 27 ] && [ $s = trt ]; then continue; fi
      run --nvel $nvel --size 160 160 160 --mode $mode --scheme $s --steps 60 --warmup 5
    done
  done
done
