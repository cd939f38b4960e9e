#!/bin/bash
# bench lines per collision scheme (m10 | bgk | trt) at two sizes
run() { timeout -k 10 300 python bench.py --cpu-baseline 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline'] or {}; print('%-60s %9.1f MLUPS  %.5f ms/step  kernel %s  frac %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac')))" "$@"; }
for s in m10 bgk trt; do
  run --size 64 64 64 --scheme $s --steps 2000 --warmup 50
  run --size 256 256 256 --scheme $s --steps 100
done
run --nvel 27 --size 128 128 128 --scheme bgk --steps 200
run --nvel 27 --size 128 128 128 --scheme m10 --steps 200
