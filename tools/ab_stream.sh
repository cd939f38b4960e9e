#!/bin/bash
# A/B: compute on torch's current (legacy default) stream vs the library's own non-blocking stream
run() { timeout -k 10 300 python bench.py --cpu-baseline 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline'] or {}; print('%-60s %9.1f MLUPS  %.5f ms/step  kernel %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r.get('avg_launch_ms')))" "$@"; }
for os in 0 1 0 1; do
  run --size 64 64 64 --scheme bgk --steps 2000 --warmup 50 --own-stream $os
done
for os in 0 1 0 1; do
  run --size 32 256 256 --selfring 1 --steps 400 --own-stream $os
done
for os in 0 1; do
  run --size 128 128 128 --steps 1000 --own-stream $os
  run --size 256 256 256 --steps 200 --own-stream $os
done
