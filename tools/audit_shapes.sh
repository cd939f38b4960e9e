#!/bin/bash
# lattice shapes away from the headline cube: looking for cliffs (fraction of the 8 TB/s roofline per shape)
run() { timeout -k 10 300 python bench.py --cpu-baseline 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline'] or {}; print('%-50s %9.1f MLUPS  %.5f ms/step  kernel %s  frac %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac')))" "$@"; }
run --size 255 255 255 --steps 60
run --size 257 257 257 --steps 60
run --size 100 100 100 --steps 200
run --size 512 512 32 --steps 100
run --size 1024 1024 8 --steps 100
run --size 32 512 512 --steps 100
run --size 2048 64 64 --steps 100
run --size 64 64 2048 --steps 100
run --size 256 256 256 --nhalo 2 --steps 60
run --size 384 384 384 --steps 30
