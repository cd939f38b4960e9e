#!/bin/bash
# rocprofv3 kernel stats of the execution modes other than the default FUSED line: EAGER (the reference's three
# stages) and FUSED_HALO (the binding's default), both with the hydro arrays read and written by every collision
# (--hydro 1), D3Q19 M10 256^3: gpurun_out/<tag>_{eager,halo}_stats/, <tag>_{eager,halo}_bench.json
set -e
tag=${1:-modes}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for m in eager fused_halo; do
  s=${m/fused_/}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${s}_stats -o s -- python3 bench.py --mode $m --hydro 1 --steps 60 --warmup 10 --cpu-baseline 0 > $out/${tag}_${s}.log 2>&1
  grep '^{' $out/${tag}_${s}.log | tail -1 > $out/${tag}_${s}_bench.json
done
ls $out/${tag}_eager_stats $out/${tag}_halo_stats
