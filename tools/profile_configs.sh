#!/bin/bash
# rocprofv3 kernel stats of the other BASELINE configurations (1: 64^3 BGK, 4: 128^3 symmetric FE,
# 5: D3Q27 64x512x256 per GPU) and the FUSED_HALO mode: gpurun_out/cfg<k>_{stats/,bench.json}
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
prof() {
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 bench.py --cpu-baseline 0 "$@" > $out/${tag}.log 2>&1
  grep '^{' $out/${tag}.log | tail -1 > $out/${tag}_bench.json
}
prof cfg1 --size 64 64 64 --scheme bgk --steps 2000 --warmup 50
prof cfg4 --size 128 128 128 --fe symmetric --nhalo 2 --steps 500
prof cfg5 --nvel 27 --size 64 512 256 --steps 200
prof cfgh --mode fused_halo --steps 100
ls $out/cfg*_stats
