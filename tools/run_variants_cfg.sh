#!/bin/bash
# R&D: bench.py of a BASELINE configuration against each build in tools/variants, three rounds on one box:
#   bash tools/run_variants_cfg.sh 4
cd "$(dirname "$0")/.."
cfg=${1:-4}
for round in 1 2 3; do
  for lib in tools/variants/liblbmi_*.so; do
    name=$(basename $lib .so); name=${name#liblbmi_}
    LBMI_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config $cfg --steps 200 --warmup 20 --cpu-baseline 0 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s config $cfg  %9.1f MLUPS  %.5f ms/step' % ('$name', d['value'], d['ms_per_step']))"
  done
done
