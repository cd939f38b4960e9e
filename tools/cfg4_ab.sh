#!/bin/bash
# BASELINE config 4 through lbmi_symmetric_lb_step: the shipped library against a variant (LBMI_LIB), interleaved
V=${1:-ludwig_amd/liblbmi_w4.so}
for rep in 1 2 3; do
  for lib in ludwig_amd/liblbmi.so $V; do
    echo -n "$lib: "
    LBMI_LIB=$PWD/$lib python bench.py --config 4 --steps 300 --warmup 20 --cpu-baseline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"
  done
done
echo "256^3:"
for lib in ludwig_amd/liblbmi.so $V; do
  echo -n "$lib: "
  LBMI_LIB=$PWD/$lib python bench.py --fe symmetric --nhalo 2 --hydro 1 --size 256 256 256 --steps 100 --warmup 10 --cpu-baseline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"
done
