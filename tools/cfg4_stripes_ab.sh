#!/bin/bash
# BASELINE config 4 through lbmi_symmetric_lb_step: every XCD an eighth of every x plane (fe_stripes 1) against runs of
# the 1-d site order dealt out to the XCDs (0, with xcd_group 32 and 8), interleaved; then 256^3
run() { python bench.py "$@" --cpu-baseline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"; }
for rep in 1 2 3; do
  for t in "fe_stripes=0" "fe_stripes=0,xcd_group=8" "fe_stripes=1"; do
    echo -n "128^3 $t: "; run --config 4 --steps 300 --warmup 20 --tune $t
  done
done
for t in "fe_stripes=0" "fe_stripes=1"; do
  echo -n "256^3 $t: "; run --fe symmetric --nhalo 2 --hydro 1 --size 256 256 256 --steps 100 --warmup 10 --tune $t
done
