#!/bin/bash
# LBMI_MODE_FUSED_HALO, D3Q19 M10 256^3: the halo shell computed by the collision kernel (halo_fold 1, the default)
# against three k_halo_copy launches after it (halo_fold 0), interleaved
run() { python bench.py "$@" --steps 100 --warmup 10 --cpu-baseline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"; }
for rep in 1 2 3; do for fold in 0 1; do for hyd in 1 lazy; do
  echo -n "fused_halo hydro $hyd halo_fold=$fold: "; run --mode fused_halo --hydro $hyd --tune halo_fold=$fold
done; done; done
