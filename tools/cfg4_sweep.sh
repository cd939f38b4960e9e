#!/bin/bash
# BASELINE config 4 (128^3 D3Q19 + symmetric free energy) through lbmi_symmetric_lb_step: launch tuning of k_symm_lb_step
for cap in 0 32768 49152 65536; do for grp in 8 32 128; do
  echo -n "lds_cap=$cap xcd_group=$grp: "
  python bench.py --config 4 --steps 200 --warmup 20 --cpu-baseline 0 --tune lds_cap=$cap,xcd_group=$grp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value'])"
done; done
