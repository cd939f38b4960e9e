#!/bin/bash
# 5000 steps of the droplet (tests/golden/inputs/iodrop7.inp with N_cycles 5000, reports every 1000) through the
# reference's executable: the free-energy sector folded into lb_collide (nothing set), bound call by call (LBMI_FE=1),
# the reference's own kernels (LBMI_FE=0). The reports of the three, side by side.
R=$PWD/oracle/_ref
for fe in "" 1 0; do
  d=$(mktemp -d); cp tools/soak_iodrop7.inp $d/input
  ( cd $d && env -u LBMI_MODE -u LBMI_HYDRO -u LBMI_FE ${fe:+LBMI_FE=$fe} LBMI_REPORT=1 timeout -k 10 250 $R/ludwig_hip_d3q19_shim > log 2> err
    echo "== LBMI_FE=${fe:-unset}: $(grep -c 'folded into lb_collide' log) $(grep 'execution mode' err | sed 's/.*; free/free/')"
    grep -E "^\[rho\]|^\[phi\]|^\[fed\]|^\[maximum \]" log | tail -8 )
  rm -rf $d
done
