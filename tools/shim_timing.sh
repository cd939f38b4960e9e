#!/bin/bash
# The reference's own HIP back end on this GPU against the same program with liblbmi bound in (oracle/_ref, built
# by `make -C oracle hip`): D3Q19 M10 256^3, per step hydro_f_zero + hydro_u_zero + lb_collide + lb_halo +
# lb_propagation as ludwig.c's loop issues them (ludwig.c:537, 791, 802-860), wall clock incl. final sync.
# The FIRST bound line is the one that counts: no LBMI_* variable set.
R=oracle/_ref
for k in LBMI_MODE LBMI_HYDRO LBMI_FE LBMI_REPORT; do unset $k; done
echo "reference HIP target as it is:"; timeout -k 10 300 $R/ref_driver_hip_d3q19 time 256 256 256 m10 0.1 0.3 20
echo "with the binding, no LBMI_* variable set (200 steps):"; timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 200
echo "with the binding, no LBMI_* variable set (40 steps, as round 2 measured):"; timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 40
echo "with the binding, LBMI_HYDRO=store:"; LBMI_HYDRO=store timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 200
for m in halo eager; do
  echo "with the binding, LBMI_MODE=$m:"; LBMI_MODE=$m timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 100
  echo "with the binding, LBMI_MODE=$m LBMI_HYDRO=store:"; LBMI_MODE=$m LBMI_HYDRO=store timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 100
done
echo "D3Q27 reference:"; timeout -k 10 300 $R/ref_driver_hip_d3q27 time 192 192 192 m10 0.1 0.3 20
echo "D3Q27 with the binding, no LBMI_* variable set:"; timeout -k 10 300 $R/ref_driver_hip_d3q27_shim time 192 192 192 m10 0.1 0.3 200
