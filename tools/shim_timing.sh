#!/bin/bash
# The reference's own HIP back end on this GPU against the same program with liblbmi bound in (oracle/_ref, built
# by `make -C oracle hip`): D3Q19 M10 256^3, lb_collide + lb_halo + lb_propagation, wall clock incl. final sync
R=oracle/_ref
echo "reference HIP target as it is:"; timeout -k 10 300 $R/ref_driver_hip_d3q19 time 256 256 256 m10 0.1 0.3 20
for m in eager halo fused; do echo "with the binding, LBMI_MODE=$m:"; LBMI_MODE=$m timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 40; done
echo "with the binding, LBMI_MODE unset (= halo), LBMI_HYDRO=lazy:"; LBMI_HYDRO=lazy timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 40
echo "with the binding, LBMI_MODE=fused LBMI_HYDRO=lazy:"; LBMI_MODE=fused LBMI_HYDRO=lazy timeout -k 10 300 $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 40
echo "D3Q27 reference:"; timeout -k 10 300 $R/ref_driver_hip_d3q27 time 192 192 192 m10 0.1 0.3 20
echo "D3Q27 binding fused:"; LBMI_MODE=fused timeout -k 10 300 $R/ref_driver_hip_d3q27_shim time 192 192 192 m10 0.1 0.3 40
