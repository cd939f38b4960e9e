#!/bin/bash
# The reference's executable (main.c, ludwig.c) on one MI355X, as it is and with the binding: its own timer report.
#   D3Q19 SIZE^3 (default 128) single fluid, 100 steps; D3Q19 128^3 symmetric free energy (BASELINE config 4), 100 steps
R=$PWD/oracle/_ref
run() {  # name exe mode
  d=$(mktemp -d); cp $1 $d/input
  ( cd $d && LBMI_MODE=$3 LBMI_FE=${4:-0} LBMI_HYDRO=${5:-} timeout -k 10 150 $2 > log 2>&1; echo "exit $?"; echo "== $(basename $1) $(basename $2) LBMI_MODE=$3 LBMI_FE=${4:-0} LBMI_HYDRO=${5:-}"; grep -E "Time step loop|Collision:|Propagation:|Lattice halos|Force calculation|phi update|phi gradients|finished|Total:" log; tail -3 log )
  rm -rf $d
}
t=$(mktemp -d)
cat > $t/single256 <<EOT
N_cycles 100
size ${SIZE:-128}_${SIZE:-128}_${SIZE:-128}
lb_halo_scheme lb_halo_target
viscosity 0.1
free_energy none
distribution_initialisation 3d_uniform_u
distribution_uniform_u 0.002_0.003_0.004
colloid_init none
periodicity 1_1_1
freq_statistics 100
config_at_end no
EOT
cat > $t/binary128 <<EOT
N_cycles 100
size 128_128_128
lb_halo_scheme lb_halo_target
viscosity 0.00625
free_energy symmetric
A -0.00625
B 0.00625
K 0.004
phi0 0.0
phi_initialisation drop
phi_init_drop_radius 32.0
mobility 1.25
fd_gradient_calculation 3d_7pt_fluid
fd_advection_scheme_order 1
colloid_init no_colloids
periodicity 1_1_1
freq_statistics 100
config_at_end no
EOT
run $t/single256 $R/ludwig_hip_d3q19 eager
for m in eager halo fused; do run $t/single256 $R/ludwig_hip_d3q19_shim $m; done
run $t/single256 $R/ludwig_hip_d3q19_shim fused 0 lazy
run $t/binary128 $R/ludwig_hip_d3q19 eager
for m in halo fused; do run $t/binary128 $R/ludwig_hip_d3q19_shim $m; done
run $t/binary128 $R/ludwig_hip_d3q19_shim fused 1
rm -rf $t
