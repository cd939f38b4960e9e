#!/bin/bash
# The reference's executable with the binding and NO LBMI_* variable set (what a drop-in user runs), by its own timer
# report: D3Q19 single fluid 256^3 and the symmetric free energy at 128^3 (BASELINE config 4), 1000 steps each,
# statistics at steps 0 and 1000. Beside it, for the second: LBMI_FE=1 (the free-energy sector bound call by call, not
# folded into the collision) and LBMI_FE=0 (left to the reference's own kernels).
R=$PWD/oracle/_ref
run() {  # input exe [VAR=value ...]
  d=$(mktemp -d); cp $1 $d/input; inp=$1; exe=$2; shift 2
  ( cd $d && env -u LBMI_MODE -u LBMI_FE -u LBMI_HYDRO LBMI_REPORT=1 "$@" timeout -k 10 280 $exe > log 2> err; echo "exit $?"
    echo "== $(basename $inp) $(basename $exe) $*"
    grep -E "Time step loop|Collision:|Propagation:|Lattice halos|Force calculation|phi update|phi gradients|Diagnostics|Total:" log
    grep -E "execution mode" err; tail -2 log )
  rm -rf $d
}
t=$(mktemp -d)
cat > $t/single256 <<EOT
N_cycles 1000
size 256_256_256
lb_halo_scheme lb_halo_target
viscosity 0.1
free_energy none
distribution_initialisation 3d_uniform_u
distribution_uniform_u 0.002_0.003_0.004
colloid_init none
periodicity 1_1_1
freq_statistics 1000
config_at_end no
EOT
cat > $t/binary128 <<EOT
N_cycles 1000
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
freq_statistics 1000
config_at_end no
EOT
run $t/single256 $R/ludwig_hip_d3q19_shim
run $t/binary128 $R/ludwig_hip_d3q19_shim
run $t/binary128 $R/ludwig_hip_d3q19_shim LBMI_FE=1
run $t/binary128 $R/ludwig_hip_d3q19_shim LBMI_FE=0
rm -rf $t
