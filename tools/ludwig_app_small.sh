#!/bin/bash
# The reference's executable with the unconfigured binding on a SMALL lattice (BASELINE config 1: D3Q19 64^3), 20000
# steps, one report at the end: is the step bound by the host that issues it (ludwig.c's loop + the binding) or by the
# device? Its timer report; LBMI_REPORT=1.
R=$PWD/oracle/_ref
d=$(mktemp -d)
cat > $d/input <<EOT
N_cycles 20000
size 64_64_64
lb_halo_scheme lb_halo_target
viscosity 0.1
lb_relaxation_scheme bgk
free_energy none
distribution_initialisation 3d_uniform_u
distribution_uniform_u 0.002_0.003_0.004
colloid_init none
periodicity 1_1_1
freq_statistics 20000
config_at_end no
EOT
( cd $d && env -u LBMI_MODE -u LBMI_FE -u LBMI_HYDRO LBMI_REPORT=1 timeout -k 10 280 $R/ludwig_hip_d3q19_shim > log 2> err; echo "exit $?"
  grep -E "Time step loop|Collision:|Propagation:|Lattice halos|Diagnostics|Total:|Relaxation|relaxation" log; grep "execution mode" err; tail -2 log )
rm -rf $d
