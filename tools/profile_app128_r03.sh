#!/bin/bash
# Kernel trace of the reference's application with the binding at 128^3 (symmetric free energy, BASELINE config 4's
# physics, 60 steps), LBMI_MODE unset; LBMI_FE as given (default unset):
#   bash tools/profile_app128_r03.sh <tag> [LBMI_FE]  -> gpurun_out/<tag>_trace/
set -e
tag=${1:-app128}
root="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
d=$(mktemp -d)
cat > $d/input <<EOT
N_cycles 60
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
freq_statistics 60
config_at_end no
EOT
cd $d
unset LBMI_MODE LBMI_HYDRO LBMI_FE
if [ -n "$2" ]; then export LBMI_FE=$2; fi
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_trace -o t -- $root/oracle/_ref/ludwig_hip_d3q19_shim > $root/gpurun_out/${tag}.log 2>&1
cd $root; rm -rf $d
