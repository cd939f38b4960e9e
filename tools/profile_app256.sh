#!/bin/bash
# rocprofv3 kernel stats of the reference's application with the UNCONFIGURED binding, D3Q19 single fluid 256^3, 200 steps:
# what runs per step of ludwig.c's main loop beside the library's one kernel.
#   bash tools/profile_app256.sh <tag>      -> gpurun_out/<tag>_stats/, gpurun_out/<tag>.log
set -e
tag=${1:-app256}
root="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
d=$(mktemp -d)
cat > $d/input <<EOT
N_cycles 200
size 256_256_256
lb_halo_scheme lb_halo_target
viscosity 0.1
free_energy none
distribution_initialisation 3d_uniform_u
distribution_uniform_u 0.002_0.003_0.004
colloid_init none
periodicity 1_1_1
freq_statistics 200
config_at_end no
EOT
cd $d
unset LBMI_MODE LBMI_FE LBMI_HYDRO
export LBMI_REPORT=1
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o s -- $root/oracle/_ref/ludwig_hip_d3q19_shim > $root/gpurun_out/${tag}.log 2>&1
cd $root; rm -rf $d
tail -3 gpurun_out/${tag}.log
