#!/bin/bash
# rocprofv3 kernel stats of the reference's application (main.c + ludwig.c) with the binding, relaxing droplet
# (tests/golden/inputs/iodrop.inp, 20 steps, LBMI_MODE=fused): which kernels are the library's, which the reference's
set -e
root="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
d=$(mktemp -d); cp $root/tests/golden/inputs/iodrop.inp $d/input
cd $d
export LBMI_MODE=fused
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/app_stats -o s -- $root/oracle/_ref/ludwig_hip_d3q19_shim > $root/gpurun_out/app_prof.log 2>&1
cd $root; rm -rf $d
tail -3 gpurun_out/app_prof.log
