#!/bin/bash
# rocprofv3 kernel stats of the reference's application (main.c + ludwig.c) with the binding, relaxing droplet
# (tests/golden/inputs/iodrop.inp, 20 steps): which kernels are the library's, which the reference's.
#   bash tools/profile_app.sh <tag> [LBMI_MODE [LBMI_FE]]      -> gpurun_out/<tag>_stats/, gpurun_out/<tag>.log
set -e
tag=${1:-app}
root="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
d=$(mktemp -d); cp $root/tests/golden/inputs/iodrop.inp $d/input
cd $d
export LBMI_MODE=${2:-halo}
export LBMI_FE=${3:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o s -- $root/oracle/_ref/ludwig_hip_d3q19_shim > $root/gpurun_out/${tag}.log 2>&1
cd $root; rm -rf $d
tail -3 gpurun_out/${tag}.log
