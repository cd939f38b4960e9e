#!/bin/bash
# rocprofv3 kernel stats of the reference's HIP back end and of the same program with the binding (fused)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
R=oracle/_ref
rocprofv3 --kernel-trace --stats --output-format csv -d $out/refhip_stats -o s -- $R/ref_driver_hip_d3q19 time 256 256 256 m10 0.1 0.3 20 > $out/refhip.log 2>&1
export LBMI_MODE=fused
rocprofv3 --kernel-trace --stats --output-format csv -d $out/shimhip_stats -o s -- $R/ref_driver_hip_d3q19_shim time 256 256 256 m10 0.1 0.3 40 > $out/shimhip.log 2>&1
ls $out/refhip_stats $out/shimhip_stats
