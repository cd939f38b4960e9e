#!/bin/bash
# The reference's own unit-test suites around the hot path (oracle/unit_main.c), each in its own process, against the
# reference's HIP target as it is and with the binding; under the table, per suite, how many calls of each bound
# symbol the library took and how many it handed back (LBMI_REPORT=1):   bash tools/unit_suites.sh [d3q19|d3q27]
nv=${1:-d3q19}
R=oracle/_ref
out=gpurun_out/unit_$nv
mkdir -p $out
export LBMI_REPORT=1
printf "%-20s %-28s %-28s\n" "suite ($nv)" "unbound" "bound"
suites="lb_d3q19 lb_d3q27 lb_model model halo prop lb_bc_inflow_rhou lb_bc_outflow_rhou wall hydro field field_grad map le phi_ch"
for s in $suites; do
  line=$(printf "%-20s" $s)
  for leg in "" _shim; do
    (cd $out && timeout -k 10 120 ../../$R/unit_hip_${nv}${leg} $s > ${s}${leg}.log 2>&1)
    rc=$?
    npass=$(grep -c "^PASS" $out/${s}${leg}.log)
    if [ $rc -eq 0 ] && grep -q "^DONE     $s" $out/${s}${leg}.log; then
      line="$line $(printf "%-28s" "passed ($npass PASS lines)")"
    else
      line="$line $(printf "%-28s" "FAILED rc=$rc")"
    fi
  done
  echo "$line"
done
echo
for s in $suites; do
  if grep -q "^liblbmi report" $out/${s}_shim.log; then
    echo "# $s (bound):"
    grep "^liblbmi report" $out/${s}_shim.log | sed 's/^liblbmi report: /#   /'
  fi
done
