#!/bin/bash
# Build ablation variants of liblbmi.so into tools/variants/ (R&D only).
# usage: tools/build_variants.sh name1="-DFLAG..." name2="..."
set -e
cd "$(dirname "$0")/../ludwig_amd/csrc"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  make -s OUT=../../tools/variants/liblbmi_$name.so VARIANT="$flags" OBJSFX=_$name
  echo "built $name: $flags"
done
