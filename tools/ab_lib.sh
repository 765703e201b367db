#!/bin/bash
# A/B of a variant library (tools/build_variant.sh) against the working build on ONE box: tools/ab_lib.sh <variant> "<script and args>" [rounds]
v="$1"; cmd="$2"; n="${3:-2}"
for i in $(seq 1 $n); do
  echo "== $v"; FHELIN_LIB=$GRAFT_REPO_ROOT/tmp_variants/$v/libfhelin_amd.so python $cmd 2>/dev/null | cut -c1-200
  echo "== working build"; python $cmd 2>/dev/null | cut -c1-200
done
