#!/bin/bash
# A/B of an environment knob on ONE box: tools/ab_env.sh VAR "<script and args>" [rounds]; prints the script's output for VAR=0 / VAR=1 alternately
var="$1"; cmd="$2"; n="${3:-2}"
for i in $(seq 1 $n); do
  for v in 0 1; do
    echo "== $var=$v"; env $var=$v python $cmd 2>/dev/null | cut -c1-400
  done
done
