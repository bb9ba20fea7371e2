#!/bin/bash
# batch throughput of several builds / workgroup sizes: bash tools/batch_ab.sh CFG "lib:block" ...
CFG=$1; shift
for spec in "$@"; do
  L=${spec%%:*}; B=${spec##*:}
  echo "== $L block $B"
  BLU_HIP_LIB=$PWD/blu_amd/$L timeout -k 10 400 python tools/batch_probe.py 1280 $B $CFG 2>&1 | grep "rep[12]"
done
