#!/bin/bash
# A/B of builds of the library on a C3 batch of 1536 bases: bash tools/batch_lib_ab.sh libA.so libB.so ...  (files under blu_amd/)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/batch_lib_ab.log; : > $OUT
for lib in "$@"; do
  echo "== $lib" >> $OUT
  BLU_HIP_LIB=$R/blu_amd/$lib timeout -k 10 300 python tools/batch_probe.py 1536 256 C3 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT || exit 1
done
cat $OUT
