#!/bin/bash
# A/B of builds of the two-wave kernel: libs given as arguments
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/w2_ab2.log; : > $OUT
for lib in "$@"; do
  echo "== $lib" >> $OUT
  BLU_HIP_LIB=$R/blu_amd/$lib BLU_PIVOT_KERNEL=3 timeout -k 10 300 python tools/batch_probe.py 1536 256 C3 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT || exit 1
  BLU_HIP_LIB=$R/blu_amd/$lib BLU_PIVOT_KERNEL=3 timeout -k 10 200 python tools/batch_probe.py 2048 256 C2 1 2>&1 | tail -1 >> $OUT || exit 1
done
cat $OUT
