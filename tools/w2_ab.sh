#!/bin/bash
# A/B of the batch pivot kernels on one box: kernels $1 (e.g. "1 3"), C3 x 1536 and C2 x 2048; then C2 x 4096 and C4 x 3072 with kernel 1
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/w2_ab.log; : > $OUT
for k in $1; do
  BLU_PIVOT_KERNEL=$k timeout -k 10 300 python tools/batch_probe.py 1536 256 C3 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT || exit 1
  BLU_PIVOT_KERNEL=$k timeout -k 10 200 python tools/batch_probe.py 2048 256 C2 1 2>&1 | tail -1 >> $OUT || exit 1
done
BLU_PIVOT_KERNEL=1 timeout -k 10 200 python tools/batch_probe.py 4096 256 C2 1 2>&1 | tail -1 >> $OUT
BLU_PIVOT_KERNEL=1 timeout -k 10 300 python tools/batch_probe.py 3072 256 C4 2>&1 | tail -1 >> $OUT
cat $OUT
