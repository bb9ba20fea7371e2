#!/bin/bash
# A/B of batch runs on one GPU box: bash tools/batch_env_ab.sh cfg:B:kernel:hintdiv:ENV=VAL,ENV=VAL ...  (one tools/batch_probe.py line each;
# kernel = BLU_PIVOT_KERNEL, hintdiv = divisor of the capacity hint, ENV = e.g. BLU_BATCH_GRID=128)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/batch_env_ab.log; : > $OUT
for a in "$@"; do
  IFS=: read cfg B k hd ev <<< "$a"
  env BLU_PIVOT_KERNEL=$k $(echo ${ev:-X=1} | tr , ' ') timeout -k 10 300 python tools/batch_probe.py $B 256 $cfg ${hd:-2} 2>&1 | grep -v amdgpu.ids | tail -1 | awk -v p="$cfg $ev " '{print p $0}' >> $OUT || exit 1
done
cat $OUT
