#!/bin/bash
# batch probes: "cfg:B:kernel:hintdiv:ENV=VAL,ENV=VAL" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/w2_ab.log; : > $OUT
for a in "$@"; do
  IFS=: read cfg B k hd ev <<< "$a"
  env BLU_PIVOT_KERNEL=$k $(echo ${ev:-X=1} | tr , ' ') timeout -k 10 300 python tools/batch_probe.py $B 256 $cfg ${hd:-2} 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$cfg $ev /" >> $OUT || exit 1
done
cat $OUT
