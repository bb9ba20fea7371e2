#!/bin/bash
# Runs a list of step-check cases on the GPU box; stops at the first case that hangs or is killed
# (never start another GPU step after a timeout), continues past plain mismatches (rc 1).
mkdir -p gpurun_out
make -s -C oracle liborc.so || exit 1
n=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  n=$((n+1))
  log=gpurun_out/check_$n.log
  echo "== $line" > $log
  timeout -k 10 ${CASE_TIMEOUT:-240} python tools/gpu_stepcheck.py $line >> $log 2>&1
  rc=$?
  echo "rc=$rc" >> $log
  echo "case $n [$line] rc=$rc"; tail -n 12 $log | cut -c1-300
  if [ $rc -ge 124 ]; then echo "case $n was killed/hung: stopping"; exit 2; fi
done
exit 0
