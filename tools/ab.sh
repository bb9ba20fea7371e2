#!/bin/bash
# A/B of two builds of the library on the same GPU box: pivot-loop time of the C3 basis (3 steps after 1 warm-up),
# alternating, twice each.   bash tools/ab.sh libA.so libB.so [extra bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    BLU_HIP_LIB=$PWD/blu_amd/$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$L', 'ms/step %.1f  pivot %.1f ms  phases %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:round(v,1) for k,v in d['phases_last_step'].items()}))"
  done
done
