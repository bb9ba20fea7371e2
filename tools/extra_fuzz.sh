#!/bin/bash
# Sweeps of FRESH seeds outside the suite (run once on the final build of a round; results into gpurun_out/extra_fuzz.log):
# single bases on all three batch-capable pivot kernels, random batches (default dispatch and one-wave kernel forced),
# mid-size and large bases, the update path.  Each tool runs as a child with a timeout; a failure stops the script.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
set -o pipefail
L=gpurun_out/extra_fuzz.log; : > $L; n=0
run() { n=$((n+1)); echo "== $*" >> $L; timeout -k 10 ${T:-500} "$@" --log gpurun_out/extra_fuzz_$n.log 2>&1 | grep -v amdgpu.ids | tail -2 >> $L || { echo "FAILED: $*" >> $L; tail -5 $L; exit 1; }; }
run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 1)) --count 500
BLU_PIVOT_KERNEL=1 run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 2)) --count 300
BLU_PIVOT_KERNEL=3 run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 3)) --count 300
run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 4)) --count 40 --mmin 3000 --mmax 20000
BLU_PIVOT_KERNEL=1 run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 5)) --count 25 --mmin 3000 --mmax 20000
BLU_PIVOT_KERNEL=3 run python tools/fuzz_gpu.py --seed $((${SEED0:-40400} + 6)) --count 25 --mmin 3000 --mmax 20000
run python tools/fuzz_batch_gpu.py --seed $((${SEED0:-40400} + 7)) --count 40
BLU_PIVOT_KERNEL=1 run python tools/fuzz_batch_gpu.py --seed $((${SEED0:-40400} + 8)) --count 25
# the fills through buckets (k_bucket.h): the LDS window forced on (a batch this small would run without), 32 KB and natural size
BLU_LDS_WINDOW=2 BLU_LDS_WINDOW_BYTES=32768 BLU_BATCH_GRID=4 run python tools/fuzz_batch_gpu.py --seed $((${SEED0:-40400} + 10)) --count 40
BLU_LDS_WINDOW=2 run python tools/fuzz_batch_gpu.py --seed $((${SEED0:-40400} + 11)) --count 30
run python tools/fuzz_update_gpu.py --seed $((${SEED0:-40400} + 9)) --count 120
cat $L
