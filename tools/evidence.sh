#!/bin/bash
# The round's evidence in two GPU calls (outputs under gpurun_out/ev/; copied into profiles/ afterwards):
#   bash tools/evidence.sh 1   bench lines of C3 (default run, with the batched legs), C2, C4, C5; rocprofv3 kernel stats of
#                              the single C3 bench and of a C3 batch; the phase tables of the batch pivot kernels (prof build)
#                              and of k_prep / k_setup / k_finish (fprof build)
#   bash tools/evidence.sh 2   SQ counters of k_pivot_loop_wave2 / _wave; FETCH/WRITE_SIZE calibration; traffic of the pivot
#                              kernels on every bench leg (profiles/pivot_loop_traffic.json, keyed by the hash of the kernel
#                              sources); traffic of every kernel of a C3 batch step; L2 counters of the O(nnz) kernels
# Before the call (in the container; the built libraries travel with the snapshot): python -c 'import __graft_entry__ as g; g.build()'
# and make -C blu_amd/csrc prof fprof.  Afterwards: copy gpurun_out/ev/* into profiles/ (names as in BASELINE.md section 9).
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/ev
mkdir -p $E
cd $R
if [ "${1:-1}" = "1" ]; then
timeout -k 10 600 python bench.py > $E/bench_c3.json 2> $E/bench_c3.err || { echo "bench C3 failed"; tail -5 $E/bench_c3.err; exit 1; }
echo "bench C3 done"
timeout -k 10 300 python bench.py --config C2 --steps 5 --batch 4096 --no-batch-sizes > $E/bench_c2.json 2> $E/bench_c2.err || echo "bench C2 failed"
timeout -k 10 300 python bench.py --config C4 --steps 5 --batch 3072 --no-batch-sizes > $E/bench_c4.json 2> $E/bench_c4.err || echo "bench C4 failed"
timeout -k 10 300 python bench.py --config C5 --steps 1 --warmup 0 > $E/bench_c5.json 2> $E/bench_c5.err || echo "bench C5 failed"
echo "benches done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $E/prof_c3 -o c3 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --batch 0 --no-cpu-baseline > $E/prof_c3.log 2>&1 )
find $E/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/c3_kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $E/prof_b3 -o b3 --output-format csv -- python3 $R/tools/batch_probe.py 1536 256 C3 2 > $E/prof_b3.log 2>&1 )
find $E/prof_b3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/batch_c3_kernel_stats.csv
echo "kernel stats done"
BLU_HIP_LIB=$R/blu_amd/libblu_hip_prof.so timeout -k 10 300 python tools/wave_phases.py 1536 C3 2>&1 | grep -v "amdgpu.ids\|k_stats chain" > $E/wave_phases_c3.txt
BLU_PIVOT_KERNEL=1 BLU_HIP_LIB=$R/blu_amd/libblu_hip_prof.so timeout -k 10 300 python tools/wave_phases.py 3072 C4 2>&1 | grep -v "amdgpu.ids\|k_stats chain" > $E/wave_phases_c4_onewave.txt
BLU_HIP_LIB=$R/blu_amd/libblu_hip_fprof.so timeout -k 10 300 python tools/batch_probe.py 1536 256 C3 2 2 2>&1 | grep -v amdgpu.ids | tail -24 > $E/fill_phases_c3.txt
rm -rf $E/prof_c3 $E/prof_b3
else
bash tools/batch_pmc.sh C2 4096 k_pivot_loop_wave 2 > $E/wave_sq_counters.txt 2>&1
bash tools/batch_pmc.sh C3 1536 k_pivot_loop_wave2 2 > $E/wave2_sq_counters.txt 2>&1
bash tools/pmc_calib.sh > $E/calib.log 2>&1
bash tools/pmc_traffic.sh 1536 > $E/traffic.log 2>&1
cp gpurun_out/pivot_loop_traffic.json gpurun_out/pmc_calib.json $E/ 2>/dev/null
bash tools/kernel_traffic.sh C3 1536 2 > $E/kernel_traffic_c3.txt 2>&1
bash tools/l2_pmc.sh "k_prep|k_finish|k_setup|k_stats" C3 1536 2 > $E/l2_counters_batch_onnz.txt 2>&1
fi
echo "evidence part ${1:-1} done"; ls $E
