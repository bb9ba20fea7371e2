#!/bin/bash
# SQ counters of the batch pivot kernel ($2 bases of config $1, default 1280 x C2; $3 = kernel name regex, default
# k_pivot_loop_batch -- pass k_pivot_loop_wave for the one-wave kernel, k_pivot_loop_wave2 for the two-wave one;
# $4 = divisor of the capacity hint, default 1, 2 for a batch that fills the card), two passes of 8 counters.
# rocprofv3 --pmc must not be combined with tracing; the program itself follows "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-C2}
NB=${2:-1280}
KRE=${3:-k_pivot_loop_batch}
HD=${4:-1}
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
n=0
for set in "$A" "$B"; do
  n=$((n+1)); out=$R/gpurun_out/bpmc_$n; rm -rf $out; mkdir -p $out
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex $KRE -d $out -o pmc --output-format csv -- python3 $R/tools/batch_probe.py $NB 256 $CFG $HD > $out/run.log 2>&1 || { echo "pass $n failed"; tail -5 $out/run.log; exit 1; }
done
python3 - "$R" "$KRE" <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/gpurun_out/bpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if sys.argv[2] in kn and not (sys.argv[2].endswith("_wave") and "_wave2" in kn):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-22s %.4g per dispatch (%d dispatches)" % (k, tot[k] / n[k], n[k]))
PY
