#!/bin/bash
# L2 (TCC) counters of one kernel of a batch: bash tools/l2_pmc.sh KERNEL_REGEX [cfg] [B] [hintdiv]   (GPU box)
# hit / miss / atomic requests at the L2 and what goes on to the fabric (EA): are the per-entry atomics of the O(nnz)
# kernels served from L2 or from HBM?  rocprofv3 --pmc is not combined with tracing; the program follows "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
KRE=${1:-k_prep}; CFG=${2:-C3}; NB=${3:-1536}; HD=${4:-2}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/l2pmc; rm -rf $out; mkdir -p $out
n=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "$KRE" -d $out/p$n -o pmc --output-format csv -- python3 $R/tools/batch_probe.py $NB 256 $CFG $HD > $out/run$n.log 2>&1 || { echo "pass $n failed"; tail -3 $out/run$n.log; }
done
python3 - "$out" <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot): print("%-42s %-24s %.4g per dispatch (%d)" % (k[0], k[1], tot[k] / n[k], n[k]))
PY
