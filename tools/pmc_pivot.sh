#!/bin/bash
# PMC counters of the pivot-loop kernel on the C3 basis (one pass per counter group; rocprofv3 --pmc
# must not be combined with tracing).  Usage on the GPU box: bash tools/pmc_pivot.sh "SQC_ICACHE_REQ SQC_ICACHE_MISSES" tag
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_$2
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc $1 --kernel-include-regex k_pivot_loop -d $out -o pmc --output-format csv -- python3 $R/tools/prof_phases.py C3 > $out/run.log 2>&1
python3 - "$out" <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pivot_loop" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(k, "total %.4g over %d dispatches" % (tot[k], n[k]))
PY
