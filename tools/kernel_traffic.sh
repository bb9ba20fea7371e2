#!/bin/bash
# HBM traffic and duration of EVERY kernel of a batch step: bash tools/kernel_traffic.sh [cfg] [B] [hintdiv]   (GPU box)
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (never combined with tracing), durations from a third run
# with --kernel-trace --stats; per dispatch, the largest dispatch of each kernel (= a full warm step).  Output: table on
# stdout and gpurun_out/kernel_traffic_<cfg>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-C3}; NB=${2:-1536}; HD=${3:-2}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/ktraffic; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c -d $out/$c -o pmc --output-format csv -- python3 $R/tools/batch_probe.py $NB 256 $CFG $HD 3 > $out/run_$c.log 2>&1 || echo "pass $c rc $?"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o tr --output-format csv -- python3 $R/tools/batch_probe.py $NB 256 $CFG $HD 3 > $out/run_trace.log 2>&1 || echo "trace rc $?"
python3 - "$out" "$CFG" "$NB" <<'PY' | tee $R/gpurun_out/kernel_traffic_$2_$3.txt
import sys, glob, csv, collections
out, cfg, nb = sys.argv[1], sys.argv[2], sys.argv[3]
def short(n): return n.split("(")[0].replace("void ", "")[-44:]
val = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: val[short(r["Kernel_Name"])][c].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
print("# tools/kernel_traffic.sh %s %s: per kernel, its LARGEST dispatch (a full warm step of the batch): duration, FETCH_SIZE + WRITE_SIZE (KB counters x 1024; not doubled: scattered accesses, see profiles/r03_pmc_calibration.json), traffic rate" % (cfg, nb))
print("%-46s %9s %10s %10s %9s" % ("kernel", "seconds", "fetch GB", "write GB", "GB/s"))
for k in sorted(dur, key=lambda k: -max(dur[k])):
    d = max(dur[k]); f = max(val[k]["FETCH_SIZE"] or [0]) * 1024 / 1e9; w = max(val[k]["WRITE_SIZE"] or [0]) * 1024 / 1e9
    if d < 1e-3: continue
    print("%-46s %9.4f %10.2f %10.2f %9.0f" % (k, d, f, w, (f + w) / d))
PY
