#!/bin/bash
# HBM traffic of the pivot kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (never combined with tracing), per launch.  Writes
# profiles/pivot_loop_traffic.json keyed by the hash of the kernel sources; bench.py quotes `traffic` only
# while that hash matches the sources the library is built from.
#   bash tools/pmc_traffic.sh [BATCH]        (on the GPU box; BATCH = bases in flight of the batched leg, default 1280)
R=${GRAFT_REPO_ROOT:-/root/repo}
BATCH=${1:-1280}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/pmc_$c
  rm -rf $out; mkdir -p $out
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-include-regex k_pivot_loop -d $out -o pmc --output-format csv -- \
      python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-batch-sizes --batch $BATCH > $out/run.log 2>&1
  rc=$?
  # (rocprofv3 of this image sometimes dumps core in its exit handler AFTER the result files are written: the pass
  # counts if the counter file is there)
  if [ -z "$(find $out -name "*counter_collection.csv" | head -1)" ]; then echo "pass $c failed (rc $rc)"; tail -5 $out/run.log; exit 1; fi
  echo "pass $c done (rc $rc)"
done
python3 - "$R" "$BATCH" <<'PY'
import sys, glob, csv, json, collections, os
R, BATCH = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, R)
import bench
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(R + "/gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            k = ("k_pivot_loop_wave2" if "k_pivot_loop_wave2" in kn else "k_pivot_loop_wave" if "k_pivot_loop_wave" in kn else "k_pivot_loop_batch" if "k_pivot_loop_batch" in kn
                 else ("k_pivot_loop" if "k_pivot_loop" in kn else None))
            if k and r["Counter_Name"] == c:
                vals[k][c].append(float(r["Counter_Value"]))
# per launch: the FULL launches only -- the warm-up repetition of a batch relaunches the kernel after storage growth
# (ST_NEED_*), and such partial launches would pull an average down
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for k in vals:
    for c in vals[k]:
        full = [v for v in vals[k][c] if v >= 0.8 * max(vals[k][c])]
        tot[k][c] = sum(full); n[k][c] = len(full)
rec = {"kernel_source_sha16": bench.kernel_source_sha16(),
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --kernel-include-regex k_pivot_loop -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-batch-sizes --batch %d" % BATCH,
       "note": "FETCH_SIZE/WRITE_SIZE are reported in KB.  FETCH_SIZE is NOT doubled: the gfx950 x2 correction of MI355X_MICROARCH.md applies to wide "
               "(16 B/lane) coalesced streams; these kernels issue scattered 4- and 8-byte accesses -- see `calibration` (tools/pmc_calib.sh: what "
               "the counters report for known-byte scattered gathers / scatters and for a 16-byte stream on this box).  Infinity-Cache hits are "
               "counted, not excluded.",
       "kernels": {}}
for k in tot:
    f = tot[k]["FETCH_SIZE"] / max(1, n[k]["FETCH_SIZE"]); w = tot[k]["WRITE_SIZE"] / max(1, n[k]["WRITE_SIZE"])
    rec["kernels"][k] = {"config": "C3", "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
                         "hbm_bytes_per_launch": 1024.0 * (f + w), "launches_seen": [n[k]["FETCH_SIZE"], n[k]["WRITE_SIZE"]]}
    if k in ("k_pivot_loop_batch", "k_pivot_loop_wave", "k_pivot_loop_wave2"): rec["kernels"][k]["bases"] = BATCH
# calibration of the counters on known-byte kernels (tools/pmc_calib.sh), when it was collected on this box
try:
    rec["calibration"] = json.load(open(R + "/gpurun_out/pmc_calib.json"))
except (OSError, ValueError):
    rec["calibration"] = None
os.makedirs(R + "/gpurun_out", exist_ok=True)
json.dump(rec, open(R + "/gpurun_out/pivot_loop_traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
