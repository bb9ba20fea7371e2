#!/bin/bash
# HBM traffic of the pivot kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (never combined with tracing), per launch.  Writes
# gpurun_out/pivot_loop_traffic.json (copied to profiles/) keyed by the hash of the kernel sources; bench.py quotes
# `traffic` only while that hash matches the sources the library is built from.  Legs:
#   single   k_pivot_loop, one C3 basis                   (python3 bench.py --batch 0)
#   C3       the batch pivot kernel on BATCH C3-size bases (tools/batch_probe.py: the batch as bench.py builds it)
#   C4 / C2  the same on 3072 C4-size / 4096 C2-size bases (bench.py's batched_other_sizes)
#   bash tools/pmc_traffic.sh [BATCH]        (on the GPU box; BATCH = bases in flight of the C3-size leg, default 1536)
R=${GRAFT_REPO_ROOT:-/root/repo}
BATCH=${1:-1536}
cd /tmp && export TMPDIR=/tmp
pass() { # leg counter command...
  leg=$1; c=$2; shift 2
  out=$R/gpurun_out/pmc_${leg}_$c
  rm -rf $out; mkdir -p $out
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-include-regex k_pivot_loop -d $out -o pmc --output-format csv -- "$@" > $out/run.log 2>&1
  rc=$?
  # (rocprofv3 of this image sometimes dumps core in its exit handler AFTER the result files are written: the pass
  # counts if the counter file is there)
  if [ -z "$(find $out -name "*counter_collection.csv" | head -1)" ]; then echo "pass $leg $c failed (rc $rc)"; tail -5 $out/run.log; return 1; fi
  echo "pass $leg $c done (rc $rc)"
}
for c in FETCH_SIZE WRITE_SIZE; do
  pass single $c python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --batch 0 || exit 1
  pass C3 $c python3 $R/tools/batch_probe.py $BATCH 256 C3 2 3 || exit 1
  pass C4 $c python3 $R/tools/batch_probe.py 3072 256 C4 2 3 || exit 1
  pass C2 $c python3 $R/tools/batch_probe.py 4096 256 C2 2 3 || exit 1
done
python3 - "$R" "$BATCH" <<'PY'
import sys, glob, csv, json, collections, os
R, BATCH = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, R)
import bench
NB = {"single": 1, "C3": BATCH, "C4": 3072, "C2": 4096}
rec = {"kernel_source_sha16": bench.kernel_source_sha16(),
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex k_pivot_loop -- python3 bench.py --steps 2 --warmup 0 "
                  "--no-cpu-baseline --batch 0   |   python3 tools/batch_probe.py {%d 256 C3, 3072 256 C4, 4096 256 C2} 2 3" % BATCH,
       "note": "FETCH_SIZE/WRITE_SIZE are reported in KB.  FETCH_SIZE is NOT doubled: the gfx950 x2 correction of MI355X_MICROARCH.md applies to wide "
               "(16 B/lane) coalesced streams; these kernels issue scattered 4- and 8-byte accesses -- see `calibration` (tools/pmc_calib.sh: what "
               "the counters report for known-byte scattered gathers / scatters and for a 16-byte stream on this box).  Infinity-Cache hits are "
               "counted, not excluded.  Keys: `kernel` = the C3 legs (single basis; the C3-size batch), `kernel@C4`, `kernel@C2` = the other batched legs.",
       "kernels": {}}
for leg in ("single", "C3", "C4", "C2"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(R + "/gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (leg, c), recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r.get("Kernel_Name", "")
                k = ("k_pivot_loop_wave2" if "k_pivot_loop_wave2" in kn else "k_pivot_loop_wave" if "k_pivot_loop_wave" in kn else "k_pivot_loop_batch" if "k_pivot_loop_batch" in kn
                     else ("k_pivot_loop" if "k_pivot_loop" in kn else None))
                if k and r["Counter_Name"] == c:
                    vals[k][c].append(float(r["Counter_Value"]))
    # per launch: the FULL launches only -- the cold repetition of a batch relaunches the kernel after storage growth
    # (ST_NEED_*), and such partial launches would pull an average down
    for k in vals:
        tot, n = {}, {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            full = [v for v in vals[k][c] if v >= 0.8 * max(vals[k][c])] if vals[k][c] else []
            tot[c] = sum(full); n[c] = len(full)
        f = tot["FETCH_SIZE"] / max(1, n["FETCH_SIZE"]); w = tot["WRITE_SIZE"] / max(1, n["WRITE_SIZE"])
        cfg = "C3" if leg in ("single", "C3") else leg
        key = k if cfg == "C3" else "%s@%s" % (k, cfg)
        rec["kernels"][key] = {"config": cfg, "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
                               "hbm_bytes_per_launch": 1024.0 * (f + w), "launches_seen": [n["FETCH_SIZE"], n["WRITE_SIZE"]]}
        if leg != "single": rec["kernels"][key]["bases"] = NB[leg]
# calibration of the counters on known-byte kernels (tools/pmc_calib.sh), when it was collected on this box
try:
    rec["calibration"] = json.load(open(R + "/gpurun_out/pmc_calib.json"))
except (OSError, ValueError):
    rec["calibration"] = None
os.makedirs(R + "/gpurun_out", exist_ok=True)
json.dump(rec, open(R + "/gpurun_out/pivot_loop_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "calibration"}, indent=1))
PY
