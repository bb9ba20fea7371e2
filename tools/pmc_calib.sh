#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE (tools/pmc_calib.hip): builds the program, runs it under rocprofv3 --pmc
# (one counter per pass; --pmc is never combined with tracing; the program itself follows "--") and writes
# gpurun_out/pmc_calib.json: counter value / useful bytes for each known-byte kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/gpurun_out/pmc_calib $R/tools/pmc_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$R/gpurun_out/calib_$c; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc $c -d $out -o calib --output-format csv -- $R/gpurun_out/pmc_calib > $out/run.log 2>&1 || { echo "pass $c failed"; tail -5 $out/run.log; exit 1; }
done
python3 - "$R" <<'PY'
import sys, glob, csv, collections, json
R = sys.argv[1]
useful = {"calib_stream16": {"FETCH_SIZE": 2**30, "WRITE_SIZE": 2**30}, "calib_gather<float>": {"FETCH_SIZE": 2**24 * 8 * 4},
          "calib_gather<double>": {"FETCH_SIZE": 2**24 * 8 * 8}, "calib_scatter<float>": {"WRITE_SIZE": 2**24 * 8 * 4},
          "calib_scatter<double>": {"WRITE_SIZE": 2**24 * 8 * 8}}
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(R + "/gpurun_out/calib_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        for name in useful:
            if name.split("<")[0] in k and (("<" not in name) or (name.split("<")[1][:-1] in k)):
                tot[(name, r["Counter_Name"])] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
out = {"unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 bytes on this stack: values below are counter * 1024 / useful bytes unless the raw ratio is near 1",
       "kernels": {}}
for (name, ctr), v in sorted(tot.items()):
    per = v / n[(name, ctr)]
    ub = useful[name].get(ctr)
    rec = out["kernels"].setdefault(name, {})
    rec[ctr] = {"counter_per_launch": per, "useful_bytes": ub, "ratio_raw": (per / ub) if ub else None, "ratio_x1024": (per * 1024 / ub) if ub else None}
json.dump(out, open(R + "/gpurun_out/pmc_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
