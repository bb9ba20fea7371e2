"""Randomized parity run of the BATCH entry (blu_hip_factorize_batch): random batches of 3..24 small bases of mixed
sizes with random generator parameters, per-handle LU parameters, capacity hints (small ones force device-side storage
growth and relaunches) and numerically null columns; every member must equal its own oracle run -- status, canonical
factors, counters, pivots per pivot routine, d3 events, statistics -- bit for bit.

Which pivot kernel runs: batches this small are resident with two waves per basis, so the library's default dispatch
takes k_pivot_loop_wave2 (statistic 118 == 3).  The one-wave kernel k_pivot_loop_wave -- the default beyond 2048
members -- is put under the same batches with BLU_PIVOT_KERNEL=1 in the environment (read when a handle is created;
tests/test_gpu_fuzz.py::test_batch_fuzz_slice_one_wave_kernel_forced), and in the shape bench.py times it by
tests/test_gpu_batch_as_benched.py.  The kernel that ran is printed with the result line.

   python tools/fuzz_batch_gpu.py [--seed S] [--start A] [--count N] [--log FILE]      (needs a GPU; the oracle is the checker)

Batch n of seed S is always the same whatever slice it is run in.  Draws are those of tools/fuzz_gpu.py (draw())."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blu_amd import keys as K  # noqa: E402
from oracle import orc  # noqa: E402
from tools import fuzz_gpu as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--count", type=int, default=20)
    ap.add_argument("--log", default="")
    ap.add_argument("--factors-only", action="store_true", help="skip the statistics (the CPU emulation build steps the pivot kernels only)")
    a = ap.parse_args()
    log = open(a.log, "w") if a.log else sys.stdout
    rng = np.random.default_rng(a.seed)
    blu_amd = None
    members, fast, kinds = 0, 0, [0] * 6
    ran = set()
    for batch in range(a.start + a.count):
        n = int(rng.integers(3, 25))
        cases = []
        for _ in range(n):
            c, mat = F.draw(rng)
            c["params"][K.PARAM_SEARCH_ROWS] = c["params"][K.PARAM_SEARCH_ROWS] if rng.random() < 0.3 else 0  # mostly the default search
            o, so = F.oracle_of(c, mat)
            cases.append((c, mat, o, so))
        if batch < a.start:
            continue
        if blu_amd is None:
            import blu_amd
            print("library:", blu_amd.lib().blu_hip_version().decode(), flush=True)
        log.write("start batch %d: %s\n" % (batch, " | ".join(F.tag_of(k, c) for k, (c, _, _, _) in enumerate(cases))))
        log.flush()
        if log is not sys.stdout:
            os.fsync(log.fileno())
        hs = []
        for c, mat, o, so in cases:
            g = blu_amd.BLU(c["m"], c["hint"])
            for key, val in c["params"].items():
                g.set_param(key, val)
            hs.append(g)
        sts = blu_amd.factorize_batch(hs, [m for _, m, _, _ in cases])
        for k, (g, (c, mat, o, so)) in enumerate(zip(hs, cases)):
            tag = "batch %d member %d: %s" % (batch, k, F.tag_of(k, c))
            assert sts[k] == so, (tag, sts[k], so)
            if so in (K.OK, K.WARNING_SINGULAR_MATRIX):
                fg, fo = g.get_factors(), o.get_factors()
                for key in F.INT_KEYS + F.VAL_KEYS:
                    assert np.array_equal(fg[key], fo[key]), (tag, key)
                for cn in F.COUNTERS + (() if a.factors_only else F.FSTATS):
                    x, y = g.stat(getattr(K, "STAT_" + cn)), o.stat(getattr(K, "STAT_" + cn))
                    assert x == y or (x != x and y != y), (tag, cn, x, y)
                assert int(g.stat(50)) == o.d3_hits(), (tag, "d3_hits")
                for kind in range(6):
                    assert g.stat(51 + kind) == o.stat(51 + kind), (tag, "pivot kind", kind)
                    kinds[kind] += int(o.stat(51 + kind))
                fast += int(g.stat(110)) + int(g.stat(111))
                ran.add(int(g.stat(118)))
            members += 1
        for g in hs:
            g.close()
        log.write("done %d\n" % batch)
    msg = "all %d batches of seed %d from %d identical (%d members; pivots by path %s, %d of them on the flattened paths; pivot kernels %s)" % (
        a.count, a.seed, a.start, members, kinds, fast, sorted(ran))
    log.write(msg + "\n")
    log.flush()
    if log is not sys.stdout:
        print(msg)


if __name__ == "__main__":
    main()
