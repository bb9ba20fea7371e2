"""Randomized sweep of the update path: random small bases, random LU parameters and capacity hints, a random sequence of
column replacements each -- the HIP path (blu_hip_solve_for_update / blu_hip_update / solves on updated factors) in LOCK
STEP with its CPU twin (oracle/orc_update.c, the intended algorithm: not reference-pinned): every status, every solution
pattern (order included) and value, ten counters after every update must be identical; every solve is also checked by its
backward error against the modified matrix held in scipy (tests/util_update.py).

   python tools/fuzz_update_gpu.py [--seed S] [--start A] [--count N] [--log FILE]      (needs a GPU)

Case n of seed S is the same whichever slice it is run in (one child generator per case).  tests/test_gpu_fuzz.py runs
slices in fresh child processes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blu_amd import keys as K  # noqa: E402
from oracle import orc  # noqa: E402
from tests import util_update as U  # noqa: E402


def draw(rng):
    c = {}
    c["m"] = int(rng.integers(8, 400))
    c["k"] = int(rng.integers(2, 9))
    c["bw"] = int(rng.integers(1, 24))
    c["tri"] = float(rng.choice([0.0, 0.3, 0.6, 1.0]))
    c["offs"] = float(rng.choice([0.1, 0.3, 0.6]))
    c["seed"] = int(rng.integers(1, 10**6))
    c["nupd"] = int(rng.integers(5, 80))
    c["check_every"] = int(rng.choice([1, 2, 5]))
    c["params"] = {K.PARAM_SPARSE_THRES: float(rng.choice([0.05, 0.0, 0.5, 1.0])), K.PARAM_MAXSEARCH: int(rng.choice([1, 3, 4])),
                   K.PARAM_PAD: int(rng.choice([4, 0, 9])), K.PARAM_STRETCH: float(rng.choice([0.3, 0.0, 1.0]))}
    c["small_hint"] = bool(rng.random() < 0.4)   # tiny b_nz hint: the arenas of the update path grow on the way
    c["upd_extra"] = int(rng.choice([-1, -1, 0, 3]))  # debug knob: arena slack of the update path (forces NEED_* round trips)
    return c


def tag_of(case, c):
    return "case %d: m=%d k=%d bw=%d tri=%g offs=%g seed=%d nupd=%d every=%d hint=%s extra=%d params=%s" % (
        case, c["m"], c["k"], c["bw"], c["tri"], c["offs"], c["seed"], c["nupd"], c["check_every"], "small" if c["small_hint"] else "nnz",
        c["upd_extra"], c["params"])


def run_case(blu_amd, case, c, log):
    m = c["m"]
    cp, ri, v = orc.gen_lp_basis(m, c["k"], c["bw"], c["tri"], c["seed"], c["offs"])
    tag = tag_of(case, c)
    log.write("start " + tag + "\n")
    log.flush()
    os.fsync(log.fileno())

    def setup(o):
        o.set_fix_d3(True)
        for key, val in c["params"].items():
            o.set_param(key, val)
    o, so = orc.OracleBLU.factorize_roomy(m, 64 * len(ri) + 1024, cp[:-1], cp[1:], ri, v, setup)
    g = blu_amd.BLU(m, 4 if c["small_hint"] else len(ri))
    for key, val in c["params"].items():
        g.set_param(key, val)
    if c["upd_extra"] >= 0:
        g.dbg_set_upd_extra(c["upd_extra"])
    sg = g.factorize(cp[:-1], cp[1:], ri, v)
    assert sg == so, (tag, sg, so)
    res = dict(done=0, max_residual=0.0)
    if sg == K.OK:
        f = g.get_factors()
        pair_row = np.zeros(m, np.int64)
        pair_row[f["colperm"]] = f["rowperm"]
        cols = U.columns_of(cp, ri, v)
        res = U.run_updates(g, cols, m, c["nupd"], np.random.default_rng(c["seed"] + 1), check_every=c["check_every"], pair_row=pair_row, twin=o)
        # stable solves: the backward errors stay at rounding level as long as the basis is not close to singular
        assert res["max_residual"] < 1e-6, (tag, res)
    g.close()
    log.write("done %d %s\n" % (case, {k: res[k] for k in ("done", "max_residual")}))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=4242)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--count", type=int, default=30)
    ap.add_argument("--log", default="")
    a = ap.parse_args()
    log = open(a.log, "w") if a.log else sys.stdout
    import blu_amd
    ndone = 0
    for case in range(a.start, a.start + a.count):
        c = draw(np.random.default_rng([a.seed, case]))
        ndone += run_case(blu_amd, case, c, log)["done"]
    msg = "all %d update cases of seed %d from %d identical; %d updates applied" % (a.count, a.seed, a.start, ndone)
    log.write(msg + "\n")
    log.flush()
    if log is not sys.stdout:
        print(msg)


if __name__ == "__main__":
    main()
