"""Randomized parity run: many small generated bases with random generator and LU parameters and random
workgroup sizes, HIP path against the CPU oracle -- status, canonical factors, counters, statistics,
solve_dense and solve_sparse must all be identical.

   python tools/fuzz_gpu.py [--seed S] [--start A] [--count N] [--log FILE]     (needs a GPU; the oracle is the checker)

Case n of seed S is always the same matrix and parameters, whichever slice [A, A+N) it is run in: the
cases before A are drawn (and factorized by the oracle alone, whose status decides how many random numbers a
case consumes) but never touch the GPU.  tests/test_gpu_fuzz.py runs the slices of the sweep, each in a
fresh child process with a timeout; the tag of a case is written to the log (flushed) BEFORE its first GPU
call, so a hang or a lost box leaves the parameters that caused it.

Round 1: 300 cases (seed 12345) identical; a 1500-case run (seed 777) ended with the loss of the GPU box.
Round 2: the same 1500 cases of seed 777 in 15 slices -- see DESIGN.md section 7 for the outcome."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blu_amd import keys as K  # noqa: E402
from oracle import orc  # noqa: E402

INT_KEYS = ("rowperm", "colperm", "l_colptr", "l_rowidx", "u_colptr", "u_rowidx")
VAL_KEYS = ("l_value", "u_value")
COUNTERS = ("RANK", "RANKDEF", "MATRIX_NZ", "BUMP_SIZE", "BUMP_NZ", "L_NZ", "U_NZ", "NSEARCH_PIVOT", "FACTOR_FLOPS")
FSTATS = ("MIN_PIVOT", "MAX_PIVOT", "CONDEST_L", "CONDEST_U", "NORM_L", "NORM_U", "NORMEST_L_INV", "NORMEST_U_INV",
          "ONENORM", "INFNORM", "RESIDUAL_TEST")


FACTORS_ONLY = [False]  # --factors-only: skip the statistics and the solves (the CPU emulation build steps the pivot kernels only)
M_RANGE = [20, 700]  # --mmin / --mmax: larger bases (ring wraps and far operands of the chain pipeline, k_chain.h)


def draw(rng):
    """The generator / LU parameters of one case (consumes the random stream exactly as round 1's tool did)."""
    c = {}
    c["m"] = m = int(rng.integers(M_RANGE[0], M_RANGE[1]))
    c["k"] = int(rng.integers(2, 12))
    c["bw"] = int(rng.integers(1, 40))
    c["tri"] = float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.0]))
    c["offs"] = float(rng.choice([0.1, 0.3, 0.6, 1.0]))
    c["seed"] = int(rng.integers(1, 10**6))
    cp, ri, v = orc.gen_lp_basis(m, c["k"], c["bw"], c["tri"], c["seed"], c["offs"])
    v = v.copy()
    c["null_cols"] = []
    if rng.random() < 0.25:  # some numerically null columns -> rank deficiency, remove_col
        for j in rng.choice(m, int(rng.integers(1, 4)), replace=False):
            v[int(cp[j]):int(cp[j + 1])] *= 1e-17
            c["null_cols"].append(int(j))
    c["params"] = {K.PARAM_NZBIAS: int(rng.choice([1, -1, 0, 3])), K.PARAM_SEARCH_ROWS: int(rng.random() < 0.3),
                   K.PARAM_MAXSEARCH: int(rng.choice([1, 2, 3, 4, 7])), K.PARAM_RELTOL: float(rng.choice([0.1, 0.01, 0.5, 1.0])),
                   K.PARAM_PAD: int(rng.choice([4, 0, 1, 9])), K.PARAM_STRETCH: float(rng.choice([0.3, 0.0, 1.0])),
                   K.PARAM_SPARSE_THRES: float(rng.choice([0.05, 0.0, 0.5, 1.0]))}
    c["block"] = int(rng.choice([64, 128, 256, 512, 1024]))
    c["hint"] = len(ri) if rng.random() < 0.7 else max(1, len(ri) // int(rng.integers(2, 30)))  # small: device-side growth
    return c, (cp, ri, v)


def draw_solves(rng, m):
    out = []
    for trans in "NT":
        b = rng.standard_normal(m)
        nz = int(rng.integers(1, max(2, m // 3)))
        ir = rng.choice(m, nz, replace=False)
        xr = rng.standard_normal(nz)
        out.append((trans, b, ir, xr))
    return out


def tag_of(case, c):
    return "case %d: m=%d k=%d bw=%d tri=%g offs=%g seed=%d null=%s block=%d hint=%d params=%s" % (
        case, c["m"], c["k"], c["bw"], c["tri"], c["offs"], c["seed"], c["null_cols"], c["block"], c["hint"], c["params"])


def oracle_of(c, mat):
    cp, ri, v = mat

    def setup(o):
        o.set_fix_d3(True)  # random matrices do hit D3 now and then; d3_hits is compared with the device's count
        for key, val in c["params"].items():
            o.set_param(key, val)
    # factorize_roomy: with too small a capacity the faithful restatement enters the reference's endless
    # Reallocate loop (defect D5) and eats the host's memory -- that, in the ORACLE, is what took the GPU box
    # down in round 1's 1500-case run (case 619 of seed 777); the capacity is raised until W never grows
    # inside the bump.
    return orc.OracleBLU.factorize_roomy(c["m"], 64 * len(ri) + 1024, cp[:-1], cp[1:], ri, v, setup)


def run_case(blu_amd, case, c, mat, o, so, solves, log):
    cp, ri, v = mat
    tag = tag_of(case, c)
    log.write("start " + tag + "\n")
    log.flush()
    os.fsync(log.fileno())
    g = blu_amd.BLU(c["m"], c["hint"])
    for key, val in c["params"].items():
        g.set_param(key, val)
    g.dbg_set_block(c["block"])
    sg = g.factorize(cp[:-1], cp[1:], ri, v)
    assert sg == so, (tag, sg, so)
    if sg in (K.OK, K.WARNING_SINGULAR_MATRIX):
        fg, fo = g.get_factors(), o.get_factors()
        for key in INT_KEYS + VAL_KEYS:
            assert np.array_equal(fg[key], fo[key]), (tag, key)
        for cn in COUNTERS + (() if FACTORS_ONLY[0] else FSTATS):
            a, b = g.stat(getattr(K, "STAT_" + cn)), o.stat(getattr(K, "STAT_" + cn))
            assert a == b or (a != a and b != b), (tag, cn, a, b)
        assert int(g.stat(50)) == o.d3_hits(), (tag, "d3_hits")
        for kind in range(6):
            assert g.stat(51 + kind) == o.stat(51 + kind), (tag, "pivot kind", kind)
        for trans, b, ir, xr in ([] if FACTORS_ONLY[0] else solves):
            assert np.array_equal(g.solve_dense(b, trans), o.solve_dense(b, trans), equal_nan=True), (tag, "solve_dense", trans)
            st_o, il, lhs = o.solve_sparse(ir, xr, trans)
            assert g.solve_sparse(ir, xr, trans) == st_o == K.OK, (tag, "solve_sparse status")
            assert np.array_equal(g.ilhs[:g.nzlhs], il) and np.array_equal(g.lhs, lhs, equal_nan=True), (tag, "solve_sparse", trans)
    g.close()
    log.write("done %d\n" % case)
    return [int(o.stat(51 + k)) for k in range(6)], o.d3_hits()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--log", default="")
    ap.add_argument("--mmin", type=int, default=20)
    ap.add_argument("--mmax", type=int, default=700)
    ap.add_argument("--factors-only", action="store_true")
    a = ap.parse_args()
    FACTORS_ONLY[0] = a.factors_only
    M_RANGE[0], M_RANGE[1] = a.mmin, a.mmax
    log = open(a.log, "w") if a.log else sys.stdout
    rng = np.random.default_rng(a.seed)
    blu_amd = None
    kinds, d3 = [0] * 6, 0
    for case in range(a.start + a.count):
        c, mat = draw(rng)
        o, so = oracle_of(c, mat)
        solves = draw_solves(rng, c["m"]) if so in (K.OK, K.WARNING_SINGULAR_MATRIX) else []
        if case < a.start:
            continue
        if blu_amd is None:
            import blu_amd  # first GPU use only when the slice begins
            print("library:", blu_amd.lib().blu_hip_version().decode(), flush=True)
        kk, dd = run_case(blu_amd, case, c, mat, o, so, solves, log)
        kinds = [x + y for x, y in zip(kinds, kk)]
        d3 += dd
    log.write("all %d cases of seed %d from %d identical; pivots by path %s, d3 events %d\n" % (a.count, a.seed, a.start, kinds, d3))
    log.flush()
    if log is not sys.stdout:
        print("all %d cases of seed %d from %d identical; pivots by path %s, d3 events %d" % (a.count, a.seed, a.start, kinds, d3))


if __name__ == "__main__":
    main()
