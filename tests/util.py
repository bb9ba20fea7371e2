"""Shared helpers of the parity tests."""
import glob
import os

import numpy as np
import scipy.sparse as sp

from blu_amd import keys as K

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT_KEYS = ("rowperm", "colperm", "l_colptr", "l_rowidx", "u_colptr", "u_rowidx")
VAL_KEYS = ("l_value", "u_value")
COUNTERS = ("RANK", "MATRIX_NZ", "BUMP_SIZE", "BUMP_NZ", "L_NZ", "U_NZ", "NSEARCH_PIVOT", "FACTOR_FLOPS", "RANKDEF")
RTOL = 1e-12  # north_star: L/U numeric values within 1e-12 relative


def golden_files():
    # factorize fixtures; solve_sparse.npz holds solve vectors and has its own tests
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("solve_"))


def csc(colptr, rowidx, values, m):
    return sp.csc_matrix((np.asarray(values, float), np.asarray(rowidx, np.int64), np.asarray(colptr, np.int64)), shape=(m, m))


def check_factors(colptr, rowidx, values, f, rank=None, tol=1e-10):
    """B[rowperm, colperm] == L*U (dependent columns replaced by unit columns), structure checks."""
    m = len(colptr) - 1
    B = csc(colptr, rowidx, values, m).tocsc()
    p, q = f["rowperm"], f["colperm"]
    assert sorted(p.tolist()) == list(range(m)) and sorted(q.tolist()) == list(range(m))
    L = csc(f["l_colptr"], f["l_rowidx"], f["l_value"], m)
    U = csc(f["u_colptr"], f["u_rowidx"], f["u_value"], m)
    # L unit lower with the diagonal first in each column, rows sorted; U upper with the diagonal last
    for k in range(m):
        a, b = f["l_colptr"][k], f["l_colptr"][k + 1]
        assert f["l_rowidx"][a] == k and f["l_value"][a] == 1.0
        assert np.all(np.diff(f["l_rowidx"][a:b]) > 0)
        a, b = f["u_colptr"][k], f["u_colptr"][k + 1]
        assert f["u_rowidx"][b - 1] == k
        assert np.all(np.diff(f["u_rowidx"][a:b]) > 0)
    PBQ = B[p, :][:, q].toarray() if m <= 4000 else None
    if rank is not None and rank < m and PBQ is not None:
        for k in range(rank, m):  # columns colperm[rank..] replaced by unit columns e_{rowperm[k]}
            PBQ[:, k] = 0.0
            PBQ[k, k] = 1.0
    if PBQ is not None:
        err = np.abs((L @ U).toarray() - PBQ).max()
        scale = max(1.0, np.abs(PBQ).max())
        assert err <= tol * scale * max(1.0, np.abs(U.toarray()).max()), err
    else:
        R = (L @ U) - B[p, :][:, q]
        assert abs(R).max() <= tol * max(1.0, abs(U).max())


def assert_same_factors(got, want, rtol=RTOL):
    """Bit-exact integer arrays; values within rtol relative (elementwise, with an absolute floor of rtol*max|.|)."""
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(got[k], np.int64), np.asarray(want[k], np.int64)), k
    for k in VAL_KEYS:
        g, w = np.asarray(got[k], float), np.asarray(want[k], float)
        assert g.shape == w.shape, k
        assert np.all(np.abs(g - w) <= rtol * np.maximum(np.abs(w), 1e-300)), (k, np.abs(g - w).max())


def counters(stat):
    return {c: int(stat(getattr(K, "STAT_" + c))) for c in COUNTERS}


def oracle_factorize(orc, cp, ri, v, params=None, cap=None, allow_d3=False):
    """Factorize with the CPU oracle as the REFERENCE restates it (faithful i32 cancellation mask, D3).

    The faithful restatement has no defined result when a cancellation lands at pivot-column position
    >= 31 (the reference corrupts its row file there), so a first run with the 64-bit mask counts such
    events (d3_hits).  d3_hits == 0: the faithful oracle is run and returned -- the comparison is with
    the reference's own semantics.  d3_hits > 0: only callers that say allow_d3=True get the 64-bit-mask
    run back (the HIP path documents the same deviation); everyone else fails."""
    m = len(cp) - 1
    cap = cap if cap else 32 * len(ri) + 1024

    def run(fix):
        def setup(o):
            o.set_fix_d3(fix)
            for k, val in (params or {}).items():
                o.set_param(k, val)
        # (factorize_roomy: a capacity too small for the bump would send the faithful restatement into the
        # reference's endless Reallocate loop, defect D5; it is raised until W never grows inside the bump)
        return orc.OracleBLU.factorize_roomy(m, cap, cp[:-1], cp[1:], ri, v, setup)

    o, st = run(True)
    if o.d3_hits() == 0:
        o, st = run(False)
        assert o.d3_hits() == 0
        return o, st
    assert allow_d3, "reference defect D3 would be hit (%d times): choose another matrix or pass allow_d3" % o.d3_hits()
    return o, st
