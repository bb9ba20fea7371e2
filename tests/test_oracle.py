"""CPU tests of the oracle (oracle/, the C restatement of rwl/blu's factorize path).

The reference has no tests and no golden vectors, so the oracle is
pinned by: the known answer of examples/simple.rs, the pivot-sequence prefix
derived by hand from the reference source, self-consistency
(L*U == B[rowperm,colperm]) and the committed fixtures in tests/golden/.
"""
import os

import numpy as np
import pytest

from blu_amd import keys as K
from blu_amd.matrices import simple_rs
from tests import util


def _walk(flink, m, nz):
    out, j = [], int(flink[m + nz])
    while j < m:
        out.append(j)
        j = int(flink[j])
    assert j == m + nz
    return out


def test_simple_rs_known_answer(oracle):
    cp, ri, v, b, x = simple_rs()
    o = oracle.OracleBLU(10, len(ri))  # BLU::new(n, a.len()), examples/simple.rs:36
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    np.testing.assert_allclose(o.solve_dense(b, "N"), x, rtol=0, atol=1e-14)
    # A is symmetric, so the transposed system has the same solution
    np.testing.assert_allclose(o.solve_dense(b, "T"), x, rtol=0, atol=1e-14)
    f = o.get_factors()
    util.check_factors(cp, ri, v, f)
    assert f["rowperm"][:2].tolist() == [5, 2] and f["colperm"][:2].tolist() == [5, 2]
    c = util.counters(o.stat)
    assert c["RANK"] == 10 and c["MATRIX_NZ"] == 32 and c["BUMP_SIZE"] == 9 and c["BUMP_NZ"] == 31
    assert o.stat(K.STAT_RESIDUAL_TEST) < 1e-14


def test_simple_rs_hand_derived_trace(oracle):
    """SURVEY.md 8c item 2, derived by reading singletons.rs / setup_bump.rs / markowitz.rs / pivot.rs."""
    cp, ri, v, _, _ = simple_rs()
    m = 10
    o = oracle.OracleBLU(m, 32)
    # singletons ok, setup_bump asks for W: addmem_w = 152 - 32
    assert o.factorize_raw(cp[:-1], cp[1:], ri, v) == K.REALLOCATE
    # use the object API from scratch to walk the realloc loop, stopping before the first bump pivot
    o = oracle.OracleBLU(m, 32)
    o.set_stop(1)  # singleton phase peels exactly one pivot (row 5, col 5): rank0 = 1
    st = o.factorize(cp[:-1], cp[1:], ri, v)
    assert st == oracle.STOPPED
    assert int(o.stat(K.STAT_W_MEM)) == 228  # floor((32 + 120) * 1.5)
    c = util.counters(o.stat)
    assert c["RANK"] == 1 and c["MATRIX_NZ"] == 32 and c["BUMP_NZ"] == 31 and c["BUMP_SIZE"] == 9
    s = o.active_state()
    assert s["pinv"][5] == 0 and s["qinv"][5] == 0
    assert _walk(s["col_flink"], m, 2) == [2, 4]
    assert _walk(s["col_flink"], m, 3) == [0, 1, 6]
    assert _walk(s["col_flink"], m, 4) == [7, 8]
    assert _walk(s["col_flink"], m, 5) == [3, 9]
    # first bump pivot: (row 2, col 2), cost 1, doubleton column path; 3 columns searched
    o.set_stop(2)
    assert o.factorize_raw(None, None, None, None, c0ntinue=True) == oracle.STOPPED
    s = o.active_state()
    assert s["pinv"][2] == 1 and s["qinv"][2] == 1
    assert int(o.stat(K.STAT_NSEARCH_PIVOT)) == 3
    lu = o.partial_lu()
    assert lu["uidx"][lu["uptr"][1]:lu["uptr"][2]].tolist() == [9]
    assert lu["uval"][lu["uptr"][1]:lu["uptr"][2]].tolist() == [0.04]
    assert lu["lidx"][lu["lptr"][1]:lu["lptr"][2]].tolist() == [9]
    assert lu["lval"][lu["lptr"][1]] == 0.04 / 1.7
    assert s["colmax"][9] == 3.2 - 0.04 * (0.04 / 1.7)
    assert _walk(s["col_flink"], m, 4) == [7, 8, 9]


@pytest.mark.parametrize("path", util.golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_regression(oracle, path):
    g = np.load(path)
    m = len(g["colptr"]) - 1
    cap = len(g["rowidx"]) if "simple_rs" in path else 16 * len(g["rowidx"]) + 64
    o = oracle.OracleBLU(m, cap)
    st = o.factorize(g["colptr"][:-1], g["colptr"][1:], g["rowidx"], g["values"])
    assert st == int(g["status"])
    f = o.get_factors()
    for k in util.INT_KEYS:
        assert np.array_equal(f[k], g[k]), k
    for k in util.VAL_KEYS:
        assert np.array_equal(f[k], g[k]), k  # same code, same machine arithmetic: bit-identical
    for c in util.COUNTERS:
        assert int(o.stat(getattr(K, "STAT_" + c))) == int(g["stat_" + c]), c
    assert o.d3_hits() == 0
    util.check_factors(g["colptr"], g["rowidx"], g["values"], f)


def test_generator_fixture_matches(oracle):
    """The committed matrices are what the generator produces (guards generator drift)."""
    for path in util.golden_files():
        g = np.load(path)
        if "gen" not in g:
            continue
        m, k, bw, tri, offs, seed = g["gen"]
        cp, ri, v = oracle.gen_lp_basis(int(m), int(k), int(bw), float(tri), int(seed), float(offs))
        assert np.array_equal(cp, g["colptr"]) and np.array_equal(ri, g["rowidx"]) and np.array_equal(v, g["values"])


@pytest.mark.parametrize("m,k,bw,tri,offs,seed", [(300, 5, 4, 0.5, 0.3, 1), (800, 8, 8, 0.3, 0.5, 2), (1500, 10, 9, 0.5, 0.3, 5)])
@pytest.mark.parametrize("nzbias,search_rows", [(1, 0), (-1, 0), (1, 1)])
def test_selfconsistency_and_params(oracle, m, k, bw, tri, offs, seed, nzbias, search_rows):
    cp, ri, v = oracle.gen_lp_basis(m, k, bw, tri, seed, offs)
    o = oracle.OracleBLU(m, 16 * len(ri))
    o.set_param(K.PARAM_NZBIAS, nzbias)
    o.set_param(K.PARAM_SEARCH_ROWS, search_rows)
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    f = o.get_factors()
    util.check_factors(cp, ri, v, f)
    assert o.stat(K.STAT_RESIDUAL_TEST) < 1e-10
    rng = np.random.default_rng(0)
    xs = rng.standard_normal(m)
    B = util.csc(cp, ri, v, m)
    np.testing.assert_allclose(o.solve_dense(B @ xs, "N"), xs, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(o.solve_dense(B.T @ xs, "T"), xs, rtol=1e-6, atol=1e-8)


def test_small_capacity_realloc_path_gives_same_factors(oracle):
    """Results are layout independent: BLU::new(m, nnz) (realloc loop, file
    compression) and a roomy BLU give the same factors."""
    cp, ri, v = oracle.gen_lp_basis(400, 6, 6, 0.5, 4, 0.3)
    a = oracle.OracleBLU(400, len(ri))
    b = oracle.OracleBLU(400, 64 * len(ri))
    assert a.factorize(cp[:-1], cp[1:], ri, v) == K.OK and b.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    fa, fb = a.get_factors(), b.get_factors()
    for k in util.INT_KEYS + util.VAL_KEYS:
        assert np.array_equal(fa[k], fb[k]), k


def test_invalid_arguments(oracle):
    """singletons.rs:119-201: pointer order, index range, duplicates."""
    cp = np.array([0, 2, 4, 5], np.uint64)
    ri = np.array([0, 1, 1, 2, 2], np.uint64)
    v = np.ones(5)
    o = oracle.OracleBLU(3, 64)
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    bad = ri.copy(); bad[1] = 3  # index out of range
    assert o.factorize(cp[:-1], cp[1:], bad, v) == K.ERROR_INVALID_ARGUMENT
    dup = ri.copy(); dup[1] = 0  # duplicate (0,0),(0,0) in column 0
    assert o.factorize(cp[:-1], cp[1:], dup, v) == K.ERROR_INVALID_ARGUMENT
    be = cp[1:].copy(); bb = cp[:-1].copy(); be[0] = 0; bb[0] = 2  # b_end < b_begin
    assert o.factorize(bb, be, ri, v) == K.ERROR_INVALID_ARGUMENT
    # get_factors without a valid factorization -> ErrorInvalidCall (the reference panics on unwrap, get_factors.rs:59)
    with pytest.raises(RuntimeError):
        o.get_factors()


def test_singular_matrices(oracle):
    m = 6
    # column 3 empty, column 4 numerically zero, rows/cols otherwise diagonal + one coupling
    cols = {0: [(0, 2.0), (1, 1.0)], 1: [(1, 3.0)], 2: [(2, 1.5), (0, 0.5)], 3: [], 4: [(4, 1e-18)], 5: [(5, 4.0), (2, 1.0)]}
    cp, ri, v = [0], [], []
    for j in range(m):
        for (i, x) in cols[j]:
            ri.append(i); v.append(x)
        cp.append(len(ri))
    cp, ri, v = np.array(cp, np.uint64), np.array(ri, np.uint64), np.array(v)
    o = oracle.OracleBLU(m, 64)
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.WARNING_SINGULAR_MATRIX
    assert int(o.stat(K.STAT_RANK)) == 4
    f = o.get_factors()
    util.check_factors(cp, ri, v, f, rank=4)
    assert sorted(f["colperm"][4:].tolist()) == [3, 4]


def test_tiny_and_dense(oracle):
    o = oracle.OracleBLU(1, 4)
    assert o.factorize(np.array([0], np.uint64), np.array([1], np.uint64), np.array([0], np.uint64), np.array([2.5])) == K.OK
    assert o.solve_dense(np.array([5.0]))[0] == 2.0
    rng = np.random.default_rng(3)
    m = 80  # dense: pivot columns longer than 64 rows -> pivot_any path (pivot.rs:114)
    A = rng.standard_normal((m, m)) + 5 * np.eye(m)
    cp = np.arange(0, m * m + 1, m, dtype=np.uint64)
    ri = np.tile(np.arange(m, dtype=np.uint64), m)
    v = A.T.reshape(-1).copy()
    o = oracle.OracleBLU(m, 8 * m * m)
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    util.check_factors(cp, ri, v, o.get_factors())
    xs = rng.standard_normal(m)
    np.testing.assert_allclose(o.solve_dense(A @ xs), xs, rtol=1e-8, atol=1e-10)


def test_d3_defect_is_restated_and_detected(oracle):
    """the i32 cancellation mask.  On a matrix with drops at pivot-column position
    >= 32 the faithful oracle must NOT silently agree with the fixed one; d3_hits flags such inputs."""
    cp, ri, v = oracle.gen_lp_basis(2000, 8, 16, 0.5, 1, 1.0)
    o = oracle.OracleBLU(2000, 16 * len(ri))
    o.set_fix_d3(True)
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    assert o.d3_hits() > 0
    util.check_factors(cp, ri, v, o.get_factors(), tol=1e-7)


# ---- solve_sparse --------------------------------------------------------------------
def _golden_sparse_cases():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_sparse.npz"))
    for name in ("simple", "lp2000"):
        for n in range(int(g[name + "_ncases"])):
            key = "%s_%d" % (name, n)
            yield name, chr(int(g[key + "_trans"])), g[key + "_irhs"], g[key + "_xrhs"], g[key + "_ilhs"], g[key + "_xlhs"]


def _golden_sparse_matrix(oracle, name):
    from blu_amd.matrices import simple_rs
    return simple_rs()[:3] if name == "simple" else oracle.gen_lp_basis(2000, 8, 8, 0.5, 1, 0.3)


def test_solve_sparse_golden_and_scipy(oracle):
    """The oracle's sparse solve against its committed fixtures (pattern order included) and against
    scipy on the same systems; the dense solve of the same object agrees to rounding."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    objs = {}
    for name, trans, ir, xr, il_g, xl_g in _golden_sparse_cases():
        if name not in objs:
            cp, ri, v = _golden_sparse_matrix(oracle, name)
            m = len(cp) - 1
            o = oracle.OracleBLU(m, 16 * len(ri) + 64)
            assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
            objs[name] = (o, sp.csc_matrix((v, ri.astype(np.int64), cp.astype(np.int64)), shape=(m, m)), m)
        o, A, m = objs[name]
        st, il, lhs = o.solve_sparse(ir, xr, trans)
        assert st == K.OK and np.array_equal(il, il_g) and np.array_equal(lhs[il], xl_g)
        mask = np.zeros(m, bool)
        mask[il] = True
        assert np.array_equal(lhs != 0.0, mask) and len(set(il.tolist())) == len(il)
        b = np.zeros(m)
        b[ir] = xr
        x = spl.spsolve(A if trans == "N" else A.T.tocsc(), b)
        assert np.abs(lhs - x).max() <= 1e-11 * max(1.0, np.abs(x).max())
        assert np.abs(lhs - o.solve_dense(b, trans)).max() <= 1e-11 * max(1.0, np.abs(x).max())


def test_solve_sparse_errors(oracle):
    cp, ri, v = oracle.gen_lp_basis(50, 4, 4, 0.5, 1, 0.3)
    o = oracle.OracleBLU(50, 16 * len(ri))
    assert o.solve_sparse([1], [1.0])[0] == K.ERROR_INVALID_CALL      # solve_sparse.rs:46
    assert o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    assert o.solve_sparse([50], [1.0])[0] == K.ERROR_INVALID_ARGUMENT  # :49-59
    st, il, lhs = o.solve_sparse([], [])
    assert st == K.OK and len(il) == 0 and not lhs.any()
