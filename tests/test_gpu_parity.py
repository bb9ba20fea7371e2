"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libblu_hip.so (include/blu_hip.h), against the CPU oracle and the committed golden fixtures.

Bar (BASELINE.json north_star): permutations and all integer arrays bit-exact; L/U values within
1e-12 relative (in practice they are bit-identical: same IEEE operations in the same order).
"""
import os

import numpy as np
import pytest

from blu_amd import keys as K
from blu_amd.matrices import CONFIGS, simple_rs
from tests import util

pytestmark = pytest.mark.gpu
FSTATS = ("CONDEST_L", "CONDEST_U", "NORM_L", "NORM_U", "NORMEST_L_INV", "NORMEST_U_INV", "ONENORM", "INFNORM")


@pytest.fixture(scope="module")
def blu():
    import blu_amd
    if blu_amd.lib().blu_hip_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return blu_amd


def _both(blu, oracle, cp, ri, v, params=None, cap=None, block=None, allow_d3=False):
    """HIP path and oracle on the same input.  The oracle is the FAITHFUL restatement (reference defect D3
    included) and d3_hits == 0 is asserted on both sides, unless the test says allow_d3 (then, and only if
    the matrix does hit D3, the 64-bit-mask oracle is the checker: util.oracle_factorize)."""
    m = len(cp) - 1
    g = blu.BLU(m, len(ri))
    for k, val in (params or {}).items():
        g.set_param(k, val)
    if block:
        g.dbg_set_block(block)
    sg = g.factorize(cp[:-1], cp[1:], ri, v)
    o, so = util.oracle_factorize(oracle, cp, ri, v, params, cap, allow_d3)
    assert int(g.stat(50)) == o.d3_hits()  # stat 50 = cancellations at position >= 31 (D3 events), device count
    return g, o, sg, so


def _assert_parity(g, o, sg, so, cp, ri, v):
    assert sg == so
    fg, fo = g.get_factors(), o.get_factors()
    util.assert_same_factors(fg, fo)
    for c in util.COUNTERS:
        assert int(g.stat(getattr(K, "STAT_" + c))) == int(o.stat(getattr(K, "STAT_" + c))), c
    for c in ("MIN_PIVOT", "MAX_PIVOT"):
        a, b = g.stat(getattr(K, "STAT_" + c)), o.stat(getattr(K, "STAT_" + c))
        assert abs(a - b) <= util.RTOL * abs(b), c
    return fg


def test_simple_rs(blu, oracle):
    cp, ri, v, b, x = simple_rs()
    g, o, sg, so = _both(blu, oracle, cp, ri, v, cap=len(ri))
    assert sg == K.OK
    fg = _assert_parity(g, o, sg, so, cp, ri, v)
    assert fg["rowperm"][:2].tolist() == [5, 2] and fg["colperm"][:2].tolist() == [5, 2]
    np.testing.assert_allclose(g.solve_dense(b, "N"), x, rtol=0, atol=1e-13)
    np.testing.assert_allclose(g.solve_dense(b, "T"), x, rtol=0, atol=1e-13)


@pytest.mark.parametrize("path", util.golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_fixtures(blu, path):
    """Committed fixtures (tests/golden/, made by the oracle): no oracle call on this path."""
    gold = np.load(path)
    m = len(gold["colptr"]) - 1
    g = blu.BLU(m, len(gold["rowidx"]))
    st = g.factorize(gold["colptr"][:-1], gold["colptr"][1:], gold["rowidx"], gold["values"])
    assert st == int(gold["status"])
    f = g.get_factors()
    util.assert_same_factors(f, gold)
    for c in util.COUNTERS:
        assert int(g.stat(getattr(K, "STAT_" + c))) == int(gold["stat_" + c]), c
    util.check_factors(gold["colptr"], gold["rowidx"], gold["values"], f)


@pytest.mark.parametrize("m,k,bw,tri,offs,seed", [(300, 5, 4, 0.5, 0.3, 1), (800, 8, 8, 0.3, 0.5, 2), (1500, 10, 9, 0.5, 0.3, 5)])
@pytest.mark.parametrize("nzbias,search_rows", [(1, 0), (-1, 0), (1, 1), (-1, 1)])
@pytest.mark.parametrize("block", [1024, 256, 64])
def test_parameters_and_workgroup_sizes(blu, oracle, m, k, bw, tri, offs, seed, nzbias, search_rows, block):
    cp, ri, v = oracle.gen_lp_basis(m, k, bw, tri, seed, offs)
    g, o, sg, so = _both(blu, oracle, cp, ri, v, {K.PARAM_NZBIAS: nzbias, K.PARAM_SEARCH_ROWS: search_rows}, block=block)
    assert sg == K.OK
    _assert_parity(g, o, sg, so, cp, ri, v)


@pytest.mark.parametrize("params", [
    {K.PARAM_RELTOL: 0.5}, {K.PARAM_RELTOL: 1.0}, {K.PARAM_MAXSEARCH: 1}, {K.PARAM_MAXSEARCH: 8},
    {K.PARAM_DROPTOL: 1e-8}, {K.PARAM_ABSTOL: 1e-3}, {K.PARAM_PAD: 0, K.PARAM_STRETCH: 0.0},
], ids=lambda p: ",".join("%d=%g" % kv for kv in p.items()))
def test_tolerances_and_search_depth(blu, oracle, params):
    cp, ri, v = oracle.gen_lp_basis(1200, 8, 10, 0.4, 9, 0.6)
    g, o, sg, so = _both(blu, oracle, cp, ri, v, params)
    _assert_parity(g, o, sg, so, cp, ri, v)


def test_dense_matrix_pivot_any(blu, oracle):
    """Pivot columns with more than 64 off-diagonals take the pivot_any path (pivot.rs:114)."""
    rng = np.random.default_rng(3)
    m = 150
    A = rng.standard_normal((m, m)) + 5 * np.eye(m)
    cp = np.arange(0, m * m + 1, m, dtype=np.uint64)
    ri = np.tile(np.arange(m, dtype=np.uint64), m)
    v = A.T.reshape(-1).copy()
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == K.OK
    _assert_parity(g, o, sg, so, cp, ri, v)
    assert g.stat(55) > 0  # pivot_any executed
    xs = rng.standard_normal(m)
    np.testing.assert_allclose(g.solve_dense(A @ xs), xs, rtol=1e-8, atol=1e-9)


def test_long_rows_and_columns(blu, oracle):
    """Arrow matrix: one dense row and one dense column (long lines: multi-chunk paths, bitmap sorts)."""
    m = 700
    rng = np.random.default_rng(5)
    cols = []
    for j in range(m):
        e = {j: 4.0 + rng.random()}
        e[m - 1] = rng.random() + 0.1  # dense last row
        if j == m - 1:
            for i in range(m):
                e[i] = rng.random() + 0.1 if i != j else 10.0
        cols.append(sorted(e.items(), key=lambda t: (t[0] * 7919) % m))  # unsorted rows
    cp = np.zeros(m + 1, np.uint64)
    ri, v = [], []
    for j, c in enumerate(cols):
        for i, x in c:
            ri.append(i); v.append(x)
        cp[j + 1] = len(ri)
    ri, v = np.array(ri, np.uint64), np.array(v)
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == K.OK
    _assert_parity(g, o, sg, so, cp, ri, v)


def test_singular_matrices(blu, oracle):
    m = 6
    cols = {0: [(0, 2.0), (1, 1.0)], 1: [(1, 3.0)], 2: [(2, 1.5), (0, 0.5)], 3: [], 4: [(4, 1e-18)], 5: [(5, 4.0), (2, 1.0)]}
    cp, ri, v = [0], [], []
    for j in range(m):
        for (i, x) in cols[j]:
            ri.append(i); v.append(x)
        cp.append(len(ri))
    cp, ri, v = np.array(cp, np.uint64), np.array(ri, np.uint64), np.array(v)
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == K.WARNING_SINGULAR_MATRIX
    f = _assert_parity(g, o, sg, so, cp, ri, v)
    util.check_factors(cp, ri, v, f, rank=4)
    # larger: a well-conditioned basis with some columns zeroed / made tiny / duplicated structure removed
    cp, ri, v = oracle.gen_lp_basis(900, 7, 8, 0.5, 21, 0.4)
    v = v.copy()
    for j in (3, 77, 500, 899):
        v[int(cp[j]):int(cp[j + 1])] *= 1e-17
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == K.WARNING_SINGULAR_MATRIX
    f = _assert_parity(g, o, sg, so, cp, ri, v)
    assert int(g.stat(K.STAT_RANK)) == 896


def test_columns_that_sink_below_abstol_are_removed(blu, oracle):
    """remove_col path (pivot.rs:99-105, 1333-1381): make two columns identical so that elimination
    cancels one of them exactly."""
    cp, ri, v = oracle.gen_lp_basis(400, 6, 6, 0.0, 13, 0.5)
    cp = cp.copy(); ri = ri.copy(); v = v.copy()
    # make column 11 a copy of column 10's pattern and values (rank deficiency discovered mid-bump)
    a, b = int(cp[10]), int(cp[11])
    c, d = int(cp[11]), int(cp[12])
    n = min(b - a, d - c)
    cols = [(ri[int(cp[j]):int(cp[j + 1])].copy(), v[int(cp[j]):int(cp[j + 1])].copy()) for j in range(400)]
    cols[11] = (cols[10][0].copy(), cols[10][1].copy())
    ncp = np.zeros(401, np.uint64)
    nri, nv = [], []
    for j in range(400):
        nri.extend(cols[j][0].tolist()); nv.extend(cols[j][1].tolist())
        ncp[j + 1] = len(nri)
    nri, nv = np.array(nri, np.uint64), np.array(nv)
    g, o, sg, so = _both(blu, oracle, ncp, nri, nv)
    assert sg == K.WARNING_SINGULAR_MATRIX
    _assert_parity(g, o, sg, so, ncp, nri, nv)


def test_invalid_arguments_and_calls(blu):
    cp = np.array([0, 2, 4, 5], np.uint64)
    ri = np.array([0, 1, 1, 2, 2], np.uint64)
    v = np.ones(5)
    g = blu.BLU(3, 5)
    with pytest.raises(blu.BluError):  # get_factors before factorize: ErrorInvalidCall
        g.get_factors()
    with pytest.raises(blu.BluError):
        g.solve_dense(np.ones(3))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    bad = ri.copy(); bad[1] = 3
    assert g.factorize(cp[:-1], cp[1:], bad, v) == K.ERROR_INVALID_ARGUMENT
    with pytest.raises(blu.BluError):  # a failed factorize invalidates the old factors (lu.reset)
        g.get_factors()
    dup = ri.copy(); dup[1] = 0
    assert g.factorize(cp[:-1], cp[1:], dup, v) == K.ERROR_INVALID_ARGUMENT
    bb = cp[:-1].copy(); be = cp[1:].copy(); bb[0] = 2; be[0] = 0
    assert g.factorize(bb, be, ri, v) == K.ERROR_INVALID_ARGUMENT
    assert g.factorize(cp[:-1], cp[1:], ri, v) == K.OK  # handle stays usable
    assert blu.lib().blu_hip_new(-1, 1, 0) is None
    assert blu.lib().blu_hip_new(3, 3, 99) is None


def test_refactorize_same_handle_and_tiny(blu, oracle):
    g = blu.BLU(1, 1)
    assert g.factorize(np.array([0], np.uint64), np.array([1], np.uint64), np.array([0], np.uint64), np.array([2.5])) == K.OK
    assert g.solve_dense(np.array([5.0]))[0] == 2.0
    cp1, ri1, v1 = oracle.gen_lp_basis(600, 6, 6, 0.5, 1, 0.3)
    cp2, ri2, v2 = oracle.gen_lp_basis(600, 6, 6, 0.5, 2, 0.3)
    g = blu.BLU(600, max(len(ri1), len(ri2)))
    for cp, ri, v in ((cp1, ri1, v1), (cp2, ri2, v2), (cp1, ri1, v1)):
        o = oracle.OracleBLU(600, 32 * len(ri))
        assert g.factorize(cp[:-1], cp[1:], ri, v) == o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
        util.assert_same_factors(g.get_factors(), o.get_factors())


def test_small_initial_capacity_forces_device_side_growth(blu, oracle):
    """BLU::new(m, tiny): L/U/arenas start small; the pivot kernel exits with NEED_*, the host grows /
    compacts and relaunches (the device counterpart of blu.rs:105-115).  Results must not change."""
    cp, ri, v = oracle.gen_lp_basis(1500, 8, 16, 0.2, 3, 1.0)
    m = 1500
    g = blu.BLU(m, 16)  # b_nz hint far too small
    sg = g.factorize(cp[:-1], cp[1:], ri, v)
    o, so = util.oracle_factorize(oracle, cp, ri, v, cap=64 * len(ri), allow_d3=True)  # this matrix hits D3 once
    assert sg == so == K.OK and o.d3_hits() == g.stat(50) == 1
    util.assert_same_factors(g.get_factors(), o.get_factors())
    assert g.stat(K.STAT_DEV_RELAUNCHES) > 1


def test_config_c2_full_parity(blu, oracle):
    c = CONFIGS["C2"]
    cp, ri, v = oracle.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == K.OK and o.d3_hits() == 0
    f = _assert_parity(g, o, sg, so, cp, ri, v)
    util.check_factors(cp, ri, v, f)


def test_config_c3_full_size(blu, oracle):
    """BASELINE.json configs[2] at full size: parity against the oracle (it finishes in ~1 s) and the
    size-independent properties: permutations, triangular structure, L*U == B[p,q], solve round trip."""
    c = CONFIGS["C3"]
    cp, ri, v = oracle.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    g, o, sg, so = _both(blu, oracle, cp, ri, v, cap=16 * len(ri))
    assert sg == K.OK and o.d3_hits() == 0 and g.stat(50) == 0
    f = _assert_parity(g, o, sg, so, cp, ri, v)
    util.check_factors(cp, ri, v, f)
    m = c["m"]
    rng = np.random.default_rng(1)
    xs = rng.standard_normal(m)
    B = util.csc(cp, ri, v, m)
    x = g.solve_dense(B @ xs, "N")
    assert np.abs(B @ x - B @ xs).max() <= 1e-9 * np.abs(B @ xs).max()
    x = g.solve_dense(B.T @ xs, "T")
    assert np.abs(B.T @ x - B.T @ xs).max() <= 1e-9 * np.abs(B.T @ xs).max()


def test_config_c4_eight_bases_through_the_batch_entry(blu, oracle):
    """BASELINE.json configs[3]: the 8 independent 50k x 50k bases (seeds 1..8).  On the 8-GPU node each rank
    factorizes one of them; here all eight go through blu_hip_factorize_batch on the one GPU of the test box
    (same kernels, one workgroup per basis), and each must equal its own FAITHFUL oracle run, d3_hits == 0."""
    c = CONFIGS["C4"]
    mats = [oracle.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"] + b, c["offscale"]) for b in range(8)]
    hs = [blu.BLU(c["m"], len(ri)) for (cp, ri, v) in mats]
    sts = blu.blu.factorize_batch(hs, mats=mats)
    for b, (h, (cp, ri, v)) in enumerate(zip(hs, mats)):
        o, so = util.oracle_factorize(oracle, cp, ri, v, cap=16 * len(ri))  # asserts d3_hits == 0, faithful mask
        assert sts[b] == so == K.OK, (b, sts[b], so)
        assert h.stat(50) == 0 and o.d3_hits() == 0
        f = h.get_factors()
        util.assert_same_factors(f, o.get_factors())
        for cn in util.COUNTERS:
            assert int(h.stat(getattr(K, "STAT_" + cn))) == int(o.stat(getattr(K, "STAT_" + cn))), (b, cn)
        for cn in FSTATS + ("RESIDUAL_TEST",):
            assert h.stat(getattr(K, "STAT_" + cn)) == o.stat(getattr(K, "STAT_" + cn)), (b, cn)
        if b == 0:
            util.check_factors(cp, ri, v, f)
    # and one of them alone through the single-basis entry (what a rank of the 8-GPU run executes)
    cp, ri, v = mats[7]
    g, o, sg, so = _both(blu, oracle, cp, ri, v, cap=16 * len(ri))
    assert sg == so == K.OK
    _assert_parity(g, o, sg, so, cp, ri, v)


@pytest.mark.parametrize("no_fast", [False, True], ids=["fast", "general"])
def test_every_pivot_path_runs(blu, oracle, no_fast):
    """pivot_singleton_row (pivot.rs:835), pivot_singleton_col (:928), pivot_doubleton_col (:1027), pivot_small
    (:460), pivot_any (:114) and the empty-column step (factorize_bump.rs:24-33) are each provably taken: the
    per-path pivot counts of the device equal the oracle's and every one is positive over this set.  With
    no_fast the general implementations of pivot_small / pivot_singleton_col run instead of the LDS ones."""
    total = [0] * 6
    sets = [oracle.gen_lp_basis(*s) for s in ((2000, 3, 8, 0.0, 4, 0.3), (2000, 2, 3, 0.0, 4, 0.3), (2000, 3, 2, 0.5, 4, 0.3))]
    rng = np.random.default_rng(3)
    m = 150  # dense: pivot columns with more than 64 off-diagonals -> pivot_any
    A = rng.standard_normal((m, m)) + 5 * np.eye(m)
    sets.append((np.arange(0, m * m + 1, m, dtype=np.uint64), np.tile(np.arange(m, dtype=np.uint64), m), A.T.reshape(-1).copy()))
    cp, ri, v = oracle.gen_lp_basis(900, 7, 8, 0.5, 21, 0.4)  # numerically null columns -> empty-column steps
    v = v.copy()
    for j in (3, 77, 500, 899):
        v[int(cp[j]):int(cp[j + 1])] *= 1e-17
    sets.append((cp, ri, v))
    for cp, ri, v in sets:
        mm = len(cp) - 1
        g = blu.BLU(mm, len(ri))
        g.dbg_set_no_fast(no_fast)
        sg = g.factorize(cp[:-1], cp[1:], ri, v)
        o, so = util.oracle_factorize(oracle, cp, ri, v)
        assert sg == so and sg in (K.OK, K.WARNING_SINGULAR_MATRIX)
        util.assert_same_factors(g.get_factors(), o.get_factors())
        for kind in range(6):
            assert g.stat(51 + kind) == o.stat(51 + kind), (mm, kind, g.stat(51 + kind), o.stat(51 + kind))
            total[kind] += int(g.stat(51 + kind))
    assert all(t > 0 for t in total), total


def test_generators_agree(blu, oracle):
    """The library's generator (bench input) and the oracle's are the same function."""
    for args in ((50, 4, 3, 0.5, 1, 0.3), (3000, 10, 9, 0.5, 7, 0.3)):
        a = blu.gen_lp_basis(*args)
        b = oracle.gen_lp_basis(*args)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("regs", [3, 4])
def test_batch_of_independent_bases(blu, oracle, monkeypatch, regs):
    """blu_hip_factorize_batch: one workgroup per handle, all kernels launched once for the batch.
    Mixed sizes, one singular and one invalid member; every member must equal its own oracle run.
    regs: the wave kernels exist with the register budget of three waves per SIMD (`_r3`: what a batch whose workgroups
    are all resident at that budget runs -- any batch this small) and of four (BLU_PIVOT_REGS=4 here; statistic 120)."""
    if regs == 4:
        monkeypatch.setenv("BLU_PIVOT_REGS", "4")
    specs = [(300, 5, 4, 0.5, 1, 0.3), (1200, 8, 8, 0.5, 2, 0.3), (50, 4, 3, 0.5, 3, 0.3), (2000, 8, 8, 0.5, 4, 0.3),
             (700, 6, 6, 1.0, 5, 0.2), (900, 7, 8, 0.5, 21, 0.4), (400, 6, 6, 0.0, 13, 0.5), (1500, 8, 16, 0.2, 3, 1.0)]
    mats = [list(oracle.gen_lp_basis(*s)) for s in specs]
    mats[5][2] = mats[5][2].copy()
    for j in (3, 77, 500, 899):  # singular member
        mats[5][2][int(mats[5][0][j]):int(mats[5][0][j + 1])] *= 1e-17
    mats[2][1] = mats[2][1].copy()
    mats[2][1][5] = 50  # invalid member: row index out of range
    hs = [blu.BLU(len(m[0]) - 1, len(m[1]) if k != 7 else 16) for k, m in enumerate(mats)]  # member 7 starts far too small
    monkeypatch.delenv("BLU_PIVOT_REGS", raising=False)
    for block in (256, 64, 1024):
        sts = blu.blu.factorize_batch(hs, mats=[tuple(m) for m in mats], block=block)
        assert int(hs[0].stat(118)) == 3 and int(hs[0].stat(120)) == regs
        for k, (h, (cp, ri, v)) in enumerate(zip(hs, mats)):
            o, so = util.oracle_factorize(oracle, cp, ri, v, cap=64 * len(ri), allow_d3=(k == 7))  # member 7 hits D3 once
            assert sts[k] == so, (k, sts[k], so)
            if so in (K.OK, K.WARNING_SINGULAR_MATRIX):
                util.assert_same_factors(h.get_factors(), o.get_factors())
                for c in util.COUNTERS:
                    assert int(h.stat(getattr(K, "STAT_" + c))) == int(o.stat(getattr(K, "STAT_" + c))), (k, c)
        assert sts[2] == K.ERROR_INVALID_ARGUMENT and sts[5] == K.WARNING_SINGULAR_MATRIX
    # handles stay usable one by one after a batch, and solves work off batch results
    cp, ri, v = mats[1]
    B = util.csc(cp, ri, v, len(cp) - 1)
    xs = np.random.default_rng(0).standard_normal(len(cp) - 1)
    np.testing.assert_allclose(hs[1].solve_dense(B @ xs), xs, rtol=1e-7, atol=1e-9)
    assert hs[0].factorize(mats[0][0][:-1], mats[0][0][1:], mats[0][1], mats[0][2]) == K.OK


@pytest.mark.parametrize("window", [2048, 16384, 49152], ids=["2KB-windows", "16KB-buckets", "48KB-buckets"])
def test_batch_with_fewer_workgroups_than_bases(blu, oracle, monkeypatch, window):
    """The O(nnz) kernels of a batch (k_prep, k_setup, k_finish, k_stats_tail) run a fixed number of workgroups, each
    taking matrix after matrix (one per CU: a large batch).  Here 3 workgroups for 8 bases of different sizes (the
    grid is read when a handle is created), so every workgroup goes through several matrices with the same LDS; and the
    row / column counters of k_prep / k_finish go through their LDS window (a large batch: 144 KB), forced here and cut
    down to 2 KB so that every matrix takes several windows; with 16 KB and 48 KB the fills of k_prep / k_finish go through
    buckets of 672 / 2576 entries (k_bucket.h; statistic 119 says which did: a matrix with a line longer than a bucket's
    slack takes the windows), several to dozens per matrix: factors, counters and all statistics of every member as the
    oracle has them."""
    specs = [(300, 5, 4, 0.5, 1, 0.3), (1200, 8, 8, 0.5, 2, 0.3), (150, 4, 3, 0.5, 3, 0.3), (2000, 8, 8, 0.5, 4, 0.3),
             (700, 6, 6, 1.0, 5, 0.2), (900, 7, 8, 0.5, 21, 0.4), (400, 6, 6, 0.0, 13, 0.5), (2500, 10, 9, 0.5, 1, 0.3)]
    mats = [oracle.gen_lp_basis(*s) for s in specs]
    monkeypatch.setenv("BLU_BATCH_GRID", "3")
    monkeypatch.setenv("BLU_LDS_WINDOW", "2")
    monkeypatch.setenv("BLU_LDS_WINDOW_BYTES", str(window))
    hs = [blu.BLU(len(cp) - 1, len(ri)) for cp, ri, v in mats]
    for name in ("BLU_BATCH_GRID", "BLU_LDS_WINDOW", "BLU_LDS_WINDOW_BYTES"):
        monkeypatch.delenv(name)
    for rep in range(2):
        sts = blu.blu.factorize_batch(hs, mats=mats)
        for k, (h, (cp, ri, v)) in enumerate(zip(hs, mats)):
            o, so = util.oracle_factorize(oracle, cp, ri, v, allow_d3=True)
            assert sts[k] == so == K.OK, (k, sts[k], so)
            util.assert_same_factors(h.get_factors(), o.get_factors())
            for c in util.COUNTERS:
                assert int(h.stat(getattr(K, "STAT_" + c))) == int(o.stat(getattr(K, "STAT_" + c))), (k, c)
            for c in FSTATS + ("RESIDUAL_TEST", "MIN_PIVOT", "MAX_PIVOT"):
                assert h.stat(getattr(K, "STAT_" + c)) == o.stat(getattr(K, "STAT_" + c)), (k, c)
            assert int(h.stat(118)) == 3  # a batch this small: two waves per basis (k_pivot_loop_wave2)
        fills = [int(h.stat(119)) for h in hs]
        assert fills == [0] * 8 if window == 2048 else (all(f & 1 for f in fills) and sum(f == 3 for f in fills) >= 4), fills


def test_byte_counter_overflow_and_long_u_columns(blu, oracle, monkeypatch):
    """The one-byte counters of k_prep / k_finish (one LDS window for every line of the matrix) give up at 255 entries
    in a line and the matrix takes the 32-bit windows; a U column longer than a bucket's slack takes the window fill.
    Arrow-like bases -- a band plus dense last columns, so U ends in columns of several hundred entries and B has rows
    of 2-3 entries only, and the transposed shape (dense last ROWS: k_prep's counters overflow, U stays short) -- in a
    batch with a 48 KB window, against the oracle; statistic 119 tells which fill each kernel took."""
    rng = np.random.default_rng(5)
    m = 800
    mats = []
    for transposed in (False, True, False):
        ndense = 3 if len(mats) < 2 else 1
        cols = [[(j, 2.0 + float(rng.uniform(0, 1)))] + ([(j - 1, float(rng.uniform(-0.5, 0.5)))] if j else []) for j in range(m)]
        for d in range(ndense):
            jd = m - 1 - d
            for i in rng.choice(m - 10, 400 - 60 * d, replace=False):
                if transposed:  # dense ROW jd
                    if all(r != jd for r, _ in cols[int(i)]):
                        cols[int(i)].append((jd, float(rng.uniform(-0.3, 0.3))))
                elif all(r != int(i) for r, _ in cols[jd]):  # dense COLUMN jd
                    cols[jd].append((int(i), float(rng.uniform(-0.3, 0.3))))
        cp, ri, v = [0], [], []
        for j in range(m):
            for i, x in cols[j]:
                ri.append(i); v.append(x)
            cp.append(len(ri))
        mats.append((np.array(cp, np.uint64), np.array(ri, np.uint64), np.array(v)))
    monkeypatch.setenv("BLU_BATCH_GRID", "2")
    monkeypatch.setenv("BLU_LDS_WINDOW", "2")
    monkeypatch.setenv("BLU_LDS_WINDOW_BYTES", "49152")
    hs = [blu.BLU(m, len(ri)) for cp, ri, v in mats]
    for name in ("BLU_BATCH_GRID", "BLU_LDS_WINDOW", "BLU_LDS_WINDOW_BYTES"):
        monkeypatch.delenv(name)
    sts = blu.blu.factorize_batch(hs, mats=mats)
    fills = []
    for k, (h, (cp, ri, v)) in enumerate(zip(hs, mats)):
        o, so = util.oracle_factorize(oracle, cp, ri, v, allow_d3=True)
        assert sts[k] == so == K.OK, (k, sts[k], so)
        util.assert_same_factors(h.get_factors(), o.get_factors())
        for c in util.COUNTERS:
            assert int(h.stat(getattr(K, "STAT_" + c))) == int(o.stat(getattr(K, "STAT_" + c))), (k, c)
        for c in FSTATS + ("RESIDUAL_TEST", "MIN_PIVOT", "MAX_PIVOT"):
            if k == 1 and c in ("INFNORM", "RESIDUAL_TEST"):
                # rows of more than 256 entries: k_stats_tail sums them in storage order, the reference in pivot order of their
                # columns (statistics tail): equal to rounding, and the residual is rounding noise itself
                if c == "INFNORM":
                    assert abs(h.stat(K.STAT_INFNORM) - o.stat(K.STAT_INFNORM)) <= 1e-13 * o.stat(K.STAT_INFNORM), k
                continue
            assert h.stat(getattr(K, "STAT_" + c)) == o.stat(getattr(K, "STAT_" + c)), (k, c)
        fills.append(int(h.stat(119)))
        ucol = np.diff(o.get_factors()["u_colptr"]).max()
        brow = np.bincount(ri.astype(np.int64), minlength=m).max()
        assert (ucol > 336) == (not fills[-1] & 2), (k, ucol, fills)  # (slack of a bucket at this window: 336 entries, pivot included)
        assert (brow > 336) == (not fills[-1] & 1), (k, brow, fills)
    assert 0 in [f & 2 for f in fills] and 0 in [f & 1 for f in fills], fills  # both fallbacks were taken by some member


def test_bucket_fill_rows_of_every_length_and_duplicates(blu, oracle, monkeypatch):
    """k_bucket.h sorts a target line where it sits in LDS -- up to 32 entries by one thread, up to 256 by its wave -- and
    sees there whether a row of B holds a column twice (singletons.rs:195-197); longer rows leave in arrival order for the
    sort of the whole workgroup.  A batch with a 48 KB window whose members have dense rows of 40, 150 and 300 entries
    (all three regimes; slack of a bucket: 336), each once as it is and once with ONE entry repeated in such a row:
    status and factors as the oracle has them."""
    rng = np.random.default_rng(77)
    m = 900
    mats = []
    for rowlen, dup in ((40, False), (40, True), (150, False), (150, True), (300, False), (300, True), (7, True)):
        cp, ri, v = oracle.gen_lp_basis(m, 6, 8, 0.5, 11 + rowlen, 0.3)
        cols = [list(zip(ri[cp[j]:cp[j + 1]].tolist(), v[cp[j]:cp[j + 1]].tolist())) for j in range(m)]
        r = 450  # row r gets entries in `rowlen` columns
        for j in rng.choice(m, rowlen, replace=False):
            if all(i != r for i, _ in cols[j]):
                cols[j].append((r, float(rng.uniform(0.1, 1.0))))
        if dup:
            j = next(j for j in range(m) if any(i == r for i, _ in cols[j]))
            cols[j].append((r, 0.25))  # row r holds column j twice
        ncp, nri, nv = [0], [], []
        for j in range(m):
            for i, x in cols[j]:
                nri.append(i); nv.append(x)
            ncp.append(len(nri))
        mats.append((np.array(ncp, np.uint64), np.array(nri, np.uint64), np.array(nv)))
    monkeypatch.setenv("BLU_BATCH_GRID", "2")
    monkeypatch.setenv("BLU_LDS_WINDOW", "2")
    monkeypatch.setenv("BLU_LDS_WINDOW_BYTES", "49152")
    hs = [blu.BLU(m, len(ri)) for cp, ri, v in mats]
    for name in ("BLU_BATCH_GRID", "BLU_LDS_WINDOW", "BLU_LDS_WINDOW_BYTES"):
        monkeypatch.delenv(name)
    sts = blu.blu.factorize_batch(hs, mats=mats)
    for k, (h, (cp, ri, v)) in enumerate(zip(hs, mats)):
        o = oracle.OracleBLU(m, 64 * len(ri))
        o.set_fix_d3(True)
        so = o.factorize(cp[:-1], cp[1:], ri, v)
        assert sts[k] == so, (k, sts[k], so)
        assert so == (K.ERROR_INVALID_ARGUMENT if k in (1, 3, 5, 6) else K.OK), (k, so)
        assert int(h.stat(119)) & 1, k  # k_prep filled through buckets
        if so == K.OK:
            util.assert_same_factors(h.get_factors(), o.get_factors())


@pytest.mark.parametrize("spec", [(3000, 9, 10, 0.4, 17, 0.4), (2500, 10, 9, 0.5, 1, 0.3), (1800, 6, 30, 0.1, 9, 1.0)],
                         ids=["mixed", "c3-like", "wide-band"])
def test_one_wave_kernel_matches_workgroup_kernel_and_oracle(blu, oracle, spec):
    """A/B/C of the pivot kernels on one basis -- k_pivot_loop_wave (one wave per matrix, flattened line updates) and
    k_pivot_loop_wave2 (the same passes dealt out to two waves; the walk of the next search begun early), the kernels
    of a batch, against k_pivot_loop (sixteen waves) -- and all against the oracle; the flattened paths must have
    taken practically every small and singleton-column pivot."""
    cp, ri, v = oracle.gen_lp_basis(*spec)
    m = spec[0]
    a, b, c2 = blu.BLU(m, len(ri)), blu.BLU(m, len(ri)), blu.BLU(m, len(ri))
    a.dbg_set_pivot_kernel(1)
    b.dbg_set_pivot_kernel(2)
    c2.dbg_set_pivot_kernel(3)
    sa, sb, sc = (h.factorize(cp[:-1], cp[1:], ri, v) for h in (a, b, c2))
    o, so = util.oracle_factorize(oracle, cp, ri, v, allow_d3=True)
    assert sa == sb == sc == so == K.OK
    fa, fb, fc, fo = a.get_factors(), b.get_factors(), c2.get_factors(), o.get_factors()
    for k in util.INT_KEYS + util.VAL_KEYS:
        assert np.array_equal(fa[k], fb[k]) and np.array_equal(fa[k], fo[k]) and np.array_equal(fc[k], fo[k]), k
    for c in util.COUNTERS:
        assert a.stat(getattr(K, "STAT_" + c)) == b.stat(getattr(K, "STAT_" + c)) == c2.stat(getattr(K, "STAT_" + c)) == o.stat(getattr(K, "STAT_" + c)), c
    for kind in range(6):
        assert a.stat(51 + kind) == b.stat(51 + kind) == c2.stat(51 + kind) == o.stat(51 + kind), kind
    assert a.stat(110) + a.stat(111) >= 0.8 * (a.stat(52) + a.stat(54)) and b.stat(110) == 0  # (pivot rows beyond 64 entries: general paths)
    assert (c2.stat(110), c2.stat(111)) == (a.stat(110), a.stat(111))
    assert c2.stat(117) >= 0.9 * c2.stat(110) - 50  # the walk of the next search begun while the rows were being updated


def test_canonical_factors_inside_the_arena_and_in_their_own_buffers(blu, oracle, monkeypatch):
    """get_factors reads the canonical L / U from inside the (dead) column arena when they fit (ensure_out) and from
    buffers of their own otherwise (forced here with BLU_NO_OUT_ALIAS, read when the handle is created): same factors
    either way, also after a second factorize of another matrix on the same handles and after solves."""
    cp, ri, v = oracle.gen_lp_basis(2500, 10, 9, 0.5, 3, 0.3)
    cp2, ri2, v2 = oracle.gen_lp_basis(2500, 7, 12, 0.3, 8, 0.5)
    a = blu.BLU(2500, len(ri))
    monkeypatch.setenv("BLU_NO_OUT_ALIAS", "1")
    b = blu.BLU(2500, len(ri))
    monkeypatch.delenv("BLU_NO_OUT_ALIAS")
    for (c, r, x) in ((cp, ri, v), (cp2, ri2, v2), (cp, ri, v)):
        assert a.factorize(c[:-1], c[1:], r, x) == b.factorize(c[:-1], c[1:], r, x) == K.OK
        o, so = util.oracle_factorize(oracle, c, r, x, allow_d3=True)
        rhs = np.random.default_rng(1).standard_normal(2500)
        assert np.array_equal(a.solve_dense(rhs, "N"), o.solve_dense(rhs, "N"))  # (the solves read the canonical U)
        fa, fb, fo = a.get_factors(), b.get_factors(), o.get_factors()
        for k in util.INT_KEYS + util.VAL_KEYS:
            assert np.array_equal(fa[k], fb[k]) and np.array_equal(fa[k], fo[k]), k


def test_general_paths_only_matches_fast_paths(blu, oracle):
    """A/B of the two implementations of the pivot loop (LDS fast paths on / off)."""
    cp, ri, v = oracle.gen_lp_basis(3000, 9, 10, 0.4, 17, 0.4)
    a, b = blu.BLU(3000, len(ri)), blu.BLU(3000, len(ri))
    b.dbg_set_no_fast(True)
    assert a.factorize(cp[:-1], cp[1:], ri, v) == b.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    fa, fb = a.get_factors(), b.get_factors()
    for k in util.INT_KEYS + util.VAL_KEYS:
        assert np.array_equal(fa[k], fb[k]), k
    assert a.stat(54) > 0 and a.stat(K.STAT_NSEARCH_PIVOT) == b.stat(K.STAT_NSEARCH_PIVOT)


# (m = 6000: longer than the LDS window of the chain pipeline, k_chain.h -- operands gathered by the helper waves -- and
# with lines of more than 64 entries, which the chain wave takes straight from global memory)
@pytest.mark.parametrize("spec", [(300, 5, 4, 0.5, 1, 0.3), (2000, 8, 8, 0.5, 1, 0.3), (1000, 10, 12, 1.0, 11, 0.2),
                                  (1500, 8, 16, 0.2, 3, 1.0), (6000, 8, 8, 0.5, 1, 0.3)], ids=lambda s: "m%d" % s[0])
@pytest.mark.parametrize("block", [1024, 128])
def test_statistics_tail(blu, oracle, spec, block):
    """condest(L), condest(U), matrix norms, residual_test (factorize.rs:121-147): every
    sum is taken in the reference's order, so all of them -- residual_test, which is pure rounding noise,
    included -- are bit-identical to the oracle."""
    cp, ri, v = oracle.gen_lp_basis(*spec)
    g, o, sg, so = _both(blu, oracle, cp, ri, v, block=block, allow_d3=(spec[0] == 1500))  # the m=1500 basis hits D3 once
    assert sg == so == K.OK
    for c in FSTATS + ("RESIDUAL_TEST",):
        a, b = g.stat(getattr(K, "STAT_" + c)), o.stat(getattr(K, "STAT_" + c))
        assert a == b, (c, a, b)
    assert 0.0 < g.stat(K.STAT_RESIDUAL_TEST) < 1e-10
    a, b = g.stat(K.STAT_UPDATE_COST_DENOM), o.stat(K.STAT_UPDATE_COST_DENOM)
    assert abs(a - b) <= 1e-12 * abs(b)


def test_statistics_tail_singular_and_skip(blu, oracle):
    cp, ri, v = oracle.gen_lp_basis(900, 7, 8, 0.5, 21, 0.4)
    v = v.copy()
    for j in (3, 77, 500, 899):
        v[int(cp[j]):int(cp[j + 1])] *= 1e-17
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == so == K.WARNING_SINGULAR_MATRIX
    for c in FSTATS + ("RESIDUAL_TEST",):
        a, b = g.stat(getattr(K, "STAT_" + c)), o.stat(getattr(K, "STAT_" + c))
        assert a == b, (c, a, b)
    h = blu.BLU(900, len(ri))
    h.set_skip_stats(True)
    assert h.factorize(cp[:-1], cp[1:], ri, v) == K.WARNING_SINGULAR_MATRIX
    assert h.stat(K.STAT_CONDEST_U) == 0.0
    util.assert_same_factors(h.get_factors(), o.get_factors())


# ---- solve_sparse (src/solve_sparse.rs, lu/solve_sparse.rs, dfs.rs, solve_triangular.rs) -------------
def _sparse_rhs(m, nz, seed):
    rng = np.random.default_rng(seed)
    return rng.choice(m, nz, replace=False), rng.standard_normal(nz)


@pytest.mark.parametrize("spec", [(300, 5, 4, 0.5, 1, 0.3), (2000, 8, 8, 0.5, 1, 0.3), (1500, 8, 16, 0.2, 3, 1.0),
                                  (6000, 10, 9, 0.5, 5, 0.3)], ids=lambda s: "m%d" % s[0])
@pytest.mark.parametrize("trans", ["N", "T"])
def test_solve_sparse_matches_oracle(blu, oracle, spec, trans):
    """Pattern order (the DFS topological order / pivot order of the sequential branch), nzlhs and the
    values are identical to the reference restatement: bit-exact, including which branch runs."""
    cp, ri, v = oracle.gen_lp_basis(*spec)
    m = spec[0]
    g, o, sg, so = _both(blu, oracle, cp, ri, v, allow_d3=(m == 1500))
    assert sg == so == K.OK
    branches = set()
    for q, nz in enumerate((1, 2, 5, 17, max(1, m // 40), max(1, m // 8), m // 2)):
        ir, xr = _sparse_rhs(m, nz, 100 * q + 7)
        st_o, il_o, lhs_o = o.solve_sparse(ir, xr, trans)
        st_g = g.solve_sparse(ir, xr, trans)
        assert st_g == st_o == K.OK
        assert g.nzlhs == len(il_o), (nz, g.nzlhs, len(il_o))
        assert np.array_equal(g.ilhs[:g.nzlhs], il_o), nz
        assert np.array_equal(g.lhs, lhs_o), (nz, np.abs(g.lhs - lhs_o).max())
        branches.add(int(g.stat(43)))
        assert g.stat(K.STAT_L_FLOPS) == o.stat(K.STAT_L_FLOPS) and g.stat(K.STAT_U_FLOPS) == o.stat(K.STAT_U_FLOPS)
    assert 2 in branches
    if m in (2000, 6000):
        assert branches == {1, 2}  # both the symbolic/sparse and the sequential branch were taken


def test_solve_sparse_solves_the_system(blu, oracle):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    m = 3000
    cp, ri, v = oracle.gen_lp_basis(m, 9, 10, 0.4, 17, 0.4)
    A = sp.csc_matrix((v, ri.astype(np.int64), cp.astype(np.int64)), shape=(m, m))
    g = blu.BLU(m, len(ri))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    for trans in "NT":
        for nz in (1, 30, 900):
            ir, xr = _sparse_rhs(m, nz, nz)
            assert g.solve_sparse(ir, xr, trans) == K.OK
            b = np.zeros(m)
            b[ir] = xr
            x = spl.spsolve(A if trans == "N" else A.T.tocsc(), b)
            assert np.abs(g.lhs - x).max() <= 1e-10 * np.abs(x).max()
            mask = np.zeros(m, bool)
            mask[g.ilhs[:g.nzlhs]] = True
            assert np.array_equal(g.lhs != 0.0, mask)


def test_solve_sparse_rank_deficient_and_errors(blu, oracle):
    cp, ri, v = oracle.gen_lp_basis(900, 7, 8, 0.5, 21, 0.4)
    v = v.copy()
    for j in (3, 77, 500, 899):
        v[int(cp[j]):int(cp[j + 1])] *= 1e-17
    g, o, sg, so = _both(blu, oracle, cp, ri, v)
    assert sg == so == K.WARNING_SINGULAR_MATRIX
    for trans in "NT":
        for nz in (1, 6, 200):
            ir, xr = _sparse_rhs(900, nz, 3 * nz + 1)
            st_o, il_o, lhs_o = o.solve_sparse(ir, xr, trans)
            assert g.solve_sparse(ir, xr, trans) == st_o == K.OK
            assert np.array_equal(g.ilhs[:g.nzlhs], il_o) and np.array_equal(g.lhs, lhs_o)
    assert g.solve_sparse([900], [1.0]) == K.ERROR_INVALID_ARGUMENT  # index out of range (solve_sparse.rs:49-59)
    h = blu.BLU(10, 30)
    assert h.solve_sparse([1], [1.0]) == K.ERROR_INVALID_CALL  # no factorization yet (solve_sparse.rs:46)


def test_solve_sparse_golden_fixtures(blu):
    """The HIP path against the committed fixtures (no oracle involved)."""
    from blu_amd.matrices import simple_rs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "solve_sparse.npz"))
    for name in ("simple", "lp2000"):
        cp, ri, v = simple_rs()[:3] if name == "simple" else blu.gen_lp_basis(2000, 8, 8, 0.5, 1, 0.3)
        h = blu.BLU(len(cp) - 1, len(ri))
        assert h.factorize(cp[:-1], cp[1:], ri, v) == K.OK
        for n in range(int(g[name + "_ncases"])):
            key = "%s_%d" % (name, n)
            assert h.solve_sparse(g[key + "_irhs"], g[key + "_xrhs"], chr(int(g[key + "_trans"]))) == K.OK
            il = h.ilhs[:h.nzlhs]
            assert np.array_equal(il, g[key + "_ilhs"]) and np.array_equal(h.lhs[il], g[key + "_xlhs"]), key


@pytest.mark.parametrize("spec", [(300, 5, 4, 0.5, 1, 0.3), (2000, 8, 8, 0.5, 1, 0.3), (1500, 8, 16, 0.2, 3, 1.0), (6000, 8, 8, 0.5, 1, 0.3)],
                         ids=lambda s: "m%d" % s[0])
def test_solve_dense_identical_to_oracle(blu, oracle, spec):
    """solve_dense keeps the reference's operation order (solve_dense.rs:32-119): bit-identical results,
    both systems, full-rank and rank-deficient factors."""
    cp, ri, v = oracle.gen_lp_basis(*spec)
    m = spec[0]
    rng = np.random.default_rng(5)
    for singular in (False, True):
        vv = v.copy()
        if singular:
            for j in (3, m // 3, m - 1):
                vv[int(cp[j]):int(cp[j + 1])] *= 1e-17
        g, o, sg, so = _both(blu, oracle, cp, ri, vv, allow_d3=(m == 1500))
        assert sg == so == (K.WARNING_SINGULAR_MATRIX if singular else K.OK)
        for trans in "NT":
            b = rng.standard_normal(m)
            assert np.array_equal(g.solve_dense(b, trans), o.solve_dense(b, trans)), (singular, trans)



def test_chain_pipeline_and_one_workgroup_kernels_agree(blu, oracle, monkeypatch):
    """A single factorize runs its statistics and solve_dense on the chain pipeline (k_chain.hip); a batch, or a
    device without the LDS for it, on one workgroup per matrix (k_stats.hip, k_solve.hip).  Same numbers, bit for bit."""
    cp, ri, v = oracle.gen_lp_basis(6000, 8, 8, 0.5, 1, 0.3)
    m = 6000
    a = blu.BLU(m, len(ri))
    monkeypatch.setenv("BLU_HIP_NO_CHAIN", "1")
    b = blu.BLU(m, len(ri))
    monkeypatch.delenv("BLU_HIP_NO_CHAIN")
    assert a.factorize(cp[:-1], cp[1:], ri, v) == b.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    assert a.stat(108) > 0.0 and b.stat(108) == 0.0  # k_rows_grid ran for the first handle only
    for c in FSTATS + ("RESIDUAL_TEST",):
        assert a.stat(getattr(K, "STAT_" + c)) == b.stat(getattr(K, "STAT_" + c)), c
    rhs = np.random.default_rng(3).standard_normal(m)
    for trans in "NT":
        assert np.array_equal(a.solve_dense(rhs, trans), b.solve_dense(rhs, trans)), trans
    # solve_sparse reads the row-wise L the chain path built
    ir = np.array([5, 77, 4000], dtype=np.uint64)
    xr = np.array([1.0, -2.0, 0.5])
    for trans in "NT":
        assert a.solve_sparse(ir, xr, trans) == b.solve_sparse(ir, xr, trans) == K.OK
        assert a.nzlhs == b.nzlhs and np.array_equal(a.ilhs[:a.nzlhs], b.ilhs[:b.nzlhs]) and np.array_equal(a.lhs, b.lhs), trans
