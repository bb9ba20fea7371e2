"""GPU tests of the update path (run with -m gpu): blu_hip_solve_for_update, blu_hip_update and
the solves on an updated factorization, through the C ABI.

The reference's update code is defective as written, so the checks are (1) mathematical:
after every column replacement every solve has a rounding-level backward error against the modified matrix held
in scipy and stays close to a fresh factorization of it; (2) the CPU restatement of the same INTENDED algorithm
(oracle/orc_update.c, not reference-pinned) driven in lockstep must give bit-identical statuses, solution
patterns (order included), values and counters; (3) all three kinds of update occur.
"""
import numpy as np
import pytest

from blu_amd import keys as K
from tests import util_update as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blu():
    import blu_amd
    if blu_amd.lib().blu_hip_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return blu_amd


def _pair(blu, oracle, spec):
    cp, ri, v = oracle.gen_lp_basis(*spec)
    m = spec[0]
    g = blu.BLU(m, len(ri))
    o = oracle.OracleBLU(m, 64 * len(ri))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    f = g.get_factors()
    pair_row = np.zeros(m, np.int64)
    pair_row[f["colperm"]] = f["rowperm"]
    return g, o, U.columns_of(cp, ri, v), pair_row


@pytest.mark.parametrize("spec,nupd", [((300, 6, 6, 0.5, 1, 0.3), 150), ((1200, 8, 8, 0.5, 3, 0.3), 120), ((60, 4, 5, 0.3, 7, 0.5), 200),
                                      ((5000, 10, 9, 0.5, 2, 0.3), 60)], ids=["m300", "m1200", "m60", "m5000"])
def test_update_sequence_in_lockstep_with_the_cpu_twin(blu, oracle, spec, nupd):
    m = spec[0]
    g, o, cols, pair_row = _pair(blu, oracle, spec)
    rng = np.random.default_rng(spec[4])
    log = U.run_updates(g, cols, m, nupd, rng, refactor=lambda c: U.fresh_oracle(oracle, c, m), check_every=1 if m <= 1200 else 10,
                        pair_row=pair_row, twin=o)
    assert log["done"] >= nupd * 0.5 or log["hit_maximum_updates"], log
    assert log["max_residual"] <= 1e-8 and log["max_vs_fresh"] <= 1e8 and log["max_pivot_error"] <= 1e-8, log
    assert g.stat(K.STAT_NFORREST_TOTAL) > 0 and int(g.stat(K.STAT_NUPDATE)) == log["done"]
    if m == 300:
        assert g.stat(K.STAT_NSYMPERM_TOTAL) > 0 and g.stat(K.STAT_DEV_NUNSYMPERM_TOTAL) > 0
    with pytest.raises(blu.BluError):  # get_factors.rs:59: only a fresh factorization can be read out
        g.get_factors()
    # a new factorize on the same handle starts over
    cp, ri, v = U.csc_arrays(cols, m)
    assert g.factorize(cp[:-1], cp[1:], ri, v) in (K.OK,)
    assert g.stat(K.STAT_NUPDATE) == 0 and g.stat(K.STAT_NFORREST) == 0
    b = rng.standard_normal(m)
    assert U.backward_error(U.matrix_of(cols, m), g.solve_dense(b), b) < 1e-12


def test_permutation_updates_on_a_bidiagonal_basis(blu):
    """Hand-predictable: B = 2 I + superdiagonal.  Column 2 := 3 e_3 is an UNsymmetric permutation update (augmenting
    path 2 -> 3 -> 2), column 5 := 7 e_5 + e_1 a SYMMETRIC one; neither needs a row eta.  No oracle on this path."""
    m = 8
    cols = [(np.array([j] + ([j - 1] if j else []), np.int64), np.array([2.0] + ([1.0] if j else []))) for j in range(m)]
    cp, ri, v = U.csc_arrays(cols, m)
    g = blu.BLU(m, len(ri))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    for j, (ai, ax), key in ((2, ([3], [3.0]), K.STAT_DEV_NUNSYMPERM_TOTAL), (5, ([5, 1], [7.0, 1.0]), K.STAT_NSYMPERM_TOTAL)):
        assert g.solve_for_update([j], None, "T") == K.OK
        assert g.solve_for_update(ai, ax, "N") == K.OK
        before = g.stat(key)
        assert g.update(g.lhs[j]) == K.OK
        assert g.stat(key) == before + 1 and g.stat(K.STAT_NFORREST) == 0
        cols[j] = (np.array(ai, np.int64), np.array(ax))
        B = U.matrix_of(cols, m)
        b = np.arange(1.0, m + 1)
        assert U.backward_error(B, g.solve_dense(b, "N"), b) < 1e-15
        assert U.backward_error(B.T, g.solve_dense(b, "T"), b) < 1e-15
        for trans, A in (("N", B), ("T", B.T)):
            assert g.solve_sparse([4], [1.0], trans) == K.OK
            assert U.backward_error(A, g.lhs, np.eye(m)[4]) < 1e-15


def test_update_call_protocol(blu, oracle):
    cp, ri, v = oracle.gen_lp_basis(200, 5, 5, 0.5, 2, 0.3)
    g = blu.BLU(200, len(ri))
    assert g.solve_for_update([3], None, "T") == K.ERROR_INVALID_CALL  # no factorization yet
    assert g.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    assert g.update(1.0) == K.ERROR_INVALID_CALL                        # not prepared
    assert g.solve_for_update([200], None, "T") == K.ERROR_INVALID_ARGUMENT
    assert g.solve_for_update([1, 999], [1.0, 2.0], "N") == K.ERROR_INVALID_ARGUMENT
    assert g.solve_for_update([1], None, "N") == K.ERROR_ARGUMENT_MISSING
    assert g.solve_for_update([3], None, "T", want_solution=False) == K.OK and g.nzlhs == 0
    assert g.update(1.0) == K.ERROR_INVALID_CALL                        # forward solve still missing
    a, b = int(cp[4]), int(cp[5])
    assert g.solve_for_update(ri[a:b], v[a:b], "N") == K.OK and abs(g.lhs[3]) < 1e-12  # B^-1 (B e_4) = e_4
    assert g.update(g.lhs[3]) == K.ERROR_SINGULAR_UPDATE                # column 3 := column 4: singular, refused
    import scipy.sparse as sp
    B = sp.csc_matrix((v, ri.astype(np.int64), cp.astype(np.int64)), shape=(200, 200))
    assert U.backward_error(B, g.solve_dense(np.ones(200)), np.ones(200)) < 1e-13  # the old factorization is still valid


def test_maximum_updates_and_storage_growth(blu, oracle):
    """m Forrest-Tomlin updates exhaust the eta file: ErrorMaximumUpdates (solve_for_update.rs:87).  The handle was
    created with a tiny b_nz hint, so the arenas of the update path have to be grown by the host on the way."""
    spec = (24, 4, 4, 0.0, 5, 0.5)
    cp, ri, v = oracle.gen_lp_basis(*spec)
    m = spec[0]
    g = blu.BLU(m, 4)
    g.dbg_set_upd_extra(8)  # forces UPD_NEED_R / NEED_UC / NEED_W round trips
    o = oracle.OracleBLU(m, 256 * len(ri))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    cols = U.columns_of(cp, ri, v)
    log = U.run_updates(g, cols, m, 400, np.random.default_rng(11), check_every=5, stop_on_max=True, twin=o)
    assert log["hit_maximum_updates"] and int(g.stat(K.STAT_NFORREST)) == m, log
    assert log["max_residual"] <= 1e-7, log


def test_config_c5_column_replacement_stream(blu, oracle):
    """BASELINE.json configs[4] (the workload of `bench.py --config C5`) at its stated size -- the 100k basis and ALL
    1000 column modifications of the stream -- in lockstep with the CPU twin (every re-solve bit-identical with its
    pattern, identical update statuses and counters after every modification); then the modified basis is checked by
    backward error of dense and sparse solves."""
    from blu_amd.matrices import CONFIGS
    from blu_amd.workloads import column_modifications
    c = CONFIGS["C3"]
    cp, ri, v = oracle.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    m = c["m"]
    g = blu.BLU(m, len(ri))
    o = oracle.OracleBLU(m, 16 * len(ri))
    assert g.factorize(cp[:-1], cp[1:], ri, v) == o.factorize(cp[:-1], cp[1:], ri, v) == K.OK
    cols = U.columns_of(cp, ri, v)
    done = 0
    for j, rows, vals in column_modifications(cp, ri, 1000, c["offscale"]):
        a = U._sfu(g, [j], None, "T")
        U._same(a, U._sfu(o, [j], None, "T"), ("T", j))
        a = U._sfu(g, rows, vals, "N")
        U._same(a, U._sfu(o, rows, vals, "N"), ("N", j))
        xtbl = a[2][j]
        if abs(xtbl) < 1e-3:
            continue
        st = g.update(xtbl)
        assert st == o.update(xtbl) and st in (K.OK, K.ERROR_SINGULAR_UPDATE)
        if st == K.OK:
            cols[j] = (rows.astype(np.int64), vals)
            done += 1
            for key in (K.STAT_NFORREST, K.STAT_NSYMPERM_TOTAL, K.STAT_DEV_NUNSYMPERM_TOTAL, K.STAT_PIVOT_ERROR, K.STAT_U_NZ, K.STAT_R_NZ):
                assert g.stat(key) == o.stat(key), (key, j)
    assert done >= 900 and g.stat(K.STAT_NUPDATE) == done
    B = U.matrix_of(cols, m)
    b = np.random.default_rng(3).standard_normal(m)
    for trans, A in (("N", B), ("T", B.T)):
        x = g.solve_dense(b, trans)
        assert np.array_equal(x, o.solve_dense(b, trans))
        assert U.backward_error(A, x, b) < 1e-10
        ir, xr = np.array([17, 40000, 99999]), np.array([1.0, -2.0, 0.5])
        assert g.solve_sparse(ir, xr, trans) == K.OK
        bs = np.zeros(m)
        bs[ir] = xr
        assert U.backward_error(A, g.lhs, bs) < 1e-10
