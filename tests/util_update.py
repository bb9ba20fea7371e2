"""Driver of the update tests: a sequence of column replacements, each prepared by the two solve_for_update
calls and applied by update(), checked against the modified matrix held in scipy.  Works on any object with
the BLU methods (solve_for_update, update, solve_dense, solve_sparse, stat): the CPU oracle (tests/
test_update_oracle.py) and the HIP implementation (tests/test_gpu_update.py)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from blu_amd import keys as K


def columns_of(cp, ri, v):
    return [(ri[int(cp[j]):int(cp[j + 1])].astype(np.int64).copy(), v[int(cp[j]):int(cp[j + 1])].copy()) for j in range(len(cp) - 1)]


def matrix_of(cols, m):
    idx = np.concatenate([c[0] for c in cols])
    val = np.concatenate([c[1] for c in cols])
    ptr = np.zeros(m + 1, np.int64)
    ptr[1:] = np.cumsum([len(c[0]) for c in cols])
    return sp.csc_matrix((val, idx, ptr), shape=(m, m))


def csc_arrays(cols, m):
    B = matrix_of(cols, m)
    return B.indptr.astype(np.uint64), B.indices.astype(np.uint64), B.data.copy()


def fresh_oracle(orc, cols, m):
    cp, ri, v = csc_arrays(cols, m)
    # 64-bit cancellation mask: the modified bases do hit the reference's defect D3 now and then, where the
    # faithful restatement aborts as the reference panics; roomy: never into the endless loop of D5
    o, st = orc.OracleBLU.factorize_roomy(m, 64 * len(ri) + 1024, cp[:-1], cp[1:], ri, v, lambda o: o.set_fix_d3(True))
    return o if st == K.OK else None


def new_column(rng, cols, m, j, pair_row=None):
    """An incoming column for position j.  Mostly: a sparse combination that keeps some rows of the outgoing
    column (so the new basis is nonsingular with high probability) plus a few random rows -> Forrest-Tomlin
    updates.  Sometimes a scaled copy of the outgoing column -> the spiked U stays symmetrically permuted
    triangular (update.rs:609-650).  Sometimes a multiple of the unit vector of the row that was paired (in the
    initial factorization) with another column k which has an entry in j's own pivot row -> a spike with a zero
    diagonal and the augmenting path j -> k -> j (unsymmetric permutation, update.rs:651-818)."""
    old_i, old_v = cols[j]
    kind = rng.random()
    if kind < 0.12:
        return old_i.copy(), old_v * (0.5 + rng.random())
    if kind < 0.40 and pair_row is not None:
        i = pair_row[j]
        cand = [k for k in rng.permutation(m) if k != j and i in cols[k][0]]
        if cand:
            k = int(cand[0])
            return np.array([pair_row[k]], np.int64), np.array([1.0 + rng.random()])
    kind = rng.random()
    rows = {}
    if kind < 0.55 and len(old_i):
        keep = rng.random(len(old_i)) < 0.7
        for i, x in zip(old_i[keep], old_v[keep]):
            rows[int(i)] = float(x) * (0.5 + rng.random())
    elif kind < 0.8:
        jj = int(rng.integers(0, m))
        for i, x in zip(*cols[jj]):
            rows[int(i)] = float(x) * (0.5 + rng.random())
    for i in rng.choice(m, int(rng.integers(1, 4)), replace=False):
        rows[int(i)] = rows.get(int(i), 0.0) + float(rng.standard_normal())
    idx = np.array(sorted(rows), np.int64)
    rng.shuffle(idx)
    return idx, np.array([rows[int(i)] for i in idx])


def backward_error(A, x, b):
    """Normwise backward error |A x - b| / (|A| |x| + |b|) (infinity norms): rounding-level for a stable solve
    whatever the conditioning of A."""
    den = abs(A).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()
    return np.abs(A @ x - b).max() / max(den, 1e-300)


def _sfu(h, irhs, xrhs, trans):
    """solve_for_update on either kind of object -> (status, ilhs, lhs)."""
    out = h.solve_for_update(irhs, xrhs, trans)
    if isinstance(out, tuple):
        return out
    return (out, h.ilhs[:h.nzlhs].copy(), h.lhs.copy()) if out == K.OK else (out, None, None)


def _ss(h, irhs, xrhs, trans):
    out = h.solve_sparse(irhs, xrhs, trans)
    if isinstance(out, tuple):
        return out
    return (out, h.ilhs[:h.nzlhs].copy(), h.lhs.copy()) if out == K.OK else (out, None, None)


def _same(a, b, what):
    """(status, pattern, values) of the device and of its CPU twin: identical, bit for bit, pattern order included."""
    assert a[0] == b[0], (what, a[0], b[0])
    if a[0] == K.OK and a[1] is not None:
        assert np.array_equal(a[1], b[1]), (what, "pattern")
        assert np.array_equal(a[2], b[2]), (what, "values", np.abs(a[2] - b[2]).max())


def run_updates(h, cols, m, nupd, rng, refactor=None, check_every=1, stop_on_max=False, tol_xtbl=1e-3, pair_row=None, twin=None):
    """Returns a log dict; `cols` is modified in place to the current basis.  Replacements whose pivot
    |xtbl| = |(B^-1 a)_j| is below tol_xtbl are not applied (they would make the basis ill-conditioned and the
    residual checks meaningless).  pair_row[j] = row paired with column j in the INITIAL factorization.
    twin: a second object (the CPU restatement of the same intended algorithm) driven in lockstep; every status,
    pattern and value must be identical to h's."""
    log = dict(done=0, skipped=0, singular=0, max_residual=0.0, max_vs_fresh=0.0, max_pivot_error=0.0, hit_maximum_updates=False,
               max_sparse_diff=0.0)
    B = matrix_of(cols, m)
    for step in range(nupd):
        j = int(rng.integers(0, m))
        ai, ax = new_column(rng, cols, m, j, pair_row)
        st, il, row = _sfu(h, [j], None, "T")
        if twin is not None:
            _same((st, il, row), _sfu(twin, [j], None, "T"), ("solve_for_update T", step))
        if st == K.ERROR_MAXIMUM_UPDATES:
            log["hit_maximum_updates"] = True
            if stop_on_max:
                break
            return log
        assert st == K.OK, st
        # row = B^-T e_j: check it (it is the "sparse re-solve" of the transposed system)
        ej = np.zeros(m)
        ej[j] = 1.0
        log["max_residual"] = max(log["max_residual"], backward_error(B.T, row, ej))
        assert np.array_equal(np.sort(il), np.flatnonzero(row)), "pattern of the transposed solution"
        st, il2, lhs = _sfu(h, ai, ax, "N")
        if twin is not None:
            _same((st, il2, lhs), _sfu(twin, ai, ax, "N"), ("solve_for_update N", step))
        assert st == K.OK, st
        a = np.zeros(m)
        a[ai] = ax
        log["max_residual"] = max(log["max_residual"], backward_error(B, lhs, a))
        assert np.array_equal(np.sort(il2), np.flatnonzero(lhs)), "pattern of the forward solution"
        xtbl = lhs[j]
        if abs(xtbl) < tol_xtbl:
            log["skipped"] += 1
            continue
        st = h.update(xtbl)
        if twin is not None:
            assert twin.update(xtbl) == st, ("update status", step)
            for key in (K.STAT_NFORREST, K.STAT_NUPDATE, K.STAT_R_NZ, K.STAT_PIVOT_ERROR, K.STAT_NSYMPERM_TOTAL, K.STAT_DEV_NUNSYMPERM_TOTAL,
                        K.STAT_MIN_PIVOT, K.STAT_MAX_PIVOT, K.STAT_MAX_ETA, K.STAT_U_NZ):
                assert h.stat(key) == twin.stat(key), ("stat", key, step, h.stat(key), twin.stat(key))
        if st == K.ERROR_SINGULAR_UPDATE:
            log["singular"] += 1
            continue
        assert st == K.OK, st
        log["done"] += 1
        log["max_pivot_error"] = max(log["max_pivot_error"], h.stat(K.STAT_PIVOT_ERROR))
        cols[j] = (ai, ax)
        B = matrix_of(cols, m)
        if log["done"] % check_every == 0:
            b = rng.standard_normal(m)
            x = h.solve_dense(b, "N")
            xt = h.solve_dense(b, "T")
            if twin is not None:
                assert np.array_equal(x, twin.solve_dense(b, "N")) and np.array_equal(xt, twin.solve_dense(b, "T")), ("solve_dense", step)
            log["max_residual"] = max(log["max_residual"], backward_error(B, x, b), backward_error(B.T, xt, b))
            # sparse solves on the updated factorization
            nz = int(rng.integers(1, max(2, m // 10)))
            ir = rng.choice(m, nz, replace=False)
            xr = rng.standard_normal(nz)
            bs = np.zeros(m)
            bs[ir] = xr
            for trans, A in (("N", B), ("T", B.T)):
                out = _ss(h, ir, xr, trans)
                if twin is not None:
                    _same(out, _ss(twin, ir, xr, trans), ("solve_sparse", trans, step))
                sol = out[2]
                assert out[0] == K.OK
                log["max_residual"] = max(log["max_residual"], backward_error(A, sol, bs))
            if refactor is not None:
                f = refactor(cols)
                if f is not None:
                    # the fresh factorization's own backward error is the yardstick: the updated one may be worse
                    # by the usual growth of a Forrest-Tomlin sequence, not by orders of magnitude
                    xf = f.solve_dense(b, "N")
                    log["max_vs_fresh"] = max(log["max_vs_fresh"], backward_error(B, x, b) / max(backward_error(B, xf, b), 1e-17))
    return log
