"""maxvolume (src/maxvolume.rs:64-224) -- the loop over solve_for_update / update that SURVEY 8f ranks N4.
CPU: the Python restatement (blu_amd/maxvolume.py) driven by the oracle finds a locally maximum-volume basis.
GPU (-m gpu): the same function driven by the HIP BLU takes the very same decisions (bit-identical solves)."""
import numpy as np
import pytest
import scipy.sparse as sp

from blu_amd import keys as K
from blu_amd.maxvolume import maxvolume


def _problem(nrow, ncol, seed):
    rng = np.random.default_rng(seed)
    cols = [dict() for _ in range(ncol)]
    for j in range(nrow):  # an identity-like start so that the initial basis is nonsingular
        cols[j][j] = 1.0
    for j in range(ncol):
        for i in rng.choice(nrow, int(rng.integers(1, 5)), replace=False):
            cols[j][int(i)] = cols[j].get(int(i), 0.0) + float(rng.standard_normal()) * (3.0 if j >= nrow else 0.3)
    a_p, a_i, a_x = [0], [], []
    for c in cols:
        for i, x in c.items():
            a_i.append(i); a_x.append(x)
        a_p.append(len(a_i))
    return np.array(a_p, np.uint64), np.array(a_i, np.uint64), np.array(a_x)


def _logvol(a_p, a_i, a_x, nrow, basis):
    A = sp.csc_matrix((a_x, a_i.astype(np.int64), a_p.astype(np.int64)), shape=(nrow, len(a_p) - 1))
    return np.linalg.slogdet(A[:, np.asarray(basis)].toarray())[1]


def _run(make_obj, nrow, ncol, seed, tol):
    a_p, a_i, a_x = _problem(nrow, ncol, seed)
    basis = list(range(nrow))
    isbasic = [1] * nrow + [0] * (ncol - nrow)
    v0 = _logvol(a_p, a_i, a_x, nrow, basis)
    trace = []
    for sweep in range(30):
        obj = make_obj(nrow, len(a_i))
        st, nupd = maxvolume(obj, ncol, a_p, a_i, a_x, basis, isbasic, tol)
        assert st == K.OK, st
        trace.append((nupd, tuple(basis)))
        if nupd == 0:
            break
    assert trace[-1][0] == 0, "no locally maximal basis after 30 sweeps"
    assert sorted(j for j in range(ncol) if isbasic[j]) == sorted(basis)
    v1 = _logvol(a_p, a_i, a_x, nrow, basis)
    nchanges = sum(t[0] for t in trace)
    assert v1 >= v0 + nchanges * np.log(tol) - 1e-6  # every basis change multiplied the volume by more than tol
    # local maximality: no nonbasic column has an entry of B^-1 a_j above the tolerance
    A = sp.csc_matrix((a_x, a_i.astype(np.int64), a_p.astype(np.int64)), shape=(nrow, ncol)).toarray()
    X = np.linalg.solve(A[:, basis], A[:, [j for j in range(ncol) if not isbasic[j]]])
    assert np.abs(X).max() <= tol * (1 + 1e-9)
    return trace


@pytest.mark.parametrize("nrow,ncol,seed,tol", [(30, 90, 1, 2.0), (60, 150, 2, 1.5), (12, 40, 3, 1.0)])
def test_maxvolume_on_the_oracle(oracle, nrow, ncol, seed, tol):
    assert maxvolume(None, 1, [0, 0], [], [], [], [], 0.5)[0] == K.ERROR_INVALID_ARGUMENT
    trace = _run(lambda m, nz: oracle.OracleBLU(m, 64 * nz + 1024), nrow, ncol, seed, tol)
    assert sum(t[0] for t in trace) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nrow,ncol,seed,tol", [(30, 90, 1, 2.0), (60, 150, 2, 1.5)])
def test_maxvolume_gpu_takes_the_same_decisions(oracle, nrow, ncol, seed, tol):
    import blu_amd
    if blu_amd.lib().blu_hip_device_count() < 1:
        pytest.fail("no HIP device visible")
    tg = _run(lambda m, nz: blu_amd.BLU(m, nz), nrow, ncol, seed, tol)
    to = _run(lambda m, nz: oracle.OracleBLU(m, 64 * nz + 1024), nrow, ncol, seed, tol)
    assert tg == to
