"""maxvolume (src/maxvolume.rs:64-224): one pass of the "maximum volume" basis search over the columns of a
rectangular matrix A (ncol >= nrow), on top of BLU.factorize / solve_for_update / update.

For every column a_j that is not in the basis, lhs = B^-1 a_j is computed (solve_for_update, forward, with the
solution); if its largest entry |lhs[imax]| exceeds `volumetol`, column j replaces position imax of the basis
(solve_for_update transposed for basis position imax, update with xtbl = lhs[imax]): the volume |det B| grows by
that factor.  After every update the basis is refactorized if the eta file is full, the last update had a large
pivot error (> 1e-8) or the accumulated solve cost says so (LU::update_cost() > 1, maxvolume.rs:211-224).

Works with any object that has the BLU methods (the HIP-backed blu_amd.BLU; the tests also run it on the CPU
oracle to compare the two).  Returns (status, nupdate); `basis` and `isbasic` are updated in place."""
import numpy as np

from . import keys as K


def _solve_for_update(obj, irhs, xrhs, trans, want):
    out = obj.solve_for_update(irhs, xrhs, trans, want)
    if isinstance(out, tuple):  # the oracle's binding returns (status, ilhs, lhs)
        return out
    if out != K.OK or not want:
        return out, None, None
    return out, obj.ilhs[:obj.nzlhs], obj.lhs


def _factorize_basis(obj, a_p, a_i, a_x, basis):
    b = np.asarray(basis, dtype=np.int64)
    begin = np.asarray(a_p, dtype=np.uint64)[b]
    end = np.asarray(a_p, dtype=np.uint64)[b + 1]
    return obj.factorize(begin, end, a_i, a_x)  # maxvolume.rs:180-197


def maxvolume(obj, ncol, a_p, a_i, a_x, basis, isbasic, volumetol):
    nupdate = 0
    if volumetol < 1.0:
        return K.ERROR_INVALID_ARGUMENT, nupdate  # maxvolume.rs:86-93
    a_i = np.ascontiguousarray(a_i, dtype=np.uint64)
    a_x = np.ascontiguousarray(a_x, dtype=np.float64)
    m = len(basis)
    st = _factorize_basis(obj, a_p, a_i, a_x, basis)
    if st != K.OK:
        return st, nupdate  # WarningSingularMatrix = the algorithm failed (doc, maxvolume.rs:61-62)
    for j in range(ncol):
        if isbasic[j]:
            continue
        a, b = int(a_p[j]), int(a_p[j + 1])
        st, il, lhs = _solve_for_update(obj, a_i[a:b], a_x[a:b], "N", True)
        if st != K.OK:
            return st, nupdate
        xmax, xtbl, imax = 0.0, 0.0, 0
        for i in il:  # first largest entry in pattern order (strict >, maxvolume.rs:121-131)
            if abs(lhs[i]) > xmax:
                xtbl = float(lhs[i])
                xmax = abs(xtbl)
                imax = int(i)
        if xmax <= volumetol:
            continue
        isbasic[int(basis[imax])] = 0
        isbasic[j] = 1
        basis[imax] = j
        nupdate += 1
        st, _, _ = _solve_for_update(obj, [imax], None, "T", False)
        if st != K.OK:
            return st, nupdate
        st = obj.update(xtbl)
        if st != K.OK:
            return st, nupdate
        # refactorize_if_needed, maxvolume.rs:199-224
        if obj.stat(K.STAT_NFORREST) == m or obj.stat(K.STAT_PIVOT_ERROR) > 1e-8 or obj.stat(K.STAT_UPDATE_COST) > 1.0:
            st = _factorize_basis(obj, a_p, a_i, a_x, basis)
            if st != K.OK:
                return st, nupdate
    return K.OK, nupdate
