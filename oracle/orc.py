"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY -- see oracle/blu_oracle.h.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, REALLOCATE, WARNING_SINGULAR_MATRIX = 0, 1, 2
ERROR_INVALID_CALL, ERROR_ARGUMENT_MISSING, ERROR_INVALID_ARGUMENT = -2, -3, -4
STOPPED = 100
D5_TRAP = -98  # the reference's endless Reallocate loop (defect D5) was met: re-run with a larger initial capacity

_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


def build():
    """Compile liborc.so (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        # ORC_LIB: another build of the same sources (tools/oracle_sanitize.sh: AddressSanitizer + UBSan)
        path = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_blu_new.restype = C.c_void_p
        L.orc_blu_new.argtypes = [C.c_int64, C.c_int64]
        L.orc_blu_free.argtypes = [C.c_void_p]
        L.orc_blu_lu.restype = C.c_void_p
        L.orc_blu_lu.argtypes = [C.c_void_p]
        L.orc_blu_factorize.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, _f64p]
        L.orc_factorize.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, _f64p, C.c_int]
        L.orc_blu_get_factors.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_blu_solve_dense.argtypes = [C.c_void_p, _f64p, _f64p, C.c_char]
        L.orc_blu_solve_sparse.argtypes = [C.c_void_p, C.c_int64, _u64p, _f64p, C.c_char]
        L.orc_blu_solve_for_update.argtypes = [C.c_void_p, C.c_int64, _u64p, C.c_void_p, C.c_char, C.c_int]
        L.orc_blu_update.argtypes = [C.c_void_p, C.c_double]
        L.orc_blu_nzlhs.restype = C.c_int64
        L.orc_blu_nzlhs.argtypes = [C.c_void_p]
        L.orc_blu_get_lhs.argtypes = [C.c_void_p, C.c_void_p, _f64p]
        L.orc_get_stat.restype = C.c_double
        L.orc_get_stat.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_gen_lp_basis.restype = C.c_int64
        L.orc_gen_lp_basis.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint64, _u64p, _u64p, _f64p]
        L.orc_dbg_set_stop.argtypes = [C.c_void_p, C.c_int64]
        L.orc_dbg_set_fix_d3.argtypes = [C.c_void_p, C.c_int]
        L.orc_dbg_d3_hits.restype = C.c_int64
        L.orc_dbg_d3_hits.argtypes = [C.c_void_p]
        L.orc_dbg_active_state.argtypes = [C.c_void_p] + [C.c_void_p] * 12
        L.orc_dbg_active_nnz.restype = C.c_int64
        L.orc_dbg_active_nnz.argtypes = [C.c_void_p, C.c_int]
        L.orc_dbg_partial_lu.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_dbg_partial_nz.restype = C.c_int64
        L.orc_dbg_partial_nz.argtypes = [C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def gen_lp_basis(m, k, bw, tri_frac, seed, offscale=1.0):
    """SURVEY.md 8d generator (C implementation in oracle/orc_api.c)."""
    colptr = np.zeros(m + 1, dtype=np.uint64)
    rowidx = np.zeros(max(1, m * max(k, 1)), dtype=np.uint64)
    value = np.zeros(max(1, m * max(k, 1)), dtype=np.float64)
    nnz = lib().orc_gen_lp_basis(m, k, bw, float(tri_frac), float(offscale), seed, _p(colptr, _u64p), _p(rowidx, _u64p), _p(value, _f64p))
    return colptr, rowidx[:nnz].copy(), value[:nnz].copy()


class OracleBLU:
    """Mirror of `struct BLU` (src/blu.rs) on top of the C oracle."""

    def __init__(self, m, b_nz):
        self.m = int(m)
        self._h = lib().orc_blu_new(int(m), int(b_nz))
        if not self._h:
            raise MemoryError("orc_blu_new failed")
        self._lu = lib().orc_blu_lu(self._h)
        self._keep = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_blu_free(self._h)
            self._h = None

    def set_param(self, key, v):
        return lib().orc_set_param(self._lu, int(key), float(v))

    def stat(self, key):
        return lib().orc_get_stat(self._lu, int(key))

    @staticmethod
    def _prep(b_begin, b_end, b_i, b_x):
        bb = np.ascontiguousarray(b_begin, dtype=np.uint64)
        be = np.ascontiguousarray(b_end, dtype=np.uint64)
        bi = np.ascontiguousarray(b_i, dtype=np.uint64)
        bx = np.ascontiguousarray(b_x, dtype=np.float64)
        return bb, be, bi, bx

    def factorize(self, b_begin, b_end, b_i, b_x):
        """BLU::factorize (blu.rs:95): realloc loop included.  Returns D5_TRAP where the reference would grow W
        for ever (see oracle/orc_api.c); use factorize_roomy() to have the capacity raised until it does not."""
        self._keep = self._prep(b_begin, b_end, b_i, b_x)
        bb, be, bi, bx = self._keep
        return lib().orc_blu_factorize(self._h, _p(bb, _u64p), _p(be, _u64p), _p(bi, _u64p), _p(bx, _f64p))

    @classmethod
    def factorize_roomy(cls, m, cap, b_begin, b_end, b_i, b_x, setup=None, max_cap=1 << 28):
        """A fresh OracleBLU(m, cap) factorized; on D5_TRAP the capacity is multiplied by 8 and the run repeated
        (results do not depend on the storage layout).  setup(o) applies parameters / hooks.
        Returns (oracle, status)."""
        while True:
            o = cls(m, cap)
            if setup:
                setup(o)
            st = o.factorize(b_begin, b_end, b_i, b_x)
            if st != D5_TRAP or cap >= max_cap:
                return o, st
            cap = min(max_cap, cap * 8)

    def factorize_raw(self, b_begin, b_end, b_i, b_x, c0ntinue=False):
        """factorize() (factorize.rs:34) without the realloc loop."""
        if not c0ntinue or self._keep is None:
            self._keep = self._prep(b_begin, b_end, b_i, b_x)
        bb, be, bi, bx = self._keep
        return lib().orc_factorize(self._lu, _p(bb, _u64p), _p(be, _u64p), _p(bi, _u64p), _p(bx, _f64p), int(bool(c0ntinue)))

    def get_factors(self):
        from blu_amd import keys as K  # numbering only
        m = self.m
        l_nz = int(self.stat(K.STAT_L_NZ))
        u_nz = int(self.stat(K.STAT_U_NZ))
        out = dict(
            rowperm=np.zeros(m, np.int64), colperm=np.zeros(m, np.int64),
            l_colptr=np.zeros(m + 1, np.int64), l_rowidx=np.zeros(m + l_nz, np.int64), l_value=np.zeros(m + l_nz),
            u_colptr=np.zeros(m + 1, np.int64), u_rowidx=np.zeros(m + u_nz, np.int64), u_value=np.zeros(m + u_nz),
        )
        st = lib().orc_blu_get_factors(self._h, *[out[k].ctypes.data for k in
                                                   ("rowperm", "colperm", "l_colptr", "l_rowidx", "l_value",
                                                    "u_colptr", "u_rowidx", "u_value")])
        if st != OK:
            raise RuntimeError("get_factors status %d" % st)
        return out

    def solve_dense(self, rhs, trans="N"):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        lhs = np.zeros(self.m)
        st = lib().orc_blu_solve_dense(self._h, _p(rhs, _f64p), _p(lhs, _f64p), trans.encode()[0:1])
        if st != OK:
            raise RuntimeError("solve_dense status %d" % st)
        return lhs

    def solve_sparse(self, irhs, xrhs, trans="N"):
        """BLU::solve_sparse (blu.rs:207).  Returns (status, ilhs[0..nzlhs) in the reference's order, lhs dense)."""
        ir = np.ascontiguousarray(irhs, dtype=np.uint64)
        xr = np.ascontiguousarray(xrhs, dtype=np.float64)
        st = lib().orc_blu_solve_sparse(self._h, len(ir), _p(ir, _u64p), _p(xr, _f64p), trans.encode()[0:1])
        if st != OK:
            return st, None, None
        nz = int(lib().orc_blu_nzlhs(self._h))
        il = np.zeros(max(1, nz), np.int64)
        lhs = np.zeros(self.m)
        lib().orc_blu_get_lhs(self._h, il.ctypes.data, _p(lhs, _f64p))
        return st, il[:nz], lhs

    def solve_for_update(self, irhs, xrhs=None, trans="N", want_solution=True):
        """BLU::solve_for_update (blu.rs:257), INTENDED algorithm (oracle/orc_update.c; not reference-pinned).
        Returns (status, ilhs, lhs) like solve_sparse; ilhs/lhs are None when no solution was wanted."""
        ir = np.ascontiguousarray(irhs, dtype=np.uint64)
        xr = None if xrhs is None else np.ascontiguousarray(xrhs, dtype=np.float64)
        st = lib().orc_blu_solve_for_update(self._h, len(ir), _p(ir, _u64p), None if xr is None else xr.ctypes.data,
                                            trans.encode()[0:1], int(bool(want_solution)))
        if st != OK or not want_solution:
            return st, None, None
        nz = int(lib().orc_blu_nzlhs(self._h))
        il = np.zeros(max(1, nz), np.int64)
        lhs = np.zeros(self.m)
        lib().orc_blu_get_lhs(self._h, il.ctypes.data, _p(lhs, _f64p))
        return st, il[:nz], lhs

    def update(self, xtbl):
        """BLU::update (blu.rs:319), INTENDED algorithm (oracle/orc_update.c)."""
        return lib().orc_blu_update(self._h, float(xtbl))

    # ---- debug hooks (not in the reference) --------------------------------
    def set_stop(self, npivots):
        lib().orc_dbg_set_stop(self._lu, int(npivots))

    def set_fix_d3(self, on=True):
        """Use BASICLU's 64-bit cancellation mask instead of the reference's i32 one (D3)."""
        lib().orc_dbg_set_fix_d3(self._lu, int(bool(on)))

    def d3_hits(self):
        return int(lib().orc_dbg_d3_hits(self._lu))

    def active_state(self):
        """Layout-independent dump of the active submatrix between two pivots."""
        m = self.m
        ncol = lib().orc_dbg_active_nnz(self._lu, 0)
        nrow = lib().orc_dbg_active_nnz(self._lu, 1)
        s = dict(
            colptr=np.zeros(m + 1, np.int64), colidx=np.zeros(max(1, ncol), np.int64), colval=np.zeros(max(1, ncol)),
            rowptr=np.zeros(m + 1, np.int64), rowidx=np.zeros(max(1, nrow), np.int64),
            colmax=np.zeros(m), pinv=np.zeros(m, np.int64), qinv=np.zeros(m, np.int64),
            col_flink=np.zeros(2 * m + 2, np.int64), col_blink=np.zeros(2 * m + 2, np.int64),
            row_flink=np.zeros(2 * m + 2, np.int64), row_blink=np.zeros(2 * m + 2, np.int64),
        )
        lib().orc_dbg_active_state(self._lu, *[s[k].ctypes.data for k in
                                               ("colptr", "colidx", "colval", "rowptr", "rowidx", "colmax", "pinv",
                                                "qinv", "col_flink", "col_blink", "row_flink", "row_blink")])
        s["colidx"] = s["colidx"][:ncol]
        s["colval"] = s["colval"][:ncol]
        s["rowidx"] = s["rowidx"][:nrow]
        return s

    def partial_lu(self):
        """L columns / U rows of the stages done so far (stage order)."""
        from blu_amd import keys as K
        rank = int(self.stat(K.STAT_RANK))
        nl = lib().orc_dbg_partial_nz(self._lu, 0)
        nu = lib().orc_dbg_partial_nz(self._lu, 1)
        s = dict(lptr=np.zeros(rank + 1, np.int64), lidx=np.zeros(max(1, nl), np.int64), lval=np.zeros(max(1, nl)),
                 uptr=np.zeros(rank + 1, np.int64), uidx=np.zeros(max(1, nu), np.int64), uval=np.zeros(max(1, nu)))
        lib().orc_dbg_partial_lu(self._lu, *[s[k].ctypes.data for k in ("lptr", "lidx", "lval", "uptr", "uidx", "uval")])
        s["lidx"], s["lval"] = s["lidx"][:nl], s["lval"][:nl]
        s["uidx"], s["uval"] = s["uidx"][:nu], s["uval"][:nu]
        return s
