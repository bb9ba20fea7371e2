"""Python host-side mirror of the reference's object API (`struct BLU`, src/blu.rs) on top of the
C ABI of libblu_hip.so (include/blu_hip.h).

The reference's host language is Rust; there is no Rust toolchain in this image, so the layer a
Rust caller would use (`extern "C"` block) is exercised from Python through ctypes
with the same method names, argument meaning and error behaviour:

    BLU(m, b_nz)                                   BLU::new                 blu.rs:61
    .factorize(b_begin, b_end, b_i, b_x)           BLU::factorize           blu.rs:95
    .get_factors()                                 BLU::get_factors         blu.rs:139
    .solve_dense(rhs, trans)                       BLU::solve_dense         blu.rs:182
    .set_param / .stat                             pub fields / getters     lu.rs:11-66, 398-684

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is visible, this
module raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import keys as K

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BLU_HIP_LIB") or os.path.join(_HERE, "libblu_hip.so")  # BLU_HIP_LIB: diagnostic builds
_LIB = None

_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)

STOPPED = 100  # debug stepping only

EXPORTS = [
    "blu_hip_new", "blu_hip_free", "blu_hip_set_param", "blu_hip_get_param", "blu_hip_get_stat",
    "blu_hip_factorize", "blu_hip_factorize_device", "blu_hip_get_factors", "blu_hip_solve_dense",
    "blu_hip_factorize_batch", "blu_hip_version", "blu_hip_device_count", "blu_hip_last_error",
    "blu_hip_solve_sparse", "blu_hip_solve_for_update", "blu_hip_update", "blu_hip_set_skip_stats", "blu_hip_gen_lp_basis",
]


class BluError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("blu_hip status %d %s" % (status, msg))
        self.status = status


SELFCHECK_LIB_PATH = os.path.join(_HERE, "libblu_hip_ewcheck.so")


def build_library(verbose=False, selfcheck=False):
    """Compile libblu_hip.so for gfx950 with hipcc (works without a GPU).

    selfcheck=True builds libblu_hip_ewcheck.so instead: the same library with -DBLU_EWCHECK, in which the pivot loop
    compares every early / speculative search of the next pivot with the ordinary search (candidates, count, key,
    staged entries) and raises ST_ERROR on the first difference.  Select it with BLU_HIP_LIB=<path> (diagnostic)."""
    target = "../libblu_hip_ewcheck.so" if selfcheck else "../libblu_hip.so"
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), target]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return SELFCHECK_LIB_PATH if selfcheck else _LIB_PATH


def lib():
    """Load libblu_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError("libblu_hip.so not built: run blu_amd.build_library() / __graft_entry__.build()")
        L = C.CDLL(_LIB_PATH)
        L.blu_hip_new.restype = C.c_void_p
        L.blu_hip_new.argtypes = [C.c_int64, C.c_int64, C.c_int]
        L.blu_hip_free.argtypes = [C.c_void_p]
        L.blu_hip_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.blu_hip_get_param.restype = C.c_double
        L.blu_hip_get_param.argtypes = [C.c_void_p, C.c_int]
        L.blu_hip_get_stat.restype = C.c_double
        L.blu_hip_get_stat.argtypes = [C.c_void_p, C.c_int]
        L.blu_hip_factorize.argtypes = [C.c_void_p, _u64p, _u64p, _u64p, _f64p, C.c_uint64]
        L.blu_hip_factorize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.blu_hip_get_factors.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.blu_hip_solve_dense.argtypes = [C.c_void_p, _f64p, _f64p, C.c_char]
        L.blu_hip_solve_sparse.argtypes = [C.c_void_p, C.c_int64, _u64p, _f64p, C.c_void_p, C.c_void_p, _f64p, C.c_char]
        L.blu_hip_solve_for_update.argtypes = [C.c_void_p, C.c_int64, _u64p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char]
        L.blu_hip_update.argtypes = [C.c_void_p, C.c_double]
        L.blu_hip_version.restype = C.c_char_p
        L.blu_hip_last_error.restype = C.c_char_p
        L.blu_hip_last_error.argtypes = [C.c_void_p]
        L.blu_hip_gen_lp_basis.restype = C.c_int64
        L.blu_hip_gen_lp_basis.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint64, _u64p, _u64p, _f64p]
        L.blu_hip_dbg_set_stop.argtypes = [C.c_void_p, C.c_int64]
        L.blu_hip_dbg_set_block.argtypes = [C.c_void_p, C.c_int]
        L.blu_hip_dbg_continue.argtypes = [C.c_void_p, C.c_int64]
        L.blu_hip_dbg_count.restype = C.c_int64
        L.blu_hip_dbg_count.argtypes = [C.c_void_p, C.c_int]
        L.blu_hip_dbg_active_state.argtypes = [C.c_void_p] + [C.c_void_p] * 12
        L.blu_hip_dbg_partial_lu.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def gen_lp_basis(m, k, bw, tri_frac, seed, offscale=1.0):
    """Synthetic LP basis (host utility inside libblu_hip.so). Returns CSC (colptr, rowidx, values)."""
    colptr = np.zeros(m + 1, dtype=np.uint64)
    n = max(1, m * max(k, 1))
    rowidx = np.zeros(n, dtype=np.uint64)
    value = np.zeros(n, dtype=np.float64)
    nnz = lib().blu_hip_gen_lp_basis(m, k, bw, float(tri_frac), float(offscale), seed,
                                     _p(colptr, _u64p), _p(rowidx, _u64p), _p(value, _f64p))
    return colptr, rowidx[:nnz].copy(), value[:nnz].copy()


def factorize_batch(handles, mats=None, device_ptrs=None, block=None):
    """Factorize len(handles) independent bases concurrently on one GPU (one workgroup per basis).

    mats: list of (colptr, rowidx, values) host CSC triples, or
    device_ptrs: list of (p_begin, p_end, p_i, p_x, nnz_len) raw device pointers (inputs already in HBM).
    Returns the list of per-handle statuses (reference Status numbering)."""
    n = len(handles)
    L = lib()
    L.blu_hip_factorize_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p]
    L.blu_hip_dbg_set_batch_block.argtypes = [C.c_void_p, C.c_int]
    if block:
        L.blu_hip_dbg_set_batch_block(handles[0]._h, int(block))
    hs = (C.c_void_p * n)(*[h._h for h in handles])
    pb, pe, pi, px = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    ln = (C.c_uint64 * n)()
    keep = []
    if device_ptrs is not None:
        for k, (b, e, i, x, nn) in enumerate(device_ptrs):
            pb[k], pe[k], pi[k], px[k], ln[k] = b, e, i, x, int(nn)
        on_dev = 1
    else:
        for k, (cp, ri, v) in enumerate(mats):
            cp = np.ascontiguousarray(cp, dtype=np.uint64)
            ri = np.ascontiguousarray(ri, dtype=np.uint64)
            v = np.ascontiguousarray(v, dtype=np.float64)
            keep.append((cp, ri, v))
            pb[k], pe[k], pi[k], px[k], ln[k] = cp.ctypes.data, cp.ctypes.data + 8, ri.ctypes.data, v.ctypes.data, len(ri)
        on_dev = 0
    UNTOUCHED = -12345
    st = (C.c_int * n)(*([UNTOUCHED] * n))
    rc = L.blu_hip_factorize_batch(hs, n, pb, pe, pi, px, ln, on_dev, st)
    out = [int(s) for s in st]
    if rc < 0 and any(s == UNTOUCHED for s in out):  # refused before any handle was worked on
        raise BluError(rc, handles[0].last_error())
    if rc in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY) and all(s >= 0 for s in out):
        raise BluError(rc, handles[0].last_error())
    for h, s in zip(handles, out):
        if s in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(s, h.last_error())
    return out


class BLU:
    """`struct BLU` (src/blu.rs:9-20) backed by the HIP implementation."""

    def __init__(self, m, b_nz, device=0):
        self.m = int(m)
        self.lhs = self.ilhs = None  # BLU.lhs / BLU.ilhs / BLU.nzlhs (blu.rs:12-17): solve_sparse results
        self.nzlhs = 0
        self._h = lib().blu_hip_new(int(m), int(b_nz), int(device))
        if not self._h:
            raise BluError(K.ERROR_DEVICE, "blu_hip_new failed (no gfx950 device, bad argument or out of memory)")

    def close(self):
        if getattr(self, "_h", None):
            lib().blu_hip_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # --- parameters / statistics (lu.rs public fields and getters) -------------------------------
    def set_param(self, key, value):
        st = lib().blu_hip_set_param(self._h, int(key), float(value))
        if st != K.OK:
            raise BluError(st)

    def get_param(self, key):
        return lib().blu_hip_get_param(self._h, int(key))

    def stat(self, key):
        return lib().blu_hip_get_stat(self._h, int(key))

    def last_error(self):
        return lib().blu_hip_last_error(self._h).decode()

    # --- BLU::factorize (blu.rs:95) ----------------------------------------------------------------
    def factorize(self, b_begin, b_end, b_i, b_x):
        """Returns the status (OK / WARNING_SINGULAR_MATRIX / ERROR_INVALID_ARGUMENT ...), as the reference's
        Result<(), Status> does; device failures raise."""
        bb = np.ascontiguousarray(b_begin, dtype=np.uint64)
        be = np.ascontiguousarray(b_end, dtype=np.uint64)
        bi = np.ascontiguousarray(b_i, dtype=np.uint64)
        bx = np.ascontiguousarray(b_x, dtype=np.float64)
        if len(bb) != self.m or len(be) != self.m or len(bi) != len(bx):
            return K.ERROR_INVALID_ARGUMENT
        st = lib().blu_hip_factorize(self._h, _p(bb, _u64p), _p(be, _u64p), _p(bi, _u64p), _p(bx, _f64p), len(bi))
        if st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    def factorize_device(self, d_begin, d_end, d_i, d_x, nnz_len):
        """Same with B already in device memory (raw device pointers as ints)."""
        st = lib().blu_hip_factorize_device(self._h, d_begin, d_end, d_i, d_x, int(nnz_len))
        if st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    # --- BLU::get_factors (blu.rs:139) -------------------------------------------------------------
    def get_factors(self):
        m = self.m
        l_nz = int(self.stat(K.STAT_L_NZ))
        u_nz = int(self.stat(K.STAT_U_NZ))
        out = dict(
            rowperm=np.zeros(m, np.int64), colperm=np.zeros(m, np.int64),
            l_colptr=np.zeros(m + 1, np.int64), l_rowidx=np.zeros(m + l_nz, np.int64), l_value=np.zeros(m + l_nz),
            u_colptr=np.zeros(m + 1, np.int64), u_rowidx=np.zeros(m + u_nz, np.int64), u_value=np.zeros(m + u_nz),
        )
        st = lib().blu_hip_get_factors(self._h, *[out[k].ctypes.data for k in
                                                  ("rowperm", "colperm", "l_colptr", "l_rowidx", "l_value",
                                                   "u_colptr", "u_rowidx", "u_value")])
        if st != K.OK:
            raise BluError(st, self.last_error())
        return out

    # --- BLU::solve_dense (blu.rs:182) -------------------------------------------------------------
    def solve_dense(self, rhs, trans="N"):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        lhs = np.zeros(self.m)
        st = lib().blu_hip_solve_dense(self._h, _p(rhs, _f64p), _p(lhs, _f64p), trans.encode()[0:1])
        if st != K.OK:
            raise BluError(st, self.last_error())
        return lhs

    # --- BLU::solve_sparse (blu.rs:207) ------------------------------------------------------------
    def solve_sparse(self, irhs, xrhs, trans="N"):
        """Sparse right-hand side irhs/xrhs -> solution.  As in the reference the result stays in the
        object: self.lhs (dense, m), self.ilhs[0..self.nzlhs) the pattern in the reference's order.
        The previous solution is cleared first (lu_clear_lhs, blu.rs:380-395).  Returns the status."""
        m = self.m
        if self.lhs is None:
            self.lhs = np.zeros(m)
            self.ilhs = np.zeros(max(1, m), np.int64)
            self.nzlhs = 0
        if self.nzlhs:
            if self.nzlhs <= int(self.get_param(K.PARAM_SPARSE_THRES) * m):
                self.lhs[self.ilhs[:self.nzlhs]] = 0.0
            else:
                self.lhs[:] = 0.0
            self.nzlhs = 0
        ir = np.ascontiguousarray(irhs, dtype=np.uint64)
        xr = np.ascontiguousarray(xrhs, dtype=np.float64)
        nz = C.c_int64(0)
        st = lib().blu_hip_solve_sparse(self._h, len(ir), _p(ir, _u64p), _p(xr, _f64p), C.byref(nz),
                                        self.ilhs.ctypes.data, _p(self.lhs, _f64p), trans.encode()[0:1])
        if st == K.OK:
            self.nzlhs = int(nz.value)
        elif st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    def _clear_lhs(self):
        m = self.m
        if self.lhs is None:
            self.lhs = np.zeros(m)
            self.ilhs = np.zeros(max(1, m), np.int64)
            self.nzlhs = 0
        if self.nzlhs:  # lu_clear_lhs, blu.rs:380-395
            if self.nzlhs <= int(self.get_param(K.PARAM_SPARSE_THRES) * m):
                self.lhs[self.ilhs[:self.nzlhs]] = 0.0
            else:
                self.lhs[:] = 0.0
            self.nzlhs = 0

    # --- BLU::solve_for_update (blu.rs:257) --------------------------------------------------------
    def solve_for_update(self, irhs, xrhs=None, trans="N", want_solution=True):
        """Prepare an update.  trans 'T': irhs[0] is the column to be replaced; otherwise irhs/xrhs is the column
        to be inserted.  With want_solution the solution of the system is left in self.lhs / self.ilhs[0..nzlhs)
        as by solve_sparse.  Returns the status."""
        self._clear_lhs()
        ir = np.ascontiguousarray(irhs, dtype=np.uint64)
        xr = None if xrhs is None else np.ascontiguousarray(xrhs, dtype=np.float64)
        nz = C.c_int64(0)
        if want_solution:
            st = lib().blu_hip_solve_for_update(self._h, len(ir), _p(ir, _u64p), None if xr is None else xr.ctypes.data,
                                                C.addressof(nz), self.ilhs.ctypes.data, self.lhs.ctypes.data, trans.encode()[0:1])
        else:
            st = lib().blu_hip_solve_for_update(self._h, len(ir), _p(ir, _u64p), None if xr is None else xr.ctypes.data,
                                                None, None, None, trans.encode()[0:1])
        if st == K.OK and want_solution:
            self.nzlhs = int(nz.value)
        elif st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    # --- BLU::update (blu.rs:319) ------------------------------------------------------------------
    def update(self, xtbl):
        st = lib().blu_hip_update(self._h, float(xtbl))
        if st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    # --- test hooks (step-wise comparison with the oracle) -------------------------------------------
    def dbg_set_stop(self, n):
        lib().blu_hip_dbg_set_stop(self._h, int(n))

    def dbg_set_block(self, threads):
        st = lib().blu_hip_dbg_set_block(self._h, int(threads))
        if st != K.OK:
            raise BluError(st)

    def set_skip_stats(self, on=True):
        """Skip the statistics tail of factorize() (condest, residual_test); default: computed, as the reference."""
        lib().blu_hip_set_skip_stats.argtypes = [C.c_void_p, C.c_int]
        lib().blu_hip_set_skip_stats(self._h, int(bool(on)))

    def dbg_set_upd_extra(self, n):
        """Arena slack of the update path (entries); small values force the host-side growth loop."""
        lib().blu_hip_dbg_set_upd_extra.argtypes = [C.c_void_p, C.c_int64]
        lib().blu_hip_dbg_set_upd_extra(self._h, int(n))

    def dbg_set_grid_blocks(self, n):
        """Workgroups of the chip-wide O(nnz) phases of a single factorize (1 = one workgroup, as inside a batch)."""
        lib().blu_hip_dbg_set_grid_blocks.argtypes = [C.c_void_p, C.c_int]
        st = lib().blu_hip_dbg_set_grid_blocks(self._h, int(n))
        if st != K.OK:
            raise BluError(st)

    def dbg_set_pivot_kernel(self, which):
        """0 = default (one basis: k_pivot_loop; batch: k_pivot_loop_wave2 while every workgroup is resident, else
        k_pivot_loop_wave), 1 = one wave per matrix, 2 = multi-wave workgroups, 3 = two waves per matrix"""
        lib().blu_hip_dbg_set_pivot_kernel.argtypes = [C.c_void_p, C.c_int]
        st = lib().blu_hip_dbg_set_pivot_kernel(self._h, int(which))
        if st != K.OK:
            raise BluError(st, "dbg_set_pivot_kernel")

    def dbg_set_no_fast(self, on=True):
        """Run the general pivot paths only (k_pivot_fast.hip off): A/B of the two implementations."""
        lib().blu_hip_dbg_set_no_fast.argtypes = [C.c_void_p, C.c_int]
        lib().blu_hip_dbg_set_no_fast(self._h, int(bool(on)))

    def dbg_continue(self, stop_at):
        st = lib().blu_hip_dbg_continue(self._h, int(stop_at))
        if st in (K.ERROR_DEVICE, K.ERROR_OUT_OF_MEMORY):
            raise BluError(st, self.last_error())
        return st

    def dbg_active_state(self):
        m = self.m
        ncol = lib().blu_hip_dbg_count(self._h, 0)
        nrow = lib().blu_hip_dbg_count(self._h, 1)
        s = dict(
            colptr=np.zeros(m + 1, np.int64), colidx=np.zeros(max(1, ncol), np.int64), colval=np.zeros(max(1, ncol)),
            rowptr=np.zeros(m + 1, np.int64), rowidx=np.zeros(max(1, nrow), np.int64),
            colmax=np.zeros(m), pinv=np.zeros(m, np.int64), qinv=np.zeros(m, np.int64),
            col_flink=np.zeros(2 * m + 2, np.int64), col_blink=np.zeros(2 * m + 2, np.int64),
            row_flink=np.zeros(2 * m + 2, np.int64), row_blink=np.zeros(2 * m + 2, np.int64),
        )
        st = lib().blu_hip_dbg_active_state(self._h, *[s[k].ctypes.data for k in
                                                       ("colptr", "colidx", "colval", "rowptr", "rowidx", "colmax",
                                                        "pinv", "qinv", "col_flink", "col_blink", "row_flink", "row_blink")])
        if st != K.OK:
            raise BluError(st, self.last_error())
        s["colidx"], s["colval"], s["rowidx"] = s["colidx"][:ncol], s["colval"][:ncol], s["rowidx"][:nrow]
        return s

    def dbg_partial_lu(self):
        nl = lib().blu_hip_dbg_count(self._h, 2)
        nu = lib().blu_hip_dbg_count(self._h, 3)
        rank = int(self.stat(K.STAT_RANK))
        s = dict(lptr=np.zeros(rank + 1, np.int64), lidx=np.zeros(max(1, nl), np.int64), lval=np.zeros(max(1, nl)),
                 uptr=np.zeros(rank + 1, np.int64), uidx=np.zeros(max(1, nu), np.int64), uval=np.zeros(max(1, nu)))
        st = lib().blu_hip_dbg_partial_lu(self._h, *[s[k].ctypes.data for k in ("lptr", "lidx", "lval", "uptr", "uidx", "uval")])
        if st != K.OK:
            raise BluError(st, self.last_error())
        s["lidx"], s["lval"], s["uidx"], s["uval"] = s["lidx"][:nl], s["lval"][:nl], s["uidx"][:nu], s["uval"][:nu]
        return s
