"""Step-wise GPU-vs-oracle comparison of the pivot loop (debug tool, run on the GPU box).

  python tools/gpu_stepcheck.py <name|m,k,bw,tri,offs,seed> [--step N] [--block T] [--search-rows] [--nzbias -1]

Runs the HIP path and the CPU oracle in lock step, comparing the complete active submatrix (ordered
line contents, column maxima, count lists, pivots, partial L/U) every N pivots; on the first
mismatch it re-runs both to the last good checkpoint and single-steps to the offending pivot.
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blu_amd import BLU, keys as K  # noqa: E402
from blu_amd.matrices import simple_rs  # noqa: E402
from oracle import orc  # noqa: E402

STATE_KEYS = ["pinv", "qinv", "colptr", "colidx", "colval", "rowptr", "rowidx", "colmax", "col_flink", "col_blink"]
LU_KEYS = ["lptr", "lidx", "lval", "uptr", "uidx", "uval"]


def diff_states(g, o, search_rows):
    bad = []
    keys = STATE_KEYS + (["row_flink", "row_blink"] if search_rows else [])
    for k in keys:
        a, b = np.asarray(g[k]), np.asarray(o[k])
        if a.shape != b.shape or not np.array_equal(a, b):
            n = min(len(a), len(b))
            w = np.nonzero(a[:n] != b[:n])[0]
            bad.append((k, a.shape, b.shape, w[:5].tolist(), a[w[:5]].tolist() if len(w) else [], b[w[:5]].tolist() if len(w) else []))
    return bad


def make(args, cp, ri, v):
    m = len(cp) - 1
    g = BLU(m, len(ri))
    o = orc.OracleBLU(m, 64 * len(ri) + 1024)
    o.set_fix_d3(True)
    for h in (g, o):
        h.set_param(K.PARAM_SEARCH_ROWS, 1 if args.search_rows else 0)
        h.set_param(K.PARAM_NZBIAS, args.nzbias)
    g.dbg_set_block(args.block)
    if args.no_fast:
        g.dbg_set_no_fast(True)
    return g, o


def run_to(args, cp, ri, v, stop):
    g, o = make(args, cp, ri, v)
    g.dbg_set_stop(stop)
    o.set_stop(stop)
    sg = g.factorize(cp[:-1], cp[1:], ri, v)
    so = o.factorize(cp[:-1], cp[1:], ri, v)
    return g, o, sg, so


def compare(g, o, args, tag):
    bad = diff_states(g.dbg_active_state(), o.active_state(), args.search_rows)
    gl, ol = g.dbg_partial_lu(), o.partial_lu()
    for k in LU_KEYS:
        if not np.array_equal(gl[k], ol[k]):
            n = min(len(gl[k]), len(ol[k]))
            w = np.nonzero(np.asarray(gl[k][:n]) != np.asarray(ol[k][:n]))[0]
            bad.append((k, gl[k].shape, ol[k].shape, w[:5].tolist(), np.asarray(gl[k])[w[:5]].tolist(), np.asarray(ol[k])[w[:5]].tolist()))
    for c in ("RANK", "RANKDEF", "NSEARCH_PIVOT", "FACTOR_FLOPS", "BUMP_NZ", "MATRIX_NZ"):
        a, b = g.stat(getattr(K, "STAT_" + c)), o.stat(getattr(K, "STAT_" + c))
        if a != b:
            bad.append((c, a, b))
    if bad:
        print("MISMATCH at", tag)
        for b in bad:
            print("   ", b)
    return not bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("matrix")
    ap.add_argument("--step", type=int, default=64)
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--search-rows", action="store_true")
    ap.add_argument("--nzbias", type=int, default=1)
    ap.add_argument("--no-fast", action="store_true", help="general pivot paths only")
    args = ap.parse_args()
    if args.matrix == "simple":
        cp, ri, v, _, _ = simple_rs()
    else:
        m, k, bw, tri, offs, seed = args.matrix.split(",")
        cp, ri, v = orc.gen_lp_basis(int(m), int(k), int(bw), float(tri), int(seed), float(offs))
    m = len(cp) - 1
    print("matrix m=%d nnz=%d block=%d" % (m, len(ri), args.block), flush=True)

    g, o, sg, so = run_to(args, cp, ri, v, 0)
    print("after prep+setup: gpu status", sg, "oracle status", so, "rank0", g.stat(K.STAT_RANK), o.stat(K.STAT_RANK), flush=True)
    if sg not in (K.OK, K.WARNING_SINGULAR_MATRIX, 100):
        print("gpu error:", g.last_error(), "dev status", g.stat(58), "line", g.stat(57))
        return 1
    good = 0
    if sg == 100 and so == 100:
        if not compare(g, o, args, "setup (0 bump pivots)"):
            return 1
        stop = int(g.stat(K.STAT_RANK))
        while True:
            stop += args.step
            sg = g.dbg_continue(stop)
            o.set_stop(stop)
            so = o.factorize_raw(None, None, None, None, c0ntinue=True)
            if sg not in (K.OK, K.WARNING_SINGULAR_MATRIX, 100):
                print("gpu error at stop", stop, ":", g.last_error(), "dev status", g.stat(58), "line", g.stat(57), flush=True)
                break
            if sg != 100 or so != 100:
                break
            if not compare(g, o, args, "checkpoint %d" % stop):
                # single-step from the last good checkpoint
                g2, o2, a, b = run_to(args, cp, ri, v, good if good else 0)
                s = int(g2.stat(K.STAT_RANK)) + int(g2.stat(K.STAT_RANKDEF)) if good == 0 else good
                while s < stop:
                    s += 1
                    a = g2.dbg_continue(s)
                    o2.set_stop(s)
                    b = o2.factorize_raw(None, None, None, None, c0ntinue=True)
                    if a != 100 or b != 100 or not compare(g2, o2, args, "pivot #%d (0-based %d)" % (s, s - 1)):
                        kinds = [g2.stat(51 + i) for i in range(6)]
                        print("gpu pivot kinds so far [srow, scol, dbl, small, any, empty]:", kinds, "status", a, b)
                        return 1
                return 1
            good = stop
            print("ok through", stop, flush=True)
    print("final: gpu status", sg, "oracle status", so, flush=True)
    if sg in (K.OK, K.WARNING_SINGULAR_MATRIX) and sg == so:
        fg, fo = g.get_factors(), o.get_factors()
        okf = True
        for k in ("rowperm", "colperm", "l_colptr", "l_rowidx", "u_colptr", "u_rowidx", "l_value", "u_value"):
            if not np.array_equal(fg[k], fo[k]):
                okf = False
                d = np.abs(np.asarray(fg[k], float) - np.asarray(fo[k], float)).max() if fg[k].shape == fo[k].shape else None
                print("   factor mismatch", k, fg[k].shape, fo[k].shape, d)
        for c in ("RANK", "L_NZ", "U_NZ", "NSEARCH_PIVOT", "FACTOR_FLOPS", "BUMP_NZ", "MATRIX_NZ", "MIN_PIVOT", "MAX_PIVOT"):
            a, b = g.stat(getattr(K, "STAT_" + c)), o.stat(getattr(K, "STAT_" + c))
            if a != b:
                okf = False
                print("   stat mismatch", c, a, b)
        print("FACTORS", "IDENTICAL" if okf else "DIFFER", "d3_hits gpu", g.stat(50), "oracle", o.d3_hits(),
              "kinds", [g.stat(51 + i) for i in range(6)], "t_pivot", g.stat(K.STAT_DEV_TIME_PIVOT_LOOP), "relaunch", g.stat(K.STAT_DEV_RELAUNCHES))
        if okf:
            rng = np.random.default_rng(0)
            xs = rng.standard_normal(m)
            import scipy.sparse as sp
            B = sp.csc_matrix((v, ri.astype(np.int64), cp.astype(np.int64)), shape=(m, m))
            for tr in ("N", "T"):
                b = (B @ xs) if tr == "N" else (B.T @ xs)
                x = g.solve_dense(b, tr)
                xo = o.solve_dense(b, tr)
                print("   solve_dense", tr, "max|x-xo|", np.abs(x - xo).max(), "max|x-xs|", np.abs(x - xs).max())
        return 0 if okf else 1
    return 1


if __name__ == "__main__":
    sys.exit(main())
