"""An INDEPENDENT second restatement of the reference's factorize path, for small matrices only.

Why it exists: the CPU oracle (oracle/*.c) is a line-by-line transliteration of the Rust sources -- same arrays,
same gap-separated files, same linked lists -- so a misreading made once could sit in the oracle and in the HIP
path alike.  This model was written from the Rust sources again (file:line cited at every rule), in a different
shape: no files, no links, no workspace arrays -- every active column is a Python list of [row, value] in the
reference's entry order, every row a list of column indices, every count list a Python list in list order.
It is slow and only handles what small examples need (pivot columns up to 30 off-diagonals: the i32 mask of
defect D3 is then the same as a wide one; no storage limits -- the layout never affects results).

`python tests/golden/derive_model.py` prints the derivation of examples/simple.rs step by step (committed as
tests/golden/simple_rs_derivation.txt) -- the pivot sequence, every list move, the L columns and U rows -- so the
anchors of tests/test_oracle.py can be audited by hand against the Rust source.  tests/test_model_vs_oracle.py
compares the oracle with this model on simple.rs, on hand-made cases and on a few hundred random small matrices.
"""
import math


class Model:
    def __init__(self, m, colptr, rowidx, values, droptol=1e-20, abstol=1e-14, reltol=0.1, nzbias=1, maxsearch=3, log=None):
        self.m = m
        self.B = [[(int(rowidx[p]), float(values[p])) for p in range(int(colptr[j]), int(colptr[j + 1]))] for j in range(m)]
        self.droptol, self.abstol, self.reltol, self.nzbias, self.maxsearch = droptol, abstol, reltol, nzbias, maxsearch
        self.log = log or (lambda *a: None)
        self.pinv = [-1] * m
        self.qinv = [-1] * m
        self.prow, self.pcol = [], []          # pivot sequence
        self.L, self.U = [], []                # stage order: L column (row, value) lists, U row (col, value) lists
        self.colpiv = [0.0] * m
        self.nsearch = 0
        self.flops = 0
        self.rankdef = 0
        self.kinds = [0] * 6

    # ------------------------------------------------------------------ singletons.rs:81-264 (+ D1)
    def singletons(self):
        m = self.m
        Bt = [[] for _ in range(m)]  # row-wise copy, ascending column inside a row (singletons.rs:186-198)
        for j in range(m):
            for (i, x) in self.B[j]:
                Bt[i].append((j, x))
        self.Bt = Bt

        def cols_phase():  # singleton_cols, singletons.rs:287-396
            cnt = {j: len(self.B[j]) for j in range(m) if self.qinv[j] < 0}
            queue = [j for j in range(m) if self.qinv[j] < 0 and cnt[j] == 1]  # D1: only the INITIAL singletons (:334)
            for j in queue:
                if cnt[j] == 0:
                    continue  # emptied by an earlier pivot in the same row (:337-339)
                (i,) = [i for (i, _) in self.B[j] if self.pinv[i] < 0]
                piv = [x for (jj, x) in Bt[i] if jj == j][0]
                if piv == 0.0 or abs(piv) < self.abstol:
                    continue  # :352-355
                r = len(self.prow)
                self.qinv[j], self.pinv[i] = r, r
                urow = []
                for (j2, x) in Bt[i]:
                    if self.qinv[j2] < 0:  # :360 (the pivot column itself is no longer < 0)
                        urow.append((j2, x))
                        cnt[j2] -= 1
                self.prow.append(i)
                self.pcol.append(j)
                self.U.append(urow)
                self.L.append([])
                self.colpiv[j] = piv
                self.log("singleton column %d (row %d, pivot %r): U row %s" % (j, i, piv, urow))

        def rows_phase():  # singleton_rows, singletons.rs:398-503
            cnt = {i: len(Bt[i]) for i in range(m) if self.pinv[i] < 0}
            queue = [i for i in range(m) if self.pinv[i] < 0 and cnt[i] == 1]  # D1 (:445)
            for i in queue:
                if cnt[i] == 0:
                    continue
                (j,) = [j for (j, _) in Bt[i] if self.qinv[j] < 0]
                piv = [x for (ii, x) in self.B[j] if ii == i][0]
                if piv == 0.0 or abs(piv) < self.abstol:
                    continue
                r = len(self.prow)
                self.qinv[j], self.pinv[i] = r, r
                lcol = []
                for (i2, x) in self.B[j]:
                    if self.pinv[i2] < 0:
                        lcol.append((i2, x / piv))  # :476
                        cnt[i2] -= 1
                self.prow.append(i)
                self.pcol.append(j)
                self.L.append(lcol)
                self.U.append([])
                self.colpiv[j] = piv
                self.log("singleton row %d (column %d, pivot %r): L column %s" % (i, j, piv, lcol))

        # NOTE on the counters: the reference counts the nonzeros of a column over ALL its rows at the start of a
        # phase (b_end - b_begin, singletons.rs:318) even if some of its rows were eliminated by the other phase
        # before; `cnt` above does the same (len of the full column / row).
        if self.nzbias is not None and self.nzbias >= 0:  # lu.nzbias.is_some() (:213): more in U
            cols_phase()
            rows_phase()
        else:
            rows_phase()
            cols_phase()
        self.rank0 = len(self.prow)

    # ------------------------------------------------------------------ setup_bump.rs:123-224
    def setup_bump(self):
        m = self.m
        self.cols, self.rows, self.colmax = {}, {}, {}
        self.clist = {k: [] for k in range(m + 2)}
        self.cnt = {}
        for j in range(m):
            if self.qinv[j] >= 0:
                continue
            ent = [[i, x] for (i, x) in self.B[j] if self.pinv[i] < 0]
            cmx = max([abs(x) for (_, x) in ent], default=0.0)
            if cmx == 0.0 or cmx < self.abstol:  # :145-156: left empty
                self.cols[j], self.colmax[j] = [], 0.0
                self._add(j, 0)
            else:
                self.cols[j], self.colmax[j] = ent, cmx
                self._add(j, len(ent))
        for i in range(m):
            if self.pinv[i] < 0:
                self.rows[i] = []
        for j in range(m):  # fill rows (:216-222): ascending column
            for (i, _) in self.cols.get(j, []):
                self.rows[i].append(j)
        self.log("bump: %d columns; count lists %s" % (len(self.cols), {k: v for k, v in self.clist.items() if v}))

    def _add(self, j, nz):
        self.clist[nz].append(j)
        self.cnt[j] = nz

    def _move(self, j, nz):  # list_move = remove + append at the tail (list.rs:89-99)
        self.clist[self.cnt[j]].remove(j)
        self._add(j, nz)

    def _remove(self, j):
        self.clist[self.cnt[j]].remove(j)
        del self.cnt[j]

    # ------------------------------------------------------------------ markowitz.rs:34-123 (columns only: search_rows = 0)
    def markowitz(self):
        m = self.m
        if self.clist[0]:
            return None, self.clist[0][0]  # :73-78
        best, pr, pc, nsearch = m * m, None, None, 0
        for nz in range(1, m + 1):
            for j in self.clist[nz]:
                cmx = self.colmax[j]
                assert not (cmx == 0.0 or cmx < self.abstol), "D2"
                tol = max(self.abstol, self.reltol * cmx)
                for (i, x) in self.cols[j]:
                    if abs(x) == 0.0 or abs(x) < tol:
                        continue
                    mc = (nz - 1) * (len(self.rows[i]) - 1)
                    if mc < best:  # strict: first seen wins (:105)
                        best, pr, pc = mc, i, j
                nsearch += 1
                if nsearch >= self.maxsearch:
                    self.nsearch += nsearch
                    return pr, pc
        self.nsearch += nsearch
        return pr, pc

    # ------------------------------------------------------------------ factorize_bump.rs:12-49, pivot.rs:48-112
    def factorize_bump(self):
        m = self.m
        while len(self.prow) + self.rankdef < m:
            pr, pc = self.markowitz()
            assert pc is not None
            if pr is None:
                self._remove(pc)
                self.rankdef += 1
                self.kinds[5] += 1
                self.log("empty column %d: rank deficiency" % pc)
                continue
            nzc, nzr = len(self.cols[pc]), len(self.rows[pr])
            rank = len(self.prow)
            if nzr == 1:
                self.pivot_singleton_row(pr, pc)
                k = 0
            elif nzc == 1:
                self.pivot_singleton_col(pr, pc)
                k = 1
            elif nzc == 2:
                self.pivot_doubleton_col(pr, pc)
                k = 2
            else:
                assert nzc - 1 <= 30, "model handles small pivot columns only"
                self.pivot_small(pr, pc)
                k = 3
            self.kinds[k] += 1
            for (j, _) in list(self.U[rank]):  # :98-106
                if self.colmax[j] == 0.0 or self.colmax[j] < self.abstol:
                    self.remove_col(j)
            self.flops += (nzc - 1) * (nzr - 1)
            self.pinv[pr] = self.qinv[pc] = rank
            self.prow.append(pr)
            self.pcol.append(pc)
            self.log("pivot %d: (row %d, col %d) %s  nz_col=%d nz_row=%d  L %s  U %s" %
                     (rank, pr, pc, ["singleton_row", "singleton_col", "doubleton_col", "small"][k], nzc, nzr, self.L[rank], self.U[rank]))
            self.log("          count lists now %s" % {k2: v for k2, v in self.clist.items() if v})

    def _finish_pivot(self, pr, pc, pivot):
        self.colmax[pc] = pivot
        self.colpiv[pc] = pivot
        self.cols[pc] = []
        self.rows[pr] = []
        self._remove(pc)

    # pivot.rs:835-926
    def pivot_singleton_row(self, pr, pc):
        col = self.cols[pc]
        pivot = [x for (i, x) in col if i == pr][0]
        self.L.append([(i, x / pivot) for (i, x) in col if i != pr and abs(x / pivot) > self.droptol])
        self.U.append([])
        for (i, _) in col:
            if i == pr:
                continue
            row = self.rows[i]
            w = row.index(pc)
            row[w] = row[-1]  # last entry into the hole (:902-903)
            row.pop()
        self._finish_pivot(pr, pc, pivot)

    # pivot.rs:928-1025
    def pivot_singleton_col(self, pr, pc):
        pivot = self.cols[pc][0][1]
        urow = []
        for j in list(self.rows[pr]):
            if j == pc:
                continue
            col = self.cols[j]
            w = [t for t, (i, _) in enumerate(col) if i == pr][0]
            xrj = col[w][1]
            cmx = max([abs(x) for t, (_, x) in enumerate(col) if t != w], default=0.0)
            if abs(xrj) > self.droptol:
                urow.append((j, xrj))
            col[w] = col[-1]  # (:991-993)
            col.pop()
            self._move(j, len(col))
            self.colmax[j] = cmx
        self.U.append(urow)
        self.L.append([])
        self._finish_pivot(pr, pc, pivot)

    # pivot.rs:1027-1331
    def pivot_doubleton_col(self, pr, pc):
        col = self.cols[pc]
        if col[0][0] != pr:
            col[0], col[1] = col[1], col[0]
        pivot = col[0][1]
        other_row, other_value = col[1]
        R = self.rows[pr]
        w = R.index(pc)
        R[0], R[w] = R[w], R[0]
        urow, fill_cols, cancelled = [], [], set()
        for j in R[1:]:
            cj = self.cols[j]
            wp = [t for t, (i, _) in enumerate(cj) if i == pr][0]
            wo = [t for t, (i, _) in enumerate(cj) if i == other_row]
            wo = wo[0] if wo else None
            cmx = max([abs(x) for t, (i, x) in enumerate(cj) if i != pr and i != other_row], default=0.0)
            xrj = cj[wp][1]
            if abs(xrj) > self.droptol:
                urow.append((j, xrj))
            if wo is None:
                x = -xrj * (other_value / pivot)  # :1151
                if abs(x) > self.droptol:
                    cj[wp] = [other_row, x]  # stored where the pivot row entry was; no list move (:1153-1161)
                    fill_cols.append(j)
                    cmx = max(cmx, abs(x))
                else:
                    cj[wp] = cj[-1]
                    cj.pop()
                    self._move(j, len(cj))
            else:
                end = len(cj) - 1
                cj[wp] = cj[end]
                cj.pop()
                if wo == end:
                    wo = wp
                cj[wo][1] -= xrj * (other_value / pivot)  # :1191
                x = abs(cj[wo][1])
                if x <= self.droptol:
                    cj[wo] = cj[-1]
                    cj.pop()
                    cancelled.add(j)
                elif x > cmx:
                    cmx = x
                self._move(j, len(cj))
            self.colmax[j] = cmx
        self.U.append(urow)
        orow = self.rows[other_row]
        if cancelled:  # ordered compress without the pivot column and the cancelled columns (:1224-1246)
            orow[:] = [j for j in orow if j != pc and j not in cancelled]
        else:
            w = orow.index(pc)
            orow[w] = orow[-1]
            orow.pop()
        orow.extend(fill_cols)  # fill-in appended in pivot-row order (:1277-1284)
        x = other_value / pivot
        self.L.append([(other_row, x)] if abs(x) > self.droptol else [])
        self._finish_pivot(pr, pc, pivot)

    # pivot.rs:460-833
    def pivot_small(self, pr, pc):
        C = self.cols[pc]
        w = [t for t, (i, _) in enumerate(C) if i == pr][0]
        C[0], C[w] = C[w], C[0]  # swap (:169-170 / :524-525)
        pivot = C[0][1]
        R = self.rows[pr]
        w = R.index(pc)
        R[0], R[w] = R[w], R[0]  # (:185 / :540)
        cnz1 = len(C) - 1
        posmap = {C[p][0]: p for p in range(1, cnz1 + 1)}
        urow, masks = [], []
        for j in R[1:]:
            work = [0.0] * (cnz1 + 1)
            kept, where, cmx = [], None, 0.0
            for (i, x) in self.cols[j]:
                p = posmap.get(i, 0)
                if p > 0:
                    work[p] = x
                else:
                    if i == pr:
                        where = len(kept)
                    elif abs(x) > cmx:
                        cmx = abs(x)
                    kept.append([i, x])
            kept[0], kept[where] = kept[where], kept[0]  # pivot row entry to the front (:604-605)
            xrj = kept[0][1]
            a = xrj / pivot
            mask = 0
            for p in range(1, cnz1 + 1):
                work[p] -= a * C[p][1]  # :631-634, two roundings
                x = abs(work[p])
                if x > self.droptol:
                    kept.append([C[p][0], work[p]])
                    cmx = max(cmx, x)
                else:
                    mask |= 1 << (p - 1)  # cancellation (:656-660)
            masks.append(mask)
            if abs(xrj) > self.droptol:
                urow.append((j, xrj))
            self.cols[j] = kept[1:]  # the pivot row entry leaves (:673-674)
            self._move(j, len(self.cols[j]))
            self.colmax[j] = cmx
        self.U.append(urow)
        Rset = set(R)
        for p in range(1, cnz1 + 1):
            i = C[p][0]
            row = [j for j in self.rows[i] if j not in Rset]  # compress, order kept (:712-724)
            row += [j for t, j in enumerate(R[1:]) if not (masks[t] >> (p - 1)) & 1]  # (:748-755)
            self.rows[i] = row
        self.L.append([(C[p][0], C[p][1] / pivot) for p in range(1, cnz1 + 1) if abs(C[p][1] / pivot) > self.droptol])
        self._finish_pivot(pr, pc, pivot)

    # pivot.rs:1333-1381
    def remove_col(self, j):
        for (i, _) in self.cols[j]:
            row = self.rows[i]
            w = row.index(j)
            row[w] = row[-1]
            row.pop()
        self.colmax[j] = 0.0
        self.cols[j] = []
        self._move(j, 0)
        self.log("          column %d sank below abstol: removed" % j)

    # ------------------------------------------------------------------ build_factors.rs:179-223 + get_factors.rs:48-180
    def factors(self):
        m = self.m
        rank = len(self.prow)
        prow = self.prow + [i for i in range(m) if self.pinv[i] < 0]
        pcol = self.pcol + [j for j in range(m) if self.qinv[j] < 0]
        pinv = {i: k for k, i in enumerate(prow)}
        qinv = {j: k for k, j in enumerate(pcol)}
        L = [[(k, 1.0)] + sorted((pinv[i], x) for (i, x) in (self.L[k] if k < rank else [])) for k in range(m)]
        Ucols = [[] for _ in range(m)]
        for k in range(rank):
            for (j, x) in self.U[k]:
                if qinv[j] < rank:  # entries in columns that never became pivotal are dropped (build_factors.rs:318-337)
                    Ucols[qinv[j]].append((k, x))
        for k in range(m):
            Ucols[k] = sorted(Ucols[k]) + [(k, self.colpiv[pcol[k]] if k < rank else 1.0)]
        return dict(rowperm=prow, colperm=pcol, L=L, U=Ucols, rank=rank)

    def factorize(self):
        self.singletons()
        self.setup_bump()
        self.factorize_bump()
        return self.factors()


def main():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from blu_amd.matrices import simple_rs
    cp, ri, v, b, x = simple_rs()
    out = []
    mdl = Model(10, cp, ri, v, log=lambda *a: out.append(" ".join(str(t) for t in a)))
    f = mdl.factorize()
    out.append("rowperm %s" % f["rowperm"])
    out.append("colperm %s" % f["colperm"])
    out.append("nsearch_pivot %d  factor_flops %d  pivots by kind %s" % (mdl.nsearch, mdl.flops, mdl.kinds))
    for k in range(10):
        out.append("L[:,%d] = %s" % (k, f["L"][k]))
    for k in range(10):
        out.append("U[:,%d] = %s" % (k, f["U"][k]))
    print("\n".join(out))


if __name__ == "__main__":
    main()
