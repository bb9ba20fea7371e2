/* orc_solve_sparse.c -- TEST INFRASTRUCTURE (see blu_oracle.h).
 * Restatement of the sparse solve of /root/reference: src/solve_sparse.rs, src/lu/solve_sparse.rs,
 * src/lu/solve_symbolic.rs, src/lu/dfs.rs, src/lu/solve_triangular.rs and BLU::solve_sparse +
 * lu_clear_lhs of src/blu.rs.  Same arrays, same aliases (pattern_symb/pattern = the two halves of
 * iwork1, the DFS position stack lives in the f64 array work1), same order of every operation, so
 * that the pattern ilhs comes out in the reference's order (DFS post-order = topological order). */
#include "orc_internal.h"

/* dfs_end / dfs_begin -- dfs.rs:49-96 / :99-145 (adapted from CSparse).  `M` = the marker value. */
static lu_int dfs_end(lu_int i, const lu_int *begin, const lu_int *end, const lu_int *index, lu_int top,
                      lu_int *xi, double *pstack, lu_int *marked, lu_int M)
{
    lu_int head = 0;
    ORC_ASSERT(marked[i] != M);
    xi[0] = i;
    while (head >= 0) {
        i = xi[head];
        if (marked[i] != M) { /* node i has not been visited */
            marked[i] = M;
            pstack[head] = (double)begin[i];
        }
        int done = 1;
        for (lu_int p = (lu_int)pstack[head]; p < end[i]; p++) { /* continue dfs at node i */
            lu_int inext = index[p];
            if (marked[inext] == M) continue; /* skip visited node */
            pstack[head] = (double)(p + 1);
            xi[++head] = inext; /* start dfs at node inext */
            done = 0;
            break;
        }
        if (done) { /* node i has no unvisited neighbours */
            head--;
            xi[--top] = i;
        }
    }
    return top;
}

static lu_int dfs_begin(lu_int i, const lu_int *begin, const lu_int *index, lu_int top, lu_int *xi,
                        double *pstack, lu_int *marked, lu_int M)
{
    lu_int head = 0;
    ORC_ASSERT(marked[i] != M);
    xi[0] = i;
    while (head >= 0) {
        i = xi[head];
        if (marked[i] != M) {
            marked[i] = M;
            pstack[head] = (double)begin[i];
        }
        int done = 1;
        lu_int p = (lu_int)pstack[head];
        while (index[p] >= 0) { /* neighbour list terminated by a negative index */
            lu_int inext = index[p];
            if (marked[inext] == M) {
                p++;
                continue;
            }
            pstack[head] = (double)(p + 1);
            xi[++head] = inext;
            done = 0;
            break;
        }
        if (done) {
            head--;
            xi[--top] = i;
        }
    }
    return top;
}

/* dfs -- dfs.rs:25-46 */
lu_int orc_dfs(lu_int i, const lu_int *begin, const lu_int *end, const lu_int *index, lu_int top,
                      lu_int *xi, double *pstack, lu_int *marked, lu_int M)
{
    if (marked[i] == M) return top;
    return end ? dfs_end(i, begin, end, index, top, xi, pstack, marked, M)
               : dfs_begin(i, begin, index, top, xi, pstack, marked, M);
}

/* solve_symbolic -- solve_symbolic.rs:19-40 */
lu_int orc_solve_symbolic(lu_int m, const lu_int *begin, const lu_int *end, const lu_int *index,
                                 lu_int nrhs, const lu_int *irhs, lu_int *ilhs, double *pstack,
                                 lu_int *marked, lu_int M)
{
    lu_int top = m;
    for (lu_int n = 0; n < nrhs; n++)
        if (marked[irhs[n]] != M) top = orc_dfs(irhs[n], begin, end, index, top, ilhs, pstack, marked, M);
    return top;
}

/* solve_triangular -- solve_triangular.rs:27-136.  The four variants of the reference differ only in
 * how a column ends (end[] or a negative index) and in the division by the pivot. */
lu_int orc_solve_triangular(lu_int nz_symb, const lu_int *pattern_symb, const lu_int *begin,
                                   const lu_int *end, const lu_int *index, const double *value,
                                   const double *pivot, double droptol, double *lhs, lu_int *pattern,
                                   lu_int *flops)
{
    lu_int nz = 0, flop_count = 0;
    for (lu_int n = 0; n < nz_symb; n++) {
        lu_int ipivot = pattern_symb[n];
        if (lhs[ipivot] != 0.0) {
            double x;
            if (pivot) {
                lhs[ipivot] /= pivot[ipivot];
                flop_count++;
            }
            x = lhs[ipivot];
            if (end) {
                for (lu_int pos = begin[ipivot]; pos < end[ipivot]; pos++) {
                    lhs[index[pos]] -= x * value[pos];
                    flop_count++;
                }
            } else {
                for (lu_int pos = begin[ipivot]; index[pos] >= 0; pos++) {
                    lhs[index[pos]] -= x * value[pos];
                    flop_count++;
                }
            }
            if (fabs(x) > droptol) pattern[nz++] = ipivot;
            else lhs[ipivot] = 0.0;
        }
    }
    *flops += flop_count;
    return nz;
}

/* lu::solve_sparse -- lu/solve_sparse.rs:11-360 */
void orc_lu_solve_sparse(orc_lu *lu, lu_int nrhs, const lu_int *irhs, const double *xrhs, lu_int *p_nlhs,
                         lu_int *ilhs, double *xlhs, char trans)
{
    const lu_int m = lu->m, nforrest = lu->nforrest, pivotlen = lu->pivotlen;
    const lu_int nz_sparse = (lu_int)(lu->sparse_thres * (double)m);
    const double droptol = lu->droptol;
    const lu_int *p = P_(lu), *pmap = PMAP(lu), *qmap = QMAP(lu), *eta_row = ETA_ROW(lu);
    const lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    const lu_int *l_begin = L_BEGIN(lu), *lt_begin = LT_BEGIN(lu), *lt_begin_p = LT_BEGIN_P(lu);
    const lu_int *u_begin = lu->u_begin, *r_begin = R_BEGIN(lu), *w_begin = lu->w_begin, *w_end = lu->w_end;
    const double *col_pivot = lu->col_pivot, *row_pivot = lu->row_pivot;
    const lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    const double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = MARKED(lu);
    lu_int *pattern_symb = IWORK1(lu), *pattern = IWORK1(lu) + m; /* :55, :183 */
    double *work = lu->work0, *pstack = lu->work1;
    lu_int l_flops = 0, u_flops = 0, r_flops = 0;
    lu_int top, nz_symb, nz, M;

    if (trans == 't' || trans == 'T') {
        /* ---- transposed system (:51-179): U' (column file W), etas backwards, L' (row-wise L) */
        M = ++lu->marker;
        top = orc_solve_symbolic(m, w_begin, w_end, w_index, nrhs, irhs, pattern_symb, pstack, marked, M);
        nz_symb = m - top;
        for (lu_int n = 0; n < nrhs; n++) work[irhs[n]] = xrhs[n];
        nz = orc_solve_triangular(nz_symb, pattern_symb + top, w_begin, w_end, w_index, w_value, col_pivot,
                                  droptol, work, pattern, &u_flops);
        /* permute solution into xlhs, map pattern from column to row indices (:95-106) */
        M = ++lu->marker;
        for (lu_int n = 0; n < nz; n++) {
            lu_int j = pattern[n], i = pmap[j];
            pattern[n] = i;
            xlhs[i] = work[j];
            work[j] = 0.0;
            marked[i] = M;
        }
        /* update etas, fill-in appended to the pattern (:108-125) */
        for (lu_int t = nforrest - 1; t >= 0; t--) {
            lu_int ipivot = eta_row[t];
            if (xlhs[ipivot] != 0.0) {
                double x = xlhs[ipivot];
                for (lu_int pos = r_begin[t]; pos < r_begin[t + 1]; pos++) {
                    lu_int i = l_index[pos];
                    if (marked[i] != M) {
                        marked[i] = M;
                        pattern[nz++] = i;
                    }
                    xlhs[i] -= x * l_value[pos];
                    r_flops++;
                }
            }
        }
        if (nz <= nz_sparse) { /* sparse solve with L' (:127-158) */
            M = ++lu->marker;
            top = orc_solve_symbolic(m, lt_begin, NULL, l_index, nz, pattern, pattern_symb, pstack, marked, M);
            nz_symb = m - top;
            nz = orc_solve_triangular(nz_symb, pattern_symb + top, lt_begin, NULL, l_index, l_value, NULL,
                                      droptol, xlhs, ilhs, &l_flops);
            *p_nlhs = nz;
        } else { /* sequential solve with L' (:159-179) */
            nz = 0;
            for (lu_int k = m - 1; k >= 0; k--) {
                lu_int ipivot = p[k];
                if (xlhs[ipivot] != 0.0) {
                    double x = xlhs[ipivot];
                    for (lu_int pos = lt_begin_p[k]; l_index[pos] >= 0; pos++) {
                        xlhs[l_index[pos]] -= x * l_value[pos];
                        l_flops++;
                    }
                    if (fabs(x) > droptol) ilhs[nz++] = ipivot;
                    else xlhs[ipivot] = 0.0;
                }
            }
            *p_nlhs = nz;
        }
    } else {
        /* ---- forward system (:180-346): L (column-wise), etas, U (column-wise by row index) */
        M = ++lu->marker;
        top = orc_solve_symbolic(m, l_begin, NULL, l_index, nrhs, irhs, pattern_symb, pstack, marked, M);
        nz_symb = m - top;
        for (lu_int n = 0; n < nrhs; n++) work[irhs[n]] = xrhs[n];
        nz = orc_solve_triangular(nz_symb, pattern_symb + top, l_begin, NULL, l_index, l_value, NULL, droptol,
                                  work, pattern, &l_flops);
        /* unmark cancellation (:227-243) */
        if (nz < nz_symb) {
            lu_int t = top, n = 0;
            while (n < nz) {
                lu_int i = pattern_symb[t];
                if (i == pattern[n]) n++;
                else marked[i] -= 1;
                t++;
            }
            while (t < m) {
                marked[pattern_symb[t]] -= 1;
                t++;
            }
        }
        /* update etas, fill-in appended to the pattern (:245-262) */
        {
            lu_int pos = r_begin[0];
            for (lu_int t = 0; t < nforrest; t++) {
                lu_int ipivot = eta_row[t];
                double x = 0.0;
                while (pos < r_begin[t + 1]) {
                    x += work[l_index[pos]] * l_value[pos];
                    pos++;
                }
                work[ipivot] -= x;
                if (x != 0.0 && marked[ipivot] != M) {
                    marked[ipivot] = M;
                    pattern[nz++] = ipivot;
                }
            }
            r_flops += r_begin[nforrest] - r_begin[0];
        }
        if (nz <= nz_sparse) { /* sparse solve with U (:264-306) */
            M = ++lu->marker;
            top = orc_solve_symbolic(m, u_begin, NULL, u_index, nz, pattern, pattern_symb, pstack, marked, M);
            nz_symb = m - top;
            nz = orc_solve_triangular(nz_symb, pattern_symb + top, u_begin, NULL, u_index, u_value, row_pivot,
                                      droptol, work, ilhs, &u_flops);
            /* permute into xlhs, map pattern from row to column indices */
            for (lu_int n = 0; n < nz; n++) {
                lu_int i = ilhs[n], j = qmap[i];
                ilhs[n] = j;
                xlhs[j] = work[i];
                work[i] = 0.0;
            }
        } else { /* sequential solve with U (:307-334) */
            nz = 0;
            for (lu_int k = pivotlen - 1; k >= 0; k--) {
                lu_int ipivot = pivotrow[k], jpivot = pivotcol[k];
                if (work[ipivot] != 0.0) {
                    double x = work[ipivot] / row_pivot[ipivot];
                    work[ipivot] = 0.0;
                    for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) {
                        work[u_index[pos]] -= x * u_value[pos];
                        u_flops++;
                    }
                    if (fabs(x) > droptol) {
                        ilhs[nz++] = jpivot;
                        xlhs[jpivot] = x;
                    }
                }
            }
        }
        *p_nlhs = nz;
    }
    lu->l_flops += l_flops;
    lu->u_flops += u_flops;
    lu->r_flops += r_flops;
    lu->update_cost_numer += (double)r_flops;
}

/* solve_sparse -- solve_sparse.rs:36-68 */
int orc_solve_sparse(orc_lu *lu, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, lu_int *p_nzlhs,
                     lu_int *ilhs, double *lhs, char trans)
{
    if (lu->nupdate < 0) return ORC_ERROR_INVALID_CALL;
    int ok = nzrhs >= 0 && nzrhs <= lu->m;
    for (lu_int n = 0; ok && n < nzrhs; n++) ok = ok && irhs[n] < (uint64_t)lu->m;
    if (!ok) return ORC_ERROR_INVALID_ARGUMENT;
    lu_int *ir = (lu_int *)malloc((size_t)(nzrhs > 0 ? nzrhs : 1) * sizeof(lu_int));
    for (lu_int n = 0; n < nzrhs; n++) ir[n] = (lu_int)irhs[n];
    orc_lu_solve_sparse(lu, nzrhs, ir, xrhs, p_nzlhs, ilhs, lhs, trans);
    free(ir);
    return ORC_OK;
}

/* lu_clear_lhs -- blu.rs:380-395 */
void orc_clear_lhs(orc_blu *obj)
{
    const lu_int m = obj->lu.m;
    const lu_int nzsparse = (lu_int)(obj->lu.sparse_thres * (double)m);
    const lu_int nz = obj->nzlhs;
    if (nz != 0) {
        if (nz <= nzsparse) {
            for (lu_int q = 0; q < nz; q++) obj->lhs[obj->ilhs[q]] = 0.0;
        } else {
            for (lu_int i = 0; i < m; i++) obj->lhs[i] = 0.0;
        }
        obj->nzlhs = 0;
    }
}

/* BLU::solve_sparse -- blu.rs:207-225.  The solution stays in obj->lhs / obj->ilhs[0..nzlhs). */
int orc_blu_solve_sparse(orc_blu *obj, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, char trans)
{
    orc_clear_lhs(obj);
    return orc_solve_sparse(&obj->lu, nzrhs, irhs, xrhs, &obj->nzlhs, obj->ilhs, obj->lhs, trans);
}

/* read-out of the solution held by the object (test plumbing, not in the reference) */
lu_int orc_blu_nzlhs(const orc_blu *obj) { return obj->nzlhs; }
void orc_blu_get_lhs(const orc_blu *obj, lu_int *ilhs, double *lhs)
{
    for (lu_int q = 0; q < obj->nzlhs; q++) ilhs[q] = obj->ilhs[q];
    for (lu_int i = 0; i < obj->lu.m; i++) lhs[i] = obj->lhs[i];
}
