/* orc_markowitz.c -- CPU oracle (test infrastructure).
 * Follows src/lu/markowitz.rs and src/lu/factorize_bump.rs of /root/reference. */
#include "orc_internal.h"

/* done -- markowitz.rs:195-219 */
static int mk_done(orc_lu *lu, lu_int pivot_row, lu_int pivot_col, lu_int nsearch,
                   lu_int min_colnz, lu_int min_rownz, double tic)
{
    lu->pivot_row = pivot_row;
    lu->pivot_col = pivot_col;
    lu->nsearch_pivot += nsearch;
    if (min_colnz >= 0) lu->min_colnz = min_colnz;
    if (min_rownz >= 0) lu->min_rownz = min_rownz;
    lu->time_search_pivot += orc_now() - tic;
    return ORC_OK;
}

/* markowitz -- markowitz.rs:34-193 */
int orc_markowitz(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    const lu_int *w_index = lu->w_index;
    const double *w_value = lu->w_value;
    const lu_int *colcount_flink = lu->colcount_flink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    const double *colmax = lu->col_pivot;
    const double abstol = lu->abstol, reltol = lu->reltol;
    const lu_int maxsearch = lu->maxsearch;
    const lu_int search_rows = lu->search_rows;
    const lu_int nz_start = search_rows != 0 ? (lu->min_colnz < lu->min_rownz ? lu->min_colnz : lu->min_rownz)
                                             : lu->min_colnz;

    const int64_t m64 = m;
    double tic = orc_now();
    lu_int pivot_row = -1, pivot_col = -1; /* best pivot so far */
    int64_t mc64 = m64 * m64;              /* Markowitz cost of best pivot so far */
    lu_int nsearch = 0;
    lu_int min_colnz = -1, min_rownz = -1; /* None */
    ORC_ASSERT(nz_start >= 1);

    /* If the active submatrix contains empty columns, choose one and return
     * with pivot_row = None. (:73-78) */
    if (colcount_flink[m] != m) {
        pivot_col = colcount_flink[m];
        ORC_ASSERT(pivot_col >= 0 && pivot_col < m);
        ORC_ASSERT(w_end[pivot_col] == w_begin[pivot_col]);
        return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
    }

    for (lu_int nz = nz_start; nz <= m; nz++) {
        /* Search columns with nz nonzeros. (:81-123) */
        lu_int j = colcount_flink[m + nz];
        while (j < m) {
            if (min_colnz < 0) min_colnz = nz;
            ORC_ASSERT(w_end[j] - w_begin[j] == nz);
            double cmx = colmax[j];
            ORC_ASSERT(cmx >= 0.0);
            if (cmx == 0.0 || cmx < abstol) {
                /* D2: the reference `continue`s here without advancing j: an
                 * infinite loop.  Unreachable under the invariants; trap. */
                fprintf(stderr, "blu oracle: markowitz D2 (column %lld with colmax < abstol in a count list): "
                                "the reference would hang here\n", (long long)j);
                abort();
            }
            double tol = fmax(abstol, reltol * cmx);
            for (lu_int pos = w_begin[j]; pos < w_end[j]; pos++) {
                double x = fabs(w_value[pos]);
                if (x == 0.0 || x < tol) continue;
                lu_int i = w_index[pos];
                ORC_ASSERT(i >= 0 && i < m);
                int64_t nz1 = nz;
                int64_t nz2 = w_end[m + i] - w_begin[m + i];
                ORC_ASSERT(nz2 >= 1);
                int64_t mc = (nz1 - 1) * (nz2 - 1);
                if (mc < mc64) {
                    mc64 = mc;
                    pivot_row = i;
                    pivot_col = j;
                    if (search_rows != 0 && mc64 <= (nz1 - 1) * (nz1 - 1))
                        return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
                }
            }
            /* We have seen at least one eligible pivot in column j. */
            ORC_ASSERT(mc64 < m64 * m64);
            nsearch++;
            if (nsearch >= maxsearch)
                return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
            j = colcount_flink[j];
        }
        ORC_ASSERT(j == m + nz);

        if (search_rows == 0) continue;

        /* Search rows with nz nonzeros. (:129-190) */
        lu_int i = rowcount_flink[m + nz];
        while (i < m) {
            if (min_rownz < 0) min_rownz = nz;
            /* rowcount_flink[i] might be changed below, so keep a copy */
            lu_int inext = rowcount_flink[i];
            ORC_ASSERT(w_end[m + i] - w_begin[m + i] == nz);
            int cheap = 0; /* row has entries with Markowitz cost < MC? */
            int found = 0; /* eligible pivot found? */
            for (lu_int pos = w_begin[m + i]; pos < w_end[m + i]; pos++) {
                lu_int jj = w_index[pos];
                ORC_ASSERT(jj >= 0 && jj < m);
                int64_t nz1 = nz;
                int64_t nz2 = w_end[jj] - w_begin[jj];
                ORC_ASSERT(nz2 >= 1);
                int64_t mc = (nz1 - 1) * (nz2 - 1);
                if (mc >= mc64) continue;
                cheap = 1;
                double cmx = colmax[jj];
                ORC_ASSERT(cmx >= 0.0);
                if (cmx == 0.0 || cmx < abstol) continue;
                /* find position of pivot in column file */
                lu_int where_ = w_begin[jj];
                while (w_index[where_] != i) {
                    ORC_ASSERT(where_ < w_end[jj] - 1);
                    where_++;
                }
                double x = fabs(w_value[where_]);
                if (x >= abstol && x >= reltol * cmx) {
                    found = 1;
                    mc64 = mc;
                    pivot_row = i;
                    pivot_col = jj;
                    if (mc64 <= nz1 * (nz1 - 1))
                        return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
                }
            }
            /* If row i has cheap entries but none of them is numerically
             * acceptable, then don't search the row again until updated. */
            if (cheap != 0 && found == 0) {
                orc_list_move(i, m + 1, rowcount_flink, rowcount_blink, m, NULL);
            } else {
                ORC_ASSERT(mc64 < m64 * m64);
                nsearch++;
                if (nsearch >= maxsearch)
                    return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
            }
            i = inext;
        }
        ORC_ASSERT(i == m + nz);
    }
    return mk_done(lu, pivot_row, pivot_col, nsearch, min_colnz, min_rownz, tic);
}

/* factorize_bump -- factorize_bump.rs:12-49 */
int orc_factorize_bump(orc_lu *lu)
{
    const lu_int m = lu->m;
    while (lu->rank + lu->rankdef < m) {
        /* test hook (not in the reference) */
        if (lu->stop_after_pivots >= 0 && lu->pivot_col < 0 && lu->rank + lu->rankdef >= lu->stop_after_pivots)
            return ORC_STOPPED;

        /* Find pivot element. Markowitz search need not be called if the
         * previous call to pivot() returned for reallocation. */
        if (lu->pivot_col < 0) {
            int st = orc_markowitz(lu);
            if (st != ORC_OK) return st;
        }
        ORC_ASSERT(lu->pivot_col >= 0);

        if (lu->pivot_row < 0) {
            /* Eliminate empty column without choosing a pivot. */
            orc_list_remove(lu->colcount_flink, lu->colcount_blink, lu->pivot_col);
            lu->pivot_col = -1;
            lu->rankdef++;
            lu->npivot_kind[5]++; /* test hook */
        } else {
            /* Eliminate pivot. This may require reallocation. */
            ORC_ASSERT(lu->pinv[lu->pivot_row] == -1);
            ORC_ASSERT(lu->qinv[lu->pivot_col] == -1);
            int st = orc_pivot(lu);
            if (st != ORC_OK) return st;
            lu->pinv[lu->pivot_row] = lu->rank;
            lu->qinv[lu->pivot_col] = lu->rank;
            lu->pivot_col = -1;
            lu->pivot_row = -1;
            lu->rank++;
        }
    }
    return ORC_OK;
}
