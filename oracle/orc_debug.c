/* orc_debug.c -- CPU oracle (test infrastructure): layout-independent state
 * dumps used by the step-wise GPU parity tests.  Nothing here exists in the
 * reference; it only READS the state the restated kernels maintain. */
#include "orc_internal.h"

void orc_dbg_set_stop(orc_lu *lu, lu_int npivots) { lu->stop_after_pivots = npivots; }
void orc_dbg_set_fix_d3(orc_lu *lu, int on) { lu->fix_d3 = on; }
lu_int orc_dbg_d3_hits(const orc_lu *lu) { return lu->d3_hits; }

/* which: 0 = column file entries, 1 = row file entries (valid while task == FACTORIZE_BUMP) */
lu_int orc_dbg_active_nnz(const orc_lu *lu, int which)
{
    const lu_int m = lu->m;
    lu_int n = 0;
    for (lu_int k = 0; k < m; k++) {
        lu_int line = which ? m + k : k;
        n += lu->w_end[line] - lu->w_begin[line];
    }
    return n;
}

/* Active submatrix between two pivots: column file (ordered entries), row file
 * (ordered pattern), colmax, pinv/qinv and the four count-list link arrays. */
void orc_dbg_active_state(const orc_lu *lu, lu_int *colptr, lu_int *colidx, double *colval,
                          lu_int *rowptr, lu_int *rowidx, double *colmax, lu_int *pinv, lu_int *qinv,
                          lu_int *col_flink, lu_int *col_blink, lu_int *row_flink, lu_int *row_blink)
{
    const lu_int m = lu->m;
    lu_int put = 0;
    for (lu_int j = 0; j < m; j++) {
        colptr[j] = put;
        for (lu_int pos = lu->w_begin[j]; pos < lu->w_end[j]; pos++) {
            colidx[put] = lu->w_index[pos];
            colval[put] = lu->w_value[pos];
            put++;
        }
    }
    colptr[m] = put;
    put = 0;
    for (lu_int i = 0; i < m; i++) {
        rowptr[i] = put;
        for (lu_int pos = lu->w_begin[m + i]; pos < lu->w_end[m + i]; pos++) rowidx[put++] = lu->w_index[pos];
    }
    rowptr[m] = put;
    memcpy(colmax, lu->col_pivot, (size_t)m * sizeof(double));
    memcpy(pinv, lu->pinv, (size_t)m * sizeof(lu_int));
    memcpy(qinv, lu->qinv, (size_t)m * sizeof(lu_int));
    memcpy(col_flink, lu->colcount_flink, (size_t)(2 * m + 2) * sizeof(lu_int));
    memcpy(col_blink, lu->colcount_blink, (size_t)(2 * m + 2) * sizeof(lu_int));
    memcpy(row_flink, lu->rowcount_flink, (size_t)(2 * m + 2) * sizeof(lu_int));
    memcpy(row_blink, lu->rowcount_blink, (size_t)(2 * m + 2) * sizeof(lu_int));
}

/* which: 0 = L entries so far (without terminators), 1 = U entries so far */
lu_int orc_dbg_partial_nz(const orc_lu *lu, int which)
{
    return which ? lu->u_begin[lu->rank] : lu->l_begin_p[lu->rank] - lu->rank;
}

/* L columns (stage order, terminators stripped) and U rows of stages 0..rank-1 */
void orc_dbg_partial_lu(const orc_lu *lu, lu_int *lptr, lu_int *lidx, double *lval,
                        lu_int *uptr, lu_int *uidx, double *uval)
{
    const lu_int rank = lu->rank;
    lu_int put = 0;
    for (lu_int k = 0; k < rank; k++) {
        lptr[k] = put;
        for (lu_int pos = lu->l_begin_p[k]; lu->l_index[pos] >= 0; pos++) {
            lidx[put] = lu->l_index[pos];
            lval[put] = lu->l_value[pos];
            put++;
        }
    }
    lptr[rank] = put;
    for (lu_int k = 0; k <= rank; k++) uptr[k] = lu->u_begin[k];
    for (lu_int pos = 0; pos < lu->u_begin[rank]; pos++) {
        uidx[pos] = lu->u_index[pos];
        uval[pos] = lu->u_value[pos];
    }
}
