/* orc_build_factors.c -- CPU oracle (test infrastructure).
 * Follows src/lu/build_factors.rs and src/get_factors.rs of /root/reference. */
#include "orc_internal.h"

/* build_factors -- build_factors.rs:113-423 */
int orc_build_factors(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const lu_int l_mem = lu->l_mem, u_mem = lu->u_mem, w_mem = lu->w_mem;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    lu_int *pivotcol = PIVOTCOL(lu);
    lu_int *pivotrow = PIVOTROW(lu);
    lu_int *l_begin = L_BEGIN(lu);
    lu_int *lt_begin = LT_BEGIN(lu);
    lu_int *lt_begin_p = LT_BEGIN_P(lu);
    lu_int *r_begin = R_BEGIN(lu);
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *iwork1 = IWORK1(lu);

    /* (:151-153) */
    lu_int l_nz = lu->l_begin_p[rank];
    l_nz -= rank; /* because each column is terminated by -1 */
    lu_int u_nz = lu->u_begin[rank]; /* might be decreased when rank < m */

    /* Calculate memory and reallocate. (:160-177) */
    lu_int need = 2 * (l_nz + m);
    if (l_mem < need) {
        lu->addmem_l = need - l_mem;
        return ORC_REALLOCATE;
    }
    need = u_nz + m + 1;
    if (u_mem < need) {
        lu->addmem_u = need - u_mem;
        return ORC_REALLOCATE;
    }
    need = u_nz + orc_trunc(stretch * (double)u_nz) + m * pad;
    if (w_mem < need) {
        lu->addmem_w = need - w_mem;
        return ORC_REALLOCATE;
    }

    /* ---- Build permutations (:179-223) ---- */
    lu_int lrank = rank;
    for (lu_int i = 0; i < m; i++) {
        if (lu->pinv[i] < 0) {
            lu->pinv[i] = lrank;
            lrank++;
        }
        pivotrow[lu->pinv[i]] = i;
    }
    ORC_ASSERT(lrank == m);
    lrank = rank;
    for (lu_int j = 0; j < m; j++) {
        if (lu->qinv[j] < 0) {
            lu->qinv[j] = lrank;
            lrank++;
        }
        pivotcol[lu->qinv[j]] = j;
    }
    ORC_ASSERT(lrank == m);

    /* Dependent columns get unit pivot elements. */
    for (lu_int k = rank; k < m; k++) lu->col_pivot[pivotcol[k]] = 1.0;

    /* ---- Lower triangular factor (:225-281) ---- */
    /* L columnwise. If rank < m, then complete with unit columns. */
    lu_int put = lu->l_begin_p[rank];
    for (lu_int k = rank; k < m; k++) {
        l_index[put] = -1;
        put++;
        lu->l_begin_p[k + 1] = put;
    }
    ORC_ASSERT(lu->l_begin_p[m] == l_nz + m);
    for (lu_int i = 0; i < m; i++) l_begin[i] = lu->l_begin_p[lu->pinv[i]];

    /* L rowwise. */
    memset(iwork1, 0, (size_t)(2 * m + 2) * sizeof(lu_int)); /* iwork1.fill(0): whole 2m+2 array */
    for (lu_int get = 0; get < l_nz + m; get++) {
        lu_int i = l_index[get];
        if (i >= 0) iwork1[i]++;
    }
    put = l_nz + m; /* L rowwise starts here */
    for (lu_int k = 0; k < m; k++) {
        lu_int i = pivotrow[k];
        lt_begin_p[k] = put;
        lt_begin[i] = put;
        put += iwork1[i];
        l_index[put] = -1; /* terminate row */
        put++;
        iwork1[i] = lt_begin_p[k];
    }
    ORC_ASSERT(put == 2 * (l_nz + m));
    for (lu_int k = 0; k < m; k++) { /* fill rows */
        lu_int ipivot = pivotrow[k];
        lu_int get = lu->l_begin_p[k];
        while (l_index[get] >= 0) {
            lu_int p = iwork1[l_index[get]]; /* put into row i */
            iwork1[l_index[get]]++;
            l_index[p] = ipivot;
            l_value[p] = l_value[get];
            get++;
        }
    }
    r_begin[0] = 2 * (l_nz + m); /* beginning of update etas */

    /* ---- Upper triangular factor (:283-384) ---- */
    /* U rowwise. */
    orc_file_empty(m, lu->w_begin, lu->w_end, lu->w_flink, lu->w_blink, w_mem);
    memset(iwork1, 0, (size_t)(2 * m + 2) * sizeof(lu_int)); /* column counts */
    put = 0;

    if (rank == m) {
        for (lu_int k = 0; k < m; k++) {
            lu_int jpivot = pivotcol[k];
            lu->w_begin[jpivot] = put;
            lu_int nz = 0;
            for (lu_int pos = lu->u_begin[k]; pos < lu->u_begin[k + 1]; pos++) {
                lu_int j = u_index[pos];
                w_index[put] = j;
                w_value[put] = u_value[pos];
                put++;
                iwork1[j]++;
                nz++;
            }
            lu->w_end[jpivot] = put;
            put += orc_trunc(stretch * (double)nz) + pad;
            orc_list_move(jpivot, 0, lu->w_flink, lu->w_blink, m, NULL);
        }
    } else {
        u_nz = 0; /* actual number of nonzeros */
        for (lu_int k = 0; k < rank; k++) {
            lu_int jpivot = pivotcol[k];
            lu->w_begin[jpivot] = put;
            lu_int nz = 0;
            for (lu_int pos = lu->u_begin[k]; pos < lu->u_begin[k + 1]; pos++) {
                lu_int j = u_index[pos];
                if (lu->qinv[j] < rank) {
                    w_index[put] = j;
                    w_value[put] = u_value[pos];
                    put++;
                    iwork1[j]++;
                    nz++;
                }
            }
            lu->w_end[jpivot] = put;
            put += orc_trunc(stretch * (double)nz) + pad;
            orc_list_move(jpivot, 0, lu->w_flink, lu->w_blink, m, NULL);
            u_nz += nz;
        }
        for (lu_int k = rank; k < m; k++) {
            lu_int jpivot = pivotcol[k];
            lu->w_begin[jpivot] = put;
            lu->w_end[jpivot] = put;
            put += pad;
            orc_list_move(jpivot, 0, lu->w_flink, lu->w_blink, m, NULL);
        }
    }
    ORC_ASSERT(put <= lu->w_end[m]);
    lu->w_begin[m] = put; /* beginning of free space */

    /* U columnwise. */
    u_index[0] = -1;
    put = 1;
    for (lu_int k = 0; k < m; k++) { /* set column pointers */
        lu_int j = pivotcol[k];
        lu_int i = pivotrow[k];
        lu_int nz = iwork1[j];
        if (nz == 0) {
            lu->u_begin[i] = 0; /* empty columns all in position 0 */
        } else {
            lu->u_begin[i] = put;
            put += nz;
            u_index[put] = -1; /* terminate column */
            put++;
        }
        iwork1[j] = lu->u_begin[i];
    }
    lu->u_begin[m] = put;
    for (lu_int k = 0; k < m; k++) { /* fill columns */
        lu_int jpivot = pivotcol[k];
        lu_int i = pivotrow[k];
        for (lu_int pos = lu->w_begin[jpivot]; pos < lu->w_end[jpivot]; pos++) {
            lu_int j = w_index[pos];
            lu_int p = iwork1[j];
            iwork1[j]++;
            ORC_ASSERT(p >= 1);
            u_index[p] = i;
            u_value[p] = w_value[pos];
        }
    }

    /* ---- Build pivot sequence (:388-419) ---- */
    /* Build row-column mappings, overwriting pinv, qinv. */
    for (lu_int k = 0; k < m; k++) {
        lu_int i = pivotrow[k];
        lu_int j = pivotcol[k];
        PMAP(lu)[j] = i;
        QMAP(lu)[i] = j;
    }

    /* Build pivots by row index. */
    double max_pivot = 0.0;
    double min_pivot = INFINITY;
    for (lu_int i = 0; i < m; i++) {
        lu->row_pivot[i] = lu->col_pivot[QMAP(lu)[i]];
        double pivot = fabs(lu->row_pivot[i]);
        max_pivot = fmax(pivot, max_pivot);
        min_pivot = fmin(pivot, min_pivot);
    }

    memcpy(P_(lu), pivotrow, (size_t)m * sizeof(lu_int));

    lu->min_pivot = min_pivot;
    lu->max_pivot = max_pivot;
    lu->pivotlen = m;
    lu->l_nz = l_nz;
    lu->u_nz = u_nz;
    lu->r_nz = 0;
    return ORC_OK;
}

/* get_factors -- get_factors.rs:48-180 */
int orc_get_factors(orc_lu *lu, lu_int *rowperm, lu_int *colperm,
                    lu_int *l_colptr, lu_int *l_rowidx, double *l_value_,
                    lu_int *u_colptr, lu_int *u_rowidx, double *u_value_)
{
    /* (:59) `lu.nupdate.unwrap() != 0`: unwrap of None panics in the
     * reference; here an invalid call is reported instead of aborting. */
    if (lu->nupdate != 0) return ORC_ERROR_INVALID_CALL;
    const lu_int m = lu->m;

    if (rowperm) memcpy(rowperm, PIVOTROW(lu), (size_t)m * sizeof(lu_int));
    if (colperm) memcpy(colperm, PIVOTCOL(lu), (size_t)m * sizeof(lu_int));

    if (l_colptr && l_rowidx && l_value_) {
        const lu_int *lt_begin_p = LT_BEGIN_P(lu);
        const lu_int *l_index = lu->l_index;
        const double *l_value = lu->l_value;
        const lu_int *p = P_(lu);
        lu_int *colptr = IWORK1(lu); /* size m workspace */

        /* L[:,k] will hold the elimination factors from the k-th pivot step. (:86-99) */
        lu_int put = 0;
        for (lu_int k = 0; k < m; k++) {
            l_colptr[k] = put;
            l_rowidx[put] = k;
            l_value_[put] = 1.0;
            put++;
            colptr[p[k]] = put; /* next free position in column */
            put += lu->l_begin_p[k + 1] - lu->l_begin_p[k] - 1;
            /* subtract 1 because internal storage uses (-1) terminators */
        }
        l_colptr[m] = put;
        ORC_ASSERT(put == lu->l_nz + m);

        for (lu_int k = 0; k < m; k++) {
            lu_int pos = lt_begin_p[k];
            while (l_index[pos] >= 0) {
                lu_int i = l_index[pos];
                put = colptr[i];
                colptr[i]++;
                l_rowidx[put] = k;
                l_value_[put] = l_value[pos];
                pos++;
            }
        }
    }

    if (u_colptr && u_rowidx && u_value_) {
        const lu_int *w_index = lu->w_index;
        const double *w_value = lu->w_value;
        const lu_int *pivotcol = PIVOTCOL(lu);
        lu_int *colptr = IWORK1(lu);

        /* U[:,k] will hold the column of B from the k-th pivot step. (:136-167) */
        memset(colptr, 0, (size_t)(2 * m + 2) * sizeof(lu_int)); /* colptr.fill(0): whole iwork1 */
        for (lu_int j = 0; j < m; j++)
            for (lu_int pos = lu->w_begin[j]; pos < lu->w_end[j]; pos++) colptr[w_index[pos]]++;
        lu_int put = 0;
        for (lu_int k = 0; k < m; k++) { /* set column pointers */
            lu_int j = pivotcol[k];
            u_colptr[k] = put;
            put += colptr[j];
            colptr[j] = u_colptr[k]; /* next free position in column */
            u_rowidx[put] = k;
            u_value_[put] = lu->col_pivot[j];
            put++;
        }
        u_colptr[m] = put;
        ORC_ASSERT(put == lu->u_nz + m);
        for (lu_int k = 0; k < m; k++) { /* scatter row k */
            lu_int j = pivotcol[k];
            for (lu_int pos = lu->w_begin[j]; pos < lu->w_end[j]; pos++) {
                put = colptr[w_index[pos]];
                colptr[w_index[pos]]++;
                u_rowidx[put] = k;
                u_value_[put] = w_value[pos];
            }
        }
    }
    return ORC_OK;
}
