/* orc_singletons.c -- CPU oracle (test infrastructure).
 * Follows src/lu/singletons.rs and src/lu/setup_bump.rs of /root/reference. */
#include "orc_internal.h"

/* singleton_cols -- singletons.rs:287-396.
 * D1: the Rust loop is `for front in 0..tail` -- the range is evaluated ONCE,
 * so columns queued during the loop (tail += 1 at :375) are never visited:
 * singleton elimination does not cascade. */
static lu_int singleton_cols(lu_int m, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i,
                             const lu_int *b_tp, const lu_int *b_ti, const double *b_tx,
                             lu_int *u_p, lu_int *u_i, double *u_x,
                             lu_int *l_p, lu_int *l_i,
                             double *col_pivot, lu_int *pinv, lu_int *qinv,
                             lu_int *iset, lu_int *queue, lu_int rank, double abstol)
{
    lu_int rk = rank;

    /* Build index sets and initialize queue. (:315-330) */
    lu_int tail = 0;
    for (lu_int j = 0; j < m; j++) {
        if (qinv[j] < 0) {
            lu_int nz = (lu_int)(b_end[j] - b_begin[j]);
            uint64_t i = 0;
            for (uint64_t pos = b_begin[j]; pos < b_end[j]; pos++) i ^= b_i[pos];
            iset[j] = (lu_int)i;
            qinv[j] = -nz - 1; /* use as nonzero counter */
            if (nz == 1) queue[tail++] = j;
        }
    }

    /* Eliminate singleton columns. (:333-384) */
    lu_int put = u_p[rank];
    const lu_int tail0 = tail; /* D1: range fixed at loop entry */
    for (lu_int front = 0; front < tail0; front++) {
        lu_int j = queue[front];
        ORC_ASSERT(qinv[j] == -2 || qinv[j] == -1);
        if (qinv[j] == -1) continue; /* empty column in active submatrix */
        lu_int i = iset[j];
        ORC_ASSERT(i >= 0 && i < m);
        ORC_ASSERT(pinv[i] < 0);
        lu_int end = b_tp[i + 1];

        lu_int pos = b_tp[i];
        while (b_ti[pos] != j) { /* find pivot */
            ORC_ASSERT(pos < end - 1);
            pos++;
        }
        double piv = b_tx[pos];
        if (piv == 0.0 || fabs(piv) < abstol) continue; /* skip singularity */

        /* Eliminate pivot. */
        qinv[j] = rank;
        pinv[i] = rank;
        for (pos = b_tp[i]; pos < end; pos++) {
            lu_int j2 = b_ti[pos];
            if (qinv[j2] < 0) {
                u_i[put] = j2;
                u_x[put] = b_tx[pos];
                put++;
                iset[j2] ^= i; /* remove i from set j2 */
                qinv[j2] += 1;
                if (qinv[j2] == -2) {
                    queue[tail] = j2; /* new singleton (never visited: D1) */
                    tail++;
                }
            }
        }
        u_p[rank + 1] = put;
        col_pivot[j] = piv;
        rank++;
    }

    /* Put empty columns into L. (:387-394) */
    lu_int pos = l_p[rk];
    while (rk < rank) {
        l_i[pos] = -1;
        pos++;
        l_p[rk + 1] = pos;
        rk++;
    }
    return rank;
}

/* singleton_rows -- singletons.rs:398-503 (D1 at :445) */
static lu_int singleton_rows(lu_int m, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i,
                             const double *b_x, const lu_int *b_tp, const lu_int *b_ti,
                             lu_int *u_p, lu_int *l_p, lu_int *l_i, double *l_x,
                             double *col_pivot, lu_int *pinv, lu_int *qinv,
                             lu_int *iset, lu_int *queue, lu_int rank, double abstol)
{
    lu_int rk = rank;

    /* Build index sets and initialize queue. (:427-441) */
    lu_int tail = 0;
    for (lu_int i = 0; i < m; i++) {
        if (pinv[i] < 0) {
            lu_int nz = b_tp[i + 1] - b_tp[i];
            lu_int j = 0;
            for (lu_int pos = b_tp[i]; pos < b_tp[i + 1]; pos++) j ^= b_ti[pos];
            iset[i] = j;
            pinv[i] = -nz - 1; /* use as nonzero counter */
            if (nz == 1) queue[tail++] = i;
        }
    }

    /* Eliminate singleton rows. (:444-492) */
    lu_int put = l_p[rank];
    const lu_int tail0 = tail; /* D1 */
    for (lu_int front = 0; front < tail0; front++) {
        lu_int i = queue[front];
        ORC_ASSERT(pinv[i] == -2 || pinv[i] == -1);
        if (pinv[i] == -1) continue;
        lu_int j = iset[i];
        ORC_ASSERT(j >= 0 && j < m);
        ORC_ASSERT(qinv[j] < 0);
        lu_int end = (lu_int)b_end[j];

        lu_int pos = (lu_int)b_begin[j];
        while ((lu_int)b_i[pos] != i) { /* find pivot */
            ORC_ASSERT(pos < end - 1);
            pos++;
        }
        double piv = b_x[pos];
        if (piv == 0.0 || fabs(piv) < abstol) continue; /* skip singularity */

        /* Eliminate pivot. */
        qinv[j] = rank;
        pinv[i] = rank;
        for (pos = (lu_int)b_begin[j]; pos < end; pos++) {
            lu_int i2 = (lu_int)b_i[pos];
            if (pinv[i2] < 0) {
                l_i[put] = i2;
                l_x[put] = b_x[pos] / piv;
                put++;
                iset[i2] ^= j; /* remove j from set i2 */
                pinv[i2] += 1;
                if (pinv[i2] == -2) {
                    queue[tail] = i2; /* new singleton (never visited: D1) */
                    tail++;
                }
            }
        }
        l_i[put] = -1; /* terminate column */
        put++;
        l_p[rank + 1] = put;
        col_pivot[j] = piv;
        rank++;
    }

    /* Put empty rows into U. (:495-500) */
    lu_int pos = u_p[rk];
    while (rk < rank) {
        u_p[rk + 1] = pos;
        rk++;
    }
    return rank;
}

/* singletons -- singletons.rs:81-264 */
int orc_singletons(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x)
{
    const lu_int m = lu->m;
    const lu_int l_mem = lu->l_mem, u_mem = lu->u_mem, w_mem = lu->w_mem;
    const double abstol = lu->abstol;
    lu_int *pinv = lu->pinv, *qinv = lu->qinv;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *iwork1 = IWORK1(lu);
    lu_int *iwork2 = iwork1 + m; /* split_at_mut(m), :105 */
    lu_int *b_tp = lu->w_begin;  /* build B rowwise in W */
    lu_int *b_ti = lu->w_index;
    double *b_tx = lu->w_value;

    double tic = orc_now();

    /* Check pointers and count nnz(B). (:119-133) */
    lu_int b_nz = 0;
    int ok = 1;
    for (lu_int j = 0; j < m && ok; j++) {
        if (b_end[j] < b_begin[j])
            ok = 0;
        else
            b_nz += (lu_int)(b_end[j] - b_begin[j]);
    }
    if (!ok) return ORC_ERROR_INVALID_ARGUMENT;

    /* Check if sufficient memory in L, U, W. (:135-150) */
    ok = 1;
    if (l_mem < b_nz) { lu->addmem_l = b_nz - l_mem; ok = 0; }
    if (u_mem < b_nz) { lu->addmem_u = b_nz - u_mem; ok = 0; }
    if (w_mem < b_nz) { lu->addmem_w = b_nz - w_mem; ok = 0; }
    if (!ok) return ORC_REALLOCATE;

    /* Count nz per row, check indices. (:152-173) */
    memset(iwork1, 0, (size_t)m * sizeof(lu_int));
    ok = 1;
    for (lu_int j = 0; j < m && ok; j++) {
        for (uint64_t pos = b_begin[j]; pos < b_end[j] && ok; pos++) {
            uint64_t i = b_i[pos];
            if (i >= (uint64_t)m)
                ok = 0;
            else
                iwork1[i]++;
        }
    }
    if (!ok) return ORC_ERROR_INVALID_ARGUMENT;

    /* Pack matrix rowwise, check for duplicates. (:175-201) */
    lu_int put = 0;
    for (lu_int i = 0; i < m; i++) {
        b_tp[i] = put;
        put += iwork1[i];
        iwork1[i] = b_tp[i];
    }
    b_tp[m] = put;
    ORC_ASSERT(put == b_nz);
    ok = 1;
    for (lu_int j = 0; j < m; j++) {
        for (uint64_t pos = b_begin[j]; pos < b_end[j]; pos++) {
            lu_int i = (lu_int)b_i[pos];
            put = iwork1[i];
            iwork1[i]++;
            b_ti[put] = j;
            b_tx[put] = b_x[pos];
            if (put > b_tp[i] && b_ti[put - 1] == j) ok = 0;
        }
    }
    if (!ok) return ORC_ERROR_INVALID_ARGUMENT;

    /* No pivot rows or pivot columns so far. (:205-211) */
    for (lu_int i = 0; i < m; i++) pinv[i] = -1;
    for (lu_int j = 0; j < m; j++) qinv[j] = -1;

    lu_int rank;
    if (lu->nzbias >= 0) { /* put more in U (:213-229) */
        l_begin_p[0] = 0;
        u_begin[0] = 0;
        rank = 0;
        rank = singleton_cols(m, b_begin, b_end, b_i, b_tp, b_ti, b_tx, u_begin, lu->u_index, lu->u_value,
                              l_begin_p, lu->l_index, lu->col_pivot, pinv, qinv, iwork1, iwork2, rank, abstol);
        rank = singleton_rows(m, b_begin, b_end, b_i, b_x, b_tp, b_ti, u_begin, l_begin_p, lu->l_index,
                              lu->l_value, lu->col_pivot, pinv, qinv, iwork1, iwork2, rank, abstol);
    } else { /* put more in L (:230-246) */
        l_begin_p[0] = 0;
        u_begin[0] = 0;
        rank = 0;
        rank = singleton_rows(m, b_begin, b_end, b_i, b_x, b_tp, b_ti, u_begin, l_begin_p, lu->l_index,
                              lu->l_value, lu->col_pivot, pinv, qinv, iwork1, iwork2, rank, abstol);
        rank = singleton_cols(m, b_begin, b_end, b_i, b_tp, b_ti, b_tx, u_begin, lu->u_index, lu->u_value,
                              l_begin_p, lu->l_index, lu->col_pivot, pinv, qinv, iwork1, iwork2, rank, abstol);
    }

    /* pinv, qinv were used as nonzero counters. Reset to -1 if not pivoted. (:248-258) */
    for (lu_int i = 0; i < m; i++)
        if (pinv[i] < 0) pinv[i] = -1;
    for (lu_int j = 0; j < m; j++)
        if (qinv[j] < 0) qinv[j] = -1;

    lu->matrix_nz = b_nz;
    lu->rank = rank;
    lu->time_singletons = orc_now() - tic;
    return ORC_OK;
}

/* setup_bump -- setup_bump.rs:55-264 */
int orc_setup_bump(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const lu_int w_mem = lu->w_mem;
    const lu_int b_nz = lu->matrix_nz;
    const lu_int l_nz = lu->l_begin_p[rank] - rank;
    const lu_int u_nz = lu->u_begin[rank];
    const double abstol = lu->abstol;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    const lu_int *pinv = lu->pinv, *qinv = lu->qinv;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *w_begin2 = w_begin + m, *w_end2 = w_end + m; /* row file */
    lu_int *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    lu_int *w_index = lu->w_index;
    double *w_value = lu->w_value;
    double *colmax = lu->col_pivot;
    lu_int *iwork0 = lu->iwork0;

    lu_int bump_nz = b_nz - l_nz - u_nz - rank; /* will change if columns are dropped */
    lu_int min_rownz = 0, min_colnz = 0;
    ORC_ASSERT(l_nz >= 0 && u_nz >= 0 && bump_nz >= 0); /* usize arithmetic in the reference */

    /* Calculate memory and reallocate. (:105-113) */
    lu_int need = bump_nz + orc_trunc(stretch * (double)bump_nz) + (m - rank) * pad;
    need = 2 * need; /* rowwise + columnwise */
    if (need > w_mem) {
        lu->addmem_w = need - w_mem;
        return ORC_REALLOCATE;
    }

    orc_file_empty(2 * m, w_begin, w_end, w_flink, w_blink, w_mem);

    /* Build columnwise storage. Build row counts in iwork0. (:123-186) */
    orc_list_init(colcount_flink, colcount_blink, m, m + 2, &min_colnz);
    lu_int put = 0;
    for (lu_int j = 0; j < m; j++) {
        if (qinv[j] >= 0) continue;
        lu_int cnz = 0;   /* count nz per column */
        double cmx = 0.0; /* find column maximum */
        for (uint64_t pos = b_begin[j]; pos < b_end[j]; pos++) {
            lu_int i = (lu_int)b_i[pos];
            if (pinv[i] >= 0) continue;
            cmx = fmax(cmx, fabs(b_x[pos]));
            cnz++;
        }
        if (cmx == 0.0 || cmx < abstol) {
            /* Leave column of active submatrix empty. */
            colmax[j] = 0.0;
            orc_list_add(j, 0, colcount_flink, colcount_blink, m, &min_colnz);
            bump_nz -= cnz;
        } else {
            /* Copy column into active submatrix. */
            colmax[j] = cmx;
            orc_list_add(j, cnz, colcount_flink, colcount_blink, m, &min_colnz);
            w_begin[j] = put;
            for (uint64_t pos = b_begin[j]; pos < b_end[j]; pos++) {
                lu_int i = (lu_int)b_i[pos];
                if (pinv[i] >= 0) continue;
                w_index[put] = i;
                w_value[put] = b_x[pos];
                put++;
                iwork0[i]++;
            }
            w_end[j] = put;
            put += orc_trunc(stretch * (double)cnz) + pad;
            /* reappend line to list end */
            orc_list_move(j, 0, w_flink, w_blink, 2 * m, NULL);
        }
    }

    /* Build rowwise storage (pattern only). (:188-224) */
    orc_list_init(rowcount_flink, rowcount_blink, m, m + 2, &min_rownz);
    for (lu_int i = 0; i < m; i++) {
        if (pinv[i] >= 0) continue;
        lu_int rnz = iwork0[i];
        iwork0[i] = 0;
        orc_list_add(i, rnz, rowcount_flink, rowcount_blink, m, &min_rownz);
        w_begin2[i] = put;
        w_end2[i] = put;
        put += rnz;
        /* reappend line to list end */
        orc_list_move(m + i, 0, w_flink, w_blink, 2 * m, NULL);
        put += orc_trunc(stretch * (double)rnz) + pad;
    }
    for (lu_int j = 0; j < m; j++) { /* fill rows */
        for (lu_int pos = w_begin[j]; pos < w_end[j]; pos++) {
            lu_int i = w_index[pos];
            w_index[w_end2[i]] = j;
            w_end2[i]++;
        }
    }
    w_begin[2 * m] = put; /* set beginning of free space */
    ORC_ASSERT(w_begin[2 * m] <= w_end[2 * m]);

    /* D12: unconditional consistency checks (:228-251) */
    ORC_ASSERT(orc_file_diff(m, w_begin, w_end, w_begin2, w_end2, w_index, NULL) == 0);
    ORC_ASSERT(orc_file_diff(m, w_begin2, w_end2, w_begin, w_end, w_index, NULL) == 0);

    lu->bump_nz = bump_nz;
    lu->bump_size = m - rank;
    lu->min_colnz = min_colnz;
    lu->min_rownz = min_rownz;
    return ORC_OK;
}
