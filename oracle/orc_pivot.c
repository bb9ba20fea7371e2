/* orc_pivot.c -- CPU oracle (test infrastructure).
 * Follows src/lu/pivot.rs of /root/reference, path by path. */
#include "orc_internal.h"

#define MAXROW_SMALL 64 /* pivot.rs:22 */

static int pivot_any(orc_lu *lu);
static int pivot_small(orc_lu *lu);
static int pivot_singleton_row(orc_lu *lu);
static int pivot_singleton_col(orc_lu *lu);
static int pivot_doubleton_col(orc_lu *lu);
static void remove_col(orc_lu *lu, lu_int j);

/* pivot -- pivot.rs:48-112 */
int orc_pivot(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const lu_int l_mem = lu->l_mem, u_mem = lu->u_mem;
    const lu_int pivot_col = lu->pivot_col;
    const lu_int pivot_row = lu->pivot_row;
    const lu_int nz_col = lu->w_end[pivot_col] - lu->w_begin[pivot_col];
    const lu_int nz_row = lu->w_end[m + pivot_row] - lu->w_begin[m + pivot_row];

    double tic = orc_now();
    ORC_ASSERT(nz_row >= 1);
    ORC_ASSERT(nz_col >= 1);

    /* Check if room is available in L and U. (:70-81) */
    lu_int room = l_mem - lu->l_begin_p[rank];
    lu_int need = nz_col; /* # off-diagonals in pivot col + end marker (-1) */
    if (room < need) {
        lu->addmem_l = need - room;
        return ORC_REALLOCATE;
    }
    room = u_mem - lu->u_begin[rank];
    need = nz_row - 1; /* # off-diagonals in pivot row */
    if (room < need) {
        lu->addmem_u = need - room;
        return ORC_REALLOCATE;
    }

    /* Branch out implementation of pivot operation. (:84-94) */
    int status, kind;
    if (nz_row == 1) {
        kind = 0;
        status = pivot_singleton_row(lu);
    } else if (nz_col == 1) {
        kind = 1;
        status = pivot_singleton_col(lu);
    } else if (nz_col == 2) {
        kind = 2;
        status = pivot_doubleton_col(lu);
    } else if (nz_col - 1 <= MAXROW_SMALL) {
        kind = 3;
        status = pivot_small(lu);
    } else {
        kind = 4;
        status = pivot_any(lu);
    }
    if (status == ORC_OK) lu->npivot_kind[kind]++; /* test hook: which path ran */

    /* Remove all entries in columns whose maximum entry has dropped below
     * absolute pivot tolerance. (:98-106) */
    if (status == ORC_OK) {
        for (lu_int pos = lu->u_begin[rank]; pos < lu->u_begin[rank + 1]; pos++) {
            lu_int j = lu->u_index[pos];
            ORC_ASSERT(j != pivot_col);
            if (lu->col_pivot[j] == 0.0 || lu->col_pivot[j] < lu->abstol) remove_col(lu, j);
        }
    }

    /* (:108) added even when status is Reallocate, as in the reference */
    lu->factor_flops += (nz_col - 1) * (nz_row - 1);
    lu->time_elim_pivot += orc_now() - tic;
    return status;
}

/* Shared head of pivot_any / pivot_small: room check in W, move pivot to the
 * front of pivot column and pivot row.  pivot.rs:156-208 == :507-559.
 * Returns ORC_OK or ORC_REALLOCATE; outputs updated cbeg/cend/rbeg/rend. */
static int pivot_prepare(orc_lu *lu, lu_int *cbeg_, lu_int *cend_, lu_int *rbeg_, lu_int *rend_, double *pivot_)
{
    const lu_int m = lu->m;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *w_index = lu->w_index;
    double *w_value = lu->w_value;
    lu_int cbeg = *cbeg_, cend = *cend_, rbeg = *rbeg_, rend = *rend_;
    const lu_int cnz1 = cend - cbeg - 1;
    const lu_int rnz1 = rend - rbeg - 1;

    lu_int grow = 0;
    lu_int where_ = -1;
    for (lu_int pos = cbeg; pos < cend; pos++) {
        lu_int i = w_index[pos];
        if (i == pivot_row) {
            where_ = pos;
        } else {
            lu_int nz = w_end[m + i] - w_begin[m + i];
            grow += nz + rnz1 + orc_trunc(stretch * (double)(nz + rnz1)) + pad;
        }
    }
    ORC_ASSERT(where_ >= 0);
    orc_iswap(w_index, cbeg, where_);
    orc_fswap(w_value, cbeg, where_);
    double pivot = w_value[cbeg];
    ORC_ASSERT(pivot != 0.0);
    where_ = -1;
    for (lu_int rpos = rbeg; rpos < rend; rpos++) {
        lu_int j = w_index[rpos];
        if (j == pivot_col) {
            where_ = rpos;
        } else {
            lu_int nz = w_end[j] - w_begin[j];
            grow += nz + cnz1 + orc_trunc(stretch * (double)(nz + cnz1)) + pad;
        }
    }
    ORC_ASSERT(where_ >= 0);
    orc_iswap(w_index, rbeg, where_);
    lu_int room = w_end[2 * m] - w_begin[2 * m];
    if (grow > room) {
        orc_file_compress(2 * m, w_begin, w_end, lu->w_flink, w_index, w_value, stretch, pad);
        cbeg = w_begin[pivot_col];
        cend = w_end[pivot_col];
        rbeg = w_begin[m + pivot_row];
        rend = w_end[m + pivot_row];
        room = w_end[2 * m] - w_begin[2 * m];
        lu->ngarbage++;
    }
    if (grow > room) {
        lu->addmem_w = grow - room;
        return ORC_REALLOCATE;
    }
    *cbeg_ = cbeg; *cend_ = cend; *rbeg_ = rbeg; *rend_ = rend; *pivot_ = pivot;
    return ORC_OK;
}

/* pivot_any -- pivot.rs:114-458 */
static int pivot_any(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const double droptol = lu->droptol;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    double *colmax = lu->col_pivot;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = lu->iwork0;
    double *work = lu->work0;

    lu_int cbeg = w_begin[pivot_col], cend = w_end[pivot_col];
    lu_int rbeg = w_begin[m + pivot_row], rend = w_end[m + pivot_row];
    const lu_int cnz1 = cend - cbeg - 1; /* nz in pivot column except pivot */
    const lu_int rnz1 = rend - rbeg - 1; /* nz in pivot row except pivot */
    double pivot;

    int st = pivot_prepare(lu, &cbeg, &cend, &rbeg, &rend, &pivot);
    if (st != ORC_OK) return st;

    /* get pointer to U (:210-213) */
    lu_int u_put = u_begin[rank];
    ORC_ASSERT(u_put >= 0);
    ORC_ASSERT(u_put < lu->u_mem);

    /* ---- Column file update (:215-331) ---- */
    lu_int position = 1;
    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        lu_int i = w_index[pos];
        marked[i] = position;
        position++;
    }

    for (lu_int rpos = rbeg + 1; rpos < rend; rpos++) {
        lu_int j = w_index[rpos];
        ORC_ASSERT(j != pivot_col);
        double cmx = 0.0; /* column maximum */

        /* Compress unmodified column entries. Store entries to be updated
         * in workspace. Move pivot row entry to the front of column. */
        lu_int where_ = -1;
        lu_int put = w_begin[j];
        lu_int pos1 = w_begin[j];
        for (lu_int pos = pos1; pos < w_end[j]; pos++) {
            lu_int i = w_index[pos];
            position = marked[i];
            if (position > 0) {
                ORC_ASSERT(i != pivot_row);
                work[position] = w_value[pos];
            } else {
                ORC_ASSERT(position == 0);
                double x = fabs(w_value[pos]);
                if (i == pivot_row)
                    where_ = put;
                else if (x > cmx)
                    cmx = x;
                w_index[put] = w_index[pos];
                w_value[put] = w_value[pos];
                put++;
            }
        }
        ORC_ASSERT(where_ >= 0);
        w_end[j] = put;
        orc_iswap(w_index, pos1, where_);
        orc_fswap(w_value, pos1, where_);
        double xrj = w_value[pos1]; /* pivot row entry */

        /* Reappend column if no room for update. */
        lu_int room = w_begin[w_flink[j]] - put;
        if (room < cnz1) {
            lu_int nz = w_end[j] - w_begin[j];
            room = cnz1 + orc_trunc(stretch * (double)(nz + cnz1)) + pad;
            orc_file_reappend(j, 2 * m, w_begin, w_end, w_flink, w_blink, w_index, w_value, room);
            put = w_end[j];
            ORC_ASSERT(w_begin[w_flink[j]] - put == room);
            lu->nexpand++;
        }

        /* Compute update in workspace and append to column. */
        double a = xrj / pivot;
        for (lu_int pos = 1; pos <= cnz1; pos++) work[pos] -= a * w_value[cbeg + pos];
        for (lu_int pos = 1; pos <= cnz1; pos++) {
            w_index[put] = w_index[cbeg + pos];
            w_value[put] = work[pos];
            put++;
            double x = fabs(work[pos]);
            if (x > cmx) cmx = x;
            work[pos] = 0.0;
        }
        w_end[j] = put;

        /* Write pivot row entry to U and remove from file. */
        if (fabs(xrj) > droptol) {
            ORC_ASSERT(u_put < lu->u_mem);
            u_index[u_put] = j;
            u_value[u_put] = xrj;
            u_put++;
        }
        ORC_ASSERT(w_index[w_begin[j]] == pivot_row);
        w_begin[j]++;

        /* Move column to new list and update min_colnz. */
        lu_int nz = w_end[j] - w_begin[j];
        orc_list_move(j, nz, colcount_flink, colcount_blink, m, &lu->min_colnz);

        colmax[j] = cmx;
    }
    for (lu_int pos = cbeg + 1; pos < cend; pos++) marked[w_index[pos]] = 0;

    /* ---- Row file update (:335-401) ---- */
    for (lu_int rpos = rbeg; rpos < rend; rpos++) marked[w_index[rpos]] = 1;
    ORC_ASSERT(marked[pivot_col] == 1);

    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        lu_int i = w_index[pos];
        ORC_ASSERT(i != pivot_row);

        /* Compress unmodified row entries (not marked). Remove
         * overlap with pivot row, including pivot column entry. */
        int found = 0;
        lu_int put = w_begin[m + i];
        for (lu_int rpos = w_begin[m + i]; rpos < w_end[m + i]; rpos++) {
            lu_int j = w_index[rpos];
            if (j == pivot_col) found = 1;
            if (marked[j] == 0) {
                w_index[put] = j;
                put++;
            }
        }
        ORC_ASSERT(found != 0);
        w_end[m + i] = put;

        /* Reappend row if no room for update. Append pattern of pivot row. */
        lu_int room = w_begin[w_flink[m + i]] - put;
        if (room < rnz1) {
            lu_int nz = w_end[m + i] - w_begin[m + i];
            room = rnz1 + orc_trunc(stretch * (double)(nz + rnz1)) + pad;
            orc_file_reappend(m + i, 2 * m, w_begin, w_end, w_flink, w_blink, w_index, w_value, room);
            put = w_end[m + i];
            ORC_ASSERT(w_begin[w_flink[m + i]] - put == room);
            lu->nexpand++;
        }
        for (lu_int rpos = rbeg + 1; rpos < rend; rpos++) {
            w_index[put] = w_index[rpos];
            put++;
        }
        w_end[m + i] = put;

        /* Move to new list. The row must be reinserted even if nz are
         * unchanged since it might have been taken out in Markowitz search. */
        lu_int nz = w_end[m + i] - w_begin[m + i];
        orc_list_move(i, nz, rowcount_flink, rowcount_blink, m, &lu->min_rownz);
    }
    for (lu_int rpos = rbeg; rpos < rend; rpos++) marked[w_index[rpos]] = 0;

    /* ---- Store column in L (:404-416) ---- */
    lu_int put = l_begin_p[rank];
    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        double x = w_value[pos] / pivot;
        if (fabs(x) > droptol) {
            l_index[put] = w_index[pos];
            l_value[put] = x;
            put++;
        }
    }
    l_index[put] = -1; /* terminate column */
    put++;
    l_begin_p[rank + 1] = put;
    u_begin[rank + 1] = u_put;

    /* ---- Cleanup (:418-426) ---- */
    colmax[pivot_col] = pivot;
    w_end[pivot_col] = cbeg;
    w_end[m + pivot_row] = rbeg;
    orc_list_remove(colcount_flink, colcount_blink, pivot_col);
    orc_list_remove(rowcount_flink, rowcount_blink, pivot_row);
    return ORC_OK;
}

/* pivot_small -- pivot.rs:460-833 */
static int pivot_small(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const double droptol = lu->droptol;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    double *colmax = lu->col_pivot;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = lu->iwork0;
    double *work = lu->work0;
    double *cancelled = lu->row_pivot; /* f64 VALUES holding the masks (:488) */
    int64_t *cancelled64 = (int64_t *)lu->work1; /* fix_d3 only: exact 64-bit masks (work1 is free here) */

    lu_int cbeg = w_begin[pivot_col], cend = w_end[pivot_col];
    lu_int rbeg = w_begin[m + pivot_row], rend = w_end[m + pivot_row];
    const lu_int cnz1 = cend - cbeg - 1;
    const lu_int rnz1 = rend - rbeg - 1;
    double pivot;

    ORC_ASSERT(cnz1 <= MAXROW_SMALL);

    int st = pivot_prepare(lu, &cbeg, &cend, &rbeg, &rend, &pivot);
    if (st != ORC_OK) return st;

    lu_int u_put = u_begin[rank];
    ORC_ASSERT(u_put >= 0);
    ORC_ASSERT(u_put < lu->u_mem);

    /* ---- Column file update (:566-691) ---- */
    lu_int position = 1;
    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        lu_int i = w_index[pos];
        marked[i] = position;
        position++;
    }

    lu_int col_number = 0; /* mask cancelled[col_number] */
    for (lu_int rpos = rbeg + 1; rpos < rend; rpos++) {
        lu_int j = w_index[rpos];
        ORC_ASSERT(j != pivot_col);
        double cmx = 0.0;

        lu_int where_ = -1;
        lu_int put = w_begin[j];
        lu_int pos1 = w_begin[j];
        for (lu_int pos = pos1; pos < w_end[j]; pos++) {
            lu_int i = w_index[pos];
            position = marked[i];
            if (position > 0) {
                ORC_ASSERT(i != pivot_row);
                work[position] = w_value[pos];
            } else {
                ORC_ASSERT(position == 0);
                double x = fabs(w_value[pos]);
                if (i == pivot_row)
                    where_ = put;
                else if (x > cmx)
                    cmx = x;
                w_index[put] = w_index[pos];
                w_value[put] = w_value[pos];
                put++;
            }
        }
        ORC_ASSERT(where_ >= 0);
        w_end[j] = put;
        orc_iswap(w_index, pos1, where_);
        orc_fswap(w_value, pos1, where_);
        double xrj = w_value[pos1];

        lu_int room = w_begin[w_flink[j]] - put;
        if (room < cnz1) {
            lu_int nz = w_end[j] - w_begin[j];
            room = cnz1 + orc_trunc(stretch * (double)(nz + cnz1)) + pad;
            orc_file_reappend(j, 2 * m, w_begin, w_end, w_flink, w_blink, w_index, w_value, room);
            put = w_end[j];
            ORC_ASSERT(w_begin[w_flink[j]] - put == room);
            lu->nexpand++;
        }

        double a = xrj / pivot;
        for (lu_int pos = 1; pos <= cnz1; pos++) work[pos] -= a * w_value[cbeg + pos];

        /* D3: `let mut mask = 0; mask |= 1 << (pos - 1) as i64;` -- the
         * literal types default to i32.  Release-mode Rust masks the shift
         * count to 5 bits (wrapping_shl); debug mode would panic for
         * pos-1 >= 32.  We restate the release behaviour: i32 mask, shift
         * count & 31.  `mask as f64` then `as i64` sign-extends. */
        int32_t mask = 0;
        int64_t mask64 = 0; /* BASICLU's intended mask (used only when fix_d3 is set) */
        for (lu_int pos = 1; pos <= cnz1; pos++) {
            double x = fabs(work[pos]);
            if (x > droptol) {
                w_index[put] = w_index[cbeg + pos];
                w_value[put] = work[pos];
                put++;
                if (x > cmx) cmx = x;
            } else {
                /* cancellation in row w_index[cbeg+pos] */
                mask |= (int32_t)((uint32_t)1 << (unsigned)((pos - 1) & 31));
                mask64 |= (int64_t)((uint64_t)1 << (unsigned)(pos - 1));
                if (pos - 1 >= 31) lu->d3_hits++;
            }
            work[pos] = 0.0;
        }
        w_end[j] = put;
        cancelled[col_number] = (double)mask;
        if (lu->fix_d3) cancelled64[col_number] = mask64;

        if (fabs(xrj) > droptol) {
            ORC_ASSERT(u_put < lu->u_mem);
            u_index[u_put] = j;
            u_value[u_put] = xrj;
            u_put++;
        }
        ORC_ASSERT(w_index[w_begin[j]] == pivot_row);
        w_begin[j]++;

        lu_int nz = w_end[j] - w_begin[j];
        orc_list_move(j, nz, colcount_flink, colcount_blink, m, &lu->min_colnz);

        colmax[j] = cmx;
        col_number++;
    }
    for (lu_int pos = cbeg + 1; pos < cend; pos++) marked[w_index[pos]] = 0;

    /* ---- Row file update (:695-775) ---- */
    for (lu_int rpos = rbeg; rpos < rend; rpos++) marked[w_index[rpos]] = 1;
    ORC_ASSERT(marked[pivot_col] == 1);

    int64_t rmask = 1; /* `let mut mask = 1;` compared against `as i64`: i64 */
    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        ORC_ASSERT(rmask != 0);
        lu_int i = w_index[pos];
        ORC_ASSERT(i != pivot_row);

        int found = 0;
        lu_int put = w_begin[m + i];
        for (lu_int rpos = w_begin[m + i]; rpos < w_end[m + i]; rpos++) {
            lu_int j = w_index[rpos];
            if (j == pivot_col) found = 1;
            if (marked[j] == 0) {
                w_index[put] = j;
                put++;
            }
        }
        ORC_ASSERT(found != 0);
        w_end[m + i] = put;

        lu_int room = w_begin[w_flink[m + i]] - put;
        if (room < rnz1) {
            lu_int nz = w_end[m + i] - w_begin[m + i];
            room = rnz1 + orc_trunc(stretch * (double)(nz + rnz1)) + pad;
            orc_file_reappend(m + i, 2 * m, w_begin, w_end, w_flink, w_blink, w_index, w_value, room);
            put = w_end[m + i];
            ORC_ASSERT(w_begin[w_flink[m + i]] - put == room);
            lu->nexpand++;
        }

        col_number = 0;
        for (lu_int rpos = rbeg + 1; rpos < rend; rpos++) {
            if (((lu->fix_d3 ? cancelled64[col_number] : (int64_t)cancelled[col_number]) & rmask) == 0) {
                w_index[put] = w_index[rpos];
                put++;
            }
            col_number++;
        }
        w_end[m + i] = put;

        lu_int nz = w_end[m + i] - w_begin[m + i];
        orc_list_move(i, nz, rowcount_flink, rowcount_blink, m, &lu->min_rownz);

        /* `mask <<= 1` on i64: for cnz1 == 64 the last shift (of bit 63) is
         * an overflow-free shl (shift count 1), giving 0 after the loop. */
        rmask = (int64_t)((uint64_t)rmask << 1);
    }
    for (lu_int rpos = rbeg; rpos < rend; rpos++) marked[w_index[rpos]] = 0;

    /* ---- Store column in L (:778-790) ---- */
    lu_int put = l_begin_p[rank];
    for (lu_int pos = cbeg + 1; pos < cend; pos++) {
        double x = w_value[pos] / pivot;
        if (fabs(x) > droptol) {
            l_index[put] = w_index[pos];
            l_value[put] = x;
            put++;
        }
    }
    l_index[put] = -1;
    put++;
    l_begin_p[rank + 1] = put;
    u_begin[rank + 1] = u_put;

    /* ---- Cleanup (:792-800) ---- */
    colmax[pivot_col] = pivot;
    w_end[pivot_col] = cbeg;
    w_end[m + pivot_row] = rbeg;
    orc_list_remove(colcount_flink, colcount_blink, pivot_col);
    orc_list_remove(rowcount_flink, rowcount_blink, pivot_row);
    return ORC_OK;
}

/* pivot_singleton_row -- pivot.rs:835-926 */
static int pivot_singleton_row(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const double droptol = lu->droptol;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    double *colmax = lu->col_pivot;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *l_index = lu->l_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *w_value = lu->w_value;

    const lu_int cbeg = w_begin[pivot_col], cend = w_end[pivot_col];
    const lu_int rbeg = w_begin[m + pivot_row], rend = w_end[m + pivot_row];
    const lu_int rnz1 = rend - rbeg - 1;
    ORC_ASSERT(rnz1 == 0);

    /* Find pivot. */
    lu_int where_ = cbeg;
    while (w_index[where_] != pivot_row) {
        ORC_ASSERT(where_ < cend - 1);
        where_++;
    }
    double pivot = w_value[where_];
    ORC_ASSERT(pivot != 0.0);

    /* Store column in L. */
    lu_int put = l_begin_p[rank];
    for (lu_int pos = cbeg; pos < cend; pos++) {
        double x = w_value[pos] / pivot;
        if (pos != where_ && fabs(x) > droptol) {
            l_index[put] = w_index[pos];
            l_value[put] = x;
            put++;
        }
    }
    l_index[put] = -1;
    put++;
    l_begin_p[rank + 1] = put;
    u_begin[rank + 1] = u_begin[rank];

    /* Remove pivot column from row file. Update row lists. */
    for (lu_int pos = cbeg; pos < cend; pos++) {
        lu_int i = w_index[pos];
        if (i == pivot_row) continue;
        where_ = w_begin[m + i];
        while (w_index[where_] != pivot_col) {
            ORC_ASSERT(where_ < w_end[m + i] - 1);
            where_++;
        }
        w_end[m + i]--;
        w_index[where_] = w_index[w_end[m + i]];
        lu_int nz = w_end[m + i] - w_begin[m + i];
        orc_list_move(i, nz, rowcount_flink, rowcount_blink, m, &lu->min_rownz);
    }

    colmax[pivot_col] = pivot;
    w_end[pivot_col] = cbeg;
    w_end[m + pivot_row] = rbeg;
    orc_list_remove(colcount_flink, colcount_blink, pivot_col);
    orc_list_remove(rowcount_flink, rowcount_blink, pivot_row);
    return ORC_OK;
}

/* pivot_singleton_col -- pivot.rs:928-1025 */
static int pivot_singleton_col(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const double droptol = lu->droptol;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    double *colmax = lu->col_pivot;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *u_value = lu->u_value, *w_value = lu->w_value;

    const lu_int cbeg = w_begin[pivot_col], cend = w_end[pivot_col];
    const lu_int rbeg = w_begin[m + pivot_row], rend = w_end[m + pivot_row];
    const lu_int cnz1 = cend - cbeg - 1;
    ORC_ASSERT(cnz1 == 0);

    /* Remove pivot row from column file and store in U. Update column lists. */
    lu_int put = u_begin[rank];
    double pivot = w_value[cbeg];
    ORC_ASSERT(pivot != 0.0);
    int found = 0;
    double xrj = 0.0;
    for (lu_int rpos = rbeg; rpos < rend; rpos++) {
        lu_int j = w_index[rpos];
        if (j == pivot_col) {
            found = 1;
            continue;
        }
        lu_int where_ = -1;
        double cmx = 0.0;
        for (lu_int pos = w_begin[j]; pos < w_end[j]; pos++) {
            double x = fabs(w_value[pos]);
            if (w_index[pos] == pivot_row) {
                where_ = pos;
                xrj = w_value[pos];
            } else if (x > cmx) {
                cmx = x;
            }
        }
        ORC_ASSERT(where_ >= 0);
        if (fabs(xrj) > droptol) {
            u_index[put] = j;
            u_value[put] = xrj;
            put++;
        }
        w_end[j]--;
        w_index[where_] = w_index[w_end[j]];
        w_value[where_] = w_value[w_end[j]];
        lu_int nz = w_end[j] - w_begin[j];
        orc_list_move(j, nz, colcount_flink, colcount_blink, m, &lu->min_colnz);
        colmax[j] = cmx;
    }
    ORC_ASSERT(found != 0);
    u_begin[rank + 1] = put;

    /* Store empty column in L. */
    put = l_begin_p[rank];
    l_index[put] = -1;
    put++;
    l_begin_p[rank + 1] = put;

    colmax[pivot_col] = pivot;
    w_end[pivot_col] = cbeg;
    w_end[m + pivot_row] = rbeg;
    orc_list_remove(colcount_flink, colcount_blink, pivot_col);
    orc_list_remove(rowcount_flink, rowcount_blink, pivot_row);
    return ORC_OK;
}

/* pivot_doubleton_col -- pivot.rs:1027-1331 */
static int pivot_doubleton_col(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int rank = lu->rank;
    const double droptol = lu->droptol;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    const lu_int pivot_col = lu->pivot_col, pivot_row = lu->pivot_row;
    lu_int *colcount_flink = lu->colcount_flink, *colcount_blink = lu->colcount_blink;
    lu_int *rowcount_flink = lu->rowcount_flink, *rowcount_blink = lu->rowcount_blink;
    double *colmax = lu->col_pivot;
    lu_int *l_begin_p = lu->l_begin_p, *u_begin = lu->u_begin;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    lu_int *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = lu->iwork0;

    lu_int cbeg = w_begin[pivot_col];
    const lu_int cend = w_end[pivot_col];
    lu_int rbeg = w_begin[m + pivot_row], rend = w_end[m + pivot_row];
    const lu_int cnz1 = cend - cbeg - 1;
    const lu_int rnz1 = rend - rbeg - 1;
    ORC_ASSERT(cnz1 == 1);

    /* Move pivot element to front of pivot column and pivot row. */
    if (w_index[cbeg] != pivot_row) {
        orc_iswap(w_index, cbeg, cbeg + 1);
        orc_fswap(w_value, cbeg, cbeg + 1);
    }
    ORC_ASSERT(w_index[cbeg] == pivot_row);
    const double pivot = w_value[cbeg];
    ORC_ASSERT(pivot != 0.0);
    const lu_int other_row = w_index[cbeg + 1];
    const double other_value = w_value[cbeg + 1];
    lu_int where_ = rbeg;
    while (w_index[where_] != pivot_col) {
        ORC_ASSERT(where_ < rend - 1);
        where_++;
    }
    orc_iswap(w_index, rbeg, where_);

    /* Check if room is available in W. (:1088-1113) */
    lu_int nz = w_end[m + other_row] - w_begin[m + other_row];
    lu_int grow = nz + rnz1 + orc_trunc(stretch * (double)(nz + rnz1)) + pad;
    lu_int room = w_end[2 * m] - w_begin[2 * m];
    if (grow > room) {
        orc_file_compress(2 * m, w_begin, w_end, w_flink, w_index, w_value, stretch, pad);
        cbeg = w_begin[pivot_col];
        rbeg = w_begin[m + pivot_row];
        rend = w_end[m + pivot_row];
        room = w_end[2 * m] - w_begin[2 * m];
        lu->ngarbage++;
    }
    if (grow > room) {
        lu->addmem_w = grow - room;
        return ORC_REALLOCATE;
    }

    /* ---- Column file update (:1115-1222) ---- */
    lu_int u_put = u_begin[rank];
    lu_int put = rbeg + 1;
    lu_int ncancelled = 0;
    for (lu_int rpos = rbeg + 1; rpos < rend; rpos++) {
        lu_int j = w_index[rpos];
        ORC_ASSERT(j != pivot_col);
        double cmx = 0.0;

        /* Find position of pivot row entry and possibly other row entry in column j. */
        lu_int where_pivot = -1, where_other = -1;
        lu_int end = w_end[j];
        for (lu_int pos = w_begin[j]; pos < end; pos++) {
            double x = fabs(w_value[pos]);
            if (w_index[pos] == pivot_row)
                where_pivot = pos;
            else if (w_index[pos] == other_row)
                where_other = pos;
            else if (x > cmx)
                cmx = x;
        }
        ORC_ASSERT(where_pivot >= 0);
        double xrj = w_value[where_pivot];

        /* Store pivot row entry in U. */
        if (fabs(w_value[where_pivot]) > droptol) {
            u_index[u_put] = j;
            u_value[u_put] = w_value[where_pivot];
            u_put++;
        }

        if (where_other < 0) {
            /* Compute fill-in element. */
            double x = -xrj * (other_value / pivot);
            double xabs = fabs(x);
            if (xabs > droptol) {
                /* Store fill-in where pivot row entry was. */
                w_index[where_pivot] = other_row;
                w_value[where_pivot] = x;
                w_index[put] = j;
                put++;
                if (xabs > cmx) cmx = xabs;
            } else {
                /* Remove pivot row entry. */
                w_end[j]--;
                end = w_end[j];
                w_index[where_pivot] = w_index[end];
                w_value[where_pivot] = w_value[end];
                /* Decrease column count. */
                nz = end - w_begin[j];
                orc_list_move(j, nz, colcount_flink, colcount_blink, m, &lu->min_colnz);
            }
        } else {
            /* Remove pivot row entry and update other row entry. */
            w_end[j]--;
            end = w_end[j];
            w_index[where_pivot] = w_index[end];
            w_value[where_pivot] = w_value[end];
            if (where_other == end) where_other = where_pivot;
            w_value[where_other] -= xrj * (other_value / pivot);

            /* If we have numerical cancellation, then remove the entry and mark the column. */
            double x = fabs(w_value[where_other]);
            if (x <= droptol) {
                w_end[j]--;
                end = w_end[j];
                w_index[where_other] = w_index[end];
                w_value[where_other] = w_value[end];
                marked[j] = 1;
                ncancelled++;
            } else if (x > cmx) {
                cmx = x;
            }

            /* Decrease column count. */
            nz = w_end[j] - w_begin[j];
            orc_list_move(j, nz, colcount_flink, colcount_blink, m, &lu->min_colnz);
        }
        colmax[j] = cmx;
    }
    rend = put;
    u_begin[rank + 1] = u_put;

    /* ---- Row file update (:1224-1293) ---- */
    if (ncancelled != 0) {
        ORC_ASSERT(marked[pivot_col] == 0);
        marked[pivot_col] = 1; /* treat as cancelled */
        lu_int rput = w_begin[m + other_row]; /* compress remaining entries */
        lu_int end = w_end[m + other_row];
        for (lu_int pos = rput; pos < end; pos++) {
            lu_int j = w_index[pos];
            if (marked[j] != 0) {
                marked[j] = 0;
            } else {
                w_index[rput] = j;
                rput++;
            }
        }
        ORC_ASSERT(end - rput == ncancelled + 1);
        w_end[m + other_row] = rput;
    } else {
        where_ = w_begin[m + other_row];
        while (w_index[where_] != pivot_col) {
            ORC_ASSERT(where_ < w_end[m + other_row] - 1);
            where_++;
        }
        w_end[m + other_row]--;
        lu_int end = w_end[m + other_row];
        w_index[where_] = w_index[end];
    }

    /* Reappend row if no room for update. */
    lu_int nfill = rend - (rbeg + 1);
    room = w_begin[w_flink[m + other_row]] - w_end[m + other_row];
    if (nfill > room) {
        nz = w_end[m + other_row] - w_begin[m + other_row];
        lu_int space = nfill + orc_trunc(stretch * (double)(nz + nfill)) + pad;
        orc_file_reappend(m + other_row, 2 * m, w_begin, w_end, w_flink, w_blink, w_index, w_value, space);
        lu->nexpand++;
    }

    /* Append fill-in to row pattern. */
    put = w_end[m + other_row];
    for (lu_int pos = rbeg + 1; pos < rend; pos++) {
        w_index[put] = w_index[pos];
        put++;
    }
    w_end[m + other_row] = put;

    /* Reinsert other row into row counts. */
    nz = w_end[m + other_row] - w_begin[m + other_row];
    orc_list_move(other_row, nz, rowcount_flink, rowcount_blink, m, &lu->min_rownz);

    /* ---- Store column in L (:1295-1305) ---- */
    put = l_begin_p[rank];
    double x = other_value / pivot;
    if (fabs(x) > droptol) {
        l_index[put] = other_row;
        l_value[put] = x;
        put++;
    }
    l_index[put] = -1;
    put++;
    l_begin_p[rank + 1] = put;

    /* ---- Cleanup (:1307-1315) ---- */
    colmax[pivot_col] = pivot;
    w_end[pivot_col] = cbeg;
    w_end[m + pivot_row] = rbeg;
    orc_list_remove(colcount_flink, colcount_blink, pivot_col);
    orc_list_remove(rowcount_flink, rowcount_blink, pivot_row);
    return ORC_OK;
}

/* remove_col -- pivot.rs:1333-1381 */
static void remove_col(orc_lu *lu, lu_int j)
{
    const lu_int m = lu->m;
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end, *w_index = lu->w_index;
    const lu_int cbeg = w_begin[j], cend = w_end[j];

    /* Remove column j from row file. */
    for (lu_int pos = cbeg; pos < cend; pos++) {
        lu_int i = w_index[pos];
        lu_int where_ = w_begin[m + i];
        while (w_index[where_] != j) {
            ORC_ASSERT(where_ < w_end[m + i] - 1);
            where_++;
        }
        w_end[m + i]--;
        w_index[where_] = w_index[w_end[m + i]];
        lu_int nz = w_end[m + i] - w_begin[m + i];
        orc_list_move(i, nz, lu->rowcount_flink, lu->rowcount_blink, m, &lu->min_rownz);
    }

    /* Remove column j from column file. */
    lu->col_pivot[j] = 0.0;
    w_end[j] = cbeg;
    orc_list_move(j, 0, lu->colcount_flink, lu->colcount_blink, m, &lu->min_colnz);
}
