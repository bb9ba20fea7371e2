/* orc_support.c -- CPU oracle (test infrastructure): LU state, count lists, data file.
 * Follows src/lu/lu.rs, src/lu/list.rs, src/lu/file.rs of /root/reference. */
#include "orc_internal.h"
#include "../include/blu_hip.h" /* key numbering shared with the product ABI */
#include <time.h>

double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* LU::new -- src/lu/lu.rs:243-319                                            */
/* ------------------------------------------------------------------------- */
static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    return p;
}

int orc_lu_init(orc_lu *lu, lu_int m, lu_int b_nz)
{
    memset(lu, 0, sizeof(*lu)); /* ..Default::default() */
    lu->l_mem = b_nz;
    lu->u_mem = b_nz;
    lu->w_mem = b_nz;

    /* default parameters, lu.rs:249-259 */
    lu->droptol = 1e-20;
    lu->abstol = 1e-14;
    lu->reltol = 0.1;
    lu->nzbias = 1; /* Some(1) */
    lu->maxsearch = 3;
    lu->pad = 4;
    lu->stretch = 0.3;
    lu->compress_thres = 0.5;
    lu->sparse_thres = 0.05;
    lu->search_rows = 0; /* D4: doc says 1, code says 0 */

    lu->m = m;

    /* Option fields defaulting to None */
    lu->nupdate = -1;
    lu->pivot_row = -1;
    lu->pivot_col = -1;
    lu->ftran_for_update = -1;
    lu->btran_for_update = -1;
    lu->task = ORC_TASK_SINGLETONS; /* impl Default for Task, def.rs:14-18 */
    lu->stop_after_pivots = -1;

    lu->l_index = xcalloc((size_t)b_nz, sizeof(lu_int));
    lu->u_index = xcalloc((size_t)b_nz, sizeof(lu_int));
    lu->w_index = xcalloc((size_t)b_nz, sizeof(lu_int));
    lu->l_value = xcalloc((size_t)b_nz, sizeof(double));
    lu->u_value = xcalloc((size_t)b_nz, sizeof(double));
    lu->w_value = xcalloc((size_t)b_nz, sizeof(double));

    size_t n2 = (size_t)(2 * m + 2);
    lu->colcount_flink = xcalloc(n2, sizeof(lu_int));
    lu->colcount_blink = xcalloc(n2, sizeof(lu_int));
    lu->rowcount_flink = xcalloc(n2, sizeof(lu_int));
    lu->rowcount_blink = xcalloc(n2, sizeof(lu_int));
    lu->w_begin = xcalloc(n2, sizeof(lu_int));
    lu->w_end = xcalloc(n2, sizeof(lu_int));
    lu->w_flink = xcalloc(n2, sizeof(lu_int));
    lu->w_blink = xcalloc(n2, sizeof(lu_int));
    lu->pinv = xcalloc((size_t)m, sizeof(lu_int));
    lu->qinv = xcalloc((size_t)m, sizeof(lu_int));
    lu->l_begin_p = xcalloc((size_t)m + 1, sizeof(lu_int));
    lu->u_begin = xcalloc((size_t)m + 1, sizeof(lu_int));
    lu->iwork0 = xcalloc((size_t)m, sizeof(lu_int));
    lu->work0 = xcalloc((size_t)m, sizeof(double));
    lu->work1 = xcalloc((size_t)m, sizeof(double));
    lu->col_pivot = xcalloc((size_t)m, sizeof(double));
    lu->row_pivot = xcalloc((size_t)m, sizeof(double));

    if (!lu->l_index || !lu->u_index || !lu->w_index || !lu->l_value || !lu->u_value || !lu->w_value ||
        !lu->colcount_flink || !lu->colcount_blink || !lu->rowcount_flink || !lu->rowcount_blink ||
        !lu->w_begin || !lu->w_end || !lu->w_flink || !lu->w_blink || !lu->pinv || !lu->qinv ||
        !lu->l_begin_p || !lu->u_begin || !lu->iwork0 || !lu->work0 || !lu->work1 || !lu->col_pivot ||
        !lu->row_pivot) {
        orc_lu_destroy(lu);
        return -1;
    }

    /* lu.rs:301-305: marker overflow guard (marker is 0 here) */
    if (lu->marker > INT64_MAX - 4) {
        memset(lu->iwork0, 0, (size_t)m * sizeof(lu_int));
        lu->marker = 0;
    }
    /* lu.rs:309-313: nupdate is None here -> w_end[2m] = w_mem */
    if (lu->nupdate >= 0)
        lu->w_end[m] = lu->w_mem;
    else
        lu->w_end[2 * m] = lu->w_mem;

    orc_lu_reset(lu);
    return 0;
}

void orc_lu_destroy(orc_lu *lu)
{
    free(lu->l_index); free(lu->u_index); free(lu->w_index);
    free(lu->l_value); free(lu->u_value); free(lu->w_value);
    free(lu->colcount_flink); free(lu->colcount_blink);
    free(lu->rowcount_flink); free(lu->rowcount_blink);
    free(lu->w_begin); free(lu->w_end); free(lu->w_flink); free(lu->w_blink);
    free(lu->pinv); free(lu->qinv); free(lu->l_begin_p); free(lu->u_begin);
    free(lu->iwork0); free(lu->work0); free(lu->work1);
    free(lu->col_pivot); free(lu->row_pivot);
    memset(lu, 0, sizeof(*lu));
}

/* LU::reset -- src/lu/lu.rs:329-396 */
void orc_lu_reset(orc_lu *lu)
{
    lu->nupdate = -1; /* invalidate factorization */
    lu->nforrest = 0;
    lu->l_nz = 0;
    lu->u_nz = 0;
    lu->r_nz = 0;
    lu->min_pivot = 0.0;
    lu->max_pivot = 0.0;
    lu->max_eta = 0.0;
    lu->update_cost_numer = 0.0;
    lu->update_cost_denom = 1.0;
    lu->time_factorize = 0.0;
    lu->time_solve = 0.0;
    lu->time_update = 0.0;
    lu->l_flops = 0;
    lu->u_flops = 0;
    lu->r_flops = 0;
    lu->condest_l = 0.0;
    lu->condest_u = 0.0;
    lu->norm_l = 0.0;
    lu->norm_u = 0.0;
    lu->normest_l_inv = 0.0;
    lu->normest_u_inv = 0.0;
    lu->onenorm = 0.0;
    lu->infnorm = 0.0;
    lu->residual_test = 0.0;

    lu->matrix_nz = 0;
    lu->rank = 0;
    lu->bump_size = 0;
    lu->bump_nz = 0;
    lu->nsearch_pivot = 0;
    lu->nexpand = 0;
    lu->ngarbage = 0;
    lu->factor_flops = 0;
    lu->time_singletons = 0.0;
    lu->time_search_pivot = 0.0;
    lu->time_elim_pivot = 0.0;
    lu->pivot_error = 0.0;
    lu->d3_hits = 0; /* test hook counter, not in the reference */
    for (int k = 0; k < 6; k++) lu->npivot_kind[k] = 0;

    lu->task = ORC_TASK_NONE;
    lu->pivot_row = -1;
    lu->pivot_col = -1;
    lu->ftran_for_update = -1;
    lu->btran_for_update = -1;
    lu->marker = 0;
    lu->pivotlen = 0;
    lu->rankdef = 0;
    lu->min_colnz = 1;
    lu->min_rownz = 1;

    lu->w_end[2 * lu->m] = lu->w_mem; /* lu.rs:385 (D5: the only place besides file_empty) */

    memset(lu->iwork0, 0, (size_t)lu->m * sizeof(lu_int));
    for (lu_int i = 0; i < lu->m; i++) lu->work0[i] = 0.0;
}

/* ------------------------------------------------------------------------- */
/* count lists -- src/lu/list.rs                                              */
/* ------------------------------------------------------------------------- */

/* list.rs:36-51 */
void orc_list_init(lu_int *flink, lu_int *blink, lu_int nelem, lu_int nlist, lu_int *min_list)
{
    for (lu_int i = 0; i < nelem + nlist; i++) {
        flink[i] = i;
        blink[i] = i;
    }
    if (min_list) *min_list = nlist > 1 ? nlist : 1;
}

/* list.rs:54-77: append at the TAIL */
void orc_list_add(lu_int elem, lu_int list, lu_int *flink, lu_int *blink, lu_int nelem, lu_int *min_list)
{
    ORC_ASSERT(flink[elem] == elem);
    ORC_ASSERT(blink[elem] == elem);
    lu_int temp = blink[nelem + list];
    blink[nelem + list] = elem;
    blink[elem] = temp;
    flink[temp] = elem;
    flink[elem] = nelem + list;
    if (min_list) {
        if (list > 0 && list < *min_list) *min_list = list;
    }
}

/* list.rs:81-86 */
void orc_list_remove(lu_int *flink, lu_int *blink, lu_int elem)
{
    flink[blink[elem]] = flink[elem];
    blink[flink[elem]] = blink[elem];
    flink[elem] = elem;
    blink[elem] = elem;
}

/* list.rs:89-99 */
void orc_list_move(lu_int elem, lu_int list, lu_int *flink, lu_int *blink, lu_int nelem, lu_int *min_list)
{
    orc_list_remove(flink, blink, elem);
    orc_list_add(elem, list, flink, blink, nelem, min_list);
}

/* list.rs:104-137 */
void orc_list_swap(lu_int *flink, lu_int *blink, lu_int e1, lu_int e2)
{
    lu_int e1next = flink[e1];
    lu_int e2next = flink[e2];
    lu_int e1prev = blink[e1];
    lu_int e2prev = blink[e2];

    ORC_ASSERT(e1next != e1);
    ORC_ASSERT(e2next != e2);

    if (e1next == e2) {
        flink[e2] = e1;
        blink[e1] = e2;
        flink[e1prev] = e2;
        blink[e2] = e1prev;
        flink[e1] = e2next;
        blink[e2next] = e1;
    } else if (e2next == e1) {
        flink[e1] = e2;
        blink[e2] = e1;
        flink[e2] = e1next;
        blink[e1next] = e2;
        flink[e2prev] = e1;
        blink[e1] = e2prev;
    } else {
        flink[e2] = e1next;
        blink[e1next] = e2;
        flink[e2prev] = e1;
        blink[e1] = e2prev;
        flink[e1prev] = e2;
        blink[e2] = e1prev;
        flink[e1] = e2next;
        blink[e2next] = e1;
    }
}

/* ------------------------------------------------------------------------- */
/* data file -- src/lu/file.rs                                                */
/* ------------------------------------------------------------------------- */

/* file.rs:32-52 */
void orc_file_empty(lu_int nlines, lu_int *begin, lu_int *end, lu_int *next, lu_int *prev, lu_int fmem)
{
    begin[nlines] = 0;
    end[nlines] = fmem;
    for (lu_int i = 0; i < nlines; i++) {
        begin[i] = 0;
        end[i] = 0;
    }
    for (lu_int i = 0; i < nlines; i++) {
        next[i] = i + 1;
        prev[i + 1] = i;
    }
    next[nlines] = 0;
    prev[0] = nlines;
}

/* file.rs:56-85 */
void orc_file_reappend(lu_int line, lu_int nlines, lu_int *begin, lu_int *end, lu_int *next, lu_int *prev,
                       lu_int *index, double *value, lu_int extra_space)
{
    lu_int fmem = end[nlines];
    lu_int used = begin[nlines];
    lu_int room = fmem - used;
    lu_int ibeg = begin[line];
    lu_int iend = end[line];
    begin[line] = used;
    ORC_ASSERT(iend - ibeg <= room);
    for (lu_int pos = ibeg; pos < iend; pos++) {
        index[used] = index[pos];
        value[used] = value[pos];
        used++;
    }
    end[line] = used;
    room = fmem - used;
    ORC_ASSERT(room >= extra_space);
    used += extra_space;
    begin[nlines] = used;
    orc_list_move(line, 0, next, prev, nlines, NULL);
}

/* file.rs:92-135 */
lu_int orc_file_compress(lu_int nlines, lu_int *begin, lu_int *end, const lu_int *next,
                         lu_int *index, double *value, double stretch, lu_int pad)
{
    lu_int nz = 0;
    lu_int used = 0;
    lu_int extra_space = 0;
    lu_int i = next[nlines];
    while (i < nlines) {
        lu_int ibeg = begin[i];
        lu_int iend = end[i];
        ORC_ASSERT(ibeg >= used);
        used += extra_space;
        if (used > ibeg) used = ibeg; /* chop extra space added before */
        begin[i] = used;
        for (lu_int pos = ibeg; pos < iend; pos++) {
            index[used] = index[pos];
            value[used] = value[pos];
            used++;
        }
        end[i] = used;
        extra_space = orc_trunc(stretch * (double)(iend - ibeg)) + pad;
        nz += iend - ibeg;
        i = next[i];
    }
    ORC_ASSERT(used <= begin[nlines]);
    used += extra_space;
    if (used > begin[nlines]) used = begin[nlines];
    begin[nlines] = used;
    return nz;
}

/* file.rs:151-181 */
lu_int orc_file_diff(lu_int nrow, const lu_int *begin_row, const lu_int *end_row,
                     const lu_int *begin_col, const lu_int *end_col,
                     const lu_int *index, const double *value)
{
    lu_int ndiff = 0;
    for (lu_int i = 0; i < nrow; i++) {
        for (lu_int pos = begin_row[i]; pos < end_row[i]; pos++) {
            lu_int j = index[pos];
            lu_int where_ = begin_col[j];
            while (where_ < end_col[j] && index[where_] != i) where_++;
            if (where_ == end_col[j]) {
                ndiff++;
            } else if (value) {
                if (value[pos] != value[where_]) ndiff++;
            }
        }
    }
    return ndiff;
}

/* ------------------------------------------------------------------------- */
/* parameters / stats by key (numbering of include/blu_hip.h)                 */
/* ------------------------------------------------------------------------- */
int orc_set_param(orc_lu *lu, int key, double v)
{
    switch (key) {
    case BLU_PARAM_DROPTOL: lu->droptol = v; break;
    case BLU_PARAM_ABSTOL: lu->abstol = v; break;
    case BLU_PARAM_RELTOL: lu->reltol = v; break;
    case BLU_PARAM_NZBIAS: lu->nzbias = v < 0 ? -1 : (lu_int)v; break;
    case BLU_PARAM_MAXSEARCH: lu->maxsearch = (lu_int)v; break;
    case BLU_PARAM_PAD: lu->pad = (lu_int)v; break;
    case BLU_PARAM_STRETCH: lu->stretch = v; break;
    case BLU_PARAM_COMPRESS_THRES: lu->compress_thres = v; break;
    case BLU_PARAM_SPARSE_THRES: lu->sparse_thres = v; break;
    case BLU_PARAM_SEARCH_ROWS: lu->search_rows = (lu_int)v; break;
    default: return ORC_ERROR_INVALID_ARGUMENT;
    }
    return ORC_OK;
}

double orc_get_stat(const orc_lu *lu, int key)
{
    switch (key) {
    case BLU_STAT_M: return (double)lu->m;
    case BLU_STAT_NUPDATE: return (double)lu->nupdate;
    case BLU_STAT_NFACTORIZE: return (double)lu->nfactorize;
    case BLU_STAT_L_NZ: return (double)lu->l_nz;
    case BLU_STAT_U_NZ: return (double)lu->u_nz;
    case BLU_STAT_MIN_PIVOT: return lu->min_pivot;
    case BLU_STAT_MAX_PIVOT: return lu->max_pivot;
    case BLU_STAT_CONDEST_L: return lu->condest_l;
    case BLU_STAT_CONDEST_U: return lu->condest_u;
    case BLU_STAT_NORM_L: return lu->norm_l;
    case BLU_STAT_NORM_U: return lu->norm_u;
    case BLU_STAT_NORMEST_L_INV: return lu->normest_l_inv;
    case BLU_STAT_NORMEST_U_INV: return lu->normest_u_inv;
    case BLU_STAT_ONENORM: return lu->onenorm;
    case BLU_STAT_INFNORM: return lu->infnorm;
    case BLU_STAT_RESIDUAL_TEST: return lu->residual_test;
    case BLU_STAT_MATRIX_NZ: return (double)lu->matrix_nz;
    case BLU_STAT_RANK: return (double)lu->rank;
    case BLU_STAT_BUMP_SIZE: return (double)lu->bump_size;
    case BLU_STAT_BUMP_NZ: return (double)lu->bump_nz;
    case BLU_STAT_NSEARCH_PIVOT: return (double)lu->nsearch_pivot;
    case BLU_STAT_NEXPAND: return (double)lu->nexpand;
    case BLU_STAT_NGARBAGE: return (double)lu->ngarbage;
    case BLU_STAT_FACTOR_FLOPS: return (double)lu->factor_flops;
    case BLU_STAT_TIME_FACTORIZE: return lu->time_factorize;
    case BLU_STAT_TIME_SINGLETONS: return lu->time_singletons;
    case BLU_STAT_TIME_SEARCH_PIVOT: return lu->time_search_pivot;
    case BLU_STAT_TIME_ELIM_PIVOT: return lu->time_elim_pivot;
    case BLU_STAT_UPDATE_COST_DENOM: return lu->update_cost_denom;
    case BLU_STAT_RANKDEF: return (double)lu->rankdef;
    case BLU_STAT_L_MEM: return (double)lu->l_mem;
    case BLU_STAT_U_MEM: return (double)lu->u_mem;
    case BLU_STAT_W_MEM: return (double)lu->w_mem;
    case BLU_STAT_L_FLOPS: return (double)lu->l_flops;
    case BLU_STAT_U_FLOPS: return (double)lu->u_flops;
    case BLU_STAT_NFORREST: return (double)lu->nforrest;
    case BLU_STAT_PIVOT_ERROR: return lu->pivot_error;
    case BLU_STAT_R_NZ: return (double)lu->r_nz;
    case BLU_STAT_R_FLOPS: return (double)lu->r_flops;
    case BLU_STAT_MAX_ETA: return lu->max_eta;
    case BLU_STAT_NSYMPERM_TOTAL: return (double)lu->nsymperm_total;
    case BLU_STAT_NFORREST_TOTAL: return (double)lu->nforrest_total;
    case BLU_STAT_DEV_NUNSYMPERM_TOTAL: return (double)lu->nunsymperm_total;
    case BLU_STAT_UPDATE_COST: return lu->update_cost_numer / lu->update_cost_denom; /* lu.rs:324-326 */
    case 50: return (double)lu->d3_hits;
    case 51: case 52: case 53: case 54: case 55: case 56: return (double)lu->npivot_kind[key - 51];
    default: return NAN;
    }
}
