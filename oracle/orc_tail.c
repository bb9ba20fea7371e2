/* orc_tail.c -- CPU oracle (test infrastructure).
 * Follows src/lu/condest.rs, src/lu/residual_test.rs, src/lu/matrix_norm.rs,
 * src/lu/solve_dense.rs and src/lu/garbage_perm.rs of /root/reference. */
#include "orc_internal.h"

/* normest -- condest.rs:74-157 */
static double orc_normest(lu_int m, const lu_int *u_begin, const lu_int *u_i, const double *u_x,
                          const double *pivot, const lu_int *perm, int upper, double *work)
{
    double x1norm = 0.0, xinfnorm = 0.0;
    lu_int kbeg, kend, kinc;
    if (upper) { kbeg = 0; kend = m; kinc = 1; }
    else { kbeg = m - 1; kend = -1; kinc = -1; }

    for (lu_int k = kbeg; k != kend; k += kinc) {
        lu_int j = perm ? perm[k] : k;
        double temp = 0.0;
        for (lu_int p = u_begin[j]; u_i[p] >= 0; p++) temp -= work[u_i[p]] * u_x[p];
        temp += temp >= 0.0 ? 1.0 : -1.0; /* choose b[i] = 1 or b[i] = -1 */
        if (pivot) temp /= pivot[j];
        work[j] = temp;
        x1norm += fabs(temp);
        xinfnorm = fmax(xinfnorm, fabs(temp));
    }

    double y1norm = 0.0;
    if (upper) { kbeg = m - 1; kend = -1; kinc = -1; }
    else { kbeg = 0; kend = m; kinc = 1; }
    for (lu_int k = kbeg; k != kend; k += kinc) {
        lu_int j = perm ? perm[k] : k;
        if (pivot) work[j] /= pivot[j];
        double temp = work[j];
        for (lu_int p = u_begin[j]; u_i[p] >= 0; p++) work[u_i[p]] -= temp * u_x[p];
        y1norm += fabs(temp);
    }
    return fmax(y1norm / x1norm, xinfnorm);
}

/* condest -- condest.rs:15-56 */
double orc_condest(lu_int m, const lu_int *u_begin, const lu_int *u_i, const double *u_x,
                   const double *pivot, const lu_int *perm, int upper, double *work,
                   double *norm, double *norminv)
{
    double u_norm = 0.0;
    for (lu_int j = 0; j < m; j++) {
        double colsum = pivot ? fabs(pivot[j]) : 1.0;
        for (lu_int p = u_begin[j]; u_i[p] >= 0; p++) colsum += fabs(u_x[p]);
        u_norm = fmax(u_norm, colsum);
    }
    double u_invnorm = orc_normest(m, u_begin, u_i, u_x, pivot, perm, upper, work);
    if (norm) *norm = u_norm;
    if (norminv) *norminv = u_invnorm;
    return u_norm * u_invnorm;
}

/* matrix_norm -- matrix_norm.rs:8-48 */
void orc_matrix_norm(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x)
{
    const lu_int m = lu->m, rank = lu->rank;
    const lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    double *rowsum = lu->work1;
    ORC_ASSERT(lu->nupdate == 0);

    for (lu_int i = 0; i < m; i++) rowsum[i] = 0.0;
    double onenorm = 0.0, infnorm = 0.0;
    for (lu_int k = 0; k < rank; k++) {
        lu_int jpivot = pivotcol[k];
        double colsum = 0.0;
        for (uint64_t pos = b_begin[jpivot]; pos < b_end[jpivot]; pos++) {
            colsum += fabs(b_x[pos]);
            rowsum[b_i[pos]] += fabs(b_x[pos]);
        }
        onenorm = fmax(onenorm, colsum);
    }
    for (lu_int k = rank; k < m; k++) {
        lu_int ipivot = pivotrow[k];
        rowsum[ipivot] += 1.0;
        onenorm = fmax(onenorm, 1.0);
    }
    for (lu_int i = 0; i < m; i++) infnorm = fmax(infnorm, rowsum[i]);
    lu->onenorm = onenorm;
    lu->infnorm = infnorm;
}

static double onenorm_vec(lu_int m, const double *x) /* residual_test.rs:7-13 */
{
    double d = 0.0;
    for (lu_int i = 0; i < m; i++) d += fabs(x[i]);
    return d;
}

/* residual_test -- residual_test.rs:16-152 */
void orc_residual_test(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x)
{
    const lu_int m = lu->m, rank = lu->rank;
    const lu_int *p = P_(lu);
    const lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    const lu_int *l_begin_p = lu->l_begin_p;
    const lu_int *lt_begin_p = LT_BEGIN_P(lu);
    const lu_int *u_begin = lu->u_begin;
    const double *row_pivot = lu->row_pivot;
    const lu_int *l_index = lu->l_index, *u_index = lu->u_index;
    const double *l_value = lu->l_value, *u_value = lu->u_value;
    double *rhs = lu->work0;
    double *lhs = lu->work1;

    ORC_ASSERT(lu->nupdate == 0);

    /* ---- Residual Test with Forward System ---- */
    /* Compute lhs = L\rhs and build rhs on-the-fly. */
    for (lu_int k = 0; k < m; k++) {
        double d = 0.0;
        for (lu_int pos = lt_begin_p[k]; l_index[pos] >= 0; pos++) d += lhs[l_index[pos]] * l_value[pos];
        lu_int ipivot = p[k];
        rhs[ipivot] = d <= 0.0 ? 1.0 : -1.0;
        lhs[ipivot] = rhs[ipivot] - d;
    }
    /* Overwrite lhs by U\lhs. */
    for (lu_int k = m - 1; k >= 0; k--) {
        lu_int ipivot = pivotrow[k];
        lhs[ipivot] /= row_pivot[ipivot];
        double d = lhs[ipivot];
        for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) lhs[u_index[pos]] -= d * u_value[pos];
    }
    /* Overwrite rhs by the residual rhs-B*lhs. */
    for (lu_int k = 0; k < rank; k++) {
        lu_int ipivot = pivotrow[k];
        lu_int jpivot = pivotcol[k];
        double d = lhs[ipivot];
        for (uint64_t pos = b_begin[jpivot]; pos < b_end[jpivot]; pos++) rhs[b_i[pos]] -= d * b_x[pos];
    }
    for (lu_int k = rank; k < m; k++) {
        lu_int ipivot = pivotrow[k];
        rhs[ipivot] -= lhs[ipivot];
    }
    double norm_ftran = onenorm_vec(m, lhs);
    double norm_ftran_res = onenorm_vec(m, rhs);

    /* ---- Residual Test with Backward System ---- */
    /* Compute lhs = U'\rhs and build rhs on-the-fly. */
    for (lu_int k = 0; k < m; k++) {
        lu_int ipivot = pivotrow[k];
        double d = 0.0;
        for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) d += lhs[u_index[pos]] * u_value[pos];
        rhs[ipivot] = d <= 0.0 ? 1.0 : -1.0;
        lhs[ipivot] = (rhs[ipivot] - d) / row_pivot[ipivot];
    }
    /* Overwrite lhs by L'\lhs. */
    for (lu_int k = m - 1; k >= 0; k--) {
        double d = 0.0;
        for (lu_int pos = l_begin_p[k]; l_index[pos] >= 0; pos++) d += lhs[l_index[pos]] * l_value[pos];
        lhs[p[k]] -= d;
    }
    /* Overwrite rhs by the residual rhs-B'*lhs. */
    for (lu_int k = 0; k < rank; k++) {
        lu_int ipivot = pivotrow[k];
        lu_int jpivot = pivotcol[k];
        double d = 0.0;
        for (uint64_t pos = b_begin[jpivot]; pos < b_end[jpivot]; pos++) d += lhs[b_i[pos]] * b_x[pos];
        rhs[ipivot] -= d;
    }
    for (lu_int k = rank; k < m; k++) {
        lu_int ipivot = pivotrow[k];
        rhs[ipivot] -= lhs[ipivot];
    }
    double norm_btran = onenorm_vec(m, lhs);
    double norm_btran_res = onenorm_vec(m, rhs);

    /* ---- Finalize ---- */
    orc_matrix_norm(lu, b_begin, b_end, b_i, b_x);
    ORC_ASSERT(lu->onenorm > 0.0);
    ORC_ASSERT(lu->infnorm > 0.0);
    lu->residual_test = fmax(norm_ftran_res / ((double)m + lu->onenorm * norm_ftran),
                             norm_btran_res / ((double)m + lu->infnorm * norm_btran));

    for (lu_int i = 0; i < m; i++) lu->work0[i] = 0.0; /* reset workspace */
}

/* garbage_perm -- garbage_perm.rs:16-48 */
void orc_garbage_perm(orc_lu *lu)
{
    const lu_int m = lu->m;
    const lu_int pivotlen = lu->pivotlen;
    lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    lu_int *marked = MARKED(lu);

    if (pivotlen > m) {
        lu->marker++;
        lu_int marker = lu->marker;
        lu_int put = pivotlen;
        for (lu_int get = pivotlen - 1; get >= 0; get--) {
            if (marked[pivotcol[get]] != marker) {
                lu_int j = pivotcol[get];
                marked[j] = marker;
                put--;
                pivotcol[put] = j;
                pivotrow[put] = pivotrow[get];
            }
        }
        ORC_ASSERT(put + m == pivotlen);
        memmove(pivotcol, pivotcol + put, (size_t)m * sizeof(lu_int));
        memmove(pivotrow, pivotrow + put, (size_t)m * sizeof(lu_int));
        lu->pivotlen = m;
    }
}

/* lu::solve_dense -- lu/solve_dense.rs:7-120 */
void orc_lu_solve_dense(orc_lu *lu, const double *rhs, double *lhs, char trans)
{
    orc_garbage_perm(lu);
    ORC_ASSERT(lu->pivotlen == lu->m);

    const lu_int m = lu->m;
    const lu_int nforrest = lu->nforrest;
    const lu_int *p = P_(lu);
    const lu_int *eta_row = ETA_ROW(lu);
    const lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    const lu_int *l_begin_p = lu->l_begin_p;
    const lu_int *lt_begin_p = LT_BEGIN_P(lu);
    const lu_int *u_begin = lu->u_begin;
    const lu_int *r_begin = R_BEGIN(lu);
    const lu_int *w_begin = lu->w_begin, *w_end = lu->w_end;
    const double *col_pivot = lu->col_pivot, *row_pivot = lu->row_pivot;
    const lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    const double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    double *work1 = lu->work1;

    if (trans == 't' || trans == 'T') {
        /* Solve transposed system (:32-74) */
        memcpy(work1, rhs, (size_t)m * sizeof(double));

        /* Solve with U'. */
        for (lu_int k = 0; k < m; k++) {
            lu_int jpivot = pivotcol[k];
            lu_int ipivot = pivotrow[k];
            double x = work1[jpivot] / col_pivot[jpivot];
            for (lu_int pos = w_begin[jpivot]; pos < w_end[jpivot]; pos++) work1[w_index[pos]] -= x * w_value[pos];
            lhs[ipivot] = x;
        }
        /* Solve with update ETAs backwards. */
        for (lu_int t = nforrest - 1; t >= 0; t--) {
            lu_int ipivot = eta_row[t];
            double x = lhs[ipivot];
            for (lu_int pos = r_begin[t]; pos < r_begin[t + 1]; pos++) lhs[l_index[pos]] -= x * l_value[pos];
        }
        /* Solve with L'. */
        for (lu_int k = m - 1; k >= 0; k--) {
            double x = 0.0;
            for (lu_int pos = l_begin_p[k]; l_index[pos] >= 0; pos++) x += lhs[l_index[pos]] * l_value[pos];
            lhs[p[k]] -= x;
        }
    } else {
        /* Solve forward system (:75-119) */
        memcpy(work1, rhs, (size_t)m * sizeof(double));

        /* Solve with L. */
        for (lu_int k = 0; k < m; k++) {
            double x = 0.0;
            for (lu_int pos = lt_begin_p[k]; l_index[pos] >= 0; pos++) x += work1[l_index[pos]] * l_value[pos];
            work1[p[k]] -= x;
        }
        /* Solve with update ETAs. */
        lu_int pos = r_begin[0];
        for (lu_int t = 0; t < nforrest; t++) {
            lu_int ipivot = eta_row[t];
            double x = 0.0;
            while (pos < r_begin[t + 1]) {
                x += work1[l_index[pos]] * l_value[pos];
                pos++;
            }
            work1[ipivot] -= x;
        }
        /* Solve with U. */
        for (lu_int k = m - 1; k >= 0; k--) {
            lu_int jpivot = pivotcol[k];
            lu_int ipivot = pivotrow[k];
            double x = work1[ipivot] / row_pivot[ipivot];
            for (lu_int pos2 = u_begin[ipivot]; u_index[pos2] >= 0; pos2++) work1[u_index[pos2]] -= x * u_value[pos2];
            lhs[jpivot] = x;
        }
    }
}
