/* orc_api.c -- CPU oracle (test infrastructure).
 * Follows src/factorize.rs, src/solve_dense.rs and src/blu.rs of /root/reference,
 * plus the synthetic LP-basis generator of SURVEY.md 8d. */
#include "orc_internal.h"

/* factorize -- factorize.rs:34-182 */
int orc_factorize(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end,
                  const uint64_t *b_i, const double *b_x, int c0ntinue)
{
    double tic = orc_now();
    int status;

    if (!c0ntinue) {
        orc_lu_reset(lu);
        lu->task = ORC_TASK_SINGLETONS;
    }

#define RETURN_TO_CALLER(st)                       \
    do {                                           \
        double el_ = orc_now() - tic;              \
        lu->time_factorize += el_;                 \
        lu->time_factorize_total += el_;           \
        return (st);                               \
    } while (0)

    /* continue factorization (:61-106) */
    switch (lu->task) {
    case ORC_TASK_SINGLETONS:
        status = orc_singletons(lu, b_begin, b_end, b_i, b_x);
        if (status != ORC_OK) RETURN_TO_CALLER(status);
        lu->task = ORC_TASK_SETUP_BUMP;
        /* fall through */
    case ORC_TASK_SETUP_BUMP:
        status = orc_setup_bump(lu, b_begin, b_end, b_i, b_x);
        if (status != ORC_OK) RETURN_TO_CALLER(status);
        lu->task = ORC_TASK_FACTORIZE_BUMP;
        /* fall through */
    case ORC_TASK_FACTORIZE_BUMP:
        status = orc_factorize_bump(lu);
        if (status != ORC_OK) RETURN_TO_CALLER(status);
        break;
    case ORC_TASK_BUILD_FACTORS:
        break;
    default:
        return ORC_ERROR_INVALID_CALL;
    }

    lu->task = ORC_TASK_BUILD_FACTORS;
    status = orc_build_factors(lu);
    if (status != ORC_OK) RETURN_TO_CALLER(status);

    /* factorization successfully finished (:114-119) */
    lu->task = ORC_TASK_NONE;
    lu->nupdate = 0; /* make factorization valid */
    lu->ftran_for_update = -1;
    lu->btran_for_update = -1;
    lu->nfactorize++;

    /* (:121-144) */
    lu->condest_l = orc_condest(lu->m, L_BEGIN(lu), lu->l_index, lu->l_value, NULL, P_(lu), 0, lu->work1,
                                &lu->norm_l, &lu->normest_l_inv);
    lu->condest_u = orc_condest(lu->m, lu->u_begin, lu->u_index, lu->u_value, lu->row_pivot, P_(lu), 1, lu->work1,
                                &lu->norm_u, &lu->normest_u_inv);

    /* measure numerical stability of the factorization (:147) */
    orc_residual_test(lu, b_begin, b_end, b_i, b_x);

    /* (:160-166) */
    double factor_cost = 0.04 * (double)lu->m + 0.07 * (double)lu->matrix_nz + 0.20 * (double)lu->bump_nz +
                         0.20 * (double)lu->nsearch_pivot + 0.008 * (double)lu->factor_flops;
    lu->update_cost_denom = factor_cost * 250.0;

    if (lu->rank < lu->m) RETURN_TO_CALLER(ORC_WARNING_SINGULAR_MATRIX);
    RETURN_TO_CALLER(ORC_OK);
#undef RETURN_TO_CALLER
}

/* solve_dense -- solve_dense.rs:24-32 */
int orc_solve_dense(orc_lu *lu, const double *rhs, double *lhs, char trans)
{
    if (lu->nupdate < 0) return ORC_ERROR_INVALID_CALL;
    orc_lu_solve_dense(lu, rhs, lhs, trans);
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* struct BLU -- src/blu.rs                                                   */
/* ------------------------------------------------------------------------- */

/* BLU::new -- blu.rs:61-70 */
orc_blu *orc_blu_new(lu_int m, lu_int b_nz)
{
    if (m < 0 || b_nz < 0) return NULL;
    orc_blu *obj = calloc(1, sizeof(*obj));
    if (!obj) return NULL;
    if (orc_lu_init(&obj->lu, m, b_nz) != 0) {
        free(obj);
        return NULL;
    }
    obj->lhs = calloc((size_t)(m ? m : 1), sizeof(double));
    obj->ilhs = calloc((size_t)(m ? m : 1), sizeof(lu_int));
    obj->nzlhs = 0;
    obj->realloc_factor = 1.5;
    if (!obj->lhs || !obj->ilhs) {
        orc_blu_free(obj);
        return NULL;
    }
    return obj;
}

void orc_blu_free(orc_blu *obj)
{
    if (!obj) return;
    orc_lu_destroy(&obj->lu);
    free(obj->lhs);
    free(obj->ilhs);
    free(obj);
}

orc_lu *orc_blu_lu(orc_blu *obj) { return &obj->lu; }

/* lu_reallocix -- blu.rs:338-342: Vec::resize keeps contents, zero-fills */
static int reallocix(lu_int nz, lu_int old, lu_int **a_i, double **a_x)
{
    lu_int *ni = realloc(*a_i, (size_t)(nz ? nz : 1) * sizeof(lu_int));
    if (!ni) return -1;
    *a_i = ni;
    double *nx = realloc(*a_x, (size_t)(nz ? nz : 1) * sizeof(double));
    if (!nx) return -1;
    *a_x = nx;
    for (lu_int k = old; k < nz; k++) {
        ni[k] = 0;
        nx[k] = 0.0;
    }
    return 0;
}

/* lu_realloc_obj -- blu.rs:345-377.  D6: addmem_* are never reset.
 * D5: w_end[2m] (the file capacity) is NOT updated here. */
static int realloc_obj(orc_blu *obj)
{
    orc_lu *lu = &obj->lu;
    const lu_int addmem_l = lu->addmem_l, addmem_u = lu->addmem_u, addmem_w = lu->addmem_w;
    const double realloc_factor = fmax(1.0, obj->realloc_factor);

    if (addmem_l > 0) {
        lu_int nelem = lu->l_mem + addmem_l;
        nelem = (lu_int)((double)nelem * realloc_factor);
        if (reallocix(nelem, lu->l_mem, &lu->l_index, &lu->l_value)) return -1;
        lu->l_mem = nelem;
    }
    if (addmem_u > 0) {
        lu_int nelem = lu->u_mem + addmem_u;
        nelem = (lu_int)((double)nelem * realloc_factor);
        if (reallocix(nelem, lu->u_mem, &lu->u_index, &lu->u_value)) return -1;
        lu->u_mem = nelem;
    }
    if (addmem_w > 0) {
        lu_int nelem = lu->w_mem + addmem_w;
        nelem = (lu_int)((double)nelem * realloc_factor);
        if (reallocix(nelem, lu->w_mem, &lu->w_index, &lu->w_value)) return -1;
        lu->w_mem = nelem;
    }
    return 0;
}

/* BLU::factorize -- blu.rs:95-118 */
int orc_blu_factorize(orc_blu *obj, const uint64_t *b_begin, const uint64_t *b_end,
                      const uint64_t *b_i, const double *b_x)
{
    int c0ntinue = 0;
    int result;
    lu_int stuck_rank = -1, stuck = 0;
    for (;;) {
        result = orc_factorize(&obj->lu, b_begin, b_end, b_i, b_x, c0ntinue);
        if (result == ORC_REALLOCATE) {
            /* Test-harness guard, not in the reference.  D5: after W has been reallocated in the middle
             * of the bump, w_end[2m] (the file capacity) is stale, pivot() asks for the same memory again
             * and the reference loops, growing W by realloc_factor each time until the process dies.  The
             * restatement reproduces that; three Reallocate returns in a row from the same pivot without
             * progress are reported as ORC_D5_TRAP instead (the caller re-runs with a W that is large
             * enough from the start, where results do not depend on the layout: SURVEY 5.2-5). */
            const lu_int at = obj->lu.rank + obj->lu.rankdef;
            if (obj->lu.task == ORC_TASK_FACTORIZE_BUMP && obj->lu.addmem_w > 0 && at == stuck_rank) {
                if (++stuck >= 3) return ORC_D5_TRAP;
            } else {
                stuck_rank = at;
                stuck = 0;
            }
            if (realloc_obj(obj)) return -9;
            c0ntinue = 1;
            continue;
        }
        break;
    }
    return result;
}

/* solve_for_update -- src/solve_for_update.rs:73-119 (argument checks), then lu::solve_for_update.
 * The update path is the INTENDED algorithm, not reference-pinned: see orc_update.c. */
int orc_solve_for_update(orc_lu *lu, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, lu_int *p_nzlhs,
                         lu_int *ilhs, double *lhs, char trans)
{
    const int tr = trans == 't' || trans == 'T';
    if (!tr && !xrhs) return ORC_ERROR_ARGUMENT_MISSING;
    if (lu->nupdate < 0) return ORC_ERROR_INVALID_CALL;
    if (lu->nforrest == lu->m) return ORC_ERROR_MAXIMUM_UPDATES;
    int ok;
    if (tr) {
        ok = irhs[0] < (uint64_t)lu->m;
    } else {
        ok = nzrhs >= 0 && nzrhs <= lu->m;
        for (lu_int n = 0; ok && n < nzrhs; n++) ok = ok && irhs[n] < (uint64_t)lu->m;
    }
    if (!ok) return ORC_ERROR_INVALID_ARGUMENT;
    const lu_int cnt = tr ? 1 : nzrhs;
    lu_int *ir = (lu_int *)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(lu_int));
    for (lu_int n = 0; n < cnt; n++) ir[n] = (lu_int)irhs[n];
    int st = orc_lu_solve_for_update(lu, nzrhs, ir, xrhs, p_nzlhs, ilhs, lhs, trans);
    free(ir);
    return st;
}

/* update -- src/update.rs:49-55 */
int orc_update(orc_lu *lu, double xtbl)
{
    if (lu->nupdate < 0 || lu->ftran_for_update < 0 || lu->btran_for_update < 0) return ORC_ERROR_INVALID_CALL;
    return orc_lu_update(lu, xtbl);
}

/* BLU::solve_for_update -- blu.rs:257-288.  FIX D11: the solution is computed only when it is wanted. */
int orc_blu_solve_for_update(orc_blu *obj, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, char trans, int want_solution)
{
    int result;
    orc_clear_lhs(obj);
    for (;;) {
        lu_int nzlhs = 0;
        result = want_solution ? orc_solve_for_update(&obj->lu, nzrhs, irhs, xrhs, &nzlhs, obj->ilhs, obj->lhs, trans)
                               : orc_solve_for_update(&obj->lu, nzrhs, irhs, xrhs, NULL, NULL, NULL, trans);
        if (want_solution) obj->nzlhs = nzlhs;
        if (result == ORC_REALLOCATE) {
            if (realloc_obj(obj)) return -9;
            continue;
        }
        break;
    }
    return result;
}

/* BLU::update -- blu.rs:319-335 */
int orc_blu_update(orc_blu *obj, double xtbl)
{
    int result;
    for (;;) {
        result = orc_update(&obj->lu, xtbl);
        if (result == ORC_REALLOCATE) {
            if (realloc_obj(obj)) return -9;
            continue;
        }
        break;
    }
    return result;
}

/* BLU::get_factors -- blu.rs:139-160 */
int orc_blu_get_factors(orc_blu *obj, lu_int *rowperm, lu_int *colperm,
                        lu_int *l_colptr, lu_int *l_rowidx, double *l_value,
                        lu_int *u_colptr, lu_int *u_rowidx, double *u_value)
{
    return orc_get_factors(&obj->lu, rowperm, colperm, l_colptr, l_rowidx, l_value, u_colptr, u_rowidx, u_value);
}

/* BLU::solve_dense -- blu.rs:182-184 */
int orc_blu_solve_dense(orc_blu *obj, const double *rhs, double *lhs, char trans)
{
    return orc_solve_dense(&obj->lu, rhs, lhs, trans);
}

/* ------------------------------------------------------------------------- */
/* Synthetic LP-basis generator.  Not part of the reference.   */
/* ------------------------------------------------------------------------- */
/* PRNG = SplitMix64, state = seed; u = (next() >> 11) * 2^-53.
 * Draw order (the Python twin in blu_amd/synth.py follows it exactly):
 *   for c in 0..m:
 *     diagonal (c,c): u1 -> |v| = 1+u1 ; u2 -> sign = (u2 < 0.5) ? -1 : +1
 *     window W = [max(0,c-bw), c-1]                         if c < tri_frac*m
 *              = [max(0,c-bw), min(m-1,c+bw)] without c     otherwise
 *     n = min(k-1, |W|) distinct rows by partial Fisher-Yates over W listed
 *         ascending: for t in 0..n: r = t + floor(u*(|W|-t)); swap(W[t],W[r])
 *         then value: u -> |v| = 0.1+0.9u ; u -> sign as above
 *         (index draw, value draw, sign draw per entry, in that order)
 *   row permutation P: Fisher-Yates over 0..m (for t in (1..m).rev(): r = floor(u*(t+1)); swap)
 *   column permutation Q: same, continuing the stream
 *   output column Q[c] = generated column c with rows mapped through P,
 *   entries in generation order (diagonal first) -> row indices unsorted.
 */
static inline uint64_t sm64_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double sm64_u(uint64_t *s) { return (double)(sm64_next(s) >> 11) * (1.0 / 9007199254740992.0); }

lu_int orc_gen_lp_basis(lu_int m, lu_int k, lu_int bw, double tri_frac, double offscale, uint64_t seed,
                        uint64_t *colptr, uint64_t *rowidx, double *value)
{
    uint64_t s = seed;
    lu_int *gptr = malloc((size_t)(m + 1) * sizeof(lu_int));
    lu_int *grow = malloc((size_t)(m * (k > 0 ? k : 1) + 1) * sizeof(lu_int));
    double *gval = malloc((size_t)(m * (k > 0 ? k : 1) + 1) * sizeof(double));
    lu_int *win = malloc((size_t)(2 * bw + 2) * sizeof(lu_int));
    lu_int *P = malloc((size_t)(m ? m : 1) * sizeof(lu_int));
    lu_int *Q = malloc((size_t)(m ? m : 1) * sizeof(lu_int));
    lu_int nnz = 0;
    const double tri_cut = tri_frac * (double)m;

    for (lu_int c = 0; c < m; c++) {
        gptr[c] = nnz;
        double u1 = sm64_u(&s), u2 = sm64_u(&s);
        grow[nnz] = c;
        gval[nnz] = (u2 < 0.5 ? -1.0 : 1.0) * (1.0 + u1);
        nnz++;
        lu_int nw = 0;
        lu_int lo = c - bw < 0 ? 0 : c - bw;
        if ((double)c < tri_cut) {
            for (lu_int r = lo; r <= c - 1; r++) win[nw++] = r;
        } else {
            lu_int hi = c + bw > m - 1 ? m - 1 : c + bw;
            for (lu_int r = lo; r <= hi; r++)
                if (r != c) win[nw++] = r;
        }
        lu_int n = k - 1 < nw ? k - 1 : nw;
        for (lu_int t = 0; t < n; t++) {
            lu_int r = t + (lu_int)(sm64_u(&s) * (double)(nw - t));
            lu_int tmp = win[t]; win[t] = win[r]; win[r] = tmp;
            double uv = sm64_u(&s), us = sm64_u(&s);
            grow[nnz] = win[t];
            gval[nnz] = (us < 0.5 ? -1.0 : 1.0) * (offscale * (0.1 + 0.9 * uv));
            nnz++;
        }
    }
    gptr[m] = nnz;

    for (lu_int i = 0; i < m; i++) P[i] = i;
    for (lu_int t = m - 1; t >= 1; t--) {
        lu_int r = (lu_int)(sm64_u(&s) * (double)(t + 1));
        lu_int tmp = P[t]; P[t] = P[r]; P[r] = tmp;
    }
    for (lu_int i = 0; i < m; i++) Q[i] = i;
    for (lu_int t = m - 1; t >= 1; t--) {
        lu_int r = (lu_int)(sm64_u(&s) * (double)(t + 1));
        lu_int tmp = Q[t]; Q[t] = Q[r]; Q[r] = tmp;
    }

    /* output column Q[c] = generated column c: need inverse to lay out CSC */
    lu_int *len = calloc((size_t)(m ? m : 1), sizeof(lu_int));
    for (lu_int c = 0; c < m; c++) len[Q[c]] = gptr[c + 1] - gptr[c];
    colptr[0] = 0;
    for (lu_int j = 0; j < m; j++) colptr[j + 1] = colptr[j] + (uint64_t)len[j];
    for (lu_int c = 0; c < m; c++) {
        uint64_t put = colptr[Q[c]];
        for (lu_int pos = gptr[c]; pos < gptr[c + 1]; pos++) {
            rowidx[put] = (uint64_t)P[grow[pos]];
            value[put] = gval[pos];
            put++;
        }
    }
    free(gptr); free(grow); free(gval); free(win); free(P); free(Q); free(len);
    return nnz;
}
