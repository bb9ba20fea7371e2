/* orc_internal.h -- shared macros of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "blu_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* The reference's assert!/assert_eq! are always on (Rust does not compile them
 * out in release).  A failed assert is a panic there; here it aborts loudly. */
#define ORC_ASSERT(cond)                                                              \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "blu oracle: assertion failed: %s (%s:%d)\n", #cond,      \
                    __FILE__, __LINE__);                                              \
            abort();                                                                  \
        }                                                                             \
    } while (0)

/* alias macros, src/lu/lu.rs:173-233 */
#define PIVOTCOL(lu) ((lu)->colcount_flink)
#define PIVOTROW(lu) ((lu)->colcount_blink)
#define R_BEGIN(lu) ((lu)->rowcount_flink)
/* D13: the reference's eta_row! expands to the same array as r_begin! with no offset, so eta_row[t] = ipivot
 * overwrites r_begin[t].  With no update nothing is ever stored there (nforrest = 0), so the factorize and
 * fresh-solve paths cannot tell.  The update path (orc_update.c, the INTENDED algorithm, not reference-pinned)
 * needs both arrays: eta_row lives in the second half, as in upstream BASICLU. */
#define ETA_ROW(lu) ((lu)->rowcount_flink + (lu)->m + 1)
#define IWORK1(lu) ((lu)->rowcount_blink)
#define L_BEGIN(lu) ((lu)->w_begin + (lu)->m + 1)
#define LT_BEGIN(lu) ((lu)->w_end + (lu)->m + 1)
#define LT_BEGIN_P(lu) ((lu)->w_flink + (lu)->m + 1)
#define P_(lu) ((lu)->w_blink + (lu)->m + 1)
#define PMAP(lu) ((lu)->pinv)
#define QMAP(lu) ((lu)->qinv)
#define MARKED(lu) ((lu)->iwork0)

/* Rust `(x as f64 * s) as usize`: truncation toward zero (values are >= 0). */
static inline lu_int orc_trunc(double x) { return (lu_int)x; }

static inline void orc_iswap(lu_int *x, lu_int i, lu_int j) /* def.rs:20 */
{
    lu_int t = x[i];
    x[i] = x[j];
    x[j] = t;
}
static inline void orc_fswap(double *x, lu_int i, lu_int j) /* def.rs:26 */
{
    double t = x[i];
    x[i] = x[j];
    x[j] = t;
}

double orc_now(void);

/* kernel layer */
int orc_singletons(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x);
int orc_setup_bump(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x);
int orc_markowitz(orc_lu *lu);
int orc_pivot(orc_lu *lu);
int orc_factorize_bump(orc_lu *lu);
int orc_build_factors(orc_lu *lu);
double orc_condest(lu_int m, const lu_int *u_begin, const lu_int *u_i, const double *u_x,
                   const double *pivot, const lu_int *perm, int upper, double *work,
                   double *norm, double *norminv);
void orc_residual_test(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x);
void orc_matrix_norm(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end, const uint64_t *b_i, const double *b_x);
void orc_lu_solve_dense(orc_lu *lu, const double *rhs, double *lhs, char trans);
void orc_garbage_perm(orc_lu *lu);
lu_int orc_dfs(lu_int i, const lu_int *begin, const lu_int *end, const lu_int *index, lu_int top,
               lu_int *xi, double *pstack, lu_int *marked, lu_int M);
lu_int orc_solve_symbolic(lu_int m, const lu_int *begin, const lu_int *end, const lu_int *index,
                          lu_int nrhs, const lu_int *irhs, lu_int *ilhs, double *pstack,
                          lu_int *marked, lu_int M);
lu_int orc_solve_triangular(lu_int nz_symb, const lu_int *pattern_symb, const lu_int *begin,
                            const lu_int *end, const lu_int *index, const double *value,
                            const double *pivot, double droptol, double *lhs, lu_int *pattern,
                            lu_int *flops);
int orc_lu_solve_for_update(orc_lu *lu, lu_int nrhs, const lu_int *irhs, const double *xrhs, lu_int *p_nlhs,
                            lu_int *ilhs, double *xlhs, char trans);
int orc_lu_update(orc_lu *lu, double xtbl);
void orc_clear_lhs(orc_blu *obj);

int orc_lu_init(orc_lu *lu, lu_int m, lu_int b_nz);
void orc_lu_reset(orc_lu *lu);
void orc_lu_destroy(orc_lu *lu);

#endif
