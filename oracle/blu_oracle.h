/*
 * blu_oracle.h -- CPU restatement of the rwl/blu v0.2.1 factorize hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.  The product path is
 * blu_amd/csrc (HIP, gfx950) behind include/blu_hip.h.
 *
 * What it is: a single-threaded plain-C restatement of the reference's Rust
 * code, function by function, with the same state arrays, the same aliases and
 * the same order of every operation that can influence a result (list order,
 * entry order inside lines, f64 comparisons, no FMA contraction).  Every
 * function cites the reference file:line it follows.  The known defects of the
 * reference that touch this path are reproduced on
 * purpose and flagged "D<n>" at the spot.
 *
 * PINNING STATUS: **parity unpinned by execution.**  The reference is a Rust
 * crate with no tests, no golden vectors and no fixtures, and there is no Rust
 * toolchain in this image, so the oracle could not be compared with outputs of
 * the reference itself.  It is anchored only by
 *   (1) the known answer of examples/simple.rs (x_i = 0.1*(i+1)),
 *   (2) the pivot-sequence prefix derived by hand from the source for that
 *       matrix, and
 *   (3) self-consistency on every fixture (B[rowperm,colperm] == L*U etc.).
 * See tests/test_oracle_*.py.
 */
#ifndef BLU_ORACLE_H
#define BLU_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t lu_int; /* reference: pub type LUInt = i64 (src/lib.rs:32) */

/* Status (src/lib.rs:38-64).  The reference enum carries no numbers; these are
 * the upstream BASICLU values. */
enum {
    ORC_OK = 0,
    ORC_REALLOCATE = 1,
    ORC_WARNING_SINGULAR_MATRIX = 2,
    ORC_ERROR_INVALID_CALL = -2,
    ORC_ERROR_ARGUMENT_MISSING = -3,
    ORC_ERROR_INVALID_ARGUMENT = -4,
    ORC_ERROR_MAXIMUM_UPDATES = -5,
    ORC_ERROR_SINGULAR_UPDATE = -6,
    /* not in the reference: returned by orc_factorize when the debug hook
     * stop_after_pivots fired (test infrastructure for step-wise comparison) */
    ORC_STOPPED = 100,
    /* not in the reference: orc_blu_factorize met the endless Reallocate loop of defect D5 (W grown in the
     * middle of the bump; the reference would grow W until the process dies) and gave up */
    ORC_D5_TRAP = -98
};

/* Task (src/lu/def.rs:6-12) */
enum { ORC_TASK_NONE = 0, ORC_TASK_SINGLETONS, ORC_TASK_SETUP_BUMP, ORC_TASK_FACTORIZE_BUMP, ORC_TASK_BUILD_FACTORS };

/* struct LU (src/lu/lu.rs:9-171).  Option<usize> fields use -1 for None. */
typedef struct orc_lu {
    lu_int l_mem, u_mem, w_mem;
    double droptol, abstol, reltol;
    lu_int nzbias;      /* Option<usize>: -1 = None */
    lu_int maxsearch, pad;
    double stretch, compress_thres, sparse_thres;
    lu_int search_rows;

    lu_int m;
    lu_int addmem_l, addmem_u, addmem_w;
    lu_int nupdate;     /* Option<usize>: -1 = None */
    lu_int nforrest, nfactorize, nupdate_total, nforrest_total, nsymperm_total;
    lu_int l_nz, u_nz, r_nz;
    double min_pivot, max_pivot, max_eta;
    double update_cost_numer, update_cost_denom;
    double time_factorize, time_solve, time_update;
    double time_factorize_total, time_solve_total, time_update_total;
    lu_int l_flops, u_flops, r_flops;
    double condest_l, condest_u, norm_l, norm_u, normest_l_inv, normest_u_inv;
    double onenorm, infnorm, residual_test;

    lu_int matrix_nz, rank, bump_size, bump_nz;
    lu_int nsearch_pivot, nexpand, ngarbage, factor_flops;
    double time_singletons, time_search_pivot, time_elim_pivot;
    double pivot_error;

    int task;
    lu_int pivot_row, pivot_col; /* Option<usize>: -1 = None */
    lu_int ftran_for_update, btran_for_update;
    lu_int marker, pivotlen, rankdef, min_colnz, min_rownz;

    lu_int *l_index, *u_index, *w_index;
    double *l_value, *u_value, *w_value;

    lu_int *colcount_flink; /* alias pivotcol */
    lu_int *colcount_blink; /* alias pivotrow */
    lu_int *rowcount_flink; /* alias r_begin, eta_row (D13: same array, no offset) */
    lu_int *rowcount_blink; /* alias iwork1 (2m+2) */
    lu_int *w_begin;        /* [m+1..] alias l_begin */
    lu_int *w_end;          /* [m+1..] alias lt_begin */
    lu_int *w_flink;        /* [m+1..] alias lt_begin_p */
    lu_int *w_blink;        /* [m+1..] alias p */
    lu_int *pinv;           /* alias pmap */
    lu_int *qinv;           /* alias qmap */
    lu_int *l_begin_p, *u_begin;
    lu_int *iwork0;         /* alias marked */
    double *work0, *work1, *col_pivot, *row_pivot;

    /* --- not in the reference: test hook -------------------------------- */
    int fix_d3;       /* 0 (default): restate D3 faithfully (i32 cancellation mask);
                         1: use the 64-bit mask upstream BASICLU intends, so that
                         matrices on which the reference corrupts its row file and
                         panics can still be compared with the HIP path */
    lu_int d3_hits;   /* cancellations recorded at pivot-column position >= 32:
                         0 means the faithful and the fixed runs are identical */
    lu_int nunsymperm_total; /* updates done by an unsymmetric permutation (test hook; no counter in the reference) */
    lu_int npivot_kind[6]; /* pivots taken per path: 0 singleton row, 1 singleton col, 2 doubleton col,
                              3 small, 4 any, 5 empty column (factorize_bump.rs:24-33) */
    lu_int stop_after_pivots; /* <0: off; else factorize_bump returns
                                 ORC_STOPPED once rank+rankdef reaches it */
} orc_lu;

/* struct BLU (src/blu.rs:9-20) */
typedef struct orc_blu {
    orc_lu lu;
    double *lhs;
    lu_int *ilhs;
    lu_int nzlhs;
    double realloc_factor;
} orc_blu;

/* --- object API (src/blu.rs) -------------------------------------------- */
orc_blu *orc_blu_new(lu_int m, lu_int b_nz);                          /* blu.rs:61 */
void orc_blu_free(orc_blu *obj);
int orc_blu_factorize(orc_blu *obj, const uint64_t *b_begin, const uint64_t *b_end,
                      const uint64_t *b_i, const double *b_x);        /* blu.rs:95 */
int orc_blu_get_factors(orc_blu *obj, lu_int *rowperm, lu_int *colperm,
                        lu_int *l_colptr, lu_int *l_rowidx, double *l_value,
                        lu_int *u_colptr, lu_int *u_rowidx, double *u_value); /* blu.rs:139 */
int orc_blu_solve_dense(orc_blu *obj, const double *rhs, double *lhs, char trans); /* blu.rs:182 */
/* BLU::solve_sparse (blu.rs:207): the solution stays in the object, as in the reference
 * (obj.lhs dense, obj.ilhs[0..nzlhs) its pattern); the two functions below read it out. */
int orc_blu_solve_sparse(orc_blu *obj, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, char trans);
lu_int orc_blu_nzlhs(const orc_blu *obj);
void orc_blu_get_lhs(const orc_blu *obj, lu_int *ilhs, double *lhs);
orc_lu *orc_blu_lu(orc_blu *obj);
/* BLU::solve_for_update (blu.rs:257) / BLU::update (blu.rs:319): the INTENDED Forrest-Tomlin algorithm, NOT
 * reference-pinned (the reference is defective there: orc_update.c lists the repairs). */
int orc_blu_solve_for_update(orc_blu *obj, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, char trans, int want_solution);
int orc_blu_update(orc_blu *obj, double xtbl);
int orc_solve_for_update(orc_lu *lu, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, lu_int *p_nzlhs,
                         lu_int *ilhs, double *lhs, char trans);  /* solve_for_update.rs:73 */
int orc_update(orc_lu *lu, double xtbl);                          /* update.rs:49 */

/* --- procedural API (src/factorize.rs, get_factors.rs, solve_dense.rs) --- */
int orc_factorize(orc_lu *lu, const uint64_t *b_begin, const uint64_t *b_end,
                  const uint64_t *b_i, const double *b_x, int c0ntinue); /* factorize.rs:34 */
int orc_get_factors(orc_lu *lu, lu_int *rowperm, lu_int *colperm,
                    lu_int *l_colptr, lu_int *l_rowidx, double *l_value,
                    lu_int *u_colptr, lu_int *u_rowidx, double *u_value); /* get_factors.rs:48 */
int orc_solve_dense(orc_lu *lu, const double *rhs, double *lhs, char trans); /* solve_dense.rs:24 */
int orc_solve_sparse(orc_lu *lu, lu_int nzrhs, const uint64_t *irhs, const double *xrhs, lu_int *p_nzlhs,
                     lu_int *ilhs, double *lhs, char trans);                  /* solve_sparse.rs:36 */

/* --- stats / params by key (shared numbering with include/blu_hip.h) ----- */
double orc_get_stat(const orc_lu *lu, int key);
int orc_set_param(orc_lu *lu, int key, double v);

/* --- kernel layer, exposed for unit tests (src/lu/ *.rs) ------------------ */
void orc_list_init(lu_int *flink, lu_int *blink, lu_int nelem, lu_int nlist, lu_int *min_list);
void orc_list_add(lu_int elem, lu_int list, lu_int *flink, lu_int *blink, lu_int nelem, lu_int *min_list);
void orc_list_remove(lu_int *flink, lu_int *blink, lu_int elem);
void orc_list_move(lu_int elem, lu_int list, lu_int *flink, lu_int *blink, lu_int nelem, lu_int *min_list);
void orc_list_swap(lu_int *flink, lu_int *blink, lu_int e1, lu_int e2);
void orc_file_empty(lu_int nlines, lu_int *begin, lu_int *end, lu_int *next, lu_int *prev, lu_int fmem);
void orc_file_reappend(lu_int line, lu_int nlines, lu_int *begin, lu_int *end, lu_int *next, lu_int *prev,
                       lu_int *index, double *value, lu_int extra_space);
lu_int orc_file_compress(lu_int nlines, lu_int *begin, lu_int *end, const lu_int *next,
                         lu_int *index, double *value, double stretch, lu_int pad);
lu_int orc_file_diff(lu_int nrow, const lu_int *begin_row, const lu_int *end_row,
                     const lu_int *begin_col, const lu_int *end_col,
                     const lu_int *index, const double *value);

/* --- synthetic inputs -------------------------------------- */
/* lp_basis(m,k,bw,tri_frac,seed): writes colptr[m+1], rowidx[<=m*k], value.
 * Returns nnz.  SplitMix64, draw order documented in oracle/blu_oracle.c. */
lu_int orc_gen_lp_basis(lu_int m, lu_int k, lu_int bw, double tri_frac, double offscale, uint64_t seed,
                        uint64_t *colptr, uint64_t *rowidx, double *value);

#ifdef __cplusplus
}
#endif
#endif
