/*
 * blu_hip.h -- C ABI of libblu_hip.so: MI355X (gfx950) implementation of the
 * factorize hot path of the `blu` crate (rwl/blu v0.2.1).
 *
 * This is the drop-in boundary.  Each entry point replaces one method of the
 * reference's object API (`struct BLU`, /root/reference/src/blu.rs) and is what
 * a Rust `extern "C"` block in the crate would bind.
 * Plain pointers and sizes only; every array argument is HOST memory owned by
 * the caller and borrowed for the duration of the call.  No exceptions or
 * aborts cross the boundary: every function returns a status.
 *
 * A handle is not thread-safe (the reference takes `&mut self` everywhere);
 * distinct handles are independent and may live on different devices.
 */
#ifndef BLU_HIP_H
#define BLU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: enum Status, src/lib.rs:38-64 (numbers = upstream BASICLU) */
#define BLU_OK 0
#define BLU_REALLOCATE 1                /* never returned: handled inside, as BLU::factorize does (blu.rs:105-115) */
#define BLU_WARNING_SINGULAR_MATRIX 2   /* factorization valid, rank < m (factorize.rs:176-178) */
#define BLU_ERROR_INVALID_CALL (-2)
#define BLU_ERROR_ARGUMENT_MISSING (-3)
#define BLU_ERROR_INVALID_ARGUMENT (-4)
#define BLU_ERROR_MAXIMUM_UPDATES (-5)
#define BLU_ERROR_SINGULAR_UPDATE (-6)
#define BLU_ERROR_OUT_OF_MEMORY (-9)    /* blu.rs:46 doc: BLU_ERROR_OUT_OF_MEMORY */
#define BLU_ERROR_DEVICE (-100)         /* HIP runtime error / no gfx950 device / kernel trap */

/* ---- parameters: public fields of struct LU, src/lu/lu.rs:11-66, defaults lu.rs:249-259 */
enum blu_param {
    BLU_PARAM_DROPTOL = 0,      /* 1e-20 */
    BLU_PARAM_ABSTOL = 1,       /* 1e-14 */
    BLU_PARAM_RELTOL = 2,       /* 0.1   */
    BLU_PARAM_NZBIAS = 3,       /* Option<usize>: default Some(1); pass -1 for None */
    BLU_PARAM_MAXSEARCH = 4,    /* 3     */
    BLU_PARAM_PAD = 5,          /* 4     */
    BLU_PARAM_STRETCH = 6,      /* 0.3   */
    BLU_PARAM_COMPRESS_THRES = 7, /* 0.5 */
    BLU_PARAM_SPARSE_THRES = 8, /* 0.05  */
    BLU_PARAM_SEARCH_ROWS = 9,  /* 0 (lu.rs:259; the doc comment lu.rs:63-66 says 1, the code says 0) */
    BLU_PARAM_REALLOC_FACTOR = 10 /* BLU.realloc_factor, blu.rs:68: 1.5 */
};

/* ---- statistics: getters of struct LU, src/lu/lu.rs:398-684 */
enum blu_stat {
    BLU_STAT_M = 0,
    BLU_STAT_NUPDATE = 1,        /* -1 = None (no valid factorization) */
    BLU_STAT_NFACTORIZE = 2,
    BLU_STAT_L_NZ = 3,           /* lu.rs:466 */
    BLU_STAT_U_NZ = 4,           /* lu.rs:471 */
    BLU_STAT_MIN_PIVOT = 5,
    BLU_STAT_MAX_PIVOT = 6,
    BLU_STAT_CONDEST_L = 7,
    BLU_STAT_CONDEST_U = 8,
    BLU_STAT_NORM_L = 9,
    BLU_STAT_NORM_U = 10,
    BLU_STAT_NORMEST_L_INV = 11,
    BLU_STAT_NORMEST_U_INV = 12,
    BLU_STAT_ONENORM = 13,
    BLU_STAT_INFNORM = 14,
    BLU_STAT_RESIDUAL_TEST = 15,
    BLU_STAT_MATRIX_NZ = 16,     /* lu.rs:619 */
    BLU_STAT_RANK = 17,
    BLU_STAT_BUMP_SIZE = 18,
    BLU_STAT_BUMP_NZ = 19,
    BLU_STAT_NSEARCH_PIVOT = 20,
    BLU_STAT_NEXPAND = 21,       /* layout dependent: not part of the parity contract */
    BLU_STAT_NGARBAGE = 22,      /* layout dependent: not part of the parity contract */
    BLU_STAT_FACTOR_FLOPS = 23,  /* lu.rs:658 */
    BLU_STAT_TIME_FACTORIZE = 24,
    /* 25-27: device seconds of the last factorize: singleton phase (k_prep); Markowitz search and elimination run inside
     * one persistent kernel, whose time is split only by the diagnostic build with phase counters (otherwise
     * TIME_SEARCH_PIVOT = 0 and TIME_ELIM_PIVOT = the whole pivot kernel) */
    BLU_STAT_TIME_SINGLETONS = 25,
    BLU_STAT_TIME_SEARCH_PIVOT = 26,
    BLU_STAT_TIME_ELIM_PIVOT = 27,
    BLU_STAT_UPDATE_COST_DENOM = 28, /* factorize.rs:160-166 */
    BLU_STAT_RANKDEF = 29,
    BLU_STAT_L_MEM = 30,
    BLU_STAT_U_MEM = 31,
    BLU_STAT_W_MEM = 32,
    BLU_STAT_L_FLOPS = 33,       /* lu.l_flops, accumulated by solve_sparse (lu/solve_sparse.rs:356) */
    BLU_STAT_U_FLOPS = 34,       /* lu.u_flops (:357) */
    BLU_STAT_NFORREST = 35,      /* lu.nforrest: Forrest-Tomlin updates since the last factorize (lu.rs:92) */
    BLU_STAT_PIVOT_ERROR = 36,   /* lu.pivot_error of the last update (update.rs:942) */
    BLU_STAT_R_NZ = 37,          /* lu.r_nz: nonzeros in the row eta file */
    BLU_STAT_R_FLOPS = 38,       /* lu.r_flops (lu/solve_sparse.rs:358) */
    BLU_STAT_MAX_ETA = 39,       /* lu.max_eta */
    /* device-side extras (no reference counterpart) */
    BLU_STAT_DEV_TIME_PIVOT_LOOP = 40,  /* seconds, hipEvent, last factorize */
    BLU_STAT_DEV_TIME_TOTAL = 41,       /* seconds, hipEvent, all kernels of last factorize */
    BLU_STAT_DEV_RELAUNCHES = 42,       /* pivot-loop kernel launches of last factorize */
    /* 44..47: device seconds of k_prep / k_setup / k_finish / the statistics tail of the last factorize;
     * 108, 109: inside the statistics of a single factorize, k_rows_grid and k_stats_tail_a+b (diagnostic keys
     * without enum names, as 50..58 and 60..107 of the diagnostic build) */
    BLU_STAT_NSYMPERM_TOTAL = 48,       /* lu.nsymperm_total: updates done by a symmetric permutation alone */
    BLU_STAT_NFORREST_TOTAL = 49,       /* lu.nforrest_total */
    BLU_STAT_DEV_NUNSYMPERM_TOTAL = 59, /* updates done by an unsymmetric permutation (update.rs:794-814); no getter in the reference */
    BLU_STAT_UPDATE_COST = 124          /* LU::update_cost() = update_cost_numer / update_cost_denom (lu.rs:324-326) */
};

typedef struct blu_hip blu_hip; /* opaque: owns all device + host state (= struct LU + struct BLU) */

/* BLU::new(m, b_nz) -- src/blu.rs:61, LU::new src/lu/lu.rs:243.
 * `device` = HIP device ordinal.  Returns NULL on bad argument, missing gfx950
 * device or allocation failure. */
blu_hip *blu_hip_new(int64_t m, int64_t b_nz, int device);
void blu_hip_free(blu_hip *h);

/* public parameter fields of LU / BLU -- src/lu/lu.rs:11-66, src/blu.rs:18-20 */
int blu_hip_set_param(blu_hip *h, int key, double value);
double blu_hip_get_param(const blu_hip *h, int key);
/* getters -- src/lu/lu.rs:398-684 */
double blu_hip_get_stat(const blu_hip *h, int key);

/* BLU::factorize(&mut self, b_begin, b_end, b_i, b_x) -- src/blu.rs:95-118,
 * factorize() src/factorize.rs:34-182.  Column j of B holds
 * b_i[b_begin[j]..b_end[j]], b_x[...]; b_begin/b_end may overlap (CSC colptr,
 * colptr+1).  b_i_len = length of b_i and b_x (needed to size the upload; the
 * Rust slices carry it).  Returns BLU_OK, BLU_WARNING_SINGULAR_MATRIX,
 * BLU_ERROR_INVALID_ARGUMENT (src/lu/singletons.rs:122-201), or a device error. */
int blu_hip_factorize(blu_hip *h, const uint64_t *b_begin, const uint64_t *b_end,
                      const uint64_t *b_i, const double *b_x, uint64_t b_i_len);

/* Same, with B already resident in device memory (hipMalloc'ed on the handle's
 * device).  This is the entry bench.py times: inputs in HBM when the clock starts. */
int blu_hip_factorize_device(blu_hip *h, const uint64_t *d_b_begin, const uint64_t *d_b_end,
                             const uint64_t *d_b_i, const double *d_b_x, uint64_t b_i_len);

/* BLU::get_factors -- src/blu.rs:139, get_factors() src/get_factors.rs:48-180.
 * Any NULL pointer skips that output (all three of an L or U triple must be
 * non-NULL for the triple to be written, get_factors.rs:73,122).
 * Sizes: rowperm[m], colperm[m], l_colptr[m+1], l_rowidx/l_value[m+l_nz],
 * u_colptr[m+1], u_rowidx/u_value[m+u_nz]. */
int blu_hip_get_factors(blu_hip *h, int64_t *rowperm, int64_t *colperm,
                        int64_t *l_colptr, int64_t *l_rowidx, double *l_value,
                        int64_t *u_colptr, int64_t *u_rowidx, double *u_value);

/* BLU::solve_dense -- src/blu.rs:182, src/lu/solve_dense.rs:7-120.
 * rhs and lhs may be the same array (solve_dense.rs doc, lines 14-16). */
int blu_hip_solve_dense(blu_hip *h, const double *rhs, double *lhs, char trans);

/* solve_sparse -- src/solve_sparse.rs:36-68 (BLU::solve_sparse, src/blu.rs:207, keeps the same three
 * outputs inside the object), lu/solve_sparse.rs:11-360 with solve_symbolic.rs, dfs.rs and
 * solve_triangular.rs.  Right-hand side in compressed form: irhs[0..nzrhs) (no duplicates),
 * xrhs[0..nzrhs).  `lhs` (m doubles) must be all zero on entry, as the reference requires
 * (solve_sparse.rs:23-24); on return the solution is scattered into it, *p_nzlhs is its number of
 * nonzeros and ilhs[0..*p_nzlhs) their indices IN THE REFERENCE'S ORDER (the topological order its
 * depth-first searches produce, or pivot order when the sequential branch runs; parameters
 * BLU_PARAM_SPARSE_THRES and BLU_PARAM_DROPTOL decide as in the reference).  ilhs must have room for m.
 * Returns BLU_ERROR_INVALID_CALL without a valid factorization, BLU_ERROR_INVALID_ARGUMENT for
 * nzrhs < 0, nzrhs > m or an index out of range.  Works on fresh and on updated factorizations. */
int blu_hip_solve_sparse(blu_hip *h, int64_t nzrhs, const uint64_t *irhs, const double *xrhs,
                         int64_t *p_nzlhs, int64_t *ilhs, double *lhs, char trans);

/* solve_for_update -- src/solve_for_update.rs:73-119 (BLU::solve_for_update, src/blu.rs:257-288, keeps the
 * outputs inside the object), lu/solve_for_update.rs:12-455.  Prepares an update of the factorization:
 *   trans 't'/'T': the column to be REPLACED is irhs[0] (nzrhs, xrhs and irhs[1..] are not read); computes and
 *                  stores the row eta (partial solve with U').
 *   otherwise:     irhs[0..nzrhs) / xrhs[0..nzrhs) (no duplicates) is the column to be INSERTED; computes and
 *                  stores the spike (solve with L and the row etas).
 * If p_nzlhs, ilhs and lhs are all non-NULL the solution of the system is completed and returned as by
 * blu_hip_solve_sparse (lhs all zero on entry); otherwise only the update is prepared.  Returns
 * BLU_ERROR_INVALID_CALL without a valid factorization, BLU_ERROR_MAXIMUM_UPDATES after m Forrest-Tomlin
 * updates, BLU_ERROR_INVALID_ARGUMENT for indices out of range, BLU_ERROR_ARGUMENT_MISSING for a forward
 * solve without xrhs.  Status::Reallocate is handled inside, as BLU::solve_for_update does.
 * The reference's own code is defective on this path; what is implemented is the
 * algorithm it documents (blu_amd/csrc/k_update.hip, repairs listed in oracle/orc_update.c). */
int blu_hip_solve_for_update(blu_hip *h, int64_t nzrhs, const uint64_t *irhs, const double *xrhs,
                             int64_t *p_nzlhs, int64_t *ilhs, double *lhs, char trans);

/* update -- src/update.rs:49-55 (BLU::update, src/blu.rs:319-335), lu/update.rs:388-959: replaces the column
 * named by the last transposed blu_hip_solve_for_update by the column given to the last forward one
 * (Forrest-Tomlin update, or a pure permutation update when the spiked U is still permuted triangular).
 * xtbl = element jpivot of the forward solution (stability monitor only: BLU_STAT_PIVOT_ERROR).  Returns
 * BLU_ERROR_INVALID_CALL unless both solves were done, BLU_ERROR_SINGULAR_UPDATE if the new pivot is zero or
 * below abstol (the old factorization stays valid).  Afterwards BLU_STAT_NUPDATE is one higher,
 * blu_hip_get_factors answers BLU_ERROR_INVALID_CALL (get_factors.rs:59) and the solves work on the updated
 * factorization. */
int blu_hip_update(blu_hip *h, double xtbl);

/* Batch extension (no reference counterpart; the reference's only parallel
 * axis is independent BLU objects).  Factorizes n handles that
 * live on the same device concurrently: every kernel is launched once for the batch, the pivot loop with ONE WAVE
 * per handle (k_pivot_loop_wave; throughput grows with n up to 16 handles per CU, 4096 per MI355X) -- with TWO waves
 * per handle while n is at most half of that (k_pivot_loop_wave2: large bases, where HBM capacity limits n).  Matrix k is
 * given by the k-th pointers.  status[k] receives the per-handle status (also when the call as a whole
 * is refused: then every status[k] carries the refusal and no handle keeps usable factors).  The same
 * handle may not appear twice and all handles must live on one device (BLU_ERROR_INVALID_ARGUMENT).
 * Parameters and blu_hip_set_skip_stats are honoured per handle; the workgroup size of the batch
 * (a debug knob) is taken from h[0].  Device memory: besides what the handles own, a call of at least one handle per
 * CU borrows, for its duration, 20 bytes per U entry of its largest member for every CU (7.4 GB for bases of the
 * 100k size) when that leaves at least 1 GB free; without it the same results take a little longer. */
int blu_hip_factorize_batch(blu_hip **h, int n,
                            const uint64_t *const *b_begin, const uint64_t *const *b_end,
                            const uint64_t *const *b_i, const double *const *b_x,
                            const uint64_t *b_i_len, int inputs_on_device, int *status);

/* factorize() ends with the statistics tail of src/factorize.rs:121-147 (condest(L), condest(U),
 * residual_test; getters BLU_STAT_CONDEST_* .. BLU_STAT_RESIDUAL_TEST).  It is a chain of 8 triangular
 * sweeps (~17 % of the factorize time at 100k); a caller that never reads those getters can switch
 * it off (on != 0): the keys then return 0.  Default: computed, as the reference does. */
int blu_hip_set_skip_stats(blu_hip *h, int on);

/* Synthetic LP basis used by the benchmark and the tests (host utility, no device
 * work): CSC out, colptr[m+1], rowidx/value[<= m*k].  Returns nnz. */
int64_t blu_hip_gen_lp_basis(int64_t m, int64_t k, int64_t bw, double tri_frac, double offscale,
                             uint64_t seed, uint64_t *colptr, uint64_t *rowidx, double *value);

/* Library/device introspection */
const char *blu_hip_version(void);
int blu_hip_device_count(void);
/* Text of the last HIP/runtime error seen by this handle ("" if none). */
const char *blu_hip_last_error(const blu_hip *h);

#ifdef __cplusplus
}
#endif
#endif
