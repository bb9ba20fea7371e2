// k_update.hip -- the update path on the device: solve_for_update, the Forrest-Tomlin update with
// reordering, and the solves on an updated factorization.
//
//   solve_for_update   src/lu/solve_for_update.rs:12-455   (API: src/solve_for_update.rs:73, BLU::solve_for_update blu.rs:257)
//   update             src/lu/update.rs:388-959            (API: src/update.rs:49, BLU::update blu.rs:319)
//   solve_sparse / solve_dense with nforrest > 0           src/lu/solve_sparse.rs:11-360, src/lu/solve_dense.rs:7-120
//
// The reference is defective on this path (an update either panics or leaves factors
// whose solves are wrong), so there is nothing to be bit-exact WITH.  What is implemented is the algorithm the
// reference documents (comments update.rs:378-387, 467-483, 609-666, 695-709, 750-755), with the repairs listed
// in oracle/orc_update.c; the tests validate it by backward error against the modified matrix and by
// refactorize-and-compare, and additionally require bit-identical results to that CPU restatement of the same
// intended algorithm (same operation order everywhere).
//
// Layout (one UpdWs per handle, built from the fresh factors by k_upd_init at the first solve_for_update):
//   * U twice, both mutable: the ROW file  wbeg/wlen/wcap[j] -> widx/wval   (row of U whose pivot is in column j;
//     column indices), and the COLUMN file ucbeg/uclen[i] -> ucidx/ucval    (column of U whose pivot is in row i;
//     row indices).  Bump-pointer arenas with per-line capacity instead of the reference's gap-separated files with
//     -1 terminators and a memory-order list: results do not depend on that layout (a reappended line keeps its
//     entry order), only the order of entries inside a line matters and that follows the reference statement by
//     statement (append at the end, "last entry into the hole" deletions).
//   * pmap/qmap, col_pivot/row_pivot, the pivot sequence pvrow/pvcol[2m] (with garbage_perm, garbage_perm.rs:16-48),
//     the row etas rbeg/eta_row -> ridx/rval.
//   * L is never modified: the stage-ordered columns and the row-wise copy of k_solve_sparse.hip are used as they are.
// Everything here is ONE wave per handle: depth-first searches, the breadth-first search for the augmenting path
// and the list surgery of an update are serial pointer chases; the numerical substitutions run with the lanes on
// the entries of one column.  This path is latency-bound, not bandwidth-bound.
#include "blu_dev.h"

enum { UPD_OK = 0, UPD_NEED_R = 1, UPD_NEED_UC = 2, UPD_NEED_W = 3, UPD_SINGULAR = 4, UPD_ERROR = 5 };

struct UpdState {
    int status, need, err_line;
    int nforrest, pivotlen;
    int u_nz, r_nz;
    int wused, ucused;
    int btran_for, ftran_for; // column to be replaced (-1 = none) / 1 after the forward solve (-1 = none)
    int spike_beg, spike_len; // pending spike in the column-file arena
    int pad0;
    long long l_flops, u_flops, r_flops;
    long long nsymperm_total, nunsymperm_total, nforrest_total;
    double min_pivot, max_pivot, max_eta, pivot_error, update_cost_numer;
};

struct UpdWs {
    UpdState *st;
    int *pmap, *qmap;
    double *col_pivot, *row_pivot;
    int *wbeg, *wlen, *wcap, *widx;
    double *wval;
    int *ucbeg, *uclen, *ucidx;
    double *ucval;
    int *rbeg, *eta_row, *ridx;
    double *rval;
    int *pvrow, *pvcol; // 2m each
    int *iw1, *iw2;     // m each: path / reach, row_reach / col_reach
    double *work1;      // m: scattered row eta
    int wcapacity, uccapacity, rcapacity;
};

#define UPD_CHECK(st, cond)                  \
    do {                                     \
        if (!(cond) && (st)->status == 0) {  \
            (st)->status = UPD_ERROR;        \
            (st)->err_line = __LINE__;       \
        }                                    \
    } while (0)

// ---- the two mutable graphs of U ----------------------------------------------------------------------
struct GraphUc { // column of U under pivot row i: row indices
    static constexpr bool FILTER = false;
    const int *beg, *len, *idx;
    const double *v, *piv;
    __device__ __forceinline__ int begin(int i) const { return beg[i]; }
    __device__ __forceinline__ int end(int i) const { return beg[i] + len[i]; }
    __device__ __forceinline__ int node(int p) const { return idx[p]; }
    __device__ __forceinline__ double val(int p) const { return v[p]; }
    __device__ __forceinline__ double pivot(int i) const { return piv[i]; }
};
typedef GraphUc GraphWr; // row of U under pivot column j: column indices (same shape, other arrays)

// ordered append of the lanes' candidates: pattern[nz ..] gets the indices of the lanes with `take`, in lane order
__device__ __forceinline__ int wave_append(int *pattern, int nz, bool take, int i)
{
    const unsigned long long b = __ballot(take);
    if (take) pattern[nz + wave_prefix_count(b)] = i;
    return nz + __popcll(b);
}

// ---------------------------------------------------------------------------------------------------------
// k_upd_init: the mutable copies, from the fresh factors (build_factors.rs:283-419 gives the same contents)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_upd_init(DevLU *Ds, FinishOut *Os, UpdWs U)
{
    const DevG D(Ds[0]);
    const FinishOut &O = Os[0];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = D.m, rank = D.s->rank;
    __shared__ int sh[40];
    UpdState *st = U.st;
    typedef GPTR(const long long) gcll;
    const gcll ucp = (gcll)O.u_colptr, uri = (gcll)O.u_rowidx;
    GPTR(const double) uvl = (GPTR(const double))O.u_value;

    for (int k = tid; k < m; k += nt) {
        const int i = D.prow[k], j = D.pcol[k];
        U.pmap[j] = i;
        U.qmap[i] = j;
        U.pvrow[k] = i;
        U.pvcol[k] = j;
        const double piv = uvl[ucp[k + 1] - 1]; // unit pivots for dependent columns included (build_factors.rs:221)
        U.col_pivot[j] = piv;
        U.row_pivot[i] = piv;
        // column file: the canonical column without its pivot, rows ascending in pivot order (:354-384)
        const int b = (int)ucp[k] - k, n = (int)(ucp[k + 1] - ucp[k]) - 1;
        U.ucbeg[i] = b;
        U.uclen[i] = n;
        for (int q = 0; q < n; q++) {
            U.ucidx[b + q] = D.prow[(int)uri[ucp[k] + q]];
            U.ucval[b + q] = uvl[ucp[k] + q];
        }
    }
    // row file: stage rows in production order, entries in columns without a pivot left out (:286-351), each
    // line with room for stretch*nz + pad more entries
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int k = c0 + tid;
        int nz = 0;
        if (k < rank)
            for (int p = D.ubeg[k]; p < D.ubeg[k + 1]; p++) nz += D.qinv[D.uidx[p]] < rank;
        const int cap = k < m ? nz + stretch_of(D.stretch, nz) + D.pad : 0;
        int tot;
        const int ex = block_excl_scan_i(cap, sh, &tot);
        if (k < m) {
            const int j = D.pcol[k];
            int put = base + ex;
            U.wbeg[j] = put;
            U.wlen[j] = nz;
            U.wcap[j] = cap;
            if (k < rank)
                for (int p = D.ubeg[k]; p < D.ubeg[k + 1]; p++) {
                    const int jj = D.uidx[p];
                    if (D.qinv[jj] < rank) {
                        U.widx[put] = jj;
                        U.wval[put] = D.uval[p];
                        put++;
                    }
                }
        }
        base += tot;
    }
    if (tid == 0) {
        UpdState z;
        memset(&z, 0, sizeof z);
        z.pivotlen = m;
        z.u_nz = (int)ucp[m] - m;
        z.wused = base;
        z.ucused = (int)ucp[m] - m;
        z.btran_for = z.ftran_for = -1;
        z.min_pivot = D.s->min_pivot;
        z.max_pivot = D.s->max_pivot;
        *st = z;
        U.rbeg[0] = 0;
    }
}

// garbage_perm (garbage_perm.rs:16-48): keep the last occurrence of every column in the pivot sequence, order kept.
// Whole wave, 64 positions at a time from the back: a position is kept if its column has not been seen further back
// (marked) and no later position of the same chunk has it (the later one wins the atomicMax on lastpos[column]).
// The compacted sequence grows downwards from `put`, which never reaches the positions still to be read.
__device__ __forceinline__ void garbage_perm_wave(const UpdWs &U, int m, int *marked, int M, int *lastpos)
{
    UpdState *st = U.st;
    const int lane = lane_id();
    const int pivotlen = st->pivotlen;
    if (pivotlen <= m) return;
    int put = pivotlen;
    for (int hi = pivotlen - 1; hi >= 0; hi -= 64) {
        const int get = hi - lane;
        int j = 0, r = 0;
        bool cand = false;
        if (get >= 0) {
            j = U.pvcol[get];
            r = U.pvrow[get];
            cand = marked[j] != M;
        }
        if (cand) __hip_atomic_store(&lastpos[j], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_mem_sync();
        if (cand) (void)__hip_atomic_fetch_max(&lastpos[j], get, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_mem_sync();
        const bool keep = cand && __hip_atomic_load(&lastpos[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == get;
        const unsigned long long kb = __ballot(keep);
        if (keep) {
            const int d = put - 1 - wave_prefix_count(kb); // lane 0 holds the latest position
            U.pvcol[d] = j;
            U.pvrow[d] = r;
            marked[j] = M;
        }
        put -= __popcll(kb);
        wave_mem_sync();
    }
    if (lane == 0) UPD_CHECK(st, put + m == pivotlen);
    for (int k0 = 0; k0 < m; k0 += 64) { // (ascending: source index >= destination index; a chunk is read before it is written)
        const int k = k0 + lane;
        int c = 0, r = 0;
        if (k < m) {
            c = U.pvcol[put + k];
            r = U.pvrow[put + k];
        }
        wave_mem_sync();
        if (k < m) {
            U.pvcol[k] = c;
            U.pvrow[k] = r;
        }
        wave_mem_sync();
    }
    if (lane == 0) st->pivotlen = m;
    wave_mem_sync();
}

// ---------------------------------------------------------------------------------------------------------
// k_solve_upd: solve_sparse on an updated factorization (mode 0) and solve_for_update (mode 1), both systems.
// out[0] nz of the solution, [1] l_flops, [2] u_flops, [3] branch (1 sparse, 2 sequential), [4] r_flops
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_solve_upd(DevLU *Ds, SparseWs W, UpdWs U, int mode, int want_solution, int nrhs, const int *irhs,
                                                  const double *xrhs, int trans, int marker, int nz_sparse)
{
    __shared__ DfsRing dfs_ring;
    const DevG D(Ds[0]);
    const int lane = lane_id();
    const int m = D.m;
    const double droptol = D.droptol;
    UpdState *st = U.st;
    const int nforrest = st->nforrest;
    long long l_flops = 0, u_flops = 0, r_flops = 0;
    int nz = 0, branch = 1, top, nz_symb, M;
    const GraphWr GW{U.wbeg, U.wlen, U.widx, U.wval, U.col_pivot};
    const GraphUc GU{U.ucbeg, U.uclen, U.ucidx, U.ucval, U.row_pivot};
    const GraphL GL{D.pinv, D.lbeg, D.lidx, D.lval};
    const GraphLt GT{W.lt_ptr, W.lt_idx, W.lt_val};
    if (lane == 0) st->status = UPD_OK;
    bool have_solution = true;

    if (trans) {
        // =========== transposed system ===========
        if (mode == 1) {
            // ---- row eta (solve_for_update.rs:70-135): U' x = (row jpivot of U), no dropping
            const int jpivot = irhs[0];
            const int ipivot = U.pmap[jpivot];
            const int jb = U.wbeg[jpivot], je = jb + U.wlen[jpivot];
            M = marker + 1;
            top = solve_symbolic(GW, m, je - jb, U.widx + jb, W, M, &dfs_ring);
            nz_symb = m - top;
            const int rput = U.rbeg[nforrest];
            if (U.rcapacity - rput < nz_symb) { // not enough room for the row eta: the host grows it and calls again
                if (lane == 0) {
                    st->status = UPD_NEED_R;
                    st->need = nz_symb;
                }
                return;
            }
            for (int p = jb + lane; p < je; p += 64) W.work[U.widx[p]] = U.wval[p];
            wave_mem_sync();
            (void)solve_triangular<true>(GW, nz_symb, W.psym + top, 0.0, W.work, W.pat, u_flops);
            // compress the row eta into R, pattern mapped from column to row indices; the SYMBOLIC pattern is
            // kept (update's triangularity test needs it)
            for (int t = top + lane; t < m; t += 64) {
                const int j = W.psym[t];
                U.ridx[rput + t - top] = U.pmap[j];
                U.rval[rput + t - top] = W.work[j];
                W.work[j] = 0.0;
            }
            if (lane == 0) {
                U.rbeg[nforrest + 1] = rput + nz_symb;
                U.eta_row[nforrest] = ipivot;
                st->btran_for = jpivot;
            }
            wave_mem_sync();
            if (!want_solution) {
                have_solution = false;
            } else {
                // scatter the row eta into xlhs, scaled to the solution of U^{-T} e_jpivot; small entries dropped
                M = marker + 2;
                const double pivot = U.col_pivot[jpivot];
                const double xdrop = droptol * fabs(pivot);
                if (lane == 0) {
                    W.pat[0] = ipivot;
                    W.marked[ipivot] = M;
                    W.xlhs[ipivot] = 1.0 / pivot;
                }
                nz = 1;
                for (int c = 0; c < nz_symb; c += 64) {
                    const int p = rput + c + lane;
                    const bool v = c + lane < nz_symb;
                    const double x = v ? U.rval[p] : 0.0;
                    const bool take = v && fabs(x) > xdrop;
                    const int i = v ? U.ridx[p] : 0;
                    if (take) {
                        W.marked[i] = M;
                        W.xlhs[i] = -x / pivot;
                    }
                    nz = wave_append(W.pat, nz, take, i);
                }
                wave_mem_sync();
            }
        } else {
            // ---- solve_sparse: U' x = rhs (solve_sparse.rs:51-106)
            M = marker + 1;
            top = solve_symbolic(GW, m, nrhs, irhs, W, M, &dfs_ring);
            for (int n = lane; n < nrhs; n += 64) W.work[irhs[n]] = xrhs[n];
            wave_mem_sync();
            nz = solve_triangular<true>(GW, m - top, W.psym + top, droptol, W.work, W.pat, u_flops);
            M = marker + 2;
            for (int n = lane; n < nz; n += 64) {
                const int j = W.pat[n], i = U.pmap[j];
                W.pat[n] = i;
                W.xlhs[i] = W.work[j];
                W.work[j] = 0.0;
                W.marked[i] = M;
            }
            wave_mem_sync();
        }
        if (have_solution) {
            // ---- update etas backwards, fill-in appended to the pattern (:108-125 / solve_for_update.rs:167-185)
            for (int t = nforrest - 1; t >= 0; t--) {
                const double x = W.xlhs[U.eta_row[t]];
                if (x != 0.0) {
                    const int b = U.rbeg[t], e = U.rbeg[t + 1];
                    for (int c = b; c < e; c += 64) {
                        const int p = c + lane;
                        const bool v = p < e;
                        const int i = v ? U.ridx[p] : 0;
                        const bool fresh = v && W.marked[i] != M;
                        if (fresh) W.marked[i] = M;
                        nz = wave_append(W.pat, nz, fresh, i);
                        if (v) W.xlhs[i] = __dsub_rn(W.xlhs[i], __dmul_rn(x, U.rval[p]));
                    }
                    r_flops += e - b;
                    wave_mem_sync();
                }
            }
            // ---- L' (:127-179 / solve_for_update.rs:187-245)
            if (nz <= nz_sparse) {
                M = marker + 3;
                top = solve_symbolic(GT, m, nz, W.pat, W, M, &dfs_ring);
                nz = solve_triangular<false>(GT, m - top, W.psym + top, droptol, W.xlhs, W.ilhs, l_flops);
            } else {
                branch = 2;
                nz = 0;
                sweep_nonzeros_desc(m, W.xlhs, [&](int k) { return D.prow[k]; }, [&](int, int ipivot, double x) {
                    const int b = GT.begin(ipivot), e = GT.end(ipivot);
                    for (int p = b + lane; p < e; p += 64) {
                        const int i = GT.node(p);
                        W.xlhs[i] = __dsub_rn(W.xlhs[i], __dmul_rn(x, GT.val(p)));
                    }
                    l_flops += e - b;
                    wave_mem_sync();
                    if (fabs(x) > droptol) {
                        if (lane == 0) W.ilhs[nz] = ipivot;
                        nz++;
                    } else if (lane == 0) {
                        W.xlhs[ipivot] = 0.0;
                    }
                });
            }
        }
    } else {
        // =========== forward system ===========
        // ---- L (solve_sparse.rs:180-243 / solve_for_update.rs:257-310)
        M = marker + 1;
        top = solve_symbolic(GL, m, nrhs, irhs, W, M, &dfs_ring);
        nz_symb = m - top;
        for (int n = lane; n < nrhs; n += 64) W.work[irhs[n]] = xrhs[n];
        wave_mem_sync();
        nz = solve_triangular<false>(GL, nz_symb, W.psym + top, droptol, W.work, W.pat, l_flops);
        if (nz < nz_symb && lane == 0) { // unmark cancellation
            int t = top, n = 0;
            while (n < nz) {
                const int i = W.psym[t];
                if (i == W.pat[n]) n++;
                else W.marked[i] -= 1;
                t++;
            }
            while (t < m) {
                W.marked[W.psym[t]] -= 1;
                t++;
            }
        }
        wave_mem_sync();
        // ---- update etas, fill-in appended to the pattern (:245-262 / :312-329); the dot in storage order
        for (int t = 0; t < nforrest; t++) {
            const int ipivot = U.eta_row[t];
            const int b = U.rbeg[t], e = U.rbeg[t + 1];
            double x = 0.0; // the same on every lane
            // the ordered sum of solve_for_update.rs:312-329, without its zero terms: the vector is a sparse spike, a row eta
            // of a banded basis has tens of thousands of entries, and x + (+-0.0) == x bit for bit (x starts as +0.0 and a
            // round-to-nearest sum is -0.0 only if both operands are) -- so only the lanes with a nonzero product take part,
            // in lane order (was: 64 dependent adds through ds_bpermute per chunk, most of the forward solve's time)
            for (int c = b; c < e; c += 64) {
                const int p = c + lane;
                const int pp = p < e ? p : b; // (every lane loads: no load under a branch)
                const double w = W.work[U.ridx[pp]], rv = U.rval[pp];
                const double term = p < e ? __dmul_rn(w, rv) : 0.0;
                unsigned long long nzb = __ballot(term != 0.0); // (a NaN counts as nonzero)
                while (nzb) {
                    const int q = __ffsll((long long)nzb) - 1;
                    nzb &= nzb - 1;
                    x = __dadd_rn(x, wave_bcast_d(term, q));
                }
            }
            const bool fresh = x != 0.0 && W.marked[ipivot] != M;
            wave_mem_sync();
            if (lane == 0) {
                W.work[ipivot] = __dsub_rn(W.work[ipivot], x);
                if (fresh) {
                    W.marked[ipivot] = M;
                    W.pat[nz] = ipivot;
                }
            }
            if (fresh) nz++;
            wave_mem_sync();
        }
        if (nforrest > 0) r_flops += U.rbeg[nforrest] - U.rbeg[0];
        if (mode == 1) {
            // ---- compress the spike into the column-file arena (solve_for_update.rs:331-355); update() takes it from there
            const int put = st->ucused;
            if (U.uccapacity - put < nz + 1) {
                for (int n = lane; n < nz; n += 64) W.work[W.pat[n]] = 0.0;
                if (lane == 0) {
                    st->status = UPD_NEED_UC;
                    st->need = nz + 1;
                }
                return;
            }
            for (int n = lane; n < nz; n += 64) {
                const int i = W.pat[n];
                U.ucidx[put + n] = i;
                U.ucval[put + n] = W.work[i];
                if (!want_solution) W.work[i] = 0.0;
            }
            if (lane == 0) {
                st->spike_beg = put;
                st->spike_len = nz;
                st->ftran_for = 1;
            }
            wave_mem_sync();
            if (!want_solution) have_solution = false;
        }
        if (have_solution) {
            // ---- U (:264-334 / solve_for_update.rs:363-433)
            if (nz <= nz_sparse) {
                M = marker + 2;
                top = solve_symbolic(GU, m, nz, W.pat, W, M, &dfs_ring);
                nz = solve_triangular<true>(GU, m - top, W.psym + top, droptol, W.work, W.ilhs, u_flops);
                for (int n = lane; n < nz; n += 64) { // permute into xlhs; the pattern goes from row to column indices
                    const int i = W.ilhs[n], j = U.qmap[i];
                    W.ilhs[n] = j;
                    W.xlhs[j] = W.work[i];
                    W.work[i] = 0.0;
                }
                wave_mem_sync();
            } else { // sequential solve over the pivot sequence (duplicates allowed: a row already done holds zero)
                branch = 2;
                nz = 0;
                // (a row that occurs twice in the sequence holds zero when its earlier position is reached)
                sweep_nonzeros_desc(st->pivotlen, W.work, [&](int k) { return U.pvrow[k]; }, [&](int k, int ipivot, double w) {
                    const int jpivot = U.pvcol[k];
                    const double x = w / U.row_pivot[ipivot];
                    if (lane == 0) W.work[ipivot] = 0.0;
                    const int b = GU.begin(ipivot), e = GU.end(ipivot);
                    for (int p = b + lane; p < e; p += 64) {
                        const int i = GU.node(p);
                        W.work[i] = __dsub_rn(W.work[i], __dmul_rn(x, GU.val(p)));
                    }
                    u_flops += e - b;
                    if (fabs(x) > droptol) {
                        if (lane == 0) {
                            W.ilhs[nz] = jpivot;
                            W.xlhs[jpivot] = x;
                        }
                        nz++;
                    }
                });
            }
        }
    }
    // hand the solution out in compressed form and restore the all-zero invariant of xlhs
    if (!have_solution) nz = 0;
    for (int n = lane; n < nz; n += 64) {
        const int j = W.ilhs[n];
        W.xval[n] = W.xlhs[j];
        W.xlhs[j] = 0.0;
    }
    if (lane == 0) {
        W.out[0] = nz;
        W.out[1] = l_flops;
        W.out[2] = u_flops;
        W.out[3] = branch;
        st->l_flops += l_flops;
        st->u_flops += u_flops;
        st->r_flops += r_flops;
        st->update_cost_numer += (double)r_flops;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_update: insert the spike into U and restore triangularity (update.rs:388-959).  One wave; lane 0 does the
// list surgery, the searches run through the shared graph routines.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_in(const int *idx, int beg, int end, int key) // position of key in idx[beg..end) or end
{
    while (beg < end && idx[beg] != key) beg++;
    return beg;
}

// bfs_path (update.rs:51-105): a path j0 -> ... -> j0 in the row-file graph; nodes into jlist[top..m).  One lane.
__device__ __forceinline__ int bfs_path_lane0(const UpdWs &U, int m, int j0, int *jlist, int *marked, int *queue)
{
    int j = -1, tail = 1, top = m;
    bool found = false;
    queue[0] = j0;
    for (int front = 0; front < tail && !found; front++) {
        j = queue[front];
        const int b = U.wbeg[j], e = b + U.wlen[j];
        for (int pos = b; pos < e; pos++) {
            const int k = U.widx[pos];
            if (k == j0) {
                found = true;
                break;
            }
            if (marked[k] >= 0) {      // not in the queue yet
                marked[k] = -j - 1;    // parent[k] = j
                queue[tail++] = k;
            }
        }
    }
    if (found) {
        int guard = 0;
        while (j != j0 && guard++ <= m) {
            jlist[--top] = j;
            j = -marked[j] - 1;
        }
        jlist[--top] = j0;
    }
    for (int pos = 0; pos < tail; pos++) marked[queue[pos]] = 0;
    return top;
}

// permute (update.rs:176-314): the row-column mappings of the nswap+1 path nodes move round by one.  One lane.
__device__ __forceinline__ void permute_lane0(const UpdWs &U, const int *jlist, int nswap)
{
    UpdState *st = U.st;
    const int j0 = jlist[0], jn = jlist[nswap];
    const int i0 = U.pmap[j0], in_ = U.pmap[jn];
    UPD_CHECK(st, nswap >= 1 && U.qmap[i0] == j0 && U.qmap[in_] == jn && U.row_pivot[i0] == 0.0 && U.col_pivot[j0] == 0.0);
    // row file
    const int begn = U.wbeg[jn], lenn = U.wlen[jn], capn = U.wcap[jn];
    const double piv = U.col_pivot[jn];
    for (int n = nswap; n > 0; n--) {
        const int j = jlist[n], jprev = jlist[n - 1];
        U.wbeg[j] = U.wbeg[jprev];
        U.wlen[j] = U.wlen[jprev];
        U.wcap[j] = U.wcap[jprev];
        const int e = U.wbeg[j] + U.wlen[j];
        const int where = find_in(U.widx, U.wbeg[j], e, j);
        UPD_CHECK(st, where < e);
        if (where >= e) return;
        if (n > 1) {
            U.widx[where] = jprev;
            U.col_pivot[j] = U.wval[where];
            U.wval[where] = U.col_pivot[jprev];
        } else {
            U.col_pivot[j] = U.wval[where];
            U.wlen[j] -= 1;
            U.widx[where] = U.widx[e - 1];
            U.wval[where] = U.wval[e - 1];
        }
        UPD_CHECK(st, U.col_pivot[j] != 0.0);
        st->min_pivot = fmin(st->min_pivot, fabs(U.col_pivot[j]));
        st->max_pivot = fmax(st->max_pivot, fabs(U.col_pivot[j]));
    }
    U.wbeg[j0] = begn;
    U.wlen[j0] = lenn;
    U.wcap[j0] = capn;
    {
        const int e = begn + lenn;
        const int where = find_in(U.widx, begn, e, j0);
        UPD_CHECK(st, where < e);
        if (where >= e) return;
        U.widx[where] = jn;
        U.col_pivot[j0] = U.wval[where];
        UPD_CHECK(st, U.col_pivot[j0] != 0.0);
        U.wval[where] = piv;
        st->min_pivot = fmin(st->min_pivot, fabs(U.col_pivot[j0]));
        st->max_pivot = fmax(st->max_pivot, fabs(U.col_pivot[j0]));
    }
    // column file
    const int cbeg0 = U.ucbeg[i0], clen0 = U.uclen[i0];
    for (int n = 0; n < nswap; n++) {
        const int i = U.pmap[jlist[n]], inext = U.pmap[jlist[n + 1]];
        U.ucbeg[i] = U.ucbeg[inext];
        U.uclen[i] = U.uclen[inext];
        const int e = U.ucbeg[i] + U.uclen[i];
        const int where = find_in(U.ucidx, U.ucbeg[i], e, i);
        UPD_CHECK(st, where < e);
        if (where >= e) return;
        U.ucidx[where] = inext;
        U.row_pivot[i] = U.ucval[where];
        UPD_CHECK(st, U.row_pivot[i] != 0.0);
        U.ucval[where] = U.row_pivot[inext];
    }
    U.ucbeg[in_] = cbeg0;
    U.uclen[in_] = clen0;
    {
        const int e = cbeg0 + clen0;
        const int where = find_in(U.ucidx, cbeg0, e, in_);
        UPD_CHECK(st, where < e);
        if (where >= e) return;
        U.row_pivot[in_] = U.ucval[where];
        UPD_CHECK(st, U.row_pivot[in_] != 0.0);
        U.ucidx[where] = U.ucidx[e - 1];
        U.ucval[where] = U.ucval[e - 1];
        U.uclen[in_] = clen0 - 1;
    }
    // mappings
    for (int n = nswap; n > 0; n--) {
        const int j = jlist[n], i = U.pmap[jlist[n - 1]];
        U.pmap[j] = i;
        U.qmap[i] = j;
    }
    U.pmap[j0] = in_;
    U.qmap[in_] = j0;
}

__global__ void __launch_bounds__(64) k_update(DevLU *Ds, SparseWs W, UpdWs U, double xtbl, int marker)
{
    const DevG D(Ds[0]);
    const int lane = lane_id();
    const int m = D.m;
    UpdState *st = U.st;
    const int nforrest = st->nforrest;
    if (lane == 0) st->status = UPD_OK;
    const int jpivot = st->btran_for;
    const int ipivot = U.pmap[jpivot];
    const double oldpiv = U.col_pivot[jpivot];
    const int sb = st->spike_beg;
    int *marked = W.marked;
    const GraphWr GW{U.wbeg, U.wlen, U.widx, U.wval, U.col_pivot};

    // results of the serial part, handed to all lanes through LDS
    __shared__ int s_istri, s_nreach, s_havediag, s_top, s_rtop, s_stop, s_nzspike;
    __shared__ double s_newpiv, s_piverr, s_spike_diag;
#ifdef BLU_PROFILE
    const long long tu0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    // the row eta into the marked work vector (update.rs:467-480) -- it can have tens of thousands of entries (the
    // transposed solve through the U chain of a banded basis): every loop over it is spread over the lanes
    {
        const int rb = U.rbeg[nforrest], re = U.rbeg[nforrest + 1];
        const int M = marker + 1;
        for (int pos = rb + lane; pos < re; pos += 64) {
            const int i = U.ridx[pos];
            marked[i] = M;
            U.work1[i] = U.rval[pos];
        }
        wave_mem_sync();
    }
    if (lane == 0) {
        s_stop = 0;
        // ---- prepare: the diagonal entry of the spike moves to its end (update.rs:441-465)
        double spike_diag = 0.0;
        int have_diag = 0, put = sb;
        for (int pos = sb; pos < sb + st->spike_len; pos++) {
            const int i = U.ucidx[pos];
            if (i != ipivot) {
                U.ucidx[put] = i;
                U.ucval[put] = U.ucval[pos];
                put++;
            } else {
                spike_diag = U.ucval[pos];
                have_diag = 1;
            }
        }
        if (have_diag) {
            U.ucidx[put] = ipivot;
            U.ucval[put] = spike_diag;
        }
        const int nz_spike = put - sb; // without the diagonal
        // ---- newpiv = spike_diag - dot(spike, row eta), intersection of the patterns counted (:467-513)
        const int M = marker + 1;
        double newpiv = spike_diag;
        int intersect = 0;
        for (int pos = sb; pos < sb + nz_spike; pos++) {
            const int i = U.ucidx[pos];
            if (marked[i] == M) {
                newpiv = __dsub_rn(newpiv, __dmul_rn(U.ucval[pos], U.work1[i]));
                intersect++;
            }
        }
        if (newpiv == 0.0 || fabs(newpiv) < D.abstol) { // singularity test: nothing has been changed
            st->status = UPD_SINGULAR;
            s_stop = 1;
        }
        // ---- room in the row file (:517-536)
        if (!s_stop) {
            long long grow = 0;
            for (int pos = sb; pos < sb + nz_spike; pos++) {
                const int j = U.qmap[U.ucidx[pos]];
                if (U.wlen[j] == U.wcap[j]) {
                    const int nz = U.wlen[j];
                    grow += nz + 1 + stretch_of(D.stretch, nz + 1) + D.pad;
                }
            }
            if (grow > (long long)U.wcapacity - st->wused) {
                st->status = UPD_NEED_W;
                st->need = (int)min(grow, 0x7fffffffLL);
                s_stop = 1;
            }
        }
        if (!s_stop) {
            int u_nz = st->u_nz;
            // ---- remove column jpivot from the row file (:538-555), erase it in the column file (:557-563)
            {
                const int cb = U.ucbeg[ipivot], ce = cb + U.uclen[ipivot];
                for (int pos = cb; pos < ce; pos++) {
                    const int j = U.qmap[U.ucidx[pos]];
                    const int e = U.wbeg[j] + U.wlen[j];
                    const int where = find_in(U.widx, U.wbeg[j], e, jpivot);
                    UPD_CHECK(st, where < e);
                    if (where < e) {
                        U.widx[where] = U.widx[e - 1];
                        U.wval[where] = U.wval[e - 1];
                        U.wlen[j] -= 1;
                    }
                }
                u_nz -= ce - cb;
            }
            // ---- the spike becomes the column (:565-570); the slot after it (the diagonal) is skipped
            U.ucbeg[ipivot] = sb;
            U.uclen[ipivot] = nz_spike;
            st->ucused = sb + nz_spike + 1;
            // ---- insert the spike into the row file (:572-601)
            for (int pos = sb; pos < sb + nz_spike; pos++) {
                const int j = U.qmap[U.ucidx[pos]];
                if (U.wlen[j] == U.wcap[j]) { // no room: the line moves to the end of the arena (file_reappend)
                    const int nz = U.wlen[j];
                    const int newcap = nz + 1 + stretch_of(D.stretch, nz + 1) + D.pad;
                    const int nb = st->wused;
                    for (int q = 0; q < nz; q++) {
                        U.widx[nb + q] = U.widx[U.wbeg[j] + q];
                        U.wval[nb + q] = U.wval[U.wbeg[j] + q];
                    }
                    U.wbeg[j] = nb;
                    U.wcap[j] = newcap;
                    st->wused = nb + newcap;
                }
                const int e = U.wbeg[j] + U.wlen[j];
                U.widx[e] = jpivot;
                U.wval[e] = U.ucval[pos];
                U.wlen[j] += 1;
            }
            u_nz += nz_spike;
            st->u_nz = u_nz;
            U.col_pivot[jpivot] = spike_diag;
            U.row_pivot[ipivot] = spike_diag;
        }
        s_newpiv = newpiv;
        s_piverr = fabs(newpiv - xtbl * oldpiv);
        s_spike_diag = spike_diag;
        s_havediag = have_diag;
        s_nzspike = nz_spike;
        s_istri = have_diag ? (intersect == 0) : 1;
        s_nreach = 0;
        s_top = m;
        s_rtop = m;
    }
    wave_mem_sync();
    __syncthreads();
#ifdef BLU_PROFILE
    const long long tu1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    if (s_stop) return;
    const double newpiv = s_newpiv;
    const int nz_spike = s_nzspike;
    const int nz_roweta = U.rbeg[nforrest + 1] - U.rbeg[nforrest];
    int *row_reach = U.iw1, *col_reach = U.iw2;
    int reach_off = 0; // row_reach / col_reach start at this offset of iw1 / iw2

    // ---- test triangularity (:607-818)
    if (s_havediag) {
        if (s_istri) { // symmetric permutation: reach = ipivot, then the pattern of the row eta (topological)
            const int nreach = nz_roweta + 1;
            const int rb = U.rbeg[nforrest];
            for (int n = 1 + lane; n < nreach; n += 64) {
                const int i = U.ridx[rb + n - 1];
                row_reach[n] = i;
                col_reach[n] = U.qmap[i];
            }
            if (lane == 0) {
                st->min_pivot = fmin(st->min_pivot, fabs(newpiv));
                st->max_pivot = fmax(st->max_pivot, fabs(newpiv));
                row_reach[0] = ipivot;
                col_reach[0] = jpivot;
                st->nsymperm_total += 1;
                s_nreach = nreach;
            }
        }
    } else {
        // spike with a zero diagonal: augmenting path jpivot -> ... -> jpivot in the row-file graph (part 1)
        int *path = U.iw1, *reach = U.iw2;
        if (lane == 0) {
            const int top = bfs_path_lane0(U, m, jpivot, path, marked, reach);
            s_top = top;
            UPD_CHECK(st, top < m - 1 && path[top] == jpivot);
            if (!(top < m - 1)) s_istri = 0; // (cannot happen after the singularity test: fall back to the FT update)
        }
        wave_mem_sync();
        const int top = s_top;
        const int M = marker + 2;
        if (lane == 0 && s_istri) {
            // part 2a: reach of every path node but the last without the path edges; combined reach in reach[rtop..m)
            int rtop = m;
            bool tri = true;
            for (int t = top; t < m - 1 && tri; t++) {
                const int j = path[t], jnext = path[t + 1];
                const int e = U.wbeg[j] + U.wlen[j];
                const int where = find_in(U.widx, U.wbeg[j], e, jnext);
                UPD_CHECK(st, where < e);
                if (where >= e) {
                    tri = false;
                    break;
                }
                U.widx[where] = j; // take the path edge out for a moment
                if (marked[j] != M) rtop = dfs_reach(GW, j, rtop, reach, W.pstack, marked, M);
                UPD_CHECK(st, reach[rtop] == j);
                reach[rtop] = jnext;
                U.widx[where] = jnext; // restore
                tri = marked[jnext] != M;
            }
            // part 2b: the reach of the final path node; triangular iff the combined reach meets the spike only there
            if (tri) {
                const int j = path[m - 1];
                if (marked[j] != M) rtop = dfs_reach(GW, j, rtop, reach, W.pstack, marked, M);
                UPD_CHECK(st, reach[rtop] == j);
                reach[rtop] = jpivot;
                marked[j] -= 1; // unmark for a moment
                const int cb = U.ucbeg[ipivot], ce = cb + U.uclen[ipivot];
                for (int pos = cb; pos < ce; pos++)
                    if (marked[U.qmap[U.ucidx[pos]]] == M) tri = false;
                marked[j] += 1;
            }
            if (tri) { // permute to a zero-free diagonal; reach lists for the permutation update
                const int nswap = m - top - 1;
                permute_lane0(U, path + top, nswap);
                st->u_nz -= 1;
                st->nunsymperm_total += 1;
                const int nreach = m - rtop;
                for (int n = 0; n < nreach; n++) row_reach[rtop + n] = U.pmap[reach[rtop + n]]; // iw1[rtop..]: the path is dead now
                s_nreach = nreach;
                s_rtop = rtop;
            }
            s_istri = tri ? 1 : 0;
        }
    }
    wave_mem_sync();
    __syncthreads();
#ifdef BLU_PROFILE
    const long long tu2 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const bool istriangular = s_istri != 0;
    if (istriangular && !s_havediag) reach_off = s_rtop;

    int nreach = s_nreach;
    if (!istriangular) {
        // ---- Forrest-Tomlin update (:820-889): row ipivot leaves U, the row eta takes its place
        if (lane == 0) {
            int u_nz = st->u_nz;
            const int wb = U.wbeg[jpivot], we = wb + U.wlen[jpivot];
            for (int pos = wb; pos < we; pos++) { // remove row ipivot from the column file
                const int j = U.widx[pos];
                const int i = U.pmap[j];
                const int e = U.ucbeg[i] + U.uclen[i];
                int where = -1;
                for (int q = U.ucbeg[i]; q < e; q++)
                    if (U.ucidx[q] == ipivot) where = q;
                UPD_CHECK(st, where >= 0);
                if (where >= 0) {
                    U.ucidx[where] = U.ucidx[e - 1];
                    U.ucval[where] = U.ucval[e - 1];
                    U.uclen[i] -= 1;
                    u_nz--;
                }
            }
            U.wlen[jpivot] = 0; // remove row ipivot from the row file
            st->u_nz = u_nz;
            U.col_pivot[jpivot] = newpiv;
            U.row_pivot[ipivot] = newpiv;
            st->min_pivot = fmin(st->min_pivot, fabs(newpiv));
            st->max_pivot = fmax(st->max_pivot, fabs(newpiv));
        }
        // drop zeros from the row eta (order kept); largest eta entry
        int nz = 0;
        const int rb0 = U.rbeg[nforrest], re0 = U.rbeg[nforrest + 1];
        int put = rb0;
        double max_eta = 0.0;
        for (int c = rb0; c < re0; c += 64) {
            const int pos = c + lane;
            const double x = pos < re0 ? U.rval[pos] : 0.0;
            const int ix = pos < re0 ? U.ridx[pos] : 0;
            const unsigned long long kb = __ballot(x != 0.0);
            wave_mem_sync();
            if (x != 0.0) {
                const int d = put + wave_prefix_count(kb);
                U.ridx[d] = ix;
                U.rval[d] = x;
                max_eta = fmax(max_eta, fabs(x));
            }
            put += __popcll(kb);
            nz += __popcll(kb);
            wave_mem_sync();
        }
        max_eta = wave_max_d(max_eta);
        if (lane == 0) {
            U.rbeg[nforrest + 1] = put;
            st->r_nz += nz;
            st->max_eta = fmax(st->max_eta, max_eta);
            st->nforrest = nforrest + 1;
            st->nforrest_total += 1;
        }
        nreach = 1;
        wave_mem_sync();
    }
    // ---- update permutations (:891-911)
    if (st->pivotlen + nreach > 2 * m) garbage_perm_wave(U, m, marked, marker + 3, W.estack);
    {
        const int put = st->pivotlen;
        if (!istriangular) {
            if (lane == 0) {
                U.pvrow[put] = ipivot;
                U.pvcol[put] = jpivot;
            }
        } else {
            const int *rr = row_reach + reach_off, *cr = col_reach + reach_off;
            for (int n = lane; n < nreach; n += 64) {
                U.pvrow[put + n] = rr[n];
                U.pvcol[put + n] = cr[n];
            }
        }
        wave_mem_sync();
        if (lane == 0) {
            st->pivotlen = put + nreach;
            // (the reference compresses the two files of U here when they have shrunk, update.rs:913-937: storage
            // layout only; the device arenas are grown by the host on demand instead)
            st->pivot_error = s_piverr / (1.0 + fabs(newpiv));
            st->btran_for = -1;
            st->ftran_for = -1;
            st->update_cost_numer += (double)nz_roweta;
        }
    }
#ifdef BLU_PROFILE
    if (lane == 0)
        printf("k_update: spike %d, row eta %d, havediag %d, triangular %d, nreach %d | spike+row file %.0f us | triangularity %.0f us | rest %.0f us\n", nz_spike,
               nz_roweta, s_havediag, s_istri, s_nreach, (tu1 - tu0) / 2100.0, (tu2 - tu1) / 2100.0, ((long long)__builtin_amdgcn_s_memtime() - tu2) / 2100.0);
#endif
    (void)nz_spike;
    (void)s_spike_diag;
}

// ---------------------------------------------------------------------------------------------------------
// k_solve_dense_upd: solve_dense on an updated factorization (solve_dense.rs:7-120 with nforrest > 0), one wave,
// every sum in the reference's order.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_ordered_dot(const double *vec, const int *idx, const double *val, int b, int e)
{
    const int lane = lane_id();
    double x = 0.0;
    for (int c = b; c < e; c += 64) {
        const int p = c + lane;
        const double term = p < e ? __dmul_rn(vec[idx[p]], val[p]) : 0.0;
        const int cnt = min(64, e - c);
        for (int q = 0; q < cnt; q++) x = __dadd_rn(x, __shfl(term, q));
    }
    return x;
}

__global__ void __launch_bounds__(64) k_solve_dense_upd(DevLU *Ds, SparseWs W, UpdWs U, const double *rhs, double *lhs, int trans, int marker)
{
    const DevG D(Ds[0]);
    const int lane = lane_id();
    const int m = D.m;
    UpdState *st = U.st;
    garbage_perm_wave(U, m, W.marked, marker + 1, W.estack); // solve_dense.rs:9
    wave_mem_sync();
    const int nforrest = st->nforrest;
    double *work1 = U.work1;
    for (int k = lane; k < m; k += 64) work1[k] = rhs[k];
    wave_mem_sync();
    if (trans) {
        for (int k = 0; k < m; k++) { // U'
            const int jpivot = U.pvcol[k], ipivot = U.pvrow[k];
            const double x = work1[jpivot] / U.col_pivot[jpivot];
            const int b = U.wbeg[jpivot], e = b + U.wlen[jpivot];
            for (int p = b + lane; p < e; p += 64) work1[U.widx[p]] = __dsub_rn(work1[U.widx[p]], __dmul_rn(x, U.wval[p]));
            if (lane == 0) lhs[ipivot] = x;
            wave_mem_sync();
        }
        for (int t = nforrest - 1; t >= 0; t--) { // row etas backwards
            const double x = lhs[U.eta_row[t]];
            const int b = U.rbeg[t], e = U.rbeg[t + 1];
            for (int p = b + lane; p < e; p += 64) lhs[U.ridx[p]] = __dsub_rn(lhs[U.ridx[p]], __dmul_rn(x, U.rval[p]));
            wave_mem_sync();
        }
        for (int k = m - 1; k >= 0; k--) { // L': dot with the stage column
            const int b = D.lbeg[k], e = D.lbeg[k + 1];
            if (e > b) {
                double x = 0.0;
                for (int c = b; c < e; c += 64) {
                    const int p = c + lane;
                    const double term = p < e ? __dmul_rn(lhs[D.lidx[p]], D.lval[p]) : 0.0;
                    const int cnt = min(64, e - c);
                    for (int q = 0; q < cnt; q++) x = __dadd_rn(x, __shfl(term, q));
                }
                if (lane == 0) lhs[D.prow[k]] = __dsub_rn(lhs[D.prow[k]], x);
                wave_mem_sync();
            }
        }
    } else {
        for (int k = 0; k < m; k++) { // L: dot with the row of row-wise L
            const int i = D.prow[k];
            const int b = W.lt_ptr[i], e = W.lt_ptr[i + 1];
            if (e > b) {
                const double x = wave_ordered_dot(work1, W.lt_idx, W.lt_val, b, e);
                if (lane == 0) work1[i] = __dsub_rn(work1[i], x);
                wave_mem_sync();
            }
        }
        for (int t = 0; t < nforrest; t++) { // row etas
            const int b = U.rbeg[t], e = U.rbeg[t + 1];
            const double x = wave_ordered_dot(work1, U.ridx, U.rval, b, e);
            if (lane == 0) work1[U.eta_row[t]] = __dsub_rn(work1[U.eta_row[t]], x);
            wave_mem_sync();
        }
        for (int k = m - 1; k >= 0; k--) { // U
            const int jpivot = U.pvcol[k], ipivot = U.pvrow[k];
            const double x = work1[ipivot] / U.row_pivot[ipivot];
            const int b = U.ucbeg[ipivot], e = b + U.uclen[ipivot];
            for (int p = b + lane; p < e; p += 64) work1[U.ucidx[p]] = __dsub_rn(work1[U.ucidx[p]], __dmul_rn(x, U.ucval[p]));
            if (lane == 0) lhs[jpivot] = x;
            wave_mem_sync();
        }
    }
}
