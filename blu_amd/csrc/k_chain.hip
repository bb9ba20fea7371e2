// k_chain.hip -- single-matrix path of the statistics tail and of solve_dense on the chain pipeline of
// k_chain.h, and the row-wise copies it needs:
//   k_rows_grid      row-wise L (build_factors.rs:243-274; the copy solve_sparse / solve_dense / update use as
//                    well) and the U rows sorted descending in pivot order, chip-wide (GridScope)
//   k_stats_chains   the four sweep chains of condest(L), condest(U), residual_test (condest.rs:15-157,
//                    residual_test.rs:43-108), one workgroup each
//   k_stats_tail     norms, residuals and the final statistics (the second half of k_stats.hip)
//   k_solve_dense_chain   solve_dense.rs:7-120
// A batch keeps k_stats / k_solve_dense (one workgroup per matrix, k_sweep.h): its matrices fill the chip anyway.
#include "blu_dev.h"
#include "k_chain.h"

struct RowsWs {
    int *lt_ptr, *lt_idx; // m+1, l_nz: row i of L = (row index of the pivot of each column it has an entry in), ascending in pivot order
    double *lt_val;
    int *lt_cur;          // m scratch
    int *ur_len;          // m: entries of stage row k of U that lie in pivotal columns
    int *ur_pos;          // u_nz, at ubeg[k]: their pivot positions, DESCENDING
    double *ur_val;
};

// insertion sort of (key,val) pairs in [b,e), ascending (DESC: descending); keys distinct.  One thread.
template <bool DESC>
__device__ __forceinline__ void insertion_sort_ik(int *key, double *val, int b, int e)
{
    for (int p = b + 1; p < e; p++) {
        const int k = key[p];
        const double v = val[p];
        int q = p - 1;
        while (q >= b && (DESC ? key[q] < k : key[q] > k)) {
            key[q + 1] = key[q];
            val[q + 1] = val[q];
            q--;
        }
        key[q + 1] = k;
        val[q + 1] = v;
    }
}
// rank sort of one segment of at most WSORT_MAX pairs by one wave through its LDS slice
template <bool DESC>
__device__ void wave_sort_segment_ik(int *key, double *val, int b, int e, int *lk, double *lv)
{
    const int lane = lane_id();
    const int n = e - b;
    WAVE_LOCKSTEP(); // (the previous segment's reads of the LDS slice are done)
    for (int t = lane; t < n; t += 64) {
        lk[t] = key[b + t];
        lv[t] = val[b + t];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int t = lane; t < n; t += 64) {
        const int k = lk[t];
        int r = 0;
        for (int u = 0; u < n; u++) r += DESC ? (lk[u] > k) : (lk[u] < k);
        key[b + r] = k;
        val[b + r] = lv[t];
    }
}

template <class Scope>
__device__ __forceinline__ void rows_body(const DevG &D, const RowsWs &R, Scope &sc, int *lds_k, double *lds_v)
{
    Scalars *S = D.s;
    const int tid = sc.tid(), nt = sc.nt();
    const int m = D.m;
    if (S->status != ST_DONE) return;
    const int rank = S->rank;
    const int lend = D.lbeg[rank];
    // ---- row-wise L: count, scan, scatter keyed by stage, sort each row by stage, stage -> pivot row
    for (int i = tid; i < m; i += nt) R.lt_cur[i] = 0;
    if (sc.leader()) *sc.ctr(0) = *sc.ctr(1) = 0;
    sc.sync();
    for (int p = tid; p < lend; p += nt) atomicAdd(&R.lt_cur[D.lidx[p]], 1);
    sc.sync();
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int i = c0 + tid;
        const int c = i < m ? R.lt_cur[i] : 0;
        int tot;
        const int ex = sc.excl_scan(c, &tot);
        if (i < m) {
            R.lt_ptr[i] = base + ex;
            R.lt_cur[i] = base + ex;
        }
        base += tot;
    }
    if (sc.leader()) R.lt_ptr[m] = base;
    sc.sync();
    for (int k = tid; k < rank; k += nt)
        for (int p = D.lbeg[k]; p < D.lbeg[k + 1]; p++) {
            const int q = atomicAdd(&R.lt_cur[D.lidx[p]], 1);
            R.lt_idx[q] = k;
            R.lt_val[q] = D.lval[p];
        }
    sc.sync();
    for (int i = tid; i < m; i += nt) {
        const int b = R.lt_ptr[i], e = R.lt_ptr[i + 1];
        if (e - b > 24 && e - b <= WSORT_MAX) D.iw2[atomicAdd(sc.ctr(0), 1)] = i;
        else insertion_sort_ik<false>(R.lt_idx, R.lt_val, b, e);
    }
    // ---- U rows: entries in pivotal columns (build_factors.rs:323), keyed by pivot position, descending
    for (int k = tid; k < m; k += nt) {
        int n = 0;
        if (k < rank) {
            const int b = D.ubeg[k], e = D.ubeg[k + 1];
            for (int p = b; p < e; p++) {
                const int c = D.qinv[D.uidx[p]];
                if (c < rank) {
                    R.ur_pos[b + n] = c;
                    R.ur_val[b + n] = D.uval[p];
                    n++;
                }
            }
            if (n > 24 && n <= WSORT_MAX) D.iw1[atomicAdd(sc.ctr(1), 1)] = k; // (its own queue: a row can be in both)
            else insertion_sort_ik<true>(R.ur_pos, R.ur_val, b, b + n);
        }
        R.ur_len[k] = n;
    }
    sc.sync();
    {
        const int nl = __hip_atomic_load(sc.ctr(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nu = __hip_atomic_load(sc.ctr(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef BLU_STATS_DEBUG
        if (sc.leader()) printf("k_rows_grid: medium L rows %d, medium U rows %d, rank %d, m %d\n", nl, nu, rank, m);
#endif
        for (int r = sc.wid(); r < nl + nu; r += sc.nw()) {
            if (r < nl) {
                const int i = D.iw2[r];
                wave_sort_segment_ik<false>(R.lt_idx, R.lt_val, R.lt_ptr[i], R.lt_ptr[i + 1], &lds_k[wave_id() * WSORT_MAX],
                                            &lds_v[wave_id() * WSORT_MAX]);
            } else {
                const int k = D.iw1[r - nl];
                wave_sort_segment_ik<true>(R.ur_pos, R.ur_val, D.ubeg[k], D.ubeg[k] + R.ur_len[k], &lds_k[wave_id() * WSORT_MAX],
                                           &lds_v[wave_id() * WSORT_MAX]);
            }
        }
    }
    sc.sync();
    for (int p = tid; p < lend; p += nt) R.lt_idx[p] = D.prow[R.lt_idx[p]];
}
__global__ void __launch_bounds__(1024) k_rows_grid(DevLU *Ds, GridWs *gw, RowsWs R)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    __shared__ int lds_k[16 * WSORT_MAX];
    __shared__ double lds_v[16 * WSORT_MAX];
    const DevG D(Ds[0]);
    GridScope sc{sh, shl, gw, 0};
    rows_body(D, R, sc, lds_k, lds_v);
}

// ---- line sets of the sweeps, in PIVOT-POSITION coordinates (work vectors indexed by k) ------------------------
// canonical U column k without its pivot (rows ascending, pivot last): U' x = b, k ascending
struct ChUCols {
    GPTR(const long long) colptr;
    GPTR(const long long) rowidx;
    GPTR(const double) value;
    gdouble_p own; // own value of step k: own[k], or null (0)
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = colptr[k];
        const long long e = colptr[k + 1] - 1;
        M.len = (int)(e - M.b);
        M.w = k;
        M.diag = value[e];
        M.own = own ? ch_load_final(own + k) : 0.0;
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = (int)rowidx[b + e];
        E.gidx = E.pos;
        E.val = value[b + e];
        return E;
    }
};
// stage-ordered L column k (row indices, production order): L' x = b, k descending
struct ChLStage {
    gcint_p lbeg, lidx, pinv;
    gdouble_p lval;
    gdouble_p own;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = lbeg[k];
        M.len = lbeg[k + 1] - lbeg[k];
        M.w = k;
        M.diag = 1.0;
        M.own = own ? ch_load_final(own + k) : 0.0;
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = pinv[lidx[b + e]];
        E.gidx = E.pos;
        E.val = lval[b + e];
        return E;
    }
};
// row-wise L, row of the pivot of step k, ascending: L y = x, k ascending
struct ChLRows {
    const int *ptr, *idx;
    const double *v;
    gcint_p prow, pinv;
    gdouble_p own;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        const int i = prow[k];
        M.b = ptr[i];
        M.len = ptr[i + 1] - ptr[i];
        M.w = k;
        M.diag = 1.0;
        M.own = own ? ch_load_final(own + k) : 0.0;
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = pinv[idx[b + e]];
        E.gidx = E.pos;
        E.val = v[b + e];
        return E;
    }
};
// stage row k of U, pivotal columns, descending: U y = x, k descending
struct ChURows {
    gcint_p ubeg;
    const int *len, *pos;
    const double *v;
    GPTR(const long long) colptr;
    GPTR(const double) value;
    gdouble_p own;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = ubeg[k];
        M.len = len[k];
        M.w = k;
        M.diag = value[colptr[k + 1] - 1];
        M.own = own ? ch_load_final(own + k) : 0.0;
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = pos[b + e];
        E.gidx = E.pos;
        E.val = v[b + e];
        return E;
    }
};

#define CHAIN_THREADS 512
// blockIdx.x = chain: 0 condest(L), 1 condest(U), 2 residual test forward, 3 residual test backward.
// Results of chains 0 and 1 go to gwork[8(m+1)], gwork[8(m+1)+1]; the work vectors (positions) stay in gwork
// for k_stats_tail: lf = gwork[2(m+1)..], rf [3..], lb [4..], rb [5..].
// defect: a sweep that gave up one of its bounded waits (k_chain.h) leaves its code here (max over the four chains); the
// host then recomputes the statistics with the one-workgroup kernel instead of trusting half-finished work vectors
__global__ void __launch_bounds__(CHAIN_THREADS) k_stats_chains(DevLU *Ds, FinishOut *Os, RowsWs R, int *defect)
{
    BLU_DYN_SHARED(unsigned char, ch_smem, sizeof(ChainLds));
    ChainLds *L = (ChainLds *)ch_smem;
    const DevG D(Ds[0]);
    const FinishOut &O = Os[0];
    Scalars *S = D.s;
    const int m = D.m, lane = lane_id();
    if (S->status != ST_DONE || D.skip_stats) return;
    const size_t M1 = (size_t)(m + 1);
    gdouble_p wl = D.gwork, wu = D.gwork + M1, lf = D.gwork + 2 * M1, rf = D.gwork + 3 * M1, lb = D.gwork + 4 * M1, rb = D.gwork + 5 * M1;
    gdouble_p res = D.gwork + 8 * M1;
    const bool w0 = threadIdx.x < 64;
    const int cc = blockIdx.x;
    bool ok = true;
    if (cc == 0) {
        // condest(L): L' x = b with b = +-1 chosen on the fly, k descending (condest.rs:101-116), then L y = x (135-154)
        double x1 = 0.0, xinf = 0.0, y1 = 0.0;
        const ChLStage A1{D.lbeg, D.lidx, D.pinv, D.lval, nullptr};
        ok = ok && chain_sweep<false, false>(A1, L, m - 1, -1, m, wl, [&](int, bool has, double dot, double, double) {
            double temp = has ? -dot : 0.0;
            temp += temp >= 0.0 ? 1.0 : -1.0;
            x1 += fabs(temp);
            xinf = fmax(xinf, fabs(temp));
            return temp;
        });
        const ChLRows A2{R.lt_ptr, R.lt_idx, R.lt_val, D.prow, D.pinv, wl};
        ok = ok && chain_sweep<true, true>(A2, L, 0, 1, m, wl, [&](int, bool, double acc, double, double) {
            y1 += fabs(acc);
            return acc;
        });
        if (w0 && lane == 0) res[0] = fmax(y1 / x1, xinf);
    } else if (cc == 1) {
        // condest(U): U' x = b, k ascending, then U y = x, k descending
        double x1 = 0.0, xinf = 0.0, y1 = 0.0;
        const ChUCols A1{(GPTR(const long long))O.u_colptr, (GPTR(const long long))O.u_rowidx, (GPTR(const double))O.u_value, nullptr};
        ok = ok && chain_sweep<false, false>(A1, L, 0, 1, m, wu, [&](int, bool has, double dot, double, double diag) {
            double temp = has ? -dot : 0.0;
            temp += temp >= 0.0 ? 1.0 : -1.0;
            temp /= diag;
            x1 += fabs(temp);
            xinf = fmax(xinf, fabs(temp));
            return temp;
        });
        const ChURows A2{D.ubeg, R.ur_len, R.ur_pos, R.ur_val, (GPTR(const long long))O.u_colptr, (GPTR(const double))O.u_value, wu};
        ok = ok && chain_sweep<true, true>(A2, L, m - 1, -1, m, wu, [&](int, bool, double acc, double, double diag) {
            const double temp = acc / diag;
            y1 += fabs(temp);
            return temp;
        });
        if (w0 && lane == 0) res[1] = fmax(y1 / x1, xinf);
    } else if (cc == 2) {
        // residual test, forward system (residual_test.rs:43-66): lhs = L\rhs with rhs = +-1 on the fly, then U\lhs
        const ChLRows A1{R.lt_ptr, R.lt_idx, R.lt_val, D.prow, D.pinv, nullptr};
        ok = ok && chain_sweep<false, false>(A1, L, 0, 1, m, lf, [&](int k, bool, double d, double, double) {
            const double r = d <= 0.0 ? 1.0 : -1.0;
            if (lane == 0) rf[k] = r;
            return r - d;
        });
        const ChURows A2{D.ubeg, R.ur_len, R.ur_pos, R.ur_val, (GPTR(const long long))O.u_colptr, (GPTR(const double))O.u_value, lf};
        ok = ok && chain_sweep<true, true>(A2, L, m - 1, -1, m, lf, [&](int, bool, double acc, double, double diag) { return acc / diag; });
    } else {
        // residual test, backward system (residual_test.rs:85-108): lhs = U'\rhs, then L'\lhs
        const ChUCols A1{(GPTR(const long long))O.u_colptr, (GPTR(const long long))O.u_rowidx, (GPTR(const double))O.u_value, nullptr};
        ok = ok && chain_sweep<false, false>(A1, L, 0, 1, m, lb, [&](int k, bool has, double dot, double, double diag) {
            const double d = has ? dot : 0.0;
            const double r = d <= 0.0 ? 1.0 : -1.0;
            if (lane == 0) rb[k] = r;
            return (r - d) / diag;
        });
        const ChLStage A2{D.lbeg, D.lidx, D.pinv, D.lval, lb};
        ok = ok && chain_sweep<false, false>(A2, L, m - 1, -1, m, lb, [&](int, bool has, double dot, double own, double) { return has ? own - dot : own; });
    }
    if (!ok && threadIdx.x == 0) atomicMax(defect, L->abort ? L->abort : 99);
}

// The passes over the columns and rows of B, L, U on TAIL_BLOCKS workgroups; per-workgroup maxima to the grid scratch
#define TAIL_BLOCKS 64
__global__ void __launch_bounds__(1024) k_stats_tail_a(DevLU *Ds, FinishOut *Os, GridWs *gw)
{
    const DevG D(Ds[0]);
    const FinishOut &O = Os[0];
    Scalars *S = D.s;
    __shared__ double red[4][40];
    if (S->status != ST_DONE || D.skip_stats) return;
    double v[4];
    stats_tail_loops<false>(D, O, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x), v[0], v[1], v[2], v[3]);
    for (int q = 0; q < 4; q++) {
        const double x = wave_max_d(v[q]);
        if (lane_id() == 0) red[q][wave_id()] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double a = red[threadIdx.x][0];
        for (int ww = 1; ww < num_waves(); ww++) a = fmax(a, red[threadIdx.x][ww]);
        ((double *)gw->partll[0])[4 * blockIdx.x + threadIdx.x] = a; // (4 * TAIL_BLOCKS <= SCOPE_MAX_BLOCKS slots)
    }
}
__global__ void __launch_bounds__(1024) k_stats_tail_b(DevLU *Ds, FinishOut *Os, GridWs *gw)
{
    const DevG D(Ds[0]);
    Scalars *S = D.s;
    __shared__ double red[4][40];
    __shared__ double chain_out[16];
    if (S->status != ST_DONE || D.skip_stats) return;
    const size_t M1 = (size_t)(D.m + 1);
    if (threadIdx.x < 2) chain_out[threadIdx.x] = D.gwork[8 * M1 + threadIdx.x];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < TAIL_BLOCKS)
        for (int q = 0; q < 4; q++) v[q] = ((const double *)gw->partll[0])[4 * threadIdx.x + q];
    __syncthreads();
    stats_tail_finish(D, red, chain_out, v[0], v[1], v[2], v[3]);
    // restore the all-zero invariant of the pivot_any work area
    const size_t ng = 7 * M1;
    for (size_t e = threadIdx.x; e < ng; e += blockDim.x) D.gwork[e] = 0.0;
    if (threadIdx.x < 2) D.gwork[8 * M1 + threadIdx.x] = 0.0;
}

// ---- solve_dense (solve_dense.rs:7-120) on the chain pipeline.  The work vectors keep the caller's numbering as in
// k_solve.hip: y = the right-hand side (row indices for B x = b, column indices for B' x = b), lhs the solution.
// forward, L: y[pivotrow[k]] -= ordered dot of row pivotrow[k] of L with y (solve_dense.rs:79-86)
struct ChSolveLRows {
    const int *ptr, *idx;
    const double *v;
    gcint_p prow, pinv;
    gdouble_p y;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        const int i = prow[k];
        M.b = ptr[i];
        M.len = ptr[i + 1] - ptr[i];
        M.w = i;
        M.diag = 1.0;
        M.own = ch_load_final(y + i);
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.gidx = idx[b + e];
        E.pos = pinv[E.gidx];
        E.val = v[b + e];
        return E;
    }
};
// forward, U: x = y[pivotrow[k]] / pivot after the columns k'' > k took their share of it, k descending
// (solve_dense.rs:88-98, as a gather over row k of U, descending); lhs[pivotcol[k]] = x
struct ChSolveURows {
    gcint_p ubeg;
    const int *len, *pos;
    const double *v;
    GPTR(const long long) colptr;
    GPTR(const double) value;
    gcint_p prow, pcol;
    gdouble_p y;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = ubeg[k];
        M.len = len[k];
        M.w = pcol[k];
        M.diag = value[colptr[k + 1] - 1];
        M.own = ch_load_final(y + prow[k]);
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = pos[b + e];
        E.gidx = pcol[E.pos];
        E.val = v[b + e];
        return E;
    }
};
// transposed, U': x = y[pivotcol[k]] / pivot after the rows k' < k took their share, k ascending
// (solve_dense.rs:36-49, as a gather over column k of U, ascending); lhs[pivotrow[k]] = x
struct ChSolveUCols {
    GPTR(const long long) colptr;
    GPTR(const long long) rowidx;
    GPTR(const double) value;
    gcint_p prow, pcol;
    gdouble_p y;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = colptr[k];
        const long long e = colptr[k + 1] - 1;
        M.len = (int)(e - M.b);
        M.w = prow[k];
        M.diag = value[e];
        M.own = ch_load_final(y + pcol[k]);
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.pos = (int)rowidx[b + e];
        E.gidx = prow[E.pos];
        E.val = value[b + e];
        return E;
    }
};
// transposed, L': lhs[pivotrow[k]] -= ordered dot of stage column k of L with lhs, k descending (solve_dense.rs:51-60)
struct ChSolveLStage {
    gcint_p lbeg, lidx, pinv, prow;
    gdouble_p lval;
    gdouble_p x;
    __device__ __forceinline__ ChMeta meta(int k) const
    {
        ChMeta M;
        M.b = lbeg[k];
        M.len = lbeg[k + 1] - lbeg[k];
        M.w = prow[k];
        M.diag = 1.0;
        M.own = ch_load_final(x + M.w);
        return M;
    }
    __device__ __forceinline__ ChEnt ent(int, long long b, int e) const
    {
        ChEnt E;
        E.gidx = lidx[b + e];
        E.pos = pinv[E.gidx];
        E.val = lval[b + e];
        return E;
    }
};

// *defect: 0, or 9100 + the code of the bounded wait that gave up (the host then fails the call)
__global__ void __launch_bounds__(CHAIN_THREADS) k_solve_dense_chain(DevLU *Ds, FinishOut *Os, RowsWs R, const double *rhs, double *lhs, int trans,
                                                                     int *defect)
{
    BLU_DYN_SHARED(unsigned char, ch_smem, sizeof(ChainLds));
    ChainLds *L = (ChainLds *)ch_smem;
    const DevG D(Ds[0]);
    const FinishOut &O = Os[0];
    const int m = D.m;
    gdouble_p y = D.txrj; // m+2 doubles of scratch
    gdouble_p x_out = (gdouble_p)lhs;
    for (int k = threadIdx.x; k < m; k += blockDim.x) y[k] = rhs[k]; // solve_dense.rs:34 / :77
    ch_vm_drain();
    __syncthreads();
    const auto sub_dot = [](int, bool has, double dot, double own, double) { return has ? own - dot : own; };
    const auto over_diag = [](int, bool, double acc, double, double diag) { return acc / diag; };
    bool ok = true;
    if (!trans) {
        const ChSolveLRows A1{R.lt_ptr, R.lt_idx, R.lt_val, D.prow, D.pinv, y};
        ok = ok && chain_sweep<false, false>(A1, L, 0, 1, m, y, sub_dot);
        const ChSolveURows A2{D.ubeg, R.ur_len, R.ur_pos, R.ur_val, (GPTR(const long long))O.u_colptr, (GPTR(const double))O.u_value, D.prow, D.pcol, y};
        ok = ok && chain_sweep<true, true>(A2, L, m - 1, -1, m, x_out, over_diag);
    } else {
        const ChSolveUCols A1{(GPTR(const long long))O.u_colptr, (GPTR(const long long))O.u_rowidx, (GPTR(const double))O.u_value, D.prow, D.pcol, y};
        ok = ok && chain_sweep<true, true>(A1, L, 0, 1, m, x_out, over_diag);
        const ChSolveLStage A2{D.lbeg, D.lidx, D.pinv, D.prow, D.lval, x_out};
        ok = ok && chain_sweep<false, false>(A2, L, m - 1, -1, m, x_out, sub_dot);
    }
    if (threadIdx.x == 0) *defect = ok ? 0 : 9100 + L->abort;
}
