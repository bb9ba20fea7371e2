// k_stats.hip -- the statistics tail of factorize() (src/factorize.rs:121-147), one workgroup per matrix:
//   condest(L), condest(U)   src/lu/condest.rs:15-157   (LINPACK 1-norm condition estimates)
//   residual_test            src/lu/residual_test.rs:16-152 (+ matrix_norm, src/lu/matrix_norm.rs:8-48)
//
// The four triangular-sweep chains (condest L, condest U, residual forward, residual backward) are
// independent of each other and run concurrently on four waves; each chain is a dependent sequence of
// m pivot steps (lanes over the entries of the step's column).  Afterwards the whole workgroup forms the
// residuals, norms and maxima in parallel.  Everything works in PIVOT-ORDER coordinates on the canonical
// factors k_finish wrote (O: L unit lower CSC diagonal first, U upper CSC pivot last), i.e. on
//   B[rowperm, colperm] = L * U,  vector index k <-> row rowperm[k] / column colperm[k].
//
// Floating point: dots are accumulated by ONE lane in storage order (the reference's loop order for the
// U sweeps and the forward L sweep; the L dots of condest/backward use the sorted column instead of the
// stage order), so values agree with the reference to rounding (tested at 1e-9 relative), and the
// +-1 right-hand-side choices (`temp >= 0`, `d <= 0`) are made on identically defined quantities.
#include "blu_dev.h"

typedef GPTR(const long long) gcll_p;
typedef GPTR(const double) gcdouble_p;

// ------------------------------------------------------------------------------------------------
// Pipelined triangular sweeps.  A sweep is a chain of m dependent steps: step k reads entries of the
// work vector that step k-1 may just have written.  What does NOT depend on the previous step -- the
// column pointers, the column's (index, value) entries, its diagonal / pivot row -- is fetched one and
// two steps ahead, so that a step costs ONE memory round trip (the gather of the work vector) instead
// of four.  Lanes hold one entry each of the first 64 of a column; longer columns (rare) take the slow
// tail loops.  One lane's store to the work vector is seen by the other lanes' later loads without a
// drain: a wave's memory operations are performed in order (wave_mem_sync = wavefront-scope fence).
// ------------------------------------------------------------------------------------------------
struct ColPtr {
    long long b, e; // entries [b, e)
    double diag;    // U: pivot (last entry of the column); unused for L
    int aux;        // stage L: prow[k]
};
struct ColEnt {
    int idx;
    double val;
};
// U columns of the canonical factors: off-diagonals [colptr[k], colptr[k+1]-1), pivot last
struct UCols {
    gcll_p colptr, rowidx;
    gcdouble_p value;
    int m;
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        ColPtr P;
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        P.b = colptr[k];
        P.e = colptr[k + 1] - 1;
        P.diag = 0.0;
        P.aux = 0;
        return P;
    }
    __device__ __forceinline__ void diag(ColPtr &P) const { P.diag = value[P.e]; }
    __device__ __forceinline__ ColEnt ent(const ColPtr &P, long long off) const
    {
        ColEnt E;
        E.idx = 0;
        E.val = 0.0;
        const long long p = P.b + off + lane_id();
        if (p < P.e) {
            E.idx = (int)rowidx[p];
            E.val = value[p];
        }
        return E;
    }
};
// sorted L columns of the canonical factors without the unit diagonal: (colptr[k], colptr[k+1])
struct LCols {
    gcll_p colptr, rowidx;
    gcdouble_p value;
    int m;
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        ColPtr P;
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        P.b = colptr[k] + 1;
        P.e = colptr[k + 1];
        P.diag = 1.0;
        P.aux = 0;
        return P;
    }
    __device__ __forceinline__ void diag(ColPtr &) const {}
    __device__ __forceinline__ ColEnt ent(const ColPtr &P, long long off) const
    {
        ColEnt E;
        E.idx = 0;
        E.val = 0.0;
        const long long p = P.b + off + lane_id();
        if (p < P.e) {
            E.idx = (int)rowidx[p];
            E.val = value[p];
        }
        return E;
    }
};
// stage-ordered L columns as the pivot loop wrote them (row indices of B): the reference's own storage
// and summation order (l_begin_p, pivot.rs:404-416); `map` (or null) takes a row index to its position
struct LStage {
    gcint_p lbeg, lidx, prow, map;
    gcdouble_p lval;
    int m;
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        ColPtr P;
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        P.b = lbeg[k];
        P.e = lbeg[k + 1];
        P.diag = 1.0;
        P.aux = prow[k];
        return P;
    }
    __device__ __forceinline__ void diag(ColPtr &) const {}
    __device__ __forceinline__ ColEnt ent(const ColPtr &P, long long off) const
    {
        ColEnt E;
        E.idx = 0;
        E.val = 0.0;
        const long long p = P.b + off + lane_id();
        if (p < P.e) {
            const int i = lidx[p];
            E.idx = map ? map[i] : i;
            E.val = lval[p];
        }
        return E;
    }
};

// sum of prod over lanes 0..n-1 in lane order, every lane gets it (the reference's sequential loop)
__device__ __forceinline__ double wave_ordered_sum(double prod, int n, double acc)
{
    const unsigned lo = (unsigned)__double_as_longlong(prod), hi = (unsigned)(__double_as_longlong(prod) >> 32);
    for (int t = 0; t < n; t++) {
        const unsigned a = __builtin_amdgcn_readlane(lo, t), b = __builtin_amdgcn_readlane(hi, t);
        acc = __dadd_rn(acc, __longlong_as_double((long long)(((unsigned long long)b << 32) | a)));
    }
    return acc;
}

// ordered dot of column P with the work vector x (first chunk E already in registers)
template <class Cols>
__device__ __forceinline__ double col_dot(const Cols &C, const ColPtr &P, const ColEnt &E, gdouble_p x)
{
    const long long len = P.e - P.b;
    if (len <= 0) return 0.0;
    const int lane = lane_id();
    const int n0 = len < 64 ? (int)len : 64;
    double acc = wave_ordered_sum(lane < n0 ? __dmul_rn(x[E.idx], E.val) : 0.0, n0, 0.0);
    for (long long off = 64; off < len; off += 64) {
        const ColEnt E2 = C.ent(P, off);
        const int n = (len - off) < 64 ? (int)(len - off) : 64;
        acc = wave_ordered_sum(lane < n ? __dmul_rn(x[E2.idx], E2.val) : 0.0, n, acc);
    }
    return acc;
}
// x[idx] = x[idx] -/+ t * val over column P (sub: minus)
template <bool SUB, class Cols>
__device__ __forceinline__ void col_scatter(const Cols &C, const ColPtr &P, const ColEnt &E, gdouble_p x, double t)
{
    const long long len = P.e - P.b;
    const int lane = lane_id();
    if (lane < len) {
        const double pr = __dmul_rn(t, E.val);
        x[E.idx] = SUB ? __dsub_rn(x[E.idx], pr) : __dadd_rn(x[E.idx], pr);
    }
    for (long long off = 64; off < len; off += 64) {
        const ColEnt E2 = C.ent(P, off);
        if (off + lane < len) {
            const double pr = __dmul_rn(t, E2.val);
            x[E2.idx] = SUB ? __dsub_rn(x[E2.idx], pr) : __dadd_rn(x[E2.idx], pr);
        }
    }
}

// for k = k0, k0+dir, .. (n steps): body(k, P_k, E_k) with the pointers of step k+2 and the entries of
// step k+1 in flight
template <class Cols, class Body>
__device__ __forceinline__ void sweep(const Cols &C, int k0, int dir, int n, Body body)
{
    if (n <= 0) return;
    ColPtr P1 = C.ptr(k0);
    C.diag(P1);
    ColEnt E1 = C.ent(P1, 0);
    ColPtr P2 = C.ptr(k0 + dir);
    for (int s = 0, k = k0; s < n; s++, k += dir) {
        C.diag(P2);
        const ColEnt E2 = C.ent(P2, 0);
        const ColPtr P3 = C.ptr(k + 2 * dir);
        body(k, P1, E1);
        wave_mem_sync();
        P1 = P2;
        E1 = E2;
        P2 = P3;
    }
}

__global__ void __launch_bounds__(1024) k_stats(DevLU *Ds, FinishOut *Os)
{
    const DevG D(Ds[blockIdx.x]);
    const FinishOut &O = Os[blockIdx.x];
    Scalars *S = D.s;
    __shared__ double red[4][40];
    __shared__ double chain_out[16];
    const int tid = threadIdx.x, nt = blockDim.x, w = wave_id(), lane = lane_id(), nw = num_waves();
    const int m = D.m;
    if (S->status != ST_DONE) return;
    const int rank = S->rank;
    // six m-vectors in the (all-zero) pivot_any work area; re-zeroed at the end
    gdouble_p wl = D.gwork, wu = D.gwork + (size_t)(m + 1), lf = D.gwork + 2 * (size_t)(m + 1),
              rf = D.gwork + 3 * (size_t)(m + 1), lb = D.gwork + 4 * (size_t)(m + 1), rb = D.gwork + 5 * (size_t)(m + 1);
    gdouble_p rs = D.gwork + 6 * (size_t)(m + 1); // row sums of |B|

    const UCols CU{(gcll_p)O.u_colptr, (gcll_p)O.u_rowidx, (gcdouble_p)O.u_value, m};
    const LCols CL{(gcll_p)O.l_colptr, (gcll_p)O.l_rowidx, (gcdouble_p)O.l_value, m};
    const LStage CS{D.lbeg, D.lidx, D.prow, nullptr, D.lval, m};
    const LStage CSmap{D.lbeg, D.lidx, D.prow, D.pinv, D.lval, m};
    for (int cc = 0; cc < 4; cc++) {
        // with fewer than 4 waves the chains run one after the other on wave 0
        const bool mine = nw >= 4 ? (w == cc) : (w == 0);
        if (!mine) continue;
        if (cc == 0) {
            // ---- condest(L): L' x = b with b = +-1 chosen on the fly, k descending (condest.rs:101-116, upper = 0)
            // This chain works in ROW-INDEX coordinates on the stage-ordered L columns (wl[i], i = row of B).
            double x1 = 0.0, xinf = 0.0;
            sweep(CS, m - 1, -1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                double temp = 0.0;
                if (P.e > P.b) temp = -col_dot(CS, P, E, wl); // temp -= work[i]*x
                temp += temp >= 0.0 ? 1.0 : -1.0;
                if (lane == 0) wl[P.aux] = temp;
                x1 += fabs(temp);
                xinf = fmax(xinf, fabs(temp));
            });
            // L y = x, k ascending, scatter (condest.rs:135-154)
            double y1 = 0.0;
            sweep(CS, 0, 1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                const double temp = wl[P.aux];
                col_scatter<true>(CS, P, E, wl, temp);
                y1 += fabs(temp);
            });
            if (lane == 0) chain_out[0] = fmax(y1 / x1, xinf); // normest_l_inv
        } else if (cc == 1) {
            // ---- condest(U): U' x = b, k ascending, then U y = x, k descending (upper = 1, pivots = diagonal)
            double x1 = 0.0, xinf = 0.0;
            sweep(CU, 0, 1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                double temp = 0.0;
                if (P.e > P.b) temp = -col_dot(CU, P, E, wu);
                temp += temp >= 0.0 ? 1.0 : -1.0;
                temp /= P.diag;
                if (lane == 0) wu[k] = temp;
                x1 += fabs(temp);
                xinf = fmax(xinf, fabs(temp));
            });
            double y1 = 0.0;
            sweep(CU, m - 1, -1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                const double temp = wu[k] / P.diag;
                wave_mem_sync(); // every lane has read wu[k] before lane 0 rewrites it
                if (lane == 0) wu[k] = temp;
                col_scatter<true>(CU, P, E, wu, temp);
                y1 += fabs(temp);
            });
            if (lane == 0) chain_out[1] = fmax(y1 / x1, xinf); // normest_u_inv
        } else if (cc == 2) {
            // ---- residual test, forward system (residual_test.rs:43-66): lhs = L\rhs with rhs = +-1 on the fly.
            // The reference takes row dots of L; the column scatter below adds the same products to each
            // accumulator in the same (ascending stage) order.  lf[k] first accumulates d, then holds lhs.
            sweep(CL, 0, 1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                const double d = lf[k];
                const double r = d <= 0.0 ? 1.0 : -1.0;
                const double x = r - d;
                wave_mem_sync();
                if (lane == 0) {
                    rf[k] = r;
                    lf[k] = x;
                }
                col_scatter<false>(CL, P, E, lf, x);
            });
            // overwrite lhs by U\lhs, k descending (residual_test.rs:57-66)
            sweep(CU, m - 1, -1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                const double d = lf[k] / P.diag;
                wave_mem_sync();
                if (lane == 0) lf[k] = d;
                col_scatter<true>(CU, P, E, lf, d);
            });
        } else {
            // ---- residual test, backward system (residual_test.rs:85-108): lhs = U'\rhs, then L'\lhs
            sweep(CU, 0, 1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                double d = 0.0;
                if (P.e > P.b) d = col_dot(CU, P, E, lb);
                const double r = d <= 0.0 ? 1.0 : -1.0;
                if (lane == 0) {
                    rb[k] = r;
                    lb[k] = (r - d) / P.diag;
                }
            });
            // dots with the stage-ordered L columns (rows mapped to positions), k descending
            sweep(CSmap, m - 1, -1, m, [&](int k, const ColPtr &P, const ColEnt &E) {
                if (P.e > P.b) {
                    const double d = col_dot(CSmap, P, E, lb);
                    const double v = lb[k] - d;
                    wave_mem_sync();
                    if (lane == 0) lb[k] = v;
                }
            });
        }
    }
    __syncthreads();

    // ---- norms of L and U (condest.rs:27-44), 1-norm = max column sum
    double nl = 0.0, nu = 0.0;
    for (int k = tid; k < m; k += nt) {
        double s = 1.0;
        for (long long p = O.l_colptr[k] + 1; p < O.l_colptr[k + 1]; p++) s += fabs(O.l_value[p]);
        nl = fmax(nl, s);
        const long long e = O.u_colptr[k + 1] - 1;
        double t = fabs(O.u_value[e]);
        for (long long p = O.u_colptr[k]; p < e; p++) t += fabs(O.u_value[p]);
        nu = fmax(nu, t);
    }
    // ---- residuals (residual_test.rs:68-83, 110-126) and matrix norms (matrix_norm.rs), pivot coordinates:
    // column k of the factorized matrix is column colperm[k] of B for k < rank, the unit vector e_k otherwise
    for (int i = tid; i < m; i += nt) rs[i] = 0.0;
    __syncthreads();
    double one = 0.0;
    for (int k = tid; k < m; k += nt) {
        if (k < rank) {
            const int j = D.pcol[k];
            double cs = 0.0, d = 0.0;
            for (int p = D.bc_ptr[j]; p < D.bc_ptr[j + 1]; p++) {
                const double a = D.bc_val[p];
                cs += fabs(a);
                d = __dadd_rn(d, __dmul_rn(lb[D.pinv[D.bc_idx[p]]], a)); // B' * lhs, column order of B
            }
            one = fmax(one, cs);
            rb[k] = rb[k] - d;
        } else {
            one = fmax(one, 1.0);
            rb[k] = rb[k] - lb[k];
        }
    }
    // forward residual rhs - B*lhs and row sums: one thread per ROW of B (bt_* = B row-wise, sorted by column)
    for (int i = tid; i < m; i += nt) {
        const int kr = D.pinv[i];
        double acc = rf[kr], rsum = 0.0;
        for (int p = D.bt_ptr[i]; p < D.bt_ptr[i + 1]; p++) {
            const int kc = D.qinv[D.bt_idx[p]];
            if (kc < rank) {
                const double a = D.bt_val[p];
                acc = __dsub_rn(acc, __dmul_rn(lf[kc], a));
                rsum += fabs(a);
            }
        }
        if (kr >= rank) {
            acc = acc - lf[kr];
            rsum += 1.0;
        }
        rf[kr] = acc;
        rs[i] = rsum;
    }
    __syncthreads();
    double s_lf = 0.0, s_rf = 0.0, s_lb = 0.0, s_rb = 0.0, inf = 0.0;
    for (int k = tid; k < m; k += nt) {
        s_lf += fabs(lf[k]);
        s_rf += fabs(rf[k]);
        s_lb += fabs(lb[k]);
        s_rb += fabs(rb[k]);
        inf = fmax(inf, rs[k]);
    }
    // workgroup reductions: sums and maxima
    double vals[8] = {s_lf, s_rf, s_lb, s_rb, nl, nu, one, inf};
    for (int q = 0; q < 8; q++) {
        double v = vals[q];
        if (q < 4) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        } else {
            v = wave_max_d(v);
        }
        if (lane == 0) red[q & 3][w] = v;
        __syncthreads();
        if (tid == 0) {
            double a = red[q & 3][0];
            for (int ww = 1; ww < nw; ww++) a = q < 4 ? a + red[q & 3][ww] : fmax(a, red[q & 3][ww]);
            chain_out[4 + q] = a;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double nf = chain_out[4], nrf = chain_out[5], nb = chain_out[6], nrb = chain_out[7];
        S->norm_l = chain_out[8];
        S->norm_u = chain_out[9];
        S->onenorm = chain_out[10];
        S->infnorm = chain_out[11];
        S->normest_l_inv = chain_out[0];
        S->normest_u_inv = chain_out[1];
        S->condest_l = chain_out[8] * chain_out[0];
        S->condest_u = chain_out[9] * chain_out[1];
        S->residual_test = fmax(nrf / ((double)m + chain_out[10] * nf), nrb / ((double)m + chain_out[11] * nb));
    }
    // restore the all-zero invariant of the pivot_any work area
    const size_t ng = (size_t)7 * (m + 1);
    for (size_t e = tid; e < ng; e += nt) D.gwork[e] = 0.0;
}
