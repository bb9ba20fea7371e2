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
// Floating point: every sum is taken in the reference's order -- dots over a line in its storage order
// (lane order of wave_ordered_sum), scatter updates of one entry in ascending pivot order, the residual
// terms of a row ascending in the pivot position of their columns, the 1-norms sequentially over the row
// indices -- so all statistics, including residual_test (pure rounding noise), are bit-identical to it
// (tests: test_statistics_tail*).
#include "blu_dev.h"

#include "k_sweep.h"

// Second half of the statistics: norms of L and U, the residuals of both systems, matrix norms and the final
// numbers (condest.rs:27-44, 60-76; residual_test.rs:68-152; matrix_norm.rs:8-48).
// chain_out[0], [1] = normest_l_inv, normest_u_inv of the condest chains; lf/rf/lb/rb as the chains left them.
//
// stats_tail_loops: the per-column / per-row passes, thread `tid` of `nt` (one workgroup of a batch, or the grid of
// k_stats_tail_a); returns this thread's maxima.  stats_tail_finish: the four ordered 1-norms, the reductions
// and the final numbers, one workgroup.
// The forward residual of ONE row of at most N entries without a dependent chain of memory accesses: every entry's
// pivot position, value and the lhs entry it multiplies into registers (loads issued together), the entries ranked by
// pivot position there, the terms applied in that order.  (The loop it replaces selects the next entry by a scan of
// the whole row, through memory, once per entry: ~2 n^2 dependent loads per row -- 0.39 of the 0.74 s of the
// statistics of 1536 bases of the 100k size were this tail.)
template <int N>
__device__ __forceinline__ void forward_row_reg(const DevG &D, gdouble_p lf, int b, int n, int rank, double &acc, double &rsum)
{
    int kc[N];
    double a[N], l[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        kc[i] = 0x7fffffff;
        a[i] = 0.0;
        if (i < n) {
            kc[i] = D.qinv[D.bt_idx[b + i]];
            a[i] = D.bt_val[b + i];
        }
    }
    int nv = 0; // entries in pivotal columns (the others are skipped: residual_test.rs:68-76 runs over the pivotal columns)
#pragma unroll
    for (int i = 0; i < N; i++) {
        l[i] = 0.0;
        if (kc[i] < rank) {
            l[i] = lf[kc[i]];
            nv++;
        } else {
            kc[i] = 0x7fffffff;
        }
    }
#pragma unroll
    for (int t = 0; t < N; t++) {
        // the entry with t smaller positions before it (positions are distinct: distinct columns)
        double lt = 0.0, at = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < N; j++) r += (kc[j] < kc[i]) ? 1 : 0;
            if (r == t && kc[i] != 0x7fffffff) {
                lt = l[i];
                at = a[i];
            }
        }
        if (t < nv) {
            acc = __dsub_rn(acc, __dmul_rn(lt, at));
            rsum += fabs(at);
        }
    }
}

// REG: rows of at most 32 entries by forward_row_reg (the 256-thread workgroups of k_stats_tail: their register budget allows it)
template <bool REG>
__device__ __forceinline__ void stats_tail_loops(const DevG &D, const FinishOut &O, int tid, int nt, double &nl, double &nu, double &one,
                                                 double &inf)
{
    Scalars *S = D.s;
    const int m = D.m;
    const int rank = S->rank;
    gdouble_p lf = D.gwork + 2 * (size_t)(m + 1), rf = D.gwork + 3 * (size_t)(m + 1), lb = D.gwork + 4 * (size_t)(m + 1),
              rb = D.gwork + 5 * (size_t)(m + 1);
    // ---- norms of L and U (condest.rs:27-44), 1-norm = max column sum
    nl = 0.0;
    nu = 0.0;
    for (int k = tid; k < m; k += nt) {
        double s = 1.0; // (stage-ordered column: the reference's storage and summation order)
        line4(D.lbeg[k], D.lbeg[k + 1], [&](int p) { return D.lval[p]; }, [&](int, double v) { s += fabs(v); });
        nl = fmax(nl, s);
        const long long e = O.u_colptr[k + 1] - 1;
        double t = fabs(O.u_value[e]);
        line4((int)O.u_colptr[k], (int)e, [&](int p) { return O.u_value[p]; }, [&](int, double v) { t += fabs(v); });
        nu = fmax(nu, t);
    }
    // ---- residuals (residual_test.rs:68-83, 110-126) and matrix norms (matrix_norm.rs), pivot coordinates:
    // column k of the factorized matrix is column colperm[k] of B for k < rank, the unit vector e_k otherwise
    one = 0.0;
    for (int k = tid; k < m; k += nt) {
        if (k < rank) {
            const int j = D.pcol[k];
            double cs = 0.0, d = 0.0;
            line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int p) { return make_double2(D.bc_val[p], lb[D.pinv[D.bc_idx[p]]]); },
                  [&](int, const double2 &av) {
                      cs += fabs(av.x);
                      d = __dadd_rn(d, __dmul_rn(av.y, av.x)); // B' * lhs, column order of B
                  });
            one = fmax(one, cs);
            rb[k] = rb[k] - d;
        } else {
            one = fmax(one, 1.0);
            rb[k] = rb[k] - lb[k];
        }
    }
    // forward residual rhs - B*lhs and row sums: one thread per ROW of B (bt_* = B row-wise).  The
    // reference scatters column after column in pivot order (residual_test.rs:68-76, matrix_norm.rs:
    // 26-36), so a row receives its terms ascending in the pivot position of their columns: the entries
    // of the row are taken in that order (selection by repeated minimum; rows are short).
    inf = 0.0;
    for (int i = tid; i < m; i += nt) {
        const int kr = D.pinv[i];
        double acc = rf[kr], rsum = 0.0;
        const int b = D.bt_ptr[i], e = D.bt_ptr[i + 1];
        if (REG && e - b <= 16) {
            forward_row_reg<16>(D, lf, b, e - b, rank, acc, rsum);
        } else if (REG && e - b <= 32) {
            forward_row_reg<32>(D, lf, b, e - b, rank, acc, rsum);
        } else if (e - b <= 256) {
            int last = -1;
            for (int t = b; t < e; t++) {
                int best = 0x7fffffff, bp = -1;
                for (int p = b; p < e; p++) {
                    const int kc = D.qinv[D.bt_idx[p]];
                    if (kc > last && kc < best) {
                        best = kc;
                        bp = p;
                    }
                }
                if (bp < 0 || best >= rank) break;
                const double a = D.bt_val[bp];
                acc = __dsub_rn(acc, __dmul_rn(lf[best], a));
                rsum += fabs(a);
                last = best;
            }
        } else { // a very long row: storage order (the sums then agree with the reference to rounding only)
            for (int p = b; p < e; p++) {
                const int kc = D.qinv[D.bt_idx[p]];
                if (kc < rank) {
                    const double a = D.bt_val[p];
                    acc = __dsub_rn(acc, __dmul_rn(lf[kc], a));
                    rsum += fabs(a);
                }
            }
        }
        if (kr >= rank) {
            acc = acc - lf[kr];
            rsum += 1.0;
        }
        rf[kr] = acc;
        inf = fmax(inf, rsum); // infinity norm = max row sum of |B|
    }
}
__device__ __forceinline__ void stats_tail_finish(const DevG &D, double (*red)[40], double *chain_out, double nl, double nu, double one, double inf)
{
    Scalars *S = D.s;
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id(), nw = num_waves();
    const int m = D.m;
    gdouble_p lf = D.gwork + 2 * (size_t)(m + 1), rf = D.gwork + 3 * (size_t)(m + 1), lb = D.gwork + 4 * (size_t)(m + 1),
              rb = D.gwork + 5 * (size_t)(m + 1);
    // the four 1-norms (residual_test.rs:7-13: a sequential sum over the ROW indices 0..m-1): one wave
    // each, 64 terms fetched together and added in lane order
    for (int q = 0; q < 4; q++) {
        if (w != (nw >= 4 ? q : 0)) continue;
        gdouble_p vec = q == 0 ? lf : (q == 1 ? rf : (q == 2 ? lb : rb));
        // (two chunks in flight: the gather of the next 64 terms is issued before the current 64 are added)
        const auto term = [&](int i0) { return i0 + lane < m ? fabs(vec[D.pinv[i0 + lane]]) : 0.0; };
        const auto cnt = [&](int i0) { return m - i0 < 64 ? (m - i0 < 0 ? 0 : m - i0) : 64; };
        double s = 0.0;
        double a = term(0);
        for (int i0 = 0; i0 < m; i0 += 128) {
            const double b = term(i0 + 64);
            s = wave_ordered_sum(a, cnt(i0), s);
            a = term(i0 + 128);
            s = wave_ordered_sum(b, cnt(i0 + 64), s);
        }
        if (lane == 0) chain_out[4 + q] = s;
    }
    // workgroup reductions: maxima
    double vals[8] = {0.0, 0.0, 0.0, 0.0, nl, nu, one, inf};
    for (int q = 4; q < 8; q++) {
        double v = vals[q];
        v = wave_max_d(v);
        if (lane == 0) red[q & 3][w] = v;
        __syncthreads();
        if (tid == 0) {
            double a = red[q & 3][0];
            for (int ww = 1; ww < nw; ww++) a = fmax(a, red[q & 3][ww]);
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
}
template <bool REG> __device__ __forceinline__ void stats_tail(const DevG &D, const FinishOut &O, double (*red)[40], double *chain_out)
{
    double nl, nu, one, inf;
    stats_tail_loops<REG>(D, O, threadIdx.x, blockDim.x, nl, nu, one, inf);
    __syncthreads();
    stats_tail_finish(D, red, chain_out, nl, nu, one, inf);
}

// do_tail = 0 (a batch): the chains only; their two estimates go to the scalars, k_stats_tail does the rest
__global__ void __launch_bounds__(1024) k_stats(DevLU *Ds, FinishOut *Os, int do_tail)
{
    const DevG D(Ds[blockIdx.x]);
    const FinishOut &O = Os[blockIdx.x];
    Scalars *S = D.s;
    __shared__ double red[4][40];
    __shared__ double chain_out[16];
    const int tid = threadIdx.x, nt = blockDim.x, w = wave_id(), lane = lane_id(), nw = num_waves();
    const int m = D.m;
    if (S->status != ST_DONE || D.skip_stats) return;
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
#ifdef BLU_PROFILE
        const long long t_chain0 = (long long)__builtin_amdgcn_s_memtime();
#endif
        const auto at_aux = [](int, const ColPtr &P) { return P.aux; };
        const auto at_k = [](int k, const ColPtr &) { return k; };
        if (cc == 0) {
            // ---- condest(L): L' x = b with b = +-1 chosen on the fly, k descending (condest.rs:101-116, upper = 0)
            // This chain works in ROW-INDEX coordinates on the stage-ordered L columns (wl[i], i = row of B).
            double x1 = 0.0, xinf = 0.0;
            sweep_dot(CS, m - 1, -1, m, wl, at_aux, [&](int, const ColPtr &P, double dot, double, bool &) {
                double temp = 0.0;
                if (P.e > P.b) temp = -dot; // temp -= work[i]*x
                temp += temp >= 0.0 ? 1.0 : -1.0;
                x1 += fabs(temp);
                xinf = fmax(xinf, fabs(temp));
                return temp;
            });
            // L y = x, k ascending, scatter (condest.rs:135-154)
            double y1 = 0.0;
            sweep_scatter<true>(CS, 0, 1, m, wl, at_aux, [&](int, const ColPtr &, double own) {
                y1 += fabs(own);
                return own;
            });
            if (lane == 0) chain_out[0] = fmax(y1 / x1, xinf); // normest_l_inv
        } else if (cc == 1) {
            // ---- condest(U): U' x = b, k ascending, then U y = x, k descending (upper = 1, pivots = diagonal)
            double x1 = 0.0, xinf = 0.0;
            sweep_dot(CU, 0, 1, m, wu, at_k, [&](int, const ColPtr &P, double dot, double, bool &) {
                double temp = 0.0;
                if (P.e > P.b) temp = -dot;
                temp += temp >= 0.0 ? 1.0 : -1.0;
                temp /= P.diag;
                x1 += fabs(temp);
                xinf = fmax(xinf, fabs(temp));
                return temp;
            });
            double y1 = 0.0;
            sweep_scatter<true>(CU, m - 1, -1, m, wu, at_k, [&](int k, const ColPtr &P, double own) {
                const double temp = own / P.diag;
                if (lane == 0) wu[k] = temp;
                y1 += fabs(temp);
                return temp;
            });
            if (lane == 0) chain_out[1] = fmax(y1 / x1, xinf); // normest_u_inv
        } else if (cc == 2) {
            // ---- residual test, forward system (residual_test.rs:43-66): lhs = L\rhs with rhs = +-1 on the fly.
            // The reference takes row dots of L; the column scatter below adds the same products to each
            // accumulator in the same (ascending stage) order.  lf[k] first accumulates d, then holds lhs.
            sweep_scatter<false>(CL, 0, 1, m, lf, at_k, [&](int k, const ColPtr &, double own) {
                const double d = own;
                const double r = d <= 0.0 ? 1.0 : -1.0;
                const double x = r - d;
                if (lane == 0) {
                    rf[k] = r;
                    lf[k] = x;
                }
                return x;
            });
            // overwrite lhs by U\lhs, k descending (residual_test.rs:57-66)
            sweep_scatter<true>(CU, m - 1, -1, m, lf, at_k, [&](int k, const ColPtr &P, double own) {
                const double d = own / P.diag;
                if (lane == 0) lf[k] = d;
                return d;
            });
        } else {
            // ---- residual test, backward system (residual_test.rs:85-108): lhs = U'\rhs, then L'\lhs
            sweep_dot(CU, 0, 1, m, lb, at_k, [&](int k, const ColPtr &P, double dot, double, bool &) {
                const double d = P.e > P.b ? dot : 0.0;
                const double r = d <= 0.0 ? 1.0 : -1.0;
                if (lane == 0) rb[k] = r;
                return (r - d) / P.diag;
            });
            // dots with the stage-ordered L columns (rows mapped to positions), k descending
            sweep_dot_mapped(CSmap, m - 1, -1, m, lb, at_k, [&](int, const ColPtr &P, double dot, double own, bool &store) {
                store = P.e > P.b;
                return store ? own - dot : own;
            });
        }
#ifdef BLU_PROFILE
        if (lane == 0) printf("k_stats chain %d: %.1f ms at 2.1 GHz\n", cc, ((long long)__builtin_amdgcn_s_memtime() - t_chain0) / 2.1e6);
#endif
    }
    __syncthreads();

#ifdef BLU_STATS_DEBUG
    return; // the work vectors of the chains stay in gwork (blu_hip_dbg_get_gwork)
#endif
    if (!do_tail) {
        if (tid == 0) {
            S->normest_l_inv = chain_out[0];
            S->normest_u_inv = chain_out[1];
        }
        return;
    }
    stats_tail<false>(D, O, red, chain_out);
    // restore the all-zero invariant of the pivot_any work area
    const size_t ng = (size_t)7 * (m + 1);
    for (size_t e = tid; e < ng; e += nt) D.gwork[e] = 0.0;
}

// The second half of the statistics of a batch as a kernel of its own: it streams over the whole matrix like k_finish,
// so it runs one workgroup per CU that takes matrix after matrix (blu_driver.inc: batch_grid) and has the registers for
// forward_row_reg -- inside k_stats it shared the register budget of the chains, which want 24 waves per CU.
template <int NT> __global__ void __launch_bounds__(NT) k_stats_tail(DevLU *Ds, FinishOut *Os, int nmat)
{
    __shared__ double red[4][40];
    __shared__ double chain_out[16];
    for (int b = blockIdx.x; b < nmat; b += gridDim.x) {
        const DevG D(Ds[b]);
        Scalars *S = D.s;
        if (S->status == ST_DONE && !D.skip_stats) { // (uniform)
            if (threadIdx.x == 0) {
                chain_out[0] = S->normest_l_inv;
                chain_out[1] = S->normest_u_inv;
            }
            __syncthreads();
            stats_tail<NT <= 512>(D, Os[b], red, chain_out);
            const size_t ng = (size_t)7 * (D.m + 1);
            for (size_t e = threadIdx.x; e < ng; e += blockDim.x) D.gwork[e] = 0.0;
        }
        __syncthreads();
    }
}
