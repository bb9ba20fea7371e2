// k_solve.hip -- solve_dense (src/lu/solve_dense.rs:7-120) for a fresh factorization (nforrest = 0),
// one workgroup per matrix.
//
// The reference's operation order is kept exactly, so the solution is bit-identical to it:
//   forward     L: for k ascending   x = sum over ROW pivotrow[k] of L (row-wise copy, ascending in the
//                                    pivot order of the columns) of work[i]*l;  work[pivotrow[k]] -= x
//               U: for k descending  x = work[pivotrow[k]] / pivot; column pivotcol[k] of U scattered:
//                                    work[i] -= x*u (rows ascending in pivot order);  lhs[pivotcol[k]] = x
//   transposed  U': for k ascending  x = work[pivotcol[k]] / pivot; ROW k of U scattered in production
//                                    order: work[j] -= x*u;  lhs[pivotrow[k]] = x
//               L': for k descending x = sum over stage COLUMN k of L (production order) of lhs[i]*l;
//                                    lhs[pivotrow[k]] -= x
// (work = the right-hand side in the caller's numbering: no permutation pass.)  The four line sets are
// the ones solve_sparse uses (k_solve_sparse.hip): row-wise L from k_build_lt, canonical U columns,
// stage-ordered U rows and L columns as the pivot loop wrote them.
// Each sweep is a chain of m dependent steps on ONE wave; pointers and entries of the next steps are
// fetched ahead (k_sweep.h), the ordered dot is accumulated in lane order.
#include "blu_dev.h"
#include "k_sweep.h"

// row-wise L: the row of the step's pivot row
struct LtRows {
    const int *rptr, *idx;
    const double *v;
    gcint_p prow;
    int m;
    __device__ __forceinline__ ColPtr ptr_of(int i) const
    {
        ColPtr P;
        P.b = rptr[i];
        P.e = rptr[i + 1];
        P.diag = 1.0;
        P.aux = i;
        P.aux2 = 0;
        return P;
    }
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        return ptr_of(prow[k]);
    }
    __device__ __forceinline__ void diag(ColPtr &) const {}
    __device__ __forceinline__ ColEnt ent(const ColPtr &P, long long off) const
    {
        ColEnt E;
        E.idx = 0;
        E.val = 0.0;
        const long long p = P.b + off + lane_id();
        if (p < P.e) {
            E.idx = idx[p];
            E.val = v[p];
        }
        return E;
    }
};
// canonical U column k (rows ascending in pivot order, pivot last), row positions mapped to row indices
struct UColsRow {
    gcll_p colptr, rowidx;
    gcdouble_p value;
    gcint_p prow, pcol;
    int m;
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        ColPtr P;
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        P.b = colptr[k];
        P.e = colptr[k + 1] - 1;
        P.diag = 0.0;
        P.aux = prow[k];
        P.aux2 = pcol[k];
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
            E.idx = prow[(int)rowidx[p]];
            E.val = value[p];
        }
        return E;
    }
};
// stage-ordered U row k (column indices, production order); entries in columns without a pivot are
// absent from the reference's copy (build_factors.rs:323) and skipped here
struct WRows {
    gcint_p ubeg, uidx, qinv, prow, pcol;
    gdouble_p uval;
    gcll_p colptr;
    gcdouble_p value;
    int m, rank;
    __device__ __forceinline__ ColPtr ptr(int k) const
    {
        ColPtr P;
        k = k < 0 ? 0 : (k >= m ? m - 1 : k);
        P.b = k < rank ? ubeg[k] : 0;
        P.e = k < rank ? ubeg[k + 1] : 0;
        P.diag = value[colptr[k + 1] - 1]; // col_pivot
        P.aux = prow[k];
        P.aux2 = pcol[k];
        return P;
    }
    __device__ __forceinline__ void diag(ColPtr &) const {}
    __device__ __forceinline__ ColEnt ent(const ColPtr &P, long long off) const
    {
        ColEnt E;
        E.idx = -1;
        E.val = 0.0;
        const long long p = P.b + off + lane_id();
        if (p < P.e) {
            const int j = uidx[p];
            E.idx = qinv[j] < rank ? j : -1;
            E.val = uval[p];
        }
        return E;
    }
};

__global__ void __launch_bounds__(1024) k_solve_dense(DevLU *Ds, FinishOut *Os, const double *rhs, double *lhs, int trans, const int *lt_ptr,
                                                      const int *lt_idx, const double *lt_val)
{
    const DevG D(Ds[blockIdx.x]);
    const FinishOut &O = Os[blockIdx.x];
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id();
    const int m = D.m;
    const int rank = D.s->rank;
    gdouble_p y = D.txrj;      // m+2 doubles of scratch: work1
    gdouble_p x_out = (gdouble_p)lhs;

    for (int k = tid; k < m; k += nt) y[k] = rhs[k]; // solve_dense.rs:34 / :77
    __syncthreads();
    if (wave_id() != 0) return;

    const auto at_aux = [](int, const ColPtr &P) { return P.aux; };
    const auto at_aux2 = [](int, const ColPtr &P) { return P.aux2; };
    const auto sub_dot = [](int, const ColPtr &P, double dot, double own, bool &store) {
        store = P.e > P.b;
        return store ? own - dot : own;
    };
    if (!trans) {
        const LtRows CL{lt_ptr, lt_idx, lt_val, D.prow, m};
        sweep_dot(CL, 0, 1, m, y, at_aux, sub_dot);
        const UColsRow CU{(gcll_p)O.u_colptr, (gcll_p)O.u_rowidx, (gcdouble_p)O.u_value, D.prow, D.pcol, m};
        sweep_scatter<true>(CU, m - 1, -1, m, y, at_aux, [&](int, const ColPtr &P, double own) {
            const double x = own / P.diag;
            if (lane == 0) x_out[P.aux2] = x;
            return x;
        });
    } else {
        const WRows CW{D.ubeg, D.uidx, D.qinv, D.prow, D.pcol, D.uval, (gcll_p)O.u_colptr, (gcdouble_p)O.u_value, m, rank};
        sweep_scatter<true>(CW, 0, 1, m, y, at_aux2, [&](int, const ColPtr &P, double own) {
            const double x = own / P.diag;
            if (lane == 0) x_out[P.aux] = x;
            return x;
        });
        const LStage CS{D.lbeg, D.lidx, D.prow, nullptr, D.lval, m};
        sweep_dot(CS, m - 1, -1, m, x_out, at_aux, sub_dot);
    }
}
