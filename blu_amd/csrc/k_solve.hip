// k_solve.hip -- solve_dense (src/lu/solve_dense.rs:7-120) for a fresh factorization (nforrest = 0),
// one workgroup per matrix, working on the canonical factors k_finish wrote (B[p,q] = L*U, L unit
// lower CSC with the diagonal first, U upper CSC with the pivot last, both in pivot order).
//
// Round-1 form: the two triangular sweeps are executed by ONE wave, a pivot step at a time (lanes
// over the entries of the step's column).  It is correct and stays on the device, but it is a
// latency chain of m steps; a level-scheduled version is the planned replacement.
// Floating point: the forward system follows the reference's operation order for U (column scatter)
// but uses column scatter for L where the reference uses row dots (and vice versa for the transposed
// system), so results agree to rounding, not bit for bit.
#include "blu_dev.h"

__global__ void __launch_bounds__(1024) k_solve_dense(DevLU *Ds, FinishOut *Os, const double *rhs_all, double *lhs_all, int trans)
{
    const DevG D(Ds[blockIdx.x]);
    const FinishOut &O = Os[blockIdx.x];
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id();
    const int m = D.m;
    const double *rhs = rhs_all + (size_t)blockIdx.x * m;
    double *lhs = lhs_all + (size_t)blockIdx.x * m;
    gdouble_p y = D.txrj; // m+2 doubles of scratch, permuted coordinates

    // gather the right-hand side into pivot order
    for (int k = tid; k < m; k += nt) y[k] = rhs[trans ? O.colperm[k] : O.rowperm[k]];
    __syncthreads();

    if (wave_id() == 0) {
        if (!trans) {
            // L z = y: for k ascending, scatter column k (entries below the diagonal)
            for (int k = 0; k < m; k++) {
                const long long b = O.l_colptr[k] + 1, e = O.l_colptr[k + 1];
                if (e > b) {
                    const double zk = y[k];
                    for (long long p = b + lane; p < e; p += 64) {
                        const int r = (int)O.l_rowidx[p];
                        y[r] = __dsub_rn(y[r], __dmul_rn(zk, O.l_value[p]));
                    }
                    wave_mem_sync();
                }
            }
            // U w = z: for k descending, w_k = z_k / u_kk, scatter the column above the diagonal
            for (int k = m - 1; k >= 0; k--) {
                const long long b = O.u_colptr[k], e = O.u_colptr[k + 1] - 1;
                const double wk = y[k] / O.u_value[e];
                if (lane == 0) y[k] = wk;
                for (long long p = b + lane; p < e; p += 64) {
                    const int r = (int)O.u_rowidx[p];
                    y[r] = __dsub_rn(y[r], __dmul_rn(wk, O.u_value[p]));
                }
                wave_mem_sync();
            }
        } else {
            // U' z = y: for k ascending, z_k = (y_k - sum_{r<k} u_rk z_r) / u_kk (dot with column k)
            for (int k = 0; k < m; k++) {
                const long long b = O.u_colptr[k], e = O.u_colptr[k + 1] - 1;
                double s = 0.0;
                for (long long p = b + lane; p < e; p += 64) s += y[(int)O.u_rowidx[p]] * O.u_value[p];
                if (e > b) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                }
                if (lane == 0) y[k] = (y[k] - s) / O.u_value[e];
                wave_mem_sync();
            }
            // L' w = z: for k descending, w_k = z_k - sum_{r>k} l_rk w_r (dot with column k)
            for (int k = m - 1; k >= 0; k--) {
                const long long b = O.l_colptr[k] + 1, e = O.l_colptr[k + 1];
                if (e > b) {
                    double s = 0.0;
                    for (long long p = b + lane; p < e; p += 64) s += y[(int)O.l_rowidx[p]] * O.l_value[p];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                    if (lane == 0) y[k] = y[k] - s;
                    wave_mem_sync();
                }
            }
        }
    }
    __syncthreads();
    // scatter the solution back to the caller's numbering
    for (int k = tid; k < m; k += nt) lhs[trans ? O.rowperm[k] : O.colperm[k]] = y[k];
}
