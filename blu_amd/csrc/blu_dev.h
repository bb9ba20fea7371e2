// blu_dev.h -- device-side state layout and wave/workgroup primitives (gfx950, wave64).
//
// One factorization = one `DevLU` (all state resident in HBM) worked on by ONE
// workgroup; a batch is a grid of workgroups, one per DevLU.  Because a matrix
// never leaves its workgroup (= one CU, one vector L1), every hand-off between
// waves is a plain store -> __syncthreads() -> plain load; no agent-scope
// fences are needed anywhere in this library.
//
// Layout freedom: the reference keeps row/column "files" with gaps, a memory-order
// linked list and a compaction pass (src/lu/file.rs).  Results do not depend on that
// layout (file_reappend/file_compress preserve line and entry order), so the device
// keeps per-line (begin,len,cap) triples over bump-pointer arenas instead and 32-bit
// indices (m, nnz < 2^31); order of entries inside a line and order of elements inside
// the count lists -- the things results DO depend on -- follow the reference exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BLU_WAVE 64

// Kernel exit / progress codes (Scalars::status)
enum {
    ST_RUNNING = 0,
    ST_DONE = 1,          // pivot loop finished (rank + rankdef == m)
    ST_STOPPED = 2,       // debug step limit reached
    ST_NEED_L = 3,        // L storage exhausted   (host grows, relaunches)
    ST_NEED_U = 4,        // U storage exhausted
    ST_NEED_CW = 5,       // column arena exhausted (host compacts/grows, relaunches)
    ST_NEED_RW = 6,       // row arena exhausted
    ST_INVALID_ARG = 7,   // singletons.rs:119-201 checks failed
    ST_ERROR = 8          // invariant violated (a reference assert! would have fired)
};

struct Scalars {
    int status;
    int err_line;
    int rank;
    int rankdef;
    int rank0;            // rank after the singleton phase
    int min_colnz;
    int min_rownz;
    int pivot_row;        // pending pivot (kept across a NEED_* exit), -1 = none
    int pivot_col;
    int bump_size;
    int cused, rused;     // arena bump pointers
    int lused, uused;     // L / U entries written
    int need;             // entries requested by a NEED_* exit
    int l_nz, u_nz;       // final counts (build_factors)
    int pad0;
    long long matrix_nz;
    long long bump_nz;
    long long nsearch_pivot;
    long long factor_flops;
    long long nexpand;
    long long ngarbage;
    long long d3_hits;    // cancellations at pivot-column position >= 32 (reference defect D3 would diverge)
    long long npivot_kind[6]; // counters: 0 singleton row, 1 singleton col, 2 doubleton, 3 small, 4 any, 5 empty col
    double min_pivot, max_pivot;
    double onenorm, infnorm;
    double norm_l, norm_u, normest_l_inv, normest_u_inv, condest_l, condest_u, residual_test;
    long long prof[8];    // diagnostic build only (-DBLU_PROFILE): shader-clock ticks per phase of the pivot loop
};

struct DevLU {
    // dimensions and parameters (public fields of struct LU, src/lu/lu.rs:11-66)
    int m;
    int nzbias;           // -1 = None
    int maxsearch;
    int pad;
    int search_rows;
    int no_fast;          // debug: 1 = general pivot paths only (k_pivot_fast.hip off)
    double droptol, abstol, reltol, stretch;

    // input matrix as handed over by the caller (device copies of the uint64 arrays)
    const unsigned long long *b_begin, *b_end, *b_i;
    const double *b_x;
    long long b_i_len;

    // packed B: columnwise (bc_*) and rowwise sorted by column (bt_*)
    int *bc_ptr, *bc_idx;
    double *bc_val;
    int *bt_ptr, *bt_idx;
    double *bt_val;
    int nzcap;

    // pivot sequence
    int *pinv, *qinv;     // inverse permutations (-1 = not pivoted)
    int *prow, *pcol;     // pivot row / column of stage k

    // active submatrix: column file (index+value), row file (index only)
    int *cbeg, *clen, *ccap;
    int *cidx;
    double *cval;
    int carena_cap;
    int *rbeg, *rlen, *rcap;
    int *ridx;
    int rarena_cap;
    double *colmax;       // col_pivot in the reference: column maximum, later the pivot

    // count lists, same representation as src/lu/list.rs (heads at m+nz), 2m+2 entries
    int *cflink, *cblink, *rflink, *rblink;

    // scratch, all-zero between pivots
    int *rowmark, *colmark;
    int *tnew;            // per pivot-row column: new column count
    int *tnewr;           // per pivot-column row: new row count
    double *txrj;         // per pivot-row column: pivot-row entry
    unsigned long long *tmask; // per pivot-row column: cancellation mask (pivot_small)
    double *gwork;        // nwaves * (m+1) doubles: pivot_any dense work columns
    int *iw0, *iw1, *iw2; // m+2 ints each: prep/finish scratch

    // factors, stage order: L column k = lidx/lval[lbeg[k]..lbeg[k+1]), U row k likewise
    int *lbeg, *ubeg;     // m+1
    int *lidx, *uidx;
    double *lval, *uval;
    int lcap, ucap;

    Scalars *s;
};

// ---------------------------------------------------------------------------------------------
// wave64 primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ int num_waves() { return blockDim.x >> 6; }

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Orders this wave's earlier global/LDS accesses before its later ones (s_waitcnt vmcnt(0) lgkmcnt(0))
// and is a compiler barrier.  Needed where one lane's store feeds another lane's later load.
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ long long wave_min_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// max of non-negative doubles (|x| values): plain compare, NaN never selected (matches `if x > cmx`)
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}
// inclusive scan over the wave
__device__ __forceinline__ int wave_incl_scan_i(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (l >= o) v += t;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// workgroup primitives (blockDim.x multiple of 64, <= 1024).  `sh` = 34 ints of LDS.
// ---------------------------------------------------------------------------------------------
// Exclusive scan of one value per thread; returns the exclusive prefix, *total = workgroup sum.
__device__ __forceinline__ int block_excl_scan_i(int v, int *sh, int *total)
{
    const int l = lane_id(), w = wave_id(), nw = num_waves();
    int inc = wave_incl_scan_i(v);
    if (l == 63) sh[w] = inc;
    __syncthreads();
    if (w == 0) {
        int x = l < nw ? sh[l] : 0;
        int xi = wave_incl_scan_i(x);
        if (l < nw) sh[l] = xi - x;
        if (l == nw - 1) sh[32] = xi;
    }
    __syncthreads();
    int res = inc - v + sh[w];
    *total = sh[32];
    __syncthreads();
    return res;
}
__device__ __forceinline__ long long block_sum_ll(long long v, long long *shl)
{
    const int l = lane_id(), w = wave_id(), nw = num_waves();
    long long s = wave_sum_ll(v);
    if (l == 0) shl[w] = s;
    __syncthreads();
    if (w == 0) {
        long long x = l < nw ? shl[l] : 0;
        x = wave_sum_ll(x);
        if (l == 0) shl[16] = x;
    }
    __syncthreads();
    long long r = shl[16];
    __syncthreads();
    return r;
}
__device__ __forceinline__ int block_or_i(int v, int *sh)
{
    if (threadIdx.x == 0) sh[33] = 0;
    __syncthreads();
    if (v) atomicOr(&sh[33], v);
    __syncthreads();
    int r = sh[33];
    __syncthreads();
    return r;
}

// status helpers: first error wins
__device__ __forceinline__ void set_error(Scalars *s, int st, int line)
{
    if (atomicCAS(&s->status, ST_RUNNING, st) == ST_RUNNING) s->err_line = line;
}
#define DEV_CHECK(S, cond)                                   \
    do {                                                     \
        if (!(cond)) set_error((S), ST_ERROR, __LINE__);     \
    } while (0)

// Rust `(stretch * n as f64) as usize`
__device__ __forceinline__ int stretch_of(double stretch, int n) { return (int)(stretch * (double)n); }

// strict IEEE mul-sub without contraction: w - a*c   (pivot.rs:287-291 `work[pos] -= a * col[pos]`)
__device__ __forceinline__ double mulsub(double w, double a, double c) { return __dsub_rn(w, __dmul_rn(a, c)); }
