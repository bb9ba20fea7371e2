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
#include <hip/hip_cooperative_groups.h>
#include <rocprim/warp/warp_reduce.hpp>
#include <rocprim/warp/warp_scan.hpp>

#define BLU_WAVE 64

// dynamic LDS of a kernel (sized at the launch); the CPU emulation build (emu/hip/hip_runtime.h) defines its own form
#ifndef BLU_DYN_SHARED
#define BLU_DYN_SHARED(T, name, bytes) extern __shared__ __attribute__((aligned(16))) T name[]
#endif
// Diagnostic build (-DBLU_PROFILE_FILLS, `make fprof`; tools/fill_phases.py): thread 0 of a workgroup of k_prep / k_finish adds
// the shader-clock ticks since its previous stamp to Scalars::prof[k].  The product build contains no stamps.
#ifdef BLU_PROFILE_FILLS
#define FILL_STAMP_BEGIN() long long fill_t0_ = (long long)__builtin_amdgcn_s_memtime()
#define FILL_STAMP(S, k)                                                 \
    do {                                                                 \
        if (threadIdx.x == 0) {                                          \
            const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
            (S)->prof[k] += t_ - fill_t0_;                               \
            fill_t0_ = t_;                                               \
        }                                                                \
    } while (0)
#else
#define FILL_STAMP_BEGIN() \
    do {                   \
    } while (0)
#define FILL_STAMP(S, k) \
    do {                 \
    } while (0)
#endif
// occupancy hint of a kernel (a device-only attribute)
#ifdef BLU_EMU_BUILD
#define BLU_WAVES_PER_EU(lo, hi)
#else
#define BLU_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

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
    int fill_paths;       // (diagnostic) bit 0 / bit 1: the transposing fill of k_prep / k_finish went through buckets (k_bucket.h)
    long long matrix_nz;
    long long bump_nz;
    long long nsearch_pivot;
    long long factor_flops;
    long long nexpand;
    long long ngarbage;
    long long d3_hits;    // cancellations at pivot-column position >= 32 (reference defect D3 would diverge)
    long long npivot_kind[6];  // counters: 0 singleton row, 1 singleton col, 2 doubleton, 3 small, 4 any, 5 empty col
    long long nfast[4];        // k_pivot_loop_wave: pivots taken by its flattened paths (small, singleton col), searches handed over by the previous pivot
    double min_pivot, max_pivot;
    double onenorm, infnorm;
    double norm_l, norm_u, normest_l_inv, normest_u_inv, condest_l, condest_u, residual_test;
    long long prof[48];    // diagnostic build only (-DBLU_PROFILE): shader-clock ticks per phase of the pivot loop
};

// Scalar members of the descriptor: dimensions, parameters (public fields of struct LU,
// src/lu/lu.rs:11-66) and capacities.
#define DEVLU_SCALARS(X)                                                                               \
    X(int, m)                                                                                          \
    X(int, nzbias)      /* -1 = None */                                                                \
    X(int, maxsearch)                                                                                  \
    X(int, pad)                                                                                        \
    X(int, search_rows)                                                                                \
    X(int, no_fast)     /* debug: 1 = general pivot paths only (k_pivot_fast.hip off) */               \
    X(int, skip_stats)  /* 1 = k_stats leaves this handle alone (blu_hip_set_skip_stats) */            \
    X(double, droptol)                                                                                 \
    X(double, abstol)                                                                                  \
    X(double, reltol)                                                                                  \
    X(double, stretch)                                                                                 \
    X(long long, b_i_len)                                                                              \
    X(int, nzcap)                                                                                      \
    X(int, carena_cap)                                                                                 \
    X(int, rarena_cap)                                                                                 \
    X(int, lcap)                                                                                       \
    X(int, ucap)

// Array members (all in HBM):
//   b_*            input matrix as handed over by the caller (device copies of the uint64 arrays)
//   bc_*, bt_*     packed B: columnwise and rowwise-sorted-by-column
//   pinv,qinv      inverse permutations (-1 = not pivoted); prow,pcol = pivot row / column of stage k
//   crec, rrec     LINE RECORDS, one 32-byte LineRec per column / row: begin, length, capacity of the line in its arena
//                  (cidx/cval: column file, index + value; ridx: row file, index only), its two count-list links and --
//                  columns -- its maximum (colmax = col_pivot).  One record = half a 64-byte sector: what a pivot needs
//                  of a line comes with ONE access instead of five or six gathers from as many arrays (round 4; the
//                  gathers were ~45 KB of sectors per small pivot against 11 KB algorithmic).  The kernels read and
//                  write the fields through the views cbeg / clen / ccap / cflink / cblink / colmax, rbeg / rlen / rcap /
//                  rflink / rblink of DevG / DevGP (RecField, LinkField below), or take whole records (crec[j]).
//   chead, rhead   heads of the count lists, m+2 HeadRecs each: list nz is element m+nz of the link space, exactly the
//                  representation of src/lu/list.rs (an element's links live in its record, a head's links here)
//   rowmark,colmark,iw2   all-zero scratch of the general pivot paths
//   tnew,tnewr,txrj,tmask per pivot-row column / pivot-column row results of the general pivot paths
//   gwork          16 * (m+1) doubles: pivot_any dense work columns
//   iw0,iw1        m+2 ints each: prep / finish scratch
//   lbeg..uval     factors in stage order: L column k = lidx/lval[lbeg[k]..lbeg[k+1]), U row k likewise
// HOT = what the low-latency paths of the pivot loop touch at every pivot; COLD = inputs, the packed
// copies of B, and the scratch of the general pivot paths (see DevGP below).
#define DEVLU_ARRAYS_HOT(X)                                                                            \
    X(int, pinv) X(int, qinv) X(int, prow) X(int, pcol)                                                \
    X(LineRec, crec) X(LineRec, rrec) X(HeadRec, chead) X(HeadRec, rhead)                              \
    X(int, cidx) X(double, cval) X(int, ridx)                                                          \
    X(int, lbeg) X(int, ubeg) X(int, lidx) X(int, uidx) X(double, lval) X(double, uval)
#define DEVLU_ARRAYS_COLD(X)                                                                           \
    X(const unsigned long long, b_begin) X(const unsigned long long, b_end) X(const unsigned long long, b_i) \
    X(const double, b_x)                                                                               \
    X(int, bc_ptr) X(int, bc_idx) X(double, bc_val) X(int, bt_ptr) X(int, bt_idx) X(double, bt_val)    \
    X(int, rowmark) X(int, colmark) X(int, tnew) X(int, tnewr) X(double, txrj)                         \
    X(unsigned long long, tmask) X(double, gwork) X(int, iw0) X(int, iw1) X(int, iw2)
#define DEVLU_ARRAYS(X) DEVLU_ARRAYS_HOT(X) DEVLU_ARRAYS_COLD(X)

// One line (column or row) of the active submatrix; `max` is used by columns only (colmax).
struct __attribute__((aligned(32))) LineRec {
    int beg, len, cap;   // the line's entries are arena[beg .. beg+len), room for cap
    int flink, blink;    // count-list links (list.rs), element space 0..m-1, heads m..2m+1
    int spare;
    double max;          // columns: maximum |value| of the column (col_pivot); the pivot once the column is pivotal
};
struct __attribute__((aligned(8))) HeadRec {
    int flink, blink;
};
static_assert(sizeof(LineRec) == 32 && sizeof(HeadRec) == 8, "record layout");
#define LINEREC_BEG 0
#define LINEREC_LEN 4
#define LINEREC_CAP 8
#define LINEREC_FLINK 12
#define LINEREC_BLINK 16
#define LINEREC_MAX 24

// The descriptor as the host fills it and as it lives in HBM: plain (generic) pointers.
struct DevLU {
#define X(T, n) T n;
    DEVLU_SCALARS(X)
#undef X
#define X(T, n) T *n;
    DEVLU_ARRAYS(X)
#undef X
    Scalars *s;
};

// The same descriptor as the kernels use it: every array pointer typed as GLOBAL address space.
// A pointer loaded from a struct in memory is a generic ("flat") pointer to the compiler, and every
// access through it becomes flat_load / flat_store, which are slower than global_* and tie the LDS
// and vector-memory wait counters together.  Kernels build a DevG from their DevLU once.
#define GPTR(T) __attribute__((address_space(1))) T *
typedef GPTR(int) gint_p;
typedef GPTR(const int) gcint_p;
typedef GPTR(double) gdouble_p;
// Views of the line records that index like the arrays they replace: D.cbeg[j] is crec[j].beg, and so on.
template <class T, int OFF> struct RecField {
    GPTR(char) base; // the LineRec array
    __device__ __forceinline__ __attribute__((address_space(1))) T &operator[](int i) const
    {
        return *(GPTR(T))(base + (size_t)(unsigned)i * sizeof(LineRec) + OFF);
    }
};
// count-list links over the link space of list.rs: elements 0..m-1 (their records), heads m..2m+1 (the HeadRec array)
template <int OFF, int HOFF> struct LinkField {
    GPTR(char) rec;  // the LineRec array
    GPTR(char) head; // the HeadRec array
    int m;
    __device__ __forceinline__ __attribute__((address_space(1))) int &operator[](int x) const
    {
        GPTR(char) p = x < m ? rec + (size_t)(unsigned)x * sizeof(LineRec) + OFF : head + (size_t)(unsigned)(x - m) * sizeof(HeadRec) + HOFF;
        return *(GPTR(int))p;
    }
    // the same where the caller knows what x is: an element (x < m) / the head of list k (x = m + k) -- no select, and an
    // element's link then merges with the other fields of its record into one wide load
    __device__ __forceinline__ __attribute__((address_space(1))) int &el(int x) const
    {
        return *(GPTR(int))(rec + (size_t)(unsigned)x * sizeof(LineRec) + OFF);
    }
    __device__ __forceinline__ __attribute__((address_space(1))) int &hd(int k) const
    {
        return *(GPTR(int))(head + (size_t)(unsigned)k * sizeof(HeadRec) + HOFF);
    }
};
typedef RecField<int, LINEREC_BEG> RecBeg;
typedef RecField<int, LINEREC_LEN> RecLen;
typedef RecField<int, LINEREC_CAP> RecCap;
typedef RecField<double, LINEREC_MAX> RecMax;
typedef LinkField<LINEREC_FLINK, 0> LinkF;
typedef LinkField<LINEREC_BLINK, 4> LinkB;
#define DEV_LINE_VIEWS                                                                                   \
    RecBeg cbeg, rbeg;                                                                                   \
    RecLen clen, rlen;                                                                                   \
    RecCap ccap, rcap;                                                                                   \
    RecMax colmax;                                                                                       \
    LinkF cflink, rflink;                                                                                \
    LinkB cblink, rblink;
#define DEV_LINE_VIEWS_INIT                                                                              \
    cbeg{(GPTR(char))crec}, rbeg{(GPTR(char))rrec}, clen{(GPTR(char))crec}, rlen{(GPTR(char))rrec},      \
        ccap{(GPTR(char))crec}, rcap{(GPTR(char))rrec}, colmax{(GPTR(char))crec},                        \
        cflink{(GPTR(char))crec, (GPTR(char))chead, m}, rflink{(GPTR(char))rrec, (GPTR(char))rhead, m},  \
        cblink{(GPTR(char))crec, (GPTR(char))chead, m}, rblink{(GPTR(char))rrec, (GPTR(char))rhead, m}
// atomics on global-address-space pointers (the HIP atomicAdd/atomicMin overloads take generic pointers)
__device__ __forceinline__ int g_atomic_add(gint_p p, int v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int g_atomic_min(gint_p p, int v)
{
    return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct DevG {
#define X(T, n) T n;
    DEVLU_SCALARS(X)
#undef X
#define X(T, n) GPTR(T) n;
    DEVLU_ARRAYS(X)
#undef X
    Scalars *s;
    DEV_LINE_VIEWS
    __device__ __forceinline__ explicit DevG(const DevLU &d)
        :
#define X(T, n) n(d.n),
          DEVLU_SCALARS(X)
#undef X
#define X(T, n) n((GPTR(T))d.n),
              DEVLU_ARRAYS(X)
#undef X
                  s(d.s),
          DEV_LINE_VIEWS_INIT
    {
    }
};

// View for the pivot loop.  The kernel keeps its base pointers in scalar registers; with all ~40 of them
// live the allocator spills SGPRs into VGPR lanes and reloads them (v_readlane) all over the hot path.
// The COLD arrays -- touched only by the general pivot paths, or not at all in this kernel -- are
// therefore not held: a ColdArr fetches its pointer from the descriptor (one scalar load) where it is used.
template <class T> struct ColdArr {
    typedef T *ptr_t;
    const __attribute__((address_space(1))) ptr_t *slot; // where the pointer sits inside the descriptor (global memory)
    __device__ __forceinline__ GPTR(T) get() const { return (GPTR(T))(*slot); }
    __device__ __forceinline__ operator GPTR(T)() const { return get(); }
    __device__ __forceinline__ __attribute__((address_space(1))) T &operator[](int i) const { return get()[i]; }
    __device__ __forceinline__ __attribute__((address_space(1))) T &operator[](size_t i) const { return get()[i]; }
    __device__ __forceinline__ GPTR(T) operator+(int i) const { return get() + i; }
};
struct DevGP {
#define X(T, n) T n;
    DEVLU_SCALARS(X)
#undef X
#define X(T, n) GPTR(T) n;
    DEVLU_ARRAYS_HOT(X)
#undef X
#define X(T, n) ColdArr<T> n;
    DEVLU_ARRAYS_COLD(X)
#undef X
    Scalars *s;
    DEV_LINE_VIEWS
    __device__ __forceinline__ explicit DevGP(const DevLU *d)
        :
#define X(T, n) n(d->n),
          DEVLU_SCALARS(X)
#undef X
#define X(T, n) n((GPTR(T))d->n),
              DEVLU_ARRAYS_HOT(X)
#undef X
#define X(T, n) n{(const __attribute__((address_space(1))) ColdArr<T>::ptr_t *)&d->n},
                  DEVLU_ARRAYS_COLD(X)
#undef X
                      s(d->s),
          DEV_LINE_VIEWS_INIT
    {
    }
};

// ---------------------------------------------------------------------------------------------
// wave64 primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ int num_waves() { return blockDim.x >> 6; }

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }
// number of set bits of a ballot below THIS lane: v_mbcnt_lo / v_mbcnt_hi, two instructions (the shift-mask-popcount
// form costs six; these sit in every compaction of the line updates, which are bound by instruction throughput)
__device__ __forceinline__ int wave_prefix_count(unsigned long long b)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}

// Orders this wave's earlier global/LDS accesses before its later ones, for the lanes of THIS wave:
// needed where one lane's store feeds another lane's later load.  A wave's memory instructions are
// performed in issue order (LDS unit and vector L1 are in-order per wave), so at wavefront scope the
// fence emits no instruction -- it only keeps the compiler from moving accesses across it.  (A
// workgroup-scope fence here would drain vmcnt/lgkmcnt: one store round trip per call.)
// Communication BETWEEN waves goes through __syncthreads().
#ifdef BLU_EMU_BUILD // (a wave barrier that reports its caller)
__device__ __forceinline__ void wave_mem_sync(const char *f = __builtin_FILE(), int l = __builtin_LINE()) { emu_fence("wavefront", f, l); }
#else
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
#endif
// A thread's walk over the entries [b, e) of a line, several entries per turn: load(p) -- everything entry p needs from
// memory, dependent gathers included -- is called for eight (four) entries before use(p, what load gave) is for any, so their
// round trips overlap.  Entry by entry such a walk is a chain of dependent round trips, and the O(nnz) kernels of a
// batch run one workgroup per CU, which hides none of it.  use() is called in entry order.
template <class Load, class Use> __device__ __forceinline__ void line4(int b, int e, Load load, Use use)
{
    int p = b;
    for (; p + 8 <= e; p += 8) { // (eight, then four, then one: 8 against 4 alone is another 1 % of a batch step)
        const auto a0 = load(p), a1 = load(p + 1), a2 = load(p + 2), a3 = load(p + 3), a4 = load(p + 4), a5 = load(p + 5), a6 = load(p + 6), a7 = load(p + 7);
        use(p, a0);
        use(p + 1, a1);
        use(p + 2, a2);
        use(p + 3, a3);
        use(p + 4, a4);
        use(p + 5, a5);
        use(p + 6, a6);
        use(p + 7, a7);
    }
    for (; p + 4 <= e; p += 4) {
        const auto a0 = load(p), a1 = load(p + 1), a2 = load(p + 2), a3 = load(p + 3);
        use(p, a0);
        use(p + 1, a1);
        use(p + 2, a2);
        use(p + 3, a3);
    }
    for (; p < e; p++) use(p, load(p));
}
struct IdxVal { // (what most such walks load per entry)
    int i, g;   // the entry's index, something gathered through it
    double v;
};

// A workgroup barrier that orders LDS only: the waves meet once each has its LDS operations behind it; its global
// stores may still be in flight unless `drain` (a per-wave choice: only a wave whose stores another wave will load has
// to wait for them).  __syncthreads() drains every wave's global stores, ~1-2 k cycles the waves mostly do not owe.
#ifdef BLU_EMU_BUILD
__device__ __forceinline__ void wg_barrier_lds(bool) { __syncthreads(); }
#else
__device__ __forceinline__ void wg_barrier_lds(bool drain)
{
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif
// The lanes of a wave execute every instruction together: "lane A loads X, then lane B stores X" needs nothing on
// the GPU.  The CPU emulation build (emu/hip/hip_runtime.h) runs the lanes of a wave one after the other between two
// collectives; WAVE_LOCKSTEP() marks the places that rely on lockstep and is a wave barrier there, nothing here.
#ifdef BLU_EMU_BUILD
#define WAVE_LOCKSTEP() emu_wave_lockstep()
#else
#define WAVE_LOCKSTEP() \
    do {                \
    } while (0)
#endif

// Broadcast of one lane's value when the source lane is the same for the whole wave (derived from a ballot):
// v_readlane with the lane number in a scalar register -- a few cycles -- instead of __shfl, which compiles
// to ds_bpermute (an LDS-crossbar round trip, ~100 cycles of latency on a dependent chain).
__device__ __forceinline__ int wave_bcast_i(int v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
__device__ __forceinline__ double wave_bcast_d(double v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), s), hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
}

// Wave-wide all-reductions through rocPRIM's DPP implementation (row_shr / row_bcast data-parallel
// primitives: ~6 VALU ops per 32-bit word) instead of __shfl_xor butterflies, which compile to
// ds_bpermute (an LDS-crossbar round trip per step) -- measured 3-4x cheaper on the pivot loop's
// critical path.  Every lane receives the result.
struct OpMinI { __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct OpMaxI { __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct OpSumI { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
struct OpMinLL { __device__ __forceinline__ long long operator()(long long a, long long b) const { return a < b ? a : b; } };
struct OpSumLL { __device__ __forceinline__ long long operator()(long long a, long long b) const { return a + b; } };
// max of non-negative doubles (|x| values): plain compare, NaN never selected (matches `if x > cmx`)
struct OpMaxD { __device__ __forceinline__ double operator()(double a, double b) const { return b > a ? b : a; } };
#ifdef BLU_EMU_BUILD
template <class T, class Op> __device__ __forceinline__ T wave_allreduce(T v, Op op) { return emu_wave_allreduce(v, op); }
#else
template <class T, class Op> __device__ __forceinline__ T wave_allreduce(T v, Op op)
{
    using WR = rocprim::warp_reduce<T, 64, true>;
    typename WR::storage_type st; // empty for the DPP implementation
    T out;
    WR().reduce(v, out, st, op);
    return out;
}
#endif
__device__ __forceinline__ int wave_min_i(int v) { return wave_allreduce(v, OpMinI()); }
__device__ __forceinline__ int wave_max_i(int v) { return wave_allreduce(v, OpMaxI()); }
__device__ __forceinline__ int wave_sum_i(int v) { return wave_allreduce(v, OpSumI()); }
__device__ __forceinline__ long long wave_min_ll(long long v) { return wave_allreduce(v, OpMinLL()); }
__device__ __forceinline__ long long wave_sum_ll(long long v) { return wave_allreduce(v, OpSumLL()); }
__device__ __forceinline__ double wave_max_d(double v) { return wave_allreduce(v, OpMaxD()); }
// inclusive scan over the wave: rocPRIM's DPP implementation (row_shr / row_bcast steps, VALU only) -- a __shfl_up
// ladder is six ds_bpermute round trips through the LDS crossbar, ~700 cycles on a wave that has nothing else to do
#ifdef BLU_EMU_BUILD
struct OpSumIEmu { int lane; };
__device__ __forceinline__ int wave_incl_scan_i(int v)
{
    int acc = 0;
    const int l = lane_id();
    for (int k = 0; k < 64; k++) { // (every lane takes part in all 64 exchanges: uniform control flow)
        const int x = __shfl(v, k);
        if (k <= l) acc += x;
    }
    return acc;
}
#else
__device__ __forceinline__ int wave_incl_scan_i(int v)
{
    using WS = rocprim::warp_scan<int, 64>;
    typename WS::storage_type st; // empty for the DPP implementation
    int out;
    WS().inclusive_scan(v, out, st, rocprim::plus<int>());
    return out;
}
#endif

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

// ---------------------------------------------------------------------------------------------
// Scopes: the O(nnz) phases around the pivot loop (k_prep, k_setup, k_finish) are written once against
// this small interface and instantiated twice:
//   BlockScope  one workgroup per matrix (a batch = a grid of independent workgroups); the sync is
//               __syncthreads(), scans and counters live in LDS.
//   GridScope   ONE matrix on a cooperative launch of many workgroups (the single-basis path): the sync
//               is the grid barrier (agent-scope release / acquire inside), scans combine per-workgroup
//               partials through global memory, counters are global words.  A single workgroup covers
//               one CU; these phases are bandwidth-bound and want the chip.
// Every method must be called by ALL threads of the scope with uniform control flow.
// ---------------------------------------------------------------------------------------------
#define SCOPE_MAX_BLOCKS 256
struct GridWs {                        // global scratch of a GridScope (zeroed by the host before the launch)
    int part[2][SCOPE_MAX_BLOCKS];
    long long partll[2][SCOPE_MAX_BLOCKS];
    int ctr[8];
};
struct BlockScope {
    int *sh;        // 40 ints of LDS
    long long *shl; // 20 long longs of LDS
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nt() const { return blockDim.x; }
    __device__ __forceinline__ int wid() const { return wave_id(); }
    __device__ __forceinline__ int nw() const { return num_waves(); }
    __device__ __forceinline__ bool leader() const { return threadIdx.x == 0; }
    __device__ __forceinline__ int unit() const { return blockIdx.x; } // which matrix of the batch
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ int excl_scan(int v, int *total) { return block_excl_scan_i(v, sh, total); }
    __device__ __forceinline__ long long sum_ll(long long v) { return block_sum_ll(v, shl); }
    __device__ __forceinline__ int any(int v) { return block_or_i(v, sh); }
    __device__ __forceinline__ int *ctr(int k) const { return &sh[34 + k]; } // k = 0..4; zero it, sync, then atomicAdd
    // which share of the column / row count lists this wave builds (k_setup): up to three waves -- wave 0 all column lists,
    // wave 1 all row lists
    __device__ __forceinline__ void list_roles(int &part, int &nparts, bool &cols, bool &rows) const
    {
        const int nw = num_waves(), w = wave_id();
        if (nw >= 4) { // (half of the waves the column lists, half the row lists, by key modulo their number)
            nparts = nw / 2;
            cols = w < nparts;
            rows = w >= nparts && w < 2 * nparts;
            part = cols ? w : w - nparts;
            return;
        }
        part = 0;
        nparts = 1;
        cols = wave_id() == 0;
        rows = num_waves() == 1 ? wave_id() == 0 : wave_id() == 1;
    }
    // maximum / minimum of NON-NEGATIVE doubles over the scope
    __device__ __forceinline__ double max_d(double v, double *shd)
    {
        v = wave_max_d(v);
        if (lane_id() == 0) shd[wave_id()] = v;
        __syncthreads();
        double r = shd[0];
        for (int w = 1; w < num_waves(); w++) r = r > shd[w] ? r : shd[w];
        __syncthreads();
        return r;
    }
    __device__ __forceinline__ double min_d(double v, double *shd)
    {
        v = -wave_max_d(-v); // (OpMaxD is a plain compare: fine for non-positive values too)
        if (lane_id() == 0) shd[wave_id()] = v;
        __syncthreads();
        double r = shd[0];
        for (int w = 1; w < num_waves(); w++) r = r < shd[w] ? r : shd[w];
        __syncthreads();
        return r;
    }
};
struct GridScope {
    int *sh;
    long long *shl;
    GridWs *g;
    int parity;
    __device__ __forceinline__ int tid() const { return blockIdx.x * blockDim.x + threadIdx.x; }
    __device__ __forceinline__ int nt() const { return gridDim.x * blockDim.x; }
    __device__ __forceinline__ int wid() const { return blockIdx.x * num_waves() + wave_id(); }
    __device__ __forceinline__ int nw() const { return gridDim.x * num_waves(); }
    __device__ __forceinline__ bool leader() const { return blockIdx.x == 0 && threadIdx.x == 0; }
    __device__ __forceinline__ int unit() const { return 0; }
    __device__ __forceinline__ void sync() { cooperative_groups::this_grid().sync(); }
    // per-workgroup partials through global memory; two buffers alternate so that one grid barrier per call is enough
    __device__ __forceinline__ int excl_scan(int v, int *total)
    {
        int bt;
        const int ex = block_excl_scan_i(v, sh, &bt);
        if (threadIdx.x == 0) __hip_atomic_store(&g->part[parity][blockIdx.x], bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sync();
        if (wave_id() == 0) {
            int before = 0, all = 0;
            for (int b = lane_id(); b < (int)gridDim.x; b += 64) {
                const int x = __hip_atomic_load(&g->part[parity][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all += x;
                if (b < (int)blockIdx.x) before += x;
            }
            before = wave_sum_i(before);
            all = wave_sum_i(all);
            if (lane_id() == 0) {
                sh[36] = before;
                sh[37] = all;
            }
        }
        __syncthreads();
        const int off = sh[36];
        *total = sh[37];
        __syncthreads();
        parity ^= 1;
        return off + ex;
    }
    __device__ __forceinline__ long long sum_ll(long long v)
    {
        const long long bs = block_sum_ll(v, shl);
        if (threadIdx.x == 0) __hip_atomic_store(&g->partll[parity][blockIdx.x], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sync();
        long long all = 0;
        if (wave_id() == 0) {
            for (int b = lane_id(); b < (int)gridDim.x; b += 64)
                all += __hip_atomic_load(&g->partll[parity][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            all = wave_sum_ll(all);
            if (lane_id() == 0) shl[17] = all;
        }
        __syncthreads();
        const long long r = shl[17];
        __syncthreads();
        parity ^= 1;
        return r;
    }
    __device__ __forceinline__ int any(int v)
    {
        int tot;
        (void)excl_scan(v ? 1 : 0, &tot);
        return tot != 0;
    }
    __device__ __forceinline__ int *ctr(int k) const { return &g->ctr[k]; }
    // one builder wave per workgroup; the first half of the workgroups share out the column lists (by key modulo
    // their number), the second half the row lists.  Every builder scans all keys, so few builders beat many.
    __device__ __forceinline__ void list_roles(int &part, int &nparts, bool &cols, bool &rows) const
    {
        const int G = gridDim.x, b = blockIdx.x, half = G / 2;
        if (G == 1) {
            part = 0;
            nparts = 1;
            cols = wave_id() == 0;
            rows = wave_id() == 1;
            return;
        }
        nparts = half;
        part = b < half ? b : b - half;
        cols = wave_id() == 0 && b < half;
        rows = wave_id() == 0 && b >= half && b < 2 * half;
    }
    // maximum / minimum of NON-NEGATIVE doubles over the scope: they order like their bit patterns
    __device__ __forceinline__ double max_d(double v, double *shd)
    {
        v = wave_max_d(v);
        if (lane_id() == 0) shd[wave_id()] = v;
        __syncthreads();
        double r = shd[0];
        for (int w = 1; w < num_waves(); w++) r = r > shd[w] ? r : shd[w];
        __syncthreads();
        return __longlong_as_double(-min_ll_blocks(-__double_as_longlong(r)));
    }
    __device__ __forceinline__ double min_d(double v, double *shd)
    {
        v = -wave_max_d(-v);
        if (lane_id() == 0) shd[wave_id()] = v;
        __syncthreads();
        double r = shd[0];
        for (int w = 1; w < num_waves(); w++) r = r < shd[w] ? r : shd[w];
        __syncthreads();
        return __longlong_as_double(min_ll_blocks(__double_as_longlong(r)));
    }
    // min over the grid of one value per workgroup (passed by every thread of the workgroup alike)
    __device__ __forceinline__ long long min_ll_blocks(long long v)
    {
        if (threadIdx.x == 0) __hip_atomic_store(&g->partll[parity][blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sync();
        long long mn = 0x7fffffffffffffffLL;
        if (wave_id() == 0) {
            for (int b = lane_id(); b < (int)gridDim.x; b += 64) {
                const long long x = __hip_atomic_load(&g->partll[parity][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mn = x < mn ? x : mn;
            }
            mn = wave_min_ll(mn);
            if (lane_id() == 0) shl[17] = mn;
        }
        __syncthreads();
        const long long r = shl[17];
        __syncthreads();
        parity ^= 1;
        return r;
    }
};

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
