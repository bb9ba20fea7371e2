// k_sweep.h -- pipelined triangular sweeps shared by k_stats.hip and k_solve.hip (included once each;
// both are part of the single translation unit blu_hip.hip).
#pragma once
#include "blu_dev.h"

typedef GPTR(const long long) gcll_p;
typedef GPTR(const double) gcdouble_p;
// What a sweep step reads at a WAVE-UNIFORM address -- the column pointers of the step, its pivot, its pivot row -- goes
// through the SCALAR data path: pointers into the constant address space make the compiler emit s_load_* for them.  As
// vector loads (all 64 lanes, one address) they went through the CU's vector memory pipeline, which the sweep chains of
// a batch -- 24 waves per CU, seven loads per step -- are bound by (round 4: neither a deeper look-ahead nor fewer ALU
// instructions changed k_stats; loads issued by every lane instead of the column's lanes made it 30 % slower).  The factors
// are read-only in these kernels (written by earlier launches), so the scalar cache holds nothing stale.
#ifdef BLU_EMU_BUILD
#define SWEEP_UNIFORM(T, p) (p)
#else
#define SWEEP_UNIFORM(T, p) ((__attribute__((address_space(4))) const T *)(p))
#endif

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
    int aux, aux2;  // row / column index of the step's pivot, where the sweep needs it
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
        P.b = SWEEP_UNIFORM(long long, colptr)[k];
        P.e = SWEEP_UNIFORM(long long, colptr)[k + 1] - 1;
        P.diag = 0.0;
        P.aux = 0;
        P.aux2 = 0;
        return P;
    }
    __device__ __forceinline__ void diag(ColPtr &P) const { P.diag = SWEEP_UNIFORM(double, value)[P.e]; }
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
        P.b = SWEEP_UNIFORM(long long, colptr)[k] + 1;
        P.e = SWEEP_UNIFORM(long long, colptr)[k + 1];
        P.diag = 1.0;
        P.aux = 0;
        P.aux2 = 0;
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
        P.b = SWEEP_UNIFORM(int, lbeg)[k];
        P.e = SWEEP_UNIFORM(int, lbeg)[k + 1];
        P.diag = 1.0;
        P.aux = SWEEP_UNIFORM(int, prow)[k];
        P.aux2 = 0;
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
    // the same in two halves, for a sweep that fetches them one step apart (sweep_dot_mapped): the row index as stored,
    // then its position
    __device__ __forceinline__ ColEnt ent_raw(const ColPtr &P) const
    {
        ColEnt E;
        E.idx = 0;
        E.val = 0.0;
        const long long p = P.b + lane_id();
        if (p < P.e) {
            E.idx = lidx[p];
            E.val = lval[p];
        }
        return E;
    }
    __device__ __forceinline__ ColEnt ent_map(const ColPtr &P, ColEnt E) const
    {
        if (P.b + lane_id() < P.e) E.idx = map[E.idx];
        return E;
    }
};

// sum of prod over lanes 0..n-1 in lane order, every lane gets it (the reference's sequential loop)
__device__ __forceinline__ double wave_ordered_sum(double prod, int n, double acc)
{
    const unsigned lo = (unsigned)__double_as_longlong(prod), hi = (unsigned)(__double_as_longlong(prod) >> 32);
#define WOS_TERM(T)                                                                                           \
    {                                                                                                         \
        const unsigned a_ = __builtin_amdgcn_readlane(lo, (T)), b_ = __builtin_amdgcn_readlane(hi, (T));      \
        acc = __dadd_rn(acc, __longlong_as_double((long long)(((unsigned long long)b_ << 32) | a_)));         \
    }
    int t = 0;
    for (; t + 4 <= n; t += 4) { // (unrolled: two v_readlane and the add per term, no loop overhead)
        WOS_TERM(t)
        WOS_TERM(t + 1)
        WOS_TERM(t + 2)
        WOS_TERM(t + 3)
    }
    for (; t < n; t++) WOS_TERM(t)
#undef WOS_TERM
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
    if (lane < len && E.idx >= 0) { // (idx < 0: an entry the column set filters out)
        const double pr = __dmul_rn(t, E.val);
        x[E.idx] = SUB ? __dsub_rn(x[E.idx], pr) : __dadd_rn(x[E.idx], pr);
    }
    for (long long off = 64; off < len; off += 64) {
        const ColEnt E2 = C.ent(P, off);
        if (off + lane < len && E2.idx >= 0) {
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


// ------------------------------------------------------------------------------------------------
// The same sweeps with the memory round trip taken OFF the step-to-step dependency.
// (Round 2 tried the look-ahead three steps deep -- gathers and own values fetched three steps ahead, the terms of
// the steps in between kept as products and applied in step order: bit-identical, and SLOWER, k_stats 177 -> 197 ms
// at C3.  A step is not waiting for memory any more: it is ~200 dependent instructions of one wave -- ordered sum,
// f64 division, 64-bit address arithmetic of five prefetches -- at ~7 cycles each.)
//
// sweep_dot: step k forms  s_k = ordered dot(column k, x),  v_k = f(k, P_k, s_k, own_k)  and stores
// x[w_k] = v_k.  The gather x[idx] of step k+1 is issued BEFORE step k is worked (entries two steps
// ahead, pointers three); everything it reads is final except possibly x[w_k], and that value is in a
// register when step k ends: the lanes of step k+1 whose index is w_k take it from there.  So a step costs
// sum + f, not sum + f + a memory round trip.  own_k = x[w_k] before the sweep touches it (the sweeps
// that subtract the dot from their own entry), fetched ahead likewise.
// f: double f(int k, const ColPtr &P, double dot, double own, bool &store)
// widx: int widx(int k, const ColPtr &P)   index of x the step writes (and the index space of the entries)
// ------------------------------------------------------------------------------------------------
template <class Cols, class WIdx, class F>
__device__ __forceinline__ void sweep_dot(const Cols &C, int k0, int dir, int n, gdouble_p x, WIdx widx, F f)
{
    if (n <= 0) return;
    const int lane = lane_id();
    ColPtr Pa = C.ptr(k0);
    C.diag(Pa);
    ColEnt Ea = C.ent(Pa, 0);
    ColPtr Pb = C.ptr(k0 + dir);
    C.diag(Pb);
    ColEnt Eb = C.ent(Pb, 0);
    ColPtr Pc = C.ptr(k0 + 2 * dir);
    double Ga = (lane < Pa.e - Pa.b) ? x[Ea.idx] : 0.0;
    double owna = x[widx(k0, Pa)];
    for (int s = 0, k = k0; s < n; s++, k += dir) {
        // ahead: gather and own value of step k+1, entries of k+2, pointers of k+3
        const int kb = k + dir;
        const bool has_b = s + 1 < n;
        const int wb = widx(kb, Pb);
        double Gb = (has_b && lane < Pb.e - Pb.b) ? x[Eb.idx] : 0.0;
        const double ownb = has_b ? x[wb] : 0.0;
        C.diag(Pc);
        const ColEnt Ec = C.ent(Pc, 0);
        const ColPtr Pd = C.ptr(k + 3 * dir);
        // step k
        const long long len = Pa.e - Pa.b;
        double dot = 0.0;
        if (len > 0) {
            const int n0 = len < 64 ? (int)len : 64;
            dot = wave_ordered_sum(lane < n0 ? __dmul_rn(Ga, Ea.val) : 0.0, n0, 0.0);
            for (long long off = 64; off < len; off += 64) { // long column: the rest straight from memory
                const ColEnt E2 = C.ent(Pa, off);
                const int n2 = (len - off) < 64 ? (int)(len - off) : 64;
                dot = wave_ordered_sum(lane < n2 ? __dmul_rn(x[E2.idx], E2.val) : 0.0, n2, dot);
            }
        }
        bool store = true;
        const double v = f(k, Pa, dot, owna, store);
        const int wa = widx(k, Pa);
        if (store && lane == 0) x[wa] = v;
        if (store && Eb.idx == wa) Gb = v; // the one value the early gather could not have seen
        wave_mem_sync();
        Pa = Pb;
        Ea = Eb;
        Ga = Gb;
        owna = ownb;
        Pb = Pc;
        Eb = Ec;
        Pc = Pd;
    }
}

// sweep_dot over a column set whose entry indices go through a map (LStage with map: row index -> position).  The
// entries of a step are then TWO dependent loads, and fetched two steps ahead as in sweep_dot the second load has to
// wait for the first inside the step that issues both -- a whole round trip on the step-to-step chain (the backward
// residual chain of a batch took 2x the time of the other three for it).  Here the pipeline is one stage deeper:
// pointers four steps ahead, stored indices three, their positions two, the gather of x one.
template <class Cols, class WIdx, class F>
__device__ __forceinline__ void sweep_dot_mapped(const Cols &C, int k0, int dir, int n, gdouble_p x, WIdx widx, F f)
{
    if (n <= 0) return;
    const int lane = lane_id();
    ColPtr Pa = C.ptr(k0);
    ColEnt Ea = C.ent(Pa, 0);
    ColPtr Pb = C.ptr(k0 + dir);
    ColEnt Eb = C.ent(Pb, 0);
    ColPtr Pc = C.ptr(k0 + 2 * dir);
    ColEnt Ecr = C.ent_raw(Pc);
    ColPtr Pd = C.ptr(k0 + 3 * dir);
    double Ga = (lane < Pa.e - Pa.b) ? x[Ea.idx] : 0.0;
    double owna = x[widx(k0, Pa)];
    for (int s = 0, k = k0; s < n; s++, k += dir) {
        // ahead: gather and own value of step k+1, positions of k+2, stored indices of k+3, pointers of k+4
        const int kb = k + dir;
        const bool has_b = s + 1 < n;
        const int wb = widx(kb, Pb);
        double Gb = (has_b && lane < Pb.e - Pb.b) ? x[Eb.idx] : 0.0;
        const double ownb = has_b ? x[wb] : 0.0;
        const ColEnt Ec = C.ent_map(Pc, Ecr);
        const ColEnt Edr = C.ent_raw(Pd);
        const ColPtr Pe = C.ptr(k + 4 * dir);
        // step k
        const long long len = Pa.e - Pa.b;
        double dot = 0.0;
        if (len > 0) {
            const int n0 = len < 64 ? (int)len : 64;
            dot = wave_ordered_sum(lane < n0 ? __dmul_rn(Ga, Ea.val) : 0.0, n0, 0.0);
            for (long long off = 64; off < len; off += 64) { // long column: the rest straight from memory
                const ColEnt E2 = C.ent(Pa, off);
                const int n2 = (len - off) < 64 ? (int)(len - off) : 64;
                dot = wave_ordered_sum(lane < n2 ? __dmul_rn(x[E2.idx], E2.val) : 0.0, n2, dot);
            }
        }
        bool store = true;
        const double v = f(k, Pa, dot, owna, store);
        const int wa = widx(k, Pa);
        if (store && lane == 0) x[wa] = v;
        if (store && Eb.idx == wa) Gb = v; // the one value the early gather could not have seen
        wave_mem_sync();
        Pa = Pb;
        Ea = Eb;
        Ga = Gb;
        owna = ownb;
        Pb = Pc;
        Eb = Ec;
        Pc = Pd;
        Ecr = Edr;
        Pd = Pe;
    }
}

// sweep_scatter: step k takes  t_k = f(k, P_k, own_k)  (own_k = x[w_k] as the earlier steps left it) and
// scatters  x[idx] -/+= t_k * val  over column k.  own_{k+1} is fetched before step k's scatter is issued
// (it then reflects steps < k: a wave's memory operations are performed in order) and the one term step k
// may add to it -- the lane of column k whose index is w_{k+1} -- is applied from registers with the same
// two roundings.  The scatter itself stays a read-modify-write through memory, off the dependency chain.
// f: double f(int k, const ColPtr &P, double own)   (may store the step's own results, e.g. x[w_k])
// ------------------------------------------------------------------------------------------------
template <bool SUB, class Cols, class WIdx, class F>
__device__ __forceinline__ void sweep_scatter(const Cols &C, int k0, int dir, int n, gdouble_p x, WIdx widx, F f)
{
    if (n <= 0) return;
    const int lane = lane_id();
    ColPtr Pa = C.ptr(k0);
    C.diag(Pa);
    ColEnt Ea = C.ent(Pa, 0);
    ColPtr Pb = C.ptr(k0 + dir);
    C.diag(Pb);
    ColEnt Eb = C.ent(Pb, 0);
    ColPtr Pc = C.ptr(k0 + 2 * dir);
    double owna = x[widx(k0, Pa)];
    for (int s = 0, k = k0; s < n; s++, k += dir) {
        const int kb = k + dir;
        const bool has_b = s + 1 < n;
        const int wb = widx(kb, Pb);
        double ownb = has_b ? x[wb] : 0.0; // before this step's scatter is issued
        C.diag(Pc);
        const ColEnt Ec = C.ent(Pc, 0);
        const ColPtr Pd = C.ptr(k + 3 * dir);
        // step k
        const double t = f(k, Pa, owna);
        const long long len = Pa.e - Pa.b;
        const bool mine = lane < len && Ea.idx >= 0; // (idx < 0: an entry the column set filters out)
        if (mine) {
            const double pr = __dmul_rn(t, Ea.val);
            x[Ea.idx] = SUB ? __dsub_rn(x[Ea.idx], pr) : __dadd_rn(x[Ea.idx], pr);
        }
        const unsigned long long hit = __ballot(mine && Ea.idx == wb);
        if (hit) { // step k changes the next step's own entry: the same update, from registers
            const int l = __ffsll((long long)hit) - 1;
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double_as_longlong(Ea.val), l);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(__double_as_longlong(Ea.val) >> 32), l);
            const double pr = __dmul_rn(t, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
            ownb = SUB ? __dsub_rn(ownb, pr) : __dadd_rn(ownb, pr);
        }
        if (len > 64) { // long column: the rest through memory, and the next own entry afterwards
            for (long long off = 64; off < len; off += 64) {
                const ColEnt E2 = C.ent(Pa, off);
                if (off + lane < len && E2.idx >= 0) {
                    const double pr = __dmul_rn(t, E2.val);
                    x[E2.idx] = SUB ? __dsub_rn(x[E2.idx], pr) : __dadd_rn(x[E2.idx], pr);
                }
            }
            wave_mem_sync();
            if (has_b) ownb = x[wb];
        }
        wave_mem_sync();
        Pa = Pb;
        Ea = Eb;
        owna = ownb;
        Pb = Pc;
        Eb = Ec;
        Pc = Pd;
    }
}
