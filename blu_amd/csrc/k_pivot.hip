// k_pivot.hip -- the bump factorization: Markowitz search + pivot elimination, persistent kernel,
// ONE workgroup per matrix, the whole pivot loop on the device (no host round trip per pivot).
//
// Reference: factorize_bump (src/lu/factorize_bump.rs:12-49), markowitz (src/lu/markowitz.rs:34-219),
// pivot and its five paths (src/lu/pivot.rs:48-1381), list_* (src/lu/list.rs).
//
// What is parallel: inside one pivot, every column of the pivot row is updated by its own wave
// (lanes over entries; the <= 64 updated positions of pivot_small map to the 64 lanes, the
// cancellation bitmask of pivot.rs:645-664 is one __ballot), every row of the pivot column likewise;
// the Markowitz scan is a wave min-reduction over (cost, position).  What is NOT parallel: pivots.
// Pivot k+1 is chosen from the count lists as pivot k left them, so the loop is a dependent chain.
//
// Result-affecting order is kept exactly:
//   * entries inside a line: compress-keep-order, swap pivot-row entry to front and drop it,
//     append updated entries in pivot-column order (pivot.rs:238-314)
//   * count lists: every touched line is re-appended at the tail, in pivot-row / pivot-column
//     order (pivot.rs:317-325); done here as one batched remove + ordered tail-append per pivot
//   * arithmetic: a = xrj / pivot; w -= a * c with separate roundings (no FMA)
#include "blu_dev.h"

// This file is compiled TWICE into one translation unit (blu_hip.hip), each time inside its own namespace:
//   pv_single   k_pivot_loop        one matrix, one 1024-thread workgroup, the full-size LDS working set
//   pv_batch    k_pivot_loop_batch  many matrices, small workgroups, BLU_CFG_BATCH: a working set sized for the
//                                    shapes that make up practically all pivots, so that many workgroups share a CU
// (shapes beyond the small working set take the general paths, as in the other configuration).
namespace BLU_NS {

// A failed check also raises this LDS flag, so the pivot loop can stop at the next pivot boundary
// without polling the status word in HBM every iteration.
__shared__ int g_pivot_err;
__shared__ int g_pivot_err_line; // source line of a bounded loop that overran (k_pivot_fast.hip: probe_overrun)
#undef DEV_CHECK
#define DEV_CHECK(S, cond)                               \
    do {                                                 \
        if (!(cond)) {                                   \
            set_error((S), ST_ERROR, __LINE__);          \
            g_pivot_err = 1;                             \
        }                                                \
    } while (0)

#if BLU_CFG_WAVE
#include "k_pivot_wave_types.h" // one wave per matrix (k_pivot_wave.hip)
#else
#include "k_pivot_fast_types.h" // (re-included per configuration: no include guard)
#endif

// Diagnostic build (-DBLU_PROFILE, `make prof`): thread 0 stamps the shader clock at phase boundaries
// of the pivot loop.  The product build contains no stamps.
#ifdef BLU_PROFILE
__shared__ long long g_pstamp[48];
#define PROF_STAMP(k)                                                      \
    do {                                                                   \
        if (threadIdx.x == 0) g_pstamp[k] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
// PROF_STAMP_L0: the same from code that a single wave other than wave 0 runs
#define PROF_STAMP_L0(k)                                                           \
    do {                                                                           \
        if (lane_id() == 0) g_pstamp[k] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
// PROF_WAIT: drain this wave's vector-memory operations, so that the stamp that follows separates
// "loads issued + arrived" from the arithmetic behind them (changes the timing a little: diagnostic only)
#define PROF_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PROF_STAMP(k) \
    do {              \
    } while (0)
#define PROF_STAMP_L0(k) \
    do {                 \
    } while (0)
#define PROF_WAIT() \
    do {            \
    } while (0)
#endif

struct alignas(16) Sm {
    Fast fa; // begins with kind, where, anycancel, ncand
    // what every wave needs right after the search barrier sits in 16-byte groups (with fa's first four
    // ints): three LDS reads issued together instead of a chain of read -> branch -> read
    struct alignas(16) {
        int pr, pc, exit_code, head_exit;
    };
    struct alignas(16) {
        int nzc, nzr, pcb, prb;
    };
#ifdef BLU_PROFILE
    long long prof[48];
#else
    long long prof[1];
#endif
    int rank, rankdef, min_colnz, min_rownz;
    int cused, rused, lused, uused;
    int need;
    int flag_small;
    int other_row, where, ncancel, nfill;
    int stop_at, need_search;
    double pivot, other_value;
    long long nsearch, flops, nexpand, d3;
    long long kinds[6];
    long long nfast[4];
    int sh[40];
    long long shl[20];
#if BLU_CFG_WAVE
    unsigned long long wmax[BLU_CFG_WAVE]; // (one or two waves per matrix)
#else
    unsigned long long wmax[16]; // per wave: maximum of a line through an LDS atomic, zero between uses
#endif
#if BLU_CFG_WAVE
    double swork[WV_WCAP]; // the dense work column of the one wave (general paths: 64 entries) = the matrix of old values of k_pivot_wave.hip; all zero between pivots
#else
    double swork[16 * 64]; // one dense work column per wave; LAST member: the batch kernel allocates 4 of the 16
#endif
};

// ------------------------------------------------------------------------------------------------
// list primitives (src/lu/list.rs), single-lane versions
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void list_remove1(const LinkF &flink, const LinkB &blink, int e)
{
    const int f = flink[e], b = blink[e];
    flink[b] = f;
    blink[f] = b;
    flink[e] = e;
    blink[e] = e;
}
__device__ __forceinline__ void list_add1(int e, int list, const LinkF &flink, const LinkB &blink, int nelem)
{
    const int t = blink[nelem + list];
    blink[nelem + list] = e;
    blink[e] = t;
    flink[t] = e;
    flink[e] = nelem + list;
}

// ------------------------------------------------------------------------------------------------
// Batched `for q in order { list_move(elem[q], key[q]) }` (list.rs:89-99) by ONE wave.
// Sequential list_moves leave every list as: untouched elements in their old relative order,
// then the moved ones in move order -- so "unlink all, then tail-append in order" is identical.
// elem(q) / key(q) are read through the two arrays with the given offsets; key < 0 = not moved.
// `mark` is an all-zero int[m] scratch (restored to zero).  Returns min key > 0 (or big).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_list_move_batch(const LinkF &flink, const LinkB &blink, int nelem, gcint_p elems, gcint_p keys, int n,
                                                    gint_p mark, int big)
{
    const int lane = lane_id();
    int minkey = big;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int q = c0 + lane;
        if (q < n && keys[q] >= 0) mark[elems[q]] = 1;
    }
    wave_mem_sync();
    // unlink: the first element of every maximal run of marked neighbours links pred -> succ
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int q = c0 + lane;
        if (q < n && keys[q] >= 0) {
            const int e = elems[q];
            const int p = blink[e];
            const bool prev_marked = p < nelem && mark[p] != 0;
            if (!prev_marked) {
                int nx = flink[e];
                for (int guard = 0; nx < nelem && mark[nx] != 0 && guard <= n; guard++) nx = flink[nx];
                flink[p] = nx;
                blink[nx] = p;
            }
        }
    }
    wave_mem_sync();
    // ordered tail-append, chunk by chunk
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int q = c0 + lane;
        int key = q < n ? keys[q] : -1;
        const int e = key >= 0 ? elems[q] : 0;
        bool act = key >= 0;
        if (act) mark[e] = 0;
        if (act && key > 0) minkey = min(minkey, key);
        unsigned long long active = __ballot(act);
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int k = __shfl(key, leader);
            const unsigned long long grp = __ballot(act && key == k);
            const int tail = blink[nelem + k];
            const unsigned long long below = grp & lanes_below(lane);
            const unsigned long long above = grp & ~((2ull << lane) - 1ull);
            const int prevl = below ? 63 - __clzll((long long)below) : 0;
            const int nextl = above ? __ffsll((long long)above) - 1 : 0;
            const int pe = __shfl(e, prevl), ne = __shfl(e, nextl);
            if (act && key == k) {
                blink[e] = below ? pe : tail;
                flink[e] = above ? ne : nelem + k;
                if (!below) flink[tail] = e;
                if (!above) blink[nelem + k] = e;
                act = false;
            }
            active &= ~grp;
        }
        wave_mem_sync();
    }
    return wave_min_i(minkey);
}

// position of `key` in idx[beg .. beg+len) or -1; all lanes of the wave return the same value
__device__ __forceinline__ int wave_find(gcint_p idx, int beg, int len, int key)
{
    const int lane = lane_id();
    for (int c = 0; c < len; c += 64) {
        const int e = c + lane;
        const bool hit = e < len && idx[beg + e] == key;
        const unsigned long long b = __ballot(hit);
        if (b) return beg + c + __ffsll((long long)b) - 1;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// Markowitz search, columns only (search_rows == 0, the reference default lu.rs:259), ONE wave.
// markowitz.rs:34-123, done() :195-219.
// ------------------------------------------------------------------------------------------------
// COLD marks the general (rarely taken) paths.  They are still inlined: measured on MI355X, keeping
// them out of line (__noinline__) shrinks the kernel from 106 KB to 27 KB of code but makes the loop
// 15 % SLOWER -- the calls force the ~50 descriptor pointers and the live state through scratch
// (156 scratch_load in the loop) -- so the instruction-cache footprint is the lesser evil.
#define COLD __forceinline__
__device__ COLD void markowitz_wave(const DevGP &D, Sm *sm)
{
    const int lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    const long long m64 = m;
    const long long BIG = 0x7fffffffffffffffLL;
    int best_r = -1, best_c = -1;
    long long best = m64 * m64;
    int nsearch = 0, found_min = -1;

    const int h0 = D.cflink[m];
    if (h0 != m) { // empty column in the active submatrix: chosen immediately (markowitz.rs:73-78)
        if (lane == 0) {
            sm->pc = h0;
            sm->pr = -1;
        }
        return;
    }
    int nz = sm->min_colnz;
    DEV_CHECK(S, nz >= 1);
    bool done = false;
    while (nz <= m && !done) {
        // skip empty count lists 64 at a time
        const int k = nz + lane;
        const int h = k <= m ? D.cflink[m + k] : m + k;
        const unsigned long long ne = __ballot(k <= m && h != m + k);
        if (!ne) {
            nz += 64;
            continue;
        }
        const int f = __ffsll((long long)ne) - 1;
        nz += f;
        int j = __shfl(h, f);
        int guard = 0;
        while (j < m) {
            if (++guard > m + 2) { // corrupted list: never spin
                DEV_CHECK(S, false);
                done = true;
                break;
            }
            if (found_min < 0) found_min = nz;
            const int cb = D.cbeg[j], cl = D.clen[j];
            const double cmx = D.colmax[j];
            if (cl != nz || cmx == 0.0 || !(cmx >= D.abstol)) { // reference: assert / D2 endless loop
                DEV_CHECK(S, false);
                done = true;
                break;
            }
            const double tol = fmax(D.abstol, D.reltol * cmx);
            for (int c = 0; c < cl; c += 64) {
                const int e = c + lane;
                const bool v = e < cl;
                const double x = v ? fabs(D.cval[cb + e]) : 0.0;
                const bool el = v && !(x == 0.0 || x < tol);
                const int i = v ? D.cidx[cb + e] : 0;
                long long mc = BIG;
                if (el) mc = (long long)(nz - 1) * (long long)(D.rlen[i] - 1);
                const long long mn = wave_min_ll(mc);
                if (mn < best) { // strict: the first-seen entry wins ties (markowitz.rs:105)
                    const unsigned long long bm = __ballot(el && mc == mn);
                    const int src = __ffsll((long long)bm) - 1;
                    best = mn;
                    best_r = __shfl(i, src);
                    best_c = j;
                }
            }
            DEV_CHECK(S, best < m64 * m64);
            nsearch++;
            if (nsearch >= D.maxsearch) {
                done = true;
                break;
            }
            j = D.cflink[j];
        }
        if (!done) nz++;
    }
    if (lane == 0) {
        sm->pr = best_r;
        sm->pc = best_c;
        sm->nsearch += nsearch;
        if (found_min >= 0) sm->min_colnz = found_min;
    }
}

// Markowitz search with row search enabled (search_rows != 0): verbatim single-lane restatement of
// markowitz.rs:34-193.  Not the default; kept simple.
__device__ COLD void markowitz_serial(const DevGP &D, Sm *sm)
{
    const int m = D.m;
    Scalars *S = D.s;
    const long long m64 = m;
    int pivot_row = -1, pivot_col = -1;
    long long mc64 = m64 * m64;
    int nsearch = 0, min_colnz = -1, min_rownz = -1;
    const int nz_start = min(sm->min_colnz, sm->min_rownz);
    if (D.cflink[m] != m) {
        sm->pc = D.cflink[m];
        sm->pr = -1;
        return;
    }
    bool fin = false;
    for (int nz = nz_start; nz <= m && !fin; nz++) {
        int j = D.cflink[m + nz];
        int guard = 0;
        while (j < m && !fin) {
            if (++guard > m + 2) {
                DEV_CHECK(S, false);
                fin = true;
                break;
            }
            if (min_colnz < 0) min_colnz = nz;
            const double cmx = D.colmax[j];
            if (D.clen[j] != nz || cmx == 0.0 || !(cmx >= D.abstol)) {
                DEV_CHECK(S, false);
                fin = true;
                break;
            }
            const double tol = fmax(D.abstol, D.reltol * cmx);
            for (int pos = D.cbeg[j]; pos < D.cbeg[j] + nz; pos++) {
                const double x = fabs(D.cval[pos]);
                if (x == 0.0 || x < tol) continue;
                const int i = D.cidx[pos];
                const long long nz1 = nz, nz2 = D.rlen[i];
                const long long mc = (nz1 - 1) * (nz2 - 1);
                if (mc < mc64) {
                    mc64 = mc;
                    pivot_row = i;
                    pivot_col = j;
                    if (mc64 <= (nz1 - 1) * (nz1 - 1)) {
                        fin = true;
                        break;
                    }
                }
            }
            if (fin) break;
            nsearch++;
            if (nsearch >= D.maxsearch) {
                fin = true;
                break;
            }
            j = D.cflink[j];
        }
        if (fin) break;
        int i = D.rflink[m + nz];
        guard = 0;
        while (i < m && !fin) {
            if (++guard > m + 2) {
                DEV_CHECK(S, false);
                fin = true;
                break;
            }
            if (min_rownz < 0) min_rownz = nz;
            const int inext = D.rflink[i];
            int cheap = 0, found = 0;
            for (int pos = D.rbeg[i]; pos < D.rbeg[i] + nz; pos++) {
                const int jj = D.ridx[pos];
                const long long nz1 = nz, nz2 = D.clen[jj];
                const long long mc = (nz1 - 1) * (nz2 - 1);
                if (mc >= mc64) continue;
                cheap = 1;
                const double cmx = D.colmax[jj];
                if (cmx == 0.0 || cmx < D.abstol) continue;
                int where = D.cbeg[jj];
                const int wend = D.cbeg[jj] + D.clen[jj] - 1;
                while (where < wend && D.cidx[where] != i) where++;
                const double x = fabs(D.cval[where]);
                if (x >= D.abstol && x >= D.reltol * cmx) {
                    found = 1;
                    mc64 = mc;
                    pivot_row = i;
                    pivot_col = jj;
                    if (mc64 <= nz1 * (nz1 - 1)) {
                        fin = true;
                        break;
                    }
                }
            }
            if (fin) break;
            if (cheap != 0 && found == 0) {
                list_remove1(D.rflink, D.rblink, i);
                list_add1(i, m + 1, D.rflink, D.rblink, m);
            } else {
                nsearch++;
                if (nsearch >= D.maxsearch) {
                    fin = true;
                    break;
                }
            }
            i = inext;
        }
    }
    sm->pr = pivot_row;
    sm->pc = pivot_col;
    sm->nsearch += nsearch;
    if (min_colnz >= 0) sm->min_colnz = min_colnz;
    if (min_rownz >= 0) sm->min_rownz = min_rownz;
}

// ------------------------------------------------------------------------------------------------
// remove_col (pivot.rs:1333-1381): verbatim, one lane (rare: a column maximum fell below abstol)
// ------------------------------------------------------------------------------------------------
__device__ COLD void remove_col_serial(const DevGP &D, Sm *sm, int j)
{
    const int m = D.m;
    const int cbeg = D.cbeg[j], cend = cbeg + D.clen[j];
    for (int pos = cbeg; pos < cend; pos++) {
        const int i = D.cidx[pos];
        int where = D.rbeg[i];
        const int wend = D.rbeg[i] + D.rlen[i] - 1;
        while (where < wend && D.ridx[where] != j) where++;
        const int nl = D.rlen[i] - 1;
        D.rlen[i] = nl;
        D.ridx[where] = D.ridx[D.rbeg[i] + nl];
        if (D.search_rows) {
            list_remove1(D.rflink, D.rblink, i);
            list_add1(i, nl, D.rflink, D.rblink, m);
            if (nl > 0 && nl < sm->min_rownz) sm->min_rownz = nl;
        }
    }
    D.colmax[j] = 0.0;
    D.clen[j] = 0;
    list_remove1(D.cflink, D.cblink, j);
    list_add1(j, 0, D.cflink, D.cblink, m);
}

// ------------------------------------------------------------------------------------------------
// pivot_any / pivot_small: one target column per wave (pivot.rs:219-331 / :566-691)
// q = position of the column in the pivot row (>= 1), work = this wave's dense work column
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gen_update_col(const DevGP &D, Sm *sm, int q, bool small, double *work)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    const int pr = sm->pr, pcb = sm->pcb, cnz1 = sm->nzc - 1;
    const double pivot = sm->pivot;
    const int j = D.ridx[sm->prb + q];
    const int cb = D.cbeg[j], cl = D.clen[j], cap = D.ccap[j];

    // pass 1: scatter entries to be updated into the work column, count the kept ones
    int nkept = 0, where = -1, first_idx = 0;
    double xrj = 0.0, first_val = 0.0, cmxl = 0.0;
    for (int c = 0; c < cl; c += 64) {
        const int e = c + lane;
        const bool v = e < cl;
        const int idx = v ? D.cidx[cb + e] : 0;
        const double val = v ? D.cval[cb + e] : 0.0;
        const int mk = v ? D.rowmark[idx] : 0;
        const bool keep = v && mk == 0;
        if (v && mk > 0) work[mk - 1] = val;
        const unsigned long long kb = __ballot(keep);
        const int t = nkept + wave_prefix_count(kb);
        const bool ispr = keep && idx == pr;
        const unsigned long long pb = __ballot(ispr);
        if (pb) {
            const int src = __ffsll((long long)pb) - 1;
            where = __shfl(t, src);
            xrj = __shfl(val, src);
        }
        const unsigned long long fb = __ballot(keep && t == 0);
        if (fb) {
            const int src = __ffsll((long long)fb) - 1;
            first_idx = __shfl(idx, src);
            first_val = __shfl(val, src);
        }
        if (keep && !ispr) {
            const double x = fabs(val);
            if (x > cmxl) cmxl = x;
        }
        nkept += __popcll(kb);
    }
    DEV_CHECK(S, where >= 0);
    const int nk1 = nkept - 1;
    const int need_max = nk1 + cnz1;
    const bool reloc = need_max > cap;
    int dst = cb, newcap = cap;
    if (reloc) { // file_reappend (file.rs:56-85): move the line to fresh space at the arena end
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->cused, newcap);
        dst = __shfl(nb, 0);
    }
    // pass 2: compress kept entries; kept[0] goes where the pivot-row entry was (the swap of
    // pivot.rs:261-262), the pivot-row entry itself leaves the line (:314)
    if (cl <= 64) { // single chunk: everything still in registers
        const int e = lane;
        const bool v = e < cl;
        const int idx = v ? D.cidx[cb + e] : 0;
        const double val = v ? D.cval[cb + e] : 0.0;
        const bool keep = v && D.rowmark[idx] == 0;
        const unsigned long long kb = __ballot(keep);
        const int t = wave_prefix_count(kb);
        if (keep && t != where && t > 0) {
            D.cidx[dst + t - 1] = idx;
            D.cval[dst + t - 1] = val;
        }
    } else {
        int nk = 0;
        for (int c = 0; c < cl; c += 64) {
            const int e = c + lane;
            const bool v = e < cl;
            const int idx = v ? D.cidx[cb + e] : 0;
            const double val = v ? D.cval[cb + e] : 0.0;
            const bool keep = v && D.rowmark[idx] == 0;
            const unsigned long long kb = __ballot(keep);
            const int t = nk + wave_prefix_count(kb);
            if (keep && t != where && t > 0) {
                D.cidx[dst + t - 1] = idx;
                D.cval[dst + t - 1] = val;
            }
            nk += __popcll(kb);
        }
    }
    if (where > 0 && lane == 0) {
        D.cidx[dst + where - 1] = first_idx;
        D.cval[dst + where - 1] = first_val;
    }
    if (!small) wave_mem_sync(); // work column lives in global memory: scatter above, gather below

    // update and append in pivot-column order
    const double a = xrj / pivot;
    const int put = dst + nk1;
    int nadd = 0;
    unsigned long long mask = 0;
    if (small) {
        const bool p = lane < cnz1;
        double x = 0.0;
        int ridx = 0;
        if (p) {
            x = mulsub(work[lane], a, D.cval[pcb + 1 + lane]);
            ridx = D.cidx[pcb + 1 + lane];
            work[lane] = 0.0;
        }
        const double ax = fabs(x);
        const bool kx = p && ax > D.droptol;
        const unsigned long long kb = __ballot(kx);
        const int d = wave_prefix_count(kb);
        if (kx) {
            D.cidx[put + d] = ridx;
            D.cval[put + d] = x;
            if (ax > cmxl) cmxl = ax;
        }
        mask = __ballot(p && !kx); // cancellation in row position lane+1 (pivot.rs:656-660)
        nadd = __popcll(kb);
    } else {
        for (int c = 0; c < cnz1; c += 64) {
            const int e = c + lane;
            if (e < cnz1) {
                const double x = mulsub(work[e], a, D.cval[pcb + 1 + e]);
                work[e] = 0.0;
                D.cidx[put + e] = D.cidx[pcb + 1 + e];
                D.cval[put + e] = x;
                const double ax = fabs(x);
                if (ax > cmxl) cmxl = ax;
            }
        }
        nadd = cnz1;
    }
    const double cmx = wave_max_d(cmxl);
    if (lane == 0) {
        const int newlen = nk1 + nadd;
        D.cbeg[j] = dst;
        D.clen[j] = newlen;
        D.ccap[j] = newcap;
        D.colmax[j] = cmx;
        D.tnew[q] = newlen;
        D.txrj[q] = xrj;
        D.tmask[q] = mask;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
        if (mask >> 31) atomicAdd((unsigned long long *)&sm->d3, (unsigned long long)__popcll(mask >> 31));
        if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
    }
}

// one target row per wave (pivot.rs:335-398 / :704-771); p = position in the pivot column (>= 1)
__device__ __forceinline__ void gen_update_row(const DevGP &D, Sm *sm, int p, bool small)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    const int pc = sm->pc, prb = sm->prb, rnz1 = sm->nzr - 1;
    const int i = D.cidx[sm->pcb + p];
    const int rb = D.rbeg[i], rl = D.rlen[i], cap = D.rcap[i];

    int nk = 0;
    bool found = false;
    for (int c = 0; c < rl; c += 64) {
        const int e = c + lane;
        const bool v = e < rl;
        const int j = v ? D.ridx[rb + e] : -1;
        const bool keep = v && D.colmark[j] == 0;
        if (__ballot(v && j == pc)) found = true;
        nk += __popcll(__ballot(keep));
    }
    DEV_CHECK(S, found);
    const int need_max = nk + rnz1;
    const bool reloc = need_max > cap;
    int dst = rb, newcap = cap;
    if (reloc) {
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->rused, newcap);
        dst = __shfl(nb, 0);
    }
    int t0 = 0;
    for (int c = 0; c < rl; c += 64) {
        const int e = c + lane;
        const bool v = e < rl;
        const int j = v ? D.ridx[rb + e] : -1;
        const bool keep = v && D.colmark[j] == 0;
        const unsigned long long kb = __ballot(keep);
        const int t = t0 + wave_prefix_count(kb);
        if (keep) D.ridx[dst + t] = j;
        t0 += __popcll(kb);
    }
    // append the pivot row pattern, minus cancelled positions (pivot.rs:752-758)
    int na = 0;
    const int put = dst + nk;
    for (int c = 0; c < rnz1; c += 64) {
        const int q = 1 + c + lane;
        const bool v = q <= rnz1;
        const int jq = v ? D.ridx[prb + q] : 0;
        bool ok = v;
        if (small && v) ok = ((D.tmask[q] >> (p - 1)) & 1ull) == 0;
        const unsigned long long kb = __ballot(ok);
        if (ok) D.ridx[put + na + wave_prefix_count(kb)] = jq;
        na += __popcll(kb);
    }
    if (lane == 0) {
        D.rbeg[i] = dst;
        D.rlen[i] = nk + na;
        D.rcap[i] = newcap;
        D.tnewr[p] = nk + na;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
    }
}

// U row of this stage: pivot-row entries with |xrj| > droptol in pivot-row order (pivot.rs:306-312),
// ONE wave.  elems = pivot row line, positions q0..q1 (inclusive), txrj[q] = value, skipq = position to skip.
__device__ __forceinline__ void wave_write_u(const DevGP &D, Sm *sm, int q0, int q1, int skipq)
{
    const int lane = lane_id();
    int put = sm->uused;
    for (int c = q0; c <= q1; c += 64) {
        const int q = c + lane;
        const bool v = q <= q1 && q != skipq;
        const double x = v ? D.txrj[q] : 0.0;
        const bool k = v && fabs(x) > D.droptol;
        const unsigned long long kb = __ballot(k);
        if (k) {
            const int d = put + wave_prefix_count(kb);
            D.uidx[d] = D.ridx[sm->prb + q];
            D.uval[d] = x;
        }
        put += __popcll(kb);
    }
    if (lane == 0) {
        D.ubeg[sm->rank + 1] = put;
        sm->uused = put;
    }
}

// L column of this stage: x = val / pivot for the pivot-column entries p0..p1 except skipp, kept if
// |x| > droptol (pivot.rs:404-416).  ONE wave.
__device__ __forceinline__ void wave_write_l(const DevGP &D, Sm *sm, int p0, int p1, int skipp)
{
    const int lane = lane_id();
    int put = sm->lused;
    const double pivot = sm->pivot;
    for (int c = p0; c <= p1; c += 64) {
        const int p = c + lane;
        const bool v = p <= p1 && p != skipp;
        const double x = v ? D.cval[sm->pcb + p] / pivot : 0.0;
        const bool k = v && fabs(x) > D.droptol;
        const unsigned long long kb = __ballot(k);
        if (k) {
            const int d = put + wave_prefix_count(kb);
            D.lidx[d] = D.cidx[sm->pcb + p];
            D.lval[d] = x;
        }
        put += __popcll(kb);
    }
    if (lane == 0) {
        D.lbeg[sm->rank + 1] = put;
        sm->lused = put;
    }
}

// ------------------------------------------------------------------------------------------------
// pivot_any / pivot_small, whole workgroup.  Returns false if the kernel must exit (NEED_*).
// ------------------------------------------------------------------------------------------------
__device__ COLD bool pivot_general(const DevGP &D, Sm *sm, bool small)
{
    const int tid = threadIdx.x, nt = blockDim.x, w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    const int pr = sm->pr, pc = sm->pc;
    const int pcb = sm->pcb, prb = sm->prb, cnz1 = sm->nzc - 1, rnz1 = sm->nzr - 1;

    // move pivot to the front of pivot column (swap, pivot.rs:169-170) and pivot row (:185)
    if (w == 0) {
        const int where = wave_find(D.cidx, pcb, cnz1 + 1, pr);
        DEV_CHECK(S, where >= 0);
        if (lane == 0 && where >= 0) {
            const int ti = D.cidx[pcb];
            const double tv = D.cval[pcb];
            const double pv = D.cval[where];
            D.cidx[pcb] = pr;
            D.cval[pcb] = pv;
            D.cidx[where] = ti;
            D.cval[where] = tv;
            sm->pivot = pv;
        }
    }
    if (w == (nw > 1 ? 1 : 0)) {
        const int where = wave_find(D.ridx, prb, rnz1 + 1, pc);
        DEV_CHECK(S, where >= 0);
        if (lane == 0 && where >= 0) {
            const int tj = D.ridx[prb];
            D.ridx[prb] = pc;
            D.ridx[where] = tj;
        }
    }
    __syncthreads();

    // room in the arenas: every updated line may have to be re-appended (pivot.rs:156-208)
    long long gc = 0, gr = 0;
    for (int p = 1 + tid; p <= cnz1; p += nt) {
        const int i = D.cidx[pcb + p];
        const int n = D.rlen[i] + rnz1;
        gr += n + stretch_of(D.stretch, n) + D.pad;
    }
    for (int q = 1 + tid; q <= rnz1; q += nt) {
        const int j = D.ridx[prb + q];
        const int n = D.clen[j] + cnz1;
        gc += n + stretch_of(D.stretch, n) + D.pad;
    }
    gc = block_sum_ll(gc, sm->shl);
    gr = block_sum_ll(gr, sm->shl);
    if ((long long)sm->cused + gc > (long long)D.carena_cap) {
        if (tid == 0) {
            sm->exit_code = ST_NEED_CW;
            sm->need = (int)min(gc, 0x7fffffffLL);
        }
        __syncthreads();
        return false;
    }
    if ((long long)sm->rused + gr > (long long)D.rarena_cap) {
        if (tid == 0) {
            sm->exit_code = ST_NEED_RW;
            sm->need = (int)min(gr, 0x7fffffffLL);
        }
        __syncthreads();
        return false;
    }
    DEV_CHECK(S, sm->pivot != 0.0);

    // marks: rowmark[i] = position in the pivot column (pivot.rs:219-224), colmark[j] = 1 for the
    // pattern of the pivot row (:337-339).  Both arrays are disjoint here, so both sets at once.
    for (int p = 1 + tid; p <= cnz1; p += nt) D.rowmark[D.cidx[pcb + p]] = p;
    for (int q = tid; q <= rnz1; q += nt) D.colmark[D.ridx[prb + q]] = 1;
    if (tid == 0) sm->flag_small = 0;
    __syncthreads();

    // column file update
    double *work = small ? &sm->swork[w * 64] : (double *)&D.gwork[(size_t)w * (m + 1)];
    for (int q = 1 + w; q <= rnz1; q += nw) gen_update_col(D, sm, q, small, work);
    __syncthreads();

    // U row, L column, column count lists: one wave each; every wave then joins the row file update
    if (w == 0) wave_write_u(D, sm, 1, rnz1, -1);
    if (w == 1 % nw) wave_write_l(D, sm, 1, cnz1, -1);
    if (w == 2 % nw) {
        if (lane == 0) list_remove1(D.cflink, D.cblink, pc);
        wave_mem_sync();
        const int mn = wave_list_move_batch(D.cflink, D.cblink, m, D.ridx + prb + 1, D.tnew + 1, rnz1, D.iw2, m + 2);
        if (lane == 0 && mn < sm->min_colnz) sm->min_colnz = mn;
    }
    for (int p = 1 + w; p <= cnz1; p += nw) gen_update_row(D, sm, p, small);
    __syncthreads();

    // cleanup (pivot.rs:418-426)
    for (int p = 1 + tid; p <= cnz1; p += nt) D.rowmark[D.cidx[pcb + p]] = 0;
    for (int q = tid; q <= rnz1; q += nt) D.colmark[D.ridx[prb + q]] = 0;
    if (D.search_rows && w == 0) {
        if (lane == 0) list_remove1(D.rflink, D.rblink, pr);
        wave_mem_sync();
        const int mn = wave_list_move_batch(D.rflink, D.rblink, m, D.cidx + pcb + 1, D.tnewr + 1, cnz1, D.iw2, m + 2);
        if (lane == 0 && mn < sm->min_rownz) sm->min_rownz = mn;
    }
    if (tid == 0) {
        D.colmax[pc] = sm->pivot;
        D.clen[pc] = 0;
        D.rlen[pr] = 0;
        sm->kinds[small ? 3 : 4]++;
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------
// pivot_singleton_row (pivot.rs:835-926)
// ------------------------------------------------------------------------------------------------
__device__ COLD bool pivot_singleton_row(const DevGP &D, Sm *sm)
{
    const int tid = threadIdx.x, w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    const int pr = sm->pr, pc = sm->pc, pcb = sm->pcb, cl = sm->nzc;

    if (w == 0) {
        const int where = wave_find(D.cidx, pcb, cl, pr);
        DEV_CHECK(S, where >= 0);
        if (lane == 0) {
            sm->where = where - pcb;
            sm->pivot = where >= 0 ? D.cval[where] : 1.0;
        }
    }
    __syncthreads();
    const int wherep = sm->where;
    DEV_CHECK(S, sm->pivot != 0.0);
    // L column in column order, pivot skipped; U row empty
    if (w == 0) {
        wave_write_l(D, sm, 0, cl - 1, wherep);
        if (lane == 0) D.ubeg[sm->rank + 1] = sm->uused;
    }
    // remove the pivot column from the row file: last entry moves into the hole (pivot.rs:902-903)
    for (int p = w; p < cl; p += nw) {
        if (p == wherep) continue;
        const int i = D.cidx[pcb + p];
        const int rb = D.rbeg[i], rl = D.rlen[i];
        const int where = wave_find(D.ridx, rb, rl, pc);
        DEV_CHECK(S, where >= 0);
        if (lane == 0 && where >= 0) {
            D.ridx[where] = D.ridx[rb + rl - 1];
            D.rlen[i] = rl - 1;
            D.tnewr[p] = rl - 1;
        }
    }
    if (tid == 0) D.tnewr[wherep] = -1;
    __syncthreads();
    if (w == 0) {
        if (lane == 0) list_remove1(D.cflink, D.cblink, pc);
        if (D.search_rows) {
            if (lane == 0) list_remove1(D.rflink, D.rblink, pr);
            wave_mem_sync();
            const int mn = wave_list_move_batch(D.rflink, D.rblink, m, D.cidx + pcb, D.tnewr, cl, D.iw2, m + 2);
            if (lane == 0 && mn < sm->min_rownz) sm->min_rownz = mn;
        }
        if (lane == 0) {
            D.colmax[pc] = sm->pivot;
            D.clen[pc] = 0;
            D.rlen[pr] = 0;
            sm->kinds[0]++;
        }
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------
// pivot_singleton_col (pivot.rs:928-1025)
// ------------------------------------------------------------------------------------------------
__device__ COLD bool pivot_singleton_col(const DevGP &D, Sm *sm)
{
    const int tid = threadIdx.x, w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    const int pr = sm->pr, pc = sm->pc, pcb = sm->pcb, prb = sm->prb, rl = sm->nzr;

    if (tid == 0) {
        sm->pivot = D.cval[pcb];
        DEV_CHECK(S, D.cidx[pcb] == pr);
    }
    // remove the pivot row from the column file; one column per wave
    for (int q = w; q < rl; q += nw) {
        const int j = D.ridx[prb + q];
        if (j == pc) {
            if (lane == 0) {
                D.tnew[q] = -1;
                sm->where = q;
            }
            continue;
        }
        const int cb = D.cbeg[j], cl = D.clen[j];
        int where = -1;
        double xrj = 0.0, cmxl = 0.0;
        for (int c = 0; c < cl; c += 64) {
            const int e = c + lane;
            const bool v = e < cl;
            const int idx = v ? D.cidx[cb + e] : -1;
            const double val = v ? D.cval[cb + e] : 0.0;
            const unsigned long long hb = __ballot(v && idx == pr);
            if (hb) {
                const int src = __ffsll((long long)hb) - 1;
                where = c + src;
                xrj = __shfl(val, src);
            }
            if (v && idx != pr) {
                const double x = fabs(val);
                if (x > cmxl) cmxl = x;
            }
        }
        DEV_CHECK(S, where >= 0);
        const double cmx = wave_max_d(cmxl);
        if (lane == 0 && where >= 0) {
            D.cidx[cb + where] = D.cidx[cb + cl - 1]; // last entry into the hole (pivot.rs:991-993)
            D.cval[cb + where] = D.cval[cb + cl - 1];
            D.clen[j] = cl - 1;
            D.colmax[j] = cmx;
            D.tnew[q] = cl - 1;
            D.txrj[q] = xrj;
            if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
        }
    }
    __syncthreads();
    DEV_CHECK(S, sm->pivot != 0.0);
    if (w == 0) {
        wave_write_u(D, sm, 0, rl - 1, sm->where);
        if (lane == 0) D.lbeg[sm->rank + 1] = sm->lused; // empty column in L
    }
    if (w == 1 % nw) {
        if (lane == 0) list_remove1(D.cflink, D.cblink, pc);
        wave_mem_sync();
        const int mn = wave_list_move_batch(D.cflink, D.cblink, m, D.ridx + prb, D.tnew, rl, D.iw2, m + 2);
        if (lane == 0) {
            if (mn < sm->min_colnz) sm->min_colnz = mn;
            if (D.search_rows) list_remove1(D.rflink, D.rblink, pr);
            D.colmax[pc] = sm->pivot;
            D.clen[pc] = 0;
            D.rlen[pr] = 0;
            sm->kinds[1]++;
        }
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------
// pivot_doubleton_col (pivot.rs:1027-1331)
// ------------------------------------------------------------------------------------------------
__device__ COLD bool pivot_doubleton_col(const DevGP &D, Sm *sm)
{
    const int tid = threadIdx.x, w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    const int pr = sm->pr, pc = sm->pc, pcb = sm->pcb, prb = sm->prb, rnz1 = sm->nzr - 1;

    // pivot to the front of the pivot column (2 entries) and of the pivot row (pivot.rs:1069-1086)
    if (w == 0) {
        if (lane == 0) {
            if (D.cidx[pcb] != pr) {
                const int ti = D.cidx[pcb];
                const double tv = D.cval[pcb];
                D.cidx[pcb] = D.cidx[pcb + 1];
                D.cval[pcb] = D.cval[pcb + 1];
                D.cidx[pcb + 1] = ti;
                D.cval[pcb + 1] = tv;
            }
            DEV_CHECK(S, D.cidx[pcb] == pr);
            sm->pivot = D.cval[pcb];
            sm->other_row = D.cidx[pcb + 1];
            sm->other_value = D.cval[pcb + 1];
            sm->flag_small = 0;
            sm->ncancel = 0;
        }
    }
    if (w == (nw > 1 ? 1 : 0)) {
        const int where = wave_find(D.ridx, prb, rnz1 + 1, pc);
        DEV_CHECK(S, where >= 0);
        if (lane == 0 && where >= 0) {
            const int tj = D.ridx[prb];
            D.ridx[prb] = pc;
            D.ridx[where] = tj;
        }
    }
    __syncthreads();
    const int other_row = sm->other_row;
    const double pivot = sm->pivot, other_value = sm->other_value;
    DEV_CHECK(S, pivot != 0.0);
    // room for the other row to grow by the fill-in (pivot.rs:1088-1113)
    {
        const int n = D.rlen[other_row] + rnz1;
        const int grow = n + stretch_of(D.stretch, n) + D.pad;
        if ((long long)sm->rused + grow > (long long)D.rarena_cap) {
            if (tid == 0) {
                sm->exit_code = ST_NEED_RW;
                sm->need = grow;
            }
            __syncthreads();
            return false;
        }
    }
    const double ratio = other_value / pivot;

    // column file update (pivot.rs:1115-1222), one column per wave.
    // tmask[q]: bit0 = fill-in created (column goes into the other row's pattern), bit1 = cancelled
    for (int q = 1 + w; q <= rnz1; q += nw) {
        const int j = D.ridx[prb + q];
        const int cb = D.cbeg[j], cl = D.clen[j];
        int where_pivot = -1, where_other = -1;
        double xrj = 0.0, xo = 0.0, cmxl = 0.0;
        for (int c = 0; c < cl; c += 64) {
            const int e = c + lane;
            const bool v = e < cl;
            const int idx = v ? D.cidx[cb + e] : -1;
            const double val = v ? D.cval[cb + e] : 0.0;
            const unsigned long long hp = __ballot(v && idx == pr);
            if (hp) {
                const int src = __ffsll((long long)hp) - 1;
                where_pivot = c + src;
                xrj = __shfl(val, src);
            }
            const unsigned long long ho = __ballot(v && idx == other_row);
            if (ho) {
                const int src = __ffsll((long long)ho) - 1;
                where_other = c + src;
                xo = __shfl(val, src);
            }
            if (v && idx != pr && idx != other_row) {
                const double x = fabs(val);
                if (x > cmxl) cmxl = x;
            }
        }
        DEV_CHECK(S, where_pivot >= 0);
        double cmx = wave_max_d(cmxl);
        if (lane == 0 && where_pivot >= 0) {
            int newkey = -1;
            unsigned long long flags = 0;
            if (where_other < 0) {
                const double x = __dmul_rn(-xrj, ratio); // fill-in element (pivot.rs:1151)
                const double xabs = fabs(x);
                if (xabs > D.droptol) { // stored where the pivot row entry was; count unchanged, no list move
                    D.cidx[cb + where_pivot] = other_row;
                    D.cval[cb + where_pivot] = x;
                    flags = 1;
                    if (xabs > cmx) cmx = xabs;
                } else {
                    const int end = cl - 1;
                    D.cidx[cb + where_pivot] = D.cidx[cb + end];
                    D.cval[cb + where_pivot] = D.cval[cb + end];
                    D.clen[j] = end;
                    newkey = end;
                }
            } else {
                int end = cl - 1;
                D.cidx[cb + where_pivot] = D.cidx[cb + end];
                D.cval[cb + where_pivot] = D.cval[cb + end];
                if (where_other == end) where_other = where_pivot;
                const double nv = __dsub_rn(xo, __dmul_rn(xrj, ratio)); // pivot.rs:1191
                D.cval[cb + where_other] = nv;
                const double x = fabs(nv);
                if (x <= D.droptol) { // numerical cancellation: remove the entry, mark the column
                    end--;
                    D.cidx[cb + where_other] = D.cidx[cb + end];
                    D.cval[cb + where_other] = D.cval[cb + end];
                    flags = 2;
                    D.colmark[j] = 1;
                    atomicAdd(&sm->ncancel, 1);
                } else if (x > cmx) {
                    cmx = x;
                }
                D.clen[j] = end;
                newkey = end;
            }
            D.colmax[j] = cmx;
            D.tnew[q] = newkey;
            D.txrj[q] = xrj;
            D.tmask[q] = flags;
            if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
        }
    }
    __syncthreads();

    // U row; column count lists
    if (w == 0) wave_write_u(D, sm, 1, rnz1, -1);
    if (w == 1 % nw) {
        if (lane == 0) list_remove1(D.cflink, D.cblink, pc);
        wave_mem_sync();
        const int mn = wave_list_move_batch(D.cflink, D.cblink, m, D.ridx + prb + 1, D.tnew + 1, rnz1, D.iw2, m + 2);
        if (lane == 0 && mn < sm->min_colnz) sm->min_colnz = mn;
    }
    // row file update of the other row (pivot.rs:1224-1293), ONE wave
    if (w == 2 % nw) {
        int rb = D.rbeg[other_row], rl = D.rlen[other_row];
        const int ncancel = sm->ncancel;
        if (ncancel != 0) { // ordered compress without the pivot column and the cancelled columns
            int t0 = 0;
            for (int c = 0; c < rl; c += 64) {
                const int e = c + lane;
                const bool v = e < rl;
                const int j = v ? D.ridx[rb + e] : -1;
                const bool drop = v && (j == pc || D.colmark[j] != 0);
                if (drop && j != pc) D.colmark[j] = 0;
                const bool keep = v && !drop;
                const unsigned long long kb = __ballot(keep);
                if (keep) D.ridx[rb + t0 + wave_prefix_count(kb)] = j;
                t0 += __popcll(kb);
            }
            DEV_CHECK(S, rl - t0 == ncancel + 1);
            rl = t0;
        } else { // last entry into the hole
            const int where = wave_find(D.ridx, rb, rl, pc);
            DEV_CHECK(S, where >= 0);
            if (lane == 0 && where >= 0) D.ridx[where] = D.ridx[rb + rl - 1];
            rl--;
            wave_mem_sync();
        }
        // count fill-in, re-append if no room, then append the fill-in columns in pivot-row order
        int nfill = 0;
        for (int c = 0; c < rnz1; c += 64) {
            const int q = 1 + c + lane;
            nfill += __popcll(__ballot(q <= rnz1 && (D.tmask[q] & 1ull)));
        }
        int newcap = D.rcap[other_row];
        if (rl + nfill > newcap) {
            newcap = rl + nfill + stretch_of(D.stretch, rl + nfill) + D.pad;
            int nb = 0;
            if (lane == 0) nb = atomicAdd(&sm->rused, newcap);
            nb = __shfl(nb, 0);
            for (int c = 0; c < rl; c += 64) {
                const int e = c + lane;
                if (e < rl) D.ridx[nb + e] = D.ridx[rb + e];
            }
            rb = nb;
            if (lane == 0) sm->nexpand++;
        }
        int na = 0;
        for (int c = 0; c < rnz1; c += 64) {
            const int q = 1 + c + lane;
            const bool ok = q <= rnz1 && (D.tmask[q] & 1ull);
            const unsigned long long kb = __ballot(ok);
            if (ok) D.ridx[rb + rl + na + wave_prefix_count(kb)] = D.ridx[prb + q];
            na += __popcll(kb);
        }
        if (lane == 0) {
            D.rbeg[other_row] = rb;
            D.rlen[other_row] = rl + na;
            D.rcap[other_row] = newcap;
            // L column (pivot.rs:1295-1305)
            int put = sm->lused;
            const double x = ratio;
            if (fabs(x) > D.droptol) {
                D.lidx[put] = other_row;
                D.lval[put] = x;
                put++;
            }
            D.lbeg[sm->rank + 1] = put;
            sm->lused = put;
            if (D.search_rows) {
                list_remove1(D.rflink, D.rblink, other_row);
                list_add1(other_row, rl + na, D.rflink, D.rblink, m);
                if (rl + na > 0 && rl + na < sm->min_rownz) sm->min_rownz = rl + na;
                list_remove1(D.rflink, D.rblink, pr);
            }
            D.colmax[pc] = pivot;
            D.clen[pc] = 0;
            D.rlen[pr] = 0;
            sm->kinds[2]++;
        }
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------
// the persistent pivot loop: factorize_bump (factorize_bump.rs:12-49) + pivot() (pivot.rs:48-112)
// ------------------------------------------------------------------------------------------------
#if !BLU_CFG_WAVE
#include "k_pivot_fast.hip"
#endif

// set-up of a pivot for the general paths (after the general searches, or for a pivot that was
// pending when the kernel left with NEED_*): line positions and the L/U room check of pivot.rs:70-81
__device__ COLD void setup_pivot_general(const DevGP &D, Sm *sm)
{
    Scalars *S = D.s;
    const int pr = sm->pr, pc = sm->pc;
    sm->fa.kind = 0;
    if (pc < 0 || pr < 0) return;
    sm->pcb = D.cbeg[pc];
    sm->prb = D.rbeg[pr];
    sm->nzc = D.clen[pc];
    sm->nzr = D.rlen[pr];
    sm->flag_small = 0;
    sm->ncancel = 0;
    DEV_CHECK(S, D.pinv[pr] == -1 && D.qinv[pc] == -1);
    DEV_CHECK(S, sm->nzc >= 1 && sm->nzr >= 1);
    if (sm->lused + (sm->nzc - 1) > D.lcap) {
        sm->exit_code = ST_NEED_L;
        sm->need = sm->nzc - 1;
    } else if (sm->uused + (sm->nzr - 1) > D.ucap) {
        sm->exit_code = ST_NEED_U;
        sm->need = sm->nzr - 1;
    }
}

#if BLU_CFG_WAVE
#include "k_pivot_wave.hip" // its own pivot loop and kernel
#else
// BATCH: the 4-wave workgroups of the batch kernel cannot spare a wave for the split list update and the
// early search; leaving that code out also relieves its tighter register budget.
// mc: the metadata cache of the single-matrix kernel (k_pivot_fast_types.h), nullptr in the batch kernel
template <bool BATCH>
__device__ __forceinline__ void pivot_loop_body(DevLU *Ds, int stop_at, Sm *sm, Mc *mc)
{
    const DevGP D(&Ds[blockIdx.x]);
    Scalars *S = D.s;
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const int m = D.m;

    if (S->status != ST_RUNNING) return; // finished or failed in an earlier launch (batch relaunch)
    if (tid == 0) {
        sm->rank = S->rank;
        sm->rankdef = S->rankdef;
        sm->min_colnz = S->min_colnz;
        sm->min_rownz = S->min_rownz;
        sm->cused = S->cused;
        sm->rused = S->rused;
        sm->lused = S->lused;
        sm->uused = S->uused;
        sm->pr = S->pivot_row;
        sm->pc = S->pivot_col;
        sm->exit_code = 0;
        sm->need = 0;
        sm->nsearch = 0;
        sm->flops = 0;
        sm->nexpand = 0;
        sm->d3 = 0;
        sm->stop_at = stop_at;
        sm->fa.kind = 0;
        sm->fa.ewValid = 0;
        sm->fa.spOk = 0;
        for (int k = 0; k < 6; k++) sm->kinds[k] = 0;
#ifdef BLU_PROFILE
        for (int k = 0; k < 48; k++) sm->prof[k] = 0;
        for (int k = 0; k < 48; k++) g_pstamp[k] = 0;
#endif
    }
    for (int k = tid; k < (int)blockDim.x; k += blockDim.x) sm->swork[k] = 0.0; // num_waves() x 64
    if (tid < 16) sm->wmax[tid] = 0ull;
    for (int k = tid; k < 2 * KGMAX; k += blockDim.x) sm->fa.kg[0][k] = 0ull;
    if (tid == 0) {
        g_pivot_err = 0;
        g_pivot_err_line = 0;
    }
    if (mc) mc_reset(D, mc, tid, (int)blockDim.x);
    __syncthreads();

    // Three workgroup barriers per pivot: wave 0 alone runs [record previous pivot -> loop head -> search
    // + set-up] while the other waves wait at the barrier below; the pivot functions hold the other two
    // (after the line updates, after the finalize step).
    long long ew_mcb = 0; // wave 0: per-lane result of the early search, kept for its next search
    int ew_fb = 0;
    for (;;) {
        if (w == 0) {
            // ---- loop head: done / stop / error?
            if (lane == 0) {
                if (g_pivot_err) sm->exit_code = ST_ERROR;
                else if (sm->rank + sm->rankdef >= m) sm->exit_code = ST_DONE;
                else if (sm->stop_at >= 0 && sm->pc < 0 && sm->rank + sm->rankdef >= sm->stop_at) sm->exit_code = ST_STOPPED;
                sm->need_search = sm->pc < 0;
                sm->head_exit = sm->exit_code;
            }
            wave_mem_sync();
            // something wrote list links straight to global memory since the last search (a general pivot path,
            // remove_col, the empty-column step): reload the LDS copies of the list heads
            const bool heads_stale = mc && mc->dirty;
            WAVE_LOCKSTEP(); // (every lane has read the flag before lane 0 clears it)
            if (heads_stale) {
                mc_reset(D, mc, lane, 64);
                wave_mem_sync();
            }
            PROF_STAMP(0);
            // ---- find pivot (skipped when a pivot is pending from a NEED_* exit, factorize_bump.rs:19-21)
            // The searching wave also lays out the pivot (positions, L/U room check, and for the two
            // common pivot kinds the LDS working set of k_pivot_fast.hip) before the barrier.
            if (!sm->head_exit) {
                bool handled = false;
                if (sm->need_search) {
                    if (D.search_rows == 0 && !D.no_fast) handled = markowitz_fast<BATCH>(D, sm, mc, ew_mcb, ew_fb);
                    if (!handled) {
                        if (D.search_rows == 0) markowitz_wave(D, sm);
                        else if (lane == 0) markowitz_serial(D, sm);
                        wave_mem_sync();
                    }
                }
                if (!handled && lane == 0) setup_pivot_general(D, sm);
            }
        }
        __syncthreads();
        const int4 dA = *reinterpret_cast<const int4 *>(&sm->pr);      // pr, pc, exit_code, head_exit
        const int4 dB = *reinterpret_cast<const int4 *>(&sm->nzc);     // nzc, nzr, pcb, prb
        const int4 dC = *reinterpret_cast<const int4 *>(&sm->fa.kind); // kind, where, anycancel, ncand
        if (dA.w) break;
        const int pr = dA.x, pc = dA.y;
        if (pc < 0) { // no pivot found: the reference asserts (factorize_bump.rs:22)
            if (tid == 0) {
                DEV_CHECK(S, false);
                sm->exit_code = ST_ERROR;
            }
            __syncthreads();
            break;
        }
        if (pr < 0) { // eliminate empty column without choosing a pivot (factorize_bump.rs:24-33)
            __syncthreads(); // every thread has read sm->pr / sm->pc before thread 0 rewrites them
            if (tid == 0) {
                list_remove1(D.cflink, D.cblink, pc);
                if (mc) mc->dirty = 1;
                sm->pc = -1;
                sm->rankdef++;
                sm->kinds[5]++;
            }
            __syncthreads();
            continue;
        }

        // ---- pivot(): the room check of pivot.rs:70-81 was made by the searching wave; dispatch (:84-94)
        if (dA.z) break;
        PROF_STAMP(1);
        const int nz_col = dB.x, nz_row = dB.y;
        const int kind = dC.x;
        bool ok = true;
        if (kind == 1) fast_small<BATCH>(D, sm, mc, pr, pc, nz_col, nz_row, ew_mcb, ew_fb);
        else if (kind == 2) fast_scol<BATCH>(D, sm, mc, pr, pc, nz_row, dC.y, ew_mcb, ew_fb);
        else {
            if (mc && tid == 0) mc->dirty = 1; // the general paths work on global memory only
            if (nz_row == 1) ok = pivot_singleton_row(D, sm);
            else if (nz_col == 1) ok = pivot_singleton_col(D, sm);
            else if (nz_col == 2) ok = pivot_doubleton_col(D, sm);
            else ok = pivot_general(D, sm, nz_col - 1 <= 64);
        }
        if (!ok) break; // exit_code set, pivot stays pending
        PROF_STAMP(2);
#ifdef BLU_PROFILE
        // a stamp of a code path that did not run in this pivot is stale (or was never written): only differences of
        // stamps taken in order inside THIS pivot are added
#define PROF_ADD(k, a, b)                                                                                           \
    do {                                                                                                            \
        const long long d_ = g_pstamp[a] - g_pstamp[b];                                                             \
        if (g_pstamp[b] >= g_pstamp[0] && d_ >= 0 && d_ < (1LL << 28)) sm->prof[k] += d_;                            \
    } while (0)
        if (tid == 0) {
            const int kk = sm->fa.kind == 1 ? 1 : (sm->fa.kind == 2 ? 2 : 3);
            PROF_ADD(0, 1, 0);  // search + set-up (incl. barrier)
            PROF_ADD(kk, 2, 1); // pivot: 1 fast small, 2 fast singleton col, 3 general paths
            sm->prof[3 + kk] += 1;                         // counts at 4,5,6
            if (kk == 1) {
                PROF_ADD(7, 3, 1);   // fast small: line updates (rest = finalize)
                // the finalize step, relative to the barrier after the line updates (stamp 3)
                PROF_ADD(22, 6, 3);   // wave 0: U row written
                PROF_ADD(23, 24, 3);  // wave 1: L column written
                PROF_ADD(24, 25, 3);  // wave 2: list update entered
                PROF_ADD(25, 26, 25); //   links + tails loaded
                PROF_ADD(26, 27, 26); //   runs resolved (LDS pointer jumping)
                PROF_ADD(27, 28, 27); //   stores issued
                PROF_ADD(30, 30, 27); //     of which: runs unlinked
                PROF_ADD(31, 31, 30); //     tails resolved, same-key groups found
                PROF_ADD(28, 29, 28); //   stores drained
                PROF_ADD(29, 2, 29);  // list wave done -> all waves past the last barrier
                // the line updates, seen by wave 1 (its first three tasks; the first is a column)
                sm->prof[32] += sm->nzr - 1;                 // tasks: columns
                sm->prof[33] += sm->nzc - 1;                 //        rows
                PROF_ADD(34, 33, 1);  // loads of the first three tasks issued
                PROF_ADD(35, 34, 33); // ... arrived
                PROF_ADD(36, 35, 34); // first task done
                PROF_ADD(37, 36, 35); // second task done
                PROF_ADD(38, 37, 36); // third task done
                PROF_ADD(39, 38, 37); // all of wave 1's tasks done, stores drained
                PROF_ADD(40, 3, 38);  // ... until every wave is past the barrier
                if (g_pstamp[44] > g_pstamp[1]) { // speculative search of the next pivot on the unlink wave, relative to stamp 1
                    PROF_ADD(41, 41, 1);  // columns of the pivot row unlinked
                    PROF_ADD(42, 42, 41); // walk
                    PROF_ADD(43, 43, 42); // staging
                    PROF_ADD(44, 44, 43); // reduction, result published
                    sm->prof[45] += 1;
                }
            }
            if (kk != 3) { // stages of the flattened search (stamps 8..14 set inside markowitz_fast)
                PROF_ADD(8, 8, 0);   // head barrier -> search entered
                PROF_ADD(9, 9, 8);   // walk: list heads + K link/meta loads
                PROF_ADD(10, 10, 9); // candidate entries + row metadata, costs
                PROF_ADD(11, 11, 10); // argmin
                PROF_ADD(12, 12, 11); // pivot column to LDS + pivot row load
                PROF_ADD(13, 13, 12); // column metadata + column hash
                PROF_ADD(14, 14, 13); // row hash + room sums
                PROF_ADD(15, 1, 14); // barrier after the search
                // inside "candidate entries + row metadata": loads drained separately (PROF_WAIT)
                if (g_pstamp[16] > g_pstamp[8]) { // (not a column-singleton search: those take mk_express)
                    PROF_ADD(16, 17, 16); // entries: address arithmetic + load + drain
                    PROF_ADD(17, 18, 17); // row metadata: load + drain
                    PROF_ADD(18, 10, 18); // LDS stores + costs
                    // inside the walk
                    PROF_ADD(19, 19, 8);  // list heads loaded
                    PROF_ADD(20, 20, 19); // first candidate's link + metadata loaded
                    sm->prof[21] += 1;
                }
            }
        }
#endif

        // ---- remove columns whose maximum dropped below abstol (pivot.rs:98-106), record the pivot
        if (tid == 0) {
            const int rank = sm->rank;
            if (sm->flag_small && nz_row > 1) {
                for (int pos = D.ubeg[rank]; pos < D.ubeg[rank + 1]; pos++) {
                    const int j = D.uidx[pos];
                    if (D.colmax[j] == 0.0 || D.colmax[j] < D.abstol) {
                        remove_col_serial(D, sm, j);
                        if (mc) mc->dirty = 1;
                    }
                }
            }
            sm->flops += (long long)(nz_col - 1) * (long long)(nz_row - 1);
            D.pinv[pr] = rank;
            D.qinv[pc] = rank;
            D.prow[rank] = pr;
            D.pcol[rank] = pc;
            sm->pc = -1;
            sm->pr = -1;
            sm->rank = rank + 1;
        }
        // no barrier: the other waves touch nothing until the barrier that follows the next search
    }

    if (tid == 0) {
        S->rank = sm->rank;
        S->rankdef = sm->rankdef;
        S->min_colnz = sm->min_colnz;
        S->min_rownz = sm->min_rownz;
        S->cused = sm->cused;
        S->rused = sm->rused;
        S->lused = sm->lused;
        S->uused = sm->uused;
        S->pivot_row = sm->pr;
        S->pivot_col = sm->pc;
        S->need = sm->need;
        S->nsearch_pivot += sm->nsearch;
        S->factor_flops += sm->flops;
        S->nexpand += sm->nexpand;
        S->d3_hits += sm->d3;
        for (int k = 0; k < 6; k++) S->npivot_kind[k] += sm->kinds[k];
#ifdef BLU_PROFILE
        for (int k = 0; k < 48; k++) S->prof[k] += sm->prof[k];
#endif
        if (sm->exit_code == ST_ERROR && g_pivot_err_line) set_error(S, ST_ERROR, g_pivot_err_line);
        if (S->status == ST_RUNNING) S->status = sm->exit_code;
    }
}

#endif // !BLU_CFG_WAVE
#if BLU_CFG_WAVE
#elif !BLU_CFG_BATCH
// One matrix: all 16 waves of a CU, 128 VGPRs.
__global__ void __launch_bounds__(1024) k_pivot_loop(DevLU *Ds, int stop_at)
{
    __shared__ Sm smem;
    __shared__ Mc mcache;
    pivot_loop_body<false>(Ds, stop_at, &smem, &mcache);
}
#else
// Many matrices (batch): workgroups of <= BLU_BATCH_THREADS threads, so only that many of the 16 per-wave work
// columns at the end of Sm are allocated, and a register budget for BLU_BATCH_WAVES waves per SIMD.  LDS is handed
// out in 1280-byte granules.
#ifndef BLU_BATCH_WAVES
#define BLU_BATCH_WAVES 6
#endif
#ifndef BLU_BATCH_THREADS
#define BLU_BATCH_THREADS 256
#endif
__global__ void __launch_bounds__(BLU_BATCH_THREADS) BLU_WAVES_PER_EU(BLU_BATCH_WAVES, BLU_BATCH_WAVES)
k_pivot_loop_batch(DevLU *Ds, int stop_at)
{
    __shared__ __attribute__((aligned(16))) char raw[sizeof(Sm) - (16 - BLU_BATCH_THREADS / 64) * 64 * sizeof(double)];
    pivot_loop_body<true>(Ds, stop_at, reinterpret_cast<Sm *>(raw), nullptr);
}
#endif

} // namespace BLU_NS
#undef PROF_STAMP
#undef PROF_STAMP_L0
#undef PROF_WAIT
#undef COLD
// the other kernels of the translation unit use the plain check of blu_dev.h (no LDS flag)
#undef DEV_CHECK
#define DEV_CHECK(S, cond)                                   \
    do {                                                     \
        if (!(cond)) set_error((S), ST_ERROR, __LINE__);     \
    } while (0)
