// k_finish.hip -- after the pivot loop; written against the Scope interface of blu_dev.h (one workgroup per
// matrix in a batch: k_finish; the whole chip for a single matrix: k_finish_grid):
//   k_finish  = the result-defining part of build_factors (src/lu/build_factors.rs:179-223, 395-419)
//               fused with get_factors (src/get_factors.rs:48-180): completes the permutations and
//               writes the canonical read-out (L: CSC, unit diagonal first, rows sorted; U: CSC,
//               pivot last, rows sorted; both in pivot order) straight from the stage-ordered L
//               columns / U rows the pivot loop produced.  The reference's intermediate row-wise L
//               and column-wise U copies (build_factors.rs:229-384) exist there only to make that
//               read-out (and the CPU solves) sequential; they are not materialised here.
//   k_compact = file_compress (src/lu/file.rs:92-135) for the bump-pointer arenas: lines are copied
//               in index order into a fresh arena with stretch*len+pad room each.
#include "blu_dev.h"
#include "k_bucket.h"

struct FinishOut {
    long long *rowperm, *colperm;             // m
    long long *l_colptr, *l_rowidx;           // m+1, l_nz+m
    double *l_value;
    long long *u_colptr, *u_rowidx;           // m+1, u_nz+m
    double *u_value;
};

// sort the (key,val) pairs in [b,e) ascending by key; keys distinct.  One thread.
__device__ __forceinline__ void insertion_sort_pairs(long long *key, double *val, int b, int e)
{
    for (int p = b + 1; p < e; p++) {
        const long long k = key[p];
        const double v = val[p];
        int q = p - 1;
        while (q >= b && key[q] > k) {
            key[q + 1] = key[q];
            val[q + 1] = val[q];
            q--;
        }
        key[q + 1] = k;
        val[q + 1] = v;
    }
}


// The same for a segment of at most SSORT_MAX pairs, one thread, without a dependent chain of memory accesses: all
// pairs are loaded into registers first (the loads are issued together), every pair is ranked against the others in
// registers and stored where it belongs.  (The insertion sort above goes through memory for every comparison: ~n^2/4
// dependent round trips per segment -- 0.14 of the 0.48 s of k_finish for 1536 bases of the 100k size.  Registers are
// free here: these kernels run one workgroup per CU.)
#define SSORT_MAX 32
__device__ __forceinline__ void small_sort_pairs(long long *key, double *val, int b, int e)
{
    const int n = e - b;
    int k[SSORT_MAX];
    double v[SSORT_MAX];
#pragma unroll
    for (int i = 0; i < SSORT_MAX; i++) {
        k[i] = 0x7fffffff; // (keys are indices < 2^31 - 1: the padding is never below a key)
        v[i] = 0.0;
        if (i < n) {
            k[i] = (int)key[b + i];
            v[i] = val[b + i];
        }
    }
#pragma unroll
    for (int i = 0; i < SSORT_MAX; i++) {
        if (i < n) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < SSORT_MAX; j++) r += (k[j] < k[i]) ? 1 : 0;
            key[b + r] = k[i];
            val[b + r] = v[i];
        }
    }
}

// L column k of the canonical factors in ONE pass for a column of at most SSORT_MAX entries: stage entries and their new
// row numbers (pinv) into registers, rank there, store every pair where it belongs -- the unsorted copy that small_sort_pairs
// reads back and rewrites never goes through memory (k_finish of a batch is bound by its HBM sector traffic: round 4).
__device__ __forceinline__ void small_sort_gather(gcint_p lidx, gdouble_p lval, gcint_p pinv, int b, int e, long long *okey, double *oval, int ob)
{
    const int n = e - b;
    int k[SSORT_MAX];
    double v[SSORT_MAX];
#pragma unroll
    for (int i = 0; i < SSORT_MAX; i++) {
        k[i] = 0x7fffffff; // (new row numbers are < 2^31 - 1: the padding is never below a key)
        v[i] = 0.0;
        if (i < n) {
            k[i] = lidx[b + i];
            v[i] = lval[b + i];
        }
    }
#pragma unroll
    for (int i = 0; i < SSORT_MAX; i++)
        if (i < n) k[i] = pinv[k[i]];
#pragma unroll
    for (int i = 0; i < SSORT_MAX; i++) {
        if (i < n) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < SSORT_MAX; j++) r += (k[j] < k[i]) ? 1 : 0;
            okey[ob + r] = k[i];
            oval[ob + r] = v[i];
        }
    }
}

// Rank sort of one segment [b,e) of at most WSORT_MAX pairs by ONE wave, staged through this wave's
// LDS slice (keys distinct).  Each lane ranks its elements against the whole segment.
#define WSORT_MAX 256
__device__ void wave_sort_segment(long long *key, double *val, int b, int e, int *lk, double *lv)
{
    const int lane = lane_id();
    const int n = e - b;
    WAVE_LOCKSTEP(); // (the previous segment's reads of the LDS slice are done)
    for (int t = lane; t < n; t += 64) {
        lk[t] = (int)key[b + t];
        lv[t] = val[b + t];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // LDS stores above before the LDS reads below (same wave)
    for (int t = lane; t < n; t += 64) {
        const int k = lk[t];
        int r = 0;
        for (int u = 0; u < n; u++) r += (lk[u] < k);
        key[b + r] = k;
        val[b + r] = lv[t];
    }
}

// Workgroup-wide sort of one long segment [b,e) with distinct keys in [0,m): presence bitmap -> rank.
// flag/rnk: int[m] scratch, stage_k/stage_v: scratch of >= e-b entries.
template <class Scope>
__device__ void scope_sort_segment(Scope &sc, long long *key, double *val, int b, int e, int m, int *flag, int *rnk, int *stage_k,
                                   double *stage_v)
{
    const int tid = sc.tid(), nt = sc.nt();
    for (int j = tid; j < m; j += nt) flag[j] = 0;
    sc.sync();
    for (int p = b + tid; p < e; p += nt) flag[(int)key[p]] = 1;
    sc.sync();
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int j = c0 + tid;
        const int f = j < m ? flag[j] : 0;
        int tot;
        const int ex = sc.excl_scan(f, &tot);
        if (j < m) rnk[j] = base + ex;
        base += tot;
    }
    sc.sync();
    for (int p = b + tid; p < e; p += nt) {
        const int k = (int)key[p];
        stage_k[rnk[k]] = k;
        stage_v[rnk[k]] = val[p];
    }
    sc.sync();
    for (int p = b + tid; p < e; p += nt) {
        key[p] = stage_k[p - b];
        val[p] = stage_v[p - b];
    }
    sc.sync();
}

// REGSORT: short segments by small_sort_pairs (the 256-thread workgroups of a batch: their register budget allows it),
// else by insertion_sort_pairs
template <bool REGSORT, class Scope>
__device__ __forceinline__ void finish_body(const DevG &D, const FinishOut &O, Scope &sc, double *shd, int *lds_k, double *lds_v,
                                            int *win = nullptr, int wincap = 0, // (win: an LDS window for the column counters, see prep_body)
                                            int nslice = 1 << 30,              // (waves of a workgroup that have a sort slice in lds_k / lds_v)
                                            char *fscr = nullptr, long long fscr_bytes = 0) // (this workgroup's scratch for the fill through buckets, k_bucket.h)
{
    Scalars *S = D.s;
    const int tid = sc.tid(), nt = sc.nt();
    const int m = D.m;
    if (S->status != ST_DONE) return;
    const int rank = S->rank;
    FILL_STAMP_BEGIN();

    // ---- complete the permutations: unpivoted rows / columns in index order (build_factors.rs:192-209)
    for (int pass = 0; pass < 2; pass++) {
        gint_p inv = pass == 0 ? D.pinv : D.qinv;
        gint_p seq = pass == 0 ? D.prow : D.pcol;
        int base = rank;
        for (int c0 = 0; c0 < m; c0 += nt) {
            const int e = c0 + tid;
            const int f = (e < m && inv[e] < 0) ? 1 : 0;
            int tot;
            const int ex = sc.excl_scan(f, &tot);
            if (f) {
                inv[e] = base + ex;
                seq[base + ex] = e;
            }
            base += tot;
        }
        if (base != m && sc.leader()) DEV_CHECK(S, false);
    }
    sc.sync();
    // dependent columns get unit pivots (build_factors.rs:221-223); empty L columns / U rows for them
    for (int k = rank + tid; k < m; k += nt) {
        D.colmax[D.pcol[k]] = 1.0;
        D.lbeg[k + 1] = D.lbeg[rank];
        D.ubeg[k + 1] = D.ubeg[rank];
    }
    for (int k = tid; k < m; k += nt) {
        O.rowperm[k] = D.prow[k];
        O.colperm[k] = D.pcol[k];
    }
    sc.sync();

    FILL_STAMP(S, 8); // permutations
    // ---- L: column k = unit diagonal, then the stage-k column with rows renumbered by pinv and sorted
    // (get_factors.rs:86-113 scatters the row-wise copy in row order, which sorts each column)
    const int l_nz = D.lbeg[rank];
    if (sc.leader()) *sc.ctr(0) = *sc.ctr(1) = 0;
    sc.sync();
    for (int k = tid; k <= m; k += nt) O.l_colptr[k] = (long long)D.lbeg[k] + k;
    for (int k = tid; k < m; k += nt) {
        const int b = D.lbeg[k], e = D.lbeg[k + 1];
        const int ob = b + k;
        O.l_rowidx[ob] = k;
        O.l_value[ob] = 1.0;
        if (REGSORT && e - b <= SSORT_MAX) { // (nearly every column of an LP basis: renumbered and sorted in registers, written once)
            small_sort_gather(D.lidx, D.lval, D.pinv, b, e, O.l_rowidx, O.l_value, ob + 1);
            continue;
        }
        // (four entries per turn: their loads -- index, then the gather through pinv -- are in flight together; a thread
        // walking its line entry by entry is a chain of dependent round trips, and one workgroup per CU hides none)
        int p = b;
        for (; p + 4 <= e; p += 4) {
            const int i0 = D.lidx[p], i1 = D.lidx[p + 1], i2 = D.lidx[p + 2], i3 = D.lidx[p + 3];
            const double v0 = D.lval[p], v1 = D.lval[p + 1], v2 = D.lval[p + 2], v3 = D.lval[p + 3];
            const int r0 = D.pinv[i0], r1 = D.pinv[i1], r2 = D.pinv[i2], r3 = D.pinv[i3];
            const int o = ob + 1 + (p - b);
            O.l_rowidx[o] = r0;
            O.l_rowidx[o + 1] = r1;
            O.l_rowidx[o + 2] = r2;
            O.l_rowidx[o + 3] = r3;
            O.l_value[o] = v0;
            O.l_value[o + 1] = v1;
            O.l_value[o + 2] = v2;
            O.l_value[o + 3] = v3;
        }
        for (; p < e; p++) {
            O.l_rowidx[ob + 1 + (p - b)] = D.pinv[D.lidx[p]];
            O.l_value[ob + 1 + (p - b)] = D.lval[p];
        }
        if (e - b > WSORT_MAX) D.iw2[m - 1 - atomicAdd(sc.ctr(1), 1)] = k; // long: from the top of iw2
        else if (e - b > SSORT_MAX) D.iw2[atomicAdd(sc.ctr(0), 1)] = k;     // medium: from the bottom
        else if (REGSORT) small_sort_pairs(O.l_rowidx, O.l_value, ob + 1, ob + 1 + (e - b));
        else insertion_sort_pairs(O.l_rowidx, O.l_value, ob + 1, ob + 1 + (e - b));
    }
    sc.sync();
    FILL_STAMP(S, 9); // L columns
    {
        const int nmed = __hip_atomic_load(sc.ctr(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nlong = __hip_atomic_load(sc.ctr(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sc.sync();
        // (a 512-thread workgroup of a batch has sort slices for `nslice` of its waves only: those share out the segments)
        const bool limited = nslice < num_waves();
        if (!limited || wave_id() < nslice)
            for (int r = sc.wid(); r < nmed; r += limited ? nslice : sc.nw()) {
                const int k = D.iw2[r];
                wave_sort_segment(O.l_rowidx, O.l_value, D.lbeg[k] + k + 1, D.lbeg[k + 1] + k + 1, &lds_k[wave_id() * WSORT_MAX],
                                  &lds_v[wave_id() * WSORT_MAX]);
            }
        for (int r = 0; r < nlong; r++) {
            const int k = D.iw2[m - 1 - r];
            const int b = D.lbeg[k] + k + 1, e = D.lbeg[k + 1] + k + 1;
            scope_sort_segment(sc, O.l_rowidx, O.l_value, b, e, m, (int *)D.iw0, (int *)D.iw1, (int *)D.tnew, (double *)D.txrj);
        }
    }
    sc.sync();

    FILL_STAMP(S, 10); // L: medium and long columns
    // ---- U: column k collects the entries (stage k', column pcol[k]) of all U rows, ascending k'
    // (get_factors.rs:136-167); entries in columns that never became pivotal are dropped
    // (build_factors.rs:318-337, `qinv[j] < rank`)
    for (int k = tid; k < m; k += nt) D.iw0[k] = 0;
    sc.sync();
    // (a batch: the target column qinv[uidx[p]] of every entry is a gather; the first sweep leaves it in the workgroup's
    // scratch, ctgt[p], and the later sweeps read that in place of index + gather.  Counters of one byte hold every column of
    // the matrix in ONE window (147 456 columns), so the count is a single sweep; a column of 255 or more entries sends the
    // matrix to the 32-bit windows.)
    const int unz4 = (D.ubeg[rank] + 3) & ~3;
    const bool keep_c = win && fscr && 20LL * unz4 <= fscr_bytes; // (uniform; 4 bytes per target + 16 per record of the fill)
    int *ctgt = (int *)fscr;
    bool have_c = false;
    const auto colof = [&](int p) { return have_c ? ctgt[p] : D.qinv[D.uidx[p]]; };
    if (win) { // column counts through the LDS window
        bool counted = false;
        if (rank <= 4 * wincap) {
            unsigned *winb = (unsigned *)win;
            for (int i = tid; i < (rank + 3) / 4; i += nt) winb[i] = 0;
            sc.sync();
            int ovf = 0;
            for (int k = tid; k < rank; k += nt)
                line4(D.ubeg[k], D.ubeg[k + 1], [&](int p) { return D.qinv[D.uidx[p]]; },
                      [&](int p, int c) {
                          if (keep_c) ctgt[p] = c;
                          if (c < rank) {
                              const int sh = (c & 3) * 8;
                              const unsigned was = atomicAdd(&winb[c >> 2], 1u << sh);
                              ovf |= ((was >> sh) & 255u) == 255u;
                          }
                      });
            have_c = keep_c;
            ovf = sc.any(ovf); // (a barrier: every counter is final behind it)
            if (!ovf) {
                for (int i = tid; i < rank; i += nt) D.iw0[i] = (int)((winb[i >> 2] >> ((i & 3) * 8)) & 255u);
                counted = true;
            }
            sc.sync();
        }
        for (int w0 = 0; !counted && w0 < rank; w0 += wincap) {
            const int wn = rank - w0 < wincap ? rank - w0 : wincap;
            const bool put_c = keep_c && !have_c; // (the first sweep over the entries)
            for (int i = tid; i < wn; i += nt) win[i] = 0;
            sc.sync();
            for (int k = tid; k < rank; k += nt)
                line4(D.ubeg[k], D.ubeg[k + 1], colof, [&](int p, int c) {
                    if (put_c) ctgt[p] = c;
                    const unsigned d = (unsigned)(c - w0);
                    if (d < (unsigned)wn) atomicAdd(&win[d], 1);
                });
            have_c = keep_c;
            sc.sync();
            for (int i = tid; i < wn; i += nt) D.iw0[w0 + i] = win[i];
            sc.sync();
        }
    } else {
        for (int k = tid; k < rank; k += nt) {
            const int e = D.ubeg[k + 1];
            int p = D.ubeg[k];
            for (; p + 4 <= e; p += 4) {
                const int j0 = D.uidx[p], j1 = D.uidx[p + 1], j2 = D.uidx[p + 2], j3 = D.uidx[p + 3];
                const int c0 = D.qinv[j0], c1 = D.qinv[j1], c2 = D.qinv[j2], c3 = D.qinv[j3];
                if (c0 < rank) g_atomic_add(&D.iw0[c0], 1);
                if (c1 < rank) g_atomic_add(&D.iw0[c1], 1);
                if (c2 < rank) g_atomic_add(&D.iw0[c2], 1);
                if (c3 < rank) g_atomic_add(&D.iw0[c3], 1);
            }
            for (; p < e; p++) {
                const int c = D.qinv[D.uidx[p]];
                if (c < rank) g_atomic_add(&D.iw0[c], 1);
            }
        }
    }
    sc.sync();
    FILL_STAMP(S, 11); // U: column counts
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int k = c0 + tid;
        const int cnt = k < m ? D.iw0[k] + 1 : 0;
        int tot;
        const int ex = sc.excl_scan(cnt, &tot);
        if (k < m) {
            O.u_colptr[k] = base + ex;
            D.iw1[k] = base + ex; // fill cursor
        }
        base += tot;
    }
    const int u_tot = base; // u_nz + m
    if (sc.leader()) {
        O.u_colptr[m] = u_tot;
        *sc.ctr(0) = *sc.ctr(1) = 0;
    }
    sc.sync();
    FILL_STAMP(S, 12); // U: column pointers
    // the fill of a batch in two phases (k_bucket.h): the records behind the targets in the workgroup's scratch
    bool bucketed = false;
    if (win && have_c) {
        static_assert(BKT_WSORT == WSORT_MAX, "lines the buckets leave sorted = lines the pass below skips");
        Buckets BK = buckets_in(win, wincap);
        if (buckets_plan(sc, BK, D.iw1, m, u_tot)) {
            BktRec *scr = (BktRec *)(fscr + 4LL * unz4);
            buckets_open(sc, BK, D.iw1, m, 1);
            for (int k = tid; k < rank; k += nt)
                line4(D.ubeg[k], D.ubeg[k + 1], [&](int p) { return IdxVal{0, ctgt[p], D.uval[p]}; },
                      [&](int, const IdxVal &a) {
                          if (a.g < rank) bucket_put(BK, scr, a.g, k, a.v);
                      });
            sc.sync();
            FILL_STAMP(S, 13); // U: plan + phase A
            for (int b = 0; b < BK.nb; b++)
                bucket_flush<false>(
                    sc, BK, b, D.iw1, m, u_tot, 1, scr,
                    [&](int t, int *key, double *val) { // pivot last
                        *key = t;
                        *val = D.colmax[D.pcol[t]];
                    },
                    [&](int pos, int key, double val) {
                        O.u_rowidx[pos] = key;
                        O.u_value[pos] = val;
                    });
            bucketed = true;
            if (sc.leader()) S->fill_paths |= 2;
        }
    }
    if (bucketed) {
        // (filled above)
    } else if (win) { // column cursors in the LDS window
        for (int w0 = 0; w0 < rank; w0 += wincap) {
            const int wn = rank - w0 < wincap ? rank - w0 : wincap;
            for (int i = tid; i < wn; i += nt) win[i] = D.iw1[w0 + i];
            sc.sync();
            for (int k = tid; k < rank; k += nt)
                line4(D.ubeg[k], D.ubeg[k + 1], [&](int p) { return IdxVal{0, colof(p), D.uval[p]}; },
                      [&](int, const IdxVal &a) {
                          const unsigned d = (unsigned)(a.g - w0);
                          if (d < (unsigned)wn) {
                              const int pos = atomicAdd(&win[d], 1);
                              O.u_rowidx[pos] = k;
                              O.u_value[pos] = a.v;
                          }
                      });
            sc.sync();
        }
    } else {
        for (int k = tid; k < rank; k += nt) {
            const int e = D.ubeg[k + 1];
            int p = D.ubeg[k];
            for (; p + 4 <= e; p += 4) { // (a column's entries are sorted afterwards: the order the cursors are taken in is free)
                const int j0 = D.uidx[p], j1 = D.uidx[p + 1], j2 = D.uidx[p + 2], j3 = D.uidx[p + 3];
                const double v0 = D.uval[p], v1 = D.uval[p + 1], v2 = D.uval[p + 2], v3 = D.uval[p + 3];
                const int c0 = D.qinv[j0], c1 = D.qinv[j1], c2 = D.qinv[j2], c3 = D.qinv[j3];
                const int q0 = c0 < rank ? g_atomic_add(&D.iw1[c0], 1) : -1;
                const int q1 = c1 < rank ? g_atomic_add(&D.iw1[c1], 1) : -1;
                const int q2 = c2 < rank ? g_atomic_add(&D.iw1[c2], 1) : -1;
                const int q3 = c3 < rank ? g_atomic_add(&D.iw1[c3], 1) : -1;
                if (q0 >= 0) {
                    O.u_rowidx[q0] = k;
                    O.u_value[q0] = v0;
                }
                if (q1 >= 0) {
                    O.u_rowidx[q1] = k;
                    O.u_value[q1] = v1;
                }
                if (q2 >= 0) {
                    O.u_rowidx[q2] = k;
                    O.u_value[q2] = v2;
                }
                if (q3 >= 0) {
                    O.u_rowidx[q3] = k;
                    O.u_value[q3] = v3;
                }
            }
            for (; p < e; p++) {
                const int c = D.qinv[D.uidx[p]];
                if (c < rank) {
                    const int pos = g_atomic_add(&D.iw1[c], 1);
                    O.u_rowidx[pos] = k;
                    O.u_value[pos] = D.uval[p];
                }
            }
        }
    }
    sc.sync();
    FILL_STAMP(S, 14); // U: phase B (or the window sweeps)
    double pmin = INFINITY, pmax = 0.0;
    for (int k = tid; k < m; k += nt) {
        const int b = (int)O.u_colptr[k], e = b + D.iw0[k];
        const double piv = D.colmax[D.pcol[k]];
        if (!bucketed) {
            O.u_rowidx[e] = k; // pivot last
            O.u_value[e] = piv;
        }
        pmin = fmin(pmin, fabs(piv));
        pmax = fmax(pmax, fabs(piv));
        if (e - b > WSORT_MAX) D.iw2[m - 1 - atomicAdd(sc.ctr(1), 1)] = k;
        else if (bucketed) continue; // (sorted in LDS before it was written)
        else if (e - b > SSORT_MAX) D.iw2[atomicAdd(sc.ctr(0), 1)] = k;
        else if (REGSORT) small_sort_pairs(O.u_rowidx, O.u_value, b, e);
        else insertion_sort_pairs(O.u_rowidx, O.u_value, b, e);
    }
    sc.sync();
    FILL_STAMP(S, 15); // U: pivots, short columns
    {
        const int nmed = __hip_atomic_load(sc.ctr(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nlong = __hip_atomic_load(sc.ctr(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sc.sync();
        const bool limited = nslice < num_waves();
        if (!limited || wave_id() < nslice)
            for (int r = sc.wid(); r < nmed; r += limited ? nslice : sc.nw()) {
                const int k = D.iw2[r];
                const int b = (int)O.u_colptr[k];
                wave_sort_segment(O.u_rowidx, O.u_value, b, b + D.iw0[k], &lds_k[wave_id() * WSORT_MAX], &lds_v[wave_id() * WSORT_MAX]);
            }
        for (int r = 0; r < nlong; r++) {
            const int k = D.iw2[m - 1 - r];
            const int b = (int)O.u_colptr[k];
            // (the segment's own length lives in iw0, which the sort must not use as scratch: iw1 / tnewr / tnew / txrj)
            scope_sort_segment(sc, O.u_rowidx, O.u_value, b, b + D.iw0[k], m, (int *)D.iw1, (int *)D.tnewr, (int *)D.tnew, (double *)D.txrj);
        }
    }
    FILL_STAMP(S, 16); // U: medium and long columns
    // min / max pivot (build_factors.rs:403-419): |pivots| are non-negative, so they order like their bit patterns
    const double amin = sc.min_d(pmin, shd), amax = sc.max_d(pmax, shd);
    if (sc.leader()) {
        S->min_pivot = amin;
        S->max_pivot = amax;
        S->l_nz = l_nz;
        S->u_nz = u_tot - m;
    }
}
// NT = threads of the workgroup at most (256 or 512: a batch -- two waves per SIMD still leave a thread the 200-250
// registers of the register sorts; 1024: one matrix without a cooperative launch)
// fscr: scratch of fscr_bytes per workgroup for the fill through buckets, or null
template <int NT> __global__ void __launch_bounds__(NT) k_finish(DevLU *Ds, FinishOut *Os, int nmat, int winbytes, char *fscr, long long fscr_bytes)
{
    BLU_DYN_SHARED(unsigned char, finish_win, 144 * 1024); // (the counter window: winbytes of dynamic LDS, or none)
    __shared__ int sh[40];
    __shared__ long long shl[20];
    __shared__ double shd[40];
    constexpr int NSLICE = NT == 512 ? 4 : NT / 64; // (512 threads: sort slices for four of the eight waves -- the window takes the rest of the LDS)
    __shared__ int lds_k[NSLICE * WSORT_MAX];
    __shared__ double lds_v[NSLICE * WSORT_MAX];
    for (int b = blockIdx.x; b < nmat; b += gridDim.x) { // (see k_prep)
        const DevG D(Ds[b]);
        BlockScope sc{sh, shl};
        finish_body<NT <= 512>(D, Os[b], sc, shd, lds_k, lds_v, winbytes > 0 ? (int *)finish_win : nullptr, winbytes / 4, NSLICE,
                               fscr ? fscr + (long long)blockIdx.x * fscr_bytes : nullptr, fscr_bytes);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_finish_grid(DevLU *Ds, FinishOut *Os, GridWs *gw)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    __shared__ double shd[40];
    __shared__ int lds_k[16 * WSORT_MAX];
    __shared__ double lds_v[16 * WSORT_MAX];
    const DevG D(Ds[0]);
    GridScope sc{sh, shl, gw, 0};
    finish_body<false>(D, Os[0], sc, shd, lds_k, lds_v);
}

// ---------------------------------------------------------------------------------------------
// k_compact: copy every line of one file into a new arena (which = 0 column file, 1 row file)
// ---------------------------------------------------------------------------------------------
// One workgroup per handle of the batch; whichv[b]: 0 = column file, 1 = row file, < 0 = nothing to do for handle b.
__global__ void __launch_bounds__(1024) k_compact(DevLU *Ds, const int *whichv, int *const *new_idx, double *const *new_val,
                                                  const int *new_cap)
{
    const int which = whichv[blockIdx.x];
    if (which < 0) return;
    const DevG D(Ds[blockIdx.x]);
    Scalars *S = D.s;
    __shared__ int sh[40];
    const int tid = threadIdx.x, nt = blockDim.x, w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    const RecBeg beg = which ? D.rbeg : D.cbeg;
    const RecLen len = which ? D.rlen : D.clen;
    const RecCap cap = which ? D.rcap : D.ccap;
    gcint_p old_idx = which ? D.ridx : D.cidx;
    int *nidx = new_idx[blockIdx.x];
    double *nval = which ? nullptr : new_val[blockIdx.x];
    // new offsets; lines that hold nothing get no room (they are dead: pivoted or emptied)
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int e = c0 + tid;
        const int l = e < m ? len[e] : 0;
        const int c = l > 0 ? l + stretch_of(D.stretch, l) + D.pad : 0;
        int tot;
        const int ex = block_excl_scan_i(c, sh, &tot);
        if (e < m) {
            D.iw0[e] = base + ex;
            D.iw1[e] = c;
        }
        base += tot;
    }
    if (base > new_cap[blockIdx.x]) {
        if (tid == 0) DEV_CHECK(S, false);
        return;
    }
    __syncthreads();
    for (int e = w; e < m; e += nw) {
        const int l = len[e], ob = beg[e], nb = D.iw0[e];
        for (int p = lane; p < l; p += 64) {
            nidx[nb + p] = old_idx[ob + p];
            if (!which) nval[nb + p] = D.cval[ob + p];
        }
    }
    __syncthreads();
    for (int e = tid; e < m; e += nt) {
        beg[e] = D.iw0[e];
        cap[e] = D.iw1[e];
    }
    if (tid == 0) {
        if (which) S->rused = base; else S->cused = base;
        S->ngarbage++;
    }
}
