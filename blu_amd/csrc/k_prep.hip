// k_prep.hip -- O(nnz) phases in front of the pivot loop.  Written once against the Scope interface of
// blu_dev.h: one workgroup per matrix in a batch (k_prep, k_setup), the whole chip for a single matrix
// (k_prep_grid, k_setup_grid: cooperative launch, grid barriers between the passes).
//
//   k_prep   = singletons()  (src/lu/singletons.rs:81-264): validate B, build the row-wise copy,
//              peel singleton columns/rows (without cascade: reference defect D1)
//   k_setup  = setup_bump()  (src/lu/setup_bump.rs:55-264): column file, row file, count lists
//
// Both are restated data-parallel: the reference's sequential loops are replaced by
// count / scan / ordered-fill passes that produce the same order of entries and the same
// order of elements in the count lists (ascending index inside a list, list.rs:65-70).
#include "blu_dev.h"
#include "k_bucket.h"

// ---------------------------------------------------------------------------------------------
// ordered tail-append of elements 0..n-1 (ascending) into count lists by key, ONE wave.
// keys[e] < 0 : element is not inserted.  Equivalent to `for e in 0..n { list_add(e, keys[e]) }`
// (list.rs:54-77).  Returns the minimum key > 0 seen (or big).
// ---------------------------------------------------------------------------------------------
// The lists are independent of each other, so the waves of the scope share them out: wave `part` of `nparts`
// builds the lists with key % nparts == part (every wave scans all elements; the keys are read coalesced).
// ltail / kcap (a batch: LDS of the workgroup, or null): the tail of every list k < kcap lives in ltail[k] while the lists
// are built and goes to its head, blink[n + k], at the end -- the tail is what every step of the build waits for (a
// dependent round trip to L2 per distinct key of every 64 elements: 0.028 of the 0.081 s of k_setup for 1536 bases of
// the 100k size), and the keys of an LP basis are line counts of a few entries.
__device__ int wave_list_build(const LinkF &flink, const LinkB &blink, int n, gcint_p keys, int big, int part, int nparts, int *ltail = nullptr,
                               int kcap = 0)
{
    const int lane = lane_id();
    int minkey = big;
    if (!ltail) kcap = 0;
    for (int k = part + nparts * lane; k < kcap; k += nparts * 64) ltail[k] = n + k; // (empty: the head itself)
    WAVE_LOCKSTEP();
    int key_next = lane < n ? keys[lane] : -1;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int e = c0 + lane;
        const int key = key_next;
        key_next = e + 64 < n ? keys[e + 64] : -1; // (in flight while this chunk is linked)
        bool act = key >= 0 && key % nparts == part;
        if (act && key > 0) minkey = min(minkey, key);
        unsigned long long active = __ballot(act);
        bool through_memory = false;
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int k = wave_bcast_i(key, leader);
            const unsigned long long grp = __ballot(act && key == k);
            const bool inl = k < kcap; // (uniform)
            const int tail = inl ? ltail[k] : blink[n + k];
            through_memory = through_memory || !inl;
            WAVE_LOCKSTEP(); // (every lane has the old tail before the group's last lane replaces it)
            if (act && key == k) {
                const unsigned long long below = grp & lanes_below(lane);
                const unsigned long long above = grp & ~((2ull << lane) - 1ull);
                const int prevl = below ? 63 - __clzll((long long)below) : -1;
                const int nextl = above ? __ffsll((long long)above) - 1 : -1;
                blink[e] = prevl >= 0 ? c0 + prevl : tail;
                flink[e] = nextl >= 0 ? c0 + nextl : n + k;
                if (prevl < 0) flink[tail] = e;
                if (nextl < 0) {
                    if (inl) ltail[k] = e;
                    else blink[n + k] = e;
                }
                act = false;
            }
            active &= ~grp;
        }
        if (through_memory) wave_mem_sync(); // the next chunk reads blink[n+k] written here
        WAVE_LOCKSTEP();
    }
    for (int k = part + nparts * lane; k < kcap; k += nparts * 64)
        if (ltail[k] != n + k) blink[n + k] = ltail[k];
    return wave_min_i(minkey);
}

// ---------------------------------------------------------------------------------------------
// k_prep
// ---------------------------------------------------------------------------------------------
// Sort of a short row (n <= N pairs, keys possibly equal: ties keep their order, so duplicates end up neighbours) by
// one thread without a dependent chain of memory accesses: every pair into registers (the loads are issued together),
// ranked there, stored where it belongs.  The insertion sort through memory costs ~n^2/4 dependent round trips.
template <int N> __device__ __forceinline__ void reg_sort_row(gint_p idx, gdouble_p val, int b, int n)
{
    int k[N];
    double v[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        k[i] = 0x7fffffff;
        v[i] = 0.0;
        if (i < n) {
            k[i] = idx[b + i];
            v[i] = val[b + i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (i < n) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < N; j++) r += (k[j] < k[i] || (j < i && k[j] == k[i])) ? 1 : 0;
            idx[b + r] = k[i];
            val[b + r] = v[i];
        }
    }
}

// REGSORT: short rows by reg_sort_row (the 256-thread workgroups of a batch: their register budget allows it)
// win / wincap: an LDS window of wincap ints, or null.  Global atomics are performed at the memory side on this part
// (TCC_EA0_ATOMIC == TCC_ATOMIC, whatever their scope): one round trip to HBM per entry of the matrix for the row
// counts and again for the fill cursors.  A workgroup that has a CU's LDS to itself (a batch: one workgroup per CU)
// counts and fills through LDS atomics instead, one window of rows at a time (three windows for 100 000 rows), reading
// the packed row indices once per window.
template <bool REGSORT, class Scope> __device__ __forceinline__ void prep_body(const DevG &D, Scope &sc, int *win = nullptr, int wincap = 0)
{
    Scalars *S = D.s;
    const int tid = sc.tid(), nt = sc.nt();
    const int m = D.m;
    if (S->status != ST_RUNNING) return;
    FILL_STAMP_BEGIN();

    // ---- check pointers, count nnz(B), column pointers of the packed copy (singletons.rs:119-133)
    int bad = 0;
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int j = c0 + tid;
        int len = 0;
        if (j < m) {
            const unsigned long long b = D.b_begin[j], e = D.b_end[j];
            if (e < b) bad = 1;
            else if (e - b > 0x7fffffffull || e > (unsigned long long)D.b_i_len) bad = 1; // outside the caller's b_i/b_x
            else len = (int)(e - b);
        }
        // the chunk total in 64 bits FIRST: up to 1024 lengths of up to 2^31-1 each (columns may overlap) can wrap
        // a 32-bit scan, and a wrapped total would pass the range check below
        const long long tot64 = sc.sum_ll((long long)len);
        if ((long long)base + tot64 > 0x7fffffffLL) {
            bad = 1;
            break; // (uniform: every thread sees the same total)
        }
        int tot;
        int ex = sc.excl_scan(len, &tot);
        if (j < m) D.bc_ptr[j] = base + ex;
        base += tot;
    }
    bad = sc.any(bad);
    if (bad) {
        if (sc.leader()) set_error(S, ST_INVALID_ARG, __LINE__);
        return;
    }
    const int b_nz = base;
    if (sc.leader()) {
        D.bc_ptr[m] = b_nz;
        S->matrix_nz = b_nz;
    }
    if (b_nz > D.nzcap) { // host sized the packed copies too small (overlapping columns): ask for more
        if (sc.leader()) {
            S->need = b_nz;
            set_error(S, ST_NEED_CW, __LINE__);
        }
        return;
    }
    for (int i = tid; i < m; i += nt) D.iw0[i] = 0;
    // (a window that holds every row as a counter of one byte: the rows are counted while the columns are packed; a row of
    // 255 or more entries sends the matrix to the sweeps with 32-bit windows below)
    const bool bytes = win && m <= 4 * wincap; // (uniform)
    unsigned *winb = (unsigned *)win;
    if (bytes)
        for (int i = tid; i < (m + 3) / 4; i += nt) winb[i] = 0;
    int ovf = 0;
    sc.sync();

    FILL_STAMP(S, 0); // column pointers
    // ---- count nz per row, check indices, pack columns (singletons.rs:152-173)
    for (int j = tid; j < m; j += nt) {
        const unsigned long long b = D.b_begin[j], e = D.b_end[j];
        int put = D.bc_ptr[j];
        const auto take = [&](unsigned long long i, double x) {
            if (i >= (unsigned long long)m) {
                bad = 1;
            } else {
                if (!win) {
                    g_atomic_add(&D.iw0[(int)i], 1);
                } else if (bytes) {
                    const int sh = ((int)i & 3) * 8;
                    const unsigned was = atomicAdd(&winb[(int)i >> 2], 1u << sh);
                    ovf |= ((was >> sh) & 255u) == 255u;
                }
                D.bc_idx[put] = (int)i;
                D.bc_val[put] = x;
            }
            put++;
        };
        unsigned long long pos = b;
        for (; pos + 4 <= e; pos += 4) { // (four entries per turn, their loads in flight together: see line4, blu_dev.h)
            const unsigned long long i0 = D.b_i[pos], i1 = D.b_i[pos + 1], i2 = D.b_i[pos + 2], i3 = D.b_i[pos + 3];
            const double x0 = D.b_x[pos], x1 = D.b_x[pos + 1], x2 = D.b_x[pos + 2], x3 = D.b_x[pos + 3];
            take(i0, x0);
            take(i1, x1);
            take(i2, x2);
            take(i3, x3);
        }
        for (; pos < e; pos++) take(D.b_i[pos], D.b_x[pos]);
    }
    bad = sc.any(bad); // (a grid barrier in the wide scope: the row counts of every workgroup are complete behind it)
    if (bad) {
        if (sc.leader()) set_error(S, ST_INVALID_ARG, __LINE__);
        return;
    }
    bool counted = false;
    if (bytes) {
        ovf = sc.any(ovf);
        if (!ovf) {
            for (int i = tid; i < m; i += nt) D.iw0[i] = (int)((winb[i >> 2] >> ((i & 3) * 8)) & 255u);
            counted = true;
        }
        sc.sync();
    }
    if (win && !counted) { // row counts through the LDS window
        for (int r0 = 0; r0 < m; r0 += wincap) {
            const int wn = m - r0 < wincap ? m - r0 : wincap;
            for (int i = tid; i < wn; i += nt) win[i] = 0;
            sc.sync();
            for (int j = tid; j < m; j += nt)
                line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int pos) { return D.bc_idx[pos]; },
                      [&](int, int i) {
                          const unsigned d = (unsigned)(i - r0);
                          if (d < (unsigned)wn) atomicAdd(&win[d], 1);
                      });
            sc.sync();
            for (int i = tid; i < wn; i += nt) D.iw0[r0 + i] = win[i];
            sc.sync();
        }
    }

    FILL_STAMP(S, 1); // pack + row counts
    // ---- row pointers (singletons.rs:176-183)
    base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int i = c0 + tid;
        const int cnt = i < m ? D.iw0[i] : 0;
        int tot;
        int ex = sc.excl_scan(cnt, &tot);
        if (i < m) {
            D.bt_ptr[i] = base + ex;
            D.iw1[i] = base + ex; // fill cursor
        }
        base += tot;
    }
    if (sc.leader()) D.bt_ptr[m] = base;
    sc.sync();

    FILL_STAMP(S, 2); // row pointers
    // ---- fill rows in arbitrary order, then sort each row by column index: the reference fills
    // rows for j = 0..m-1 in turn (singletons.rs:186-198), i.e. ascending column inside a row.
    // a batch: in two phases through buckets (k_bucket.h); the 16-byte records go to the value array of the column arena
    // (nothing is in the arenas before k_setup)
    bool bucketed = false;
    if (win) {
        static_assert(BKT_WSORT >= 48, "rows the buckets leave sorted (up to BKT_WSORT entries) cover the short rows of the pass below");
        Buckets BK = buckets_in(win, wincap);
        const bool room = 2LL * b_nz <= (long long)D.carena_cap; // (uniform)
        if (room && buckets_plan(sc, BK, D.iw1, m, b_nz)) {
            BktRec *scr = (BktRec *)D.cval;
            buckets_open(sc, BK, D.iw1, m, 0);
            for (int j = tid; j < m; j += nt)
                line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int pos) { return IdxVal{D.bc_idx[pos], 0, D.bc_val[pos]}; },
                      [&](int, const IdxVal &a) { bucket_put(BK, scr, a.i, j, a.v); });
            sc.sync();
            FILL_STAMP(S, 3); // fill: plan + phase A
            for (int b = 0; b < BK.nb; b++)
                bad |= bucket_flush<true>(
                    sc, BK, b, D.iw1, m, b_nz, 0, scr, [&](int, int *, double *) {},
                    [&](int pos, int key, double val) {
                        D.bt_idx[pos] = key;
                        D.bt_val[pos] = val;
                    });
            bucketed = true;
            if (sc.leader()) S->fill_paths |= 1;
        }
    }
    if (bucketed) {
        // (filled above)
    } else if (win) { // fill cursors in the LDS window
        for (int r0 = 0; r0 < m; r0 += wincap) {
            const int wn = m - r0 < wincap ? m - r0 : wincap;
            for (int i = tid; i < wn; i += nt) win[i] = D.iw1[r0 + i];
            sc.sync();
            for (int j = tid; j < m; j += nt)
                line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int pos) { return IdxVal{D.bc_idx[pos], 0, D.bc_val[pos]}; },
                      [&](int, const IdxVal &a) {
                          const unsigned d = (unsigned)(a.i - r0);
                          if (d < (unsigned)wn) {
                              const int p = atomicAdd(&win[d], 1);
                              D.bt_idx[p] = j;
                              D.bt_val[p] = a.v;
                          }
                      });
            sc.sync();
        }
    } else {
        for (int j = tid; j < m; j += nt)
            line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int pos) { return IdxVal{D.bc_idx[pos], 0, D.bc_val[pos]}; },
                  [&](int, const IdxVal &a) {
                      const int p = g_atomic_add(&D.iw1[a.i], 1);
                      D.bt_idx[p] = j;
                      D.bt_val[p] = a.v;
                  });
    }
    FILL_STAMP(S, 4); // fill: phase B (or the window sweeps)
    if (sc.leader()) *sc.ctr(0) = 0; // number of long rows
    sc.sync();
    // short rows: insertion sort by one thread; long rows (> 48): bitmap rank sort by the whole scope
    for (int i = tid; i < m; i += nt) {
        const int b = D.bt_ptr[i], e = D.bt_ptr[i + 1];
        if (bucketed && e - b <= BKT_WSORT) continue; // (sorted and checked for duplicates in LDS before it was written)
        if (e - b > 48) {
            const int k = atomicAdd(sc.ctr(0), 1);
            D.iw2[k] = i; // list of long rows (order irrelevant)
            continue;
        }
        if (REGSORT && e - b <= 16) {
            reg_sort_row<16>(D.bt_idx, D.bt_val, b, e - b);
        } else if (REGSORT && e - b <= 32) {
            reg_sort_row<32>(D.bt_idx, D.bt_val, b, e - b);
        } else {
            for (int p = b + 1; p < e; p++) {
                const int kj = D.bt_idx[p];
                const double kv = D.bt_val[p];
                int q = p - 1;
                while (q >= b && D.bt_idx[q] > kj) {
                    D.bt_idx[q + 1] = D.bt_idx[q];
                    D.bt_val[q + 1] = D.bt_val[q];
                    q--;
                }
                D.bt_idx[q + 1] = kj;
                D.bt_val[q + 1] = kv;
            }
        }
        for (int p = b + 1; p < e; p++)
            if (D.bt_idx[p] == D.bt_idx[p - 1]) bad = 1; // duplicate (singletons.rs:195-197)
    }
    sc.sync();
    FILL_STAMP(S, 5); // rows of 33..48 entries, duplicates
    const int nlong = __hip_atomic_load(sc.ctr(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sc.sync();
    for (int r = 0; r < nlong; r++) {
        const int i = D.iw2[r];
        const int b = D.bt_ptr[i], e = D.bt_ptr[i + 1];
        // bitmap of columns present in row i -> rank of each column -> ordered rewrite
        for (int j = tid; j < m; j += nt) D.iw0[j] = 0;
        sc.sync();
        for (int p = b + tid; p < e; p += nt)
            if (g_atomic_add(&D.iw0[D.bt_idx[p]], 1) != 0) bad = 1; // duplicate
        sc.sync();
        int rb = 0;
        for (int c0 = 0; c0 < m; c0 += nt) {
            const int j = c0 + tid;
            const int f = j < m ? (D.iw0[j] ? 1 : 0) : 0;
            int tot;
            int ex = sc.excl_scan(f, &tot);
            if (j < m) D.iw1[j] = rb + ex; // rank of column j inside the row
            rb += tot;
        }
        sc.sync();
        // stage (idx,val) in the column arena's scratch-free tail: use tnew/txrj (m entries each)
        for (int p = b + tid; p < e; p += nt) {
            const int j = D.bt_idx[p];
            const int rk = D.iw1[j];
            D.tnew[rk] = j;
            D.txrj[rk] = D.bt_val[p];
        }
        sc.sync();
        for (int p = b + tid; p < e; p += nt) {
            D.bt_idx[p] = D.tnew[p - b];
            D.bt_val[p] = D.txrj[p - b];
        }
        sc.sync();
    }
    bad = sc.any(bad);
    if (bad) {
        if (sc.leader()) set_error(S, ST_INVALID_ARG, __LINE__);
        return;
    }

    FILL_STAMP(S, 6); // long rows
    // ---- pivot singletons (singletons.rs:203-258), no cascade (D1)
    for (int i = tid; i < m; i += nt) {
        D.pinv[i] = -1;
        D.qinv[i] = -1;
    }
    if (sc.leader()) {
        D.lbeg[0] = 0;
        D.ubeg[0] = 0;
    }
    sc.sync();
    int rank = 0, lused = 0, uused = 0;
    const double abstol = D.abstol;
    for (int phase = 0; phase < 2; phase++) {
        const bool cols = (D.nzbias >= 0) ? (phase == 0) : (phase == 1);
        // winner table: per row the smallest singleton column with an acceptable pivot (or per column the
        // smallest singleton row).  In queue order (ascending index, singletons.rs:318-329) the first
        // acceptable one eliminates the line and empties the others (:337-339, :352-355).
        for (int i = tid; i < m; i += nt) D.iw0[i] = 0x7fffffff;
        sc.sync();
        for (int e = tid; e < m; e += nt) {
            if (cols) {
                const int j = e;
                if (D.qinv[j] < 0 && D.bc_ptr[j + 1] - D.bc_ptr[j] == 1) {
                    const int i = D.bc_idx[D.bc_ptr[j]];
                    const double piv = D.bc_val[D.bc_ptr[j]];
                    DEV_CHECK(S, D.pinv[i] < 0);
                    if (!(piv == 0.0 || fabs(piv) < abstol)) g_atomic_min(&D.iw0[i], j);
                }
            } else {
                const int i = e;
                if (D.pinv[i] < 0 && D.bt_ptr[i + 1] - D.bt_ptr[i] == 1) {
                    const int j = D.bt_idx[D.bt_ptr[i]];
                    const double piv = D.bt_val[D.bt_ptr[i]];
                    DEV_CHECK(S, D.qinv[j] < 0);
                    if (!(piv == 0.0 || fabs(piv) < abstol)) g_atomic_min(&D.iw0[j], i);
                }
            }
        }
        sc.sync();
        // ranks of the winners in ascending index
        int nwin = 0;
        for (int c0 = 0; c0 < m; c0 += nt) {
            const int e = c0 + tid;
            int win = 0, other = -1;
            double piv = 0.0;
            if (e < m) {
                if (cols) {
                    if (D.qinv[e] < 0 && D.bc_ptr[e + 1] - D.bc_ptr[e] == 1) {
                        other = D.bc_idx[D.bc_ptr[e]];
                        piv = D.bc_val[D.bc_ptr[e]];
                        win = (D.iw0[other] == e);
                    }
                } else {
                    if (D.pinv[e] < 0 && D.bt_ptr[e + 1] - D.bt_ptr[e] == 1) {
                        other = D.bt_idx[D.bt_ptr[e]];
                        piv = D.bt_val[D.bt_ptr[e]];
                        win = (D.iw0[other] == e);
                    }
                }
            }
            int tot;
            int ex = sc.excl_scan(win, &tot);
            if (win) {
                const int r = rank + nwin + ex;
                const int i = cols ? other : e, j = cols ? e : other;
                D.prow[r] = i;
                D.pcol[r] = j;
                D.iw1[r] = e; // winners in rank order
                D.colmax[j] = piv;
            }
            nwin += tot;
        }
        sc.sync();
        for (int r = rank + tid; r < rank + nwin; r += nt) {
            D.pinv[D.prow[r]] = r;
            D.qinv[D.pcol[r]] = r;
        }
        sc.sync();
        // factor entries of the new stages, in stage order
        int put0 = cols ? uused : lused;
        for (int c0 = 0; c0 < nwin; c0 += nt) {
            const int r = rank + c0 + tid;
            int cnt = 0;
            if (r < rank + nwin) {
                if (cols) { // U row = entries of row i in still-active columns (singletons.rs:360-377)
                    const int i = D.prow[r];
                    for (int p = D.bt_ptr[i]; p < D.bt_ptr[i + 1]; p++) cnt += (D.qinv[D.bt_idx[p]] < 0);
                } else { // L column = entries of column j in still-active rows, divided by the pivot (:469-487)
                    const int j = D.pcol[r];
                    for (int p = D.bc_ptr[j]; p < D.bc_ptr[j + 1]; p++) cnt += (D.pinv[D.bc_idx[p]] < 0);
                }
            }
            int tot;
            int ex = sc.excl_scan(cnt, &tot);
            if (r < rank + nwin) {
                int put = put0 + ex;
                const bool fits = cols ? (put0 + tot <= D.ucap) : (put0 + tot <= D.lcap);
                if (cols) {
                    D.ubeg[r + 1] = put + cnt;
                    D.lbeg[r + 1] = lused;
                    if (fits) {
                        const int i = D.prow[r];
                        for (int p = D.bt_ptr[i]; p < D.bt_ptr[i + 1]; p++) {
                            const int j2 = D.bt_idx[p];
                            if (D.qinv[j2] < 0) {
                                D.uidx[put] = j2;
                                D.uval[put] = D.bt_val[p];
                                put++;
                            }
                        }
                    }
                } else {
                    D.lbeg[r + 1] = put + cnt;
                    D.ubeg[r + 1] = uused;
                    if (fits) {
                        const int j = D.pcol[r];
                        const double piv = D.colmax[j];
                        for (int p = D.bc_ptr[j]; p < D.bc_ptr[j + 1]; p++) {
                            const int i2 = D.bc_idx[p];
                            if (D.pinv[i2] < 0) {
                                D.lidx[put] = i2;
                                D.lval[put] = D.bc_val[p] / piv;
                                put++;
                            }
                        }
                    }
                }
            }
            put0 += tot;
        }
        if (cols) uused = put0; else lused = put0;
        rank += nwin;
        sc.sync();
    }
    // singletons.rs:135-150 guarantees l_mem, u_mem >= nnz(B) up front; here L/U are sized by the host
    if (uused > D.ucap || lused > D.lcap) {
        if (sc.leader()) {
            S->need = max(uused, lused);
            set_error(S, uused > D.ucap ? ST_NEED_U : ST_NEED_L, __LINE__);
        }
        return;
    }
    if (sc.leader()) {
        S->rank = rank;
        S->rank0 = rank;
        S->lused = lused;
        S->uused = uused;
    }
    FILL_STAMP(S, 7); // singletons
}
// One workgroup per matrix AT A TIME: the grid is smaller than a large batch and each workgroup takes matrices
// blockIdx.x, + gridDim.x, ... (blu_driver.inc: batch_grid).
// NT = threads of the workgroup at most (256 or 512: a batch; 1024: one matrix without a cooperative launch)
// winbytes = dynamic LDS of the launch (the window of prep_body), 0: none
template <int NT> __global__ void __launch_bounds__(NT) k_prep(DevLU *Ds, int nmat, int winbytes)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    BLU_DYN_SHARED(unsigned char, prep_win, 144 * 1024);
    for (int b = blockIdx.x; b < nmat; b += gridDim.x) {
        const DevG D(Ds[b]);
        BlockScope sc{sh, shl};
        prep_body<NT <= 512>(D, sc, winbytes > 0 ? (int *)prep_win : nullptr, winbytes / 4);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_prep_grid(DevLU *Ds, GridWs *gw)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    const DevG D(Ds[0]);
    GridScope sc{sh, shl, gw, 0};
    prep_body<false>(D, sc);
}

// ---------------------------------------------------------------------------------------------
// k_setup = setup_bump (setup_bump.rs:55-264)
// ---------------------------------------------------------------------------------------------
// win / wincap: an LDS window of wincap ints, or null.  The four passes over the entries of B ask of every entry
// whether its row is pivotal already (columns) / whether its column is active (rows): a scattered 4-byte gather each,
// 64 bytes of HBM traffic per entry where the entry itself is 12, and what bounds these passes in a batch (400 KB of
// flags per matrix x 256 matrices in flight is far beyond the L2).  One BIT per line answers the question: with the
// window the flags are a bitmap in LDS (12.5 KB for 100 000 lines), built by ballots from one coalesced read.
template <class Scope> __device__ __forceinline__ void setup_body(const DevG &D, Scope &sc, int *win = nullptr, int wincap = 0)
{
    Scalars *S = D.s;
    const int tid = sc.tid(), nt = sc.nt();
    const int m = D.m;
    if (S->status != ST_RUNNING) return;
    const int rank = S->rank;
    const double abstol = D.abstol, stretch = D.stretch;
    const int pad = D.pad;
    const bool usebm = win != nullptr && (long long)wincap * 32 >= (long long)m + 64; // (uniform)
    const auto build_bm = [&](auto pred) { // bit e of the window = pred(e), e < m
        for (int b0 = sc.wid() * 64; b0 < m; b0 += sc.nw() * 64) {
            const int e = b0 + lane_id();
            const unsigned long long bits = __ballot(e < m && pred(e));
            if (lane_id() == 0) {
                win[b0 >> 5] = (int)(unsigned)bits;
                win[(b0 >> 5) + 1] = (int)(unsigned)(bits >> 32);
            }
        }
        sc.sync();
    };
    const auto bm = [&](int e) { return (int)(((unsigned)win[e >> 5] >> (e & 31)) & 1u); };
    const auto row_g = [&](int i) { return usebm ? (bm(i) ? 0 : -1) : D.pinv[i]; }; // >= 0: row i is pivotal already
    const auto col_g = [&](int j) { return usebm ? bm(j) : D.iw0[j]; };             // > 0: column j is active
    FILL_STAMP_BEGIN();
    if (usebm) build_bm([&](int e) { return D.pinv[e] >= 0; });

    // ---- columns: count, maximum, capacity (setup_bump.rs:131-186).  iw0[j] = list key:
    //      -2 column not active, 0 dropped (cmx == 0 or < abstol), else cnz
    int base = 0;
    long long dropped_nz = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int j = c0 + tid;
        int cap = 0, cnz = 0, key = -2;
        double cmx = 0.0;
        if (j < m && D.qinv[j] < 0) {
            line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int p) { const int i = D.bc_idx[p]; return IdxVal{i, row_g(i), D.bc_val[p]}; },
                  [&](int, const IdxVal &a) {
                      if (a.g >= 0) return;
                      cmx = fmax(cmx, fabs(a.v));
                      cnz++;
                  });
            if (cmx == 0.0 || cmx < abstol) {
                key = 0;
                dropped_nz += cnz;
                cmx = 0.0;
                cnz = 0;
            } else {
                key = cnz;
                cap = cnz + stretch_of(stretch, cnz) + pad;
            }
        }
        int tot;
        int ex = sc.excl_scan(cap, &tot);
        if (j < m) {
            D.iw0[j] = key;
            D.cbeg[j] = key > 0 ? base + ex : 0;
            D.clen[j] = key > 0 ? cnz : 0;
            D.ccap[j] = cap;
            if (key >= 0) D.colmax[j] = cmx;
        }
        if ((long long)base + tot > (long long)D.carena_cap) {
            if (sc.leader()) {
                S->need = base + tot;
                set_error(S, ST_NEED_CW, __LINE__);
            }
            return; // uniform: base/tot are the same for every thread of the scope
        }
        base += tot;
    }
    FILL_STAMP(S, 17); // setup: columns counted
    const int cused = base;
    dropped_nz = sc.sum_ll(dropped_nz);
    sc.sync();
    for (int j = tid; j < m; j += nt) {
        if (D.iw0[j] <= 0) continue;
        int put = D.cbeg[j];
        line4(D.bc_ptr[j], D.bc_ptr[j + 1], [&](int p) { const int i = D.bc_idx[p]; return IdxVal{i, row_g(i), D.bc_val[p]}; },
              [&](int, const IdxVal &a) {
                  if (a.g >= 0) return;
                  D.cidx[put] = a.i;
                  D.cval[put] = a.v;
                  put++;
              });
    }

    FILL_STAMP(S, 18); // setup: columns copied
    // ---- rows: pattern of the copied columns in ascending column order (setup_bump.rs:188-224)
    if (usebm) {
        sc.sync(); // (the column passes are done with the rows' bitmap)
        build_bm([&](int e) { return D.iw0[e] > 0; });
    }
    base = 0;
    for (int c0 = 0; c0 < m; c0 += nt) {
        const int i = c0 + tid;
        int cap = 0, rnz = 0, key = -2;
        if (i < m && D.pinv[i] < 0) {
            line4(D.bt_ptr[i], D.bt_ptr[i + 1], [&](int p) { const int j = D.bt_idx[p]; return IdxVal{j, col_g(j), 0.0}; },
                  [&](int, const IdxVal &a) { rnz += (a.g > 0); });
            key = rnz;
            cap = rnz + stretch_of(stretch, rnz) + pad;
        }
        int tot;
        int ex = sc.excl_scan(cap, &tot);
        if (i < m) {
            D.iw1[i] = key;
            D.rbeg[i] = key >= 0 ? base + ex : 0;
            D.rlen[i] = key >= 0 ? rnz : 0;
            D.rcap[i] = cap;
        }
        if ((long long)base + tot > (long long)D.rarena_cap) {
            if (sc.leader()) {
                S->need = base + tot;
                set_error(S, ST_NEED_RW, __LINE__);
            }
            return;
        }
        base += tot;
    }
    FILL_STAMP(S, 19); // setup: rows counted
    const int rused = base;
    sc.sync();
    for (int i = tid; i < m; i += nt) {
        if (D.iw1[i] < 0) continue;
        int put = D.rbeg[i];
        line4(D.bt_ptr[i], D.bt_ptr[i + 1], [&](int p) { const int j = D.bt_idx[p]; return IdxVal{j, col_g(j), 0.0}; },
              [&](int, const IdxVal &a) {
                  if (a.g > 0) D.ridx[put++] = a.i;
              });
    }

    FILL_STAMP(S, 20); // setup: rows copied
    // ---- count lists (setup_bump.rs:124-130, 188-194): list_init, then list_add in ascending index
    for (int e = tid; e < 2 * m + 2; e += nt) {
        D.cflink[e] = e;
        D.cblink[e] = e;
        D.rflink[e] = e;
        D.rblink[e] = e;
    }
    // scratch that the pivot loop expects all-zero (freshly hipMalloc'ed memory is NOT zero when the
    // allocator recycles a block); gwork is zeroed by the host at allocation and kept zero by its users
    for (int e = tid; e < m; e += nt) {
        D.rowmark[e] = 0;
        D.colmark[e] = 0;
        D.iw2[e] = 0;
    }
    if (sc.leader()) { // list_init sets min_list = max(1, nlist) = m + 2 (list.rs:48-50)
        S->min_colnz = m + 2;
        S->min_rownz = m + 2;
    }
    sc.sync();
    FILL_STAMP(S, 21); // setup: lists and marks initialised
    { // the builder waves of the scope share out the lists (Scope::list_roles)
        int part, nparts;
        bool cols, rows;
        sc.list_roles(part, nparts, cols, rows);
        // (a batch: the window -- its bitmaps are done with -- holds the tails of the lists of small keys, half of it each)
        const int kcap = win ? min(wincap / 2, m + 2) : 0;
        if (cols) {
            const int mn = wave_list_build(D.cflink, D.cblink, m, D.iw0, m + 2, part, nparts, win, kcap);
            if (lane_id() == 0 && mn < m + 2) atomicMin(&S->min_colnz, mn);
        }
        if (rows) {
            const int mn = wave_list_build(D.rflink, D.rblink, m, D.iw1, m + 2, part, nparts, win ? win + wincap / 2 : nullptr, kcap);
            if (lane_id() == 0 && mn < m + 2) atomicMin(&S->min_rownz, mn);
        }
    }
#ifdef BLU_PROFILE_FILLS
    sc.sync(); // (the builder waves are done)
#endif
    FILL_STAMP(S, 22); // setup: count lists built
    if (sc.leader()) {
        const long long l_nz = S->lused, u_nz = S->uused;
        S->bump_nz = S->matrix_nz - l_nz - u_nz - rank - dropped_nz; // setup_bump.rs:89, :155
        S->bump_size = m - rank;
        S->cused = cused;
        S->rused = rused;
        S->pivot_row = -1;
        S->pivot_col = -1;
        S->rankdef = 0;
    }
}
// NT = threads of the workgroup at most (512: a batch -- the register budget of two waves per SIMD; 1024 otherwise)
template <int NT> __global__ void __launch_bounds__(NT) k_setup(DevLU *Ds, int nmat, int winbytes)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    BLU_DYN_SHARED(unsigned char, setup_win, 144 * 1024); // (the flag bitmaps: winbytes of dynamic LDS, or none)
    for (int b = blockIdx.x; b < nmat; b += gridDim.x) {
        const DevG D(Ds[b]);
        BlockScope sc{sh, shl};
        setup_body(D, sc, winbytes > 0 ? (int *)setup_win : nullptr, winbytes / 4);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_setup_grid(DevLU *Ds, GridWs *gw)
{
    __shared__ int sh[40];
    __shared__ long long shl[20];
    const DevG D(Ds[0]);
    GridScope sc{sh, shl, gw, 0};
    setup_body(D, sc);
}
