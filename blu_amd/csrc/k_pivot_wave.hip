// k_pivot_wave.hip -- the bump factorization with ONE WAVE PER MATRIX (included by k_pivot.hip, configuration
// pv_wave): the pivot loop of a BATCH of independent bases, built for residency.
//
// Reference: factorize_bump (src/lu/factorize_bump.rs:12-49), markowitz (src/lu/markowitz.rs:34-123),
// pivot_small (src/lu/pivot.rs:460-833), pivot_singleton_col (:928-1025), list_move (src/lu/list.rs:89-99).
//
// Why: the four-wave workgroups of k_pivot_loop_batch keep three waves waiting while one searches, and give every
// line of a pivot a whole wave (a 20-entry column uses 20 of 64 lanes for ~270 vector instructions).  A batch is
// bound by the instruction issue of the CUs, so this kernel (a) runs a matrix on a single wave -- no workgroup
// barrier, every resident wave is an active instruction stream -- and (b) packs the lines of a pivot into the lanes:
//
//   * FLATTENED line updates.  The entries of all columns of the pivot row form one index space (column after
//     column); a pass handles 64 of its entries, whichever columns they belong to.  Which column a lane is in comes
//     from a 64-bit word of segment-head bits (one LDS word per pass; the columns set their bits with one LDS
//     atomic); ranks inside a column are ballot prefixes relative to the column's first lane.  The rows of the
//     pivot column are handled the same way.
//   * The update proper (pivot.rs:581-690: work -= a * col, drop below droptol, append in pivot-column order) runs
//     over (column, position) pairs, floor(64 / cnz1) columns per pass; the old values of the entries being
//     updated cross from the first pass to the second through a small LDS matrix, group of columns by group.
//   * The appended part of the rows (pivot.rs:752-758) runs over (row, position) pairs likewise.
//
// Everything result-affecting is as in the general paths of k_pivot.hip (entry order inside lines, list order,
// arithmetic); those remain the fallback for every other shape -- they run here with a workgroup of one wave.
//
// Lanes of a wave run in lockstep on the GPU; the CPU emulation build (emu/hip/hip_runtime.h) needs the places that
// rely on it marked: WAVE_LOCKSTEP().

// Diagnostic build (-DBLU_PROFILE, `make prof`): lane 0 adds the shader-clock ticks since the previous stamp to phase k
// (tools/wave_phases.py prints the table).  The product build contains no stamps.
#ifdef BLU_PROFILE
#define WV_T(k)                                                                \
    do {                                                                       \
        if (threadIdx.x == 0) {                                                \
            const long long t_ = (long long)__builtin_amdgcn_s_memtime();      \
            sm->prof[k] += t_ - g_pstamp[0];                                   \
            sm->prof[24 + (k)] += 1;                                           \
            g_pstamp[0] = t_;                                                  \
        }                                                                      \
    } while (0)
#else
#define WV_T(k) \
    do {        \
    } while (0)
#endif

__device__ __forceinline__ unsigned wv_hslot(int k) { return ((unsigned)k * 2654435761u) >> (32 - WV_HBITS); }
__device__ __forceinline__ void wv_probe_overrun(int line)
{
    g_pivot_err = 1;
    g_pivot_err_line = line;
}
// every lane: empty the table (two words per lane)
__device__ __forceinline__ void wv_hclear_t(unsigned long long *tab)
{
    const int lane = lane_id();
    tab[lane] = ~0ull;
    tab[lane + 64] = ~0ull;
}
__device__ __forceinline__ void wv_hinsert_t(unsigned long long *tab, int k, int v)
{
    unsigned s = wv_hslot(k);
    const unsigned long long want = ((unsigned long long)(unsigned)k << 32) | (unsigned)v;
    for (int n = 0; n < WV_HASH; n++) {
        const unsigned long long old = atomicCAS(&tab[s], ~0ull, want);
        if (old == ~0ull || (int)(old >> 32) == k) return;
        s = (s + 1) & (WV_HASH - 1);
    }
    wv_probe_overrun(__LINE__);
}
// value of key k, -1 if absent.  The first two probes are read TOGETHER (one LDS round trip; the table is at most
// half full, so they decide practically every look-up) and combined without a branch; a third probe is rare.
__device__ __forceinline__ int wv_hfind_t(const unsigned long long *tab, int k)
{
    const unsigned s0 = wv_hslot(k), s1 = (s0 + 1) & (WV_HASH - 1);
    const unsigned long long x0 = tab[s0], x1 = tab[s1];
    const bool h0 = (int)(x0 >> 32) == k, e0 = x0 == ~0ull, h1 = (int)(x1 >> 32) == k, e1 = x1 == ~0ull;
    int r = h0 ? (int)(x0 & 0xffffffffull) : ((!e0 && h1) ? (int)(x1 & 0xffffffffull) : -1);
    if (!(h0 || e0 || h1 || e1)) { // both slots taken by other keys: go on probing
        unsigned s = (s1 + 1) & (WV_HASH - 1);
        r = -1;
        bool found = false;
        for (int n = 2; n < WV_HASH && !found; n++) {
            const unsigned long long x = tab[s];
            if ((int)(x >> 32) == k) {
                r = (int)(x & 0xffffffffull);
                found = true;
            } else if (x == ~0ull) {
                found = true;
            }
            s = (s + 1) & (WV_HASH - 1);
        }
        if (!found) wv_probe_overrun(__LINE__);
    }
    return r;
}
__device__ __forceinline__ void wv_hclear(Fast *fa) { wv_hclear_t(fa->hsh); }
__device__ __forceinline__ void wv_hinsert(Fast *fa, int k, int v) { wv_hinsert_t(fa->hsh, k, v); }
__device__ __forceinline__ int wv_hfind(const Fast *fa, int k) { return wv_hfind_t(fa->hsh, k); }

// exclusive prefix sum over the wave; *total = sum
__device__ __forceinline__ int wv_excl_scan(int v, int *total)
{
    const int inc = wave_incl_scan_i(v);
    *total = __builtin_amdgcn_readlane(inc, 63);
    return inc - v;
}
// 64-bit mask helpers (per-lane constants of the paired passes)
__device__ __forceinline__ unsigned long long wv_bits_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// What the search lays out for the pivot functions, one value per lane:
//   lanes c < rnz1: column slot c+1 of the pivot row;  lane rnz1: the pivot column (its list links only)
//   lanes p < cnz1: row slot p+1 of the pivot column (kind 1)
// what the search hands to the pivot loop in (wave-uniform) registers: the same values it leaves in LDS for the
// general paths, without the LDS round trip
struct WvPick {
    int pr, pc, nzc, nzr, kind, exit_code;
};
struct WvLines {
    int j, cb, cl, cap, fl, bl; // column: index, begin, length, capacity, count-list links
    int i, rb, rl, rc;          // row: index, begin, length, capacity
    double pv;                  // pivot-column value of that row
};

// ------------------------------------------------------------------------------------------------
// Batched list_move of the pivot row's columns + removal of the pivot column (list.rs:81-99), ONE pass.
// Lane c < n moves element e to the list of `key` (key < 0: not moved); lane n (isgone) only unlinks its
// element.  fl/bl = the element's links as loaded by the search.  The membership test "is this neighbour moved
// too" is the column hash (value = slot; slot s lives in lane s-1, slot 0 = the pivot column in lane n).
// Sequential list_moves leave every list as [untouched elements in old order][moved ones in move order], so
// "unlink all, append all in lane order" is the same thing (see wave_list_move_batch, k_pivot.hip).
// Returns the smallest key > 0 (or big).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wv_lane_of(const Fast *fa, int e, bool want, int n)
{
    if (!want) return -1;
    const int s = wv_hfind(fa, e);
    return s < 0 ? -1 : (s == 0 ? n : s - 1);
}
// t = the tail of the lane's new list, D.cblink[m + key], as loaded by the caller BEFORE the unlinks (a load the caller
// can issue as soon as the keys are known and leave in flight over other work)
__device__ __forceinline__ int wv_list_move(const DevGP &D, Fast *fa, int e, int key, bool act, bool isgone, int fl, int bl, int n, int big,
                                            int t)
{
    const int lane = lane_id();
    const int m = D.m;
    const bool unl = act || isgone;
    // lanes of equal key meet in an LDS word indexed by the key (all-zero between calls); large keys by ballots
    unsigned long long mygrp = 0ull;
    if (!__ballot(act && key >= WV_ZW)) {
        if (act) atomicOr(&fa->zw[key], 1ull << lane);
        wave_mem_sync();
        if (act) mygrp = fa->zw[key];
        WAVE_LOCKSTEP();
        if (act) fa->zw[key] = 0ull;
    } else {
        unsigned long long active = __ballot(act);
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int k = __builtin_amdgcn_readlane(key, leader);
            const unsigned long long grp = __ballot(act && key == k);
            if (key == k) mygrp = grp;
            active &= ~grp;
        }
    }
    const unsigned long long below = mygrp & lanes_below(lane);
    const unsigned long long above = lane < 63 ? mygrp & ~((2ull << lane) - 1ull) : 0ull;
    const int prevl = below ? 63 - __clzll((long long)below) : -1;
    const int nextl = above ? __ffsll((long long)above) - 1 : -1;
    const int eprev = __shfl(e, prevl >= 0 ? prevl : lane), enext = __shfl(e, nextl >= 0 ? nextl : lane);
    const int minall = wave_min_i(act && key > 0 ? key : big);
    // lanes holding my successor / my predecessor / the old tail of my new list
    int sl = wv_lane_of(fa, fl, unl && fl < m, n), pl = wv_lane_of(fa, bl, unl && bl < m, n), tl = wv_lane_of(fa, t, act && t < m, n);
    const bool first = unl && pl < 0; // first of a run of moved neighbours
    // first unmoved element after / before every moved one: pointer doubling over the lanes
    int fs = fl, fp = bl;
    for (int round = 0; round < 7; round++) {
        if (!__ballot((unl && sl >= 0) || (unl && pl >= 0))) break;
        const int srcs = sl >= 0 ? sl : lane, srcp = pl >= 0 ? pl : lane;
        const int fs2 = __shfl(fs, srcs), sl2 = __shfl(sl, srcs);
        const int fp2 = __shfl(fp, srcp), pl2 = __shfl(pl, srcp);
        if (sl >= 0) {
            fs = fs2;
            sl = sl2;
        }
        if (pl >= 0) {
            fp = fp2;
            pl = pl2;
        }
    }
    if (first) { // link the run's unmoved predecessor to its unmoved successor
        D.cflink[bl] = fs;
        D.cblink[fs] = bl;
    }
    // the old tail is being moved itself: the real tail is its nearest unmoved predecessor
    const int tfix = __shfl(fp, tl >= 0 ? tl : lane);
    if (act && tl >= 0) t = tfix;
    if (isgone) { // list.rs:84-85: a removed element links to itself
        D.cflink.el(e) = e;
        D.cblink.el(e) = e;
    }
    WAVE_LOCKSTEP(); // unlink stores before append stores (one address may get both; a wave's stores keep their order)
    if (act) {
        D.cblink.el(e) = prevl >= 0 ? eprev : t;
        D.cflink.el(e) = nextl >= 0 ? enext : m + key;
        if (prevl < 0) D.cflink[t] = e;
        if (nextl < 0) D.cblink.hd(key) = e;
    }
    wave_mem_sync();
    return minall;
}

// ------------------------------------------------------------------------------------------------
// The list walk of a search (markowitz.rs:73-123: the first maxsearch columns in count-list order) as a resumable
// sequence of steps, one memory round trip each: the list heads, then link + metadata of one candidate column at a
// time.  A search runs the steps back to back.  A pivot_small STARTS the walk of the NEXT search as soon as its own
// list move is done (lists and column metadata are final then, unless a column has to be removed) and advances it
// one step at each stage of the rest of its work -- row epilogue, row append, L column, clean-up, pivot record --
// so that the next search finds its candidates waiting: the walk is the longest dependent chain of a search.
// ------------------------------------------------------------------------------------------------
// MEASURED (round 3, MI355X): with the early start the walk disappears from the search (7 240 -> 1 070 ticks per search)
// but every stage that advances it gets ~2 000 ticks slower, and the pivot kernel as a whole 7 % slower (C2: 0.146 ->
// 0.156 s at 1024 bases, C3: 2.10 -> 2.23 s at 1536): a step has to wait for its loads with s_waitcnt vmcnt, and on
// gfx950 loads and stores share that in-order counter -- the wait also drains the scattered stores the stage before
// has just issued, which nothing waits for otherwise.  So the early start is OFF (WV_EARLY_WALK 0); a search runs the
// steps back to back.
#ifndef WV_EARLY_WALK
#define WV_EARLY_WALK 0
#endif
struct WvWalk {
    int st;     // 0 idle, 1 list heads in flight, 2 a candidate in flight, 3 done: candidates in fa->c*, 4 not handled here (the
                // general search decides), 5 an empty column heads list 0, 6 a column singleton heads list 1
    int nz0, h; // h: one list head per lane (lane 0: list 0; lane l: list nz0 + l - 1)
    unsigned long long ne;
    int j, znz, ncand, total, found_nz, guard;
    int fl, cb, cl; // in flight: link, begin, length, maximum of column j
    double cmx;
};
__device__ __forceinline__ void ew_issue(const DevGP &D, WvWalk &E)
{
    E.fl = D.cflink.el(E.j);
    E.cb = D.cbeg[E.j];
    E.cl = D.clen[E.j];
    E.cmx = D.colmax[E.j];
}
__device__ __forceinline__ void ew_begin(const DevGP &D, WvWalk &E, int nz0)
{
    const int lane = lane_id();
    const int m = D.m;
    E.nz0 = nz0;
    const int kk = lane == 0 ? 0 : nz0 + lane - 1;
    E.h = kk <= m ? D.cflink.hd(kk) : m + kk;
    E.ne = 0ull;
    E.j = 0;
    E.znz = E.ncand = E.total = E.guard = 0;
    E.found_nz = -1;
    E.fl = E.cb = E.cl = 0;
    E.cmx = 0.0;
    E.st = nz0 >= 1 ? 1 : 4;
}
// the next list with a member, or the end of the walk
__device__ __forceinline__ void ew_next_list(const DevGP &D, WvWalk &E)
{
    if (E.ne) {
        const int b = __ffsll((long long)E.ne) - 1;
        E.ne &= E.ne - 1;
        E.j = __builtin_amdgcn_readlane(E.h, b);
        E.znz = E.nz0 + b - 1;
        E.guard = 0;
        ew_issue(D, E);
        E.st = 2;
    } else {
        E.st = 3;
    }
}
__device__ __forceinline__ void ew_step(const DevGP &D, Sm *sm, WvWalk &E, int K)
{
    const int lane = lane_id();
    const int m = D.m;
    Fast *fa = &sm->fa;
    if (E.st == 1) {
        const int kk = lane == 0 ? 0 : E.nz0 + lane - 1;
        const int h0 = __builtin_amdgcn_readlane(E.h, 0);
        E.ne = __ballot(lane >= 1 && kk <= m && E.h != m + kk);
        if (h0 != m) E.st = 5;                         // empty column: chosen immediately (markowitz.rs:73-78)
        else if (!E.ne) E.st = 4;                      // (a long stretch of empty lists: the general search skips them 64 at a time)
        else if (E.nz0 == 1 && (E.ne & 2ull)) E.st = 6; // a column singleton
        else ew_next_list(D, E);
    } else if (E.st == 2) {
        if (E.cl != E.znz || E.cmx == 0.0 || !(E.cmx >= D.abstol) || ++E.guard > m + 2) {
            E.st = 4; // reference: assert / D2; the general search raises it
            return;
        }
        if (lane == 0) {
            fa->cJ[E.ncand] = E.j;
            fa->cNz[E.ncand] = E.znz;
            fa->cB[E.ncand] = E.cb;
            fa->cL[E.ncand] = E.cl;
            fa->cMx[E.ncand] = E.cmx;
            fa->cOff[E.ncand] = E.total;
        }
        if (E.found_nz < 0) E.found_nz = E.znz;
        E.total += E.cl;
        E.ncand++;
        if (E.ncand >= K) {
            E.st = 3;
        } else if (E.fl < m) {
            E.j = E.fl;
            ew_issue(D, E);
        } else {
            ew_next_list(D, E);
        }
    }
}

#if WV_NW == 2
// Two waves per matrix: what the search (wave 0) found goes to LDS for both waves -- the metadata of every line of the
// pivot, both membership tables, and which lines each wave takes (by weight: entries to read + entries to append).
#ifndef WV2_SHARE0
#define WV2_SHARE0 50 // percent of the columns' weight wave 0 takes (wave 1 then has all the rows)
#endif
__device__ __forceinline__ void wv2_publish(Fast *fa, const WvLines &L, int rnz1, int cnz1)
{
    const int lane = lane_id();
    wv_hclear_t(fa->hshr); // (the table of the columns, fa->hsh: built by wave 1, wv2_small)
    wave_mem_sync();
    if (lane <= rnz1) {
        fa->lJ[lane] = L.j;
        fa->lFl[lane] = L.fl;
        fa->lBl[lane] = L.bl;
    }
    if (lane < rnz1) {
        fa->lCb[lane] = L.cb;
        fa->lCl[lane] = L.cl;
        fa->lCap[lane] = L.cap;
    }
    if (lane < cnz1) {
        wv_hinsert_t(fa->hshr, L.i, lane);
        fa->lI[lane] = L.i;
        fa->lRb[lane] = L.rb;
        fa->lRl[lane] = L.rl;
        fa->lRc[lane] = L.rc;
    }
    // a line goes to wave 0 if it BEGINS inside wave 0's share (the first line always does)
    int tc;
    const int cw = wv_excl_scan(lane < rnz1 ? L.cl + cnz1 : 0, &tc);
    const int cs = __popcll(__ballot(lane < rnz1 && (long long)cw * 100 < (long long)tc * WV2_SHARE0));
    if (lane == 0) {
        fa->csplit = cs;
        fa->tiny = 0;
    }
}
#endif

// Second half of the search: room in L and U, the kind of pivot, and -- for the two flattened kinds -- the pivot row
// (and column) in slot order with the metadata of every line they touch.  pv1 = the pivot value of a column singleton.
__device__ __forceinline__ bool wv_layout(const DevGP &D, Sm *sm, WvLines &L, WvPick &P, int pc, int pr, int nzc, int pcb, int nzr, int prb, int where,
                                          int found_nz, int nsearched, double pv1)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    int exit_code = 0, need = 0;
    // room in L and U (pivot.rs:70-81)
    if (sm->lused + (nzc - 1) > D.lcap) {
        exit_code = ST_NEED_L;
        need = nzc - 1;
    } else if (sm->uused + (nzr - 1) > D.ucap) {
        exit_code = ST_NEED_U;
        need = nzr - 1;
    }
    if (lane == 0) {
        sm->pr = pr;
        sm->pc = pc;
        sm->pcb = pcb;
        sm->prb = prb;
        sm->nzc = nzc;
        sm->nzr = nzr;
        sm->nsearch += nsearched;
        sm->min_colnz = found_nz;
        sm->flag_small = 0;
        sm->ncancel = 0;
        fa->anycancel = 0;
        if (exit_code) {
            sm->exit_code = exit_code;
            sm->need = need;
        }
    }
    DEV_CHECK(S, nzr >= 1 && nzc >= 1);
    int kind = 0;
    if (nzr >= 2 && nzr <= WV_SLOTS && !exit_code) {
        if (nzc == 1) kind = 2;
        else if (nzc >= 3 && nzc <= WV_SLOTS) kind = 1; // (64 lanes hold the pivot column: cnz1 <= 63)
    }
    P.pr = pr;
    P.pc = pc;
    P.nzc = nzc;
    P.nzr = nzr;
    P.exit_code = exit_code;
    P.kind = 0;
    if (kind == 0) {
        if (lane == 0) fa->kind = 0;
        wave_mem_sync();
        return true;
    }
    // ---- lay out the pivot: pivot row (and column) into slot order, metadata of every line they touch
    const int rnz1 = nzr - 1, cnz1 = nzc - 1;
    const int jq = lane < nzr ? D.ridx[prb + lane] : -1;
    int ci = -1;
    double cv = 0.0;
    if (kind == 1 && lane < nzc) {
        ci = D.cidx[pcb + lane];
        cv = D.cval[pcb + lane];
    }
    const unsigned long long hb = __ballot(jq == pc);
    if (!hb) {
        DEV_CHECK(S, false);
        if (lane == 0) {
            sm->pc = -1;
            sm->pr = -1;
            fa->kind = 0;
        }
        P.pc = P.pr = -1;
        wave_mem_sync();
        return true;
    }
    const int wpos = __ffsll((long long)hb) - 1;
    if (lane < nzr) {
        // kind 1: pivot column swapped with the first entry (pivot.rs:185); kind 2: taken out, order kept (:959-965)
        const int slot = kind == 1 ? (lane == wpos ? 0 : (lane == 0 ? wpos : lane)) : (lane == wpos ? 0 : (lane < wpos ? lane + 1 : lane));
        fa->tJ[slot] = jq;
    }
    if (kind == 1 && lane < nzc) {
        const int slot = lane == where ? 0 : (lane == 0 ? where : lane); // pivot.rs:169-170
        fa->pI[slot] = ci;
        fa->pV[slot] = cv;
    }
    if (kind == 2 && lane == 0) fa->pV[0] = pv1;
    wave_mem_sync();
    WV_T(4);
    L.j = lane <= rnz1 ? fa->tJ[lane < rnz1 ? lane + 1 : 0] : -1; // lane rnz1: the pivot column
    L.cb = L.cl = L.cap = L.fl = L.bl = 0;
    if (lane <= rnz1) {
        L.fl = D.cflink.el(L.j);
        L.bl = D.cblink.el(L.j);
    }
    if (lane < rnz1) {
        L.cb = D.cbeg[L.j];
        L.cl = D.clen[L.j];
        L.cap = D.ccap[L.j];
    }
    L.i = -1;
    L.rb = L.rl = L.rc = 0;
    L.pv = 0.0;
    if (kind == 1) {
        if (lane < cnz1) {
            L.i = fa->pI[lane + 1];
            L.pv = fa->pV[lane + 1];
            L.rb = D.rbeg[L.i];
            L.rl = D.rlen[L.i];
            L.rc = D.rcap[L.i];
        }
        if (fa->pI[0] != pr) DEV_CHECK(S, false);
    }
    // sizes of the flattened phases; room in the arenas if every line had to be re-appended (pivot.rs:156-208)
    long long gc = 0, gr = 0;
    if (kind == 1 && lane < rnz1) {
        const int n = L.cl + cnz1;
        gc = n + stretch_of(D.stretch, n) + D.pad;
    }
    if (kind == 1 && lane < cnz1) {
        const int n = L.rl + rnz1;
        gr = n + stretch_of(D.stretch, n) + D.pad;
    }
    const long long tboth = wave_sum_ll(((long long)L.cl << 32) | (long long)(unsigned)L.rl);
    if ((tboth >> 32) > WV_TMAX || (tboth & 0xffffffffLL) > WV_TMAX) kind = 0;
    if (D.carena_cap >= (1 << 29) || D.rarena_cap >= (1 << 29)) kind = 0; // (32-bit byte offsets into the arenas: wv_ld / wv_st)
    if (kind == 1) {
        const long long both = wave_sum_ll((gc << 32) | (gr & 0xffffffffLL));
        if ((long long)sm->cused + (both >> 32) > (long long)D.carena_cap || (long long)sm->rused + (both & 0xffffffffLL) > (long long)D.rarena_cap)
            kind = 0; // the general path makes the exact check and leaves with NEED_CW / NEED_RW
    }
    WV_T(5);
#if WV_NW == 2
    if (kind == 1) wv2_publish(fa, L, rnz1, cnz1);
#endif
    if (lane == 0) {
        fa->kind = kind;
        fa->where = wpos;
    }
    P.kind = kind;
    wave_mem_sync();
    return true;
}

// ------------------------------------------------------------------------------------------------
// Search (markowitz.rs:34-123, search_rows == 0) + pivot set-up.  Returns false if the shape is outside what this
// path handles (nothing modified: the caller runs markowitz_wave).  On true: sm->pr / sm->pc are set (pr < 0: an
// empty column was chosen; both < 0: error raised), fa->kind says which pivot function runs, L holds the lines.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wv_search(const DevGP &D, Sm *sm, WvLines &L, WvPick &P, WvWalk &E)
{
    const int lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int K = D.maxsearch;
    if (K < 1 || K > KCMAX || m >= (1 << 27)) { // (cost * 256 + position must fit 64 bits)
        E.st = 0;
        return false;
    }
    const bool handed = fa->nxValid != 0;
    WAVE_LOCKSTEP();
    if (handed) {
        // ---- the previous pivot (a column singleton) left the next one: see wv_scol
        E.st = 0;
        if (lane == 0) {
            fa->nxValid = 0;
            sm->nfast[2]++;
        }
        const int pc = fa->nxPc, pr = fa->nxPr;
        const int left = m - sm->rank - sm->rankdef;
        return wv_layout(D, sm, L, P, pc, pr, 1, fa->nxPcb, D.rlen[pr], D.rbeg[pr], 0, 1, left < K ? left : K, fa->nxVal);
    }
    // ---- the list walk: started by the previous pivot (see WvWalk), or from scratch
    if (E.st == 0) ew_begin(D, E, sm->min_colnz);
    else if (lane == 0) sm->nfast[3]++;
    while (E.st == 1 || E.st == 2) ew_step(D, sm, E, K);
    const int wst = E.st;
    E.st = 0;
    if (wst == 4) return false;
    if (wst == 5) {
        const int h0 = __builtin_amdgcn_readlane(E.h, 0);
        if (lane == 0) {
            sm->pc = h0;
            sm->pr = -1;
            fa->kind = 0;
        }
        P.pc = h0;
        P.pr = -1;
        P.nzc = P.nzr = P.kind = P.exit_code = 0;
        wave_mem_sync();
        return true;
    }
    WV_T(2);
    const int left = m - sm->rank - sm->rankdef; // every active column is in a count list; list 0 is empty
    const int nsearched = left < K ? left : K;

    int pc, pr, nzc, pcb, nzr, prb, where, found_nz;
    double pv1 = 0.0;
    if (wst == 6) {
        // ---- column singleton: its one entry costs 0 and no later candidate can be strictly cheaper
        // (markowitz.rs:105); the reference still looks at maxsearch columns, which only shows in nsearch_pivot
        pc = __builtin_amdgcn_readlane(E.h, 1);
        pcb = D.cbeg[pc];
        const int cl = D.clen[pc];
        const double cmx = D.colmax[pc];
        if (cl != 1 || cmx == 0.0 || !(cmx >= D.abstol)) return false; // (the general search raises the error)
        pr = D.cidx[pcb];
        pv1 = D.cval[pcb];
        const double x = fabs(pv1);
        const double tol = fmax(D.abstol, D.reltol * cmx);
        if (x == 0.0 || x < tol) return false;
        nzc = 1;
        nzr = D.rlen[pr];
        prb = D.rbeg[pr];
        where = 0;
        found_nz = 1;
    } else {
        const int ncand = E.ncand, total = E.total;
        found_nz = E.found_nz;
        if (ncand < nsearched) return false; // more columns exist in lists beyond nz0+62
        if (total > WV_STG) return false;
        if (lane == 0) fa->cOff[ncand] = total;
        wave_mem_sync();
        const int off1 = ncand > 1 ? fa->cOff[1] : 0x7fffffff, off2 = ncand > 2 ? fa->cOff[2] : 0x7fffffff,
                  off3 = ncand > 3 ? fa->cOff[3] : 0x7fffffff;
        // ---- all candidate entries in one flattened pass (two for > 64): cost of every eligible entry; the
        // reference's sequential strict-< scan is the lexicographic minimum over (cost, flat position)
        const long long BIG = 0x7fffffffffffffffLL;
        long long mcb = BIG;
        int idx0 = 0, idx1 = 0, rl0 = 0, rl1 = 0, rb0 = 0, rb1 = 0;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int f = u * 64 + lane;
            if (f < total) {
                const int c = (f >= off1) + (f >= off2) + (f >= off3);
                const int pos = fa->cB[c] + f - fa->cOff[c];
                const int idx = D.cidx[pos];
                const double val = D.cval[pos];
                const int rl = D.rlen[idx], rbg = D.rbeg[idx]; // (the winner's row begin comes with its length: one round trip less)
                const double tol = fmax(D.abstol, D.reltol * fa->cMx[c]);
                const double x = fabs(val);
                if (!(x == 0.0 || x < tol)) {
                    const long long key = (long long)(fa->cNz[c] - 1) * (long long)(rl - 1) * 256LL + (long long)f;
                    if (key < mcb) mcb = key;
                }
                if (u == 0) {
                    idx0 = idx;
                    rl0 = rl;
                    rb0 = rbg;
                } else {
                    idx1 = idx;
                    rl1 = rl;
                    rb1 = rbg;
                }
            }
        }
        const long long best = wave_min_ll(mcb);
        if (best == BIG) return false; // no eligible entry: cannot happen; the general search raises it
        WV_T(3);
        const int fsel = (int)(best & 255LL);
        const int csel = (fsel >= off1) + (fsel >= off2) + (fsel >= off3);
        pc = fa->cJ[csel];
        nzc = fa->cL[csel];
        pcb = fa->cB[csel];
        where = fsel - fa->cOff[csel];
        pr = __builtin_amdgcn_readlane(fsel < 64 ? idx0 : idx1, fsel & 63);
        nzr = __builtin_amdgcn_readlane(fsel < 64 ? rl0 : rl1, fsel & 63);
        prb = __builtin_amdgcn_readlane(fsel < 64 ? rb0 : rb1, fsel & 63);
    }
    WV_T(1);
    return wv_layout(D, sm, L, P, pc, pr, nzc, pcb, nzr, prb, where, found_nz, nsearched, pv1);
}

// Segment bookkeeping of a flattened pass.  `hw` = head bits of this pass, `cbv` = (slot of the last line begun
// before this pass); gives this lane's slot, whether it is the last lane of its line IN THIS PASS, and what the
// rank of an entry inside its line needs: the lanes below the line's first lane of this pass, and whether the line
// continues from the previous pass (its count so far is then carried in a scalar, not read back from LDS).
struct WvSeg {
    int c;
    bool tail, first_seg;
    unsigned long long below_h;
};
__device__ __forceinline__ WvSeg wv_segment(unsigned long long hw, int cbv, bool valid, bool lastflat)
{
    const int lane = lane_id();
    const unsigned long long le = (2ull << lane) - 1ull; // lanes at or below this one
    const int own = (int)((hw >> lane) & 1ull);
    WvSeg s;
    s.c = cbv + wave_prefix_count(hw) + own;
    const unsigned long long mine = (hw | 1ull) & le;
    const int h = 63 - __clzll((long long)mine); // first lane of this lane's line in this pass
    s.below_h = (1ull << h) - 1ull;
    s.first_seg = (hw & le) == 0ull;
    s.tail = valid && (lane == 63 || lastflat || (((hw >> 1) >> lane) & 1ull));
    return s;
}
// rank of a kept entry among the kept entries of its line (kb = ballot of the kept ones)
__device__ __forceinline__ int wv_rank(const WvSeg &sg, unsigned long long kb, int carry)
{
    return wave_prefix_count(kb) - __popcll(kb & sg.below_h) + (sg.first_seg ? carry : 0);
}

// Arena accesses of the passes with a 32-bit BYTE offset from the (scalar) array base: `global_load v, v_off, s[base]`
// instead of a 64-bit address built per lane (sign extension + two 64-bit shift-adds per access).  The flattened
// paths are taken only while the arenas are below 2^29 entries (wv_layout), so the offsets fit.
template <class T> __device__ __forceinline__ T wv_ld(GPTR(const T) base, int i)
{
    return *(GPTR(const T))((GPTR(const char))base + (unsigned)i * (unsigned)sizeof(T));
}
template <class T> __device__ __forceinline__ void wv_st(GPTR(T) base, int i, T v)
{
    *(GPTR(T))((GPTR(char))base + (unsigned)i * (unsigned)sizeof(T)) = v;
}

// One pass of a flattened phase, loads issued: which line each lane is in, the entry it holds.  The passes are
// software-pipelined -- pass k+1 is fetched before pass k is worked on -- because a wave alone on its matrix has
// nothing else to hide a memory round trip behind, and a small pivot has ~16 such passes.  (The in-place stores of
// pass k go to positions below the entries pass k read, so they never touch what pass k+1 has already loaded.)
struct WvPass {
    WvSeg sg;
    bool valid;
    int2 bo;
    int e, idx;
    double val;
};
// The loads are issued by EVERY lane on EVERY call (lanes past the end, and calls past the last pass, read entry 0 of
// the arena): with a fixed number of loads in flight the compiler waits for "all but the newest two" when pass k is
// worked on; loads under a branch would make it wait for everything, i.e. for the pass just fetched.
template <bool VALUES>
__device__ __forceinline__ WvPass wv_fetch_z(const DevGP &D, Fast *fa, unsigned long long *zw, const int2 *sbo, gcint_p idxarr, int k, int T, int f0, int &cbv)
{
    const int lane = lane_id();
    WvPass P;
    const int kw = k < WV_ZW ? k : WV_ZW - 1; // (a call past the last pass: a word that is zero already)
    const unsigned long long hw = zw[kw];
    WAVE_LOCKSTEP();
    if (lane == 0) zw[kw] = 0ull;
    const int f = k * 64 + lane;
    P.valid = f < T;
    P.sg = wv_segment(hw, cbv, P.valid, f == T - 1);
    cbv += __popcll(hw);
    P.bo = sbo[P.sg.c]; // (a lane past the end: the slot of the last line -- a valid slot, its entry is not used)
    P.e = f + f0 - P.bo.y;
    const int pos = P.valid ? P.bo.x + P.e : 0;
    P.idx = wv_ld<int>(idxarr, pos);
    P.val = 0.0;
    if (VALUES) P.val = wv_ld<double>(D.cval, pos);
    return P;
}
template <bool VALUES>
__device__ __forceinline__ WvPass wv_fetch(const DevGP &D, Fast *fa, gcint_p idxarr, int k, int T, int f0, int &cbv)
{
    return wv_fetch_z<VALUES>(D, fa, fa->zw, fa->sBO, idxarr, k, T, f0, cbv);
}

#if WV_NW == 1
// ------------------------------------------------------------------------------------------------
// pivot_small (pivot.rs:460-833), pivot row of <= 64 entries
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wv_small(const DevGP &D, Sm *sm, const WvLines &L, WvWalk &E, int pr, int pc, int nz_col, int nz_row)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int cnz1 = nz_col - 1, rnz1 = nz_row - 1;
    const double pivot = fa->pV[0];
    DEV_CHECK(S, pivot != 0.0);
    const double droptol = D.droptol;

    // ---- rows of the pivot column -> position (the reference's `marked`, pivot.rs:219-224)
    wv_hclear(fa);
    wave_mem_sync();
    if (lane < cnz1) wv_hinsert(fa, L.i, lane);
    int Tc;
    const int coff = wv_excl_scan(lane < rnz1 ? L.cl : 0, &Tc);
    if (lane < rnz1) {
        fa->sBO[lane] = make_int2(L.cb, coff);
        fa->sCnt[lane] = 0;
    }
    wave_mem_sync();

    WV_T(6);
    // ================= column file update (pivot.rs:566-691) =================
    // per-lane constants of the paired pass: lane = (column cl_l of the pass, position p_l)
    const int Cper = 64 / cnz1;
    const int cl_l = lane / cnz1, p_l = lane - cl_l * cnz1;
    const bool pairlane = cl_l < Cper;
    const int segs = cl_l * cnz1;
    const unsigned long long segbelow = wv_bits_below(segs);
    const unsigned long long segmask = pairlane ? (wv_bits_below(cnz1) << segs) : 0ull;
    const double pv_l = fa->pV[1 + p_l];
    const int pi_l = fa->pI[1 + p_l];
    int Gc = WV_WCAP / cnz1; // columns per group (their old values fit the LDS matrix)
    if (Gc < 1) Gc = 1;
    int nk1 = 0, dst = L.cb, newcap = L.cap; // lane = column: kept entries, new begin, new capacity
    double xrj = 0.0;
    int cused = sm->cused, nexp = 0, nd3 = 0;
    bool anyc = false;

    for (int c0 = 0; c0 < rnz1; c0 += Gc) {
        const int c1 = min(c0 + Gc, rnz1);
        const int f0 = __builtin_amdgcn_readlane(coff, c0);
        const int f1 = c1 < rnz1 ? __builtin_amdgcn_readlane(coff, c1) : Tc;
        const int Tg = f1 - f0;
        const bool ing = lane >= c0 && lane < c1;
        if (ing) {
            const int o = coff - f0;
            atomicOr(&fa->zw[o >> 6], 1ull << (o & 63));
        }
        wave_mem_sync();
        // ---- pass A: every entry of the group's columns.  Entries whose row is in the pivot column leave the
        // column (their value goes to W); the others are compressed in place, keeping their order
        int cbv = c0 - 1, carry = 0;
        const auto work_a = [&](const WvPass &P) {
            const WvSeg sg = P.sg;
            const bool valid = P.valid;
            const int2 bo = P.bo;
            const int idx = P.idx;
            const double val = P.val;
            const int p = valid ? wv_hfind(fa, idx) : -1;
            const bool hit = valid && p >= 0;
            const bool keep = valid && !hit;
            const bool ispr = keep && idx == pr;
            if (hit) sm->swork[(sg.c - c0) * cnz1 + p] = val;
            const unsigned long long kb = __ballot(keep);
            const int t = valid ? wv_rank(sg, kb, carry) : 0;
            carry = __builtin_amdgcn_readlane(t + (keep ? 1 : 0), 63); // (used only if the line at lane 63 goes on)
            if (sg.tail) fa->sCnt[sg.c] = t + (keep ? 1 : 0);
            if (ispr) {
                fa->sX[sg.c] = val;
                fa->sW[sg.c] = t;
            } else if (keep) {
                if (t == 0) { // it goes where the pivot-row entry was (the swap of pivot.rs:261-262): at the line's end
                    fa->sK0i[sg.c] = idx;
                    fa->sK0v[sg.c] = val;
                } else {
                    wv_st<int>(D.cidx, bo.x + t - 1, idx);
                    wv_st<double>(D.cval, bo.x + t - 1, val);
                }
                atomicMax(&fa->sMax[sg.c], (unsigned long long)__double_as_longlong(fabs(val)));
            }
        };
        // two passes per iteration, each with registers of its own: the pass being fetched never has to be copied
        // into the registers of the pass being worked on (a copy waits for the loads: no overlap)
        const int npass = (Tg + 63) >> 6;
        WvPass PA = wv_fetch<true>(D, fa, D.cidx, 0, Tg, f0, cbv), PB = PA;
        for (int k = 0; k < npass; k += 2) {
            PB = wv_fetch<true>(D, fa, D.cidx, k + 1, Tg, f0, cbv);
            work_a(PA);
            PA = wv_fetch<true>(D, fa, D.cidx, k + 2, Tg, f0, cbv);
            work_a(PB); // (past the last pass: every lane invalid, nothing happens)
        }
        WV_T(7);
        wave_mem_sync();
        // ---- per column: kept count, room (file_reappend, file.rs:56-85), the deferred first entry, multiplier
        bool reloc = false;
        if (ing) {
            const int cnt = fa->sCnt[lane], w = fa->sW[lane];
            DEV_CHECK(S, cnt >= 1);
            nk1 = cnt - 1;
            const int need = nk1 + cnz1;
            reloc = need > L.cap;
            if (reloc) newcap = need + stretch_of(D.stretch, need) + D.pad;
            if (w > 0) {
                D.cidx[L.cb + w - 1] = fa->sK0i[lane];
                D.cval[L.cb + w - 1] = fa->sK0v[lane];
            }
            xrj = fa->sX[lane];
        }
        const unsigned long long rlb = __ballot(reloc);
        if (rlb) { // (uniform)
            int tot;
            const int ex = wv_excl_scan(reloc ? newcap : 0, &tot);
            if (reloc) dst = cused + ex;
            cused += tot;
            nexp += __popcll(rlb);
            wave_mem_sync();
            unsigned long long rest = rlb;
            while (rest) { // copy the kept part of a re-appended column (one column at a time: not the common case)
                const int b = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const int src = __builtin_amdgcn_readlane(L.cb, b), nn = __builtin_amdgcn_readlane(nk1, b), dd = __builtin_amdgcn_readlane(dst, b);
                for (int t = lane; t < nn; t += 64) {
                    D.cidx[dd + t] = D.cidx[src + t];
                    D.cval[dd + t] = D.cval[src + t];
                }
            }
        }
        if (ing) {
            fa->sX[lane] = xrj / pivot;
            fa->sDst[lane] = dst + nk1;
        }
        wave_mem_sync();
        WV_T(8);
        // ---- pass B: (column, position) pairs: work -= a * col (pivot.rs:623-625), append what stays above
        // droptol in pivot-column order (:630-664)
        int wbase = 0;
        for (int cb0 = c0; cb0 < c1; cb0 += Cper, wbase += Cper * cnz1) {
            const int c = cb0 + cl_l;
            const bool act = pairlane && c < c1;
            double old = 0.0, a = 0.0;
            int dd = 0;
            if (act) {
                old = sm->swork[wbase + lane];
                sm->swork[wbase + lane] = 0.0;
                a = fa->sX[c];
                dd = fa->sDst[c];
            }
            const double x = mulsub(old, a, pv_l);
            const double ax = fabs(x);
            const bool kx = act && ax > droptol;
            const unsigned long long kxb = __ballot(kx);
            if (kx) {
                const int rank = wave_prefix_count(kxb) - __popcll(kxb & segbelow);
                wv_st<int>(D.cidx, dd + rank, pi_l);
                wv_st<double>(D.cval, dd + rank, x);
                atomicMax(&fa->sMax[c], (unsigned long long)__double_as_longlong(ax));
            }
            if (act && p_l == 0) {
                const unsigned long long cm = (~kxb & segmask) >> segs; // cancelled positions (pivot.rs:656-660)
                fa->sNew[c] = __popcll(kxb & segmask);
                fa->sM[c] = cm;
                if (cm) {
                    anyc = true;
                    nd3 += __popcll(cm >> 31);
                }
            }
        }
        wave_mem_sync();
    }

    WV_T(9);
    // ---- per column: new metadata, U row (pivot.rs:666-672), new list key
    int newlen = -1;
    bool tiny = false;
    if (lane < rnz1) {
        newlen = nk1 + fa->sNew[lane];
        const double cmx = __longlong_as_double((long long)fa->sMax[lane]);
        fa->sMax[lane] = 0ull;
        D.cbeg[L.j] = dst;
        D.clen[L.j] = newlen;
        D.ccap[L.j] = newcap;
        D.colmax[L.j] = cmx;
        tiny = cmx == 0.0 || cmx < D.abstol;
    }
    const bool ku = lane < rnz1 && fabs(xrj) > droptol;
    const unsigned long long kub = __ballot(ku);
    int uused = sm->uused;
    if (ku) {
        const int d = uused + wave_prefix_count(kub);
        D.uidx[d] = L.j;
        D.uval[d] = xrj;
    }
    uused += __popcll(kub);
    const unsigned long long tinyb = __ballot(tiny);
    const unsigned long long anycb = __ballot(anyc);
    // tails of the columns' new count lists: the keys are final here, the lists are not touched until the list move
    // below -- the load stays in flight over the whole row file update
    int ltail = 0;
    if (lane < rnz1) ltail = D.cblink.hd(newlen);

    WV_T(10);
    // ================= row file update (pivot.rs:695-775) =================
    // columns of the pivot row -> slot (slot 0 = the pivot column): membership for the rows and for the list move
    wv_hclear(fa);
    wave_mem_sync();
    if (lane <= rnz1) wv_hinsert(fa, L.j, lane < rnz1 ? lane + 1 : 0);
    int Tr;
    const int roff = wv_excl_scan(lane < cnz1 ? L.rl : 0, &Tr);
    if (lane < cnz1) {
        fa->sBO[lane] = make_int2(L.rb, roff);
        fa->sCnt[lane] = 0;
        atomicOr(&fa->zw[roff >> 6], 1ull << (roff & 63));
    }
    wave_mem_sync();
    WV_T(11);
    {
        int cbv = -1, carry = 0;
        const auto work_r = [&](const WvPass &P) {
            const WvSeg sg = P.sg;
            const bool valid = P.valid;
            const int j = P.idx;
            const bool keep = valid && wv_hfind(fa, j) < 0; // overlap with the pivot row leaves, pivot column included
            const unsigned long long kb = __ballot(keep);
            const int t = valid ? wv_rank(sg, kb, carry) : 0;
            carry = __builtin_amdgcn_readlane(t + (keep ? 1 : 0), 63);
            if (sg.tail) fa->sCnt[sg.c] = t + (keep ? 1 : 0);
            if (keep && t != P.e) wv_st<int>(D.ridx, P.bo.x + t, j);
        };
        const int npass = (Tr + 63) >> 6;
        WvPass PA = wv_fetch<false>(D, fa, D.ridx, 0, Tr, 0, cbv), PB = PA;
        for (int k = 0; k < npass; k += 2) {
            PB = wv_fetch<false>(D, fa, D.ridx, k + 1, Tr, 0, cbv);
            work_r(PA);
            PA = wv_fetch<false>(D, fa, D.ridx, k + 2, Tr, 0, cbv);
            work_r(PB);
        }
    }
    wave_mem_sync();
    WV_T(12);
    // ---- column count lists (pivot.rs:682-683, :797): every column of the pivot row to the list of its new count, in
    // pivot-row order; the pivot column leaves.  (Here, not at the end: the tails arrived during the rows pass, and
    // from here on the lists are final -- the walk of the next search can start.)
    const int mn = wv_list_move(D, fa, L.j, newlen, lane < rnz1, lane == rnz1, L.fl, L.bl, rnz1, D.m + 2, ltail);
    {
        const int K = D.maxsearch;
        E.st = 0;
        if (WV_EARLY_WALK && !tinyb && D.search_rows == 0 && !D.no_fast && K >= 1 && K <= KCMAX) { // (a column to be removed would change the lists again)
            const int cur = sm->min_colnz;
            ew_begin(D, E, mn < cur ? mn : cur);
        }
    }
    WV_T(17);
    int rnk = 0, rdst = L.rb, rnewcap = L.rc;
    int rused = sm->rused;
    {
        bool reloc = false;
        if (lane < cnz1) {
            rnk = fa->sCnt[lane];
            DEV_CHECK(S, rnk < L.rl); // the pivot-column entry at least has left
            const int need = rnk + rnz1;
            reloc = need > L.rc;
            if (reloc) rnewcap = need + stretch_of(D.stretch, need) + D.pad;
        }
        const unsigned long long rlb = __ballot(reloc);
        if (rlb) {
            int tot;
            const int ex = wv_excl_scan(reloc ? rnewcap : 0, &tot);
            if (reloc) rdst = rused + ex;
            rused += tot;
            nexp += __popcll(rlb);
            unsigned long long rest = rlb;
            while (rest) {
                const int b = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const int src = __builtin_amdgcn_readlane(L.rb, b), nn = __builtin_amdgcn_readlane(rnk, b), dd = __builtin_amdgcn_readlane(rdst, b);
                for (int t = lane; t < nn; t += 64) D.ridx[dd + t] = D.ridx[src + t];
            }
        }
        if (lane < cnz1) fa->sDst[lane] = rdst + rnk;
    }
    wave_mem_sync();
    if (WV_EARLY_WALK && (E.st == 1 || E.st == 2)) ew_step(D, sm, E, D.maxsearch);
    WV_T(13);
    // ---- append the pattern of the pivot row, minus the cancelled positions (pivot.rs:752-758): (row, position) pairs
    int rnew = rnk + rnz1;
    {
        const int Rper = 64 / rnz1;
        const int pl_l = lane / rnz1, q_l = lane - pl_l * rnz1;
        const bool rpair = pl_l < Rper;
        const int rsegs = pl_l * rnz1;
        const unsigned long long rsegbelow = wv_bits_below(rsegs);
        const unsigned long long rsegmask = rpair ? (wv_bits_below(rnz1) << rsegs) : 0ull;
        const int tj_l = fa->tJ[1 + q_l];
        const unsigned long long cm_l = anycb ? fa->sM[q_l] : 0ull;
        for (int p0 = 0; p0 < cnz1; p0 += Rper) {
            const int p = p0 + pl_l;
            const bool act = rpair && p < cnz1;
            const bool ok = act && ((cm_l >> p) & 1ull) == 0ull;
            const unsigned long long okb = __ballot(ok);
            if (ok) {
                const int rank = wave_prefix_count(okb) - __popcll(okb & rsegbelow);
                wv_st<int>(D.ridx, fa->sDst[p] + rank, tj_l);
            }
            if (anycb && act && q_l == 0) fa->sNew[p] = __popcll(okb & rsegmask);
        }
        if (anycb) {
            wave_mem_sync();
            if (lane < cnz1) rnew = rnk + fa->sNew[lane];
        }
    }
    if (lane < cnz1) {
        D.rbeg[L.i] = rdst;
        D.rlen[L.i] = rnew;
        D.rcap[L.i] = rnewcap;
    }

    if (WV_EARLY_WALK && (E.st == 1 || E.st == 2)) ew_step(D, sm, E, D.maxsearch);
    WV_T(14);
    // ---- L column (pivot.rs:778-790)
    double lx = 0.0;
    if (lane < cnz1) lx = L.pv / pivot;
    const bool kl = lane < cnz1 && fabs(lx) > droptol;
    const unsigned long long klb = __ballot(kl);
    int lused = sm->lused;
    if (kl) {
        const int d = lused + wave_prefix_count(klb);
        D.lidx[d] = L.i;
        D.lval[d] = lx;
    }
    lused += __popcll(klb);
#ifdef WV_DEBUG_HAND
    if (lane == 0) printf("L column: rank %d cnz1 %d klb %llx sm->lused %d lused(after) %d lidx %p\n", sm->rank, cnz1, klb, sm->lused, lused, (void *)(int *)D.lidx);
#endif

    WV_T(15);
    if (WV_EARLY_WALK && (E.st == 1 || E.st == 2)) ew_step(D, sm, E, D.maxsearch);
    WV_T(16);
    // ---- cleanup (pivot.rs:792-800)
    WAVE_LOCKSTEP(); // (every lane has read sm->lused above before lane 0 replaces it)
    if (lane == 0) {
        const int rank = sm->rank;
        D.ubeg[rank + 1] = uused;
        D.lbeg[rank + 1] = lused;
        sm->uused = uused;
        sm->lused = lused;
        sm->cused = cused;
        sm->rused = rused;
        sm->nexpand += nexp;
        if (mn < sm->min_colnz) sm->min_colnz = mn;
        if (tinyb) sm->flag_small = 1;
        D.colmax[pc] = pivot;
        D.clen[pc] = 0;
        D.rlen[pr] = 0;
        sm->kinds[3]++;
        sm->nfast[0]++;
    }
    const int d3all = wave_sum_i(nd3);
    if (lane == 0 && d3all) sm->d3 += d3all;
    wave_mem_sync();
    if (WV_EARLY_WALK && (E.st == 1 || E.st == 2)) ew_step(D, sm, E, D.maxsearch);
    WV_T(16);
}
#endif // WV_NW == 1

// ------------------------------------------------------------------------------------------------
// pivot_singleton_col (pivot.rs:928-1025), pivot row of <= 64 entries: every column of the pivot row loses its
// pivot-row entry (the column's last entry moves into the hole, :991-993); no values change.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wv_scol(const DevGP &D, Sm *sm, const WvLines &L, int pr, int pc, int nz_row)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int rnz1 = nz_row - 1;
    const double pivot = fa->pV[0]; // (the search left it there)
    DEV_CHECK(S, pivot != 0.0);

    wv_hclear(fa);
    wave_mem_sync();
    if (lane <= rnz1) wv_hinsert(fa, L.j, lane < rnz1 ? lane + 1 : 0);
    int Tc;
    const int coff = wv_excl_scan(lane < rnz1 ? L.cl : 0, &Tc);
    if (lane < rnz1) {
        fa->sBO[lane] = make_int2(L.cb, coff);
        fa->sCnt[lane] = L.cl;
        fa->sW[lane] = -1;
        fa->sNew[lane] = -1;
        atomicOr(&fa->zw[coff >> 6], 1ull << (coff & 63));
    }
    wave_mem_sync();
    // tails of the columns' new count lists (every column loses exactly one entry): in flight over the pass below
    int ltail = 0;
    if (lane < rnz1) ltail = D.cblink.hd(L.cl - 1);
    WV_T(18);
    int cbv = -1;
    for (int k = 0; k * 64 < Tc; k++) {
        const unsigned long long hw = fa->zw[k];
        WAVE_LOCKSTEP();
        if (lane == 0) fa->zw[k] = 0ull;
        const int f = k * 64 + lane;
        const bool valid = f < Tc;
        const WvSeg sg = wv_segment(hw, cbv, valid, f == Tc - 1);
        cbv += __popcll(hw);
        if (valid) {
            const int2 bo = fa->sBO[sg.c];
            const int e = f - bo.y;
            const int idx = D.cidx[bo.x + e];
            const double val = D.cval[bo.x + e];
            if (idx == pr) {
                fa->sX[sg.c] = val;
                fa->sW[sg.c] = e;
            } else {
                atomicMax(&fa->sMax[sg.c], (unsigned long long)__double_as_longlong(fabs(val)));
                if (fa->sCnt[sg.c] == 2) { // the entry a column of two keeps: if that column is the next pivot, this is its pivot
                    fa->sNew[sg.c] = idx;
                    fa->sM[sg.c] = (unsigned long long)__double_as_longlong(val);
                }
            }
            // the last entry fills the hole -- needed only if it is not the pivot-row entry itself; (for a column of
            // two it is then the entry recorded above: same bytes, same values)
            if (e == fa->sCnt[sg.c] - 1 && idx != pr) {
                fa->sK0i[sg.c] = idx;
                fa->sK0v[sg.c] = val;
            }
        }
    }
    wave_mem_sync();
    WV_T(19);
    int newlen = -1;
    double xrj = 0.0;
    bool tiny = false;
    if (lane < rnz1) {
        const int w = fa->sW[lane];
        DEV_CHECK(S, w >= 0);
        xrj = fa->sX[lane];
        newlen = L.cl - 1;
        if (w >= 0 && w != newlen) {
            D.cidx[L.cb + w] = fa->sK0i[lane];
            D.cval[L.cb + w] = fa->sK0v[lane];
        }
        const double cmx = __longlong_as_double((long long)fa->sMax[lane]);
        fa->sMax[lane] = 0ull;
        D.clen[L.j] = newlen;
        D.colmax[L.j] = cmx;
        tiny = cmx == 0.0 || cmx < D.abstol;
    }
    const bool ku = lane < rnz1 && fabs(xrj) > D.droptol;
    const unsigned long long kub = __ballot(ku);
    int uused = sm->uused;
    if (ku) {
        const int d = uused + wave_prefix_count(kub);
        D.uidx[d] = L.j;
        D.uval[d] = xrj;
    }
    uused += __popcll(kub);
    const unsigned long long tinyb = __ballot(tiny);
    WV_T(20);
    const int mn = wv_list_move(D, fa, L.j, newlen, lane < rnz1, lane == rnz1, L.fl, L.bl, rnz1, D.m + 2, ltail);
    // ---- hand-over to the next search.  A chain of column singletons (the triangular part of an LP basis: half of all
    // pivots) uncovers one singleton per pivot: if the count-1 list held nothing but this pivot column, its head is now the
    // first column this pivot left with one entry -- and that entry passed through the lanes above.  The next search
    // then starts at the pivot row (6 dependent round trips less: list heads, column, entry).
    {
        // (the pivot column itself was the only member of the count-1 list: both its links were the list head)
        const bool only_pc = __builtin_amdgcn_readlane(L.fl, rnz1) == D.m + 1 && __builtin_amdgcn_readlane(L.bl, rnz1) == D.m + 1;
        const bool cand = lane < rnz1 && newlen == 1 && only_pc;
        const unsigned long long candb = __ballot(cand);
        const unsigned long long zerob = __ballot(lane < rnz1 && newlen == 0);
#ifdef WV_DEBUG_HAND
        const int dfl = __builtin_amdgcn_readlane(L.fl, rnz1), dbl = __builtin_amdgcn_readlane(L.bl, rnz1);
        if (lane == 0 && sm->rank < 40) printf("rank %d pc %d: only_pc %d (fl %d bl %d m+1 %d) candb %llx tinyb %llx zerob %llx\n", sm->rank, pc, (int)only_pc, dfl, dbl, D.m + 1, candb, tinyb, zerob);
#endif
        if (candb && !tinyb && !zerob) { // (uniform)
            const int first = __ffsll((long long)candb) - 1;
            if (lane == first) {
                const int oi = fa->sNew[lane];
                const double ov = __longlong_as_double((long long)fa->sM[lane]);
                const double x = fabs(ov);
                const double tol = fmax(D.abstol, D.reltol * x); // (the column's maximum is this entry)
                if (oi >= 0 && !(x == 0.0 || x < tol)) {
                    fa->nxPc = L.j;
                    fa->nxPr = oi;
                    fa->nxPcb = L.cb;
                    fa->nxVal = ov;
                    fa->nxValid = 1;
                }
            }
        }
    }
    if (lane == 0) {
        const int rank = sm->rank;
        D.ubeg[rank + 1] = uused;
        D.lbeg[rank + 1] = sm->lused; // empty column in L
        sm->uused = uused;
        if (mn < sm->min_colnz) sm->min_colnz = mn;
        if (tinyb) sm->flag_small = 1;
        D.colmax[pc] = pivot;
        D.clen[pc] = 0;
        D.rlen[pr] = 0;
        sm->kinds[1]++;
        sm->nfast[1]++;
    }
    wave_mem_sync();
    WV_T(21);
}

#if WV_NW == 2
#include "k_pivot_wave2.inc" // pivot_small dealt out to two waves, the pivot loop and kernel of that configuration
#else
// ------------------------------------------------------------------------------------------------
// the pivot loop of one matrix on one wave: factorize_bump (factorize_bump.rs:12-49) + pivot() (pivot.rs:48-112)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pivot_loop_wave(DevLU *Ds, int stop_at, Sm *sm)
{
    const DevGP D(&Ds[blockIdx.x]);
    Scalars *S = D.s;
    const int lane = lane_id();
    const int m = D.m;

    if (S->status != ST_RUNNING) return; // finished or failed in an earlier launch (batch relaunch)
    Fast *fa = &sm->fa;
    if (lane == 0) {
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
        sm->flag_small = 0;
        fa->kind = 0;
        fa->nxValid = 0;
#ifdef BLU_PROFILE
        for (int k = 0; k < 48; k++) sm->prof[k] = 0;
        g_pstamp[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
        for (int k = 0; k < 6; k++) sm->kinds[k] = 0;
        sm->nfast[0] = sm->nfast[1] = sm->nfast[2] = sm->nfast[3] = 0;
        g_pivot_err = 0;
        g_pivot_err_line = 0;
    }
    for (int k = lane; k < WV_WCAP; k += 64) sm->swork[k] = 0.0;
    if (lane == 0) sm->wmax[0] = 0ull;
    fa->zw[lane] = 0ull; // WV_ZW == 64
    fa->sMax[lane] = 0ull;
    wave_mem_sync();

    WvLines L;
    WvWalk E;
    E.st = 0;
    int rank = sm->rank, rankdef = sm->rankdef;
    bool pending = sm->pc >= 0; // a pivot left pending by a NEED_* exit of the previous launch
    for (;;) {
        // ---- loop head: done / stop / error?
        // (rank / rankdef / pending are wave-uniform registers kept beside their LDS copies, which the general paths read)
        int head_exit = 0;
        if (g_pivot_err) head_exit = ST_ERROR;
        else if (rank + rankdef >= m) head_exit = ST_DONE;
        else if (stop_at >= 0 && !pending && rank + rankdef >= stop_at) head_exit = ST_STOPPED;
        const bool need_search = !pending;
        pending = false;
        WAVE_LOCKSTEP();
        if (head_exit) {
            if (lane == 0) sm->exit_code = head_exit;
            break;
        }
        WV_T(0);
        // ---- find pivot (skipped when a pivot is pending from a NEED_* exit, factorize_bump.rs:19-21)
        bool handled = false;
        WvPick P;
        if (need_search) {
            if (D.search_rows == 0 && !D.no_fast) handled = wv_search(D, sm, L, P, E);
            if (!handled) {
                if (D.search_rows == 0) markowitz_wave(D, sm);
                else if (lane == 0) markowitz_serial(D, sm);
                wave_mem_sync();
            }
        }
        if (!handled) {
            if (lane == 0) setup_pivot_general(D, sm);
            wave_mem_sync();
        }
        if (!handled) { // (the general search / a pending pivot: through LDS)
            P.pr = sm->pr;
            P.pc = sm->pc;
            P.exit_code = sm->exit_code;
            P.nzc = sm->nzc;
            P.nzr = sm->nzr;
            P.kind = fa->kind;
            WAVE_LOCKSTEP();
        }
        const int pr = P.pr, pc = P.pc, exit_code = P.exit_code, nz_col = P.nzc, nz_row = P.nzr, kind = P.kind;
        if (pc < 0) { // no pivot found: the reference asserts (factorize_bump.rs:22)
            if (lane == 0) {
                DEV_CHECK(S, false);
                sm->exit_code = ST_ERROR;
            }
            break;
        }
        if (pr < 0) { // eliminate empty column without choosing a pivot (factorize_bump.rs:24-33)
            if (lane == 0) {
                list_remove1(D.cflink, D.cblink, pc);
                sm->pc = -1;
                sm->rankdef = rankdef + 1;
                sm->kinds[5]++;
            }
            rankdef++;
            wave_mem_sync();
            continue;
        }
        // ---- pivot(): the room check of pivot.rs:70-81 was made by the search; dispatch (:84-94)
        if (exit_code) break;
        bool ok = true;
        if (kind == 1) wv_small(D, sm, L, E, pr, pc, nz_col, nz_row);
        else if (kind == 2) wv_scol(D, sm, L, pr, pc, nz_row);
        else if (nz_row == 1) ok = pivot_singleton_row(D, sm);
        else if (nz_col == 1) ok = pivot_singleton_col(D, sm);
        else if (nz_col == 2) ok = pivot_doubleton_col(D, sm);
        else ok = pivot_general(D, sm, nz_col - 1 <= 64);
        if (kind == 0) WV_T(23);
        if (!ok) break; // exit_code set, pivot stays pending
        // ---- remove columns whose maximum dropped below abstol (pivot.rs:98-106), record the pivot
        if (lane == 0) {
            if (sm->flag_small && nz_row > 1) {
                for (int pos = D.ubeg[rank]; pos < D.ubeg[rank + 1]; pos++) {
                    const int j = D.uidx[pos];
                    if (D.colmax[j] == 0.0 || D.colmax[j] < D.abstol) remove_col_serial(D, sm, j);
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
        rank++;
        wave_mem_sync();
        if (WV_EARLY_WALK && (E.st == 1 || E.st == 2)) ew_step(D, sm, E, D.maxsearch);
        WV_T(22);
    }
    wave_mem_sync();
    if (lane == 0) {
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
        for (int k = 0; k < 4; k++) S->nfast[k] += sm->nfast[k];
#ifdef BLU_PROFILE
        for (int k = 0; k < 48; k++) S->prof[k] += sm->prof[k];
#endif
        if (sm->exit_code == ST_ERROR && g_pivot_err_line) set_error(S, ST_ERROR, g_pivot_err_line);
        if (S->status == ST_RUNNING) S->status = sm->exit_code;
    }
}

#ifndef BLU_WAVE_OCC
#define BLU_WAVE_OCC 4 // waves per SIMD the register budget is set for
#endif
__global__ void __launch_bounds__(64) BLU_WAVES_PER_EU(BLU_WAVE_OCC, BLU_WAVE_OCC) k_pivot_loop_wave(DevLU *Ds, int stop_at)
{
    __shared__ Sm smem;
    pivot_loop_wave(Ds, stop_at, &smem);
}
// The same loop with the registers of THREE waves per SIMD (168 VGPRs), for a batch of at most twelve bases per CU (3072):
// pivot kernel 1.129 -> 1.109 s at C4 x 3072.
__global__ void __launch_bounds__(64) BLU_WAVES_PER_EU(3, 3) k_pivot_loop_wave_r3(DevLU *Ds, int stop_at)
{
    __shared__ Sm smem;
    pivot_loop_wave(Ds, stop_at, &smem);
}
#endif // WV_NW
