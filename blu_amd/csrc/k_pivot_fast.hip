// k_pivot_fast.hip -- low-latency paths of the pivot loop (included by k_pivot.hip).
//
// The general paths in k_pivot.hip walk every line through HBM-resident metadata and mark arrays:
// ~40 dependent memory round trips and 12 barriers per pivot.  For the two pivot kinds that make up
// practically all pivots of LP-like bases -- pivot_small with a short pivot row, and
// pivot_singleton_col -- this file keeps the per-pivot working set in LDS instead:
//
//   * Markowitz: the (<= 4) candidate columns are found by walking the count list first, then ALL
//     their entries are evaluated in one flattened pass (lexicographic wave min over (cost, flat
//     position) == the reference's sequential strict-< scan, markowitz.rs:80-122); the candidates'
//     entries and the (begin,len,cap) of their rows are staged in LDS on the way.
//   * The chosen pivot column, the pivot row and the (begin,len,cap) of every line they touch are
//     laid out in LDS by the searching wave; membership tests "row in pivot column" / "column in
//     pivot row" (the reference's `marked` array, pivot.rs:219-224, 337-339) are LDS hash sets.
//   * Column and row updates of one pivot run concurrently (one line per wave); the rare numerical
//     cancellation (pivot.rs:656-660) is repaired afterwards.
//   * 4 barriers per pivot.
//
// Everything result-affecting (entry order inside lines, list order, arithmetic) is identical to
// the general paths, which remain the fallback for all other shapes.

__device__ __forceinline__ unsigned hslot(int k, int bits) { return ((unsigned)k * 2654435761u) >> (32 - bits); }
// Every probe sequence is bounded by the table size: the tables are sized for at most half load (PCMAX - 1
// keys in HROW slots, PRMAX keys in HCOL slots), so a sequence that visits every slot without finding the
// key or an empty slot means a violated invariant.  That raises the pivot loop's error flag (the loop
// leaves with ST_ERROR at the next pivot boundary) instead of spinning.
__device__ __forceinline__ void probe_overrun(int line)
{
    g_pivot_err = 1;
    g_pivot_err_line = line;
}
__device__ __forceinline__ void hrow_insert(Fast *f, int k, int v)
{
    unsigned s = hslot(k, HROW_BITS);
    const unsigned long long want = ((unsigned long long)(unsigned)k << 32) | (unsigned)v;
    for (int n = 0; n < HROW; n++) {
        const unsigned long long old = atomicCAS(&f->hRow[s], ~0ull, want);
        if (old == ~0ull || (int)(old >> 32) == k) return;
        s = (s + 1) & (HROW - 1);
    }
    probe_overrun(__LINE__);
}
__device__ __forceinline__ int hrow_lookup(const Fast *f, int k)
{
    unsigned s = hslot(k, HROW_BITS);
    for (int n = 0; n < HROW; n++) {
        const unsigned long long x = f->hRow[s];
        if ((int)(x >> 32) == k) return (int)(x & 0xffffffffull);
        if (x == ~0ull) return 0;
        s = (s + 1) & (HROW - 1);
    }
    probe_overrun(__LINE__);
    return 0;
}
__device__ __forceinline__ void hcol_insert(Fast *f, int k, int slot)
{
    unsigned s = hslot(k, HCOL_BITS);
    const unsigned long long want = ((unsigned long long)(unsigned)k << 32) | (unsigned)slot;
    for (int n = 0; n < HCOL; n++) {
        const unsigned long long old = atomicCAS(&f->hCol[s], ~0ull, want);
        if (old == ~0ull || (int)(old >> 32) == k) return;
        s = (s + 1) & (HCOL - 1);
    }
    probe_overrun(__LINE__);
}
__device__ __forceinline__ bool hcol_has(const Fast *f, int k)
{
    unsigned s = hslot(k, HCOL_BITS);
    for (int n = 0; n < HCOL; n++) {
        const unsigned long long x = f->hCol[s];
        if ((int)(x >> 32) == k) return true;
        if (x == ~0ull) return false;
        s = (s + 1) & (HCOL - 1);
    }
    probe_overrun(__LINE__);
    return false;
}
// slot of column k in the pivot row laid out by the last mk_pick, or -1
__device__ __forceinline__ int hcol_slot(const Fast *f, int k)
{
    unsigned s = hslot(k, HCOL_BITS);
    for (int n = 0; n < HCOL; n++) {
        const unsigned long long x = f->hCol[s];
        if ((int)(x >> 32) == k) return (int)(x & 0xffffffffull);
        if (x == ~0ull) return -1;
        s = (s + 1) & (HCOL - 1);
    }
    probe_overrun(__LINE__);
    return -1;
}
// Three look-ups at once in a table of (key << 32 | value) words: the probes of the three keys are
// issued together, so the whole costs about one LDS round trip.  r = value, or -1 if the key is absent
// (or its look-up not wanted).
template <int BITS>
__device__ __forceinline__ void hash_find3(const unsigned long long *H, int k1, int k2, int k3, bool w1, bool w2, bool w3, int &r1,
                                           int &r2, int &r3)
{
    const unsigned mask = (1u << BITS) - 1u;
    unsigned s1 = hslot(k1, BITS), s2 = hslot(k2, BITS), s3 = hslot(k3, BITS);
    r1 = r2 = r3 = -1;
    bool d1 = !w1, d2 = !w2, d3 = !w3;
    int n = 0;
    while (!(d1 && d2 && d3)) {
        if (++n > (1 << BITS)) {
            probe_overrun(__LINE__);
            break;
        }
        const unsigned long long x1 = H[s1], x2 = H[s2], x3 = H[s3];
        if (!d1) {
            if ((int)(x1 >> 32) == k1) r1 = (int)(x1 & 0xffffffffull);
            d1 = (int)(x1 >> 32) == k1 || x1 == ~0ull;
            s1 = (s1 + 1) & mask;
        }
        if (!d2) {
            if ((int)(x2 >> 32) == k2) r2 = (int)(x2 & 0xffffffffull);
            d2 = (int)(x2 >> 32) == k2 || x2 == ~0ull;
            s2 = (s2 + 1) & mask;
        }
        if (!d3) {
            if ((int)(x3 >> 32) == k3) r3 = (int)(x3 & 0xffffffffull);
            d3 = (int)(x3 >> 32) == k3 || x3 == ~0ull;
            s3 = (s3 + 1) & mask;
        }
    }
}
// ------------------------------------------------------------------------------------------------
// Links of the column count lists with the heads / tails of the short lists in LDS (Mc, k_pivot_fast_types.h).
// mc == nullptr (batch kernel): plain global accesses; every branch on mc folds away after inlining.
// ------------------------------------------------------------------------------------------------
struct LinksC {
    const DevGP &D;
    Mc *mc;
    __device__ __forceinline__ int fl(int e) const
    {
        const int k = e - D.m;
        if (mc && k >= 0 && k < MC_HEADS) return mc->hf[k];
        return D.cflink[e];
    }
    __device__ __forceinline__ int bl(int e) const
    {
        const int k = e - D.m;
        if (mc && k >= 0 && k < MC_HEADS) return mc->hb[k];
        return D.cblink[e];
    }
    __device__ __forceinline__ void set_fl(int e, int v) const
    {
        D.cflink[e] = v;
        const int k = e - D.m;
        if (mc && k >= 0 && k < MC_HEADS) mc->hf[k] = v;
    }
    __device__ __forceinline__ void set_bl(int e, int v) const
    {
        D.cblink[e] = v;
        const int k = e - D.m;
        if (mc && k >= 0 && k < MC_HEADS) mc->hb[k] = v;
    }
};
// the same interface over two plain arrays (row count lists)
struct LinksG {
    LinkF f;
    LinkB b;
    __device__ __forceinline__ int fl(int e) const { return f[e]; }
    __device__ __forceinline__ int bl(int e) const { return b[e]; }
    __device__ __forceinline__ void set_fl(int e, int v) const { f[e] = v; }
    __device__ __forceinline__ void set_bl(int e, int v) const { b[e] = v; }
};
// (re)loads the LDS copies of the list heads: at the start of a launch (every thread) and, by wave 0 alone,
// after anything that wrote links straight to global memory (general pivot paths, remove_col)
__device__ __forceinline__ void mc_reset(const DevGP &D, Mc *mc, int first, int step)
{
    for (int k = first; k < MC_HEADS; k += step) {
        const bool in = k <= D.m + 1;
        mc->hf[k] = in ? D.cflink.hd(k) : 0;
        mc->hb[k] = in ? D.cblink.hd(k) : 0;
    }
    if (first == 0) {
        mc->dirty = 0;
        mc->prevValid = 0;
    }
}

// The element sets of the batched list moves: membership, and the lanes that handle three elements in
// wave_list_move_batch_set (lane q holds elems[q]; lane n the element that is only removed).
struct InHCol {
    const Fast *f;
    int base, gone_slot; // elems = tJ + base; the pivot column sits at slot gone_slot of tJ
    __device__ __forceinline__ bool operator()(int e) const { return hcol_has(f, e); }
    __device__ __forceinline__ int lane_of_slot(int s, int n) const { return s < 0 ? -1 : (s == gone_slot ? n : s - base); }
    __device__ __forceinline__ void lanes_of(int e1, int e2, int e3, bool w1, bool w2, bool w3, int n, int &l1, int &l2, int &l3) const
    {
        int a, b, c;
        hash_find3<HCOL_BITS>(f->hCol, e1, e2, e3, w1, w2, w3, a, b, c);
        l1 = lane_of_slot(a, n);
        l2 = lane_of_slot(b, n);
        l3 = lane_of_slot(c, n);
    }
};
struct InHRow {
    const Fast *f;
    __device__ __forceinline__ bool operator()(int e) const { return hrow_lookup(f, e) > 0; }
    __device__ __forceinline__ void lanes_of(int e1, int e2, int e3, bool w1, bool w2, bool w3, int n, int &l1, int &l2, int &l3) const
    {
        int a, b, c; // position p >= 1 in the cached pivot column; elems = pcI + 1
        hash_find3<HROW_BITS>(f->hRow, e1, e2, e3, w1, w2, w3, a, b, c);
        l1 = a < 0 ? -1 : a - 1;
        l2 = b < 0 ? -1 : b - 1;
        l3 = c < 0 ? -1 : c - 1;
    }
};

// The batched list_move in two halves (n < 64 elements).  WHICH elements leave their lists is known as
// soon as the pivot is chosen -- every column of the pivot row -- while their new lists (keys = new
// counts) are known only after the line updates.  So the unlink half runs BESIDE the line updates, on a
// wave of its own, and only the append half is left for the finalize step:
//   wave_list_unlink_set  links of every element loaded together (one round trip); runs of moved
//                         neighbours resolved by hash look-ups + pointer doubling over the lanes; the first
//                         of each run links its unmoved predecessor to the run's unmoved successor;
//                         `gone` (lane n) links to itself (list.rs:84-85)
//   wave_list_append_set  (after a workgroup barrier: the unlink stores are complete) the tails in memory
//                         ARE the real tails; lanes of equal key meet in the LDS key table; ordered
//                         tail-append.  Returns the smallest key > 0.
// skip = index in elems of an element that is not moved (the pivot column inside a singleton-column
// pivot row, which is `gone`), or -1.
// sfl/sbl (optional): the links of the elements as staged in LDS by the pivot set-up, indexed like elems (the
// element that is only removed: index gone_at)
template <class Links, class InSet>
__device__ __forceinline__ void wave_list_unlink_set(const Links &L, const int *elems, int n, int skip, InSet inS, int gone,
                                                     const int *sfl = nullptr, const int *sbl = nullptr, int gone_at = 0)
{
    const int lane = lane_id();
    const bool mov = lane < n && lane != skip;
    const bool unl = mov || (lane == n && gone >= 0);
    const int e = mov ? elems[lane] : (unl ? gone : 0);
    int p = 0, nx = 0;
    if (unl) {
        if (sfl) {
            const int at = mov ? lane : gone_at;
            p = sbl[at];
            nx = sfl[at];
        } else {
            p = L.bl(e);
            nx = L.fl(e);
        }
    }
    int sl, pl, dummy;
    inS.lanes_of(nx, p, 0, unl, unl, false, n, sl, pl, dummy);
    const bool first = unl && pl < 0; // first of a run of moved neighbours
    int fs = nx;
    for (int round = 0; round < 7; round++) {
        if (!__ballot(sl >= 0)) break;
        const int src = sl >= 0 ? sl : lane;
        const int fs2 = __shfl(fs, src), sl2 = __shfl(sl, src);
        if (sl >= 0) {
            fs = fs2;
            sl = sl2;
        }
    }
    if (first) {
        L.set_fl(p, fs);
        L.set_bl(fs, p);
    }
    if (lane == n && gone >= 0) {
        L.set_fl(gone, gone);
        L.set_bl(gone, gone);
    }
}
// pfl (optional): pfl[q] receives the new forward link of element q (for the next search: Mc::pFl)
template <class Links>
__device__ __forceinline__ int wave_list_append_set(const Links &L, int nelem, const int *elems, const int *keys, int n, int big,
                                                    unsigned long long *kg /* KGMAX words of LDS, all zero between calls */, int *pfl = nullptr)
{
    const int lane = lane_id();
    const int key = lane < n ? keys[lane] : -1;
    const bool act = key >= 0;
    const int e = act ? elems[lane] : 0;
    int t = 0;
    if (act) t = L.bl(nelem + key);
    // while the load is in flight: neighbours inside the new list = nearest lanes below / above with the same key
    unsigned long long mygrp = 0ull;
    if (!__ballot(act && key >= KGMAX)) {
        if (act) atomicOr(&kg[key], 1ull << lane);
        wave_mem_sync();
        if (act) {
            mygrp = kg[key];
            kg[key] = 0ull;
        }
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
    const int eprev = prevl >= 0 ? elems[prevl] : 0, enext = nextl >= 0 ? elems[nextl] : 0;
    const int minall = wave_min_i(act && key > 0 ? key : big);
    if (act) {
        L.set_bl(e, prevl >= 0 ? eprev : t);
        L.set_fl(e, nextl >= 0 ? enext : nelem + key);
        if (pfl) pfl[lane] = nextl >= 0 ? enext : nelem + key;
        if (prevl < 0) L.set_fl(t, e);
        if (nextl < 0) L.set_bl(nelem + key, e);
    }
    return minall;
}

// Batched list_move (see wave_list_move_batch in k_pivot.hip) with the set of moved elements given
// as a membership predicate instead of a mark array.  elems/keys may live in LDS.
// `gone` (>= 0): one more element of the set that is only unlinked, not re-appended (the pivot
// column / pivot row, list.rs:81-86 at the end of every pivot path).
template <class Links, class InSet>
__device__ __forceinline__ int wave_list_move_batch_set(const Links &L, int nelem, const int *elems, const int *keys, int n,
                                        InSet inS, int big, int gone,
                                        unsigned long long *kg /* KGMAX words of LDS, all zero between calls */)
{
    const int lane = lane_id();
    int minkey = big;
    if (n < 64) {
        // Single pass, one memory round trip.  Lane n unlinks `gone`.
        // Single pass: the three loads per element (its two links and the tail of its new list) are
        // independent, so the whole batch costs one memory round trip plus the stores.  The tail read
        // here is the tail BEFORE the unlinks; if that element is itself being moved, the real tail is
        // its nearest unmoved predecessor (links of moved elements are not modified by the unlinks).
        // Unlink stores are issued before append stores; stores of one wave to one address keep order.
        const int key = lane < n ? keys[lane] : -1;
        bool act = key >= 0;
        const bool unl = act || (lane == n && gone >= 0); // elements to unlink
        const int e = act ? elems[lane] : (unl ? gone : 0);
        int p = 0, nx = 0, t = 0;
        if (unl) {
            p = L.bl(e);
            nx = L.fl(e);
        }
        if (act) {
            t = L.bl(nelem + key);
            if (key > 0) minkey = key;
        }
        // While the three loads are in flight: neighbours inside the new list = nearest lanes below /
        // above with the same key.  One ballot per distinct key (a handful), the rest per lane.
        unsigned long long mygrp = 0ull;
        if (!__ballot(act && key >= KGMAX)) {
            // small keys (new counts): the lanes of one key meet in an LDS word indexed by the key;
            // the table is all-zero between calls
            if (act) atomicOr(&kg[key], 1ull << lane);
            wave_mem_sync();
            if (act) {
                mygrp = kg[key];
                kg[key] = 0ull;
            }
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
        const int eprev = prevl >= 0 ? elems[prevl] : 0, enext = nextl >= 0 ? elems[nextl] : 0;
        const int minall = wave_min_i(minkey);
        // Runs of adjacent moved elements are common (lines updated together were appended together),
        // so "next unmoved element" / "previous unmoved element" are found by pointer doubling over the
        // wave's own lanes, never by chasing links through memory.
        PROF_WAIT();
        PROF_STAMP_L0(26);
        // lanes holding my successor / my predecessor / the old tail of my new list: three hash look-ups
        // (a scan over the batch costs ~100 cycles per element in branches and cross-lane reads)
        int sl, pl, tl;
        inS.lanes_of(nx, p, t, unl, unl, act, n, sl, pl, tl);
        const bool first = unl && pl < 0; // first of a run of moved neighbours
        PROF_STAMP_L0(27);
        // first unmoved element after / before every moved one, by pointer doubling over the lanes:
        // (fs, sl) = (element after the stretch skipped so far, its lane if it is moved too)
        int fs = nx, fp = p;
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
            L.set_fl(p, fs);
            L.set_bl(fs, p);
        }
        PROF_STAMP_L0(30);
        // the old tail is being moved itself: the real tail is its nearest unmoved predecessor
        const int tfix = __shfl(fp, tl >= 0 ? tl : lane);
        if (act && tl >= 0) t = tfix;
        if (lane == n && gone >= 0) { // list.rs:84-85: a removed element links to itself
            L.set_fl(gone, gone);
            L.set_bl(gone, gone);
        }
        PROF_STAMP_L0(31);
        if (act) {
            L.set_bl(e, prevl >= 0 ? eprev : t);
            L.set_fl(e, nextl >= 0 ? enext : nelem + key);
            if (prevl < 0) L.set_fl(t, e);
            if (nextl < 0) L.set_bl(nelem + key, e);
        }
        PROF_STAMP_L0(28);
        return minall; // no drain here: the workgroup barrier that follows waits for the stores
    }
    if (gone >= 0) { // long batches: unlink `gone` first, as the general path does
        if (lane == 0) { // list_remove1 through the accessor
            const int f = L.fl(gone), b = L.bl(gone);
            L.set_fl(b, f);
            L.set_bl(f, b);
            L.set_fl(gone, gone);
            L.set_bl(gone, gone);
        }
        wave_mem_sync();
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int q = c0 + lane;
        if (q < n && keys[q] >= 0) {
            const int e = elems[q];
            const int p = L.bl(e);
            int nx = L.fl(e);
            const bool prev_marked = p < nelem && inS(p);
            if (!prev_marked) {
                for (int guard = 0; nx < nelem && inS(nx) && guard <= n; guard++) nx = L.fl(nx);
                L.set_fl(p, nx);
                L.set_bl(nx, p);
            }
        }
    }
    wave_mem_sync();
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int q = c0 + lane;
        int key = q < n ? keys[q] : -1;
        const int e = key >= 0 ? elems[q] : 0;
        bool act = key >= 0;
        if (act && key > 0) minkey = min(minkey, key);
        unsigned long long active = __ballot(act);
        while (active) {
            const int leader = __ffsll((long long)active) - 1;
            const int k = __shfl(key, leader);
            const unsigned long long grp = __ballot(act && key == k);
            const int tail = L.bl(nelem + k);
            const unsigned long long below = grp & lanes_below(lane);
            const unsigned long long above = grp & ~((2ull << lane) - 1ull);
            const int prevl = below ? 63 - __clzll((long long)below) : 0;
            const int nextl = above ? __ffsll((long long)above) - 1 : 0;
            const int pe = __shfl(e, prevl), ne = __shfl(e, nextl);
            if (act && key == k) {
                L.set_bl(e, below ? pe : tail);
                L.set_fl(e, above ? ne : nelem + k);
                if (!below) L.set_fl(tail, e);
                if (!above) L.set_bl(nelem + k, e);
                act = false;
            }
            active &= ~grp;
        }
        wave_mem_sync();
    }
    return wave_min_i(minkey);
}

// ------------------------------------------------------------------------------------------------
// Flattened Markowitz search + pivot set-up, ONE wave, in three parts so that the first two can run
// EARLY -- beside the finalize work of the previous pivot -- on the list state as it was BEFORE that
// pivot's list update:
//   mk_walk   the first K columns in list order (with `skip`: members of the previous pivot's column
//             set are passed over -- they are leaving these lists), their (begin,len,max)
//   mk_stage  all candidate entries + the (begin,len,cap) of their rows -> LDS, cost of every eligible
//             entry, per-lane best
//   mk_pick   lexicographic wave min (cost, flat position) == the reference's sequential strict-<
//             scan (markowitz.rs:80-122); pivot row/column and the metadata of every line they touch ->
//             LDS, hash sets, room checks, fa->kind
// mk_walk returns: 0 candidates found, 1 empty column chosen (pr = -1), 2 error, 3 shape not handled
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mk_walk(const DevGP &D, Sm *sm, Mc *mc)
{
    const int lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int K = D.maxsearch;
    if (K < 1 || K > KCMAX || m >= (1 << 27)) return 3; // (cost*256 + position must fit 64 bits)
    const LinksC LC{D, mc};
    const int h0 = LC.fl(m);
    if (h0 != m) { // empty column: chosen immediately (markowitz.rs:73-78)
        if (lane == 0) {
            sm->pc = h0;
            sm->pr = -1;
        }
        return 1;
    }
    int ncand = 0, total = 0;
    int nz = sm->min_colnz;
    bool bad = false;
    while (ncand < K && nz <= m && !bad) {
        const int k = nz + lane;
        const int h = k <= m ? LC.fl(m + k) : m + k;
        unsigned long long ne = __ballot(k <= m && h != m + k);
        if (ncand == 0) {
            PROF_WAIT();
            PROF_STAMP(19);
        }
        while (ne && ncand < K && !bad) {
            const int b = __ffsll((long long)ne) - 1;
            ne &= ne - 1;
            int j = wave_bcast_i(h, b);
            const int znz = nz + b;
            int guard = 0;
            while (j < m && ncand < K) {
                // a column moved by the previous pivot: new length, begin, maximum and link are still in LDS
                int fl, cb, cl;
                double cmx;
                int ps = -1;
                if (mc && mc->prevValid) {
                    ps = hcol_slot(fa, j);
                    if (ps < mc->prevBase || ps >= MC_PREV || fa->tNew[ps] < 0) ps = -1;
                }
                if (ps >= 0) {
                    fl = mc->pFl[ps];
                    cb = fa->tB[ps];
                    cl = fa->tNew[ps];
                    cmx = fa->tMx[ps];
                } else {
                    fl = D.cflink.el(j);
                    cb = D.cbeg[j];
                    cl = D.clen[j];
                    cmx = D.colmax[j];
                }
                if (ncand == 0) {
                    PROF_WAIT();
                    PROF_STAMP(20);
                }
                if (cl != znz || cmx == 0.0 || !(cmx >= D.abstol) || ++guard > m + 2) {
                    bad = true;
                    break;
                }
                if (lane == 0) {
                    fa->cJ[ncand] = j;
                    fa->cNz[ncand] = znz;
                    fa->cB[ncand] = cb;
                    fa->cL[ncand] = cl;
                    fa->cMx[ncand] = cmx;
                    fa->cOff[ncand] = total;
                }
                total += cl;
                ncand++;
                j = fl;
            }
        }
        nz += 64;
    }
    if (bad || ncand == 0) { // reference: assert / D2 / "no pivot found" assert
        DEV_CHECK(S, false);
        if (lane == 0) {
            sm->pc = -1;
            sm->pr = -1;
        }
        return 2;
    }
    if (lane == 0) {
        fa->cOff[ncand] = total;
        fa->ncand = ncand;
    }
    wave_mem_sync();
    if (total > STGMAX) return 3;
    return 0;
}

__device__ __forceinline__ void mk_stage(const DevGP &D, Sm *sm, Mc *mc, long long &mcb, int &fb)
{
    const int lane = lane_id();
    Fast *fa = &sm->fa;
    const int ncand = fa->ncand, total = fa->cOff[ncand];
    const int off1 = ncand > 1 ? fa->cOff[1] : 0x7fffffff, off2 = ncand > 2 ? fa->cOff[2] : 0x7fffffff,
              off3 = ncand > 3 ? fa->cOff[3] : 0x7fffffff;
    const long long BIG = 0x7fffffffffffffffLL;
    mcb = BIG;
    fb = 0x7fffffff;
    PROF_STAMP(16);
    for (int base = 0; base < total; base += 64) {
        const int f = base + lane;
        if (f < total) {
            const int c = (f >= off1) + (f >= off2) + (f >= off3);
            const int e = f - fa->cOff[c];
            const int pos = fa->cB[c] + e;
            const int idx = D.cidx[pos];
            const double val = D.cval[pos];
            if (base == 0) {
                PROF_WAIT();
                PROF_STAMP(17);
            }
            const int rb = D.rbeg[idx], rl = D.rlen[idx], rc = D.rcap[idx];
            if (base == 0) {
                PROF_WAIT();
                PROF_STAMP(18);
            }
            fa->sI[f] = idx;
            fa->sV[f] = val;
            fa->sB[f] = rb;
            fa->sL[f] = rl;
            fa->sC[f] = rc;
            const double cmx = fa->cMx[c];
            const double tol = fmax(D.abstol, D.reltol * cmx);
            const double x = fabs(val);
            if (!(x == 0.0 || x < tol)) {
                const long long mc = (long long)(fa->cNz[c] - 1) * (long long)(rl - 1);
                if (mc < mcb) { // f grows with base: strict < keeps the earliest position per lane
                    mcb = mc;
                    fb = f;
                }
            }
        }
    }
}

// Column singletons.  While the count-1 list is not empty its first column is the pivot column: its one
// entry costs (1-1)*(r-1) = 0 and no later candidate can be strictly cheaper (markowitz.rs:105); the
// reference still looks at maxsearch columns (or all that are left), which only shows in nsearch_pivot.
// So for these -- half of all pivots of an LP basis -- the other candidates are neither walked nor
// staged: 4 dependent loads (list heads, column, entry, row) instead of 6, and no reduction.
// Returns false (nothing modified) if the count-1 list is empty or anything is unusual; the caller
// then takes the ordinary route, which also raises the errors.
__device__ __forceinline__ bool mk_express(const DevGP &D, Sm *sm, Mc *mc, int &nsearched)
{
    const int lane = lane_id();
    const int m = D.m;
    Fast *fa = &sm->fa;
    const int K = D.maxsearch;
    if (K < 1 || K > KCMAX || m >= (1 << 27)) return false;
    const LinksC LC{D, mc};
    const int h0 = LC.fl(m), j = LC.fl(m + 1);
    if (h0 != m || j >= m) return false;
    int cb, cl, idx = -1;
    double cmx, val = 0.0;
    int ps = -1;
    if (mc && mc->prevValid) {
        ps = hcol_slot(fa, j);
        if (ps < mc->prevBase || ps >= MC_PREV || fa->tNew[ps] < 0) ps = -1;
    }
    if (ps >= 0) { // moved by the previous pivot: everything about it is still in LDS
        cb = fa->tB[ps];
        cl = fa->tNew[ps];
        cmx = fa->tMx[ps];
        idx = mc->e1i[ps];
        val = mc->e1v[ps];
    } else {
        cb = D.cbeg[j];
        cl = D.clen[j];
        cmx = D.colmax[j];
    }
    if (cl != 1 || cmx == 0.0 || !(cmx >= D.abstol)) return false;
    int rb, rl, rc;
    if (idx >= 0 && mc->e1rl[ps] >= 0) { // (left with the entry by the singleton-column pivot that made it a singleton)
        rb = mc->e1rb[ps];
        rl = mc->e1rl[ps];
        rc = mc->e1rc[ps];
    } else {
        if (idx < 0) {
            idx = D.cidx[cb];
            val = D.cval[cb];
        }
        rb = D.rbeg[idx];
        rl = D.rlen[idx];
        rc = D.rcap[idx];
    }
    const double tol = fmax(D.abstol, D.reltol * cmx);
    const double x = fabs(val);
    if (x == 0.0 || x < tol) return false;
    const int left = m - sm->rank - sm->rankdef; // every active column is in a count list, list 0 is empty
    nsearched = left < K ? left : K;
    if (lane == 0) {
        fa->ncand = 1;
        fa->cJ[0] = j;
        fa->cNz[0] = 1;
        fa->cB[0] = cb;
        fa->cL[0] = 1;
        fa->cMx[0] = cmx;
        fa->cOff[0] = 0;
        fa->cOff[1] = 1;
        fa->sI[0] = idx;
        fa->sV[0] = val;
        fa->sB[0] = rb;
        fa->sL[0] = rl;
        fa->sC[0] = rc;
    }
    wave_mem_sync();
    return true;
}

// single: one candidate entry (a column singleton): nothing to reduce
// key_given >= 0: the reduction was done already (spec_finish)
__device__ __forceinline__ void mk_pick(const DevGP &D, Sm *sm, Mc *mc, long long mcb, int fb, int nsearched, bool single, long long key_given = -1)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int ncand = fa->ncand;
    const int off1 = ncand > 1 ? fa->cOff[1] : 0x7fffffff, off2 = ncand > 2 ? fa->cOff[2] : 0x7fffffff,
              off3 = ncand > 3 ? fa->cOff[3] : 0x7fffffff;
    const long long BIG = 0x7fffffffffffffffLL;
    if (lane == 0) fa->kind = 0;
    // key = cost * 256 + position (position < STGMAX <= 256, cost < 2^55); first-seen entry wins ties
    const long long bestkey = key_given >= 0 ? key_given : (single ? 0LL : wave_min_ll(mcb != BIG ? mcb * 256LL + (long long)fb : BIG));
    wave_mem_sync();
    if (bestkey == BIG) { // no eligible entry: cannot happen when colmax is the column maximum
        DEV_CHECK(S, false);
        if (lane == 0) {
            sm->pc = -1;
            sm->pr = -1;
        }
        return;
    }
    const int fsel = (int)(bestkey & 255LL);
    const int csel = (fsel >= off1) + (fsel >= off2) + (fsel >= off3);
    const int pc = fa->cJ[csel], pr = fa->sI[fsel];
    const int nzc = fa->cL[csel], pcb = fa->cB[csel], where = fsel - fa->cOff[csel];
    const int nzr = fa->sL[fsel], prb = fa->sB[fsel];
    if (lane == 0) {
        sm->pr = pr;
        sm->pc = pc;
        sm->pcb = pcb;
        sm->prb = prb;
        sm->nzc = nzc;
        sm->nzr = nzr;
        sm->nsearch += nsearched;
        sm->min_colnz = fa->cNz[0];
        sm->flag_small = 0;
        sm->ncancel = 0;
        fa->anycancel = 0;
        // room in L and U (pivot.rs:70-81)
        if (sm->lused + (nzc - 1) > D.lcap) {
            sm->exit_code = ST_NEED_L;
            sm->need = nzc - 1;
        } else if (sm->uused + (nzr - 1) > D.ucap) {
            sm->exit_code = ST_NEED_U;
            sm->need = nzr - 1;
        }
    }
    DEV_CHECK(S, nzr >= 1 && nzc >= 1);
    PROF_STAMP(11); // pivot chosen
    int kind = 0;
    if (nzr >= 2 && nzr <= PRMAX) {
        if (nzc == 1) kind = 2;
        else if (nzc >= 3 && nzc <= PCMAX) kind = 1;
    }
    if (kind == 0) return;

    if (nzr <= 64 && nzc <= 64) {
        // ---- the common shape in straight-line form: one chunk of the pivot row, one of the pivot column, the first
        // hash probe without a loop.  (The searching wave is alone on the critical path: its time is its instruction
        // count.)  Same steps, same order of the loads as below.
        const int jq0 = lane < nzr ? D.ridx[prb + lane] : -1;
        if (kind == 1) {
            for (int s = lane; s < HROW; s += 64) fa->hRow[s] = ~0ull;
        }
        for (int s = lane; s < HCOL; s += 64) fa->hCol[s] = ~0ull;
        const int coff = fa->cOff[csel];
        int hr_idx0 = 0, hr_slot0 = 0, hr_len0 = 0;
        if (lane < nzc) {
            const int slot = (lane == where) ? 0 : (lane == 0 ? where : lane);
            const int idx = fa->sI[coff + lane], rlv = fa->sL[coff + lane];
            fa->pcI[slot] = idx;
            fa->pcV[slot] = fa->sV[coff + lane];
            fa->prB[slot] = fa->sB[coff + lane];
            fa->prL[slot] = rlv;
            fa->prC[slot] = fa->sC[coff + lane];
            hr_slot0 = slot;
            hr_idx0 = idx;
            hr_len0 = rlv;
        }
        const unsigned long long hb = __ballot(jq0 == pc);
        if (!hb) {
            DEV_CHECK(S, false);
            if (lane == 0) sm->pc = -1;
            return;
        }
        const int wpos = __ffsll((long long)hb) - 1;
        PROF_STAMP(12);
        int tb = 0, tl = 0, tc = 0, tfl = 0, tbl = 0;
        if (lane < nzr) {
            tb = D.cbeg[jq0];
            tl = D.clen[jq0];
            tc = D.ccap[jq0];
#if !BLU_CFG_BATCH
            tfl = D.cflink.el(jq0); // for the unlink wave: same round trip, one less on its own chain
            tbl = D.cblink.el(jq0);
#endif
        }
        int gc = 0, gr = 0;
        if (kind == 1 && hr_slot0 >= 1) {
            hrow_insert(fa, hr_idx0, hr_slot0);
            const int n = hr_len0 + nzr - 1;
            gr = n + stretch_of(D.stretch, n) + D.pad;
        }
        if (lane < nzr) {
            const int slot = kind == 1 ? ((lane == wpos) ? 0 : (lane == 0 ? wpos : lane)) : lane;
            fa->tJ[slot] = jq0;
            fa->tB[slot] = tb;
            fa->tL[slot] = tl;
            fa->tC[slot] = tc;
#if !BLU_CFG_BATCH // (the batch kernel has no unlink wave and no speculative search)
            fa->tFl[slot] = tfl;
            fa->tBl[slot] = tbl;
#endif
            hcol_insert(fa, jq0, slot);
            if (kind == 1 && lane != wpos) {
                const int n = tl + nzc - 1;
                gc = n + stretch_of(D.stretch, n) + D.pad;
            }
        }
        PROF_STAMP(13);
        if (kind == 1) {
            // (each estimate is below 2^31: at most 64 lines of < 2^22 entries here)
            const long long both = wave_sum_ll(((long long)gc << 32) | (long long)(unsigned)gr);
            if ((long long)sm->cused + (both >> 32) > (long long)D.carena_cap || (long long)sm->rused + (both & 0xffffffffLL) > (long long)D.rarena_cap)
                kind = 0; // the general path makes the exact check and leaves with NEED_CW / NEED_RW
        }
        if (lane == 0) {
            fa->kind = kind;
            fa->where = wpos;
            fa->tLnk = BLU_CFG_BATCH ? 0 : 1;
        }
        PROF_STAMP(14);
        return;
    }

    // Order of the steps below: every global load is issued as early as its address is known and the LDS work
    // that does not depend on it runs while it is in flight (pivot row || pivot column copy; column metadata ||
    // row hash).
    // ---- pivot row: loads issued first; kind 1: pivot column swapped to the front later (pivot.rs:185)
    int jq[PRMAX / 64];
#pragma unroll
    for (int c = 0; c < PRMAX / 64; c++) {
        jq[c] = -1;
        if (c * 64 < nzr) { // (rows are mostly shorter than 64: one chunk)
            const int q = c * 64 + lane;
            if (q < nzr) jq[c] = D.ridx[prb + q];
        }
    }
    if (kind == 1) {
        for (int s = lane; s < HROW; s += 64) fa->hRow[s] = ~0ull;
    }
    for (int s = lane; s < HCOL; s += 64) fa->hCol[s] = ~0ull;
    // ---- pivot column into LDS; kind 1: pivot swapped to the front (pivot.rs:169-170)
    const int coff = fa->cOff[csel];
    int hr_idx[(PCMAX + 63) / 64], hr_slot[(PCMAX + 63) / 64], hr_len[(PCMAX + 63) / 64];
#pragma unroll
    for (int c = 0; c < (PCMAX + 63) / 64; c++) {
        const int e = c * 64 + lane;
        hr_slot[c] = 0;
        hr_idx[c] = 0;
        hr_len[c] = 0;
        if (e < nzc) {
            const int slot = (e == where) ? 0 : (e == 0 ? where : e);
            const int idx = fa->sI[coff + e], rlv = fa->sL[coff + e];
            fa->pcI[slot] = idx;
            fa->pcV[slot] = fa->sV[coff + e];
            fa->prB[slot] = fa->sB[coff + e];
            fa->prL[slot] = rlv;
            fa->prC[slot] = fa->sC[coff + e];
            hr_slot[c] = slot;
            hr_idx[c] = idx;
            hr_len[c] = rlv;
        }
    }
    int wpos = -1;
#pragma unroll
    for (int c = 0; c < PRMAX / 64; c++) {
        if (c * 64 < nzr) {
            const unsigned long long hb = __ballot(jq[c] == pc);
            if (hb) wpos = c * 64 + __ffsll((long long)hb) - 1;
        }
    }
    if (wpos < 0) {
        DEV_CHECK(S, false);
        if (lane == 0) sm->pc = -1;
        return;
    }
    PROF_STAMP(12); // pivot column copied, pivot row loaded
    // ---- (begin,len,cap) of the pivot row's columns: loads issued, the row hash is built meanwhile
    int tbq[PRMAX / 64], tlq[PRMAX / 64], tcq[PRMAX / 64];
#pragma unroll
    for (int c = 0; c < PRMAX / 64; c++) {
        const int q = c * 64 + lane;
        tbq[c] = tlq[c] = tcq[c] = 0;
        if (c * 64 < nzr && q < nzr) {
            const int j = jq[c];
            tbq[c] = D.cbeg[j];
            tlq[c] = D.clen[j];
            tcq[c] = D.ccap[j];
        }
    }
    long long gc = 0, gr = 0;
    if (kind == 1) {
#pragma unroll
        for (int c = 0; c < (PCMAX + 63) / 64; c++) {
            if (hr_slot[c] >= 1) {
                hrow_insert(fa, hr_idx[c], hr_slot[c]);
                const int n = hr_len[c] + nzr - 1;
                gr += n + stretch_of(D.stretch, n) + D.pad;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < PRMAX / 64; c++) {
        const int q = c * 64 + lane;
        if (c * 64 < nzr && q < nzr) {
            const int j = jq[c];
            const int slot = kind == 1 ? ((q == wpos) ? 0 : (q == 0 ? wpos : q)) : q;
            const int tb = tbq[c], tl = tlq[c], tc = tcq[c];
            fa->tJ[slot] = j;
            fa->tB[slot] = tb;
            fa->tL[slot] = tl;
            fa->tC[slot] = tc;
            hcol_insert(fa, j, slot);
            if (kind == 1 && q != wpos) {
                const int n = tl + nzc - 1;
                gc += n + stretch_of(D.stretch, n) + D.pad;
            }
        }
    }
    PROF_STAMP(13); // line metadata of the pivot row's columns loaded, both hashes built
    if (kind == 1) {
        // one reduction for both room estimates (each below 2^31: <= 256 lines of < 2^22 entries)
        const long long both = wave_sum_ll((gc << 32) | (gr & 0xffffffffLL));
        gc = both >> 32;
        gr = both & 0xffffffffLL;
        if ((long long)sm->cused + gc > (long long)D.carena_cap || (long long)sm->rused + gr > (long long)D.rarena_cap)
            kind = 0; // the general path makes the exact check and leaves with NEED_CW / NEED_RW
    }
    if (lane == 0) {
        fa->kind = kind;
        fa->where = wpos;
        fa->tLnk = 0;
    }
    PROF_STAMP(14);
}

// the complete search on the current list state.  Returns false if the shape is outside what this path
// handles (nothing has been modified then; the caller runs the general search).
template <bool BATCH>
__device__ __forceinline__ bool markowitz_fast(const DevGP &D, Sm *sm, Mc *mc, long long ew_mcb, int ew_fb)
{
    Fast *fa = &sm->fa;
    if (lane_id() == 0) fa->kind = 0;
    PROF_STAMP(8);
    int nsr = 0;
    if (BLU_EARLY && !BATCH && fa->ewValid) { // found and staged while the previous pivot was being finished (early_search)
        const int lane = lane_id();
        const bool whole = fa->ewValid == 2; // a whole search (spec_walk + spec_finish): the winner's key is in spKey
        const long long key = whole ? fa->spKey : -1;
        const long long mcb = ew_mcb;
        const int fb = ew_fb;
        nsr = fa->ewNsr;
        wave_mem_sync();
        if (lane == 0) fa->ewValid = 0;
#ifdef BLU_EWCHECK
        { // self-checking build: the ordinary search must give the same candidates, costs and count
            const int sN = fa->ncand;
            const int myc = lane < sN ? fa->cJ[lane] : -1;
            // (whole search: every staged entry with its row's begin, length and capacity as well)
            const int sT = whole ? fa->cOff[sN] : 1;
            int kI[2], kB[2], kL[2], kC[2];
            double kV[2];
            for (int u = 0; u < 2; u++) {
                const int f = lane + 64 * u;
                kI[u] = f < sT ? fa->sI[f] : 0;
                kB[u] = f < sT ? fa->sB[f] : 0;
                kL[u] = f < sT ? fa->sL[f] : 0;
                kC[u] = f < sT ? fa->sC[f] : 0;
                kV[u] = f < sT ? fa->sV[f] : 0.0;
            }
            wave_mem_sync();
            int nsr2 = 0, fb2 = 0;
            long long mcb2 = 0;
            bool okc = true;
            if (whole || !mk_express(D, sm, mc, nsr2)) {
                const int r2 = mk_walk(D, sm, mc);
                if (r2 == 0) {
                    mk_stage(D, sm, mc, mcb2, fb2);
                    nsr2 = fa->ncand;
                } else {
                    okc = false;
                }
            }
            if (whole) {
                const long long BIGK = 0x7fffffffffffffffLL;
                const long long key2 = wave_min_ll(mcb2 != BIGK ? mcb2 * 256LL + (long long)fb2 : BIGK);
                okc = okc && key2 == key && fa->cOff[fa->ncand] == sT;
                mcb2 = mcb;
                fb2 = fb;
            }
            for (int u = 0; u < 2; u++) { // (a singleton found early: its one entry; if the ordinary search walked, that is entry 0 too)
                const int f = lane + 64 * u;
                if (okc && f < sT) okc = fa->sI[f] == kI[u] && fa->sB[f] == kB[u] && fa->sL[f] == kL[u] && fa->sC[f] == kC[u] && fa->sV[f] == kV[u];
            }
            okc = okc && fa->ncand == sN && nsr2 == nsr && mcb2 == mcb && fb2 == fb && (lane >= sN || fa->cJ[lane] == myc);
            if (__ballot(!okc)) {
                if (lane == 0 && D.s->prof[0] == 0) {
                    long long *P = D.s->prof;
                    P[0] = 1 + sm->rank;
                    P[1] = sN * 10 + fa->ncand;
                    P[2] = nsr * 10 + nsr2;
                    for (int i = 0; i < 3; i++) P[7 + i] = fa->cJ[i] * 1000LL + fa->cNz[i];
                }
                if (lane < 3) D.s->prof[10 + lane] = myc;
                DEV_CHECK(D.s, false);
            }
        }
#endif
        PROF_STAMP(9);
        PROF_STAMP(10);
        mk_pick(D, sm, mc, mcb, fb, nsr, true, key);
        return true;
    }
    if (mk_express(D, sm, mc, nsr)) {
        PROF_STAMP(9);
        PROF_STAMP(10);
        mk_pick(D, sm, mc, 0, 0, nsr, true);
        return true;
    }
    const int r = mk_walk(D, sm, mc);
    PROF_STAMP(9); // candidates walked (list heads + up to K link/meta loads)
    if (r == 3) return false;
    if (r != 0) return true; // empty column chosen, or error raised
    long long mcb;
    int fb;
    mk_stage(D, sm, mc, mcb, fb);
    PROF_STAMP(10); // candidate entries + their row metadata loaded and costed
    mk_pick(D, sm, mc, mcb, fb, fa->ncand, false);
    return true;
}

// ------------------------------------------------------------------------------------------------
// kind 1: pivot_small (pivot.rs:460-833), column q of the pivot row, ONE wave
// ------------------------------------------------------------------------------------------------
// idx_first/val_first: this lane's entry of the first 64-entry chunk, loaded by the caller ahead of
// time (the caller issues the loads of all its tasks before processing any of them)
__device__ __forceinline__ void fast_col(const DevGP &D, Sm *sm, Mc *mc, int q, double *work, int idx_first, double val_first, int pr, int cnz1,
                                         double pivot)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int j = fa->tJ[q], cb = fa->tB[q], cl = fa->tL[q], cap = fa->tC[q];

    int nkept = 0, where = -1, first_idx = 0;
    double xrj = 0.0, first_val = 0.0, cmxl = 0.0;
    // registers of the first chunk are reused in pass 2 when the column fits one chunk
    int idx0 = 0, t0r = 0;
    double val0 = 0.0;
    bool keep0 = false;
    for (int c = 0; c < cl; c += 64) {
        const int e = c + lane;
        const bool v = e < cl;
        const int idx = c == 0 ? idx_first : (v ? D.cidx[cb + e] : 0);
        const double val = c == 0 ? val_first : (v ? D.cval[cb + e] : 0.0);
        const int mk = v ? hrow_lookup(fa, idx) : 0;
        const bool keep = v && mk == 0;
        if (v && mk > 0) work[mk - 1] = val;
        const unsigned long long kb = __ballot(keep);
        const int t = nkept + wave_prefix_count(kb);
        const bool ispr = keep && idx == pr;
        const unsigned long long pb = __ballot(ispr);
        if (pb) {
            const int src = __ffsll((long long)pb) - 1;
            where = wave_bcast_i(t, src);
            xrj = wave_bcast_d(val, src);
        }
        const unsigned long long fb = __ballot(keep && t == 0);
        if (fb) {
            const int src = __ffsll((long long)fb) - 1;
            first_idx = wave_bcast_i(idx, src);
            first_val = wave_bcast_d(val, src);
        }
        if (keep && !ispr) {
            const double x = fabs(val);
            if (x > cmxl) cmxl = x;
        }
        if (c == 0) {
            idx0 = idx;
            val0 = val;
            keep0 = keep;
            t0r = t;
        }
        nkept += __popcll(kb);
    }
    DEV_CHECK(S, where >= 0);
    const int nk1 = nkept - 1;
    const int need_max = nk1 + cnz1;
    const bool reloc = need_max > cap;
    int dst = cb, newcap = cap;
    if (reloc) {
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->cused, newcap);
        dst = __builtin_amdgcn_readfirstlane(nb);
    }
    if (cl <= 64) {
        if (keep0 && t0r != where && t0r > 0) {
            D.cidx[dst + t0r - 1] = idx0;
            D.cval[dst + t0r - 1] = val0;
        }
    } else {
        int nk = 0;
        for (int c = 0; c < cl; c += 64) {
            const int e = c + lane;
            const bool v = e < cl;
            const int idx = v ? D.cidx[cb + e] : 0;
            const double val = v ? D.cval[cb + e] : 0.0;
            const bool keep = v && hrow_lookup(fa, idx) == 0;
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
    const double a = xrj / pivot;
    const int put = dst + nk1;
    const bool p = lane < cnz1;
    double x = 0.0;
    int ri = 0;
    if (p) {
        x = mulsub(work[lane], a, fa->pcV[1 + lane]);
        ri = fa->pcI[1 + lane];
        work[lane] = 0.0;
    }
    const double ax = fabs(x);
    const bool kx = p && ax > D.droptol;
    const unsigned long long kb = __ballot(kx);
    if (kx) {
        const int d = wave_prefix_count(kb);
        D.cidx[put + d] = ri;
        D.cval[put + d] = x;
        if (ax > cmxl) cmxl = ax;
    }
    const unsigned long long mask = __ballot(p && !kx);
    const int nadd = __popcll(kb);
    // column maximum: LDS atomic max on the bit patterns (non-negative doubles order like integers);
    // one LDS round trip instead of a 6-step cross-lane reduction
    const double cmx = wave_max_d(cmxl);
    if (lane == 0) {
        const int newlen = nk1 + nadd;
        D.cbeg[j] = dst;
        D.clen[j] = newlen;
        D.ccap[j] = newcap;
        D.colmax[j] = cmx;
        fa->tNew[q] = newlen;
        if (q < 64) { // new begin and maximum: the (early) search of the next pivot reads them
            fa->tB[q] = dst;
            fa->tMx[q] = cmx;
            if (mc) mc->e1i[q] = -1;
        }
        fa->tX[q] = xrj;
        fa->tM[q] = mask;
        if (mask) fa->anycancel = 1;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
        if (mask >> 31) atomicAdd((unsigned long long *)&sm->d3, (unsigned long long)__popcll(mask >> 31));
        if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
    }
}

// fast_col for a column of at most 64 entries (all but a handful): one chunk, no loops, the look-up of the common
// case (first or second slot of the probe sequence) in straight-line code.  The line-update phase of a small pivot is
// bound by the instruction throughput of the CU, so every instruction here is paid 20 times per
// pivot.
__device__ __forceinline__ int hrow_lookup2(const Fast *f, int k)
{
    const unsigned s0 = hslot(k, HROW_BITS), s1 = (s0 + 1) & (HROW - 1);
    const unsigned long long x0 = f->hRow[s0], x1 = f->hRow[s1];
    const bool h0 = (int)(x0 >> 32) == k, e0 = x0 == ~0ull, h1 = (int)(x1 >> 32) == k, e1 = x1 == ~0ull;
    int r = h0 ? (int)(x0 & 0xffffffffull) : ((!e0 && h1) ? (int)(x1 & 0xffffffffull) : 0);
    if (!(h0 || e0 || h1 || e1)) r = hrow_lookup(f, k); // a third probe: rare (tables at most a quarter full)
    return r;
}
__device__ __forceinline__ void fast_col_short(const DevGP &D, Sm *sm, Mc *mc, int q, double *work, int idx, double val, int pr, int cnz1,
                                               double pivot)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int j = fa->tJ[q], cb = fa->tB[q], cl = fa->tL[q], cap = fa->tC[q];
    const bool v = lane < cl;
    const int mk = v ? hrow_lookup2(fa, idx) : 0;
    const bool keep = v && mk == 0;
    if (v && mk > 0) work[mk - 1] = val;
    const unsigned long long kb = __ballot(keep);
    const int t = wave_prefix_count(kb);
    const unsigned long long pb = __ballot(keep && idx == pr);
    DEV_CHECK(S, pb != 0ull);
    const int psrc = pb ? __ffsll((long long)pb) - 1 : 0;
    const int where = wave_bcast_i(t, psrc);
    const double xrj = wave_bcast_d(val, psrc);
    const int fsrc = kb ? __ffsll((long long)kb) - 1 : 0; // the first kept entry (t == 0)
    const int first_idx = wave_bcast_i(idx, fsrc);
    const double first_val = wave_bcast_d(val, fsrc);
    double cmxl = (keep && lane != psrc) ? fabs(val) : 0.0;
    const int nk1 = __popcll(kb) - 1;
    const int need_max = nk1 + cnz1;
    const bool reloc = need_max > cap;
    int dst = cb, newcap = cap;
    if (reloc) {
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->cused, newcap);
        dst = __builtin_amdgcn_readfirstlane(nb);
    }
    if (keep && t != where && t > 0) {
        D.cidx[dst + t - 1] = idx;
        D.cval[dst + t - 1] = val;
    }
    if (where > 0 && lane == 0) {
        D.cidx[dst + where - 1] = first_idx;
        D.cval[dst + where - 1] = first_val;
    }
    const double a = xrj / pivot;
    const int put = dst + nk1;
    const bool p = lane < cnz1;
    double x = 0.0;
    int ri = 0;
    if (p) {
        x = mulsub(work[lane], a, fa->pcV[1 + lane]);
        ri = fa->pcI[1 + lane];
        work[lane] = 0.0;
    }
    const double ax = fabs(x);
    const bool kx = p && ax > D.droptol;
    const unsigned long long kxb = __ballot(kx);
    if (kx) {
        const int d = wave_prefix_count(kxb);
        D.cidx[put + d] = ri;
        D.cval[put + d] = x;
        if (ax > cmxl) cmxl = ax;
    }
    const unsigned long long mask = __ballot(p && !kx);
    // column maximum: one LDS atomic per lane with a candidate (non-negative doubles order like their bit patterns)
    // instead of a 64-bit cross-lane reduction (~30 vector instructions); only lane 0 needs the result
    unsigned long long *wm = &sm->wmax[wave_id()];
    if (cmxl > 0.0) atomicMax(wm, (unsigned long long)__double_as_longlong(cmxl));
    wave_mem_sync();
    if (lane == 0) {
        const double cmx = __longlong_as_double((long long)*wm);
        *wm = 0ull;
        const int newlen = nk1 + __popcll(kxb);
        D.cbeg[j] = dst;
        D.clen[j] = newlen;
        D.ccap[j] = newcap;
        D.colmax[j] = cmx;
        fa->tNew[q] = newlen;
        if (q < 64) {
            fa->tB[q] = dst;
            fa->tMx[q] = cmx;
            if (mc) mc->e1i[q] = -1;
        }
        fa->tX[q] = xrj;
        fa->tM[q] = mask;
        if (mask) fa->anycancel = 1;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
        if (mask >> 31) atomicAdd((unsigned long long *)&sm->d3, (unsigned long long)__popcll(mask >> 31));
        if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
    }
}

// kind 1: row p of the pivot column, ONE wave.  Appends the whole pivot-row pattern; positions
// cancelled by fast_col are removed afterwards by fast_fixrow.
__device__ __forceinline__ void fast_row(const DevGP &D, Sm *sm, Mc *mc, int p, int j_first, int pc, int rnz1)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int i = fa->pcI[p], rb = fa->prB[p], rl = fa->prL[p], cap = fa->prC[p];

    int nk = 0;
    bool found = false;
    int j0 = -1, t0r = 0;
    bool keep0 = false;
    for (int c = 0; c < rl; c += 64) {
        const int e = c + lane;
        const bool v = e < rl;
        const int j = c == 0 ? j_first : (v ? D.ridx[rb + e] : -1);
        const bool keep = v && !hcol_has(fa, j);
        if (__ballot(v && j == pc)) found = true;
        const unsigned long long kb = __ballot(keep);
        if (c == 0) {
            j0 = j;
            keep0 = keep;
            t0r = wave_prefix_count(kb);
        }
        nk += __popcll(kb);
    }
    DEV_CHECK(S, found);
    const int need_max = nk + rnz1;
    const bool reloc = need_max > cap;
    int dst = rb, newcap = cap;
    if (reloc) {
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->rused, newcap);
        dst = __builtin_amdgcn_readfirstlane(nb);
    }
    if (rl <= 64) {
        if (keep0) D.ridx[dst + t0r] = j0;
    } else {
        int t0 = 0;
        for (int c = 0; c < rl; c += 64) {
            const int e = c + lane;
            const bool v = e < rl;
            const int j = v ? D.ridx[rb + e] : -1;
            const bool keep = v && !hcol_has(fa, j);
            const unsigned long long kb = __ballot(keep);
            if (keep) D.ridx[dst + t0 + wave_prefix_count(kb)] = j;
            t0 += __popcll(kb);
        }
    }
    for (int q = 1 + lane; q <= rnz1; q += 64) D.ridx[dst + nk + q - 1] = fa->tJ[q];
    if (lane == 0) {
        D.rbeg[i] = dst;
        D.rlen[i] = nk + rnz1;
        D.rcap[i] = newcap;
        fa->rNew[p] = nk + rnz1;
        fa->rKept[p] = nk;
        fa->rDst[p] = dst;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
    }
}

// fast_row for a row of at most 64 entries: one chunk, straight-line
__device__ __forceinline__ bool hcol_has2(const Fast *f, int k)
{
    const unsigned s0 = hslot(k, HCOL_BITS), s1 = (s0 + 1) & (HCOL - 1);
    const unsigned long long x0 = f->hCol[s0], x1 = f->hCol[s1];
    const bool h0 = (int)(x0 >> 32) == k, e0 = x0 == ~0ull, h1 = (int)(x1 >> 32) == k, e1 = x1 == ~0ull;
    bool r = h0 || (!e0 && h1);
    if (!(h0 || e0 || h1 || e1)) r = hcol_has(f, k);
    return r;
}
__device__ __forceinline__ void fast_row_short(const DevGP &D, Sm *sm, Mc *mc, int p, int j, int pc, int rnz1)
{
    const int lane = lane_id();
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    const int i = fa->pcI[p], rb = fa->prB[p], rl = fa->prL[p], cap = fa->prC[p];
    const bool v = lane < rl;
    const bool keep = v && !hcol_has2(fa, j);
    DEV_CHECK(S, __ballot(v && j == pc) != 0ull);
    const unsigned long long kb = __ballot(keep);
    const int t0 = wave_prefix_count(kb);
    const int nk = __popcll(kb);
    const int need_max = nk + rnz1;
    const bool reloc = need_max > cap;
    int dst = rb, newcap = cap;
    if (reloc) {
        newcap = need_max + stretch_of(D.stretch, need_max) + D.pad;
        int nb = 0;
        if (lane == 0) nb = atomicAdd(&sm->rused, newcap);
        dst = __builtin_amdgcn_readfirstlane(nb);
    }
    if (keep) D.ridx[dst + t0] = j;
    for (int q = 1 + lane; q <= rnz1; q += 64) D.ridx[dst + nk + q - 1] = fa->tJ[q];
    if (lane == 0) {
        D.rbeg[i] = dst;
        D.rlen[i] = nk + rnz1;
        D.rcap[i] = newcap;
        fa->rNew[p] = nk + rnz1;
        fa->rKept[p] = nk;
        fa->rDst[p] = dst;
        if (reloc) atomicAdd((unsigned long long *)&sm->nexpand, 1ull);
    }
}

// (Round 2 built "two lines per wave": pair forms of fast_col / fast_row running one instruction stream on two
// lines, lanes 0-31 and 32-63, ballots split into halves, half-wide broadcasts by two v_readlane and a select,
// rows of up to 64 entries in two register chunks -- 18 pair tasks per small pivot instead of 35 line tasks.
// Bit-identical (87 parity tests), and no faster: a pair costs 5 500 cycles against 3 500 for one line (the
// half-wide selects, twice the LDS set-up per task, more spilled registers) and the task set-up doubles, so a
// wave's share takes as long as before: C3 pivot loop 820 -> 842 ms, batch kernel 1.21 -> 1.18 s.  Not kept.
// Tried again after the straight-line forms below, for the rows only (the cheap kind: pattern, no values): 745 -> 764
// ms, batch kernel 1.03 -> 1.10 s -- one row in eight has more than 32 entries and its pair falls back to two
// single updates with loads that were not issued ahead, and the per-lane selects of the pair set-up cost what the
// shared instruction stream saves.  Not kept.)
// rewrite the appended part of row p without the cancelled positions (pivot.rs:752-758)
__device__ __forceinline__ void fast_fixrow(const DevGP &D, Sm *sm, Mc *mc, int p)
{
    const int lane = lane_id();
    Fast *fa = &sm->fa;
    const int rnz1 = sm->nzr - 1;
    const int dst = fa->rDst[p], nk = fa->rKept[p];
    int na = 0;
    for (int c = 0; c < rnz1; c += 64) {
        const int q = 1 + c + lane;
        const bool ok = q <= rnz1 && ((fa->tM[q] >> (p - 1)) & 1ull) == 0;
        const unsigned long long kb = __ballot(ok);
        if (ok) D.ridx[dst + nk + na + wave_prefix_count(kb)] = fa->tJ[q];
        na += __popcll(kb);
    }
    if (lane == 0) {
        D.rlen[fa->pcI[p]] = nk + na;
        fa->rNew[p] = nk + na;
    }
}

// smallest key >= 0 (new count of a moved column), for the queue's validity test
__device__ __forceinline__ int wave_min_key(const int *keys, int n, int big)
{
    int v = big;
    for (int c = lane_id(); c < n; c += 64) {
        const int k = keys[c];
        if (k >= 0 && k < v) v = k;
    }
    return wave_min_i(v);
}

// U row from the LDS copies (pivot.rs:306-312): slots q0..q1 of the pivot row except skipq
__device__ __forceinline__ void fast_write_u(const DevGP &D, Sm *sm, int q0, int q1, int skipq)
{
    const int lane = lane_id();
    Fast *fa = &sm->fa;
    int put = sm->uused;
    for (int c = q0; c <= q1; c += 64) {
        const int q = c + lane;
        const bool v = q <= q1 && q != skipq;
        const double x = v ? fa->tX[q] : 0.0;
        const bool k = v && fabs(x) > D.droptol;
        const unsigned long long kb = __ballot(k);
        if (k) {
            const int d = put + wave_prefix_count(kb);
            D.uidx[d] = fa->tJ[q];
            D.uval[d] = x;
        }
        put += __popcll(kb);
    }
    if (lane == 0) {
        D.ubeg[sm->rank + 1] = put;
        sm->uused = put;
    }
}

// L column from the LDS copy (pivot.rs:404-416): slots 1..cnz1
__device__ __forceinline__ void fast_write_l(const DevGP &D, Sm *sm)
{
    const int lane = lane_id();
    Fast *fa = &sm->fa;
    const int cnz1 = sm->nzc - 1;
    const double pivot = fa->pcV[0];
    int put = sm->lused;
    for (int c = 1; c <= cnz1; c += 64) {
        const int p = c + lane;
        const bool v = p <= cnz1;
        const double x = v ? fa->pcV[p] / pivot : 0.0;
        const bool k = v && fabs(x) > D.droptol;
        const unsigned long long kb = __ballot(k);
        if (k) {
            const int d = put + wave_prefix_count(kb);
            D.lidx[d] = fa->pcI[p];
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
// Early search: when the next pivot is a column singleton (half of all pivots of an LP basis), wave 0 finds
// and stages it WHILE the other waves write out pivot k (U row, L column, tail-append of the moved
// columns).  After the barrier that ends the line updates everything the next search reads is final --
// column entries, line metadata, and the count lists with the columns of the pivot row already unlinked
// (wave_list_unlink_set ran beside the line updates) -- except that those columns are being re-appended
// at the tails of their new lists at this very moment.  The first column of list 1 AFTER the update is
// nevertheless known: the head's link in memory if it leads to an unmoved column (an unmoved column
// keeps its place; a moved one met there has just been appended: no unmoved member), else the first
// column of the pivot row whose new count is 1 (new counts, begins and maxima are in LDS; moved columns
// are appended in pivot-row order).  The result equals mk_express's; `make ewcheck` compares every early
// result with the ordinary search.  Not attempted: a column became empty or numerically null, a
// cancellation fix-up is pending, row search is on, the long form of the list update was taken.
// ------------------------------------------------------------------------------------------------
// ebase = slot of elems[0] in the pivot row (the e1 arrays are indexed by slot)
__device__ __forceinline__ void early_search(const DevGP &D, Sm *sm, Mc *mc, const int *elems, const int *keys, const int *begs, const double *maxs,
                                             int n, int ebase, long long &ew_mcb, int &ew_fb)
{
    const int lane = lane_id();
    const int m = D.m;
    Fast *fa = &sm->fa;
    const int K = D.maxsearch;
    if (!BLU_EARLY || D.search_rows || D.no_fast || K < 1 || K > KCMAX || m >= (1 << 27) || n >= 64) return;
    if (sm->flag_small || fa->anycancel) return;
    const int left = m - (sm->rank + 1) - sm->rankdef; // active columns at the next search
    if (left < 1) return;
    // my moved column (lane q < n): its new count
    const int kq = lane < n ? keys[lane] : -1;
    if (__ballot(kq == 0)) return; // an empty column: the ordinary search takes it (markowitz.rs:73-78)
    int nsr = 0;

    // ---- column singletons first (mk_express's case, half of all pivots): the head of list 1 if it is an
    // unmoved column, else the first moved column whose new count is 1
    {
        const int h1 = LinksC{D, mc}.fl(m + 1);
        const unsigned long long one = __ballot(kq == 1);
        int j = -1, cb = 0, fs = -1;
        double cmx = 0.0;
        if (h1 < m && !hcol_has(fa, h1)) {
            j = h1;
            cb = D.cbeg[j];
            cmx = D.colmax[j];
            if (D.clen[j] != 1) return;
        } else if (one) {
            const int l = __ffsll((long long)one) - 1;
            j = elems[l];
            cb = begs[l];
            cmx = maxs[l];
            // a column of two that lost its pivot-row entry: the wave that updated it left the entry that stays and
            // the metadata of its row in LDS (fast_scol): no round trip at all
            if (mc && ebase + l < MC_PREV && mc->e1i[ebase + l] >= 0 && mc->e1rl[ebase + l] >= 0) fs = ebase + l;
        }
        if (j >= 0) {
            if (cmx == 0.0 || !(cmx >= D.abstol)) return;
            int idx, rb, rl, rc;
            double val;
            if (fs >= 0) {
                idx = mc->e1i[fs];
                val = mc->e1v[fs];
                rb = mc->e1rb[fs];
                rl = mc->e1rl[fs];
                rc = mc->e1rc[fs];
            } else {
                idx = D.cidx[cb];
                val = D.cval[cb];
                rb = D.rbeg[idx];
                rl = D.rlen[idx];
                rc = D.rcap[idx];
            }
            const double tol = fmax(D.abstol, D.reltol * cmx);
            const double x = fabs(val);
            if (x == 0.0 || x < tol) return;
            nsr = left < K ? left : K;
            if (lane == 0) {
                fa->ncand = 1;
                fa->cJ[0] = j;
                fa->cNz[0] = 1;
                fa->cB[0] = cb;
                fa->cL[0] = 1;
                fa->cMx[0] = cmx;
                fa->cOff[0] = 0;
                fa->cOff[1] = 1;
                fa->sI[0] = idx;
                fa->sV[0] = val;
                fa->sB[0] = rb;
                fa->sL[0] = rl;
                fa->sC[0] = rc;
                fa->ewNsr = nsr;
                fa->ewValid = 1;
            }
            ew_mcb = 0;
            ew_fb = 0;
            wave_mem_sync();
            return;
        }
    }
    // No singleton: a full walk HERE (3 dependent candidate loads, then the staging) takes ~8 000 cycles
    // against ~5 500 in the ordinary search and the 2 000 of the finalize step it would hide behind: measured
    // as a net loss.  The walk that pays is the one that starts beside the line updates (spec_walk below).
}

// ------------------------------------------------------------------------------------------------
// Speculative search: the search of pivot k+1 beside the line updates of pivot k (kind 1), on the wave that has
// just unlinked the columns of the pivot row from their count lists and would otherwise wait for the barrier.
//   spec_walk    (beside the line updates) the list walk of the next search on the list state WITHOUT the columns
//                being updated: the first K unmoved columns in list order with their (begin, len, max).  Four
//                dependent round trips -- the long pole of a search -- off the critical path.
//   spec_cond    (after the barrier) K unmoved candidates were found, nothing cancelled, no column fell below
//                abstol, no column became empty.
//   spec_finish  (finalize step, same wave) MERGE: the updated columns are appended at the TAILS of their new
//                lists, in pivot-row order, so the candidates of the next search are the first K of [the walk's
//                unmoved columns] merged with [the updated columns whose new count is below that of the walk's last
//                candidate], by (count, unmoved before updated, order) -- all of it in LDS.  Then what mk_stage and
//                the reduction of mk_pick would do: entries, row metadata, costs, winner, on data that is final by
//                then (the barrier has drained the line updates).
// At C3 the next search is done this way after 47 400 of the 49 927 kind-1 pivots (23 322 of them with updated
// columns merged in): pivot loop 748 -> 667 ms.  The wave runs at raised priority (s_setprio): its chain of loads
// is the longest thing in the phase and it issues few instructions.
// Nothing here is read by the current pivot: the candidate and staging arrays were consumed by its own set-up.
// `make ewcheck` compares every such result -- candidates, count, key and every staged entry -- with the ordinary
// search.  Steps on the way: the whole search before the barrier; without the merge (the walk
// is the next search's own only if no updated column has a smaller count than its last candidate: 48 % of the
// pivots, 698 ms); reuse of the current search's surviving candidates and their staged entries instead of walking
// them again (never more than two of four survive and the bookkeeping costs more instructions on a contended SIMD
// than the two round trips it saves: 715 -> 767 ms, not kept).
// ------------------------------------------------------------------------------------------------
#ifdef BLU_SPEC_STATS
#define SPEC_STAT(i) do { if (lane_id() == 0) atomicAdd((unsigned long long *)&D.s->prof[40 + (i)], 1ull); } while (0)
#else
#define SPEC_STAT(i) do { } while (0)
#endif
__device__ __forceinline__ void spec_walk(const DevGP &D, Sm *sm, Mc *mc)
{
    const int lane = lane_id();
    const int m = D.m;
    Fast *fa = &sm->fa;
    const int K = D.maxsearch;
    if (lane == 0) fa->spOk = 0;
    if (!mc || D.no_fast || K < 1 || K > KCMAX || m >= (1 << 27)) return;
    if (m - (sm->rank + 1) - sm->rankdef < K) return;
    const LinksC LC{D, mc};
    if (LC.fl(m) != m) return;
    SPEC_STAT(0);
    PROF_STAMP_L0(41);
    int ncand = 0, total = 0, lastnz = 0;
    int nz = sm->min_colnz; // (the current search's: a lower bound of every unmoved column's count)
    bool bad = false;
    while (ncand < K && nz <= m && !bad) {
        const int k = nz + lane;
        const int h = k <= m ? LC.fl(m + k) : m + k;
        unsigned long long ne = __ballot(k <= m && h != m + k);
        while (ne && ncand < K && !bad) {
            const int b = __ffsll((long long)ne) - 1;
            ne &= ne - 1;
            int j = wave_bcast_i(h, b);
            const int znz = nz + b;
            int guard = 0;
            while (j < m && ncand < K) {
                const int fl = D.cflink.el(j), cb = D.cbeg[j], cl = D.clen[j];
                const double cmx = D.colmax[j];
                if (cl != znz || cmx == 0.0 || !(cmx >= D.abstol) || ++guard > m + 2) {
                    bad = true;
                    break;
                }
                if (lane == 0) {
                    fa->cJ[ncand] = j;
                    fa->cNz[ncand] = znz;
                    fa->cB[ncand] = cb;
                    fa->cL[ncand] = cl;
                    fa->cMx[ncand] = cmx;
                    fa->cOff[ncand] = total;
                }
                total += cl;
                ncand++;
                lastnz = znz;
                j = fl;
            }
        }
        nz += 64;
    }
    if (bad || ncand < K || total > STGMAX) return;
    SPEC_STAT(1);
    if (lane == 0) {
        fa->cOff[ncand] = total;
        fa->ncand = ncand;
        fa->spLastNz = lastnz;
        fa->spOk = 1;
    }
    wave_mem_sync();
    PROF_STAMP_L0(42);
}

// After the barrier that ends the line updates: can the walk above still become the next search?  Evaluated by the
// wave that walked (it goes on to stage the entries, spec_finish) and by wave 0 (it looks for a column singleton
// instead if not, early_search): the same LDS words, the same answer.  keys = new counts of the n moved columns.
__device__ __forceinline__ bool spec_cond(const Sm *sm, const int *keys, int n)
{
    const int lane = lane_id();
    const Fast *fa = &sm->fa;
    if (!fa->spOk || sm->flag_small || fa->anycancel || n >= 64) return false;
    const int kq = lane < n ? keys[lane] : 0x7fffffff;
    return __ballot(kq <= 0) == 0ull; // (an empty column: the ordinary search takes it, markowitz.rs:73-78)
}

// The second half, in the finalize step (this wave has no other job there): the candidates' entries and the
// metadata of their rows -- final now, also for the rows this pivot rewrote: the barrier has drained the stores of
// the line updates -- the cost of every eligible entry, the winner.  Publishes the result for the next search
// (ewValid = 2).  (Staging before the barrier and only the costs here was measured too: the barrier comes later
// by more than this step gets shorter, 699 -> 707 ms.)
__device__ __forceinline__ void spec_finish(const DevGP &D, Sm *sm, int n)
{
    const int lane = lane_id();
    Fast *fa = &sm->fa;
    const int ncand = fa->ncand;
    // ---- merge.  The next search walks, count by count, the unmoved columns of a list in their old order and then
    // the columns this pivot appended to it, in pivot-row order.  The walk has the first K unmoved ones (counts
    // ascending, the last one's = spLastNz); an updated column comes before the K-th candidate exactly if its new
    // count is below spLastNz, and then every unmoved column of its count or less is among the K.  Positions:
    //   updated column q:    #(unmoved with count <= its count) + #(entering updated columns before it by (count, q))
    //   unmoved candidate i: i + #(entering updated columns with a count below its own)
    // and the first K positions are the candidates of the next search.
    {
        const int lastnz = fa->spLastNz;
        const int key = lane < n ? fa->tNew[1 + lane] : 0x7fffffff;
        const unsigned long long inb = __ballot(key < lastnz);
        if (inb) {
            int uJ = 0, uNz = 0x7fffffff, uB = 0, uL = 0;
            double uMx = 0.0;
            if (lane < ncand) {
                uJ = fa->cJ[lane];
                uNz = fa->cNz[lane];
                uB = fa->cB[lane];
                uL = fa->cL[lane];
                uMx = fa->cMx[lane];
            }
            int ule = 0;
            for (int i = 0; i < ncand; i++) ule += fa->cNz[i] <= key;
            int before = 0, below = 0;
            unsigned long long mb = inb;
            while (mb) {
                const int b = __ffsll((long long)mb) - 1;
                mb &= mb - 1;
                const int kb = __builtin_amdgcn_readlane(key, b);
                before += (kb < key) || (kb == key && b < lane);
                below += kb < uNz;
            }
            const int posM = ule + before, posU = lane + below;
            wave_mem_sync();
            if (lane < ncand && posU < ncand) {
                fa->cJ[posU] = uJ;
                fa->cNz[posU] = uNz;
                fa->cB[posU] = uB;
                fa->cL[posU] = uL;
                fa->cMx[posU] = uMx;
            }
            if (((inb >> lane) & 1ull) && posM < ncand) {
                fa->cJ[posM] = fa->tJ[1 + lane];
                fa->cNz[posM] = key;
                fa->cB[posM] = fa->tB[1 + lane];
                fa->cL[posM] = key;
                fa->cMx[posM] = fa->tMx[1 + lane];
            }
            wave_mem_sync();
            if (lane == 0) {
                int off = 0;
                for (int i = 0; i < ncand; i++) {
                    fa->cOff[i] = off;
                    off += fa->cL[i];
                }
                fa->cOff[ncand] = off;
            }
            wave_mem_sync();
            SPEC_STAT(4);
        }
    }
    const int total = fa->cOff[ncand];
    if (total > STGMAX) return;
    const int off1 = ncand > 1 ? fa->cOff[1] : 0x7fffffff, off2 = ncand > 2 ? fa->cOff[2] : 0x7fffffff,
              off3 = ncand > 3 ? fa->cOff[3] : 0x7fffffff;
    const long long BIG = 0x7fffffffffffffffLL;
    long long mcb = BIG;
    int fb = 0x7fffffff;
    for (int base = 0; base < total; base += 64) {
        const int f = base + lane;
        if (f < total) {
            const int c = (f >= off1) + (f >= off2) + (f >= off3);
            const int pos = fa->cB[c] + (f - fa->cOff[c]);
            const int idx = D.cidx[pos];
            const double val = D.cval[pos];
            const int rb = D.rbeg[idx], rl = D.rlen[idx], rc = D.rcap[idx];
            fa->sI[f] = idx;
            fa->sV[f] = val;
            fa->sB[f] = rb;
            fa->sL[f] = rl;
            fa->sC[f] = rc;
            const double tol = fmax(D.abstol, D.reltol * fa->cMx[c]);
            const double x = fabs(val);
            if (!(x == 0.0 || x < tol)) {
                const long long cost = (long long)(fa->cNz[c] - 1) * (long long)(rl - 1);
                if (cost < mcb) {
                    mcb = cost;
                    fb = f;
                }
            }
        }
    }
    PROF_WAIT();
    PROF_STAMP_L0(43);
    SPEC_STAT(2);
    const long long key = wave_min_ll(mcb != BIG ? mcb * 256LL + (long long)fb : BIG);
    if (key == BIG) return;
    SPEC_STAT(3);
    if (lane == 0) {
        fa->spKey = key;
        fa->ewNsr = ncand;
        fa->ewValid = 2;
    }
    wave_mem_sync();
    PROF_STAMP_L0(44);
}

// ------------------------------------------------------------------------------------------------
// kind 1, whole workgroup
// ------------------------------------------------------------------------------------------------
template <bool BATCH>
__device__ __forceinline__ void fast_small(const DevGP &D, Sm *sm, Mc *mc, int pr, int pc, int nzc, int nzr, long long &ew_mcb, int &ew_fb)
{
    const int w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Fast *fa = &sm->fa;
    const int cnz1 = nzc - 1, rnz1 = nzr - 1;
    const double pivot = fa->pcV[0];
    DEV_CHECK(D.s, pivot != 0.0);

    double *work = &sm->swork[w * 64];
    // tasks 0..rnz1-1 = columns of the pivot row, rnz1.. = rows of the pivot column.  Three tasks at a
    // time: all first-chunk loads are issued before any of the lines is processed, so their latencies
    // overlap instead of adding up.  A column costs about twice a row, so the second of the three goes
    // round the waves in reverse: the waves that got the last columns get no second column.
    const int ntask = rnz1 + cnz1;
    PROF_STAMP(4);
    // With 8 or more waves (the single-matrix configuration; a 4-wave batch workgroup cannot spare one) the
    // last wave does not take line updates: it unlinks the columns of the pivot row from their count lists
    // meanwhile (their new lists are known only after the updates: the append
    // half follows in the finalize step).
    const bool split = !BATCH && nw >= 8 && rnz1 < 64;
    const int nwt = split ? nw - 1 : nw;
    const LinksC LC{D, mc};
    const bool early = BLU_EARLY && split && !D.search_rows;
    if (split && w == nw - 1) {
        // this wave's chain of dependent loads is the longest thing in the phase and it issues few instructions:
        // it goes first on its SIMD
        if (early && BLU_SPEC) __builtin_amdgcn_s_setprio(3);
        const bool lnk = fa->tLnk != 0; // (elems = tJ + 1: staged links from slot 1, the pivot column's at slot 0 = index -1)
        wave_list_unlink_set(LC, fa->tJ + 1, rnz1, -1, InHCol{fa, 1, 0}, pc, lnk ? fa->tFl + 1 : nullptr, lnk ? fa->tBl + 1 : nullptr, -1);
        if (early && BLU_SPEC) {
            wave_mem_sync();
            spec_walk(D, sm, mc);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    for (int base = 0; base < ntask && w < nwt; base += 3 * nwt) {
        int li[3], tt[3];
        double lv[3];
        tt[0] = base + w;
        tt[1] = base + 2 * nwt - 1 - w;
        tt[2] = base + 2 * nwt + w;
        // (begin, len) of the three lines first, unconditionally (slot 0 stands in for "no task"): the LDS
        // reads are independent and return together
        int lb[3], ll[3];
        bool isc[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int t = tt[u];
            isc[u] = t < rnz1;
            const bool isr = !isc[u] && t < ntask;
            const int q = isc[u] ? t + 1 : 0, p = isr ? t - rnz1 + 1 : 0;
            const int cb_ = fa->tB[q], cl_ = fa->tL[q], rb_ = fa->prB[p], rl_ = fa->prL[p];
            lb[u] = isc[u] ? cb_ : rb_;
            ll[u] = isc[u] ? cl_ : (isr ? rl_ : 0);
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            li[u] = -1;
            lv[u] = 0.0;
            if (lane < ll[u]) {
                if (isc[u]) {
                    li[u] = D.cidx[lb[u] + lane];
                    lv[u] = D.cval[lb[u] + lane];
                } else {
                    li[u] = D.ridx[lb[u] + lane];
                }
            }
        }
#ifdef BLU_PROFILE
        if (w == 1 && base == 0) {
            PROF_STAMP_L0(33);
            PROF_WAIT();
            PROF_STAMP_L0(34);
        }
#endif
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int t = tt[u];
            if (t < rnz1) {
                if (ll[u] <= 64) fast_col_short(D, sm, mc, t + 1, work, li[u], lv[u], pr, cnz1, pivot);
                else fast_col(D, sm, mc, t + 1, work, li[u], lv[u], pr, cnz1, pivot);
            } else if (t < ntask) {
                if (ll[u] <= 64) fast_row_short(D, sm, mc, t - rnz1 + 1, li[u], pc, rnz1);
                else fast_row(D, sm, mc, t - rnz1 + 1, li[u], pc, rnz1);
            }
#ifdef BLU_PROFILE
            if (w == 1 && base == 0) PROF_STAMP_L0(35 + u);
#endif
        }
    }
#ifdef BLU_PROFILE
    if (w == 1) {
        PROF_WAIT();
        PROF_STAMP_L0(38);
    }
#endif
    PROF_STAMP(5);
    __syncthreads();
    PROF_STAMP(3);
    if (fa->anycancel) {
        for (int p = 1 + w; p <= cnz1; p += nw) fast_fixrow(D, sm, mc, p);
        __syncthreads();
    }
    // finalize step, one job per wave: [0] the search of the NEXT pivot (early_search), [1] L column,
    // [2] count lists, [3] U row and the pivot's own bookkeeping; with fewer than 4 waves (or row search)
    // wave 0 writes the U row instead and the next search waits for the barrier
    const bool spec = BLU_SPEC && early && (w == 0 || w == nw - 1) && spec_cond(sm, fa->tNew + 1, rnz1);
    if (w == 0 && early && !spec) early_search(D, sm, mc, fa->tJ + 1, fa->tNew + 1, fa->tB + 1, fa->tMx + 1, rnz1, 1, ew_mcb, ew_fb);
    if (w == nw - 1 && spec) spec_finish(D, sm, rnz1);
    if (w == (early ? 3 : 0)) {
        fast_write_u(D, sm, 1, rnz1, -1);
        if (lane == 0) {
            D.colmax[pc] = fa->pcV[0];
            D.clen[pc] = 0;
            D.rlen[pr] = 0;
            sm->kinds[3]++;
        }
        PROF_STAMP_L0(6);
    }
    if (w == 1 % nw) {
        fast_write_l(D, sm);
        PROF_STAMP_L0(24);
    }
    if (w == 2 % nw) {
        PROF_STAMP_L0(25);
        const int mn = split ? wave_list_append_set(LC, m, fa->tJ + 1, fa->tNew + 1, rnz1, m + 2, fa->kg[0], mc ? mc->pFl + 1 : nullptr)
                             : wave_list_move_batch_set(LC, m, fa->tJ + 1, fa->tNew + 1, rnz1, InHCol{fa, 1, 0}, m + 2, pc, fa->kg[0]);
        if (lane == 0 && mn < sm->min_colnz) sm->min_colnz = mn;
        if (mc && lane == 0) {
            mc->prevValid = split ? 1 : 0;
            mc->prevBase = 1;
        }
        PROF_WAIT();
        PROF_STAMP_L0(29);
    }
    if (D.search_rows && w == 3 % nw) {
        if (lane == 0) list_remove1(D.rflink, D.rblink, pr); // pr is not in the row hash set: unlink it first
        wave_mem_sync();
        const int mn = wave_list_move_batch_set(LinksG{D.rflink, D.rblink}, m, fa->pcI + 1, fa->rNew + 1, cnz1, InHRow{fa}, m + 2, -1, fa->kg[1]);
        if (lane == 0 && mn < sm->min_rownz) sm->min_rownz = mn;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// kind 2: pivot_singleton_col (pivot.rs:928-1025), whole workgroup
// (Round 2 also built this pivot kind -- half of all pivots, in long runs -- as a single-wave route with search
// and elimination fused, no workgroup barrier, the pivot row in the registers of wave 0 and a register stash
// from one pivot to the next.  Bit-identical, and no faster: a lone wave issues one instruction every ~7 cycles,
// and the ~2 000 instructions of such a pivot -- column entries 4 600 cycles, batched list update 4 300, the
// dependent look-ups of the search 3 300, stores and bookkeeping 1 800 -- cost the 6.8 us that the sixteen-wave
// route with its three barriers costs: 820 -> 885 ms at C3.  Not kept.)
// ------------------------------------------------------------------------------------------------
template <bool BATCH>
__device__ __forceinline__ void fast_scol(const DevGP &D, Sm *sm, Mc *mc, int pr, int pc, int rl, int wq, long long &ew_mcb, int &ew_fb)
{
    const int w = wave_id(), nw = num_waves(), lane = lane_id();
    const int m = D.m;
    Scalars *S = D.s;
    Fast *fa = &sm->fa;
    DEV_CHECK(S, fa->pcV[0] != 0.0 && fa->pcI[0] == pr);

    // (as in fast_small: with 8 or more waves the last one unlinks the row's columns from their count lists
    // while the others take the column updates)
    const bool split = !BATCH && nw >= 8 && rl < 64;
    const int nwt = split ? nw - 1 : nw;
    const LinksC LC{D, mc};
    if (split && w == nw - 1) {
        const bool lnk = fa->tLnk != 0;
        wave_list_unlink_set(LC, fa->tJ, rl, wq, InHCol{fa, 0, wq}, pc, lnk ? fa->tFl : nullptr, lnk ? fa->tBl : nullptr, wq);
    }
    for (int q = w; q < rl && w < nwt; q += nwt) {
        if (q == wq) {
            if (lane == 0) fa->tNew[q] = -1;
            continue;
        }
        const int j = fa->tJ[q], cb = fa->tB[q], cl = fa->tL[q];
        if (cl <= 64) { // (all but a handful) one chunk, straight-line; the maximum through an LDS atomic
            const bool v = lane < cl;
            const int idx = v ? D.cidx[cb + lane] : -1;
            const double val = v ? D.cval[cb + lane] : 0.0;
            const unsigned long long hb = __ballot(v && idx == pr);
            DEV_CHECK(S, hb != 0ull);
            const int src = hb ? __ffsll((long long)hb) - 1 : 0;
            const double xrj = wave_bcast_d(val, src);
            unsigned long long *wm = &sm->wmax[w];
            // the one entry that stays in a column of two: the column is a singleton now, very likely the next pivot
            // column.  The next search finds the entry in LDS, and the metadata of its row as well: loaded here, by
            // the one lane that holds the entry (rows do not change in a singleton-column pivot), stored at the end.
            const bool fw1 = mc && cl == 2 && q < MC_PREV && v && lane != src;
            int f_rb = 0, f_rl = 0, f_rc = 0;
            if (fw1) {
                f_rb = D.rbeg[idx];
                f_rl = D.rlen[idx];
                f_rc = D.rcap[idx];
                mc->e1i[q] = idx;
                mc->e1v[q] = val;
            }
            if (v && lane != src) {
                const double x = fabs(val);
                if (x > 0.0) atomicMax(wm, (unsigned long long)__double_as_longlong(x));
            }
            // last entry into the hole (pivot.rs:991-993): it is in lane cl - 1
            const int last_i = wave_bcast_i(idx, cl - 1);
            const double last_v = wave_bcast_d(val, cl - 1);
            wave_mem_sync();
            if (lane == 0 && hb) {
                const double cmx = __longlong_as_double((long long)*wm);
                *wm = 0ull;
                D.cidx[cb + src] = last_i;
                D.cval[cb + src] = last_v;
                D.clen[j] = cl - 1;
                D.colmax[j] = cmx;
                fa->tNew[q] = cl - 1;
                if (q < 64) fa->tMx[q] = cmx;
                if (mc && cl != 2 && q < MC_PREV) mc->e1i[q] = -1;
                fa->tX[q] = xrj;
                if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
            }
            if (fw1) {
                mc->e1rb[q] = f_rb;
                mc->e1rl[q] = f_rl;
                mc->e1rc[q] = f_rc;
            }
            continue;
        }
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
                xrj = wave_bcast_d(val, src);
            }
            if (v && idx != pr) {
                const double x = fabs(val);
                if (x > cmxl) cmxl = x;
                if (mc && cl == 2 && q < MC_PREV) { // (never here: a column of two takes the short form above)
                    mc->e1i[q] = idx;
                    mc->e1v[q] = val;
                    mc->e1rl[q] = -1;
                }
            }
        }
        DEV_CHECK(S, where >= 0);
        const double cmx = wave_max_d(cmxl);
        if (lane == 0 && where >= 0) {
            D.cidx[cb + where] = D.cidx[cb + cl - 1]; // last entry into the hole (pivot.rs:991-993)
            D.cval[cb + where] = D.cval[cb + cl - 1];
            D.clen[j] = cl - 1;
            D.colmax[j] = cmx;
            fa->tNew[q] = cl - 1;
            if (q < 64) fa->tMx[q] = cmx;
            if (mc && cl != 2 && q < MC_PREV) mc->e1i[q] = -1;
            fa->tX[q] = xrj;
            if (cmx == 0.0 || cmx < D.abstol) sm->flag_small = 1;
        }
    }
    __syncthreads();
    // finalize step: [0] the search of the next pivot, [1] count lists, [2] U row and bookkeeping
    const bool early = BLU_EARLY && split && !D.search_rows;
    if (w == 0 && early) early_search(D, sm, mc, fa->tJ, fa->tNew, fa->tB, fa->tMx, rl, 0, ew_mcb, ew_fb);
    if (w == (early ? 2 : 0)) {
        fast_write_u(D, sm, 0, rl - 1, wq);
        if (lane == 0) {
            D.lbeg[sm->rank + 1] = sm->lused; // empty column in L
            D.colmax[pc] = fa->pcV[0];
            D.clen[pc] = 0;
            D.rlen[pr] = 0;
            sm->kinds[1]++;
        }
    }
    if (w == 1 % nw) {
        if (D.search_rows && lane == 0) list_remove1(D.rflink, D.rblink, pr);
        // the pivot column sits at slot `where` of the row with key -1: it is unlinked as `gone`
        const int mn = split ? wave_list_append_set(LC, m, fa->tJ, fa->tNew, rl, m + 2, fa->kg[0], mc ? mc->pFl : nullptr)
                             : wave_list_move_batch_set(LC, m, fa->tJ, fa->tNew, rl, InHCol{fa, 0, wq}, m + 2, pc, fa->kg[0]);
        if (lane == 0 && mn < sm->min_colnz) sm->min_colnz = mn;
        if (mc && lane == 0) { // what the next search may reuse (a column that sank below abstol cancels it: dirty)
            mc->prevValid = split ? 1 : 0;
            mc->prevBase = 0;
        }
    }
    __syncthreads();
}
