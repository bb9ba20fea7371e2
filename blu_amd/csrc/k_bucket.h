// k_bucket.h -- the transposing fill of a batch's O(nnz) kernels in two phases (k_prep: the row-wise copy of B,
// singletons.rs:186-198; k_finish: the columns of U, get_factors.rs:136-167).
//
// A fill sends entry (line, target, value) to a cursor of its TARGET line.  With the cursors of a 36 864-line window in LDS
// (prep_body, finish_body) the sweep over the entries runs once per window, and its stores -- 8-byte pieces, each at the
// cursor of another line -- meet as many open cache lines as the window has target lines: far more than the L2 of an XCD
// keeps for the 32 workgroups behind it, so lines leave it partly written and come back.  Here instead:
//
//   phase A  ONE sweep.  The output is cut into buckets of `ch` consecutive entries (a few hundred buckets per matrix); an
//            entry goes, as one 16-byte record, to the cursor of the BUCKET its target line starts in.  A few hundred
//            append fronts per workgroup stay in L2 until their lines are full.
//   phase B  bucket by bucket: the records (contiguous, read once) are scattered to their lines inside LDS, short lines
//            are sorted there, and the bucket's stretch of the output is written once, in order.
//
// LDS (the window of the launch, `wincap` ints):
//   [0, 1024)            bnd[b] = first target line that starts in bucket b (bnd[nb..] = number of lines)
//   [1024, 1536)         phase B: the lines of the bucket that a whole wave sorts (their number, then start and length)
//   phase A  [1536, 2560) the bucket cursors; from 2560: tb[g] = bucket of line 64 g (16 bits each)
//   phase B  from 1536:   `span` values, `span` keys, `span` line cursors
// A target line may hang over the end of its bucket by its own length: `slack` = span / 8 entries are kept for that, and
// a matrix with a longer line, more than 1023 buckets or more lines in one bucket than `span` takes the window sweeps.
#pragma once
#include "blu_dev.h"

struct alignas(16) BktRec {
    int t, k; // target line, source line
    double v;
};
#define BKT_NBMAX 1023
#define BKT_SSORT 32
#define BKT_WSORT 256

struct Buckets {
    int span, slack, ch, nb;
    int *bnd, *cur;
    unsigned short *tb;
    int tbcap;
    double *lv;
    int *lk, *lcur, *wl;
    __device__ __forceinline__ int of(int t) const
    {
        int b = tb[t >> 6];
        while (t >= bnd[b + 1]) b++;
        return b;
    }
};

__device__ __forceinline__ Buckets buckets_in(int *win, int wincap)
{
    Buckets B;
    B.span = wincap >= 4096 ? ((wincap - 1536) / 4) & ~63 : 0;
    B.slack = B.span / 8;
    B.ch = B.span - B.slack;
    B.nb = 0;
    B.bnd = win;
    B.wl = win + 1024;
    B.cur = win + 1536;
    B.tb = (unsigned short *)(win + 2560);
    B.tbcap = wincap >= 4096 ? (wincap - 2560) * 2 : 0;
    B.lv = (double *)(win + 1536);
    B.lk = win + 1536 + 2 * B.span;
    B.lcur = win + 1536 + 3 * B.span;
    return B;
}

// The buckets of `nlines` target lines that start at tstart[0..nlines) (ascending; `total` = end of the last line).
// false (uniform): this matrix does not fit the scheme.  Leaves a barrier behind.
template <class Scope> __device__ __forceinline__ bool buckets_plan(Scope &sc, Buckets &B, gcint_p tstart, int nlines, int total)
{
    const int tid = sc.tid(), nt = sc.nt();
    if (B.span == 0 || nlines <= 0 || total <= 0) return false;
    const int nb = (total - 1) / B.ch + 1;
    if (nb > BKT_NBMAX || (nlines >> 6) + 1 > B.tbcap) return false;
    B.nb = nb;
    for (int b = tid; b <= BKT_NBMAX; b += nt) B.bnd[b] = nlines;
    sc.sync();
    int bad = 0;
    for (int k = tid; k < nlines; k += nt) {
        const int s = tstart[k];
        const int b1 = s / B.ch, b0 = k ? tstart[k - 1] / B.ch : -1;
        for (int b = b0 + 1; b <= b1 && b <= BKT_NBMAX; b++) B.bnd[b] = k;
        bad |= (k + 1 < nlines ? tstart[k + 1] : total) - s > B.slack;
        if ((k & 63) == 0) B.tb[k >> 6] = (unsigned short)b1;
    }
    sc.sync();
    for (int b = tid; b < nb; b += nt) bad |= B.bnd[b + 1] - B.bnd[b] > B.span;
    return !sc.any(bad);
}

// Phase A: cursors.  perline: entries per target line that are NOT records (the pivot of a U column): records of bucket b
// start at tstart[bnd[b]] - perline * bnd[b].
template <class Scope> __device__ __forceinline__ void buckets_open(Scope &sc, const Buckets &B, gcint_p tstart, int nlines, int perline)
{
    for (int b = sc.tid(); b < B.nb; b += sc.nt()) {
        const int t0 = B.bnd[b];
        B.cur[b] = t0 < nlines ? tstart[t0] - perline * t0 : 0;
    }
    sc.sync();
}
__device__ __forceinline__ void bucket_put(const Buckets &B, BktRec *scr, int t, int k, double v)
{
    const int pos = atomicAdd(&B.cur[B.of(t)], 1);
    BktRec r;
    r.t = t;
    r.k = k;
    r.v = v;
    scr[pos] = r;
}

// A line of at most 64 J pairs in LDS ranked by the whole wave, in place: every lane holds up to J pairs in registers,
// counts the keys below each of them (the keys are read by all lanes together, eight loads in flight: LDS broadcasts) and
// stores the pairs where they belong once every lane has read.  Ties keep their order; returns whether this lane saw a key
// twice.  DUPS = false: the keys are known to be distinct (the columns of U).
template <int J, bool DUPS> __device__ __forceinline__ int wave_rank_sort_lds(int *k, double *v, int n)
{
    const int lane = lane_id();
    int kk[J], r[J];
    double vv[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int q = lane + 64 * j;
        kk[j] = q < n ? k[q] : 0x7fffffff;
        vv[j] = q < n ? v[q] : 0.0;
        r[j] = 0;
    }
    int dup = 0;
    const auto rank_against = [&](int ku, int u) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            if (DUPS) {
                const bool same = ku == kk[j] && u < lane + 64 * j;
                r[j] += (ku < kk[j] || same) ? 1 : 0;
                dup |= same ? 1 : 0;
            } else {
                r[j] += ku < kk[j] ? 1 : 0;
            }
        }
    };
    int u = 0;
    for (; u + 8 <= n; u += 8) {
        const int k0 = k[u], k1 = k[u + 1], k2 = k[u + 2], k3 = k[u + 3], k4 = k[u + 4], k5 = k[u + 5], k6 = k[u + 6], k7 = k[u + 7];
        rank_against(k0, u);
        rank_against(k1, u + 1);
        rank_against(k2, u + 2);
        rank_against(k3, u + 3);
        rank_against(k4, u + 4);
        rank_against(k5, u + 5);
        rank_against(k6, u + 6);
        rank_against(k7, u + 7);
    }
    for (; u < n; u++) rank_against(k[u], u);
    WAVE_LOCKSTEP(); // (every lane has read the keys)
#pragma unroll
    for (int j = 0; j < J; j++)
        if (lane + 64 * j < n) {
            k[r[j]] = kk[j];
            v[r[j]] = vv[j];
        }
    WAVE_LOCKSTEP();
    return dup;
}

// Phase B: bucket b.  extra(t, &key, &val): the entry a line ends with beyond its records (perline = 1), out(pos, key, val):
// the store of output entry pos.  Lines of at most BKT_WSORT records leave sorted by key (ties in arrival order: up to
// BKT_SSORT records by their thread, more by its wave); longer ones in arrival order.  Returns (per thread) whether a sorted
// line holds a key twice.
template <bool DUPS, class Scope, class Extra, class Out>
__device__ __forceinline__ int bucket_flush(Scope &sc, const Buckets &B, int b, gcint_p tstart, int nlines, int total, int perline,
                                             const BktRec *scr, Extra extra, Out out)
{
    const int tid = sc.tid(), nt = sc.nt();
    const int t0 = B.bnd[b], t1 = B.bnd[b + 1];
    if (t0 >= t1) return 0; // (uniform)
    int dup = 0;
    const int out0 = tstart[t0], out1 = t1 < nlines ? tstart[t1] : total;
    const int s0 = out0 - perline * t0, s1 = out1 - perline * t1;
    for (int i = tid; i < t1 - t0; i += nt) B.lcur[i] = tstart[t0 + i] - out0;
    if (tid == 0) B.wl[0] = 0;
    sc.sync();
    for (int s = s0 + tid; s < s1; s += nt) {
        const BktRec r = scr[s];
        const int pos = atomicAdd(&B.lcur[r.t - t0], 1);
        B.lk[pos] = r.k;
        B.lv[pos] = r.v;
    }
    sc.sync();
    for (int i = tid; i < t1 - t0; i += nt) {
        const int lb = tstart[t0 + i] - out0, le = B.lcur[i]; // (the records of line t0 + i)
        if (perline) {
            int key;
            double val;
            extra(t0 + i, &key, &val);
            B.lk[le] = key;
            B.lv[le] = val;
        }
        const int n = le - lb;
        if (n > 1 && n <= BKT_SSORT) { // (as small_sort_pairs: everything into registers, ranked there, stored where it belongs)
            int k[BKT_SSORT];
            double v[BKT_SSORT];
#pragma unroll
            for (int q = 0; q < BKT_SSORT; q++) {
                k[q] = 0x7fffffff;
                v[q] = 0.0;
                if (q < n) {
                    k[q] = B.lk[lb + q];
                    v[q] = B.lv[lb + q];
                }
            }
#pragma unroll
            for (int q = 0; q < BKT_SSORT; q++) {
                if (q < n) {
                    int r = 0;
#pragma unroll
                    for (int j = 0; j < BKT_SSORT; j++) {
                        if (DUPS) {
                            const bool same = j < q && k[j] == k[q];
                            r += (k[j] < k[q] || same) ? 1 : 0;
                            dup |= same ? 1 : 0;
                        } else {
                            r += k[j] < k[q] ? 1 : 0;
                        }
                    }
                    B.lk[lb + r] = k[q];
                    B.lv[lb + r] = v[q];
                }
            }
        }
        if (n > BKT_SSORT && n <= BKT_WSORT) B.wl[1 + atomicAdd(&B.wl[0], 1)] = lb * BKT_WSORT + (n - 1); // (at most span / 33 < 511 of them)
    }
    sc.sync();
    // the longer lines, dealt out to the waves (they come in runs: the columns of the bump are neighbours)
    const int nwl = __builtin_amdgcn_readfirstlane(B.wl[0]);
    for (int e = wave_id(); e < nwl; e += num_waves()) {
        const int w = __builtin_amdgcn_readfirstlane(B.wl[1 + e]), mlb = w / BKT_WSORT, mn = w % BKT_WSORT + 1; // (scalars)
        dup |= mn <= 64 ? wave_rank_sort_lds<1, DUPS>(B.lk + mlb, B.lv + mlb, mn) : wave_rank_sort_lds<4, DUPS>(B.lk + mlb, B.lv + mlb, mn);
    }
    sc.sync();
    for (int q = tid; q < out1 - out0; q += nt) out(out0 + q, B.lk[q], B.lv[q]);
    sc.sync();
    return dup;
}
