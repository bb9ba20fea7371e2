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
//   phase A  [1024, 2048) the bucket cursors; from 2048: tb[g] = bucket of line 64 g (16 bits each)
//   phase B  from 1024:   `span` values, `span` keys, `span` line cursors
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

struct Buckets {
    int span, slack, ch, nb;
    int *bnd, *cur;
    unsigned short *tb;
    int tbcap;
    double *lv;
    int *lk, *lcur;
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
    B.span = wincap >= 4096 ? ((wincap - 1024) / 4) & ~63 : 0;
    B.slack = B.span / 8;
    B.ch = B.span - B.slack;
    B.nb = 0;
    B.bnd = win;
    B.cur = win + 1024;
    B.tb = (unsigned short *)(win + 2048);
    B.tbcap = wincap >= 4096 ? (wincap - 2048) * 2 : 0;
    B.lv = (double *)(win + 1024);
    B.lk = win + 1024 + 2 * B.span;
    B.lcur = win + 1024 + 3 * B.span;
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

// Phase B: bucket b.  extra(t, &key, &val): the entry a line ends with beyond its records (perline = 1), out(pos, key, val):
// the store of output entry pos.  Lines of at most BKT_SSORT records leave sorted by key (ties in arrival order); longer
// ones in arrival order.  Returns (per thread) whether a sorted line holds a key twice.
template <class Scope, class Extra, class Out>
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
                        const bool same = j < q && k[j] == k[q];
                        r += (k[j] < k[q] || same) ? 1 : 0;
                        dup |= same ? 1 : 0;
                    }
                    B.lk[lb + r] = k[q];
                    B.lv[lb + r] = v[q];
                }
            }
        }
    }
    sc.sync();
    for (int q = tid; q < out1 - out0; q += nt) out(out0 + q, B.lk[q], B.lv[q]);
    sc.sync();
    return dup;
}
