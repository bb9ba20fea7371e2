// k_pivot_fast_types.h -- LDS working set of the low-latency pivot paths (see k_pivot_fast.hip)
#pragma once
#define PCMAX 65    // cached pivot column entries (pivot_small: <= 64 off-diagonals)
#define PRMAX 256   // cached pivot row entries
#define STGMAX 104  // staged Markowitz candidate entries
#define KCMAX 4     // candidate columns handled by the flattened search (maxsearch <= KCMAX)
#define HROW 256    // hash slots, rows of the pivot column (<= 64 keys)
#define HCOL 512    // hash slots, columns of the pivot row (<= 256 keys)
#define KGMAX 128    // batched list moves: keys (new counts) below this meet in an LDS table
#define HROW_BITS 8
#define HCOL_BITS 9
// Early search (early_search in k_pivot_fast.hip): when the next pivot is a column singleton, wave 0 finds
// and stages it while the other waves write out the current pivot.  -DBLU_EARLY=0 switches it off; `make
// ewcheck` builds a library that verifies every early result against the ordinary search.
#ifndef BLU_EARLY
#define BLU_EARLY 1
#endif

struct Fast {
    int kind;  // 0 none (general paths), 1 pivot_small, 2 pivot_singleton_col
    int where; // singleton col: slot of the pivot column in the (unswapped) pivot row
    int anycancel;
    int ncand;
    int cJ[KCMAX], cNz[KCMAX], cB[KCMAX], cL[KCMAX], cOff[KCMAX + 1];
    double cMx[KCMAX];
    // early search: hand-over from the list wave (unlinked runs: predecessor -> first unmoved successor)
    // and the result kept for the next search
    int ewValid, ewNsr;
#if BLU_EARLY
    double tMx[64]; // new maximum of the first 64 columns of the pivot row (the early search takes rows < 64 only)
#else
    double tMx[1];
#endif
    // pivot column, pivot at slot 0 (kind 1), with the (begin,len,cap) of each row
    int pcI[PCMAX], prB[PCMAX], prL[PCMAX], prC[PCMAX], rNew[PCMAX], rKept[PCMAX], rDst[PCMAX];
    double pcV[PCMAX];
    // pivot row, pivot column at slot 0 (kind 1), with the (begin,len,cap) of each column
    int tJ[PRMAX], tB[PRMAX], tL[PRMAX], tC[PRMAX], tNew[PRMAX];
    double tX[PRMAX]; // pivot-row value of the column
    unsigned long long tM[PRMAX];
    // staged candidate entries
    int sI[STGMAX], sB[STGMAX], sL[STGMAX], sC[STGMAX];
    double sV[STGMAX];
    unsigned long long hRow[HROW]; // (row index << 32) | position, ~0 = empty: one LDS read per probe
    unsigned long long hCol[HCOL]; // (column index << 32) | slot in tJ, ~0 = empty
    unsigned long long kg[2][KGMAX]; // key -> lanes with that key, all zero between list moves
};

