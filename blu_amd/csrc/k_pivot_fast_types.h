// k_pivot_fast_types.h -- LDS working set of the low-latency pivot paths (see k_pivot_fast.hip)
#pragma once
#define PCMAX 65    // cached pivot column entries (pivot_small: <= 64 off-diagonals)
#define PRMAX 256   // cached pivot row entries
#define STGMAX 192  // staged Markowitz candidate entries
#define KCMAX 4     // candidate columns handled by the flattened search (maxsearch <= KCMAX)
#define HROW 256    // hash slots, rows of the pivot column (<= 64 keys)
#define HCOL 512    // hash slots, columns of the pivot row (<= 256 keys)
#define KGMAX 256    // batched list moves: keys (new counts) below this meet in an LDS table
#define HROW_BITS 8
#define HCOL_BITS 9
#define QMAX 8      // candidate queue: leading columns of the count lists kept across pivots
// The queue is compiled out by default: on banded LP bases most queued columns sit in the very next
// pivot row (24 % of the searches of the 100k benchmark basis were served from it, the top-up loads
// cost more than those saved).  -DBLU_QUEUE=1 enables it, `make qcheck` verifies it against the lists.
#ifndef BLU_QUEUE
#define BLU_QUEUE 0
#endif

struct Fast {
    int kind;  // 0 none (general paths), 1 pivot_small, 2 pivot_singleton_col
    int where; // singleton col: slot of the pivot column in the (unswapped) pivot row
    int anycancel;
    int ncand;
    int cJ[KCMAX], cNz[KCMAX], cB[KCMAX], cL[KCMAX], cOff[KCMAX + 1];
    double cMx[KCMAX];
    // candidate queue (see q_prepare in k_pivot_fast.hip): the first qN active columns in search order
    // (count lists 1,2,.. each from its head), with their (count, begin, len, max).  qCont = successor of
    // the last one in its list if known (>= m: end of list qContNz seen during this search; -1 unknown).
    int qN, qCont, qContNz, qMinNew;
    int qJ[QMAX], qNz[QMAX], qB[QMAX], qL[QMAX];
    double qMx[QMAX];
    // pivot column, pivot at slot 0 (kind 1), with the (begin,len,cap) of each row
    int pcI[PCMAX], prB[PCMAX], prL[PCMAX], prC[PCMAX], rNew[PCMAX], rKept[PCMAX], rDst[PCMAX];
    double pcV[PCMAX];
    // pivot row, pivot column at slot 0 (kind 1), with the (begin,len,cap) of each column
    int tJ[PRMAX], tB[PRMAX], tL[PRMAX], tC[PRMAX], tNew[PRMAX];
    double tX[PRMAX];
    unsigned long long tM[PRMAX];
    // staged candidate entries
    int sI[STGMAX], sB[STGMAX], sL[STGMAX], sC[STGMAX];
    double sV[STGMAX];
    unsigned long long hRow[HROW]; // (row index << 32) | position, ~0 = empty: one LDS read per probe
    unsigned long long hCol[HCOL]; // (column index << 32) | slot in tJ, ~0 = empty
    int ls[2][320]; // scratch of the batched list moves (0: column lists, 1: row lists)
    unsigned long long kg[2][KGMAX]; // key -> lanes with that key, all zero between list moves
};

