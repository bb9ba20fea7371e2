// k_pivot_fast_types.h -- LDS working set of the low-latency pivot paths (see k_pivot_fast.hip)
#pragma once
#define PCMAX 65    // cached pivot column entries (pivot_small: <= 64 off-diagonals)
#define PRMAX 256   // cached pivot row entries
#define STGMAX 192  // staged Markowitz candidate entries
#define KCMAX 4     // candidate columns handled by the flattened search (maxsearch <= KCMAX)
#define HROW 256    // hash slots, rows of the pivot column (<= 64 keys)
#define HCOL 1024   // hash slots, columns of the pivot row (<= 256 keys)

struct Fast {
    int kind;  // 0 none (general paths), 1 pivot_small, 2 pivot_singleton_col
    int where; // singleton col: slot of the pivot column in the (unswapped) pivot row
    int anycancel;
    int ncand;
    int cJ[KCMAX], cNz[KCMAX], cB[KCMAX], cL[KCMAX], cOff[KCMAX + 1];
    double cMx[KCMAX];
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
    int hColK[HCOL];
    int ls[2][320]; // scratch of the batched list moves (0: column lists, 1: row lists)
};

