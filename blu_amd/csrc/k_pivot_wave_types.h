// k_pivot_wave_types.h -- LDS working set of the one-wave-per-matrix pivot kernel (k_pivot_wave.hip).
// (no include guard: included inside the pv_wave namespace of k_pivot.hip)
//
// One wave = one matrix, so nothing here is shared between waves and no workgroup barrier is needed; the
// whole struct is sized so that many workgroups of one wave each share a CU (LDS is the limit next to
// registers).  "slot" = position in the pivot row (columns, slot 0 = pivot column) or in the pivot
// column (rows, slot 0 = pivot row) after the swap of pivot.rs:169-185; a line update handles slots
// 1..n, which live in lanes 0..n-1.
#undef WV_SLOTS
#undef WV_HBITS
#undef WV_HASH
#undef WV_ZW
#undef WV_TMAX
#undef WV_WCAP
#undef WV_STG
#undef KCMAX
#undef WV_NW
#undef WV_WHALF
#define WV_NW BLU_CFG_WAVE // waves per matrix: 1 (k_pivot_loop_wave) or 2 (k_pivot_loop_wave2, k_pivot_wave2.inc)
#define WV_SLOTS 64   // lines per phase: pivot rows and pivot columns of <= 64 entries (one lane each)
#define WV_HBITS 7
#define WV_HASH 128   // hash slots: <= 64 keys, at most half full
#define WV_ZW 64      // zero-words: segment-head bits of a flattened pass / key groups of a list move
#define WV_TMAX (64 * WV_ZW) // entries of all lines of one flattened phase
#ifndef WV_WCAP_SET
#define WV_WCAP_SET (256 * WV_NW)
#endif
#define WV_WCAP WV_WCAP_SET // old values of the entries being updated, columns of one group x pivot-column positions
#define WV_WHALF (WV_WCAP / 2) // ... of which each wave of the two-wave kernel has one half
#define WV_STG 128    // entries of the candidate columns of one search
#define KCMAX 4       // candidate columns of a search (maxsearch <= KCMAX)

struct Fast {
    int kind;  // 0 general paths, 1 pivot_small, 2 pivot_singleton_col
    int where, anycancel, ncand;
    // hand-over from a singleton-column pivot: the next pivot is a column singleton whose one entry that pivot saw
    int nxValid, nxPc, nxPr, nxPcb;
    double nxVal;
    int cJ[KCMAX], cNz[KCMAX], cB[KCMAX], cL[KCMAX], cOff[KCMAX + 1];
    double cMx[KCMAX];
    unsigned long long hsh[WV_HASH]; // (key << 32) | value, ~0 = empty
    unsigned long long zw[WV_ZW];    // all zero between uses
    int2 sBO[WV_SLOTS];              // line of slot s: {begin, offset of its first entry in the flattened index space}
    int sCnt[WV_SLOTS];              // entries kept (written by the last lane of the line in each pass)
    double sX[WV_SLOTS];             // pivot-row entry xrj, then the multiplier xrj / pivot
    unsigned long long sMax[WV_SLOTS]; // bit pattern of the line's new maximum, zero between uses
#if WV_NW == 2
    // Two waves: the lines of a pivot_small are dealt out to the waves, so what the search (wave 0) lays out goes through
    // LDS, one entry per line (index = the lane of the one-wave layout: column slot c+1 <-> c, row slot p+1 <-> p;
    // index rnz1 of lJ / lFl / lBl = the pivot column).  lCl turns into the column's NEW length once its wave is done
    // with it (the key of the list move), lXr is its pivot-row entry (the U row): both for wave 0, which finishes the pivot.
    int csplit;         // wave 0 takes the columns below, wave 1 the rest (and all the rows)
    int tiny;           // a column's maximum fell below abstol (pivot.rs:98-106)
    int lJ[WV_SLOTS], lCb[WV_SLOTS], lCl[WV_SLOTS], lCap[WV_SLOTS], lFl[WV_SLOTS], lBl[WV_SLOTS];
    int lI[WV_SLOTS], lRb[WV_SLOTS], lRl[WV_SLOTS], lRc[WV_SLOTS];
    double lXr[WV_SLOTS];
    unsigned long long hshr[WV_HASH]; // rows of the pivot column -> position (hsh holds the columns of the pivot row)
    unsigned long long zw2[WV_ZW];    // the zero-words of wave 1
    double pV[WV_SLOTS];              // (read by both waves while slots are being reused: no overlay here)
    int pI[WV_SLOTS];
    union {
        double sK0v[WV_SLOTS];
        unsigned long long sM[WV_SLOTS];
    };
    union {
        int sK0i[WV_SLOTS];
        int sNew[WV_SLOTS];
    };
#else
    // Three arrays live one after the other in the same bytes (slot by slot: a slot's earlier use is over before its
    // later one begins -- the pivot column is read out when wv_small starts; a column's first kept entry is consumed
    // by its group's epilogue before its pass B writes the mask / count):
    union {
        double pV[WV_SLOTS];             // pivot column values, slot order (slot 0 = the pivot)
        double sK0v[WV_SLOTS];           // first kept entry of the line (it goes where the pivot-row entry was, pivot.rs:261)
        unsigned long long sM[WV_SLOTS]; // cancellation mask of the column (bit p = position p+1 of the pivot column, pivot.rs:645-664)
    };
    union {
        int pI[WV_SLOTS];                // pivot column row indices, slot order
        int sK0i[WV_SLOTS];
        int sNew[WV_SLOTS];              // entries appended
    };
#endif
    union {
        int sW[WV_SLOTS];                // rank of the pivot-row entry among the kept entries (read by the epilogue ...)
        int sDst[WV_SLOTS];              // ... which then writes where the appended part of the line begins)
    };
    int tJ[WV_SLOTS];                // pivot row, slot order
};
