// k_pivot_fast_types.h -- LDS working set of the low-latency pivot paths (see k_pivot_fast.hip)
// (no include guard: k_pivot.hip includes this once per configuration, inside its namespace)
#undef PCMAX
#undef PRMAX
#undef STGMAX
#undef KCMAX
#undef HROW
#undef HCOL
#undef KGMAX
#undef HROW_BITS
#undef HCOL_BITS
#undef MC_HEADS
#undef MC_PREV
#define PCMAX 65    // cached pivot column entries (pivot_small: <= 64 off-diagonals)
#define KCMAX 4     // candidate columns handled by the flattened search (maxsearch <= KCMAX)
#if BLU_CFG_BATCH
// batch configuration: pivot rows up to 64 entries (all of them on the banded LP bases; longer rows take the
// general paths), hash tables at half load for that, ~11 KB of LDS per workgroup instead of ~24 KB
#define PRMAX 64
#define STGMAX 80
#define HROW 128
#define HCOL 128
#define KGMAX 64
#define HROW_BITS 7
#define HCOL_BITS 7
#else
#define PRMAX 256   // cached pivot row entries
#define STGMAX 104  // staged Markowitz candidate entries
#define HROW 256    // hash slots, rows of the pivot column (<= 64 keys)
#define HCOL 512    // hash slots, columns of the pivot row (<= 256 keys)
#define KGMAX 128   // batched list moves: keys (new counts) below this meet in an LDS table
#define HROW_BITS 8
#define HCOL_BITS 9
#endif
// Early search (early_search in k_pivot_fast.hip): when the next pivot is a column singleton, wave 0 finds
// and stages it while the other waves write out the current pivot.  -DBLU_EARLY=0 switches it off; `make
// ewcheck` builds a library that verifies every early result against the ordinary search.
#ifndef BLU_EARLY
#define BLU_EARLY 1
#endif
// Speculative search (spec_walk / spec_finish in k_pivot_fast.hip): the search of the next pivot beside the line
// updates and the finalize step of a kind-1 pivot.  -DBLU_SPEC=0 switches it off.
#ifndef BLU_SPEC
#define BLU_SPEC 1
#endif

// ---- list heads of the single-matrix kernel in LDS (k_pivot_loop; the batch kernel has none) ---------------
// Heads and tails of the column count lists 0..MC_HEADS-1 (cflink/cblink[m + k]) live in LDS: every search
// starts at them and every batched list update ends at them, so each pivot saves two dependent global round
// trips.  WRITE-THROUGH: global memory is always current.  Everything the general pivot paths do goes straight
// to global memory; they raise `dirty` and wave 0 reloads the copies before the next search.
// (A direct-mapped LDS cache of whole line records -- begin, length, capacity, maximum, links -- was built and
// measured in round 2: 66 % hits on the banded C3 basis, but the per-lane look-ups in the gathers of the pivot
// set-up and the write-through in every line update cost more than the hits saved: pivot loop +12 %.  Removed.)
// The same struct carries what the NEXT search may reuse of the pivot just finished ("prev"): 70 % of the
// candidate columns of a search on a banded basis are columns of the previous pivot row, whose new length,
// begin and maximum are still in the Fast arrays (slot-indexed: tNew, tB, tMx) and whose membership is still
// in hCol; with their new forward links (pFl, recorded by the list wave) the walk over them needs no global
// round trip at all, and a column left with ONE entry by a singleton-column pivot has that entry in e1i/e1v.
#define MC_HEADS 128
#define MC_PREV 64 // slots of the previous pivot row that can be reused (rows shorter than this)
struct Mc {
    int hf[MC_HEADS], hb[MC_HEADS];
    int dirty;
    int prevValid; // 1: the Fast arrays + pFl/e1 describe the columns moved by the previous pivot; slots >= prevBase
    int prevBase, pad0;
    int pFl[MC_PREV];
    int e1i[MC_PREV]; // row index of the single remaining entry, -1 = not known
    double e1v[MC_PREV];
    int e1rb[MC_PREV], e1rl[MC_PREV], e1rc[MC_PREV]; // (begin, len, cap) of that row, loaded by the wave that updated the column
};

struct Fast {
    int kind;  // 0 none (general paths), 1 pivot_small, 2 pivot_singleton_col
    int where; // singleton col: slot of the pivot column in the (unswapped) pivot row
    int anycancel;
    int ncand;
    int cJ[KCMAX], cNz[KCMAX], cB[KCMAX], cL[KCMAX], cOff[KCMAX + 1];
    double cMx[KCMAX];
    // early search: hand-over from the list wave (unlinked runs: predecessor -> first unmoved successor)
    // and the result kept for the next search
    int ewValid, ewNsr; // ewValid: 1 = a column singleton (early_search), 2 = a whole search (spec_finish), key in spKey
    int spOk, spLastNz; // speculative search of the next pivot: walk complete / count of its last candidate
    long long spKey;    // its winner: cost * 256 + flat position
    double tMx[64]; // new maximum of the first 64 columns of the pivot row (the next search reuses rows < 64 only)
    // pivot column, pivot at slot 0 (kind 1), with the (begin,len,cap) of each row
    int pcI[PCMAX], prB[PCMAX], prL[PCMAX], prC[PCMAX], rNew[PCMAX], rKept[PCMAX], rDst[PCMAX];
    double pcV[PCMAX];
    // pivot row, pivot column at slot 0 (kind 1), with the (begin,len,cap) of each column
    int tJ[PRMAX], tB[PRMAX], tL[PRMAX], tC[PRMAX], tNew[PRMAX];
#if BLU_CFG_BATCH
    int tFl[1], tBl[1], tLnk; // (single-matrix kernel only)
#else
    int tFl[64], tBl[64], tLnk; // tLnk: the count-list links of the row's columns (slots < 64) were loaded with their metadata
#endif
    double tX[PRMAX]; // pivot-row value of the column
    unsigned long long tM[PRMAX];
    // staged candidate entries
    int sI[STGMAX], sB[STGMAX], sL[STGMAX], sC[STGMAX];
    double sV[STGMAX];
    unsigned long long hRow[HROW]; // (row index << 32) | position, ~0 = empty: one LDS read per probe
    unsigned long long hCol[HCOL]; // (column index << 32) | slot in tJ, ~0 = empty
    unsigned long long kg[2][KGMAX]; // key -> lanes with that key, all zero between list moves
};

