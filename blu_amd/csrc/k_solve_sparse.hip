// k_solve_sparse.hip -- solve_sparse for a fresh factorization (nforrest = 0): the hypersparse
// Gilbert-Peierls solve of src/lu/solve_sparse.rs:11-360 with src/lu/solve_symbolic.rs:19-40,
// src/lu/dfs.rs:25-145 and src/lu/solve_triangular.rs:27-136, one wave per matrix.
//
// The reference's result is more than the solution vector: the pattern ilhs[0..nzlhs) comes out in the
// order the depth-first searches leave it (reverse post-order over each line's entries in STORAGE
// order), and which branch runs (symbolic + sparse substitution, or the sequential sweep) depends on
// nnz of the intermediate vector against sparse_thres*m.  Both are reproduced: the four graphs below
// present the factors with the entry order build_factors gives them (build_factors.rs:229-384):
//   GL   L column of row i   = stage column pinv[i], row indices, production order       (forward, 1st)
//   GU   U column under row i = canonical column pinv[i], rows ascending in pivot order   (forward, 2nd)
//   GW   U row under column j = stage row qinv[j], production order; entries in columns
//        without a pivot (rank-deficient factorization) are absent                        (transposed, 1st)
//   GLt  L row i, ascending in the pivot order of its columns: built by k_build_lt        (transposed, 2nd)
// The depth-first search is inherently serial: lane 0 runs it.  The numerical substitution visits the
// columns in that topological order; inside one column the 64 lanes update distinct entries, so every
// entry sees its updates in the reference's order and the values are bit-identical.
#include "blu_dev.h"

struct SparseWs {
    int *marked;     // m, 0 <= marked[i] <= marker (src/lu/lu.rs:164)
    int *psym;       // m: DFS stack from the front, topological order from the back (pattern_symb)
    int *pat;        // m: pattern of the intermediate vector
    int *pstack;     // m: position stack of the DFS (the reference keeps it in work1)
    int *estack;     // m: end of the adjacency list of each stack level (dfs_reach_wave)
    double *work;    // m, all zero between calls (work0)
    double *xlhs;    // m, all zero between calls
    int *ilhs;       // m: OUT pattern of the solution
    double *xval;    // m: OUT xval[n] = lhs[ilhs[n]]
    long long *out;  // OUT [0] nzlhs, [1] l_flops, [2] u_flops, [3] branch taken (1 sparse, 2 sequential)
    int *lt_ptr;     // m+1   row-wise L (k_build_lt)
    int *lt_idx;     // l_nz  row index of the pivot of the entry's column
    double *lt_val;  // l_nz
    int *lt_cur;     // m     scratch of k_build_lt
};

// ---- the four graphs ------------------------------------------------------------------------------
struct GraphL {
    static constexpr bool FILTER = false;
    gcint_p pinv, lbeg, lidx;
    gdouble_p lval;
    __device__ __forceinline__ int begin(int i) const { return lbeg[pinv[i]]; }
    __device__ __forceinline__ int end(int i) const { return lbeg[pinv[i] + 1]; }
    __device__ __forceinline__ int node(int p) const { return lidx[p]; }
    __device__ __forceinline__ double val(int p) const { return lval[p]; }
    __device__ __forceinline__ double pivot(int) const { return 1.0; }
};
struct GraphU {
    static constexpr bool FILTER = false;
    gcint_p pinv, prow;
    GPTR(const long long) colptr;
    GPTR(const long long) rowidx;
    GPTR(const double) value;
    __device__ __forceinline__ int begin(int i) const { return (int)colptr[pinv[i]]; }
    __device__ __forceinline__ int end(int i) const { return (int)colptr[pinv[i] + 1] - 1; } // pivot last
    __device__ __forceinline__ int node(int p) const { return prow[(int)rowidx[p]]; }
    __device__ __forceinline__ double val(int p) const { return value[p]; }
    __device__ __forceinline__ double pivot(int i) const { return value[colptr[pinv[i] + 1] - 1]; } // row_pivot[i]
};
struct GraphW {
    static constexpr bool FILTER = true;
    gcint_p qinv, ubeg, uidx;
    gdouble_p uval;
    GPTR(const long long) colptr;
    GPTR(const double) value;
    int rank;
    __device__ __forceinline__ int begin(int j) const
    {
        const int k = qinv[j];
        return k < rank ? ubeg[k] : 0;
    }
    __device__ __forceinline__ int end(int j) const
    {
        const int k = qinv[j];
        return k < rank ? ubeg[k + 1] : 0;
    }
    __device__ __forceinline__ int node(int p) const
    {
        const int j = uidx[p];
        return qinv[j] < rank ? j : -1; // build_factors.rs:323
    }
    __device__ __forceinline__ double val(int p) const { return uval[p]; }
    __device__ __forceinline__ double pivot(int j) const { return value[colptr[qinv[j] + 1] - 1]; } // col_pivot[j]
};
struct GraphLt {
    static constexpr bool FILTER = false;
    const int *ptr, *idx;
    const double *v;
    __device__ __forceinline__ int begin(int i) const { return ptr[i]; }
    __device__ __forceinline__ int end(int i) const { return ptr[i + 1]; }
    __device__ __forceinline__ int node(int p) const { return idx[p]; }
    __device__ __forceinline__ double val(int p) const { return v[p]; }
    __device__ __forceinline__ double pivot(int) const { return 1.0; }
};

// dfs (dfs.rs:49-145, after CSparse): reach of node i in topological order into xi[newtop..top), ONE lane
template <class G>
__device__ __forceinline__ int dfs_reach(const G &g, int i, int top, int *xi, int *pstack, int *marked, int M)
{
    int head = 0;
    xi[0] = i;
    while (head >= 0) {
        i = xi[head];
        if (marked[i] != M) { // node i has not been visited
            marked[i] = M;
            pstack[head] = g.begin(i);
        }
        bool done = true;
        const int e = g.end(i);
        for (int p = pstack[head]; p < e; p++) { // continue the search at node i
            const int inext = g.node(p);
            if (inext < 0 || marked[inext] == M) continue;
            pstack[head] = p + 1;
            xi[++head] = inext;
            done = false;
            break;
        }
        if (done) { // node i has no unvisited neighbours
            head--;
            xi[--top] = i;
        }
    }
    return top;
}

// The same search by the whole wave (dfs.rs:49-145: same visiting order, same output).  The lone-lane form above
// pays a dependent global load or two for every ENTRY it looks at (~1 us each); here the 64 lanes look at 64 entries
// of the adjacency list at once -- neighbour, its mark, and where its own list begins and ends, all issued together
// -- and the first unvisited one (lowest lane = the one the sequential scan would stop at; marks only change through
// this search) is descended into with its list bounds already in hand: two to four dependent loads per node VISIT
// instead of per entry.  The top of the stack lives in registers, the levels below in an LDS ring (write-through to
// xi / pstack / estack in global memory, which the ring is refilled from, 64 levels at a time, when a deep stack
// unwinds past it).
#define DFS_RING 2048
struct DfsRing {
    int i[DFS_RING], p[DFS_RING], e[DFS_RING];
};
template <class G>
__device__ __forceinline__ int dfs_reach_wave(const G &g, int i0, int top, int *xi, int *pstack, int *estack, int *marked, int M, DfsRing *R)
{
    const int lane = lane_id();
    int head = 0, lo = 0; // ring slots of the levels lo..head-1 are valid
    int i = i0, p = g.begin(i0), e = g.end(i0);
    if (lane == 0) {
        marked[i0] = M;
        xi[0] = i0;
    }
    wave_mem_sync();
    for (;;) {
        // first unvisited neighbour of i at or behind position p
        int child = -1, cb = 0, ce = 0, cpos = e;
        for (int q0 = p; q0 < e; q0 += 64) {
            const int q = q0 + lane;
            int inext = -1, mk = M, nb = 0, ne = 0;
            if (q < e) {
                inext = g.node(q);
                if (inext >= 0) {
                    mk = marked[inext];
                    nb = g.begin(inext);
                    ne = g.end(inext);
                }
            }
            const unsigned long long b = __ballot(inext >= 0 && mk != M);
            if (b) {
                const int l = __ffsll((long long)b) - 1;
                child = wave_bcast_i(inext, l);
                cb = wave_bcast_i(nb, l);
                ce = wave_bcast_i(ne, l);
                cpos = q0 + l + 1;
                // marks are only ever set during a search: if nothing else in the rest of the list is unvisited now,
                // nothing will be when the search comes back -- the level is finished, no second look
                if ((b & ~(1ull << l)) == 0ull && q0 + 64 >= e) cpos = e;
                break;
            }
        }
        if (child >= 0) { // descend: level `head` continues at cpos later
            if (lane == 0) {
                pstack[head] = cpos;
                estack[head] = e;
                xi[head + 1] = child;
                marked[child] = M;
                R->i[head & (DFS_RING - 1)] = i;
                R->p[head & (DFS_RING - 1)] = cpos;
                R->e[head & (DFS_RING - 1)] = e;
            }
            if (head - (DFS_RING - 1) > lo) lo = head - (DFS_RING - 1);
            head++;
            i = child;
            p = cb;
            e = ce;
            wave_mem_sync();
        } else { // node i has no unvisited neighbours
            top--;
            if (lane == 0) xi[top] = i;
            head--;
            if (head < 0) break;
            if (head < lo) { // the stack has unwound past the ring: refill 64 levels
                const int h = head - lane;
                if (h >= 0) {
                    R->i[h & (DFS_RING - 1)] = xi[h];
                    R->p[h & (DFS_RING - 1)] = pstack[h];
                    R->e[h & (DFS_RING - 1)] = estack[h];
                }
                lo = head - 63 < 0 ? 0 : head - 63;
                wave_mem_sync();
            }
            i = R->i[head & (DFS_RING - 1)];
            p = R->p[head & (DFS_RING - 1)];
            e = R->e[head & (DFS_RING - 1)];
        }
    }
    wave_mem_sync();
    return top;
}

// solve_symbolic (solve_symbolic.rs:19-40), whole wave; returns top.  The roots are taken 64 at a time: one that is
// marked when its chunk is loaded stays marked (marks are only ever set during a symbolic phase), the others are
// looked at again when their turn comes.
template <class G>
__device__ __forceinline__ int solve_symbolic(const G &g, int m, int nrhs, const int *irhs, const SparseWs &W, int M, DfsRing *R)
{
    const int lane = lane_id();
    int top = m;
    for (int n0 = 0; n0 < nrhs; n0 += 64) {
        const int n = n0 + lane;
        const int ir = n < nrhs ? irhs[n] : 0;
        unsigned long long todo = __ballot(n < nrhs && W.marked[ir] != M);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int i = wave_bcast_i(ir, l);
            if (W.marked[i] != M) top = dfs_reach_wave(g, i, top, W.psym, W.pstack, W.estack, W.marked, M, R);
        }
    }
    wave_mem_sync();
    return top;
}

// solve_triangular (solve_triangular.rs:27-136), whole wave: columns in the given order, lanes over the
// entries of a column.  Returns nz; pattern[0..nz) = nonzeros kept.
template <bool PIV, class G>
__device__ __forceinline__ int solve_triangular(const G &g, int nz_symb, const int *psym, double droptol, double *lhs, int *pattern,
                                                long long &flops)
{
    // The symbolic reach can be far longer than the numerical one (on the banded LP bases a unit vector reaches
    // tens of thousands of nodes through the U chain while the values fall below droptol after a few dozen), and a
    // dependent load per position just to find a zero is ~0.7 us: 64 positions are looked at together, the nonzero
    // ones are worked in order, and after each of them the rest of the chunk is read again (it may have been filled).
    // What does not depend on the values -- list bounds and pivot of every position of the chunk, and the first 64
    // entries of the NEXT nonzero position -- is fetched ahead, so that a column costs the read-modify-write of its
    // targets and the re-read of the chunk, not five more dependent loads.
    const int lane = lane_id();
    int nz = 0;
    for (int n0 = 0; n0 < nz_symb; n0 += 64) {
        const int n = n0 + lane;
        const bool in = n < nz_symb;
        const int ip = in ? psym[n] : 0;
        double xl = in ? lhs[ip] : 0.0;
        const int bl = in ? g.begin(ip) : 0, el = in ? g.end(ip) : 0;
        const double pl = (PIV && in) ? g.pivot(ip) : 1.0;
        unsigned long long todo = __ballot(xl != 0.0);
        // entries of the first candidate
        int cand = todo ? __ffsll((long long)todo) - 1 : -1;
        int ci = -1;
        double cv = 0.0;
        if (cand >= 0) {
            const int cb = wave_bcast_i(bl, cand), ce = wave_bcast_i(el, cand);
            if (cb + lane < ce) {
                ci = g.node(cb + lane);
                cv = g.val(cb + lane);
            }
        }
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            const int ipivot = wave_bcast_i(ip, l);
            const int b = wave_bcast_i(bl, l), e = wave_bcast_i(el, l);
            double x = wave_bcast_d(xl, l);
            // this column's first entries: prefetched if the guess was right
            int i0 = ci;
            double v0 = cv;
            if (l != cand) {
                i0 = -1;
                v0 = 0.0;
                if (b + lane < e) {
                    i0 = g.node(b + lane);
                    v0 = g.val(b + lane);
                }
            }
            // guess the next position (the next one that is nonzero NOW) and fetch its first entries
            const unsigned long long rest = todo & ~((2ull << l) - 1ull);
            cand = (l < 63 && rest) ? __ffsll((long long)rest) - 1 : -1;
            ci = -1;
            cv = 0.0;
            if (cand >= 0) {
                const int cb = wave_bcast_i(bl, cand), ce = wave_bcast_i(el, cand);
                if (cb + lane < ce) {
                    ci = g.node(cb + lane);
                    cv = g.val(cb + lane);
                }
            }
            if (PIV) {
                x = x / wave_bcast_d(pl, l);
                wave_mem_sync();
                if (lane == 0) lhs[ipivot] = x;
                flops++;
            }
            if (b + lane < e && i0 >= 0) lhs[i0] = __dsub_rn(lhs[i0], __dmul_rn(x, v0));
            for (int p = b + 64 + lane; p < e; p += 64) {
                const int i = g.node(p);
                if (i >= 0) lhs[i] = __dsub_rn(lhs[i], __dmul_rn(x, g.val(p)));
            }
            if (G::FILTER) { // flop count = entries present
                flops += __popcll(__ballot(b + lane < e && i0 >= 0));
                for (int p = b + 64; p < e; p += 64) {
                    const int q = p + lane;
                    flops += __popcll(__ballot(q < e && g.node(q) >= 0));
                }
            } else {
                flops += e - b;
            }
            wave_mem_sync();
            if (fabs(x) > droptol) {
                if (lane == 0) pattern[nz] = ipivot;
                nz++;
            } else if (lane == 0) {
                lhs[ipivot] = 0.0;
            }
            wave_mem_sync();
            xl = (in && lane > l) ? lhs[ip] : 0.0;
            todo = __ballot(xl != 0.0);
        }
    }
    return nz;
}

// The sequential branches of the sparse solves (lu/solve_sparse.rs:159-179, 307-334): a sweep over the whole
// pivot sequence, k descending, that does something only where the vector is nonzero.  One dependent global load
// per k is ~1 us on a lone wave -- 0.1 s for m = 100 000 even if the vector is nearly empty -- so the sweep takes
// 64 sequence positions at a time: their entries are loaded together, the zero ones cost nothing, and after
// every step that was worked the remaining positions of the chunk are re-read (the step may have filled them).
// The order of the steps, and with it every sum, is the reference's.
//   seq(k)      vector index of sequence position k
//   step(k, i, x)  work one position whose entry x = vec[i] is nonzero (uniform arguments); may modify vec
template <class Seq, class Step>
__device__ __forceinline__ void sweep_nonzeros_desc(int len, const double *vec, Seq seq, Step step)
{
    const int lane = lane_id();
    for (int k0 = len - 1; k0 >= 0; k0 -= 64) {
        const int k = k0 - lane;
        const int i = k >= 0 ? seq(k) : 0;
        double x = k >= 0 ? vec[i] : 0.0;
        unsigned long long todo = __ballot(x != 0.0);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1; // lowest lane = highest position
            step(k0 - l, wave_bcast_i(i, l), wave_bcast_d(x, l));
            wave_mem_sync();
            x = k >= 0 ? vec[i] : 0.0;
            todo = __ballot(x != 0.0) & ~((2ull << l) - 1ull);
            if (l == 63) todo = 0ull;
        }
    }
}

__global__ void __launch_bounds__(64) k_solve_sparse(DevLU *Ds, FinishOut *Os, SparseWs W, int nrhs, const int *irhs, const double *xrhs,
                                                     int trans, int marker, int nz_sparse)
{
    __shared__ DfsRing dfs_ring;
    const DevG D(Ds[0]);
    const FinishOut &O = Os[0];
    const int lane = lane_id();
    const int m = D.m;
    const int rank = D.s->rank;
    const double droptol = D.droptol;
    long long l_flops = 0, u_flops = 0;
    int nz = 0, branch = 1;
    typedef GPTR(const long long) gcll;
    typedef GPTR(const double) gcd;

    if (trans) {
        // ---- transposed system (solve_sparse.rs:51-179): U', then L'
        const GraphW GW{D.qinv, D.ubeg, D.uidx, D.uval, (gcll)O.u_colptr, (gcd)O.u_value, rank};
        int M = marker + 1;
#ifdef BLU_PROFILE
        const long long tp0 = (long long)__builtin_amdgcn_s_memtime();
#endif
        int top = solve_symbolic(GW, m, nrhs, irhs, W, M, &dfs_ring);
#ifdef BLU_PROFILE
        const long long tp1 = (long long)__builtin_amdgcn_s_memtime();
#endif
        for (int n = lane; n < nrhs; n += 64) W.work[irhs[n]] = xrhs[n];
        wave_mem_sync();
        nz = solve_triangular<true>(GW, m - top, W.psym + top, droptol, W.work, W.pat, u_flops);
#ifdef BLU_PROFILE
        if (lane == 0)
            printf("solve_sparse T, U' part: reach %d nodes in %.0f us (%.2f us each), numeric %d kept, %lld flops in %.0f us\n", m - top,
                   (tp1 - tp0) / 2100.0, (tp1 - tp0) / 2100.0 / (m - top), nz, u_flops, ((long long)__builtin_amdgcn_s_memtime() - tp1) / 2100.0);
#endif
        // permute into xlhs; the pattern goes from column to row indices (:95-106)
        M = marker + 2;
        for (int n = lane; n < nz; n += 64) {
            const int j = W.pat[n], i = D.prow[D.qinv[j]]; // pmap[j]
            W.pat[n] = i;
            W.xlhs[i] = W.work[j];
            W.work[j] = 0.0;
            W.marked[i] = M;
        }
        wave_mem_sync();
        const GraphLt GT{W.lt_ptr, W.lt_idx, W.lt_val};
        if (nz <= nz_sparse) {
            M = marker + 3;
            top = solve_symbolic(GT, m, nz, W.pat, W, M, &dfs_ring);
            nz = solve_triangular<false>(GT, m - top, W.psym + top, droptol, W.xlhs, W.ilhs, l_flops);
        } else { // sequential solve with L' (:159-179)
            branch = 2;
            const int nin = nz;
            (void)nin;
            nz = 0;
            sweep_nonzeros_desc(m, W.xlhs, [&](int k) { return D.prow[k]; }, [&](int, int ipivot, double x) {
                const int b = GT.begin(ipivot), e = GT.end(ipivot);
                for (int p = b + lane; p < e; p += 64) {
                    const int i = GT.node(p);
                    W.xlhs[i] = __dsub_rn(W.xlhs[i], __dmul_rn(x, GT.val(p)));
                }
                l_flops += e - b;
                wave_mem_sync();
                if (fabs(x) > droptol) {
                    if (lane == 0) W.ilhs[nz] = ipivot;
                    nz++;
                } else if (lane == 0) {
                    W.xlhs[ipivot] = 0.0;
                }
            });
        }
    } else {
        // ---- forward system (solve_sparse.rs:180-346): L, then U
        const GraphL GL{D.pinv, D.lbeg, D.lidx, D.lval};
        int M = marker + 1;
        int top = solve_symbolic(GL, m, nrhs, irhs, W, M, &dfs_ring);
        const int nz_symb = m - top;
        for (int n = lane; n < nrhs; n += 64) W.work[irhs[n]] = xrhs[n];
        wave_mem_sync();
        nz = solve_triangular<false>(GL, nz_symb, W.psym + top, droptol, W.work, W.pat, l_flops);
        // unmark cancellation (:227-243)
        if (nz < nz_symb && lane == 0) {
            int t = top, n = 0;
            while (n < nz) {
                const int i = W.psym[t];
                if (i == W.pat[n]) n++;
                else W.marked[i] -= 1;
                t++;
            }
            while (t < m) {
                W.marked[W.psym[t]] -= 1;
                t++;
            }
        }
        wave_mem_sync();
        const GraphU GU{D.pinv, D.prow, (gcll)O.u_colptr, (gcll)O.u_rowidx, (gcd)O.u_value};
        if (nz <= nz_sparse) {
            M = marker + 2;
            top = solve_symbolic(GU, m, nz, W.pat, W, M, &dfs_ring);
            nz = solve_triangular<true>(GU, m - top, W.psym + top, droptol, W.work, W.ilhs, u_flops);
            // permute into xlhs; the pattern goes from row to column indices (:299-306)
            for (int n = lane; n < nz; n += 64) {
                const int i = W.ilhs[n], j = D.pcol[D.pinv[i]]; // qmap[i]
                W.ilhs[n] = j;
                W.xlhs[j] = W.work[i];
                W.work[i] = 0.0;
            }
            wave_mem_sync();
        } else { // sequential solve with U (:307-334)
            branch = 2;
            nz = 0;
            sweep_nonzeros_desc(m, W.work, [&](int k) { return D.prow[k]; }, [&](int k, int ipivot, double w) {
                const int jpivot = D.pcol[k];
                const double x = w / GU.pivot(ipivot);
                if (lane == 0) W.work[ipivot] = 0.0;
                const int b = GU.begin(ipivot), e = GU.end(ipivot);
                for (int p = b + lane; p < e; p += 64) {
                    const int i = GU.node(p);
                    W.work[i] = __dsub_rn(W.work[i], __dmul_rn(x, GU.val(p)));
                }
                u_flops += e - b;
                if (fabs(x) > droptol) {
                    if (lane == 0) {
                        W.ilhs[nz] = jpivot;
                        W.xlhs[jpivot] = x;
                    }
                    nz++;
                }
            });
        }
    }
    // hand the solution out in compressed form and restore the all-zero invariant of xlhs
    for (int n = lane; n < nz; n += 64) {
        const int j = W.ilhs[n];
        W.xval[n] = W.xlhs[j];
        W.xlhs[j] = 0.0;
    }
    if (lane == 0) {
        W.out[0] = nz;
        W.out[1] = l_flops;
        W.out[2] = u_flops;
        W.out[3] = branch;
    }
}

// Row-wise L in the reference's order (build_factors.rs:243-274): row i holds, for every column it has
// an entry in, the row index of that column's pivot, ascending in the columns' pivot order.
// One workgroup: count, scan, scatter (unordered), then sort each short row by pivot position.
__global__ void __launch_bounds__(1024) k_build_lt(DevLU *Ds, SparseWs W)
{
    const DevG D(Ds[0]);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = D.m;
    __shared__ int carry;
    __shared__ int part[1024];
    for (int i = tid; i < m; i += nt) W.lt_cur[i] = 0;
    if (tid == 0) carry = 0;
    __syncthreads();
    const int lend = D.lbeg[m];
    for (int p = tid; p < lend; p += nt) atomicAdd(&W.lt_cur[D.lidx[p]], 1);
    __syncthreads();
    // exclusive scan of the row counts, 1024 at a time
    for (int base = 0; base < m; base += nt) {
        const int i = base + tid;
        const int c = i < m ? W.lt_cur[i] : 0;
        part[tid] = c;
        __syncthreads();
        for (int o = 1; o < nt; o <<= 1) {
            const int v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const int excl = carry + part[tid] - c;
        if (i < m) {
            W.lt_ptr[i] = excl;
            W.lt_cur[i] = excl;
        }
        __syncthreads();
        if (tid == nt - 1) carry += part[tid];
        __syncthreads();
    }
    if (tid == 0) W.lt_ptr[m] = carry;
    __syncthreads();
    // scatter: entry (row i, stage k) -> row i, keyed by k for now
    for (int k = tid; k < m; k += nt) {
        for (int p = D.lbeg[k]; p < D.lbeg[k + 1]; p++) {
            const int q = atomicAdd(&W.lt_cur[D.lidx[p]], 1);
            W.lt_idx[q] = k;
            W.lt_val[q] = D.lval[p];
        }
    }
    __syncthreads();
    // sort every row by k (rows are short), then replace k by the pivot row of stage k
    for (int i = tid; i < m; i += nt) {
        const int b = W.lt_ptr[i], e = W.lt_ptr[i + 1];
        for (int p = b + 1; p < e; p++) {
            const int k = W.lt_idx[p];
            const double v = W.lt_val[p];
            int q = p - 1;
            while (q >= b && W.lt_idx[q] > k) {
                W.lt_idx[q + 1] = W.lt_idx[q];
                W.lt_val[q + 1] = W.lt_val[q];
                q--;
            }
            W.lt_idx[q + 1] = k;
            W.lt_val[q + 1] = v;
        }
        for (int p = b; p < e; p++) W.lt_idx[p] = D.prow[W.lt_idx[p]];
    }
}
