/* orc_update.c -- TEST INFRASTRUCTURE (see blu_oracle.h).
 *
 * solve_for_update + Forrest-Tomlin update: the INTENDED algorithm of /root/reference's
 *   src/lu/solve_for_update.rs:12-455, src/lu/update.rs:23-959,
 *   src/solve_for_update.rs:73-119, src/update.rs:49-55, BLU::solve_for_update / BLU::update src/blu.rs:257-335.
 *
 * >>> NOT REFERENCE-PINNED, and deliberately NOT a faithful restatement. <<<  The reference's update path is
 * defective as written: an FT update either panics or leaves a factorization whose solves
 * are wrong.  This file restates what the code documents itself as doing (comments update.rs:378-387,
 * 467-483, 609-666, 695-709, 750-755; solve_for_update.rs:75-80, 122-135, 342-355), statement by statement
 * in the reference's order, with exactly these repairs, each marked "FIX Dn" where it is made:
 *   D7  update.rs:422-423, 877-878  row_reach / col_reach of an FT update are the one-element lists
 *                                   (ipivot) / (jpivot), not zero vectors of that LENGTH
 *   D8  update.rs:634-635           row_reach / col_reach of the symmetric-permutation case hold nreach entries
 *   D9  update.rs:797               permute() receives the nswap + 1 path nodes
 *   D10 update.rs:69                the breadth-first search runs until the queue is empty (the Rust range
 *                                   `0..tail` is evaluated once, so only the root was explored)
 *   D11 blu.rs:271-283              BLU::solve_for_update passes None for the solution when want_solution == 0
 *   D13 lu.rs:184-193               eta_row does not alias r_begin (orc_internal.h)
 *   D5' update.rs:532               w_end[m] (file capacity) is refreshed from w_mem on entry, so a Reallocate
 *                                   answered by the caller ends the loop (the reference would loop for ever)
 * Validation is by residual and by refactorize-and-compare (tests/test_update_oracle.py), not by parity. */
#include "orc_internal.h"

#define GAP (-1)
#define FLIP(i) (-(i) - 1)

/* update.rs:26-42 */
static lu_int find(lu_int j, const lu_int *index, lu_int start, lu_int end)
{
    if (end >= 0) {
        while (start < end && index[start] != j) start++;
        return start;
    }
    while (index[start] != j && index[start] >= 0) start++;
    return index[start] == j ? start : end;
}

/* bfs_path -- update.rs:51-105 */
static lu_int bfs_path(lu_int m, lu_int j0, const lu_int *begin, const lu_int *end, const lu_int *index,
                       lu_int *jlist, lu_int *marked, lu_int *queue)
{
    lu_int j = -1, tail = 1, top = m;
    int found = 0;
    queue[0] = j0;
    for (lu_int front = 0; front < tail && !found; front++) { /* FIX D10 */
        j = queue[front];
        for (lu_int pos = begin[j]; pos < end[j]; pos++) {
            lu_int k = index[pos];
            if (k == j0) {
                found = 1;
                break;
            }
            if (marked[k] >= 0) {     /* not in queue yet */
                marked[k] = FLIP(j);  /* parent[k] = j */
                queue[tail++] = k;    /* append to queue */
            }
        }
    }
    if (found) { /* build path (j0,..,j) */
        while (j != j0) {
            jlist[--top] = j;
            j = FLIP(marked[j]); /* go to parent */
            ORC_ASSERT(j >= 0);
        }
        jlist[--top] = j0;
    }
    for (lu_int pos = 0; pos < tail; pos++) marked[queue[pos]] = 0; /* reset */
    return top;
}

/* compress_packed -- update.rs:115-162 */
static lu_int compress_packed(lu_int m, lu_int *begin, lu_int *index, double *value)
{
    lu_int nz = 0;
    const lu_int end = begin[m];
    for (lu_int i = 0; i < m; i++) { /* mark the beginning of each nonempty line */
        lu_int p = begin[i];
        if (index[p] == GAP) {
            begin[i] = 0;
        } else {
            ORC_ASSERT(index[p] > GAP);
            begin[i] = index[p];     /* temporarily store index here */
            index[p] = GAP - i - 1;  /* mark beginning of line i */
        }
    }
    ORC_ASSERT(index[0] == GAP);
    lu_int i = -1, put = 1;
    for (lu_int get = 1; get < end; get++) {
        if (index[get] > GAP) { /* shift entry of line i */
            ORC_ASSERT(i >= 0);
            index[put] = index[get];
            value[put++] = value[get];
            nz++;
        } else if (index[get] < GAP) { /* beginning of line i */
            ORC_ASSERT(i == -1);
            i = GAP - index[get] - 1;
            index[put] = begin[i]; /* store back */
            begin[i] = put;
            value[put++] = value[get];
            nz++;
        } else if (i >= 0) { /* line i ended at a gap */
            i = -1;
            index[put++] = GAP;
        }
    }
    ORC_ASSERT(i == -1);
    begin[m] = put;
    return nz;
}

/* permute -- update.rs:176-314.  jlist holds nswap + 1 nodes. */
static void permute(orc_lu *lu, const lu_int *jlist, lu_int nswap)
{
    lu_int *pmap = PMAP(lu), *qmap = QMAP(lu);
    lu_int *u_begin = lu->u_begin, *w_begin = lu->w_begin, *w_end = lu->w_end, *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    double *col_pivot = lu->col_pivot, *row_pivot = lu->row_pivot;
    lu_int *u_index = lu->u_index, *w_index = lu->w_index;
    double *u_value = lu->u_value, *w_value = lu->w_value;

    const lu_int j0 = jlist[0], jn = jlist[nswap];
    const lu_int i0 = pmap[j0], in_ = pmap[jn];
    ORC_ASSERT(nswap >= 1);
    ORC_ASSERT(qmap[i0] == j0);
    ORC_ASSERT(qmap[in_] == jn);
    ORC_ASSERT(row_pivot[i0] == 0.0);
    ORC_ASSERT(col_pivot[j0] == 0.0);

    /* Update row file */
    lu_int begin = w_begin[jn], end = w_end[jn]; /* keep for later */
    const double piv = col_pivot[jn];
    for (lu_int n = nswap; n > 0; n--) {
        lu_int j = jlist[n], jprev = jlist[n - 1];
        /* When row i was indexed by jprev in the row file before, then it is indexed by j now. */
        w_begin[j] = w_begin[jprev];
        w_end[j] = w_end[jprev];
        orc_list_swap(w_flink, w_blink, j, jprev);
        /* That row must have an entry in column j because (jprev,j) is an edge in the augmenting path.
         * This entry becomes a pivot element.  If jprev is not the first node in the path, then it has
         * an entry in the row (the old pivot) which becomes an off-diagonal entry now. */
        lu_int where_ = find(j, w_index, w_begin[j], w_end[j]);
        ORC_ASSERT(where_ < w_end[j]);
        if (n > 1) {
            ORC_ASSERT(jprev != j0);
            w_index[where_] = jprev;
            col_pivot[j] = w_value[where_];
            ORC_ASSERT(col_pivot[j] != 0.0);
            w_value[where_] = col_pivot[jprev];
        } else {
            ORC_ASSERT(jprev == j0);
            col_pivot[j] = w_value[where_];
            ORC_ASSERT(col_pivot[j] != 0.0);
            w_end[j]--;
            w_index[where_] = w_index[w_end[j]];
            w_value[where_] = w_value[w_end[j]];
        }
        lu->min_pivot = fmin(lu->min_pivot, fabs(col_pivot[j]));
        lu->max_pivot = fmax(lu->max_pivot, fabs(col_pivot[j]));
    }
    w_begin[j0] = begin;
    w_end[j0] = end;
    lu_int where_ = find(j0, w_index, w_begin[j0], w_end[j0]);
    ORC_ASSERT(where_ < w_end[j0]);
    w_index[where_] = jn;
    col_pivot[j0] = w_value[where_];
    ORC_ASSERT(col_pivot[j0] != 0.0);
    w_value[where_] = piv;
    lu->min_pivot = fmin(lu->min_pivot, fabs(col_pivot[j0]));
    lu->max_pivot = fmax(lu->max_pivot, fabs(col_pivot[j0]));

    /* Update column file */
    begin = u_begin[i0]; /* keep for later */
    for (lu_int n = 0; n < nswap; n++) {
        lu_int i = pmap[jlist[n]], inext = pmap[jlist[n + 1]];
        /* When column j indexed by inext in the column file before, then it is indexed by i now. */
        u_begin[i] = u_begin[inext];
        /* That column must have an entry in row i because there is an edge in the augmenting path.  This
         * entry becomes a pivot element.  There is also an entry in row inext (the old pivot), which now
         * becomes an off-diagonal entry. */
        where_ = find(i, u_index, u_begin[i], -1);
        ORC_ASSERT(where_ >= 0);
        u_index[where_] = inext;
        row_pivot[i] = u_value[where_];
        ORC_ASSERT(row_pivot[i] != 0.0);
        u_value[where_] = row_pivot[inext];
    }
    u_begin[in_] = begin;
    where_ = find(in_, u_index, u_begin[in_], -1);
    ORC_ASSERT(where_ >= 0);
    row_pivot[in_] = u_value[where_];
    ORC_ASSERT(row_pivot[in_] != 0.0);
    end = where_;
    while (u_index[end] >= 0) end++;
    u_index[where_] = u_index[end - 1];
    u_value[where_] = u_value[end - 1];
    u_index[end - 1] = -1;

    /* Update row-column mappings */
    for (lu_int n = nswap; n > 0; n--) {
        lu_int j = jlist[n], i = pmap[jlist[n - 1]];
        pmap[j] = i;
        qmap[i] = j;
    }
    pmap[j0] = in_;
    qmap[in_] = j0;
}

/* lu::update -- update.rs:388-959 */
int orc_lu_update(orc_lu *lu, double xtbl)
{
    const lu_int m = lu->m;
    const lu_int nforrest = lu->nforrest;
    lu_int u_nz = lu->u_nz;
    const lu_int pad = lu->pad;
    const double stretch = lu->stretch;
    lu_int *pmap = PMAP(lu), *qmap = QMAP(lu);
    lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    lu_int *u_begin = lu->u_begin, *r_begin = R_BEGIN(lu);
    lu_int *w_begin = lu->w_begin, *w_end = lu->w_end, *w_flink = lu->w_flink, *w_blink = lu->w_blink;
    double *col_pivot = lu->col_pivot, *row_pivot = lu->row_pivot;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = MARKED(lu);
    lu_int *iwork1 = IWORK1(lu), *iwork2 = IWORK1(lu) + m;
    double *work1 = lu->work1;

    const lu_int jpivot = lu->btran_for_update;
    const lu_int ipivot = pmap[jpivot];
    const double oldpiv = col_pivot[jpivot];
    lu_int ipivot_vec = ipivot, jpivot_vec = jpivot; /* FIX D7: the one-element reach lists of an FT update */
    lu_int nreach = 0, *row_reach = NULL, *col_reach = NULL;
    int istriangular;

    ORC_ASSERT(nforrest < m);
    w_end[m] = lu->w_mem; /* FIX D5' */

    /* ---- Prepare: if present, move diagonal element to end of spike (:441-465) */
    double spike_diag = 0.0;
    int have_diag = 0;
    lu_int put = u_begin[m];
    for (lu_int pos = put; u_index[pos] >= 0; pos++) {
        lu_int i = u_index[pos];
        if (i != ipivot) {
            u_index[put] = i;
            u_value[put++] = u_value[pos];
        } else {
            spike_diag = u_value[pos];
            have_diag = 1;
        }
    }
    if (have_diag) {
        u_index[put] = ipivot;
        u_value[put] = spike_diag;
    }
    const lu_int nz_spike = put - u_begin[m]; /* nz excluding diagonal */
    const lu_int nz_roweta = r_begin[nforrest + 1] - r_begin[nforrest];

    /* ---- Compute pivot (:467-513): newpiv = spike_diag - dot(spike, row eta), intersection counted */
    lu_int M = ++lu->marker;
    for (lu_int pos = r_begin[nforrest]; pos < r_begin[nforrest + 1]; pos++) {
        lu_int i = l_index[pos];
        marked[i] = M;
        work1[i] = l_value[pos];
    }
    double newpiv = spike_diag;
    lu_int intersect = 0;
    for (lu_int pos = u_begin[m]; pos < u_begin[m] + nz_spike; pos++) {
        lu_int i = u_index[pos];
        ORC_ASSERT(i != ipivot);
        if (marked[i] == M) {
            newpiv -= u_value[pos] * work1[i];
            intersect++;
        }
    }
    if (newpiv == 0.0 || fabs(newpiv) < lu->abstol) return ORC_ERROR_SINGULAR_UPDATE; /* singularity test */
    const double piverr = fabs(newpiv - xtbl * oldpiv);                                  /* stability measure */

    /* ---- Insert spike (:515-605) */
    lu_int grow = 0; /* bound on file growth */
    for (lu_int pos = u_begin[m]; pos < u_begin[m] + nz_spike; pos++) {
        lu_int i = u_index[pos];
        lu_int j = qmap[i], jnext = w_flink[j];
        if (w_end[j] == w_begin[jnext]) {
            lu_int nz = w_end[j] - w_begin[j];
            grow += nz + 1;                                        /* row including spike entry */
            grow += orc_trunc(stretch * (double)(nz + 1)) + pad;   /* extra room */
        }
    }
    lu_int room = w_end[m] - w_begin[m];
    if (grow > room) {
        lu->addmem_w = grow - room;
        return ORC_REALLOCATE;
    }
    /* remove column jpivot from row file */
    lu_int nz = 0;
    for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) {
        lu_int i = u_index[pos];
        lu_int j = qmap[i];
        lu_int end = w_end[j]--;
        lu_int where_ = find(jpivot, w_index, w_begin[j], end);
        ORC_ASSERT(where_ < end);
        w_index[where_] = w_index[end - 1];
        w_value[where_] = w_value[end - 1];
        nz++;
    }
    u_nz -= nz;
    /* erase column jpivot in column file */
    for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) u_index[pos] = GAP;
    /* set column pointers to spike, chop off diagonal */
    u_begin[ipivot] = u_begin[m];
    u_begin[m] += nz_spike;
    u_index[u_begin[m]++] = GAP;
    /* insert spike into row file */
    for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++) {
        lu_int i = u_index[pos];
        lu_int j = qmap[i], jnext = w_flink[j];
        if (w_end[j] == w_begin[jnext]) {
            nz = w_end[j] - w_begin[j];
            room = 1 + orc_trunc(stretch * (double)(nz + 1)) + pad;
            orc_file_reappend(j, m, w_begin, w_end, w_flink, w_blink, w_index, w_value, room);
        }
        lu_int end = w_end[j]++;
        w_index[end] = jpivot;
        w_value[end] = u_value[pos];
    }
    u_nz += nz_spike;
    /* insert diagonal */
    col_pivot[jpivot] = spike_diag;
    row_pivot[ipivot] = spike_diag;

    /* ---- Test triangularity (:607-818) */
    if (have_diag) {
        /* Spike with a nonzero diagonal: the spiked matrix is symmetrically permuted triangular iff the
         * patterns of the row eta and the spike do not intersect.  row_reach = ipivot followed by the
         * pattern of the row eta (computed by a dfs: topological order), col_reach through qmap. */
        istriangular = intersect == 0;
        if (istriangular) {
            lu->min_pivot = fmin(lu->min_pivot, fabs(newpiv));
            lu->max_pivot = fmax(lu->max_pivot, fabs(newpiv));
            nreach = nz_roweta + 1;
            row_reach = iwork1; /* FIX D8: nreach entries */
            col_reach = iwork2;
            row_reach[0] = ipivot;
            col_reach[0] = jpivot;
            lu_int pos = r_begin[nforrest];
            for (lu_int n = 1; n < nreach; n++) {
                lu_int i = l_index[pos++];
                row_reach[n] = i;
                col_reach[n] = qmap[i];
            }
            lu->nsymperm_total++;
        }
    } else {
        /* Spike with a zero diagonal: only an UNsymmetric permutation can restore triangularity.
         * Part 1: an augmenting path in U[pmap,:] starting from jpivot -> path[top..m-1]. */
        lu_int *path = iwork1, *reach = iwork2;
        double *pstack = work1;
        lu_int top = bfs_path(m, jpivot, w_begin, w_end, w_index, path, marked, reach);
        ORC_ASSERT(top < m - 1);
        ORC_ASSERT(path[top] == jpivot);
        /* Part 2a: reach of every path node (except the last) in U[pmap,:] without the path edges; the
         * combined reach in U[pmap_new,:] is assembled in topological order in reach[rtop..m-1]. */
        istriangular = 1;
        lu_int rtop = m;
        M = ++lu->marker;
        for (lu_int t = top; t < m - 1 && istriangular; t++) {
            lu_int j = path[t], jnext = path[t + 1];
            lu_int where_ = find(jnext, w_index, w_begin[j], w_end[j]);
            ORC_ASSERT(where_ < w_end[j]);
            w_index[where_] = j; /* take out for a moment */
            rtop = orc_dfs(j, w_begin, w_end, w_index, rtop, reach, pstack, marked, M);
            ORC_ASSERT(reach[rtop] == j);
            reach[rtop] = jnext;
            w_index[where_] = jnext; /* restore */
            istriangular = marked[jnext] != M;
        }
        /* Part 2b: the reach of the final path node = reach(jpivot) in U[pmap_new,:]; triangular iff the
         * combined reach does not intersect the spike pattern except in the final path index. */
        if (istriangular) {
            lu_int j = path[m - 1];
            rtop = orc_dfs(j, w_begin, w_end, w_index, rtop, reach, pstack, marked, M);
            ORC_ASSERT(reach[rtop] == j);
            reach[rtop] = jpivot;
            marked[j]--; /* unmark for a moment */
            for (lu_int pos = u_begin[ipivot]; u_index[pos] >= 0; pos++)
                if (marked[qmap[u_index[pos]]] == M) istriangular = 0;
            marked[j]++; /* restore */
        }
        /* If U is permuted triangular, permute to a zero-free diagonal and set up the reach lists. */
        if (istriangular) {
            lu_int nswap = m - top - 1;
            permute(lu, path + top, nswap); /* FIX D9: nswap + 1 nodes */
            u_nz--;
            lu->nunsymperm_total++; /* test hook */
            ORC_ASSERT(reach[rtop] == jpivot);
            col_reach = reach + rtop;   /* stored in iwork2 */
            row_reach = iwork1 + rtop;
            nreach = m - rtop;
            for (lu_int n = 0; n < nreach; n++) row_reach[n] = pmap[col_reach[n]];
        }
    }

    /* ---- Forrest-Tomlin update (:820-889) */
    if (!istriangular) {
        /* remove row ipivot from column file */
        for (lu_int pos = w_begin[jpivot]; pos < w_end[jpivot]; pos++) {
            lu_int j = w_index[pos];
            ORC_ASSERT(j != jpivot);
            lu_int where_ = -1, end;
            for (end = u_begin[pmap[j]]; u_index[end] >= 0; end++)
                if (u_index[end] == ipivot) where_ = end;
            ORC_ASSERT(where_ >= 0);
            u_index[where_] = u_index[end - 1];
            u_value[where_] = u_value[end - 1];
            u_index[end - 1] = -1;
            u_nz--;
        }
        /* remove row ipivot from row file */
        w_end[jpivot] = w_begin[jpivot];
        /* replace pivot */
        col_pivot[jpivot] = newpiv;
        row_pivot[ipivot] = newpiv;
        lu->min_pivot = fmin(lu->min_pivot, fabs(newpiv));
        lu->max_pivot = fmax(lu->max_pivot, fabs(newpiv));
        /* drop zeros from row eta; update max entry of row etas */
        nz = 0;
        put = r_begin[nforrest];
        double max_eta = 0.0;
        for (lu_int pos = put; pos < r_begin[nforrest + 1]; pos++) {
            if (l_value[pos] != 0.0) {
                max_eta = fmax(max_eta, fabs(l_value[pos]));
                l_index[put] = l_index[pos];
                l_value[put++] = l_value[pos];
                nz++;
            }
        }
        r_begin[nforrest + 1] = put;
        lu->r_nz += nz;
        lu->max_eta = fmax(lu->max_eta, max_eta);
        /* prepare permutation update */
        nreach = 1;
        row_reach = &ipivot_vec; /* FIX D7 */
        col_reach = &jpivot_vec;
        lu->nforrest++;
        lu->nforrest_total++;
    }

    /* ---- Update permutations (:891-911) */
    if (lu->pivotlen + nreach > 2 * m) orc_garbage_perm(lu);
    put = lu->pivotlen;
    for (lu_int n = 0; n < nreach; n++) pivotrow[put++] = row_reach[n];
    put = lu->pivotlen;
    for (lu_int n = 0; n < nreach; n++) pivotcol[put++] = col_reach[n];
    lu->pivotlen += nreach;

    /* ---- Clean up (:913-957) */
    lu_int used = u_begin[m];
    if (used - u_nz - m > orc_trunc(lu->compress_thres * (double)used)) {
        nz = compress_packed(m, u_begin, u_index, u_value);
        ORC_ASSERT(nz == u_nz);
    }
    used = w_begin[m];
    lu_int need = u_nz + orc_trunc(stretch * (double)u_nz) + m * pad;
    if (used - need > orc_trunc(lu->compress_thres * (double)used)) {
        nz = orc_file_compress(m, w_begin, w_end, w_flink, w_index, w_value, stretch, pad);
        ORC_ASSERT(nz == u_nz);
    }
    lu->pivot_error = piverr / (1.0 + fabs(newpiv));
    lu->u_nz = u_nz;
    lu->btran_for_update = -1;
    lu->ftran_for_update = -1;
    lu->update_cost_numer += (double)nz_roweta;
    lu->nupdate++;
    lu->nupdate_total++;
    return ORC_OK;
}

/* lu::solve_for_update -- lu/solve_for_update.rs:12-455.  xlhs == NULL: the solution is not wanted. */
int orc_lu_solve_for_update(orc_lu *lu, lu_int nrhs, const lu_int *irhs, const double *xrhs, lu_int *p_nlhs,
                            lu_int *ilhs, double *xlhs, char trans)
{
    const lu_int m = lu->m;
    const lu_int nforrest = lu->nforrest, pivotlen = lu->pivotlen;
    const lu_int nz_sparse = orc_trunc(lu->sparse_thres * (double)m);
    const double droptol = lu->droptol;
    const lu_int *p = P_(lu), *pmap = PMAP(lu), *qmap = QMAP(lu);
    lu_int *eta_row = ETA_ROW(lu), *r_begin = R_BEGIN(lu);
    const lu_int *pivotcol = PIVOTCOL(lu), *pivotrow = PIVOTROW(lu);
    const lu_int *l_begin = L_BEGIN(lu), *lt_begin = LT_BEGIN(lu), *lt_begin_p = LT_BEGIN_P(lu);
    const lu_int *u_begin = lu->u_begin, *w_begin = lu->w_begin, *w_end = lu->w_end;
    const double *col_pivot = lu->col_pivot, *row_pivot = lu->row_pivot;
    lu_int *l_index = lu->l_index, *u_index = lu->u_index, *w_index = lu->w_index;
    double *l_value = lu->l_value, *u_value = lu->u_value, *w_value = lu->w_value;
    lu_int *marked = MARKED(lu);
    lu_int *pattern_symb = IWORK1(lu), *pattern = IWORK1(lu) + m;
    double *work = lu->work0, *pstack = lu->work1;
    const int want_solution = p_nlhs && ilhs && xlhs;
    lu_int l_flops = 0, u_flops = 0, r_flops = 0;
    lu_int M, top, nz, nz_symb;

    if (trans == 't' || trans == 'T') {
        /* ---- transposed system (:59-245) */
        const lu_int jpivot = irhs[0];
        const lu_int ipivot = pmap[jpivot];
        const lu_int jbegin = w_begin[jpivot], jend = w_end[jpivot];
        /* Compute row eta vector.  Symbolic pattern in pattern_symb[top..m-1], values scattered into work. */
        M = ++lu->marker;
        top = orc_solve_symbolic(m, w_begin, w_end, w_index, jend - jbegin, w_index + jbegin, pattern_symb, pstack, marked, M);
        nz_symb = m - top;
        /* reallocate if not enough memory in Li, Lx (where we store R) */
        lu_int room = lu->l_mem - r_begin[nforrest];
        if (room < nz_symb) {
            lu->addmem_l = nz_symb - room;
            return ORC_REALLOCATE;
        }
        for (lu_int pos = jbegin; pos < jend; pos++) work[w_index[pos]] = w_value[pos];
        orc_solve_triangular(nz_symb, pattern_symb + top, w_begin, w_end, w_index, w_value, col_pivot, 0.0, work, pattern, &u_flops);
        /* Compress row eta into L, pattern mapped from column to row indices (the triangularity test in
         * update needs the symbolic pattern). */
        lu_int put = r_begin[nforrest];
        for (lu_int t = top; t < m; t++) {
            lu_int j = pattern_symb[t];
            l_index[put] = pmap[j];
            l_value[put++] = work[j];
            work[j] = 0.0;
        }
        r_begin[nforrest + 1] = put;
        eta_row[nforrest] = ipivot;
        lu->btran_for_update = jpivot;
        if (!want_solution) goto done;

        /* Scatter the row eta into xlhs and scale it to become the solution to U^{-1}*[unit vector]. */
        M = ++lu->marker;
        pattern[0] = ipivot;
        marked[ipivot] = M;
        const double pivot = col_pivot[jpivot];
        xlhs[ipivot] = 1.0 / pivot;
        const double xdrop = droptol * fabs(pivot);
        nz = 1;
        for (lu_int pos = r_begin[nforrest]; pos < r_begin[nforrest + 1]; pos++) {
            if (fabs(l_value[pos]) > xdrop) {
                lu_int i = l_index[pos];
                pattern[nz++] = i;
                marked[i] = M;
                xlhs[i] = -l_value[pos] / pivot;
            }
        }
        /* Solve with update etas.  Append fill-in to pattern. */
        for (lu_int t = nforrest - 1; t >= 0; t--) {
            lu_int ip = eta_row[t];
            if (xlhs[ip] != 0.0) {
                double x = xlhs[ip];
                for (lu_int pos = r_begin[t]; pos < r_begin[t + 1]; pos++) {
                    lu_int i = l_index[pos];
                    if (marked[i] != M) {
                        marked[i] = M;
                        pattern[nz++] = i;
                    }
                    xlhs[i] -= x * l_value[pos];
                    r_flops++;
                }
            }
        }
        if (nz <= nz_sparse) { /* sparse triangular solve with L' */
            M = ++lu->marker;
            top = orc_solve_symbolic(m, lt_begin, NULL, l_index, nz, pattern, pattern_symb, pstack, marked, M);
            nz_symb = m - top;
            nz = orc_solve_triangular(nz_symb, pattern_symb + top, lt_begin, NULL, l_index, l_value, NULL, droptol, xlhs, ilhs, &l_flops);
            *p_nlhs = nz;
        } else { /* sequential triangular solve with L' */
            nz = 0;
            for (lu_int k = m - 1; k >= 0; k--) {
                lu_int ip = p[k];
                if (xlhs[ip] != 0.0) {
                    double x = xlhs[ip];
                    for (lu_int pos = lt_begin_p[k]; l_index[pos] >= 0; pos++) {
                        xlhs[l_index[pos]] -= x * l_value[pos];
                        l_flops++;
                    }
                    if (fabs(x) > droptol) ilhs[nz++] = ip;
                    else xlhs[ip] = 0.0;
                }
            }
            *p_nlhs = nz;
        }
    } else {
        /* ---- forward system (:246-435) */
        M = ++lu->marker;
        top = orc_solve_symbolic(m, l_begin, NULL, l_index, nrhs, irhs, pattern_symb, pstack, marked, M);
        nz_symb = m - top;
        for (lu_int n = 0; n < nrhs; n++) work[irhs[n]] = xrhs[n];
        nz = orc_solve_triangular(nz_symb, pattern_symb + top, l_begin, NULL, l_index, l_value, NULL, droptol, work, pattern, &l_flops);
        /* unmark cancellation */
        if (nz < nz_symb) {
            lu_int t = top, n = 0;
            while (n < nz) {
                lu_int i = pattern_symb[t];
                if (i == pattern[n]) n++;
                else marked[i]--;
                t++;
            }
            while (t < m) marked[pattern_symb[t++]]--;
        }
        /* Solve with update etas.  Append fill-in to pattern. */
        lu_int pos = r_begin[0];
        for (lu_int t = 0; t < nforrest; t++) {
            lu_int ip = eta_row[t];
            double x = 0.0;
            while (pos < r_begin[t + 1]) {
                x += work[l_index[pos]] * l_value[pos];
                pos++;
            }
            work[ip] -= x;
            if (x != 0.0 && marked[ip] != M) {
                marked[ip] = M;
                pattern[nz++] = ip;
            }
        }
        r_flops += r_begin[nforrest] - r_begin[0];
        /* reallocate if not enough memory in U */
        lu_int room = lu->u_mem - u_begin[m];
        lu_int need = nz + 1;
        if (room < need) {
            for (lu_int n = 0; n < nz; n++) work[pattern[n]] = 0.0;
            lu->addmem_u = need - room;
            return ORC_REALLOCATE;
        }
        /* Compress spike into U. */
        lu_int put = u_begin[m];
        for (lu_int n = 0; n < nz; n++) {
            lu_int i = pattern[n];
            u_index[put] = i;
            u_value[put++] = work[i];
            if (!want_solution) work[i] = 0.0;
        }
        u_index[put] = -1; /* terminate column */
        lu->ftran_for_update = 0;
        if (!want_solution) goto done;

        if (nz <= nz_sparse) { /* sparse triangular solve with U */
            M = ++lu->marker;
            top = orc_solve_symbolic(m, u_begin, NULL, u_index, nz, pattern, pattern_symb, pstack, marked, M);
            nz_symb = m - top;
            nz = orc_solve_triangular(nz_symb, pattern_symb + top, u_begin, NULL, u_index, u_value, row_pivot, droptol, work, ilhs, &u_flops);
            /* Permute solution into xlhs.  Map pattern from row indices to column indices. */
            for (lu_int n = 0; n < nz; n++) {
                lu_int i = ilhs[n];
                lu_int j = qmap[i];
                ilhs[n] = j;
                xlhs[j] = work[i];
                work[i] = 0.0;
            }
        } else { /* sequential triangular solve with U */
            nz = 0;
            for (lu_int k = pivotlen - 1; k >= 0; k--) {
                lu_int ip = pivotrow[k], jp = pivotcol[k];
                if (work[ip] != 0.0) {
                    double x = work[ip] / row_pivot[ip];
                    work[ip] = 0.0;
                    for (lu_int pos2 = u_begin[ip]; u_index[pos2] >= 0; pos2++) {
                        work[u_index[pos2]] -= x * u_value[pos2];
                        u_flops++;
                    }
                    if (fabs(x) > droptol) {
                        ilhs[nz++] = jp;
                        xlhs[jp] = x;
                    }
                }
            }
        }
        *p_nlhs = nz;
    }
done:
    lu->l_flops += l_flops;
    lu->u_flops += u_flops;
    lu->r_flops += r_flops;
    lu->update_cost_numer += (double)r_flops;
    return ORC_OK;
}
