// k_chain.h -- triangular sweeps as a decoupled access / execute pipeline, one workgroup per sweep chain.
//
// A triangular sweep (condest.rs, residual_test.rs, solve_dense.rs) is a chain of m dependent steps.  On the
// banded LP bases the chain is real: the U of the C3 basis has a dependency path of 50 380 steps (SURVEY 8d's
// tri_frac = 0.5 part is one long chain; L has depth 571), so level scheduling has nothing to offer there and
// what counts is the time of ONE step.  k_sweep.h runs a sweep on one wave that also fetches its own operands
// from global memory one and two steps ahead: a step then takes a global round trip (~0.85 us), because the
// wave has to wait for the loads of this step before it can start the next.
//
// Here the wave that walks the chain never loads from global memory.  The other waves of the workgroup
// (helpers) stream the operands of the coming steps into LDS rings -- step records (length, pivot, own value,
// output index) and entries (value, and WHERE the operand of the work vector is found) -- hundreds of steps
// ahead of the chain wave; the chain wave reads LDS only (the records of a block once, a step's fields by v_readlane;
// the entries two steps ahead) and its step is the arithmetic: products, the ordered sum in the reference's order,
// the step function (one f64 division), one LDS and one global store.  Work-vector operands come from
//   * a window of the last CH_W results in LDS (xwin, written by the chain wave itself),
//   * the previous step's result, forwarded in a register,
//   * for producers at least CH_FAR steps back: the value gathered from global memory by the helper (final by
//     then: a helper stages a block only after the chain wave has published -- stores drained -- a progress
//     that puts every such producer behind it).  The leading terms of a step that are of this kind are added by
//     the helper itself, in the step's order and with the chain wave's roundings: the chain wave starts behind them.
// Every sweep is in GATHER form: step k reads results of earlier steps only.  The reference's scatter-form
// loops (x[i] -= t_k * a_ik over column k, k in sweep order) are run over the transposed storage -- row-wise L
// ascending, U rows descending in the pivot order (k_rows_grid below) -- accumulating into the step's own
// entry in the same order with the same two roundings per term, so all results stay bit-identical.
//
// Flow control (LDS, in-order per wave): helpers take blocks of CH_SB steps round-robin; the entry-ring base of a
// block is handed from the helper of the previous block as soon as that one knows its entry count; a block is
// staged only when its step-ring slots (chain CH_NB blocks behind) and its entry-ring space are free; the chain
// wave waits for blk_ready of the block it enters.  Steps with more than 64 entries are not staged: the chain
// wave takes them straight from global memory (its own stores are ordered before its later loads).
#pragma once
#include "blu_dev.h"

#define CH_SB 32                 // steps per block
#define CH_NB 8                  // blocks in the step ring
#define CH_CS (CH_SB * CH_NB)    // step ring
#define CH_CE 4096               // entry ring (two full blocks of 64-entry steps)
#define CH_W 2048                // window of results kept in LDS (positions)
#define CH_FAR 320               // a producer this many steps back is final and visible when a helper stages (> CH_CS)
#define CH_HAS 0x10000           // record: the step has entries (some or all may have gone into `init` already)
#define CH_HT 16                 // hand-off ring of entry bases
#define CH_SEL_FAR (-1)          // operand: the helper's gathered value
#define CH_SEL_PREV (-2)         // operand: the previous step's result (register)

struct __attribute__((aligned(16))) ChRecA { // what the chain wave reads of a step, in two 16-byte LDS reads
    int n, eb, w, k;
};
struct __attribute__((aligned(16))) ChRecB {
    double diag, own;
};
struct __attribute__((aligned(16))) ChOpsV {
    double val, xv;
};
struct ChainLds {
    ChOpsV ev[CH_CE];
    int sel[CH_CE];
    double xwin[CH_W];
    ChRecA ra[CH_CS];
    ChRecB rb[CH_CS];
    double rinit[CH_CS]; // the accumulator's start: own value or 0, plus the leading terms the helper has added already
    long long s_b[CH_CS];
    volatile int blk_ready[CH_NB];
    volatile int eb_tag[CH_HT];
    volatile int eb_val[CH_HT];
    volatile int chain_done; // blocks finished by the chain wave with their stores drained
    volatile int chain_eb;   // entry-ring position of the first entry not consumed by those blocks
    volatile int abort;      // a wait ran into its bound (a defect, never a valid state): everybody leaves
};

struct ChMeta {
    long long b; // storage offset of the step's entries
    int len;     // number of entries
    int w;       // index of the output vector the step writes
    double diag, own;
};
struct ChEnt {
    int pos;  // sweep position k of the producer (0..m-1)
    int gidx; // its index in the output vector
    double val;
};

// Bounded wait on LDS state written by another wave: 2^24 polls of s_sleep(1) (of the order of a second) and the
// sweep is abandoned with an error instead of hanging the GPU; the callers report the code to the host, which
// never uses the results of an abandoned sweep.
#define CH_WAIT(L, cond, code)                                         \
    do {                                                            \
        int it_ = 0;                                                \
        while (!(cond)) {                                           \
            if ((L)->abort || ++it_ > (1 << 24)) {                  \
                if (!(L)->abort) (L)->abort = (code);               \
                break;                                              \
            }                                                       \
            __builtin_amdgcn_s_sleep(1);                            \
        }                                                           \
    } while (0)

#ifdef BLU_EMU_BUILD // (CPU emulation build: memory is always current; the DPP sum below through the emulator's exchange)
__device__ __forceinline__ void ch_lds_fence() {}
__device__ __forceinline__ void ch_vm_drain() {}
#else
__device__ __forceinline__ void ch_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ch_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
__device__ __forceinline__ double ch_load_final(gdouble_p p)
{
    // a value another wave of this workgroup stored some time ago: read past this CU's vector cache
    const long long b = __hip_atomic_load((GPTR(const long long))p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double(b);
}

// ---- helper side: stage block b ---------------------------------------------------------------------------
template <bool INIT_OWN, bool SUB, class A>
__device__ __forceinline__ void ch_stage_block(const A &ad, ChainLds *L, int b, int k0, int dir, int nsteps, gdouble_p out)
{
    const int lane = lane_id();
    const int s0 = b * CH_SB;
    const int ns = nsteps - s0 < CH_SB ? nsteps - s0 : CH_SB;
    const int k = k0 + dir * (s0 + (lane < ns ? lane : 0));
    ChMeta M = ad.meta(k);
    if (M.len < 0) M.len = 0; // (defensive: a negative length would make the loops below unbounded)
    const bool mine = lane < ns;
    const int st = (mine && M.len <= 64) ? M.len : 0;
    const int incl = wave_incl_scan_i(st);
    const int off = incl - st;
    const int cnt = __builtin_amdgcn_readlane(incl, 63);
    // entry-ring base: from the helper of the previous block
    int base = 0;
    if (b > 0) {
        CH_WAIT(L, L->eb_tag[b & (CH_HT - 1)] == b, 1);
        base = L->eb_val[b & (CH_HT - 1)];
    }
    if (lane == 0) {
        L->eb_val[(b + 1) & (CH_HT - 1)] = base + cnt;
        ch_lds_fence();
        L->eb_tag[(b + 1) & (CH_HT - 1)] = b + 1;
    }
    // room: step-ring slots of block b - CH_NB, entry-ring space; this also makes every result further back
    // than the window final and visible (see the header)
    CH_WAIT(L, L->chain_done >= b - CH_NB + 1 && base + cnt - L->chain_eb <= CH_CE, 2);
    asm volatile("" ::: "memory");
    if (L->abort) return;
    const int slot0 = s0 & (CH_CS - 1);
    if (mine) {
        const int sl = slot0 + lane;
        ChRecA a;
        a.n = M.len <= 64 ? (M.len > 0 ? (M.len | CH_HAS) : 0) : -1;
        a.eb = base + off;
        a.w = M.w;
        a.k = k;
        L->ra[sl] = a;
        ChRecB bq;
        bq.diag = M.diag;
        bq.own = M.own;
        L->rb[sl] = bq;
        L->rinit[sl] = INIT_OWN ? M.own : 0.0;
        L->s_b[sl] = M.b;
    }
    ch_lds_fence();
    // entries, flat over the block: 4 x 64 per pass, loads of a pass issued together
    for (int f0 = 0; f0 < cnt; f0 += 256) {
        int tt[4], ee[4];
        ChEnt E[4];
        bool ok[4], far[4];
        double xo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = f0 + u * 64 + lane;
            ok[u] = f < cnt;
            // step of flat entry f: the last t with s_eb[t] - base <= f (binary search over the block's records)
            int lo = 0, hi = ns - 1;
            const int target = base + (ok[u] ? f : 0);
#pragma unroll
            for (int it = 0; it < 5; it++) {
                const int mid = (lo + hi + 1) >> 1;
                const bool ge = (L->ra[slot0 + mid].eb <= target) && (mid <= hi);
                lo = ge ? mid : lo;
                hi = ge ? hi : mid - 1;
            }
            // skip unstaged / empty steps that share the same base: take the LAST step with s_eb <= target that has entries
            tt[u] = lo;
            ee[u] = target - L->ra[slot0 + lo].eb;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            E[u].pos = 0;
            E[u].gidx = 0;
            E[u].val = 0.0;
            if (ok[u]) E[u] = ad.ent(L->ra[slot0 + tt[u]].k, L->s_b[slot0 + tt[u]], ee[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = L->ra[slot0 + tt[u]].k;
            const int d = (kk - E[u].pos) * dir;
            far[u] = ok[u] && d >= CH_FAR;
            xo[u] = 0.0;
            if (far[u]) xo[u] = ch_load_final(out + E[u].gidx);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (ok[u]) {
                const int kk = L->ra[slot0 + tt[u]].k;
                const int r = (base + f0 + u * 64 + lane) & (CH_CE - 1);
                ChOpsV ov;
                ov.val = E[u].val;
                ov.xv = xo[u];
                L->ev[r] = ov;
                L->sel[r] = far[u] ? CH_SEL_FAR : (E[u].pos == kk - dir ? CH_SEL_PREV : (E[u].pos & (CH_W - 1)));
            }
        }
    }
    ch_lds_fence();
    // The leading terms of a step whose producers are that far back are final: this lane (one per step) adds them in
    // the step's order, with the chain wave's two roundings per term, and the chain wave starts behind them.  (Lines
    // sorted in sweep order -- canonical U columns, row-wise L, sorted U rows -- have all such terms in front.)
    if (mine && M.len > 0 && M.len <= 64) {
        const int sl = slot0 + lane;
        const int eb = base + off;
        double acc = INIT_OWN ? M.own : 0.0;
        int p = 0;
        while (p < M.len && L->sel[(eb + p) & (CH_CE - 1)] == CH_SEL_FAR) {
            const ChOpsV ov = L->ev[(eb + p) & (CH_CE - 1)];
            const double term = __dmul_rn(ov.xv, ov.val);
            acc = SUB ? __dsub_rn(acc, term) : __dadd_rn(acc, term);
            p++;
        }
        if (p > 0) {
            L->ra[sl].n = (M.len - p) | CH_HAS;
            L->ra[sl].eb = eb + p;
            L->rinit[sl] = acc;
        }
    }
    ch_lds_fence();
    if (lane == 0) L->blk_ready[b & (CH_NB - 1)] = b + 1;
}

// ---- chain side -------------------------------------------------------------------------------------------
struct ChRec {
    int n, eb, w, k; // n: entries left for the chain wave | CH_HAS, or -1 (long step)
    double diag, own, init;
};
struct ChOps {
    double val, xv;
    int sel;
};
// (the four integers stay in vector registers while the record travels through the look-ahead stages: turning them
// into scalars right after the LDS read would make the wave wait for the read in the middle of every step)
__device__ __forceinline__ ChRec ch_read_rec(ChainLds *L, int s)
{
    const int sl = s & (CH_CS - 1);
    const ChRecA a = L->ra[sl];
    const ChRecB b = L->rb[sl];
    ChRec R;
    R.n = a.n;
    R.eb = a.eb;
    R.w = a.w;
    R.k = a.k;
    R.diag = b.diag;
    R.own = b.own;
    R.init = L->rinit[sl];
    return R;
}
__device__ __forceinline__ ChOps ch_read_ops(ChainLds *L, const ChRec &R)
{
    ChOps E;
    E.val = 0.0;
    E.xv = 0.0;
    E.sel = CH_SEL_FAR;
    if (lane_id() < (R.n & (CH_HAS - 1)) && R.n > 0) {
        const int r = (R.eb + lane_id()) & (CH_CE - 1);
        const ChOpsV v = L->ev[r];
        E.val = v.val;
        E.xv = v.xv;
        E.sel = L->sel[r];
    }
    return E;
}
// acc -/+= the products of lanes 0..n-1, in lane order (the reference's sequential loop)
template <bool SUB>
__device__ __forceinline__ double ch_accumulate(double acc, double prod, int n)
{
    const unsigned lo = (unsigned)__double_as_longlong(prod), hi = (unsigned)(__double_as_longlong(prod) >> 32);
#define CH_TERM(T)                                                                                              \
    {                                                                                                           \
        const unsigned a_ = __builtin_amdgcn_readlane(lo, (T)), b_ = __builtin_amdgcn_readlane(hi, (T));        \
        const double term_ = __longlong_as_double((long long)(((unsigned long long)b_ << 32) | a_));            \
        acc = SUB ? __dsub_rn(acc, term_) : __dadd_rn(acc, term_);                                              \
    }
    int t = 0;
    for (; t + 4 <= n; t += 4) {
        CH_TERM(t)
        CH_TERM(t + 1)
        CH_TERM(t + 2)
        CH_TERM(t + 3)
    }
    for (; t < n; t++) CH_TERM(t)
#undef CH_TERM
    return acc;
}

// The same for at most 16 terms when every row of 16 lanes holds the products (lane l: term l & 15): one instruction
// per term.  v_fmac_f64 is the one 64-bit VALU operation of gfx90a+ that takes a DPP operand, with row_newbcast:t
// (lane t of the row to the whole row); acc = fma(term_t, +-1.0, acc) rounds once, exactly like acc +- term_t.  Three
// instructions per term otherwise (two v_readlane and the add).  An s_nop covers the VALU-write -> DPP-read hazards (2 wait states after a write of the operand, 5 after a VALU write of EXEC)
// of the products, which the compiler cannot see inside the asm.
template <bool SUB>
__device__ __forceinline__ double ch_accumulate_rows(double acc, double prod, int n)
{
    const double sg = SUB ? -1.0 : 1.0;
#ifdef BLU_EMU_BUILD
#define CH_D(T) acc = fma(__shfl(prod, (lane_id() & ~15) + T), sg, acc);
#define CH_D0 CH_D(0)
#else
#define CH_D(T) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(prod), "v"(sg));
// (term 0 is the first in every path: the wait states ride in the same asm statement, where no scheduler can move them)
#define CH_D0 asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(prod), "v"(sg));
#endif
#define CH_TAIL(A, B, C) \
    if (n > A) {         \
        CH_D(A)          \
        if (n > B) {     \
            CH_D(B)      \
            if (n > C) { \
                CH_D(C)  \
            }            \
        }                \
    }
    if (n >= 4) {
        CH_D0 CH_D(1) CH_D(2) CH_D(3)
        if (n >= 8) {
            CH_D(4) CH_D(5) CH_D(6) CH_D(7)
            if (n >= 12) {
                CH_D(8) CH_D(9) CH_D(10) CH_D(11)
                if (n >= 16) {
                    CH_D(12) CH_D(13) CH_D(14) CH_D(15)
                } else {
                    CH_TAIL(12, 13, 14)
                }
            } else {
                CH_TAIL(8, 9, 10)
            }
        } else {
            CH_TAIL(4, 5, 6)
        }
    } else if (n > 0) {
        CH_D0
        CH_TAIL(1, 2, 3)
    }
#undef CH_TAIL
#undef CH_D0
#undef CH_D
    return acc;
}

// One sweep by the whole workgroup: wave 0 walks the chain, the other waves stage.  All waves must call it.
//   INIT_OWN: the accumulator starts at the step's own value (scatter-form loops of the reference), else at 0
//   SUB:      terms are subtracted, else added
//   f(k, has_entries, acc, own, diag) -> the step's result, stored to out[w] and the window; f may store more
// The helpers add the leading terms whose producers are at least CH_FAR steps back (ch_stage_block).
// On return every store of the sweep has completed and the workgroup is synchronised; false: abandoned (defect).
template <bool INIT_OWN, bool SUB, class A, class F>
__device__ __forceinline__ bool chain_sweep(const A &ad, ChainLds *L, int k0, int dir, int nsteps, gdouble_p out, F f)
{
    const int w = wave_id(), nw = num_waves(), lane = lane_id();
    const int nblk = (nsteps + CH_SB - 1) / CH_SB;
    if (threadIdx.x < CH_NB) L->blk_ready[threadIdx.x] = 0;
    if (threadIdx.x < CH_HT) L->eb_tag[threadIdx.x] = -1;
    if (threadIdx.x == 0) {
        L->chain_done = 0;
        L->chain_eb = 0;
        L->abort = 0;
    }
    __syncthreads();
    if (w > 0) {
        for (int b = w - 1; b < nblk && !L->abort; b += nw - 1) ch_stage_block<INIT_OWN, SUB>(ad, L, b, k0, dir, nsteps, out);
    } else if (nsteps > 0) {
#ifdef BLU_PROFILE
        long long t_wait = 0, t_drain = 0;
        const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
        // The records of the steps past the end read as empty: the ring slots behind the last block are written
        // as "n = 0" by nobody, so the look-ahead below is clamped to the last step instead.
        const auto wait_block = [&](int b) {
            if (b < nblk) {
                CH_WAIT(L, L->blk_ready[b & (CH_NB - 1)] == b + 1, 3);
                asm volatile("" ::: "memory");
            }
        };
        const int last = nsteps - 1;
        // The records of a block are read once, 64 lanes at a time: lane t holds the record of step bs + t (lanes 32,
        // 33 reach into the next block for the look-ahead) and a step takes its fields by v_readlane -- a handful of
        // scalar moves instead of LDS reads, address arithmetic and a three-stage rotation of ten registers per step.
        const auto rl_d = [](double x, int t) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), t), hi = __builtin_amdgcn_readlane(__double2hiint(x), t);
            return __hiloint2double(hi, lo);
        };
        const auto read_ops = [&](int eb, int nf) {
            ChOps E;
            E.val = 0.0;
            E.xv = 0.0;
            E.sel = CH_SEL_FAR;
            const int n = nf < 0 ? 0 : (nf & (CH_HAS - 1));
            const int li = n <= 16 ? (lane & 15) : lane; // (up to 16 entries: every row of 16 lanes holds them all, ch_accumulate_rows)
            if (li < n) {
                const int r = (eb + li) & (CH_CE - 1);
                const ChOpsV v = L->ev[r];
                E.val = v.val;
                E.xv = v.xv;
                E.sel = L->sel[r];
            }
            return E;
        };
        wait_block(0);
        ChRec BR;
        ChOps E1, E2;
        double xw1 = 0.0, vprev = 0.0;
        for (int b = 0; b < nblk && !L->abort; b++) {
            // the look-ahead of this block's steps reaches two steps into the next block
#ifdef BLU_PROFILE
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime();
#endif
            wait_block(b + 1);
#ifdef BLU_PROFILE
            t_wait += (long long)__builtin_amdgcn_s_memtime() - tw0;
#endif
            const int bs = b * CH_SB;
            {
                const int sl = bs + lane;
                BR = ch_read_rec(L, sl < last ? sl : last);
            }
            if (b == 0) {
                E1 = read_ops(__builtin_amdgcn_readlane(BR.eb, 0), __builtin_amdgcn_readlane(BR.n, 0));
                xw1 = E1.sel >= 0 ? L->xwin[E1.sel] : 0.0;
                E2 = read_ops(__builtin_amdgcn_readlane(BR.eb, 1), __builtin_amdgcn_readlane(BR.n, 1));
            }
            const int ns = (bs + CH_SB < nsteps ? CH_SB : nsteps - bs);
            for (int t = 0; t < ns; t++) {
                // ahead: window operands of step s+1, entries of s+2
                const double xw2 = E2.sel >= 0 ? L->xwin[E2.sel] : 0.0;
                const ChOps E3 = read_ops(__builtin_amdgcn_readlane(BR.eb, t + 2), __builtin_amdgcn_readlane(BR.n, t + 2));
                // step s = bs + t
                const int nf = __builtin_amdgcn_readlane(BR.n, t), k1 = __builtin_amdgcn_readlane(BR.k, t), w1 = __builtin_amdgcn_readlane(BR.w, t);
                const double own1 = rl_d(BR.own, t), diag1 = rl_d(BR.diag, t);
                const int n1 = nf < 0 ? -1 : (nf & (CH_HAS - 1));
                double acc = rl_d(BR.init, t);
                if (n1 > 0) {
                    const double x = E1.sel == CH_SEL_PREV ? vprev : (E1.sel >= 0 ? xw1 : E1.xv);
                    if (n1 <= 16) acc = ch_accumulate_rows<SUB>(acc, (lane & 15) < n1 ? __dmul_rn(x, E1.val) : 0.0, n1);
                    else acc = ch_accumulate<SUB>(acc, lane < n1 ? __dmul_rn(x, E1.val) : 0.0, n1);
                } else if (n1 < 0) { // long step: straight from global memory
                    const ChMeta M = ad.meta(k1);
                    for (int o = 0; o < M.len; o += 64) {
                        ChEnt E;
                        E.pos = 0;
                        E.gidx = 0;
                        E.val = 0.0;
                        const int nn = M.len - o < 64 ? M.len - o : 64;
                        if (lane < nn) E = ad.ent(k1, M.b, o + lane);
                        acc = ch_accumulate<SUB>(acc, lane < nn ? __dmul_rn(out[E.gidx], E.val) : 0.0, nn);
                    }
                }
                const double v = f(k1, nf != 0, acc, own1, diag1);
                if (lane == 0) {
                    L->xwin[k1 & (CH_W - 1)] = v;
                    out[w1] = v;
                }
                vprev = v;
                E1 = E2;
                xw1 = xw2;
                E2 = E3;
            }
            // block finished: drain the stores, publish (lane 32 of the records = the first step of the next block)
#ifdef BLU_PROFILE
            const long long td0 = (long long)__builtin_amdgcn_s_memtime();
#endif
            ch_vm_drain();
#ifdef BLU_PROFILE
            t_drain += (long long)__builtin_amdgcn_s_memtime() - td0;
#endif
            ch_lds_fence();
            const int eb_next = __builtin_amdgcn_readlane(BR.eb, CH_SB);
            if (lane == 0 && bs + CH_SB < nsteps) {
                L->chain_eb = eb_next;
                L->chain_done = b + 1;
            }
        }
        ch_vm_drain();
#ifdef BLU_PROFILE
        if (lane == 0)
            printf("chain sweep (block %d): %d steps, %.0f cycles/step, waiting for blocks %.0f, draining stores %.0f\n", (int)blockIdx.x, nsteps,
                   (double)((long long)__builtin_amdgcn_s_memtime() - t_begin) / nsteps, (double)t_wait / nsteps, (double)t_drain / nsteps);
#endif
    }
    __syncthreads();
    return L->abort == 0; // (the code of the wait that gave up stays in L->abort for the caller's error line)
}
