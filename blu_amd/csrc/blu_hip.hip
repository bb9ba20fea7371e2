// blu_hip.hip -- host side of libblu_hip.so: the C ABI of include/blu_hip.h.
//
// Mirrors `struct BLU` (src/blu.rs): BLU::new / factorize (with the internal realloc loop,
// blu.rs:95-118, 345-377) / get_factors / solve_dense, plus the parameter fields and getters of
// `struct LU` (src/lu/lu.rs).  All numerical work happens in the HIP kernels of this directory;
// there is no CPU fallback -- without a gfx950 device blu_hip_new returns NULL.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/blu_hip.h"
#include "blu_dev.h"
#include "k_finish.hip"
#define BLU_NS pv_single
#define BLU_CFG_BATCH 0
#define BLU_CFG_WAVE 0
#include "k_pivot.hip"
#undef BLU_NS
#undef BLU_CFG_BATCH
#define BLU_NS pv_batch
#define BLU_CFG_BATCH 1
#include "k_pivot.hip"
#undef BLU_NS
#undef BLU_CFG_BATCH
#undef BLU_CFG_WAVE
#define BLU_NS pv_wave
#define BLU_CFG_BATCH 1
#define BLU_CFG_WAVE 1
#include "k_pivot.hip"
#undef BLU_NS
#undef BLU_CFG_BATCH
#undef BLU_CFG_WAVE
#define BLU_NS pv_wave2
#define BLU_CFG_BATCH 1
#define BLU_CFG_WAVE 2
#include "k_pivot.hip"
#undef BLU_NS
#undef BLU_CFG_BATCH
#undef BLU_CFG_WAVE
using pv_single::k_pivot_loop;
using pv_batch::k_pivot_loop_batch;
using pv_wave::k_pivot_loop_wave;
using pv_wave::k_pivot_loop_wave_r3;
using pv_wave2::k_pivot_loop_wave2;
using pv_wave2::k_pivot_loop_wave2_r3;
#include "k_prep.hip"
#include "k_solve.hip"
#include "k_solve_sparse.hip"
#include "k_stats.hip"
#include "k_chain.hip"
#include "k_update.hip"

#define BLU_STOPPED_STATUS 100 /* debug stepping only */

struct blu_hip {
    int device;
    int64_t m, b_nz_hint;
    // parameters (LU public fields + BLU.realloc_factor)
    double droptol, abstol, reltol, stretch, compress_thres, sparse_thres, realloc_factor;
    int64_t nzbias, maxsearch, pad, search_rows;
    // state
    int64_t nupdate;   // -1 = None
    int64_t nfactorize;
    DevLU D;           // host copy of the device descriptor (device pointers inside)
    char *slab;        // one allocation holding every fixed-size device array of this handle
    DevLU *dD;         // device copy (own slot)
    DevLU *dslot;      // where the descriptor currently lives on the device: dD, or a slot of a batch array
    Scalars hs;        // last downloaded scalars
    FinishOut O;       // device output buffers of get_factors
    FinishOut *dO;
    FinishOut *oslot;
    int batch_block;   // workgroup size of the pivot kernel when this handle leads a batch
    int batch_block_other, batch_block_stats; // workgroup sizes of k_prep / k_setup / k_finish and of k_stats in a batch
    int no_out_alias;  // diagnostic: 1 = canonical factors always in buffers of their own (see ensure_out)
    int pivot_kernel;  // 0 = default (one basis: k_pivot_loop; batch: k_pivot_loop_wave2 while all its workgroups are resident, else k_pivot_loop_wave), 1 = k_pivot_loop_wave, 2 = the multi-wave kernels, 3 = k_pivot_loop_wave2
    int64_t out_lcap, out_ucap;
    bool out_in_arena; // the canonical L / U of the last factorize live inside the (then dead) column arena, not in buffers of their own
    // owned device copies of the caller's B (blu_hip_factorize with host arrays)
    unsigned long long *ob_begin, *ob_end, *ob_i;
    double *ob_x;
    int64_t ob_mcap, ob_nzcap;
    // solve workspace
    double *d_rhs, *d_lhs; // solve_dense right-hand side / solution on the device (allocated at the first solve_dense)
    int gwork_cols;        // columns of m + 1 doubles D.gwork holds: 9 (statistics) or the waves of the pivot kernel, whichever is more
    // solve_sparse workspace (allocated at the first call)
    SparseWs sw;
    bool sw_ready;
    int64_t sw_ltcap;       // entries of the row-wise L buffers
    int64_t lt_for_nfact;   // nfactorize the row-wise L was built for (-1: none)
    int marker;             // lu.marker (src/lu/lu.rs:128)
    int *d_irhs;
    double *d_xrhs;
    int64_t rhs_cap;
    // update path (k_update.hip): mutable copies of U, maps, pivot sequence, row etas; built at the first solve_for_update
    UpdWs uw;
    UpdState ust;           // last downloaded state
    int64_t upd_alloc_m;    // m the fixed-size arrays of uw were allocated for (-1: none)
    int64_t upd_for_nfact;  // nfactorize uw was built for (-1: none)
    int64_t upd_extra;      // debug: arena slack of the update path (-1: default)
    int64_t sp_l_flops, sp_u_flops; // lu.l_flops / lu.u_flops
    int sp_branch;                  // 1 sparse, 2 sequential: branch of the last solve_sparse (diagnostic)
    // timing
    hipStream_t stream;
    hipEvent_t ev[4];
    double t_total, t_pivot;
    double t_phase[6]; // k_prep, k_setup, k_finish, statistics (all of it), and inside the statistics: k_rows_grid, k_stats_tail (seconds, HIP events)
    // single-matrix statistics / solve_dense on the chain pipeline (k_chain.hip)
    int chain_ok;           // 1: the device grants the LDS the chain kernels need
    int64_t chain_defects;  // statistics chains abandoned (a bounded wait gave up) and recomputed by the one-workgroup kernel
    int *ur_len, *ur_pos;   // U rows sorted descending in pivot order (k_rows_grid)
    double *ur_val;
    int64_t ur_cap;
    int64_t rows_for_nfact; // nfactorize the sorted U rows were built for (-1: none)
    int relaunches;
    int block_threads; // workgroup size of the pivot kernel
    int no_fast;       // debug: disable the LDS fast paths
    int skip_stats;    // 1: do not compute condest / residual_test inside factorize (keys return 0)
    GridWs *gw;        // scratch of the chip-wide O(nnz) phases (single-basis path)
    int grid_blocks;   // workgroups of their cooperative launches (0/1: one workgroup, as in a batch)
    int last_pivot_kernel, last_pivot_regs; // (statistics 118, 120)
    int lds_window, lds_window_mode; // bytes of dynamic LDS the batch forms of k_prep / k_finish may use as a counter window (0: not available), and when (env BLU_LDS_WINDOW)
    int num_cus, batch_grid; // CUs of the device; workgroups of k_prep / k_setup / k_finish in a batch (0: one per CU; env BLU_BATCH_GRID)
    int wave2_max;     // bases the card holds at once with TWO waves each (k_pivot_loop_wave2): a batch up to this size takes that kernel
    int wave2_r3_max, wave_r3_max; // ... and what it holds of the variants with the registers of three waves per SIMD (_r3)
    std::string err;
    int64_t stop_at;   // debug: -1 off
};

// ---------------------------------------------------------------------------------------------
static bool hip_ok(blu_hip *h, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (h) h->err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(h, call)                                \
    do {                                                \
        if (!hip_ok((h), (call), #call)) return false;  \
    } while (0)

template <class T> static bool dalloc(blu_hip *h, T **p, size_t n)
{
    *p = nullptr;
    return hip_ok(h, hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)), "hipMalloc");
}
template <class T> static void dfree(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}
// grow a device array keeping the first `keep` elements
template <class T> static bool dgrow(blu_hip *h, T **p, size_t keep, size_t n)
{
    T *q;
    if (!dalloc(h, &q, n)) return false;
    if (*p && keep) {
        if (!hip_ok(h, hipMemcpy(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice), "hipMemcpy d2d")) return false;
    }
    if (*p) (void)hipFree((void *)*p);
    *p = q;
    return true;
}

static const int64_t kIntMax = 0x7ffffff0;

static void free_upd(blu_hip *h);
static void free_all(blu_hip *h)
{
    DevLU &D = h->D;
    // growable storage: separate allocations
    dfree(D.bc_idx); dfree(D.bc_val); dfree(D.bt_idx); dfree(D.bt_val);
    dfree(D.cidx); dfree(D.cval); dfree(D.ridx);
    dfree(D.lidx); dfree(D.uidx); dfree(D.lval); dfree(D.uval);
    if (!h->out_in_arena) { dfree(h->O.l_rowidx); dfree(h->O.l_value); dfree(h->O.u_rowidx); dfree(h->O.u_value); }
    dfree(h->ob_begin); dfree(h->ob_end); dfree(h->ob_i); dfree(h->ob_x);
    SparseWs &W = h->sw;
    dfree(W.marked); dfree(W.psym); dfree(W.pat); dfree(W.pstack); dfree(W.estack); dfree(W.work); dfree(W.xlhs); dfree(W.ilhs);
    dfree(W.xval); dfree(W.out); dfree(W.lt_ptr); dfree(W.lt_idx); dfree(W.lt_val); dfree(W.lt_cur);
    dfree(h->d_irhs); dfree(h->d_xrhs);
    dfree(h->d_rhs); dfree(h->d_lhs); dfree(D.gwork);
    dfree(h->ur_len); dfree(h->ur_pos); dfree(h->ur_val);
    free_upd(h);
    // everything else lives in the slab
    dfree(h->slab);
}

// host copy of the descriptor brought up to date with the handle's parameters
static void fill_desc(blu_hip *h)
{
    DevLU &D = h->D;
    D.m = (int)h->m;
    D.nzbias = (int)h->nzbias;
    D.maxsearch = (int)std::min<int64_t>(h->maxsearch, kIntMax);
    D.pad = (int)h->pad;
    D.search_rows = (int)h->search_rows;
    D.no_fast = h->no_fast;
    D.skip_stats = h->skip_stats;
    D.droptol = h->droptol;
    D.abstol = h->abstol;
    D.reltol = h->reltol;
    D.stretch = h->stretch;
}
static bool upload_desc(blu_hip *h)
{
    fill_desc(h);
    HIP_TRY(h, hipMemcpy(h->dslot, &h->D, sizeof(DevLU), hipMemcpyHostToDevice));
    return true;
}
static bool download_scalars(blu_hip *h)
{
    HIP_TRY(h, hipMemcpy(&h->hs, h->D.s, sizeof(Scalars), hipMemcpyDeviceToHost));
    return true;
}
static bool set_status(blu_hip *h, int st)
{
    HIP_TRY(h, hipMemcpy(&h->D.s->status, &st, sizeof(int), hipMemcpyHostToDevice));
    return true;
}

// ---------------------------------------------------------------------------------------------
// BLU::new -- src/blu.rs:61, LU::new src/lu/lu.rs:243-319
// ---------------------------------------------------------------------------------------------
extern "C" blu_hip *blu_hip_new(int64_t m, int64_t b_nz, int device)
{
    if (m < 0 || b_nz < 0 || m > kIntMax / 2 - 4 || b_nz > kIntMax / 8) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return nullptr; // kernels are built for gfx950 only

    blu_hip *h = new blu_hip();
    h->device = device;
    h->m = m;
    h->b_nz_hint = b_nz;
    // defaults, lu.rs:249-259 and blu.rs:68
    h->droptol = 1e-20;
    h->abstol = 1e-14;
    h->reltol = 0.1;
    h->nzbias = 1;
    h->maxsearch = 3;
    h->pad = 4;
    h->stretch = 0.3;
    h->compress_thres = 0.5;
    h->sparse_thres = 0.05;
    h->search_rows = 0;
    h->realloc_factor = 1.5;
    h->nupdate = -1;
    h->nfactorize = 0;
    memset(&h->sw, 0, sizeof(SparseWs));
    h->sw_ready = false;
    h->sw_ltcap = 0;
    h->lt_for_nfact = -1;
    h->marker = 0;
    h->d_irhs = nullptr;
    h->d_xrhs = nullptr;
    h->rhs_cap = 0;
    h->sp_l_flops = h->sp_u_flops = 0;
    h->sp_branch = 0;
    memset(&h->uw, 0, sizeof(UpdWs));
    memset(&h->ust, 0, sizeof(UpdState));
    h->upd_alloc_m = -1;
    h->upd_for_nfact = -1;
    h->upd_extra = -1;
    h->chain_ok = 0;
    h->chain_defects = 0;
    h->ur_len = h->ur_pos = nullptr;
    h->ur_val = nullptr;
    h->ur_cap = 0;
    h->rows_for_nfact = -1;
    h->stop_at = -1;
    h->block_threads = 1024;
    h->no_fast = 0;
    memset(&h->D, 0, sizeof(DevLU));
    memset(&h->hs, 0, sizeof(Scalars));
    memset(&h->O, 0, sizeof(FinishOut));

    DevLU &D = h->D;
    const size_t M = (size_t)m;
    // The reference starts with l_mem = u_mem = w_mem = b_nz and grows on demand (blu.rs:345-377).
    // HBM is plentiful: start roomy so that the pivot loop rarely has to leave the device.
    D.nzcap = (int)std::max<int64_t>(b_nz, 1);
    // L and U: sized for the fill an LP basis typically has (C3: l_nz = 0.7 nnz, u_nz = 1.5 nnz), not for the worst case:
    // a batch is limited by HBM capacity, and a factor that outgrows its storage costs one relaunch (ST_NEED_L / _U)
    D.lcap = (int)std::min<int64_t>(2 * b_nz + 2 * m + 64, kIntMax);
    D.ucap = (int)std::min<int64_t>(3 * b_nz + b_nz / 2 + 2 * m + 64, kIntMax);
    D.carena_cap = (int)std::min<int64_t>(6 * b_nz + 8 * m + 64, kIntMax);
    // (the row file re-appends more than the column file -- every pivot appends its row pattern to all rows of its
    // column: measured on the LP bases 3.99 M entries of row arena against 3.56 M of column arena at the 100k size.  An
    // eighth more from the start is 2 MB; outgrowing the arena costs a relaunch of the pivot kernel and, grown by
    // realloc_factor, 7 MB more per handle for good)
    D.rarena_cap = (int)std::min<int64_t>((int64_t)D.carena_cap + D.carena_cap / 8, kIntMax);
    bool ok = true;
    // Every fixed-size (m-dependent) array lives in ONE allocation: the pivot loop hops randomly over
    // all of them, and one large hipMalloc is mapped with large page-table fragments, while ~35
    // separate sub-megabyte allocations are not (TLB reach).  Growable storage stays separate.
    size_t slab_bytes = 0;
    std::vector<std::pair<void **, size_t>> reqs;
    auto want = [&](auto **p, size_t n) {
        reqs.push_back({(void **)p, slab_bytes});
        slab_bytes += (std::max<size_t>(n, 1) * sizeof(**p) + 255) & ~(size_t)255;
    };
    want(&D.bc_ptr, M + 1); want(&D.bt_ptr, M + 1);
    want(&D.pinv, M); want(&D.qinv, M); want(&D.prow, M + 1); want(&D.pcol, M + 1);
    want(&D.crec, M); want(&D.rrec, M); want(&D.chead, M + 2); want(&D.rhead, M + 2); // line records, count-list heads (blu_dev.h)
    want(&D.rowmark, M); want(&D.colmark, M);
    want(&D.tnew, M + 2); want(&D.tnewr, M + 2); want(&D.txrj, M + 2); want(&D.tmask, M + 2);
    want(&D.iw0, M + 2); want(&D.iw1, M + 2); want(&D.iw2, M + 2);
    want(&D.lbeg, M + 1); want(&D.ubeg, M + 1);
    want(&D.s, 1); want(&h->dD, 1); want(&h->dO, 1); want(&h->gw, 1);
    want(&h->O.rowperm, M); want(&h->O.colperm, M); want(&h->O.l_colptr, M + 1); want(&h->O.u_colptr, M + 1);
    ok = ok && dalloc(h, &h->slab, slab_bytes);
    if (ok)
        for (auto &r : reqs) *r.first = (void *)(h->slab + r.second);
    ok = ok && dalloc(h, &D.bc_idx, D.nzcap) && dalloc(h, &D.bc_val, D.nzcap) && dalloc(h, &D.bt_idx, D.nzcap) && dalloc(h, &D.bt_val, D.nzcap);
    ok = ok && dalloc(h, &D.cidx, D.carena_cap) && dalloc(h, &D.cval, D.carena_cap) && dalloc(h, &D.ridx, D.rarena_cap);
    ok = ok && dalloc(h, &D.lidx, D.lcap) && dalloc(h, &D.lval, D.lcap) && dalloc(h, &D.uidx, D.ucap) && dalloc(h, &D.uval, D.ucap);
    // gwork: the statistics use 8 columns of m + 1 doubles and two words more, pivot_any one dense work column per wave of
    // the pivot kernel (16 for a single basis, 1 or 2 in a batch: ensure_gwork grows it before such a launch).  A batch is
    // limited by HBM capacity: 9 columns instead of 16 are 5.6 MB less per handle of the 100k size.
    h->gwork_cols = 9;
    ok = ok && dalloc(h, &D.gwork, (size_t)h->gwork_cols * (M + 1));
    if (ok) ok = hip_ok(h, hipMemset(D.gwork, 0, (size_t)h->gwork_cols * (M + 1) * sizeof(double)), "hipMemset");
    h->d_rhs = h->d_lhs = nullptr;
    h->dslot = h->dD;
    h->oslot = h->dO;
    h->batch_block = 256;
    {   // diagnostic: which pivot kernel this handle launches (read once; blu_hip_dbg_set_pivot_kernel overrides)
        const char *bo = getenv("BLU_BATCH_OTHER"), *bs = getenv("BLU_BATCH_STATS");
        h->batch_block_other = bo ? atoi(bo) : 512; // (512 = two waves per SIMD with the registers of the register sorts: 8-15 % faster than 256, round 4)
        h->batch_block_stats = bs ? atoi(bs) : 256;
        if (h->batch_block_other < 64 || h->batch_block_other > 1024 || (h->batch_block_other & 63)) h->batch_block_other = 512;
        if (h->batch_block_stats < 64 || h->batch_block_stats > 1024 || (h->batch_block_stats & 63)) h->batch_block_stats = 256;
        h->no_out_alias = getenv("BLU_NO_OUT_ALIAS") ? 1 : 0;
        const char *pk = getenv("BLU_PIVOT_KERNEL");
        h->pivot_kernel = pk ? atoi(pk) : 0;
        if (h->pivot_kernel < 0 || h->pivot_kernel > 3) h->pivot_kernel = 0;
    }
    if (ok) { // chip-wide phases: as many workgroups as are certainly co-resident, at most 64 (one per CU of two XCDs' worth)
        int nb = 0, best = 1 << 30;
        const void *fns[4] = {(const void *)k_prep_grid, (const void *)k_setup_grid, (const void *)k_finish_grid, (const void *)k_rows_grid};
        for (int k = 0; k < 4; k++) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fns[k], 1024, 0) != hipSuccess) nb = 0;
            best = std::min(best, nb * prop.multiProcessorCount);
        }
        h->grid_blocks = std::max(1, std::min(best, 64));
        if (!prop.cooperativeLaunch) h->grid_blocks = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_pivot_loop_wave2, 128, 0) != hipSuccess) nb = 0;
        h->wave2_max = nb * prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_pivot_loop_wave2_r3, 128, 0) != hipSuccess) nb = 0;
        h->wave2_r3_max = nb * prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_pivot_loop_wave_r3, 64, 0) != hipSuccess) nb = 0;
        h->wave_r3_max = nb * prop.multiProcessorCount;
        if (const char *pr = getenv("BLU_PIVOT_REGS")) // (diagnostic) 4: never the _r3 variants
            if (atoi(pr) == 4) h->wave2_r3_max = h->wave_r3_max = 0;
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        {   // 144 KB of the CU's 160 KB as the counter window of k_prep / k_finish in a large batch.  Diagnostics:
            // BLU_LDS_WINDOW = 0 never, 1 (default) a batch of at least one basis per CU, 2 every batch;
            // BLU_LDS_WINDOW_BYTES = a smaller window (tests: several windows on small bases)
            const char *lw = getenv("BLU_LDS_WINDOW"), *lb = getenv("BLU_LDS_WINDOW_BYTES");
            h->lds_window_mode = lw ? atoi(lw) : 1;
            const int want = 144 * 1024;
            h->lds_window = 0;
            if ((size_t)prop.sharedMemPerBlock >= (size_t)want + 4096 &&
                hipFuncSetAttribute((const void *)k_prep<256>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                hipFuncSetAttribute((const void *)k_finish<256>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                hipFuncSetAttribute((const void *)k_prep<512>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                hipFuncSetAttribute((const void *)k_finish<512>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                hipFuncSetAttribute((const void *)k_setup<512>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                hipFuncSetAttribute((const void *)k_setup<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess)
                h->lds_window = want;
            if (lb && h->lds_window) {
                const int v = atoi(lb) & ~15;
                if (v >= 64 && v < want) h->lds_window = v;
            }
        }
        const char *bg = getenv("BLU_BATCH_GRID");
        h->batch_grid = bg ? atoi(bg) : 0;
        // the chain kernels keep ~106 KB of LDS rings per workgroup (k_chain.h)
        h->chain_ok = h->grid_blocks > 1 && (size_t)prop.sharedMemPerBlock >= sizeof(ChainLds) &&
                      hipFuncSetAttribute((const void *)k_stats_chains, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainLds)) == hipSuccess &&
                      hipFuncSetAttribute((const void *)k_solve_dense_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ChainLds)) == hipSuccess;
        if (!h->chain_ok) (void)hipGetLastError();
        if (getenv("BLU_HIP_NO_CHAIN")) h->chain_ok = 0; // diagnostic: the one-workgroup kernels of a batch
    }
    if (ok) ok = hip_ok(h, hipStreamCreate(&h->stream), "hipStreamCreate");
    for (int k = 0; ok && k < 4; k++) ok = hip_ok(h, hipEventCreate(&h->ev[k]), "hipEventCreate");
    if (!ok) {
        free_all(h);
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void blu_hip_free(blu_hip *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_all(h);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (int k = 0; k < 4; k++)
        if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    delete h;
}

extern "C" int blu_hip_set_param(blu_hip *h, int key, double v)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    switch (key) {
    case BLU_PARAM_DROPTOL: h->droptol = v; break;
    case BLU_PARAM_ABSTOL: h->abstol = v; break;
    case BLU_PARAM_RELTOL: h->reltol = v; break;
    case BLU_PARAM_NZBIAS: h->nzbias = v < 0 ? -1 : (int64_t)v; break;
    case BLU_PARAM_MAXSEARCH: h->maxsearch = (int64_t)v; break;
    case BLU_PARAM_PAD: h->pad = (int64_t)v; break;
    case BLU_PARAM_STRETCH: h->stretch = v; break;
    case BLU_PARAM_COMPRESS_THRES: h->compress_thres = v; break;
    case BLU_PARAM_SPARSE_THRES: h->sparse_thres = v; break;
    case BLU_PARAM_SEARCH_ROWS: h->search_rows = (int64_t)v; break;
    case BLU_PARAM_REALLOC_FACTOR: h->realloc_factor = v; break;
    default: return BLU_ERROR_INVALID_ARGUMENT;
    }
    return BLU_OK;
}
extern "C" double blu_hip_get_param(const blu_hip *h, int key)
{
    if (!h) return NAN;
    switch (key) {
    case BLU_PARAM_DROPTOL: return h->droptol;
    case BLU_PARAM_ABSTOL: return h->abstol;
    case BLU_PARAM_RELTOL: return h->reltol;
    case BLU_PARAM_NZBIAS: return (double)h->nzbias;
    case BLU_PARAM_MAXSEARCH: return (double)h->maxsearch;
    case BLU_PARAM_PAD: return (double)h->pad;
    case BLU_PARAM_STRETCH: return h->stretch;
    case BLU_PARAM_COMPRESS_THRES: return h->compress_thres;
    case BLU_PARAM_SPARSE_THRES: return h->sparse_thres;
    case BLU_PARAM_SEARCH_ROWS: return (double)h->search_rows;
    case BLU_PARAM_REALLOC_FACTOR: return h->realloc_factor;
    default: return NAN;
    }
}

extern "C" double blu_hip_get_stat(const blu_hip *h, int key)
{
    if (!h) return NAN;
    const Scalars &s = h->hs;
    switch (key) {
    case BLU_STAT_M: return (double)h->m;
    case BLU_STAT_NUPDATE: return (double)h->nupdate;
    case BLU_STAT_NFACTORIZE: return (double)h->nfactorize;
    case BLU_STAT_L_NZ: return (double)s.l_nz;
    // after an update these three live in the update state (update.rs:242-255, 625-626, 853-854, 943)
    case BLU_STAT_U_NZ: return h->nupdate > 0 ? (double)h->ust.u_nz : (double)s.u_nz;
    case BLU_STAT_MIN_PIVOT: return h->nupdate > 0 ? h->ust.min_pivot : s.min_pivot;
    case BLU_STAT_MAX_PIVOT: return h->nupdate > 0 ? h->ust.max_pivot : s.max_pivot;
    case BLU_STAT_CONDEST_L: return s.condest_l;
    case BLU_STAT_CONDEST_U: return s.condest_u;
    case BLU_STAT_NORM_L: return s.norm_l;
    case BLU_STAT_NORM_U: return s.norm_u;
    case BLU_STAT_NORMEST_L_INV: return s.normest_l_inv;
    case BLU_STAT_NORMEST_U_INV: return s.normest_u_inv;
    case BLU_STAT_ONENORM: return s.onenorm;
    case BLU_STAT_INFNORM: return s.infnorm;
    case BLU_STAT_RESIDUAL_TEST: return s.residual_test;
    case BLU_STAT_MATRIX_NZ: return (double)s.matrix_nz;
    case BLU_STAT_RANK: return (double)s.rank;
    case BLU_STAT_BUMP_SIZE: return (double)s.bump_size;
    case BLU_STAT_BUMP_NZ: return (double)s.bump_nz;
    case BLU_STAT_NSEARCH_PIVOT: return (double)s.nsearch_pivot;
    case BLU_STAT_NEXPAND: return (double)s.nexpand;
    case BLU_STAT_NGARBAGE: return (double)s.ngarbage;
    case BLU_STAT_FACTOR_FLOPS: return (double)s.factor_flops;
    case BLU_STAT_TIME_FACTORIZE: return h->t_total;
    // lu.time_singletons / time_search_pivot / time_elim_pivot (lu.rs:560-572): device seconds of the last factorize.  The
    // singleton phase is k_prep.  Search and elimination run inside ONE persistent kernel; its device time is split by
    // the shader-clock phase counters of the diagnostic build (`make prof`: prof[0] = search + set-up of the multi-wave
    // kernel) when they were collected, otherwise the whole kernel time is reported as elimination and the search as 0.
    case BLU_STAT_TIME_SINGLETONS: return h->t_phase[0];
    case BLU_STAT_TIME_SEARCH_PIVOT: {
        double tot = 0.0;
        for (int k = 0; k < 4; k++) tot += (double)s.prof[k];
        return tot > 0.0 ? h->t_pivot * (double)s.prof[0] / tot : 0.0;
    }
    case BLU_STAT_TIME_ELIM_PIVOT: {
        double tot = 0.0;
        for (int k = 0; k < 4; k++) tot += (double)s.prof[k];
        return tot > 0.0 ? h->t_pivot * (1.0 - (double)s.prof[0] / tot) : h->t_pivot;
    }
    case BLU_STAT_UPDATE_COST_DENOM: // factorize.rs:160-166
        return 250.0 * (0.04 * (double)h->m + 0.07 * (double)s.matrix_nz + 0.20 * (double)s.bump_nz +
                        0.20 * (double)s.nsearch_pivot + 0.008 * (double)s.factor_flops);
    case BLU_STAT_RANKDEF: return (double)s.rankdef;
    case BLU_STAT_L_MEM: return (double)h->D.lcap;
    case BLU_STAT_U_MEM: return (double)h->D.ucap;
    case BLU_STAT_W_MEM: return (double)h->D.carena_cap + (double)h->D.rarena_cap;
    case BLU_STAT_L_FLOPS: return (double)h->sp_l_flops;
    case BLU_STAT_U_FLOPS: return (double)h->sp_u_flops;
    case BLU_STAT_NFORREST: return h->upd_for_nfact == h->nfactorize ? (double)h->ust.nforrest : 0.0;
    case BLU_STAT_PIVOT_ERROR: return h->upd_for_nfact == h->nfactorize ? h->ust.pivot_error : 0.0; // (per factorization: lu.rs:330-346)
    case BLU_STAT_R_NZ: return h->upd_for_nfact == h->nfactorize ? (double)h->ust.r_nz : 0.0;
    case BLU_STAT_R_FLOPS: return h->upd_for_nfact == h->nfactorize ? (double)h->ust.r_flops : 0.0;
    case BLU_STAT_MAX_ETA: return h->upd_for_nfact == h->nfactorize ? h->ust.max_eta : 0.0;
    case BLU_STAT_NSYMPERM_TOTAL: return (double)h->ust.nsymperm_total;
    case BLU_STAT_NFORREST_TOTAL: return (double)h->ust.nforrest_total;
    case BLU_STAT_DEV_NUNSYMPERM_TOTAL: return (double)h->ust.nunsymperm_total;
    case BLU_STAT_UPDATE_COST: // lu.rs:324-326; the numerator is reset by factorize (lu.rs:346)
        return (h->upd_for_nfact == h->nfactorize ? h->ust.update_cost_numer : 0.0) / blu_hip_get_stat(h, BLU_STAT_UPDATE_COST_DENOM);
    case 44: case 45: case 46: case 47: return h->t_phase[key - 44]; // device seconds of k_prep / k_setup / k_finish / k_stats
    case 108: case 109: return h->t_phase[key - 108 + 4]; // inside the statistics: k_rows_grid, k_stats_tail
    case 43: return (double)h->sp_branch; // branch of the last solve_sparse: 1 sparse, 2 sequential
    case BLU_STAT_DEV_TIME_PIVOT_LOOP: return h->t_pivot;
    case BLU_STAT_DEV_TIME_TOTAL: return h->t_total;
    case BLU_STAT_DEV_RELAUNCHES: return (double)h->relaunches;
    case 50: return (double)s.d3_hits;       // BLU_STAT_DEV_D3_HITS
    case 51: return (double)s.npivot_kind[0];
    case 52: return (double)s.npivot_kind[1];
    case 53: return (double)s.npivot_kind[2];
    case 54: return (double)s.npivot_kind[3];
    case 55: return (double)s.npivot_kind[4];
    case 56: return (double)s.npivot_kind[5];
    case 110: case 111: case 116: case 117: return (double)s.nfast[key < 116 ? key - 110 : key - 114]; // pivots taken by the flattened paths of k_pivot_loop_wave
    case 112: return (double)s.cused;  // entries of the column / row arena handed out (bump pointers)
    case 113: return (double)s.rused;
    case 114: return (double)h->D.carena_cap;
    case 115: return (double)h->D.lcap;
    case 120: return (double)h->last_pivot_regs; // waves per SIMD its register budget was set for (the _r3 variants of the wave kernels: 3; else 4)
    case 118: return (double)h->last_pivot_kernel; // which pivot kernel the last factorize of this handle ran: 0 k_pivot_loop, 1 k_pivot_loop_wave, 2 k_pivot_loop_batch, 3 k_pivot_loop_wave2
    case 119: return (double)s.fill_paths; // bit 0 / bit 1: k_prep / k_finish filled through buckets (k_bucket.h)
    case 57: return (double)s.err_line;
    case 58: return (double)s.status;
    case 60: case 61: case 62: case 63: case 64: case 65: case 66: case 67: case 68: case 69: case 70: case 71: case 72: case 73: case 74: case 75:
    case 76: case 77: case 78: case 79: case 80: case 81: case 82: case 83:
    case 84: case 85: case 86: case 87: case 88: case 89: case 90: case 91:
    case 92: case 93: case 94: case 95: case 96: case 97: case 98: case 99: case 100: case 101: case 102: case 103: case 104: case 105: case 106: case 107:
        return (double)s.prof[key - 60]; // diagnostic build only
    default: return NAN;
    }
}

extern "C" const char *blu_hip_last_error(const blu_hip *h) { return h ? h->err.c_str() : "null handle"; }
extern "C" const char *blu_hip_version(void)
{
#if defined(BLU_EWCHECK)
    return "blu_hip 0.2 (gfx950; self-checking build: early / speculative searches verified in the kernel)";
#elif defined(BLU_PROFILE)
    return "blu_hip 0.2 (gfx950; phase-timing build)";
#else
    return "blu_hip 0.2 (gfx950)";
#endif
}
extern "C" int blu_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// memory growth (the device-side counterpart of lu_realloc_obj, blu.rs:345-377)
// ---------------------------------------------------------------------------------------------
// gwork holds `cols` all-zero columns of m + 1 doubles (the state its users expect and restore)
static bool ensure_gwork(blu_hip *h, int cols)
{
    if (cols <= h->gwork_cols) return true;
    DevLU &D = h->D;
    const size_t n = (size_t)cols * ((size_t)h->m + 1);
    dfree(D.gwork);
    h->gwork_cols = 0;
    if (!dalloc(h, &D.gwork, n) || !hip_ok(h, hipMemset(D.gwork, 0, n * sizeof(double)), "hipMemset")) return false;
    h->gwork_cols = cols;
    return true;
}
static int64_t grown(blu_hip *h, int64_t cap, int64_t need_extra)
{
    const double f = std::max(1.0, h->realloc_factor);
    int64_t n = (int64_t)((double)(cap + need_extra) * f) + 64;
    return std::min<int64_t>(n, kIntMax);
}
static bool grow_l(blu_hip *h, int need)
{
    DevLU &D = h->D;
    const int64_t n = grown(h, D.lcap, need);
    if (n <= D.lcap) { h->err = "L storage limit (2^31 entries) reached"; return false; }
    if (!dgrow(h, &D.lidx, (size_t)h->hs.lused, (size_t)n) || !dgrow(h, &D.lval, (size_t)h->hs.lused, (size_t)n)) return false;
    D.lcap = (int)n;
    return true;
}
static bool grow_u(blu_hip *h, int need)
{
    DevLU &D = h->D;
    const int64_t n = grown(h, D.ucap, need);
    if (n <= D.ucap) { h->err = "U storage limit (2^31 entries) reached"; return false; }
    if (!dgrow(h, &D.uidx, (size_t)h->hs.uused, (size_t)n) || !dgrow(h, &D.uval, (size_t)h->hs.uused, (size_t)n)) return false;
    D.ucap = (int)n;
    return true;
}
// compact files into new, larger arenas (file_compress + realloc), for every handle of a batch that asked for it, in
// ONE launch (a batch of similar bases asks for it together: one launch per handle cost 3.3 ms each at C3 size, 5 s
// for 1536 handles).  req[k] = 0 column file, 1 row file, < 0 nothing; need[k] = entries wanted beyond the old capacity;
// slots = the batch's descriptor array; ok[k] = false where the new arena could not be had.
static bool compact_files(blu_hip *const *hs, int n, DevLU *slots, hipStream_t stream, const std::vector<int> &req, const std::vector<int> &need,
                          std::vector<char> &ok)
{
    blu_hip *h0 = hs[0];
    std::vector<int *> nidx(n, nullptr);
    std::vector<double *> nval(n, nullptr);
    std::vector<int> ncap(n, 0), which(n, -1);
    ok.assign(n, 1);
    int **d_pi = nullptr;
    double **d_pv = nullptr;
    int *d_cap = nullptr, *d_which = nullptr;
    bool good = true;
    // Old and new arena of a handle exist side by side until its files are moved: a batch that fills the card (3072 bases
    // of the 50k size) cannot have that for all its handles at once.  So: rounds, each as many handles as fit in most of
    // the memory that is free now.
    for (int start = 0; start < n && good;) {
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) freeb = 0;
        const size_t budget = freeb - freeb / 4;
        size_t used = 0;
        bool any = false;
        int k = start;
        std::fill(which.begin(), which.end(), -1);
        for (; k < n; k++) {
            if (req[k] < 0) continue;
            blu_hip *h = hs[k];
            DevLU &D = h->D;
            const int64_t oldcap = req[k] ? D.rarena_cap : D.carena_cap;
            const int64_t nn = grown(h, oldcap, need[k]);
            if (nn <= oldcap) { h->err = "arena limit (2^31 entries) reached"; ok[k] = 0; continue; }
            const size_t bytes = (size_t)nn * (req[k] ? sizeof(int) : sizeof(int) + sizeof(double));
            if (any && used + bytes > budget) break; // the next round (a round takes one handle at least)
            if (!dalloc(h, &nidx[k], (size_t)nn) || (!req[k] && !dalloc(h, &nval[k], (size_t)nn))) {
                dfree(nidx[k]);
                nidx[k] = nullptr;
                ok[k] = 0;
                continue;
            }
            used += bytes;
            ncap[k] = (int)nn;
            which[k] = req[k];
            any = true;
        }
        const int end = k;
        if (any) {
            good = (d_pi || (dalloc(h0, &d_pi, n) && dalloc(h0, &d_pv, n) && dalloc(h0, &d_cap, n) && dalloc(h0, &d_which, n))) &&
                   hip_ok(h0, hipMemcpy(d_pi, nidx.data(), sizeof(int *) * n, hipMemcpyHostToDevice), "h2d arenas") &&
                   hip_ok(h0, hipMemcpy(d_pv, nval.data(), sizeof(double *) * n, hipMemcpyHostToDevice), "h2d arenas") &&
                   hip_ok(h0, hipMemcpy(d_cap, ncap.data(), sizeof(int) * n, hipMemcpyHostToDevice), "h2d arenas") &&
                   hip_ok(h0, hipMemcpy(d_which, which.data(), sizeof(int) * n, hipMemcpyHostToDevice), "h2d arenas");
            if (good) {
                hipLaunchKernelGGL(k_compact, dim3(n), dim3(1024), 0, stream, slots, d_which, d_pi, d_pv, d_cap);
                good = hip_ok(h0, hipStreamSynchronize(stream), "k_compact");
            }
            for (int q = start; q < end; q++) {
                if (which[q] < 0) continue;
                blu_hip *h = hs[q];
                DevLU &D = h->D;
                if (!good) {
                    dfree(nidx[q]); dfree(nval[q]);
                    ok[q] = 0;
                    continue;
                }
                if (which[q]) {
                    dfree(D.ridx);
                    D.ridx = nidx[q];
                    D.rarena_cap = ncap[q];
                } else {
                    dfree(D.cidx);
                    dfree(D.cval);
                    D.cidx = nidx[q];
                    D.cval = nval[q];
                    D.carena_cap = ncap[q];
                }
                if (!upload_desc(h)) ok[q] = 0;
            }
        }
        start = end;
    }
    dfree(d_pi); dfree(d_pv); dfree(d_cap); dfree(d_which);
    return good;
}

// Room for the canonical factors (get_factors.rs:35-43: m + l_nz and m + u_nz entries of 16 bytes).  When the pivot loop
// is over the active submatrix is gone and its column arena is dead storage until the next factorize -- which
// invalidates these factors anyway -- so the canonical L goes into the arena's index array and the canonical U into its
// value array whenever they fit (at C3: 13 of 15 MB and 25 of 30 MB; 38 MB less per handle, and HBM capacity is what
// limits the number of bases of a batch).  Nothing but the pivot loop, k_compact and the debug dumps of a STOPPED
// factorization reads the arenas.  Buffers of their own otherwise.
// upload = false: the caller copies h->O to the handle's slot itself (a batch stages all of them in one copy)
static bool ensure_out(blu_hip *h, int64_t ln, int64_t un, bool upload = true)
{
    const DevLU &D = h->D;
    const bool fits = !h->no_out_alias && (size_t)ln * 16 <= (size_t)D.carena_cap * sizeof(int) && (size_t)un * 16 <= (size_t)D.carena_cap * sizeof(double);
    if (fits) {
        if (!h->out_in_arena) { // give the separate buffers back
            dfree(h->O.l_rowidx); dfree(h->O.l_value); dfree(h->O.u_rowidx); dfree(h->O.u_value);
            h->out_lcap = h->out_ucap = 0;
        }
        h->O.l_rowidx = (long long *)D.cidx;
        h->O.l_value = (double *)((char *)D.cidx + (size_t)ln * 8);
        h->O.u_rowidx = (long long *)D.cval;
        h->O.u_value = (double *)((char *)D.cval + (size_t)un * 8);
        h->out_in_arena = true;
    } else {
        if (h->out_in_arena) {
            h->O.l_rowidx = h->O.u_rowidx = nullptr;
            h->O.l_value = h->O.u_value = nullptr;
            h->out_lcap = h->out_ucap = 0;
            h->out_in_arena = false;
        }
        if (ln > h->out_lcap) {
            dfree(h->O.l_rowidx); dfree(h->O.l_value);
            if (!dalloc(h, &h->O.l_rowidx, (size_t)ln) || !dalloc(h, &h->O.l_value, (size_t)ln)) return false;
            h->out_lcap = ln;
        }
        if (un > h->out_ucap) {
            dfree(h->O.u_rowidx); dfree(h->O.u_value);
            if (!dalloc(h, &h->O.u_rowidx, (size_t)un) || !dalloc(h, &h->O.u_value, (size_t)un)) return false;
            h->out_ucap = un;
        }
    }
    if (upload) HIP_TRY(h, hipMemcpy(h->oslot, &h->O, sizeof(FinishOut), hipMemcpyHostToDevice));
    return true;
}

static int ensure_rows_ws(blu_hip *h, int64_t lnz, int64_t unz);
static RowsWs rows_ws_of(blu_hip *h);
static bool launch_rows(blu_hip *h, DevLU *dD, hipStream_t stream);
#include "blu_driver.inc"

// BLU::get_factors -- src/blu.rs:139, get_factors.rs:48-180
extern "C" int blu_hip_get_factors(blu_hip *h, int64_t *rowperm, int64_t *colperm,
                                   int64_t *l_colptr, int64_t *l_rowidx, double *l_value,
                                   int64_t *u_colptr, int64_t *u_rowidx, double *u_value)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    if (h->nupdate != 0) return BLU_ERROR_INVALID_CALL; // get_factors.rs:59
    if (h->m == 0) {
        if (l_colptr) l_colptr[0] = 0;
        if (u_colptr) u_colptr[0] = 0;
        return BLU_OK;
    }
    if (hipSetDevice(h->device) != hipSuccess) return BLU_ERROR_DEVICE;
    const size_t M = (size_t)h->m;
    const size_t ln = (size_t)h->hs.l_nz + M, un = (size_t)h->hs.u_nz + M;
    bool ok = true;
    if (rowperm) ok = ok && hip_ok(h, hipMemcpy(rowperm, h->O.rowperm, M * 8, hipMemcpyDeviceToHost), "d2h rowperm");
    if (colperm) ok = ok && hip_ok(h, hipMemcpy(colperm, h->O.colperm, M * 8, hipMemcpyDeviceToHost), "d2h colperm");
    if (l_colptr && l_rowidx && l_value) {
        ok = ok && hip_ok(h, hipMemcpy(l_colptr, h->O.l_colptr, (M + 1) * 8, hipMemcpyDeviceToHost), "d2h l_colptr");
        ok = ok && hip_ok(h, hipMemcpy(l_rowidx, h->O.l_rowidx, ln * 8, hipMemcpyDeviceToHost), "d2h l_rowidx");
        ok = ok && hip_ok(h, hipMemcpy(l_value, h->O.l_value, ln * 8, hipMemcpyDeviceToHost), "d2h l_value");
    }
    if (u_colptr && u_rowidx && u_value) {
        ok = ok && hip_ok(h, hipMemcpy(u_colptr, h->O.u_colptr, (M + 1) * 8, hipMemcpyDeviceToHost), "d2h u_colptr");
        ok = ok && hip_ok(h, hipMemcpy(u_rowidx, h->O.u_rowidx, un * 8, hipMemcpyDeviceToHost), "d2h u_rowidx");
        ok = ok && hip_ok(h, hipMemcpy(u_value, h->O.u_value, un * 8, hipMemcpyDeviceToHost), "d2h u_value");
    }
    return ok ? BLU_OK : BLU_ERROR_DEVICE;
}

// workspace of the solves that walk the factors line by line (allocated at the first use)
static int ensure_sparse_ws(blu_hip *h)
{
    if (h->sw_ready) return BLU_OK;
    const size_t M = (size_t)h->m;
    SparseWs &W = h->sw;
    bool a = dalloc(h, &W.marked, M) && dalloc(h, &W.psym, M) && dalloc(h, &W.pat, M) && dalloc(h, &W.pstack, M) && dalloc(h, &W.estack, M) &&
             dalloc(h, &W.work, M) && dalloc(h, &W.xlhs, M) && dalloc(h, &W.ilhs, M) && dalloc(h, &W.xval, M) &&
             dalloc(h, &W.out, 4) && dalloc(h, &W.lt_ptr, M + 1) && dalloc(h, &W.lt_cur, M);
    a = a && hip_ok(h, hipMemset(W.marked, 0, M * sizeof(int)), "hipMemset") &&
        hip_ok(h, hipMemset(W.work, 0, M * sizeof(double)), "hipMemset") &&
        hip_ok(h, hipMemset(W.xlhs, 0, M * sizeof(double)), "hipMemset");
    if (!a) return BLU_ERROR_OUT_OF_MEMORY;
    h->sw_ready = true;
    h->marker = 0;
    return BLU_OK;
}
// row-wise L of THIS factorization (build_factors.rs:243-274), built on the device at the first solve that needs it
static int ensure_lt(blu_hip *h)
{
    const int st = ensure_sparse_ws(h);
    if (st != BLU_OK) return st;
    if (h->lt_for_nfact == h->nfactorize) return BLU_OK;
    SparseWs &W = h->sw;
    const int64_t lnz = std::max<int64_t>((int64_t)h->hs.lused, 1);
    if (lnz > h->sw_ltcap) {
        dfree(W.lt_idx);
        dfree(W.lt_val);
        if (!dalloc(h, &W.lt_idx, (size_t)lnz) || !dalloc(h, &W.lt_val, (size_t)lnz)) return BLU_ERROR_OUT_OF_MEMORY;
        h->sw_ltcap = lnz;
    }
    hipLaunchKernelGGL(k_build_lt, dim3(1), dim3(1024), 0, h->stream, h->dD, W);
    if (!hip_ok(h, hipStreamSynchronize(h->stream), "k_build_lt")) return BLU_ERROR_DEVICE;
    h->lt_for_nfact = h->nfactorize;
    return BLU_OK;
}

// Row-wise copies for the chain pipeline (single-matrix path): row-wise L into the buffers ensure_lt owns, U rows
// sorted descending; k_rows_grid on `stream`.  `nfact` = the nfactorize value these factors will carry.
static int ensure_rows_ws(blu_hip *h, int64_t lnz, int64_t unz)
{
    const int st = ensure_sparse_ws(h);
    if (st != BLU_OK) return st;
    SparseWs &W = h->sw;
    lnz = std::max<int64_t>(lnz, 1);
    unz = std::max<int64_t>(unz, 1);
    if (lnz > h->sw_ltcap) {
        dfree(W.lt_idx);
        dfree(W.lt_val);
        if (!dalloc(h, &W.lt_idx, (size_t)lnz) || !dalloc(h, &W.lt_val, (size_t)lnz)) return BLU_ERROR_OUT_OF_MEMORY;
        h->sw_ltcap = lnz;
        h->lt_for_nfact = -1;
    }
    if (!h->ur_len && !dalloc(h, &h->ur_len, (size_t)h->m)) return BLU_ERROR_OUT_OF_MEMORY;
    if (unz > h->ur_cap) {
        dfree(h->ur_pos);
        dfree(h->ur_val);
        if (!dalloc(h, &h->ur_pos, (size_t)unz) || !dalloc(h, &h->ur_val, (size_t)unz)) return BLU_ERROR_OUT_OF_MEMORY;
        h->ur_cap = unz;
        h->rows_for_nfact = -1;
    }
    return BLU_OK;
}
static RowsWs rows_ws_of(blu_hip *h)
{
    RowsWs R;
    R.lt_ptr = h->sw.lt_ptr;
    R.lt_idx = h->sw.lt_idx;
    R.lt_val = h->sw.lt_val;
    R.lt_cur = h->sw.lt_cur;
    R.ur_len = h->ur_len;
    R.ur_pos = h->ur_pos;
    R.ur_val = h->ur_val;
    return R;
}
static bool launch_rows(blu_hip *h, DevLU *dD, hipStream_t stream)
{
    if (hipMemsetAsync(h->gw, 0, sizeof(GridWs), stream) != hipSuccess) return false;
    RowsWs R = rows_ws_of(h);
    void *a0 = (void *)dD, *a1 = (void *)h->gw;
    void *args[3] = {&a0, &a1, &R};
    if (hipLaunchCooperativeKernel((const void *)k_rows_grid, dim3(h->grid_blocks), dim3(1024), args, 0, stream) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

#include "blu_update.inc"

// BLU::solve_dense -- src/blu.rs:182, lu/solve_dense.rs:7-120
extern "C" int blu_hip_solve_dense(blu_hip *h, const double *rhs, double *lhs, char trans)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    if (h->nupdate < 0) return BLU_ERROR_INVALID_CALL; // solve_dense.rs:25-27
    if (!rhs || !lhs) return BLU_ERROR_ARGUMENT_MISSING;
    if (h->m == 0) return BLU_OK;
    if (hipSetDevice(h->device) != hipSuccess) return BLU_ERROR_DEVICE;
    const size_t M = (size_t)h->m;
    if (!h->d_rhs && (!dalloc(h, &h->d_rhs, M) || !dalloc(h, &h->d_lhs, M))) return BLU_ERROR_OUT_OF_MEMORY;
    if (!hip_ok(h, hipMemcpy(h->d_rhs, rhs, M * 8, hipMemcpyHostToDevice), "h2d rhs")) return BLU_ERROR_DEVICE;
    const int tr = (trans == 't' || trans == 'T') ? 1 : 0;
    if (h->nupdate > 0) { // updated factorization: mutable U, row etas, pivot sequence (k_update.hip)
        const int st = solve_dense_updated(h, tr);
        if (st != BLU_OK) return st;
        if (!hip_ok(h, hipMemcpy(lhs, h->d_lhs, M * 8, hipMemcpyDeviceToHost), "d2h lhs")) return BLU_ERROR_DEVICE;
        return BLU_OK;
    }
    // a factorization whose row-wise copies exist (a single factorize on a device with the LDS for it): the chain pipeline
    if (h->chain_ok && h->rows_for_nfact == h->nfactorize && h->lt_for_nfact == h->nfactorize) {
        hipLaunchKernelGGL(k_solve_dense_chain, dim3(1), dim3(CHAIN_THREADS), sizeof(ChainLds), h->stream, h->dD, h->dO, rows_ws_of(h), h->d_rhs,
                           h->d_lhs, tr, &h->gw->ctr[7]);
        if (!hip_ok(h, hipStreamSynchronize(h->stream), "k_solve_dense_chain")) return BLU_ERROR_DEVICE;
        int defect = 0;
        if (!hip_ok(h, hipMemcpy(&defect, &h->gw->ctr[7], sizeof(int), hipMemcpyDeviceToHost), "d2h defect")) return BLU_ERROR_DEVICE;
        if (defect) { // a bounded wait of the chain pipeline gave up (k_chain.h): never a valid state
            h->err = "k_solve_dense_chain: internal wait abandoned, code " + std::to_string(defect);
            return BLU_ERROR_DEVICE;
        }
        if (!hip_ok(h, hipMemcpy(lhs, h->d_lhs, M * 8, hipMemcpyDeviceToHost), "d2h lhs")) return BLU_ERROR_DEVICE;
        return BLU_OK;
    }
    if (!tr) { // the forward L solve takes row dots (solve_dense.rs:79-86)
        const int st = ensure_lt(h);
        if (st != BLU_OK) return st;
    }
    hipLaunchKernelGGL(k_solve_dense, dim3(1), dim3(1024), 0, h->stream, h->dD, h->dO, h->d_rhs, h->d_lhs, tr, h->sw.lt_ptr, h->sw.lt_idx,
                       h->sw.lt_val);
    if (!hip_ok(h, hipStreamSynchronize(h->stream), "k_solve_dense")) return BLU_ERROR_DEVICE;
    if (!hip_ok(h, hipMemcpy(lhs, h->d_lhs, M * 8, hipMemcpyDeviceToHost), "d2h lhs")) return BLU_ERROR_DEVICE;
    return BLU_OK;
}

// solve_sparse -- src/solve_sparse.rs:36-68, lu/solve_sparse.rs:11-360 (fresh factorization: nforrest == 0)
extern "C" int blu_hip_solve_sparse(blu_hip *h, int64_t nzrhs, const uint64_t *irhs, const double *xrhs, int64_t *p_nzlhs,
                                    int64_t *ilhs, double *lhs, char trans)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    if (h->nupdate < 0) return BLU_ERROR_INVALID_CALL; // solve_sparse.rs:46-47
    if (!p_nzlhs || !ilhs || !lhs || (nzrhs > 0 && (!irhs || !xrhs))) return BLU_ERROR_ARGUMENT_MISSING;
    // check RHS indices (solve_sparse.rs:49-59)
    bool ok = nzrhs >= 0 && nzrhs <= h->m;
    for (int64_t n = 0; ok && n < nzrhs; n++) ok = irhs[n] < (uint64_t)h->m;
    if (!ok) return BLU_ERROR_INVALID_ARGUMENT;
    *p_nzlhs = 0;
    if (h->m == 0) return BLU_OK;
    if (hipSetDevice(h->device) != hipSuccess) return BLU_ERROR_DEVICE;
    const size_t M = (size_t)h->m;
    SparseWs &W = h->sw;
    {
        const int st = ensure_sparse_ws(h);
        if (st != BLU_OK) return st;
    }
    if (h->marker > 0x7fffffff - 8) { // lu.rs:301-305: reset the marks before the marker overflows
        if (!hip_ok(h, hipMemset(W.marked, 0, M * sizeof(int)), "hipMemset")) return BLU_ERROR_DEVICE;
        h->marker = 0;
    }
    const int tr = (trans == 't' || trans == 'T') ? 1 : 0;
    if (h->nupdate > 0) return solve_sparse_updated(h, nzrhs, irhs, xrhs, p_nzlhs, ilhs, lhs, tr);
    if (tr) { // the transposed system ends with L': row-wise L
        const int st = ensure_lt(h);
        if (st != BLU_OK) return st;
    }
    if (nzrhs > h->rhs_cap) {
        dfree(h->d_irhs);
        dfree(h->d_xrhs);
        if (!dalloc(h, &h->d_irhs, (size_t)nzrhs) || !dalloc(h, &h->d_xrhs, (size_t)nzrhs)) return BLU_ERROR_OUT_OF_MEMORY;
        h->rhs_cap = nzrhs;
    }
    if (nzrhs > 0) {
        std::vector<int> ir((size_t)nzrhs);
        for (int64_t n = 0; n < nzrhs; n++) ir[(size_t)n] = (int)irhs[n];
        if (!hip_ok(h, hipMemcpy(h->d_irhs, ir.data(), (size_t)nzrhs * sizeof(int), hipMemcpyHostToDevice), "h2d irhs") ||
            !hip_ok(h, hipMemcpy(h->d_xrhs, xrhs, (size_t)nzrhs * sizeof(double), hipMemcpyHostToDevice), "h2d xrhs"))
            return BLU_ERROR_DEVICE;
    }
    const int nz_sparse = (int)(h->sparse_thres * (double)h->m); // lu/solve_sparse.rs:24
    hipLaunchKernelGGL(k_solve_sparse, dim3(1), dim3(64), 0, h->stream, h->dD, h->dO, W, (int)nzrhs, h->d_irhs, h->d_xrhs, tr,
                       h->marker, nz_sparse);
    if (!hip_ok(h, hipStreamSynchronize(h->stream), "k_solve_sparse")) return BLU_ERROR_DEVICE;
    h->marker += 3;
    long long out[4];
    if (!hip_ok(h, hipMemcpy(out, W.out, sizeof out, hipMemcpyDeviceToHost), "d2h out")) return BLU_ERROR_DEVICE;
    const size_t nz = (size_t)out[0];
    h->sp_l_flops += out[1];
    h->sp_u_flops += out[2];
    h->sp_branch = (int)out[3];
    if (nz > 0) {
        std::vector<int> il(nz);
        std::vector<double> xv(nz);
        if (!hip_ok(h, hipMemcpy(il.data(), W.ilhs, nz * sizeof(int), hipMemcpyDeviceToHost), "d2h ilhs") ||
            !hip_ok(h, hipMemcpy(xv.data(), W.xval, nz * sizeof(double), hipMemcpyDeviceToHost), "d2h xval"))
            return BLU_ERROR_DEVICE;
        for (size_t n = 0; n < nz; n++) { // scatter into the caller's (all-zero) lhs
            ilhs[n] = il[n];
            lhs[il[n]] = xv[n];
        }
    }
    *p_nzlhs = (int64_t)nz;
    return BLU_OK;
}

// ---------------------------------------------------------------------------------------------
// debug / test hooks (step-wise comparison with the oracle); not part of the drop-in surface
// ---------------------------------------------------------------------------------------------
extern "C" int blu_hip_dbg_set_stop(blu_hip *h, int64_t stop_at)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    h->stop_at = stop_at;
    return BLU_OK;
}
// 1 = skip the statistics tail of factorize() (condest, residual_test); default 0 = compute, as the reference
#ifdef BLU_STATS_DEBUG
extern "C" int blu_hip_dbg_get_rows(blu_hip *h, int *lt_ptr, int *lt_idx, double *lt_val, int *ur_len, int *ur_pos, double *ur_val, int *ubeg)
{
    const size_t M = (size_t)h->m, ln = (size_t)h->hs.lused, un = (size_t)h->hs.uused;
    bool ok = hipMemcpy(lt_ptr, h->sw.lt_ptr, (M + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(lt_idx, h->sw.lt_idx, ln * 4, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(lt_val, h->sw.lt_val, ln * 8, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(ur_len, h->ur_len, M * 4, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(ur_pos, h->ur_pos, un * 4, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(ur_val, h->ur_val, un * 8, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(ubeg, h->D.ubeg, (M + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess;
    return ok ? 0 : -1;
}
extern "C" int blu_hip_dbg_get_gwork(blu_hip *h, double *out, int64_t n)
{
    return hipMemcpy(out, h->D.gwork, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int blu_hip_set_skip_stats(blu_hip *h, int on)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    h->skip_stats = on ? 1 : 0;
    return BLU_OK;
}
extern "C" int blu_hip_dbg_set_batch_block(blu_hip *h, int threads)
{
    if (!h || threads < 64 || threads > 1024 || (threads & 63)) return BLU_ERROR_INVALID_ARGUMENT;
    h->batch_block = threads;
    return BLU_OK;
}
// workgroups of the chip-wide O(nnz) phases of a single factorize (1: one workgroup, as inside a batch)
extern "C" int blu_hip_dbg_set_grid_blocks(blu_hip *h, int nblocks)
{
    if (!h || nblocks < 1 || nblocks > SCOPE_MAX_BLOCKS) return BLU_ERROR_INVALID_ARGUMENT;
    h->grid_blocks = nblocks;
    return BLU_OK;
}
// 0 = default, 1 = one wave per matrix (k_pivot_loop_wave), 2 = multi-wave workgroups (k_pivot_loop / k_pivot_loop_batch),
// 3 = two waves per matrix (k_pivot_loop_wave2)
extern "C" int blu_hip_dbg_set_pivot_kernel(blu_hip *h, int which)
{
    if (!h || which < 0 || which > 3) return BLU_ERROR_INVALID_ARGUMENT;
    h->pivot_kernel = which;
    return BLU_OK;
}
extern "C" int blu_hip_dbg_set_no_fast(blu_hip *h, int on)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    h->no_fast = on ? 1 : 0;
    return BLU_OK;
}
extern "C" int blu_hip_dbg_set_block(blu_hip *h, int threads)
{
    if (!h || threads < 64 || threads > 1024 || (threads & 63)) return BLU_ERROR_INVALID_ARGUMENT;
    h->block_threads = threads;
    return BLU_OK;
}
// continue a factorization that returned 100 (stopped)
extern "C" int blu_hip_dbg_continue(blu_hip *h, int64_t stop_at)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    if (hipSetDevice(h->device) != hipSuccess) return BLU_ERROR_DEVICE;
    h->stop_at = stop_at;
    if (!download_scalars(h)) return BLU_ERROR_DEVICE;
    if (h->hs.status != ST_STOPPED) return BLU_ERROR_INVALID_CALL;
    if (!set_status(h, ST_RUNNING)) return BLU_ERROR_DEVICE;
    return continue_pivot_single(h);
}
template <class T> static bool d2h_vec(blu_hip *h, std::vector<T> &v, const T *d, size_t n)
{
    v.resize(std::max<size_t>(n, 1));
    if (!n) return true;
    return hip_ok(h, hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost), "d2h dbg");
}
// the line records and list heads of a handle on the host (debug dumps)
struct HostLines {
    std::vector<LineRec> crec, rrec;
    std::vector<HeadRec> chead, rhead;
};
static bool d2h_lines(blu_hip *h, HostLines &L)
{
    const size_t M = (size_t)h->m;
    return d2h_vec(h, L.crec, h->D.crec, M) && d2h_vec(h, L.rrec, h->D.rrec, M) && d2h_vec(h, L.chead, h->D.chead, M + 2) &&
           d2h_vec(h, L.rhead, h->D.rhead, M + 2);
}
// which: 0 column-file entries, 1 row-file entries, 2 L entries so far, 3 U entries so far
extern "C" int64_t blu_hip_dbg_count(blu_hip *h, int which)
{
    if (!h) return -1;
    (void)hipSetDevice(h->device);
    if (!download_scalars(h)) return -1;
    if (which == 2) return h->hs.lused;
    if (which == 3) return h->hs.uused;
    std::vector<LineRec> rec;
    if (!d2h_vec(h, rec, which ? h->D.rrec : h->D.crec, (size_t)h->m)) return -1;
    int64_t n = 0;
    for (int64_t k = 0; k < h->m; k++) n += rec[k].len;
    return n;
}
// same layout as oracle/orc_debug.c: orc_dbg_active_state
extern "C" int blu_hip_dbg_active_state(blu_hip *h, int64_t *colptr, int64_t *colidx, double *colval,
                                        int64_t *rowptr, int64_t *rowidx, double *colmax, int64_t *pinv, int64_t *qinv,
                                        int64_t *col_flink, int64_t *col_blink, int64_t *row_flink, int64_t *row_blink)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    (void)hipSetDevice(h->device);
    const size_t M = (size_t)h->m;
    DevLU &D = h->D;
    std::vector<int> cbeg(M), clen(M), rbeg(M), rlen(M), cidx, ridx, pi, qi, cf(2 * M + 2), cb(2 * M + 2), rf(2 * M + 2), rb(2 * M + 2);
    std::vector<double> cval;
    HostLines HL;
    bool ok = d2h_lines(h, HL);
    if (ok) { // the arrays of list.rs / file.rs out of the records: elements 0..m-1, heads m..2m+1
        for (size_t j = 0; j < M; j++) {
            cbeg[j] = HL.crec[j].beg; clen[j] = HL.crec[j].len; rbeg[j] = HL.rrec[j].beg; rlen[j] = HL.rrec[j].len;
            cf[j] = HL.crec[j].flink; cb[j] = HL.crec[j].blink; rf[j] = HL.rrec[j].flink; rb[j] = HL.rrec[j].blink;
            colmax[j] = HL.crec[j].max;
        }
        for (size_t k = 0; k < M + 2; k++) {
            cf[M + k] = HL.chead[k].flink; cb[M + k] = HL.chead[k].blink; rf[M + k] = HL.rhead[k].flink; rb[M + k] = HL.rhead[k].blink;
        }
    }
    ok = ok && d2h_vec(h, cidx, D.cidx, (size_t)D.carena_cap) && d2h_vec(h, cval, D.cval, (size_t)D.carena_cap) && d2h_vec(h, ridx, D.ridx, (size_t)D.rarena_cap);
    ok = ok && d2h_vec(h, pi, D.pinv, M) && d2h_vec(h, qi, D.qinv, M);
    if (!ok) return BLU_ERROR_DEVICE;
    int64_t put = 0;
    for (size_t j = 0; j < M; j++) {
        colptr[j] = put;
        for (int p = 0; p < clen[j]; p++) {
            colidx[put] = cidx[cbeg[j] + p];
            colval[put] = cval[cbeg[j] + p];
            put++;
        }
    }
    colptr[M] = put;
    put = 0;
    for (size_t i = 0; i < M; i++) {
        rowptr[i] = put;
        for (int p = 0; p < rlen[i]; p++) rowidx[put++] = ridx[rbeg[i] + p];
    }
    rowptr[M] = put;
    for (size_t k = 0; k < M; k++) {
        pinv[k] = pi[k];
        qinv[k] = qi[k];
    }
    for (size_t k = 0; k < 2 * M + 2; k++) {
        col_flink[k] = cf[k];
        col_blink[k] = cb[k];
        row_flink[k] = rf[k];
        row_blink[k] = rb[k];
    }
    return BLU_OK;
}
// same layout as orc_dbg_partial_lu
extern "C" int blu_hip_dbg_partial_lu(blu_hip *h, int64_t *lptr, int64_t *lidx, double *lval,
                                      int64_t *uptr, int64_t *uidx, double *uval)
{
    if (!h) return BLU_ERROR_ARGUMENT_MISSING;
    (void)hipSetDevice(h->device);
    if (!download_scalars(h)) return BLU_ERROR_DEVICE;
    const size_t R = (size_t)h->hs.rank;
    std::vector<int> lb, ub, li, ui;
    bool ok = d2h_vec(h, lb, h->D.lbeg, R + 1) && d2h_vec(h, ub, h->D.ubeg, R + 1);
    ok = ok && d2h_vec(h, li, h->D.lidx, (size_t)h->hs.lused) && d2h_vec(h, ui, h->D.uidx, (size_t)h->hs.uused);
    if (h->hs.lused) ok = ok && hip_ok(h, hipMemcpy(lval, h->D.lval, (size_t)h->hs.lused * 8, hipMemcpyDeviceToHost), "d2h lval");
    if (h->hs.uused) ok = ok && hip_ok(h, hipMemcpy(uval, h->D.uval, (size_t)h->hs.uused * 8, hipMemcpyDeviceToHost), "d2h uval");
    if (!ok) return BLU_ERROR_DEVICE;
    for (size_t k = 0; k <= R; k++) {
        lptr[k] = lb[k];
        uptr[k] = ub[k];
    }
    for (int k = 0; k < h->hs.lused; k++) lidx[k] = li[k];
    for (int k = 0; k < h->hs.uused; k++) uidx[k] = ui[k];
    return BLU_OK;
}

// ---------------------------------------------------------------------------------------------
// synthetic LP-basis generator, host utility for benchmarks and tests.
// SplitMix64; draw order documented in DESIGN.md ("Synthetic inputs").
// ---------------------------------------------------------------------------------------------
static inline uint64_t sm64_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double sm64_u(uint64_t *s) { return (double)(sm64_next(s) >> 11) * (1.0 / 9007199254740992.0); }

extern "C" int64_t blu_hip_gen_lp_basis(int64_t m, int64_t k, int64_t bw, double tri_frac, double offscale,
                                        uint64_t seed, uint64_t *colptr, uint64_t *rowidx, double *value)
{
    uint64_t s = seed;
    std::vector<int64_t> gptr((size_t)m + 1), grow, win((size_t)(2 * bw + 2)), P((size_t)m), Q((size_t)m);
    std::vector<double> gval;
    grow.reserve((size_t)(m * std::max<int64_t>(k, 1)));
    gval.reserve((size_t)(m * std::max<int64_t>(k, 1)));
    const double tri_cut = tri_frac * (double)m;
    for (int64_t c = 0; c < m; c++) {
        gptr[(size_t)c] = (int64_t)grow.size();
        const double u1 = sm64_u(&s), u2 = sm64_u(&s);
        grow.push_back(c);
        gval.push_back((u2 < 0.5 ? -1.0 : 1.0) * (1.0 + u1));
        int64_t nw = 0;
        const int64_t lo = c - bw < 0 ? 0 : c - bw;
        if ((double)c < tri_cut) {
            for (int64_t r = lo; r <= c - 1; r++) win[(size_t)nw++] = r;
        } else {
            const int64_t hi = c + bw > m - 1 ? m - 1 : c + bw;
            for (int64_t r = lo; r <= hi; r++)
                if (r != c) win[(size_t)nw++] = r;
        }
        const int64_t n = std::min<int64_t>(k - 1, nw);
        for (int64_t t = 0; t < n; t++) {
            const int64_t r = t + (int64_t)(sm64_u(&s) * (double)(nw - t));
            std::swap(win[(size_t)t], win[(size_t)r]);
            const double uv = sm64_u(&s), us = sm64_u(&s);
            grow.push_back(win[(size_t)t]);
            gval.push_back((us < 0.5 ? -1.0 : 1.0) * (offscale * (0.1 + 0.9 * uv)));
        }
    }
    gptr[(size_t)m] = (int64_t)grow.size();
    for (int64_t i = 0; i < m; i++) P[(size_t)i] = i;
    for (int64_t t = m - 1; t >= 1; t--) std::swap(P[(size_t)t], P[(size_t)(sm64_u(&s) * (double)(t + 1))]);
    for (int64_t i = 0; i < m; i++) Q[(size_t)i] = i;
    for (int64_t t = m - 1; t >= 1; t--) std::swap(Q[(size_t)t], Q[(size_t)(sm64_u(&s) * (double)(t + 1))]);
    std::vector<int64_t> len((size_t)m, 0);
    for (int64_t c = 0; c < m; c++) len[(size_t)Q[(size_t)c]] = gptr[(size_t)c + 1] - gptr[(size_t)c];
    colptr[0] = 0;
    for (int64_t j = 0; j < m; j++) colptr[j + 1] = colptr[j] + (uint64_t)len[(size_t)j];
    for (int64_t c = 0; c < m; c++) {
        uint64_t put = colptr[Q[(size_t)c]];
        for (int64_t pos = gptr[(size_t)c]; pos < gptr[(size_t)c + 1]; pos++) {
            rowidx[put] = (uint64_t)P[(size_t)grow[(size_t)pos]];
            value[put] = gval[(size_t)pos];
            put++;
        }
    }
    return (int64_t)grow.size();
}
