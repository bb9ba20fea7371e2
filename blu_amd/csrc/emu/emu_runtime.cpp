// emu_runtime.cpp -- fiber scheduler of the CPU emulation build (see emu/hip/hip_runtime.h).
#include "hip/hip_runtime.h"
#include <sys/mman.h>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#endif

namespace emu {

Block *g_blk = nullptr;
Fiber *g_cur = nullptr;
ucontext_t g_sched;
static const size_t kStack = 512 << 10;
static std::vector<char *> g_stacks; // reused across launches
static const std::function<void()> *g_body = nullptr;
static const void *g_sched_bottom = nullptr;
static size_t g_sched_size = 0;

static char *stack_for(size_t k)
{
    while (g_stacks.size() <= k) {
        void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            fprintf(stderr, "emu: mmap of a fiber stack failed\n");
            abort();
        }
        g_stacks.push_back((char *)p);
    }
    return g_stacks[k];
}

static void to_scheduler()
{
    Fiber *me = g_cur;
#ifdef EMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(me->state == 3 ? nullptr : &fake, g_sched_bottom, g_sched_size);
#endif
    swapcontext(&me->ctx, &g_sched);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, &g_sched_bottom, &g_sched_size);
#endif
}

void yield_to_scheduler() { to_scheduler(); }

static void fiber_main()
{
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
    (*g_body)();
    Fiber *me = g_cur;
    me->state = 3;
    Block *B = g_blk;
    Wave &W = B->waves[me->wave];
    W.nlive--;
    B->nlive--;
    // a collective (or the barrier) that was only waiting for this lane completes now
    if (W.nlive > 0 && W.arrived == W.nlive) {
        W.arrived = 0;
        W.gen++;
    }
    if (B->nlive > 0 && B->arrived == B->nlive) {
        B->arrived = 0;
        B->gen++;
    }
    to_scheduler();
    abort(); // never resumed
}

static int g_hist[1024][16];
static long g_hcount[1024];
void wave_exchange(uint64_t mine, uint64_t out[64], uint64_t *mask, const char *file, int line)
{
    Fiber *me = g_cur;
    g_hist[me->tid.x][g_hcount[me->tid.x]++ & 15] = line;
    Wave &W = g_blk->waves[me->wave];
    const int par = W.gen & 1;
    if (W.arrived == 0) W.depmask[par] = 0;
    W.dep[par][me->lane] = mine;
    W.depmask[par] |= 1ull << me->lane;
    W.site_line[me->lane] = line;
    me->site_file = file;
    me->site_line = line;
    W.arrived++;
    if (W.arrived == W.nlive) {
        // call-site check: every lane of a collective must come from the same source line
        for (int k = 0; k < 64; k++)
            if (((W.depmask[par] >> k) & 1) && W.site_line[k] != line) {
                fprintf(stderr, "emu: wave %d of block %u: lanes meet in DIFFERENT collectives: lane %d at %s:%d, lane %d at %s:%d\n",
                        me->wave, g_blk->bid.x, k, g_blk->fib[me->wave * 64 + k].site_file, W.site_line[k], me->lane, file, line);
                for (int q : {k, me->lane}) {
                    const int t = me->wave * 64 + q;
                    fprintf(stderr, "   thread %d: %ld collectives, last lines:", t, g_hcount[t]);
                    for (long u = g_hcount[t] > 16 ? g_hcount[t] - 16 : 0; u < g_hcount[t]; u++) fprintf(stderr, " %d", g_hist[t][u & 15]);
                    fprintf(stderr, "\n");
                }
                for (int q = 0; q < 64; q++)
                    if ((W.depmask[par] >> q) & 1) fprintf(stderr, "   lane %d: %s:%d\n", q, g_blk->fib[me->wave * 64 + q].site_file, W.site_line[q]);
                abort();
            }
        W.arrived = 0;
        W.gen++;
    } else {
        me->state = 1;
        me->wait_gen = W.gen;
        to_scheduler();
    }
    memcpy(out, W.dep[par], sizeof(uint64_t) * 64);
    *mask = W.depmask[par];
}

void block_barrier(const char *file, int line)
{
    Fiber *me = g_cur;
    Block *B = g_blk;
    me->site_file = file;
    me->site_line = line;
    B->arrived++;
    if (B->arrived == B->nlive) {
        B->arrived = 0;
        B->gen++;
    } else {
        me->state = 2;
        me->wait_gen = B->gen;
        to_scheduler();
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    if (g_blk) {
        fprintf(stderr, "emu: nested launch\n");
        abort();
    }
    const unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || nt > 1024 || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "emu: unsupported launch shape\n");
        abort();
    }
    Block B;
    g_body = &body;
    B.bdim = Idx{block.x, 1, 1};
    B.gdim = Idx{grid.x, 1, 1};
    for (unsigned b = 0; b < grid.x; b++) {
        B.bid = Idx{b, 0, 0};
        B.fib.assign(nt, Fiber());
        B.waves.assign((nt + 63) / 64, Wave());
        B.nlive = (int)nt;
        B.arrived = 0;
        B.gen = 0;
        for (unsigned t = 0; t < nt; t++) {
            Fiber &F = B.fib[t];
            F.tid = Idx{t, 0, 0};
            F.lane = (int)(t & 63);
            F.wave = (int)(t >> 6);
            F.state = 0;
            F.stack = stack_for(t);
            B.waves[F.wave].nlive++;
            getcontext(&F.ctx);
            F.ctx.uc_stack.ss_sp = F.stack;
            F.ctx.uc_stack.ss_size = kStack;
            F.ctx.uc_link = nullptr;
            makecontext(&F.ctx, (void (*)())fiber_main, 0);
        }
        g_blk = &B;
        memset(g_hcount, 0, sizeof g_hcount);
        int done = 0;
        while (done < (int)nt) {
            bool progress = false;
            for (unsigned t = 0; t < nt; t++) {
                Fiber &F = B.fib[t];
                if (F.state == 3) continue;
                if (F.state == 1 && B.waves[F.wave].gen == F.wait_gen) continue;
                if (F.state == 2 && B.gen == F.wait_gen) continue;
                F.state = 0;
                g_cur = &F;
#ifdef EMU_ASAN
                void *fake = nullptr;
                __sanitizer_start_switch_fiber(&fake, F.stack, kStack);
#endif
                swapcontext(&g_sched, &F.ctx);
#ifdef EMU_ASAN
                __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
                g_cur = nullptr;
                progress = true;
                if (F.state == 3) done++;
            }
            if (!progress) {
                fprintf(stderr, "emu: DEADLOCK in block %u: no fiber can run.  Waiting fibers:\n", b);
                int shown = 0;
                for (unsigned t = 0; t < nt && shown < 24; t++) {
                    Fiber &F = B.fib[t];
                    if (F.state == 1 || F.state == 2) {
                        fprintf(stderr, "  thread %u (wave %d lane %d): %s at %s:%d\n", t, F.wave, F.lane,
                                F.state == 1 ? "wave collective" : "__syncthreads", F.site_file, F.site_line);
                        shown++;
                    }
                }
                abort();
            }
        }
        g_blk = nullptr;
    }
    g_body = nullptr;
}

} // namespace emu
