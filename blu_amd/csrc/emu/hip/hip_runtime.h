// emu/hip/hip_runtime.h -- CPU stand-in for <hip/hip_runtime.h>: a DIAGNOSTIC build of the library
// (`make emu` -> libblu_emu.so, never loaded by the product path) that runs the HIP kernels of this
// directory on the host, one fiber per GPU thread, so that kernel logic can be stepped against the CPU
// oracle, run under AddressSanitizer, and debugged with ordinary tools before it goes to a GPU box.
//
// Model: a launch runs its workgroups one after the other; the threads of a workgroup are ucontext
// fibers scheduled round-robin by ONE host thread.  A fiber runs until it reaches a collective
// (__ballot, __shfl, readlane, a wave reduction, wave_mem_sync, __syncthreads, ...): every collective
// is a barrier of its wave (or workgroup), so between two collectives the lanes of a wave run one after
// the other instead of in lockstep.  Code that relies on lockstep across lanes WITHOUT a collective in
// between (lane A loads X, lane B then stores X) marks the spot with WAVE_LOCKSTEP() (blu_dev.h), a
// wave barrier here and nothing on the GPU.  Collectives must be reached by all live lanes of a wave
// (wave-uniform control flow); the scheduler reports lanes that wait at different call sites, and
// deadlocks, with the source lines.
// Atomics are plain read-modify-writes (one host thread).  "Device memory" is the host heap, filled
// with a pattern at allocation so that reads of uninitialised memory show.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <time.h>
#include <ucontext.h>
#include <vector>

#define BLU_EMU_BUILD 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
// dynamic LDS (`extern __shared__ T x[];`): a fixed static array in this build
#define BLU_DYN_SHARED(T, name, bytes) static __attribute__((aligned(16))) T name[bytes]

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct int2 { int x, y; };
inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
struct double2 { double x, y; };
inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
struct alignas(16) int4 { int x, y, z, w; };

namespace emu {

struct Idx { unsigned x, y, z; };
struct Wave;
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    Idx tid{0, 0, 0};
    int lane = 0, wave = 0;
    int state = 0;   // 0 runnable, 1 waiting for a wave collective, 2 waiting for the workgroup barrier, 3 done
    int wait_gen = 0;
    const char *site_file = "";
    int site_line = 0;
};
struct Wave {
    int nlive = 0, arrived = 0, gen = 0;
    uint64_t dep[2][64];
    uint64_t depmask[2] = {0, 0};
    int site_line[64];
};
struct Block {
    std::vector<Fiber> fib;
    std::vector<Wave> waves;
    int nlive = 0, arrived = 0, gen = 0;
    Idx bid{0, 0, 0};
    Idx bdim{1, 1, 1}, gdim{1, 1, 1};
};
extern Block *g_blk;
extern Fiber *g_cur;
extern ucontext_t g_sched;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void yield_to_scheduler();
// every live lane of the calling fiber's wave deposits `mine`; returns once all have, with all 64 deposits
// in out[] and the mask of lanes that took part
void wave_exchange(uint64_t mine, uint64_t out[64], uint64_t *mask, const char *file, int line);
void block_barrier(const char *file, int line);

inline int lane() { return g_cur->lane; }

template <class T> inline uint64_t to_bits(T v)
{
    static_assert(sizeof(T) <= 8, "emu: value wider than 64 bits in a collective");
    uint64_t b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T from_bits(uint64_t b)
{
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}
} // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

// ---- collectives ---------------------------------------------------------------------------------
inline void __syncthreads(const char *f = __builtin_FILE(), int l = __builtin_LINE()) { emu::block_barrier(f, l); }
inline unsigned long long __ballot(int pred, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(pred ? 1 : 0, out, &mask, f, l);
    unsigned long long b = 0;
    for (int k = 0; k < 64; k++)
        if (((mask >> k) & 1) && out[k]) b |= 1ull << k;
    return b;
}
template <class T> inline T __shfl(T v, int src, int width = 64, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    return emu::from_bits<T>(out[src & 63]);
}
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    const int s = emu::lane() - (int)delta;
    return s >= 0 ? emu::from_bits<T>(out[s]) : v;
}
template <class T> inline T __shfl_xor(T v, int lm, int width = 64, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    return emu::from_bits<T>(out[(emu::lane() ^ lm) & 63]);
}
inline int emu_readlane(int v, int src, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    return emu::from_bits<int>(out[src & 63]);
}
inline int emu_readfirstlane(int v, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    return emu::from_bits<int>(out[__builtin_ctzll(mask)]);
}
#define __builtin_amdgcn_readlane(v, s) emu_readlane((int)(v), (int)(s))
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane((int)(v))
// a fence at wavefront scope orders one lane's stores before another lane's loads: here, a wave barrier
inline void emu_fence(const char *scope, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    if (scope[0] == 'w') { // "wavefront" / "workgroup": both are used to order the lanes of one wave
        uint64_t out[64], mask;
        emu::wave_exchange(0, out, &mask, f, l);
    }
}
#define __builtin_amdgcn_fence(order, scope) emu_fence(scope)
inline void emu_wave_lockstep(const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(0, out, &mask, f, l);
}
// wave all-reduce used by blu_dev.h in this build
template <class T, class Op> inline T emu_wave_allreduce(T v, Op op, const char *f = __builtin_FILE(), int l = __builtin_LINE())
{
    uint64_t out[64], mask;
    emu::wave_exchange(emu::to_bits(v), out, &mask, f, l);
    bool first = true;
    T acc = v;
    for (int k = 0; k < 64; k++)
        if ((mask >> k) & 1) {
            const T x = emu::from_bits<T>(out[k]);
            acc = first ? x : op(acc, x);
            first = false;
        }
    return acc;
}

// ---- per-lane builtins ---------------------------------------------------------------------------
inline unsigned emu_mbcnt_lo(unsigned m, unsigned base)
{
    const int l = emu::lane();
    const unsigned below = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
    return base + (unsigned)__builtin_popcount(m & below);
}
inline unsigned emu_mbcnt_hi(unsigned m, unsigned base)
{
    const int l = emu::lane();
    const unsigned below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return base + (unsigned)__builtin_popcount(m & below);
}
#define __builtin_amdgcn_mbcnt_lo(m, b) emu_mbcnt_lo((unsigned)(m), (unsigned)(b))
#define __builtin_amdgcn_mbcnt_hi(m, b) emu_mbcnt_hi((unsigned)(m), (unsigned)(b))
inline unsigned long long emu_memtime()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec;
}
#define __builtin_amdgcn_s_memtime() emu_memtime()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu::yield_to_scheduler()
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __ddiv_rn(double a, double b) { return a / b; }
inline long long __double_as_longlong(double x) { return emu::from_bits<long long>(emu::to_bits(x)); }
inline double __longlong_as_double(long long x) { return emu::from_bits<double>((uint64_t)x); }
inline int __double2loint(double x) { return (int)(uint32_t)(emu::to_bits(x) & 0xffffffffull); }
inline int __double2hiint(double x) { return (int)(uint32_t)(emu::to_bits(x) >> 32); }
inline double __hiloint2double(int hi, int lo) { return emu::from_bits<double>(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline long long min(long long a, int b) { return a < b ? a : (long long)b; }
inline long long min(int a, long long b) { return a < b ? (long long)a : b; }
inline long long max(long long a, int b) { return a > b ? a : (long long)b; }
inline long long max(int a, long long b) { return a > b ? (long long)a : b; }

// ---- atomics (one host thread: plain read-modify-write) -------------------------------------------
template <class T, class V> inline T atomicAdd(T *p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> inline T atomicMax(T *p, V v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> inline T atomicMin(T *p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class V> inline T atomicOr(T *p, V v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class V> inline T atomicAnd(T *p, V v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class V> inline T atomicExch(T *p, V v) { T o = *p; *p = (T)v; return o; }
template <class T, class C, class V> inline T atomicCAS(T *p, C c, V v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }
template <class P, class V> inline auto emu_afetch_add(P p, V v) { auto o = *p; *p = o + v; return o; }
template <class P, class V> inline auto emu_afetch_min(P p, V v) { auto o = *p; if (v < o) *p = v; return o; }
template <class P, class V> inline auto emu_afetch_max(P p, V v) { auto o = *p; if (v > o) *p = v; return o; }
template <class P, class V> inline auto emu_afetch_or(P p, V v) { auto o = *p; *p = o | v; return o; }
template <class P> inline auto emu_aload(P p) { return *p; }
template <class P, class V> inline void emu_astore(P p, V v) { *p = v; }
#define __hip_atomic_fetch_add(p, v, o, s) emu_afetch_add((p), (v))
#define __hip_atomic_fetch_min(p, v, o, s) emu_afetch_min((p), (v))
#define __hip_atomic_fetch_max(p, v, o, s) emu_afetch_max((p), (v))
#define __hip_atomic_fetch_or(p, v, o, s) emu_afetch_or((p), (v))
#define __hip_atomic_load(p, o, s) emu_aload((p))
#define __hip_atomic_store(p, v, o, s) emu_astore((p), (v))

// ---- runtime API ---------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct emu_stream *hipStream_t;
struct emu_event { double t; };
typedef emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToHost = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    int multiProcessorCount;
    int cooperativeLaunch;
    int maxThreadsPerBlock;
    int warpSize;
    int clockRate;
};
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu: error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "CPU emulation (fibers)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:emu");
    p->totalGlobalMem = (size_t)8 << 30;
    p->sharedMemPerBlock = 160 << 10;
    p->multiProcessorCount = 1;
    p->cooperativeLaunch = 0;
    p->maxThreadsPerBlock = 1024;
    p->warpSize = 64;
    p->clockRate = 1000000;
    return hipSuccess;
}
template <class T> inline hipError_t hipMalloc(T **p, size_t bytes)
{
    void *q = malloc(bytes ? bytes : 1);
    if (!q) return hipErrorOutOfMemory;
    memset(q, 0xA5, bytes);
    *p = (T *)q;
    return hipSuccess;
}
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)1 << 34; *t = (size_t)1 << 35; return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = 1e-6 * (double)emu_memtime(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int, size_t) { *n = 1; return hipSuccess; }
inline hipError_t hipLaunchCooperativeKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipErrorNotSupported; }
#define hipLaunchKernelGGL(K, G, B, SH, ST, ...) emu::launch((G), (B), [=]() { K(__VA_ARGS__); })
