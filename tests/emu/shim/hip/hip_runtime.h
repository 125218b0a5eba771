/* TEST INFRASTRUCTURE: a CPU stand-in for <hip/hip_runtime.h>, enough to compile the library's HIP translation units with g++
 * and run their kernels on the host — one workgroup at a time, every work-item a fiber (ucontext) on ONE OS thread, scheduled
 * round-robin and meeting at emulated barriers:
 *   __syncthreads()                       all fibers of the workgroup
 *   wave operations (readlane, ballot, DPP, shuffles, the wave barrier)   the 64 fibers of a wave, values exchanged through a
 *                                         per-wave slot array — which is also what makes the emulation honest about wave64
 *                                         semantics: a lane sees exactly what the other lanes published at the same operation.
 * LDS is one static array filled with a poison pattern before every workgroup (reads of uninitialised LDS show up);
 * "device" memory is host memory.  A workgroup whose fibers all wait without any barrier completing is a deadlock: the run
 * aborts with a message (divergent barriers are bugs on the GPU too).
 * Compiled with the ROCm clang++ as a plain C++ compiler (it knows ext_vector_type and the __hip_atomic builtins).
 * Only tests/ builds against this (tests/emu/Makefile → libxgm_emu.so); the product is always built by hipcc for gfx950. */
#ifndef XGM_EMU_HIP_RUNTIME_H
#define XGM_EMU_HIP_RUNTIME_H

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <ucontext.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>

#define XGM_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
/* LDS: everything runs on one OS thread, one workgroup at a time, so a thread_local variable is exactly "shared by the
 * work-items of the workgroup": `extern __shared__ unsigned char smem[];` binds to the translation unit's LDS array below,
 * and a statically sized `__shared__` array becomes a function-local (implicitly static) thread_local. */
#define __shared__ thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

constexpr unsigned kMaxThreads = 1024, kMaxWaves = kMaxThreads / 64;
constexpr size_t kStackBytes = 256 * 1024;
constexpr size_t kLdsBytes = 160 * 1024;

struct State {
    ucontext_t sched;
    ucontext_t ctx[kMaxThreads];
    bool done[kMaxThreads];
    char* stacks = nullptr;
    unsigned n = 0, cur = 0, live = 0;
    unsigned blk_arrived = 0, blk_gen = 0;
    unsigned wav_arrived[kMaxWaves] = {}, wav_gen[kMaxWaves] = {}, wav_live[kMaxWaves] = {};
    const void* want[kMaxThreads] = {};         /* per work-item: the wave operation it is heading for (diagnostics) */
    const void* wav_site[kMaxWaves] = {};       /* call site of the wave operation in progress (its return address) */
    uint64_t wav_mask[kMaxWaves] = {};          /* lanes waiting at the wave operation in progress */
    uint64_t wav_active[kMaxWaves] = {};        /* lanes that took part in the wave operation just completed */
    uint64_t xch[kMaxWaves][64];
    std::function<void()> body;
    unsigned long progress = 0;
    bool in_kernel = false;
};
inline State S;
inline std::mutex launch_mu;        /* one launch at a time: host threads that search concurrently take turns on the one emulated device */

inline void yield() { swapcontext(&S.ctx[S.cur], &S.sched); }

inline unsigned wave_of() { return threadIdx.x >> 6; }
inline unsigned lane_of() { return threadIdx.x & 63u; }

inline void complete_block_barrier() { S.blk_arrived = 0; ++S.blk_gen; ++S.progress; }
inline void complete_wave_op(unsigned w) { S.wav_active[w] = S.wav_mask[w]; S.wav_mask[w] = 0; S.wav_arrived[w] = 0; S.wav_site[w] = nullptr; ++S.wav_gen[w]; ++S.progress; }

/* work-items that have returned no longer take part in barriers (as on the hardware) */
inline void block_barrier() {
    const unsigned g = S.blk_gen;
    if (++S.blk_arrived == S.live) { complete_block_barrier(); return; }
    while (S.blk_gen == g) yield();
}
/* A wave operation is identified by its call site.  It completes when every live lane of the wave has arrived at THAT site —
 * or, when the workgroup cannot make progress otherwise, with the lanes that did (launch() below): the others are inside
 * another branch (they wait here until this operation is over, then form their own), at a workgroup barrier, or gone, i.e.
 * inactive for this instruction — which is how the hardware executes a wave operation under divergence.  What this cannot
 * model is reconvergence: a lane that skipped ahead to a LATER execution of the same site joins the earlier one.  Kernels
 * keep wave operations in wave-uniform control flow; XGM_EMU_TRACE reports every operation that ran with a subset. */
inline void wave_barrier_at(const void* site) {
    const unsigned w = wave_of();
    S.want[threadIdx.x] = site;
    while (S.wav_site[w] && S.wav_site[w] != site) yield();
    const unsigned g = S.wav_gen[w];
    S.wav_site[w] = site;
    S.wav_mask[w] |= 1ull << lane_of();
    if (++S.wav_arrived[w] == S.wav_live[w]) { complete_wave_op(w); return; }
    while (S.wav_gen[w] == g) yield();
}
inline uint64_t active_lanes() { return S.wav_active[wave_of()]; }
/* every lane publishes v, then reads what it needs: f(slots) → result.  Two rendezvous: after publishing, after reading
 * (the slots may then be overwritten by the next operation); the second is a distinct site (site + 1). */
template <class F>
inline auto exchange(const void* site, uint64_t v, F&& f) {
    const unsigned w = wave_of();
    S.xch[w][lane_of()] = v;
    wave_barrier_at(site);
    auto r = f(S.xch[w]);
    wave_barrier_at((const char*)site + 1);
    return r;
}

inline void fiber_entry() {
    S.body();
    const unsigned w = S.cur >> 6;
    S.done[S.cur] = true;
    --S.live; --S.wav_live[w];
    ++S.progress;
    if (S.live && S.blk_arrived == S.live) complete_block_barrier();
    if (S.wav_live[w] && S.wav_arrived[w] == S.wav_live[w]) complete_wave_op(w);
    swapcontext(&S.ctx[S.cur], &S.sched);
}

inline void on_fault(int sig) {
    static const char msg[] = "emu: fatal signal inside an emulated kernel; work-item / backtrace follow\n";
    if (write(2, msg, sizeof msg - 1) < 0) {}
    fprintf(stderr, "emu: signal %d, workgroup (%u,%u,%u), work-item %u\n", sig, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x);
    void* bt[48];
    backtrace_symbols_fd(bt, backtrace(bt, 48), 2);
    _exit(134);
}
inline void install_fault_handler() {
    static bool done = false;
    if (done) return;
    done = true;
    static char alt[64 * 1024];
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fault; sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t smem_bytes, unsigned char* lds, F&& body) {
    std::lock_guard<std::mutex> launch_lk(launch_mu);
    if (getenv("XGM_EMU_FAULT_TRACE")) install_fault_handler();
    if (S.in_kernel) { fprintf(stderr, "emu: nested launch\n"); abort(); }
    if (block.x * block.y * block.z > kMaxThreads || block.y != 1 || block.z != 1 || smem_bytes > kLdsBytes) { fprintf(stderr, "emu: unsupported launch shape\n"); abort(); }
    if (!S.stacks) S.stacks = (char*)malloc(kStackBytes * kMaxThreads);
    S.in_kernel = true;
    S.n = block.x;
    blockDim = block; gridDim = grid;
    S.body = body;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        memset(lds, 0xCD, kLdsBytes);
        const size_t canary_from = (smem_bytes + 15) & ~(size_t)15;          /* LDS the launch did not ask for must stay untouched */
        for (size_t i = canary_from; i < kLdsBytes; ++i) lds[i] = 0xA5;
        S.blk_arrived = 0;
        S.live = S.n;
        for (unsigned w = 0; w < kMaxWaves; ++w) {
            S.wav_arrived[w] = 0; S.wav_mask[w] = 0; S.wav_active[w] = 0;
            S.wav_live[w] = w * 64u >= S.n ? 0u : (S.n - w * 64u < 64u ? S.n - w * 64u : 64u);
        }
        for (unsigned i = 0; i < S.n; ++i) {
            S.done[i] = false;
            getcontext(&S.ctx[i]);
            S.ctx[i].uc_stack.ss_sp = S.stacks + (size_t)i * kStackBytes;
            S.ctx[i].uc_stack.ss_size = kStackBytes;
            S.ctx[i].uc_link = nullptr;
            makecontext(&S.ctx[i], (void (*)())fiber_entry, 0);
        }
        for (;;) {
            bool any = false;
            const unsigned long before = S.progress;
            for (unsigned i = 0; i < S.n; ++i) {
                if (S.done[i]) continue;
                any = true;
                S.cur = i;
                threadIdx = dim3(i, 0, 0);
                swapcontext(&S.sched, &S.ctx[i]);
            }
            if (!any) {
                for (size_t i = canary_from; i < kLdsBytes; ++i)
                    if (lds[i] != 0xA5) { fprintf(stderr, "emu: workgroup (%u,%u,%u) wrote LDS byte %zu, beyond the %zu bytes the launch asked for\n", bx, by, bz, i, smem_bytes); abort(); }
                break;
            }
            if (S.progress == before) {
                /* nobody can move: a wave operation reached by only some lanes runs with those (divergence) */
                bool released = false;
                for (unsigned w = 0; w < kMaxWaves && !released; ++w)
                    if (S.wav_arrived[w]) {
                        if (getenv("XGM_EMU_TRACE")) { Dl_info di; const bool ok = dladdr(S.wav_site[w], &di) != 0;
                          fprintf(stderr, "emu: wave %u runs the wave operation at +0x%lx (%s) with %u of %u live lanes (mask %016llx)\n", w,
                                  ok ? (unsigned long)((const char*)S.wav_site[w] - (const char*)di.dli_fbase) : 0ul, ok && di.dli_sname ? di.dli_sname : "?",
                                  S.wav_arrived[w], S.wav_live[w], (unsigned long long)S.wav_mask[w]);
                          for (unsigned l = 0; l < 64u && w * 64u + l < S.n; ++l) {
                              const unsigned t = w * 64u + l;
                              if (S.done[t] || ((S.wav_mask[w] >> l) & 1ull)) continue;
                              const bool ok2 = S.want[t] && dladdr(S.want[t], &di) != 0;
                              fprintf(stderr, "emu:    lane %u is elsewhere, last wave operation +0x%lx\n", l, ok2 ? (unsigned long)((const char*)S.want[t] - (const char*)di.dli_fbase) : 0ul);
                          } }
                        complete_wave_op(w); released = true;
                    }
                if (!released) { fprintf(stderr, "emu: deadlock in workgroup (%u,%u,%u): every work-item waits at a workgroup barrier that cannot complete\n", bx, by, bz); abort(); }
            }
        }
    }
    S.in_kernel = false;
}

}  // namespace emu

/* the dynamic LDS of this translation unit's kernels: `extern __shared__ unsigned char smem[];` inside a kernel binds to it */
namespace { alignas(64) thread_local unsigned char smem[emu::kLdsBytes]; inline unsigned char* emu_lds() { return smem; } }
#define hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, ...) emu::launch((grid), (block), (lds_bytes), emu_lds(), [&]() { kern(__VA_ARGS__); })
#define HIP_SYMBOL(x) (&(x))

/* ---- work-item functions ---- */
inline void __syncthreads() { emu::block_barrier(); }
#define EMU_SITE __builtin_return_address(0)
#define EMU_WAVE_OP __attribute__((noinline, convergent, noduplicate))
EMU_WAVE_OP inline int __builtin_amdgcn_readlane(int v, int lane) { return emu::exchange(EMU_SITE, (uint32_t)v, [&](const uint64_t* s) { return (int)(uint32_t)s[lane & 63]; }); }
EMU_WAVE_OP inline int __builtin_amdgcn_readfirstlane(int v) { return emu::exchange(EMU_SITE, (uint32_t)v, [&](const uint64_t* s) { const uint64_t act = emu::active_lanes(); return (int)(uint32_t)s[act ? __builtin_ctzll(act) : 0]; }); }
EMU_WAVE_OP inline unsigned long long __ballot(int pred) {
    return emu::exchange(EMU_SITE, pred ? 1u : 0u, [&](const uint64_t* s) { unsigned long long m = 0; const uint64_t act = emu::active_lanes(); for (unsigned i = 0; i < 64u; ++i) if (((act >> i) & 1ull) && s[i]) m |= 1ull << i; return m; });
}
template <class T> EMU_WAVE_OP inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    const unsigned lane = emu::lane_of(), base = lane & ~(unsigned)(width - 1);
    const uint64_t got = emu::exchange(EMU_SITE, raw, [&](const uint64_t* s) { return s[base + ((unsigned)src & (unsigned)(width - 1))]; });
    T r; memcpy(&r, &got, sizeof(T)); return r;
}
template <class T> EMU_WAVE_OP inline T __shfl_xor(T v, int mask, int width = 64) {
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    const unsigned lane = emu::lane_of(), base = lane & ~(unsigned)(width - 1);
    const uint64_t got = emu::exchange(EMU_SITE, raw, [&](const uint64_t* s) { const unsigned o = (lane ^ (unsigned)mask) & (unsigned)(width - 1); return s[base + o]; });
    T r; memcpy(&r, &got, sizeof(T)); return r;
}
template <class T> EMU_WAVE_OP inline T __shfl_up(T v, unsigned delta, int width = 64) {
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    const unsigned lane = emu::lane_of(), in = lane & (unsigned)(width - 1);
    const uint64_t got = emu::exchange(EMU_SITE, raw, [&](const uint64_t* s) { return in >= delta ? s[lane - delta] : s[lane]; });
    T r; memcpy(&r, &got, sizeof(T)); return r;
}
/* DPP with bound_ctrl = false: a lane without a source (or disabled by row_mask) keeps `old`.  Controls used by the library:
 * 0x111-0x11F row_shr:1-15, 0x142 row_bcast:15, 0x143 row_bcast:31 (CDNA ISA guide, DPP_CTRL) */
EMU_WAVE_OP inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)bound_ctrl;
    const unsigned lane = emu::lane_of(), row = lane >> 4, in = lane & 15u;
    return emu::exchange(EMU_SITE, (uint32_t)v, [&](const uint64_t* s) -> int {
        if (!((row_mask >> row) & 1) || !((bank_mask >> (in >> 2)) & 1)) return old;
        if (ctrl >= 0x111 && ctrl <= 0x11F) { const unsigned n = (unsigned)ctrl - 0x110u; return in >= n ? (int)(uint32_t)s[lane - n] : old; }
        if (ctrl == 0x142) return row >= 1 ? (int)(uint32_t)s[row * 16u - 1u] : old;
        if (ctrl == 0x143) return row >= 2 ? (int)(uint32_t)s[31] : old;
        fprintf(stderr, "emu: DPP control 0x%x not emulated\n", ctrl); abort();
    });
}
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t base) { const unsigned l = emu::lane_of(); const uint32_t m = l >= 32 ? mask : (mask & ((1u << l) - 1u)); return base + (uint32_t)__builtin_popcount(m); }
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t base) { const unsigned l = emu::lane_of(); const uint32_t m = l <= 32 ? 0u : (mask & ((1u << (l - 32)) - 1u)); return base + (uint32_t)__builtin_popcount(m); }
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __builtin_amdgcn_s_waitcnt(int) {}
EMU_WAVE_OP inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier_at(EMU_SITE); }
inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (shift & 31u)); }
inline uint32_t __builtin_amdgcn_sad_u8(uint32_t a, uint32_t b, uint32_t c) {
    for (int i = 0; i < 4; ++i) { const int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255; c += (uint32_t)(x > y ? x - y : y - x); }
    return c;
}
/* (__builtin_readcyclecounter is clang's own: the host's cycle counter; diagnostics only) */
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
inline unsigned long long wall_clock64() { return (unsigned long long)__builtin_readcyclecounter(); }      /* (diagnostics only) */
inline int __popc(uint32_t v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }

/* one OS thread: plain read-modify-write is atomic enough */
template <class T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, T v) { const T o = *p; *p = o & v; return o; }
template <class T> inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
/* __hip_atomic_load & co. are clang builtins on every target; their scope arguments need clang's values */
#ifndef __HIP_MEMORY_SCOPE_SINGLETHREAD
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif

/* ---- host runtime: "device" memory is host memory, everything is synchronous ---- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidDevice = 101, hipErrorNotReady = 600 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipFuncAttributes { size_t sharedSizeBytes = 0, localSizeBytes = 0; int numRegs = 0, maxThreadsPerBlock = 1024; size_t maxDynamicSharedSizeBytes = 160 * 1024; };
inline const char* hipGetErrorString(hipError_t) { return "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)64 << 30; return hipSuccess; }
/* XGM_EMU_GUARD=1: every device allocation ends 16-byte aligned right before an inaccessible page, so a kernel that reads or
 * writes past the end of a buffer faults (XGM_EMU_FAULT_TRACE=1 says where) instead of finding host memory there */
#include <sys/mman.h>
#include <map>
namespace emu {
inline std::map<void*, std::pair<void*, size_t>>& guard_map() { static std::map<void*, std::pair<void*, size_t>> m; return m; }
inline bool guard_on() { static const bool on = getenv("XGM_EMU_GUARD") != nullptr; return on; }
}
inline hipError_t hipMalloc(void** p, size_t n) {
    if (emu::guard_on()) {
        const size_t page = 4096, body = (n + 15) & ~(size_t)15, span = (body + page - 1) / page * page;
        char* base = (char*)mmap(nullptr, span + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) return 2;
        mprotect(base + span, page, PROT_NONE);
        *p = base + span - body;
        memset(*p, 0xCD, body);
        emu::guard_map()[*p] = std::make_pair((void*)base, span + page);
        return hipSuccess;
    }
    *p = aligned_alloc(256, (n + 255 + 256) & ~(size_t)255);
    if (!*p) return 2;
    memset(*p, 0xCD, n);
    return hipSuccess;
}
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    if (emu::guard_on()) {
        auto it = emu::guard_map().find(p);
        if (it != emu::guard_map().end()) { munmap(it->second.first, it->second.second); emu::guard_map().erase(it); return hipSuccess; }
    }
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
template <class T> inline hipError_t hipMemcpyFromSymbol(void* d, T* sym, size_t n) { memcpy(d, sym, n); return hipSuccess; }
template <class T> inline hipError_t hipMemcpyToSymbol(T* sym, const void* s, size_t n) { memcpy(sym, s, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = malloc(1); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(1); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(1); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = malloc(1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class F> inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, F) { *a = hipFuncAttributes(); return hipSuccess; }
template <class F> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }

#endif
