// TEST INFRASTRUCTURE (not product code): a host stand-in for <hip/hip_runtime.h> that lets g++ compile one of OUR OWN kernel files
// (contrastboundary_amd/csrc/pt_layer.hip) for the CPU and run it with wave semantics: every thread of a workgroup is a ucontext fibre,
// cross-lane operations (MFMA, DPP, bpermute) and barriers are rendezvous points at which the fibres of a wave / workgroup meet.
// Only what that file and cbl_common.h use is provided.  Used by tests/test_pt_layer_host.py on small shapes.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define CBL_HOST_WAVE_EMULATION 1      // sections of a kernel file that only make sense on the device (dynamic LDS, runtime API calls, entry points of other translation units) are compiled out

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void* hipStream_t;
static inline hipError_t hipGetLastError() { return hipSuccess; }
// what cbl_common.h's launch helpers ask the runtime (one "device" with 256 compute units, four workgroups of any kernel resident per unit)
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = 256; return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* blocks, const void*, int, size_t) { *blocks = 4; return hipSuccess; }

using std::max;
using std::min;
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }

// Switching fibres: glibc's swapcontext saves and restores the signal mask — one system call per switch, a third of an emulated kernel's run time.  On x86-64
// (without a sanitizer, which has to see the stack change through its swapcontext interceptor) the switch is the callee-saved registers and the stack pointer.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(CBL_EMUL_UCONTEXT)
#define CBL_EMUL_FAST_SWITCH 1
#endif

namespace emul {

#ifdef CBL_EMUL_FAST_SWITCH
// saves the caller's callee-saved state on its stack, leaves that stack pointer in *from, continues on the stack `to` (prepared by a previous switch or by fresh_stack)
__attribute__((naked, noinline, unused)) static void switch_stack(void** /*from: rdi*/, void* /*to: rsi*/)
{
    __asm__ volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "subq $8, %rsp\n\tstmxcsr (%rsp)\n\tfnstcw 4(%rsp)\n\t"
        "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
        "ldmxcsr (%rsp)\n\tfldcw 4(%rsp)\n\taddq $8, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
// a stack on which switch_stack "returns" into entry() (which never returns), with the launching thread's floating-point control words
inline void* fresh_stack(char* stack, size_t size, void (*entry)())
{
    unsigned long long* sp = (unsigned long long*)(((unsigned long long)(stack + size)) & ~15ull);
    unsigned int csr; unsigned short cw;
    __asm__ volatile("stmxcsr %0" : "=m"(csr)); __asm__ volatile("fnstcw %0" : "=m"(cw));
    *--sp = 0;                                   // where entry's return address would be: entry starts, as after a call, 8 below a 16-byte boundary
    *--sp = (unsigned long long)entry;
    for (int i = 0; i < 6; i++) *--sp = 0;
    *--sp = (unsigned long long)csr | ((unsigned long long)cw << 32);
    return sp;
}
#endif

constexpr int WAVE = 64;
struct Wave;
struct Thunk { void (*call)(void*, Wave&) = nullptr; void* ctx = nullptr; };
template <class F> inline void invoke_thunk(void* c, Wave& w) { (*static_cast<F*>(c))(w); }
struct Wave {
    int alive = 0, arrived = 0, at_barrier = 0;                     // at_barrier: lanes of this wave parked in __syncthreads (they take part in no wave collective)
    float in[WAVE][8]; float out[WAVE][4]; bool present[WAVE];      // present: takes part in the collective being computed
    float pay[WAVE][8];
    // per waiting lane: which collective it waits in (tag), whether that one spans the wave (scope 1) or a lane group (scope 0), and how to compute it
    bool waiting[WAVE]; unsigned tag[WAVE]; int scope[WAVE]; Thunk thunk[WAVE];
};
struct Fiber {
    ucontext_t ctx; void* sp = nullptr; dim3 tid; int lin = 0, wave = 0, lane = 0; bool done = false; char* stack = nullptr; size_t stack_size = 0;
};
struct State {
    ucontext_t main_ctx; void* main_sp = nullptr;
    std::vector<Fiber> fibers; std::vector<Wave> waves;
    Fiber* cur = nullptr;
    dim3 block_idx, block_dim, grid_dim;
    int block_alive = 0, block_arrived = 0; unsigned block_gen = 0;
    const std::function<void()>* body = nullptr;
    size_t stack_bytes = 96 * 1024;
    State() { if (const char* e = std::getenv("CBL_EMUL_STACK_KB")) { const long kb = std::atol(e); if (kb >= 64 && kb <= 8192) stack_bytes = (size_t)kb * 1024; } }
};
inline State& S() { static State s; return s; }

#ifdef CBL_EMUL_FAST_SWITCH
inline void to_main(Fiber* f) { switch_stack(&f->sp, S().main_sp); }
inline void to_fiber(Fiber* f) { switch_stack(&S().main_sp, f->sp); }
#else
inline void to_main(Fiber* f) { swapcontext(&f->ctx, &S().main_ctx); }
inline void to_fiber(Fiber* f) { swapcontext(&S().main_ctx, &f->ctx); }
#endif
inline void yield() { to_main(S().cur); }

inline void trampoline()
{
    State& s = S();
    (*s.body)();
    Fiber* f = s.cur;
    f->done = true;
    s.waves[f->wave].alive--;
    s.block_alive--;
    to_main(f);
    std::abort();                                 // a finished fibre is never resumed
}

// Every alive lane of the calling wave deposits `np` floats and waits; when all alive lanes wait, the collective is computed (`compute(wave)` fills wave.out)
// and each lane takes its `nr` results.
// Divergence: a wave executes the two sides of a branch one after the other and meets again behind it, so lanes may wait in DIFFERENT collectives — the idiom in
// our kernels is a lane GROUP (16 / 32 lanes that share a point) skipping a group reduction the other groups of the wave take (cbl.hip contrast_row: a point
// without both kinds of neighbours returns before the DPP sums), all of them meeting at the next wave-wide operation.  When every alive lane waits and their
// (scope, tag) differ, the lanes of ONE collective run it among themselves — lanes outside count as inactive, as the exec mask makes them on the device — and go
// on; group-scoped collectives (DPP, shuffles narrower than the wave) go before wave-wide ones, which are the meeting points.  The others keep waiting.
inline void resolve(Wave& w)
{
    int pick = -1;
    for (int l = 0; l < WAVE; l++) if (w.waiting[l] && (pick < 0 || w.scope[l] < w.scope[pick])) pick = l;
    if (pick < 0) return;
    for (int l = 0; l < WAVE; l++) {
        w.present[l] = w.waiting[l] && w.scope[l] == w.scope[pick] && w.tag[l] == w.tag[pick];
        if (!w.present[l]) for (int i = 0; i < 8; i++) w.pay[l][i] = 0.f; else std::memcpy(w.pay[l], w.in[l], sizeof(w.pay[l]));
    }
    std::swap(w.in, w.pay);                                         // compute reads `in`: the participants' payloads, zeros elsewhere (waiting lanes keep theirs in `pay`)
    w.thunk[pick].call(w.thunk[pick].ctx, w);
    std::swap(w.in, w.pay);
    for (int l = 0; l < WAVE; l++) if (w.present[l]) { w.waiting[l] = false; w.present[l] = false; w.arrived--; }
}

template <class F>
inline void wave_collective(const float* payload, int np, float* result, int nr, F compute, unsigned tag = 0, int scope = 1)
{
    State& s = S(); Fiber* f = s.cur; Wave& w = s.waves[f->wave];
    const int me = f->lane;
    for (int i = 0; i < np; i++) w.in[me][i] = payload[i];
    w.waiting[me] = true; w.tag[me] = tag; w.scope[me] = scope;
    w.thunk[me].call = &invoke_thunk<F>; w.thunk[me].ctx = (void*)&compute;
    w.arrived++;
    while (w.waiting[me]) {
        if (w.arrived == w.alive - w.at_barrier) resolve(w);       // lanes at the workgroup barrier are behind the meeting point: the others go on among themselves
        // (also the lane that completed a rendezvous steps aside once: the wave's lanes then run the code up to the next rendezvous in ASCENDING lane order.
        //  A wave executes in lockstep, so "lane 0 stores to LDS, every lane loads it" needs no barrier on the device; here it needs the storing lane
        //  to run first, which this order gives for the idiom's usual writer — the first lane of a wave or of a lane group.)
        yield();
    }
    for (int i = 0; i < nr; i++) result[i] = w.out[me][i];
}

inline void block_barrier()
{
    State& s = S();
    Wave& w = s.waves[s.cur->wave];
    const unsigned g = s.block_gen;
    s.block_arrived++; w.at_barrier++;
    while (s.block_gen == g) {
        if (s.block_arrived == s.block_alive) {                      // released: every lane of the workgroup is past the barrier from here on, whenever it runs next
            s.block_arrived = 0; s.block_gen++;
            for (auto& x : s.waves) x.at_barrier = 0;
            break;
        }
        yield();
    }
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    State& s = S();
    const int nt = (int)(block.x * block.y * block.z);
    const int nw = (nt + WAVE - 1) / WAVE;
    // (the state is ONE per process even when several host builds are loaded — inline statics are unique symbols — so a fibre made by an earlier launch
    //  may carry a smaller stack than this launch asks for: every fibre remembers its own size)
    if ((int)s.fibers.size() < nt) s.fibers.resize(nt);
    for (int i = 0; i < nt; i++) {
        Fiber& f = s.fibers[i];
        if (!f.stack || f.stack_size < s.stack_bytes) { std::free(f.stack); f.stack = (char*)std::malloc(s.stack_bytes); f.stack_size = s.stack_bytes; }
    }
    s.body = &body; s.block_dim = block; s.grid_dim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        s.block_idx = dim3(bx, by, bz);
        s.waves.assign(nw, Wave());
        for (auto& w : s.waves) for (int l = 0; l < WAVE; l++) w.present[l] = w.waiting[l] = false;
        s.block_alive = nt; s.block_arrived = 0; s.block_gen = 0;
        for (int t = 0; t < nt; t++) {
            Fiber& f = s.fibers[t];
            f.lin = t; f.wave = t / WAVE; f.lane = t % WAVE; f.done = false;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            s.waves[f.wave].alive++;
#ifdef CBL_EMUL_FAST_SWITCH
            f.sp = fresh_stack(f.stack, f.stack_size, trampoline);
#else
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = f.stack_size; f.ctx.uc_link = &s.main_ctx;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
#endif
        }
        int remaining = nt; long rounds = 0;
        while (remaining > 0) {
            remaining = 0;
            for (int t = 0; t < nt; t++) {
                Fiber& f = s.fibers[t];
                if (f.done) continue;
                s.cur = &f;
                to_fiber(&f);
                if (!f.done) remaining++;
            }
            if (++rounds > 50000000) { std::fprintf(stderr, "emul: no progress (divergent rendezvous?)\n"); std::abort(); }
        }
    }
    s.cur = nullptr;
}

}  // namespace emul

#define threadIdx (emul::S().cur->tid)
#define blockIdx (emul::S().block_idx)
#define blockDim (emul::S().block_dim)
#define gridDim (emul::S().grid_dim)
static inline void __syncthreads() { emul::block_barrier(); }
// CBL_EMUL_TRACE=1 names every kernel as it is launched (which one a "no progress" abort belongs to)
namespace emul { inline void trace(const char* name) { static const bool on = std::getenv("CBL_EMUL_TRACE") != nullptr; if (on) std::fprintf(stderr, "emul: launch %s\n", name); } }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) (emul::trace(#kern), emul::launch((grid), (block), [&]() { kern(__VA_ARGS__); }))
