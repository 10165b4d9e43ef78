// xc_device.h (EMULATOR build) -- TEST INFRASTRUCTURE ONLY.
//
// Same vocabulary as x_clip_amd/csrc/hw/xc_device.h, implemented on a functional wave64 emulator so
// that the CPU test-suite (no GPU in the build container) executes the *same kernel sources* the
// product compiles for gfx950.  One OS thread; every GPU thread of a workgroup is a fibre (own stack, a six-register switch);
// workgroups run one after another.  Fibres switch only at __syncthreads() and at wave collectives
// (shuffles, MFMA), waves of a workgroup are scheduled in a seeded random order and the dynamic LDS is
// poisoned per workgroup, so a missing barrier or a read of unwritten LDS shows up as a wrong result.
// MFMA follows the operand / accumulator lane maps documented in cdna_hip_programming.md section 3.
//
// Built by tests/emu/build_emu.py into tests/emu/libxclip_emu.so with the host clang++; it is never
// loaded by the x_clip_amd package itself.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define XC_DEV inline
#define XC_FOUR_WAVES_PER_SIMD
#define XC_HOST_DEV inline
#define XC_LDS_DYNAMIC(name) unsigned char* name = xcemu::dyn_lds()
#define XC_ALLOW_LDS(kernel, bytes) ((void)0)

namespace xcemu {

struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

constexpr int kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;
constexpr int kExBytes = 64;          // bytes one lane may deposit in a wave collective

// Fibre switch.  glibc's swapcontext saves and restores the signal mask with a system call on every switch, and a wave collective
// costs 128 switches: on x86-64 a switch here is the six callee-saved registers and the stack pointer (nothing in the kernels
// touches the signal mask, MXCSR or the x87 control word).  Other hosts keep ucontext.
#if defined(__x86_64__)
#define XCEMU_FAST_SWITCH 1
struct Context { void* sp = nullptr; };
__attribute__((naked, noinline)) static void ctx_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t pushq %rbx\n\t pushq %r12\n\t pushq %r13\n\t pushq %r14\n\t pushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t movq %rsi, %rsp\n\t"
        "popq %r15\n\t popq %r14\n\t popq %r13\n\t popq %r12\n\t popq %rbx\n\t popq %rbp\n\t ret\n\t");
}
inline void ctx_swap(Context& from, Context& to) { ctx_switch(&from.sp, to.sp); }
// a fresh context that enters `entry` (which never returns) on the given stack: six zeroed registers, the entry address as the
// `ret` target, and a null return address above it so that the entry sees the stack alignment of a called function
inline void ctx_make(Context& c, char* stack, size_t bytes, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;
    *--sp = (void*)entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    c.sp = sp;
}
#else
struct Context { ucontext_t uc; };
inline void ctx_swap(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
inline void ctx_make(Context& c, char* stack, size_t bytes, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = bytes;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
    Context ctx;
    bool done = true;
    int tid = 0;
    unsigned coll = 0;                // number of wave collectives this lane has completed
};
struct WaveState {
    alignas(16) unsigned char buf[2][64][kExBytes];
    unsigned slot_gen[2];
    int arrived[2];
    int live;
};
struct State {
    Dim3 threadIdx, blockIdx, blockDim, gridDim;
    Fiber fibers[kMaxThreads];
    WaveState waves[kMaxThreads / 64];
    Context sched;
    char* stacks = nullptr;
    unsigned char* lds = nullptr;
    int cur = -1, nthreads = 0, live = 0;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    bool progressed = false;
    const std::function<void()>* body = nullptr;
    unsigned rng = 12345u;
};
inline State& S() {
    static State* s = nullptr;
    if (!s) {
        s = new State();
        s->stacks = (char*)aligned_alloc(4096, (size_t)kStack * kMaxThreads);
        s->lds = (unsigned char*)aligned_alloc(4096, 160 * 1024);
        const char* e = getenv("XCEMU_SEED");
        if (e) s->rng = (unsigned)atoi(e) * 2654435761u + 1u;
    }
    return *s;
}
inline unsigned char* dyn_lds() { return S().lds; }
inline unsigned next_rand() {
    State& s = S();
    s.rng = s.rng * 1664525u + 1013904223u;
    return s.rng >> 8;
}

inline void yield() {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    ctx_swap(f.ctx, s.sched);
}

inline void fiber_main() {
    State& s = S();
    (*s.body)();
    Fiber& f = s.fibers[s.cur];
    f.done = true;
    s.live--;
    s.waves[f.tid >> 6].live--;
    s.progressed = true;
    // a thread that has exited no longer takes part in barriers (as on the hardware)
    if (s.live > 0 && s.bar_arrived == s.live) {
        s.bar_arrived = 0;
        s.bar_gen++;
    }
    ctx_swap(f.ctx, s.sched);
    abort();                                              // (a finished fibre is never resumed)
}

inline void block_barrier() {
    State& s = S();
    unsigned gen = s.bar_gen;
    s.bar_arrived++;
    s.progressed = true;
    if (s.bar_arrived == s.live) {
        s.bar_arrived = 0;
        s.bar_gen++;
        return;
    }
    while (s.bar_gen == gen) yield();
}

// Deposit `bytes` from this lane, wait until every live lane of the wave has deposited, return the table
// (indexed by lane).  Double buffered: a lane can be at most one collective ahead of its wave mates.
inline const unsigned char (*wave_exchange(const void* mine, int bytes))[kExBytes] {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    WaveState& w = s.waves[f.tid >> 6];
    const int slot = f.coll & 1;
    if (w.slot_gen[slot] != f.coll) {
        w.slot_gen[slot] = f.coll;
        w.arrived[slot] = 0;
    }
    memcpy(w.buf[slot][f.tid & 63], mine, bytes);
    w.arrived[slot]++;
    s.progressed = true;
    while (w.arrived[slot] < w.live) yield();
    f.coll++;
    return w.buf[slot];
}

inline void run_block(const std::function<void()>& body, Dim3 grid, Dim3 block, Dim3 bidx) {
    State& s = S();
    const int n = block.x * block.y * block.z;
    if (n > kMaxThreads) { fprintf(stderr, "xcemu: block too large\n"); abort(); }
    s.body = &body;
    s.blockIdx = bidx; s.blockDim = block; s.gridDim = grid;
    s.nthreads = n; s.live = n; s.bar_arrived = 0; s.bar_gen = 0;
    memset(s.lds, 0xFF, 160 * 1024);                      // poison: bf16/f32 NaN patterns
    const int nw = (n + 63) / 64;
    for (int w = 0; w < nw; ++w) {
        WaveState& ws = s.waves[w];
        ws.slot_gen[0] = ws.slot_gen[1] = 0xFFFFFFFFu;
        ws.arrived[0] = ws.arrived[1] = 0;
        ws.live = (w == nw - 1) ? n - 64 * w : 64;
    }
    for (int t = 0; t < n; ++t) {
        Fiber& f = s.fibers[t];
        f.done = false; f.tid = t; f.coll = 0;
        ctx_make(f.ctx, s.stacks + (size_t)t * kStack, kStack, fiber_main);
    }
    std::vector<int> order(nw);
    for (int w = 0; w < nw; ++w) order[w] = w;
    while (s.live > 0) {
        s.progressed = false;
        for (int i = nw - 1; i > 0; --i) {                // fresh random wave order every sweep
            int j = next_rand() % (i + 1);
            int tmp = order[i]; order[i] = order[j]; order[j] = tmp;
        }
        for (int oi = 0; oi < nw; ++oi) {
            const int w = order[oi];
            const int lanes = (w == nw - 1) ? n - 64 * w : 64;
            const int rot = next_rand() & 63;
            for (int li = 0; li < lanes; ++li) {
                const int t = w * 64 + (li + rot) % lanes;
                Fiber& f = s.fibers[t];
                if (f.done) continue;
                s.cur = t;
                s.threadIdx = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                ctx_swap(s.sched, f.ctx);
            }
        }
        if (!s.progressed) {
            fprintf(stderr, "xcemu: deadlock in block (%u,%u,%u): a barrier or wave collective is not reached by all "
                            "threads (divergent __syncthreads / shuffle / MFMA?)\n", bidx.x, bidx.y, bidx.z);
            abort();
        }
    }
}

template <class F>
inline void launch(F&& f, Dim3 grid, Dim3 block) {
    std::function<void()> body(f);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) run_block(body, grid, block, Dim3(x, y, z));
}

}  // namespace xcemu

// ---- HIP surface used by kernels and by the C-ABI launch code ----------------------------------------
typedef xcemu::Dim3 dim3;
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulator"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipFuncSetAttributeMaxDynLds(const void*, int) { return 0; }
#define threadIdx (xcemu::S().threadIdx)
#define blockIdx (xcemu::S().blockIdx)
#define blockDim (xcemu::S().blockDim)
#define gridDim (xcemu::S().gridDim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    xcemu::launch([=]() { kernel(__VA_ARGS__); }, grid, block)
inline void __syncthreads() { xcemu::block_barrier(); }

// small on purpose: persistent kernels wrap around in the tests.  XCLIP_EMU_CUS=256 (diagnostics) makes the host-side grid / split
// heuristics take the decisions they take on an MI355X.
inline int xc_num_cus() {
    static const int v = [] { const char* e = getenv("XCLIP_EMU_CUS"); return e != nullptr && atoi(e) > 0 ? atoi(e) : 3; }();
    return v;
}

// split-K policies plan for an MI355X whatever the emulated grid (hw/xc_device.h: the device's own count)
inline int xc_policy_cus() {
    static const int v = [] { const char* e = getenv("XCLIP_EMU_CUS"); return e != nullptr && atoi(e) > 0 ? atoi(e) : 256; }();
    return v;
}

namespace xc {

typedef uint16_t bf16_t;
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

inline float bf2f(bf16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline bf16_t f2bf(float f) {           // round to nearest even, NaN preserved
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

inline uint32_t f2bf_pk(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

inline int lane_id() { return threadIdx.x & 63; }
inline int wave_id() { return threadIdx.x >> 6; }
inline void sync() { xcemu::block_barrier(); }

template <class V>
inline V shfl_generic(V v, int src_lane) {
    auto tab = xcemu::wave_exchange(&v, sizeof(V));
    V out;
    memcpy(&out, tab[src_lane & 63], sizeof(V));
    return out;
}
inline float shfl_xor(float v, int mask) { return shfl_generic(v, lane_id() ^ mask); }
inline int shfl_xor(int v, int mask) { return shfl_generic(v, lane_id() ^ mask); }
inline float shfl(float v, int src) { return shfl_generic(v, src); }
inline int shfl(int v, int src) { return shfl_generic(v, src); }

inline bool wave_all(bool pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= shfl_xor(v, m);
    return v != 0;
}
inline uint32_t wave_ballot32(bool pred) {                 // bit l = the predicate of lane l, lanes 0..31
    int v = (pred && lane_id() < 32) ? (int)(1u << lane_id()) : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= shfl_xor(v, m);
    return (uint32_t)v;
}
inline uint64_t wave_ballot64(bool pred) {
    int lo = (pred && lane_id() < 32) ? (int)(1u << lane_id()) : 0, hi = (pred && lane_id() >= 32) ? (int)(1u << (lane_id() - 32)) : 0;
    for (int m = 32; m >= 1; m >>= 1) { lo |= shfl_xor(lo, m); hi |= shfl_xor(hi, m); }
    return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
}
inline int popc64(uint64_t v) { return __builtin_popcountll(v); }
inline bool wave_any(bool pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= shfl_xor(v, m);
    return v != 0;
}

inline float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
inline float wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}
inline void wave_max_min(float& hi, float& lo) {
    for (int m = 32; m >= 1; m >>= 1) {
        const float h2 = shfl_xor(hi, m), l2 = shfl_xor(lo, m);
        hi = fmaxf(hi, h2);
        lo = fminf(lo, l2);
    }
}

inline float dot2_bf16(uint32_t a, uint32_t b, float c) {
    auto lo = [](uint32_t v) { uint32_t u = v << 16; float f; memcpy(&f, &u, 4); return f; };
    auto hi = [](uint32_t v) { uint32_t u = v & 0xffff0000u; float f; memcpy(&f, &u, 4); return f; };
    return c + lo(a) * lo(b) + hi(a) * hi(b);
}

inline f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    struct Dep { float a[8], b[8]; } mine;                // (converted once per lane, not once per product)
    for (int k = 0; k < 8; ++k) { mine.a[k] = bf2f((bf16_t)a[k]); mine.b[k] = bf2f((bf16_t)b[k]); }
    auto tab = xcemu::wave_exchange(&mine, sizeof(Dep));
    const int l = lane_id();
    const int j = l & 31;
    float bj[16];
    for (int k = 0; k < 16; ++k) bj[k] = reinterpret_cast<const Dep*>(tab[j + 32 * (k >> 3)])->b[k & 7];
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const float* a0 = reinterpret_cast<const Dep*>(tab[i])->a;
        const float* a1 = reinterpret_cast<const Dep*>(tab[i + 32])->a;
        float acc = c[r];
        for (int k = 0; k < 8; ++k) acc += a0[k] * bj[k];
        for (int k = 0; k < 8; ++k) acc += a1[k] * bj[8 + k];
        d[r] = acc;
    }
    return d;
}
inline f32x16 mfma_32x32x16_bf16_zero(s16x8 a, s16x8 b) {
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return mfma_32x32x16_bf16(a, b, z);
}
inline f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    struct Dep { float a, b; } mine{a, b};
    auto tab = xcemu::wave_exchange(&mine, sizeof(Dep));
    const int l = lane_id();
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const int j = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            Dep da, db;
            memcpy(&da, tab[i + 32 * k], sizeof(Dep));
            memcpy(&db, tab[j + 32 * k], sizeof(Dep));
            acc = fmaf(da.a, db.b, acc);
        }
        d[r] = acc;
    }
    return d;
}

// LDS DMA: lane-linear 16-byte pieces (performed immediately; the real one completes at wait_vmem() + barrier)
inline void glds16(const void* gsrc, void* lds_wave_base) {
    memcpy((unsigned char*)lds_wave_base + 16 * lane_id(), gsrc, 16);
}
inline void glds16_raw(const void* gsrc, void* lds_wave_base) { glds16(gsrc, lds_wave_base); }
inline void glds4(const void* gsrc, void* lds_wave_base) { memcpy((unsigned char*)lds_wave_base + 4 * lane_id(), gsrc, 4); }
// buffer resources (csrc/hw/xc_device.h): base + byte count; out-of-range accesses read zero / are dropped
struct BufRsrc { const unsigned char* base; uint32_t bytes; };
inline BufRsrc make_rsrc(const void* base, uint32_t bytes) { return BufRsrc{static_cast<const unsigned char*>(base), bytes}; }
inline void buf_glds16(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const uint64_t off = (uint64_t)voff + soff;
    unsigned char* dst = (unsigned char*)lds_wave_base + 16 * lane_id();
    if (off + 16 <= r.bytes) memcpy(dst, r.base + off, 16); else memset(dst, 0, 16);
}
inline const void* uniform_ptr(const void* p) { return p; }
inline void buf_glds16_raw(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) { buf_glds16(r, voff, soff, lds_wave_base); }
template <int IMM>
inline u32x4 buf_ld16(BufRsrc r, uint32_t voff, uint32_t soff) {
    const uint64_t off = (uint64_t)voff + soff + IMM;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (off + 16 <= r.bytes) memcpy(&v, r.base + off, 16);
    return v;
}
template <int IMM>
inline void buf_st16(BufRsrc r, uint32_t voff, uint32_t soff, u32x4 v) {
    const uint64_t off = (uint64_t)voff + soff + IMM;
    if (off + 16 <= r.bytes) memcpy(const_cast<unsigned char*>(r.base) + off, &v, 16);
}
template <int IMM>
inline void buf_st16_nt(BufRsrc r, uint32_t voff, uint32_t soff, u32x4 v) { buf_st16<IMM>(r, voff, soff, v); }
inline void buf_st16_nt_tracked(BufRsrc r, uint32_t voff, u32x4 v) { buf_st16<0>(r, voff, 0u, v); }
template <int IMM>
inline u32x4 buf_ld16_nt(BufRsrc r, uint32_t voff, uint32_t soff) { return buf_ld16<IMM>(r, voff, soff); }
inline u32x4 ld16_nt(const void* p) { u32x4 v; memcpy(&v, p, 16); return v; }
inline void st16_nt(void* p, u32x4 v) { memcpy(p, &v, 16); }
inline void wait_vmem() {}
// transpose read: lane c of a 16-lane group, slot j <- element (c & 3) at the address supplied by lane 4j + (c >> 2)
inline s16x4 lds_read_tr16(const void* p) {
    auto tab = xcemu::wave_exchange(&p, sizeof(p));
    const int l = lane_id(), g = l >> 4, c = l & 15;
    s16x4 out;
    for (int j = 0; j < 4; ++j) {
        const void* src;
        memcpy(&src, tab[16 * g + 4 * j + (c >> 2)], sizeof(src));
        out[j] = ((const short*)src)[c & 3];
    }
    return out;
}
inline int uniform(int v) { return v; }
inline void wave_sync() { int z = 0; (void)xcemu::wave_exchange(&z, sizeof(z)); }      // lanes are fibres: rendezvous
inline void lds_fence() { wave_sync(); }
inline void lds_drain() { wave_sync(); }
inline uint64_t realtime_10ns() { static uint64_t t = 0; return t += 1000; }
inline uint64_t shader_cycles() { return 0; }
inline uint32_t lds_base_granule() { return 0; }
inline void nap() {}
template <class T> inline void reg_keep(T&) {}
inline uint32_t opaque(uint32_t v) { return v; }

template <int OFF>
inline u32x4 lds_read16_async(const void* p) { return *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(p) + OFF); }
inline u32x2 lds_read_tr16_async(const void* p) { return __builtin_bit_cast(u32x2, lds_read_tr16(p)); }
template <int N>
inline void lds_wait(u32x4 (&)[4], u32x4 (&)[2]) {}
template <int N>
inline void lds_wait4(u32x4 (&)[4], u32x4 (&)[4]) {}
#define XC_WAIT_VMEM_LE(N) ((void)0)
inline void barrier_nodrain() { xcemu::block_barrier(); }
inline void mfma_prio(int) {}
inline void sched_fence() {}
inline void permlane32_swap(uint32_t& a, uint32_t& b) {
    struct P { uint32_t a, b; } mine{a, b};
    auto tab = xcemu::wave_exchange(&mine, sizeof(P));
    const int l = lane_id();
    P other;
    memcpy(&other, tab[l ^ 32], sizeof(P));
    if (l >= 32) a = other.b; else b = other.a;      // upper half of a <-> lower half of b
}

// whole-kernel asm units (csrc/hw/xc_device.h): not executable here -- XC_ASM_UNITS = false keeps the host from selecting such kernels
constexpr bool XC_ASM_UNITS = false;
#define XC_ASM_UNIT(TEXT, OPERANDS, CLOBBERS) abort()
inline uint32_t lds_addr(const void*) { return 0; }        // (only feeds asm units)

inline void atomic_add(float* p, float v) { *p += v; }
inline void lds_atomic_add(float* p, float v) { *p += v; }      // (fibres switch only at collectives / barriers)
inline void lds_atomic_add(int* p, int v) { *p = (int)((unsigned)*p + (unsigned)v); }
inline float fast_exp(float x) { return expf(x); }
inline float fast_exp2(float x) { return exp2f(x); }
inline float fast_rsqrt(float x) { return 1.0f / sqrtf(x); }
inline float fast_rcp(float x) { return 1.0f / x; }

}  // namespace xc

inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
