// xc_device.h (gfx950 build) -- the thin vocabulary every kernel in csrc/kernels is written in.
//
// This is the only header that names HIP/CDNA4 builtins.  tests/emu/xc_device.h provides the same
// vocabulary on top of a wave64 fibre emulator so the CPU test-suite can execute the very same kernel
// sources; the include path (-I csrc/hw  vs  -I tests/emu) selects which one a build sees.  The product
// library (libxclip_hip.so) is only ever built from THIS header by hipcc --offload-arch=gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XC_DEV __device__ __forceinline__
// kernel attribute: plan for four waves per SIMD, i.e. at most 128 VGPRs per lane
#define XC_FOUR_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(4)))
#define XC_HOST_DEV __host__ __device__ __forceinline__
// dynamic LDS carve base, 16-byte aligned (cdna_hip_programming.md Guideline 17)
#define XC_LDS_DYNAMIC(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

// kernels that carve more than the default 64 KiB of dynamic LDS must opt in once per function
#define XC_ALLOW_LDS(kernel, bytes)                                                                          \
    do {                                                                                                     \
        static bool done_ = false;                                                                           \
        if (!done_) {                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));             \
            done_ = true;                                                                                    \
        }                                                                                                    \
    } while (0)

// compute units of the current device (persistent kernels launch one work-group per CU)
inline int xc_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// the CU count work-SPLITTING policies plan for (split-K slice counts): the device's.  (The emulator twin answers 256 here while running
// its persistent grids on 3 "CUs": the CPU suite then takes the MI355X's split decisions and exercises the same slabs.)
inline int xc_policy_cus() { return xc_num_cus(); }

namespace xc {

typedef uint16_t bf16_t;                                            // raw bfloat16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));            // 8 x bf16  (MFMA A/B operand, 4 VGPR)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));          // 32x32 MFMA accumulator
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));     // one 16-byte global/LDS transaction
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

XC_DEV float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; clang lowers the __bf16 conversion to v_cvt_pk_bf16_f32 on gfx950
XC_DEV bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// two floats -> one dword of two bf16 (lo in bits 0-15), round-to-nearest-even: ONE v_cvt_pk_bf16_f32
XC_DEV uint32_t f2bf_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

XC_DEV int lane_id() { return threadIdx.x & 63; }
XC_DEV int wave_id() { return threadIdx.x >> 6; }
XC_DEV void sync() { __syncthreads(); }

// ---- wave-level data exchange (64 lanes) ---------------------------------------------------------
XC_DEV float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
XC_DEV int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
XC_DEV float shfl(float v, int src) { return __shfl(v, src, 64); }
XC_DEV int shfl(int v, int src) { return __shfl(v, src, 64); }

// wave-wide votes (the result is uniform)
XC_DEV bool wave_all(bool pred) { return __all(pred) != 0; }
XC_DEV bool wave_any(bool pred) { return __any(pred) != 0; }
// bit l = the predicate of lane l, lanes 0..31 (wave-uniform result)
XC_DEV uint32_t wave_ballot32(bool pred) { return (uint32_t)__ballot(pred); }
// bit l = the predicate of lane l, all 64 lanes
XC_DEV uint64_t wave_ballot64(bool pred) { return (uint64_t)__ballot(pred); }
XC_DEV int popc64(uint64_t v) { return __popcll(v); }

XC_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
XC_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// hi <- max over the wave of hi, lo <- min over the wave of lo (both wave-uniform afterwards; the two butterfly chains are independent,
// so their cross-lane latencies overlap)
XC_DEV void wave_max_min(float& hi, float& lo) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float h2 = __shfl_xor(hi, m, 64), l2 = __shfl_xor(lo, m, 64);
        hi = fmaxf(hi, h2);
        lo = fminf(lo, l2);
    }
}

// ---- matrix cores ---------------------------------------------------------------------------------
// D = A*B + C, one wave.  32x32x16 bf16: lane l supplies A[i = l&31][k = 8*(l>>5) + 0..7] and
// B[k = 8*(l>>5) + 0..7][j = l&31]; D/C: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
// (cdna_hip_programming.md section 3).
XC_DEV f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the same with C = 0 (the first k-block of a tile: no accumulator initialisation pass; the zero is an inline constant)
XC_DEV f32x16 mfma_32x32x16_bf16_zero(s16x8 a, s16x8 b) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), z, 0, 0, 0);
}
// 32x32x2 f32 (exact fp32 fma chain): lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
XC_DEV f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// c + a.lo * b.lo + a.hi * b.hi of two packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16): dot products of single rows that are not
// worth an MFMA block (the 257th token of the text encoder in attention3.h)
XC_DEV float dot2_bf16(uint32_t a, uint32_t b, float c) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// ---- LDS DMA and transpose read (gfx950) -----------------------------------------------------------------
// glds16: every lane copies 16 bytes from its own global address to LDS at lds_wave_base + 16 * lane
// (global_load_lds_dwordx4: the destination is wave-uniform base + lane * 16, the source is per lane; no VGPR
// round trip, completion is tracked by vmcnt).  Measured layout: tools/probes/tr_read_probe.txt.
XC_DEV void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// glds16 as an asm unit the compiler's wait-count pass does not see (see buf_glds16_raw below): for prefetches that must stay in flight
// across later LDS reads.  Every wait for the piece is the caller's.
XC_DEV void glds16_raw(const void* gsrc, void* lds_wave_base) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(m) : "memory");
}
// glds4: 4 bytes per lane (global_load_lds_dword) -- used as an L2 PREFETCH: one lane per 128-byte line, landing in a
// scratch LDS area nobody reads; costs no VGPR and is tracked by vmcnt like any other DMA piece
XC_DEV void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// ---- buffer resources: wave-uniform base + 32-bit byte offsets, hardware bounds check ---------------------------------------
// A raw buffer descriptor (128 bits in SGPRs) built from wave-uniform values (cdna_hip_programming.md T8 / T20): memory instructions
// through it take a per-lane 32-bit byte offset in ONE VGPR plus a scalar byte offset in an SGPR -- no 64-bit per-lane address
// arithmetic -- and an access that reaches past `bytes` reads zero / is dropped (ragged tile edges need no clamps or masks).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
XC_DEV BufRsrc make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// LDS DMA through a descriptor (buffer_load_dwordx4 ... offen lds): lane copies 16 bytes from base + voff + soff to
// lds_wave_base + 16 * lane; tracked by vmcnt like glds16
XC_DEV void buf_glds16(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// The same LDS DMA as ONE asm unit, i.e. invisible to the compiler's wait-count pass: behind the builtin form the pass treats every later
// LDS read that might alias the landing zone as dependent and inserts s_waitcnt vmcnt(0) in front of it -- a prefetch issued at the top of
// a step was waited for in the middle of the same step (attention6.h, round 6: the streamed ring ran at the DMA's latency).  With this
// form EVERY wait for the piece is the caller's (wait_vmem / XC_WAIT_VMEM_LE + a barrier before another wave reads the landing zone).
// `lds_wave_base` must be wave-uniform (it goes to M0 through an SGPR).
// (a wave-uniform pointer the compiler cannot prove uniform, forced into scalar registers: the asm's descriptor operand must be one)
XC_DEV const void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
XC_DEV void buf_glds16_raw(BufRsrc r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(r), "s"(so), "s"(m) : "memory");
}
// 16-byte store at base + voff + soff + IMM (IMM: the instruction's 12-bit immediate offset)
// 16-byte load from base + voff + soff + IMM (zero past the descriptor's extent); counted by the compiler's own vmcnt bookkeeping
template <int IMM>
XC_DEV u32x4 buf_ld16(BufRsrc r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff + IMM, (int)soff, 0));
}
// the same load with the non-temporal hint (a line read once: it need not displace what the L2 holds)
template <int IMM>
XC_DEV u32x4 buf_ld16_nt(BufRsrc r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff + IMM, (int)soff, 2));
}
// 16-byte store at base + voff + soff + IMM (IMM: the instruction's 12-bit immediate offset), as one asm unit WITH two wait states
// behind it (so the store is not part of the compiler's vmcnt bookkeeping: every wait around it is explicit or only ever too long).
// The wait states are load-bearing: a 128-bit buffer store reads its four data VGPRs a little after it issues, and a
// VALU write to the first of them in the very next instruction reached memory instead of the store's value -- rarely, in 4-lane
// groups, only with a REGISTER soffset (for which hipcc / ROCm 7.2 inserts no wait state: its hazard table covers the immediate-soffset
// form only), first seen as garbage in one dword of a few rows of the residual epilogue (tools/debug/res_epilogue_check.py).
// The five wait states IN FRONT are load-bearing too: the compiler does not look into an asm statement, and when it has spilled the
// scalar offset (or the descriptor) to a VGPR lane it reloads it with v_readlane right before the statement -- a VALU write of an SGPR
// that a vector-memory instruction reads needs 5 wait states.  Without them the fp32 slab epilogue stored whole tiles to stale offsets
// (tools/debug/slab_epilogue_check.py; the emulator cannot see it).
template <int IMM>
XC_DEV void buf_st16(BufRsrc r, uint32_t voff, uint32_t soff, u32x4 v) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
}
// wait until every outstanding vector-memory operation of this wave (LDS DMA included) has completed
// the same store with the non-temporal hint (streaming: the line is the first to leave the L2)
template <int IMM>
XC_DEV void buf_st16_nt(BufRsrc r, uint32_t voff, uint32_t soff, u32x4 v) {
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4 nt\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
}
// The same non-temporal store as a BUILTIN with an immediate-zero scalar offset (the form the compiler's own hazard table covers), i.e. one
// the compiler's wait-count pass KNOWS about.  For code that mixes such stores with compiler-tracked loads (buf_ld16 / plain loads): behind
// an asm store the pass under-counts -- it waits with vmcnt(#its own younger loads), and since the counter retires in order that drains the
// asm stores issued in between as well: every use of a prefetched load then waits for the previous batch of stores to be ACKNOWLEDGED
// (gemm9.h's epilogue: one store round trip per 32-row group).  With the builtin the pass counts the stores and leaves them in flight.
XC_DEV void buf_st16_nt_tracked(BufRsrc r, uint32_t voff, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 2); }
// plain 16-byte global accesses with the non-temporal hint (streamed once: first use is last use)
XC_DEV u32x4 ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
XC_DEV void st16_nt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
XC_DEV void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// lds_read_tr16 (ds_read_b64_tr_b16): within each 16-lane group, lane c (slot j) receives the 16-bit element
// at addr[lane 4j + (c >> 2) of the group] + (c & 3): a 4 x 16 block whose rows are addressed by the lanes is
// returned transposed -- 4 consecutive ROWS of one column per lane (tools/probes/tr_read_probe.txt).
XC_DEV s16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// ---- LDS reads outside the compiler's wait-count model -----------------------------------------------------------
// hipcc's waitcnt pass drains lgkmcnt to 0 before the first use of ANY pending ds_read result in the GEMM loops (measured:
// every fragment batch was followed by s_waitcnt lgkmcnt(0) ahead of the MFMAs it was supposed to overlap).  These
// variants issue the read as volatile asm -- invisible to that pass -- and lds_wait<N>() is the hand-placed s_waitcnt; the
// registers passed to it are "modified" by it as far as the compiler knows, so no consumer can be moved above the wait.
// Rule for users: nothing but MFMAs on OTHER registers between a read and its wait (straight-line code, no loop back-edge).
XC_DEV uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <int OFF>
XC_DEV u32x4 lds_read16_async(const void* p) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr(p)), "n"(OFF));
    return v;
}
XC_DEV u32x2 lds_read_tr16_async(const void* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr(p)));
    return v;
}
template <int N>
XC_DEV void lds_wait(u32x4 (&a)[4], u32x4 (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
}
// the same wait for a 4 + 4 fragment set (gemm7.h)
template <int N>
XC_DEV void lds_wait4(u32x4 (&a)[4], u32x4 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
XC_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// orders this wave's LDS traffic around a wave-private hand-off (lanes exchange data through LDS without a work-group
// barrier): the hardware executes a wave's LDS instructions in order; this only stops the compiler from reordering.
XC_DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }
// between a wave's LDS stores and its loads of what OTHER lanes stored (the LDS serves a wave's operations in order: nothing to wait
// for, the compiler must only keep the order)
// every LDS operation of this wave has completed (before something else overwrites what it read)
XC_DEV void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// measurement builds: keep a register value alive / opaque to the optimiser
template <class T> XC_DEV void reg_keep(T& v) { asm volatile("" : "+v"(v)); }
// the value, as something the optimiser cannot see through: what is computed from it is recomputed where it is used instead of being
// hoisted out of a loop and carried in registers (store addresses of an epilogue that has none to spare)
XC_DEV uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// constant-rate (100 MHz) timestamp
XC_DEV uint64_t realtime_10ns() { return __builtin_amdgcn_s_memrealtime(); }
XC_DEV uint64_t shader_cycles() { return __builtin_amdgcn_s_memtime(); }
// first LDS granule of this work-group on its CU (HW_REG_LDS_ALLOC[7:0]): 0 for the work-group that got the CU's LDS first
XC_DEV uint32_t lds_base_granule() { return __builtin_amdgcn_s_getreg((8 - 1) << 11 | 0 << 6 | 6); }
XC_DEV void nap() { __builtin_amdgcn_s_sleep(8); }
XC_DEV void lds_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

// counted wait: returns when at most N of this wave's vector-memory operations (LDS DMA included) are still outstanding
#define XC_WAIT_VMEM_LE(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// bare s_barrier (no implicit vmcnt(0) drain, unlike __syncthreads() with LDS DMA in flight -- cdna_hip_programming.md
// section 5 "Pipelining across barriers"); pending LDS reads of this wave are drained first
XC_DEV void barrier_nodrain() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// pins the instruction order at this point (the compiler may not move anything across it)
XC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
XC_DEV void mfma_prio(int on) { if (on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
// lanes 32-63 of `a` swap with lanes 0-31 of `b` (v_permlane32_swap): widens row-per-lane epilogue stores to 16 bytes
XC_DEV void permlane32_swap(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// ---- whole-kernel asm units (gemm8.h) -----------------------------------------------------------------------------------------------
// A kernel body written as ONE asm statement: TEXT names every register it uses by hand, OPERANDS are its inputs (macro name expanding to the
// "v"(..), "s"(..) list), CLOBBERS the macro name of its clobber list (which is also what makes the kernel descriptor allocate the
// registers).  The emulator twin cannot execute such a body: there XC_ASM_UNITS is false and the host never selects these kernels.
constexpr bool XC_ASM_UNITS = true;
#define XC_ASM_UNIT(TEXT, OPERANDS, CLOBBERS) asm volatile(TEXT : : OPERANDS : CLOBBERS)

XC_DEV void atomic_add(float* p, float v) { atomicAdd(p, v); }
// adds into LDS without a return value (counted by lgkmcnt).  Measured on MI355X (tools/probes/lds_atomic_rate_probe.hip, 8 waves per CU issuing,
// conflict-free addresses): ds_add_u32 4.1 cycles per wave-instruction (= ds_write_b32), ds_add_u64 8.0, ds_add_f32 192 -- the fp32 form
// is served one lane at a time (3 cycles per lane) and is 47 x slower than the integer form: accumulate in fixed point (attention6.h)
XC_DEV void lds_atomic_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XC_DEV void lds_atomic_add(int* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

XC_DEV float fast_exp(float x) { return __expf(x); }
XC_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }        // bare v_exp_f32
XC_DEV float fast_rsqrt(float x) { return rsqrtf(x); }
XC_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

}  // namespace xc
