// TEST INFRASTRUCTURE ONLY: a lockstep wave64 simulator that runs the REAL kernel source
// (octopus_amd/csrc/phmm_kernels.hpp, compiled for the host with -DOCTPHMM_SIM) on CPU coroutines, so the
// device code's logic (indexing, staging, packing, DPP shifts, reductions, walk) is unit-tested without a GPU.
// It emulates exactly the functions of octopus_amd/csrc/phmm_hw.hpp. Never compiled into liboct_phmm.so.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ucontext.h>
#include <functional>
#include <vector>

#define OCT_DEVICE inline
#define OCT_DEVICE_NOINLINE inline
#define OCT_HD inline
#define OCT_KERNEL(name) inline void name
#define OCT_MAX_THREADS(n)
#define OCT_DYN_SMEM(ptr) unsigned char* ptr = hipsim::S().smem
#define __shared__ static   /* workgroups run one after another in the simulator */

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

// ThreadSanitizer builds (tests/test_sim_sanitizers.py): every lane coroutine is announced as a fiber, or the first swapcontext onto a malloc'ed stack kills the tool
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define HIPSIM_TSAN 1
extern "C" { void* __tsan_get_current_fiber(void); void* __tsan_create_fiber(unsigned flags); void __tsan_destroy_fiber(void* fiber); void __tsan_switch_to_fiber(void* fiber, unsigned flags); }
#endif
#endif
#ifdef HIPSIM_TSAN
#define HIPSIM_TO_FIBER(f) __tsan_switch_to_fiber((f), 0)
#else
#define HIPSIM_TO_FIBER(f) ((void)0)
#endif

// Lane switches: glibc's swapcontext saves and restores the signal mask with a system call on every switch, and a simulated cross-lane operation is four switches per lane -
// the CPU suite spent most of its time there. Plain builds on x86-64 switch with a dozen instructions instead (callee-saved registers and the stack pointer; nothing here
// changes the signal mask or the floating-point control words between lanes); sanitizer builds keep ucontext, which their fibre annotations know.
#if defined(__x86_64__) && !defined(HIPSIM_TSAN) && !defined(HIPSIM_UCONTEXT)
#if defined(__has_feature)
#if !__has_feature(address_sanitizer)
#define HIPSIM_FAST_SWITCH 1
#endif
#else
#define HIPSIM_FAST_SWITCH 1
#endif
#endif
#ifdef HIPSIM_FAST_SWITCH
extern "C" void hipsim_switch(void** save_sp, void* const* load_sp);
asm(R"(
    .text
    .p2align 4
    .globl hipsim_switch
    .type hipsim_switch,@function
hipsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipsim_switch, .-hipsim_switch
)");
#endif

namespace hipsim {

enum Wait { kRun = 0, kWave = 1, kBlock = 2, kDone = 3 };

#ifdef HIPSIM_FAST_SWITCH
struct Lane { void* sp = nullptr; char* stack = nullptr; int wait = kRun; void* fiber = nullptr; };
#else
struct Lane { ucontext_t ctx; char* stack = nullptr; int wait = kRun; void* fiber = nullptr; };
#endif

struct State {
    std::vector<Lane> lanes;
#ifdef HIPSIM_FAST_SWITCH
    void* sched_sp = nullptr;
#else
    ucontext_t sched;
#endif
    void* sched_fiber = nullptr;
    uint32_t cur = 0, bid = 0, bdim = 0, gdim = 0;
    unsigned char* smem = nullptr;
    uint32_t xchg[1024];
    std::function<void()> body;
};
inline State& S() { static State s; return s; }

#ifdef HIPSIM_FAST_SWITCH
inline void yield_to_sched(int why) { State& s = S(); Lane& l = s.lanes[s.cur]; l.wait = why; hipsim_switch(&l.sp, &s.sched_sp); }
#else
inline void yield_to_sched(int why) { State& s = S(); s.lanes[s.cur].wait = why; HIPSIM_TO_FIBER(s.sched_fiber); swapcontext(&s.lanes[s.cur].ctx, &s.sched); }
#endif
inline void wave_sync() { yield_to_sched(kWave); }
inline void block_sync_impl() { yield_to_sched(kBlock); }

#ifdef HIPSIM_FAST_SWITCH
inline void lane_entry() { State& s = S(); s.body(); Lane& l = s.lanes[s.cur]; l.wait = kDone; hipsim_switch(&l.sp, &s.sched_sp); abort(); }      // (a finished lane is never resumed)
#else
inline void lane_entry() { State& s = S(); s.body(); s.lanes[s.cur].wait = kDone; HIPSIM_TO_FIBER(s.sched_fiber); swapcontext(&s.lanes[s.cur].ctx, &s.sched); }
#endif

// run one workgroup of `bdim` threads
inline void run_block(uint32_t bid, uint32_t bdim, uint32_t gdim, size_t smem_bytes)
{
    State& s = S();
    constexpr size_t kStack = 256 * 1024;
    s.bid = bid; s.bdim = bdim; s.gdim = gdim;
    std::vector<unsigned char> smem(smem_bytes + 64, 0xAB);
    s.smem = (unsigned char*)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
    if (s.lanes.size() < bdim) s.lanes.resize(bdim);
    for (uint32_t t = 0; t < bdim; ++t) {
        Lane& l = s.lanes[t];
        if (!l.stack) l.stack = (char*)malloc(kStack);
#ifdef HIPSIM_FAST_SWITCH
        {   // a fresh lane's stack as hipsim_switch expects to find it: six zeroed callee-saved registers, then lane_entry as the return address; at its entry
            // the stack pointer is 8 below a 16-byte boundary, as after a call
            void** top = (void**)(((uintptr_t)l.stack + kStack) & ~(uintptr_t)15);
            top[-1] = nullptr; top[-2] = (void*)lane_entry;
            for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
            l.sp = (void*)(top - 8);
        }
#else
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack; l.ctx.uc_stack.ss_size = kStack; l.ctx.uc_link = &s.sched;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
#endif
        l.wait = kRun;
#ifdef HIPSIM_TSAN
        l.fiber = __tsan_create_fiber(0);
#endif
    }
#ifdef HIPSIM_TSAN
    s.sched_fiber = __tsan_get_current_fiber();
#endif
    for (;;) {
        bool progressed = false, all_done = true;
        for (uint32_t t = 0; t < bdim; ++t) {
#ifdef HIPSIM_FAST_SWITCH
            if (s.lanes[t].wait == kRun) { s.cur = t; hipsim_switch(&s.sched_sp, &s.lanes[t].sp); progressed = true; }
#else
            if (s.lanes[t].wait == kRun) { s.cur = t; HIPSIM_TO_FIBER(s.lanes[t].fiber); swapcontext(&s.sched, &s.lanes[t].ctx); progressed = true; }
#endif
            if (s.lanes[t].wait != kDone) all_done = false;
        }
        if (all_done) break;
        // release waves whose live lanes all wait at a wave sync
        for (uint32_t w0 = 0; w0 < bdim; w0 += 64) {
            bool all = true, any = false;
            for (uint32_t t = w0; t < w0 + 64 && t < bdim; ++t) {
                if (s.lanes[t].wait == kDone) continue;
                any = true;
                if (s.lanes[t].wait != kWave) all = false;
            }
            if (any && all) { for (uint32_t t = w0; t < w0 + 64 && t < bdim; ++t) if (s.lanes[t].wait == kWave) s.lanes[t].wait = kRun; progressed = true; }
        }
        {
            bool all = true, any = false;
            for (uint32_t t = 0; t < bdim; ++t) { if (s.lanes[t].wait == kDone) continue; any = true; if (s.lanes[t].wait != kBlock) all = false; }
            if (any && all) { for (uint32_t t = 0; t < bdim; ++t) if (s.lanes[t].wait == kBlock) s.lanes[t].wait = kRun; progressed = true; }
        }
        if (!progressed) { fprintf(stderr, "hipsim: deadlock (divergent cross-lane op or barrier) in block %u\n", bid); abort(); }
    }
#ifdef HIPSIM_TSAN
    for (uint32_t t = 0; t < bdim; ++t) { __tsan_destroy_fiber(s.lanes[t].fiber); s.lanes[t].fiber = nullptr; }
#endif
}

template <class F>
inline void launch(uint32_t grid, uint32_t block, size_t smem_bytes, F&& f)
{
    S().body = f;
    for (uint32_t b = 0; b < grid; ++b) run_block(b, block, grid, smem_bytes);
}

inline uint32_t xchg_read(uint32_t v, int src_off_fn(uint32_t lane), uint32_t fill)
{
    State& s = S();
    const uint32_t t = s.cur, lane = t & 63;
    s.xchg[t] = v;
    wave_sync();
    const int src = src_off_fn(lane);
    const uint32_t r = src < 0 ? fill : s.xchg[(t & ~63u) + (uint32_t)src];
    wave_sync();
    return r;
}

} // namespace hipsim

namespace octphmm { namespace hw {

inline uint32_t dpp_row_shr1(uint32_t fill, uint32_t v)  { return hipsim::xchg_read(v, [](uint32_t l) { return (l & 15) == 0 ? -1 : (int)l - 1; }, fill); }
inline uint32_t dpp_row_shl1(uint32_t fill, uint32_t v)  { return hipsim::xchg_read(v, [](uint32_t l) { return (l & 15) == 15 ? -1 : (int)l + 1; }, fill); }
inline uint32_t dpp_row_shr1_z(uint32_t v) { return dpp_row_shr1(0u, v); }
inline uint32_t dpp_row_shl1_z(uint32_t v) { return dpp_row_shl1(0u, v); }
inline uint32_t dpp_wave_shr1_z(uint32_t v);
inline uint32_t dpp_wave_shl1_z(uint32_t v);
inline uint32_t dpp_wave_shr1(uint32_t fill, uint32_t v) { return hipsim::xchg_read(v, [](uint32_t l) { return l == 0 ? -1 : (int)l - 1; }, fill); }
inline uint32_t dpp_wave_shl1(uint32_t fill, uint32_t v) { return hipsim::xchg_read(v, [](uint32_t l) { return l == 63 ? -1 : (int)l + 1; }, fill); }
inline uint32_t dpp_wave_shr1_z(uint32_t v) { return dpp_wave_shr1(0u, v); }
inline uint32_t dpp_row_rol1(uint32_t v) { return hipsim::xchg_read(v, [](uint32_t l) { return (int)((l & ~15u) | ((l + 1) & 15u)); }, 0u); }
inline uint32_t dpp_wave_ror1(uint32_t v) { return hipsim::xchg_read(v, [](uint32_t l) { return (int)((l + 63) & 63); }, 0u); }
inline uint32_t dpp_wave_rol1(uint32_t v) { return hipsim::xchg_read(v, [](uint32_t l) { return (int)((l + 1) & 63); }, 0u); }
inline uint32_t dpp_wave_shl1_z(uint32_t v) { return dpp_wave_shl1(0u, v); }

inline uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int n = 0; n < 4; ++n) {
        const uint32_t s = (sel >> (8 * n)) & 0xff;
        uint32_t byte;
        if (s < 8) byte = (uint32_t)(src >> (8 * s)) & 0xff;
        else if (s == 0x0c) byte = 0;
        else if (s > 0x0c) byte = 0xff;
        else byte = ((src >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xffu : 0u;      // 8..11: sign of source byte 1, 3, 5, 7
        r |= byte << (8 * n);
    }
    return r;
}
inline uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t lo = ((a & 0xffff) * (b & 0xffff) + (c & 0xffff)) & 0xffff;
    const uint32_t hi = ((a >> 16) * (b >> 16) + (c >> 16)) & 0xffff;
    return lo | hi << 16;
}
inline void wave_lds_fence() { hipsim::wave_sync(); }
inline uint32_t shfl_xor(uint32_t v, int mask)
{
    hipsim::State& s = hipsim::S();
    const uint32_t t = s.cur;
    s.xchg[t] = v;
    hipsim::wave_sync();
    const uint32_t r = s.xchg[(t & ~63u) + ((t & 63) ^ (uint32_t)mask)];
    hipsim::wave_sync();
    return r;
}
inline uint32_t shfl(uint32_t v, int src)
{
    hipsim::State& s = hipsim::S();
    const uint32_t t = s.cur;
    s.xchg[t] = v;
    hipsim::wave_sync();
    const uint32_t r = s.xchg[(t & ~63u) + (uint32_t)src];
    hipsim::wave_sync();
    return r;
}
inline uint32_t readfirstlane(uint32_t v) { return shfl(v, 0); }
inline uint32_t readlane(uint32_t v, uint32_t src) { return shfl(v, (int)src); }
inline uint64_t ballot(bool p)
{
    hipsim::State& s = hipsim::S();
    const uint32_t t = s.cur;
    s.xchg[t] = p ? 1u : 0u;
    hipsim::wave_sync();
    uint64_t m = 0;
    for (uint32_t l = 0; l < 64; ++l) if ((t & ~63u) + l < s.bdim && s.lanes[(t & ~63u) + l].wait != hipsim::kDone && s.xchg[(t & ~63u) + l]) m |= 1ull << l;
    hipsim::wave_sync();
    return m;
}
inline uint32_t wave_max_u32(uint32_t v)
{
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
inline uint32_t wave_min_u32(uint32_t v)
{
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
inline uint32_t wave_sum_u32(uint32_t v)
{
    for (int m = 1; m < 64; m <<= 1) v += shfl_xor(v, m);
    return v;
}
inline uint32_t atomic_max_lds_u32(uint32_t* p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomic_add_lds_u32(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline unsigned long long atomic_cas_lds_u64(unsigned long long* p, unsigned long long expected, unsigned long long v) { const auto o = *p; if (o == expected) *p = v; return o; }
inline void atomic_min_lds_u64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
inline void block_sync() { hipsim::block_sync_impl(); }
inline int atomic_min_i32(int32_t* p, int32_t v) { const int32_t o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomic_min_u64(unsigned long long* p, unsigned long long v) { const auto o = *p; if (v < o) *p = v; return o; }
inline uint32_t atomic_or_u32(uint32_t* p, uint32_t v) { const auto o = *p; *p = o | v; return o; }
inline unsigned long long atomic_max_u64(unsigned long long* p, unsigned long long v) { const auto o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) { const auto o = *p; *p = o + v; return o; }
inline uint32_t atomic_and_u32(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o & v; return o; }
inline uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { const uint32_t o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomic_cas_u64(unsigned long long* p, unsigned long long expected, unsigned long long v) { const auto o = *p; if (o == expected) *p = v; return o; }
inline void lds_store_u64(unsigned long long* p, unsigned long long v) { *(volatile unsigned long long*)p = v; }
inline unsigned long long lds_load_u64(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
inline unsigned long long load_device_u64(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
inline uint32_t load_device_u32(const uint32_t* p) { return *(const volatile uint32_t*)p; }
inline void spin_pause() { hipsim::yield_to_sched(hipsim::kRun); }      // a lane that waits for another wave's mailbox: let the other lanes run
inline void store_streaming_u4(uint4* p, uint4 v) { *p = v; }
template <int P> inline void wave_priority() {}
inline uint32_t thread_idx() { return hipsim::S().cur; }
inline uint32_t block_idx() { return hipsim::S().bid; }
inline uint32_t block_dim() { return hipsim::S().bdim; }
inline uint32_t grid_dim() { return hipsim::S().gdim; }

}} // namespace octphmm::hw
