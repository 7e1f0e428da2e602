// Thin runtime layer under the host API (oct_phmm.hip): device memory, copies, stream, events, kernel launch.
// Product build = HIP runtime. With -DOCTPHMM_SIM (tests/sim only) the same host code drives the CPU wave
// simulator instead, so the whole pipeline's logic is unit-tested without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(OCTPHMM_SIM)
#include "rt_sim.hpp"
#else
#include <dlfcn.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>

namespace octphmm { namespace rt {

typedef hipStream_t Stream;
typedef hipEvent_t  Event;

inline int last_error_code = 0;
#define OCT_RT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { octphmm::rt::last_error_code = (int)e_; return false; } } while (0)

inline bool device_count(int* n) { OCT_RT_CHECK(hipGetDeviceCount(n)); return true; }
inline bool set_device(int d) { OCT_RT_CHECK(hipSetDevice(d)); return true; }
inline bool device_is_gfx950(int d)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) return false;
    return __builtin_strncmp(p.gcnArchName, "gfx950", 6) == 0;
}
inline bool stream_create(Stream* s) { OCT_RT_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); return true; }
// high = true: the device's highest stream priority (its workgroups are dispatched before those of normal streams when both wait for a CU)
inline bool stream_create_priority(Stream* s, bool high)
{
    int least = 0, greatest = 0;
    if (!high || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least) { (void)hipGetLastError(); return stream_create(s); }
    OCT_RT_CHECK(hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest));
    return true;
}
inline void stream_destroy(Stream s) { (void)hipStreamDestroy(s); }
inline bool stream_sync(Stream s) { OCT_RT_CHECK(hipStreamSynchronize(s)); return true; }
inline bool stream_idle(Stream s) { const hipError_t e = hipStreamQuery(s); if (e == hipErrorNotReady) { (void)hipGetLastError(); return false; } return true; }   // everything enqueued so far has run (errors: let the wait report them)
inline bool dev_malloc(void** p, size_t n) { OCT_RT_CHECK(hipMalloc(p, n ? n : 16)); return true; }
inline void dev_free(void* p) { if (p) (void)hipFree(p); }
inline bool mem_info(size_t* free_b, size_t* total_b) { OCT_RT_CHECK(hipMemGetInfo(free_b, total_b)); return true; }
inline bool host_pinned_malloc(void** p, size_t n) { OCT_RT_CHECK(hipHostMalloc(p, n ? n : 16, hipHostMallocDefault)); return true; }
inline void host_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
// does the runtime know this host address as page-locked memory (hipHostMalloc / hipHostRegister)? Then the DMA engines read and write it directly.
inline bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // (an address the runtime has never seen: an error, and a sticky one)
    return a.type == hipMemoryTypeHost;
}
// ... the whole array [p, p + bytes): a hipHostRegister'ed range may end before the array does, and the copy engines are pointed at all of it
inline bool host_is_pinned(const void* p, size_t bytes)
{
    if (!host_is_pinned(p)) return false;
    return bytes <= 1 || host_is_pinned((const char*)p + bytes - 1);
}
inline bool h2d(void* d, const void* h, size_t n, Stream s) { if (n) OCT_RT_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); return true; }
inline bool d2h(void* h, const void* d, size_t n, Stream s) { if (n) OCT_RT_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); return true; }
inline bool dev_memset(void* d, int v, size_t n, Stream s) { if (n) OCT_RT_CHECK(hipMemsetAsync(d, v, n, s)); return true; }
inline bool event_create(Event* e) { OCT_RT_CHECK(hipEventCreate(e)); return true; }
inline void event_destroy(Event e) { (void)hipEventDestroy(e); }
inline bool event_record(Event e, Stream s) { OCT_RT_CHECK(hipEventRecord(e, s)); return true; }
inline bool event_sync(Event e) { OCT_RT_CHECK(hipEventSynchronize(e)); return true; }
inline bool stream_wait_event(Stream s, Event e) { OCT_RT_CHECK(hipStreamWaitEvent(s, e, 0)); return true; }
inline bool event_elapsed_ms(float* ms, Event a, Event b) { OCT_RT_CHECK(hipEventElapsedTime(ms, a, b)); return true; }
inline bool launch_ok() { OCT_RT_CHECK(hipGetLastError()); return true; }
inline void clear_error() { (void)hipGetLastError(); }       // after a failed allocation that the caller recovers from: the runtime's last-error slot is sticky and the next launch_ok() would report it
template <class K> inline bool allow_lds(K kernel, size_t bytes)
{
    OCT_RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return true;
}
constexpr size_t kMaxLdsBytes = 160 * 1024;

// roctx ranges around the host-side phases (upload, run, per-slice phases, download) for rocprofv3 --marker-trace timelines. The tracer library
// is looked up at run time and only when OCT_PHMM_ROCTX is set: the product has no link-time dependency on it and pays one branch otherwise.
struct Range {
    typedef int (*PushFn)(const char*); typedef int (*PopFn)();
    static PushFn& push_fn() { static PushFn f = nullptr; return f; }
    static PopFn& pop_fn() { static PopFn f = nullptr; return f; }
    static bool enabled()
    {
        static const bool on = [] {
            if (!getenv("OCT_PHMM_ROCTX")) return false;
            void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) return false;
            push_fn() = (PushFn)dlsym(lib, "roctxRangePushA"); pop_fn() = (PopFn)dlsym(lib, "roctxRangePop");
            return push_fn() && pop_fn();
        }();
        return on;
    }
    bool live;
    explicit Range(const char* name) : live(enabled()) { if (live) push_fn()(name); }
    ~Range() { if (live) pop_fn()(); }
    Range(const Range&) = delete; Range& operator=(const Range&) = delete;
};

}} // namespace octphmm::rt

#define OCT_LAUNCH(kernel, grid, block, smem, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), smem, stream, __VA_ARGS__)
#endif
