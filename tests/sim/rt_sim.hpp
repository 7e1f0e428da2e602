// TEST INFRASTRUCTURE ONLY: host stand-in for octopus_amd/csrc/phmm_rt.hpp so that the library's host API can
// drive the CPU wave simulator (tests/sim/hipsim.hpp). "Device" memory is plain host memory.
#pragma once
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include "hipsim.hpp"

namespace octphmm { namespace rt {

typedef int Stream;
typedef int Event;
inline int last_error_code = 0;

// OCTSIM_DEVICES=n (tests/test_sim_server.py): the simulator reports n "devices" - nothing but ordinals here, but the host code's per-device bookkeeping (a region server's
// workers and handles per device, the pools' per-device cache accounting and trimming) then runs with DISTINCT device ids, which a one-GPU box cannot offer either
inline int sim_devices() { const char* e = getenv("OCTSIM_DEVICES"); const int n = e ? atoi(e) : 1; return n >= 1 && n <= 64 ? n : 1; }
inline bool device_count(int* n) { *n = sim_devices(); return true; }
inline bool set_device(int d) { return d >= 0 && d < sim_devices(); }
inline bool device_is_gfx950(int) { return true; }
inline bool stream_create(Stream* s) { *s = 0; return true; }
inline bool stream_create_priority(Stream* s, bool) { return stream_create(s); }
inline void stream_destroy(Stream) {}
inline bool stream_sync(Stream) { return true; }
inline bool stream_idle(Stream) { return true; }
// octsim_fail_next_mallocs(min_bytes, count), exported by the simulator's build of the library only: the next `count` device allocations of at least `min_bytes` "fail" (the device is
// full). What a pool does then - trim its own cache, then its siblings' (DevPool::trim_device) - has no other way to run without a GPU. The counter is shared by all handles.
inline std::atomic<long>& sim_fail_left() { static std::atomic<long> n {0}; return n; }
inline std::atomic<size_t>& sim_fail_min() { static std::atomic<size_t> n {0}; return n; }
inline std::atomic<long>& sim_fail_seen() { static std::atomic<long> n {0}; return n; }
inline bool sim_malloc_should_fail(size_t n)
{
    if (sim_fail_left().load() <= 0 || n < sim_fail_min().load()) return false;
    if (sim_fail_left().fetch_sub(1) > 0) { ++sim_fail_seen(); return true; }
    return false;
}
inline bool dev_malloc(void** p, size_t n) { if (sim_malloc_should_fail(n)) { *p = nullptr; return false; } *p = malloc(n ? n : 16); if (*p) memset(*p, 0xCD, n ? n : 16); return *p != nullptr; }
inline void dev_free(void* p) { free(p); }
inline bool mem_info(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)288 << 30; return true; }
// "pinned" allocations are remembered, so that the library's fast paths for page-locked caller buffers run in the CPU suite too
inline std::mutex& pinned_mu() { static std::mutex m; return m; }
inline std::map<const char*, size_t>& pinned_blocks() { static std::map<const char*, size_t> b; return b; }
inline bool host_pinned_malloc(void** p, size_t n) { *p = malloc(n ? n : 16); if (*p) { std::lock_guard<std::mutex> g(pinned_mu()); pinned_blocks()[(const char*)*p] = n ? n : 16; } return *p != nullptr; }
inline void host_pinned_free(void* p) { if (p) { std::lock_guard<std::mutex> g(pinned_mu()); pinned_blocks().erase((const char*)p); } free(p); }
inline bool host_is_pinned(const void* p)
{
    std::lock_guard<std::mutex> g(pinned_mu());
    auto it = pinned_blocks().upper_bound((const char*)p);
    if (it == pinned_blocks().begin()) return false;
    --it;
    return (const char*)p < it->first + it->second;
}
inline bool host_is_pinned(const void* p, size_t bytes) { return host_is_pinned(p) && (bytes <= 1 || host_is_pinned((const char*)p + bytes - 1)); }
inline bool h2d(void* d, const void* h, size_t n, Stream) { if (n) memcpy(d, h, n); return true; }
inline bool d2h(void* h, const void* d, size_t n, Stream) { if (n) memcpy(h, d, n); return true; }
inline bool dev_memset(void* d, int v, size_t n, Stream) { if (n) memset(d, v, n); return true; }
inline bool event_create(Event* e) { *e = 0; return true; }
inline void event_destroy(Event) {}
inline bool event_record(Event, Stream) { return true; }
inline bool event_sync(Event) { return true; }
inline bool stream_wait_event(Stream, Event) { return true; }
inline bool event_elapsed_ms(float* ms, Event, Event) { *ms = 0.f; return true; }
inline bool launch_ok() { return true; }
inline void clear_error() {}
template <class K> inline bool allow_lds(K, size_t) { return true; }
constexpr size_t kMaxLdsBytes = 160 * 1024;
struct Range { explicit Range(const char*) {} };

}} // namespace octphmm::rt

#define OCT_LAUNCH(kernel, grid, block, smem, stream, ...) hipsim::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })

extern "C" inline __attribute__((used, visibility("default"))) long octsim_fail_next_mallocs(size_t min_bytes, long count)     // returns how many allocations have "failed" so far
{
    octphmm::rt::sim_fail_min().store(min_bytes); octphmm::rt::sim_fail_left().store(count);
    return octphmm::rt::sim_fail_seen().load();
}
