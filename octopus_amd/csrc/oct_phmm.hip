// liboct_phmm.so — host side of the C ABI in include/oct_phmm.h: validation, HBM layout, kernel sequencing.
// All compute happens in the HIP kernels of phmm_kernels.hpp; there is no CPU compute path here.
#include "../../include/oct_phmm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <memory>
#include <unordered_map>
#include <new>
#include <vector>

#include "phmm_rt.hpp"
#include "phmm_kernels.hpp"
#include "phmm_readout.hpp"
#include "phmm_error_model.hpp"
#include "phmm_error_model_tables.hpp"
#include "phmm_custom_error_model.h"

using namespace octphmm;

// ---------------------------------------------------------------------------------------------------------------
// handle / batch objects
// ---------------------------------------------------------------------------------------------------------------
// Size-class cache of device allocations, one per handle: a populate call per active region makes dozens of small allocations, and
// hipMalloc / hipFree take a process-wide lock and synchronise the device, which serialises the caller's region threads. Blocks are
// returned here instead and handed out again; everything goes back to the runtime when the handle is destroyed (or past the cap).
struct DevPool {
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> live;
    size_t cached = 0;
    std::mutex mu;                                       // a pool is its handle's, i.e. one thread's - except when ANOTHER handle's allocation fails and that handle trims its siblings' caches
    // What all the pools of one device may hold back between them (a region server runs several handles per GPU, INTEGRATION's populate patch one per caller thread:
    // with a cap per handle a handle could report out-of-memory while its siblings sat on tens of GB of free blocks - ADVICE r04; round 5's 2,000-scenario shape fuzz met exactly that).
    static std::atomic<size_t>& device_cached(int dev) { static std::atomic<size_t> c[64]; return c[(unsigned)dev & 63u]; }
    static std::mutex& registry_mu() { static std::mutex m; return m; }
    static std::vector<DevPool*>& registry() { static std::vector<DevPool*> r; return r; }
    int device = 0;                                      // written once, under registry_mu (set_device), before the handle's first allocation; trim_device reads it under the same lock
    void set_device(int dev) { std::lock_guard<std::mutex> lk(registry_mu()); device = dev; }
    static constexpr size_t kDeviceCacheCap = (size_t)128 << 30;
    static constexpr size_t kCacheCap = (size_t)64 << 30;       // (288 GB of HBM: a handle that streams 6,250-region batches - 20 GB resident each - paid a 20 GB hipMalloc + hipFree, 0.4 s, per call with the cap at 16 GB)
    DevPool() { std::lock_guard<std::mutex> lk(registry_mu()); registry().push_back(this); }
    ~DevPool() { std::lock_guard<std::mutex> lk(registry_mu()); auto& r = registry(); r.erase(std::remove(r.begin(), r.end(), this), r.end()); }
    DevPool(const DevPool&) = delete; DevPool& operator=(const DevPool&) = delete;
    // Powers of two up to 1 GB (a thread's region calls differ in size by orders of magnitude - 20 to 5,000 reads, 1 to 200 haplotypes: with finer classes most
    // calls of a run's first thousands met a size nobody had freed yet and paid a hipMalloc, which synchronises the device), eight classes per octave beyond
    // (resident many-gigabyte batches are not rounded up by half of themselves).
    static size_t size_class(size_t n)
    {
        if (n < 4096) return 4096;
        size_t c = 4096; while (c < n && c < ((size_t)1 << 30)) c <<= 1;
        if (c >= n) return c;
        size_t p2 = (size_t)1 << 30; while ((p2 << 1) <= n) p2 <<= 1;      // largest power of two <= n
        const size_t step = p2 >> 3;
        return (n + step - 1) / step * step;
    }
    // the device has no room: every pool of this device gives its cached (free) blocks back to the runtime - this one's first, then its siblings'
    static void trim_device(int dev)
    {
        std::lock_guard<std::mutex> lk(registry_mu());
        for (DevPool* q : registry()) if (q->device == dev) q->trim();
    }
    bool alloc(void** p, size_t n)
    {
        const size_t c = size_class(n);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_blocks.lower_bound(c);             // the smallest cached block that fits, if it is not wastefully large (small blocks: up to 8x, nobody misses those bytes)
            if (it != free_blocks.end() && (it->first <= c + c / 2 || it->first <= std::min<size_t>(8 * c, (size_t)64 << 20))) {
                *p = it->second; const size_t got = it->first; free_blocks.erase(it); cached -= got; device_cached(device) -= got; live[*p] = got; return true;
            }
        }
        if (!rt::dev_malloc(p, c)) {
            rt::clear_error();
            trim();                                     // give this pool's cached blocks back and retry ...
            if (!rt::dev_malloc(p, c)) {
                rt::clear_error();
                trim_device(device);                    // ... then every sibling's
                if (!rt::dev_malloc(p, c)) { rt::clear_error(); return false; }
            }
        }
        std::lock_guard<std::mutex> lk(mu);
        live[*p] = c;
        return true;
    }
    void release(void* p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) { rt::dev_free(p); return; }
        const size_t c = it->second; live.erase(it);
        if (cached + c > kCacheCap || device_cached(device).load() + c > kDeviceCacheCap) { rt::dev_free(p); return; }
        free_blocks.emplace(c, p); cached += c; device_cached(device) += c;
    }
    void trim()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : free_blocks) rt::dev_free(kv.second);
        free_blocks.clear(); device_cached(device) -= cached; cached = 0;
    }
};

struct oct_phmm_handle {
    oct_phmm_config cfg;
    DevPool pool;
    void* stage = nullptr; size_t stage_bytes = 0;       // pinned host staging: all input arrays of a batch go up in ONE copy
    void* out_stage = nullptr; size_t out_stage_bytes = 0;   // pinned landing zone for result copies (oct_phmm_populate)
    std::vector<void*> stat_stage_free;                  // pinned landing blocks for a run's counters (one per batch in flight, recycled)
    void* get_stat_stage(size_t bytes) { if (!stat_stage_free.empty()) { void* p = stat_stage_free.back(); stat_stage_free.pop_back(); return p; } void* p = nullptr; return rt::host_pinned_malloc(&p, bytes) ? p : nullptr; }
    std::vector<rt::Event> ev_pool;                      // recycled timing / completion events
    bool timing = false;                                 // HIP-event timing of the DP launches (oct_phmm_set_timing; bench.py's roofline leg)
    bool get_event(rt::Event* e) { if (!ev_pool.empty()) { *e = ev_pool.back(); ev_pool.pop_back(); return true; } return rt::event_create(e); }
    void put_event(rt::Event e) { ev_pool.push_back(e); }
    int band = 0;
    bool wide = false;                                   // int32 lanes (Config::use_int_scores)
    int  lanes_c = 1;                                    // band diagonals per lane on the streaming path (band / 64) for bands 128, 256
#ifndef OCT_MAX_SLICES
#define OCT_MAX_SLICES 8                                 // (a build-time knob for A/B libraries: tools/build_variant.sh)
#endif
    static constexpr int kMaxSlices = OCT_MAX_SLICES;
    rt::Stream stream {};                                 // slice 0 / uploads / downloads
    rt::Stream extra_streams[kMaxSlices] {};              // further slices run on their own streams so that latency-bound and VALU-bound kernels overlap
    bool main_stream_high_priority = false;
                                                          // batch of several slices - 12.8 M-pair step 29.2 -> 29.5 ms, stream-hq 22.4 -> 22.8 - but two calls in flight lose more without it); set by oct_phmm_batch_run
    rt::Event ev_ready {};
    uint32_t* bp[kMaxSlices] {}; size_t bp_bytes[kMaxSlices] {};   // traceback scratch per slice, grown on demand
    rt::Stream slice_stream(int i) const { return i == 0 ? stream : extra_streams[i - 1]; }
    // traceback scratch budget: large, so that all traceback tasks of a batch run in ONE DP launch and ONE walk launch (the walk
    // is a latency-bound pointer chase that needs every task in flight to hide it); MI355X has 288 GB. OCT_PHMM_BP_BUDGET_GB overrides.
    size_t bp_budget = (size_t)96 << 30;
    // error model for in-call penalty vectors (oct_phmm_set_error_model)
    bool has_model = false; oct_phmm_error_model model {};
    std::shared_ptr<const em::CustomIndelModel> custom;   // oct_phmm_set_custom_error_model: gap penalties from a model file's rows (host threads only), SNV vectors from `model`
    std::vector<uint8_t> last_align_counts; bool last_align_device_map = false;   // oct_phmm_align_candidate_counts
    int fail_bp_allocs = 0;                              // test hook, see ensure_bp
    bool probe_ready = false; rt::Stream probe_stream {}; unsigned long long* d_probe = nullptr; unsigned long long* h_probe = nullptr;   // oct_phmm_probe_clock
    oct_phmm_error_model* d_model = nullptr;             // device copy, made on first use
    // canonical-window pass of an upload (exact de-duplication of pairs): scratch and the two power tables, kept and grown on demand
    void* dedup_scratch = nullptr; size_t dedup_scratch_bytes = 0; uint64_t* d_pw = nullptr; uint64_t* d_pwinv = nullptr; size_t pw_n = 0;
};

struct oct_phmm_batch {
    double* out_landing = nullptr;   // oct_phmm_populate with a page-locked `out`: results are copied there by the DMA engine, no landing zone of the handle's in between
    DevBatch d {};
    std::vector<void*> allocs;
    // host-side shape + small copies needed for error reporting
    uint32_t n_reads = 0, n_haps = 0, n_rows = 0, n_regions = 0, t_cap = 0, lh_cap = 0, n_hap_bases = 0;
    uint64_t n_pairs = 0, n_out = 0;
    std::vector<uint32_t> h_roff, h_hoff, h_blk_hap, h_blk_read0; std::vector<int64_t> h_rbegin, h_hbegin;
    std::vector<uint32_t> h_hap_region, h_reg_hap0; std::vector<uint64_t> h_hap_out_off;      // for the genotype read-out
    // run state
    struct Slice {                     // whole haplotypes [hap0, hap1) = pairs [pair0, pair1) = outputs [out0, out1)
        uint32_t hap0 = 0, hap1 = 0, blk0 = 0, blk1 = 0, n_tiles = 0; uint64_t pair0 = 0, pair1 = 0, out0 = 0, out1 = 0;
        uint4* cnt = nullptr; uint4* tile_sums = nullptr; uint4* d_totals = nullptr; uint4 totals {};
        bool scan_fused = false;      // this run scanned the counts tile-locally (k_scan_fused): k_emit adds the tile prefixes, a flavour's traceback and late-start lists share one launch
        uint4* cnt_late = nullptr; uint4* tile_sums_late = nullptr; uint4* d_totals_late = nullptr; uint4 totals_late {};   // right-flank-only traceback tasks (x fast, y generic)
        DevTask* d_tasks = nullptr; size_t tasks_cap = 0; TraceEnd* d_ends = nullptr; size_t ends_cap = 0;
        DevTask* d_tasks_sorted = nullptr; size_t sorted_cap = 0;    // the fast-cost lists after k_pair_sort (window pairing)
        unsigned long long* d_keys = nullptr; size_t keys_cap = 0;   // align mode: per traceback task
        uint32_t seg0 = 0, n_segs = 0, n_seg_tiles = 0;              // k_dedup_match: this slice's (region, haplotype range) segments and their 64-read tiles
        bool resumes = false; rt::Event matched {};                  // its first region began in the previous slice: its matcher waits for that slice's, its epilogue for the earlier slices' results
        rt::Event done {};
    };
    std::vector<Slice> slices;
    uint4* d_hap_base = nullptr; uint4* d_totals = nullptr;
    bool late_ok = false; uint4* d_pair_cnt_late = nullptr; uint4* d_hap_base_late = nullptr; uint4* d_totals_late = nullptr;
    double* d_out = nullptr;
    uint32_t n_tasks[kNumKinds] = {0, 0, 0, 0};
    unsigned long long h_stats[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> h_win_blocks;                                   // canonical windows: (region, key class) of every k_window_region workgroup (upload)
    std::vector<uint32_t> h_tab_base, h_tab_mask;                     // canonical windows: first slot and mask of every region's hash table (upload)
    bool dedup = false, dedup_tables = false; std::vector<DedupSeg> h_segs; DedupSeg* d_segs = nullptr;   // exact de-duplication of pairs (phmm_kernels.hpp)
    std::vector<unsigned long long> h_stat_stripes;
    unsigned long long* stat_stage = nullptr;                   // pinned landing block of the counters' copy (the handle's; pageable destinations cost a staged copy per call)
    bool synced = false;                                        // oct_phmm_batch_wait has seen the handle's streams idle since the last run
    unsigned long long h_err_key = ~0ull;
    bool ran = false, device_map = false, stats_clear = false;       // stats_clear: the upload's table kernel left the counters zeroed (the first run skips its memset)
    // device-sized launches (one slice, scratch for the host-known task bound fits): no host read-back of the task counts in the middle of a step
    bool dsl = false; uint32_t dsl_list_bound = 0; size_t dsl_total_bound = 0; int dsl_flavours = 3;
    uint32_t dsl_trace_cap = 0;   // tasks a traceback list may hold (the scratch provisioned for it); a batch that needs more is repeated with host-sized launches   // tasks one list / all six lists can hold at most (padding included)
    rt::Event ev_fork {}, ev_join {}, ev_hashes {};
    // align mode (oct_phmm_align)
    bool align_mode = false; uint32_t cig_cap = 0;
    double* d_aln_lik = nullptr; uint32_t* d_aln_mpos = nullptr; uint32_t* d_aln_n = nullptr; uint32_t* d_aln_ops = nullptr; uint32_t* d_err_flags = nullptr;
    double* early_out = nullptr;  // oct_phmm_populate: copy every slice's rows to the caller as soon as its epilogue is done
    bool stream = false;          // streaming DP path (k_dp_wide): band 128/256, or band 64 with reads/haplotypes too long for LDS
    bool rows32 = false;          // ... its row form k_dp_rows: band 16 with int32 lanes (four tasks per wave, fast-cost and generic lists, read record rows, operands shared along the row)
    bool multi_wave = false;      // ... its multi-wave form k_dp_mw: bands 128 / 256 with int32 lanes (one task per workgroup, fast-cost and generic lists)
    bool map_big = false;         // haplotypes too long for the LDS-resident k-mer mapper
    int  map_lanes = 0;           // > 0: k_kmer_map_lanes with this many lanes (= reads) per workgroup
    bool fast_adds = false;       // no int16 lane of this batch can wrap (bounds below): k_dp may add with v_add_u32
    bool pair_ok = false;         // window pairing (k_pair_sort + the PAIRED segments of k_dp): big host-sized batches on the packed int16 fast-cost kernels
    uint32_t* d_paired_end[3] = {nullptr, nullptr, nullptr};   // per haplotype: score-only fast, traceback fast, late-start fast
    uint32_t* d_blk_hap = nullptr; uint32_t* d_blk_read0 = nullptr; uint32_t n_map_blocks = 0;
    uint32_t map_reads_per_block = 64;   // reads one k_kmer_map workgroup walks with the haplotype's bins staged once; fewer for small batches (latency)
    double dp_ms = 0; uint32_t dp_launches = 0;
    std::vector<std::pair<rt::Event, rt::Event>> timers;       // one (start, stop) pair per DP launch
    std::vector<int> timer_kind;
    double kind_ms[kNumKinds] = {0, 0, 0, 0}; uint32_t kind_launches[kNumKinds] = {0, 0, 0, 0};
    oct_phmm_handle* owner = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// Every environment switch of the library, in one place (documented for callers in INTEGRATION.md section 7). None is needed in
// production. They are read when a handle is created or a batch is uploaded - never by a kernel - and fall in three groups:
//   profiling    OCT_PHMM_TIMING, OCT_PHMM_ROCTX (phmm_rt.hpp), OCT_PHMM_SERVER_PROFILE, OCT_PHMM_MAP_STATS, OCT_PHMM_UPLOAD_PROFILE
//   A/B choices between paths with identical results    OCT_PHMM_SLICES, OCT_PHMM_EXACT_ADDS, OCT_PHMM_PENALTIES, OCT_PHMM_MAP_COUNT_ONLY, OCT_PHMM_LANE_MAPPER, OCT_PHMM_BP_BUDGET_GB,
//                OCT_PHMM_DEDUP, OCT_PHMM_DEVICE_SIZED, OCT_PHMM_WALK_STAGE, OCT_PHMM_MULTI_WAVE, OCT_PHMM_MW_PLANES, OCT_PHMM_DP_ROWS, OCT_PHMM_DSL_MERGE_DP, OCT_PHMM_HOST_MAPPED, OCT_PHMM_SERVER_WORKERS,
//                OCT_PHMM_JOIN_LATE, OCT_PHMM_LATE_START, OCT_PHMM_REC_CHUNK, OCT_PHMM_PAIRED (round 6). Switches whose A/B is recorded as lost were retired in round 6 (DESIGN.md section 9 lists the survivors).
//   test hooks that push SMALL batches through the code paths only large ones take    OCT_PHMM_LATE_MIN_PAIRS, OCT_PHMM_BP_BUDGET_KB,
//                OCT_PHMM_STAGE_MAX_KB, OCT_PHMM_BIG_MAPPER, OCT_PHMM_DEDUP_HASH_BITS (both de-duplication hashes cut to a few bits: collisions),
//                OCT_PHMM_DSL_TRACE_PER_PAIR, OCT_PHMM_TEST_FAIL_BP_ALLOCS (the first traceback-scratch allocations "fail"), OCT_PHMM_SCAN_ONE_LAUNCH_MAX
// ---------------------------------------------------------------------------------------------------------------
namespace tune {
// Switches reach the library in two ways, neither by accident:
//   oct_phmm_test_set(name, value)   a process-wide override table (tests, bench.py's single-slice roofline leg, A/B tools);
//   the environment                  ONLY when OCT_PHMM_ENV_SWITCHES is set in it (tests/conftest.py, tools/*.sh): a variant caller's environment that happens to
//                                    hold an OCT_PHMM_* variable does not steer the product.
// The profiling switches (stderr reports, HIP-event timing, roctx ranges) are read from the environment directly: they change no result and no code path.
inline std::mutex& switch_mu() { static std::mutex m; return m; }
inline std::map<std::string, std::string>& switch_table() { static std::map<std::string, std::string> t; return t; }
inline std::atomic<bool>& switch_table_used() { static std::atomic<bool> u {false}; return u; }
inline const char* get(const char* name)
{
    static const bool env_ok = getenv("OCT_PHMM_ENV_SWITCHES") != nullptr;
    if (!env_ok && !switch_table_used().load(std::memory_order_acquire)) return nullptr;     // production: no table, no environment - a call asks ~30 times, from every region thread
    {
        std::lock_guard<std::mutex> lk(switch_mu());
        auto it = switch_table().find(name);
        if (it != switch_table().end()) return it->second.c_str();          // (stays valid: oct_phmm_test_set retires replaced strings instead of freeing them)
    }
    return env_ok ? getenv(name) : nullptr;
}
inline bool prof_flag(const char* name) { return getenv(name) != nullptr; }
inline bool flag(const char* name) { return get(name) != nullptr; }
inline bool number(const char* name, long long* v) { const char* e = get(name); if (!e) return false; *v = atoll(e); return true; }
inline bool timing()          { return prof_flag("OCT_PHMM_TIMING"); }             // HIP events around every DP launch (bench.py's roofline leg)
inline bool server_profile()  { return prof_flag("OCT_PHMM_SERVER_PROFILE"); }     // region server: where the workers' time goes, printed at destroy
inline bool map_stats()       { return prof_flag("OCT_PHMM_MAP_STATS"); }          // k-mer mapper: pairs decided by the shortcut / counted, printed per run
inline bool exact_adds()      { return flag("OCT_PHMM_EXACT_ADDS"); }         // keep v_pk_add_u16 even where the host bound allows v_add_u32
inline size_t pinned_min_bytes(size_t dflt) { long long kb; return number("OCT_PHMM_PINNED_MIN_KB", &kb) && kb >= 0 ? (size_t)kb << 10 : dflt; }   // test hook: arrays / results from this size on are asked whether they are page-locked
inline bool map_count_only()  { return flag("OCT_PHMM_MAP_COUNT_ONLY"); }     // k-mer mapper without the exact shortcut
inline bool big_mapper()      { return flag("OCT_PHMM_BIG_MAPPER"); }         // test hook: the long-haplotype mapper on short haplotypes
inline bool window_lds()      { const char* e = get("OCT_PHMM_WINDOW_LDS"); return !e || atoi(e) != 0; }      // 0: canonical windows through per-region hash tables in global memory (k_window_insert x 2 + k_window_candidate) instead of k_window_region (A/B, tests)
inline bool map_mismatches()  { const char* e = get("OCT_PHMM_MAP_MISMATCHES"); return !e || atoi(e) != 0; }   // 0: k_classify compares the bases of every candidate itself (A/B, tests)
inline int  penalties_where() { const char* e = get("OCT_PHMM_PENALTIES"); return !e ? 0 : (e[0] == 'd' || e[0] == 'l' ? 2 : 1); }   // 0 by size, 1 host threads, 2 device
inline int  dedup()           { const char* e = get("OCT_PHMM_DEDUP"); return !e ? -1 : atoi(e); }                                  // -1 by shape, 0 never, 1 wherever it is possible
inline uint32_t dedup_hash_mask() { long long n; return number("OCT_PHMM_DEDUP_HASH_BITS", &n) && n >= 1 && n < 32 ? (1u << n) - 1u : 0xffffffffu; }   // test hook: collisions
inline int  device_sized()    { const char* e = get("OCT_PHMM_DEVICE_SIZED"); return !e ? -1 : atoi(e); }                           // -1 by shape, 0 never (host-sized launches: the mid-step read-back), 1 wherever possible
inline bool trace_per_pair(long long* v) { return number("OCT_PHMM_DSL_TRACE_PER_PAIR", v); }                                      // test hook: traceback tasks per pair the device-sized path provisions scratch for (-1: one task group, so that every batch overflows and is repeated host-sized)
inline bool late_start()      { const char* e = get("OCT_PHMM_LATE_START"); return !e || atoi(e) != 0; }                          // 0: every traceback task writes all of its backpointer tiles (A/B)
inline int  join_late()       { const char* e = get("OCT_PHMM_JOIN_LATE"); return !e ? -1 : atoi(e); }                                // a flavour's traceback and late-start lists in one DP launch and one walk: -1 one-slice batches only, 0 never, 1 always
inline bool dp_rows()         { const char* e = get("OCT_PHMM_DP_ROWS"); return !e || atoi(e) != 0; }                             // 0: long reads at band 16 with int32 lanes keep k_dp_wide (generic cost for every task, operands per lane) instead of k_dp_rows
inline bool multi_wave()      { const char* e = get("OCT_PHMM_MULTI_WAVE"); return !e || atoi(e) != 0; }                          // 0: bands 128 / 256 with int32 lanes keep one wave per task (k_dp_wide) instead of k_dp_mw
inline int  mw_planes()       { const char* e = get("OCT_PHMM_MW_PLANES"); return !e ? -1 : atoi(e); }                              // k_dp_mw: -1 by task count, 0 one plane per wave (B / 64 waves per task), 1 all planes in one wave
inline bool host_mapped()     { const char* e = get("OCT_PHMM_HOST_MAPPED"); return !e || atoi(e) != 0; }                                    // region-sized one-shot calls: inputs read and results written through mapped pinned host memory by kernels (0: DMA copies)
inline int  dsl_merge_dp()    { const char* e = get("OCT_PHMM_DSL_MERGE_DP"); return !e ? -1 : atoi(e); }                                         // device-sized step: traceback and score-only list of a flavour in one launch (k_dp_pair): -1 by batch size, 0 never (two launches on two streams), 1 always
inline int  walk_stage()      { const char* e = get("OCT_PHMM_WALK_STAGE"); return !e ? -1 : atoi(e); }                             // -1 by launch size; 0 lockstep walker out of registers, 1 lockstep out of LDS-staged tiles, 2 one walk per 16-lane row (k_walk_rows; k_walk_long at bands 128 / 256 for 1 and 2)
inline bool penalties_report() { return getenv("OCT_PHMM_PENALTIES_REPORT") != nullptr; }                                       // one stderr line per device generation
inline bool penalties_lane_kernel() { const char* e = get("OCT_PHMM_PENALTIES"); return e && e[0] == 'l'; }               // "lanes": one lane per haplotype even where a wave's LDS would do
}

namespace {

int fail(oct_phmm_status* st, int code, const char* msg)
{
    if (st) {
        memset(st, 0, sizeof(*st));
        st->code = code;
        if (code == OCT_PHMM_EHIP) st->hip_error = rt::last_error_code;
        if (msg) snprintf(st->message, sizeof(st->message), "%s", msg);
    }
    return code;
}
int ok(oct_phmm_status* st) { if (st) { memset(st, 0, sizeof(*st)); } return OCT_PHMM_OK; }

int band_for(int max_indel_error)   // simd_pair_hmm_wrapper.hpp:219-241
{
    for (int b = 8; b <= 256; b *= 2) if (max_indel_error <= b) return b;
    return -1;
}

#define RT(expr) do { if (!(expr)) return fail(status, OCT_PHMM_EHIP, #expr); } while (0)

// All device memory of a batch is ONE pool block: `upload` / `dalloc` only record what is needed, `commit` allocates, fills in the
// pointers and sends every input array up in a single copy out of the handle's pinned staging buffer (each array keeps a zeroed
// 16-byte tail pad, as the kernels' vector loads expect).
// Run f(lo, hi) over [0, n) on a few host threads (memory-bound passes over a big batch's arrays); small n stays on the caller's thread.
template <class F> void host_parallel(size_t n, size_t grain, F&& f)
{
    static const unsigned kCores = std::thread::hardware_concurrency();     // (asked once: glibc reads /sys for it, ~15 us per call - five calls were a third of a region call's host time)
    unsigned T = kCores > 4 ? 4 : (kCores ? kCores : 1);
    if (n / grain < T) T = (unsigned)(n / grain);
    if (T <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th; th.reserve(T - 1);
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&f, n, t, T] { f(n * t / T, n * (t + 1) / T); });
    f((size_t)0, n / T);
    for (auto& x : th) x.join();
}

struct Packer {
    struct Item { const void* src; size_t bytes; size_t off; void** dst; };
    std::vector<Item> items;
    size_t in_bytes = 0, total = 0;
    static size_t aligned(size_t n) { return (n + 16 + 255) & ~(size_t)255; }
    template <class T> void upload(const T* host, size_t n, const T** dev) { items.push_back({host, n * sizeof(T), 0, (void**)dev}); }
    template <class T> void dalloc(T** dev, size_t n) { items.push_back({nullptr, n * sizeof(T), 0, (void**)dev}); }
    bool commit(oct_phmm_handle* h, oct_phmm_batch* b, rt::Stream s);
};
// Inputs up to this size are packed into the pinned staging buffer and copied in one piece; larger ones stream through its two halves.
// OCT_PHMM_STAGE_MAX_KB: test hook (small batches through the streaming path).
static size_t stage_max()
{
    long long kb; if (tune::number("OCT_PHMM_STAGE_MAX_KB", &kb) && kb >= 2) return (size_t)kb << 10;
    return (size_t)64 << 20;
}

constexpr size_t kHostMappedCopyMax = (size_t)1 << 20;     // inputs up to here go up through k_copy_from_host, results of up to kHostMappedOutMax values (one region's) come back through the epilogue's own stores
constexpr uint64_t kHostMappedOutMax = 12288;             // (the epilogue's stores over the host link: 4 us for one region's 58 KB, 30 us for four regions', 82 for eight - a DMA copy wins from two regions on)
bool Packer::commit(oct_phmm_handle* h, oct_phmm_batch* b, rt::Stream s)
{
    std::stable_partition(items.begin(), items.end(), [](const Item& it) { return it.src != nullptr; });   // inputs first, contiguous
    total = 0;
    for (auto& it : items) { it.off = total; total += aligned(it.bytes); if (it.src) in_bytes = total; }
    void* base = nullptr;
    if (!h->pool.alloc(&base, total)) return false;
    b->allocs.push_back(base);
    for (auto& it : items) *it.dst = (char*)base + it.off;
    if (!in_bytes) return true;
    const size_t kStageMax = stage_max();
    if (in_bytes <= kStageMax) {
        if (h->stage_bytes < in_bytes) {
            rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
            size_t want = (size_t)1 << 20; while (want < in_bytes) want <<= 1;
            if (!rt::host_pinned_malloc(&h->stage, want)) return false;
            h->stage_bytes = want;
        }
        // the image of the input arrays in the pinned buffer, copied by a few host threads once it is worth their start-up (one thread moves
        // ~10 GB/s: the 30 MB of a 100k x 128 batch took 3 ms of the call on one thread)
        size_t n_in = 0; while (n_in < items.size() && items[n_in].src) ++n_in;
        host_parallel(in_bytes, (size_t)2 << 20, [&](size_t lo, size_t hi) {
            for (size_t i = 0; i < n_in; ++i) {
                const Item& it = items[i];
                const size_t slot_end = it.off + aligned(it.bytes), a = std::max(lo, it.off), z = std::min(hi, slot_end);
                if (a >= z) continue;
                const size_t data_end = it.off + it.bytes;
                if (a < data_end) memcpy((char*)h->stage + a, (const char*)it.src + (a - it.off), std::min(z, data_end) - a);
                if (z > data_end) { const size_t p0 = std::max(a, data_end); memset((char*)h->stage + p0, 0, z - p0); }
            }
        });
        if (in_bytes <= kHostMappedCopyMax && tune::host_mapped()) {                                // region-sized: a copy kernel reads the pinned image itself
            const uint32_t n16 = (uint32_t)((in_bytes + 15) / 16);
            OCT_LAUNCH(k_copy_from_host, (n16 + 255) / 256, 256, 0, s, (uint4*)base, (const uint4*)h->stage, n16);
            return rt::launch_ok();
        }
        return rt::h2d(base, h->stage, in_bytes, s);
    }
    size_t n_in = 0; while (n_in < items.size() && items[n_in].src) ++n_in;
    {   // Big batch with arrays in page-locked caller memory (oct_phmm_host_alloc, hipHostMalloc, hipHostRegister): the DMA engine reads those arrays themselves; the
        // others (the library's own small tables, pageable caller arrays) go through the staging halves one by one
        std::vector<char> direct(n_in, 0); bool any = false;
        for (size_t i = 0; i < n_in; ++i) if (items[i].bytes >= tune::pinned_min_bytes((size_t)1 << 20) && rt::host_is_pinned(items[i].src, items[i].bytes)) { direct[i] = 1; any = true; }
        if (any) {
            if (h->stage_bytes < kStageMax) {
                rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
                if (!rt::host_pinned_malloc(&h->stage, kStageMax)) return false;
                h->stage_bytes = kStageMax;
            }
            const size_t half = (kStageMax / 2) & ~(size_t)255;
            rt::Event ev[2] {}; bool used[2] = {false, false};
            if (!h->get_event(&ev[0]) || !h->get_event(&ev[1])) return false;
            bool ok = true; int k = 0;
            for (size_t i = 0; i < n_in && ok; ++i) {
                const Item& it = items[i];
                const size_t slot = aligned(it.bytes);
                if (direct[i]) {
                    ok = rt::dev_memset((char*)base + it.off + it.bytes, 0, slot - it.bytes, s) && rt::h2d((char*)base + it.off, it.src, it.bytes, s);   // (kernels read up to 16 bytes past an array)
                    continue;
                }
                for (size_t pos = 0; pos < slot && ok; pos += half, k ^= 1) {
                    const size_t len = slot - pos < half ? slot - pos : half;
                    char* buf = (char*)h->stage + (size_t)k * half;
                    if (used[k]) ok = rt::event_sync(ev[k]);
                    host_parallel(len, (size_t)4 << 20, [&](size_t lo, size_t hi) {
                        const size_t a = pos + lo, z = pos + hi;                  // bytes [a, z) of the slot: payload, then zero padding
                        if (a < it.bytes) memcpy(buf + lo, (const char*)it.src + a, (z < it.bytes ? z : it.bytes) - a);
                        if (z > it.bytes) { const size_t p0 = a > it.bytes ? a : it.bytes; memset(buf + (p0 - pos), 0, z - p0); }
                    });
                    ok = ok && rt::h2d((char*)base + it.off + pos, buf, len, s) && rt::event_record(ev[k], s);
                    used[k] = true;
                }
            }
            for (int i = 0; i < 2; ++i) { if (used[i]) ok = rt::event_sync(ev[i]) && ok; h->put_event(ev[i]); }
            return ok;
        }
    }
    // Big batch: the device image [0, in_bytes) goes through the two halves of the pinned staging buffer. While the DMA drains one half
    // a few host threads fill the other (one thread copies at ~10 GB/s, a pageable hipMemcpy no faster; PCIe takes ~50 GB/s).
    if (h->stage_bytes < kStageMax) {
        rt::host_pinned_free(h->stage); h->stage = nullptr; h->stage_bytes = 0;
        if (!rt::host_pinned_malloc(&h->stage, kStageMax)) return false;
        h->stage_bytes = kStageMax;
    }
    const size_t half = (kStageMax / 2) & ~(size_t)255;
    rt::Event ev[2] {}; bool used[2] = {false, false};
    if (!h->get_event(&ev[0]) || !h->get_event(&ev[1])) return false;
    auto fill = [&](char* dst, size_t lo, size_t hi) {       // image of device bytes [lo, hi): item payloads, zero padding between them
        size_t i = (size_t)(std::upper_bound(items.begin(), items.begin() + n_in, lo, [](size_t v, const Item& it) { return v < it.off; }) - items.begin());
        i = i ? i - 1 : 0;
        for (size_t pos = lo; pos < hi; ) {
            const Item& it = items[i];
            const size_t end = i + 1 < n_in ? items[i + 1].off : in_bytes;      // this item's slot (payload + padding)
            const size_t stop = end < hi ? end : hi;
            if (pos < it.off + it.bytes) {
                const size_t n = (it.off + it.bytes < stop ? it.off + it.bytes : stop) - pos;
                memcpy(dst + (pos - lo), (const char*)it.src + (pos - it.off), n);
                pos += n;
            }
            if (pos < stop) { memset(dst + (pos - lo), 0, stop - pos); pos = stop; }
            if (pos >= end) ++i;
        }
    };
    bool ok = true; int k = 0;
    for (size_t lo = 0; lo < in_bytes && ok; lo += half, k ^= 1) {
        const size_t len = in_bytes - lo < half ? in_bytes - lo : half;
        char* buf = (char*)h->stage + (size_t)k * half;
        if (used[k]) ok = rt::event_sync(ev[k]);
        host_parallel(len, kStageMax >= ((size_t)32 << 20) ? (size_t)4 << 20 : 256, [&](size_t a, size_t z) { fill(buf + a, lo + a, lo + z); });
        ok = ok && rt::h2d((char*)base + lo, buf, len, s) && rt::event_record(ev[k], s);
        used[k] = true;
    }
    for (int i = 0; i < 2; ++i) { if (used[i]) ok = rt::event_sync(ev[i]) && ok; h->put_event(ev[i]); }   // the staging buffer is the handle's: drained before anyone reuses it
    return ok;
}

// Byte-set questions over the input arrays, eight bytes per step (the compiler left the byte loops scalar: 0.17 ms of a 16-region upload, the only thing that made a
// device-sized batch of 150 k pairs slower than a host-sized one). high bit of every byte of the result: clear where the byte of x equals c.
inline uint64_t swar_ne(uint64_t x, uint8_t c) { const uint64_t y = x ^ (0x0101010101010101ull * c); return ((y & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | y; }
bool any_byte_outside_acgt(const uint8_t* p, size_t n)
{
    uint64_t bad = 0; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t x; memcpy(&x, p + i, 8); bad |= swar_ne(x, 'A') & swar_ne(x, 'C') & swar_ne(x, 'G') & swar_ne(x, 'T'); }
    uint32_t tail = 0;
    for (; i < n; ++i) tail |= ((p[i] == 'A') | (p[i] == 'C') | (p[i] == 'G') | (p[i] == 'T')) ? 0u : 1u;
    return (bad & 0x8080808080808080ull) != 0 || tail != 0;
}
bool any_byte_equals(const uint8_t* p, size_t n, uint8_t c)
{
    uint64_t all_ne = ~0ull; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t x; memcpy(&x, p + i, 8); all_ne &= swar_ne(x, c); }
    uint32_t tail = 0;
    for (; i < n; ++i) tail |= p[i] == c ? 1u : 0u;
    return (~all_ne & 0x8080808080808080ull) != 0 || tail != 0;
}

// What an upload must know about EVERY byte of its input before it packs it: the contract's range checks (quality <= 127, penalties >= 0, no empty read), the bounds the
// FASTADD decision needs (largest per-read quality sum, largest gap penalties) and - for device-sized batches - whether any base is outside ACGT / any SNV mask byte '0'.
// upload_impl makes them itself (one pass per array, threaded from ~2 MB on); the region server's CALLERS make them for their own region before they queue - 64 threads that would
// otherwise sleep - and a device batch inherits the merge (facts_of_reads / facts_of_haps are what both run).
struct InputFacts {
    uint32_t q_or = 0, pen_or = 0, gomax = 0, gemax = 0, t_min = 0xffffffffu; uint64_t sum_q_max = 0;
    int dirty = -1;                                       // -1 not looked at, 0 every base ACGT and every SNV mask byte set, 1 not so
    bool have_haps = false;                               // the penalty vectors were looked at (false: the library makes them)
    void merge(const InputFacts& o)
    {
        q_or |= o.q_or; pen_or |= o.pen_or; gomax = std::max(gomax, o.gomax); gemax = std::max(gemax, o.gemax); t_min = std::min(t_min, o.t_min); sum_q_max = std::max(sum_q_max, o.sum_q_max);
        dirty = (dirty < 0 || o.dirty < 0) ? -1 : (dirty | o.dirty);
    }
};
void facts_of_reads(const oct_phmm_reads* R, size_t r0, size_t r1, bool want_dirty, InputFacts* f)
{
    uint32_t v = 0, shortest = 0xffffffffu; uint64_t best = 0;
    for (size_t r = r0; r < r1; ++r) {
        const uint8_t* q = R->qualities + R->offsets[r]; const uint32_t n = R->offsets[r + 1] - R->offsets[r];
        uint32_t sq = 0, o = 0;                           // reads are < 32,768 bases of quality <= 127
        for (uint32_t i = 0; i < n; ++i) { sq += q[i]; o |= q[i]; }
        v |= o; best = std::max<uint64_t>(best, sq); shortest = std::min(shortest, n);
    }
    f->q_or |= v; f->sum_q_max = std::max(f->sum_q_max, best); f->t_min = std::min(f->t_min, shortest);
    if (want_dirty && r1 > r0 && any_byte_outside_acgt((const uint8_t*)R->bases + R->offsets[r0], (size_t)R->offsets[r1] - R->offsets[r0])) f->dirty = 1;
}
void facts_of_haps(const oct_phmm_haplotypes* H, size_t lo, size_t hi, bool want_dirty, InputFacts* f)      // bases [lo, hi) of the concatenated haplotypes, vectors given
{
    uint32_t v = 0, a = 0, e = 0;
    for (size_t i = lo; i < hi; ++i) {
        const uint32_t go = (uint8_t)H->gap_open[i], ge = (uint8_t)H->gap_extend[i];
        v |= go | ge | (uint8_t)H->snv_prior_fwd[i] | (uint8_t)H->snv_prior_rev[i];
        a = std::max(a, go); e = std::max(e, ge);           // (as bytes: with no sign bit anywhere - checked by the caller - these are the values)
    }
    f->pen_or |= v; f->gomax = std::max(f->gomax, a); f->gemax = std::max(f->gemax, e);
    if (want_dirty && (any_byte_outside_acgt((const uint8_t*)H->bases + lo, hi - lo) || any_byte_equals((const uint8_t*)H->snv_mask_fwd + lo, hi - lo, '0')
                       || any_byte_equals((const uint8_t*)H->snv_mask_rev + lo, hi - lo, '0'))) f->dirty = 1;
}

bool monotone(const uint32_t* off, uint32_t n) { for (uint32_t i = 0; i < n; ++i) if (off[i + 1] < off[i]) return false; return true; }

// kernel dispatch over (band, traceback, generic bytes, 32-bit adds)
template <int B, bool TR, bool GEN, bool FA>
bool launch_dp_inst(const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp<B, TR, GEN, FA>), lds)) return false;    // up to 64 KB needs no opt-in (and the call is a driver round trip)
    OCT_LAUNCH((k_dp<B, TR, GEN, FA>), n_blocks, kBlockWaves * 64, lds, s, p);
    return rt::launch_ok();
}
template <int B, bool FA>
bool launch_dp_band(bool tr, bool gen, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (tr) return gen ? launch_dp_inst<B, true, true, FA>(p, n_blocks, lds, s) : launch_dp_inst<B, true, false, FA>(p, n_blocks, lds, s);
    return gen ? launch_dp_inst<B, false, true, FA>(p, n_blocks, lds, s) : launch_dp_inst<B, false, false, FA>(p, n_blocks, lds, s);
}
bool launch_dp(int band, bool tr, bool gen, bool fa, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return fa ? launch_dp_band<8, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<8, false>(tr, gen, p, n_blocks, lds, s);
        case 16: return fa ? launch_dp_band<16, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<16, false>(tr, gen, p, n_blocks, lds, s);
        case 32: return fa ? launch_dp_band<32, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<32, false>(tr, gen, p, n_blocks, lds, s);
        case 64: return fa ? launch_dp_band<64, true>(tr, gen, p, n_blocks, lds, s) : launch_dp_band<64, false>(tr, gen, p, n_blocks, lds, s);
        default: return false;
    }
}
// the traceback list and the score-only list of one flavour in one launch (device-sized steps)
template <int B, bool GEN, bool FA>
bool launch_dp_pair_inst(const DpParams& pt, const DpParams& ps, uint32_t n_blocks_t, uint32_t n_blocks_s, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp_pair<B, GEN, FA>), lds)) return false;
    OCT_LAUNCH((k_dp_pair<B, GEN, FA>), n_blocks_t + n_blocks_s, kBlockWaves * 64, lds, s, pt, ps, n_blocks_t);
    return rt::launch_ok();
}
template <int B>
bool launch_dp_pair_band(bool gen, bool fa, const DpParams& pt, const DpParams& ps, uint32_t nt, uint32_t ns, size_t lds, rt::Stream s)
{
    if (gen) return fa ? launch_dp_pair_inst<B, true, true>(pt, ps, nt, ns, lds, s) : launch_dp_pair_inst<B, true, false>(pt, ps, nt, ns, lds, s);
    return fa ? launch_dp_pair_inst<B, false, true>(pt, ps, nt, ns, lds, s) : launch_dp_pair_inst<B, false, false>(pt, ps, nt, ns, lds, s);
}
bool launch_dp_pair(int band, bool gen, bool fa, const DpParams& pt, const DpParams& ps, uint32_t nt, uint32_t ns, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return launch_dp_pair_band<8>(gen, fa, pt, ps, nt, ns, lds, s);
        case 16: return launch_dp_pair_band<16>(gen, fa, pt, ps, nt, ns, lds, s);
        case 32: return launch_dp_pair_band<32>(gen, fa, pt, ps, nt, ns, lds, s);
        case 64: return launch_dp_pair_band<64>(gen, fa, pt, ps, nt, ns, lds, s);
        default: return false;
    }
}
template <int B, bool TR>
bool launch_dp32_inst(const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    if (lds > 64 * 1024 && !rt::allow_lds((k_dp32<B, TR>), lds)) return false;
    OCT_LAUNCH((k_dp32<B, TR>), n_blocks, kBlockWaves * 64, lds, s, p);
    return rt::launch_ok();
}
bool launch_dp32(int band, bool tr, const DpParams& p, uint32_t n_blocks, size_t lds, rt::Stream s)
{
    switch (band) {
        case 8:  return tr ? launch_dp32_inst<8, true>(p, n_blocks, lds, s) : launch_dp32_inst<8, false>(p, n_blocks, lds, s);
        case 16: return tr ? launch_dp32_inst<16, true>(p, n_blocks, lds, s) : launch_dp32_inst<16, false>(p, n_blocks, lds, s);
        case 32: return tr ? launch_dp32_inst<32, true>(p, n_blocks, lds, s) : launch_dp32_inst<32, false>(p, n_blocks, lds, s);
        case 64: return tr ? launch_dp32_inst<64, true>(p, n_blocks, lds, s) : launch_dp32_inst<64, false>(p, n_blocks, lds, s);
        default: return false;
    }
}
template <int B, int TPR, int C>
bool launch_walk_inst(const WalkParams& w, rt::Stream s, int stage)     // stage: 0 lockstep walker out of registers, 1 lockstep out of LDS-staged tiles, 2 one walk per 16-lane row
{
    const uint32_t blocks = (w.n_tasks + 255) / 256;
    const size_t lds = 256 * kWalkEvents * sizeof(uint32_t);
    if constexpr (C == 1) {
        const size_t stage_lds = walk_stage_lds_bytes(B, TPR);
        if (stage == 2) { const uint32_t th = 64; OCT_LAUNCH((k_walk_rows<B, TPR>), (w.n_tasks + th / 16 - 1) / (th / 16), th, walk_rows_lds_bytes(B, th), s, w); }   // region-sized launch: four walks per wave, runs of matches in one move
        else if (stage && stage_lds <= rt::kMaxLdsBytes) {          // one wave per workgroup, the tiles staged in LDS
            if (stage_lds > 64 * 1024 && !rt::allow_lds((k_walk<B, TPR, C, true>), stage_lds)) return false;
            OCT_LAUNCH((k_walk<B, TPR, C, true>), (w.n_tasks + 63) / 64, 64, stage_lds, s, w);
        } else OCT_LAUNCH((k_walk<B, TPR, C, false>), blocks, 256, lds, s, w);
    } else {
        // bands 128 / 256: one wave walks one task out of LDS-staged lines (k_walk_long) unless the launch is big enough for the lockstep walker to fill its waves
        if (stage) OCT_LAUNCH((k_walk_long<B, C>), w.n_tasks, 64, walk_long_lds_bytes(), s, w);
        else OCT_LAUNCH((k_walk<B, TPR, C>), blocks, 256, lds, s, w);
    }
    if (w.out_align1 != nullptr) OCT_LAUNCH((k_walk_strings<B, TPR, C>), blocks, 256, 0, s, w);   // test seam: gapped strings from the simple per-step walker
    if (w.pair_key != nullptr) OCT_LAUNCH((k_walk_cigar<B, TPR, C>), blocks, 256, 0, s, w);       // align mode: the pairs' winning tasks write their CIGARs
    return rt::launch_ok();
}
bool launch_walk(int band, bool one_per_row, const WalkParams& w, rt::Stream s, int stage)
{
    switch (band) {
        case 8:   return one_per_row ? launch_walk_inst<8, 1, 1>(w, s, stage) : launch_walk_inst<8, 2, 1>(w, s, stage);
        case 16:  return one_per_row ? launch_walk_inst<16, 1, 1>(w, s, stage) : launch_walk_inst<16, 2, 1>(w, s, stage);
        case 32:  return one_per_row ? launch_walk_inst<32, 1, 1>(w, s, stage) : launch_walk_inst<32, 2, 1>(w, s, stage);
        case 64:  return one_per_row ? launch_walk_inst<64, 1, 1>(w, s, stage) : launch_walk_inst<64, 2, 1>(w, s, stage);
        case 128: return launch_walk_inst<128, 1, 2>(w, s, stage);
        case 256: return launch_walk_inst<256, 1, 4>(w, s, stage);
        default: return false;
    }
}
template <int B, bool TR>
bool launch_dp_wide_inst(bool w16, const DpParams& p, rt::Stream s)
{
    constexpr uint32_t ROWS = B < 64 ? 64 / B : 1;                       // tasks per wave
    const uint32_t waves = (p.n_tasks + ROWS - 1) / ROWS, blocks = (waves + kBlockWaves - 1) / kBlockWaves;
    if (w16) OCT_LAUNCH((k_dp_wide<B, TR, true>), blocks, kBlockWaves * 64, 0, s, p);
    else     OCT_LAUNCH((k_dp_wide<B, TR, false>), blocks, kBlockWaves * 64, 0, s, p);
    return rt::launch_ok();
}
bool launch_dp_rows(bool tr, bool gen, const DpParams& p, rt::Stream s)      // long reads at band 16, int32 lanes: one task per row of 16 lanes (k_dp_rows)
{
    const uint32_t waves = (p.n_tasks + 3) / 4, blocks = (waves + kBlockWaves - 1) / kBlockWaves;
    if (tr) { if (gen) OCT_LAUNCH((k_dp_rows<true, true>), blocks, kBlockWaves * 64, 0, s, p); else OCT_LAUNCH((k_dp_rows<true, false>), blocks, kBlockWaves * 64, 0, s, p); }
    else    { if (gen) OCT_LAUNCH((k_dp_rows<false, true>), blocks, kBlockWaves * 64, 0, s, p); else OCT_LAUNCH((k_dp_rows<false, false>), blocks, kBlockWaves * 64, 0, s, p); }
    return rt::launch_ok();
}
bool launch_dp_wide(int band, bool tr, bool w16, const DpParams& p, rt::Stream s)
{
    switch (band) {
        case 8:   return tr ? launch_dp_wide_inst<8, true>(w16, p, s) : launch_dp_wide_inst<8, false>(w16, p, s);
        case 16:  return tr ? launch_dp_wide_inst<16, true>(w16, p, s) : launch_dp_wide_inst<16, false>(w16, p, s);
        case 32:  return tr ? launch_dp_wide_inst<32, true>(w16, p, s) : launch_dp_wide_inst<32, false>(w16, p, s);
        case 64:  return tr ? launch_dp_wide_inst<64, true>(w16, p, s) : launch_dp_wide_inst<64, false>(w16, p, s);
        case 128: return tr ? launch_dp_wide_inst<128, true>(w16, p, s) : launch_dp_wide_inst<128, false>(w16, p, s);
        case 256: return tr ? launch_dp_wide_inst<256, true>(w16, p, s) : launch_dp_wide_inst<256, false>(w16, p, s);
        default: return false;
    }
}

template <int B, int PL>
bool launch_dp_mw_band(bool tr, bool gen, const DpParams& p, rt::Stream s)
{
    const uint32_t blocks = p.n_tasks, threads = B / PL;                 // one task per workgroup of B / (64 PL) waves
    if (tr) { if (gen) OCT_LAUNCH((k_dp_mw<B, PL, true, true>), blocks, threads, 0, s, p); else OCT_LAUNCH((k_dp_mw<B, PL, true, false>), blocks, threads, 0, s, p); }
    else    { if (gen) OCT_LAUNCH((k_dp_mw<B, PL, false, true>), blocks, threads, 0, s, p); else OCT_LAUNCH((k_dp_mw<B, PL, false, false>), blocks, threads, 0, s, p); }
    return rt::launch_ok();
}
// one_wave: all planes of a task in one wave (a launch with a task for every SIMD of the chip), else one plane per wave (few tasks: spread them out)
bool launch_dp_mw(int band, bool one_wave, bool tr, bool gen, const DpParams& p, rt::Stream s)
{
    if (band == 128) return one_wave ? launch_dp_mw_band<128, 2>(tr, gen, p, s) : launch_dp_mw_band<128, 1>(tr, gen, p, s);
    if (band == 256) return one_wave ? launch_dp_mw_band<256, 4>(tr, gen, p, s) : launch_dp_mw_band<256, 1>(tr, gen, p, s);
    return false;
}

// Read records of the packed int16 kernels: whole reads in LDS while three workgroups then fit on a CU (150-base reads: 47 KB with traceback tiles, 30 KB without), else
// the largest chunk of iterations (a multiple of 32, at least 64) with which three do, else two - 500-base chunks of long reads against 1.8 kb haplotypes: 115 KB -> 75 KB
// with traceback (one wave per SIMD -> two), 98 -> 51 KB without (-> three). OCT_PHMM_REC_CHUNK=n forces a chunk (test hook: restaging on small reads; 0 = never).
// dense (round 5): a mid-size batch (a region server's device batch: a few rounds of workgroups, both DP forms on the chip at once) takes 64-iteration chunks where that lets a
// fourth traceback workgroup (46 -> 39 KB) and a sixth or seventh score-only one (30 -> 22 KB) onto a CU - the launch is bound by rounds x wave latency, not by issue slots:
// 16 regions of the configs[3] stream 1.054 -> 0.983 ms per populate; the 12.8 M-pair step does not care (traceback 8.60 -> 8.58 ms per launch, score-only 6.25 -> 6.48).
uint32_t dp_rec_chunk(uint32_t t_cap, uint32_t lh_cap, uint32_t B, bool trace, bool dense = false)
{
    long long v;
    if (tune::number("OCT_PHMM_REC_CHUNK", &v)) return v > 0 ? (uint32_t)((v + 3) & ~3ll) : 0u;
    if (dense && t_cap > 100 && dp_lds_bytes(t_cap, lh_cap, B, trace, 64) <= rt::kMaxLdsBytes / (trace ? 4 : 6) - 512
        && dp_lds_bytes(t_cap, lh_cap, B, trace) > rt::kMaxLdsBytes / (trace ? 4 : 6) - 512) return 64u;
    if (t_cap <= 128) return 0;
    for (size_t per_cu : {(size_t)3, (size_t)2}) {
        const size_t budget = rt::kMaxLdsBytes / per_cu - 1024;
        if (dp_lds_bytes(t_cap, lh_cap, B, trace) <= budget) return 0;
        for (uint32_t c = ((t_cap - 1) / 32) * 32; c >= 64; c -= 32) if (dp_lds_bytes(t_cap, lh_cap, B, trace, c) <= budget) return c;
    }
    return dp_lds_bytes(t_cap, lh_cap, B, trace) <= rt::kMaxLdsBytes ? 0u : 64u;      // (no chunk gives two per CU: whole reads if they fit at all)
}

bool ensure_bp(oct_phmm_handle* h, int slice, size_t bytes)
{
    if (h->bp_bytes[slice] >= bytes) return true;
    // grow by half at least (below 4 GB): a region thread's calls differ in size, and every regrowth is a hipFree + hipMalloc that stalls the device
    // (doubling: three workers of a region server each met their biggest batch late in a run of 8,000 calls, and every regrowth of a multi-gigabyte block took
    // 0.1 - 1 s of hipMalloc: profiles/r04_step3_server_api_trace.txt); never beyond the handle's budget
    const size_t old = h->bp_bytes[slice];
    size_t roomy = old < ((size_t)4 << 30) ? std::max(bytes, 2 * old) : bytes;
    if (roomy > h->bp_budget) roomy = std::max(bytes, h->bp_budget);
    rt::dev_free(h->bp[slice]); h->bp[slice] = nullptr; h->bp_bytes[slice] = 0;
    void* p = nullptr; size_t got = roomy;
    if (h->fail_bp_allocs > 0) { --h->fail_bp_allocs; return false; }      // test hook (OCT_PHMM_TEST_FAIL_BP_ALLOCS): the device "has no room": the caller halves its chunk
    if (!rt::dev_malloc(&p, roomy)) {
        rt::clear_error();
        h->pool.trim();                                 // cached blocks of earlier batches may be in the way
        got = bytes;
        if (!rt::dev_malloc(&p, bytes)) {
            rt::clear_error();
            DevPool::trim_device(h->pool.device);       // ... or the sibling handles' (a region server's, a caller's other threads')
            if (!rt::dev_malloc(&p, bytes)) { rt::clear_error(); return false; }
        }
    }
    h->bp[slice] = (uint32_t*)p; h->bp_bytes[slice] = got;
    return true;
}

// Run one kind's task list through the DP kernel (+ walk for traceback kinds), chunked so the traceback scratch fits.
// `ref` (device-sized launch): `tasks` is the array that holds all six lists, `n_tasks` the host's bound for one list; the kernels take the list itself
// from the totals in device memory.
// One workgroup scans ~10 us per tile of 8,192 pairs (41 us at four regions, 82 at eight: profiles/r03_step7_multi_region_timelines.txt); the four launches of the tiled scan
// cost ~20 us whatever the size
constexpr size_t   kPinnedOutMinBytes = (size_t)8 << 20;  // results from here on: is the caller's `out` page-locked? (the question costs microseconds: not asked for region-sized calls)
constexpr uint64_t kLaneMapMinPairs = 200000;          // k-mer mapper: one lane per pair from here on (k_kmer_map_lanes), one wave per pair below
constexpr uint64_t kDslMaxPairs = 400000;              // device-sized launches (no read-back inside the step, grids sized by the host's bound) up to here. Round 4 stopped at 100 k: first 6 / 8 / 12 / 16 / 64
                                                       // regions of the configs[3] stream (50 k / 70 k / 110 k / 150 k / 660 k pairs), one populate from host buffers: 0.54 / 0.60 / 1.04 / 1.31 / 4.14 ms
                                                       // device-sized, 0.57 / 0.57 / 0.97 / 1.21 / 3.56 host-sized. Round 5 (gpurun_out/r05_s01): on the DEVICE the two forms take the same time (16 regions:
                                                       // run + wait 0.643 against 0.650 ms, 64 regions 2.21 against 2.24) - the difference was the upload's byte-by-byte scan for the cost flavours
                                                       // (0.17 / 0.29 ms), now eight bytes per step (any_byte_outside_acgt). What the device-sized form buys a caller who has other work - the region
                                                       // server's workers - is that oct_phmm_batch_run never waits.
constexpr uint64_t kWalkRowsMaxPairs = 49152;          // traceback walks of batches up to here: one walk per 16-lane row (k_walk_rows)
constexpr uint64_t kDslMergeMaxPairs = 12000;          // device-sized step: up to here the traceback and the score-only list of a flavour share one launch (k_dp_pair)
constexpr uint32_t kDslMaxBlocks = 1024;               // grid of a device-sized DP launch: the bound, at most this (workgroups stride over the groups)
int run_dp_kind(oct_phmm_handle* h, oct_phmm_batch* b, int slice, int kind, const DevTask* tasks, uint32_t n_tasks, TraceEnd* ends,
                int nuc_prior, const WalkParams* seam_walk, oct_phmm_status* status, const rt::Stream* on_stream = nullptr, bool late = false,
                TaskListRef ref = TaskListRef {nullptr, nullptr, 0, nullptr}, const rt::Event* after_first_dp = nullptr,
                int paired_score_list = -1, uint32_t paired_score_bound = 0,     // device-sized traceback launch: the score-only list of the same flavour rides along (k_dp_pair)
                uint32_t joined_late_from = 0xffffffffu,                         // host-sized traceback launch: the flavour's late-start list lies behind the list proper (this many tasks) and is part of `n_tasks`
                const uint32_t* paired_end = nullptr)                            // `tasks` went through k_pair_sort: per haplotype, the list index up to which tasks 2i and 2i + 1 share a window
{
    if (!n_tasks) return OCT_PHMM_OK;
    const bool dsl = ref.totals != nullptr;
    rt::Stream st = on_stream ? *on_stream : h->slice_stream(slice);
    const int B = h->band;
    const uint32_t C = (uint32_t)h->lanes_c;
    const uint32_t G = b->stream ? (B < 64 ? 64u / (uint32_t)B : 1u) : (h->wide ? 1 : 2) * (64 / B);
    const bool tr = kind == kTraceFast || kind == kTraceGen, gen = kind == kScoreGen || kind == kTraceGen;
    const bool dense = b->n_pairs > 20000 && b->n_pairs <= kDslMaxPairs;        // (a region-sized call is one round of workgroups: nothing to gain from a restage every 64 iterations)
    const uint32_t rec_chunk = (b->stream || h->wide) ? 0u : dp_rec_chunk(b->t_cap, b->lh_cap, (uint32_t)B, tr, dense);
    const size_t lds = b->stream ? 0 : dp_lds_bytes(b->t_cap, b->lh_cap, (uint32_t)B, tr, rec_chunk, paired_end != nullptr);
    if (lds > rt::kMaxLdsBytes) return fail(status, OCT_PHMM_EUNSUPPORTED, "read/haplotype too long for the LDS-resident DP kernel");
    DpParams p {};
    p.rec_chunk = rec_chunk;
    p.ref = ref;
    p.rbases = b->d.rbases; p.rquals = b->d.rquals; p.roff = b->d.roff; p.rrev = b->d.rrev; p.hoff = b->d.hoff;
    p.rrec = b->d.rrec; p.rrec_stride = b->d.rrec_stride; p.rrecW = b->d.rrecW;
    p.tabF = gen ? b->d.tabGenF : b->d.tabFastF; p.tabR = gen ? b->d.tabGenR : b->d.tabFastR;
    p.pair_best = b->d.pair_best;
    p.k_cap = bp_tiles(b->t_cap, (uint32_t)B); p.t_cap = b->t_cap; p.lh_cap = b->lh_cap;
    const uint32_t n4 = ((uint32_t)(int8_t)nuc_prior << 2) & 0xffffu;       // vectorise_left_shift_bits(int8_t), simd_pair_hmm.hpp:74-78,257
    p.nuc4 = n4 | n4 << 16;
    const uint32_t n_groups = n_tasks / G;
    // workgroups walk kGroupsPerWave groups per wave to amortise the haplotype-table staging; a small launch (one active region) instead
    // spreads over the chip: one group per wave until there are enough workgroups for every CU
    p.groups_per_block = kBlockWaves * std::max<uint32_t>(1, std::min<uint32_t>(kGroupsPerWave, n_groups / (kBlockWaves * 2048)));
    if (dsl) p.groups_per_block = kBlockWaves;             // (region-sized by construction)
    // late traceback start is PERMITTED wherever the walk may stop early (below); which task groups take it is geometry (dp_groups). `late`: the launch is a late-start list.
    // (a launch of a traceback list proper, beside late-start lists of its own, has no such group by construction: p.late stays 0 and its groups skip the question)
    const bool may_start_late = tr && !seam_walk && !b->align_mode && b->fast_adds && !b->stream && !h->wide && tune::late_start();
    p.late = (may_start_late && (late || ref.join_late || joined_late_from != 0xffffffffu)) ? 1 : 0;
    p.late_from = late ? 0u : joined_late_from;
    p.hap_region = b->d.hap_region; p.reg_rhs = b->d.reg_rhs; p.reg_lhs = b->d.reg_lhs;
    p.paired_end = paired_end; p.task0 = 0;
    uint32_t chunk_groups = n_groups;
    if (tr) {
        const size_t per_group = (size_t)p.k_cap * 4096 * (b->stream ? C : 1);
        const size_t fit = dsl ? n_groups : std::max<size_t>(1, h->bp_budget / std::max<size_t>(1, b->slices.size()) / per_group);   // (device-sized: the bound was checked at upload)
        chunk_groups = (uint32_t)std::min<size_t>(n_groups, fit);
        if (chunk_groups < n_groups) chunk_groups = std::max<uint32_t>(p.groups_per_block, chunk_groups / p.groups_per_block * p.groups_per_block);   // several launches: whole workgroups each
        // the device may not have the budget free (other handles, other processes): fall back to smaller chunks of whole workgroups
        while (!ensure_bp(h, slice, (size_t)std::min(chunk_groups, n_groups) * per_group)) {
            if (chunk_groups <= p.groups_per_block || b->align_mode || dsl) return fail(status, OCT_PHMM_EHIP, "traceback scratch allocation");
            chunk_groups = std::max<uint32_t>(p.groups_per_block, chunk_groups / 2 / p.groups_per_block * p.groups_per_block);
        }
    }
    for (uint32_t g0 = 0; g0 < n_groups; g0 += chunk_groups) {
        const uint32_t ng = std::min(chunk_groups, n_groups - g0);
        p.tasks = tasks + (size_t)g0 * G; p.n_tasks = ng * G; p.task0 = g0 * G;
        if (!late && joined_late_from != 0xffffffffu) p.late_from = joined_late_from > g0 * G ? joined_late_from - g0 * G : 0u;      // (relative to this chunk's first task)
        p.bp = h->bp[slice]; p.ends = tr ? ends + (size_t)g0 * G : nullptr;
        uint32_t n_blocks = (ng + p.groups_per_block - 1) / p.groups_per_block;
        const long long dsl_blocks = kDslMaxBlocks;
        if (dsl) n_blocks = std::min<uint32_t>(n_blocks, (uint32_t)std::max<long long>(64, dsl_blocks));
        rt::Event e0 {}, e1 {};
        if (h->timing) { RT(h->get_event(&e0)); RT(h->get_event(&e1)); RT(rt::event_record(e0, st)); }
        // (a device-sized launch does not know its task count: region-sized, so the spread-out form)
        const bool one_wave = tune::mw_planes() >= 0 ? tune::mw_planes() != 0 : (!dsl && p.n_tasks >= 640);
        if (paired_score_list >= 0 && g0 == 0) {
            DpParams ps = p;                                   // same tables (same flavour), the score-only list of the same task array, no traceback scratch
            ps.ref.list = (uint32_t)paired_score_list; ps.tasks = tasks; ps.n_tasks = paired_score_bound / G * G; ps.bp = nullptr; ps.ends = nullptr; ps.late = 0;
            const uint32_t n_blocks_s = std::min<uint32_t>((ps.n_tasks / G + ps.groups_per_block - 1) / ps.groups_per_block, (uint32_t)std::max<long long>(64, dsl_blocks));
            if (!launch_dp_pair(B, gen, b->fast_adds, p, ps, n_blocks, n_blocks_s, lds, st)) return fail(status, OCT_PHMM_EHIP, "DP kernel launch");
        } else
        if (!(b->multi_wave ? launch_dp_mw(B, one_wave, tr, gen, p, st) : b->rows32 ? launch_dp_rows(tr, gen, p, st) : b->stream ? launch_dp_wide(B, tr, !h->wide, p, st)
                        : h->wide ? launch_dp32(B, tr, p, n_blocks, lds, st) : launch_dp(B, tr, gen, b->fast_adds, p, n_blocks, lds, st)))
            return fail(status, OCT_PHMM_EHIP, "DP kernel launch");
        if (h->timing) { RT(rt::event_record(e1, st)); b->timers.emplace_back(e0, e1); b->timer_kind.push_back(kind); }
        if (after_first_dp && g0 == 0) RT(rt::event_record(*after_first_dp, st));      // whoever waits for it runs beside this launch's walk, not beside its DP
        if (tr) {
            WalkParams w {};
            if (seam_walk) w = *seam_walk;
            w.ref = ref;
            w.tasks = p.tasks; w.n_tasks = p.n_tasks; w.ends = p.ends; w.bp = h->bp[slice]; w.k_cap = p.k_cap; w.band = B;
            w.rbases = b->d.rbases; w.rquals = b->d.rquals; w.roff = b->d.roff; w.rrev = b->d.rrev;
            w.hbases = b->d.hbases; w.hoff = b->d.hoff; w.go = b->d.go; w.ge = b->d.ge;
            w.maskF = b->d.maskF; w.priorF = b->d.priorF; w.maskR = b->d.maskR; w.priorR = b->d.priorR;
            w.hap_region = b->d.hap_region; w.reg_lhs = b->d.reg_lhs; w.reg_rhs = b->d.reg_rhs;
            w.nuc_prior = nuc_prior; w.pair_best = b->d.pair_best;
            w.early_stop = (!seam_walk && !b->align_mode && b->fast_adds && !b->stream && !h->wide) ? 1 : 0;
            if (seam_walk) {   // seam outputs are indexed by task: advance to this chunk
                const size_t o = (size_t)g0 * G;
                w.out_first_pos += o; w.out_align_off += o;
                if (w.seam_lhs) { w.seam_lhs += o; w.seam_rhs += o; w.out_flank += o; w.out_mask_size += o; }
            }
            if (b->align_mode) {
                if (ng != n_groups) return fail(status, OCT_PHMM_EUNSUPPORTED, "alignment batch too large for the traceback scratch (raise OCT_PHMM_BP_BUDGET_GB or split the batch)");
                w.pair_key = b->d.pair_key; w.task_key = b->slices[slice].d_keys; w.pos = b->d.pos; w.npos = b->d.npos; w.max_pos = b->d.max_pos;
                w.err_flags = b->d_err_flags; w.cig_ops = b->d_aln_ops; w.cig_n = b->d_aln_n; w.cig_mpos = b->d_aln_mpos; w.cig_cap = b->cig_cap;
            }
            // region-sized launches (a few hundred waves at most) stage their tiles in LDS; big ones hide the line fetches behind other waves
            // ... and a few regions' worth of walks (up to kWalkRowsMaxPairs pairs) get a 16-lane row each: measured per 300 x 24 call 20 against 60 us, and a 1k x 64 batch
            // 0.52 against 0.46 ms (profiles/r03_step7_small_batch_walkers_ab.log) - from there on the lockstep walker's 64 walks per wave win again
            const bool small = dsl || (size_t)p.n_tasks <= 64 * 1024;
            const int stage = tune::walk_stage() >= 0 ? tune::walk_stage() : (small ? (b->n_pairs <= kWalkRowsMaxPairs ? 2 : 1) : 0);
            if (!launch_walk(B, h->wide || b->stream, w, st, stage)) return fail(status, OCT_PHMM_EHIP, "walk kernel launch");
        }
    }
    return OCT_PHMM_OK;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// per-haplotype penalty vectors (phmm_error_model.hpp): host threads, or one device lane per haplotype
// ---------------------------------------------------------------------------------------------------------------
struct PenaltyOut { int8_t* go; int8_t* ge; uint8_t* mf; int8_t* pf; uint8_t* mr; int8_t* pr; };
constexpr size_t kPenaltyLdsBytes = 64 * 1024;     // LDS a wave of k_penalty_vectors_wave may take: the 160 KB of a CU then hold two haplotypes

OCT_KERNEL(k_penalty_vectors)(const oct_phmm_error_model* model, const uint8_t* hbases, const uint32_t* hoff, uint32_t hap0, uint32_t hap1,
                              const uint8_t* sub_mask, uint32_t* workspace, size_t words_per_hap, PenaltyOut out, uint32_t* overflow)
{
    const uint32_t h = hap0 + hw::block_idx() * hw::block_dim() + hw::thread_idx();
    if (h >= hap1) return;
    const uint32_t o = hoff[h], n = hoff[h + 1] - o;
    uint32_t* w = workspace + (size_t)(h - hap0) * words_per_hap;
    const int rc = em::penalty_vectors(*model, hbases + o, n, sub_mask ? sub_mask + o : nullptr, w, 1, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o);
    if (rc != em::kOk) overflow[h] = 1;
}

// One wave per haplotype, everything but the six output vectors in LDS: the haplotype's bases and the flat workspace of
// phmm_error_model.hpp at its tight sizing. The wave's lanes share the parallel phases; lane 0 runs the sequential ones at LDS latency.
OCT_KERNEL(k_penalty_vectors_wave)(const oct_phmm_error_model* model, const uint8_t* hbases, const uint32_t* hoff, uint32_t n_haps,
                                   const uint8_t* sub_mask, uint32_t lds_words, PenaltyOut out, uint32_t* overflow, unsigned long long* prof)
{
    OCT_DYN_SMEM(lds_raw);
    uint32_t* w = (uint32_t*)lds_raw;
    const uint32_t h = hw::block_idx();
    if (h >= n_haps) return;
    em::Wave x;
    x.prof = prof;
    x.tick(0);
    const uint32_t o = hoff[h], n = hoff[h + 1] - o;
    const size_t need = em::workspace_words(n, 0);
    uint8_t* s = (uint8_t*)(w + need);
    if (need + (n + 3) / 4 > lds_words) { if (x.lane() == 0) overflow[h] = 1; return; }
    for (uint32_t i = x.lane(); i < n; i += 64) s[i] = hbases[o + i];
    x.sync();
    const int rc = em::penalty_vectors(x, *model, s, n, sub_mask ? sub_mask + o : nullptr, w, 0, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o);
    if (rc != em::kOk && x.lane() == 0) overflow[h] = 1;
}

namespace {

// one haplotype on the calling thread; the workspace grows until the run lists fit (pathological repeat structure only)
void host_penalty_vectors_one(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* sub, std::vector<uint32_t>& w, PenaltyOut out, size_t o,
                              const em::CustomIndelModel* custom = nullptr)
{
    for (uint32_t grow = 1; ; grow *= 4) {
        const size_t need = em::workspace_words(n, grow);
        if (w.size() < need) w.resize(need);
        if (!custom) {
            if (em::penalty_vectors(m, s, n, sub, w.data(), grow, out.go + o, out.ge + o, out.mf + o, out.pf + o, out.mr + o, out.pr + o) == em::kOk) return;
        } else if (em::custom_indel_penalties(*custom, s, n, w.data(), grow, out.go + o, out.ge + o) == em::kOk) {     // the file's rows for the gaps, `m` for the SNV vectors only
            em::snv_priors(em::Seq {}, m, s, n, sub, w.data(), out.mf + o, out.pf + o, out.mr + o, out.pr + o);
            return;
        }
    }
}

void host_penalty_vectors(const oct_phmm_error_model& m, uint32_t n_haps, const uint8_t* bases, const uint32_t* off, const uint8_t* sub, PenaltyOut out,
                          const em::CustomIndelModel* custom = nullptr)
{
    static const unsigned kCores = std::thread::hardware_concurrency();     // (asked once, see host_parallel)
    unsigned T = kCores ? std::min(kCores, 16u) : 1;
    const uint32_t n_bases = n_haps ? off[n_haps] : 0;
    if (n_bases < 2000 || n_haps < 4) T = 1;                           // a thread start costs more than a few short haplotypes
    T = std::min<unsigned>(T, std::max<uint32_t>(1, n_haps / 2));
    std::atomic<uint32_t> next {0};
    auto work = [&] {
        std::vector<uint32_t> w;
        for (uint32_t h = next.fetch_add(1); h < n_haps; h = next.fetch_add(1))
            host_penalty_vectors_one(m, bases + off[h], off[h + 1] - off[h], sub ? sub + off[h] : nullptr, w, out, off[h], custom);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

bool model_is_valid(const oct_phmm_error_model* m)       // every table entry a penalty in [0, 127]; table by table (the struct's padding bytes are the caller's)
{
    bool ok = true;
    auto table = [&](const int8_t* t, size_t n) { for (size_t i = 0; i < n; ++i) ok = ok && t[i] >= 0; };
    table(m->at_homopolymer_open, OCT_PHMM_INDEL_TABLE); table(m->cg_homopolymer_open, OCT_PHMM_INDEL_TABLE);
    table(m->dinucleotide_open, OCT_PHMM_INDEL_TABLE); table(m->trinucleotide_open, OCT_PHMM_INDEL_TABLE);
    table(m->homopolymer_extend, OCT_PHMM_INDEL_TABLE); table(m->dinucleotide_extend, OCT_PHMM_INDEL_TABLE); table(m->trinucleotide_extend, OCT_PHMM_INDEL_TABLE);
    for (int k = 0; k < 3; ++k) table(m->snv_caps[k], OCT_PHMM_SNV_TABLE);
    return ok;
}

} // namespace

extern "C" void oct_phmm_error_model_expand(int8_t* dst, uint32_t capacity, const int8_t* src, uint32_t n)
{
    if (!dst || !src || !n) return;
    for (uint32_t i = 0; i < capacity; ++i) dst[i] = src[i < n ? i : n - 1];
}

// The reference's built-in parameter sets (error_model_factory.cpp:220-517) as data: phmm_error_model_tables.hpp, generated from the reference's source by
// tools/make_error_model_tables.py. Names are matched the way the reference's operator>> does (:88-104, :158-182): capitalised, "PCR-FREE" also "PCRF".
namespace {
int library_by_name(const char* name)
{
    if (!name || !*name) return emt::kDefaultLibrary;
    std::string t(name); for (char& c : t) c = (char)toupper((unsigned char)c);          // utils::capitalise
    if (t == "PCR") return emt::pcr;
    if (t == "PCR-FREE" || t == "PCRF") return emt::pcr_free;
    if (t == "10X") return emt::tenx;
    if (t == "MDA") return emt::mda;
    return -1;                                                                            // UnknownLibraryPreparation
}
int sequencer_by_name(const char* name)
{
    if (!name || !*name) return emt::kDefaultSequencer;
    std::string t(name); for (char& c : t) c = (char)toupper((unsigned char)c);
    static const char* const names[emt::kSequencers] = {"HISEQ-2000", "HISEQ-2500", "HISEQ-4000", "X10", "NOVASEQ", "BGISEQ-500", "PACBIO", "PACBIOCCS"};
    for (int i = 0; i < emt::kSequencers; ++i) if (t == names[i]) return i;
    return -1;                                                                            // UnknownSequencer
}
int builtin_model(int lib, int seq, oct_phmm_error_model* m)
{
    if (!m || lib < 0 || seq < 0) return OCT_PHMM_EINVAL;
    if (emt::indel_open[lib][seq][0] < 0) return OCT_PHMM_EINVAL;                          // builtin_indel_models.at() throws: 10X / MDA have no PacBio entries (:366-473)
    memset(m, 0, sizeof(*m));
    auto put = [&](int8_t* dst, uint32_t cap, int row) { const emt::Row& r = emt::rows[row]; oct_phmm_error_model_expand(dst, cap, r.v, r.n); };
    put(m->at_homopolymer_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][0]); put(m->cg_homopolymer_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][1]);
    put(m->dinucleotide_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][2]); put(m->trinucleotide_open, OCT_PHMM_INDEL_TABLE, emt::indel_open[lib][seq][3]);
    put(m->homopolymer_extend, OCT_PHMM_INDEL_TABLE, emt::extend[0]); put(m->dinucleotide_extend, OCT_PHMM_INDEL_TABLE, emt::extend[1]); put(m->trinucleotide_extend, OCT_PHMM_INDEL_TABLE, emt::extend[2]);
    for (int k = 0; k < 3; ++k) put(m->snv_caps[k], OCT_PHMM_SNV_TABLE, emt::snv_caps[lib][k]);
    m->use_snv_model = (seq == emt::pacbio || seq == emt::pacbio_ccs) ? 0 : 1;             // use_snv_error_model :480-483
    return OCT_PHMM_OK;
}
} // namespace

extern "C" void oct_phmm_error_model_default(oct_phmm_error_model* m)
{
    if (m) builtin_model(emt::kDefaultLibrary, emt::kDefaultSequencer, m);                 // default_model_config = {PCR-free, HiSeq-2500} (error_model_factory.hpp:26-28)
}

extern "C" int oct_phmm_error_model_by_name(const char* library_preparation, const char* sequencer, oct_phmm_error_model* m)
{
    return builtin_model(library_by_name(library_preparation), sequencer_by_name(sequencer), m);
}

extern "C" int oct_phmm_error_model_by_label(const char* label, oct_phmm_error_model* m)   // parse_model_config :628-644: "<library>[.<sequencer>]", either part may be empty
{
    if (!label) return OCT_PHMM_EINVAL;
    const std::string l(label);
    const size_t dot = l.find('.');
    const std::string lib = l.substr(0, dot), seq = dot == std::string::npos ? std::string() : l.substr(dot + 1);
    return builtin_model(library_by_name(lib.c_str()), sequencer_by_name(seq.c_str()), m);
}

extern "C" int oct_phmm_penalty_vectors(const oct_phmm_error_model* model, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                                        const uint8_t* substitution_mask, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                        int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status)
{
    if (!model || (n_haps && (!bases || !offsets || !gap_open || !gap_extend || !snv_mask_fwd || !snv_prior_fwd || !snv_mask_rev || !snv_prior_rev)))
        return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (n_haps && !monotone(offsets, n_haps)) return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    if (!model_is_valid(model)) return fail(status, OCT_PHMM_EINVAL, "negative penalty in the error model's tables");
    try {
        host_penalty_vectors(*model, n_haps, (const uint8_t*)bases, offsets, substitution_mask,
                             PenaltyOut {gap_open, gap_extend, (uint8_t*)snv_mask_fwd, snv_prior_fwd, (uint8_t*)snv_mask_rev, snv_prior_rev});
    } catch (const std::exception&) { return fail(status, OCT_PHMM_EHIP, "host allocation"); }
    return ok(status);
}

extern "C" int oct_phmm_set_error_model(oct_phmm_handle* h, const oct_phmm_error_model* model)
{
    if (!h) return OCT_PHMM_EINVAL;
    if (model && !model_is_valid(model)) return OCT_PHMM_EINVAL;
    h->has_model = model != nullptr;
    h->custom.reset();
    if (model) h->model = *model;
    if (h->d_model) { rt::set_device(h->cfg.device_id); rt::stream_sync(h->stream); h->pool.release(h->d_model); h->d_model = nullptr; }
    return OCT_PHMM_OK;
}

// ---- CustomRepeatBasedIndelErrorModel: the indel model of `--sequence-error-model <file>` (phmm_custom_error_model.h) -----------------------------------------
struct oct_phmm_custom_indel_model { std::shared_ptr<const em::CustomIndelModel> m; };

extern "C" int oct_phmm_custom_indel_model_parse(const char* text, size_t len, oct_phmm_custom_indel_model** out)
{
    if (!out || (len && !text)) return OCT_PHMM_EINVAL;
    *out = nullptr;
    try {
        auto m = std::make_shared<em::CustomIndelModel>();
        if (!em::parse_custom_indel_model(text, len, *m)) return OCT_PHMM_EINVAL;     // "Bad model" / MalformedErrorModelFile
        *out = new oct_phmm_custom_indel_model {std::move(m)};
    } catch (const std::exception&) { return OCT_PHMM_EHIP; }
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_custom_indel_model_create(const oct_phmm_motif_penalties* open, uint32_t n_open, int8_t default_open,
                                                  const oct_phmm_motif_penalties* extend, uint32_t n_extend, int32_t has_extend, int8_t default_extend,
                                                  oct_phmm_custom_indel_model** out)
{
    if (!out || (n_open && !open) || (n_extend && !extend) || (n_extend && !has_extend)) return OCT_PHMM_EINVAL;
    *out = nullptr;
    try {
        auto m = std::make_shared<em::CustomIndelModel>();
        auto put = [](em::CustomIndelModel::Map& rows, const oct_phmm_motif_penalties* r, uint32_t n) {
            for (uint32_t i = 0; i < n; ++i) {
                if (!r[i].motif || !r[i].motif_len || !r[i].penalties || !r[i].n_penalties) return false;
                rows.emplace(std::string(r[i].motif, r[i].motif_len), std::vector<int8_t>(r[i].penalties, r[i].penalties + r[i].n_penalties));
            }
            return true;
        };
        if (!put(m->open, open, n_open) || !put(m->extend, extend, n_extend)) return OCT_PHMM_EINVAL;
        m->has_extend = has_extend != 0; m->default_open = default_open; m->default_extend = default_extend;
        *out = new oct_phmm_custom_indel_model {std::move(m)};
    } catch (const std::exception&) { return OCT_PHMM_EHIP; }
    return OCT_PHMM_OK;
}

extern "C" void oct_phmm_custom_indel_model_destroy(oct_phmm_custom_indel_model* m) { delete m; }

extern "C" int oct_phmm_custom_indel_model_info(const oct_phmm_custom_indel_model* m, int8_t* default_open, int8_t* default_extend, uint32_t* n_open_rows, uint32_t* n_extend_rows, int32_t* has_extend)
{
    if (!m) return OCT_PHMM_EINVAL;
    if (default_open) *default_open = m->m->default_open;
    if (default_extend) *default_extend = m->m->default_extend;
    if (n_open_rows) *n_open_rows = (uint32_t)m->m->open.size();
    if (n_extend_rows) *n_extend_rows = (uint32_t)m->m->extend.size();
    if (has_extend) *has_extend = m->m->has_extend ? 1 : 0;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_custom_penalty_vectors(const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv, uint32_t n_haps, const char* bases, const uint32_t* offsets,
                                               const uint8_t* substitution_mask, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                               int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status)
{
    if (!indel || (n_haps && (!bases || !offsets || !gap_open || !gap_extend || !snv_mask_fwd || !snv_prior_fwd || !snv_mask_rev || !snv_prior_rev)))
        return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (n_haps && !monotone(offsets, n_haps)) return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    oct_phmm_error_model dflt;
    if (!snv) { oct_phmm_error_model_default(&dflt); snv = &dflt; }                    // make_snv_error_model(default_model_config), error_model_factory.cpp:587
    if (!model_is_valid(snv)) return fail(status, OCT_PHMM_EINVAL, "negative penalty in the error model's tables");
    try {
        host_penalty_vectors(*snv, n_haps, (const uint8_t*)bases, offsets, substitution_mask,
                             PenaltyOut {gap_open, gap_extend, (uint8_t*)snv_mask_fwd, snv_prior_fwd, (uint8_t*)snv_mask_rev, snv_prior_rev}, indel->m.get());
    } catch (const std::exception&) { return fail(status, OCT_PHMM_EHIP, "host allocation"); }
    return ok(status);
}

extern "C" int oct_phmm_set_custom_error_model(oct_phmm_handle* h, const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv)
{
    if (!h || !indel) return OCT_PHMM_EINVAL;
    oct_phmm_error_model dflt;
    if (!snv) { oct_phmm_error_model_default(&dflt); snv = &dflt; }
    const int rc = oct_phmm_set_error_model(h, snv);
    if (rc != OCT_PHMM_OK) return rc;
    h->custom = indel->m;                                  // shared: the caller may destroy its model object
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_device_count(void)
{
    int n = 0, ok = 0;
    if (!rt::device_count(&n) || n <= 0) return 0;
    for (int d = 0; d < n; ++d) ok += rt::device_is_gfx950(d) ? 1 : 0;
    return ok == n ? n : 0;            // ordinals are HIP's: a node with anything but gfx950 devices is not a target
}

extern "C" void oct_phmm_config_default(oct_phmm_config* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->max_indel_error = 8; c->use_int_scores = 0; c->use_mapping_quality = 1; c->mapping_quality_cap = 120;
    c->mapping_quality_cap_trigger = -1; c->use_flank_state = 1; c->nuc_prior = 2; c->max_mapping_positions = 10; c->device_id = 0;
}

extern "C" const char* oct_phmm_strerror(int code)
{
    switch (code) {
        case OCT_PHMM_OK: return "ok";
        case OCT_PHMM_EINVAL: return "invalid batch";
        case OCT_PHMM_EBAND: return "requested band size is too large";
        case OCT_PHMM_ESHORT_HAPLOTYPE: return "Haplotype is too short for alignment";
        case OCT_PHMM_EHIP: return "HIP runtime error";
        case OCT_PHMM_ENODEVICE: return "no gfx950 device";
        case OCT_PHMM_EUNSUPPORTED: return "unsupported configuration";
        case OCT_PHMM_EOVERFLOW: return "Pair HMM alignment overflowed";
        default: return "unknown";
    }
}

extern "C" int oct_phmm_create(const oct_phmm_config* cfg, oct_phmm_handle** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(oct_phmm_config)) return OCT_PHMM_EINVAL;
    *out = nullptr;
    const int band = band_for(cfg->max_indel_error);
    if (band < 0) return OCT_PHMM_EBAND;
    if (cfg->max_mapping_positions < 0 || cfg->max_mapping_positions >= kMaxSlots) return OCT_PHMM_EUNSUPPORTED;
    int n = 0;
    if (!rt::device_count(&n) || n <= 0 || cfg->device_id < 0 || cfg->device_id >= n) return OCT_PHMM_ENODEVICE;
    if (!rt::device_is_gfx950(cfg->device_id)) return OCT_PHMM_ENODEVICE;
    if (!rt::set_device(cfg->device_id)) return OCT_PHMM_EHIP;
    std::unique_ptr<oct_phmm_handle> h(new (std::nothrow) oct_phmm_handle());
    if (!h) return OCT_PHMM_EHIP;
    h->cfg = *cfg; h->band = band; h->wide = cfg->use_int_scores != 0; h->lanes_c = band > 64 ? band / 64 : 1;
    h->pool.set_device(cfg->device_id);
    if (h->cfg.mapping_quality_cap_trigger >= 0 && h->cfg.mapping_quality_cap_trigger >= h->cfg.mapping_quality_cap)
        h->cfg.mapping_quality_cap_trigger = -1;                                     // model.cpp:50-52
    h->timing = tune::timing();
    {   // never plan for more than 60 % of what the device has free now (two server handles per device, other processes)
        size_t free_b = 0, total_b = 0;
        if (rt::mem_info(&free_b, &total_b) && free_b) h->bp_budget = std::min<size_t>(h->bp_budget, free_b / 10 * 6);
    }
    { long long v; if (tune::number("OCT_PHMM_BP_BUDGET_GB", &v) && v > 0) h->bp_budget = (size_t)v << 30;
      if (tune::number("OCT_PHMM_BP_BUDGET_KB", &v) && v > 0) h->bp_budget = (size_t)v << 10; }   // KB: test hook, forces chunked traceback launches on small batches
    { long long v; if (tune::number("OCT_PHMM_TEST_FAIL_BP_ALLOCS", &v) && v > 0) h->fail_bp_allocs = (int)v; }
    // The handle's own stream carries the chain a caller waits for (mapper -> classifier -> traceback DP -> walk -> epilogue); the score-only DP of a region-sized or
    // mid-size batch runs beside it on the second stream and is off the critical path as long as the traceback DP gets its CUs first: stream 0 at the device's highest
    // priority, the others normal.
    if (!rt::stream_create_priority(&h->stream, true)) return OCT_PHMM_EHIP;
    h->main_stream_high_priority = true;
    for (auto& es : h->extra_streams) if (!rt::stream_create(&es)) return OCT_PHMM_EHIP;
    if (!rt::event_create(&h->ev_ready)) return OCT_PHMM_EHIP;
    *out = h.release();
    return OCT_PHMM_OK;
}

extern "C" void oct_phmm_destroy(oct_phmm_handle* h)
{
    if (!h) return;
    rt::set_device(h->cfg.device_id);
    rt::stream_sync(h->stream);
    for (auto& es : h->extra_streams) { rt::stream_sync(es); rt::stream_destroy(es); }
    for (int i = 0; i < oct_phmm_handle::kMaxSlices; ++i) rt::dev_free(h->bp[i]);
    for (auto& kv : h->pool.live) rt::dev_free(kv.first);
    h->pool.live.clear(); h->pool.trim();
    rt::host_pinned_free(h->stage); rt::host_pinned_free(h->out_stage);
    for (void* p : h->stat_stage_free) rt::host_pinned_free(p);
    if (h->probe_ready) { rt::stream_sync(h->probe_stream); rt::stream_destroy(h->probe_stream); rt::dev_free(h->d_probe); rt::host_pinned_free(h->h_probe); }
    for (rt::Event e : h->ev_pool) rt::event_destroy(e);
    rt::event_destroy(h->ev_ready);
    rt::stream_destroy(h->stream);
    delete h;
}

extern "C" int oct_phmm_band_size(const oct_phmm_handle* h) { return h ? h->band : -1; }

extern "C" int oct_phmm_set_timing(oct_phmm_handle* h, int enabled)
{
    if (!h) return OCT_PHMM_EINVAL;
    h->timing = enabled != 0;
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------------------------------------------
extern "C" void oct_phmm_batch_free(oct_phmm_handle* h, oct_phmm_batch* b)
{
    if (!b) return;
    if (h && !b->synced) { rt::set_device(h->cfg.device_id); rt::stream_sync(h->stream); for (auto& es : h->extra_streams) rt::stream_sync(es); }   // (a waited batch has nothing in flight)
    if (!h) h = b->owner;
    if (b->stat_stage) h->stat_stage_free.push_back(b->stat_stage);
    for (auto& t : b->timers) { h->put_event(t.first); h->put_event(t.second); }
    for (void* p : b->allocs) h->pool.release(p);
    for (auto& sl : b->slices) { h->pool.release(sl.d_tasks); h->pool.release(sl.d_tasks_sorted); h->pool.release(sl.d_ends); h->pool.release(sl.d_keys); h->put_event(sl.done); if (b->dedup) h->put_event(sl.matched); }
    if (b->ev_fork) { h->put_event(b->ev_fork); h->put_event(b->ev_join); h->put_event(b->ev_hashes); }
    delete b;
}

static int upload_impl(oct_phmm_handle* h, const oct_phmm_reads* R, const oct_phmm_haplotypes* H,
                       const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                       const oct_phmm_positions* positions, oct_phmm_batch** out, oct_phmm_status* status, bool align_mode, uint32_t max_cigar_ops,
                       bool one_shot = false, const InputFacts* pre = nullptr);

extern "C" int oct_phmm_batch_upload(oct_phmm_handle* h, const oct_phmm_reads* R, const oct_phmm_haplotypes* H,
                                     const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                                     const oct_phmm_positions* positions, oct_phmm_batch** out, oct_phmm_status* status)
{
    return upload_impl(h, R, H, regions, flank, positions, out, status, false, 0);
}

static int upload_impl_body(oct_phmm_handle* h, const oct_phmm_reads* R, const oct_phmm_haplotypes* H_in,
                            const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                            const oct_phmm_positions* positions, oct_phmm_batch** out, oct_phmm_status* status, bool align_mode, uint32_t max_cigar_ops,
                            bool one_shot, const InputFacts* pre);
static int upload_impl(oct_phmm_handle* h, const oct_phmm_reads* R, const oct_phmm_haplotypes* H_in,
                       const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                       const oct_phmm_positions* positions, oct_phmm_batch** out, oct_phmm_status* status, bool align_mode, uint32_t max_cigar_ops,
                       bool one_shot, const InputFacts* pre)
{
    const int rc = upload_impl_body(h, R, H_in, regions, flank, positions, out, status, align_mode, max_cigar_ops, one_shot, pre);
    // An upload that fails after its copies were enqueued returns to a caller who may free the arrays at once - and page-locked arrays are read by the copy
    // engines directly (Packer::commit): nothing of this handle is in flight any more when the error is reported (the error path only).
    if (rc != OCT_PHMM_OK && h) { rt::set_device(h->cfg.device_id); rt::stream_sync(h->stream); rt::clear_error(); }
    return rc;
}
static int upload_impl_body(oct_phmm_handle* h, const oct_phmm_reads* R, const oct_phmm_haplotypes* H_in,
                            const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                            const oct_phmm_positions* positions, oct_phmm_batch** out, oct_phmm_status* status, bool align_mode, uint32_t max_cigar_ops,
                            bool one_shot, const InputFacts* pre)    // pre: somebody has looked at every byte already (the region server's callers, each at its own region on its own thread)
{
    if (!h || !R || !H_in || !out) return fail(status, OCT_PHMM_EINVAL, "null argument");
    *out = nullptr;
    rt::Range range_("oct_phmm upload");
    // OCT_PHMM_UPLOAD_PROFILE: where a big upload's host time goes (one stderr line per upload)
    const bool up_prof = tune::prof_flag("OCT_PHMM_UPLOAD_PROFILE");
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_up0 = up_prof ? now_ms() : 0; double t_up1 = 0, t_up2 = 0;
    oct_phmm_haplotypes Hv = *H_in;
    const oct_phmm_haplotypes* H = &Hv;
    const uint8_t* sub_mask = H->substitution_mask;                          // only read where the library makes the vectors
    const int n_vec = (H->gap_open ? 1 : 0) + (H->gap_extend ? 1 : 0) + (H->snv_mask_fwd ? 1 : 0) + (H->snv_prior_fwd ? 1 : 0) + (H->snv_mask_rev ? 1 : 0) + (H->snv_prior_rev ? 1 : 0);
    const bool generate = H->n_haps && n_vec == 0;                           // HaplotypeLikelihoodModel::reset inside the call (oct_phmm_set_error_model)
    if (generate && !h->has_model) return fail(status, OCT_PHMM_EINVAL, "penalty vectors are NULL and the handle has no error model");
    if ((R->n_reads && (!R->bases || !R->qualities || !R->offsets || !R->mapping_quality || !R->reverse_strand || !R->ref_begin))
        || (H->n_haps && (!H->bases || !H->offsets || !H->ref_begin || (!generate && n_vec != 6))))
        return fail(status, OCT_PHMM_EINVAL, "null array");
    if (!R->offsets || !H->offsets || !monotone(R->offsets, R->n_reads) || !monotone(H->offsets, H->n_haps))
        return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    // Where the vectors are made: region-sized calls on host threads (one haplotype takes a host core 30 us and a wave 0.5 ms, and the call is
    // latency-bound), batches on the device, one haplotype per wave (five waves per CU beat sixteen cores from a few hundred haplotypes on).
    // OCT_PHMM_PENALTIES=host|device|lanes overrides.
    bool gen_device = generate && H->n_haps >= 512;
    if (tune::penalties_where()) gen_device = generate && tune::penalties_where() == 2;
    if (h->custom) gen_device = false;                                       // a model file's rows are looked up by motif string: host threads at every size
    std::vector<int8_t> gen_go, gen_ge, gen_pf, gen_pr; std::vector<char> gen_mf, gen_mr;
    if (generate && !gen_device) {
        const size_t nb = H->offsets[H->n_haps];
        try {
            gen_go.resize(nb + 1); gen_ge.resize(nb + 1); gen_pf.resize(nb + 1); gen_pr.resize(nb + 1); gen_mf.resize(nb + 1); gen_mr.resize(nb + 1);
            host_penalty_vectors(h->model, H->n_haps, (const uint8_t*)H->bases, H->offsets, sub_mask,
                                 PenaltyOut {gen_go.data(), gen_ge.data(), (uint8_t*)gen_mf.data(), gen_pf.data(), (uint8_t*)gen_mr.data(), gen_pr.data()}, h->custom.get());
        } catch (const std::exception&) { return fail(status, OCT_PHMM_EHIP, "host allocation"); }
        Hv.gap_open = gen_go.data(); Hv.gap_extend = gen_ge.data(); Hv.snv_mask_fwd = gen_mf.data(); Hv.snv_prior_fwd = gen_pf.data();
        Hv.snv_mask_rev = gen_mr.data(); Hv.snv_prior_rev = gen_pr.data();
    }
    const uint32_t n_rows = R->row_offsets ? R->n_rows : R->n_reads;
    if (R->row_offsets && (!monotone(R->row_offsets, n_rows) || R->row_offsets[0] != 0 || R->row_offsets[n_rows] != R->n_reads))
        return fail(status, OCT_PHMM_EINVAL, "row_offsets must partition the reads");
    const uint32_t n_read_bases = R->offsets[R->n_reads], n_hap_bases = H->offsets[H->n_haps];
    // regions
    uint32_t one_row[2] = {0, n_rows}, one_hap[2] = {0, H->n_haps};
    uint8_t one_hf = flank ? 1 : 0; oct_phmm_flank_state one_fl = flank ? *flank : oct_phmm_flank_state {0, 0};
    uint32_t G = 1; const uint32_t* g_row = one_row; const uint32_t* g_hap = one_hap;
    const uint8_t* g_hf = &one_hf; const oct_phmm_flank_state* g_fl = &one_fl;
    if (regions) {
        G = regions->n_regions; g_row = regions->row_offsets; g_hap = regions->hap_offsets; g_hf = regions->has_flank; g_fl = regions->flank;
        if (!g_row || !g_hap || !monotone(g_row, G) || !monotone(g_hap, G) || g_row[0] != 0 || g_hap[0] != 0
            || g_row[G] != n_rows || g_hap[G] != H->n_haps || (g_hf && !g_fl))
            return fail(status, OCT_PHMM_EINVAL, "region tables must partition rows and haplotypes");
    }
    auto first_read = [&](uint32_t row) { return R->row_offsets ? R->row_offsets[row] : row; };

    struct BatchDel { void operator()(oct_phmm_batch* p) const { oct_phmm_batch_free(p->owner, p); } };   // a failed upload returns its blocks and events to the handle
    std::unique_ptr<oct_phmm_batch, BatchDel> b(new (std::nothrow) oct_phmm_batch());
    if (!b) return fail(status, OCT_PHMM_EHIP, "host allocation");
    b->owner = h; b->n_reads = R->n_reads; b->n_haps = H->n_haps; b->n_rows = n_rows; b->n_regions = G; b->n_hap_bases = n_hap_bases;
    std::vector<uint32_t> hap_region(H->n_haps + 1, 0), reg_row0(G + 1), reg_read0(G + 1), reg_lhs(G + 1, 0), reg_rhs(G + 1, 0);
    std::vector<uint64_t> hap_out_off(H->n_haps + 1, 0), hap_pair_off(H->n_haps + 1, 0);
    for (uint32_t g = 0; g < G; ++g) {
        reg_row0[g] = g_row[g]; reg_read0[g] = first_read(g_row[g]);
        const uint32_t rows = g_row[g + 1] - g_row[g], nreads = first_read(g_row[g + 1]) - first_read(g_row[g]);
        if (h->cfg.use_flank_state && g_hf && g_hf[g]) { reg_lhs[g] = g_fl[g].lhs_flank; reg_rhs[g] = g_fl[g].rhs_flank; }   // model.cpp:276-282
        int64_t first_begin = INT64_MAX;
        for (uint32_t r = reg_read0[g]; r < reg_read0[g] + nreads; ++r) first_begin = std::min(first_begin, R->ref_begin[r]);
        for (uint32_t hp = g_hap[g]; hp < g_hap[g + 1]; ++hp) {
            hap_region[hp] = g; hap_out_off[hp + 1] = hap_out_off[hp] + rows; hap_pair_off[hp + 1] = hap_pair_off[hp] + nreads;
            if (first_begin < H->ref_begin[hp]) return fail(status, OCT_PHMM_EINVAL, "read begins before its haplotype (contains() violated)");
        }
    }
    reg_row0[G] = g_row[G]; reg_read0[G] = first_read(g_row[G]);
    b->h_hap_region = hap_region; b->h_hap_out_off = hap_out_off; b->h_reg_hap0.assign(g_hap, g_hap + G + 1);
    b->n_out = hap_out_off[H->n_haps]; b->n_pairs = hap_pair_off[H->n_haps];
    if (b->n_pairs >= 0xffffffffull) return fail(status, OCT_PHMM_EUNSUPPORTED, "more than 2^32-1 pairs in one batch");
    // The contract's range checks and the bounds the FASTADD decision below needs, in ONE pass over the read qualities and one over the penalty vectors. These loops are
    // the first touch of every byte of the batch - memory-bound on a core (a 64-region device batch of the region server: 12.5 MB, 1.4 ms of a 2.2 ms upload when each
    // check was its own single-threaded pass) - so a batch from ~2 MB on is cut over up to four threads. Inner loops are branch-free and vectorise.
    // A batch that may take device-sized launches also learns here whether every base is one of ACGT and every SNV mask byte set (a clean batch launches no generic kernels).
    uint32_t q_or = 0, pen_or = 0, gomax = 0, gemax = 0, t_min = 0xffffffffu, dirty = 0; uint64_t sum_q_max = 0; bool any_empty = false;
    const bool dsl_wanted = !align_mode && tune::device_sized() != 0 && (b->n_pairs <= kDslMaxPairs || tune::device_sized() > 0);
    const bool pre_ok = pre && (gen_device || generate || pre->have_haps);      // (vectors the library makes on host threads are looked at here, below)
    const int flavour_hint = pre_ok ? pre->dirty : -1;
    // (region-sized calls only: from a few regions on the scan - the first touch of every base and mask byte, 0.1 ms of a 16-region upload although it takes eight bytes
    // per step - costs the host more than the three near-empty launches of the generic lists cost the device, and the region server's workers are bound by their host work)
    const bool want_dirty = dsl_wanted && !gen_device && b->n_pairs <= 20000 && flavour_hint < 0;
    if (flavour_hint >= 0) dirty = flavour_hint ? 1u : 0u;
    {
        std::mutex mx;
        InputFacts all;
        if (pre_ok) { all = *pre; all.dirty = 0; }
        else {
            const size_t read_grain = std::max<size_t>(1, (size_t)R->n_reads / std::max<size_t>(1, (size_t)n_read_bases >> 20));      // reads per ~1 MB of qualities
            host_parallel(R->n_reads, read_grain, [&](size_t r0, size_t r1) {
                InputFacts f; f.dirty = 0; facts_of_reads(R, r0, r1, want_dirty, &f);
                std::lock_guard<std::mutex> lk(mx); all.merge(f); if (f.dirty > 0) dirty = 1;
            });
        }
        if (!gen_device && !(pre_ok && pre->have_haps)) host_parallel(n_hap_bases, (size_t)1 << 18, [&](size_t lo, size_t hi) {       // (device-made vectors come out of validated tables)
            InputFacts f; f.dirty = 0; facts_of_haps(H, lo, hi, want_dirty, &f);
            std::lock_guard<std::mutex> lk(mx); all.merge(f); if (f.dirty > 0) dirty = 1;
        });
        q_or = all.q_or; pen_or = all.pen_or; gomax = all.gomax; gemax = all.gemax; t_min = all.t_min; sum_q_max = all.sum_q_max;
        any_empty = R->n_reads && t_min == 0;
        if (q_or & 0x80u) return fail(status, OCT_PHMM_EINVAL, "base quality > 127");
        if (pen_or & 0x80u) return fail(status, OCT_PHMM_EINVAL, "negative penalty");
    }

    for (uint32_t r = 0; r < R->n_reads; ++r) b->t_cap = std::max(b->t_cap, R->offsets[r + 1] - R->offsets[r]);
    for (uint32_t hp = 0; hp < H->n_haps; ++hp) b->lh_cap = std::max(b->lh_cap, H->offsets[hp + 1] - H->offsets[hp]);
    {
        const bool fits = h->band <= 64 && dp_lds_bytes(b->t_cap, b->lh_cap, (uint32_t)h->band, true, h->wide ? 0u : dp_rec_chunk(b->t_cap, b->lh_cap, (uint32_t)h->band, true)) <= rt::kMaxLdsBytes;
        b->stream = h->band > 64 || !fits;      // long reads at any band stream their operands (PacBioCCS.config: max-indel-errors=16 with 10-20 kb reads)
        b->multi_wave = h->band >= 128 && h->wide && tune::multi_wave();
        b->rows32 = b->stream && h->band == 16 && h->wide && tune::dp_rows();
        if (b->t_cap + 2 * (uint32_t)h->band >= 32768) return fail(status, OCT_PHMM_EUNSUPPORTED, "read too long (walk events hold 15-bit coordinates)");
    }
    {
        // Can any biased int16 lane exceed 0xFFFF (= the reference's own lane wrapping)? Every finite cell is bounded by the pure-match
        // path along its diagonal plus one gap opening, every not-yet-initialised ("infinite") cell by infinity_ plus the deletion chain's
        // growth (the 0x7FF tolerance the reference itself relies on, simd_pair_hmm.hpp:55). If neither can, a 32-bit add of two packed
        // halves never carries between them and k_dp uses v_add_u32 (FASTADD); otherwise it keeps v_pk_add_u16. Results are identical.
        if (gen_device) {                                   // the vectors do not exist yet: bound them by the model's tables
            const oct_phmm_error_model& m = h->model;
            for (int i = 0; i < OCT_PHMM_INDEL_TABLE; ++i) {
                gomax = std::max<uint32_t>(gomax, std::max(std::max(m.at_homopolymer_open[i], m.cg_homopolymer_open[i]), std::max(m.dinucleotide_open[i], m.trinucleotide_open[i])));
                gemax = std::max<uint32_t>(gemax, std::max(m.homopolymer_extend[i], std::max(m.dinucleotide_extend[i], m.trinucleotide_extend[i])));
            }
        }
        const uint64_t B64 = (uint64_t)h->band, nuc = (uint64_t)std::max(0, h->cfg.nuc_prior);
        // A read shorter than its wave's longest keeps iterating (padding quality 64) after its end cells were captured: its rows past the end
        // grow by at most one insertion step (gap extend + nuc_prior) per iteration. An uninitialised lane runs its insertion chain
        // (gap extend + nuc_prior per step) and its deletion chain for up to 2 B steps before the rolling initialiser reaches it.
        const uint64_t tail = R->n_reads ? (uint64_t)(b->t_cap - std::min(b->t_cap, t_min)) * (gemax + nuc) : 0;
        // (+ nuc once more: window-paired segments add nuc_prior to BOTH candidates of an insertion's minimum before they compare, not to the winner after it)
        const uint64_t finite = 4 * (sum_q_max + 2 * 64 * B64 + gomax + gemax + 2 * nuc + tail) + 1024;
        const uint64_t garbage = 4 * (2 * B64 * (gemax + nuc) + 64 + gomax + gemax + 2 * nuc) + 64;
        b->fast_adds = finite < 0xF800u && garbage < 0x7FFu && h->cfg.nuc_prior >= 0 && !tune::exact_adds();
    }
    if (any_empty) return fail(status, OCT_PHMM_EINVAL, "empty read");
    std::vector<uint32_t> h_pos; std::vector<uint8_t> h_npos;
    const uint32_t S = (uint32_t)h->cfg.max_mapping_positions;
    if (positions) {
        if (b->n_pairs && (!positions->offsets || !positions->positions)) return fail(status, OCT_PHMM_EINVAL, "positions arrays null");
        h_pos.assign((size_t)b->n_pairs * S + 1, 0); h_npos.assign((size_t)b->n_pairs + 1, 0);
        for (uint64_t e = 0; e < b->n_pairs; ++e) {
            const uint64_t p0 = positions->offsets[e], p1 = positions->offsets[e + 1];
            if (p1 < p0 || p1 - p0 > S) return fail(status, OCT_PHMM_EINVAL, "more mapping positions than max_mapping_positions");
            h_npos[e] = (uint8_t)(p1 - p0);
            for (uint64_t j = p0; j < p1; ++j) h_pos[e * S + (j - p0)] = positions->positions[j];
        }
    } else {
        b->device_map = true;
        b->map_big = kmer_map_lds_bytes(b->lh_cap) > rt::kMaxLdsBytes || tune::big_mapper();
        if (b->lh_cap >= 65536 || (size_t)b->lh_cap * 4 + 64 > rt::kMaxLdsBytes)
            return fail(status, OCT_PHMM_EUNSUPPORTED, "haplotype too long for the k-mer mapper (>= 40k bases)");
    }
    b->h_roff.assign(R->offsets, R->offsets + R->n_reads + 1); b->h_hoff.assign(H->offsets, H->offsets + H->n_haps + 1);
    b->h_rbegin.assign(R->ref_begin, R->ref_begin + R->n_reads); b->h_hbegin.assign(H->ref_begin, H->ref_begin + H->n_haps);

    RT(rt::set_device(h->cfg.device_id));
    rt::Stream s = h->stream;
    DevBatch& d = b->d;
    d.n_reads = R->n_reads; d.n_rows = n_rows; d.n_haps = H->n_haps; d.n_regions = G; d.n_pairs = b->n_pairs;
    d.band = h->band; d.nuc_prior = h->cfg.nuc_prior; d.max_pos = h->cfg.max_mapping_positions; d.wide = ((h->wide || b->stream) && !b->multi_wave && !b->rows32) ? 1 : 0;   // 1: every task takes the generic lists
    d.use_mapq = h->cfg.use_mapping_quality; d.mapq_cap = h->cfg.mapping_quality_cap; d.mapq_trigger = h->cfg.mapping_quality_cap_trigger;
    oct_phmm_batch* bp = b.get();
    Packer pk;
    pk.upload((const uint8_t*)R->bases, n_read_bases, &d.rbases);
    pk.upload(R->qualities, n_read_bases, &d.rquals);
    pk.upload(R->offsets, (size_t)R->n_reads + 1, &d.roff);
    pk.upload(R->mapping_quality, R->n_reads, &d.rmapq);
    pk.upload(R->reverse_strand, R->n_reads, &d.rrev);
    pk.upload(R->ref_begin, R->n_reads, &d.rbegin);
    d.row_off = nullptr;
    if (R->row_offsets) pk.upload(R->row_offsets, (size_t)n_rows + 1, &d.row_off);
    pk.upload((const uint8_t*)H->bases, n_hap_bases, &d.hbases);
    pk.upload(H->offsets, (size_t)H->n_haps + 1, &d.hoff);
    pk.upload(H->ref_begin, H->n_haps, &d.hbegin);
    const uint8_t* d_sub_mask = nullptr;
    if (gen_device) {                                       // written by k_penalty_vectors below
        pk.dalloc((int8_t**)&d.go, (size_t)n_hap_bases + 16); pk.dalloc((int8_t**)&d.ge, (size_t)n_hap_bases + 16);
        pk.dalloc((uint8_t**)&d.maskF, (size_t)n_hap_bases + 16); pk.dalloc((int8_t**)&d.priorF, (size_t)n_hap_bases + 16);
        pk.dalloc((uint8_t**)&d.maskR, (size_t)n_hap_bases + 16); pk.dalloc((int8_t**)&d.priorR, (size_t)n_hap_bases + 16);
        if (sub_mask) pk.upload(sub_mask, n_hap_bases, &d_sub_mask);
    } else {
    pk.upload(H->gap_open, n_hap_bases, &d.go);
    pk.upload(H->gap_extend, n_hap_bases, &d.ge);
    pk.upload((const uint8_t*)H->snv_mask_fwd, n_hap_bases, &d.maskF);
    pk.upload(H->snv_prior_fwd, n_hap_bases, &d.priorF);
    pk.upload((const uint8_t*)H->snv_mask_rev, n_hap_bases, &d.maskR);
    pk.upload(H->snv_prior_rev, n_hap_bases, &d.priorR);
    }
    pk.upload(hap_region.data(), hap_region.size(), &d.hap_region);
    pk.upload(hap_out_off.data(), hap_out_off.size(), &d.hap_out_off);
    pk.upload(hap_pair_off.data(), hap_pair_off.size(), &d.hap_pair_off);
    pk.upload(reg_row0.data(), reg_row0.size(), &d.reg_row0);
    pk.upload(reg_read0.data(), reg_read0.size(), &d.reg_read0);
    pk.upload(reg_lhs.data(), reg_lhs.size(), &d.reg_lhs);
    pk.upload(reg_rhs.data(), reg_rhs.size(), &d.reg_rhs);
    pk.dalloc(&d.pos, (size_t)b->n_pairs * S + 1); pk.dalloc(&d.npos, (size_t)b->n_pairs + 1);
    d.bin_start = nullptr; d.bin_idx = nullptr; d.rhash = nullptr; d.bin32 = nullptr; d.hhash = nullptr; d.map_count_only = 0; d.map_stats = 0; d.pair_mm = nullptr; d.rhash_rows = nullptr; d.rhash_stride = 0; d.rcode = nullptr; d.rcode_words = 0;
    if (!positions) {
        if (b->map_big) pk.dalloc(&d.bin_start, (size_t)H->n_haps * (kKmerBins + 1) + 1);      // the u16 table of k_kmer_map_big only
        pk.dalloc(&d.bin_idx, (size_t)n_hap_bases + 1);
        pk.dalloc(&d.hhash, (size_t)n_hap_bases + 1);
        d.map_count_only = tune::map_count_only(); d.map_stats = tune::map_stats();
        b->map_reads_per_block = b->n_pairs < 500000 ? 16 : 256;      // the haplotype's tables are staged once per workgroup: big batches amortise them over more reads
        pk.dalloc(&d.bin32, (size_t)H->n_haps * kKmerBins + 4);
        // lane-per-pair mapper (k_kmer_map_lanes: 256 reads of one haplotype per workgroup, the exact shortcut per lane): batches big enough to fill the chip with
        // 256-pair workgroups; region-sized calls keep one wave per pair (more, shorter waves). OCT_PHMM_LANE_MAPPER=0 / 1 forces one or the other.
        {
            long long want = -1; tune::number("OCT_PHMM_LANE_MAPPER", &want);
            const uint32_t nq_cap = b->t_cap >= kKmer ? b->t_cap - kKmer + 1 : 0;
            const bool can = !b->map_big && nq_cap >= 1 && nq_cap <= kLaneMapMaxKmers && kmer_map_lanes_lds_bytes(b->lh_cap) <= rt::kMaxLdsBytes;
            if (can && (want >= 0 ? want != 0 : b->n_pairs >= kLaneMapMinPairs)) b->map_lanes = (int)kLaneMapThreads;
        }
        if (!b->map_lanes) pk.dalloc(&d.rhash, (size_t)n_read_bases + 1);        // (the lane mapper reads rhash_rows instead)
        if (b->map_lanes) {
            b->map_reads_per_block = (uint32_t)b->map_lanes;
            d.rhash_stride = rhash_row_stride(b->t_cap);
            pk.dalloc(&d.rhash_rows, (size_t)R->n_reads * d.rhash_stride + 64);
            d.rcode_words = rcode_row_words(b->t_cap);                                  // the reads' 2-bit codes in tiles of 64 reads (the bit-parallel pass)
            pk.dalloc(&d.rcode, (size_t)((R->n_reads + 63) / 64) * d.rcode_words * 64 + 64);
            if (tune::map_mismatches()) pk.dalloc(&d.pair_mm, (size_t)b->n_pairs + 2);   // k_kmer_map_lanes tells k_classify what it saw along the mapped position
        }
        std::vector<uint32_t> blk_hap, blk_read0;           // one k_kmer_map workgroup per (haplotype, read chunk of its region)
        for (uint32_t hp = 0; hp < H->n_haps; ++hp) {
            const uint32_t g = hap_region[hp];
            for (uint32_t r = reg_read0[g]; r < first_read(g_row[g + 1]); r += b->map_reads_per_block) { blk_hap.push_back(hp); blk_read0.push_back(r); }
        }
        b->n_map_blocks = (uint32_t)blk_hap.size();
        b->h_blk_hap = blk_hap;
        b->h_blk_read0 = blk_read0;
        pk.upload(b->h_blk_hap.data(), b->h_blk_hap.size(), (const uint32_t**)&b->d_blk_hap);
        pk.upload(b->h_blk_read0.data(), b->h_blk_read0.size(), (const uint32_t**)&b->d_blk_read0);
    }
    pk.dalloc(&d.racgt, (size_t)R->n_reads);
    d.rrec = nullptr; d.rrec_stride = 0; d.rrecW = nullptr;
    if (!b->stream && !h->wide && R->n_reads) {           // the LDS-resident int16 kernels read their read-side operands from per-read record rows
        d.rrec_stride = dp_rec_n(b->t_cap, (uint32_t)h->band);
        pk.dalloc(&d.rrec, (size_t)R->n_reads * d.rrec_stride);
    }
    if ((b->multi_wave || b->rows32) && R->n_reads) {                    // ... and so does the multi-wave streaming kernel (16 bytes per entry: both cost flavours)
        d.rrec_stride = dp_rec_n(b->t_cap, (uint32_t)h->band);
        pk.dalloc(&d.rrecW, (size_t)R->n_reads * d.rrec_stride);
    }
    pk.dalloc(&d.tabFastF, (size_t)n_hap_bases + 16); pk.dalloc(&d.tabFastR, (size_t)n_hap_bases + 16);      // (+16: k_dp_mw's last operand chunks run past a window)
    pk.dalloc(&d.tabGenF, (size_t)n_hap_bases + 16);  pk.dalloc(&d.tabGenR, (size_t)n_hap_bases + 16);
    pk.dalloc(&d.pair_best, (size_t)b->n_pairs); pk.dalloc(&d.pair_cls, (size_t)b->n_pairs);
    pk.dalloc(&d.pair_extra, (size_t)b->n_pairs); pk.dalloc(&d.pair_cnt, (size_t)b->n_pairs + oct_phmm_handle::kMaxSlices + 1);
    // Exact de-duplication of pairs (phmm_kernels.hpp): populate on the LDS-resident int16 path, where some region has several haplotypes and
    // the batch is big enough for the matcher's walk over a region's haplotypes (one after the other, ~1.5 us each) not to show: a 1k x 64
    // call went from 0.48 to 0.71 ms with it, the 100k x 128 batch from 32.4 to 30.9 ms, the 2,000-region stream from 49.6 to 44.5 ms.
    d.dedup_hash_mask = tune::dedup_hash_mask();
    d.canon = nullptr; d.pair_rep = nullptr; d.pair_hash = nullptr; d.pair_fast = nullptr; d.dd_hash = d.dd_hap = d.dd_n = nullptr; d.window_len = b->t_cap + 2 * (uint32_t)h->band - 1;
    b->dedup = !align_mode && !b->stream && !h->wide && H->n_haps > G && b->lh_cap <= 8192 && b->n_pairs >= 500000 && tune::dedup() != 0;
    if (tune::dedup() > 0) b->dedup = !align_mode && !b->stream && !h->wide && H->n_haps > G;
    for (uint32_t g = 0; g < G && b->dedup; ++g) if (g_hap[g + 1] - g_hap[g] > 65535) b->dedup = false;     // (the matcher's table holds 16-bit haplotype numbers within a region)
    if (b->dedup) { pk.dalloc(&d.canon, (size_t)n_hap_bases + 1); pk.dalloc(&d.pair_rep, (size_t)b->n_pairs + 1); pk.dalloc(&d.pair_hash, (size_t)b->n_pairs + 1); pk.dalloc(&d.pair_fast, (size_t)b->n_pairs + 1); }
    pk.dalloc(&d.stats, (size_t)kStatSlots * kStatStride + 8);
    pk.dalloc(&b->d_hap_base, (size_t)H->n_haps + 1);
    // late traceback start (k_dp): packed int16 kernels only, no lane can wrap (a failed traceback is then impossible,
    // so a walk may stop once it has left the right flank), populate only, and only where the three extra scan launches do not show
    uint64_t late_min_pairs = 100000;
    { long long v; if (tune::number("OCT_PHMM_LATE_MIN_PAIRS", &v)) late_min_pairs = (uint64_t)v; }   // test hook (0 = always, a huge value = never)
    b->late_ok = b->fast_adds && !align_mode && !b->stream && !h->wide && b->n_pairs >= late_min_pairs;
    {
        // Window pairing: host-sized multi-slice batches (a haplotype's task runs are thousands long there: 100 k reads over ~240 (offset, strand) classes), packed int16 lanes with
        // plain adds, fast-cost flavour; the 20-byte columns must leave the traceback form its three workgroups per CU. OCT_PHMM_PAIRED=0 / 1: off / forced (tests: small batches).
        long long want = -1; tune::number("OCT_PHMM_PAIRED", &want);
        const uint32_t Bw = (uint32_t)h->band;
        const size_t lds_tr = dp_lds_bytes(b->t_cap, b->lh_cap, Bw, true, dp_rec_chunk(b->t_cap, b->lh_cap, Bw, true), true);
        const bool can = !h->wide && !b->stream && b->fast_adds && !align_mode && b->lh_cap <= kPairSortMaxLh && b->n_pairs > 0 && lds_tr <= rt::kMaxLdsBytes;
        b->pair_ok = can && (want >= 0 ? want != 0 : (b->n_pairs >= 4000000 && lds_tr * 3 <= rt::kMaxLdsBytes));
        if (b->pair_ok) for (int k = 0; k < 3; ++k) pk.dalloc(&b->d_paired_end[k], (size_t)H->n_haps + 1);
    }
    if (b->late_ok) {
        pk.dalloc(&b->d_pair_cnt_late, (size_t)b->n_pairs + oct_phmm_handle::kMaxSlices + 1); pk.dalloc(&b->d_hap_base_late, (size_t)H->n_haps + 1);
    }
    {
        // Slices of whole haplotypes, each on its own stream: while slice i is in its VALU-bound DP kernels, slice i+1 runs its
        // latency-bound mapper/classifier and slice i-1 its latency-bound walk. Small batches stay in one slice.
        // (at most four by default: every slice costs a front-end chain and a host read-back of its task counts, and round 6's sweep on the 12.8 M-pair step and the 2,000-region
        // stream put 4 ahead of 2, 3, 5, 6, 8, 10, 12 and 16 - 28.5 against 29.0 ms at 8; profiles/r06_slice_count_sweep.txt. OCT_PHMM_SLICES asks for up to kMaxSlices.)
        int n_slices = (int)std::min<uint64_t>(4, std::max<uint64_t>(1, b->n_pairs / 1000000));
        { long long v; if (tune::number("OCT_PHMM_SLICES", &v)) n_slices = std::max(1, std::min(oct_phmm_handle::kMaxSlices, (int)v)); }
        n_slices = (int)std::min<uint32_t>((uint32_t)n_slices, std::max<uint32_t>(1, H->n_haps));
        pk.dalloc(&b->d_totals, (size_t)n_slices);
        if (b->late_ok) pk.dalloc(&b->d_totals_late, (size_t)n_slices);
        b->slices.reserve((size_t)n_slices);                   // the packer keeps addresses of the slices' pointers
        uint32_t hap = 0;
        for (int i = 0; i < n_slices; ++i) {
            oct_phmm_batch::Slice sl;
            sl.hap0 = hap;
            const uint64_t target = b->n_pairs * (uint64_t)(i + 1) / (uint64_t)n_slices;
            while (hap < H->n_haps && (i == n_slices - 1 || hap_pair_off[hap + 1] <= target || hap == sl.hap0)) ++hap;
            if (i == n_slices - 1) hap = H->n_haps;
            sl.hap1 = hap;
            sl.pair0 = hap_pair_off[sl.hap0]; sl.pair1 = hap_pair_off[sl.hap1]; sl.out0 = hap_out_off[sl.hap0]; sl.out1 = hap_out_off[sl.hap1];
            sl.n_tiles = (uint32_t)((sl.pair1 - sl.pair0 + 1 + kScanLocalTile - 1) / kScanLocalTile);
            RT(h->get_event(&sl.done));
            sl.blk0 = (uint32_t)(std::lower_bound(b->h_blk_hap.begin(), b->h_blk_hap.end(), sl.hap0) - b->h_blk_hap.begin());
            sl.blk1 = (uint32_t)(std::lower_bound(b->h_blk_hap.begin(), b->h_blk_hap.end(), sl.hap1) - b->h_blk_hap.begin());
            if (b->dedup) {                                       // the slice's haplotypes region by region, the region's reads in tiles of 64
                sl.seg0 = (uint32_t)b->h_segs.size();
                for (uint32_t hp = sl.hap0; hp < sl.hap1;) {
                    const uint32_t g = hap_region[hp];
                    uint32_t e = hp; while (e < sl.hap1 && hap_region[e] == g) ++e;
                    const uint32_t nreads = first_read(g_row[g + 1]) - reg_read0[g];
                    // a region cut by a slice border: its reads carry their tables over (a single haplotype on one side still takes part)
                    const uint32_t resumes = hp > g_hap[g] ? 1u : 0u, continues = e < g_hap[g + 1] ? 1u : 0u;
                    if ((e - hp >= 2 || resumes || continues) && nreads) {
                        b->h_segs.push_back(DedupSeg {g, g_hap[g], hp, e, reg_read0[g], nreads, sl.n_seg_tiles, resumes, continues}); sl.n_seg_tiles += (nreads + 63) / 64;
                        if (resumes) sl.resumes = true;
                        if (resumes || continues) b->dedup_tables = true;
                    }
                    hp = e;
                }
                sl.n_segs = (uint32_t)b->h_segs.size() - sl.seg0;
                RT(h->get_event(&sl.matched));
            }
            b->slices.push_back(sl);
            const size_t n_tile_sums = std::max<size_t>((size_t)sl.n_tiles, (size_t)((sl.pair1 - sl.pair0 + 1 + kScanLocalTile - 1) / kScanLocalTile)) + 1;   // (256-pair tiles of the workgroup-local scan)
            pk.dalloc(&b->slices.back().tile_sums, n_tile_sums);
            if (b->late_ok) pk.dalloc(&b->slices.back().tile_sums_late, n_tile_sums);
        }
    }
    {
        // Device-sized launches: with ONE slice and traceback scratch for the most tasks the pairs can emit (max_mapping_positions + 1 each, plus the
        // padding of every haplotype's runs), nothing on the host depends on the task counts: the step is a fixed launch sequence without a read-back.
        const uint32_t Bw = (uint32_t)h->band, Gs = b->stream ? (Bw < 64 ? 64u / Bw : 1u) : (h->wide ? 1u : 2u) * (64u / Bw);
        const uint64_t raw = b->n_pairs * (uint64_t)(S + 1), pad = (uint64_t)H->n_haps * (Gs - 1);
        const uint64_t list_bound = (raw + pad + Gs - 1) / Gs * Gs, total_bound = raw + 6 * pad;
        // Traceback scratch is provisioned for two traceback tasks per pair, not for the bound of eleven (a 300 x 24 region: 86 MB instead of 475 MB per
        // handle; this generator's regions need 0.9): the scan flags a batch that needs more and oct_phmm_batch_wait repeats it host-sized.
        long long per_pair = 2; tune::trace_per_pair(&per_pair);
        const uint64_t trace_cap = per_pair < 0 ? Gs : std::min<uint64_t>(list_bound, (b->n_pairs * (uint64_t)per_pair + pad + Gs - 1) / Gs * Gs + Gs);   // (negative: one task group, test hook)
        const uint64_t bp_bytes = trace_cap / Gs * ((uint64_t)bp_tiles(b->t_cap, Bw) * 4096u * (b->stream ? (uint64_t)h->lanes_c : 1u));
        const uint64_t cap = std::min<uint64_t>((uint64_t)8 << 30, h->bp_budget);
        b->dsl = b->slices.size() == 1 && dsl_wanted && b->n_pairs > 0 && bp_bytes <= cap && list_bound < 0x7fffffffull;
        b->dsl_list_bound = b->dsl ? (uint32_t)list_bound : 0; b->dsl_total_bound = b->dsl ? (size_t)total_bound : 0; b->dsl_trace_cap = b->dsl ? (uint32_t)trace_cap : 0;
        if (b->dsl && (want_dirty || (flavour_hint >= 0 && !gen_device)) && !d.wide) {
            // which of the two cost flavours can occur at all (k_hap_tables / read_flags_thread decide per read and haplotype): a clean region launches no generic kernels
            b->dsl_flavours = dirty ? 3 : 1;
        } else b->dsl_flavours = d.wide ? 2 : 3;            // bit 0: fast-cost lists may hold tasks, bit 1: generic lists may
    }
    if (b->dedup && !b->h_segs.empty()) pk.upload(b->h_segs.data(), b->h_segs.size(), (const DedupSeg**)&b->d_segs);
    if (b->dedup_tables) { pk.dalloc(&d.dd_hash, (size_t)kDedupReps * R->n_reads + 1); pk.dalloc(&d.dd_hap, (size_t)kDedupReps * R->n_reads + 1); pk.dalloc(&d.dd_n, (size_t)R->n_reads + 1); }
    pk.dalloc(&b->d_out, (size_t)b->n_out);
    d.align_mode = align_mode ? 1 : 0; d.pair_key = nullptr;
    if (align_mode) {
        if (R->row_offsets) return fail(status, OCT_PHMM_EINVAL, "alignments are per read: row_offsets must be NULL");
        b->align_mode = true;
        b->cig_cap = (uint32_t)std::min<uint64_t>(max_cigar_ops, 2ull * (b->t_cap + (uint32_t)h->band) + 1);   // an alignment has at most 2 (T + B) columns
        pk.dalloc(&d.pair_key, (size_t)b->n_pairs); pk.dalloc(&b->d_aln_lik, (size_t)b->n_pairs); pk.dalloc(&b->d_aln_mpos, (size_t)b->n_pairs);
        pk.dalloc(&b->d_aln_n, (size_t)b->n_pairs); pk.dalloc(&b->d_aln_ops, (size_t)b->n_pairs * b->cig_cap); pk.dalloc(&b->d_err_flags, 4);
    }
    std::vector<uint32_t> ones(H->n_haps + 1, 1u);
    pk.upload(ones.data(), (size_t)H->n_haps, (const uint32_t**)&d.hclean);
    RT(h->get_event(&b->ev_fork)); RT(h->get_event(&b->ev_join)); RT(h->get_event(&b->ev_hashes));
    if (up_prof) t_up1 = now_ms();
    RT(pk.commit(h, bp, s));
    if (up_prof) t_up2 = now_ms();
    d.err_key = d.stats + (size_t)kStatSlots * kStatStride; d.dsl_overflow = d.err_key + 1; d.dsl_trace_cap = b->dsl_trace_cap;
    for (size_t i = 0; i < b->slices.size(); ++i) {
        b->slices[i].cnt = d.pair_cnt + b->slices[i].pair0 + i;      // each slice owns pair1 - pair0 + 1 scan entries
        b->slices[i].d_totals = b->d_totals + i;
        if (b->late_ok) { b->slices[i].cnt_late = b->d_pair_cnt_late + b->slices[i].pair0 + i; b->slices[i].d_totals_late = b->d_totals_late + i; }
    }
    if (positions) {
        RT(rt::h2d(d.pos, h_pos.data(), (size_t)b->n_pairs * S * sizeof(uint32_t), s));
        RT(rt::h2d(d.npos, h_npos.data(), (size_t)b->n_pairs, s));
    }
    if (gen_device) {
        // HaplotypeLikelihoodModel::reset for every haplotype on the device; a haplotype whose run lists outgrow the fixed workspace
        // (pathological repeats) is redone on the host
        if (!h->d_model) {
            void* p = nullptr; RT(h->pool.alloc(&p, sizeof(oct_phmm_error_model))); h->d_model = (oct_phmm_error_model*)p;
            RT(rt::h2d(h->d_model, &h->model, sizeof(oct_phmm_error_model), s)); RT(rt::stream_sync(s));
        }
        void* ovf = nullptr; RT(h->pool.alloc(&ovf, ((size_t)H->n_haps + 1) * 4));
        RT(rt::dev_memset(ovf, 0, ((size_t)H->n_haps + 1) * 4, s));
        const PenaltyOut po {(int8_t*)d.go, (int8_t*)d.ge, (uint8_t*)d.maskF, (int8_t*)d.priorF, (uint8_t*)d.maskR, (int8_t*)d.priorR};
        const size_t lds_words = em::workspace_words(b->lh_cap, 0) + (b->lh_cap + 3) / 4;
        if (lds_words * 4 <= kPenaltyLdsBytes && !tune::penalties_lane_kernel()) {
            void* prof = nullptr;
            if (tune::penalties_report()) { RT(h->pool.alloc(&prof, 16 * 8)); RT(rt::dev_memset(prof, 0, 16 * 8, s)); }
            // one wave per haplotype, workspace in LDS (at least two waves per CU)
            OCT_LAUNCH(k_penalty_vectors_wave, H->n_haps, 64, lds_words * 4, s, (const oct_phmm_error_model*)h->d_model, d.hbases, d.hoff, H->n_haps, d_sub_mask,
                       (uint32_t)lds_words, po, (uint32_t*)ovf, (unsigned long long*)prof);
            RT(rt::launch_ok());
            if (prof) {
                unsigned long long t[16];
                RT(rt::d2h(t, prof, sizeof t, s)); RT(rt::stream_sync(s));
                h->pool.release(prof);
                fprintf(stderr, "oct_phmm: k_penalty_vectors_wave lane-0 clocks per haplotype by phase:");
                for (int k = 1; k <= 12; ++k) fprintf(stderr, " %d:%llu", k, t[k] / std::max<uint32_t>(1, H->n_haps));
                fprintf(stderr, "\n");
            }
        } else {
            // long haplotypes: one lane per haplotype, its workspace in HBM, in chunks that keep the workspace below 2 GB
            const size_t words = em::workspace_words(b->lh_cap, 1);
            const uint32_t chunk = (uint32_t)std::max<size_t>(256, std::min<size_t>(H->n_haps, (((size_t)2 << 30) / (words * 4)) / 256 * 256));
            void* ws = nullptr; RT(h->pool.alloc(&ws, (size_t)chunk * words * 4));
            for (uint32_t h0 = 0; h0 < H->n_haps; h0 += chunk) {
                const uint32_t h1 = std::min<uint32_t>(H->n_haps, h0 + chunk);
                OCT_LAUNCH(k_penalty_vectors, (h1 - h0 + 63) / 64, 64, 0, s, (const oct_phmm_error_model*)h->d_model, d.hbases, d.hoff, h0, h1, d_sub_mask,
                           (uint32_t*)ws, words, po, (uint32_t*)ovf);
                RT(rt::launch_ok());
            }
            RT(rt::stream_sync(s));
            h->pool.release(ws);
        }
        std::vector<uint32_t> flags(H->n_haps);
        RT(rt::d2h(flags.data(), ovf, (size_t)H->n_haps * 4, s)); RT(rt::stream_sync(s));
        h->pool.release(ovf);
        if (tune::penalties_report()) {
            size_t redo = 0; for (uint32_t f : flags) redo += f;
            fprintf(stderr, "oct_phmm: penalty vectors of %u haplotypes on the device (LDS words %zu), %zu redone on the host\n", H->n_haps, lds_words, redo);
        }
        std::vector<uint32_t> w;
        for (uint32_t hp = 0; hp < H->n_haps; ++hp) if (flags[hp]) {
            const uint32_t o = H->offsets[hp], n = H->offsets[hp + 1] - o;
            std::vector<int8_t> go(n), ge(n), pf(n), pr(n); std::vector<uint8_t> mf(n), mr(n);
            host_penalty_vectors_one(h->model, (const uint8_t*)H->bases + o, n, sub_mask ? sub_mask + o : nullptr, w, PenaltyOut {go.data(), ge.data(), mf.data(), pf.data(), mr.data(), pr.data()}, 0);
            RT(rt::h2d((void*)(d.go + o), go.data(), n, s)); RT(rt::h2d((void*)(d.ge + o), ge.data(), n, s)); RT(rt::h2d((void*)(d.maskF + o), mf.data(), n, s));
            RT(rt::h2d((void*)(d.priorF + o), pf.data(), n, s)); RT(rt::h2d((void*)(d.maskR + o), mr.data(), n, s)); RT(rt::h2d((void*)(d.priorR + o), pr.data(), n, s));
            RT(rt::stream_sync(s));
        }
    }
    // per-read flags and per-base DP tables (once per batch; HaplotypeLikelihoodModel::reset analogue)
    {
        const uint32_t table_blocks = (n_hap_bases + 255) / 256, flag_blocks = (R->n_reads + 3) / 4;      // tables: a thread per base; flags: a wave per read
        const uint64_t rec_blocks64 = (d.rrec || d.rrecW) ? ((uint64_t)R->n_reads * d.rrec_stride + 255) / 256 : 0;
        if (table_blocks + flag_blocks + rec_blocks64 >= 0x7fffffffull) return fail(status, OCT_PHMM_EUNSUPPORTED, "batch too large for one table launch");
        const uint32_t rec_blocks = (uint32_t)rec_blocks64;
        if (table_blocks + flag_blocks + rec_blocks) {
            OCT_LAUNCH(k_hap_tables, table_blocks + flag_blocks + rec_blocks, 256, 0, s, d, n_hap_bases, table_blocks, flag_blocks); RT(rt::launch_ok());
            b->stats_clear = true;                             // (the kernel's last workgroup cleared the counters)
        }
    }
    if (b->dedup) {
        // canonical band windows: polynomial prefix sums per haplotype, a hash table from window key to the first window with that key,
        // then every window is compared byte by byte with the table's (phmm_kernels.hpp)
        if (h->pw_n < (size_t)b->lh_cap + 2) {                       // powers of the hash base and of its inverse mod 2^64, up to the longest haplotype seen
            const uint64_t base = 0x9e3779b97f4a7c15ull;              // odd: invertible
            uint64_t inv = base; for (int it = 0; it < 6; ++it) inv *= 2 - base * inv;   // Newton: inv * base == 1 (mod 2^64)
            size_t n = 1024; while (n < (size_t)b->lh_cap + 2) n <<= 1;
            std::vector<uint64_t> pw(n), pwinv(n);
            pw[0] = 1; pwinv[0] = 1;
            for (size_t i = 1; i < n; ++i) { pw[i] = pw[i - 1] * base; pwinv[i] = pwinv[i - 1] * inv; }
            RT(rt::stream_sync(s));                                  // earlier uploads on this stream may still read the old tables
            h->pool.release(h->d_pw); h->pool.release(h->d_pwinv); h->d_pw = h->d_pwinv = nullptr; h->pw_n = 0;
            void* p1 = nullptr; void* p2 = nullptr; RT(h->pool.alloc(&p1, n * 8)); RT(h->pool.alloc(&p2, n * 8));
            h->d_pw = (uint64_t*)p1; h->d_pwinv = (uint64_t*)p2;
            RT(rt::h2d(h->d_pw, pw.data(), n * 8, s)); RT(rt::h2d(h->d_pwinv, pwinv.data(), n * 8, s)); RT(rt::stream_sync(s));
            h->pw_n = n;
        }
        if (tune::window_lds()) {
            // keys, table and candidates of a region in ONE workgroup with the table in LDS (k_window_region), then the confirmation by runs
            const size_t n_prefix = ((size_t)n_hap_bases + H->n_haps + 2) & ~(size_t)1;
            b->h_win_blocks.clear();                                                             // one workgroup per (region, class of its keys)
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t np = window_passes((uint64_t)H->offsets[g_hap[g + 1]] - H->offsets[g_hap[g]]);
                for (uint32_t p = 0; p < np; ++p) { b->h_win_blocks.push_back(g); b->h_win_blocks.push_back(p); }
            }
            const size_t n_wblk = b->h_win_blocks.size() / 2;
            const size_t need = (n_prefix + n_wblk + 1) * 8 + ((size_t)G + 2) * 4 + 64;
            if (h->dedup_scratch_bytes < need) {
                RT(rt::stream_sync(s));
                h->pool.release(h->dedup_scratch); h->dedup_scratch = nullptr; h->dedup_scratch_bytes = 0;
                RT(h->pool.alloc(&h->dedup_scratch, need + need / 4)); h->dedup_scratch_bytes = need + need / 4;
            }
            uint64_t* d_prefix = (uint64_t*)h->dedup_scratch;
            uint2* d_win_blocks = (uint2*)(d_prefix + n_prefix); uint32_t* d_reg_hap0 = (uint32_t*)(d_win_blocks + n_wblk + 1);
            RT(rt::h2d(d_reg_hap0, b->h_reg_hap0.data(), ((size_t)G + 1) * 4, s));             // (both live as long as the batch)
            if (n_wblk) RT(rt::h2d(d_win_blocks, b->h_win_blocks.data(), n_wblk * 8, s));
            OCT_LAUNCH(k_window_prefix, (H->n_haps + 3) / 4, 256, 0, s, d, (const uint64_t*)h->d_pw, d_prefix); RT(rt::launch_ok());   // one wave per haplotype
            if (n_hap_bases && n_wblk) {
                const size_t lds = window_region_lds_bytes();
                RT(rt::allow_lds(k_window_region, lds));
                OCT_LAUNCH(k_window_region, (uint32_t)n_wblk, kWinThreads, lds, s, d, (const uint64_t*)h->d_pwinv, (const uint64_t*)d_prefix, (const uint32_t*)d_reg_hap0, (const uint2*)d_win_blocks); RT(rt::launch_ok());
                OCT_LAUNCH(k_window_confirm, (H->n_haps + 3) / 4, 256, 0, s, d); RT(rt::launch_ok());                                  // one wave per haplotype
            }
        } else {
        // one table per region (phmm_kernels.hpp, k_window_insert): a power of two of slots >= 1.25 x the region's windows, one behind the other
        std::vector<uint32_t> tab_base(G + 1, 0), tab_mask(G + 1, 0);
        size_t tsize = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const size_t w = (size_t)H->offsets[g_hap[g + 1]] - H->offsets[g_hap[g]];
            size_t n = 16; while (n < w + w / 4) n <<= 1;
            tab_base[g] = (uint32_t)tsize; tab_mask[g] = (uint32_t)(n - 1); tsize += n;
        }
        if (tsize >= 0xffffffffull) { b->dedup = false; d.canon = nullptr; }       // (more than 2^32 table slots: no sharing for this batch)
        tab_base[G] = (uint32_t)tsize;
      if (b->dedup) {
        b->h_tab_base = std::move(tab_base); b->h_tab_mask = std::move(tab_mask);           // (the copies below read them: they live as long as the batch)
        const size_t n_wblk = ((size_t)n_hap_bases + 255) / 256;
        b->h_tab_base.reserve((size_t)G + 1 + n_wblk);                                      // behind the bases: the haplotype of every 256-window workgroup's first window
        for (size_t blk = 0, hp = 0; blk < n_wblk; ++blk) { while (hp + 1 < H->n_haps && H->offsets[hp + 1] <= blk * 256) ++hp; b->h_tab_base.push_back((uint32_t)hp); }
        const size_t n_tab = ((size_t)G + 2 + n_wblk) & ~(size_t)1;
        const size_t n_prefix = ((size_t)n_hap_bases + H->n_haps + 2) & ~(size_t)1, n_wkey = ((size_t)n_hap_bases + 2) & ~(size_t)1;
        const size_t need = (n_prefix + n_wkey + tsize) * 8 + tsize * 4 + 2 * n_tab * 4 + 64;
        if (h->dedup_scratch_bytes < need) {
            RT(rt::stream_sync(s));
            h->pool.release(h->dedup_scratch); h->dedup_scratch = nullptr; h->dedup_scratch_bytes = 0;
            RT(h->pool.alloc(&h->dedup_scratch, need + need / 4)); h->dedup_scratch_bytes = need + need / 4;
        }
        uint64_t* d_prefix = (uint64_t*)h->dedup_scratch; unsigned long long* d_wkey = (unsigned long long*)(d_prefix + n_prefix);
        unsigned long long* d_tkeys = d_wkey + n_wkey; uint32_t* d_tvals = (uint32_t*)(d_tkeys + tsize);
        uint32_t* d_tab_base = d_tvals + ((tsize + 1) & ~(size_t)1); uint32_t* d_tab_mask = d_tab_base + n_tab;
        RT(rt::h2d(d_tab_base, b->h_tab_base.data(), b->h_tab_base.size() * 4, s)); RT(rt::h2d(d_tab_mask, b->h_tab_mask.data(), ((size_t)G + 1) * 4, s));
        const uint32_t* d_blk_hap_w = d_tab_base + G + 1;
        RT(rt::dev_memset(d_tkeys, 0, tsize * 8, s)); RT(rt::dev_memset(d_tvals, 0xff, tsize * 4, s));
        OCT_LAUNCH(k_window_prefix, (H->n_haps + 3) / 4, 256, 0, s, d, (const uint64_t*)h->d_pw, d_prefix); RT(rt::launch_ok());   // one wave per haplotype
        if (n_hap_bases) {
            for (int phase = 0; phase < 2; ++phase) {                       // every region's first haplotype, then the rest (k_window_insert)
                OCT_LAUNCH(k_window_insert, (n_hap_bases + 255) / 256, 256, 0, s, d, (const uint64_t*)h->d_pwinv, (const uint64_t*)d_prefix, n_hap_bases,
                           d_wkey, d_tkeys, d_tvals, (const uint32_t*)d_tab_base, (const uint32_t*)d_tab_mask, d_blk_hap_w, phase); RT(rt::launch_ok());
            }
            OCT_LAUNCH(k_window_candidate, (n_hap_bases + 255) / 256, 256, 0, s, d, n_hap_bases, (const unsigned long long*)d_wkey,
                       (const unsigned long long*)d_tkeys, (const uint32_t*)d_tvals, (const uint32_t*)d_tab_base, (const uint32_t*)d_tab_mask, d_blk_hap_w); RT(rt::launch_ok());
            OCT_LAUNCH(k_window_confirm, (H->n_haps + 3) / 4, 256, 0, s, d); RT(rt::launch_ok());                                          // one wave per haplotype
        }
      }
        }
    }
    // The copies above read this call's host-side staging (pinned buffer, position vectors): a caller of the split API may upload the next batch
    // right away, so they must have landed. A one-shot call (populate, align) runs on the same stream at once and does not return before its
    // results are back, which covers the pinned buffer; it only waits here when it brought pageable position arrays.
    if (!one_shot || positions) RT(rt::stream_sync(s));
    if (up_prof) fprintf(stderr, "{\"upload_profile_ms\": {\"validate_and_tables\": %.2f, \"pack_and_copy\": %.2f, \"kernels_enqueue%s\": %.2f, \"input_MB\": %.1f}}\n",
                         t_up1 - t_up0, t_up2 - t_up1, (!one_shot || positions) ? "_and_wait" : "", now_ms() - t_up2, (double)pk.in_bytes / 1e6);
    *out = b.release();
    return ok(status);
}

// ---------------------------------------------------------------------------------------------------------------
// run
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_batch_run(oct_phmm_handle* h, oct_phmm_batch* b, oct_phmm_status* status)
{
    if (!h || !b || b->owner != h) return fail(status, OCT_PHMM_EINVAL, "bad handle/batch");
    rt::Range range_("oct_phmm run");
    RT(rt::set_device(h->cfg.device_id));
    rt::Stream s0 = h->stream;
    DevBatch& d = b->d;
    for (auto& t : b->timers) { h->put_event(t.first); h->put_event(t.second); }
    b->timers.clear(); b->timer_kind.clear(); b->dp_ms = 0; b->dp_launches = 0; b->ran = false;
    const uint32_t G = b->stream ? (h->band < 64 ? 64u / (uint32_t)h->band : 1u) : (h->wide ? 1u : 2u) * (64 / (uint32_t)h->band);
    const int S = (int)b->slices.size();
    if (b->dsl && b->dsl_trace_cap) {     // the scratch of the device-sized launches, before anything is enqueued: without it the batch simply runs host-sized (chunked if need be)
        const size_t per_group = (size_t)bp_tiles(b->t_cap, (uint32_t)h->band) * 4096 * (b->stream ? (size_t)h->lanes_c : 1);
        if (!ensure_bp(h, 0, (size_t)b->dsl_trace_cap / G * per_group)) b->dsl = false;
    }
    d.dsl_trace_cap = b->dsl ? b->dsl_trace_cap : 0;
    constexpr size_t kStatWords = (size_t)kStatSlots * kStatStride + 2;   // counters + the inverted error key + the overflow flag, one copy
    b->h_stat_stripes.assign(kStatWords, 0);
    if (!b->stat_stage) b->stat_stage = (unsigned long long*)h->get_stat_stage(kStatWords * sizeof(unsigned long long));
    // A one-shot region-sized call (oct_phmm_populate set early_out; one slice): no copy behind the last kernel. The epilogue stores the results into the pinned
    // landing zone itself (mapped into the device) and leaves the sums of the counter stripes beside them; the host waits once.
    const bool mapped_out = b->early_out && !b->out_landing && S == 1 && !b->align_mode && b->n_out <= kHostMappedOutMax && b->stat_stage && tune::host_mapped();
    const uint32_t mapped_stripes = tune::map_stats() ? kStatSlots : (uint32_t)std::min<uint64_t>(kStatSlots, (b->n_pairs + 255) / 256);   // (k_classify's workgroups own the counters; the mapper's only with OCT_PHMM_MAP_STATS)
    if (mapped_out) memset(b->stat_stage, 0, kStatWords * sizeof(unsigned long long));
    if (!b->stats_clear) RT(rt::dev_memset(d.stats, 0, ((size_t)kStatSlots * kStatStride + 2) * sizeof(unsigned long long), s0));   // counters + the (inverted) error key + the overflow flag behind them
    b->stats_clear = false;
    if (b->align_mode) RT(rt::dev_memset(b->d_err_flags, 0, 16, s0));
    if (b->dedup) RT(rt::dev_memset(d.pair_rep, 0xff, (size_t)b->n_pairs * sizeof(uint32_t), s0));      // kNoPair: every pair is computed itself until k_dedup_verify says otherwise
    for (int k = 0; k < kNumKinds; ++k) b->n_tasks[k] = 0;
    // (slice 0 stays on the handle's high-priority stream: with ALL slices on normal-priority streams the single step is 0.2-0.3 ms faster and two calls in flight fall from 0.92-0.95 x to 0.88-0.92 x the resident rate, profiles/r05_priority_big_batches.md)
    const int first_aside = 1;    // the first slice that runs on a stream other than the handle's own
    if (S > 1) {
        RT(rt::event_record(h->ev_ready, s0));
        for (int i = first_aside; i < S; ++i) RT(rt::stream_wait_event(h->slice_stream(i), h->ev_ready));
    }

    // phase 1 of a slice: candidate mapping, classification + scalar fast path, task counts -> slot offsets (everything up to the one
    // host read-back that sizes the DP launches)
    int hash_slice = -1;                                  // the slice whose table launch also hashed the reads
    auto phase1 = [&](int i) -> int {
        rt::Range range_p1("slice phase 1: map, classify, scan");
        oct_phmm_batch::Slice& sl = b->slices[i];
        rt::Stream s = h->slice_stream(i);
        const uint64_t np = sl.pair1 - sl.pair0;
        if (!np) { sl.totals = make_uint4(0, 0, 0, 0); sl.totals_late = make_uint4(0, 0, 0, 0); return OCT_PHMM_OK; }
        if (b->device_map) {                              // HaplotypeLikelihoodArray::populate maps per haplotype (array.cpp:118-158)
            // the first slice that has pairs also hashes every read of the batch once (array.cpp:118-131); the later slices' mappers wait for it
            const bool hashes_here = hash_slice < 0;
            const uint32_t hash_blocks = hashes_here ? (b->n_reads + 3) / 4 : 0;        // one wave per read
            OCT_LAUNCH(k_kmer_tables, sl.hap1 - sl.hap0 + hash_blocks, 256, (kKmerBins + 256) * sizeof(uint32_t), s, d, sl.hap0, sl.hap1 - sl.hap0); RT(rt::launch_ok());
            if (hashes_here) { hash_slice = i; if (S > 1) RT(rt::event_record(b->ev_hashes, s)); }
            else RT(rt::stream_wait_event(s, b->ev_hashes));
            if (b->map_big) {
                const size_t lds = (size_t)b->lh_cap * 4 + 64;
                if (lds > 64 * 1024) RT(rt::allow_lds(k_kmer_map_big, lds));
                OCT_LAUNCH(k_kmer_map_big, (uint32_t)np, 256, lds, s, d, sl.pair0); RT(rt::launch_ok());
            } else if (sl.blk1 > sl.blk0 && b->map_lanes) {
                const size_t lds = kmer_map_lanes_lds_bytes(b->lh_cap);
                if (lds > 64 * 1024) RT(rt::allow_lds(k_kmer_map_lanes, lds));
                OCT_LAUNCH(k_kmer_map_lanes, sl.blk1 - sl.blk0, kLaneMapThreads, lds, s, d, (const uint32_t*)b->d_blk_hap + sl.blk0, (const uint32_t*)b->d_blk_read0 + sl.blk0, b->lh_cap);
                RT(rt::launch_ok());
            } else if (sl.blk1 > sl.blk0) {
                const size_t lds = kmer_map_lds_bytes(b->lh_cap);
                const uint32_t nq_cap = b->t_cap >= kKmer ? b->t_cap - kKmer + 1 : 0;
                if (nq_cap <= 192) {                          // reads up to 197 bases: three 64-lane rounds hold a read's k-mers
                    if (lds > 64 * 1024) RT(rt::allow_lds(k_kmer_map<3>, lds));
                    OCT_LAUNCH(k_kmer_map<3>, sl.blk1 - sl.blk0, kBlockWaves * 64, lds, s, d, (const uint32_t*)b->d_blk_hap + sl.blk0,
                               (const uint32_t*)b->d_blk_read0 + sl.blk0, b->lh_cap, b->map_reads_per_block);
                } else {
                    if (lds > 64 * 1024) RT(rt::allow_lds(k_kmer_map<4>, lds));
                    OCT_LAUNCH(k_kmer_map<4>, sl.blk1 - sl.blk0, kBlockWaves * 64, lds, s, d, (const uint32_t*)b->d_blk_hap + sl.blk0,
                               (const uint32_t*)b->d_blk_read0 + sl.blk0, b->lh_cap, b->map_reads_per_block);
                }
                RT(rt::launch_ok());
            }
        }
        // The scan of the task counts starts in the kernel that makes them: its workgroups store tile-local prefixes and tile totals (k_scan_finish does the rest in
        // one workgroup). The grid then covers pair1 itself, the scan's extra entry.
        const uint64_t n_scan = np + 1;
        sl.scan_fused = true;                                 // (the scan that starts in the classifier; round 4's chain of scan launches was retired in round 6)
        const bool verify_runs = b->dedup && sl.n_seg_tiles;
        const uint32_t pair_blocks = (uint32_t)(((sl.scan_fused ? n_scan : np) + 255) / 256);
        uint4* const ts = sl.scan_fused ? sl.tile_sums : nullptr; uint4* const ts_late = sl.scan_fused ? sl.tile_sums_late : nullptr;
        OCT_LAUNCH(k_classify, pair_blocks, 256, 0, s, d, sl.pair0, sl.pair1, sl.cnt, sl.cnt_late, verify_runs ? nullptr : ts, verify_runs ? nullptr : ts_late); RT(rt::launch_ok());
        if (verify_runs) {                                    // pairs whose candidates equal an earlier pair's of the same read drop their tasks
            if (sl.resumes && i > 0) RT(rt::stream_wait_event(s, b->slices[i - 1].matched));     // its reads' tables and the earlier pairs' classes
            OCT_LAUNCH(k_dedup_match, sl.n_seg_tiles, 64, (size_t)kDedupSlots * 64 * (sizeof(uint32_t) + sizeof(uint16_t)), s, d, (const DedupSeg*)b->d_segs + sl.seg0, sl.n_segs); RT(rt::launch_ok());
            OCT_LAUNCH(k_dedup_verify, pair_blocks, 256, 0, s, d, sl.pair0, sl.pair1, sl.cnt, sl.cnt_late, ts, ts_late); RT(rt::launch_ok());
        }
        if (b->dedup && S > 1) RT(rt::event_record(sl.matched, s));   // the next slice's matcher may resume a region of this one
        {                                                     // any size: tile prefixes, haplotype bases and totals of both count arrays in ONE single-workgroup launch
            OCT_LAUNCH(k_scan_finish, sl.cnt_late ? 2 : 1, kHapBaseThreads, 16 * sizeof(uint4), s, d, sl.hap0, sl.hap1, (const uint4*)sl.cnt, (const uint4*)sl.cnt_late, sl.pair0, pair_blocks,
                       sl.tile_sums, sl.tile_sums_late, b->d_hap_base, b->d_hap_base_late, sl.d_totals, sl.d_totals_late, G); RT(rt::launch_ok());
            sl.totals_late = make_uint4(0, 0, 0, 0);
            if (!b->dsl) { RT(rt::d2h(&sl.totals, sl.d_totals, sizeof(uint4), s)); if (sl.cnt_late) RT(rt::d2h(&sl.totals_late, sl.d_totals_late, sizeof(uint4), s)); }
            return OCT_PHMM_OK;
        }
    };
    // phase 2 of the one slice of a device-sized batch: the same launches with grids from the host's bound; the kernels find their task lists through
    // the totals k_hap_bases left in device memory
    auto phase2_device_sized = [&]() -> int {
        rt::Range range_p2("phase 2 (device-sized): emit, DP, walk, epilogue");
        oct_phmm_batch::Slice& sl = b->slices[0];
        rt::Stream s = h->slice_stream(0);
        const uint64_t np = sl.pair1 - sl.pair0;
        if (np) {
            if (b->dsl_total_bound > sl.tasks_cap) {
                h->pool.release(sl.d_tasks); sl.d_tasks = nullptr; sl.tasks_cap = 0;
                void* p = nullptr; RT(h->pool.alloc(&p, b->dsl_total_bound * sizeof(DevTask))); sl.d_tasks = (DevTask*)p; sl.tasks_cap = b->dsl_total_bound;
            }
            if (b->dsl_trace_cap > sl.ends_cap) {
                h->pool.release(sl.d_ends); sl.d_ends = nullptr; sl.ends_cap = 0;
                void* p = nullptr; RT(h->pool.alloc(&p, (size_t)b->dsl_trace_cap * sizeof(TraceEnd))); sl.d_ends = (TraceEnd*)p; sl.ends_cap = b->dsl_trace_cap;
            }
            TaskArrays ta {}; ta.t[0] = sl.d_tasks; TaskArrays tl {};
            TaskListRef ref {sl.d_totals, sl.cnt_late ? sl.d_totals_late : nullptr, 0, d.dsl_overflow};
            const bool join = sl.scan_fused && tune::join_late() != 0;   // a flavour's traceback and late-start lists in one launch (k_scan_finish checked that BOTH fit the scratch)
            OCT_LAUNCH(k_emit, (uint32_t)((np + 255) / 256), 256, 0, s, d, sl.pair0, sl.pair1, (const uint4*)sl.cnt, (const uint4*)b->d_hap_base, ta,
                       (const uint4*)sl.cnt_late, (const uint4*)b->d_hap_base_late, tl, ref, G,
                       sl.scan_fused ? (const uint4*)sl.tile_sums : nullptr, sl.scan_fused ? (const uint4*)sl.tile_sums_late : nullptr); RT(rt::launch_ok());
            // Region-sized and latency-bound: the score-only DP runs on a second stream beside the traceback DP (a region's two lists together are about one
            // wave per SIMD; OCT_PHMM_DSL_FORK_EARLY=0: beside the traceback WALK instead, as round 2's lockstep walker wanted it).
            rt::Stream aux = h->slice_stream(1);
            auto flavour_live = [&](int list) { const bool gen = list == kScoreGen || list == kTraceGen || list == 5; return (b->dsl_flavours & (gen ? 2 : 1)) != 0; };
            // ... unless both fit ONE launch (k_dp_pair: packed int16 lanes, LDS-resident kernels): the score-only list of a flavour rides with that flavour's first
            // traceback launch, no second stream, no events.
            // Measured on one box, three interleaved repetitions each (profiles/r03_step7_dp_launch_forms_ab.log): one 300 x 24 region per call 0.242 ms merged, 0.250 forked
            // early, 0.280 forked after the traceback DP; 16 callers on the region server (3.5 - 5 regions per device batch) 11.6 k / 13.7 k / 12.3 k regions/s - in the
            // merged launch the score-only workgroups hold the traceback form's LDS and registers, which costs occupancy once a batch fills the chip. So: merged up
            // to kDslMergeMaxPairs pairs, two launches side by side beyond.
            const int want_merge = tune::dsl_merge_dp();
            const bool merge = (want_merge >= 0 ? want_merge != 0 : np <= kDslMergeMaxPairs) && !h->wide && !b->stream && !b->multi_wave && !b->align_mode;
            bool forked = false, score_done[2] = {false, false};
            if (!merge) { RT(rt::event_record(b->ev_fork, s)); forked = true; }     // (long reads: see the host-sized path)
            for (int list : {4, 5, (int)kTraceFast, (int)kTraceGen}) {
                if (list >= 4 && (!sl.cnt_late || join)) continue;
                if (!flavour_live(list)) continue;
                ref.list = list; ref.join_late = (join && sl.cnt_late) ? 1 : 0;
                const int fl = (list == kTraceGen || list == 5) ? 1 : 0;
                const bool ride = merge && !score_done[fl];
                const int rc = run_dp_kind(h, b, 0, list == 4 ? kTraceFast : list == 5 ? kTraceGen : list, sl.d_tasks, b->dsl_trace_cap, sl.d_ends, h->cfg.nuc_prior, nullptr, status,
                                           nullptr, list >= 4, ref, (merge || forked) ? nullptr : &b->ev_fork, ride ? (fl ? (int)kScoreGen : (int)kScoreFast) : -1, b->dsl_list_bound);
                if (rc != OCT_PHMM_OK) return rc;
                forked = true; if (ride) score_done[fl] = true;
            }
            if (!(merge && score_done[0] == ((b->dsl_flavours & 1) != 0) && score_done[1] == ((b->dsl_flavours & 2) != 0))) {
                if (!forked || merge) RT(rt::event_record(b->ev_fork, s));
                RT(rt::stream_wait_event(aux, b->ev_fork));
                for (int list : {(int)kScoreFast, (int)kScoreGen}) {
                    if (!flavour_live(list) || score_done[list == kScoreGen ? 1 : 0]) continue;
                    ref.list = list; ref.join_late = 0;
                    const int rc = run_dp_kind(h, b, 0, list, sl.d_tasks, b->dsl_list_bound, sl.d_ends, h->cfg.nuc_prior, nullptr, status, &aux, false, ref);
                    if (rc != OCT_PHMM_OK) return rc;
                }
                RT(rt::event_record(b->ev_join, aux)); RT(rt::stream_wait_event(s, b->ev_join));
            }
        }
        if (sl.out1 > sl.out0) { OCT_LAUNCH(k_epilogue, (uint32_t)((sl.out1 - sl.out0 + 255) / 256), 256, 0, s, d, mapped_out ? (double*)h->out_stage : b->d_out, sl.out0, sl.out1,
                                            mapped_out ? b->stat_stage : nullptr, mapped_stripes); RT(rt::launch_ok()); }
        if (b->early_out && !mapped_out && sl.out1 > sl.out0)    // one-shot call: the results land in the handle's pinned zone behind the epilogue, no second synchronisation
            RT(rt::d2h((b->out_landing ? b->out_landing : (double*)h->out_stage) + sl.out0, b->d_out + sl.out0, (size_t)(sl.out1 - sl.out0) * sizeof(double), s));
        RT(rt::event_record(sl.done, s));
        return OCT_PHMM_OK;
    };
    // phase 2: task emission, the DP kernels (+ traceback walk), epilogue for the slice's rows
    auto phase2 = [&](int i) -> int {
        rt::Range range_p2("slice phase 2: emit, DP, walk, epilogue");
        oct_phmm_batch::Slice& sl = b->slices[i];
        rt::Stream s = h->slice_stream(i);
        const uint64_t np = sl.pair1 - sl.pair0;
        const uint4 totals = sl.totals;
        b->n_tasks[0] += totals.x; b->n_tasks[1] += totals.y; b->n_tasks[2] += totals.z; b->n_tasks[3] += totals.w;
        const uint4 late = sl.totals_late;                      // x: fast-cost kernel, y: generic kernel
        const size_t total = (size_t)totals.x + totals.y + totals.z + totals.w + late.x + late.y;
        if (total > sl.tasks_cap) {
            h->pool.release(sl.d_tasks); sl.d_tasks = nullptr; sl.tasks_cap = 0;
            void* p = nullptr; RT(h->pool.alloc(&p, (total + total / 8) * sizeof(DevTask))); sl.d_tasks = (DevTask*)p; sl.tasks_cap = total + total / 8;
        }
        // a flavour's traceback list and its late-start list (which lies right behind it) in ONE DP launch and ONE walk: one-slice batches, where the step is a chain of
        // dependent launches (a region server's device batch; 16 regions: the second traceback launch and its walk were 233 of 868 us). Batches of several slices keep
        // the two launches: the 12.8 M-pair step lost 6 % of its traceback DP with them joined (17.8 against 2 x 8.36 ms per launch; profiles/EXPERIMENTS.md)
        const bool join = sl.scan_fused && !b->pair_ok && (tune::join_late() >= 0 ? tune::join_late() != 0 : (S == 1 && np <= 2000000));   // (window pairing keeps the lists apart: a haplotype's run is per list)
        const size_t n_trace = join ? (size_t)std::max(totals.y + late.x, totals.w + late.y) : (size_t)std::max(std::max(totals.y, totals.w), std::max(late.x, late.y));
        if (n_trace > sl.ends_cap) {
            h->pool.release(sl.d_ends); sl.d_ends = nullptr; sl.ends_cap = 0;
            void* p = nullptr; RT(h->pool.alloc(&p, (n_trace + n_trace / 8) * sizeof(TraceEnd))); sl.d_ends = (TraceEnd*)p; sl.ends_cap = n_trace + n_trace / 8;
        }
        if (b->align_mode && n_trace > sl.keys_cap) {
            h->pool.release(sl.d_keys); sl.d_keys = nullptr; sl.keys_cap = 0;
            void* p = nullptr; RT(h->pool.alloc(&p, (n_trace + n_trace / 8) * sizeof(unsigned long long))); sl.d_keys = (unsigned long long*)p; sl.keys_cap = n_trace + n_trace / 8;
        }
        if (total) {
            // physical order (task_list_range): score-only fast, traceback fast, LATE fast, score-only generic, traceback generic, LATE generic
            TaskArrays ta, tl;                                   // tl: late-start traceback tasks, [0] fast-cost kernel, [1] generic
            ta.t[0] = sl.d_tasks; ta.t[1] = ta.t[0] + totals.x; tl.t[0] = ta.t[1] + totals.y; ta.t[2] = tl.t[0] + late.x; ta.t[3] = ta.t[2] + totals.z;
            tl.t[1] = ta.t[3] + totals.w; tl.t[2] = tl.t[1] + late.y; tl.t[3] = tl.t[2];
            OCT_LAUNCH(k_emit, (uint32_t)((np + 255) / 256), 256, 0, s, d, sl.pair0, sl.pair1, (const uint4*)sl.cnt, (const uint4*)b->d_hap_base, ta,
                       (const uint4*)sl.cnt_late, (const uint4*)b->d_hap_base_late, tl, TaskListRef {nullptr, nullptr, 0, nullptr}, G,
                       sl.scan_fused ? (const uint4*)sl.tile_sums : nullptr, sl.scan_fused ? (const uint4*)sl.tile_sums_late : nullptr); RT(rt::launch_ok());
            // window pairing: the three fast-cost lists re-ordered per haplotype, out of place (k_pair_sort); the DP and the walk then read the sorted copy
            // ... where a haplotype's runs are long enough to hold pairs: ~240 (offset, strand) classes per 150-base read on a 300-base haplotype - from ~1,000 fast-cost tasks per
            // haplotype of the slice on. (The 2,000-region stream has ~500 per haplotype over three lists: a sort workgroup per run would cost more than the few pairs give.)
            long long min_run = 1024; tune::number("OCT_PHMM_PAIRED_MIN_RUN", &min_run);      // (test hook: 0 = every slice of a batch that may pair)
            const bool pairing = b->pair_ok && !join && !b->dsl && (totals.x + totals.y + late.x) > 0 &&
                                 (uint64_t)(totals.x + totals.y + late.x) >= (uint64_t)min_run * (sl.hap1 - sl.hap0);
            TaskArrays ts = ta, tsl = tl;
            if (pairing) {
                if (total > sl.sorted_cap) {
                    h->pool.release(sl.d_tasks_sorted); sl.d_tasks_sorted = nullptr; sl.sorted_cap = 0;
                    void* q = nullptr; RT(h->pool.alloc(&q, (total + total / 8) * sizeof(DevTask))); sl.d_tasks_sorted = (DevTask*)q; sl.sorted_cap = total + total / 8;
                }
                ts.t[0] = sl.d_tasks_sorted; ts.t[1] = ts.t[0] + totals.x; tsl.t[0] = ts.t[1] + totals.y;
                const PairSortList none {nullptr, nullptr, nullptr, nullptr, 0, 0};
                const PairSortList l0 = totals.x ? PairSortList {ta.t[0], ts.t[0], b->d_paired_end[0], (const uint4*)b->d_hap_base, 0u, totals.x} : none;
                const PairSortList l1 = totals.y ? PairSortList {ta.t[1], ts.t[1], b->d_paired_end[1], (const uint4*)b->d_hap_base, 1u, totals.y} : none;
                const PairSortList l2 = late.x ? PairSortList {tl.t[0], tsl.t[0], b->d_paired_end[2], (const uint4*)b->d_hap_base_late, 0u, late.x} : none;
                OCT_LAUNCH(k_pair_sort, 3 * (sl.hap1 - sl.hap0), kPairSortThreads, pair_sort_lds_bytes(b->lh_cap), s, l0, l1, l2, d.rrev, sl.hap0, sl.hap1 - sl.hap0, pair_sort_keys(b->lh_cap));
                RT(rt::launch_ok());
            }
            static const int order[kNumKinds] = {kTraceFast, kTraceGen, kScoreFast, kScoreGen};   // traceback first: its walk then overlaps the score-only DP of the next slice
            // A single-slice (region-sized) batch is latency-bound: its score-only DP runs beside the traceback DP + walk on a second stream.
            const bool side = S == 1 && total < 200000 && (totals.x + totals.z) > 0 && (totals.y + totals.w + late.x + late.y) > 0;   // big launches fill the chip on their own
            rt::Stream aux = h->slice_stream(1);
            bool forked = false;                                 // (side) the score-only DP starts beside the first traceback launch's walk: see phase2_device_sized
            // ... except for long reads: a traceback launch of ~10^2 tasks is a few hundred latency-bound waves that leave the chip's issue slots to the score-only DP
            // ... and (round 5) for every such batch: joined with its late-start list the first traceback launch is the whole traceback DP, and a score-only DP that
            // waits for it runs behind it instead of beside it (16 regions: 204 + 190 us one after the other). OCT_PHMM_DSL_FORK_EARLY=0: beside the first walk.
            if (side) { RT(rt::event_record(b->ev_fork, s)); forked = true; }
            for (int lk = 0; lk < 2 && !join; ++lk) {            // late-start traceback launches first (the longest walks of the slice start earliest)
                const uint32_t n = lk ? late.y : late.x;
                const int rc = run_dp_kind(h, b, i, lk ? kTraceGen : kTraceFast, tsl.t[lk], n, sl.d_ends, h->cfg.nuc_prior, nullptr, status, nullptr, true,
                                           TaskListRef {nullptr, nullptr, 0, nullptr}, side && !forked && n ? &b->ev_fork : nullptr, -1, 0, 0xffffffffu,
                                           pairing && lk == 0 ? b->d_paired_end[2] : nullptr);
                if (rc != OCT_PHMM_OK) return rc;
                forked = forked || (side && n);
            }
            for (int k : order) {
                const bool score_kind = k == kScoreFast || k == kScoreGen;
                const uint32_t n = k == 0 ? totals.x : k == 1 ? totals.y + (join ? late.x : 0u) : k == 2 ? totals.z : totals.w + (join ? late.y : 0u);   // (join: the late-start list lies right behind)
                if (side && score_kind && !forked) { RT(rt::event_record(b->ev_fork, s)); forked = true; }
                if (side && score_kind && n) RT(rt::stream_wait_event(aux, b->ev_fork));
                const int rc = run_dp_kind(h, b, i, k, ts.t[k], n, sl.d_ends, h->cfg.nuc_prior, nullptr, status, side && score_kind ? &aux : nullptr, false,
                                           TaskListRef {nullptr, nullptr, 0, nullptr}, side && !score_kind && !forked && n ? &b->ev_fork : nullptr, -1, 0,
                                           join && !score_kind ? (k == kTraceFast ? totals.y : totals.w) : 0xffffffffu,
                                           pairing && k == kScoreFast ? b->d_paired_end[0] : pairing && k == kTraceFast ? b->d_paired_end[1] : nullptr);
                if (rc != OCT_PHMM_OK) return rc;
                forked = forked || (side && !score_kind && n);
            }
            if (side) { RT(rt::event_record(b->ev_join, aux)); RT(rt::stream_wait_event(s, b->ev_join)); }
        }
        if (b->align_mode) {
            if (np) { OCT_LAUNCH(k_epilogue_align, (uint32_t)((np + 255) / 256), 256, 0, s, d, sl.pair0, sl.pair1, b->d_aln_lik, b->d_aln_mpos, b->d_aln_n, b->d_aln_ops, b->cig_cap); RT(rt::launch_ok()); }
        } else if (sl.out1 > sl.out0) {
            if (b->dedup && sl.resumes) for (int j = 0; j < i; ++j) RT(rt::stream_wait_event(s, b->slices[j].done));   // pairs of a resumed region may share results of earlier slices
            OCT_LAUNCH(k_epilogue, (uint32_t)((sl.out1 - sl.out0 + 255) / 256), 256, 0, s, d, mapped_out ? (double*)h->out_stage : b->d_out, sl.out0, sl.out1,
                       mapped_out ? b->stat_stage : nullptr, mapped_stripes); RT(rt::launch_ok()); }
        if (b->early_out && !mapped_out && sl.out1 > sl.out0)
            RT(rt::d2h((b->out_landing ? b->out_landing : (double*)h->out_stage) + sl.out0, b->d_out + sl.out0, (size_t)(sl.out1 - sl.out0) * sizeof(double), s));
        RT(rt::event_record(sl.done, s));
        return OCT_PHMM_OK;
    };
    auto deliver = [&](int i) -> int {                        // finished slice -> the caller's buffer (host copy overlaps the later slices' kernels)
        const oct_phmm_batch::Slice& sl = b->slices[i];
        if (!b->early_out || sl.out1 <= sl.out0 || S == 1 || b->out_landing) return OCT_PHMM_OK;   // (a one-slice batch: oct_phmm_populate copies after its one wait; a page-locked `out`: the DMA wrote it)
        RT(rt::event_sync(sl.done));
        const char* src = (const char*)((const double*)h->out_stage + sl.out0); char* dst = (char*)(b->early_out + sl.out0);
        host_parallel((size_t)(sl.out1 - sl.out0) * sizeof(double), (size_t)2 << 20, [&](size_t lo, size_t hi) { memcpy(dst + lo, src + lo, hi - lo); });
        return OCT_PHMM_OK;
    };
    // software pipeline over slices: phase 1 of slice i+1 is enqueued before the host waits for slice i's task counts
    int rc = S ? phase1(0) : OCT_PHMM_OK;
    if (b->dsl) {                                             // one slice, no read-back: phase 2 follows at once
        if (rc == OCT_PHMM_OK) rc = phase2_device_sized();
    } else
    for (int i = 0; i < S && rc == OCT_PHMM_OK; ++i) {
        if (i + 1 < S) rc = phase1(i + 1);
        if (rc != OCT_PHMM_OK) break;
        RT(rt::stream_sync(h->slice_stream(i)));              // the host read-back that sizes this slice's launches
        rc = phase2(i);
        if (rc == OCT_PHMM_OK && i >= 2) rc = deliver(i - 2);
    }
    if (rc != OCT_PHMM_OK) return rc;
    for (int i = std::max(0, S - 2); i < S; ++i) { rc = deliver(i); if (rc != OCT_PHMM_OK) return rc; }
    for (int i = first_aside; i < S; ++i) RT(rt::stream_wait_event(s0, b->slices[i].done));
    if (!mapped_out) RT(rt::d2h(b->stat_stage ? b->stat_stage : b->h_stat_stripes.data(), d.stats, kStatWords * sizeof(unsigned long long), s0));   // (else: the epilogue left the sums there)
    b->ran = true; b->synced = false;
    return ok(status);
}

extern "C" int oct_phmm_batch_wait(oct_phmm_handle* h, oct_phmm_batch* b, oct_phmm_status* status)
{
    if (!h || !b || b->owner != h || !b->ran) return fail(status, OCT_PHMM_EINVAL, "batch was not run");
    RT(rt::set_device(h->cfg.device_id));
    RT(rt::stream_sync(h->stream));
    if (b->stat_stage) memcpy(b->h_stat_stripes.data(), b->stat_stage, b->h_stat_stripes.size() * sizeof(unsigned long long));
    if (b->dsl && b->h_stat_stripes[(size_t)kStatSlots * kStatStride + 1]) {
        // a traceback list outgrew the scratch provisioned for the device-sized launches: every list read as empty. Once more, host-sized.
        b->dsl = false;
        const int rc = oct_phmm_batch_run(h, b, status);
        if (rc != OCT_PHMM_OK) return rc;
        RT(rt::stream_sync(h->stream));
        if (b->stat_stage) memcpy(b->h_stat_stripes.data(), b->stat_stage, b->h_stat_stripes.size() * sizeof(unsigned long long));
    }
    b->synced = true;                                           // (every slice stream joined the handle's before the counters were copied)
    for (int k = 0; k < 12; ++k) { b->h_stats[k] = 0; for (uint32_t sl = 0; sl < kStatSlots; ++sl) b->h_stats[k] += b->h_stat_stripes[(size_t)sl * kStatStride + k]; }
    b->h_err_key = ~b->h_stat_stripes[(size_t)kStatSlots * kStatStride];
    if (tune::map_stats()) {
        unsigned long long dec = 0, cnt = 0;
        for (uint32_t sl = 0; sl < kStatSlots; ++sl) { dec += b->h_stat_stripes[(size_t)sl * kStatStride + 6]; cnt += b->h_stat_stripes[(size_t)sl * kStatStride + 7]; }
        fprintf(stderr, "{\"mapper_pairs_decided_by_shortcut\": %llu, \"mapper_pairs_counted\": %llu}\n", dec, cnt);
    }
    b->dp_ms = 0; b->dp_launches = 0;
    for (int k = 0; k < kNumKinds; ++k) { b->kind_ms[k] = 0; b->kind_launches[k] = 0; }
    for (size_t i = 0; i < b->timers.size(); ++i) {
        float ms = 0; RT(rt::event_elapsed_ms(&ms, b->timers[i].first, b->timers[i].second));
        b->dp_ms += ms; ++b->dp_launches; b->kind_ms[b->timer_kind[i]] += ms; ++b->kind_launches[b->timer_kind[i]];
    }
    if (b->h_err_key != ~0ull) {
        // ShortHaplotypeError: recompute required_extension for the first offending (haplotype, read) — model.cpp:238-253
        const uint32_t hp = (uint32_t)(b->h_err_key >> 32), r = (uint32_t)b->h_err_key;
        fail(status, OCT_PHMM_ESHORT_HAPLOTYPE, "Haplotype is too short for alignment");
        if (status) {
            const uint64_t T = b->h_roff[r + 1] - b->h_roff[r], Lh = b->h_hoff[hp + 1] - b->h_hoff[hp], B = (uint64_t)h->band;
            const uint64_t orig = (uint64_t)(b->h_rbegin[r] - b->h_hbegin[hp]);
            int32_t min_shift;
            if (orig < B) min_shift = (int32_t)(B - orig); else { const uint64_t e = orig + T + B; min_shift = e > Lh ? (int32_t)Lh - (int32_t)e : 0; }
            status->hap_index = hp; status->read_index = r;
            status->required_extension = min_shift > 0 ? (uint32_t)min_shift : (uint32_t)((uint32_t)(-min_shift) - orig);
        }
        return OCT_PHMM_ESHORT_HAPLOTYPE;
    }
    return ok(status);
}

extern "C" int oct_phmm_batch_download(oct_phmm_handle* h, oct_phmm_batch* b, double* out, oct_phmm_status* status)
{
    if (!out && b && b->n_out) return fail(status, OCT_PHMM_EINVAL, "null output");
    rt::Range range_("oct_phmm download");
    const int rc = oct_phmm_batch_wait(h, b, status);
    if (rc != OCT_PHMM_OK) return rc;
    RT(rt::d2h(out, b->d_out, (size_t)b->n_out * sizeof(double), h->stream));
    RT(rt::stream_sync(h->stream));
    return ok(status);
}

extern "C" int oct_phmm_batch_candidate_positions(oct_phmm_handle* h, oct_phmm_batch* b, uint8_t* counts, uint32_t* positions, oct_phmm_status* status)
{
    if (!h || !b) return fail(status, OCT_PHMM_EINVAL, "null handle or batch");
    if (b->n_pairs && (!counts || !positions)) return fail(status, OCT_PHMM_EINVAL, "null output");
    const int rc = oct_phmm_batch_wait(h, b, status);
    if (rc != OCT_PHMM_OK) return rc;
    RT(rt::d2h(counts, b->d.npos, (size_t)b->n_pairs, h->stream));
    RT(rt::d2h(positions, b->d.pos, (size_t)b->n_pairs * (size_t)b->d.max_pos * sizeof(uint32_t), h->stream));
    RT(rt::stream_sync(h->stream));
    return ok(status);
}

extern "C" int oct_phmm_batch_penalty_vectors(oct_phmm_handle* h, oct_phmm_batch* b, int8_t* gap_open, int8_t* gap_extend, char* snv_mask_fwd,
                                              int8_t* snv_prior_fwd, char* snv_mask_rev, int8_t* snv_prior_rev, oct_phmm_status* status)
{
    if (!h || !b || b->owner != h) return fail(status, OCT_PHMM_EINVAL, "bad handle/batch");
    const size_t n = b->n_hap_bases;
    if (n && (!gap_open || !gap_extend || !snv_mask_fwd || !snv_prior_fwd || !snv_mask_rev || !snv_prior_rev)) return fail(status, OCT_PHMM_EINVAL, "null output");
    RT(rt::set_device(h->cfg.device_id));
    RT(rt::d2h(gap_open, b->d.go, n, h->stream)); RT(rt::d2h(gap_extend, b->d.ge, n, h->stream));
    RT(rt::d2h(snv_mask_fwd, b->d.maskF, n, h->stream)); RT(rt::d2h(snv_prior_fwd, b->d.priorF, n, h->stream));
    RT(rt::d2h(snv_mask_rev, b->d.maskR, n, h->stream)); RT(rt::d2h(snv_prior_rev, b->d.priorR, n, h->stream));
    RT(rt::stream_sync(h->stream));
    return ok(status);
}

extern "C" int oct_phmm_batch_stats(const oct_phmm_batch* b, oct_phmm_stats* st)
{
    if (!b || !st) return OCT_PHMM_EINVAL;
    st->n_candidates = b->h_stats[0]; st->n_fast_path = b->h_stats[1]; st->n_dp_score_only = b->h_stats[2];
    st->n_dp_traceback = b->h_stats[3]; st->band_cells = b->h_stats[4]; st->n_pairs = b->h_stats[5];
    st->n_dp_score_only_shared = b->h_stats[8]; st->n_dp_traceback_shared = b->h_stats[9]; st->band_cells_shared = b->h_stats[10]; st->n_pairs_shared = b->h_stats[11];
    return OCT_PHMM_OK;
}

extern "C" size_t oct_phmm_batch_out_size(const oct_phmm_batch* b) { return b ? (size_t)b->n_out : 0; }

extern "C" int oct_phmm_batch_device_sized(const oct_phmm_batch* b) { return b && b->dsl ? 1 : 0; }

extern "C" int oct_phmm_test_set(const char* name, const char* value)
{
    if (!name || strncmp(name, "OCT_PHMM_", 9) != 0) return OCT_PHMM_EINVAL;
    std::lock_guard<std::mutex> lk(tune::switch_mu());
    tune::switch_table_used().store(true, std::memory_order_release);
    // tune::get hands out pointers into the table's strings and its callers read them after the lock is gone (atoll on another thread's upload): a value that is
    // replaced or removed moves to a list that is never freed instead of dying under a reader (a few bytes per oct_phmm_test_set call, tests and tools only)
    // (the NODE is kept, not a moved-to string: a short value lives inside its std::string object, so moving it copies the characters and the reader's pointer would be left
    // pointing into the erased node - ADVICE r04)
    static std::list<std::map<std::string, std::string>::node_type> retired;
    auto it = tune::switch_table().find(name);
    if (it != tune::switch_table().end()) retired.push_back(tune::switch_table().extract(it));
    if (value) tune::switch_table().emplace(name, value);
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// diagnostic: the shader clock while other work runs (bench.py prices its VALU roofline at the clock the DP kernels actually get)
// ---------------------------------------------------------------------------------------------------------------
#if !defined(OCTPHMM_SIM)
__global__ void k_clock_probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = clock64(), r0 = wall_clock64();          // s_memtime: shader cycles; s_memrealtime: the constant reference clock
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
#endif
extern "C" int oct_phmm_probe_clock(oct_phmm_handle* h, double window_ms, double* shader_ghz)
{
    if (!h || !shader_ghz || !(window_ms > 0) || window_ms > 1000) return OCT_PHMM_EINVAL;
#if defined(OCTPHMM_SIM)
    return OCT_PHMM_EUNSUPPORTED;
#else
    if (!rt::set_device(h->cfg.device_id)) return OCT_PHMM_EHIP;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id) != hipSuccess || khz <= 0) return OCT_PHMM_EHIP;
    // its own stream (made once per handle: creating one, like hipMalloc, synchronises the device): the probe wave runs beside whatever the handle's streams are doing
    if (!h->probe_ready) {
        if (!rt::stream_create(&h->probe_stream) || !rt::dev_malloc((void**)&h->d_probe, 16) || !rt::host_pinned_malloc((void**)&h->h_probe, 16)) return OCT_PHMM_EHIP;
        h->probe_ready = true;
    }
    rt::Stream s = h->probe_stream; unsigned long long* v = h->h_probe;
    v[0] = v[1] = 0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, h->d_probe, (unsigned long long)(window_ms * khz));
    const bool ok = rt::launch_ok() && rt::d2h(v, h->d_probe, 16, s) && rt::stream_sync(s);
    if (!ok || !v[1]) return OCT_PHMM_EHIP;
    *shader_ghz = (double)v[0] / (double)v[1] * khz * 1e-6;
    return OCT_PHMM_OK;
#endif
}

extern "C" int oct_phmm_batch_kernel_time(const oct_phmm_batch* b, double* ms, uint32_t* launches)
{
    if (!b) return OCT_PHMM_EINVAL;
    if (ms) *ms = b->dp_ms;
    if (launches) *launches = b->dp_launches;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_batch_kernel_time_by_kind(const oct_phmm_batch* b, double ms[4], uint32_t launches[4])
{
    if (!b) return OCT_PHMM_EINVAL;
    for (int k = 0; k < kNumKinds; ++k) { if (ms) ms[k] = b->kind_ms[k]; if (launches) launches[k] = b->kind_launches[k]; }
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// genotype read-out (phmm_readout.hpp)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_batch_genotype_likelihoods(oct_phmm_handle* h, oct_phmm_batch* b, const oct_phmm_genotype_sets* gs,
                                                   double* out, oct_phmm_status* status)
{
    if (!h || !b || b->owner != h || !gs) return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (!b->ran) return fail(status, OCT_PHMM_EINVAL, "batch was not run");
    if (gs->n_sets == 0) return ok(status);
    if (!gs->ploidy || !gs->gt_offsets || gs->gt_offsets[0] != 0) return fail(status, OCT_PHMM_EINVAL, "genotype set tables");
    const uint32_t n_gt = gs->gt_offsets[gs->n_sets];
    if (n_gt == 0) return ok(status);
    if (!gs->hap_indices || !out) return fail(status, OCT_PHMM_EINVAL, "null genotype indices or output");
    const int rc = oct_phmm_batch_wait(h, b, status);
    if (rc != OCT_PHMM_OK) return rc;

    constexpr uint32_t kTargetBlocks = 2048;                     // >= 8 workgroups per CU before rows are split
    constexpr size_t kTileBytes = 96 * 1024;
    std::vector<ReadoutSet> sets(gs->n_sets);
    std::vector<uint4> blocks, sum_blocks;
    uint64_t idx_off = 0, partial_off = 0; size_t lds = 0;
    uint64_t total_blocks_unsplit = 0;
    for (uint32_t s = 0; s < gs->n_sets; ++s) total_blocks_unsplit += (gs->gt_offsets[s + 1] - gs->gt_offsets[s] + kReadoutThreads - 1) / kReadoutThreads;
    for (uint32_t s = 0; s < gs->n_sets; ++s) {
        ReadoutSet& q = sets[s];
        if (gs->gt_offsets[s + 1] < gs->gt_offsets[s]) return fail(status, OCT_PHMM_EINVAL, "genotype offsets must not decrease");
        q.gt0 = gs->gt_offsets[s]; q.n_genotypes = gs->gt_offsets[s + 1] - q.gt0; q.ploidy = gs->ploidy[s];
        q.gt_idx_off = idx_off; q.partial_off = partial_off;
        if (q.ploidy < 1 || q.ploidy > OCT_PHMM_MAX_PLOIDY) return fail(status, OCT_PHMM_EUNSUPPORTED, "ploidy outside 1..16");
        const uint32_t* gi = gs->hap_indices + idx_off;
        idx_off += (uint64_t)q.n_genotypes * q.ploidy;
        if (q.n_genotypes == 0) { q.n_splits = 1; continue; }
        if (gi[0] >= b->n_haps) return fail(status, OCT_PHMM_EINVAL, "haplotype index out of range");
        const uint32_t reg = b->h_hap_region[gi[0]];
        q.hap0 = b->h_reg_hap0[reg]; q.n_haps = b->h_reg_hap0[reg + 1] - q.hap0;
        for (uint64_t i = 0; i < (uint64_t)q.n_genotypes * q.ploidy; ++i) {
            if (gi[i] < q.hap0 || gi[i] >= q.hap0 + q.n_haps) return fail(status, OCT_PHMM_EINVAL, "genotypes of one set must use haplotypes of one region");
            if (i % q.ploidy && gi[i] < gi[i - 1]) return fail(status, OCT_PHMM_EINVAL, "genotype haplotype indices must be sorted");
        }
        const uint32_t rows = (uint32_t)(b->h_hap_out_off[q.hap0 + 1] - b->h_hap_out_off[q.hap0]);
        q.row_begin = gs->row_begin ? gs->row_begin[s] : 0; q.row_end = gs->row_end ? gs->row_end[s] : rows;
        if (q.row_begin > q.row_end || q.row_end > rows) return fail(status, OCT_PHMM_EINVAL, "row range outside the region");
        const uint32_t nrows = q.row_end - q.row_begin;
        const size_t fit = kTileBytes / (8 * (size_t)q.n_haps);
        if (fit < 3) return fail(status, OCT_PHMM_EUNSUPPORTED, "too many haplotypes in one region for the read-out tile");
        q.tile_rows = (uint32_t)std::min<size_t>(32, fit - 1);
        lds = std::max(lds, (size_t)q.n_haps * (q.tile_rows + 1) * 8);
        const uint32_t n_chunks = (q.n_genotypes + kReadoutThreads - 1) / kReadoutThreads;
        const uint32_t n_tiles = std::max(1u, (nrows + q.tile_rows - 1) / q.tile_rows);
        uint32_t want = total_blocks_unsplit >= kTargetBlocks ? 1 : (uint32_t)((kTargetBlocks + total_blocks_unsplit - 1) / total_blocks_unsplit);
        want = std::min(want, n_tiles);
        const uint32_t tiles_per_split = (n_tiles + want - 1) / want;
        q.rows_per_split = tiles_per_split * q.tile_rows;
        q.n_splits = (n_tiles + tiles_per_split - 1) / tiles_per_split;
        if (q.n_splits > 1) partial_off += (uint64_t)q.n_splits * q.n_genotypes;
        for (uint32_t sp = 0; sp < q.n_splits; ++sp)
            for (uint32_t c = 0; c < n_chunks; ++c) blocks.push_back(uint4{s, c, sp, 0});
        if (q.n_splits > 1) for (uint32_t c = 0; c < n_chunks; ++c) sum_blocks.push_back(uint4{s, c, 0, 0});
    }
    if (blocks.empty()) return ok(status);
    if (blocks.size() > 0x7fffffffull) return fail(status, OCT_PHMM_EUNSUPPORTED, "too many genotypes in one call");

    RT(rt::set_device(h->cfg.device_id));
    rt::Stream st = h->stream;
    struct Tmp { oct_phmm_handle* h; std::vector<void*> v; ~Tmp() { for (void* p : v) h->pool.release(p); } } tmp {h, {}};
    auto put = [&](const void* host, size_t bytes, void** dev) {
        if (!h->pool.alloc(dev, bytes)) return false;
        tmp.v.push_back(*dev);
        return host ? rt::h2d(*dev, host, bytes, st) : true;
    };
    void *d_gt = nullptr, *d_sets = nullptr, *d_blocks = nullptr, *d_sum = nullptr, *d_partial = nullptr, *d_res = nullptr;
    RT(put(gs->hap_indices, idx_off * sizeof(uint32_t), &d_gt));
    RT(put(sets.data(), sets.size() * sizeof(ReadoutSet), &d_sets));
    RT(put(blocks.data(), blocks.size() * sizeof(uint4), &d_blocks));
    if (!sum_blocks.empty()) { RT(put(sum_blocks.data(), sum_blocks.size() * sizeof(uint4), &d_sum)); RT(put(nullptr, partial_off * sizeof(double), &d_partial)); }
    RT(put(nullptr, (size_t)n_gt * sizeof(double), &d_res));
    ReadoutParams p {};
    p.lik = b->d_out; p.hap_out_off = b->d.hap_out_off; p.gt = (const uint32_t*)d_gt; p.sets = (const ReadoutSet*)d_sets;
    p.blocks = (const uint4*)d_blocks; p.partial = (double*)d_partial; p.out = (double*)d_res;
    if (lds > 64 * 1024) RT(rt::allow_lds(k_genotype_lik, lds));
    OCT_LAUNCH(k_genotype_lik, (uint32_t)blocks.size(), kReadoutThreads, lds, st, p);
    RT(rt::launch_ok());
    if (!sum_blocks.empty()) {
        OCT_LAUNCH(k_genotype_sum, (uint32_t)sum_blocks.size(), kReadoutThreads, 0, st, p, (const uint4*)d_sum);
        RT(rt::launch_ok());
    }
    RT(rt::d2h(out, d_res, (size_t)n_gt * sizeof(double), st));
    RT(rt::stream_sync(st));
    return ok(status);
}

extern "C" void* oct_phmm_host_alloc(size_t bytes) { void* p = nullptr; return rt::host_pinned_malloc(&p, bytes) ? p : nullptr; }
extern "C" void oct_phmm_host_free(void* p) { rt::host_pinned_free(p); }

// oct_phmm_populate in two halves, so that a caller who owns the handle (the region server's workers) can prepare its next batch on another handle while this one
// computes: populate_begin returns when the step is enqueued (device-sized batches: no wait at all; host-sized ones wait once, for the task counts), populate_end waits
// for the results. in_place (populate_end): a one-slice batch leaves its results in the handle's pinned landing zone and is NOT copied into `out` - *in_place points at
// them, valid until the handle's next call (the server scatters them straight into its callers' matrices).
struct PopulateCall { oct_phmm_batch* b = nullptr; bool early = false; };
static int populate_begin(oct_phmm_handle* h, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps, const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                          const oct_phmm_positions* positions, double* out, oct_phmm_status* status, PopulateCall* pc, const InputFacts* pre = nullptr)
{
    oct_phmm_batch* b = nullptr;
    int rc = upload_impl(h, reads, haps, regions, flank, positions, &b, status, false, 0, true, pre);
    bool early = false;
    if (rc == OCT_PHMM_OK && out && b->n_out) {                // results come back through a pinned landing zone: slice by slice while a big batch computes, behind the
        const size_t bytes = (size_t)b->n_out * sizeof(double);  // epilogue of a small one - one stream synchronisation per call, no staged copy into pageable memory
        if (bytes >= tune::pinned_min_bytes(kPinnedOutMinBytes) && rt::host_is_pinned(out, bytes)) { b->early_out = out; b->out_landing = out; early = true; }   // the caller's own page-locked buffer IS the landing zone
        else if (h->out_stage_bytes < bytes) {
            const size_t roomy = std::max(bytes + bytes / 2, (size_t)1 << 20);     // (a region thread's calls differ in size: no regrowth per call)
            rt::host_pinned_free(h->out_stage); h->out_stage = nullptr; h->out_stage_bytes = 0;
            if (rt::host_pinned_malloc(&h->out_stage, roomy)) h->out_stage_bytes = roomy;
            else if (rt::host_pinned_malloc(&h->out_stage, bytes)) h->out_stage_bytes = bytes;
        }
        if (!early && h->out_stage_bytes >= bytes) { b->early_out = out; early = true; }
    }
    if (rc == OCT_PHMM_OK) rc = oct_phmm_batch_run(h, b, status);
    if (rc != OCT_PHMM_OK) { oct_phmm_batch_free(h, b); b = nullptr; }
    pc->b = b; pc->early = early;
    return rc;
}
static int populate_end(oct_phmm_handle* h, PopulateCall* pc, double* out, oct_phmm_status* status, const double** in_place = nullptr)
{
    oct_phmm_batch* b = pc->b;
    if (!b) return fail(status, OCT_PHMM_EINVAL, "no call in flight");
    int rc = pc->early ? oct_phmm_batch_wait(h, b, status) : oct_phmm_batch_download(h, b, out, status);
    if (in_place) *in_place = nullptr;
    if (rc == OCT_PHMM_OK && pc->early && b->slices.size() == 1 && !b->out_landing) {
        if (in_place) *in_place = (const double*)h->out_stage;
        else memcpy(out, h->out_stage, (size_t)b->n_out * sizeof(double));
    }
    oct_phmm_batch_free(h, b);
    pc->b = nullptr;
    return rc;
}

extern "C" int oct_phmm_populate(oct_phmm_handle* h, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
                                 const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                                 const oct_phmm_positions* positions, double* out, oct_phmm_status* status)
{
    PopulateCall pc;
    int rc = populate_begin(h, reads, haps, regions, flank, positions, out, status, &pc);
    if (rc == OCT_PHMM_OK) rc = populate_end(h, &pc, out, status);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// region server: calls from many threads -> multi-region batches on one handle
// ---------------------------------------------------------------------------------------------------------------
struct oct_phmm_server {
    struct Request {
        const oct_phmm_reads* R; const oct_phmm_haplotypes* H; const oct_phmm_flank_state* flank; const oct_phmm_positions* pos;
        double* out; oct_phmm_status st; int rc = OCT_PHMM_OK; bool done = false;
        InputFacts facts; bool have_facts = false;        // what an upload must know about every byte of the call (range checks, bounds, cost flavours): made by the CALLER's thread before it queues
        std::mutex m; std::condition_variable cv;         // one pair per call: finishing a batch wakes exactly its callers, and nobody queues for the server's lock to return
    };
#if defined(OCTPHMM_SIM)
    static constexpr int kWorkers = 1;                   // the CPU wave simulator is single-threaded
#else
    static constexpr int kWorkers = 2;                   // worker threads per GPU, each with kSlots handles (round 5, pipelined workers on the configs[3] regions, 16 / 64 / 128 callers: 1 worker 12.0 / 18.7 / 16.7 k
                                                         // regions/s, 2: 11.6 / 20.6 - 23.6 / 18.3, 3: 8.4 / 18.2 / 17.9 - gpurun_out/r05_s03; round 4, one handle per worker: 2 workers 13.6 / 12.7 / 15.5, 3: 13.9 / 17.0 / 16.8,
                                                         // 4: 13.5 / 16.7 / 18.4). More workers mean smaller batches, and a batch of 6 costs the device what one of 12 does.
#endif
    static constexpr int kSlots = 2;                     // handles per worker (round 5): while the batch on one computes, the worker gathers, checks, packs and enqueues the next on the other
    std::vector<oct_phmm_handle*> hs;                    // kSlots handles per worker, worker-major; kWorkers workers per device, device-major
    uint32_t max_regions = 256;
    std::mutex mu; std::condition_variable cv_work;
    std::deque<Request*> queue;
    bool stop = false;
    std::vector<std::thread> workers;                    // all of them drain the one queue, so an idle device takes the next calls
    std::atomic<uint64_t> n_calls {0}, n_batches {0};    // counted when the calls are taken / the batch is enqueued: a caller that has its answer finds itself counted
    std::vector<uint64_t> n_calls_by_device;
    int busy_workers = 0;                                // workers between taking calls and answering them (under mu)
                                                         // (profiles/r04_step3_server_sweep.log): a bigger device batch is not cheaper per region, the step is a chain of ~25 small launches either way
    std::atomic<bool> has_model {false};                 // oct_phmm_server_set_error_model: calls may leave their penalty vectors NULL
    // the model travels to the handles through their own worker threads (under mu): a handle is only ever touched by its worker
    oct_phmm_error_model pending_model {}; bool pending_has_model = false; uint64_t model_version = 0; std::vector<uint64_t> worker_version;
    std::shared_ptr<const em::CustomIndelModel> pending_custom;     // oct_phmm_server_set_custom_error_model
    // OCT_PHMM_SERVER_PROFILE=1: where a worker's time goes (ns, summed over workers), printed by oct_phmm_server_destroy
    bool profile = tune::server_profile();
    std::atomic<uint64_t> ns_idle {0}, ns_concat {0}, ns_begin {0}, ns_end {0}, ns_scatter {0}, ns_single {0};
    static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::vector<int> device_of;                          // worker -> index into the device list

    static uint32_t rows_of(const oct_phmm_reads* R) { return R->row_offsets ? R->n_rows : R->n_reads; }
    static void answer(Request* q) { std::lock_guard<std::mutex> lk(q->m); q->done = true; q->cv.notify_one(); }      // under the call's own lock: the request lives on its caller's stack

    void serve_one(oct_phmm_handle* h, Request* q)
    {
        try { q->rc = oct_phmm_populate(h, q->R, q->H, nullptr, q->flank, q->pos, q->out, &q->st); }
        catch (const std::exception&) { q->rc = fail(&q->st, OCT_PHMM_EHIP, "host allocation"); }
    }

    // The calls of one device batch concatenated into one flat batch with a region per call. The buffers belong to a worker's slot and keep their capacity from batch to batch
    // (round 4 grew fresh std::strings and vectors per batch, element by element: a third of the 0.6 - 0.8 ms a worker spent between two of its batches).
    struct Concat {
        std::vector<char> rb, hb, mf, mr; std::vector<uint8_t> rq, mq, rv, has_flank, sub; std::vector<uint32_t> roff, hoff, row_off, reg_rows, reg_haps;
        std::vector<int64_t> rbeg, hbeg; std::vector<int8_t> go, ge, pf, pr; std::vector<oct_phmm_flank_state> fl;
        std::vector<double> spill;                        // results of a batch of several slices (the landing zone of a one-slice batch is read in place)
        oct_phmm_reads R {}; oct_phmm_haplotypes H {}; oct_phmm_regions G {}; size_t n_out = 0;
        template <class V, class T> static void put(V& v, const T* src, size_t n) { const size_t o = v.size(); v.resize(o + n); if (n) memcpy(v.data() + o, src, n * sizeof(T)); }
        void build(const std::vector<Request*>& qs)
        {
            bool any_sub = false, templates = false;
            size_t nb = 0, hn = 0, nr = 0, nh = 0, nrows = 0;
            for (Request* q : qs) {
                if (q->H->substitution_mask) any_sub = true;
                if (q->R->row_offsets) templates = true;
                nb += q->R->n_reads ? q->R->offsets[q->R->n_reads] : 0; hn += q->H->n_haps ? q->H->offsets[q->H->n_haps] : 0; nr += q->R->n_reads; nh += q->H->n_haps; nrows += rows_of(q->R);
            }
            const bool given = qs.front()->H->gap_open != nullptr;   // (a device batch holds either calls with vectors or calls without, run())
            for (auto* v : {&rb, &hb, &mf, &mr}) v->clear();
            for (auto* v : {&rq, &mq, &rv, &has_flank, &sub}) v->clear();
            for (auto* v : {&roff, &hoff, &row_off, &reg_rows, &reg_haps}) v->clear();
            rbeg.clear(); hbeg.clear(); fl.clear(); for (auto* v : {&go, &ge, &pf, &pr}) v->clear();
            rb.reserve(nb); rq.reserve(nb); roff.reserve(nr + 1); mq.reserve(nr); rv.reserve(nr); rbeg.reserve(nr); if (templates) row_off.reserve(nrows + 1);
            hb.reserve(hn); hoff.reserve(nh + 1); hbeg.reserve(nh); if (any_sub && !given) sub.reserve(hn);
            if (given) { go.reserve(hn); ge.reserve(hn); mf.reserve(hn); mr.reserve(hn); pf.reserve(hn); pr.reserve(hn); }
            roff.push_back(0); hoff.push_back(0); row_off.push_back(0); reg_rows.assign(1, 0); reg_haps.assign(1, 0);
            n_out = 0;
            for (Request* q : qs) {
                const oct_phmm_reads* Rq = q->R; const oct_phmm_haplotypes* Hq = q->H;
                const uint32_t b1 = Rq->n_reads ? Rq->offsets[Rq->n_reads] : 0, h1 = Hq->n_haps ? Hq->offsets[Hq->n_haps] : 0;
                const uint32_t rbase = roff.back() - (Rq->n_reads ? Rq->offsets[0] : 0), hbase = hoff.back() - (Hq->n_haps ? Hq->offsets[0] : 0);
                const uint32_t rb0 = Rq->n_reads ? Rq->offsets[0] : 0, hb0 = Hq->n_haps ? Hq->offsets[0] : 0;
                put(rb, Rq->bases + rb0, b1 - rb0); put(rq, Rq->qualities + rb0, b1 - rb0);
                { const size_t o = roff.size(); roff.resize(o + Rq->n_reads); for (uint32_t r = 0; r < Rq->n_reads; ++r) roff[o + r] = rbase + Rq->offsets[r + 1]; }
                put(mq, Rq->mapping_quality, Rq->n_reads); put(rv, Rq->reverse_strand, Rq->n_reads); put(rbeg, Rq->ref_begin, Rq->n_reads);
                const uint32_t read0 = (uint32_t)mq.size() - Rq->n_reads;
                if (templates) for (uint32_t row = 0; row < rows_of(Rq); ++row) row_off.push_back(read0 + (Rq->row_offsets ? Rq->row_offsets[row + 1] : row + 1));
                put(hb, Hq->bases + hb0, h1 - hb0);
                { const size_t o = hoff.size(); hoff.resize(o + Hq->n_haps); for (uint32_t k = 0; k < Hq->n_haps; ++k) hoff[o + k] = hbase + Hq->offsets[k + 1]; }
                put(hbeg, Hq->ref_begin, Hq->n_haps);
                if (any_sub && !given) { if (Hq->substitution_mask) put(sub, Hq->substitution_mask + hb0, h1 - hb0); else sub.resize(sub.size() + (h1 - hb0), (uint8_t)0); }
                if (given) {
                    put(go, Hq->gap_open + hb0, h1 - hb0); put(ge, Hq->gap_extend + hb0, h1 - hb0); put(mf, Hq->snv_mask_fwd + hb0, h1 - hb0); put(mr, Hq->snv_mask_rev + hb0, h1 - hb0);
                    put(pf, Hq->snv_prior_fwd + hb0, h1 - hb0); put(pr, Hq->snv_prior_rev + hb0, h1 - hb0);
                }
                reg_rows.push_back(reg_rows.back() + rows_of(Rq)); reg_haps.push_back(reg_haps.back() + Hq->n_haps);
                has_flank.push_back(q->flank ? 1 : 0); fl.push_back(q->flank ? *q->flank : oct_phmm_flank_state {0, 0});
                n_out += (size_t)rows_of(Rq) * Hq->n_haps;
            }
            const uint32_t n_reads = (uint32_t)mq.size(), n_rows = reg_rows.back();
            R = oct_phmm_reads {n_reads, rb.data(), rq.data(), roff.data(), mq.data(), rv.data(), rbeg.data(), templates ? n_rows : 0, templates ? row_off.data() : nullptr};
            H = oct_phmm_haplotypes {(uint32_t)hbeg.size(), hb.data(), hoff.data(), hbeg.data(), given ? go.data() : nullptr, given ? ge.data() : nullptr,
                                     given ? mf.data() : nullptr, given ? pf.data() : nullptr, given ? mr.data() : nullptr, given ? pr.data() : nullptr,
                                     !given && any_sub ? sub.data() : nullptr};
            G = oct_phmm_regions {(uint32_t)qs.size(), reg_rows.data(), reg_haps.data(), has_flank.data(), fl.data()};
            if (spill.size() < n_out + 1) spill.resize(n_out + 1);
        }
    };
    // one device batch between populate_begin and populate_end
    struct Flight { std::vector<Request*> qs; PopulateCall pc; oct_phmm_handle* h = nullptr; int slot = 0; bool active = false; };

    // gather -> check -> pack -> enqueue; the calls' arrays are not read after this returns (upload_impl packed them into the handle's pinned image)
    bool begin_many(oct_phmm_handle* h, Concat& c, std::vector<Request*>& qs, Flight& f)
    {
        const uint64_t t0 = profile ? now_ns() : 0;
        c.build(qs);
        const uint64_t t1 = profile ? now_ns() : 0;
        oct_phmm_status st;
        InputFacts all; all.dirty = 0; bool have = true;
        for (Request* q : qs) { if (!q->have_facts) { have = false; break; } all.merge(q->facts); all.have_haps = true; }
        const int rc = populate_begin(h, &c.R, &c.H, &c.G, nullptr, nullptr, c.spill.data(), &st, &f.pc, have ? &all : nullptr);
        if (profile) { ns_concat += t1 - t0; ns_begin += now_ns() - t1; }
        if (rc != OCT_PHMM_OK) return false;
        f.qs = std::move(qs); f.h = h; f.active = true;
        return true;
    }
    // wait -> scatter -> wake the callers. One region's error must not reach the others: a failed batch is answered call by call.
    void end_many(Concat& c, Flight& f)
    {
        const uint64_t t0 = profile ? now_ns() : 0;
        oct_phmm_status st; const double* in_place = nullptr;
        const int rc = populate_end(f.h, &f.pc, c.spill.data(), &st, &in_place);
        const uint64_t t1 = profile ? now_ns() : 0;
        if (rc != OCT_PHMM_OK) { for (Request* q : f.qs) { serve_one(f.h, q); answer(q); } }
        else {
            const double* p = in_place ? in_place : c.spill.data();
            for (Request* q : f.qs) {
                const size_t n = (size_t)rows_of(q->R) * q->H->n_haps;
                if (n) memcpy(q->out, p, n * sizeof(double));
                p += n; q->rc = OCT_PHMM_OK; memset(&q->st, 0, sizeof(q->st));
                answer(q);
            }
        }
        if (profile) { ns_end += t1 - t0; ns_scatter += now_ns() - t1; }
        f.qs.clear(); f.active = false;
    }

    // A worker = two threads around kSlots handles. The GATHERER takes calls, concatenates, checks, packs and enqueues them on a free slot (begin_many: no wait on the
    // device for device-sized batches); the FINISHER waits for the slots' batches in the order they were begun, scatters the results and wakes the callers at once -
    // a batch that has left the device is never held up by the next one's preparation (round 5's first pipelined form finished a batch only between two steps of the
    // gatherer: up to 0.6 ms of a ~3 ms call). OCT_PHMM_SERVER_PIPELINE=0: one slot, i.e. round 4's take - run - answer loop.
    struct Slot { Concat concat; Flight flight; bool flying = false; };
    struct Worker {
        Slot slot[kSlots]; std::mutex m; std::condition_variable cv_free, cv_flying; bool quit = false; std::thread finisher;
        int n_flying() const { int n = 0; for (const Slot& s : slot) n += s.flying ? 1 : 0; return n; }
    };
    std::vector<std::unique_ptr<Worker>> wk;
#if defined(OCTPHMM_SIM)
    std::mutex sim_mu;                                     // the wave simulator runs one kernel at a time: the workers of several "devices" take turns
#endif

    void finish_loop(int w)
    {
        Worker& W = *wk[(size_t)w];
        const int n_slots = kSlots;
        for (int k = 0;; k = (k + 1) % n_slots) {          // slots fly in turn
            {
                std::unique_lock<std::mutex> lk(W.m);
                W.cv_flying.wait(lk, [&] { return W.quit || W.slot[k].flying; });
                if (!W.slot[k].flying) return;              // (quit, and nothing left in the air)
            }
            {
#if defined(OCTPHMM_SIM)
                std::lock_guard<std::mutex> sim_lk(sim_mu);
#endif
                end_many(W.slot[k].concat, W.slot[k].flight);
            }
            { std::lock_guard<std::mutex> lk(W.m); W.slot[k].flying = false; }
            W.cv_free.notify_all();
            cv_work.notify_all();                           // (a gatherer that waits for calls OR for its batch to land)
        }
    }

    void run(int w)
    {
        Worker& W = *wk[(size_t)w];
        const int n_slots = kSlots;
        oct_phmm_handle* hslot[kSlots]; for (int k = 0; k < kSlots; ++k) hslot[k] = hs[(size_t)w * kSlots + k];
        std::deque<std::vector<Request*>> groups;          // batches taken from the queue that wait for a slot
        int next_slot = 0; size_t last_batch = 1;
        bool w_busy = false;                               // counted in busy_workers
        auto flying = [&] { std::lock_guard<std::mutex> lk(W.m); return W.n_flying(); };
        auto wait_all_landed = [&] { std::unique_lock<std::mutex> lk(W.m); W.cv_free.wait(lk, [&] { return W.n_flying() == 0; }); };
        for (;;) {
            if (groups.empty()) {
                std::vector<Request*> take;
                {
                    const uint64_t t_idle = profile ? now_ns() : 0;
                    std::unique_lock<std::mutex> lk(mu);
                    if (flying() == 0) {
                        if (w_busy) { --busy_workers; w_busy = false; }
                        cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                    } else {
                        // A batch of this worker is on the device: the calls that have arrived since are the first of the next burst (its own callers come back when it
                        // lands, the other workers' when theirs do). A batch of two costs the device what one of ten does, so there is no hurry - but the next batch should
                        // be enqueued when this one ends. Wait until as many calls wait as the last batch held (two batches of a size, turn and turn about: what a steady
                        // crowd of callers settles into), or until nothing of this worker's is on the device any more (few callers: take what has come).
                        const size_t want = std::max<size_t>(1, std::min<size_t>(max_regions, last_batch));
                        while (!stop && queue.size() < want && flying() > 0)
                            cv_work.wait_for(lk, std::chrono::microseconds(50), [&] { return stop || queue.size() >= want; });
                    }
                    if (profile) ns_idle += now_ns() - t_idle;
                    if (queue.empty() && stop) { lk.unlock(); wait_all_landed(); return; }
                    if (worker_version[(size_t)w] != model_version) {      // a new error model since this worker's last batch: install it (nothing of ours in the air) before taking calls
                        lk.unlock(); wait_all_landed(); lk.lock();
                        for (int k = 0; k < kSlots; ++k) {
                            if (pending_has_model && pending_custom) { const oct_phmm_custom_indel_model cm {pending_custom}; oct_phmm_set_custom_error_model(hslot[k], &cm, &pending_model); }
                            else oct_phmm_set_error_model(hslot[k], pending_has_model ? &pending_model : nullptr);
                        }
                        worker_version[(size_t)w] = model_version;
                    }
                    while (!queue.empty() && take.size() < max_regions) { take.push_back(queue.front()); queue.pop_front(); }
                    if (!take.empty() && !w_busy) { ++busy_workers; w_busy = true; }
                    n_calls += take.size(); n_calls_by_device[(size_t)device_of[(size_t)w]] += take.size();
                }
                if (take.empty()) continue;
                std::vector<Request*> batchable, batchable_gen, single;       // calls that leave their penalty vectors to the library batch among themselves
                for (Request* q : take) (q->pos || !q->R || !q->H || !q->R->n_reads || !q->H->n_haps ? single : q->H->gap_open ? batchable : batchable_gen).push_back(q);
                if (!single.empty()) {                         // calls with positions of their own, empty calls: one by one, on a slot that is on the ground
                    const uint64_t t_single = profile ? now_ns() : 0;
                    { std::unique_lock<std::mutex> lk(W.m); W.cv_free.wait(lk, [&] { return !W.slot[next_slot].flying; }); }
                    for (Request* q : single) {
#if defined(OCTPHMM_SIM)
                        std::lock_guard<std::mutex> sim_lk(sim_mu);
#endif
                        ++n_batches; serve_one(hslot[next_slot], q); answer(q);
                    }
                    if (profile) ns_single += now_ns() - t_single;
                }
                if (!batchable.empty()) groups.push_back(std::move(batchable));
                if (!batchable_gen.empty()) groups.push_back(std::move(batchable_gen));
                if (groups.empty()) continue;
            }
            // the next batch goes onto the next slot as soon as that slot's last batch has landed; the other slot's batch keeps computing meanwhile
            std::vector<Request*> qs = std::move(groups.front()); groups.pop_front();
            const int k = next_slot;
            { std::unique_lock<std::mutex> lk(W.m); W.cv_free.wait(lk, [&] { return !W.slot[k].flying; }); }
            Slot& S = W.slot[k];
            S.flight = Flight {}; S.flight.slot = k;
            last_batch = qs.size();
            bool started = false;
            ++n_batches;
            {
#if defined(OCTPHMM_SIM)
                std::lock_guard<std::mutex> sim_lk(sim_mu);
#endif
                try { started = begin_many(hslot[k], S.concat, qs, S.flight); } catch (const std::exception&) { started = false; }      // e.g. bad_alloc while concatenating
            }
            if (started) {
                { std::lock_guard<std::mutex> lk(W.m); S.flying = true; }
                W.cv_flying.notify_all();
                next_slot = (k + 1) % n_slots;
            } else for (Request* q : qs) {                     // the batch could not be uploaded as one (an error in one of its regions, no memory): every call on its own
#if defined(OCTPHMM_SIM)
                std::lock_guard<std::mutex> sim_lk(sim_mu);
#endif
                serve_one(hslot[k], q); answer(q);
            }
        }
    }
};

extern "C" int oct_phmm_server_create_multi(const oct_phmm_config* cfg, const int32_t* device_ids, uint32_t n_devices,
                                            uint32_t max_regions_per_batch, oct_phmm_server** out)
{
    if (!out || !cfg || !device_ids || !n_devices) return OCT_PHMM_EINVAL;
    *out = nullptr;
    oct_phmm_server* s = new (std::nothrow) oct_phmm_server();
    if (!s) return OCT_PHMM_EHIP;
    for (uint32_t dv = 0; dv < n_devices; ++dv) {
        oct_phmm_config c = *cfg; c.device_id = device_ids[dv];
        int n_workers = oct_phmm_server::kWorkers;
        { long long v; if (tune::number("OCT_PHMM_SERVER_WORKERS", &v) && v >= 1 && v <= 8) n_workers = (int)v; }      // A/B switch: device queues (worker threads + handles) per device (the simulator's
                                                                                                                         // default is 1; with more, its workers take turns at sim_mu - the ThreadSanitizer run uses 2)
        for (int w = 0; w < n_workers * oct_phmm_server::kSlots; ++w) {
            oct_phmm_handle* h = nullptr;
            const int rc = oct_phmm_create(&c, &h);
            if (rc != OCT_PHMM_OK) { for (auto* k : s->hs) oct_phmm_destroy(k); delete s; return rc; }
            // A worker's traceback scratch: capped (a device batch that needs more runs its traceback lists in chunks) and reserved now - a multi-gigabyte
            // hipMalloc in the middle of a run stalled every caller for up to a second, once per worker and growth step
            { long long gb = 4; tune::number("OCT_PHMM_SERVER_BP_BUDGET_GB", &gb); if (gb >= 1) h->bp_budget = std::min<size_t>(h->bp_budget, (size_t)gb << 30); }
#if !defined(OCTPHMM_SIM)
            // (all of it: device-sized batches provision two traceback tasks per pair, and a worker's biggest batch comes late in a run. A device that other processes - or this
            // process's own per-thread handles - have filled gives what it has: the budget is halved until the reservation succeeds, and the handle then lives within that, its
            // bigger batches running their traceback lists in chunks, instead of failing or trimming its neighbours' caches at the first big call)
            while (!ensure_bp(h, 0, h->bp_budget) && h->bp_budget > ((size_t)256 << 20)) h->bp_budget >>= 1;
#endif
            s->hs.push_back(h); if (w % oct_phmm_server::kSlots == 0) s->device_of.push_back((int)dv);
        }
    }
    const size_t n_workers_total = s->hs.size() / oct_phmm_server::kSlots;
    s->n_calls_by_device.assign(n_devices, 0); s->worker_version.assign(n_workers_total, 0);
    if (max_regions_per_batch) s->max_regions = max_regions_per_batch;
    for (size_t w = 0; w < n_workers_total; ++w) s->wk.emplace_back(new oct_phmm_server::Worker());
    for (size_t w = 0; w < n_workers_total; ++w) {
        s->wk[w]->finisher = std::thread([s, w] { s->finish_loop((int)w); });
        s->workers.emplace_back([s, w] { s->run((int)w); });
    }
    *out = s;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_server_create(const oct_phmm_config* cfg, uint32_t max_regions_per_batch, oct_phmm_server** out)
{
    if (!cfg) return OCT_PHMM_EINVAL;
    const int32_t dev = cfg->device_id;
    return oct_phmm_server_create_multi(cfg, &dev, 1, max_regions_per_batch, out);
}

extern "C" void oct_phmm_server_destroy(oct_phmm_server* s)
{
    if (!s) return;
    { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; }
    s->cv_work.notify_all();
    for (auto& t : s->workers) if (t.joinable()) t.join();          // (every gatherer leaves with nothing of its own in the air)
    for (auto& W : s->wk) { { std::lock_guard<std::mutex> lk(W->m); W->quit = true; } W->cv_flying.notify_all(); if (W->finisher.joinable()) W->finisher.join(); }
    for (auto* h : s->hs) oct_phmm_destroy(h);
    if (s->profile)
        fprintf(stderr, "{\"server_profile_ms\": {\"workers\": %zu, \"calls\": %llu, \"batches\": %llu, \"idle\": %.2f, \"concat\": %.2f, \"check_pack_enqueue\": %.2f, "
                        "\"wait_for_results\": %.2f, \"scatter_and_wake\": %.2f, \"single_calls\": %.2f}}\n", s->workers.size(), (unsigned long long)s->n_calls.load(), (unsigned long long)s->n_batches.load(),
                s->ns_idle / 1e6, s->ns_concat / 1e6, s->ns_begin / 1e6, s->ns_end / 1e6, s->ns_scatter / 1e6, s->ns_single / 1e6);
    delete s;
}

extern "C" int oct_phmm_server_populate(oct_phmm_server* s, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
                                        const oct_phmm_flank_state* flank, const oct_phmm_positions* positions, double* out, oct_phmm_status* status)
{
    if (!s || !reads || !haps) return fail(status, OCT_PHMM_EINVAL, "null argument");
    // the workers concatenate queued calls before the library proper validates them: a malformed call is answered here, not in a worker thread
    if ((reads->n_reads && (!reads->bases || !reads->qualities || !reads->offsets || !reads->mapping_quality || !reads->reverse_strand || !reads->ref_begin))
        || (haps->n_haps && (!haps->bases || !haps->offsets || !haps->ref_begin)))
        return fail(status, OCT_PHMM_EINVAL, "null array");
    {
        const int n_vec = (haps->gap_open ? 1 : 0) + (haps->gap_extend ? 1 : 0) + (haps->snv_mask_fwd ? 1 : 0) + (haps->snv_prior_fwd ? 1 : 0)
                        + (haps->snv_mask_rev ? 1 : 0) + (haps->snv_prior_rev ? 1 : 0);
        if (haps->n_haps && n_vec != 6 && !(n_vec == 0 && s->has_model.load())) return fail(status, OCT_PHMM_EINVAL, "null array");
    }
    if ((reads->n_reads && !monotone(reads->offsets, reads->n_reads)) || (haps->n_haps && !monotone(haps->offsets, haps->n_haps)))
        return fail(status, OCT_PHMM_EINVAL, "offsets not monotone");
    {
        const uint32_t rows = reads->row_offsets ? reads->n_rows : reads->n_reads;
        if (reads->row_offsets && (!monotone(reads->row_offsets, rows) || reads->row_offsets[0] != 0 || reads->row_offsets[rows] != reads->n_reads))
            return fail(status, OCT_PHMM_EINVAL, "row_offsets must partition the reads");
        if (!out && (size_t)rows * haps->n_haps) return fail(status, OCT_PHMM_EINVAL, "null output");
    }
    oct_phmm_server::Request q; q.R = reads; q.H = haps; q.flank = flank; q.pos = positions; q.out = out; memset(&q.st, 0, sizeof(q.st));
    if (haps->gap_open && reads->n_reads && haps->n_haps) {      // everything an upload has to know about the call's bytes: looked up HERE, on the caller's thread (the workers are what a busy server waits for)
        q.facts.dirty = 0; q.facts.have_haps = true;
        facts_of_reads(reads, 0, reads->n_reads, true, &q.facts);
        facts_of_haps(haps, haps->offsets[0], haps->offsets[haps->n_haps], true, &q.facts);
        q.have_facts = true;
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->stop) return fail(status, OCT_PHMM_EINVAL, "server is shutting down");
        s->queue.push_back(&q);
        s->cv_work.notify_all();                                // (all: one notification could land on a gatherer whose predicate is "as many calls as my last batch" and be lost on it while an idle worker - of another GPU, say - sleeps on)
    }
    { std::unique_lock<std::mutex> lk(q.m); q.cv.wait(lk, [&] { return q.done; }); }      // (the call's own lock: a batch's callers do not queue for the server's to return)
    if (status) *status = q.st;
    return q.rc;
}

extern "C" int oct_phmm_server_set_error_model(oct_phmm_server* s, const oct_phmm_error_model* model)
{
    if (!s) return OCT_PHMM_EINVAL;
    if (model && !model_is_valid(model)) return OCT_PHMM_EINVAL;
    std::lock_guard<std::mutex> lk(s->mu);
    s->pending_has_model = model != nullptr;
    s->pending_custom.reset();
    if (model) s->pending_model = *model;
    ++s->model_version;                                    // every worker installs it on its own handle before its next batch (run())
    s->has_model = model != nullptr;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_server_set_custom_error_model(oct_phmm_server* s, const oct_phmm_custom_indel_model* indel, const oct_phmm_error_model* snv)
{
    if (!s || !indel) return OCT_PHMM_EINVAL;
    oct_phmm_error_model dflt;
    if (!snv) { oct_phmm_error_model_default(&dflt); snv = &dflt; }
    if (!model_is_valid(snv)) return OCT_PHMM_EINVAL;
    std::lock_guard<std::mutex> lk(s->mu);
    s->pending_has_model = true; s->pending_model = *snv; s->pending_custom = indel->m;
    ++s->model_version;
    s->has_model = true;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_server_device_calls(const oct_phmm_server* s, uint64_t* calls_by_device, uint32_t n_devices)
{
    if (!s || !calls_by_device) return OCT_PHMM_EINVAL;
    std::lock_guard<std::mutex> lk(const_cast<oct_phmm_server*>(s)->mu);
    for (uint32_t i = 0; i < n_devices; ++i) calls_by_device[i] = i < s->n_calls_by_device.size() ? s->n_calls_by_device[i] : 0;
    return OCT_PHMM_OK;
}

extern "C" int oct_phmm_server_stats(const oct_phmm_server* s, uint64_t* n_calls, uint64_t* n_batches)
{
    if (!s) return OCT_PHMM_EINVAL;
    std::lock_guard<std::mutex> lk(const_cast<oct_phmm_server*>(s)->mu);
    if (n_calls) *n_calls = s->n_calls.load();
    if (n_batches) *n_batches = s->n_batches.load();
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// realignment: best alignment per (read, haplotype) pair
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_align(oct_phmm_handle* h, const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
                              const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
                              const oct_phmm_positions* positions, oct_phmm_alignments* out, oct_phmm_status* status)
{
    if (!out || !out->mapping_position || !out->likelihood || !out->n_cigar_ops || (!out->cigar && out->max_cigar_ops))
        return fail(status, OCT_PHMM_EINVAL, "null output");
    oct_phmm_batch* b = nullptr;
    int rc = upload_impl(h, reads, haps, regions, flank, positions, &b, status, true, out->max_cigar_ops, true);
    struct Guard { oct_phmm_handle* h; oct_phmm_batch*& b; ~Guard() { oct_phmm_batch_free(h, b); } } guard {h, b};
    if (rc != OCT_PHMM_OK) return rc;
    rc = oct_phmm_batch_run(h, b, status);
    if (rc == OCT_PHMM_OK) rc = oct_phmm_batch_wait(h, b, status);
    if (rc != OCT_PHMM_OK) return rc;
    const size_t np = (size_t)b->n_pairs, cap = b->cig_cap;
    uint32_t flags = 0;
    std::vector<uint32_t> ops(np * cap + 1);
    RT(rt::d2h(&flags, b->d_err_flags, sizeof(flags), h->stream));
    RT(rt::d2h(out->likelihood, b->d_aln_lik, np * sizeof(double), h->stream));
    RT(rt::d2h(out->mapping_position, b->d_aln_mpos, np * sizeof(uint32_t), h->stream));
    RT(rt::d2h(out->n_cigar_ops, b->d_aln_n, np * sizeof(uint32_t), h->stream));
    RT(rt::d2h(ops.data(), b->d_aln_ops, np * cap * sizeof(uint32_t), h->stream));
    h->last_align_counts.assign(np, 0); h->last_align_device_map = b->device_map;
    RT(rt::d2h(h->last_align_counts.data(), b->d.npos, np, h->stream));        // oct_phmm_align_candidate_counts: did the mapper's output reach max_mapping_positions?
    RT(rt::stream_sync(h->stream));
    if (flags & 1u) return fail(status, OCT_PHMM_EOVERFLOW, "Pair HMM alignment overflowed");
    uint32_t needed = 0;
    for (size_t e = 0; e < np; ++e) {                       // the device wrote each alignment last column first
        const uint32_t n = out->n_cigar_ops[e];
        if (n > out->max_cigar_ops) { needed = std::max(needed, n); continue; }
        for (uint32_t k = 0; k < n; ++k) out->cigar[e * (size_t)out->max_cigar_ops + k] = ops[e * cap + (n - 1 - k)];
    }
    if (needed) { fail(status, OCT_PHMM_EINVAL, "max_cigar_ops too small"); if (status) status->required_extension = needed; return OCT_PHMM_EINVAL; }
    return ok(status);
}

extern "C" int oct_phmm_align_candidate_counts(const oct_phmm_handle* h, uint8_t* counts, size_t n_pairs, uint32_t* n_saturated)
{
    if (!h || (!counts && !n_saturated)) return OCT_PHMM_EINVAL;
    if (counts && n_pairs != h->last_align_counts.size()) return OCT_PHMM_EINVAL;
    if (counts && n_pairs) memcpy(counts, h->last_align_counts.data(), n_pairs);
    if (n_saturated) {
        uint32_t n = 0;
        if (h->last_align_device_map) for (uint8_t c : h->last_align_counts) if ((int)c >= h->cfg.max_mapping_positions) ++n;
        *n_saturated = n;
    }
    return OCT_PHMM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// test seam: raw band kernel on explicit windows
// ---------------------------------------------------------------------------------------------------------------
extern "C" int oct_phmm_align_windows(oct_phmm_handle* h, uint32_t n,
                                      const char* truth, const uint32_t* truth_offsets,
                                      const char* target, const uint8_t* qualities, const uint32_t* target_offsets,
                                      const int8_t* gap_open, const int8_t* gap_extend, int32_t gap_extend_scalar,
                                      const char* snv_mask, const int8_t* snv_prior,
                                      int32_t nuc_prior, int32_t traceback,
                                      int32_t* scores, int32_t* first_pos,
                                      char* align1, char* align2, const uint32_t* align_offsets,
                                      const int32_t* lhs_flank, const int32_t* rhs_flank,
                                      int32_t* flank_score, int32_t* target_mask_size,
                                      oct_phmm_status* status)
{
    if (!h || !truth || !truth_offsets || !target || !qualities || !target_offsets || !gap_open || !scores
        || (!!snv_mask != !!snv_prior) || (traceback && (!first_pos || !align1 || !align2 || !align_offsets))
        || (lhs_flank && (!traceback || !rhs_flank || !flank_score || !target_mask_size || !snv_mask)))
        return fail(status, OCT_PHMM_EINVAL, "null argument");
    if (!n) return ok(status);
    const uint32_t B = (uint32_t)h->band;
    const uint32_t n_truth = truth_offsets[n];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t L = truth_offsets[i + 1] - truth_offsets[i], T = target_offsets[i + 1] - target_offsets[i];
        if (T == 0 || L != T + 2 * B - 1) return fail(status, OCT_PHMM_EINVAL, "truth_len must equal target_len + 2*band - 1");
        if (traceback && align_offsets[i + 1] - align_offsets[i] < 2 * (T + B) + 1) return fail(status, OCT_PHMM_EINVAL, "alignment buffer too small");
    }
    // the windows become one-off "haplotypes", the targets "reads", one task each at offset 0
    std::vector<int8_t> ge_arr, prior_arr; std::vector<uint8_t> zeros8(n, 0), mapq(n, 0); std::vector<int64_t> zeros64(n, 0);
    if (!gap_extend) { ge_arr.assign(n_truth, (int8_t)gap_extend_scalar); gap_extend = ge_arr.data(); }
    if (!snv_mask) { snv_mask = truth; prior_arr.assign(n_truth, 0); snv_prior = prior_arr.data(); }   // mask == truth never fires (cost 0 on equality)
    oct_phmm_reads R {}; R.n_reads = n; R.bases = target; R.qualities = qualities; R.offsets = target_offsets;
    R.mapping_quality = mapq.data(); R.reverse_strand = zeros8.data(); R.ref_begin = zeros64.data(); R.n_rows = n; R.row_offsets = nullptr;
    oct_phmm_haplotypes H {}; H.n_haps = n; H.bases = truth; H.offsets = truth_offsets; H.ref_begin = zeros64.data();
    H.gap_open = gap_open; H.gap_extend = gap_extend; H.snv_mask_fwd = snv_mask; H.snv_prior_fwd = snv_prior;
    H.snv_mask_rev = snv_mask; H.snv_prior_rev = snv_prior;
    // regions: every window is its own region (one row x one haplotype), no pairs beyond the diagonal
    std::vector<uint32_t> reg(n + 1); for (uint32_t i = 0; i <= n; ++i) reg[i] = i;
    oct_phmm_regions RG {}; RG.n_regions = n; RG.row_offsets = reg.data(); RG.hap_offsets = reg.data(); RG.has_flank = nullptr; RG.flank = nullptr;
    std::vector<uint64_t> poff(n + 1, 0); uint32_t dummy_pos = 0;
    oct_phmm_positions P {poff.data(), &dummy_pos};
    oct_phmm_batch* b = nullptr;
    int rc = oct_phmm_batch_upload(h, &R, &H, &RG, nullptr, &P, &b, status);
    if (rc != OCT_PHMM_OK) return rc;
    struct Guard { oct_phmm_handle* h; oct_phmm_batch* b; std::vector<void*> extra; ~Guard() { rt::stream_sync(h->stream); for (void* p : extra) h->pool.release(p); oct_phmm_batch_free(h, b); } } guard {h, b, {}};
    rt::Stream s = h->stream;
    const uint32_t G = b->stream ? (B < 64 ? 64u / (uint32_t)B : 1u) : (h->wide ? 1u : 2u) * (64 / B);
    // route each window to the fast or generic kernel exactly as k_classify would
    std::vector<uint8_t> racgt(n); std::vector<uint32_t> hclean(n);
    RT(rt::d2h(racgt.data(), b->d.racgt, n, s)); RT(rt::d2h(hclean.data(), b->d.hclean, n * sizeof(uint32_t), s)); RT(rt::stream_sync(s));
    std::vector<DevTask> tasks[2]; std::vector<uint32_t> origin[2];
    // every window is its own haplotype and a DP task group must stay within one haplotype: give each window a whole
    // group (one real task + G-1 padding copies). This is a test seam, not the throughput path.
    for (uint32_t i = 0; i < n; ++i) {
        const int gen = ((h->wide || b->stream) && !b->multi_wave && !b->rows32) || !(racgt[i] && hclean[i]);
        tasks[gen].push_back(DevTask {i, i, i, 0}); origin[gen].push_back(i);
        for (uint32_t k = 1; k < G; ++k) tasks[gen].push_back(DevTask {kPadTask, i, i, 0});
    }
    std::vector<int32_t> init(n, kNoScore);
    RT(rt::h2d(b->d.pair_best, init.data(), n * sizeof(int32_t), s));
    for (int gen = 0; gen < 2; ++gen) {
        std::vector<DevTask>& t = tasks[gen];
        if (t.empty()) continue;
        const uint32_t nt = (uint32_t)t.size(), real = nt / G;
        void* d_tasks = nullptr; RT(h->pool.alloc(&d_tasks, nt * sizeof(DevTask))); guard.extra.push_back(d_tasks);
        RT(rt::h2d(d_tasks, t.data(), nt * sizeof(DevTask), s));
        void* d_ends = nullptr; RT(h->pool.alloc(&d_ends, nt * sizeof(TraceEnd))); guard.extra.push_back(d_ends);
        WalkParams w {}; std::vector<uint32_t> aoff(nt + 1, 0); std::vector<int32_t> l(nt, 0), r(nt, 0);
        void *d_fp = nullptr, *d_a1 = nullptr, *d_a2 = nullptr, *d_aoff = nullptr, *d_l = nullptr, *d_r = nullptr, *d_fl = nullptr, *d_ms = nullptr;
        size_t aln_bytes = 0;
        if (traceback) {
            for (uint32_t j = 0; j < nt; ++j) {
                const uint32_t i = t[j].read; const uint32_t T = target_offsets[i + 1] - target_offsets[i];
                aoff[j + 1] = aoff[j] + 2 * (T + B) + 1;
                if (lhs_flank) { l[j] = lhs_flank[i]; r[j] = rhs_flank[i]; }
            }
            aln_bytes = aoff[nt];
            RT(h->pool.alloc(&d_fp, nt * sizeof(int32_t))); guard.extra.push_back(d_fp);
            RT(h->pool.alloc(&d_a1, aln_bytes)); guard.extra.push_back(d_a1); RT(h->pool.alloc(&d_a2, aln_bytes)); guard.extra.push_back(d_a2);
            RT(rt::dev_memset(d_a1, 0, aln_bytes, s)); RT(rt::dev_memset(d_a2, 0, aln_bytes, s));
            RT(h->pool.alloc(&d_aoff, (nt + 1) * sizeof(uint32_t))); guard.extra.push_back(d_aoff);
            RT(rt::h2d(d_aoff, aoff.data(), (nt + 1) * sizeof(uint32_t), s));
            w.out_first_pos = (int32_t*)d_fp; w.out_align1 = (char*)d_a1; w.out_align2 = (char*)d_a2; w.out_align_off = (const uint32_t*)d_aoff;
            if (lhs_flank) {
                RT(h->pool.alloc(&d_l, nt * 4)); guard.extra.push_back(d_l); RT(h->pool.alloc(&d_r, nt * 4)); guard.extra.push_back(d_r);
                RT(h->pool.alloc(&d_fl, nt * 4)); guard.extra.push_back(d_fl); RT(h->pool.alloc(&d_ms, nt * 4)); guard.extra.push_back(d_ms);
                RT(rt::h2d(d_l, l.data(), nt * 4, s)); RT(rt::h2d(d_r, r.data(), nt * 4, s));
                w.seam_lhs = (const int32_t*)d_l; w.seam_rhs = (const int32_t*)d_r; w.out_flank = (int32_t*)d_fl; w.out_mask_size = (int32_t*)d_ms;
            }
        }
        const int kind = traceback ? (gen ? kTraceGen : kTraceFast) : (gen ? kScoreGen : kScoreFast);
        rc = run_dp_kind(h, b, 0, kind, (const DevTask*)d_tasks, nt, (TraceEnd*)d_ends, nuc_prior, traceback ? &w : nullptr, status);
        if (rc != OCT_PHMM_OK) return rc;
        if (traceback) {
            std::vector<TraceEnd> ends(nt); std::vector<int32_t> fp(nt), fl(nt), ms(nt); std::vector<char> a1(aln_bytes), a2(aln_bytes);
            RT(rt::d2h(ends.data(), d_ends, nt * sizeof(TraceEnd), s)); RT(rt::d2h(fp.data(), d_fp, nt * 4, s));
            RT(rt::d2h(a1.data(), d_a1, aln_bytes, s)); RT(rt::d2h(a2.data(), d_a2, aln_bytes, s));
            if (lhs_flank) { RT(rt::d2h(fl.data(), d_fl, nt * 4, s)); RT(rt::d2h(ms.data(), d_ms, nt * 4, s)); }
            RT(rt::stream_sync(s));
            for (uint32_t w = 0; w < real; ++w) {
                const uint32_t i = origin[gen][w], j = w * G;
                scores[i] = ends[j].score; first_pos[i] = fp[j];
                const uint32_t len = aoff[j + 1] - aoff[j];
                memcpy(align1 + align_offsets[i], a1.data() + aoff[j], len); memcpy(align2 + align_offsets[i], a2.data() + aoff[j], len);
                if (lhs_flank) { flank_score[i] = fl[j]; target_mask_size[i] = ms[j]; }
            }
        }
    }
    if (!traceback) { RT(rt::d2h(scores, b->d.pair_best, n * sizeof(int32_t), s)); RT(rt::stream_sync(s)); }
    return ok(status);
}
