// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - device block pools, the handle and batch objects, the switch table.
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
    // tables_pending: a device-sized batch's per-base DP tables, read flags and read records are made by the FIRST step's table launch (k_tables: one launch with the k-mer tables), not by the upload
    bool tables_pending = false; uint32_t tp_n_bases = 0, tp_table_blocks = 0, tp_flag_blocks = 0, tp_rec_blocks = 0;
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

