// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - oct_phmm_batch_run / wait / download / stats: the launch sequence of a step.
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
            // a haplotype's table is one workgroup's work (one histogram in LDS): 1,024 threads from 4 k bases on (a 16 kb haplotype with 256: most of the 0.27 ms the launch took on ccs256x12)
            const uint32_t kt_threads = (b->lh_cap >= 4096 && !b->tables_pending) ? 1024u : 256u, kt_waves = kt_threads / 64;
            const uint32_t hash_blocks = hashes_here ? (uint32_t)(((uint64_t)b->n_reads * d.hash_segs + kt_waves - 1) / kt_waves) : 0;   // one wave per read and 1,024 bases
            if (b->tables_pending) {                        // first step of a device-sized batch: the upload left its table kernel to this launch
                const uint32_t n_kmer_blocks = sl.hap1 - sl.hap0 + hash_blocks;
                OCT_LAUNCH(k_tables, n_kmer_blocks + b->tp_table_blocks + b->tp_flag_blocks + b->tp_rec_blocks, 256, (kKmerBins + 256) * sizeof(uint32_t), s, d, sl.hap0, sl.hap1 - sl.hap0,
                           n_kmer_blocks, b->tp_n_bases, b->tp_table_blocks, b->tp_flag_blocks); RT(rt::launch_ok());
                b->tables_pending = false;
            } else {
            OCT_LAUNCH(k_kmer_tables, sl.hap1 - sl.hap0 + hash_blocks, kt_threads, (kKmerBins + 256) * sizeof(uint32_t), s, d, sl.hap0, sl.hap1 - sl.hap0); RT(rt::launch_ok());
            }
            if (hashes_here) { hash_slice = i; if (S > 1) RT(rt::event_record(b->ev_hashes, s)); }
            else RT(rt::stream_wait_event(s, b->ev_hashes));
            if (b->map_big) {
                const size_t lds = kmer_map_big_lds_bytes(b->lh_cap);
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

