// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - the region server: one device queue behind many calling threads.
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

