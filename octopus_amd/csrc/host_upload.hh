// Part of liboct_phmm.so's host side (one translation unit: octopus_amd/csrc/oct_phmm.hip includes this file in place) - oct_phmm_batch_upload: validation, HBM layout, slices, device-sized bounds.
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
        // a walk event holds a window coordinate in 20 bits (phmm_kernels.hpp: kMaxWindowBases; 15 bits, i.e. reads below 32 k bases, until round 6)
        if ((uint64_t)b->t_cap + 2 * (uint32_t)h->band >= kMaxWindowBases) return fail(status, OCT_PHMM_EUNSUPPORTED, "read too long (T + 2B must stay below 2^20: walk events hold 20-bit coordinates)");
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
        if (b->lh_cap >= 65536 || kmer_map_big_lds_bytes(b->lh_cap) > rt::kMaxLdsBytes)      // (16-bit bin tables; k_kmer_map_big's counters are 16-bit halves since round 6: 40 k bases were the limit before)
            return fail(status, OCT_PHMM_EUNSUPPORTED, "haplotype too long for the k-mer mapper (>= 65,536 bases)");
        if (b->t_cap >= 65536) return fail(status, OCT_PHMM_EUNSUPPORTED, "read too long for the k-mer mapper (>= 65,536 bases: a diagonal's votes are counted in 16 bits)");
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
    d.hash_segs = std::max<uint32_t>(1u, (b->t_cap + kHashSegment - 1) / kHashSegment);
    d.bin_idx = nullptr; d.rhash = nullptr; d.bin32 = nullptr; d.hhash = nullptr; d.map_count_only = 0; d.map_stats = 0; d.pair_mm = nullptr; d.rcode = nullptr; d.rcode_words = 0;
    if (!positions) {
        pk.dalloc(&d.bin_idx, (size_t)n_hap_bases + 8);          // (+ 8: k_kmer_map_big fetches a bin's first four entries with one 8-byte load)
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
        if (!b->map_lanes) pk.dalloc(&d.rhash, (size_t)n_read_bases + 1);        // (the lane mapper reads the reads' 2-bit code rows instead)
        if (b->map_lanes) {
            b->map_reads_per_block = (uint32_t)b->map_lanes;
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
        // Window pairing: host-sized batches with >= 2,048 pairs per haplotype (a haplotype's task runs are thousands long there: 100 k reads over ~240 (offset, strand) classes), packed int16 lanes with
        // plain adds, fast-cost flavour; the 20-byte columns must leave the traceback form its three workgroups per CU. OCT_PHMM_PAIRED=0 / 1: off / forced (tests: small batches).
        long long want = -1; tune::number("OCT_PHMM_PAIRED", &want);
        const uint32_t Bw = (uint32_t)h->band;
        const size_t lds_tr = dp_lds_bytes(b->t_cap, b->lh_cap, Bw, true, dp_rec_chunk(b->t_cap, b->lh_cap, Bw, true), true);
        const bool can = !h->wide && !b->stream && b->fast_adds && !align_mode && b->lh_cap <= kPairSortMaxLh && b->n_pairs > 0 && lds_tr <= rt::kMaxLdsBytes;
        b->pair_ok = can && (want >= 0 ? want != 0 : (b->n_pairs > kDslMaxPairs && b->n_pairs / std::max<uint32_t>(1, H->n_haps) >= 2048 && lds_tr * 3 <= rt::kMaxLdsBytes));
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
            // a device-sized batch that maps on the device: the first step's k-mer table launch makes these too (k_tables: upload 0.355-0.375 -> 0.335 ms for 16 regions, one launch fewer;
            // the A/B switch went with the measurement, profiles/EXPERIMENTS.md)
            if (b->dsl && b->device_map && !b->dedup && b->lh_cap < 4096) {      // (long haplotypes: k_kmer_tables runs 1,024 threads per haplotype)
                b->tables_pending = true; b->tp_n_bases = n_hap_bases; b->tp_table_blocks = table_blocks; b->tp_flag_blocks = flag_blocks; b->tp_rec_blocks = rec_blocks;
            } else { OCT_LAUNCH(k_hap_tables, table_blocks + flag_blocks + rec_blocks, 256, 0, s, d, n_hap_bases, table_blocks, flag_blocks); RT(rt::launch_ok()); }
            b->stats_clear = true;                             // (the kernel's last workgroup clears the counters)
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

