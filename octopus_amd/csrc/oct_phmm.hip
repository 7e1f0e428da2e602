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

#include "host_state.hh"
#include "host_launch.hh"
#include "host_error_model.hh"
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

#include "host_upload.hh"
#include "host_step.hh"
#include "host_readout.hh"
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

#include "host_server.hh"
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
