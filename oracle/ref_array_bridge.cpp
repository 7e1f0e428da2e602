// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own batch driver, compiled in place from /root/reference/src:
//   core/models/haplotype_likelihood_array.cpp   HaplotypeLikelihoodArray::populate(ReadMap ...) :51-103, populate(TemplateMap ...) :105-199
//                                                (read hashes once per batch, k-mer table per haplotype, map_query_to_target per pair,
//                                                model reset per haplotype, template sums, optional thread-pool fan-out), the read-back
//                                                accessors operator()(sample, IndexedHaplotype) / operator[] when primed, merge_samples :357-409
//   core/models/haplotype_likelihood_model.cpp, utils/kmer_mapper.hpp, utils/thread_pool.cpp, utils/parallel_transform.hpp
// on stand-in Haplotype / AlignedRead / container types (oracle/ref_shim) and with every haplotype's six penalty vectors supplied through
// two trivial error-model subclasses (the vectors are the C-ABI batch's). This pins the oracle's populate driver loop (oracle_populate).
// Linked into its own library (_ref/libref_array.so): the genotype bridge in libref_phmm.so uses a stand-in class of the same name.
#include <cstdint>
#include <cstdio>
#include <memory>
#include <numeric>
#include <string>
#include <vector>
#include REF_ARRAY_HPP
#include "core/models/error/error_model_factory.hpp"

using namespace octopus;

namespace octopus { namespace config { const std::string HelpForum {}, BugReport {"(bridge)"}; } }   // named by ProgramError::do_help only
namespace octopus {
std::unique_ptr<SnvErrorModel> make_snv_error_model() { return nullptr; }
std::unique_ptr<IndelErrorModel> make_indel_error_model() { return nullptr; }
ErrorModel make_error_model(const std::string&) { return {}; }
}

namespace {
struct Vectors { std::vector<char> mask_f, mask_r; std::vector<std::int8_t> prior_f, prior_r, go, ge; };
const Vectors& vectors_of(const Haplotype& h) { return *static_cast<const Vectors*>(h.payload_); }

class GivenSnvModel : public SnvErrorModel
{
    std::unique_ptr<SnvErrorModel> do_clone() const override { return std::make_unique<GivenSnvModel>(*this); }
    void do_evaluate(const Haplotype& h, MutationVector& fm, PenaltyVector& fp, MutationVector& rm, PenaltyVector& rp) const override
    { const auto& v = vectors_of(h); fm = v.mask_f; fp = v.prior_f; rm = v.mask_r; rp = v.prior_r; }
};
class GivenIndelModel : public IndelErrorModel
{
    std::unique_ptr<IndelErrorModel> do_clone() const override { return std::make_unique<GivenIndelModel>(*this); }
    void do_set_penalties(const Haplotype& h, PenaltyVector& go, PenaltyType& ge) const override { const auto& v = vectors_of(h); go = v.go; ge = v.ge.empty() ? 0 : v.ge.front(); }
    void do_set_penalties(const Haplotype& h, PenaltyVector& go, PenaltyVector& ge) const override { const auto& v = vectors_of(h); go = v.go; ge = v.ge; }
};
} // namespace

struct ref_array_args {
    int32_t max_indel_error, use_int_scores, use_mapping_quality, mapping_quality_cap, mapping_quality_cap_trigger, use_flank_state;
    uint32_t n_haps; const char* hap_bases; const uint32_t* hap_off; const int64_t* hap_begin;
    const int8_t* gap_open; const int8_t* gap_extend; const char* mask_f; const int8_t* prior_f; const char* mask_r; const int8_t* prior_r;
    int32_t has_flank; uint32_t lhs_flank, rhs_flank;
    uint32_t n_reads; const char* read_bases; const uint8_t* quals; const uint32_t* read_off; const int64_t* read_begin; const uint8_t* mapq; const uint8_t* reverse;
    uint32_t n_rows; const uint32_t* row_off;          // NULL: the ReadMap overload (a row per read); else the TemplateMap overload, row r = reads [row_off[r], row_off[r+1])
    uint32_t n_samples; const uint32_t* sample_row_off;  // rows of sample s = [sample_row_off[s], sample_row_off[s+1])
    int32_t n_threads;                                   // > 2: hand populate a ThreadPool of that size
};

// Timing entry for bench.py's cpu_baseline: `reps` populate() calls of the reference on the same inputs (containers built once, outside the
// clock); returns the seconds spent inside populate(), or a negative value on ShortHaplotypeError.
extern "C" double ref_array_time_populate(const ref_array_args* a, int reps);

// out[h * n_rows + row] read back through operator()(sample, IndexedHaplotype); merged (optional, same shape) through merge_samples() + operator[].
// returns 0 ok, 1 ShortHaplotypeError (*err_hap = index of the haplotype it names, *ext = required_extension)
extern "C" int ref_array_populate(const ref_array_args* a, double* out, double* merged, uint32_t* err_hap, uint32_t* ext)
{
    std::vector<Vectors> vecs(a->n_haps);
    MappableBlock<Haplotype> haps(a->n_haps);
    for (uint32_t h = 0; h < a->n_haps; ++h) {
        const uint32_t o = a->hap_off[h], n = a->hap_off[h + 1] - o;
        auto& v = vecs[h];
        v.go.assign(a->gap_open + o, a->gap_open + o + n); v.ge.assign(a->gap_extend + o, a->gap_extend + o + n);
        v.mask_f.assign(a->mask_f + o, a->mask_f + o + n); v.mask_r.assign(a->mask_r + o, a->mask_r + o + n);
        v.prior_f.assign(a->prior_f + o, a->prior_f + o + n); v.prior_r.assign(a->prior_r + o, a->prior_r + o + n);
        haps[h].sequence_.assign(a->hap_bases + o, a->hap_bases + o + n); haps[h].begin_ = a->hap_begin[h]; haps[h].payload_ = &v;
    }
    auto make_read = [&](uint32_t r) {
        AlignedRead x; const uint32_t o = a->read_off[r], n = a->read_off[r + 1] - o;
        x.sequence_.assign(a->read_bases + o, a->read_bases + o + n); x.base_qualities_.assign(a->quals + o, a->quals + o + n);
        x.mapping_quality_ = a->mapq[r]; x.reverse_ = a->reverse[r] != 0; x.begin_ = a->read_begin[r];
        return x;
    };
    std::vector<SampleName> samples;
    for (uint32_t s = 0; s < a->n_samples; ++s) { char nm[16]; std::snprintf(nm, sizeof nm, "s%04u", s); samples.emplace_back(nm); }
    HaplotypeLikelihoodModel::Config cfg;
    cfg.use_mapping_quality = a->use_mapping_quality != 0; cfg.mapping_quality_cap = static_cast<std::uint8_t>(a->mapping_quality_cap);
    if (a->mapping_quality_cap_trigger >= 0) cfg.mapping_quality_cap_trigger = static_cast<std::uint8_t>(a->mapping_quality_cap_trigger);
    cfg.use_flank_state = a->use_flank_state != 0; cfg.max_indel_error = static_cast<unsigned>(a->max_indel_error); cfg.use_int_scores = a->use_int_scores != 0;
    HaplotypeLikelihoodModel model {std::make_unique<GivenSnvModel>(), std::make_unique<GivenIndelModel>(), cfg};
    const bool arr_can_use_flank = model.can_use_flank_state();
    HaplotypeLikelihoodArray arr {std::move(model), a->n_haps, samples};
    boost::optional<HaplotypeLikelihoodArray::FlankState> fs;
    if (a->has_flank && arr_can_use_flank) fs = HaplotypeLikelihoodArray::FlankState {a->lhs_flank, a->rhs_flank};   // the caller's gate: Caller::compute_haplotype_likelihoods, caller.cpp:1172-1173
    std::unique_ptr<ThreadPool> pool;
    HaplotypeLikelihoodArray::OptionalThreadPool workers;
    if (a->n_threads > 2) { pool = std::make_unique<ThreadPool>(static_cast<std::size_t>(a->n_threads)); workers = *pool; }
    try {
        if (!a->row_off) {
            ReadMap reads;
            for (uint32_t s = 0; s < a->n_samples; ++s) {
                auto& dst = reads[samples[s]];
                for (uint32_t r = a->sample_row_off[s]; r < a->sample_row_off[s + 1]; ++r) dst.push_back(make_read(r));
            }
            arr.populate(reads, haps, fs, workers);
        } else {
            TemplateMap reads;
            for (uint32_t s = 0; s < a->n_samples; ++s) {
                auto& dst = reads[samples[s]];
                for (uint32_t row = a->sample_row_off[s]; row < a->sample_row_off[s + 1]; ++row) {
                    AlignedTemplate t;
                    for (uint32_t r = a->row_off[row]; r < a->row_off[row + 1]; ++r) t.push_back(make_read(r));
                    dst.push_back(std::move(t));
                }
            }
            arr.populate(reads, haps, fs, workers);
        }
    } catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) {
        *ext = static_cast<uint32_t>(e.required_extension()); *err_hap = ~0u;
        for (uint32_t h = 0; h < a->n_haps; ++h) if (&e.haplotype() == &haps[h] || e.haplotype() == haps[h]) { *err_hap = h; break; }
        return 1;
    }
    for (uint32_t h = 0; h < a->n_haps; ++h)
        for (uint32_t s = 0; s < a->n_samples; ++s) {
            const auto& v = arr(samples[s], IndexedHaplotype<> {h});
            if (v.size() != a->sample_row_off[s + 1] - a->sample_row_off[s]) return 2;
            for (std::size_t i = 0; i < v.size(); ++i) out[static_cast<std::size_t>(h) * a->n_rows + a->sample_row_off[s] + i] = v[i];
        }
    if (merged) {
        const auto m = arr.merge_samples();
        for (uint32_t h = 0; h < a->n_haps; ++h) {
            const auto& v = m[IndexedHaplotype<> {h}];
            if (v.size() != a->n_rows) return 3;
            for (std::size_t i = 0; i < v.size(); ++i) merged[static_cast<std::size_t>(h) * a->n_rows + i] = v[i];
        }
    }
    return 0;
}

#include <chrono>
extern "C" double ref_array_time_populate(const ref_array_args* a, int reps)
{
    std::vector<Vectors> vecs(a->n_haps);
    MappableBlock<Haplotype> haps(a->n_haps);
    for (uint32_t h = 0; h < a->n_haps; ++h) {
        const uint32_t o = a->hap_off[h], n = a->hap_off[h + 1] - o;
        auto& v = vecs[h];
        v.go.assign(a->gap_open + o, a->gap_open + o + n); v.ge.assign(a->gap_extend + o, a->gap_extend + o + n);
        v.mask_f.assign(a->mask_f + o, a->mask_f + o + n); v.mask_r.assign(a->mask_r + o, a->mask_r + o + n);
        v.prior_f.assign(a->prior_f + o, a->prior_f + o + n); v.prior_r.assign(a->prior_r + o, a->prior_r + o + n);
        haps[h].sequence_.assign(a->hap_bases + o, a->hap_bases + o + n); haps[h].begin_ = a->hap_begin[h]; haps[h].payload_ = &v;
    }
    TemplateMap reads;                                   // one read per template: the overload that fans haplotypes out over the thread pool
    auto& dst = reads["s0000"];
    for (uint32_t r = 0; r < a->n_reads; ++r) {
        AlignedRead x; const uint32_t o = a->read_off[r], n = a->read_off[r + 1] - o;
        x.sequence_.assign(a->read_bases + o, a->read_bases + o + n); x.base_qualities_.assign(a->quals + o, a->quals + o + n);
        x.mapping_quality_ = a->mapq[r]; x.reverse_ = a->reverse[r] != 0; x.begin_ = a->read_begin[r];
        dst.push_back(AlignedTemplate {std::move(x)});
    }
    HaplotypeLikelihoodModel::Config cfg;
    cfg.use_mapping_quality = a->use_mapping_quality != 0; cfg.mapping_quality_cap = static_cast<std::uint8_t>(a->mapping_quality_cap);
    if (a->mapping_quality_cap_trigger >= 0) cfg.mapping_quality_cap_trigger = static_cast<std::uint8_t>(a->mapping_quality_cap_trigger);
    cfg.use_flank_state = a->use_flank_state != 0; cfg.max_indel_error = static_cast<unsigned>(a->max_indel_error); cfg.use_int_scores = a->use_int_scores != 0;
    boost::optional<HaplotypeLikelihoodArray::FlankState> fs;
    if (a->has_flank && cfg.use_flank_state) fs = HaplotypeLikelihoodArray::FlankState {a->lhs_flank, a->rhs_flank};
    std::unique_ptr<ThreadPool> pool;
    HaplotypeLikelihoodArray::OptionalThreadPool workers;
    if (a->n_threads > 2) { pool = std::make_unique<ThreadPool>(static_cast<std::size_t>(a->n_threads)); workers = *pool; }
    double seconds = 0.0;
    for (int rep = 0; rep < reps; ++rep) {
        HaplotypeLikelihoodModel model {std::make_unique<GivenSnvModel>(), std::make_unique<GivenIndelModel>(), cfg};
        HaplotypeLikelihoodArray arr {std::move(model), a->n_haps, {"s0000"}};
        const auto t0 = std::chrono::steady_clock::now();
        try { arr.populate(reads, haps, fs, workers); }
        catch (const HaplotypeLikelihoodModel::ShortHaplotypeError&) { return -1.0; }
        seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return seconds;
}



// Every read-back method of the class after one populate(), for tests/test_integration_patch.py (the same bridge is compiled against the
// unpatched reference class and against the class with INTEGRATION.md's patch applied, oracle/make_patched_tree.py):
//   sec[0] operator()(sample, Haplotype)            sec[1] extract_sample(sample).at(haplotype)
//   sec[2] prime(sample) + operator[](Haplotype)    sec[3] merge_samples(all) + operator[](IndexedHaplotype)
//   sec[4] merge_samples({first, last sample}) + operator[](Haplotype)   (rows of those two samples only, row-major per haplotype)
//   sec[5] after reset(kept haplotypes): operator()(sample, IndexedHaplotype) of the kept rows, [n_keep][n_rows]
// each of sec[0..3] is [n_haps][n_rows]; flags[0] = num_likelihoods() when primed with sample 0, flags[1] = haplotypes().size() after reset,
// flags[2] = contains(dropped haplotype) after reset, flags[3] = contains(kept haplotype), flags[4] = is_empty() after clear().
// returns 0 ok, 1 ShortHaplotypeError (as ref_array_populate), 2.. shape mismatch
extern "C" int ref_array_exercise(const ref_array_args* a, const uint32_t* keep, uint32_t n_keep, double* sec, uint32_t* flags, uint32_t* err_hap, uint32_t* ext)
{
    std::vector<Vectors> vecs(a->n_haps);
    MappableBlock<Haplotype> haps(a->n_haps);
    for (uint32_t h = 0; h < a->n_haps; ++h) {
        const uint32_t o = a->hap_off[h], n = a->hap_off[h + 1] - o;
        auto& v = vecs[h];
        v.go.assign(a->gap_open + o, a->gap_open + o + n); v.ge.assign(a->gap_extend + o, a->gap_extend + o + n);
        v.mask_f.assign(a->mask_f + o, a->mask_f + o + n); v.mask_r.assign(a->mask_r + o, a->mask_r + o + n);
        v.prior_f.assign(a->prior_f + o, a->prior_f + o + n); v.prior_r.assign(a->prior_r + o, a->prior_r + o + n);
        haps[h].sequence_.assign(a->hap_bases + o, a->hap_bases + o + n); haps[h].begin_ = a->hap_begin[h]; haps[h].payload_ = &v;
    }
    auto make_read = [&](uint32_t r) {
        AlignedRead x; const uint32_t o = a->read_off[r], n = a->read_off[r + 1] - o;
        x.sequence_.assign(a->read_bases + o, a->read_bases + o + n); x.base_qualities_.assign(a->quals + o, a->quals + o + n);
        x.mapping_quality_ = a->mapq[r]; x.reverse_ = a->reverse[r] != 0; x.begin_ = a->read_begin[r];
        return x;
    };
    std::vector<SampleName> samples;
    for (uint32_t s = 0; s < a->n_samples; ++s) { char nm[16]; std::snprintf(nm, sizeof nm, "s%04u", s); samples.emplace_back(nm); }
    HaplotypeLikelihoodModel::Config cfg;
    cfg.use_mapping_quality = a->use_mapping_quality != 0; cfg.mapping_quality_cap = static_cast<std::uint8_t>(a->mapping_quality_cap);
    if (a->mapping_quality_cap_trigger >= 0) cfg.mapping_quality_cap_trigger = static_cast<std::uint8_t>(a->mapping_quality_cap_trigger);
    cfg.use_flank_state = a->use_flank_state != 0; cfg.max_indel_error = static_cast<unsigned>(a->max_indel_error); cfg.use_int_scores = a->use_int_scores != 0;
    HaplotypeLikelihoodModel model {std::make_unique<GivenSnvModel>(), std::make_unique<GivenIndelModel>(), cfg};
    const bool can_flank = model.can_use_flank_state();
    HaplotypeLikelihoodArray arr {std::move(model), a->n_haps, samples};
    boost::optional<HaplotypeLikelihoodArray::FlankState> fs;
    if (a->has_flank && can_flank) fs = HaplotypeLikelihoodArray::FlankState {a->lhs_flank, a->rhs_flank};
    std::unique_ptr<ThreadPool> pool;
    HaplotypeLikelihoodArray::OptionalThreadPool workers;
    if (a->n_threads > 2) { pool = std::make_unique<ThreadPool>(static_cast<std::size_t>(a->n_threads)); workers = *pool; }
    try {
        if (!a->row_off) {
            ReadMap reads;
            for (uint32_t s = 0; s < a->n_samples; ++s) { auto& dst = reads[samples[s]]; for (uint32_t r = a->sample_row_off[s]; r < a->sample_row_off[s + 1]; ++r) dst.push_back(make_read(r)); }
            arr.populate(reads, haps, fs, workers);
        } else {
            TemplateMap reads;
            for (uint32_t s = 0; s < a->n_samples; ++s) {
                auto& dst = reads[samples[s]];
                for (uint32_t row = a->sample_row_off[s]; row < a->sample_row_off[s + 1]; ++row) {
                    AlignedTemplate t;
                    for (uint32_t r = a->row_off[row]; r < a->row_off[row + 1]; ++r) t.push_back(make_read(r));
                    dst.push_back(std::move(t));
                }
            }
            arr.populate(reads, haps, fs, workers);
        }
    } catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) {
        *ext = static_cast<uint32_t>(e.required_extension()); *err_hap = ~0u;
        for (uint32_t h = 0; h < a->n_haps; ++h) if (&e.haplotype() == &haps[h] || e.haplotype() == haps[h]) { *err_hap = h; break; }
        return 1;
    }
    const std::size_t H = a->n_haps, NR = a->n_rows, plane = H * NR;
    auto rows_of = [&](uint32_t s) { return a->sample_row_off[s + 1] - a->sample_row_off[s]; };
    for (uint32_t s = 0; s < a->n_samples; ++s) {
        const auto ex = arr.extract_sample(samples[s]);
        arr.prime(samples[s]);
        if (!arr.is_primed() || arr.num_likelihoods() != rows_of(s) || arr.num_likelihoods(samples[s]) != rows_of(s)) return 2;
        for (uint32_t h = 0; h < H; ++h) {
            const auto& v0 = arr(samples[s], haps[h]);
            const auto& v1 = ex.at(haps[h]).get();
            const auto& v2 = arr[haps[h]];
            if (v0.size() != rows_of(s) || v1.size() != rows_of(s) || v2.size() != rows_of(s)) return 3;
            for (std::size_t i = 0; i < v0.size(); ++i) {
                const std::size_t at = h * NR + a->sample_row_off[s] + i;
                sec[at] = v0[i]; sec[plane + at] = v1[i]; sec[2 * plane + at] = v2[i];
            }
        }
        arr.unprime();
        if (arr.is_primed()) return 4;
    }
    arr.prime(samples[0]); flags[0] = static_cast<uint32_t>(arr.num_likelihoods()); arr.unprime();
    {
        const auto m = arr.merge_samples();
        for (uint32_t h = 0; h < H; ++h) {
            const auto& v = m[IndexedHaplotype<> {h}];
            if (v.size() != NR) return 5;
            for (std::size_t i = 0; i < NR; ++i) sec[3 * plane + h * NR + i] = v[i];
        }
        const std::vector<SampleName> two {samples.front(), samples.back()};
        const auto m2 = arr.merge_samples(two, SampleName {"both"});
        const std::size_t n2 = rows_of(0) + rows_of(a->n_samples - 1);
        for (uint32_t h = 0; h < H; ++h) {
            const auto& v = m2[haps[h]];
            if (v.size() != n2) return 6;
            for (std::size_t i = 0; i < n2; ++i) sec[4 * plane + h * NR + i] = v[i];
        }
    }
    MappableBlock<Haplotype> kept;
    for (uint32_t k = 0; k < n_keep; ++k) kept.push_back(haps[keep[k]]);
    arr.reset(kept);
    flags[1] = static_cast<uint32_t>(arr.haplotypes().size());
    uint32_t dropped = ~0u;
    for (uint32_t h = 0; h < H && dropped == ~0u; ++h) { bool in = false; for (uint32_t k = 0; k < n_keep; ++k) in = in || keep[k] == h; if (!in) dropped = h; }
    flags[2] = dropped != ~0u && arr.contains(haps[dropped]) ? 1 : 0;     // the reference leaves dropped haplotypes in its index map (cpp:331-355); whatever it says, the patched class must say the same
    flags[3] = n_keep && arr.contains(haps[keep[0]]) ? 1 : 0;
    for (uint32_t k = 0; k < n_keep; ++k)
        for (uint32_t s = 0; s < a->n_samples; ++s) {
            const auto& v = arr(samples[s], IndexedHaplotype<> {k});
            const auto& w = arr(samples[s], haps[keep[k]]);
            if (v.size() != rows_of(s) || &v != &w) return 7;
            for (std::size_t i = 0; i < v.size(); ++i) sec[5 * plane + k * NR + a->sample_row_off[s] + i] = v[i];
        }
    arr.clear();
    flags[4] = arr.is_empty() ? 1 : 0;
    return 0;
}
