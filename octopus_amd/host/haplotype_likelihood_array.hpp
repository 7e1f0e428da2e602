// Host-side C++ mirror of the reference's interface for the hot path, on top of the C ABI (include/oct_phmm.h).
//
//   octopus::HaplotypeLikelihoodModel   ref: src/core/models/haplotype_likelihood_model.{hpp,cpp}
//   octopus::HaplotypeLikelihoodArray   ref: src/core/models/haplotype_likelihood_array.{hpp,cpp}
//
// Same names, argument meaning, index order and error behaviour (ShortHaplotypeError, TooLargeBandSizeError), so that the body of
// the reference's populate() can be swapped for this one. AlignedRead / Haplotype are minimal stand-ins for the
// reference's domain types (only the fields the path touches); the per-haplotype penalty vectors normally produced by the reference's
// error models (src/core/models/error/*) are INPUTS of the path and come from a user-supplied PenaltyModel. All arithmetic happens in
// liboct_phmm.so on the GPU; this header only packs and scatters.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <cmath>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/oct_phmm.h"

namespace octopus_amd {

using SampleName = std::string;

struct AlignedRead                                  // ref: src/basics/aligned_read.hpp (fields used by the path)
{
    std::string sequence_;
    std::vector<std::uint8_t> base_qualities_;
    std::uint8_t mapping_quality_ = 60;
    bool reverse_ = false;
    std::int64_t begin_ = 0;                        // mapped_region(read).begin()
    const std::string& sequence() const noexcept { return sequence_; }
    const std::vector<std::uint8_t>& base_qualities() const noexcept { return base_qualities_; }
    std::uint8_t mapping_quality() const noexcept { return mapping_quality_; }
    bool is_marked_reverse_mapped() const noexcept { return reverse_; }
};
using AlignedTemplate = std::vector<AlignedRead>;   // ref: src/basics/aligned_template.hpp

struct Haplotype                                    // ref: src/core/types/haplotype.hpp
{
    std::string sequence_;
    std::int64_t begin_ = 0;                        // mapped_region(haplotype).begin()
    const std::string& sequence() const noexcept { return sequence_; }
    bool operator==(const Haplotype& o) const noexcept { return begin_ == o.begin_ && sequence_ == o.sequence_; }
};
struct HaplotypeHash { std::size_t operator()(const Haplotype& h) const noexcept { return std::hash<std::string> {}(h.sequence_) ^ std::hash<std::int64_t> {}(h.begin_); } };

// iteration order of the reference's MappableMap<SampleName, …> made explicit: (sample, items) in container order
using ReadMap     = std::vector<std::pair<SampleName, std::vector<AlignedRead>>>;       // ref: src/config/common.hpp:27-37
using TemplateMap = std::vector<std::pair<SampleName, std::vector<AlignedTemplate>>>;

// The six per-haplotype vectors HaplotypeLikelihoodModel::reset prepares (model.cpp:60-78)
struct PenaltyVectors
{
    std::vector<std::int8_t> gap_open, gap_extend, snv_forward_priors, snv_reverse_priors;
    std::string snv_forward_mask, snv_reverse_mask;
};
using PenaltyModel = std::function<PenaltyVectors(const Haplotype&)>;

// Stand-in for "no error model" (model.cpp:68-73: mask = haplotype, prior = 100) with flat gap penalties. The reference's
// repeat-aware models (error/*.cpp) stay on the host and plug in here.
inline PenaltyVectors flat_penalties(const Haplotype& h, std::int8_t open = 45, std::int8_t extend = 3)
{
    const std::size_t n = h.sequence().size();
    PenaltyVectors p;
    p.gap_open.assign(n, open); p.gap_extend.assign(n, extend);
    p.snv_forward_priors.assign(n, 100); p.snv_reverse_priors.assign(n, 100);
    p.snv_forward_mask = h.sequence(); p.snv_reverse_mask = h.sequence();
    return p;
}

class HaplotypeLikelihoodModel
{
public:
    using LogProbability = double;
    struct Config                                   // ref: haplotype_likelihood_model.hpp:36-44
    {
        bool use_mapping_quality = true;
        int  mapping_quality_cap_trigger = -1;      // boost::none
        std::uint8_t mapping_quality_cap = 120;
        bool use_flank_state = true;
        unsigned max_indel_error = 8;
        bool use_int_scores = false;
        int  device_id = 0;                         // new: which GPU
    };
    struct FlankState { std::uint32_t lhs_flank, rhs_flank; };     // ref: :46-49

    class ShortHaplotypeError : public std::runtime_error        // ref: :123-139, model.cpp:17-33
    {
    public:
        ShortHaplotypeError(std::size_t haplotype_index, std::size_t required_extension)
        : std::runtime_error {"Haplotype is too short for alignment"}, haplotype_index_ {haplotype_index}, required_extension_ {required_extension} {}
        std::size_t haplotype_index() const noexcept { return haplotype_index_; }
        std::size_t required_extension() const noexcept { return required_extension_; }
    private:
        std::size_t haplotype_index_, required_extension_;
    };
    class TooLargeBandSizeError : public std::runtime_error      // ref: simd_pair_hmm_wrapper.hpp:45-61
    {
    public:
        explicit TooLargeBandSizeError(int requested) : std::runtime_error {"requested band size is too large"}, requested_ {requested} {}
        int requested() const noexcept { return requested_; }
        int max() const noexcept { return 256; }
    private:
        int requested_;
    };
    class DeviceError : public std::runtime_error { public: using std::runtime_error::runtime_error; };

    HaplotypeLikelihoodModel() : HaplotypeLikelihoodModel {Config {}} {}
    explicit HaplotypeLikelihoodModel(Config config, PenaltyModel penalties = {})
    : config_ {config}, penalties_ {penalties ? std::move(penalties) : PenaltyModel {[] (const Haplotype& h) { return flat_penalties(h); }}}
    {
        band_ = 0;
        for (int b = 8; b <= 256 && !band_; b *= 2) if (static_cast<int>(config.max_indel_error) <= b) band_ = b;     // simd_pair_hmm_wrapper.hpp:219-241
        if (!band_) throw TooLargeBandSizeError {static_cast<int>(config.max_indel_error)};
    }
    // One oct_phmm_handle per model OBJECT: the handle owns streams, staging buffers and scratch and must not be shared between threads
    // (include/oct_phmm.h), while the reference copies its model into every worker task (haplotype_likelihood_array.cpp:172) and into every
    // array. A copy therefore starts without a handle and creates its own on first use.
    HaplotypeLikelihoodModel(const HaplotypeLikelihoodModel& other) : config_ {other.config_}, penalties_ {other.penalties_}, band_ {other.band_} {}
    HaplotypeLikelihoodModel& operator=(const HaplotypeLikelihoodModel& other)
    {
        if (this != &other) { config_ = other.config_; penalties_ = other.penalties_; band_ = other.band_; handle_.reset(); }
        return *this;
    }
    HaplotypeLikelihoodModel(HaplotypeLikelihoodModel&&) = default;
    HaplotypeLikelihoodModel& operator=(HaplotypeLikelihoodModel&&) = default;
    struct Alignment { std::size_t mapping_position; std::string cigar; double likelihood; };   // ref: haplotype_likelihood_model.hpp:57-62 (CigarString as text)
    class HMMOverflow : public std::runtime_error { public: HMMOverflow() : std::runtime_error {"Pair HMM alignment overflowed"} {} };   // ref: pair_hmm.hpp:47-64

    // ref: HaplotypeLikelihoodModel::align(read) after reset(haplotype, flank_state), model.cpp:322-431 - here for a whole vector of reads
    // against one haplotype, as read_realigner.cpp:114-139 needs it; candidate positions come from the device k-mer mapper.
    std::vector<Alignment> align(const std::vector<AlignedRead>& reads, const Haplotype& haplotype, const FlankState* flank_state = nullptr) const
    {
        std::string rb; std::vector<std::uint8_t> q, mq, rev; std::vector<std::uint32_t> off {0}; std::vector<std::int64_t> beg;
        for (const AlignedRead& r : reads) {
            rb += r.sequence(); q.insert(q.end(), r.base_qualities().begin(), r.base_qualities().end()); off.push_back(static_cast<std::uint32_t>(rb.size()));
            mq.push_back(r.mapping_quality()); rev.push_back(r.is_marked_reverse_mapped() ? 1 : 0); beg.push_back(r.begin_);
        }
        const PenaltyVectors v = penalties_(haplotype);
        const std::uint32_t hoff[2] = {0, static_cast<std::uint32_t>(haplotype.sequence().size())}; const std::int64_t hbeg[1] = {haplotype.begin_};
        oct_phmm_reads R {static_cast<std::uint32_t>(reads.size()), rb.data(), q.data(), off.data(), mq.data(), rev.data(), beg.data(), 0, nullptr};
        oct_phmm_haplotypes H {1, haplotype.sequence().data(), hoff, hbeg, v.gap_open.data(), v.gap_extend.data(), v.snv_forward_mask.data(),
                               v.snv_forward_priors.data(), v.snv_reverse_mask.data(), v.snv_reverse_priors.data()};
        oct_phmm_flank_state fs {0, 0}; if (flank_state) fs = {flank_state->lhs_flank, flank_state->rhs_flank};
        std::uint32_t cap = 32;
        for (;;) {
            std::vector<std::uint32_t> mpos(reads.size() + 1), n_ops(reads.size() + 1), ops(reads.size() * cap + 1); std::vector<double> lik(reads.size() + 1);
            oct_phmm_alignments out {cap, mpos.data(), lik.data(), n_ops.data(), ops.data()};
            oct_phmm_status st;
            const int rc = oct_phmm_align(handle(), &R, &H, nullptr, flank_state ? &fs : nullptr, nullptr, &out, &st);
            if (rc == OCT_PHMM_EINVAL && st.required_extension > cap) { cap = st.required_extension; continue; }   // CIGARs longer than guessed: once more with room
            if (rc == OCT_PHMM_ESHORT_HAPLOTYPE) throw ShortHaplotypeError {st.hap_index, st.required_extension};
            if (rc == OCT_PHMM_EOVERFLOW) throw HMMOverflow {};
            if (rc != OCT_PHMM_OK) throw DeviceError {std::string {"oct_phmm_align: "} + oct_phmm_strerror(rc) + " (" + st.message + ")"};
            std::vector<Alignment> result(reads.size());
            for (std::size_t i = 0; i < reads.size(); ++i) {
                result[i].mapping_position = mpos[i]; result[i].likelihood = lik[i];
                for (std::uint32_t k = 0; k < n_ops[i]; ++k) {
                    const std::uint32_t op = ops[i * cap + k];
                    result[i].cigar += std::to_string(op >> 4);
                    result[i].cigar += (op & 15u) == OCT_PHMM_CIGAR_EQ ? '=' : (op & 15u) == OCT_PHMM_CIGAR_X ? 'X' : (op & 15u) == OCT_PHMM_CIGAR_INS ? 'I' : 'D';
                }
            }
            return result;
        }
    }

    const Config& config() const noexcept { return config_; }
    unsigned pad_requirement() const noexcept { return static_cast<unsigned>(band_); }   // ref: model.cpp:55-58 (== oct_phmm_band_size)
    bool can_use_flank_state() const noexcept { return config_.use_flank_state; }
    PenaltyVectors penalties(const Haplotype& h) const { return penalties_(h); }
    // this object's own device handle, created on first use; shared_ptr so that a device-resident matrix can keep it alive
    const std::shared_ptr<oct_phmm_handle>& shared_handle() const
    {
        if (!handle_) {
            oct_phmm_config c; oct_phmm_config_default(&c);
            c.max_indel_error = static_cast<int>(config_.max_indel_error); c.use_int_scores = config_.use_int_scores;
            c.use_mapping_quality = config_.use_mapping_quality; c.mapping_quality_cap = config_.mapping_quality_cap;
            c.mapping_quality_cap_trigger = config_.mapping_quality_cap_trigger; c.use_flank_state = config_.use_flank_state;
            c.device_id = config_.device_id;
            oct_phmm_handle* h = nullptr;
            const int rc = oct_phmm_create(&c, &h);
            if (rc == OCT_PHMM_EBAND) throw TooLargeBandSizeError {static_cast<int>(config_.max_indel_error)};
            if (rc != OCT_PHMM_OK) throw DeviceError {std::string {"oct_phmm_create: "} + oct_phmm_strerror(rc)};
            handle_.reset(h, [] (oct_phmm_handle* p) { oct_phmm_destroy(p); });
        }
        return handle_;
    }
    oct_phmm_handle* handle() const { return shared_handle().get(); }
private:
    Config config_;
    PenaltyModel penalties_;
    int band_ = 0;
    mutable std::shared_ptr<oct_phmm_handle> handle_;
};

class HaplotypeLikelihoodArray
{
public:
    using FlankState = HaplotypeLikelihoodModel::FlankState;
    using LogProbability = double;
    using LikelihoodVector = std::vector<LogProbability>;

    using SampleLikelihoodMap = std::unordered_map<Haplotype, LikelihoodVector, HaplotypeHash>;   // ref: hpp:46 (by value here)

    HaplotypeLikelihoodArray() = default;
    HaplotypeLikelihoodArray(HaplotypeLikelihoodModel model, std::vector<SampleName> samples)
    : likelihood_model_ {std::move(model)}, samples_ {std::move(samples)} {}
    // A copy gets the host-side matrix and its own model (own device handle); the device-resident matrix stays with the original, whose
    // handle it lives on - two arrays on two threads never touch one handle.
    HaplotypeLikelihoodArray(const HaplotypeLikelihoodArray& o)
    : likelihood_model_ {o.likelihood_model_}, likelihoods_ {o.likelihoods_}, haplotype_indices_ {o.haplotype_indices_}, sample_indices_ {o.sample_indices_},
      samples_ {o.samples_}, haplotypes_ {o.haplotypes_}, primed_sample_ {o.primed_sample_}, primed_ {o.primed_}, sample_row_begin_ {o.sample_row_begin_} {}
    HaplotypeLikelihoodArray& operator=(const HaplotypeLikelihoodArray& o) { if (this != &o) { HaplotypeLikelihoodArray tmp {o}; *this = std::move(tmp); } return *this; }
    HaplotypeLikelihoodArray(HaplotypeLikelihoodArray&&) = default;
    HaplotypeLikelihoodArray& operator=(HaplotypeLikelihoodArray&&) = default;

    // ref: haplotype_likelihood_array.cpp:51-103
    void populate(const ReadMap& reads, const std::vector<Haplotype>& haplotypes, const FlankState* flank_state = nullptr)
    {
        Packed p;
        std::vector<std::size_t> rows_per_sample;
        sample_indices_.clear();
        for (const auto& s : reads) {
            sample_indices_.emplace(s.first, rows_per_sample.size());
            for (const AlignedRead& r : s.second) p.add(r);
            rows_per_sample.push_back(s.second.size());
        }
        run(p, nullptr, rows_per_sample, haplotypes, flank_state);
    }
    // ref: haplotype_likelihood_array.cpp:105-199
    void populate(const TemplateMap& reads, const std::vector<Haplotype>& haplotypes, const FlankState* flank_state = nullptr)
    {
        Packed p;
        std::vector<std::uint32_t> row_off {0};
        std::vector<std::size_t> rows_per_sample;
        sample_indices_.clear();
        for (const auto& s : reads) {
            sample_indices_.emplace(s.first, rows_per_sample.size());
            for (const AlignedTemplate& t : s.second) { for (const AlignedRead& r : t) p.add(r); row_off.push_back(static_cast<std::uint32_t>(p.mapq.size())); }
            rows_per_sample.push_back(s.second.size());
        }
        run(p, &row_off, rows_per_sample, haplotypes, flank_state);
    }

    std::size_t num_likelihoods(const SampleName& sample) const { return likelihoods_.front()[sample_indices_.at(sample)].size(); }
    std::size_t num_likelihoods() const { return likelihoods_.front()[primed_sample_].size(); }
    const LikelihoodVector& operator()(const SampleName& sample, const Haplotype& haplotype) const { return likelihoods_[haplotype_indices_.at(haplotype)][sample_indices_.at(sample)]; }
    const LikelihoodVector& operator()(const SampleName& sample, std::size_t haplotype_index) const { return likelihoods_[haplotype_index][sample_indices_.at(sample)]; }   // IndexedHaplotype<>
    const LikelihoodVector& operator[](const Haplotype& haplotype) const { return likelihoods_[haplotype_indices_.at(haplotype)][primed_sample_]; }
    const LikelihoodVector& operator[](std::size_t haplotype_index) const noexcept { return likelihoods_[haplotype_index][primed_sample_]; }
    std::vector<SampleName> samples() const { return samples_; }
    const std::vector<Haplotype>& haplotypes() const noexcept { return haplotypes_; }
    SampleLikelihoodMap extract_sample(const SampleName& sample) const                        // ref: cpp:248-257
    {
        const auto sample_index = sample_indices_.at(sample);
        SampleLikelihoodMap result {haplotype_indices_.size()};
        for (const auto& p : haplotype_indices_) result.emplace(p.first, likelihoods_[p.second][sample_index]);
        return result;
    }
    bool contains(const Haplotype& haplotype) const noexcept { return haplotype_indices_.count(haplotype) == 1; }
    bool is_empty() const noexcept { return likelihoods_.empty(); }
    void clear() noexcept { likelihoods_.clear(); haplotype_indices_.clear(); sample_indices_.clear(); haplotypes_.clear(); resident_.reset(); resident_handle_.reset(); unprime(); }
    // the device-resident matrix of the last populate (null after reset()/merge_samples(), whose row layout it no longer matches)
    oct_phmm_batch* resident_batch() const noexcept { return resident_.get(); }
    oct_phmm_handle* handle() const { return resident_handle_ ? resident_handle_.get() : likelihood_model_.handle(); }   // the handle the resident matrix lives on
    std::pair<std::uint32_t, std::uint32_t> primed_rows() const { return {sample_row_begin_.at(primed_sample_), sample_row_begin_.at(primed_sample_ + 1)}; }
    bool is_primed() const noexcept { return primed_; }
    void prime(const SampleName& sample) const { primed_sample_ = sample_indices_.at(sample); primed_ = true; }
    void unprime() const noexcept { primed_ = false; }

    // ref: :331-355 — keep a (sorted) subset of the haplotypes, erasing the other rows in place
    void reset(const std::vector<Haplotype>& keep)
    {
        if (keep.empty()) { clear(); return; }
        if (keep.size() >= haplotypes_.size()) return;
        std::vector<std::vector<LikelihoodVector>> kept; kept.reserve(keep.size());
        std::unordered_map<Haplotype, std::size_t, HaplotypeHash> idx;
        for (std::size_t i = 0; i < keep.size(); ++i) { kept.push_back(std::move(likelihoods_[haplotype_indices_.at(keep[i])])); idx.emplace(keep[i], i); }
        likelihoods_ = std::move(kept); haplotype_indices_ = std::move(idx); haplotypes_ = keep; resident_.reset();
    }
    // ref: :357-409 — concatenate the samples' likelihood vectors under one new sample
    HaplotypeLikelihoodArray merge_samples(const std::vector<SampleName>& which, SampleName new_sample = {}) const
    {
        if (new_sample.empty()) for (const auto& s : which) new_sample += s;
        HaplotypeLikelihoodArray result {likelihood_model_, {new_sample}};
        result.haplotypes_ = haplotypes_; result.haplotype_indices_ = haplotype_indices_;
        result.likelihoods_.resize(haplotypes_.size(), std::vector<LikelihoodVector>(1));
        for (std::size_t h = 0; h < haplotypes_.size(); ++h)
            for (const auto& s : which) { const auto& src = likelihoods_[h][sample_indices_.at(s)]; auto& dst = result.likelihoods_[h][0]; dst.insert(dst.end(), src.begin(), src.end()); }
        result.sample_indices_.emplace(new_sample, 0);
        result.prime(new_sample);
        return result;
    }

private:
    struct Packed
    {
        std::string bases; std::vector<std::uint8_t> quals, mapq, reverse; std::vector<std::uint32_t> off {0}; std::vector<std::int64_t> begin;
        void add(const AlignedRead& r)
        {
            bases += r.sequence(); quals.insert(quals.end(), r.base_qualities().begin(), r.base_qualities().end());
            off.push_back(static_cast<std::uint32_t>(bases.size())); mapq.push_back(r.mapping_quality());
            reverse.push_back(r.is_marked_reverse_mapped() ? 1 : 0); begin.push_back(r.begin_);
        }
    };

    void run(const Packed& p, const std::vector<std::uint32_t>* row_off, const std::vector<std::size_t>& rows_per_sample,
             const std::vector<Haplotype>& haplotypes, const FlankState* flank_state)
    {
        std::string hb, mf, mr; std::vector<std::uint32_t> hoff {0}; std::vector<std::int64_t> hbegin;
        std::vector<std::int8_t> go, ge, pf, pr;
        for (const Haplotype& h : haplotypes) {              // HaplotypeLikelihoodModel::reset per haplotype (model.cpp:60-78)
            const PenaltyVectors v = likelihood_model_.penalties(h);
            hb += h.sequence(); hoff.push_back(static_cast<std::uint32_t>(hb.size())); hbegin.push_back(h.begin_);
            go.insert(go.end(), v.gap_open.begin(), v.gap_open.end()); ge.insert(ge.end(), v.gap_extend.begin(), v.gap_extend.end());
            mf += v.snv_forward_mask; mr += v.snv_reverse_mask;
            pf.insert(pf.end(), v.snv_forward_priors.begin(), v.snv_forward_priors.end()); pr.insert(pr.end(), v.snv_reverse_priors.begin(), v.snv_reverse_priors.end());
        }
        const std::uint32_t n_rows = row_off ? static_cast<std::uint32_t>(row_off->size() - 1) : static_cast<std::uint32_t>(p.mapq.size());
        oct_phmm_reads R {static_cast<std::uint32_t>(p.mapq.size()), p.bases.data(), p.quals.data(), p.off.data(), p.mapq.data(), p.reverse.data(),
                          p.begin.data(), n_rows, row_off ? row_off->data() : nullptr};
        oct_phmm_haplotypes H {static_cast<std::uint32_t>(haplotypes.size()), hb.data(), hoff.data(), hbegin.data(), go.data(), ge.data(),
                               mf.data(), pf.data(), mr.data(), pr.data()};
        oct_phmm_flank_state fs {0, 0};
        if (flank_state) fs = {flank_state->lhs_flank, flank_state->rhs_flank};
        std::vector<double> out(haplotypes.size() * static_cast<std::size_t>(n_rows) + 1);
        oct_phmm_status st;
        // upload + run + download instead of the one-shot oct_phmm_populate: the matrix stays in HBM behind the array so that the genotype
        // models can read it out there (ConstantMixtureGenotypeLikelihoodModel below)
        resident_.reset(); sample_row_begin_.assign(1, 0);
        for (std::size_t n : rows_per_sample) sample_row_begin_.push_back(sample_row_begin_.back() + static_cast<std::uint32_t>(n));
        const std::shared_ptr<oct_phmm_handle> keep = likelihood_model_.shared_handle();   // the batch holds its handle alive
        oct_phmm_handle* hd = keep.get();
        oct_phmm_batch* batch = nullptr;
        int rc = oct_phmm_batch_upload(hd, &R, &H, nullptr, flank_state ? &fs : nullptr, nullptr, &batch, &st);
        if (batch) { resident_.reset(batch, [keep] (oct_phmm_batch* b) { oct_phmm_batch_free(keep.get(), b); }); resident_handle_ = keep; }
        if (rc == OCT_PHMM_OK) rc = oct_phmm_batch_run(hd, batch, &st);
        if (rc == OCT_PHMM_OK) rc = oct_phmm_batch_download(hd, batch, out.data(), &st);
        if (rc != OCT_PHMM_OK) resident_.reset();
        if (rc == OCT_PHMM_ESHORT_HAPLOTYPE) throw HaplotypeLikelihoodModel::ShortHaplotypeError {st.hap_index, st.required_extension};
        if (rc != OCT_PHMM_OK) throw HaplotypeLikelihoodModel::DeviceError {std::string {"oct_phmm_populate: "} + oct_phmm_strerror(rc) + " (" + st.message + ")"};
        likelihoods_.assign(haplotypes.size(), std::vector<LikelihoodVector>(rows_per_sample.size()));
        const double* q = out.data();
        for (std::size_t h = 0; h < haplotypes.size(); ++h)
            for (std::size_t s = 0; s < rows_per_sample.size(); ++s) { likelihoods_[h][s].assign(q, q + rows_per_sample[s]); q += rows_per_sample[s]; }
        haplotype_indices_.clear();
        for (std::size_t h = 0; h < haplotypes.size(); ++h) haplotype_indices_.emplace(haplotypes[h], h);
        haplotypes_ = haplotypes;
    }

    HaplotypeLikelihoodModel likelihood_model_;
    std::vector<std::vector<LikelihoodVector>> likelihoods_;     // [haplotype][sample][row], ref: hpp:123
    std::unordered_map<Haplotype, std::size_t, HaplotypeHash> haplotype_indices_;
    std::unordered_map<SampleName, std::size_t> sample_indices_;
    std::vector<SampleName> samples_;
    std::vector<Haplotype> haplotypes_;
    mutable std::size_t primed_sample_ = 0;
    mutable bool primed_ = false;
    std::shared_ptr<oct_phmm_batch> resident_;
    std::shared_ptr<oct_phmm_handle> resident_handle_;
    std::vector<std::uint32_t> sample_row_begin_ {0};
};

// ref: core/models/genotype/constant_mixture_genotype_likelihood_model.hpp:16-76. A genotype is its haplotypes' indices in the array
// (IndexedHaplotype<>), sorted. evaluate(genotypes) = the free evaluate(genotypes, model) helper (:58-75): with the matrix still on the
// device it is one oct_phmm_batch_genotype_likelihoods call.
class ConstantMixtureGenotypeLikelihoodModel
{
public:
    using LogProbability = double;
    using Genotype = std::vector<std::size_t>;
    explicit ConstantMixtureGenotypeLikelihoodModel(const HaplotypeLikelihoodArray& likelihoods) : likelihoods_ {likelihoods} {}
    const HaplotypeLikelihoodArray& cache() const noexcept { return likelihoods_; }

    std::vector<LogProbability> evaluate(const std::vector<Genotype>& genotypes) const
    {
        std::vector<LogProbability> result(genotypes.size(), 0.0);
        if (genotypes.empty()) return result;
        const std::size_t ploidy = genotypes.front().size();
        if (ploidy == 0) return result;                                     // :36
        if (oct_phmm_batch* batch = likelihoods_.resident_batch()) {
            std::vector<std::uint32_t> idx; idx.reserve(genotypes.size() * ploidy);
            for (const Genotype& g : genotypes) { if (g.size() != ploidy) throw std::invalid_argument {"mixed ploidies in one evaluate call"}; for (std::size_t h : g) idx.push_back(static_cast<std::uint32_t>(h)); }
            const std::uint32_t pl = static_cast<std::uint32_t>(ploidy), offs[2] = {0, static_cast<std::uint32_t>(genotypes.size())};
            const auto rows = likelihoods_.primed_rows();
            oct_phmm_genotype_sets sets {1, &pl, offs, idx.data(), &rows.first, &rows.second};
            oct_phmm_status st;
            const int rc = oct_phmm_batch_genotype_likelihoods(likelihoods_.handle(), batch, &sets, result.data(), &st);
            if (rc != OCT_PHMM_OK) throw HaplotypeLikelihoodModel::DeviceError {std::string {"oct_phmm_batch_genotype_likelihoods: "} + oct_phmm_strerror(rc) + " (" + st.message + ")"};
            return result;
        }
        // reset()/merge_samples() changed the row layout on the host only: there is no device matrix to read out (and no host
        // re-implementation here) - the reference's own genotype models keep working on the array's accessors
        throw HaplotypeLikelihoodModel::DeviceError {"ConstantMixtureGenotypeLikelihoodModel: the array has no device-resident matrix (populate it again)"};
    }
    LogProbability evaluate(const Genotype& genotype) const { return evaluate(std::vector<Genotype> {genotype}).front(); }

private:
    const HaplotypeLikelihoodArray& likelihoods_;
};

} // namespace octopus_amd
