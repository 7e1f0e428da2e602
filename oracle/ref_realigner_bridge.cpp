// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the third seam of INTEGRATION.md: the REFERENCE's own realignment of the reads assigned to a haplotype, the functions of
// src/core/tools/read_realigner.cpp:83-155 (compute_read_hashes, the two realign(read, haplotype, ...) helpers and
// realign(reads, haplotype, model, log_likelihoods, workers): k-mer table, model.reset, model.align read by read, AlignedRead::realign), cut out of a copy of
// that file by oracle/make_patched_tree.py and compiled HERE between stand-in types - once as they are (SEAM_INC = read_realigner_seam_ref.inc ->
// _ref/libref_realigner.so), once with the last function replaced by integration/read_realigner_on_device.inc (-> _ref/libref_realigner_patched_*.so, linked
// against the product's C ABI). Around them, compiled in place from /root/reference/src: core/models/haplotype_likelihood_model.cpp (reset, align), the
// repeat-based indel / SNV error models with the tandem library, utils/kmer_mapper.hpp, utils/parallel_transform.hpp, utils/thread_pool.cpp, basics/cigar_string.cpp.
// Stand-ins (oracle/ref_shim + below): Haplotype, AlignedRead (its realign() records region and CIGAR), GenomicRegion.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include REF_MODEL_HPP
#include "core/models/error/error_model_factory.hpp"
#include "core/models/error/basic_repeat_based_indel_error_model.hpp"
#include "core/models/error/repeat_based_snv_error_model.hpp"
#include "core/models/pairhmm/pair_hmm.hpp"
#include "utils/kmer_mapper.hpp"
#include "utils/parallel_transform.hpp"
#if defined(REALIGNER_PATCHED)
#include "oct_phmm.h"
#endif

namespace octopus { namespace config { const std::string HelpForum {}, BugReport {"(bridge)"}; } }   // named by ProgramError::do_help only
namespace octopus {
std::unique_ptr<SnvErrorModel> make_snv_error_model() { return nullptr; }
std::unique_ptr<IndelErrorModel> make_indel_error_model() { return nullptr; }
ErrorModel make_error_model(const std::string&) { return {}; }

// ---- stand-ins for what the seam touches outside the model ----
struct GenomicRegion { using Position = std::int64_t; using Size = std::uint32_t; std::string contig; Position begin, end; };
inline std::string contig_name(const AlignedRead&) { return {}; }
namespace {
#include SEAM_INC
} // namespace
} // namespace octopus

using namespace octopus;

struct ref_realigner_args {
    int32_t max_indel_error, use_int_scores, use_mapping_quality, mapping_quality_cap, mapping_quality_cap_trigger;
    const int8_t* tables; const uint32_t* table_lens;      // the error models' ten tables, as ref_error_models takes them
    const char* hap_bases; uint32_t hap_len; int64_t hap_begin;
    uint32_t n_reads; const char* read_bases; const uint8_t* quals; const uint32_t* read_off; const int64_t* read_begin; const uint8_t* mapq; const uint8_t* reverse;
    int32_t want_likelihoods;                              // the overload with the log-likelihood vector (:159-167) or the one without (:169-174)
};

// Per read: begin / end of its new region, its CIGAR (n_ops[i] operations, length << 8 | flag character, at ops + i * max_ops), its log-likelihood.
// Returns 0 ok, 1 ShortHaplotypeError (*ext = required_extension), 3 hmm::HMMOverflow, 4 a CIGAR longer than max_ops, 2 any other exception.
extern "C" int ref_realigner_realign(const ref_realigner_args* a, int64_t* out_begin, int64_t* out_end, uint32_t* n_ops, uint32_t* ops, uint32_t max_ops,
                                     double* loglik, uint32_t* ext)
{
    std::vector<std::vector<std::int8_t>> t(10);
    for (int i = 0, o = 0; i < 10; o += a->table_lens[i], ++i) t[i].assign(a->tables + o, a->tables + o + a->table_lens[i]);
    BasicRepeatBasedIndelErrorModel::Parameters ip {t[0], t[1], t[2], t[3], t[4], t[5], t[6]};
    BasicRepeatBasedSNVErrorModel::Parameters sp {t[7], t[8], t[9]};
    HaplotypeLikelihoodModel::Config cfg;
    cfg.use_mapping_quality = a->use_mapping_quality != 0; cfg.mapping_quality_cap = static_cast<std::uint8_t>(a->mapping_quality_cap);
    if (a->mapping_quality_cap_trigger >= 0) cfg.mapping_quality_cap_trigger = static_cast<std::uint8_t>(a->mapping_quality_cap_trigger);
    cfg.use_flank_state = false; cfg.max_indel_error = static_cast<unsigned>(a->max_indel_error); cfg.use_int_scores = a->use_int_scores != 0;
    HaplotypeLikelihoodModel model {std::make_unique<BasicRepeatBasedSNVErrorModel>(sp), std::make_unique<BasicRepeatBasedIndelErrorModel>(ip), cfg};
    Haplotype haplotype;
    haplotype.sequence_.assign(a->hap_bases, a->hap_bases + a->hap_len); haplotype.begin_ = a->hap_begin;
    haplotype.cigar_.emplace_back(static_cast<CigarOperation::Size>(a->hap_len), CigarOperation::Flag::sequenceMatch);
    std::vector<AlignedRead> reads(a->n_reads);
    for (uint32_t r = 0; r < a->n_reads; ++r) {
        AlignedRead& x = reads[r]; const uint32_t o = a->read_off[r], n = a->read_off[r + 1] - o;
        x.sequence_.assign(a->read_bases + o, a->read_bases + o + n); x.base_qualities_.assign(a->quals + o, a->quals + o + n);
        x.mapping_quality_ = a->mapq[r]; x.reverse_ = a->reverse[r] != 0; x.begin_ = a->read_begin[r];
    }
    try {
        std::vector<HaplotypeLikelihoodModel::LogProbability> ll(a->n_reads, 0.0);
        boost::optional<std::vector<HaplotypeLikelihoodModel::LogProbability>&> ll_ref;
        if (a->want_likelihoods) ll_ref = ll;
        realign(reads, haplotype, model, ll_ref, boost::none);        // the seam's function (the reference's serial branch: its thread-pool branch waits on futures it never set, :128-147)
        for (uint32_t r = 0; r < a->n_reads; ++r) {
            out_begin[r] = reads[r].begin_; out_end[r] = reads[r].realigned_end_;
            if (reads[r].realigned_cigar_.size() > max_ops) return 4;
            n_ops[r] = static_cast<uint32_t>(reads[r].realigned_cigar_.size());
            std::copy(reads[r].realigned_cigar_.begin(), reads[r].realigned_cigar_.end(), ops + static_cast<std::size_t>(r) * max_ops);
            loglik[r] = ll[r];
        }
    } catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) {
        *ext = static_cast<uint32_t>(e.required_extension());
        return 1;
    } catch (const hmm::HMMOverflow&) {
        return 3;
    } catch (const std::exception&) {
        return 2;
    }
    return 0;
}
