// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the second seam of INTEGRATION.md: the REFERENCE's own functions of src/core/tools/read_assigner.cpp:145-287
// (estimate_max_indel_size, compute_read_hashes, expand_for_alignment, calculate_likelihoods: ploidy haplotypes x reads), cut out of a copy of that file by
// oracle/make_patched_tree.py and compiled HERE between stand-in types — once as they are (SEAM_INC = read_assigner_seam_ref.inc ->
// _ref/libref_assigner.so), once with the last function replaced by integration/read_assigner_on_device.inc (-> _ref/libref_assigner_patched_*.so,
// linked against the product's C ABI). Around them, compiled in place from /root/reference/src: core/models/haplotype_likelihood_model.cpp (reset,
// evaluate, pad_requirement), the repeat-based indel / SNV error models with the tandem library, utils/kmer_mapper.hpp, utils/parallel_transform.hpp,
// utils/thread_pool.cpp. Stand-ins (oracle/ref_shim + below): Haplotype, AlignedRead, Genotype, and the handful of region functions the seam calls; a
// haplotype's flanking reference bases (what expand() reads from the reference genome) travel in its payload.
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
#include "core/types/genotype.hpp"
#include "utils/kmer_mapper.hpp"
#include "utils/parallel_transform.hpp"
#if defined(ASSIGNER_PATCHED)
#include "oct_phmm.h"
#endif

namespace octopus { namespace config { const std::string HelpForum {}, BugReport {"(bridge)"}; } }   // named by ProgramError::do_help only
namespace octopus {
std::unique_ptr<SnvErrorModel> make_snv_error_model() { return nullptr; }
std::unique_ptr<IndelErrorModel> make_indel_error_model() { return nullptr; }
ErrorModel make_error_model(const std::string&) { return {}; }

// ---- stand-ins for what the seam touches outside the model: regions, and expand() ----
struct GenomicRegion { using Size = std::uint32_t; using Distance = std::int64_t; std::int64_t begin, end; };
struct HaplotypeContext { std::string left, right; GenomicRegion::Size region_size; };   // reference bases either side of the haplotype; its span on the reference
inline GenomicRegion::Size region_size(const Haplotype& h) { return h.payload_ ? static_cast<const HaplotypeContext*>(h.payload_)->region_size : static_cast<GenomicRegion::Size>(h.sequence_.size()); }
inline GenomicRegion::Size region_size(const AlignedRead& r) { return static_cast<GenomicRegion::Size>(r.sequence_.size()); }
inline GenomicRegion mapped_region(const Haplotype& h) { return {h.begin_, h.begin_ + region_size(h)}; }
inline GenomicRegion region_of(const AlignedRead& r) { return {r.begin_, r.begin_ + static_cast<std::int64_t>(region_size(r))}; }
inline bool begins_before(const GenomicRegion& a, const GenomicRegion& b) { return a.begin < b.begin; }
inline bool ends_before(const GenomicRegion& a, const GenomicRegion& b) { return a.end < b.end; }
inline GenomicRegion::Distance begin_distance(const GenomicRegion& a, const GenomicRegion& b) { return b.begin - a.begin; }
inline GenomicRegion::Distance end_distance(const GenomicRegion& a, const GenomicRegion& b) { return b.end - a.end; }
inline GenomicRegion encompassing_region(const std::vector<AlignedRead>& reads)
{
    GenomicRegion r {INT64_MAX, INT64_MIN};
    for (const auto& x : reads) { const auto q = region_of(x); r.begin = std::min(r.begin, q.begin); r.end = std::max(r.end, q.end); }
    return r;
}
inline GenomicRegion encompassing_region(const std::vector<AlignedTemplate>& templates)
{
    GenomicRegion r {INT64_MAX, INT64_MIN};
    for (const auto& t : templates) for (const auto& x : t) { const auto q = region_of(x); r.begin = std::min(r.begin, q.begin); r.end = std::max(r.end, q.end); }
    return r;
}
inline Haplotype expand(const Haplotype& h, const std::size_t n)                  // core/types/haplotype.hpp: n reference bases more on both sides
{
    const auto* ctx = static_cast<const HaplotypeContext*>(h.payload_);
    if (!ctx || ctx->left.size() < n || ctx->right.size() < n) throw std::runtime_error {"bridge: not enough flanking reference bases for expand()"};
    Haplotype r;
    r.sequence_ = ctx->left.substr(ctx->left.size() - n) + h.sequence_ + ctx->right.substr(0, n);
    r.begin_ = h.begin_ - static_cast<std::int64_t>(n);
    r.cigar_.emplace_back(static_cast<CigarOperation::Size>(r.sequence_.size()), CigarOperation::Flag::sequenceMatch);
    return r;
}
using OptionalThreadPool = boost::optional<ThreadPool&>;                          // core/tools/read_assigner.hpp:69
namespace {
using HaplotypeLikelihoods = std::vector<std::vector<double>>;                    // read_assigner.cpp:26
#include SEAM_INC
} // namespace
} // namespace octopus

using namespace octopus;

struct ref_assigner_args {
    int32_t max_indel_error, use_int_scores, use_mapping_quality, mapping_quality_cap, mapping_quality_cap_trigger;
    const int8_t* tables; const uint32_t* table_lens;      // the error models' ten tables, as ref_error_models takes them
    uint32_t ploidy; const char* hap_bases; const uint32_t* hap_off; const int64_t* hap_begin; const uint32_t* hap_region_size;
    const char* left_bases; const uint32_t* left_off; const char* right_bases; const uint32_t* right_off;   // flanking reference bases per haplotype
    uint32_t n_reads; const char* read_bases; const uint8_t* quals; const uint32_t* read_off; const int64_t* read_begin; const uint8_t* mapq; const uint8_t* reverse;
    uint32_t n_rows; const uint32_t* row_off;              // NULL: a row per read; else templates, row r = reads [row_off[r], row_off[r + 1])
    int32_t n_threads;                                     // > 2: a ThreadPool of that size (the reference's parallel branch; the patched seam ignores it)
};

// out[k * n_rows + row] = calculate_likelihoods(genotype, reads, model, workers)[k][row]; returns 0 ok, 1 ShortHaplotypeError (*ext = required_extension), 2 other exception
extern "C" int ref_assigner_likelihoods(const ref_assigner_args* a, double* out, uint32_t* ext)
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
    std::vector<HaplotypeContext> ctx(a->ploidy);
    Genotype<Haplotype> genotype;
    for (uint32_t k = 0; k < a->ploidy; ++k) {
        Haplotype h;
        h.sequence_.assign(a->hap_bases + a->hap_off[k], a->hap_bases + a->hap_off[k + 1]); h.begin_ = a->hap_begin[k];
        ctx[k].left.assign(a->left_bases + a->left_off[k], a->left_bases + a->left_off[k + 1]);
        ctx[k].right.assign(a->right_bases + a->right_off[k], a->right_bases + a->right_off[k + 1]);
        ctx[k].region_size = a->hap_region_size[k];
        h.payload_ = &ctx[k];
        genotype.haplotypes_.push_back(std::move(h));
    }
    auto make_read = [&](uint32_t r) {
        AlignedRead x; const uint32_t o = a->read_off[r], n = a->read_off[r + 1] - o;
        x.sequence_.assign(a->read_bases + o, a->read_bases + o + n); x.base_qualities_.assign(a->quals + o, a->quals + o + n);
        x.mapping_quality_ = a->mapq[r]; x.reverse_ = a->reverse[r] != 0; x.begin_ = a->read_begin[r];
        return x;
    };
    std::unique_ptr<ThreadPool> pool;
    OptionalThreadPool workers;
    if (a->n_threads > 2) { pool = std::make_unique<ThreadPool>(static_cast<std::size_t>(a->n_threads)); workers = *pool; }
    try {
        HaplotypeLikelihoods result;
        if (!a->row_off) {
            std::vector<AlignedRead> reads;
            for (uint32_t r = 0; r < a->n_reads; ++r) reads.push_back(make_read(r));
            result = calculate_likelihoods(genotype, reads, model, workers);
        } else {
            std::vector<AlignedTemplate> templates;
            for (uint32_t row = 0; row < a->n_rows; ++row) {
                AlignedTemplate tpl;
                for (uint32_t r = a->row_off[row]; r < a->row_off[row + 1]; ++r) tpl.push_back(make_read(r));
                templates.push_back(std::move(tpl));
            }
            result = calculate_likelihoods(genotype, templates, model, workers);
        }
        for (uint32_t k = 0; k < a->ploidy; ++k) std::memcpy(out + (size_t)k * a->n_rows, result[k].data(), a->n_rows * sizeof(double));
    } catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) {
        *ext = static_cast<uint32_t>(e.required_extension());
        return 1;
    } catch (const std::exception&) {
        return 2;
    }
    return 0;
}
