// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own L3 layer, compiled in place from /root/reference/src:
//   core/models/haplotype_likelihood_model.cpp   HaplotypeLikelihoodModel::reset / evaluate(read, positions) / align(read, positions)
//                                                (max_score :211-259, is_in_range / num_out_of_range_bases :187-207, the mapping-quality
//                                                mixture :285-303 / :416-429, compute_optimal_alignment :335-395, ShortHaplotypeError)
//   core/models/error/{snv,indel}_error_model.cpp  the abstract model bases it calls
// on stand-in Haplotype / AlignedRead types (oracle/ref_shim: the real classes need HTSlib and the reference-genome machinery) and with
// the haplotype's six penalty vectors supplied through two trivial error-model subclasses, so that the vectors are the C-ABI batch's.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "core/models/haplotype_likelihood_model.hpp"
#include "core/models/error/error_model_factory.hpp"

using namespace octopus;

namespace octopus {
std::unique_ptr<SnvErrorModel> make_snv_error_model() { return nullptr; }
std::unique_ptr<IndelErrorModel> make_indel_error_model() { return nullptr; }
ErrorModel make_error_model(const std::string&) { return {}; }
}

namespace {
struct Vectors { std::vector<char> mask_f, mask_r; std::vector<std::int8_t> prior_f, prior_r, go, ge; };

class GivenSnvModel : public SnvErrorModel
{
    const Vectors* v_;
    std::unique_ptr<SnvErrorModel> do_clone() const override { return std::make_unique<GivenSnvModel>(*this); }
    void do_evaluate(const Haplotype&, MutationVector& fm, PenaltyVector& fp, MutationVector& rm, PenaltyVector& rp) const override
    { fm = v_->mask_f; fp = v_->prior_f; rm = v_->mask_r; rp = v_->prior_r; }
public:
    explicit GivenSnvModel(const Vectors* v) : v_ {v} {}
};
class GivenIndelModel : public IndelErrorModel
{
    const Vectors* v_;
    std::unique_ptr<IndelErrorModel> do_clone() const override { return std::make_unique<GivenIndelModel>(*this); }
    void do_set_penalties(const Haplotype&, PenaltyVector& go, PenaltyType& ge) const override { go = v_->go; ge = v_->ge.empty() ? 0 : v_->ge.front(); }
    void do_set_penalties(const Haplotype&, PenaltyVector& go, PenaltyVector& ge) const override { go = v_->go; ge = v_->ge; }
public:
    explicit GivenIndelModel(const Vectors* v) : v_ {v} {}
};

struct Call {
    Vectors v; Haplotype hap; AlignedRead read; HaplotypeLikelihoodModel::MappingPositionVector pos;
    std::unique_ptr<HaplotypeLikelihoodModel> model;
};

} // namespace

// plain-argument entry points -------------------------------------------------------------------------------------------------------
struct ref_model_args {
    int32_t max_indel_error, use_int_scores, use_mapping_quality, mapping_quality_cap, mapping_quality_cap_trigger, use_flank_state;
    const char* hap; uint32_t hap_len; int64_t hap_begin;
    const int8_t* gap_open; const int8_t* gap_extend; const char* mask_f; const int8_t* prior_f; const char* mask_r; const int8_t* prior_r;
    int32_t has_flank; uint32_t lhs_flank, rhs_flank;
    const char* read; const uint8_t* quals; uint32_t read_len; int64_t read_begin; uint8_t mapq; uint8_t reverse;
    const uint32_t* positions; uint32_t n_positions;
};

namespace {
void build(Call& c, const ref_model_args& a)
{
    c.v.go.assign(a.gap_open, a.gap_open + a.hap_len); c.v.ge.assign(a.gap_extend, a.gap_extend + a.hap_len);
    c.v.mask_f.assign(a.mask_f, a.mask_f + a.hap_len); c.v.mask_r.assign(a.mask_r, a.mask_r + a.hap_len);
    c.v.prior_f.assign(a.prior_f, a.prior_f + a.hap_len); c.v.prior_r.assign(a.prior_r, a.prior_r + a.hap_len);
    c.hap.sequence_.assign(a.hap, a.hap + a.hap_len); c.hap.begin_ = a.hap_begin;
    c.read.sequence_.assign(a.read, a.read + a.read_len); c.read.base_qualities_.assign(a.quals, a.quals + a.read_len);
    c.read.mapping_quality_ = a.mapq; c.read.reverse_ = a.reverse != 0; c.read.begin_ = a.read_begin;
    c.pos.assign(a.positions, a.positions + a.n_positions);
    HaplotypeLikelihoodModel::Config cfg;
    cfg.use_mapping_quality = a.use_mapping_quality != 0; cfg.mapping_quality_cap = static_cast<std::uint8_t>(a.mapping_quality_cap);
    if (a.mapping_quality_cap_trigger >= 0) cfg.mapping_quality_cap_trigger = static_cast<std::uint8_t>(a.mapping_quality_cap_trigger);
    cfg.use_flank_state = a.use_flank_state != 0; cfg.max_indel_error = static_cast<unsigned>(a.max_indel_error); cfg.use_int_scores = a.use_int_scores != 0;
    c.model = std::make_unique<HaplotypeLikelihoodModel>(std::make_unique<GivenSnvModel>(&c.v), std::make_unique<GivenIndelModel>(&c.v), cfg);
    boost::optional<HaplotypeLikelihoodModel::FlankState> fs;
    if (a.has_flank && cfg.use_flank_state) fs = HaplotypeLikelihoodModel::FlankState {a.lhs_flank, a.rhs_flank};   // what populate passes (array.cpp:140-145)
    c.model->reset(c.hap, fs);
}
}

// returns 0 ok (ln likelihood in *out), 1 ShortHaplotypeError (*ext = required_extension)
extern "C" int ref_model_evaluate(const ref_model_args* a, double* out, uint32_t* ext)
{
    Call c; build(c, *a);
    try { *out = c.model->evaluate(c.read, c.pos); return 0; }
    catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) { *ext = static_cast<uint32_t>(e.required_extension()); return 1; }
}

// returns 0 ok, 1 ShortHaplotypeError, 2 HMMOverflow; CIGAR as len << 4 | op (I 1, D 2, = 7, X 8)
extern "C" int ref_model_align(const ref_model_args* a, double* likelihood, uint32_t* mapping_position, uint32_t* ops, uint32_t cap, uint32_t* n_ops, uint32_t* ext)
{
    Call c; build(c, *a);
    try {
        const auto r = c.model->align(c.read, c.pos);
        *likelihood = r.likelihood; *mapping_position = static_cast<uint32_t>(r.mapping_position);
        uint32_t k = 0;
        for (const auto& op : r.cigar) {
            uint32_t code = 15;
            switch (op.flag()) {
                case CigarOperation::Flag::insertion: code = 1; break;
                case CigarOperation::Flag::deletion: code = 2; break;
                case CigarOperation::Flag::sequenceMatch: code = 7; break;
                case CigarOperation::Flag::substitution: code = 8; break;
                default: break;
            }
            if (k < cap) ops[k] = static_cast<uint32_t>(op.size()) << 4 | code;
            ++k;
        }
        *n_ops = k;
        return 0;
    }
    catch (const HaplotypeLikelihoodModel::ShortHaplotypeError& e) { *ext = static_cast<uint32_t>(e.required_extension()); return 1; }
    catch (const hmm::HMMOverflow&) { return 2; }
}
