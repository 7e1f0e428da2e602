// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own L2 pair-HMM layer, compiled in place from /root/reference/src:
//   hmm::PairHMM<hmm::MutationModel>::evaluate / align     core/models/pairhmm/pair_hmm.hpp:892-1032
//   (try_naive_evaluate :278-319, simd_evaluate_helper :722-766, try_naive_align :321-341, simd_align :788-823, make_cigar :152-188,
//    use_adjusted_alignment_score :123-137, calculate_flank_score :505-623, discount_flank_score :625-694)
// exactly as HaplotypeLikelihoodModel::evaluate / align instantiate it (haplotype_likelihood_model.hpp:106, .cpp:270-283, :404-415).
// Boost and the reference's maths.hpp are replaced by the few-line shims in oracle/ref_shim (see each shim's header).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "core/models/pairhmm/pair_hmm.hpp"

namespace octopus { namespace config { const std::string HelpForum {}, BugReport {"(bridge)"}; } }   // named by ProgramError::do_help only

using namespace octopus;

namespace {
struct Inputs {
    std::string truth, target; std::vector<std::uint8_t> quals;
    hmm::PenaltyVector gap_open, gap_extend, snv_priors; hmm::NucleotideVector snv_mask;
    Inputs(const char* truth_, int truth_len, const char* target_, int target_len, const uint8_t* q, const int8_t* go, const int8_t* ge,
           const char* mask, const int8_t* prior)
    : truth(truth_, truth_ + truth_len), target(target_, target_ + target_len), quals(q, q + target_len), gap_open(go, go + truth_len),
      gap_extend(ge, ge + truth_len), snv_priors(prior, prior + truth_len), snv_mask(mask, mask + truth_len) {}
};
}

// returns ln likelihood (hmm.evaluate(read sequence, haplotype sequence, qualities, position), model.cpp:217)
extern "C" double ref_hmm_evaluate(int band, int score_bits, const char* truth, int truth_len, const char* target, int target_len,
                                   const uint8_t* quals, uint32_t target_offset, const int8_t* gap_open, const int8_t* gap_extend,
                                   const char* snv_mask, const int8_t* snv_prior, uint32_t lhs_flank, uint32_t rhs_flank, int nuc_prior)
{
    const Inputs in(truth, truth_len, target, target_len, quals, gap_open, gap_extend, snv_mask, snv_prior);
    hmm::MutationModel model {in.gap_open, in.gap_extend, in.snv_mask, in.snv_priors};
    model.lhs_flank_size = lhs_flank; model.rhs_flank_size = rhs_flank; model.nuc_prior = static_cast<short>(nuc_prior);
    hmm::PairHMM<hmm::MutationModel> h {model, static_cast<unsigned>(band), score_bits == 32 ? hmm::PairHMM<hmm::MutationModel>::ScoreType::int32 : hmm::PairHMM<hmm::MutationModel>::ScoreType::int16};
    return h.evaluate(in.target, in.truth, in.quals, target_offset);
}

// hmm.align(...): returns 0, fills likelihood, target_offset and the CIGAR as BAM-encoded operations (len << 4 | op, I 1, D 2, = 7, X 8);
// returns 2 on hmm::HMMOverflow
extern "C" int ref_hmm_align(int band, int score_bits, const char* truth, int truth_len, const char* target, int target_len,
                             const uint8_t* quals, uint32_t target_offset, const int8_t* gap_open, const int8_t* gap_extend,
                             const char* snv_mask, const int8_t* snv_prior, uint32_t lhs_flank, uint32_t rhs_flank, int nuc_prior,
                             double* likelihood, uint32_t* out_offset, uint32_t* ops, uint32_t cap, uint32_t* n_ops)
{
    const Inputs in(truth, truth_len, target, target_len, quals, gap_open, gap_extend, snv_mask, snv_prior);
    hmm::MutationModel model {in.gap_open, in.gap_extend, in.snv_mask, in.snv_priors};
    model.lhs_flank_size = lhs_flank; model.rhs_flank_size = rhs_flank; model.nuc_prior = static_cast<short>(nuc_prior);
    hmm::PairHMM<hmm::MutationModel> h {model, static_cast<unsigned>(band), score_bits == 32 ? hmm::PairHMM<hmm::MutationModel>::ScoreType::int32 : hmm::PairHMM<hmm::MutationModel>::ScoreType::int16};
    try {
        const auto a = h.align(in.target, in.truth, in.quals, target_offset);
        *likelihood = a.likelihood; *out_offset = static_cast<uint32_t>(a.target_offset);
        uint32_t k = 0;
        for (const auto& op : a.cigar) {
            uint32_t code = 0;
            switch (op.flag()) {
                case CigarOperation::Flag::insertion: code = 1; break;
                case CigarOperation::Flag::deletion: code = 2; break;
                case CigarOperation::Flag::sequenceMatch: code = 7; break;
                case CigarOperation::Flag::substitution: code = 8; break;
                default: code = 15;
            }
            if (k < cap) ops[k] = static_cast<uint32_t>(op.size()) << 4 | code;
            ++k;
        }
        *n_ops = k;
        return 0;
    } catch (const hmm::HMMOverflow&) {
        return 2;
    }
}
