// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own error-model classes, compiled in place from /root/reference/src/core/models/error:
//   BasicRepeatBasedIndelErrorModel (basic_repeat_based_indel_error_model.cpp over repeat_based_indel_error_model.cpp)
//   BasicRepeatBasedSNVErrorModel   (repeat_based_snv_error_model.cpp)
//   CustomRepeatBasedIndelErrorModel + make_penalty_map (custom_repeat_based_indel_error_model.cpp): the model read from a file
// with the reference's tandem library, on the stand-in Haplotype of oracle/ref_shim. Pins oracle/error_model_oracle.c.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "core/models/error/basic_repeat_based_indel_error_model.hpp"
#include "core/models/error/repeat_based_snv_error_model.hpp"

using namespace octopus;

// tables: the seven indel tables of n_indel entries each (AT open, CG open, dinucleotide open, trinucleotide open, homopolymer extend,
// dinucleotide extend, trinucleotide extend), then the three SNV cap tables of n_snv entries each - unexpanded vectors, as in the factory
extern "C" void ref_error_models(const int8_t* tables, const uint32_t* lens /* [10] */, const char* seq, uint32_t n, const uint8_t* substitution_mask,
                                 int8_t* gap_open, int8_t* gap_extend, char* mask_f, int8_t* prior_f, char* mask_r, int8_t* prior_r)
{
    std::vector<std::vector<std::int8_t>> t(10);
    for (int i = 0, o = 0; i < 10; o += lens[i], ++i) t[i].assign(tables + o, tables + o + lens[i]);
    BasicRepeatBasedIndelErrorModel::Parameters ip {t[0], t[1], t[2], t[3], t[4], t[5], t[6]};
    BasicRepeatBasedSNVErrorModel::Parameters sp {t[7], t[8], t[9]};
    const BasicRepeatBasedIndelErrorModel indel {ip};
    const BasicRepeatBasedSNVErrorModel snv {sp};
    Haplotype h; h.sequence_.assign(seq, seq + n);
    if (substitution_mask) {                                   // runs of substitution / sequence-match operations covering the haplotype
        for (uint32_t i = 0; i < n;) {
            uint32_t j = i; while (j < n && (substitution_mask[j] != 0) == (substitution_mask[i] != 0)) ++j;
            h.cigar_.emplace_back(j - i, substitution_mask[i] ? CigarOperation::Flag::substitution : CigarOperation::Flag::sequenceMatch);
            i = j;
        }
    } else if (n) h.cigar_.emplace_back(n, CigarOperation::Flag::sequenceMatch);
    std::vector<std::int8_t> go, ge, pf, pr; std::vector<char> mf, mr;
    indel.set_penalties(h, go, ge);
    snv.evaluate(h, mf, pf, mr, pr);
    std::memcpy(gap_open, go.data(), n); std::memcpy(gap_extend, ge.data(), n);
    std::memcpy(mask_f, mf.data(), n); std::memcpy(prior_f, pf.data(), n); std::memcpy(mask_r, mr.data(), n); std::memcpy(prior_r, pr.data(), n);
}

// std::sort exactly as sort_by_length uses it (repeat_based_indel_error_model.cpp:20-23), for pinning the oracle's restatement of libstdc++'s algorithm
#include <algorithm>
#include "tandem/tandem.hpp"
extern "C" void ref_sort_by_length(const uint32_t* lengths, uint32_t n, uint32_t* out_ids)
{
    std::vector<tandem::Repeat> v; v.reserve(n);
    for (uint32_t i = 0; i < n; ++i) v.emplace_back(i, lengths[i], i);
    std::sort(std::begin(v), std::end(v), [] (const auto& lhs, const auto& rhs) { return lhs.length < rhs.length; });
    for (uint32_t i = 0; i < n; ++i) out_ids[i] = v[i].period;
}

// The model `--sequence-error-model <file>` builds: make_penalty_map on the file's text, then the construction of make_error_model(path)
// (error_model_factory.cpp:572-590: no open map -> MalformedErrorModelFile; two-map or one-map constructor), then set_penalties (vector overload).
// Returns 0, 1 if the reference throws while reading the text, 2 for its MalformedErrorModelFile.
#include <memory>
#include "core/models/error/custom_repeat_based_indel_error_model.hpp"
extern "C" int ref_custom_indel_penalties(const char* text, size_t len, const char* seq, uint32_t n, int8_t* gap_open, int8_t* gap_extend)
{
    try {
        auto params = make_penalty_map(std::string(text, len));
        if (!params.open) return 2;
        std::unique_ptr<CustomRepeatBasedIndelErrorModel> model;
        if (params.extend) model = std::make_unique<CustomRepeatBasedIndelErrorModel>(std::move(*params.open), std::move(*params.extend));
        else model = std::make_unique<CustomRepeatBasedIndelErrorModel>(std::move(*params.open));
        Haplotype h; h.sequence_.assign(seq, seq + n);
        std::vector<std::int8_t> go, ge;
        model->set_penalties(h, go, ge);
        if (n) { std::memcpy(gap_open, go.data(), n); std::memcpy(gap_extend, ge.data(), n); }
        return 0;
    } catch (const std::exception&) { return 1; }
}
