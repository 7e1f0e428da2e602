// TEST INFRASTRUCTURE ONLY: stand-in for core/types/haplotype.hpp (the real class needs the reference-genome / allele machinery) with the
// members and free functions core/models/haplotype_likelihood_model.cpp and the error-model base classes touch.
#pragma once
#include <cstdint>
#include <string>
#include "basics/aligned_read.hpp"
#include "basics/cigar_string.hpp"
namespace octopus {
class Haplotype
{
public:
    using NucleotideSequence = std::string;
    NucleotideSequence sequence_; std::int64_t begin_ = 0; CigarString cigar_;     // cigar_: the haplotype against the reference (substitution runs matter to the SNV model)
    const NucleotideSequence& sequence() const noexcept { return sequence_; }
    CigarString cigar() const { return cigar_; }
};
inline std::size_t sequence_size(const Haplotype& h) noexcept { return h.sequence_.size(); }
inline bool contains(const Haplotype&, const AlignedRead&) noexcept { return true; }                       // concepts/mappable.hpp
inline std::int64_t begin_distance(const Haplotype& h, const AlignedRead& r) noexcept { return r.begin_ - h.begin_; }
} // namespace octopus
