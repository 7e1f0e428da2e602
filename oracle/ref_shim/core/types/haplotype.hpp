// TEST INFRASTRUCTURE ONLY: stand-in for core/types/haplotype.hpp (the real class needs the reference-genome / allele machinery) with the
// members and free functions core/models/haplotype_likelihood_model.cpp and the error-model base classes touch.
#pragma once
#include <cstdint>
#include <string>
#include <functional>
#include "basics/aligned_read.hpp"
#include "basics/cigar_string.hpp"
namespace octopus {
class Haplotype
{
public:
    using NucleotideSequence = std::string;
    NucleotideSequence sequence_; std::int64_t begin_ = 0; CigarString cigar_;     // cigar_: the haplotype against the reference (substitution runs matter to the SNV model)
    const void* payload_ = nullptr;                                                // test bridges: the penalty vectors that belong to this haplotype
    const NucleotideSequence& sequence() const noexcept { return sequence_; }
    CigarString cigar() const { return cigar_; }
};
// what core/models/haplotype_likelihood_array.hpp needs of the real header: equality, the two hashes, the debug printer's declaration
inline bool operator==(const Haplotype& a, const Haplotype& b) noexcept { return a.begin_ == b.begin_ && a.sequence_ == b.sequence_; }
struct HaplotypeHash { std::size_t operator()(const Haplotype& h) const noexcept { return std::hash<std::string> {}(h.sequence_) * 31u + static_cast<std::size_t>(h.begin_); } };
namespace debug { template <typename S> void print_variant_alleles(S&, const Haplotype&) {} }
inline std::size_t sequence_size(const Haplotype& h) noexcept { return h.sequence_.size(); }
inline bool contains(const Haplotype&, const AlignedRead&) noexcept { return true; }                       // concepts/mappable.hpp
inline std::int64_t begin_distance(const Haplotype& h, const AlignedRead& r) noexcept { return r.begin_ - h.begin_; }
inline std::int64_t mapped_begin(const Haplotype& h) noexcept { return h.begin_; }                        // concepts/mappable.hpp:181
inline std::int64_t mapped_begin(const AlignedRead& r) noexcept { return r.begin_; }
} // namespace octopus
namespace std {
template <> struct hash<reference_wrapper<const octopus::Haplotype>>
{ size_t operator()(reference_wrapper<const octopus::Haplotype> h) const noexcept { return octopus::HaplotypeHash {}(h.get()); } };
}
