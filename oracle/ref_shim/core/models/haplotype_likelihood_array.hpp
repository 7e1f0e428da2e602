// TEST INFRASTRUCTURE ONLY: stand-in for core/models/haplotype_likelihood_array.hpp with what the genotype likelihood model reads:
// one primed likelihood vector per haplotype, addressed by Haplotype (looked up by position in `haplotypes`) or by IndexedHaplotype.
#pragma once
#include <cstddef>
#include <functional>
#include <vector>
#include "core/types/haplotype.hpp"
#include "core/types/indexed_haplotype.hpp"
namespace octopus {
class HaplotypeLikelihoodArray
{
public:
    using LogProbability = double;
    using LikelihoodVector = std::vector<LogProbability>;
    using LikelihoodVectorRef = std::reference_wrapper<const LikelihoodVector>;
    std::vector<LikelihoodVector> columns; std::vector<Haplotype> haplotypes;
    bool is_primed() const noexcept { return true; }
    std::size_t num_likelihoods() const noexcept { return columns.empty() ? 0 : columns.front().size(); }
    const LikelihoodVector& operator[](const Haplotype& h) const { for (std::size_t i = 0; i < haplotypes.size(); ++i) if (haplotypes[i].begin_ == h.begin_ && haplotypes[i].sequence_ == h.sequence_) return columns[i]; return columns.front(); }
    template <typename I> const LikelihoodVector& operator[](const IndexedHaplotype<I>& h) const noexcept { return columns[h.index_]; }
};
} // namespace octopus
