// TEST INFRASTRUCTURE ONLY: stand-in for core/types/genotype.hpp with the members and free functions
// core/models/genotype/constant_mixture_genotype_likelihood_model.cpp touches: an ordered multiset of haplotypes.
#pragma once
#include <cstddef>
#include <vector>
#include "core/types/haplotype.hpp"
#include "core/types/indexed_haplotype.hpp"
namespace octopus {
inline bool operator!=(const Haplotype& a, const Haplotype& b) noexcept { return !(a == b); }
template <typename T>
class Genotype
{
public:
    std::vector<T> haplotypes_;                             // kept sorted by the caller, as the reference's Genotype keeps them
    unsigned ploidy() const noexcept { return static_cast<unsigned>(haplotypes_.size()); }
    const T& operator[](std::size_t i) const noexcept { return haplotypes_[i]; }
    auto begin() const noexcept { return haplotypes_.cbegin(); }
    auto end() const noexcept { return haplotypes_.cend(); }
    auto cbegin() const noexcept { return haplotypes_.cbegin(); }
    auto cend() const noexcept { return haplotypes_.cend(); }
};
template <typename T> unsigned zygosity(const Genotype<T>& g)   // number of distinct haplotypes (sorted)
{
    unsigned z = g.ploidy() ? 1 : 0;
    for (unsigned i = 1; i < g.ploidy(); ++i) if (g[i] != g[i - 1]) ++z;
    return z;
}
template <typename T> bool is_homozygous(const Genotype<T>& g) { return zygosity(g) <= 1; }
} // namespace octopus
