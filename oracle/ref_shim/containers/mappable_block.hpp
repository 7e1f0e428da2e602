// TEST INFRASTRUCTURE ONLY: stand-in for containers/mappable_block.hpp: a block of haplotypes is a vector here (the real one adds the
// shared genomic region, which core/models/haplotype_likelihood_array.cpp never reads).
#pragma once
#include <vector>
namespace octopus {
template <typename T> class MappableBlock : public std::vector<T> { public: using std::vector<T>::vector; };
} // namespace octopus
