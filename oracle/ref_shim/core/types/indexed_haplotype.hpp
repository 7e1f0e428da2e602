// TEST INFRASTRUCTURE ONLY: stand-in for core/types/indexed_haplotype.hpp: a haplotype known by its index in the likelihood array.
#pragma once
#include <cstddef>
#include "core/types/haplotype.hpp"
namespace octopus {
template <typename IndexTp = std::size_t>
class IndexedHaplotype
{
public:
    IndexTp index_ {};
    IndexedHaplotype() = default;
    explicit IndexedHaplotype(IndexTp i) : index_ {i} {}
    IndexTp index() const noexcept { return index_; }
};
template <typename I> bool operator==(const IndexedHaplotype<I>& a, const IndexedHaplotype<I>& b) noexcept { return a.index_ == b.index_; }
template <typename I> bool operator!=(const IndexedHaplotype<I>& a, const IndexedHaplotype<I>& b) noexcept { return a.index_ != b.index_; }
template <typename I> I index_of(const IndexedHaplotype<I>& h) noexcept { return h.index_; }
} // namespace octopus
