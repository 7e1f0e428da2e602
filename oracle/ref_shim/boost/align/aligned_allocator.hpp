// TEST INFRASTRUCTURE ONLY (oracle/): minimal stand-in for <boost/align/aligned_allocator.hpp>.
// Boost is not installed in this image; the reference's simd_pair_hmm.hpp:18,32 only needs an
// allocator that over-aligns std::vector<VectorType> storage. This file is ours, not a copy of Boost.
#ifndef ORACLE_SHIM_BOOST_ALIGNED_ALLOCATOR_HPP
#define ORACLE_SHIM_BOOST_ALIGNED_ALLOCATOR_HPP

#include <cstddef>
#include <cstdlib>
#include <new>

namespace boost { namespace alignment {

template <typename T, std::size_t Alignment = 64>
struct aligned_allocator
{
    using value_type = T;
    aligned_allocator() noexcept = default;
    template <typename U> aligned_allocator(const aligned_allocator<U, Alignment>&) noexcept {}
    template <typename U> struct rebind { using other = aligned_allocator<U, Alignment>; };
    T* allocate(std::size_t n)
    {
        void* p = nullptr;
        constexpr std::size_t al = Alignment < alignof(T) ? alignof(T) : Alignment;
        if (posix_memalign(&p, al < sizeof(void*) ? sizeof(void*) : al, n * sizeof(T) + al) != 0) throw std::bad_alloc {};
        return static_cast<T*>(p);
    }
    void deallocate(T* p, std::size_t) noexcept { std::free(p); }
    template <typename U> bool operator==(const aligned_allocator<U, Alignment>&) const noexcept { return true; }
    template <typename U> bool operator!=(const aligned_allocator<U, Alignment>&) const noexcept { return false; }
};

}} // namespace boost::alignment

#endif
