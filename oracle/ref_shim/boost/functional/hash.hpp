// TEST INFRASTRUCTURE ONLY: the Boost.Functional/Hash names basics/cigar_string.* uses (Boost is not in this image).
#pragma once
#include <cstddef>
#include <functional>
#include <iterator>
namespace boost {
template <class T> struct hash { std::size_t operator()(const T& v) const { return std::hash<T> {}(v); } };
template <class T> inline void hash_combine(std::size_t& seed, const T& v) { seed ^= hash<T> {}(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
template <class It> inline std::size_t hash_range(It first, It last)
{
    std::size_t seed = 0;
    for (; first != last; ++first) hash_combine(seed, *first);
    return seed;
}
}
