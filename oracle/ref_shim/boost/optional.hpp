// TEST INFRASTRUCTURE ONLY: boost::optional / boost::none as the reference's model, array and config headers use them, on std::optional.
// boost::optional<T&> (HaplotypeLikelihoodArray::OptionalThreadPool) has no std counterpart: a pointer-backed stand-in with the members used.
#pragma once
#include <optional>
#include <type_traits>
namespace boost {
constexpr std::nullopt_t none = std::nullopt;
namespace shim {
template <class T> class optional_ref
{
    T* p_ = nullptr;
public:
    optional_ref() = default;
    optional_ref(std::nullopt_t) noexcept {}
    optional_ref(T& x) noexcept : p_ {&x} {}
    optional_ref& operator=(std::nullopt_t) noexcept { p_ = nullptr; return *this; }
    optional_ref& operator=(T& x) noexcept { p_ = &x; return *this; }
    explicit operator bool() const noexcept { return p_ != nullptr; }
    T& operator*() const noexcept { return *p_; }
    T* operator->() const noexcept { return p_; }
    T& get() const noexcept { return *p_; }
};
template <class T> struct pick { using type = std::optional<T>; };
template <class T> struct pick<T&> { using type = optional_ref<T>; };
} // namespace shim
template <class T> using optional = typename shim::pick<T>::type;
} // namespace boost
