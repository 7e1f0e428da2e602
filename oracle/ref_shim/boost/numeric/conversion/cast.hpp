// TEST INFRASTRUCTURE ONLY: boost::numeric_cast as custom_repeat_based_indel_error_model.cpp:138 uses it (int -> int8_t): the value or a bad_numeric_cast
// (Boost.NumericConversion's documented behaviour: negative_overflow / positive_overflow, both derived from bad_numeric_cast, itself a std::bad_cast).
#pragma once
#include <limits>
#include <typeinfo>
namespace boost { namespace numeric {
class bad_numeric_cast : public std::bad_cast { public: const char* what() const noexcept override { return "bad numeric conversion: overflow"; } };
class negative_overflow : public bad_numeric_cast { public: const char* what() const noexcept override { return "bad numeric conversion: negative overflow"; } };
class positive_overflow : public bad_numeric_cast { public: const char* what() const noexcept override { return "bad numeric conversion: positive overflow"; } };
template <class T, class S> T numeric_cast(S s)
{
    static_assert(std::numeric_limits<T>::is_integer && std::numeric_limits<S>::is_integer, "stand-in: integers only");
    if (static_cast<long long>(s) < static_cast<long long>(std::numeric_limits<T>::min())) throw negative_overflow {};
    if (static_cast<long long>(s) > static_cast<long long>(std::numeric_limits<T>::max())) throw positive_overflow {};
    return static_cast<T>(s);
}
} using numeric::numeric_cast; using numeric::bad_numeric_cast; }
