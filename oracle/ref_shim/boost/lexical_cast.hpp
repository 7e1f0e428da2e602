// TEST INFRASTRUCTURE ONLY: boost::lexical_cast for the CIGAR parser in basics/cigar_string.cpp (unused by the bridges) and for the model-file reader
// custom_repeat_based_indel_error_model.cpp:138, whose accept / refuse behaviour the product must share. Boost.LexicalCast's documented behaviour for a
// character source and an integral target: an optional sign, then decimal digits, the WHOLE source consumed (no blanks before or after), the value
// representable in the target - anything else throws bad_lexical_cast (a std::bad_cast).
#pragma once
#include <cassert>      // (the real header brings it along: custom_repeat_based_indel_error_model.cpp uses assert without including it)
#include <limits>
#include <sstream>
#include <string>
#include <type_traits>
#include <typeinfo>
namespace boost {
class bad_lexical_cast : public std::bad_cast { public: const char* what() const noexcept override { return "bad lexical cast: source type value could not be interpreted as target"; } };
template <class T, class S> T lexical_cast(const S& s)
{
    const std::string str {s};
    if constexpr (std::is_integral_v<T> && !std::is_same_v<T, bool> && !std::is_same_v<T, char>) {
        size_t i = 0;
        bool neg = false;
        if (i < str.size() && (str[i] == '+' || str[i] == '-')) { neg = str[i] == '-'; ++i; }
        if (i == str.size()) throw bad_lexical_cast {};
        unsigned long long v = 0;
        for (; i < str.size(); ++i) {
            if (str[i] < '0' || str[i] > '9') throw bad_lexical_cast {};
            if (v > (std::numeric_limits<unsigned long long>::max() - 9) / 10) throw bad_lexical_cast {};
            v = v * 10 + (unsigned long long)(str[i] - '0');
        }
        if constexpr (std::is_signed_v<T>) {
            const unsigned long long lim = neg ? (unsigned long long)std::numeric_limits<T>::max() + 1ull : (unsigned long long)std::numeric_limits<T>::max();
            if (v > lim) throw bad_lexical_cast {};
            return neg ? (T)(0 - (long long)v) : (T)v;
        } else {
            if (v > (unsigned long long)std::numeric_limits<T>::max()) throw bad_lexical_cast {};
            return neg ? (T)(0 - v) : (T)v;
        }
    } else {
        std::istringstream is {str}; T v {}; is >> v;
        if (is.fail()) throw bad_lexical_cast {};
        return v;
    }
}
}
