// TEST INFRASTRUCTURE ONLY: boost::optional / boost::none as config/config.hpp uses them (default-initialised members), on std::optional.
#pragma once
#include <optional>
namespace boost {
template <class T> using optional = std::optional<T>;
constexpr std::nullopt_t none = std::nullopt;
}
