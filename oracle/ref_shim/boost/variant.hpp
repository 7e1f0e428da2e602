// TEST INFRASTRUCTURE ONLY: boost::variant + apply_visitor as simd_pair_hmm_wrapper.hpp uses them, on std::variant.
#pragma once
#include <utility>
#include <variant>
namespace boost {
template <class... Ts> using variant = std::variant<Ts...>;
template <class Visitor, class Variant> decltype(auto) apply_visitor(Visitor&& v, Variant&& x) { return std::visit(std::forward<Visitor>(v), std::forward<Variant>(x)); }
}
