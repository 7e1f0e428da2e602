// TEST INFRASTRUCTURE ONLY: shadows the reference's utils/maths.hpp (which needs Boost.Math and htslib) for the one thing
// core/models/pairhmm/pair_hmm.hpp takes from it: the constant ln10Div10 (utils/maths.hpp:41 in the reference, same digits).
#pragma once
namespace octopus { namespace maths { namespace constants {
template <typename T = double>
constexpr T ln10Div10 = T {0.230258509299404568401799145468436420760110148862877297603};
}}}
