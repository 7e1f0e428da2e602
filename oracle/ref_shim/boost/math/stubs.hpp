// TEST INFRASTRUCTURE ONLY: DECLARATIONS (no definitions) of the Boost.Math names the reference's utils/maths.hpp mentions, so that the
// real header can be compiled in place without Boost. Nothing on the pair-HMM path calls any of them; a use would fail at link time.
#pragma once
namespace boost { namespace math {
template <class T> T factorial(unsigned);
template <class T> T binomial_coefficient(unsigned, unsigned);
template <class T> T digamma(T);
template <class T> int sign(const T&);
template <class A, class B> auto gamma_p(A, B) -> decltype(A {} + B {});
template <class A, class B> auto gamma_q(A, B) -> decltype(A {} + B {});
template <class A, class B, class C> auto ibeta_inv(A, B, C) -> decltype(A {} + B {} + C {});
template <class T = double> struct beta_distribution { beta_distribution(T, T); };
template <class T = double> struct geometric_distribution { geometric_distribution(T); };
template <class T = double> struct binomial_distribution { binomial_distribution(T, T); };
template <class T = double> struct normal_distribution { normal_distribution(T, T); };
template <class D, class T> struct complemented2 { };
template <class D, class T> complemented2<D, T> complement(const D&, const T&);
template <class D, class T> double pdf(const D&, const T&);
template <class D, class T> double cdf(const D&, const T&);
template <class D, class T> double cdf(const complemented2<D, T>&);
}}
