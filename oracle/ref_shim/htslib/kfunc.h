/* TEST INFRASTRUCTURE ONLY: declarations of the htslib kfunc.h functions utils/maths.hpp mentions (unused on the pair-HMM path). */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
double kf_lgamma(double z); double kf_erfc(double x); double kf_gammap(double s, double z); double kf_gammaq(double s, double z); double kf_betai(double a, double b, double x);
double kt_fisher_exact(int n11, int n12, int n21, int n22, double* _left, double* _right, double* two);
#ifdef __cplusplus
}
#endif
