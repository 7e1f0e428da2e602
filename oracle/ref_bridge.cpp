// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or executed from the product path.
//
// extern "C" bridge over the REFERENCE's own SIMD pair-HMM headers, compiled in place from
// /root/reference/src (nothing is copied into this repo). Built by oracle/Makefile into
// oracle/_ref/libref_phmm.so. One translation unit per ISA (REF_ISA = sse2 | avx2 | avx512) because
// each needs different -m flags; oracle/ref_dispatch.c picks at run time.
//
// Reference entry points wrapped:
//   simd::PairHMM<ISA,Init>::align                  src/core/models/pairhmm/simd_pair_hmm.hpp:438-509
//   simd::PairHMM<ISA,Init>::calculate_flank_score  src/core/models/pairhmm/simd_pair_hmm.hpp:511-549
//   SSE2PairHMM / AVX2PairHMM / AVX512PairHMM       src/core/models/pairhmm/simd_pair_hmm_factory.hpp:18-45
#include <cstdint>
#include <cstddef>
#include "core/models/pairhmm/simd_pair_hmm_factory.hpp"

using namespace octopus::hmm::simd;

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#if defined(REF_ISA_SSE2)
  #define ISA_NAME sse2
  template <unsigned B, typename S> using Hmm = SSE2PairHMM<B, S>;
  template <unsigned B, typename S> constexpr bool viable = true;
#elif defined(REF_ISA_AVX2)
  #define ISA_NAME avx2
  template <unsigned B, typename S> using Hmm = AVX2PairHMM<B, S>;
  template <unsigned B, typename S> constexpr bool viable = (B % (32 / sizeof(S)) == 0);
#elif defined(REF_ISA_AVX512)
  #define ISA_NAME avx512
  template <unsigned B, typename S> using Hmm = AVX512PairHMM<B, S>;
  template <unsigned B, typename S> constexpr bool viable = (B % (64 / sizeof(S)) == 0);
#else
  #error "define REF_ISA_SSE2 / REF_ISA_AVX2 / REF_ISA_AVX512"
#endif

namespace {

struct Args
{
    const char* truth; const char* target; const std::int8_t* quals; int truth_len; int target_len;
    const char* snv_mask; const std::int8_t* snv_prior;      // both null => no-mask overload
    const std::int8_t* gap_open;                             // always an array
    const std::int8_t* gap_extend; int gap_extend_scalar;    // array, or null => scalar
    short nuc_prior;
    int traceback; int* first_pos; char* align1; char* align2;
};

template <unsigned B, typename S, bool Viable = viable<B, S>>
struct Run
{
    static int align(const Args& a, int* status)
    {
        Hmm<B, S> hmm;
        *status = 0;
        if (a.snv_mask) {
            if (!a.gap_extend) { *status = -2; return 0; }
            if (a.traceback)
                return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len, a.snv_mask, a.snv_prior,
                                 a.gap_open, a.gap_extend, a.nuc_prior, *a.first_pos, a.align1, a.align2);
            return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len, a.snv_mask, a.snv_prior,
                             a.gap_open, a.gap_extend, a.nuc_prior);
        }
        if (a.gap_extend) {
            if (a.traceback)
                return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len,
                                 a.gap_open, a.gap_extend, a.nuc_prior, *a.first_pos, a.align1, a.align2);
            return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len,
                             a.gap_open, a.gap_extend, a.nuc_prior);
        }
        const auto ge = static_cast<typename Hmm<B, S>::ScoreType>(a.gap_extend_scalar);
        if (a.traceback)
            return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len,
                             a.gap_open, ge, a.nuc_prior, *a.first_pos, a.align1, a.align2);
        return hmm.align(a.truth, a.target, a.quals, a.truth_len, a.target_len, a.gap_open, ge, a.nuc_prior);
    }
    static int flank(int truth_len, int lhs, int rhs, const char* target, const std::int8_t* quals,
                     const char* snv_mask, const std::int8_t* snv_prior,
                     const std::int8_t* gap_open, const std::int8_t* gap_extend, short nuc_prior,
                     int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status)
    {
        Hmm<B, S> hmm;
        *status = 0;
        if (snv_mask)
            return hmm.calculate_flank_score(truth_len, lhs, rhs, target, quals, snv_mask, snv_prior,
                                             gap_open, gap_extend, nuc_prior, first_pos, aln1, aln2, *mask_size);
        // The reference's no-mask overload (simd_pair_hmm.hpp:511-528) does not compile when instantiated
        // (it passes NullType as `target` to get_mismatch_quality, :388); upstream never instantiates it.
        *status = -2;
        return 0;
    }
};

template <unsigned B, typename S>
struct Run<B, S, false>
{
    static int align(const Args&, int* status) { *status = -1; return 0; }
    static int flank(int, int, int, const char*, const std::int8_t*, const char*, const std::int8_t*,
                     const std::int8_t*, const std::int8_t*, short, int, const char*, const char*, int*, int* status)
    { *status = -1; return 0; }
};

template <typename S>
int align_band(int band, const Args& a, int* status)
{
    switch (band) {
        case 8:   return Run<8, S>::align(a, status);
        case 16:  return Run<16, S>::align(a, status);
        case 32:  return Run<32, S>::align(a, status);
        case 64:  return Run<64, S>::align(a, status);
        case 128: return Run<128, S>::align(a, status);
        case 256: return Run<256, S>::align(a, status);
        default: *status = -1; return 0;
    }
}

template <typename S, typename... Ts>
int flank_band(int band, int* status, Ts... ts)
{
    switch (band) {
        case 8:   return Run<8, S>::flank(ts..., status);
        case 16:  return Run<16, S>::flank(ts..., status);
        case 32:  return Run<32, S>::flank(ts..., status);
        case 64:  return Run<64, S>::flank(ts..., status);
        case 128: return Run<128, S>::flank(ts..., status);
        case 256: return Run<256, S>::flank(ts..., status);
        default: *status = -1; return 0;
    }
}

} // namespace

// status: 0 ok, -1 (band, score_bits) not available for this ISA, -2 unsupported overload combination
extern "C" int CAT(ref_phmm_align_, ISA_NAME)(int band, int score_bits,
        const char* truth, const char* target, const std::int8_t* quals, int truth_len, int target_len,
        const char* snv_mask, const std::int8_t* snv_prior,
        const std::int8_t* gap_open, const std::int8_t* gap_extend, int gap_extend_scalar, int nuc_prior,
        int traceback, int* first_pos, char* align1, char* align2, int* status)
{
    Args a {truth, target, quals, truth_len, target_len, snv_mask, snv_prior, gap_open, gap_extend,
            gap_extend_scalar, static_cast<short>(nuc_prior), traceback, first_pos, align1, align2};
    return score_bits == 32 ? align_band<int>(band, a, status) : align_band<short>(band, a, status);
}

extern "C" int CAT(ref_phmm_flank_, ISA_NAME)(int band, int score_bits,
        int truth_len, int lhs_flank, int rhs_flank, const char* target, const std::int8_t* quals,
        const char* snv_mask, const std::int8_t* snv_prior,
        const std::int8_t* gap_open, const std::int8_t* gap_extend, int nuc_prior,
        int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status)
{
    const short np = static_cast<short>(nuc_prior);
    return score_bits == 32
        ? flank_band<int>(band, status, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, gap_extend, np, first_pos, aln1, aln2, mask_size)
        : flank_band<short>(band, status, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, gap_extend, np, first_pos, aln1, aln2, mask_size);
}
