/* TEST INFRASTRUCTURE ONLY. Run-time ISA dispatch for the reference bridge (oracle/ref_bridge.cpp).
 * isa: 0 = SSE2 ("the reference SSE pair-HMM"), 1 = AVX2, 2 = AVX-512, -1 = what upstream's
 * PairHMMSelector would pick with -march=native on this host (simd_pair_hmm_factory.hpp:55-76). */
#include <stdint.h>

#define DECL(isa) \
  int ref_phmm_align_##isa(int, int, const char*, const char*, const int8_t*, int, int, const char*, const int8_t*, \
                           const int8_t*, const int8_t*, int, int, int, int*, char*, char*, int*); \
  int ref_phmm_flank_##isa(int, int, int, int, int, const char*, const int8_t*, const char*, const int8_t*, \
                           const int8_t*, const int8_t*, int, int, const char*, const char*, int*, int*);
DECL(sse2) DECL(avx2) DECL(avx512)

int ref_phmm_isa_supported(int isa)
{
    __builtin_cpu_init();
    switch (isa) {
        case 0: return __builtin_cpu_supports("sse4.1") != 0;
        case 1: return __builtin_cpu_supports("avx2") != 0;
        case 2: return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")
                       && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
        default: return 0;
    }
}

static int pick(int isa, int band, int score_bits)
{
    if (isa >= 0) return isa;
    const int words512 = 64 / (score_bits / 8), words256 = 32 / (score_bits / 8);
    if (ref_phmm_isa_supported(2) && band % words512 == 0) return 2;
    if (ref_phmm_isa_supported(1) && band % words256 == 0) return 1;
    return 0;
}

int ref_phmm_align(int isa, int band, int score_bits,
        const char* truth, const char* target, const int8_t* quals, int truth_len, int target_len,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int gap_extend_scalar, int nuc_prior,
        int traceback, int* first_pos, char* align1, char* align2, int* status)
{
    isa = pick(isa, band, score_bits);
    if (!ref_phmm_isa_supported(isa)) { *status = -3; return 0; }
    switch (isa) {
        case 0: return ref_phmm_align_sse2(band, score_bits, truth, target, quals, truth_len, target_len, snv_mask, snv_prior, gap_open, gap_extend, gap_extend_scalar, nuc_prior, traceback, first_pos, align1, align2, status);
        case 1: return ref_phmm_align_avx2(band, score_bits, truth, target, quals, truth_len, target_len, snv_mask, snv_prior, gap_open, gap_extend, gap_extend_scalar, nuc_prior, traceback, first_pos, align1, align2, status);
        default: return ref_phmm_align_avx512(band, score_bits, truth, target, quals, truth_len, target_len, snv_mask, snv_prior, gap_open, gap_extend, gap_extend_scalar, nuc_prior, traceback, first_pos, align1, align2, status);
    }
}

int ref_phmm_flank(int isa, int band, int score_bits,
        int truth_len, int lhs_flank, int rhs_flank, const char* target, const int8_t* quals,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int nuc_prior,
        int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status)
{
    isa = pick(isa, band, score_bits);
    if (!ref_phmm_isa_supported(isa)) { *status = -3; return 0; }
    switch (isa) {
        case 0: return ref_phmm_flank_sse2(band, score_bits, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, gap_extend, nuc_prior, first_pos, aln1, aln2, mask_size, status);
        case 1: return ref_phmm_flank_avx2(band, score_bits, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, gap_extend, nuc_prior, first_pos, aln1, aln2, mask_size, status);
        default: return ref_phmm_flank_avx512(band, score_bits, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, gap_extend, nuc_prior, first_pos, aln1, aln2, mask_size, status);
    }
}

int ref_phmm_picked_isa(int band, int score_bits) { return pick(-1, band, score_bits); }

/* ISA-bound entry points with the oracle_align_fn / oracle_flank_fn signatures (oracle/phmm_oracle.h), so the
 * oracle's L2/L3 restatement can drive the reference's own L1 kernels through plain function pointers. */
#define BOUND(name, isa) \
  int ref_phmm_align_##name(int band, int score_bits, const char* truth, const char* target, const int8_t* quals, \
        int truth_len, int target_len, const char* snv_mask, const int8_t* snv_prior, const int8_t* gap_open, \
        const int8_t* gap_extend, int gap_extend_scalar, int nuc_prior, int traceback, int* first_pos, char* align1, \
        char* align2, int* status) \
  { return ref_phmm_align(isa, band, score_bits, truth, target, quals, truth_len, target_len, snv_mask, snv_prior, gap_open, \
                          gap_extend, gap_extend_scalar, nuc_prior, traceback, first_pos, align1, align2, status); } \
  int ref_phmm_flank_##name(int band, int score_bits, int truth_len, int lhs_flank, int rhs_flank, const char* target, \
        const int8_t* quals, const char* snv_mask, const int8_t* snv_prior, const int8_t* gap_open, const int8_t* gap_extend, \
        int nuc_prior, int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status) \
  { return ref_phmm_flank(isa, band, score_bits, truth_len, lhs_flank, rhs_flank, target, quals, snv_mask, snv_prior, gap_open, \
                          gap_extend, nuc_prior, first_pos, aln1, aln2, mask_size, status); }
BOUND(bound_sse2, 0)
BOUND(bound_avx2, 1)
BOUND(bound_avx512, 2)
BOUND(bound_native, -1)
