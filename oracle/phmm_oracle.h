/*
 * phmm_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference algorithm for the pair-HMM haplotype-likelihood path of
 * luntergroup/octopus v0.7.4. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product library (octopus_amd/csrc) never links, imports or executes it.
 *
 * Parity pinning (everything below is compiled in place from /root/reference into oracle/_ref/libref_phmm.so):
 *   L1  align / traceback / flank score: every golden vector of the reference's own test/unit/core/models/pair_hmm_tests.cpp
 *       (tests/golden/pair_hmm_tests.json) and the reference's SIMD headers (SSE2 / AVX2 / AVX-512) on random inputs.
 *   L2  hmm::evaluate (fast path, score-only, flank-adjusted traceback) and hmm::align (exact-match shortcut, simd_align,
 *       make_cigar, flank discount): the reference's own core/models/pairhmm/pair_hmm.hpp, hmm::PairHMM<hmm::MutationModel>
 *       (tests/test_oracle_l2.py; Boost and maths.hpp replaced by the few-line shims in oracle/ref_shim).
 *   k-mer mapper: the reference's own utils/kmer_mapper.hpp (tests/test_oracle_mapper.py).
 *   L3  HaplotypeLikelihoodModel::evaluate / align over candidate positions (in-range test, original-position rule, shifted fallback,
 *       ShortHaplotypeError, mapping-quality mixture): the reference's own core/models/haplotype_likelihood_model.cpp on stand-in
 *       Haplotype / AlignedRead types (tests/test_oracle_l3.py).
 *   Genotype read-out: the reference's own constant_mixture_genotype_likelihood_model.cpp on stand-in array / genotype types.
 *   Populate driver (rows x haplotypes loop, read hashes once per batch, k-mer table + model reset per haplotype, template sums, sample
 *       concatenation = merge_samples, ShortHaplotypeError): the reference's own core/models/haplotype_likelihood_array.cpp (both populate
 *       overloads, with and without its thread pool) on stand-in containers, oracle/_ref/libref_array.so (tests/test_oracle_populate_driver.py).
 * Nothing on the path is "parity unpinned" any more; oracle_align_batch's loop over pairs is the same loop as oracle_populate's.
 */
#ifndef PHMM_ORACLE_H
#define PHMM_ORACLE_H

#include <stdint.h>
#include "../include/oct_phmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* L1 kernel signatures — identical to oracle/_ref/libref_phmm.so's ref_phmm_align / ref_phmm_flank minus
 * the leading `isa` argument, so either can drive the upper layers. */
typedef int (*oracle_align_fn)(int band, int score_bits,
        const char* truth, const char* target, const int8_t* quals, int truth_len, int target_len,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int gap_extend_scalar, int nuc_prior,
        int traceback, int* first_pos, char* align1, char* align2, int* status);
typedef int (*oracle_flank_fn)(int band, int score_bits,
        int truth_len, int lhs_flank, int rhs_flank, const char* target, const int8_t* quals,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int nuc_prior,
        int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status);

/* simd::PairHMM::align_helper, simd_pair_hmm.hpp:240-324 (+ update_traceback :147-163, set_alignments :165-231) */
int oracle_align(int band, int score_bits,
        const char* truth, const char* target, const int8_t* quals, int truth_len, int target_len,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int gap_extend_scalar, int nuc_prior,
        int traceback, int* first_pos, char* align1, char* align2, int* status);

/* simd::PairHMM::calculate_flank_score_helper, simd_pair_hmm.hpp:347-430 */
int oracle_flank(int band, int score_bits,
        int truth_len, int lhs_flank, int rhs_flank, const char* target, const int8_t* quals,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int nuc_prior,
        int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status);

/* The L1 backend used by the layers above (default: the restatement). isa_bound_* let the caller bind the
 * reference .so's functions with a fixed ISA through small trampolines (see oracle/__init__.py). */
void oracle_set_l1_backend(oracle_align_fn align, oracle_flank_fn flank);

/* smallest B in {8,...,256} >= max_indel_error, or -1 (simd_pair_hmm_wrapper.hpp:219-241) */
int oracle_band_size(int max_indel_error);

/* hmm::detail::try_naive_evaluate, pair_hmm.hpp:278-319. Returns 1 if handled; *penalty = phred penalty
 * whose ln-probability (-ln10/10 * penalty) the reference returns. */
int oracle_try_naive_evaluate(const char* truth, int truth_len, const char* target, int target_len,
        const uint8_t* quals, uint32_t target_offset,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        uint32_t lhs_flank, uint32_t rhs_flank, int* penalty);

/* hmm::evaluate (pair_hmm.hpp:831-841) for hmm::MutationModel: fast path else simd_evaluate_helper<false_type>
 * (:722-766). kind (optional out): 0 fast path, 1 score-only DP, 2 traceback DP, 3 window overrun (lowest()). */
double oracle_evaluate(const char* truth, int truth_len, const char* target, int target_len,
        const uint8_t* quals, uint32_t target_offset, int band, int score_bits,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        uint32_t lhs_flank, uint32_t rhs_flank, int nuc_prior, int* kind);

/* compute_kmer_hashes<6> + make_kmer_hash_table<6> + map_query_to_target, utils/kmer_mapper.hpp:25-159.
 * Writes up to max_positions offsets (ascending), returns how many. */
int oracle_map_query_to_target(const char* query, int query_len, const char* target, int target_len,
        int max_positions, uint32_t* out_positions);

/* HaplotypeLikelihoodArray::populate (haplotype_likelihood_array.cpp:51-199) over the flat C-ABI batch
 * format, n_threads pthreads over haplotypes (the reference's per-haplotype tasks, :167-184).
 * Same outputs, status and stats conventions as oct_phmm_populate. */
int oracle_populate(const oct_phmm_config* cfg,
        const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
        const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
        const oct_phmm_positions* positions,
        double* out, oct_phmm_status* status, oct_phmm_stats* stats, int n_threads);

/* HaplotypeLikelihoodModel::align (haplotype_likelihood_model.cpp:322-431) for every (haplotype, read) pair of the flat batch:
 * compute_optimal_alignment :335-395 over hmm::align (pair_hmm.hpp:861-872 = try_naive_align :321-341 | simd_align :788-823,
 * make_cigar :152-188, discount_flank_score :646-673). Same conventions as oct_phmm_align. The per-position hmm::align is pinned on the reference's pair_hmm.hpp; the loop over positions is not. */
int oracle_align_batch(const oct_phmm_config* cfg,
        const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
        const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
        const oct_phmm_positions* positions,
        const oct_phmm_alignments* out, oct_phmm_status* status, int n_threads);

/* Raw-kernel timing loop for the CPU baseline: runs the current L1 backend's score-only or traceback align over
 * n windows `reps` times with n_threads threads; returns seconds. */
double oracle_time_align_windows(int band, int score_bits, uint32_t n,
        const char* truth, const uint32_t* truth_offsets,
        const char* target, const uint8_t* quals, const uint32_t* target_offsets,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        int nuc_prior, int traceback, int reps, int n_threads, int64_t* checksum);

/* ConstantMixtureGenotypeLikelihoodModel::evaluate(const Genotype<IndexedHaplotype<>>&)
 * (core/models/genotype/constant_mixture_genotype_likelihood_model.cpp:65-330) for n_genotypes genotypes of `ploidy`
 * sorted haplotype indices each, over rows [row_begin, row_end) of the columns lik[hap_out_off[h] + row].
 * Pinned on the reference's own constant_mixture_genotype_likelihood_model.cpp built in place (tests/test_oracle_l3.py). */
int oracle_genotype_likelihoods(const double* lik, const uint64_t* hap_out_off, uint32_t n_genotypes, uint32_t ploidy,
        const uint32_t* hap_indices, uint32_t row_begin, uint32_t row_end, double* out);

#ifdef __cplusplus
}
#endif
#endif
