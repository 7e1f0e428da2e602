/* CPU ORACLE (test infrastructure only): per-haplotype penalty vectors, see error_model_oracle.c. */
#ifndef ERROR_MODEL_ORACLE_H
#define ERROR_MODEL_ORACLE_H
#include <stdint.h>
#include "../include/oct_phmm.h"
#ifdef __cplusplus
extern "C" {
#endif
/* What HaplotypeLikelihoodModel::reset (haplotype_likelihood_model.cpp:60-78) asks of its error models, as data:
 * BasicRepeatBasedIndelErrorModel's seven tables and BasicRepeatBasedSNVErrorModel's three cap tables, ALREADY EXPANDED as
 * their constructors do (copy the first min(size, N) entries, fill the rest with the last one:
 * basic_repeat_based_indel_error_model.cpp:15-34, repeat_based_snv_error_model.cpp:20-34), e.g. from
 * error_model_factory.cpp:220-517. */
#define OCT_PHMM_INDEL_TABLE 50
#define OCT_PHMM_SNV_TABLE   51
typedef struct oct_phmm_error_model {
    int8_t  at_homopolymer_open[OCT_PHMM_INDEL_TABLE], cg_homopolymer_open[OCT_PHMM_INDEL_TABLE];
    int8_t  dinucleotide_open[OCT_PHMM_INDEL_TABLE], trinucleotide_open[OCT_PHMM_INDEL_TABLE];
    int8_t  homopolymer_extend[OCT_PHMM_INDEL_TABLE], dinucleotide_extend[OCT_PHMM_INDEL_TABLE], trinucleotide_extend[OCT_PHMM_INDEL_TABLE];
    int8_t  snv_caps[3][OCT_PHMM_SNV_TABLE];   /* homopolymer, dinucleotide, trinucleotide penalty caps */
    int32_t use_snv_model;                     /* 0: masks = the haplotype itself, priors = 100 (model.cpp:69-73) */
} oct_phmm_error_model;


/* tandem::extract_exact_tandem_repeats(str, min_period, max_period): (pos, length, period) triples in the library's output order.
 * Returns the number of repeats (written while they fit). */
int oracle_tandem_repeats(const char* str, uint32_t n, uint32_t min_period, uint32_t max_period, uint32_t* out_pos_len_period, uint32_t capacity);
/* test hook: ids 0..n-1 in the order sort_by_length (std::sort on length) leaves them */
void oracle_sort_by_length(const uint32_t* lengths, uint32_t n, uint32_t* out_ids);
/* RepeatBasedIndelErrorModel::set_penalties(haplotype, gap_open, gap_extend) with BasicRepeatBasedIndelErrorModel tables */
void oracle_indel_penalties(const oct_phmm_error_model* m, const char* seq, uint32_t n, int8_t* gap_open, int8_t* gap_extend);
/* BasicRepeatBasedSNVErrorModel::evaluate; substitution_mask (may be NULL): 1 where the haplotype's own CIGAR has a substitution */
void oracle_snv_priors(const oct_phmm_error_model* m, const char* seq, uint32_t n, const uint8_t* substitution_mask,
                       char* mask_fwd, int8_t* prior_fwd, char* mask_rev, int8_t* prior_rev);
#ifdef __cplusplus
}
#endif
#endif
