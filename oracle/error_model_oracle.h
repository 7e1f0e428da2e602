/* CPU ORACLE (test infrastructure only): per-haplotype penalty vectors, see error_model_oracle.c. */
#ifndef ERROR_MODEL_ORACLE_H
#define ERROR_MODEL_ORACLE_H
#include <stdint.h>
#include "../include/oct_phmm.h"
#ifdef __cplusplus
extern "C" {
#endif
/* oct_phmm_error_model (the tables, already expanded) is declared by the C ABI header. */

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
