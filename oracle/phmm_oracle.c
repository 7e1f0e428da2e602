/*
 * phmm_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE, NOT PRODUCT CODE (see phmm_oracle.h).
 *
 * Scalar restatement of the reference's banded min-plus pair-HMM and the layers that drive it. Every
 * function cites the reference lines it follows (paths relative to /root/reference/src). Arithmetic is
 * carried in int32 and wrapped to the reference's lane type (int16 or int32) after every add, so that
 * even overflowing inputs follow the SIMD lanes bit for bit.
 *
 * PINNED (tests/test_oracle_*.py): to all 24 records of the reference's own unit tests (test/unit/core/models/pair_hmm_tests.cpp ->
 * tests/golden/pair_hmm_tests.json, needs nothing but the fixture), and - exact equality on thousands of seeded cases per layer - to the
 * reference's own sources compiled where they lie (oracle/Makefile -> oracle/_ref; that build stands on few-line stand-ins for the headers this
 * image lacks, oracle/ref_shim: disclosed in DESIGN.md section 7).
 */
#define _GNU_SOURCE
#include "phmm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

#define MAX_BAND 256
#define TRACE_BITS 2                 /* simd_pair_hmm.hpp:57 */
#define N_SCORE (2 << TRACE_BITS)    /* :58 */
#define MAX_QUALITY 64               /* :60 */
#define LN10_DIV_10 0.230258509299404568401799145468436420760110148862877297603 /* utils/maths.hpp:41 */

/* ------------------------------------------------------------------------------------------------ */
/* L1: simd::PairHMM::align_helper                                                                  */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
    int bits;
    int32_t inf;   /* infinity_ = max - 0x7FF, simd_pair_hmm.hpp:55-56 */
    int32_t nul;   /* null_score_ = min, :61 */
} lane_mode;

static lane_mode make_mode(int score_bits)
{
    lane_mode m;
    m.bits = score_bits;
    if (score_bits == 32) { m.inf = INT32_MAX - 0x7FF; m.nul = INT32_MIN; }
    else                  { m.inf = INT16_MAX - 0x7FF; m.nul = INT16_MIN; }
    return m;
}

/* wrap a wide value to the lane type (what _mm_add_epi16/_mm_add_epi32 and insert-truncation do) */
static inline int32_t wr(int64_t v, int bits)
{
    return bits == 16 ? (int32_t)(int16_t)(uint16_t)(uint64_t)v : (int32_t)(uint32_t)(uint64_t)v;
}
static inline int32_t min32(int32_t a, int32_t b) { return a < b ? a : b; }

int oracle_band_size(int max_indel_error)
{
    /* simd_pair_hmm_wrapper.hpp:219-241: smallest of 8..256 that is >= the request */
    for (int b = 8; b <= MAX_BAND; b *= 2) if (max_indel_error <= b) return b;
    return -1;
}

typedef struct {
    const char* truth; const char* target; const int8_t* quals; int L, T;
    const char* mask; const int8_t* prior;          /* NULL, NULL => no-mask overload */
    const int8_t* go; const int8_t* ge; int ge_scalar;
    lane_mode md;
} l1_in;

/* value held by lane i of each sliding window, expressed by position instead of by shifting
 * (windows: simd_pair_hmm.hpp:258-265 initial fill, :273-281 target side, :298-306 truth side) */
static inline int32_t win_target(const l1_in* in, int t)
{
    if (t < 0) return in->md.inf;                     /* _targetwin starts as _inf (:259) */
    if (t < in->T) return (int32_t)in->target[t];      /* char sign-extends like _mm_set_epi16(char) */
    return (int32_t)'0';                               /* :279 */
}
static inline int32_t win_qual(const l1_in* in, int t)
{
    if (t < 0 || t >= in->T) return wr((int64_t)MAX_QUALITY << TRACE_BITS, in->md.bits); /* :260,280 */
    return wr((int64_t)in->quals[t] * 4, in->md.bits);                                     /* :277 */
}
static inline int32_t win_truth(const l1_in* in, int x) { return x < in->L ? (int32_t)in->truth[x] : (int32_t)'N'; } /* :300 */
static inline int32_t win_gap(const l1_in* in, const int8_t* arr, int scalar, int x)
{
    if (!arr) return wr((int64_t)(int8_t)scalar * 4, in->md.bits);         /* vectorise_left_shift_bits(int8_t), :74-78 */
    const int idx = x < in->L ? x : in->L - 1;                              /* :303 */
    return wr((int64_t)arr[idx] * 4, in->md.bits);                          /* :103 */
}
static inline int32_t win_mask(const l1_in* in, int x) { return x < in->L ? (int32_t)in->mask[x] : (int32_t)'N'; } /* :116 */
static inline int32_t win_prior(const l1_in* in, int x)
{
    if (x < in->L) return wr((int64_t)in->prior[x] * 4, in->md.bits);
    return wr((int64_t)in->md.inf * 4, in->md.bits);                        /* :117, truncated on insert */
}

/* update_match_state, simd_pair_hmm.hpp:121-142 */
static inline int32_t match_cost(const l1_in* in, int t, int x)
{
    const int32_t tw = win_target(in, t), hw = win_truth(in, x), qw = win_qual(in, t);
    const int32_t nq = (hw == (int32_t)'N') ? N_SCORE : in->md.inf;         /* _truthnqual :265,302 */
    int32_t c;
    if (in->mask) {
        const int32_t inner = (tw == win_mask(in, x)) ? win_prior(in, x) : qw;
        c = min32(qw, inner);
    } else {
        c = qw;
    }
    if (tw == hw) c = 0;                                                    /* _andnot(_cmpeq(target, truth), ...) */
    return min32(c, nq);
}

int oracle_align(int band, int score_bits,
        const char* truth, const char* target, const int8_t* quals, int truth_len, int target_len,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int gap_extend_scalar, int nuc_prior,
        int traceback, int* first_pos, char* align1, char* align2, int* status)
{
    if (status) *status = 0;
    const int B = band, T = target_len, L = truth_len;
    if (B < 8 || B > MAX_BAND || (B & (B - 1)) || (score_bits != 16 && score_bits != 32)
        || T <= 0 || L != T + 2 * B - 1 || !gap_open || (!!snv_mask != !!snv_prior)) {
        if (status) *status = -1;
        return 0;
    }
    l1_in in = {truth, target, quals, L, T, snv_mask, snv_prior, gap_open, gap_extend, gap_extend_scalar, make_mode(score_bits)};
    const int bits = score_bits;
    const int32_t inf = in.md.inf, nul = in.md.nul;
    const int32_t nuc4 = wr((int64_t)(int8_t)nuc_prior * 4, bits);          /* :257 via the int8_t overload :74-78 */
    int32_t M1[MAX_BAND], I1[MAX_BAND], D1[MAX_BAND], M2[MAX_BAND], I2[MAX_BAND], D2[MAX_BAND], tmp[MAX_BAND];
    for (int i = 0; i < B; ++i) M1[i] = I1[i] = D1[i] = M2[i] = I2[i] = D2[i] = inf;   /* :267 */
    const int n_diag = 2 * (T + B) + 1;                                     /* make_traceback_array :144 */
    int32_t* bp = NULL;
    if (traceback) {
        bp = (int32_t*)calloc((size_t)n_diag * B, sizeof(int32_t));
        if (!bp) { if (status) *status = -4; return 0; }
    }
    int32_t minscore = inf;                                                 /* :269 */
    int minscoreidx = -1;
    for (int k = 0; k < T + B; ++k) {                                       /* s = 2k, :271 */
        const int s = 2 * k;
        /* rolling initialiser: lane k of both M vectors := null_score_ while k < B (rolling_initializer.hpp:39-51, :282-283) */
        if (k < B) { M1[k] = nul; M2[k] = nul; }
        /* ---- even diagonal: lane i is cell (t = k-i, x = k+i) ---- */
        for (int i = 0; i < B; ++i) M1[i] = min32(M1[i], min32(I1[i], D1[i]));          /* :284 */
        if (k >= T) {                                                       /* :285-291 */
            const int32_t cur = M1[k - T];
            if (cur < minscore) { minscore = cur; minscoreidx = s; }
        }
        for (int i = 0; i < B; ++i) M1[i] = wr((int64_t)M1[i] + match_cost(&in, k - i, k + i), bits);   /* :292 */
        /* :293-294: D1 = insert_bottom(left_shift_word(min(D2 + rshift(GE), min(M2,I2) + rshift(GO))), inf) */
        for (int i = 0; i < B; ++i) {
            /* right-shifted penalty window: lane i sees lane i+1's value, top lane sees 0 (only for arrays, :92-99) */
            const int32_t ges = gap_extend ? (i + 1 < B ? win_gap(&in, gap_extend, 0, k + i + 1) : 0)
                                           : win_gap(&in, NULL, gap_extend_scalar, 0);
            const int32_t gos = (i + 1 < B ? win_gap(&in, gap_open, 0, k + i + 1) : 0);
            tmp[i] = min32(wr((int64_t)D2[i] + ges, bits), wr((int64_t)min32(M2[i], I2[i]) + gos, bits));
        }
        for (int i = B - 1; i >= 1; --i) D1[i] = tmp[i - 1];
        D1[0] = inf;
        for (int i = 0; i < B; ++i) {                                       /* :295 */
            const int32_t ge = win_gap(&in, gap_extend, gap_extend_scalar, k + i), go = win_gap(&in, gap_open, 0, k + i);
            I1[i] = wr((int64_t)min32(wr((int64_t)I2[i] + ge, bits), wr((int64_t)M2[i] + go, bits)) + nuc4, bits);
        }
        if (traceback) {                                                    /* update_traceback :147-163 */
            for (int i = 0; i < B; ++i) {
                bp[(size_t)s * B + i] = (M1[i] & 3) | ((I1[i] & 3) << 2) | ((D1[i] & 3) << 6);
                M1[i] &= ~3; I1[i] = (I1[i] & ~3) | 1; D1[i] = (D1[i] & ~3) | 3;
            }
        }
        /* ---- odd diagonal: lane i is cell (t = k-i, x = k+1+i) (:298-320) ---- */
        for (int i = 0; i < B; ++i) M2[i] = min32(M2[i], min32(I2[i], D2[i]));          /* :308 */
        if (k >= T) {
            const int32_t cur = M2[k - T];
            if (cur < minscore) { minscore = cur; minscoreidx = s + 1; }
        }
        for (int i = 0; i < B; ++i) M2[i] = wr((int64_t)M2[i] + match_cost(&in, k - i, k + 1 + i), bits); /* :316 */
        for (int i = 0; i < B; ++i) {                                       /* :317 */
            const int32_t ge = win_gap(&in, gap_extend, gap_extend_scalar, k + 1 + i), go = win_gap(&in, gap_open, 0, k + 1 + i);
            D2[i] = min32(wr((int64_t)D1[i] + ge, bits), wr((int64_t)min32(M1[i], I1[i]) + go, bits));
        }
        for (int i = 0; i < B; ++i) {                                       /* :318-319 */
            if (i == B - 1) { I2[i] = inf; continue; }
            const int32_t ge = win_gap(&in, gap_extend, gap_extend_scalar, k + 1 + i), go = win_gap(&in, gap_open, 0, k + 1 + i);
            I2[i] = wr((int64_t)min32(wr((int64_t)I1[i + 1] + ge, bits), wr((int64_t)M1[i + 1] + go, bits)) + nuc4, bits);
        }
        if (traceback) {
            for (int i = 0; i < B; ++i) {
                bp[(size_t)(s + 1) * B + i] = (M2[i] & 3) | ((I2[i] & 3) << 2) | ((D2[i] & 3) << 6);
                M2[i] &= ~3; I2[i] = (I2[i] & ~3) | 1; D2[i] = (D2[i] & ~3) | 3;
            }
        }
    }
    if (traceback) {
        /* set_alignments, simd_pair_hmm.hpp:165-231 */
        const int64_t n_flat = (int64_t)n_diag * B;
        int fp = 0;
        int done = 0;
        if (minscoreidx < 0) fp = -1;                                       /* :176-179 (falls through) */
        int sidx = minscoreidx;
        int i = sidx / 2 - T;                                               /* C division truncates toward zero, as C++ */
        int y = T;
        int x = sidx - y;
        int alnidx = 0;
        const int64_t flat0 = (int64_t)sidx * B + i;
        if (flat0 < 0 || flat0 >= n_flat) { fp = -1; done = 1; }            /* :186-190 */
        if (!done) {
            int state = bp[flat0] & 3;                                      /* :191, match_label_ = 0 */
            sidx -= 2;
            while (y > 0) {                                                 /* :194 */
                if (sidx < 0 || i < 0) { fp = -1; done = 1; break; }        /* :195-199 */
                const int64_t f = (int64_t)sidx * B + i;
                if (f >= n_flat) { fp = -1; done = 1; break; }              /* reference would read past the array (UB) */
                const int new_state = (bp[f] >> (2 * state)) & 3;           /* :200 */
                if (state == 0) {                                           /* match :201-204 */
                    sidx -= 2;
                    align1[alnidx] = truth[--x];
                    align2[alnidx] = target[--y];
                } else if (state == 1) {                                    /* insert :205-209 */
                    i += sidx & 1;
                    sidx -= 1;
                    align1[alnidx] = '-';
                    align2[alnidx] = target[--y];
                } else {                                                    /* delete :210-215 */
                    sidx -= 1;
                    i -= sidx & 1;
                    align1[alnidx] = truth[--x];
                    align2[alnidx] = '-';
                }
                state = new_state;
                ++alnidx;
            }
            if (!done) {
                align1[alnidx] = 0; align2[alnidx] = 0;
                fp = x;
                for (int a = 0, b = alnidx - 1; a < b; ++a, --b) {          /* :223-230 */
                    char c = align1[a]; align1[a] = align1[b]; align1[b] = c;
                    c = align2[a]; align2[a] = align2[b]; align2[b] = c;
                }
            }
        }
        if (first_pos) *first_pos = fp;
        free(bp);
    }
    /* :323  (minscore - null_score_) >> trace_bits_, in int; for int32 lanes the subtraction wraps */
    if (bits == 16) return ((int32_t)minscore - (int32_t)nul) >> TRACE_BITS;
    return (int32_t)((uint32_t)minscore - (uint32_t)nul) >> TRACE_BITS;
}

/* calculate_flank_score_helper, simd_pair_hmm.hpp:347-430 (masked overload :530-549; the unmasked one does not instantiate) */
int oracle_flank(int band, int score_bits,
        int truth_len, int lhs_flank, int rhs_flank, const char* target, const int8_t* quals,
        const char* snv_mask, const int8_t* snv_prior,
        const int8_t* gap_open, const int8_t* gap_extend, int nuc_prior,
        int first_pos, const char* aln1, const char* aln2, int* mask_size, int* status)
{
    (void)band; (void)score_bits;
    if (status) *status = 0;
    char prev = 'M';
    int truth_idx = first_pos, target_idx = 0, a = 0, result = 0, msz = 0;
    const int rhs_begin = truth_len - rhs_flank;
    const short np = (short)nuc_prior;
    while (aln1[a]) {
        char st = 'M';
        if (aln1[a] == '-') st = 'I'; else if (aln2[a] == '-') st = 'D';
        const int in_flank = truth_idx < lhs_flank || truth_idx >= rhs_begin;
        if (st == 'M') {
            if (in_flank) {
                if (aln1[a] != aln2[a]) {
                    if (aln1[a] != 'N') {
                        /* get_mismatch_quality :329-336 (std::min on int8) */
                        int8_t q = quals[target_idx];
                        if (snv_mask[truth_idx] == target[target_idx] && snv_prior[truth_idx] < q) q = snv_prior[truth_idx];
                        result += q;
                    } else {
                        result += N_SCORE >> TRACE_BITS;
                    }
                }
                ++msz;
            }
            ++truth_idx; ++target_idx;
        } else if (st == 'I') {
            if (in_flank) {
                /* :402-407; index truth_idx-1 can be -1 when the alignment opens with an insertion at window
                 * start: the reference then reads one byte before its (offset) penalty pointer. We clamp to 0;
                 * documented divergence on reference UB. */
                const int gi = truth_idx - 1 < 0 ? 0 : truth_idx - 1;
                result += (prev == 'I' ? gap_extend[gi] : gap_open[gi]) + np;
                ++msz;
            }
            ++target_idx;
        } else {
            if (in_flank) result += (prev == 'D' ? gap_extend[truth_idx] : gap_open[truth_idx]);
            ++truth_idx;
        }
        ++a;
        prev = st;
    }
    if (mask_size) *mask_size = msz;
    return result;
}

static oracle_align_fn g_align = oracle_align;
static oracle_flank_fn g_flank = oracle_flank;

void oracle_set_l1_backend(oracle_align_fn align, oracle_flank_fn flank)
{
    g_align = align ? align : oracle_align;
    g_flank = flank ? flank : oracle_flank;
}

/* ------------------------------------------------------------------------------------------------ */
/* L2: pair_hmm.hpp                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* is_in_flank, pair_hmm.hpp:206-214 (size_t arithmetic) */
static inline int is_in_flank(uint64_t truth_idx, uint64_t truth_length, uint64_t lhs, uint64_t rhs)
{
    return truth_idx < lhs || truth_idx >= (truth_length - rhs);
}

int oracle_try_naive_evaluate(const char* truth, int truth_len, const char* target, int target_len,
        const uint8_t* quals, uint32_t target_offset,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        uint32_t lhs_flank, uint32_t rhs_flank, int* penalty)
{
    /* try_naive_evaluate, pair_hmm.hpp:278-319 */
    const char* tr = truth + target_offset;
    const int T = target_len;
    int i1 = 0;
    while (i1 < T && target[i1] == tr[i1]) ++i1;                            /* m1 :288 */
    if (i1 == T) { *penalty = 0; return 1; }                                /* :289-291 */
    int i2 = i1 + 1;
    while (i2 < T && target[i2] == tr[i2]) ++i2;                            /* m2 :292 */
    if (i2 == T) {
        const uint64_t truth_mismatch_idx = (uint64_t)i1 + target_offset;   /* :297 */
        if (is_in_flank(truth_mismatch_idx, (uint64_t)truth_len, lhs_flank, rhs_flank)) { *penalty = 0; return 1; } /* :298 */
        /* get_mismatch_penalty :250-263 */
        uint8_t mismatch_penalty = quals[i1];
        if (snv_mask[truth_mismatch_idx] == target[i1]) {
            const uint8_t p = (uint8_t)snv_prior[truth_mismatch_idx];
            if (p < mismatch_penalty) mismatch_penalty = p;
        }
        const int gap_open_penalty = gap_open[truth_mismatch_idx];          /* int8 :301 */
        if ((int)mismatch_penalty <= gap_open_penalty) { *penalty = mismatch_penalty; return 1; } /* :302-303 */
        /* :305 std::equal(next(m1.first), cend(target), m1.second) */
        if (memcmp(target + i1 + 1, tr + i1, (size_t)(T - i1 - 1)) == 0) { *penalty = gap_open_penalty; return 1; }
        /* :309 std::equal(m1.first, cend(target), next(m1.second)) */
        if (memcmp(target + i1, tr + i1 + 1, (size_t)(T - i1)) == 0) { *penalty = gap_open_penalty; return 1; }
        if ((int)mismatch_penalty <= gap_open_penalty + (int)gap_extend[truth_mismatch_idx]) { *penalty = mismatch_penalty; return 1; } /* :313 */
    }
    return 0;
}

static double phred_to_ln(double penalty) { return -LN10_DIV_10 * penalty; } /* :716 and lookup :106-115 */

typedef struct { char* a1; char* a2; size_t cap; } aln_scratch;

static double simd_evaluate(const char* truth, int truth_len, const char* target, int T, const uint8_t* quals,
        uint32_t target_offset, int band, int score_bits,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        uint32_t lhs_flank, uint32_t rhs_flank, int nuc_prior, aln_scratch* scr, int* kind)
{
    /* simd_evaluate_helper<false_type>, pair_hmm.hpp:722-766 */
    const int pad = band;
    const int truth_alignment_size = T + 2 * pad - 1;
    int alignment_offset = (int)target_offset - pad;
    if (alignment_offset < 0) alignment_offset = 0;                         /* :735 */
    if (alignment_offset + truth_alignment_size > truth_len) {              /* :736-738 */
        if (kind) *kind = 3;
        return OCT_PHMM_LOWEST;
    }
    const int8_t* q8 = (const int8_t*)quals;                                /* reinterpret_cast :372 */
    int st = 0;
    /* use_adjusted_alignment_score :123-137 (size_t arithmetic) */
    const int adjusted = (uint64_t)target_offset < ((uint64_t)lhs_flank + (uint64_t)pad)
        || ((uint64_t)target_offset + (uint64_t)T + (uint64_t)pad) > ((uint64_t)truth_len - (uint64_t)rhs_flank);
    if (!adjusted) {
        if (kind) *kind = 1;
        const int score = g_align(band, score_bits, truth + alignment_offset, target, q8, truth_alignment_size, T,
                                  snv_mask + alignment_offset, snv_prior + alignment_offset,
                                  gap_open + alignment_offset, gap_extend + alignment_offset, 0, nuc_prior,
                                  0, NULL, NULL, NULL, &st);
        return -LN10_DIV_10 * (double)score;                                /* :741 */
    }
    if (kind) *kind = 2;
    const size_t need = (size_t)(2 * (T + pad)) + 1;                        /* :744-746 */
    if (scr->cap < need) {
        scr->a1 = (char*)realloc(scr->a1, need); scr->a2 = (char*)realloc(scr->a2, need); scr->cap = need;
    }
    memset(scr->a1, 0, need); memset(scr->a2, 0, need);
    int first_pos = 0;
    const int score = g_align(band, score_bits, truth + alignment_offset, target, q8, truth_alignment_size, T,
                              snv_mask + alignment_offset, snv_prior + alignment_offset,
                              gap_open + alignment_offset, gap_extend + alignment_offset, 0, nuc_prior,
                              1, &first_pos, scr->a1, scr->a2, &st);
    if (first_pos == -1) return OCT_PHMM_LOWEST;                            /* :750-752 */
    /* calculate_flank_score<false_type>, pair_hmm.hpp:558-602: flanks re-expressed in window coordinates */
    int lhs = (int)lhs_flank;
    if (lhs < alignment_offset) lhs = 0; else { lhs -= alignment_offset; if (lhs < 0) lhs = 0; }
    int rhs = (int)rhs_flank;
    if (alignment_offset + truth_alignment_size < truth_len - rhs) rhs = 0;
    else { rhs += alignment_offset + truth_alignment_size; rhs -= truth_len; if (rhs < 0) rhs = 0; }
    int target_mask_size = 0;
    int flank_score = g_flank(band, score_bits, truth_alignment_size, lhs, rhs, target, q8,
                              snv_mask + alignment_offset, snv_prior + alignment_offset,
                              gap_open + alignment_offset, gap_extend + alignment_offset, nuc_prior,
                              first_pos, scr->a1, scr->a2, &target_mask_size, &st);
    const int num_explained_bases = T - target_mask_size;                   /* :757-759 */
    if (num_explained_bases < 2) flank_score = 0;
    if (flank_score <= score) return -LN10_DIV_10 * (double)(score - flank_score); /* :760-761 */
    return -LN10_DIV_10 * (flank_score + score);                            /* :763 */
}

double oracle_evaluate(const char* truth, int truth_len, const char* target, int target_len,
        const uint8_t* quals, uint32_t target_offset, int band, int score_bits,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        uint32_t lhs_flank, uint32_t rhs_flank, int nuc_prior, int* kind)
{
    /* hmm::evaluate, pair_hmm.hpp:831-841 */
    int penalty = 0;
    if (oracle_try_naive_evaluate(truth, truth_len, target, target_len, quals, target_offset, gap_open, gap_extend,
                                  snv_mask, snv_prior, lhs_flank, rhs_flank, &penalty)) {
        if (kind) *kind = 0;
        return phred_to_ln((double)penalty);
    }
    aln_scratch scr = {NULL, NULL, 0};
    const double r = simd_evaluate(truth, truth_len, target, target_len, quals, target_offset, band, score_bits,
                                   gap_open, gap_extend, snv_mask, snv_prior, lhs_flank, rhs_flank, nuc_prior, &scr, kind);
    free(scr.a1); free(scr.a2);
    return r;
}

/* ------------------------------------------------------------------------------------------------ */
/* K-mer mapper: utils/kmer_mapper.hpp                                                              */
/* ------------------------------------------------------------------------------------------------ */
#define KMER 6
#define NUM_KMERS 4096   /* num_kmers(6), kmer_mapper.hpp:19-22 */

static inline unsigned base_code(char b)
{
    /* perfect_hash, kmer_mapper.hpp:25-39: A0 C1 G2 T3, everything else 0 (table is 128 entries; bytes >= 128 are
     * out of the reference's table, we give them 0 as well) */
    switch (b) { case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; }
}
static inline unsigned kmer_hash(const char* s)
{
    unsigned h = 0, k = 1;                                                  /* perfect_kmer_hash :43-53 */
    for (int j = 0; j < KMER; ++j) { h += k * base_code(s[j]); k *= 4; }
    return h;
}

typedef struct {
    uint32_t bin_start[NUM_KMERS + 1];   /* CSR of make_kmer_hash_table :85-106: bins hold ascending target indices */
    uint32_t* idx; size_t idx_cap;
    uint32_t n_kmers;                    /* table.second */
    uint32_t* counts; size_t counts_cap; /* MappedIndexCounts :108-118 */
} kmer_table;

static void kmer_table_build(kmer_table* t, const char* seq, int len)
{
    memset(t->bin_start, 0, sizeof(t->bin_start));
    t->n_kmers = len >= KMER ? (uint32_t)(len - KMER + 1) : 0;
    if (t->idx_cap < t->n_kmers) { t->idx = (uint32_t*)realloc(t->idx, sizeof(uint32_t) * t->n_kmers); t->idx_cap = t->n_kmers; }
    if (t->counts_cap < t->n_kmers) { t->counts = (uint32_t*)realloc(t->counts, sizeof(uint32_t) * t->n_kmers); t->counts_cap = t->n_kmers; }
    for (uint32_t i = 0; i < t->n_kmers; ++i) ++t->bin_start[kmer_hash(seq + i) + 1];
    for (int b = 0; b < NUM_KMERS; ++b) t->bin_start[b + 1] += t->bin_start[b];
    uint32_t* fill = (uint32_t*)calloc(NUM_KMERS, sizeof(uint32_t));
    for (uint32_t i = 0; i < t->n_kmers; ++i) { const unsigned h = kmer_hash(seq + i); t->idx[t->bin_start[h] + fill[h]++] = i; }
    free(fill);
    if (t->n_kmers) memset(t->counts, 0, sizeof(uint32_t) * t->n_kmers);
}

static int kmer_map(kmer_table* t, const uint16_t* query_hashes, int n_query, int max_positions, uint32_t* out)
{
    /* map_query_to_target, kmer_mapper.hpp:120-159 */
    unsigned max_hit_count = 0, num_max_hits = 0;
    size_t first_max_hit_index = 0;
    for (int qi = 0; qi < n_query; ++qi) {
        const unsigned h = query_hashes[qi];
        for (uint32_t e = t->bin_start[h]; e < t->bin_start[h + 1]; ++e) {
            const uint32_t target_index = t->idx[e];
            if (target_index >= (uint32_t)qi) {
                const uint32_t mb = target_index - (uint32_t)qi;
                if (++t->counts[mb] > max_hit_count) {
                    max_hit_count = t->counts[mb]; first_max_hit_index = mb; num_max_hits = 1;
                } else if (t->counts[mb] == max_hit_count) {
                    ++num_max_hits;
                    if (mb < first_max_hit_index) first_max_hit_index = mb;
                }
            }
        }
    }
    int n = 0;
    size_t maxp = (size_t)max_positions;
    if (max_hit_count > 0) {
        out[n++] = (uint32_t)first_max_hit_index++;
        --num_max_hits; --maxp;
        while (maxp > 0 && num_max_hits > 0) {
            if (t->counts[first_max_hit_index] == max_hit_count) { out[n++] = (uint32_t)first_max_hit_index; --num_max_hits; --maxp; }
            ++first_max_hit_index;
        }
    }
    if (t->n_kmers) memset(t->counts, 0, sizeof(uint32_t) * t->n_kmers);    /* reset_mapping_counts :115-118 */
    return n;
}

int oracle_map_query_to_target(const char* query, int query_len, const char* target, int target_len,
        int max_positions, uint32_t* out_positions)
{
    kmer_table t; memset(&t, 0, sizeof(t));
    kmer_table_build(&t, target, target_len);
    const int nq = query_len >= KMER ? query_len - KMER + 1 : 0;            /* compute_kmer_hashes :57-69 */
    uint16_t* qh = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(nq ? nq : 1));
    for (int i = 0; i < nq; ++i) qh[i] = (uint16_t)kmer_hash(query + i);
    const int n = kmer_map(&t, qh, nq, max_positions, out_positions);
    free(qh); free(t.idx); free(t.counts);
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* L3: HaplotypeLikelihoodModel::evaluate + HaplotypeLikelihoodArray::populate                      */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
    const oct_phmm_config* cfg;
    const oct_phmm_reads* reads;
    const oct_phmm_haplotypes* haps;
    const oct_phmm_positions* positions;
    int band, score_bits;
    uint32_t n_regions;
    const uint32_t* reg_row_off; const uint32_t* reg_hap_off;
    const uint8_t* reg_has_flank; const oct_phmm_flank_state* reg_flank;
    uint32_t one_row_off[2], one_hap_off[2]; uint8_t one_has_flank; oct_phmm_flank_state one_flank;
    uint32_t* hap_region;         /* [n_haps] */
    uint64_t* hap_out_off;        /* [n_haps+1] rows */
    uint64_t* hap_pair_off;       /* [n_haps+1] reads */
    uint16_t* read_hashes;        /* concatenated per read, offset = reads->offsets[r] */
    double* out;
    const oct_phmm_alignments* aln;   /* non-NULL: HaplotypeLikelihoodModel::align mode (oracle_align_batch) */
    int aln_error; uint32_t aln_needed_ops;   /* 1 = HMMOverflow, 2 = cigar capacity */
    /* per-haplotype error records (ShortHaplotypeError) */
    uint8_t* hap_err; uint32_t* hap_err_read; uint32_t* hap_err_ext;
    volatile uint32_t next_hap; uint32_t n_items; uint32_t* hap_item_off;   /* work items = (haplotype, ROW_CHUNK rows) */
    pthread_mutex_t mu;
    oct_phmm_stats stats;
} pop_ctx;

static inline uint32_t row_first_read(const oct_phmm_reads* r, uint32_t row) { return r->row_offsets ? r->row_offsets[row] : row; }

/* num_out_of_range_bases, haplotype_likelihood_model.cpp:187-201 */
static int num_out_of_range_bases(uint64_t mapping_position, uint64_t T, uint64_t Lh, unsigned required_pad)
{
    if (mapping_position < required_pad) return (int)(required_pad - mapping_position);
    const uint64_t mapping_end = mapping_position + T + required_pad;
    if (mapping_end > Lh) return (int)Lh - (int)mapping_end;
    return 0;
}

typedef struct {
    oct_phmm_stats st; aln_scratch scr; kmer_table kt; uint32_t mapped[64];
} worker_state;

/* HaplotypeLikelihoodModel::evaluate(read, first, last), haplotype_likelihood_model.cpp:261-304, with max_score :211-259.
 * Returns 0 ok, 1 ShortHaplotypeError (ext filled). */
static int evaluate_read(pop_ctx* c, worker_state* w, uint32_t h, uint32_t r, uint32_t lhs, uint32_t rhs,
                         const uint32_t* pos, int npos, double* result, uint32_t* ext)
{
    const oct_phmm_reads* R = c->reads; const oct_phmm_haplotypes* H = c->haps;
    const uint32_t ro = R->offsets[r], T = R->offsets[r + 1] - ro;
    const uint32_t ho = H->offsets[h], Lh = H->offsets[h + 1] - ho;
    const char* target = R->bases + ro; const uint8_t* quals = R->qualities + ro;
    const char* truth = H->bases + ho;
    const int is_forward = !R->reverse_strand[r];                           /* :269 */
    const char* mask = (is_forward ? H->snv_mask_fwd : H->snv_mask_rev) + ho;   /* :270-275 */
    const int8_t* prior = (is_forward ? H->snv_prior_fwd : H->snv_prior_rev) + ho;
    const int8_t* go = H->gap_open + ho; const int8_t* ge = H->gap_extend + ho;
    const unsigned pad = (unsigned)c->band;
    const uint64_t original = (uint64_t)(R->ref_begin[r] - H->ref_begin[h]); /* begin_distance, cast to size_t :220 */
    double best = OCT_PHMM_LOWEST;
    int orig_mapped = 0, has_in_range = 0, kind = 0;
#define EVAL(p) do { \
        double v_ = 0; int pen_ = 0; \
        ++w->st.n_candidates; \
        if (oracle_try_naive_evaluate(truth, (int)Lh, target, (int)T, quals, (uint32_t)(p), go, ge, mask, prior, lhs, rhs, &pen_)) { \
            ++w->st.n_fast_path; v_ = phred_to_ln((double)pen_); \
        } else { \
            v_ = simd_evaluate(truth, (int)Lh, target, (int)T, quals, (uint32_t)(p), c->band, c->score_bits, go, ge, mask, prior, \
                               lhs, rhs, c->cfg->nuc_prior, &w->scr, &kind); \
            if (kind == 1) { ++w->st.n_dp_score_only; w->st.band_cells += 2ull * pad * (T + pad); } \
            else if (kind == 2) { ++w->st.n_dp_traceback; w->st.band_cells += 2ull * pad * (T + pad); } \
        } \
        if (v_ > best) best = v_; \
    } while (0)
    for (int j = 0; j < npos; ++j) {                                        /* :223-232 */
        const uint64_t p = pos[j];
        if (p == original) orig_mapped = 1;
        if (num_out_of_range_bases(p, T, Lh, pad) == 0) { has_in_range = 1; EVAL(p); }
    }
    if (!orig_mapped && num_out_of_range_bases(original, T, Lh, pad) == 0) { has_in_range = 1; EVAL(original); } /* :233-237 */
    if (!has_in_range) {                                                    /* :238-256 */
        const int min_shift = num_out_of_range_bases(original, T, Lh, pad);
        uint64_t final_pos = original;
        if (min_shift > 0) {
            final_pos += (uint64_t)min_shift;
            if (num_out_of_range_bases(final_pos, T, Lh, pad) != 0) { *ext = (unsigned)min_shift; return 1; }
        } else {
            const unsigned min_left_shift = (unsigned)(-min_shift);
            if (original >= min_left_shift) final_pos -= min_left_shift;
            else { *ext = (uint32_t)(min_left_shift - original); return 1; }
        }
        best = OCT_PHMM_LOWEST;
        EVAL(final_pos);                                                    /* :255 assigns, not max */
    }
#undef EVAL
    double res;
    if (c->cfg->use_mapping_quality) {                                      /* :285-300 */
        int mq = R->mapping_quality[r];
        if (c->cfg->mapping_quality_cap_trigger >= 0 && c->cfg->mapping_quality_cap_trigger < c->cfg->mapping_quality_cap
            && mq >= c->cfg->mapping_quality_cap_trigger) mq = c->cfg->mapping_quality_cap; /* :50-52,293-295 (uint8 semantics) */
        mq &= 0xFF;
        const double ln_missmapped = -LN10_DIV_10 * mq;
        const double ln_mapped = log(1.0 - exp(ln_missmapped));
        const double a = ln_mapped + best, b = ln_missmapped;               /* maths::log_sum_exp, utils/maths.hpp:294-298 */
        const double lo = a < b ? a : b, hi = a < b ? b : a;                /* std::minmax(a,b): (b<a)?(b,a):(a,b) */
        res = hi + log1p(exp(lo - hi));
        res = res > -1e-15 ? 0.0 : res;
    } else {
        res = best > -1e-15 ? 0.0 : best;                                   /* :302 */
    }
    *result = res;
    return 0;
}

/* make_cigar, pair_hmm.hpp:152-188: run-length classes of the alignment columns. Returns the number of operations (they are
 * written while they fit in cap). */
static uint32_t make_cigar(const char* a1, const char* a2, uint32_t* ops, uint32_t cap)
{
    size_t n = strlen(a1);                                 /* last non-zero char of align1 (:157) */
    uint32_t k = 0; size_t i = 0;
    while (i < n) {
        uint32_t op; size_t j = i;
        if (a1[i] == a2[i]) { while (j < n && a1[j] == a2[j]) ++j; op = OCT_PHMM_CIGAR_EQ; }                   /* std::mismatch :162-166 */
        else if (a1[i] == '-') { while (j < n && a1[j] == '-') ++j; op = OCT_PHMM_CIGAR_INS; }                 /* :168-173 */
        else if (a2[i] == '-') { while (j < n && a2[j] == '-') ++j; op = OCT_PHMM_CIGAR_DEL; }                 /* :174-179 */
        else { ++j; while (j < n && a1[j] != a2[j] && a1[j] != '-' && a2[j] != '-') ++j; op = OCT_PHMM_CIGAR_X; }   /* :180-185 */
        if (k < cap) ops[k] = (uint32_t)(j - i) << 4 | op;
        ++k; i = j;
    }
    return k;
}

typedef struct { double likelihood; uint32_t target_offset; uint32_t n_ops; int overflow; } one_alignment;

/* hmm::align, pair_hmm.hpp:861-872 = try_naive_align :321-341, else simd_align :788-823 */
static one_alignment hmm_align(pop_ctx* c, worker_state* w, const char* truth, int Lh, const char* target, int T, const uint8_t* quals,
                               uint32_t target_offset, const int8_t* go, const int8_t* ge, const char* mask, const int8_t* prior,
                               uint32_t lhs_flank, uint32_t rhs_flank, uint32_t* ops, uint32_t cap)
{
    one_alignment r; memset(&r, 0, sizeof(r));
    if (memcmp(target, truth + target_offset, (size_t)T) == 0) {            /* try_naive_align */
        r.likelihood = 0; r.target_offset = target_offset; r.n_ops = 1;
        if (cap) ops[0] = (uint32_t)T << 4 | OCT_PHMM_CIGAR_EQ;
        return r;
    }
    const int pad = c->band, L = T + 2 * pad - 1;
    int alignment_offset = (int)target_offset - pad; if (alignment_offset < 0) alignment_offset = 0;   /* :799 */
    if (alignment_offset + L > Lh) { r.likelihood = OCT_PHMM_LOWEST; return r; }                      /* :800-805 */
    const size_t need = (size_t)(2 * (T + pad)) + 1;
    aln_scratch* scr = &w->scr;
    if (scr->cap < need) { scr->a1 = (char*)realloc(scr->a1, need); scr->a2 = (char*)realloc(scr->a2, need); scr->cap = need; }
    memset(scr->a1, 0, need); memset(scr->a2, 0, need);
    int first_pos = 0, st = 0;
    const int8_t* q8 = (const int8_t*)quals;
    int score = g_align(c->band, c->score_bits, truth + alignment_offset, target, q8, L, T, mask + alignment_offset, prior + alignment_offset,
                        go + alignment_offset, ge + alignment_offset, 0, c->cfg->nuc_prior, 1, &first_pos, scr->a1, scr->a2, &st);
    if (first_pos == -1) { r.overflow = 1; return r; }                                                /* throw HMMOverflow :811-813 */
    /* discount_flank_score :646-673 */
    const int adjusted = (uint64_t)target_offset < ((uint64_t)lhs_flank + (uint64_t)pad)
        || ((uint64_t)target_offset + (uint64_t)T + (uint64_t)pad) > ((uint64_t)Lh - (uint64_t)rhs_flank);
    if (adjusted) {
        int lhs = (int)lhs_flank;
        if (lhs < alignment_offset) lhs = 0; else { lhs -= alignment_offset; if (lhs < 0) lhs = 0; }
        int rhs = (int)rhs_flank;
        if (alignment_offset + L < Lh - rhs) rhs = 0; else { rhs += alignment_offset + L; rhs -= Lh; if (rhs < 0) rhs = 0; }
        int target_mask_size = 0;
        int flank_score = g_flank(c->band, c->score_bits, L, lhs, rhs, target, q8, mask + alignment_offset, prior + alignment_offset,
                                  go + alignment_offset, ge + alignment_offset, c->cfg->nuc_prior, first_pos, scr->a1, scr->a2,
                                  &target_mask_size, &st);
        if (T - target_mask_size < 2) flank_score = 0;
        if (flank_score <= score) score -= flank_score; else score += flank_score;
    }
    r.target_offset = target_offset - (uint32_t)pad + (uint32_t)first_pos;                            /* :817 */
    r.likelihood = -LN10_DIV_10 * (double)score;
    r.n_ops = make_cigar(scr->a1, scr->a2, ops, cap);
    return r;
}

/* HaplotypeLikelihoodModel::align(read, first, last), haplotype_likelihood_model.cpp:397-431 with compute_optimal_alignment :335-395.
 * Returns 0 ok, 1 ShortHaplotypeError (ext filled), 2 HMMOverflow. */
static int align_read(pop_ctx* c, worker_state* w, uint32_t h, uint32_t r, uint32_t lhs, uint32_t rhs, const uint32_t* pos, int npos,
                      uint64_t pair, uint32_t* ext)
{
    const oct_phmm_reads* R = c->reads; const oct_phmm_haplotypes* H = c->haps;
    const uint32_t ro = R->offsets[r], T = R->offsets[r + 1] - ro;
    const uint32_t ho = H->offsets[h], Lh = H->offsets[h + 1] - ho;
    const char* target = R->bases + ro; const uint8_t* quals = R->qualities + ro; const char* truth = H->bases + ho;
    const int is_forward = !R->reverse_strand[r];
    const char* mask = (is_forward ? H->snv_mask_fwd : H->snv_mask_rev) + ho;
    const int8_t* prior = (is_forward ? H->snv_prior_fwd : H->snv_prior_rev) + ho;
    const int8_t* go = H->gap_open + ho; const int8_t* ge = H->gap_extend + ho;
    const unsigned pad = (unsigned)c->band;
    const uint64_t original = (uint64_t)(R->ref_begin[r] - H->ref_begin[h]);
    const uint32_t cap = c->aln->max_cigar_ops;
    uint32_t* best_ops = c->aln->cigar + pair * cap;
    uint32_t* tmp = (uint32_t*)malloc(((size_t)cap + 1) * sizeof(uint32_t));
    one_alignment best; memset(&best, 0, sizeof(best)); best.likelihood = OCT_PHMM_LOWEST;
    int orig_mapped = 0, has_in_range = 0, rc = 0;
#define ALIGN_AT(p, accept_equal) do { \
        one_alignment a_ = hmm_align(c, w, truth, (int)Lh, target, (int)T, quals, (uint32_t)(p), go, ge, mask, prior, lhs, rhs, tmp, cap); \
        if (a_.overflow) { rc = 2; goto done; } \
        if ((accept_equal) ? a_.likelihood >= best.likelihood : a_.likelihood > best.likelihood) { \
            best = a_; memcpy(best_ops, tmp, (size_t)(a_.n_ops < cap ? a_.n_ops : cap) * sizeof(uint32_t)); } \
    } while (0)
    for (int j = 0; j < npos; ++j) {                                        /* :349-362 */
        const uint64_t p = pos[j];
        if (p == original) orig_mapped = 1;
        if (num_out_of_range_bases(p, T, Lh, pad) == 0) { has_in_range = 1; ALIGN_AT(p, 0); }
    }
    if (!orig_mapped && num_out_of_range_bases(original, T, Lh, pad) == 0) { has_in_range = 1; ALIGN_AT(original, 1); }   /* :363-371 */
    if (!has_in_range) {                                                    /* :372-392 */
        const int min_shift = num_out_of_range_bases(original, T, Lh, pad);
        uint64_t final_pos = original;
        if (min_shift > 0) {
            final_pos += (uint64_t)min_shift;
            if (num_out_of_range_bases(final_pos, T, Lh, pad) != 0) { *ext = (unsigned)min_shift; rc = 1; goto done; }
        } else {
            const unsigned min_left_shift = (unsigned)(-min_shift);
            if (original >= min_left_shift) final_pos -= min_left_shift;
            else { *ext = (uint32_t)(min_left_shift - original); rc = 1; goto done; }
        }
        best.likelihood = OCT_PHMM_LOWEST;
        ALIGN_AT(final_pos, 1);                                             /* assigned unconditionally :388-391 */
    }
#undef ALIGN_AT
    {
        double res = best.likelihood;
        if (c->cfg->use_mapping_quality) {                                  /* :416-427 */
            int mq = R->mapping_quality[r];
            if (c->cfg->mapping_quality_cap_trigger >= 0 && c->cfg->mapping_quality_cap_trigger < c->cfg->mapping_quality_cap
                && mq >= c->cfg->mapping_quality_cap_trigger) mq = c->cfg->mapping_quality_cap;
            mq &= 0xFF;
            const double ln_missmapped = -LN10_DIV_10 * mq, ln_mapped = log(1.0 - exp(ln_missmapped));
            const double a = ln_mapped + res, b = ln_missmapped;
            const double lo = a < b ? a : b, hi = a < b ? b : a;
            res = hi + log1p(exp(lo - hi));
        }
        res = res > -1e-15 ? 0.0 : res;
        c->aln->likelihood[pair] = res; c->aln->mapping_position[pair] = best.target_offset; c->aln->n_cigar_ops[pair] = best.n_ops;
        if (best.n_ops > cap) {
            pthread_mutex_lock(&c->mu);
            if (c->aln_error == 0) c->aln_error = 3;
            if (best.n_ops > c->aln_needed_ops) c->aln_needed_ops = best.n_ops;
            pthread_mutex_unlock(&c->mu);
        }
    }
done:
    free(tmp);
    return rc;
}

#define ROW_CHUNK 128   /* rows per work item: (haplotype, row chunk) items keep every host core busy even with few haplotypes */

static void populate_haplotype(pop_ctx* c, worker_state* w, uint32_t h, uint32_t chunk)
{
    /* body of populate_haplotype, haplotype_likelihood_array.cpp:132-160 (TemplateMap) / :78-99 (ReadMap) */
    const uint32_t g = c->hap_region[h];
    const uint32_t reg_row0 = c->reg_row_off[g], reg_row1 = c->reg_row_off[g + 1];
    const uint32_t row0 = reg_row0 + chunk * ROW_CHUNK, row1 = row0 + ROW_CHUNK < reg_row1 ? row0 + ROW_CHUNK : reg_row1;
    if (row0 >= reg_row1) return;
    const uint32_t first_read = row_first_read(c->reads, reg_row0);
    const int has_flank = c->cfg->use_flank_state && c->reg_has_flank && c->reg_has_flank[g];
    const uint32_t lhs = has_flank ? c->reg_flank[g].lhs_flank : 0, rhs = has_flank ? c->reg_flank[g].rhs_flank : 0; /* model.cpp:276-282 */
    const uint32_t ho = c->haps->offsets[h], Lh = c->haps->offsets[h + 1] - ho;
    if (!c->positions) kmer_table_build(&w->kt, c->haps->bases + ho, (int)Lh);
    for (uint32_t row = row0; row < row1; ++row) {
        const uint32_t r0 = row_first_read(c->reads, row), r1 = row_first_read(c->reads, row + 1);
        double acc = 0;                                                     /* std::inner_product from 0, model.cpp:313-320 */
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t* pos; int npos;
            if (c->positions) {
                const uint64_t e = c->hap_pair_off[h] + (r - first_read);
                pos = c->positions->positions + c->positions->offsets[e];
                npos = (int)(c->positions->offsets[e + 1] - c->positions->offsets[e]);
            } else {
                const uint32_t ro = c->reads->offsets[r], T = c->reads->offsets[r + 1] - ro;
                npos = kmer_map(&w->kt, c->read_hashes + ro, T >= KMER ? (int)(T - KMER + 1) : 0,
                                c->cfg->max_mapping_positions > 64 ? 64 : c->cfg->max_mapping_positions, w->mapped);
                pos = w->mapped;
            }
            double v = 0; uint32_t ext = 0;
            ++w->st.n_pairs;
            int erc;
            if (c->aln) {
                erc = align_read(c, w, h, r, lhs, rhs, pos, npos, c->hap_pair_off[h] + (r - first_read), &ext);
                if (erc == 2) { pthread_mutex_lock(&c->mu); c->aln_error = 2; pthread_mutex_unlock(&c->mu); erc = 0; }
            } else erc = evaluate_read(c, w, h, r, lhs, rhs, pos, npos, &v, &ext);
            if (erc) {
                pthread_mutex_lock(&c->mu);                                 /* keep the first failing read of this haplotype in serial order */
                if (!c->hap_err[h] || r < c->hap_err_read[h]) { c->hap_err[h] = 1; c->hap_err_read[h] = r; c->hap_err_ext[h] = ext; }
                pthread_mutex_unlock(&c->mu);
                return;                                                     /* the exception aborts populate */
            }
            acc = acc + v;
        }
        if (!c->aln) c->out[c->hap_out_off[h] + (row - reg_row0)] = acc;
    }
}

static void* pop_worker(void* arg)
{
    pop_ctx* c = (pop_ctx*)arg;
    worker_state w; memset(&w, 0, sizeof(w));
    for (;;) {
        const uint32_t item = __sync_fetch_and_add(&c->next_hap, 1);
        if (item >= c->n_items) break;
        /* item -> (haplotype, chunk) through the per-haplotype item offsets */
        uint32_t lo = 0, hi = c->haps->n_haps;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (c->hap_item_off[mid] <= item) lo = mid; else hi = mid; }
        populate_haplotype(c, &w, lo, item - c->hap_item_off[lo]);
    }
    pthread_mutex_lock(&c->mu);
    c->stats.n_pairs += w.st.n_pairs; c->stats.n_candidates += w.st.n_candidates; c->stats.n_fast_path += w.st.n_fast_path;
    c->stats.n_dp_score_only += w.st.n_dp_score_only; c->stats.n_dp_traceback += w.st.n_dp_traceback; c->stats.band_cells += w.st.band_cells;
    pthread_mutex_unlock(&c->mu);
    free(w.scr.a1); free(w.scr.a2); free(w.kt.idx); free(w.kt.counts);
    return NULL;
}

static int set_status(oct_phmm_status* st, int code, const char* msg)
{
    if (st) { memset(st, 0, sizeof(*st)); st->code = code; if (msg) snprintf(st->message, sizeof(st->message), "%s", msg); }
    return code;
}

static int populate_impl(const oct_phmm_config* cfg,
        const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
        const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
        const oct_phmm_positions* positions,
        double* out, const oct_phmm_alignments* aln, oct_phmm_status* status, oct_phmm_stats* stats, int n_threads)
{
    if (!cfg || !reads || !haps || (!out && !aln)) return set_status(status, OCT_PHMM_EINVAL, "null argument");
    if (aln && reads->row_offsets) return set_status(status, OCT_PHMM_EINVAL, "alignments are per read: row_offsets must be NULL");
    pop_ctx c; memset(&c, 0, sizeof(c));
    c.cfg = cfg; c.reads = reads; c.haps = haps; c.positions = positions; c.out = out; c.aln = aln;
    c.band = oracle_band_size(cfg->max_indel_error);
    if (c.band < 0) return set_status(status, OCT_PHMM_EBAND, "requested band size is too large");
    c.score_bits = cfg->use_int_scores ? 32 : 16;
    const uint32_t n_rows = reads->row_offsets ? reads->n_rows : reads->n_reads;
    if (regions) {
        c.n_regions = regions->n_regions; c.reg_row_off = regions->row_offsets; c.reg_hap_off = regions->hap_offsets;
        c.reg_has_flank = regions->has_flank; c.reg_flank = regions->flank;
    } else {
        c.n_regions = 1; c.one_row_off[0] = 0; c.one_row_off[1] = n_rows; c.one_hap_off[0] = 0; c.one_hap_off[1] = haps->n_haps;
        c.one_has_flank = flank != NULL; if (flank) c.one_flank = *flank;
        c.reg_row_off = c.one_row_off; c.reg_hap_off = c.one_hap_off; c.reg_has_flank = &c.one_has_flank; c.reg_flank = &c.one_flank;
    }
    const uint32_t nh = haps->n_haps;
    c.hap_region = (uint32_t*)calloc(nh + 1, sizeof(uint32_t));
    c.hap_out_off = (uint64_t*)calloc(nh + 1, sizeof(uint64_t));
    c.hap_pair_off = (uint64_t*)calloc(nh + 1, sizeof(uint64_t));
    c.hap_err = (uint8_t*)calloc(nh + 1, 1); c.hap_err_read = (uint32_t*)calloc(nh + 1, sizeof(uint32_t)); c.hap_err_ext = (uint32_t*)calloc(nh + 1, sizeof(uint32_t));
    for (uint32_t g = 0; g < c.n_regions; ++g) {
        const uint32_t rows = c.reg_row_off[g + 1] - c.reg_row_off[g];
        const uint32_t nreads = row_first_read(reads, c.reg_row_off[g + 1]) - row_first_read(reads, c.reg_row_off[g]);
        for (uint32_t h = c.reg_hap_off[g]; h < c.reg_hap_off[g + 1]; ++h) {
            c.hap_region[h] = g; c.hap_out_off[h + 1] = c.hap_out_off[h] + rows; c.hap_pair_off[h + 1] = c.hap_pair_off[h] + nreads;
        }
    }
    if (!positions) {
        /* compute_kmer_hashes once per read, haplotype_likelihood_array.cpp:118-131 */
        const uint32_t total = reads->offsets[reads->n_reads];
        c.read_hashes = (uint16_t*)malloc(sizeof(uint16_t) * (total ? total : 1));
        for (uint32_t r = 0; r < reads->n_reads; ++r) {
            const uint32_t ro = reads->offsets[r], T = reads->offsets[r + 1] - ro;
            for (uint32_t i = 0; i + KMER <= T; ++i) c.read_hashes[ro + i] = (uint16_t)kmer_hash(reads->bases + ro + i);
        }
    }
    c.hap_item_off = (uint32_t*)calloc(nh + 2, sizeof(uint32_t));
    for (uint32_t h = 0; h < nh; ++h) {
        const uint32_t g = c.hap_region[h], rows = c.reg_row_off[g + 1] - c.reg_row_off[g];
        c.hap_item_off[h + 1] = c.hap_item_off[h] + (rows + ROW_CHUNK - 1) / ROW_CHUNK;
    }
    c.n_items = c.hap_item_off[nh];
    pthread_mutex_init(&c.mu, NULL);
    if (n_threads < 1) n_threads = 1;
    if (n_threads == 1) pop_worker(&c);
    else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, pop_worker, &c);
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&c.mu);
    int code = OCT_PHMM_OK;
    set_status(status, OCT_PHMM_OK, NULL);
    for (uint32_t h = 0; h < nh; ++h) if (c.hap_err[h]) {
        code = set_status(status, OCT_PHMM_ESHORT_HAPLOTYPE, "Haplotype is too short for alignment");
        if (status) { status->hap_index = h; status->read_index = c.hap_err_read[h]; status->required_extension = c.hap_err_ext[h]; }
        break;
    }
    if (code == OCT_PHMM_OK && c.aln_error == 2) code = set_status(status, OCT_PHMM_EOVERFLOW, "Pair HMM alignment overflowed");
    if (code == OCT_PHMM_OK && c.aln_error == 3) {
        code = set_status(status, OCT_PHMM_EINVAL, "max_cigar_ops too small");
        if (status) status->required_extension = c.aln_needed_ops;
    }
    if (stats) *stats = c.stats;
    free(c.hap_item_off); free(c.hap_region); free(c.hap_out_off); free(c.hap_pair_off); free(c.hap_err); free(c.hap_err_read); free(c.hap_err_ext); free(c.read_hashes);
    return code;
}

int oracle_populate(const oct_phmm_config* cfg,
        const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
        const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
        const oct_phmm_positions* positions,
        double* out, oct_phmm_status* status, oct_phmm_stats* stats, int n_threads)
{
    if (!out) return set_status(status, OCT_PHMM_EINVAL, "null argument");
    return populate_impl(cfg, reads, haps, regions, flank, positions, out, NULL, status, stats, n_threads);
}

int oracle_align_batch(const oct_phmm_config* cfg,
        const oct_phmm_reads* reads, const oct_phmm_haplotypes* haps,
        const oct_phmm_regions* regions, const oct_phmm_flank_state* flank,
        const oct_phmm_positions* positions,
        const oct_phmm_alignments* out, oct_phmm_status* status, int n_threads)
{
    if (!out || !out->mapping_position || !out->likelihood || !out->n_cigar_ops || !out->cigar)
        return set_status(status, OCT_PHMM_EINVAL, "null argument");
    return populate_impl(cfg, reads, haps, regions, flank, positions, NULL, out, status, NULL, n_threads);
}

/* ------------------------------------------------------------------------------------------------ */
/* raw L1 timing loop (CPU baseline leg of bench.py)                                                */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    int band, score_bits; uint32_t n;
    const char* truth; const uint32_t* toff; const char* target; const uint8_t* quals; const uint32_t* roff;
    const int8_t* go; const int8_t* ge; const char* mask; const int8_t* prior; int nuc_prior, traceback, reps;
    int tid, nthreads; int64_t checksum;
} time_ctx;

static void* time_worker(void* arg)
{
    time_ctx* c = (time_ctx*)arg;
    int64_t sum = 0;
    char* a1 = NULL; char* a2 = NULL; size_t cap = 0;
    for (int rep = 0; rep < c->reps; ++rep) {
        for (uint32_t i = (uint32_t)c->tid; i < c->n; i += (uint32_t)c->nthreads) {
            const uint32_t to = c->toff[i], L = c->toff[i + 1] - to, ro = c->roff[i], T = c->roff[i + 1] - ro;
            int st = 0, fp = 0;
            if (c->traceback) {
                const size_t need = 2 * ((size_t)T + (size_t)c->band) + 1;
                if (cap < need) { a1 = (char*)realloc(a1, need); a2 = (char*)realloc(a2, need); cap = need; }
                memset(a1, 0, need); memset(a2, 0, need);
            }
            sum += g_align(c->band, c->score_bits, c->truth + to, c->target + ro, (const int8_t*)c->quals + ro, (int)L, (int)T,
                           c->mask ? c->mask + to : NULL, c->prior ? c->prior + to : NULL, c->go + to, c->ge ? c->ge + to : NULL, 1,
                           c->nuc_prior, c->traceback, &fp, a1, a2, &st);
        }
    }
    free(a1); free(a2);
    c->checksum = sum;
    return NULL;
}

double oracle_time_align_windows(int band, int score_bits, uint32_t n,
        const char* truth, const uint32_t* truth_offsets,
        const char* target, const uint8_t* quals, const uint32_t* target_offsets,
        const int8_t* gap_open, const int8_t* gap_extend, const char* snv_mask, const int8_t* snv_prior,
        int nuc_prior, int traceback, int reps, int n_threads, int64_t* checksum)
{
    if (n_threads < 1) n_threads = 1;
    time_ctx* cs = (time_ctx*)calloc((size_t)n_threads, sizeof(time_ctx));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; ++t) {
        time_ctx c = {band, score_bits, n, truth, truth_offsets, target, quals, target_offsets, gap_open, gap_extend, snv_mask, snv_prior,
                      nuc_prior, traceback, reps, t, n_threads, 0};
        cs[t] = c;
        pthread_create(&th[t], NULL, time_worker, &cs[t]);
    }
    int64_t sum = 0;
    for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); sum += cs[t].checksum; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (checksum) *checksum = sum;
    free(cs); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Genotype read-out: ConstantMixtureGenotypeLikelihoodModel (constant_mixture_genotype_likelihood_model.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */
static const double LN2 = 0.693147180559945309417232121458176568075500134360255254120;   /* :46-63 */
static const double LN3 = 1.098612288668109691395245236922525704647490557822749451734;
static const double LN4 = 1.386294361119890618834464242916353136151000268720510508241;

static double lse_2(double a, double b)              /* utils/maths.hpp:294-298 */
{
    const double lo = b < a ? b : a, hi = b < a ? a : b;
    return hi + log1p(exp(lo - hi));
}
static double lse_3(double a, double b, double c)    /* utils/maths.hpp:302-306 */
{
    double mx = a; if (mx < b) mx = b; if (mx < c) mx = c;
    return mx + log(exp(a - mx) + exp(b - mx) + exp(c - mx));
}
static double lse_n(const double* x, uint32_t n)     /* utils/maths.hpp:310-330 */
{
    double mx = x[0], s = 0;
    for (uint32_t i = 1; i < n; ++i) if (mx < x[i]) mx = x[i];
    for (uint32_t i = 0; i < n; ++i) s = s + exp(x[i] - mx);
    return mx + log(s);
}

int oracle_genotype_likelihoods(const double* lik, const uint64_t* hap_out_off, uint32_t n_genotypes, uint32_t ploidy,
        const uint32_t* hap_indices, uint32_t row_begin, uint32_t row_end, double* out)
{
    if (ploidy > 16) return OCT_PHMM_EUNSUPPORTED;
    for (uint32_t gi = 0; gi < n_genotypes; ++gi) {
        const uint32_t* g = hap_indices + (size_t)gi * ploidy;
        const double* L[16];
        for (uint32_t j = 0; j < ploidy; ++j) L[j] = lik + hap_out_off[g[j]];
        double result = 0;
        for (uint32_t r = row_begin; r < row_end; ++r) {
            double v;
            switch (ploidy) {
                case 0: v = 0; break;
                case 1: v = L[0][r]; break;                                                     /* evaluate_haploid :164-169 */
                case 2:                                                                         /* evaluate_diploid :171-188 */
                    v = g[0] == g[1] ? L[0][r] : lse_2(L[0][r], L[1][r]) - LN2; break;
                case 3:                                                                         /* evaluate_triploid :190-226 */
                    if (g[0] == g[1]) v = g[1] == g[2] ? L[0][r] : lse_2(LN2 + L[0][r], L[2][r]) - LN3;
                    else if (g[1] == g[2]) v = lse_2(L[0][r], LN2 + L[1][r]) - LN3;
                    else v = lse_3(L[0][r], L[1][r], L[2][r]) - LN3;
                    break;
                case 4:                                                                         /* evaluate_tetraploid :228-313 */
                    if (g[0] == g[1]) {
                        if (g[1] == g[2]) v = g[2] == g[3] ? L[0][r] : lse_2(LN3 + L[0][r], L[3][r]) - LN4;
                        else if (g[2] == g[3]) v = lse_2(L[0][r], L[2][r]) - LN2;
                        else v = lse_3(LN2 + L[0][r], L[2][r], L[3][r]) - LN4;
                    } else if (g[1] == g[2]) {
                        if (g[2] == g[3]) v = lse_2(L[0][r], LN3 + L[1][r]) - LN4;
                        else v = lse_3(L[0][r], LN2 + L[1][r], L[3][r]) - LN4;
                    } else if (g[2] == g[3]) {
                        v = lse_3(L[0][r], L[1][r], LN2 + L[2][r]) - LN4;
                    } else {
                        const double x[4] = {L[0][r], L[1][r], L[2][r], L[3][r]};
                        v = lse_n(x, 4) - LN4;
                    }
                    break;
                default: {                                                                      /* evaluate_polyploid :315-329 */
                    double x[16];
                    for (uint32_t j = 0; j < ploidy; ++j) x[j] = L[j][r];
                    v = lse_n(x, ploidy) - log((double)ploidy);
                }
            }
            result += v;
        }
        out[gi] = result;
    }
    return OCT_PHMM_OK;
}
