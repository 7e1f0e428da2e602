/* CPU ORACLE (test infrastructure only) for the per-haplotype penalty vectors, SURVEY.md 8f-3.
 *
 * Restates, citing the reference:
 *   tandem::extract_exact_tandem_repeats          lib/tandem/tandem.hpp:497-514  (naive path :392-493 for max_period <= 3,
 *                                                 Lempel-Ziv / Main / Kolpakov-Kucherov path :183-390 + tandem.cpp:35-110 otherwise)
 *   RepeatBasedIndelErrorModel::do_set_penalties  core/models/error/repeat_based_indel_error_model.cpp:67-83
 *   BasicRepeatBasedIndelErrorModel lookups       core/models/error/basic_repeat_based_indel_error_model.cpp:23-103
 *   BasicRepeatBasedSNVErrorModel::do_evaluate    core/models/error/repeat_based_snv_error_model.cpp:144-179 (count_runs :48-87)
 *
 * PINNED: the repeat extraction is checked list-for-list against the reference's own tandem library compiled in place
 * (oracle/_ref, ref_tandem_repeats) - including that library's quirks (runs touching the end of the string, period == max_period,
 * the duplicated suffix-array entry), which the error models inherit. The two model classes are pinned too: oracle/_ref compiles the reference's BasicRepeatBasedIndelErrorModel and BasicRepeatBasedSNVErrorModel
 * in place on a stand-in Haplotype (oracle/ref_errmodel_bridge.cpp); all six vectors are compared for equality, including the order
 * std::sort leaves equal-length repeats in (restated below).
 */
#include "error_model_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t pos, length, period; } rep_t;
typedef struct { rep_t* v; uint32_t n, cap; } vec_t;

static void vec_push(vec_t* a, rep_t r)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->v = (rep_t*)realloc(a->v, a->cap * sizeof(rep_t)); }
    a->v[a->n++] = r;
}
static void vec_insert_front(vec_t* a, const rep_t* src, uint32_t k)
{
    while (a->n + k > a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->v = (rep_t*)realloc(a->v, a->cap * sizeof(rep_t)); }
    memmove(a->v + k, a->v, a->n * sizeof(rep_t));
    memcpy(a->v, src, k * sizeof(rep_t));
    a->n += k;
}

/* ---- naive extraction, tandem.hpp:392-493 ---------------------------------------------------------------------- */
static void extract_homopolymers(const char* s, uint32_t n, vec_t* out)                   /* :392-409 */
{
    uint32_t cur = 0;
    while (cur < n) {
        uint32_t it = cur;                                         /* std::adjacent_find */
        while (it + 1 < n && s[it] != s[it + 1]) ++it;
        if (it + 1 >= n) break;
        uint32_t it2 = it + 1;
        while (it2 < n && s[it2] == s[it]) ++it2;
        vec_push(out, (rep_t){it, it2 - it, 1});
        cur = it2;
    }
}
static uint32_t adjacent_find_ne(const char* s, uint32_t from, uint32_t n)               /* adjacent_find(..., not_equal_to) */
{
    uint32_t it = from;
    if (it >= n) return n;
    while (it + 1 < n && s[it] == s[it + 1]) ++it;
    return it + 1 >= n ? n : it;
}
static void extract_period_n(const char* s, uint32_t n, uint32_t N, vec_t* out)          /* :411-436 */
{
    if (n < 2 * N) return;
    uint32_t it1 = adjacent_find_ne(s, 0, n);
    if (it1 == n) return;
    for (uint32_t it2 = it1 + N; it2 < n;) {
        uint32_t a = it2, b = it1;                                 /* std::mismatch(it2, last, it1) */
        while (a < n && s[a] == s[b]) { ++a; ++b; }
        if (b >= it2) { vec_push(out, (rep_t){it1, a - it1, N}); it1 = b; }
        else ++it1;
        it1 = adjacent_find_ne(s, it1, n);
        if (it1 == n) break;
        it2 = it1 + N;
    }
}
static void merge_by_pos(vec_t* dst, const vec_t* src)                                   /* detail::merge :456-466 (inplace_merge is stable) */
{
    vec_t r = {0};
    uint32_t i = 0, j = 0;
    while (i < dst->n || j < src->n) {
        if (j >= src->n || (i < dst->n && !(src->v[j].pos < dst->v[i].pos))) vec_push(&r, dst->v[i++]);
        else vec_push(&r, src->v[j++]);
    }
    free(dst->v); *dst = r;
}
static void extract_naive(const char* s, uint32_t n, uint32_t min_period, uint32_t max_period, vec_t* out)   /* :468-493 */
{
    if (min_period == max_period) {
        if (min_period == 1) extract_homopolymers(s, n, out); else extract_period_n(s, n, min_period, out);
        return;
    }
    vec_t t = {0};
    if (min_period == 1) {
        extract_homopolymers(s, n, out);
        extract_period_n(s, n, 2, &t); merge_by_pos(out, &t); t.n = 0;
        if (max_period == 3) { extract_period_n(s, n, 3, &t); merge_by_pos(out, &t); }
    } else {
        extract_period_n(s, n, 2, out);
        extract_period_n(s, n, 3, &t); merge_by_pos(out, &t);
    }
    free(t.v);
}

/* ---- Lempel-Ziv path ----------------------------------------------------------------------------------------------- */
static const unsigned char* g_sort_text; static uint32_t g_sort_n;
static int suffix_cmp(const void* a, const void* b)
{
    const uint32_t i = *(const uint32_t*)a, j = *(const uint32_t*)b;
    const uint32_t li = g_sort_n - i, lj = g_sort_n - j, m = li < lj ? li : lj;
    const int c = memcmp(g_sort_text + i, g_sort_text + j, m);
    if (c) return c;
    return li < lj ? -1 : (li > lj ? 1 : 0);
}
static uint32_t fwd_lce(const char* s, uint32_t i, uint32_t j, uint32_t n, uint32_t len)  /* forward_lce :113-120; s[len] acts as a terminator */
{
    uint32_t k = 0;
    while (i + k < n && j + k < len && s[i + k] == s[j + k]) ++k;
    return k;
}
static uint32_t bwd_lce(const char* s, uint32_t i, uint32_t j, uint32_t t)                /* backward_lce :128-135: indices i, i-1, ... >= t */
{
    uint32_t k = 0;
    while (i >= t + k && s[i - k] == s[j - k]) { ++k; if (k > i) break; }
    return k;
}

static void extract_lz(const char* s, uint32_t n, uint32_t min_period, uint32_t max_period, vec_t* out)      /* :377-390 */
{
    /* make_suffix_array(str, 1) :92-100: the real suffix array plus one extra entry holding 0 */
    const uint32_t N = n + 1;
    uint32_t* sa = (uint32_t*)calloc(N + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) sa[i] = i;
    g_sort_text = (const unsigned char*)s; g_sort_n = n;
    qsort(sa, n, sizeof(uint32_t), suffix_cmp);
    sa[n] = 0;
    /* make_lcp_array(str, sa, 1) :142-158 (Kasai) */
    uint32_t* rank = (uint32_t*)calloc(n + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) rank[sa[i]] = i;
    uint32_t* lcp = (uint32_t*)calloc(N + 1, sizeof(uint32_t));
    for (uint32_t i = 0, h = 0; i < n; ++i) {
        if (rank[i] > 0) {
            h += fwd_lce(s, i + h, sa[rank[i] - 1] + h, n, n);
            lcp[rank[i]] = h;
            if (h > 0) --h;
        }
    }
    /* make_lpf_and_prev_occ_arrays, tandem.cpp:69-110, over the N = n + 1 entries */
    const uint32_t SENT = 0xffffffffu;
    uint32_t* lpf = (uint32_t*)calloc(N, sizeof(uint32_t)); uint32_t* prev_occ = (uint32_t*)calloc(N, sizeof(uint32_t));
    sa[N] = SENT; lcp[N] = 0;
    uint32_t* st_first = (uint32_t*)malloc((N + 1) * sizeof(uint32_t)); uint32_t* st_second = (uint32_t*)malloc((N + 1) * sizeof(uint32_t));
    uint32_t sp = 0;
    st_first[sp] = 0; st_second[sp] = sa[0]; ++sp;
    for (uint32_t i = 1; i <= N; ++i) {
        uint32_t u = lcp[i];
        while (sp && (sa[i] == SENT || sa[i] < st_second[sp - 1])) {
            const uint32_t vf = st_first[sp - 1], vs = st_second[sp - 1];
            const uint32_t lo = u < vf ? u : vf, hi = u < vf ? vf : u;   /* std::tie(u, lpf[v]) = std::minmax(v.first, u) */
            u = lo; lpf[vs] = hi;
            --sp;
            if (lpf[vs] == 0) prev_occ[vs] = SENT;
            else if (vf > u) prev_occ[vs] = sp ? st_second[sp - 1] : SENT;   /* stack.top() after the pop (an empty stack there is UB in the reference) */
            else prev_occ[vs] = sa[i];
        }
        if (i < N) { st_first[sp] = u; st_second[sp] = sa[i]; ++sp; }
    }
    /* lempel_ziv_factorisation_with_prev_block_occurences :218-245 */
    uint32_t* bpos = (uint32_t*)malloc((n + 1) * sizeof(uint32_t)); uint32_t* blen = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
    uint32_t* bprev = (uint32_t*)malloc((n + 1) * sizeof(uint32_t));
    uint32_t nb = 0, end = 1;
    bpos[0] = 0; blen[0] = 1; bprev[0] = SENT; nb = 1;
    while (end < n) {
        const uint32_t m = lpf[end] > 1 ? lpf[end] : 1;
        bpos[nb] = end; blen[nb] = m; bprev[nb] = prev_occ[end]; ++nb;
        end += m;
    }
    /* find_leftmost_maximal_repetitions + add_maximal_periodicities :251-292 */
    vec_t lmrs = {0};
    for (uint32_t h = 1; h < nb; ++h) {
        const uint32_t u = bpos[h], bn = blen[h];
        const uint32_t m2 = 2 * blen[h - 1] + bn, m = u < m2 ? u : m2;
        const uint32_t t = u - m, e = u + bn;
        const uint32_t jr = bn < max_period ? bn : max_period;
        for (uint32_t j = min_period; j <= jr; ++j) {                    /* rightmax periodicities */
            const uint32_t ls = bwd_lce(s, u - 1, u + j - 1, t);
            const uint32_t lp = fwd_lce(s, u + j, u, e, n);
            if (ls + lp >= j && j + lp < bn) vec_push(&lmrs, (rep_t){u - ls, j + lp + ls, j});
        }
        const uint32_t jl = m < max_period ? m : max_period;
        for (uint32_t j = min_period; j < jl; ++j) {                     /* leftmax periodicities */
            const uint32_t ls = bwd_lce(s, u - j - 1, u - 1, t);
            const uint32_t lp = fwd_lce(s, u, u - j, e, n);
            if (ls + lp >= j) vec_push(&lmrs, (rep_t){u - (ls + j), j + lp + ls, j});
        }
    }
    /* get_end_buckets :297-312 (a run is identified by pos and length only, operator== :74-77), get_sorted_buckets :318-334 */
    vec_t* endb = (vec_t*)calloc(n ? n : 1, sizeof(vec_t)); vec_t* sorted = (vec_t*)calloc(n ? n : 1, sizeof(vec_t));
    for (uint32_t i = 0; i < lmrs.n; ++i) {
        vec_t* b = &endb[lmrs.v[i].pos + lmrs.v[i].length - 1];
        int seen = 0;
        for (uint32_t k = 0; k < b->n; ++k) if (b->v[k].pos == lmrs.v[i].pos && b->v[k].length == lmrs.v[i].length) { seen = 1; break; }
        if (!seen) vec_push(b, lmrs.v[i]);
    }
    for (uint32_t i = 0; i < n; ++i) for (uint32_t k = 0; k < endb[i].n; ++k) vec_push(&sorted[endb[i].v[k].pos], endb[i].v[k]);
    /* extract_maximal_repetitions :337-375: runs inside an earlier occurrence of a block are copied into the block */
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t block_end = bpos[k] + blen[k];
        const uint32_t delta = bpos[k] - (bprev[k] != SENT ? bprev[k] : 0);
        const uint32_t max_target_end = block_end - delta;
        for (uint32_t j = bpos[k]; j < block_end; ++j) {
            const uint32_t target_start = j - delta;
            const vec_t* target = &sorted[target_start];
            uint32_t target_end = max_target_end;
            if (sorted[j].n) { const uint32_t c = target_start + sorted[j].v[0].length; target_end = c < max_target_end ? c : max_target_end; }
            uint32_t first = 0, len = target->n;                         /* std::lower_bound on run.pos + run.length < target_end */
            while (len > 0) {
                const uint32_t half = len >> 1, mid = first + half;
                if (target->v[mid].pos + target->v[mid].length < target_end) { first = mid + 1; len = len - half - 1; } else len = half;
            }
            if (first > 0) {
                rep_t* shifted = (rep_t*)malloc(first * sizeof(rep_t));
                for (uint32_t q = 0; q < first; ++q) shifted[q] = (rep_t){target->v[q].pos + delta, target->v[q].length, target->v[q].period};
                vec_insert_front(&sorted[j], shifted, first);
                free(shifted);
            }
        }
    }
    for (uint32_t i = 0; i < n; ++i) for (uint32_t k = 0; k < sorted[i].n; ++k) vec_push(out, sorted[i].v[k]);
    for (uint32_t i = 0; i < n; ++i) { free(endb[i].v); free(sorted[i].v); }
    free(endb); free(sorted); free(lmrs.v); free(bpos); free(blen); free(bprev); free(st_first); free(st_second);
    free(lpf); free(prev_occ); free(lcp); free(rank); free(sa);
}

static void extract_repeats(const char* s, uint32_t n, uint32_t min_period, uint32_t max_period, vec_t* out)   /* :497-514 */
{
    if (min_period == 0) ++min_period;
    if (n == 0 || n < min_period) return;
    if (max_period <= 3) extract_naive(s, n, min_period, max_period, out); else extract_lz(s, n, min_period, max_period, out);
}

int oracle_tandem_repeats(const char* str, uint32_t n, uint32_t min_period, uint32_t max_period, uint32_t* out_pos_len_period, uint32_t capacity)
{
    vec_t r = {0};
    extract_repeats(str, n, min_period, max_period, &r);
    for (uint32_t i = 0; i < r.n && i < capacity; ++i) { out_pos_len_period[3 * i] = r.v[i].pos; out_pos_len_period[3 * i + 1] = r.v[i].length; out_pos_len_period[3 * i + 2] = r.v[i].period; }
    const int k = (int)r.n;
    free(r.v);
    return k;
}

/* sort_by_length (repeat_based_indel_error_model.cpp:20-23) is std::sort on `length` alone: NOT stable, and which of two equal-length
 * repeats ends up later decides the extension penalty where they overlap. The reference is built with libstdc++, so this is its
 * std::sort, step for step (bits/stl_algo.h: __introsort_loop with median-of-three pivot to *first, __unguarded_partition, depth limit
 * 2 * floor(log2 n) with the heap-sort fallback, then __final_insertion_sort with threshold 16); pinned against the real std::sort in
 * oracle/_ref (ref_sort_by_length) and through the gap_extend vectors of the reference's model class. */
static int len_less(const rep_t* a, const rep_t* b) { return a->length < b->length; }
static void rep_swap(rep_t* a, rep_t* b) { const rep_t t = *a; *a = *b; *b = t; }
static void unguarded_linear_insert(rep_t* last)
{
    const rep_t val = *last; rep_t* next = last - 1;
    while (len_less(&val, next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void insertion_sort(rep_t* first, rep_t* last)
{
    if (first == last) return;
    for (rep_t* i = first + 1; i != last; ++i) {
        if (len_less(i, first)) { const rep_t val = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(rep_t)); *first = val; }
        else unguarded_linear_insert(i);
    }
}
static void adjust_heap(rep_t* first, long hole, long len, rep_t value)      /* std::__adjust_heap + __push_heap */
{
    const long top = hole; long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (len_less(first + child, first + (child - 1))) --child;
        first[hole] = first[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && len_less(first + parent, &value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void heap_sort_range(rep_t* first, rep_t* last)                       /* std::__partial_sort(first, last, last): make_heap + sort_heap */
{
    const long len = last - first;
    if (len >= 2) for (long parent = (len - 2) / 2; ; --parent) { adjust_heap(first, parent, len, first[parent]); if (parent == 0) break; }
    while (last - first > 1) { --last; const rep_t value = *last; *last = *first; adjust_heap(first, 0, last - first, value); }
}
static void move_median_to_first(rep_t* result, rep_t* a, rep_t* b, rep_t* c)
{
    if (len_less(a, b)) { if (len_less(b, c)) rep_swap(result, b); else if (len_less(a, c)) rep_swap(result, c); else rep_swap(result, a); }
    else if (len_less(a, c)) rep_swap(result, a);
    else if (len_less(b, c)) rep_swap(result, c);
    else rep_swap(result, b);
}
static rep_t* unguarded_partition(rep_t* first, rep_t* last, rep_t* pivot)
{
    for (;;) {
        while (len_less(first, pivot)) ++first;
        --last;
        while (len_less(pivot, last)) --last;
        if (!(first < last)) return first;
        rep_swap(first, last);
        ++first;
    }
}
static void introsort_loop(rep_t* first, rep_t* last, long depth_limit)
{
    while (last - first > 16) {
        if (depth_limit == 0) { heap_sort_range(first, last); return; }
        --depth_limit;
        rep_t* mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        rep_t* cut = unguarded_partition(first + 1, last, first);
        introsort_loop(cut, last, depth_limit);
        last = cut;
    }
}
static void sort_by_length_like_libstdcxx(rep_t* v, uint32_t n)
{
    if (n < 2) return;
    long lg = 0; for (uint32_t m = n; m > 1; m >>= 1) ++lg;                  /* std::__lg */
    introsort_loop(v, v + n, 2 * lg);
    if (n > 16) { insertion_sort(v, v + 16); for (rep_t* i = v + 16; i != v + n; ++i) unguarded_linear_insert(i); }
    else insertion_sort(v, v + n);
}

/* test hook: the permutation sort_by_length produces for the given lengths (ids 0..n-1 in `period`) */
void oracle_sort_by_length(const uint32_t* lengths, uint32_t n, uint32_t* out_ids)
{
    rep_t* v = (rep_t*)malloc((n ? n : 1) * sizeof(rep_t));
    for (uint32_t i = 0; i < n; ++i) v[i] = (rep_t){i, lengths[i], i};
    sort_by_length_like_libstdcxx(v, n);
    for (uint32_t i = 0; i < n; ++i) out_ids[i] = v[i].period;
    free(v);
}

/* ---- indel error model --------------------------------------------------------------------------------------------- */
static int8_t table_at(const int8_t* t, uint32_t periodicity) { return t[periodicity < OCT_PHMM_INDEL_TABLE ? periodicity : OCT_PHMM_INDEL_TABLE - 1]; }   /* get_min_penalty :44-47 on the 50-entry arrays (:23-28) */
static int8_t cap_at(const int8_t* t, uint32_t run) { return t[run < OCT_PHMM_SNV_TABLE ? run : OCT_PHMM_SNV_TABLE - 1]; }                          /* get_penalty :115-119 on the 51-entry arrays */

static int8_t open_penalty(const oct_phmm_error_model* m, const char* motif, uint32_t period, uint32_t length)      /* :57-86 */
{
    const uint32_t periodicity = length / period;
    switch (period) {
        case 1: return table_at(motif[0] == 'A' || motif[0] == 'T' ? m->at_homopolymer_open : m->cg_homopolymer_open, periodicity);
        case 2: {
            int8_t r = table_at(m->dinucleotide_open, periodicity);
            if (r > 7 && ((motif[0] == 'C' && motif[1] == 'G') || (motif[0] == 'G' && motif[1] == 'C'))) r -= 2;
            return r;
        }
        default: return table_at(m->trinucleotide_open, periodicity);
    }
}
static int8_t extend_penalty(const oct_phmm_error_model* m, uint32_t period, uint32_t length)                        /* :94-103 */
{
    const uint32_t periodicity = length / period;
    switch (period) {
        case 1: return table_at(m->homopolymer_extend, periodicity);
        case 2: return table_at(m->dinucleotide_extend, periodicity);
        default: return table_at(m->trinucleotide_extend, periodicity);
    }
}

void oracle_indel_penalties(const oct_phmm_error_model* m, const char* seq, uint32_t n, int8_t* gap_open, int8_t* gap_extend)
{
    /* RepeatBasedIndelErrorModel::do_set_penalties (vector overload) :67-83; defaults = first dinucleotide entries (:32-33) */
    memset(gap_open, m->dinucleotide_open[0], n); memset(gap_extend, m->dinucleotide_extend[0], n);
    vec_t r = {0};
    extract_repeats(seq, n, 1, 5, &r);                                  /* :15-18 */
    sort_by_length_like_libstdcxx(r.v, r.n);                             /* sort_by_length :20-23 */
    for (uint32_t i = 0; i < r.n; ++i) {
        const rep_t* q = &r.v[i];
        const int8_t op = open_penalty(m, seq + q->pos, q->period, q->length);
        const int8_t ex = extend_penalty(m, q->period, q->length);
        for (uint32_t k = 0; k < q->length; ++k) {
            if (op < gap_open[q->pos + k]) gap_open[q->pos + k] = op;   /* fill_n_if_less */
            gap_extend[q->pos + k] = ex;                                /* std::fill_n */
        }
    }
    free(r.v);
}

/* ---- SNV error model ------------------------------------------------------------------------------------------------- */
static int8_t base_hash(char b) { switch (b) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'T': return 4; default: return 5; } }   /* :89-105 */

/* count_runs :48-87 over mask[0..n) read with stride `step` from `start`; writes runs with the same stride */
static void count_runs(const int8_t* mask, uint32_t n, int start, int step, uint32_t* runs, uint32_t max_gap)
{
    if (n == 0) return;
    int8_t prev = mask[start];
    uint32_t count = prev > 0 ? 1u : 0u, gap = 0;
    runs[start] = 0;
    for (uint32_t k = 1; k < n; ++k) {
        const int idx = start + (int)k * step;
        const int8_t x = mask[idx];
        uint32_t v = 0;
        if (x == 0) {
            ++gap;
            if (count > 0) {
                if (gap == 1) { if (max_gap >= 1) v = count; else { v = count; count = 0; } }
                else if (gap > max_gap) count = 0;
            }
        } else if (prev == x) { gap = 0; ++count; }
        else { prev = x; v = count; count = 1; }
        runs[idx] = v;
    }
}

void oracle_snv_priors(const oct_phmm_error_model* m, const char* seq, uint32_t n, const uint8_t* substitution_mask,
                       char* mask_fwd, int8_t* prior_fwd, char* mask_rev, int8_t* prior_rev)
{
    /* BasicRepeatBasedSNVErrorModel::do_evaluate :144-179, max_period_ = 3 */
    vec_t r = {0};
    extract_repeats(seq, n, 1, 3, &r);
    int8_t* masks[3];
    for (int p = 0; p < 3; ++p) masks[p] = (int8_t*)calloc(n ? n : 1, 1);
    for (uint32_t i = 0; i < r.n; ++i) {
        int8_t hsh = 0;
        for (uint32_t k = 0; k < r.v[i].period; ++k) hsh = (int8_t)(hsh + base_hash(seq[r.v[i].pos + k]));          /* repeat_hash :107-113 */
        memset(masks[r.v[i].period - 1] + r.v[i].pos, hsh, r.v[i].length);
    }
    const int8_t max_quality = m->snv_caps[0][0];
    memset(prior_fwd, max_quality, n); memset(prior_rev, max_quality, n);
    uint32_t* runs = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    for (int p = 0; p < 3; ++p) {
        const uint32_t max_gap = (uint32_t)p + 2;
        count_runs(masks[p], n, 0, 1, runs, max_gap);
        for (uint32_t i = 0; i < n; ++i) { const int8_t c = cap_at(m->snv_caps[p], runs[i]); if (c < prior_fwd[i]) prior_fwd[i] = c; }   /* set_priors :121-130 */
        if (n) count_runs(masks[p], n, (int)n - 1, -1, runs, max_gap);
        for (uint32_t i = 0; i < n; ++i) { const int8_t c = cap_at(m->snv_caps[p], runs[i]); if (c < prior_rev[i]) prior_rev[i] = c; }
    }
    if (substitution_mask) for (uint32_t i = 0; i < n; ++i) if (substitution_mask[i]) { prior_fwd[i] = max_quality; prior_rev[i] = max_quality; }   /* :168-172 */
    for (uint32_t i = 0; i < n; ++i) {                                  /* rotate_copy :173-177: masks are the sequence rotated by one base */
        mask_fwd[i] = seq[(i + n - 1) % n];
        mask_rev[i] = seq[(i + 1) % n];
    }
    free(runs); for (int p = 0; p < 3; ++p) free(masks[p]); free(r.v);
}
