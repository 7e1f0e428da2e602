// Per-haplotype penalty vectors: what HaplotypeLikelihoodModel::reset (core/models/haplotype_likelihood_model.cpp:60-78) obtains from its two
// error models, computed by the product itself — on host threads or on the device, from the same source:
//
//   gap open / gap extend     RepeatBasedIndelErrorModel::do_set_penalties (core/models/error/repeat_based_indel_error_model.cpp:67-83) with the
//                             table look-ups of BasicRepeatBasedIndelErrorModel (basic_repeat_based_indel_error_model.cpp:44-103)
//   SNV masks / prior caps    BasicRepeatBasedSNVErrorModel::do_evaluate (repeat_based_snv_error_model.cpp:144-179, count_runs :48-87)
//   exact tandem repeats      tandem::extract_exact_tandem_repeats (lib/tandem/tandem.hpp:497-514): the scans of :392-436 for periods <= 3,
//                             the Lempel-Ziv / maximal-periodicity path (:183-390, tandem.cpp:69-110) for the indel model's periods 1-5
//
// The vectors must be those of the reference bit for bit, and they inherit every habit of that library (runs that touch the end of the
// string or have period == max_period are mostly dropped, some non-maximal sub-runs are reported, an extra suffix-array entry travels
// through the LPF stack) and of libstdc++'s unstable std::sort, which decides the extension penalty where equal-length repeats overlap.
// So this is the same sequential algorithm, laid out for one flat workspace per haplotype (no allocation inside, 32-bit words only):
// suffix array -> LCP -> LPF + previous occurrences -> LZ blocks -> maximal periodicities -> per-end / per-start buckets -> runs copied
// from earlier block occurrences -> sort by length -> table look-ups.
//
// Host and device compile this file alike (plain loops, no recursion, no library calls); oct_phmm.hip runs it over host threads for
// region-sized calls and one GPU lane per haplotype for large batches (k_penalty_vectors).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/oct_phmm.h"
#include "phmm_hw.hpp"

#if defined(OCTPHMM_SIM)
#define OCT_EM inline
#else
#define OCT_EM __host__ __device__ inline
#endif

namespace octphmm { namespace em {

struct Repeat { uint32_t pos, length, period; };                 // tandem::Repeat, lib/tandem/tandem.hpp:62-72
constexpr uint32_t kNone = 0xffffffffu;

// Capacities of the two variable-size lists, in entries per haplotype base (+ a constant): the periodicities found per LZ block and the
// final run list. Real haplotypes use a few percent of this; a string that exceeds it reports kOverflow and the caller retries with `grow`.
OCT_EM uint32_t cap_found(uint32_t n, uint32_t grow) { return (4u * n + 64u) * grow; }
OCT_EM uint32_t cap_runs(uint32_t n, uint32_t grow) { return (8u * n + 64u) * grow; }
OCT_EM size_t workspace_words(uint32_t n, uint32_t grow)
{
    const size_t N = (size_t)n + 4;
    return 10 * N                                   // sa, rank | stack a, lcp, lpf, prev_occ, stack b, block pos / len / prev, counts
         + 3 * (size_t)cap_found(n, grow) * 2       // periodicities found + the kept ones in bucket order
         + 3 * (size_t)cap_runs(n, grow)            // the final run list
         + 2 * N                                    // first entry of every start position in the two lists
         + 3 * N                                    // the SNV model's three masks (one byte per base, kept in words for alignment: n / 4 each, rounded up generously)
         + N;                                       // run counts
}
enum : int { kOk = 0, kOverflow = 1 };

// ---- small string helpers --------------------------------------------------------------------------------------------------
OCT_EM int suffix_less(const uint8_t* s, uint32_t n, uint32_t a, uint32_t b)    // suffix a < suffix b (a != b); a suffix that is a proper prefix of the other is smaller
{
    while (a < n && b < n) {
        if (s[a] != s[b]) return s[a] < s[b];
        ++a; ++b;
    }
    return a >= n;
}
OCT_EM uint32_t forward_lce(const uint8_t* s, uint32_t i, uint32_t j, uint32_t i_end, uint32_t j_end)   // tandem.hpp:113-120
{
    uint32_t k = 0;
    while (i + k < i_end && j + k < j_end && s[i + k] == s[j + k]) ++k;
    return k;
}
OCT_EM uint32_t backward_lce(const uint8_t* s, uint32_t i, uint32_t j, uint32_t t)                       // tandem.hpp:128-135: positions i, i-1, ... >= t
{
    uint32_t k = 0;
    while (i >= t + k && s[i - k] == s[j - k]) { ++k; if (k > i) break; }
    return k;
}

// Suffix array of s[0, n) into sa. Strings here are a few hundred bases: a 4-byte key per suffix and a shell sort on (key, suffix)
// settle almost every comparison in one word compare; equal keys fall back to the byte loop.
OCT_EM void build_suffix_array(const uint8_t* s, uint32_t n, uint32_t* sa, uint32_t* key)
{
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t k = 0;
        for (uint32_t b = 0; b < 4; ++b) k = k << 8 | (i + b < n ? (uint32_t)s[i + b] + 1u : 0u);    // past the end sorts below every byte
        key[i] = k; sa[i] = i;
    }
    auto less = [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : (bool)suffix_less(s, n, a + 4 < n ? a + 4 : n, b + 4 < n ? b + 4 : n); };
    uint32_t gap = 1;
    while (gap < n / 3) gap = 3 * gap + 1;
    for (; gap >= 1; gap /= 3) {
        for (uint32_t i = gap; i < n; ++i) {
            const uint32_t v = sa[i];
            uint32_t j = i;
            while (j >= gap && less(v, sa[j - gap])) { sa[j] = sa[j - gap]; j -= gap; }
            sa[j] = v;
        }
        if (gap == 1) break;
    }
}

// ---- libstdc++'s std::sort on Repeat::length (sort_by_length, repeat_based_indel_error_model.cpp:20-23), step for step -----------
// bits/stl_algo.h: __introsort_loop (median of three to *first, __unguarded_partition, depth limit 2 floor(log2 n), heap sort beyond it),
// then __final_insertion_sort with threshold 16. The recursion on the right part is an explicit stack here.
OCT_EM bool shorter(const Repeat& a, const Repeat& b) { return a.length < b.length; }
OCT_EM void swap_runs(Repeat& a, Repeat& b) { const Repeat t = a; a = b; b = t; }
OCT_EM void unguarded_linear_insert(Repeat* last)
{
    const Repeat val = *last; Repeat* next = last - 1;
    while (shorter(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
OCT_EM void insertion_sort(Repeat* first, Repeat* last)
{
    if (first == last) return;
    for (Repeat* i = first + 1; i != last; ++i) {
        if (shorter(*i, *first)) { const Repeat val = *i; for (Repeat* p = i; p != first; --p) *p = *(p - 1); *first = val; }
        else unguarded_linear_insert(i);
    }
}
OCT_EM void adjust_heap(Repeat* first, long hole, long len, Repeat value)
{
    const long top = hole; long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (shorter(first[child], first[child - 1])) --child;
        first[hole] = first[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && shorter(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
OCT_EM void heap_sort(Repeat* first, Repeat* last)
{
    const long len = last - first;
    if (len >= 2) for (long parent = (len - 2) / 2; ; --parent) { adjust_heap(first, parent, len, first[parent]); if (parent == 0) break; }
    while (last - first > 1) { --last; const Repeat value = *last; *last = *first; adjust_heap(first, 0, last - first, value); }
}
OCT_EM void sort_by_length(Repeat* v, uint32_t n)
{
    if (n < 2) return;
    long lg = 0; for (uint32_t m = n; m > 1; m >>= 1) ++lg;
    struct Range { Repeat* first; Repeat* last; long depth; };
    Range todo[80]; int top = 0;                                          // depth limit 2 lg <= 64
    todo[top++] = Range {v, v + n, 2 * lg};
    while (top) {
        Range r = todo[--top];
        while (r.last - r.first > 16) {
            if (r.depth == 0) { heap_sort(r.first, r.last); break; }
            --r.depth;
            Repeat* mid = r.first + (r.last - r.first) / 2;
            Repeat *a = r.first + 1, *b = mid, *c = r.last - 1;              // __move_median_to_first
            if (shorter(*a, *b)) { if (shorter(*b, *c)) swap_runs(*r.first, *b); else if (shorter(*a, *c)) swap_runs(*r.first, *c); else swap_runs(*r.first, *a); }
            else if (shorter(*a, *c)) swap_runs(*r.first, *a);
            else if (shorter(*b, *c)) swap_runs(*r.first, *c);
            else swap_runs(*r.first, *b);
            Repeat *lo = r.first + 1, *hi = r.last;                          // __unguarded_partition around *first
            for (;;) {
                while (shorter(*lo, *r.first)) ++lo;
                --hi;
                while (shorter(*r.first, *hi)) --hi;
                if (!(lo < hi)) break;
                swap_runs(*lo, *hi);
                ++lo;
            }
            // the reference recurses into [cut, last) FIRST and then loops on [first, cut): neither part's outcome depends on the other,
            // so the right part may wait on the stack
            todo[top++] = Range {lo, r.last, r.depth};
            r.last = lo;
        }
    }
    if (n > 16) { insertion_sort(v, v + 16); for (Repeat* i = v + 16; i != v + n; ++i) unguarded_linear_insert(i); }
    else insertion_sort(v, v + n);
}

// ---- exact tandem repeats of periods min_period..max_period through the Lempel-Ziv path (tandem.hpp:377-390) ---------------------
// Writes the runs in the library's output order into `runs` (capacity cap_runs) and returns their number through n_runs.
OCT_EM int lz_tandem_repeats(const uint8_t* s, uint32_t n, uint32_t min_period, uint32_t max_period, uint32_t* w, uint32_t grow, Repeat* runs, uint32_t* n_runs)
{
    const uint32_t N = n + 1, W = n + 4;
    uint32_t* sa = w; uint32_t* rank = w + W; uint32_t* lcp = w + 2 * W; uint32_t* lpf = w + 3 * W; uint32_t* prev_occ = w + 4 * W;
    uint32_t* st_b = w + 5 * W; uint32_t* bpos = w + 6 * W; uint32_t* blen = w + 7 * W; uint32_t* bprev = w + 8 * W; uint32_t* cnt = w + 9 * W;
    uint32_t* st_a = rank;                                                // the LPF stack reuses the rank array (dead after the LCP pass)
    Repeat* found = (Repeat*)(w + 10 * W);
    const uint32_t found_cap = cap_found(n, grow), runs_cap = cap_runs(n, grow);
    Repeat* kept = found + found_cap;
    uint32_t* first_kept = (uint32_t*)(kept + found_cap);                 // [n + 1] start of every position's initial runs in `kept`
    uint32_t* first_run = first_kept + W;                                 // [n + 1] start of every position's final runs in `runs`
    *n_runs = 0;

    // make_suffix_array(str, 1) :92-100: the suffix array plus one extra entry that holds 0; Kasai's LCP (:142-158)
    build_suffix_array(s, n, sa, lcp);
    sa[n] = 0; sa[N] = kNone;
    for (uint32_t i = 0; i < n; ++i) rank[sa[i]] = i;
    for (uint32_t i = 0; i <= N; ++i) lcp[i] = 0;
    for (uint32_t i = 0, h = 0; i < n; ++i) {
        if (rank[i] > 0) {
            h += forward_lce(s, i + h, sa[rank[i] - 1] + h, n, n);
            lcp[rank[i]] = h;
            if (h > 0) --h;
        }
    }
    // longest previous factor + where it occurred, tandem.cpp:69-110, over the N entries (the sentinel flushes the stack)
    for (uint32_t i = 0; i < N; ++i) { lpf[i] = 0; prev_occ[i] = 0; }
    uint32_t sp = 0;
    st_a[sp] = 0; st_b[sp] = sa[0]; ++sp;
    for (uint32_t i = 1; i <= N; ++i) {
        uint32_t u = lcp[i];
        while (sp && (sa[i] == kNone || sa[i] < st_b[sp - 1])) {
            const uint32_t vf = st_a[sp - 1], vs = st_b[sp - 1];
            const uint32_t lo = u < vf ? u : vf, hi = u < vf ? vf : u;   // std::tie(u, lpf[v]) = std::minmax(v.first, u)
            u = lo; lpf[vs] = hi;
            --sp;
            if (lpf[vs] == 0) prev_occ[vs] = kNone;
            else if (vf > u) prev_occ[vs] = sp ? st_b[sp - 1] : kNone;
            else prev_occ[vs] = sa[i];
        }
        if (i < N) { st_a[sp] = u; st_b[sp] = sa[i]; ++sp; }
    }
    // Lempel-Ziv blocks with a previous occurrence each, :218-245
    uint32_t nb = 1, end = 1;
    bpos[0] = 0; blen[0] = 1; bprev[0] = kNone;
    while (end < n) {
        const uint32_t m = lpf[end] > 1 ? lpf[end] : 1;
        bpos[nb] = end; blen[nb] = m; bprev[nb] = prev_occ[end]; ++nb;
        end += m;
    }
    // leftmost maximal repetitions that cross or touch a block border, :251-292
    uint32_t nf = 0;
    for (uint32_t h = 1; h < nb; ++h) {
        const uint32_t u = bpos[h], bn = blen[h];
        const uint32_t m2 = 2 * blen[h - 1] + bn, m = u < m2 ? u : m2;
        const uint32_t t = u - m, e = u + bn;
        const uint32_t jr = bn < max_period ? bn : max_period;
        for (uint32_t j = min_period; j <= jr; ++j) {
            const uint32_t ls = backward_lce(s, u - 1, u + j - 1, t);
            const uint32_t lp = forward_lce(s, u + j, u, e, n);
            if (ls + lp >= j && j + lp < bn) { if (nf == found_cap) return kOverflow; found[nf++] = Repeat {u - ls, j + lp + ls, j}; }
        }
        const uint32_t jl = m < max_period ? m : max_period;
        for (uint32_t j = min_period; j < jl; ++j) {
            const uint32_t ls = backward_lce(s, u - j - 1, u - 1, t);
            const uint32_t lp = forward_lce(s, u, u - j, e, n);
            if (ls + lp >= j) { if (nf == found_cap) return kOverflow; found[nf++] = Repeat {u - (ls + j), j + lp + ls, j}; }
        }
    }
    // get_end_buckets (:297-312: a run is its position and length, the first period found stays) and get_sorted_buckets (:318-334):
    // per start position, the distinct runs in order of their end, ties in order of discovery. Two stable counting sorts.
    uint32_t nk = 0;
    {
        Repeat* by_end = kept;                                            // stable by end position
        for (uint32_t i = 0; i <= n; ++i) cnt[i] = 0;
        for (uint32_t i = 0; i < nf; ++i) ++cnt[found[i].pos + found[i].length - 1];
        for (uint32_t i = 0, run = 0; i <= n; ++i) { const uint32_t c = cnt[i]; cnt[i] = run; run += c; }
        for (uint32_t i = 0; i < nf; ++i) by_end[cnt[found[i].pos + found[i].length - 1]++] = found[i];
        // drop repeats of an earlier entry with the same end and start (entries of one end bucket are contiguous now)
        for (uint32_t i = 0; i < nf; ++i) {
            bool seen = false;
            const uint32_t e = by_end[i].pos + by_end[i].length;
            for (uint32_t k = i; k > 0 && by_end[k - 1].pos + by_end[k - 1].length == e; --k)
                if (by_end[k - 1].pos == by_end[i].pos && by_end[k - 1].period != kNone) { seen = true; break; }
            if (seen) by_end[i].period = kNone;                           // tombstone: keeps its place for the scans above
        }
        for (uint32_t i = 0; i <= n; ++i) cnt[i] = 0;
        for (uint32_t i = 0; i < nf; ++i) if (by_end[i].period != kNone) ++cnt[by_end[i].pos];
        for (uint32_t i = 0, run = 0; i <= n; ++i) { const uint32_t c = cnt[i]; cnt[i] = run; first_kept[i] = run; run += c; }
        for (uint32_t i = 0; i < nf; ++i) if (by_end[i].period != kNone) { found[cnt[by_end[i].pos]++] = by_end[i]; ++nk; }   // `found` is free again: stable by start
    }
    const Repeat* initial = found;                                        // initial[first_kept[p] .. first_kept[p + 1]) = runs that start at p
    first_kept[n] = nk;
    // extract_maximal_repetitions :337-375: inside a block, the runs that lie within the block's earlier occurrence are copied (shifted)
    // in front of the position's own runs. Positions are visited left to right and a copy source lies to the left, so every final
    // list is written once, in output order.
    uint32_t nr = 0;
    for (uint32_t k = 0; k < nb; ++k) {
        const uint32_t block_end = bpos[k] + blen[k];
        const uint32_t delta = bpos[k] - (bprev[k] != kNone ? bprev[k] : 0);
        const uint32_t max_target_end = block_end - delta;
        for (uint32_t j = bpos[k]; j < block_end; ++j) {
            first_run[j] = nr;
            const uint32_t target_start = j - delta;                      // delta > 0 for every block but the first, whose target is itself (nothing to copy yet)
            uint32_t target_end = max_target_end;
            if (first_kept[j + 1] > first_kept[j]) { const uint32_t c = target_start + initial[first_kept[j]].length; target_end = c < max_target_end ? c : max_target_end; }
            if (target_start < j) {
                const Repeat* target = runs + first_run[target_start];
                uint32_t first = 0, len = first_run[target_start + 1] - first_run[target_start];     // std::lower_bound on pos + length < target_end
                while (len > 0) {
                    const uint32_t half = len >> 1, mid = first + half;
                    if (target[mid].pos + target[mid].length < target_end) { first = mid + 1; len = len - half - 1; } else len = half;
                }
                if (nr + first > runs_cap) return kOverflow;
                for (uint32_t q = 0; q < first; ++q) runs[nr++] = Repeat {target[q].pos + delta, target[q].length, target[q].period};
            }
            const uint32_t own = first_kept[j + 1] - first_kept[j];
            if (nr + own > runs_cap) return kOverflow;
            for (uint32_t q = 0; q < own; ++q) runs[nr++] = initial[first_kept[j] + q];
            first_run[j + 1] = nr;
        }
    }
    *n_runs = nr;
    return kOk;
}

// ---- the two models ------------------------------------------------------------------------------------------------------------
OCT_EM int8_t table_at(const int8_t* t, uint32_t periodicity) { return t[periodicity < OCT_PHMM_INDEL_TABLE ? periodicity : OCT_PHMM_INDEL_TABLE - 1]; }   // get_min_penalty :44-47
OCT_EM int8_t cap_at(const int8_t* t, uint32_t run) { return t[run < OCT_PHMM_SNV_TABLE ? run : OCT_PHMM_SNV_TABLE - 1]; }                            // get_penalty :115-119

// RepeatBasedIndelErrorModel::do_set_penalties, vector overload :67-83
OCT_EM int indel_penalties(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, uint32_t* w, uint32_t grow, int8_t* gap_open, int8_t* gap_extend)
{
    for (uint32_t i = 0; i < n; ++i) { gap_open[i] = m.dinucleotide_open[0]; gap_extend[i] = m.dinucleotide_extend[0]; }   // complex_open / complex_extend :32-33
    if (n == 0) return kOk;
    Repeat* runs = (Repeat*)(w + 10 * (size_t)(n + 4) + 6 * (size_t)cap_found(n, grow) + 2 * (size_t)(n + 4));
    uint32_t nr = 0;
    const int rc = lz_tandem_repeats(s, n, 1, 5, w, grow, runs, &nr);     // extract_repeats :15-18
    if (rc != kOk) return rc;
    sort_by_length(runs, nr);                                             // :20-23
    for (uint32_t i = 0; i < nr; ++i) {
        const Repeat& q = runs[i];
        const uint32_t periodicity = q.length / q.period;
        const uint8_t* motif = s + q.pos;
        int8_t open, extend;
        switch (q.period) {                                               // get_open_penalty :57-86, get_extension_penalty :94-103
            case 1: open = table_at(motif[0] == 'A' || motif[0] == 'T' ? m.at_homopolymer_open : m.cg_homopolymer_open, periodicity);
                    extend = table_at(m.homopolymer_extend, periodicity); break;
            case 2: open = table_at(m.dinucleotide_open, periodicity);
                    if (open > 7 && ((motif[0] == 'C' && motif[1] == 'G') || (motif[0] == 'G' && motif[1] == 'C'))) open = (int8_t)(open - 2);
                    extend = table_at(m.dinucleotide_extend, periodicity); break;
            default: open = table_at(m.trinucleotide_open, periodicity);
                    extend = table_at(m.trinucleotide_extend, periodicity); break;
        }
        for (uint32_t k = 0; k < q.length; ++k) {
            if (open < gap_open[q.pos + k]) gap_open[q.pos + k] = open;   // fill_n_if_less
            gap_extend[q.pos + k] = extend;                               // fill_n: the last (longest) covering run decides
        }
    }
    return kOk;
}

OCT_EM int8_t base_hash(uint8_t b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 3 : b == 'T' ? 4 : 5; }   // :89-105

// count_runs :48-87 over mask read with stride `step` from `start`
OCT_EM void count_runs(const int8_t* mask, uint32_t n, int start, int step, uint32_t* runs, uint32_t max_gap)
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

OCT_EM uint32_t next_unequal_pair(const uint8_t* s, uint32_t from, uint32_t n)     // std::adjacent_find(..., not_equal_to): first i with s[i] != s[i + 1], else n
{
    uint32_t it = from;
    if (it >= n) return n;
    while (it + 1 < n && s[it] == s[it + 1]) ++it;
    return it + 1 >= n ? n : it;
}

// BasicRepeatBasedSNVErrorModel::do_evaluate :144-179 (max_period 3: the scans of tandem.hpp:392-436; the masks of one period only see
// that period's runs, in the order the scan finds them, so the library's merge by position is not needed)
OCT_EM void snv_priors(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* substitution_mask, uint32_t* w,
                       uint8_t* mask_fwd, int8_t* prior_fwd, uint8_t* mask_rev, int8_t* prior_rev)
{
    if (!m.use_snv_model) {                                               // model.cpp:68-73
        for (uint32_t i = 0; i < n; ++i) { mask_fwd[i] = s[i]; mask_rev[i] = s[i]; prior_fwd[i] = 100; prior_rev[i] = 100; }
        return;
    }
    const size_t W = (size_t)n + 4;
    int8_t* masks[3] = {(int8_t*)w, (int8_t*)(w + W), (int8_t*)(w + 2 * W)};
    uint32_t* runs = w + 3 * W;
    for (int p = 0; p < 3; ++p) for (uint32_t i = 0; i < n; ++i) masks[p][i] = 0;
    auto mark = [&](uint32_t pos, uint32_t length, uint32_t period) {
        int8_t h = 0;
        for (uint32_t k = 0; k < period; ++k) h = (int8_t)(h + base_hash(s[pos + k]));   // repeat_hash :107-113
        for (uint32_t k = 0; k < length; ++k) masks[period - 1][pos + k] = h;
    };
    for (uint32_t cur = 0; cur < n;) {                                    // homopolymers :392-409
        uint32_t it = cur;
        while (it + 1 < n && s[it] != s[it + 1]) ++it;
        if (it + 1 >= n) break;
        uint32_t it2 = it + 1;
        while (it2 < n && s[it2] == s[it]) ++it2;
        mark(it, it2 - it, 1);
        cur = it2;
    }
    for (uint32_t P = 2; P <= 3; ++P) {                                   // extract_exact_tandem_repeats<P> :411-436
        if (n < 2 * P) continue;
        uint32_t it1 = next_unequal_pair(s, 0, n);
        if (it1 == n) continue;
        for (uint32_t it2 = it1 + P; it2 < n;) {
            uint32_t a = it2, b = it1;                                    // std::mismatch(it2, last, it1)
            while (a < n && s[a] == s[b]) { ++a; ++b; }
            if (b >= it2) { mark(it1, a - it1, P); it1 = b; } else ++it1;
            it1 = next_unequal_pair(s, it1, n);
            if (it1 == n) break;
            it2 = it1 + P;
        }
    }
    const int8_t max_quality = m.snv_caps[0][0];
    for (uint32_t i = 0; i < n; ++i) { prior_fwd[i] = max_quality; prior_rev[i] = max_quality; }
    for (int p = 0; p < 3; ++p) {
        const uint32_t max_gap = (uint32_t)p + 2;
        count_runs(masks[p], n, 0, 1, runs, max_gap);
        for (uint32_t i = 0; i < n; ++i) { const int8_t c = cap_at(m.snv_caps[p], runs[i]); if (c < prior_fwd[i]) prior_fwd[i] = c; }   // set_priors :121-130
        if (n) count_runs(masks[p], n, (int)n - 1, -1, runs, max_gap);
        for (uint32_t i = 0; i < n; ++i) { const int8_t c = cap_at(m.snv_caps[p], runs[i]); if (c < prior_rev[i]) prior_rev[i] = c; }
    }
    if (substitution_mask) for (uint32_t i = 0; i < n; ++i) if (substitution_mask[i]) { prior_fwd[i] = max_quality; prior_rev[i] = max_quality; }   // :168-172
    for (uint32_t i = 0; i < n; ++i) {                                    // rotate_copy :173-177
        mask_fwd[i] = s[(i + n - 1) % n];
        mask_rev[i] = s[(i + 1) % n];
    }
}

// All six vectors of one haplotype. `w`: workspace_words(n, grow) words.
OCT_EM int penalty_vectors(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* substitution_mask, uint32_t* w, uint32_t grow,
                           int8_t* gap_open, int8_t* gap_extend, uint8_t* mask_fwd, int8_t* prior_fwd, uint8_t* mask_rev, int8_t* prior_rev)
{
    const int rc = indel_penalties(m, s, n, w, grow, gap_open, gap_extend);
    if (rc != kOk) return rc;
    snv_priors(m, s, n, substitution_mask, w, mask_fwd, prior_fwd, mask_rev, prior_rev);
    return kOk;
}

}} // namespace octphmm::em
