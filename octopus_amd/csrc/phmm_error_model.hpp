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
// region-sized calls, and for large batches on the device: one wave per haplotype with the workspace in LDS (k_penalty_vectors_wave), or
// one lane per haplotype with the workspace in HBM when a haplotype is too long for that (k_penalty_vectors).
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
// final run list. Real haplotypes use a few percent of this; a string that exceeds it reports kOverflow and the caller retries with a larger
// `grow`. grow == 0 is the tight sizing of the wave kernel, whose workspace must fit in LDS (an overflow there goes to the host).
OCT_EM uint32_t cap_found(uint32_t n, uint32_t grow) { return grow ? (4u * n + 64u) * grow : (n / 2 + 32u < n + 4u ? n / 2 + 32u : n + 4u); }
OCT_EM uint32_t cap_runs(uint32_t n, uint32_t grow) { return grow ? (8u * n + 64u) * grow : n + 32u; }
// The flat workspace, in 32-bit words (N = n + 4 or n + 5, even): ten arrays of N words (suffix array, ranks | LPF stack, LCPs, LPFs, previous occurrences,
// second stack array, LZ block pos / len / prev, counts), the periodicities found + the kept ones in bucket order (3 words each), the first
// entry of every start position in the two run lists (2 N), the final run list. At the tight sizing the two periodicity lists lie over
// the first six arrays, which are dead by then.
OCT_EM uint32_t stride(uint32_t n) { return (n + 5u) & ~1u; }        // words per array: n + 4 and more, even (64-bit entries stay aligned)
struct Layout { size_t found, first_kept, runs, total; };
OCT_EM Layout layout(uint32_t n, uint32_t grow)
{
    const size_t N = stride(n), lists = 6 * (size_t)cap_found(n, grow);
    Layout l;
    l.found = grow ? 10 * N : 0;
    l.first_kept = grow ? 10 * N + lists : 10 * N;
    l.runs = l.first_kept + 2 * N;
    l.total = l.runs + 3 * (size_t)cap_runs(n, grow);
    return l;
}
OCT_EM size_t workspace_words(uint32_t n, uint32_t grow) { return layout(n, grow).total; }   // (the SNV model's masks and run counts reuse the first 9 N words)
enum : int { kOk = 0, kOverflow = 1 };

// ---- who executes: one thread (a host thread, or one GPU lane per haplotype), or the 64 lanes of a wave that share one haplotype --------
// The sequential phases (LPF stack, LZ blocks, buckets, run copying, std::sort replica, the count_runs state machines) run on lane 0 (the
// six count_runs passes on six lanes); suffix ranks, LCPs, the longest-common-extension scans at the block borders and all per-position
// loops spread over the lanes. Results do not depend on the policy: every parallel phase writes what the sequential loop writes.
struct Seq {
    OCT_EM uint32_t lane() const { return 0; }
    OCT_EM uint32_t nl() const { return 1; }
    OCT_EM void sync() const {}
    OCT_EM uint32_t claim(bool valid, uint32_t& count) const { const uint32_t at = count; count += valid ? 1u : 0u; return at; }   // slot of this item in discovery order
    OCT_EM bool any(bool v) const { return v; }
    OCT_EM uint32_t take(uint32_t* counter) const { return (*counter)++; }                                                          // next free place (any order)
    OCT_EM void tick(int) const {}
};
#if defined(OCTPHMM_SIM) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
struct Wave {
    uint32_t l;
    unsigned long long* prof = nullptr;                                   // phase clocks of lane 0, summed over the grid (profiling builds of the launch only)
    mutable unsigned long long t_last = 0;
    OCT_DEVICE Wave() : l(hw::thread_idx() & 63u) {}
    OCT_DEVICE void tick(int phase) const
    {
#if !defined(OCTPHMM_SIM)
        if (prof) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (l == 0 && t_last) atomicAdd(prof + phase, now - t_last); t_last = now; }
#else
        (void)phase;
#endif
    }
    OCT_DEVICE uint32_t lane() const { return l; }
    OCT_DEVICE uint32_t nl() const { return 64; }
    OCT_DEVICE void sync() const { hw::wave_lds_fence(); }
    OCT_DEVICE uint32_t claim(bool valid, uint32_t& count) const
    {
        const uint64_t m = hw::ballot(valid);
        const uint32_t at = count + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull));
        count += (uint32_t)__builtin_popcountll(m);
        return at;
    }
    OCT_DEVICE bool any(bool v) const { return hw::ballot(v) != 0; }
    OCT_DEVICE uint32_t take(uint32_t* counter) const { return hw::atomic_add_lds_u32(counter, 1u); }
};
#endif

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
// Writes the runs in the library's output order into `runs` (capacity cap_runs) and returns their number through n_runs (valid on
// every lane). `w` and `runs` are shared by the lanes of the policy (LDS for a wave).
template <class X>
OCT_EM int lz_tandem_repeats(const X& x, const uint8_t* s, uint32_t n, uint32_t min_period, uint32_t max_period, uint32_t* w, uint32_t grow, Repeat* runs, uint32_t* n_runs)
{
    const uint32_t N = n + 1, W = stride(n), lane = x.lane(), nl = x.nl();
    uint32_t* sa = w; uint32_t* rank = w + W; uint32_t* lcp = w + 2 * W; uint32_t* lpf = w + 3 * W; uint32_t* prev_occ = w + 4 * W;
    uint32_t* st_b = w + 5 * W; uint32_t* bpos = w + 6 * W; uint32_t* blen = w + 7 * W; uint32_t* bprev = w + 8 * W; uint32_t* cnt = w + 9 * W;
    uint32_t* st_a = rank;                                                // the LPF stack reuses the rank array (dead after the LCP pass)
    const Layout lay = layout(n, grow);
    Repeat* found = (Repeat*)(w + lay.found);
    const uint32_t found_cap = cap_found(n, grow), runs_cap = cap_runs(n, grow);
    Repeat* kept = found + found_cap;
    uint32_t* first_kept = w + lay.first_kept;                            // [n + 1] start of every position's initial runs in `kept`
    uint32_t* first_run = first_kept + W;                                 // [n + 1] start of every position's final runs in `runs`
    uint32_t* shared = first_run + n + 2;                                 // [2] number of blocks / status, written by lane 0, read by all (inside the W-sized slot)
    *n_runs = 0;
    if (nl != 1 && n >= 0xffffu) return kOverflow;                        // (several lanes: positions and lengths share a word in places)

    // make_suffix_array(str, 1) :92-100: the suffix array plus one extra entry that holds 0
    if (nl == 1) build_suffix_array(s, n, sa, lcp);
    else {
        // several lanes: every suffix counts the suffixes below it (its rank IS its place in the array; the array is unique, so this is the
        // same array). 8-byte keys: ties go to the byte loop only inside repeats of 8 and more.
        uint64_t* key = (uint64_t*)lcp;                                   // over the LCP and LPF arrays (written later); 16-byte aligned
        for (uint32_t i = lane; i < W; i += nl) {
            uint32_t hi = 0xffffffffu, lo = 0xffffffffu;                 // padding behind the last suffix: never below, never equal
            if (i < n) {
                hi = 0; lo = 0;
                for (uint32_t b4 = 0; b4 < 4; ++b4) { hi = hi << 8 | (i + b4 < n ? (uint32_t)s[i + b4] + 1u : 0u); lo = lo << 8 | (i + 4 + b4 < n ? (uint32_t)s[i + 4 + b4] + 1u : 0u); }
            }
            key[i] = (uint64_t)hi << 32 | lo;
        }
        x.sync();
        // suffixes that share their 8 bytes with others form a group of consecutive ranks: each member first takes any free place of the
        // group, then counts the members below it (8 bytes at a time through the same keys), then moves to its own place
        uint32_t* taken = st_b; uint32_t* group_size = prev_occ; uint32_t* within = bpos;     // all unused so far
        for (uint32_t i = lane; i < n; i += nl) taken[i] = 0;
        x.sync();
        for (uint32_t i = lane; i < n; i += nl) {
            const uint64_t ki = key[i];
            uint32_t below = 0, same = 0;
            for (uint32_t j = 0; j < n; j += 4) {                         // (the padding makes the last group whole)
                const uint64_t k0 = key[j], k1 = key[j + 1], k2 = key[j + 2], k3 = key[j + 3];
                below += (k0 < ki ? 1u : 0u) + (k1 < ki ? 1u : 0u) + (k2 < ki ? 1u : 0u) + (k3 < ki ? 1u : 0u);
                same += (k0 == ki ? 1u : 0u) + (k1 == ki ? 1u : 0u) + (k2 == ki ? 1u : 0u) + (k3 == ki ? 1u : 0u);
            }
            rank[i] = below; group_size[i] = same;
            sa[below + (same > 1 ? x.take(taken + below) : 0u)] = i;
        }
        x.sync();
        for (uint32_t i = lane; i < n; i += nl) {
            const uint32_t same = group_size[i], base = rank[i];
            if (same == 1) continue;
            uint32_t before = 0;
            for (uint32_t t = 0; t < same; ++t) {
                const uint32_t other = sa[base + t];
                if (other == i) continue;
                for (uint32_t a = other + 8, b8 = i + 8; ; a += 8, b8 += 8) {   // the shorter suffix runs out first: its key is 0 there, the other's is not
                    const uint64_t ka = a < n ? key[a] : 0, kb = b8 < n ? key[b8] : 0;
                    if (ka != kb) { before += ka < kb ? 1u : 0u; break; }
                }
            }
            within[i] = before;
        }
        x.sync();
        for (uint32_t i = lane; i < n; i += nl)
            if (group_size[i] > 1) { const uint32_t r = rank[i] + within[i]; sa[r] = i; rank[i] = r; }
        x.sync();
    }
    if (lane == 0) { sa[n] = 0; sa[N] = kNone; }
    if (nl == 1) for (uint32_t i = 0; i < n; ++i) rank[sa[i]] = i;
    x.sync();
    x.tick(1);
    // LCP (:142-158). One thread: Kasai's walk. Several lanes: each rank compares its suffix with its predecessor's directly - the same numbers.
    if (nl == 1) {
        for (uint32_t i = 0; i <= N; ++i) lcp[i] = 0;
        for (uint32_t i = 0, h = 0; i < n; ++i) {
            if (rank[i] > 0) {
                h += forward_lce(s, i + h, sa[rank[i] - 1] + h, n, n);
                lcp[rank[i]] = h;
                if (h > 0) --h;
            }
        }
    } else {
        for (uint32_t r = lane; r <= N; r += nl) lcp[r] = (r >= 1 && r < n) ? forward_lce(s, sa[r], sa[r - 1], n, n) : 0u;
    }
    for (uint32_t i = lane; i < N; i += nl) { lpf[i] = 0; prev_occ[i] = 0; }
    x.sync();
    x.tick(2);
    if (lane == 0) {
        // longest previous factor + where it occurred, tandem.cpp:69-110, over the N entries (the sentinel flushes the stack)
        // (every entry is also in the stack arrays; the top two are held in registers so that a pop does not wait for memory)
        uint32_t sp = 1;
        uint32_t top_a = 0, top_b = sa[0], under_a = 0, under_b = 0;
        st_a[0] = top_a; st_b[0] = top_b;
        uint32_t next_lcp = lcp[1], next_sa = sa[1];
        for (uint32_t i = 1; i <= N; ++i) {
            uint32_t u = next_lcp;
            const uint32_t sai = next_sa;
            if (i < N) { next_lcp = lcp[i + 1]; next_sa = sa[i + 1]; }
            while (sp && (sai == kNone || sai < top_b)) {
                const uint32_t vf = top_a, vs = top_b;
                const uint32_t lo = u < vf ? u : vf, hi = u < vf ? vf : u;   // std::tie(u, lpf[v]) = std::minmax(v.first, u)
                u = lo; lpf[vs] = hi;
                --sp;
                top_a = under_a; top_b = under_b;                             // entry sp - 1 (meaningless when sp == 0)
                if (sp >= 2) { under_a = st_a[sp - 2]; under_b = st_b[sp - 2]; }
                if (hi == 0) prev_occ[vs] = kNone;
                else if (vf > u) prev_occ[vs] = sp ? top_b : kNone;
                else prev_occ[vs] = sai;
            }
            if (i < N) { st_a[sp] = u; st_b[sp] = sai; under_a = top_a; under_b = top_b; top_a = u; top_b = sai; ++sp; }
        }
        x.tick(3);
        // Lempel-Ziv blocks with a previous occurrence each, :218-245
        uint32_t nb0 = 1, end = 1;
        bpos[0] = 0; blen[0] = 1; bprev[0] = kNone;
        while (end < n) {
            const uint32_t m = lpf[end] > 1 ? lpf[end] : 1;
            bpos[nb0] = end; blen[nb0] = m; bprev[nb0] = prev_occ[end]; ++nb0;
            end += m;
        }
        shared[0] = nb0; shared[1] = kOk;
    }
    x.sync();
    x.tick(4);
    const uint32_t nb = shared[0];
    // leftmost maximal repetitions that cross or touch a block border, :251-292. An item = (block h, right period j = 1..5 | left period j = 1..4),
    // in the reference's discovery order h-major, right periods before left ones; each item is independent, the list keeps discovery order.
    uint32_t nf = 0;
    {
        const uint32_t per_block = 2 * max_period, n_items = nb > 1 ? (nb - 1) * per_block : 0;
        bool overflow = false;
        for (uint32_t t0 = 0; t0 < n_items; t0 += nl) {
            const uint32_t t = t0 + lane;
            Repeat r {0, 0, 0}; bool valid = false;
            if (t < n_items) {
                const uint32_t h = 1 + t / per_block, c = t % per_block;
                const uint32_t u = bpos[h], bn = blen[h];
                const uint32_t m2 = 2 * blen[h - 1] + bn, m = u < m2 ? u : m2;
                const uint32_t tt = u - m, e = u + bn;
                if (c < max_period) {                                        // right-maximal periodicity of period j
                    const uint32_t j = min_period + c, jr = bn < max_period ? bn : max_period;
                    if (j <= jr) {
                        const uint32_t ls = backward_lce(s, u - 1, u + j - 1, tt);
                        const uint32_t lp = forward_lce(s, u + j, u, e, n);
                        if (ls + lp >= j && j + lp < bn) { r = Repeat {u - ls, j + lp + ls, j}; valid = true; }
                    }
                } else {                                                     // left-maximal periodicity of period j
                    const uint32_t j = min_period + (c - max_period), jl = m < max_period ? m : max_period;
                    if (j < jl) {
                        const uint32_t ls = backward_lce(s, u - j - 1, u - 1, tt);
                        const uint32_t lp = forward_lce(s, u, u - j, e, n);
                        if (ls + lp >= j) { r = Repeat {u - (ls + j), j + lp + ls, j}; valid = true; }
                    }
                }
            }
            const uint32_t at = x.claim(valid, nf);
            if (valid) { if (at < found_cap) found[at] = r; else overflow = true; }
        }
        if (x.any(overflow)) return kOverflow;
    }
    x.sync();
    x.tick(5);
    // get_end_buckets (:297-312: a run is its position and length, the first period found stays) and get_sorted_buckets (:318-334):
    // per start position, the distinct runs in order of their end, ties in order of discovery.
    const Repeat* initial = found;                                        // initial[first_kept[p] .. first_kept[p + 1]) = runs that start at p
    if (nl == 1) {
        // two stable counting sorts
        uint32_t nk = 0;
        {
            Repeat* by_end = kept;                                        // stable by end position
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
                if (seen) by_end[i].period = kNone;                       // tombstone: keeps its place for the scans above
            }
            for (uint32_t i = 0; i <= n; ++i) cnt[i] = 0;
            for (uint32_t i = 0; i < nf; ++i) if (by_end[i].period != kNone) ++cnt[by_end[i].pos];
            for (uint32_t i = 0, run = 0; i <= n; ++i) { const uint32_t c = cnt[i]; cnt[i] = run; first_kept[i] = run; run += c; }
            for (uint32_t i = 0; i < nf; ++i) if (by_end[i].period != kNone) { found[cnt[by_end[i].pos]++] = by_end[i]; ++nk; }   // `found` is free again: stable by start
        }
        first_kept[n] = nk;
    } else {
        // several lanes: a run is dropped if an earlier one has its start and end; the place of a kept run is the number of kept runs
        // with a smaller (start, end); a position's list starts after the kept runs that start before it. (start, length) in one word.
        uint32_t* fkey = cnt; uint32_t* fkept = first_run;                // nf <= cap_found <= n + 4 entries each; first_run is written later
        for (uint32_t i = lane; i < nf; i += nl) fkey[i] = found[i].pos << 16 | found[i].length;
        x.sync();
        for (uint32_t i = lane; i < nf; i += nl) {
            const uint32_t ki = fkey[i];
            bool dup = false;
            for (uint32_t j0 = 0; j0 < i; j0 += 4) {
                const uint32_t k0 = fkey[j0], k1 = fkey[j0 + 1 < i ? j0 + 1 : j0], k2 = fkey[j0 + 2 < i ? j0 + 2 : j0], k3 = fkey[j0 + 3 < i ? j0 + 3 : j0];
                dup = dup || k0 == ki || k1 == ki || k2 == ki || k3 == ki;
            }
            fkept[i] = dup ? kNone : ki;
        }
        x.sync();
        for (uint32_t i = lane; i < nf; i += nl) {
            const uint32_t ki = fkept[i];
            if (ki == kNone) continue;
            uint32_t place = 0;
            for (uint32_t j0 = 0; j0 < nf; j0 += 4) {
                const uint32_t k0 = fkept[j0], k1 = fkept[j0 + 1 < nf ? j0 + 1 : j0], k2 = fkept[j0 + 2 < nf ? j0 + 2 : j0], k3 = fkept[j0 + 3 < nf ? j0 + 3 : j0];
                place += (k0 < ki ? 1u : 0u) + (j0 + 1 < nf && k1 < ki ? 1u : 0u) + (j0 + 2 < nf && k2 < ki ? 1u : 0u) + (j0 + 3 < nf && k3 < ki ? 1u : 0u);
            }
            kept[place] = found[i];
        }
        uint32_t nk = 0;
        for (uint32_t i0 = 0; i0 < nf; i0 += nl) x.claim(i0 + lane < nf && fkept[i0 + lane] != kNone, nk);
        x.sync();
        for (uint32_t p = lane; p <= n; p += nl) {                        // std::lower_bound on the start positions, which are sorted now
            uint32_t first = 0, len = nk;
            while (len > 0) {
                const uint32_t half = len >> 1, mid = first + half;
                if (kept[mid].pos < p) { first = mid + 1; len = len - half - 1; } else len = half;
            }
            first_kept[p] = first;
        }
        initial = kept;
        x.sync();
    }
    x.tick(6);
    if (lane == 0) {
        // extract_maximal_repetitions :337-375: inside a block, the runs that lie within the block's earlier occurrence are copied (shifted)
        // in front of the position's own runs. Positions are visited left to right and a copy source lies to the left, so every final
        // list is written once, in output order.
        uint32_t nr = 0; int rc = kOk;
        first_run[0] = 0;
        uint32_t own_begin = first_kept[0];
        for (uint32_t k = 0; k < nb && rc == kOk; ++k) {
            const uint32_t block_end = bpos[k] + blen[k];
            const uint32_t delta = bpos[k] - (bprev[k] != kNone ? bprev[k] : 0);
            const uint32_t max_target_end = block_end - delta;
            for (uint32_t j = bpos[k]; j < block_end; ++j) {              // first_run[j] == nr here
                const uint32_t target_start = j - delta;                  // delta > 0 for every block but the first, whose target is itself (nothing to copy yet)
                const uint32_t own_end = first_kept[j + 1];
                const uint32_t target_first = first_run[target_start], target_last = first_run[target_start + 1];
                const uint32_t own = own_end - own_begin;
                if (target_start < j && target_last > target_first) {
                    uint32_t target_end = max_target_end;
                    if (own) { const uint32_t c = target_start + initial[own_begin].length; target_end = c < max_target_end ? c : max_target_end; }
                    const Repeat* target = runs + target_first;
                    uint32_t first = 0, len = target_last - target_first;                                 // std::lower_bound on pos + length < target_end
                    while (len > 0) {
                        const uint32_t half = len >> 1, mid = first + half;
                        if (target[mid].pos + target[mid].length < target_end) { first = mid + 1; len = len - half - 1; } else len = half;
                    }
                    if (nr + first > runs_cap) { rc = kOverflow; break; }
                    for (uint32_t q = 0; q < first; ++q) runs[nr++] = Repeat {target[q].pos + delta, target[q].length, target[q].period};
                }
                if (nr + own > runs_cap) { rc = kOverflow; break; }
                for (uint32_t q = 0; q < own; ++q) runs[nr++] = initial[own_begin + q];
                first_run[j + 1] = nr;
                own_begin = own_end;
            }
        }
        shared[0] = nr; shared[1] = (uint32_t)rc;
    }
    x.sync();
    x.tick(7);
    *n_runs = shared[0];
    return (int)shared[1];
}

// ---- the two models ------------------------------------------------------------------------------------------------------------
OCT_EM int8_t table_at(const int8_t* t, uint32_t periodicity) { return t[periodicity < OCT_PHMM_INDEL_TABLE ? periodicity : OCT_PHMM_INDEL_TABLE - 1]; }   // get_min_penalty :44-47
OCT_EM int8_t cap_at(const int8_t* t, uint32_t run) { return t[run < OCT_PHMM_SNV_TABLE ? run : OCT_PHMM_SNV_TABLE - 1]; }                            // get_penalty :115-119

// RepeatBasedIndelErrorModel::do_set_penalties, vector overload :67-83
template <class X>
OCT_EM int indel_penalties(const X& x, const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, uint32_t* w, uint32_t grow, int8_t* gap_open, int8_t* gap_extend)
{
    const uint32_t lane = x.lane(), nl = x.nl();
    for (uint32_t i = lane; i < n; i += nl) { gap_open[i] = m.dinucleotide_open[0]; gap_extend[i] = m.dinucleotide_extend[0]; }   // complex_open / complex_extend :32-33
    if (n == 0) return kOk;
    const Layout lay = layout(n, grow);
    Repeat* runs = (Repeat*)(w + lay.runs);
    uint32_t nr = 0;
    const int rc = lz_tandem_repeats(x, s, n, 1, 5, w, grow, runs, &nr);  // extract_repeats :15-18
    if (rc != kOk) return rc;
    if (lane == 0) sort_by_length(runs, nr);                              // :20-23
    x.sync();
    x.tick(8);
    auto penalties_of = [&](const Repeat& q, int8_t& open, int8_t& extend) {   // get_open_penalty :57-86, get_extension_penalty :94-103
        const uint32_t periodicity = q.length / q.period;
        const uint8_t* motif = s + q.pos;
        switch (q.period) {
            case 1: open = table_at(motif[0] == 'A' || motif[0] == 'T' ? m.at_homopolymer_open : m.cg_homopolymer_open, periodicity);
                    extend = table_at(m.homopolymer_extend, periodicity); break;
            case 2: open = table_at(m.dinucleotide_open, periodicity);
                    if (open > 7 && ((motif[0] == 'C' && motif[1] == 'G') || (motif[0] == 'G' && motif[1] == 'C'))) open = (int8_t)(open - 2);
                    extend = table_at(m.dinucleotide_extend, periodicity); break;
            default: open = table_at(m.trinucleotide_open, periodicity);
                    extend = table_at(m.trinucleotide_extend, periodicity); break;
        }
    };
    if (nl == 1) {
        for (uint32_t i = 0; i < nr; ++i) {
            const Repeat& q = runs[i];
            int8_t open, extend;
            penalties_of(q, open, extend);
            for (uint32_t k = 0; k < q.length; ++k) {
                if (open < gap_open[q.pos + k]) gap_open[q.pos + k] = open;   // fill_n_if_less
                gap_extend[q.pos + k] = extend;                               // fill_n: the last (longest) covering run decides
            }
        }
    } else {
        // the penalties of every run once (kept in the run's own entry: start | length << 16, open | extend << 8), then per position:
        // the minimum open penalty over the covering runs, the extension penalty of the LAST covering run in sorted order
        for (uint32_t i = lane; i < nr; i += nl) {
            const Repeat q = runs[i];
            int8_t open, extend;
            penalties_of(q, open, extend);
            runs[i].pos = q.pos | q.length << 16;
            runs[i].length = (uint32_t)(uint8_t)open | (uint32_t)(uint8_t)extend << 8;
        }
        x.sync();
        for (uint32_t pos = lane; pos < n; pos += nl) {
            int8_t best_open = m.dinucleotide_open[0], last_extend = m.dinucleotide_extend[0];
            for (uint32_t i0 = 0; i0 < nr; i0 += 4) {
                uint32_t where[4], pen[4];
                for (uint32_t u = 0; u < 4; ++u) { const uint32_t i = i0 + u < nr ? i0 + u : i0; where[u] = runs[i].pos; pen[u] = runs[i].length; }
                for (uint32_t u = 0; u < 4; ++u) {
                    if (i0 + u < nr && pos - (where[u] & 0xffffu) < where[u] >> 16) {     // start <= pos < start + length
                        const int8_t open = (int8_t)(pen[u] & 0xff);
                        if (open < best_open) best_open = open;
                        last_extend = (int8_t)(pen[u] >> 8 & 0xff);
                    }
                }
            }
            gap_open[pos] = best_open; gap_extend[pos] = last_extend;
        }
    }
    x.sync();
    x.tick(9);
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
    for (uint32_t k0 = 1; k0 < n; k0 += 4) {                              // four loads ahead of the state machine (the loads do not depend on it)
        int8_t m4[4]; uint32_t v4[4];
        for (uint32_t u = 0; u < 4; ++u) m4[u] = mask[start + (int)(k0 + u < n ? k0 + u : k0) * step];
        for (uint32_t u = 0; u < 4; ++u) {
            const int8_t x = m4[u];
            uint32_t v = 0;
            if (k0 + u < n) {
                if (x == 0) {
                    ++gap;
                    if (count > 0) {
                        if (gap == 1) { if (max_gap >= 1) v = count; else { v = count; count = 0; } }
                        else if (gap > max_gap) count = 0;
                    }
                } else if (prev == x) { gap = 0; ++count; }
                else { prev = x; v = count; count = 1; }
            }
            v4[u] = v;
        }
        for (uint32_t u = 0; u < 4; ++u) if (k0 + u < n) runs[start + (int)(k0 + u) * step] = v4[u];
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
// that period's runs, in the order the scan finds them, so the library's merge by position is not needed). Several lanes: the three
// scans run on three lanes, the six count_runs passes (three periods x two strands) on six, everything per position on all.
template <class X>
OCT_EM void snv_priors(const X& x, const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* substitution_mask, uint32_t* w,
                       uint8_t* mask_fwd, int8_t* prior_fwd, uint8_t* mask_rev, int8_t* prior_rev)
{
    const uint32_t lane = x.lane(), nl = x.nl();
    if (!m.use_snv_model) {                                               // model.cpp:68-73
        for (uint32_t i = lane; i < n; i += nl) { mask_fwd[i] = s[i]; mask_rev[i] = s[i]; prior_fwd[i] = 100; prior_rev[i] = 100; }
        x.sync();
        return;
    }
    const size_t W = stride(n);
    int8_t* masks[3] = {(int8_t*)w, (int8_t*)(w + W), (int8_t*)(w + 2 * W)};
    uint32_t* runs6 = w + 3 * W;                                          // [6][W] run counts: period p forward at 2 p, backward at 2 p + 1
    for (int p = 0; p < 3; ++p) for (uint32_t i = lane; i < n; i += nl) masks[p][i] = 0;
    x.sync();
    auto mark = [&](uint32_t pos, uint32_t length, uint32_t period) {
        int8_t h = 0;
        for (uint32_t k = 0; k < period; ++k) h = (int8_t)(h + base_hash(s[pos + k]));   // repeat_hash :107-113
        for (uint32_t k = 0; k < length; ++k) masks[period - 1][pos + k] = h;
    };
    if (nl == 1) {
        for (uint32_t cur = 0; cur < n;) {                                // homopolymers :392-409
            uint32_t it = cur;
            while (it + 1 < n && s[it] != s[it + 1]) ++it;
            if (it + 1 >= n) break;
            uint32_t it2 = it + 1;
            while (it2 < n && s[it2] == s[it]) ++it2;
            mark(it, it2 - it, 1);
            cur = it2;
        }
        for (uint32_t P = 2; P <= 3; ++P) {                               // extract_exact_tandem_repeats<P> :411-436
            if (n < 2 * P) continue;
            uint32_t it1 = next_unequal_pair(s, 0, n);
            if (it1 == n) continue;
            for (uint32_t it2 = it1 + P; it2 < n;) {
                uint32_t a = it2, b = it1;                                // std::mismatch(it2, last, it1)
                while (a < n && s[a] == s[b]) { ++a; ++b; }
                if (b >= it2) { mark(it1, a - it1, P); it1 = b; } else ++it1;
                it1 = next_unequal_pair(s, it1, n);
                if (it1 == n) break;
                it2 = it1 + P;
            }
        }
    } else {
        // The same scans read per position. Homopolymers: the scan marks every maximal run of two and more equal bases. Periods 2 and 3:
        // the scan visits the positions p with s[p] != s[p + 1] in order; where the text repeats itself P further on for L >= P bases it
        // marks [p, p + P + L) and continues at the first such position >= p + L. So: all positions with L >= P in order (candidates),
        // a greedy pass over that short list, and per position the hash of the last marked run that covers it.
        for (uint32_t i = lane; i < n; i += nl)
            masks[0][i] = ((i > 0 && s[i] == s[i - 1]) || (i + 1 < n && s[i] == s[i + 1])) ? base_hash(s[i]) : (int8_t)0;
        uint32_t* cand_pos[2] = {w + 3 * W, w + 5 * W}; uint32_t* cand_len[2] = {w + 4 * W, w + 6 * W};
        uint32_t* real_pos[2] = {w + 7 * W, w + 8 * W}; uint32_t* real_lh[2] = {w + 7 * W + W / 2, w + 8 * W + W / 2};
        uint32_t* n_real = w + 9 * W;
        uint32_t nc[2] = {0, 0};
        for (uint32_t p0 = 0; p0 < n; p0 += nl) {
            const uint32_t q = p0 + lane;
            uint32_t L[2] = {0, 0};
            if (q + 1 < n && s[q] != s[q + 1])
                for (uint32_t P = 2; P <= 3; ++P) if (q + P < n) L[P - 2] = forward_lce(s, q + P, q, n, n);
            for (uint32_t P = 2; P <= 3; ++P) {
                const bool is = L[P - 2] >= P;
                const uint32_t at = x.claim(is, nc[P - 2]);
                if (is) { cand_pos[P - 2][at] = q; cand_len[P - 2][at] = L[P - 2]; }
            }
        }
        x.sync();
        if (lane < 2) {
            const uint32_t P = lane + 2, count = nc[lane];
            const uint32_t* cp = cand_pos[lane]; const uint32_t* cl = cand_len[lane];
            uint32_t limit = 0, nr = 0;
            for (uint32_t c = 0; c < count; ++c) {
                const uint32_t q = cp[c], L = cl[c];
                if (q >= limit) {
                    int8_t h = 0;
                    for (uint32_t k = 0; k < P; ++k) h = (int8_t)(h + base_hash(s[q + k]));
                    real_pos[lane][nr] = q; real_lh[lane][nr] = (P + L) << 8 | (uint32_t)(uint8_t)h; ++nr;
                    limit = q + L;
                }
            }
            n_real[lane] = nr;
        }
        x.sync();
        for (uint32_t i = lane; i < n; i += nl) {
            for (uint32_t P = 2; P <= 3; ++P) {
                int8_t h = 0;
                const uint32_t count = n_real[P - 2];
                for (uint32_t r = 0; r < count; ++r) {
                    const uint32_t lh = real_lh[P - 2][r];
                    if (i - real_pos[P - 2][r] < lh >> 8) h = (int8_t)(lh & 0xff);
                }
                masks[P - 1][i] = h;
            }
        }
    }
    x.sync();
    x.tick(10);
    const int8_t max_quality = m.snv_caps[0][0];
    for (uint32_t pass = (nl == 1 ? 0u : lane); pass < 6; pass += nl) {   // count_runs :48-87, one pass per (period, strand)
        const uint32_t p = pass >> 1;
        if (n) count_runs(masks[p], n, (pass & 1) ? (int)n - 1 : 0, (pass & 1) ? -1 : 1, runs6 + pass * W, p + 2);
    }
    x.sync();
    x.tick(11);
    for (uint32_t i = lane; i < n; i += nl) {
        int8_t pf = max_quality, pr = max_quality;
        for (uint32_t p = 0; p < 3; ++p) {                                // set_priors :121-130
            const int8_t cf = cap_at(m.snv_caps[p], runs6[(2 * p) * W + i]), cr = cap_at(m.snv_caps[p], runs6[(2 * p + 1) * W + i]);
            if (cf < pf) pf = cf;
            if (cr < pr) pr = cr;
        }
        if (substitution_mask && substitution_mask[i]) { pf = max_quality; pr = max_quality; }   // :168-172
        prior_fwd[i] = pf; prior_rev[i] = pr;
        mask_fwd[i] = s[(i + n - 1) % n];                                 // rotate_copy :173-177
        mask_rev[i] = s[(i + 1) % n];
    }
    x.sync();
    x.tick(12);
}

// All six vectors of one haplotype. `w`: workspace_words(n, grow) words, shared by the policy's lanes.
template <class X>
OCT_EM int penalty_vectors(const X& x, const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* substitution_mask, uint32_t* w, uint32_t grow,
                           int8_t* gap_open, int8_t* gap_extend, uint8_t* mask_fwd, int8_t* prior_fwd, uint8_t* mask_rev, int8_t* prior_rev)
{
    const int rc = indel_penalties(x, m, s, n, w, grow, gap_open, gap_extend);
    if (rc != kOk) return rc;
    snv_priors(x, m, s, n, substitution_mask, w, mask_fwd, prior_fwd, mask_rev, prior_rev);
    return kOk;
}
OCT_EM int penalty_vectors(const oct_phmm_error_model& m, const uint8_t* s, uint32_t n, const uint8_t* substitution_mask, uint32_t* w, uint32_t grow,
                           int8_t* gap_open, int8_t* gap_extend, uint8_t* mask_fwd, int8_t* prior_fwd, uint8_t* mask_rev, int8_t* prior_rev)
{
    return penalty_vectors(Seq {}, m, s, n, substitution_mask, w, grow, gap_open, gap_extend, mask_fwd, prior_fwd, mask_rev, prior_rev);
}

}} // namespace octphmm::em
